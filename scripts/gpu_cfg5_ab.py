"""cfg 3 / cfg 5 (L2 composite: three proofs of a client on separate lanes) latency, sequential clients: gpu_cfg5_ab.py [d] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import api, params
R.set_device(0); api.set_fp(32, 7)
d = int(sys.argv[1]) if len(sys.argv) > 1 else 55000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = np.random.default_rng(5)
mx = np.float32(((1 << 7) - 1) / 128.0)
vals = rng.uniform(-mx, mx, size=d).astype(np.float32) * np.float32(0.05)
r1 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r1[:, 31] &= 0x0F
r2 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r2[:, 31] &= 0x0F
tc, tv = [], []
for i in range(reps + 3):
    t0 = time.perf_counter()
    upd = params.EncParamsL2.encrypt(vals, r1, 8, 4, 32, nonce_seed=bytes([i + 1]) * 32, rand_scalars=r2)
    t1 = time.perf_counter()
    ok = upd.verify(verifier_seed=b"\x05" * 32)
    t2 = time.perf_counter()
    assert ok
    if i >= 3: tc.append((t1 - t0) * 1e3); tv.append((t2 - t1) * 1e3)
tc.sort(); tv.sort()
print("%s d=%d create med %.2f min %.2f max %.2f | verify med %.2f min %.2f" % ("prev" if os.environ.get("ROFL_ZK_LIB") else "new ", d, tc[len(tc) // 2], tc[0], tc[-1], tv[len(tv) // 2], tv[0]), flush=True)
