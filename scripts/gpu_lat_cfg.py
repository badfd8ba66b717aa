"""Latency of the non-headline shapes: cfg 1 (d = 5 000, 8-bit, fp16) and the cfg 5 composite (EncParamsL2 at d = 55 000)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import api, params
R.set_device(0)
def med(x): x = sorted(x); return x[len(x) // 2]
# cfg 1
api.set_fp(16, 7)
rng = np.random.default_rng(1)
mx = np.float32(127 / 128.0)
vals = np.clip(rng.uniform(-mx, mx, 5000).astype(np.float32), -mx, np.nextafter(mx, np.float32(0)))
bl = rng.integers(0, 256, size=(5000, 32), dtype=np.uint8); bl[:, 31] &= 0x0F
tc, tv = [], []
for i in range(12):
    t = time.perf_counter(); pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, 8, 4, nonce=R.Nonce.seeded(bytes([i + 1]) * 32)); t1 = time.perf_counter()
    assert R.range_proof_vec.verify_rangeproof(pr, cm, 8, verifier_seed=b"\x02" * 32); t2 = time.perf_counter()
    if i >= 3: tc.append((t1 - t) * 1e3); tv.append((t2 - t1) * 1e3)
print("cfg1 create %.2f verify %.2f" % (med(tc), med(tv)), flush=True)
# cfg 5 / cfg 3
FP = (32, 7)
for d in (25000, 55000):
    rng = np.random.default_rng(5)
    vals = (rng.integers(-3, 4, size=d) / 128.0).astype(np.float32)
    r1 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r1[:, 31] &= 0x0F
    r2 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r2[:, 31] &= 0x0F
    tc, tv = [], []
    for i in range(10):
        t = time.perf_counter(); upd = params.EncParamsL2.encrypt(vals, r1, 8, 4, 32, nonce_seed=b"\x01" * 32, rand_scalars=r2, fp=FP); t1 = time.perf_counter()
        assert upd.verify(verifier_seed=b"\x04" * 32, fp=FP); t2 = time.perf_counter()
        if i >= 3: tc.append((t1 - t) * 1e3); tv.append((t2 - t1) * 1e3)
    print("L2 composite d=%d create %.2f verify %.2f" % (d, med(tc), med(tv)), flush=True)
