"""cfg 5 with gpu_configs.py's protocol (six clients: six encrypts, then six verifies; 3 warm-ups + 6 samples): ms per client"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import api, params
R.set_device(0); api.set_fp(32, 7)
d = int(sys.argv[1]) if len(sys.argv) > 1 else 55000
ins = []
for c in range(6):
    rng = np.random.default_rng(77 + c)
    vals = (rng.integers(-3, 4, size=d) / 128.0).astype(np.float32)
    r1 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r1[:, 31] &= 0x0F
    r2 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r2[:, 31] &= 0x0F
    ins.append((vals, r1, r2))
tc, tv = [], []
for s in range(9):
    t0 = time.perf_counter(); outs = []; each = []
    for vals, r1, r2 in ins:
        t = time.perf_counter(); outs.append(params.EncParamsL2.encrypt(vals, r1, 8, 4, 32, nonce_seed=b"\x01" * 32, rand_scalars=r2)); each.append((time.perf_counter() - t) * 1e3)
    t1 = time.perf_counter()
    for upd in outs:
        assert upd.verify(verifier_seed=b"\x04" * 32)
    t2 = time.perf_counter()
    if s >= 3: tc.append((t1 - t0) * 1e3 / 6); tv.append((t2 - t1) * 1e3 / 6)
    if s == 8: print("last sample, per client create ms:", [round(x, 1) for x in each])
print("ROFL_HOST_THREADS=%s d=%d create med %.2f (min %.2f max %.2f) verify med %.2f" % (os.environ.get("ROFL_HOST_THREADS", "-"), d, float(np.median(tc)), min(tc), max(tc), float(np.median(tv))), flush=True)
