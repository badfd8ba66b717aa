"""L2 composite at d = 55 000 (cfg 5, one client): EncParamsL2.encrypt / verify steady-state times, for A/B runs of the round-2 knobs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import params
R.set_device(0)
fp = (32, 7); d = int(sys.argv[1]) if len(sys.argv) > 1 else 55000
rng = np.random.default_rng(77)
vals = (rng.integers(-3, 4, size=d) / 128.0).astype(np.float32)
r1 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r1[:, 31] &= 0x0F
r2 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r2[:, 31] &= 0x0F
te, tv = [], []
for s in range(12):
    t0 = time.perf_counter(); u = params.EncParamsL2.encrypt(vals, r1, 8, 4, 32, nonce_seed=b"\x01" * 32, rand_scalars=r2, fp=fp); t1 = time.perf_counter()
    ok = u.verify(verifier_seed=b"\x04" * 32, fp=fp); t2 = time.perf_counter()
    assert ok
    if s >= 6: te.append((t1 - t0) * 1e3); tv.append((t2 - t1) * 1e3)
print("d=%d %s encrypt %.2f verify %.2f" % (d, " ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith("ROFL_")), float(np.median(te)), float(np.median(tv))))
