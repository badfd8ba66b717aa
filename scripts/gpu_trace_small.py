"""Per-phase host trace (ROFL_TRACE=2) of one warm client at a small shape: cfg 1 (d = 5000, 8-bit, fp16/frac7) by default."""
import os, sys, time
os.environ["ROFL_TRACE"] = "2"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import api
d = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
fpb = int(sys.argv[3]) if len(sys.argv) > 3 else 16
R.set_device(0); api.set_fp(fpb, 7)
rng = np.random.default_rng(3)
mx = np.float32(((1 << (nb - 1)) - 1) / 128.0)
vals = rng.uniform(-mx, mx, d).astype(np.float32) * np.float32(0.99)
bl = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); bl[:, 31] &= 0x0F
for i in range(3):
    sys.stderr.write("=== create %d\n" % i)
    t = time.perf_counter()
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, nb, 4, nonce=R.Nonce.seeded(b"\x01" * 32))
    t1 = time.perf_counter()
    ok = R.range_proof_vec.verify_rangeproof(pr, cm, nb, verifier_seed=b"\x02" * 32)
    sys.stderr.write("=== create %.2f ms verify %.2f ms ok=%s\n" % ((t1 - t) * 1e3, (time.perf_counter() - t1) * 1e3, ok))
