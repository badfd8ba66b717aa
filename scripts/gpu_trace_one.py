"""One warm create + verify of the headline shape with the library's per-phase host timeline on stderr (ROFL_TRACE=2)."""
import os, sys
os.environ["ROFL_TRACE"] = "2"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import rofl_project_code_amd as R
from rofl_project_code_amd import api
R.set_device(0); api.set_fp(32, 7)
d, nb, P = 25000, 32, 4
rng = np.random.default_rng(3)
mx = np.float32(((1 << (nb - 1)) - 1) / 128.0)
vals = np.clip(rng.uniform(-mx, mx, d).astype(np.float32), -mx, np.nextafter(mx, np.float32(0)))
bl = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); bl[:, 31] &= 0x0F
api.bp_gens_prepare(nb, R.range_proof_vec.next_pow2(d) // P)
for i in range(6):
    if i == 5: print("==== traced step", file=sys.stderr, flush=True)
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.seeded(b"\x07" * 32))
    assert R.range_proof_vec.verify_rangeproof(pr, cm, nb, verifier_seed=b"\x08" * 32)
