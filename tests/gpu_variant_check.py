"""Helper for test_gpu_parity.py::test_msm_variants_small_sizes (run in a subprocess: the knobs are read once per process).
Proofs, commitments and verdicts vs the oracle for small shapes, so that the code paths that normally only run at
full size (fixed-base window tables, LDS-ranked scatter, two-pass sort, batched verification) are compared bit for bit."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
import rofl_project_code_amd as R
R.set_device(0)
rng = np.random.default_rng(11)
bad = 0
for d, nb, P, fb, ff in ((64, 8, 1, 16, 7), (256, 16, 2, 16, 7), (500, 32, 4, 32, 7), (700, 8, 4, 16, 7), (512, 32, 1, 32, 7), (37, 64, 2, 64, 7),
                         (2100, 8, 64, 16, 7)):       # many small chunks (the e2e n_partition): device-side Horner, folds down to 64 generators
    R.api.set_fp(fb, ff)
    mn, mx = R.conversion32.get_clip_bounds(nb)
    vals = np.clip(rng.uniform(mn, mx, size=d).astype(np.float32), mn, np.nextafter(np.float32(mx), np.float32(0)))
    if d == 700: vals[:] = 0          # structured scalars
    bl = orc.rand_scalars(rng, d)
    seed = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.seeded(seed))
    rc, opr, ocm = orc.create_rangeproof(vals, bl, nb, P, fb, ff, seed=seed)
    ok = R.range_proof_vec.verify_rangeproof(pr, cm, nb, verifier_seed=b"\x05" * 32)
    t = pr.copy(); t[0, 40] ^= 1
    nok = R.range_proof_vec.verify_rangeproof(t, cm, nb, verifier_seed=b"\x05" * 32)
    good = rc == 0 and (pr == opr).all() and (cm == ocm).all() and ok and not nok
    print(d, nb, P, "OK" if good else "MISMATCH", flush=True)
    bad += not good
print("FB_SMALL", "PASS" if bad == 0 else "FAIL")
sys.exit(bad)
