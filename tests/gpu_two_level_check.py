"""Helper for test_gpu_parity.py::test_two_level_sort_bin_shapes (subprocess: the knobs are read once per process).
One proof over a single chunk of 2^19 terms (d = 16384, 32 bits, P = 1) so that the fixed-base launches take the two-level bucket sort;
prints a digest of proof + commitments and the verdicts.  With argv[1] == "oracle" the digest is the oracle's."""
import sys, os, hashlib, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
d, nb, P, fb, ff = 16384, 32, 1, 32, 7
rng = np.random.default_rng(2024)
mx = np.float32(((1 << (nb - 1)) - 1) / float(1 << ff))
vals = np.clip(rng.uniform(-mx, mx, size=d).astype(np.float32), -mx, np.nextafter(mx, np.float32(0)))
bl = orc.rand_scalars(rng, d)
seed = b"\x21" * 32
if len(sys.argv) > 1 and sys.argv[1] == "oracle":
    rc, pr, cm = orc.create_rangeproof(vals, bl, nb, P, fb, ff, seed=seed)
    assert rc == 0
    print("DIGEST", hashlib.sha256(pr.tobytes() + cm.tobytes()).hexdigest(), "1 0")
    sys.exit(0)
import rofl_project_code_amd as R
R.set_device(0); R.api.set_fp(fb, ff)
pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.seeded(seed))
ok = R.range_proof_vec.verify_rangeproof(pr, cm, nb, verifier_seed=b"\x05" * 32)
t = pr.copy(); t[0, 100] ^= 4
nok = R.range_proof_vec.verify_rangeproof(t, cm, nb, verifier_seed=b"\x05" * 32)
print("DIGEST", hashlib.sha256(pr.tobytes() + cm.tobytes()).hexdigest(), int(ok), int(nok))
