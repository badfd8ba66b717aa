"""Helper for test_gpu_parity.py::test_two_level_sort_bin_shapes (subprocess: the knobs are read once per process).
One proof over a single chunk of 2^19 terms (16 384 values, 32 bits, P = 1) so that the fixed-base launches take the two-level bucket sort;
prints a digest of proof + commitments and the verdicts.  argv[1]: an .npz with the chunk's inputs (vals, bl, seed) -- chunk 0 of the session's
full-size cfg-4 oracle case, whose oracle bytes the test holds."""
import sys, os, hashlib, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
case = np.load(sys.argv[1])
nb, P, fb, ff = 32, 1, 32, 7
vals, bl, seed = case["vals"], case["bl"], bytes(case["seed"])
import rofl_project_code_amd as R
R.set_device(0); R.api.set_fp(fb, ff)
pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.seeded(seed))
ok = R.range_proof_vec.verify_rangeproof(pr, cm, nb, verifier_seed=b"\x05" * 32)
t = pr.copy(); t[0, 100] ^= 4
nok = R.range_proof_vec.verify_rangeproof(t, cm, nb, verifier_seed=b"\x05" * 32)
print("DIGEST", hashlib.sha256(pr.tobytes() + cm.tobytes()).hexdigest(), int(ok), int(nok))
