"""Randomised parity fuzz of the many-chunk shapes (n_partition 16 .. 128, chunks of 2^11 .. 2^15 generators: the 15-bit window tables,
the device Horner, folds down to 64 generators): three random chunks per case against the oracle's single-chunk prover (orc.prove_chunk),
round trip, tamper.  Time-boxed; run on the GPU box."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import orc
import rofl_project_code_amd as R
from rofl_project_code_amd import api
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
R.set_device(0)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
t0 = time.time(); n_ok = 0
while time.time() - t0 < budget:
    nb = int(rng.choice([8, 16, 32]))
    ff = 7; fb = 32 if nb == 32 else int(rng.choice([16, 32]))
    P = int(rng.choice([16, 32, 64, 128]))
    lgm = int(rng.integers(max(1, 11 - int(np.log2(nb))), 16 - int(np.log2(nb))))      # n*m between 2^11 and 2^15
    m = 1 << lgm
    dp = m * P
    d = int(rng.integers(dp // 2 + 1, dp + 1))
    api.set_fp(fb, ff)
    mn, mx = R.conversion32.get_clip_bounds(nb)
    vals = np.clip(rng.uniform(mn, mx, size=d).astype(np.float32), mn, np.nextafter(np.float32(mx), np.float32(0)))
    bl = orc.rand_scalars(rng, d)
    seed = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.seeded(seed))
    assert pr.shape[0] == P, (pr.shape, P)
    vp = np.zeros(dp, np.float32); vp[:d] = vals
    bp = np.zeros((dp, 32), np.uint8); bp[:d] = bl
    for c in sorted(set([0, P - 1, int(rng.integers(0, P))])):
        lo, hi = c * m, min((c + 1) * m, d)
        rc, oproof, oV = orc.prove_chunk(vp[c * m:(c + 1) * m], bp[c * m:(c + 1) * m], nb, c, ff, seed, n_real=max(hi - lo, 0))
        assert rc == 0 and (oproof == pr[c]).all(), ("chunk differs", nb, fb, P, m, d, c)
    assert R.range_proof_vec.verify_rangeproof(pr, cm, nb, verifier_seed=seed)
    bad = pr.copy(); bad[rng.integers(0, P), rng.integers(0, pr.shape[1])] ^= 1 << int(rng.integers(0, 8))
    try:
        assert R.range_proof_vec.verify_rangeproof(bad, cm, nb, verifier_seed=seed) is False
    except R.RoflError as e:
        assert e.code == 5
    n_ok += 1
    print("ok nb=%d fp=%d P=%d m=%d d=%d" % (nb, fb, P, m, d), flush=True)
print("fuzz ok: %d many-chunk cases in %.0f s" % (n_ok, time.time() - t0))
