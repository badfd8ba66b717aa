"""Worker of tests/test_gpu_chunk_split.py (a process of its own: logical devices are mapped before the library first touches them, and the
per-device fold tables are kept small with ROFL_FOLD_TAB_MB so that four device contexts share one MI355X comfortably).

  gpu_split_worker.py <case.npz> <n_devices> <n_partition>

ONE client's chunks over n_devices logical devices through rofl_set_option("devices", mask) -- SURVEY 8(e) "cfg 2/3 at > 1 GPU -> chunks over
ranks", range_proof_vec/mod.rs:54-78, 168-181.  The .npz holds the client's inputs and the oracle's answer: either whole (opr, ocm) or, for
n_partition = 64, the oracle's proofs of sampled chunks (ochunk_idx, ochunk_proofs).  Checks: split bytes == oracle bytes == unsplit bytes; the
split verifier accepts them, rejects a tampered proof whichever run it falls into, and reports a bad value in the LAST run as the call's
ValueOutOfRangeError.  Prints 'split ok: ...' and exits 0."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def l2_main(nd):
    """gpu_split_worker.py l2 <n_devices>: one L2 update (cfg 3: d = 25 000, 8-bit range leg at P = 4, square proofs, sum proof) on one device and
    with its legs dealt to 2 .. n_devices logical devices: same bytes; verdicts; tamper."""
    import rofl_project_code_amd as R
    from rofl_project_code_amd import api, params
    for k in range(1, nd):
        api.map_device(k, 0)
    R.set_device(0)
    fp = (32, 7); d = 25000
    rng = np.random.default_rng(31)
    vals = (rng.integers(-3, 4, size=d) / 128.0).astype(np.float32)
    r1 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r1[:, 31] &= 0x0F
    r2 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r2[:, 31] &= 0x0F
    one = params.EncParamsL2.encrypt(vals, r1, 8, 4, 32, nonce_seed=b"\x61" * 32, rand_scalars=r2, fp=fp)
    assert one.verify(verifier_seed=b"\x02" * 32, fp=fp) is True
    for k in sorted({2, nd}):
        R.set_option("devices", (1 << k) - 1)
        upd = params.EncParamsL2.encrypt(vals, r1, 8, 4, 32, nonce_seed=b"\x61" * 32, rand_scalars=r2, fp=fp)
        for name in ("enc_values", "square_proofs", "range_proofs", "square_range_proof"):
            assert (getattr(upd, name) == getattr(one, name)).all(), "%s differs on %d devices" % (name, k)
        assert upd.verify(verifier_seed=b"\x03" * 32, fp=fp) is True
        used = [j for j in range(k) if (R.set_device(j), api.bp_gens_table_bytes(8, 32768 // 4))[1] > 0]      # the 8-bit leg's four chunks went to min(k, 4) devices
        R.set_device(0)
        assert len(used) == min(k, 4), used
        bad = params.EncParamsL2(upd.enc_values, upd.square_proofs.copy(), upd.range_proofs, upd.square_range_proof, upd.prove_range, upd.l2_prove_range)
        bad.square_proofs[d - 3, 130] ^= 1      # a response scalar of an element of the LAST run
        assert bad.verify(verifier_seed=b"\x03" * 32, fp=fp) is False
        pr, cm = R.square_rand_proof_vec.create_l2rangeproof_vec(vals, r1, r2, nonce=R.Nonce.seeded(b"\x62" * 32), fp=fp)      # the vector call itself, split
        R.set_option("devices", 0)
        pr1, cm1 = R.square_rand_proof_vec.create_l2rangeproof_vec(vals, r1, r2, nonce=R.Nonce.seeded(b"\x62" * 32), fp=fp)
        assert (pr == pr1).all() and (cm == cm1).all()
        assert R.square_rand_proof_vec.verify_l2rangeproof_vec(pr, cm) is True
    print("split ok: L2 update of d=%d on 1, 2 and %d devices: same bytes" % (d, nd))


def main():
    if sys.argv[1] == "l2":
        return l2_main(int(sys.argv[2]))
    case, nd, P = np.load(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    import rofl_project_code_amd as R
    from rofl_project_code_amd import api
    for k in range(1, nd):
        api.map_device(k, 0)
    R.set_device(0)
    vals, bl, seed, nb = case["vals"], case["bl"], bytes(case["seed"]), int(case["nb"])
    fp = (int(case["fp"][0]), int(case["fp"][1]))
    d = vals.size
    pr1, cm1 = R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.seeded(seed), fp=fp)          # one device
    R.set_option("devices", (1 << nd) - 1)
    prs, cms = R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.seeded(seed), fp=fp)          # chunks dealt to nd devices
    assert prs.shape == pr1.shape and (prs == pr1).all() and (cms == cm1).all(), "split bytes differ from the one-device call"
    if "opr" in case.files:
        assert (prs == case["opr"]).all() and (cms == case["ocm"]).all(), "split bytes differ from the oracle"
    if "ochunk_idx" in case.files:
        for c, op in zip(case["ochunk_idx"], case["ochunk_proofs"]):
            assert (prs[int(c)] == op).all(), "chunk %d differs from the oracle" % c
    used = [k for k in range(nd) if (R.set_device(k), api.bp_gens_table_bytes(nb, (1 << (d - 1).bit_length()) // prs.shape[0]))[1] > 0]
    R.set_device(0)
    assert len(used) == min(nd, prs.shape[0]), "expected every listed device to have proved a run: %r" % used
    assert R.range_proof_vec.verify_rangeproof(prs, cms, nb, verifier_seed=b"\x07" * 32, fp=fp) is True
    for c in sorted({0, prs.shape[0] // 2, prs.shape[0] - 1}):      # a bad proof in the first, a middle and the last run
        bad = prs.copy(); bad[c, 200] ^= 4
        assert R.range_proof_vec.verify_rangeproof(bad, cms, nb, verifier_seed=b"\x07" * 32, fp=fp) is False
    badc = cms.copy(); badc[d - 1] = cms[0]                               # a swapped commitment in the last real element
    assert R.range_proof_vec.verify_rangeproof(prs, badc, nb, verifier_seed=b"\x07" * 32, fp=fp) is False
    nonc = prs.copy(); nonc[prs.shape[0] - 1, 128:160] = 0xFF           # a non-canonical scalar in the last run: the call's FormatError
    try:
        R.range_proof_vec.verify_rangeproof(nonc, cms, nb, verifier_seed=b"\x07" * 32, fp=fp); raise AssertionError("no FormatError")
    except R.RoflError as e:
        assert e.code == 5
    vbad = vals.copy(); vbad[d - 1] = np.float32(3e9)
    try:
        R.range_proof_vec.create_rangeproof(vbad, bl, nb, P, nonce=R.Nonce.seeded(seed), fp=fp); raise AssertionError("no ValueOutOfRangeError")
    except R.RoflError as e:
        assert e.code == 2
    R.set_option("devices", 0)
    assert R.range_proof_vec.verify_rangeproof(prs, cms, nb, verifier_seed=b"\x08" * 32, fp=fp) is True
    print("split ok: d=%d P=%d devices=%d proofs=%s" % (d, P, nd, prs.shape))


if __name__ == "__main__":
    main()
