"""The server's side of an L2 round (BASELINE cfg 5; rofl_service/src/flserver/params.rs:204-232 for every client of a round, server.rs:656-687,
and :474-484 -- one failing client fails the round): EncParamsL2.verify_batch = the square proofs of all clients in one launch sequence
(rofl_verify_squarerandproof_vec_batch, which also returns every client's sum of c_sq), the L-inf legs through
rofl_verify_rangeproof_batch_strided (commitments read in place from the 96-byte records) and the L2 sum proofs through
rofl_verify_rangeproof_l2_batch.  What it must reject: a tampered member in each of the three legs, malformed members, swapped
components -- verdict lists equal to the per-client verify()'s, the oracle agreeing on every tampered component."""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu
FP = (32, 7)


@pytest.fixture(scope="module")
def R():
    import rofl_project_code_amd as R
    from rofl_project_code_amd import build
    build.build()
    R.set_device(0)
    yield R
    R.set_option("verify_batch", 1)


def _client(R, seed, d, nb, P, l2n, cls=None):
    rng = np.random.default_rng(seed)
    x = (rng.integers(-3, 4, size=d) / 128.0).astype(np.float32)
    bl = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); bl[:, 31] &= 0x0F
    r2 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r2[:, 31] &= 0x0F
    return (cls or R.EncParamsL2).encrypt(x, bl, nb, P, l2n, nonce_seed=bytes([seed % 251]) * 32, rand_scalars=r2, fp=FP)


def _copy(R, u):
    return type(u).deserialize(u.serialize())


def _both(R, ups, seed=b"\x21" * 32):
    """verify_batch with one equation per batch (verify_batch = 2) and per client (1), against every client's own verify()"""
    single = [u.verify(verifier_seed=seed, fp=FP) for u in ups]
    for vb in (2, 1):
        R.set_option("verify_batch", vb)
        got = type(ups[0]).verify_batch(ups, verifier_seed=seed, fp=FP)
        assert got == single, (vb, got, single)
    R.set_option("verify_batch", 1)
    return single


@pytest.mark.parametrize("shape", [dict(d=300, nb=8, P=4, l2n=32, n=7), dict(d=5000, nb=8, P=4, l2n=32, n=5)], ids=["d300", "d5000"])
def test_l2_round_verdicts_match_the_per_client_path(R, shape):
    d, nb, P, l2n, n = (shape[k] for k in ("d", "nb", "P", "l2n", "n"))
    ups = [_client(R, 100 + i, d, nb, P, l2n) for i in range(n)]
    try:
        assert _both(R, ups) == [True] * n
        # the by-product sums are the reference's sum of c_sq, and the oracle accepts a member's sum proof against it
        ok, sums = R.square_rand_proof_vec.verify_l2rangeproof_vec_batch([u.square_proofs for u in ups], [u.enc_values for u in ups], with_csq_sums=True)
        assert ok == [True] * n
        for i in (0, n - 1):
            assert (sums[i] == ups[i]._sum_c_sq()).all()
            assert orc.verify_rangeproof_l2(ups[i].square_range_proof, sums[i], l2n, FP[0], FP[1]) == (0, True)
        # one tampered member per leg
        t = [_copy(R, u) for u in ups]
        t[1].square_proofs[d // 2, 64 + 40] ^= 1            # c_sq' of one element's square proof
        t[2].range_proofs[P - 1, 7 * 32 + 33] ^= 2          # an R of the L-inf leg's last chunk
        t[4].square_range_proof[5 * 32 + 3] ^= 1            # t_x_blinding of the L2 sum proof
        want = [True] * n
        for i in (1, 2, 4): want[i] = False
        assert _both(R, t) == want
        assert orc.sigma_verify(1, t[1].square_proofs, t[1].enc_values) in ((0, False), (5, False))
        assert orc.verify_rangeproof(t[2].range_proofs, t[2].enc_values[:, :32].copy(), nb, FP[0], FP[1])[1] is False
        assert orc.verify_rangeproof_l2(t[4].square_range_proof, t[4]._sum_c_sq(), l2n, FP[0], FP[1])[1] is False
        # the components must belong together: c_sq of one element replaced (square proof AND sum break), two clients' sum proofs swapped
        t = [_copy(R, u) for u in ups]
        t[0].enc_values[7, 64:96] = t[0].enc_values[8, 64:96]
        t[3].square_range_proof, t[4].square_range_proof = t[4].square_range_proof.copy(), t[3].square_range_proof.copy()
        want = [True] * n
        for i in (0, 3, 4): want[i] = False
        assert _both(R, t) == want
        # malformed members cost the others nothing: a non-canonical response scalar, an undecodable commitment, a non-canonical scalar in each range proof
        t = [_copy(R, u) for u in ups]
        t[0].square_proofs[3, 96:128] = 0xFF                # Z_m >= l
        t[1].enc_values[9, 32:64] = np.frombuffer(bytes([1] + [0] * 31), np.uint8)      # R: not a Ristretto encoding
        t[2].range_proofs[0, 128:160] = 0xFF                # t_x of the L-inf leg
        t[3].square_range_proof[128:160] = 0xFF             # t_x of the sum proof
        want = [False] * 4 + [True] * (n - 4)
        assert _both(R, t) == want
        # batches of one and two
        assert _both(R, ups[:1]) == [True] and _both(R, ups[:2]) == [True, True]
        # a member of another shape is verified on its own
        odd = _client(R, 999, d // 2 + 1, nb, P, l2n)
        R.set_option("verify_batch", 2)
        assert R.EncParamsL2.verify_batch(ups[:3] + [odd], verifier_seed=b"\x01" * 32, fp=FP) == [True] * 4
    finally:
        R.set_option("verify_batch", 1)


def test_l2_compressed_round(R):
    """EncParamsL2Compressed (what the paper's e2e runs use, cifar_large.yml:56,74,100): the square proofs are SquareProofs over (c.L, c_sq)"""
    d, nb, P, l2n, n = 400, 8, 4, 32, 5
    ups = [_client(R, 300 + i, d, nb, P, l2n, cls=R.EncParamsL2Compressed) for i in range(n)]
    try:
        assert _both(R, ups) == [True] * n
        t = [_copy(R, u) for u in ups]
        t[2].square_proofs[11, 70] ^= 1
        t[4].square_range_proof[40] ^= 1
        assert _both(R, t) == [True, True, False, True, False]
    finally:
        R.set_option("verify_batch", 1)


def test_sigma_batches_of_the_other_kinds_and_options(R):
    """rofl_verify_randproof_vec_batch (the EncRange arm's per-element proofs); sigma_batch = 0 keeps the per-element check, client by client"""
    rng = np.random.default_rng(5)
    d, n = 700, 6
    vecs = []
    for i in range(n):
        x = rng.uniform(-3, 3, size=d).astype(np.float32); r = orc.rand_scalars(rng, d)
        vecs.append(R.rand_proof_vec.create_randproof_vec(x, r, nonce=R.Nonce.seeded(bytes([i + 1]) * 32), fp=FP))
    pr = [p.copy() for p, _ in vecs]; cm = [c.copy() for _, c in vecs]
    assert R.rand_proof_vec.verify_randproof_vec_batch(pr, cm) == [True] * n
    pr[2][d - 1, 64] ^= 1; cm[5][0, 0] ^= 1
    want = [True, True, False, True, True, False]
    for sb in (1, 0):
        R.set_option("sigma_batch", sb)
        try:
            assert R.rand_proof_vec.verify_randproof_vec_batch(pr, cm) == want, sb
        finally:
            R.set_option("sigma_batch", 1)
    assert orc.sigma_verify(0, pr[2], cm[2])[1] is False
    assert R.rand_proof_vec.verify_randproof_vec_batch([], []) == []


def test_l2_round_at_cfg5_size(R):
    """Twelve clients of BASELINE cfg 5's shape (d = 55 000, 8-bit L-inf legs, 32-bit sum proofs, n_partition = 4): the round verifies; a late
    element's square proof, a chunk's range proof and a sum proof tampered in three different clients are each attributed to their client;
    the oracle rejects the tampered components (the square proofs on the slice around the element)."""
    d, nb, P, l2n, n = 55000, 8, 4, 32, 12
    ups = [_client(R, 500 + i, d, nb, P, l2n) for i in range(n)]
    seed = b"\x33" * 32
    try:
        R.set_option("verify_batch", 2)
        assert R.EncParamsL2.verify_batch(ups, verifier_seed=seed, fp=FP) == [True] * n
        t = [_copy(R, u) for u in ups]
        t[3].square_proofs[54321, 128 + 9] ^= 1             # Z_r1 of a late element
        t[7].range_proofs[2, 7 * 32 + 64 * 5 + 1] ^= 1
        t[11].square_range_proof[-20] ^= 1                  # b of the sum proof's inner-product argument
        want = [True] * n
        for i in (3, 7, 11): want[i] = False
        assert R.EncParamsL2.verify_batch(t, verifier_seed=seed, fp=FP) == want
        R.set_option("verify_batch", 1)
        assert R.EncParamsL2.verify_batch(t, verifier_seed=seed, fp=FP) == want
        for i in (3, 7, 11):
            assert t[i].verify(verifier_seed=seed, fp=FP) is False
        assert orc.sigma_verify(1, t[3].square_proofs[54300:54340].copy(), t[3].enc_values[54300:54340].copy())[1] is False
        assert orc.sigma_verify(1, ups[3].square_proofs[54300:54340].copy(), ups[3].enc_values[54300:54340].copy()) == (0, True)
        chunk = 65536 // P
        assert orc.verify_rangeproof(t[7].range_proofs[2:3].copy(), t[7].enc_values[2 * chunk:3 * chunk, :32].copy(), nb, FP[0], FP[1])[1] is False
        assert orc.verify_rangeproof_l2(t[11].square_range_proof, t[11]._sum_c_sq(), l2n, FP[0], FP[1])[1] is False
    finally:
        R.set_option("verify_batch", 1)


def test_encrypt_batch_is_byte_identical_to_single_encrypts(R):
    """EncParamsL2.encrypt_batch: the L-inf legs of several clients as one rofl_create_rangeproof_batch call; every container equals encrypt()'s"""
    d, nb, P, l2n, n = 700, 8, 4, 32, 5
    cl, seeds = [], []
    for i in range(n):
        rng = np.random.default_rng(700 + i)
        x = (rng.integers(-3, 4, size=d) / 128.0).astype(np.float32)
        bl = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); bl[:, 31] &= 0x0F
        r2 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r2[:, 31] &= 0x0F
        cl.append((x, bl, r2)); seeds.append(bytes([i + 9]) * 32)
    got = R.EncParamsL2.encrypt_batch(cl, nb, P, l2n, nonce_seeds=seeds, fp=FP)
    for i in range(n):
        one = R.EncParamsL2.encrypt(cl[i][0], cl[i][1], nb, P, l2n, nonce_seed=seeds[i], rand_scalars=cl[i][2], fp=FP)
        assert got[i].serialize() == one.serialize(), i
    assert R.EncParamsL2.verify_batch(got, verifier_seed=b"\x02" * 32, fp=FP) == [True] * n


def test_l2_round_over_two_logical_devices(R):
    """One server process, two devices (rofl_set_option("devices", 0b11); both logical devices are GPU 0 on the test box): the square-proof batch and
    the range legs of a round are dealt to the devices inside the library; verdicts and the sums of c_sq are those of the one-device call."""
    from rofl_project_code_amd import api
    api.map_device(1, 0)
    d, nb, P, l2n, n = 900, 8, 4, 32, 6
    ups = [_client(R, 800 + i, d, nb, P, l2n) for i in range(n)]
    t = [_copy(R, u) for u in ups]
    t[1].square_proofs[17, 5] ^= 1; t[4].range_proofs[0, 40] ^= 1
    seed = b"\x44" * 32
    try:
        R.set_option("verify_batch", 2)
        one = R.EncParamsL2.verify_batch(t, verifier_seed=seed, fp=FP)
        ok1, s1 = R.square_rand_proof_vec.verify_l2rangeproof_vec_batch([u.square_proofs for u in ups], [u.enc_values for u in ups], with_csq_sums=True)
        R.set_option("devices", 0b11)
        two = R.EncParamsL2.verify_batch(t, verifier_seed=seed, fp=FP)
        ok2, s2 = R.square_rand_proof_vec.verify_l2rangeproof_vec_batch([u.square_proofs for u in ups], [u.enc_values for u in ups], with_csq_sums=True)
        assert one == two == [True, False, True, True, False, True]
        assert ok1 == ok2 == [True] * n and (s1 == s2).all()
    finally:
        R.set_option("devices", 0); R.set_option("verify_batch", 1)
