"""Input hardening at the C ABI and its Python mirror: proof sets that do not cover every chunk, mixed shapes in a
verification batch, non-canonical scalars, forged (n, m) that would make the device build tables, thread-local fp defaults."""
import ctypes
import threading

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu
L = orc.L_ORDER


@pytest.fixture(scope="module")
def R():
    import rofl_project_code_amd as R
    from rofl_project_code_amd import build
    build.build()
    R.set_device(0)
    return R


def _client(R, seed, d=8, nb=8, P=4, fp=(16, 7)):
    rng = np.random.default_rng(seed)
    vals = rng.uniform(-0.9, 0.9, d).astype(np.float32)
    bl = orc.rand_scalars(rng, d)
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.seeded(bytes([seed]) * 32), fp=fp)
    return pr, cm


def test_proofs_must_cover_every_chunk(R):
    """range_proof_vec/mod.rs:169-176 zips chunks with proofs and silently drops what is left: 3 proofs for 8 commitments
    would check 6 of them.  Here an incomplete cover does not verify (the oracle restates the reference and accepts it)."""
    fp = (16, 7)
    rng = np.random.default_rng(1)
    d, nb = 8, 8
    vals = rng.uniform(-0.9, 0.9, d).astype(np.float32)
    bl = orc.rand_scalars(rng, d)
    pr4, cm = R.range_proof_vec.create_rangeproof(vals, bl, nb, 4, nonce=R.Nonce.seeded(b"\x01" * 32), fp=fp)      # 4 chunks of 2
    assert R.range_proof_vec.verify_rangeproof(pr4, cm, nb, fp=fp) is True
    # 3 of the 4 proofs: chunk size stays 8 // 3 = 2, the last two commitments would go unchecked
    assert orc.verify_rangeproof(pr4[:3], cm, nb, 16, 7) == (0, True)            # the reference's behaviour
    assert R.range_proof_vec.verify_rangeproof(pr4[:3], cm, nb, fp=fp) is False
    # ... even when the unchecked tail is garbage-free but was never range-proved
    cm2 = cm.copy(); cm2[6] = cm[0]
    assert R.range_proof_vec.verify_rangeproof(pr4[:3], cm2, nb, fp=fp) is False
    # dp/2 + 1 proofs (5 for 8): chunk size 1, three commitments unchecked
    pr8, cm8 = R.range_proof_vec.create_rangeproof(vals, bl, nb, 8, nonce=R.Nonce.seeded(b"\x02" * 32), fp=fp)
    assert R.range_proof_vec.verify_rangeproof(pr8, cm8, nb, fp=fp) is True
    assert R.range_proof_vec.verify_rangeproof(pr8[:5], cm8, nb, fp=fp) is False
    assert R.range_proof_vec.verify_rangeproof_batch([pr8[:5], pr8], [cm8, cm8], nb, fp=fp) == [False, True]


def test_batch_with_mixed_shapes_and_bad_members(R):
    fp = (16, 7)
    a = _client(R, 1); b = _client(R, 2); c = _client(R, 3, d=16)           # c: a different (but valid) shape
    short = (a[0][:, :-32].copy(), a[1])                                      # truncated proofs: FormatError on its own
    fewer = (b[0][:2].copy(), b[1])                                           # fewer proofs than chunks
    badc = (b[0], b[1].copy()); badc[1][0] = np.frombuffer(bytes([1] + [0] * 31), np.uint8)      # not a Ristretto encoding
    res = R.range_proof_vec.verify_rangeproof_batch([a[0], short[0], c[0], fewer[0], badc[0], b[0]],
                                                    [a[1], short[1], c[1], fewer[1], badc[1], b[1]], 8, verifier_seed=b"\x09" * 32, fp=fp)
    assert res == [True, False, True, False, False, True]
    # the raw C entry with one undecodable member: that member fails, the batch does not
    ps = [a[0], badc[0], b[0]]; cs = [a[1], badc[1], b[1]]
    pp = (ctypes.c_void_p * 3)(*[p.ctypes.data for p in ps]); cp = (ctypes.c_void_p * 3)(*[x.ctypes.data for x in cs])
    ok = (ctypes.c_int * 3)()
    rc = R.lib().rofl_verify_rangeproof_batch(ctypes.c_size_t(3), pp, ctypes.c_size_t(ps[0].shape[1]), ctypes.c_size_t(ps[0].shape[0]), cp,
                                              ctypes.c_size_t(8), ctypes.c_size_t(8), 16, 7, b"\x01" * 32, ok)
    assert rc == 0 and list(ok) == [1, 0, 1]
    with pytest.raises(R.RoflError) as e:                                     # a single set keeps the reference's FormatError
        R.range_proof_vec.verify_rangeproof(badc[0], badc[1], 8, fp=fp)
    assert e.value.code == 5


def test_forged_shape_is_rejected_before_tables_are_built(R):
    """(prove_range, chunk) of a verification come off the wire; a proof whose length does not match them must be turned away
    before the generator cache allocates anything for that shape."""
    fp = (32, 7)
    pr, cm = _client(R, 4, d=8, nb=8, P=4, fp=fp)
    # claim 64-bit proofs: N = 64 * 2 = 128 != 2^lg of these 8-bit proofs -> Ok(false), no tables for (64, 2)
    import time
    t0 = time.perf_counter()
    assert R.range_proof_vec.verify_rangeproof(pr, cm, 64, fp=(64, 7)) is False
    # one "proof" for 2^20 commitments would ask for (8, 2^20) tables: rejected on the proof length alone
    big = np.zeros((1 << 20, 32), np.uint8)
    assert R.range_proof_vec.verify_rangeproof(pr[:1], big, 8, fp=fp) is False
    assert time.perf_counter() - t0 < 5.0


def test_non_canonical_scalars_are_reduced(R):
    fp = (16, 7)
    rng = np.random.default_rng(6)
    d = 5
    bl = orc.rand_scalars(rng, d)
    small = np.stack([np.frombuffer(int(7 + i).to_bytes(32, "little"), np.uint8) for i in range(d)])
    plus_l = np.stack([np.frombuffer((int.from_bytes(b.tobytes(), "little") + L).to_bytes(32, "little"), np.uint8) for b in small])      # >= l, < 2^253
    top = np.stack([np.frombuffer(((1 << 256) - 1 - i).to_bytes(32, "little"), np.uint8) for i in range(d)])                         # >= 2^255
    top_red = np.stack([np.frombuffer((((1 << 256) - 1 - i) % L).to_bytes(32, "little"), np.uint8) for i in range(d)])
    vals = orc.rand_scalars(rng, d)
    assert (R.pedersen_ops.commit_vec(vals, plus_l) == R.pedersen_ops.commit_vec(vals, small)).all()
    assert (R.pedersen_ops.commit_vec(top, bl) == R.pedersen_ops.commit_vec(top_red, bl)).all()
    assert (R.pedersen_ops.commit_vec(top_red, bl) == orc.commit_vec(top_red, bl)).all()
    x = rng.uniform(-0.9, 0.9, d).astype(np.float32)
    pr1, cm1 = R.range_proof_vec.create_rangeproof(x, plus_l, 8, 1, nonce=R.Nonce.seeded(b"\x05" * 32), fp=fp)
    pr2, cm2 = R.range_proof_vec.create_rangeproof(x, small, 8, 1, nonce=R.Nonce.seeded(b"\x05" * 32), fp=fp)
    assert (cm1 == cm2).all() and (pr1 == pr2).all()
    assert R.range_proof_vec.verify_rangeproof(pr1, cm1, 8, fp=fp)


def test_fp_default_is_per_thread(R):
    R.api.set_fp(32, 7)
    seen = {}

    def other():
        seen["default"] = R.api.get_fp()
        R.api.set_fp(8, 3)
        seen["own"] = R.api.get_fp()
    t = threading.Thread(target=other); t.start(); t.join()
    assert seen == {"default": (16, 7), "own": (8, 3)} and R.api.get_fp() == (32, 7)
    assert R.conversion32.get_clip_bounds(8, fp=(16, 7)) != R.conversion32.get_clip_bounds(8, fp=(16, 3))
    R.api.set_fp(16, 7)


def test_wire_check_percentage_out_of_range_does_not_raise(R):
    fp = (16, 7)
    rng = np.random.default_rng(8)
    x = (rng.integers(-50, 50, size=12) / 128.0).astype(np.float32)
    enc = R.EncParamsRange.encrypt(x, orc.rand_scalars(rng, 12), 8, 4, 1.0, nonce_seed=b"\x01" * 32, fp=fp)
    assert enc.verify(fp=fp)
    for cp in (float("inf"), float("nan"), -0.5, 1.5, 1e30):
        enc.check_percentage = cp
        assert enc.verify(fp=fp) is False


def test_create_batch_is_bit_identical_to_single_calls(R):
    """rofl_create_rangeproof_batch: several clients in one launch sequence; every client's proofs and commitments equal what
    rofl_create_rangeproof returns for it (and therefore the oracle's), bad clients are singled out."""
    for (d, nb, P, fp, nc) in ((37, 8, 4, (16, 7), 5), (300, 32, 4, (32, 7), 3), (1000, 16, 8, (16, 7), 4), (5, 8, 1, (16, 7), 6)):
        rng = np.random.default_rng(d)
        mn, mx = R.conversion32.get_clip_bounds(nb, fp=fp)
        vals = [np.clip(rng.uniform(mn, mx, d).astype(np.float32), mn, np.nextafter(np.float32(mx), np.float32(0))) for _ in range(nc)]
        bls = [orc.rand_scalars(rng, d) for _ in range(nc)]
        nonces = [R.Nonce.seeded(bytes([40 + i]) * 32) for i in range(nc)]
        if d == 37:      # one client with an explicit stream
            m = 64 // 4
            nonces[2] = R.Nonce.stream(rng.integers(0, 256, 4 * m * (2 * nb + 4) * 64, dtype=np.uint8).tobytes())
        res = R.range_proof_vec.create_rangeproof_batch(vals, bls, nb, P, nonces=nonces, fp=fp)
        for i in range(nc):
            pr, cm = R.range_proof_vec.create_rangeproof(vals[i], bls[i], nb, P, nonce=nonces[i], fp=fp)
            assert (res[i][0] == pr).all() and (res[i][1] == cm).all(), (d, i)
        kw = {"stream": nonces[2]._stream.tobytes()} if d == 37 else {"seed": bytes([42]) * 32}
        rc, opr, ocm = orc.create_rangeproof(vals[2], bls[2], nb, P, fp[0], fp[1], **kw)
        assert rc == 0 and (opr == res[2][0]).all() and (ocm == res[2][1]).all()
        assert R.range_proof_vec.verify_rangeproof_batch([r[0] for r in res], [r[1] for r in res], nb, fp=fp) == [True] * nc
    # per-client failures do not sink the batch
    fp = (16, 7)
    rng = np.random.default_rng(99)
    vals = [rng.uniform(-0.9, 0.9, 20).astype(np.float32) for _ in range(4)]
    bls = [orc.rand_scalars(rng, 20) for _ in range(4)]
    vals[1] = vals[1].copy(); vals[1][7] = 5.0             # out of the 8-bit range
    vals[3] = vals[3].copy(); vals[3][0] = np.nan
    nonces = [R.Nonce.seeded(bytes([i]) * 32) for i in range(4)]
    res = R.range_proof_vec.create_rangeproof_batch(vals, bls, 8, 4, nonces=nonces, fp=fp)
    assert isinstance(res[1], R.RoflError) and res[1].code == 2 and isinstance(res[3], R.RoflError) and res[3].code == 10
    for i in (0, 2):
        pr, cm = R.range_proof_vec.create_rangeproof(vals[i], bls[i], 8, 4, nonce=nonces[i], fp=fp)
        assert (res[i][0] == pr).all() and (res[i][1] == cm).all()


def test_behaviour_options_are_abi_calls(R):
    """switches that change what a call returns are set through rofl_set_option, not the process environment.
    "verify_zip_truncate" = 1 restores the reference's zip-truncation bit for bit (the oracle restates it); "verify_batch" = 0 checks
    every proof on its own like upstream verify_multiple; unknown keys and out-of-range values are BadParameter."""
    fp = (16, 7)
    rng = np.random.default_rng(11)
    d, nb = 8, 8
    vals = rng.uniform(-0.9, 0.9, d).astype(np.float32)
    bl = orc.rand_scalars(rng, d)
    pr4, cm = R.range_proof_vec.create_rangeproof(vals, bl, nb, 4, nonce=R.Nonce.seeded(b"\x01" * 32), fp=fp)
    assert R.get_option("verify_zip_truncate") == 0 and R.get_option("verify_batch") == 1
    assert R.range_proof_vec.verify_rangeproof(pr4[:3], cm, nb, fp=fp) is False
    try:
        R.set_option("verify_zip_truncate", 1)
        assert R.range_proof_vec.verify_rangeproof(pr4[:3], cm, nb, fp=fp) is True            # == orc / the reference
        assert orc.verify_rangeproof(pr4[:3], cm, nb, 16, 7) == (0, True)
        bad = cm.copy(); bad[1] = cm[0]                                                          # a covered commitment: still caught
        assert R.range_proof_vec.verify_rangeproof(pr4[:3], bad, nb, fp=fp) is False
    finally:
        R.set_option("verify_zip_truncate", 0)
    try:
        R.set_option("verify_batch", 0)
        assert R.range_proof_vec.verify_rangeproof(pr4, cm, nb, fp=fp) is True
        t = pr4.copy(); t[2, 50] ^= 1
        assert R.range_proof_vec.verify_rangeproof(t, cm, nb, fp=fp) is False
    finally:
        R.set_option("verify_batch", 1)
    for key, val in (("no_such_option", 1), ("verify_batch", 7), ("blocking_sync", -2)):
        with pytest.raises(R.RoflError) as e:
            R.set_option(key, val)
        assert e.value.code == 11


def test_batch_member_with_non_canonical_scalar_fails_alone(R):
    """a proof whose t_x / a / b bytes are not canonical scalars is a FormatError for a single set (RangeProof::from_bytes), but in
    a batch it must only cost THAT client its verdict -- the server verifies each client on its own (server.rs:656-687)."""
    fp = (16, 7)
    a = _client(R, 21, d=16, nb=8, P=4, fp=fp); b = _client(R, 22, d=16, nb=8, P=4, fp=fp); c = _client(R, 23, d=16, nb=8, P=4, fp=fp)
    evil = b[0].copy(); evil[1, 128:160] = 0xFF                     # t_x of chunk 1 >= l
    assert R.range_proof_vec.verify_rangeproof_batch([a[0], evil, c[0]], [a[1], b[1], c[1]], 8, verifier_seed=b"\x03" * 32, fp=fp) == [True, False, True]
    evil2 = c[0].copy(); evil2[3, -32:] = 0xFF                       # the final b of the last chunk
    assert R.range_proof_vec.verify_rangeproof_batch([evil2, a[0]], [c[1], a[1]], 8, verifier_seed=b"\x03" * 32, fp=fp) == [False, True]
    with pytest.raises(R.RoflError) as e:
        R.range_proof_vec.verify_rangeproof(evil, b[1], 8, fp=fp)
    assert e.value.code == 5


def test_colliding_scalars_do_not_reach_the_gather(R):
    """a proof with a = 0 gives every G term of the verifier's generator MSM the same scalar (-z): all 2^18 items of a
    window land in ONE bucket, the coarse bin of the two-level sort overflows, and what the overflowed bin leaves behind must not be used as
    window-table indices by the accumulation that runs before the host sees the flag.  The call has to come back (verdict: false), and
    the lane must be healthy afterwards."""
    fp = (32, 7)
    rng = np.random.default_rng(77)
    d, nb, P = 16384, 32, 1                                            # one chunk of 2^19 terms: the fixed-base launches take the two-level sort
    mx = np.float32(((1 << 31) - 1) / 128.0)
    vals = np.clip(rng.uniform(-mx, mx, size=d).astype(np.float32), -mx, np.nextafter(mx, np.float32(0)))
    bl = orc.rand_scalars(rng, d)
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.seeded(b"\x61" * 32), fp=fp)
    assert R.range_proof_vec.verify_rangeproof(pr, cm, nb, verifier_seed=b"\x01" * 32, fp=fp) is True
    for which in (-64, -32):                                           # a = 0, then b = 0 (every H term then carries z + y^-k zz 2^i z^j: distinct; a = 0 is the colliding one)
        evil = pr.copy(); evil[0, which:(which + 32) or None] = 0
        assert R.range_proof_vec.verify_rangeproof(evil, cm, nb, verifier_seed=b"\x01" * 32, fp=fp) is False
    both = pr.copy(); both[0, -64:] = 0
    assert R.range_proof_vec.verify_rangeproof(both, cm, nb, verifier_seed=b"\x02" * 32, fp=fp) is False
    assert R.range_proof_vec.verify_rangeproof(pr, cm, nb, verifier_seed=b"\x03" * 32, fp=fp) is True
    pr2, cm2 = R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.seeded(b"\x61" * 32), fp=fp)
    assert (pr2 == pr).all() and (cm2 == cm).all()


def test_batch_argument_lengths_are_checked(R):
    vals = np.zeros(4, np.float32); bl = np.zeros((4, 32), np.uint8)
    with pytest.raises(ValueError):
        R.range_proof_vec.create_rangeproof_batch([vals, vals], [bl], 8, 1, fp=(16, 7))
    with pytest.raises(ValueError):
        R.range_proof_vec.create_rangeproof_batch([vals, vals], [bl, bl], 8, 1, nonces=[R.Nonce.seeded(b"\x01" * 32)], fp=(16, 7))


def test_caller_memory_is_staged_for_any_size(R):
    """Transfers between caller memory and the device go through the lane's pinned staging arena from 32 KB on and straight through
    below that (csrc/host_rt.hpp `Stage`): sizes on both sides of the threshold, a transfer larger than the arena's first chunk (the
    arena grows, then coalesces its chunks when the call ends), fresh output arrays every time, results dropped in between -- the
    bytes must be the oracle's every time.  (Pageable buffers handed to hipMemcpyAsync used to cost 20-30 ms after a free.)"""
    rng = np.random.default_rng(20261002)
    for d in (5, 900, 1100, 30000, 200000, 1100, 200000, 7):
        v = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); v[:, 31] &= 0x0F
        b = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); b[:, 31] &= 0x0F
        got = R.pedersen_ops.commit_vec(v, b)
        pick = np.unique(np.concatenate([np.arange(min(d, 40)), np.arange(max(0, d - 40), d), rng.integers(0, d, size=60)]))
        want = orc.commit_vec(v[pick], b[pick])
        assert (got[pick] == want).all(), d
        s = R.pedersen_ops.add_rp_vec(got, got)          # two staged inputs, one staged output
        assert (s[pick] == orc.add_points_vec(want, want)[1]).all(), d
        del got, s


def test_verify_with_a_free_transcript_label(R):
    """rofl_dbg_verify_labelled (debug header): verify_multiple as upstream's own tests call it -- any label, unshifted commitments, generator
    capacity >= n.  The L2 sum proof (label "L2RangeProof", no shift) and one chunk of an L-inf proof (label "RangeProof", commitments shifted
    by 2^(n-1) B as range_proof_vec/mod.rs:155-167 does) are accepted under their labels and rejected under any other."""
    import ctypes
    L = R.lib()
    fp = (32, 7)
    rng = np.random.default_rng(31)

    def labelled(label, cap, proof, commits, m, n):
        ok = ctypes.c_int(-1)
        p = np.ascontiguousarray(proof, np.uint8).reshape(-1); c = np.ascontiguousarray(commits, np.uint8).reshape(-1, 32)
        rc = L.rofl_dbg_verify_labelled(label, ctypes.c_size_t(len(label)), ctypes.c_size_t(cap), p.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(p.size),
                                        c.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(m), ctypes.c_size_t(n), b"\x09" * 32, ctypes.byref(ok))
        return rc, ok.value
    d = 40
    x = (rng.integers(-3, 4, size=d) / 128.0).astype(np.float32); r2 = orc.rand_scalars(rng, d)
    proof, commit = R.l2_range_proof_vec.create_rangeproof_l2(x, r2, 32, 4, nonce=R.Nonce.seeded(b"\x11" * 32), fp=fp)
    assert labelled(b"L2RangeProof", 64, proof, commit, 1, 32) == (0, 1)
    assert labelled(b"L2RangeProof", 32, proof, commit, 1, 32) == (0, 1)
    assert labelled(b"RangeProof", 64, proof, commit, 1, 32) == (0, 0)
    assert labelled(b"Deserialize-And-Verify Test", 64, proof, commit, 1, 32) == (0, 0)
    assert labelled(b"L2RangeProof", 16, proof, commit, 1, 32)[0] == 6          # InvalidGeneratorsLength
    bad = proof.copy(); bad[3] ^= 1
    assert labelled(b"L2RangeProof", 64, bad, commit, 1, 32) in ((0, 0), (5, 0))
    vals = rng.uniform(-0.9, 0.9, 64).astype(np.float32); bl = orc.rand_scalars(rng, 64)
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, 8, 4, nonce=R.Nonce.seeded(b"\x12" * 32), fp=(16, 7))
    shift = R.pedersen_ops.commit_no_blinding_vec(np.frombuffer((128).to_bytes(32, "little"), np.uint8).reshape(1, 32))[0]
    V = R.pedersen_ops.compute_shifted_values_rp(cm, shift)
    assert labelled(b"RangeProof", 8, pr[1], V[16:32], 16, 8) == (0, 1)
    assert labelled(b"RangeProof", 8, pr[1], cm[16:32], 16, 8) == (0, 0)       # the unshifted commitments are not what was proved
    assert labelled(b"rangeproof", 8, pr[1], V[16:32], 16, 8) == (0, 0)
    assert labelled(b"RangeProof", 8, pr[1], V[16:28], 12, 8)[0] == 11          # m must be a power of two
