"""GPU parity tests: the HIP path (through the C ABI) vs the oracle on identical seeded inputs, vs the committed
golden fixtures, plus size-independent properties at BASELINE.json's full sizes.  Integer work: bit-exact."""
import ctypes
import os

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu
H = bytes.fromhex


@pytest.fixture(scope="module")
def R():
    import rofl_project_code_amd as R
    from rofl_project_code_amd import build
    build.build()
    R.set_device(0)     # fails loudly when there is no HIP device / library
    return R


def _inputs(R, rng, d, nb, fb, ff):
    R.api.set_fp(fb, ff)
    mn, mx = R.conversion32.get_clip_bounds(nb)
    vals = rng.uniform(mn, mx, size=d).astype(np.float32)
    vals = np.clip(vals, mn, np.nextafter(np.float32(mx), np.float32(0)))   # half-open, SURVEY 8(d)
    return vals, orc.rand_scalars(rng, d)


def test_generators_match_oracle(R):
    for n, m in ((8, 4), (32, 2), (64, 1)):
        G = np.zeros((n * m, 32), np.uint8); Hh = np.zeros((n * m, 32), np.uint8)
        rc = R.lib().rofl_bp_gens_export(ctypes.c_size_t(n), ctypes.c_size_t(m), G.ctypes.data_as(ctypes.c_void_p), Hh.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0
        oG, oH = orc.bp_gens(n, m)
        assert (G == oG).all() and (Hh == oH).all()


def test_generators_match_golden(R, prim):
    G = np.zeros((9 * 6, 32), np.uint8); Hh = np.zeros((9 * 6, 32), np.uint8)
    # gens_capacity 9 is not a proof size, but the chain prefix property makes it a valid table
    assert R.lib().rofl_bp_gens_export(ctypes.c_size_t(9), ctypes.c_size_t(6), G.ctypes.data_as(ctypes.c_void_p), Hh.ctypes.data_as(ctypes.c_void_p)) == 0
    for name, lst in prim["generators"].items():
        arr = G if name[0] == "G" else Hh
        for i, enc in enumerate(lst):
            assert arr[int(name[1:]) * 9 + i].tobytes().hex() == enc


@pytest.mark.parametrize("d,nb,P,fb,ff", [(1, 8, 1, 16, 7), (3, 16, 4, 16, 7), (100, 8, 4, 16, 7), (16, 32, 4, 32, 7),
                                           (37, 16, 8, 16, 7), (300, 8, 4, 16, 7), (5, 64, 2, 64, 7), (200, 32, 1, 32, 12),
                                           (1000, 32, 4, 32, 7), (5, 8, 3, 16, 7)])
def test_create_bit_exact_vs_oracle(R, d, nb, P, fb, ff):
    rng = np.random.default_rng(d * 1000 + nb)
    vals, bl = _inputs(R, rng, d, nb, fb, ff)
    seed = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.seeded(seed))
    rc, opr, ocm = orc.create_rangeproof(vals, bl, nb, P, fb, ff, seed=seed)
    assert rc == 0
    assert (cm == ocm).all(), "commitments differ"
    assert pr.shape == opr.shape and (pr == opr).all(), "proof bytes differ"
    # cross verification both ways, and tamper tests (Ok(false), not Err)
    assert R.range_proof_vec.verify_rangeproof(pr, cm, nb, verifier_seed=b"\x01" * 32)
    assert R.range_proof_vec.verify_rangeproof(opr, ocm, nb, verifier_seed=b"\x02" * 32)
    assert orc.verify_rangeproof(pr, cm, nb, fb, ff) == (0, True)
    bad = pr.copy(); bad[0, 5 * 32 + 3] ^= 1
    assert not R.range_proof_vec.verify_rangeproof(bad, cm, nb, verifier_seed=b"\x03" * 32)
    bad = pr.copy(); bad[-1, 7 * 32 + 40] ^= 4
    assert not R.range_proof_vec.verify_rangeproof(bad, cm, nb, verifier_seed=b"\x03" * 32)
    if d > 1:
        badc = cm.copy(); badc[0] = cm[1]
        if not (cm[0] == cm[1]).all():
            assert not R.range_proof_vec.verify_rangeproof(pr, badc, nb, verifier_seed=b"\x04" * 32)


def test_explicit_nonce_stream_mode(R):
    rng = np.random.default_rng(5)
    d, nb, P, fb, ff = 6, 8, 2, 16, 7
    vals, bl = _inputs(R, rng, d, nb, fb, ff)
    n_sc = 2 * 4 * (2 * nb + 4)
    stream = rng.integers(0, 256, n_sc * 64, dtype=np.uint8).tobytes()
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.stream(stream))
    rc, opr, ocm = orc.create_rangeproof(vals, bl, nb, P, fb, ff, stream=stream)
    assert rc == 0 and (pr == opr).all() and (cm == ocm).all()
    with pytest.raises(R.RoflError) as e:
        R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.stream(stream[:64 * 10]))
    assert e.value.code == 12


def test_golden_fixtures(R, golden_proofs):
    def nonce(g):
        return R.Nonce.stream(H(g["stream"])) if g.get("nonce") == "stream" else R.Nonce.seeded(H(g["seed"]))
    for g in golden_proofs:
        if g["kind"] == "tie_cases":
            for c in g["cases"]:
                got = R.conversion32.f32_to_scalar_vec(np.array([c["v"]], np.float32), fp=(c["fp_bits"], c["fp_frac"]))
                assert got[0].tobytes().hex() == c["scalar"]
            continue
        R.api.set_fp(g["fp_bits"], g["fp_frac"])
        if g["kind"] in ("rand", "sqrand"):
            r1 = np.frombuffer(H(g["r1"]), np.uint8).reshape(-1, 32); r2 = np.frombuffer(H(g["r2"]), np.uint8).reshape(-1, 32)
            if g["kind"] == "rand":
                pr, cm = R.rand_proof_vec.create_randproof_vec(g["values"], r1, nonce=nonce(g))
                assert R.rand_proof_vec.verify_randproof_vec(pr, cm)
            else:
                pr, cm = R.square_rand_proof_vec.create_l2rangeproof_vec(g["values"], r1, r2, nonce=nonce(g))
                assert R.square_rand_proof_vec.verify_l2rangeproof_vec(pr, cm)
            assert pr.tobytes().hex() == g["proofs"] and cm.tobytes().hex() == g["commits"]
            continue
        bl = np.frombuffer(H(g["blindings"]), np.uint8).reshape(-1, 32)
        if g["kind"] == "linf":
            pr, cm = R.range_proof_vec.create_rangeproof(g["values"], bl, g["prove_range"], g["n_partition"], nonce=nonce(g))
            assert pr.tobytes().hex() == g["proofs"] and cm.tobytes().hex() == g["commits"]
            assert R.range_proof_vec.verify_rangeproof(pr, cm, g["prove_range"], verifier_seed=b"\x07" * 32)
        else:
            pr, cm = R.l2_range_proof_vec.create_rangeproof_l2(g["values"], bl, g["prove_range"], g["n_partition"], nonce=nonce(g))
            assert pr.tobytes().hex() == g["proofs"] and cm.tobytes().hex() == g["commits"]
            assert R.l2_range_proof_vec.verify_rangeproof_l2(pr, cm, g["prove_range"], verifier_seed=b"\x07" * 32)


def test_reference_semantic_tests_on_gpu(R, prim):
    # range_proof_vec/mod.rs:318-332 and :369-399 through the C ABI
    R.api.set_fp(16, 7)
    ref = prim["reference_values_fp16_frac7"]
    cms = []
    rng = np.random.default_rng(9)
    vecs = {"x": [0.25, 1.25, -1.5], "y": [-0.75, 1.25, -2.0], "z": [0.5, 1.25, -3.0]}
    for name, vec in vecs.items():
        pr, cm = R.range_proof_vec.create_rangeproof(vec, np.zeros((3, 32), np.uint8), 16, 4, nonce=R.Nonce.seeded(b"\x01" * 32))
        assert [c.tobytes().hex() for c in cm] == ref[name]
        assert R.range_proof_vec.verify_rangeproof(pr, cm, 16)
        cms.append(cm)
    tot = R.pedersen_ops.add_rp_vec_vec(cms)
    assert [c.tobytes().hex() for c in tot] == ref["sum"]
    # clipping + errors
    with pytest.raises(R.RoflError) as e:
        R.range_proof_vec.create_rangeproof([5.0], orc.rand_scalars(rng, 1), 8, 1)
    assert e.value.code == 2
    with pytest.raises(R.RoflError) as e:
        R.range_proof_vec.create_rangeproof([0.5], orc.rand_scalars(rng, 2), 8, 1)
    assert e.value.code == 1
    with pytest.raises(R.RoflError) as e:
        R.range_proof_vec.create_rangeproof([0.5] * 9, orc.rand_scalars(rng, 9), 8, 3)
    assert e.value.code == 4
    with pytest.raises(R.RoflError) as e:
        R.range_proof_vec.create_rangeproof([float("nan")], orc.rand_scalars(rng, 1), 8, 1)
    assert e.value.code == 10
    with pytest.raises(R.RoflError) as e:
        R.range_proof_vec.create_rangeproof([], np.zeros((0, 32), np.uint8), 8, 1)
    assert e.value.code == 11
    # wrap-around quirk at x == +fp_max with prove_range == N_BITS: same wrong commitment as the oracle
    R.api.set_fp(32, 7)
    v = [16777216.0]
    bl = orc.rand_scalars(rng, 1)
    pr, cm = R.range_proof_vec.create_rangeproof(v, bl, 32, 1, nonce=R.Nonce.seeded(b"\x02" * 32))
    rc, opr, ocm = orc.create_rangeproof(v, bl, 32, 1, 32, 7, seed=b"\x02" * 32)
    assert (pr == opr).all() and (cm == ocm).all()
    R.api.set_fp(16, 7)


def test_format_and_identity_rejections(R):
    R.api.set_fp(16, 7)
    rng = np.random.default_rng(2)
    pr, cm = R.range_proof_vec.create_rangeproof([0.5, 0.25], orc.rand_scalars(rng, 2), 8, 2, nonce=R.Nonce.seeded(b"\x03" * 32))
    bad = pr.copy(); bad[0, 4 * 32:5 * 32] = 0xFF
    with pytest.raises(R.RoflError) as e:
        R.range_proof_vec.verify_rangeproof(bad, cm, 8)
    assert e.value.code == 5
    with pytest.raises(R.RoflError) as e:
        R.range_proof_vec.verify_rangeproof(pr[:, :-32], cm, 8)
    assert e.value.code == 5
    bad = pr.copy(); bad[1, 0:32] = 0
    assert R.range_proof_vec.verify_rangeproof(bad, cm, 8) is False
    bad = pr.copy(); bad[0, 32:64] = np.frombuffer(bytes([1] + [0] * 31), np.uint8)      # S does not decompress -> VerificationError -> false
    assert R.range_proof_vec.verify_rangeproof(bad, cm, 8) is False
    badc = cm.copy(); badc[0] = np.frombuffer(bytes([1] + [0] * 31), np.uint8)
    with pytest.raises(R.RoflError) as e:
        R.range_proof_vec.verify_rangeproof(pr, badc, 8)
    assert e.value.code == 5


def test_l2_path(R):
    R.api.set_fp(16, 7)
    rng = np.random.default_rng(4)
    bl = orc.rand_scalars(rng, 3)
    pr, cm = R.l2_range_proof_vec.create_rangeproof_l2([1.25, 0.5, 0.25], bl, 16, 4, nonce=R.Nonce.seeded(b"\x05" * 32))
    rc, opr, ocm = orc.create_rangeproof_l2([1.25, 0.5, 0.25], bl, 16, 4, 16, 7, seed=b"\x05" * 32)
    assert rc == 0 and (pr == opr).all() and (cm == ocm).all()
    assert R.l2_range_proof_vec.verify_rangeproof_l2(pr, cm, 16)
    assert orc.verify_rangeproof_l2(pr, cm, 16, 16, 7) == (0, True)
    for vals, code in (([8.0], 8), ([6.0, 6.0], 8)):
        with pytest.raises(R.RoflError) as e:
            R.l2_range_proof_vec.create_rangeproof_l2(vals, orc.rand_scalars(rng, len(vals)), 16, 16)
        assert e.value.code == orc.create_rangeproof_l2(vals, orc.rand_scalars(rng, len(vals)), 16, 16, 16, 7, seed=b"\x00" * 32)[0] == code
    fake = orc.commit_vec(np.frombuffer((1 << 17).to_bytes(32, "little"), np.uint8).reshape(1, 32), orc.rand_scalars(rng, 1))[0]
    assert R.l2_range_proof_vec.verify_rangeproof_l2(pr, fake, 16) is False
    # BASELINE config 3 shape: d = 25 000 on the quantisation grid (bench l2rangeproof inputs), fp32
    R.api.set_fp(32, 7)
    d = 25000
    vals = (rng.integers(-3, 4, size=d) / 128.0).astype(np.float32)
    bl = orc.rand_scalars(rng, 64)
    bl = np.tile(bl, (d // 64 + 1, 1))[:d]
    pr, cm = R.l2_range_proof_vec.create_rangeproof_l2(vals, bl, 32, 4, nonce=R.Nonce.seeded(b"\x06" * 32))
    rc, opr, ocm = orc.create_rangeproof_l2(vals, bl, 32, 4, 32, 7, seed=b"\x06" * 32)
    assert rc == 0 and (pr == opr).all() and (cm == ocm).all()
    assert R.l2_range_proof_vec.verify_rangeproof_l2(pr, cm, 32)
    R.api.set_fp(16, 7)


def test_pedersen_ops(R):
    rng = np.random.default_rng(6)
    s, b = orc.rand_scalars(rng, 70), orc.rand_scalars(rng, 70)
    c1 = R.pedersen_ops.commit_vec(s, b)
    assert (c1 == orc.commit_vec(s, b)).all()
    c0 = R.pedersen_ops.commit_no_blinding_vec(s)
    assert (c0 == orc.commit_vec(s, None)).all()
    rc, osum = orc.add_points_vec(c1, c0)
    assert (R.pedersen_ops.add_rp_vec(c1, c0) == osum).all()
    sh = R.pedersen_ops.compute_shifted_values_rp(c1, c0[0])
    rc, osh = orc.add_points_vec(c1, np.tile(c0[0], (70, 1)))
    assert (sh == osh).all()
    # homomorphism: commit(a, r) + commit(b, s) == commit(a + b, r + s)
    def addsc(x, y):
        return np.stack([np.frombuffer(((int.from_bytes(x[i].tobytes(), "little") + int.from_bytes(y[i].tobytes(), "little")) % orc.L_ORDER).to_bytes(32, "little"), np.uint8) for i in range(len(x))])
    s2, b2 = orc.rand_scalars(rng, 70), orc.rand_scalars(rng, 70)
    assert (R.pedersen_ops.add_rp_vec(c1, R.pedersen_ops.commit_vec(s2, b2)) == R.pedersen_ops.commit_vec(addsc(s, s2), addsc(b, b2))).all()
    assert (R.pedersen_ops.commit_no_blinding_vec(np.zeros((4, 32), np.uint8)) == 0).all()     # zero_rp_vec


def test_batch_verify(R):
    R.api.set_fp(16, 7)
    rng = np.random.default_rng(8)
    prs, cms = [], []
    for c in range(5):
        vals, bl = _inputs(R, rng, 50, 8, 16, 7)
        pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, 8, 4, nonce=R.Nonce.seeded(bytes([c]) * 32))
        prs.append(pr); cms.append(cm)
    assert R.range_proof_vec.verify_rangeproof_batch(prs, cms, 8, verifier_seed=b"\x01" * 32) == [True] * 5
    prs[3] = prs[3].copy(); prs[3][2, 100] ^= 1
    cms[1] = cms[1].copy(); cms[1][7] = cms[1][8]
    assert R.range_proof_vec.verify_rangeproof_batch(prs, cms, 8, verifier_seed=b"\x01" * 32) == [True, False, True, False, True]


def test_full_size_properties_cfg2(R):
    """BASELINE config 2 (d = 25 000, 32-bit, P = 4): the oracle would need minutes, so check the
    size-independent properties: determinism, create -> verify round trip, tamper rejection, commitment
    homomorphism against an independent small-kernel path, and oracle parity of a transcript prefix."""
    R.api.set_fp(32, 7)
    rng = np.random.default_rng(25000)
    d, nb, P = 25000, 32, 4
    vals, _ = _inputs(R, rng, d, nb, 32, 7)
    raw = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); raw[:, 31] &= 0x0F
    bl = raw
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.seeded(b"\x11" * 32))
    pr2, cm2 = R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.seeded(b"\x11" * 32))
    assert pr.shape == (4, 1440) and (pr == pr2).all() and (cm == cm2).all()
    assert R.range_proof_vec.verify_rangeproof(pr, cm, nb, verifier_seed=b"\x01" * 32)
    assert R.range_proof_vec.verify_rangeproof(pr, cm, nb, verifier_seed=b"\x02" * 32)
    bad = pr.copy(); bad[2, 9 * 32 + 1] ^= 1
    assert not R.range_proof_vec.verify_rangeproof(bad, cm, nb, verifier_seed=b"\x01" * 32)
    badc = cm.copy(); badc[24999] = cm[0]
    assert not R.range_proof_vec.verify_rangeproof(pr, badc, nb, verifier_seed=b"\x01" * 32)
    # commitments == commit_vec(f32_to_scalar(values), blindings) (independent kernel path), and vs the oracle on a sample
    sc = R.conversion32.f32_to_scalar_vec(vals)
    assert (R.pedersen_ops.commit_vec(sc, bl) == cm).all()
    idx = rng.choice(d, 64, replace=False)
    assert (orc.commit_vec(sc[idx], bl[idx]) == cm[idx]).all()
    # chunk 0 at full size against the oracle, bit for bit (its nonces start at index 0, so it equals the single-chunk proof
    # over the first 8192 values; ~8 s of CPU)
    rc, opr, ocm = orc.create_rangeproof(vals[:8192], bl[:8192], nb, 1, 32, 7, seed=b"\x11" * 32)
    assert rc == 0 and (opr[0] == pr[0]).all() and (ocm == cm[:8192]).all()
    # a different nonce seed changes the proof but not the commitments
    pr3, cm3 = R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.seeded(b"\x12" * 32))
    assert (cm3 == cm).all() and not (pr3 == pr).all()
    assert R.range_proof_vec.verify_rangeproof(pr3, cm3, nb, verifier_seed=b"\x01" * 32)
    R.api.set_fp(16, 7)


def test_full_size_properties_cfg2_e2e_partition(R):
    """The same workload at the reference's e2e partition count (n_partition = 64, cifar_large.yml:39-46): 64 chunks of
    m = 512 -- fixed-base tables shared by all chunks, device-side Horner for the 128 problems of a round, folds down to
    64 generators.  Chunk 0 is compared with the oracle bit for bit (its nonces start at index 0, so it equals a
    single-chunk proof over the first 512 values); the rest through the size-independent properties."""
    R.api.set_fp(32, 7)
    rng = np.random.default_rng(25064)
    d, nb, P = 25000, 32, 64
    vals, _ = _inputs(R, rng, d, nb, 32, 7)
    bl = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); bl[:, 31] &= 0x0F
    seed = b"\x21" * 32
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.seeded(seed))
    assert pr.shape == (64, 32 * (9 + 2 * 14))
    pr2, cm2 = R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.seeded(seed))
    assert (pr == pr2).all() and (cm == cm2).all()
    rc, opr, ocm = orc.create_rangeproof(vals[:512], bl[:512], nb, 1, 32, 7, seed=seed)
    assert rc == 0 and (opr[0] == pr[0]).all() and (ocm == cm[:512]).all()
    assert R.range_proof_vec.verify_rangeproof(pr, cm, nb, verifier_seed=b"\x01" * 32)
    assert orc.verify_rangeproof(pr[:1], cm[:512], nb, 32, 7) == (0, True)
    bad = pr.copy(); bad[37, 9 * 32 + 1] ^= 1
    assert not R.range_proof_vec.verify_rangeproof(bad, cm, nb, verifier_seed=b"\x01" * 32)
    # the commitments do not depend on the partition
    pr4, cm4 = R.range_proof_vec.create_rangeproof(vals, bl, nb, 4, nonce=R.Nonce.seeded(seed))
    assert (cm4 == cm).all()
    assert R.range_proof_vec.verify_rangeproof_batch([pr, pr2], [cm, cm2], nb, verifier_seed=b"\x03" * 32) == [True, True]
    R.api.set_fp(16, 7)


def test_quad_parallel_point_ops(R):
    """csrc/quad26.hpp (a point spread over four lanes, used by the device-side Horner chains): 2^k P + Q for random and special
    points, one quad per pair against one thread per pair, and both against the oracle's scalar arithmetic (P = a B, Q = b B)."""
    rng = np.random.default_rng(99)
    pairs = 200
    a = orc.rand_scalars(rng, pairs); b = orc.rand_scalars(rng, pairs)
    a[0] = 0; b[1] = 0; a[2] = 0; b[2] = 0                      # identity operands
    b[3] = a[3]                                                  # P == Q
    L = orc.L_ORDER
    b[4] = np.frombuffer(((L - int.from_bytes(a[4].tobytes(), "little") * 2 ** 7) % L).to_bytes(32, "little"), np.uint8)      # 2^7 P + Q = identity
    P = orc.commit_vec(a, None); Q = orc.commit_vec(b, None)
    inp = np.ascontiguousarray(np.stack([P, Q], axis=1).reshape(pairs, 64))
    for k in (0, 1, 7, 64):
        o1 = np.zeros((pairs, 32), np.uint8); o2 = np.zeros((pairs, 32), np.uint8)
        rc = R.lib().rofl_dbg_quad_ops(inp.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(pairs), ctypes.c_uint(k), o1.ctypes.data_as(ctypes.c_void_p), o2.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0
        want = np.stack([np.frombuffer(((int.from_bytes(a[i].tobytes(), "little") * 2 ** k + int.from_bytes(b[i].tobytes(), "little")) % L).to_bytes(32, "little"), np.uint8) for i in range(pairs)])
        ref = orc.commit_vec(want, None)
        assert (o1 == ref).all(), k
        assert (o2 == ref).all(), k


def _gpu_msm(R, k, p):
    out = np.zeros(32, np.uint8)
    rc = R.lib().rofl_dbg_msm(k.ctypes.data_as(ctypes.c_void_p), p.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(k.shape[0]), out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return out


@pytest.mark.parametrize("n", [1, 5, 63, 64, 300, 513, 2000, 6000, 8192, 9000])
def test_msm_extreme_scalars(R, n):
    """The Pippenger pipeline on inputs the proof path never produces: heavy bucket skew (slot overflow list /
    two-pass fallback), scalars in [2^252, l) (split top-window digit), zeros, small and all-ones scalars."""
    rng = np.random.default_rng(n)
    pts = orc.commit_vec(orc.rand_scalars(rng, n), None)            # n random valid points
    L = orc.L_ORDER
    def sc(v):
        return np.frombuffer((v % L).to_bytes(32, "little"), np.uint8)
    cases = {
        "random": orc.rand_scalars(rng, n),
        "all_equal": np.tile(sc(int.from_bytes(rng.integers(0, 256, 32, dtype=np.uint8).tobytes(), "little")), (n, 1)),
        "top_range": np.stack([sc(L - 1 - i) for i in range(n)]),                    # >= 2^252
        "two_pow_252": np.stack([sc((1 << 252) + i * 12345) for i in range(n)]),
        "small": np.stack([sc(i) for i in range(n)]),                                 # includes 0
        "minus_one": np.tile(sc(L - 1), (n, 1)),
    }
    for name, k in cases.items():
        k = np.ascontiguousarray(k)
        assert (_gpu_msm(R, k, pts) == orc.msm(k, pts)).all(), name


def test_cfg4_shape_batch_verify(R):
    """BASELINE config 4 shape on one rank: d = 55 000 (d_pad = 65 536, m = 16 384, N = 524 288), 32-bit, P = 4;
    three clients verified in one batched call; tampered members are singled out."""
    R.api.set_fp(32, 7)
    d, nb, P = 55000, 32, 4
    mx = np.float32(16777216.0)
    prs, cms = [], []
    for c in range(3):
        rng = np.random.default_rng(7000 + c)
        vals = np.clip(rng.uniform(-mx, mx, size=d).astype(np.float32), -mx, np.nextafter(mx, np.float32(0)))
        bl = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); bl[:, 31] &= 0x0F
        pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.seeded(bytes([c + 1]) * 32))
        assert pr.shape == (4, 1504)
        prs.append(pr); cms.append(cm)
    assert R.range_proof_vec.verify_rangeproof_batch(prs, cms, nb, verifier_seed=b"\x01" * 32) == [True, True, True]
    prs[1] = prs[1].copy(); prs[1][3, 77] ^= 1
    cms[2] = cms[2].copy(); cms[2][54999] = cms[2][0]
    assert R.range_proof_vec.verify_rangeproof_batch(prs, cms, nb, verifier_seed=b"\x02" * 32) == [True, False, False]
    R.api.set_fp(16, 7)


@pytest.mark.parametrize("kind,d,fb,ff", [(0, 1, 16, 7), (0, 70, 16, 7), (1, 5, 16, 7), (1, 130, 32, 7), (0, 300, 32, 12)])
def test_sigma_proofs_bit_exact(R, kind, d, fb, ff):
    """rand_proof_vec / square_rand_proof_vec: proofs, commitments and verify bits vs the oracle (device Merlin)."""
    R.api.set_fp(fb, ff)
    rng = np.random.default_rng(100 * d + kind)
    vals = rng.uniform(-3, 3, size=d).astype(np.float32)
    vals[0] = -1.5
    r1, r2 = orc.rand_scalars(rng, d), orc.rand_scalars(rng, d)
    seed = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
    if kind == 0:
        pr, cm = R.rand_proof_vec.create_randproof_vec(vals, r1, nonce=R.Nonce.seeded(seed))
        ver = R.rand_proof_vec.verify_randproof_vec
    else:
        pr, cm = R.square_rand_proof_vec.create_l2rangeproof_vec(vals, r1, r2, nonce=R.Nonce.seeded(seed))
        ver = R.square_rand_proof_vec.verify_l2rangeproof_vec
    rc, opr, ocm = orc.sigma_create(kind, vals, r1, r2 if kind else None, fb, ff, seed=seed)
    assert rc == 0 and (cm == ocm).all() and (pr == opr).all()
    assert ver(pr, cm) is True and orc.sigma_verify(kind, pr, cm) == (0, True)
    # the ElGamal L part is the Pedersen commitment of the range proof path (commit_vec)
    assert (cm[:, :32] == R.pedersen_ops.commit_vec(R.conversion32.f32_to_scalar_vec(vals), r1)).all()
    # prove_existing with those commitments reproduces the same proof
    if kind == 0:
        pr2, cm2 = R.rand_proof_vec.create_randproof_vec_existing(vals, cm[:, :32], r1, nonce=R.Nonce.seeded(seed))
    else:
        pr2, cm2 = R.square_rand_proof_vec.create_l2rangeproof_vec_existing(vals, cm[:, :32], r1, r2, nonce=R.Nonce.seeded(seed))
    assert (pr2 == pr).all() and (cm2 == cm).all()
    # tamper: response scalar, prime commitment, real commitment
    for off in ((64 if kind == 0 else 96) + 3, 5):
        bad = pr.copy(); bad[d // 2, off] ^= 1
        try:
            res = ver(bad, cm)
        except R.RoflError as e:
            assert e.code == 5 and orc.sigma_verify(kind, bad, cm)[0] == 5      # flipped into an invalid encoding
        else:
            assert res is False and orc.sigma_verify(kind, bad, cm) == (0, False)
    badc = cm.copy(); badc[0, :32] = cm[-1, :32] if d > 1 else R.pedersen_ops.commit_no_blinding_vec(r1)[0]
    if d > 1 and not (cm[0, :32] == cm[-1, :32]).all():
        assert ver(pr, badc) is False
    # non-canonical response scalar -> FormatError
    bad = pr.copy(); bad[0, (64 if kind == 0 else 96):(96 if kind == 0 else 128)] = 0xFF
    with pytest.raises(R.RoflError) as e:
        ver(bad, cm)
    assert e.value.code == 5
    R.api.set_fp(16, 7)


@pytest.mark.parametrize("kind", [1, 2])
def test_square_proofs_with_commitments_that_are_not_the_values(R, kind):
    """square_rand_proof/party.rs (prove_existing): c_sq' = m' * L of the commitment HANDED IN.  The HIP path computes c_sq' from the opening it knows
    when the handed-in bytes ARE m B + r1 Bb, and the reference's way (decode, variable-base multiplication) for every element where they are not:
    the proof bytes equal the oracle's for consistent, inconsistent (another value's, another blinding's, the identity) and mixed inputs, and an
    invalid encoding is the reference's FormatError."""
    R.api.set_fp(16, 7)
    rng = np.random.default_rng(4242 + kind)
    d = 300
    vals = rng.uniform(-3, 3, size=d).astype(np.float32)
    r1, r2 = orc.rand_scalars(rng, d), orc.rand_scalars(rng, d)
    seed = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
    M = R.square_rand_proof_vec if kind == 1 else R.square_proof_vec
    good = R.pedersen_ops.commit_vec(R.conversion32.f32_to_scalar_vec(vals), r1)
    ex = good.copy()
    ex[3] = good[4]                                                                   # another element's commitment
    ex[70] = R.pedersen_ops.commit_vec(R.conversion32.f32_to_scalar_vec(vals[70:71]), r2[70:71])[0]   # the value under another blinding
    ex[64] = 0                                                                        # the identity
    ex[255:270] = good[100:115]                                                       # a run of them across a 64-thread block boundary
    ex[d - 1] = R.pedersen_ops.commit_no_blinding_vec(r1[:1])[0]
    for name, e_ in (("consistent", good), ("mixed", ex), ("all wrong", np.roll(good, 1, axis=0))):
        rc, opr, ocm = orc.sigma_create(kind, vals, r1, r2, 16, 7, seed=seed, existing=e_)
        pr, cm = M.create_l2rangeproof_vec_existing(vals, e_, r1, r2, nonce=R.Nonce.seeded(seed))
        assert rc == 0 and (pr == opr).all() and (cm == ocm).all(), name
        assert M.verify_l2rangeproof_vec(pr, cm) is (name == "consistent") and orc.sigma_verify(kind, pr, cm) == (0, name == "consistent")
    # the runs of elements of a split client see the same marks (ranges that start and end inside the run of wrong commitments)
    if kind == 1:
        rc, opr, ocm = orc.sigma_create(kind, vals, r1, r2, 16, 7, seed=seed, existing=ex)
        for a, b in ((0, 66), (66, 260), (260, d)):
            p_, c_ = R.api.create_sigmaproof_vec_range(kind, vals, r1, r2, a, b - a, nonce=R.Nonce.seeded(seed), existing=ex)
            assert (p_ == opr[a:b]).all() and (c_ == ocm[a:b]).all(), (a, b)
    bad = good.copy(); bad[17] = 0xFF                                                  # not a canonical field element
    rc, _, _ = orc.sigma_create(kind, vals, r1, r2, 16, 7, seed=seed, existing=bad)
    with pytest.raises(R.RoflError) as e:
        M.create_l2rangeproof_vec_existing(vals, bad, r1, r2, nonce=R.Nonce.seeded(seed))
    assert rc == e.value.code == 5
    R.api.set_fp(16, 7)


def test_sigma_reference_semantics(R):
    """l2_range_proof_vec/mod.rs:539-561: sum of the c_sq of the square proofs == the L2 range-proof commitment;
    square_rand_proof_vec/mod.rs:195-208: existing round trip; wrong lengths -> WrongNumBlindingFactors."""
    R.api.set_fp(16, 7)
    rng = np.random.default_rng(77)
    vals = np.array([1.25, -0.5, 0.25], np.float32)
    zero = np.zeros((3, 32), np.uint8)
    pr, cm = R.square_rand_proof_vec.create_l2rangeproof_vec(vals, zero, zero, nonce=R.Nonce.seeded(b"\x01" * 32))
    _, l2c = R.l2_range_proof_vec.create_rangeproof_l2(vals, zero, 16, 4, nonce=R.Nonce.seeded(b"\x02" * 32))
    s = R.pedersen_ops.add_rp_vec(R.pedersen_ops.add_rp_vec(cm[0:1, 64:], cm[1:2, 64:]), cm[2:3, 64:])
    assert (s[0] == l2c).all()
    with pytest.raises(R.RoflError) as e:
        R.rand_proof_vec.create_randproof_vec(vals, orc.rand_scalars(rng, 2))
    assert e.value.code == 1
    # explicit nonce stream
    stream = rng.integers(0, 256, 3 * 3 * 64, dtype=np.uint8).tobytes()
    r1, r2 = orc.rand_scalars(rng, 3), orc.rand_scalars(rng, 3)
    pr, cm = R.square_rand_proof_vec.create_l2rangeproof_vec(vals, r1, r2, nonce=R.Nonce.stream(stream))
    rc, opr, ocm = orc.sigma_create(1, vals, r1, r2, 16, 7, stream=stream)
    assert rc == 0 and (pr == opr).all() and (cm == ocm).all()


def test_sigma_full_size_cfg3(R):
    """BASELINE config 3 (L2, d = 25 000): per-element square proofs + the sum proof; properties at full size,
    oracle parity on a sample of elements (each element is an independent proof)."""
    R.api.set_fp(32, 7)
    rng = np.random.default_rng(3)
    d = 25000
    vals = (rng.integers(-3, 4, size=d) / 128.0).astype(np.float32)
    r1 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r1[:, 31] &= 0x0F
    r2 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r2[:, 31] &= 0x0F
    seed = b"\x21" * 32
    pr, cm = R.square_rand_proof_vec.create_l2rangeproof_vec(vals, r1, r2, nonce=R.Nonce.seeded(seed))
    assert R.square_rand_proof_vec.verify_l2rangeproof_vec(pr, cm) is True
    idx = rng.choice(d, 16, replace=False)
    for i in idx:      # element i uses nonces 3i..3i+2: reproduce with an explicit 3-scalar stream from the oracle's DRBG
        ns = orc._nonce(seed=seed)
        raw = b""
        for j in range(3):
            out = np.zeros(32, np.uint8)
            orc.lib().orc_nonce_scalar(ctypes.byref(ns), ctypes.c_uint64(3 * int(i) + j), out.ctypes.data_as(ctypes.c_void_p))
            raw += out.tobytes() + bytes(32)
        rc, opr, ocm = orc.sigma_create(1, vals[i:i + 1], r1[i:i + 1], r2[i:i + 1], 32, 7, stream=raw)
        assert rc == 0 and (opr[0] == pr[i]).all() and (ocm[0] == cm[i]).all()
    bad = pr.copy(); bad[12345, 100] ^= 1
    try:
        assert R.square_rand_proof_vec.verify_l2rangeproof_vec(bad, cm) is False
    except R.RoflError as e:
        assert e.code == 5
    R.api.set_fp(16, 7)


def test_random_shape_fuzz(R):
    """Time-boxed randomised parity (tests/gpu_fuzz.py): random (d, bits, P, fp) incl. error codes and bit-flip tampering."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_fuzz.py"), "20", "4242"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "fuzz ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("d,fb,ff", [(1, 16, 7), (90, 16, 7), (257, 32, 7)])
def test_square_proof_bit_exact(R, d, fb, ff):
    """square_proof_vec (Pedersen-only square proof, the *Compressed enc types' per-element part)."""
    R.api.set_fp(fb, ff)
    rng = np.random.default_rng(d)
    vals = rng.uniform(-3, 3, size=d).astype(np.float32)
    r1, r2 = orc.rand_scalars(rng, d), orc.rand_scalars(rng, d)
    seed = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
    pr, cm = R.square_proof_vec.create_l2rangeproof_vec(vals, r1, r2, nonce=R.Nonce.seeded(seed))
    rc, opr, ocm = orc.sigma_create(2, vals, r1, r2, fb, ff, seed=seed)
    assert rc == 0 and (pr == opr).all() and (cm == ocm).all()
    assert R.square_proof_vec.verify_l2rangeproof_vec(pr, cm) and orc.sigma_verify(2, pr, cm) == (0, True)
    pr2, cm2 = R.square_proof_vec.create_l2rangeproof_vec_existing(vals, cm[:, :32], r1, r2, nonce=R.Nonce.seeded(seed))
    assert (pr2 == pr).all() and (cm2 == cm).all()
    bad = pr.copy(); bad[d // 2, 64 + 7] ^= 1          # Z_m
    assert R.square_proof_vec.verify_l2rangeproof_vec(bad, cm) is False and orc.sigma_verify(2, bad, cm) == (0, False)
    # merge() of params.rs:776-788: ElGamal pairs of the compressed proof + square commitments share c_l
    _, pairs = R.compressed_rand_proof.helper_prove(vals, r1, nonce=R.Nonce.seeded(seed))
    assert (pairs[:, :32] == cm[:, :32]).all()
    R.api.set_fp(16, 7)


@pytest.mark.parametrize("d,fb,ff", [(1, 16, 7), (3, 16, 7), (100, 16, 7), (700, 32, 7), (5000, 32, 7)])
def test_compressed_rand_proof(R, d, fb, ff):
    """compressed_rand_proof: one proof for d ElGamal pairs; verification = two d-term MSMs sharing challenge powers."""
    R.api.set_fp(fb, ff)
    rng = np.random.default_rng(31 * d)
    vals = rng.uniform(-3, 3, size=d).astype(np.float32)
    r = orc.rand_scalars(rng, d)
    seed = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
    pf, pairs = R.compressed_rand_proof.helper_prove(vals, r, nonce=R.Nonce.seeded(seed))
    rc, opf, opairs = orc.compressed_create(vals, r, fb, ff, seed=seed)
    assert rc == 0 and (pairs == opairs).all() and (pf == opf).all()
    assert R.compressed_rand_proof.helper_verify(pf, pairs) is True and orc.compressed_verify(pf, pairs) == (0, True)
    pf2, pairs2 = R.compressed_rand_proof.helper_prove_existing(vals, pairs[:, :32], r, nonce=R.Nonce.seeded(seed))
    assert (pf2 == pf).all() and (pairs2 == pairs).all()
    bad = pf.copy(); bad[64 + 5] ^= 1
    assert R.compressed_rand_proof.helper_verify(bad, pairs) is False
    if d > 1:
        badp = pairs.copy(); badp[0], badp[1] = pairs[1], pairs[0]      # order matters: powers of the challenge
        if not (pairs[0] == pairs[1]).all():
            assert R.compressed_rand_proof.helper_verify(pf, badp) is False and orc.compressed_verify(pf, badp) == (0, False)
    bad = pf.copy(); bad[96:128] = 0xFF
    with pytest.raises(R.RoflError) as e:
        R.compressed_rand_proof.helper_verify(bad, pairs)
    assert e.value.code == 5
    stream = rng.integers(0, 256, 2 * 64, dtype=np.uint8).tobytes()
    pf3, _ = R.compressed_rand_proof.helper_prove(vals, r, nonce=R.Nonce.stream(stream))
    rc, opf3, _ = orc.compressed_create(vals, r, fb, ff, stream=stream)
    assert (pf3 == opf3).all()
    R.api.set_fp(16, 7)


def test_bsgs_discrete_log(R):
    """bsgs32.rs tests (:89-125) + the reference's dlog-based semantic tests (range_proof_vec/mod.rs:318-332, 369-399) end to end:
    create_rangeproof with zero / cancelling blindings -> add_rp_vec_vec -> default_discrete_log_vec -> scalar_to_f32."""
    R.api.set_fp(16, 7)
    rng = np.random.default_rng(5)
    fx = (rng.integers(-65535, 65536, size=400)).astype(np.float64) / 128.0
    vals = fx.astype(np.float32)
    sc = R.conversion32.f32_to_scalar_vec(vals)
    pts = R.pedersen_ops.commit_no_blinding_vec(sc)
    for m in (1 << 15, 1 << 16, 1 << 10, 3000):
        # a non-dividing table size covers only (2^16 / m) * m + m values: keep the inputs inside it
        sel = np.abs(fx * 128) <= (65536 // m) * m
        got = R.pedersen_ops.discrete_log_vec(pts[sel], m, 16)
        rc, exp = orc.bsgs_solve(pts[sel], m, 16)
        assert rc == 0 and (got == exp).all() and (got == sc[sel]).all(), m
    with pytest.raises(R.RoflError):        # 65428 > 21 * 3000 + 3000: the reference unwraps None, so does the oracle
        R.pedersen_ops.discrete_log_vec(pts, 3000, 16)
    assert orc.bsgs_solve(pts, 3000, 16)[0] == 11
    assert (R.pedersen_ops.default_discrete_log_vec(pts) == sc).all()
    assert list(R.conversion32.scalar_to_f32_vec(R.pedersen_ops.default_discrete_log_vec(pts))) == list(vals)
    # test_create_rangeproof_correct_shift
    x = np.array([0.25, 1.25, -1.5], np.float32)
    _, cm = R.range_proof_vec.create_rangeproof(x, np.zeros((3, 32), np.uint8), 16, 4, nonce=R.Nonce.seeded(b"\x01" * 32))
    assert list(R.conversion32.scalar_to_f32_vec(R.pedersen_ops.default_discrete_log_vec(cm))) == [0.25, 1.25, -1.5]
    # test_rangeproof_with_cancelling_blindings
    b0, b1 = orc.rand_scalars(rng, 3), orc.rand_scalars(rng, 3)
    b2 = np.stack([np.frombuffer(((-(int.from_bytes(b0[i].tobytes(), "little") + int.from_bytes(b1[i].tobytes(), "little"))) % orc.L_ORDER).to_bytes(32, "little"), np.uint8) for i in range(3)])
    cms = [R.range_proof_vec.create_rangeproof(v, b, 16, 4, nonce=R.Nonce.seeded(b"\x02" * 32))[1]
           for v, b in zip(([0.25, 1.25, -1.5], [-0.75, 1.25, -2.0], [0.5, 1.25, -3.0]), (b0, b1, b2))]
    tot = R.pedersen_ops.add_rp_vec_vec(cms)
    assert list(R.conversion32.scalar_to_f32_vec(R.pedersen_ops.default_discrete_log_vec(tot))) == [0.0, 3.75, -6.5]
    # out of table range -> the reference panics (unwrap on None)
    far = R.pedersen_ops.commit_no_blinding_vec(np.frombuffer((1 << 40).to_bytes(32, "little"), np.uint8).reshape(1, 32))
    with pytest.raises(R.RoflError) as e:
        R.pedersen_ops.discrete_log_vec(far, 1 << 10, 16)
    assert e.value.code == 11 and orc.bsgs_solve(far, 1 << 10, 16)[0] == 11
    # fp8 flavour: 8-bit values, table 2^(4+3)
    R.api.set_fp(8, 3)
    v8 = np.array([0.0, 1.5, -2.25, 15.875, -15.875], np.float32)
    p8 = R.pedersen_ops.commit_no_blinding_vec(R.conversion32.f32_to_scalar_vec(v8))
    assert list(R.conversion32.scalar_to_f32_vec(R.pedersen_ops.default_discrete_log_vec(p8))) == list(v8)
    R.api.set_fp(16, 7)


@pytest.mark.parametrize("env", [
    {"ROFL_MSM_FB_MIN": "64", "ROFL_MSM_LDS_MIN": "32"},                 # fixed-base tables + LDS scatter at every size
    {"ROFL_MSM_FB": "0", "ROFL_MSM_LDS_MIN": "32"},                      # generic windows through the LDS scatter
    {"ROFL_MSM_FB_MIN": "64", "ROFL_MSM_LDS": "0"},                      # fixed-base through the per-item slot scatter
    {"ROFL_MSM_SLOTS": "0", "ROFL_MSM_LR": "0"},                         # two-pass counting sort, separate L / R arrays
    {"ROFL_VERIFY_BATCH": "0", "ROFL_FOLD_PB": "64", "ROFL_FOLD_W": "4", "ROFL_LANES": "1"},
    {"ROFL_FOLD_PB": "16", "ROFL_FOLD_W": "5", "ROFL_FOLD_T1": "2", "ROFL_FOLD_MIN": "16"},
    {"ROFL_GENS_BUDGET_MB": "1", "ROFL_LANES": "2"},                      # every new (n, m) evicts the previous tables
    {"ROFL_FOLD_WNAF": "4", "ROFL_FOLD_MIN": "16"},                        # later folds over odd multiples of their sources (event-list kernel, side-stream table build)
    {"ROFL_FOLD_WNAF": "5", "ROFL_FOLD_T": "1", "ROFL_FOLD_MIN": "16"},
    {"ROFL_FOLD_TAB_EV": "0"},                                            # the first fold through the digit-scanning kernel (k_fold_gens_tab) instead of the event list
    {"ROFL_GENS_RESERVE_MB": "400000"},                                   # no big table may be allocated (reserve > HBM): fold tables of 4 GB and more are narrowed until they fit
    {"ROFL_MSM_DEV_HORNER_MIN": "1", "ROFL_MSM_FB": "0"},                 # every MSM finishes its Horner chains on the device
    {"ROFL_MSM_DEV_HORNER_MIN": "1", "ROFL_MSM_FB_MIN": "64"},
    {"ROFL_MSM_SMALL_MAX": "0"},                                          # the general pipeline at the sizes the fused small-MSM launch normally takes
])
def test_msm_variants_small_sizes(R, env):
    import subprocess, sys
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "gpu_variant_check.py")], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FB_SMALL PASS" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_two_level_sort_bin_shapes(R, full_oracle, tmp_path):
    """The two-level bucket sort of the fixed-base launches at its three shapes -- 4 windows per bucket array (256 coarse bins, the
    single-client default), 8 and 16 windows per array (512 bins; what calls in flight next to others and single-set plans take) --
    and the one-level LDS scatter: all bit-identical to the oracle on one chunk of 2^19 terms (chunk 0 of the session's full-size cfg-4
    oracle proof = the single-chunk proof over its first 16 384 values: its nonces start at index 0)."""
    import hashlib, subprocess, sys
    helper = os.path.join(os.path.dirname(__file__), "gpu_two_level_check.py")
    c = full_oracle.case("cfg4"); m = 16384
    npz = str(tmp_path / "chunk0.npz")
    np.savez(npz, vals=c["vals"][:m], bl=c["bl"][:m], seed=np.frombuffer(c["seed"], np.uint8))
    want = hashlib.sha256(c["opr"][:1].tobytes() + c["ocm"][:m].tobytes()).hexdigest()
    def run(env):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, helper, npz], env=e, capture_output=True, text=True, timeout=900)
        lines = [l for l in r.stdout.splitlines() if l.startswith("DIGEST")]
        assert r.returncode == 0 and lines, r.stdout[-2000:] + r.stderr[-2000:]
        return lines[0].split()[1:]
    for env in ({}, {"ROFL_MSM_FB_THREADS": "65536"}, {"ROFL_MSM_FB_THREADS": "32768"},      # two sets / one set per problem: 8 and 16 windows per bucket array
                {"ROFL_MSM_TWO_LEVEL": "0"}, {"ROFL_ACC_BALANCE": "0"}):
        got = run(env)
        assert got == [want, "1", "0"], (env, got)


def test_concurrent_calls_use_separate_lanes(R):
    """The reference's server calls verify from a thread pool (server.rs:656-687): concurrent calls must not disturb each
    other.  Six threads, different shapes, every proof bit-exact vs the oracle; timing and errors are per thread."""
    from concurrent.futures import ThreadPoolExecutor
    shapes = [(300, 8, 4, 16, 7), (1000, 32, 4, 32, 7), (37, 16, 8, 16, 7), (2000, 8, 2, 16, 7), (64, 64, 1, 64, 7), (500, 16, 4, 16, 7)]
    inputs = []
    for i, (d, nb, P, fb, ff) in enumerate(shapes):
        rng = np.random.default_rng(900 + i)
        R.api.set_fp(fb, ff)
        mn, mx = R.conversion32.get_clip_bounds(nb)
        vals = np.clip(rng.uniform(mn, mx, size=d).astype(np.float32), mn, np.nextafter(np.float32(mx), np.float32(0)))
        inputs.append((vals, orc.rand_scalars(rng, d), bytes(rng.integers(0, 256, 32, dtype=np.uint8))))
    lib = R.lib()

    def work(i):
        d, nb, P, fb, ff = shapes[i]
        vals, bl, seed = inputs[i]
        out = []
        for rep in range(3):
            # fp config is module state in the Python mirror: go through the C ABI directly with explicit (fp_bits, fp_frac)
            ns = R.Nonce.seeded(seed)._struct()
            npr = lib.rofl_rangeproof_chunks(ctypes.c_size_t(d), ctypes.c_size_t(P)); plen = lib.rofl_rangeproof_size(ctypes.c_size_t(nb), ctypes.c_size_t(d), ctypes.c_size_t(P))
            pr = np.zeros((npr, plen), np.uint8); cm = np.zeros((d, 32), np.uint8)
            a, b = ctypes.c_size_t(), ctypes.c_size_t()
            rc = lib.rofl_create_rangeproof(vals.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(d), bl.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(d),
                                            ctypes.c_size_t(nb), ctypes.c_size_t(P), fb, ff, ctypes.byref(ns), pr.ctypes.data_as(ctypes.c_void_p),
                                            ctypes.byref(a), ctypes.byref(b), cm.ctypes.data_as(ctypes.c_void_p))
            ok = ctypes.c_int()
            rc2 = lib.rofl_verify_rangeproof(pr.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(plen), ctypes.c_size_t(npr), cm.ctypes.data_as(ctypes.c_void_p),
                                             ctypes.c_size_t(d), ctypes.c_size_t(nb), fb, ff, (ctypes.c_uint8 * 32)(*([rep + 1] * 32)), ctypes.byref(ok))
            out.append((rc, rc2, ok.value, pr, cm))
        return out
    lib.rofl_rangeproof_chunks.restype = ctypes.c_size_t; lib.rofl_rangeproof_size.restype = ctypes.c_size_t
    with ThreadPoolExecutor(max_workers=6) as ex:
        res = list(ex.map(work, range(len(shapes))))
    for i, (d, nb, P, fb, ff) in enumerate(shapes):
        vals, bl, seed = inputs[i]
        rc, opr, ocm = orc.create_rangeproof(vals, bl, nb, P, fb, ff, seed=seed)
        assert rc == 0
        for (rc1, rc2, ok, pr, cm) in res[i]:
            assert rc1 == 0 and rc2 == 0 and ok == 1 and (pr == opr).all() and (cm == ocm).all(), shapes[i]
    R.api.set_fp(16, 7)


def test_device_resident_inputs(R):
    """values / blindings / commitments handed over as device pointers (torch CUDA tensors) give the same bytes.
    (Own process: torch has to bring up its HIP runtime before the library's is loaded, as in bench.py.)"""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "gpu_device_inputs_check.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DEVICE_INPUTS PASS" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
