"""The BASELINE configs that round 1 left without an oracle comparison ("configs not exercised in -m gpu"):
cfg 1 at full size (whole proof), cfg 4 chunk 0 at m = 16 384, and the cfg 3 / cfg 5 composites through
EncParamsL2.encrypt / verify (rofl_service/src/flserver/params.rs:608-646, 206-234) at d = 25 000 and d = 55 000 with chunk-0
parity of the 8-bit L-inf leg, the L2 sum proof bit for bit and sampled per-element square proofs.  Bit-exact: integer work."""
import ctypes
import hashlib

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    import rofl_project_code_amd as R
    from rofl_project_code_amd import build
    build.build()
    R.set_device(0)
    return R


def _uniform(R, rng, d, nb, fp):
    mn, mx = R.conversion32.get_clip_bounds(nb, fp=fp)
    vals = rng.uniform(mn, mx, size=d).astype(np.float32)
    return np.clip(vals, mn, np.nextafter(np.float32(mx), np.float32(0)))


def test_cfg1_full_size_whole_proof_vs_oracle(R):
    """BASELINE cfg 1: L-inf 8-bit, d = 5 000 (mnist_dev_intrinsic_5k), P = 4, fp16 / frac7 -- every chunk, bit for bit."""
    fp = (16, 7)
    rng = np.random.default_rng(5000)
    d, nb, P = 5000, 8, 4
    vals = _uniform(R, rng, d, nb, fp)
    bl = orc.rand_scalars(rng, d)
    seed = b"\x31" * 32
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.seeded(seed), fp=fp)
    assert pr.shape == (4, 1184)
    rc, opr, ocm = orc.create_rangeproof(vals, bl, nb, P, 16, 7, seed=seed)
    assert rc == 0 and (ocm == cm).all() and (opr == pr).all()
    assert R.range_proof_vec.verify_rangeproof(pr, cm, nb, verifier_seed=b"\x01" * 32, fp=fp)
    assert orc.verify_rangeproof(pr, cm, nb, 16, 7) == (0, True)
    bad = pr.copy(); bad[3, 700] ^= 2
    assert not R.range_proof_vec.verify_rangeproof(bad, cm, nb, verifier_seed=b"\x01" * 32, fp=fp)
    assert orc.verify_rangeproof(bad, cm, nb, 16, 7) == (0, False)


def test_cfg4_chunk0_vs_oracle(R, full_oracle):
    """BASELINE cfg 4 shape: d = 55 000 (d_pad 65 536, m = 16 384, N = 524 288, 19 rounds), 32-bit, P = 4.  Chunk 0's nonces
    start at index 0, so the oracle's chunk 0 of the four-chunk proof equals the single-chunk proof (P = 1) over the first 16 384 values:
    the HIP path proves exactly that and must return the same bytes; the oracle accepts the GPU's chunk and the GPU the oracle's."""
    c = full_oracle.case("cfg4")
    fp, nb, m = c["fp"], c["nb"], 16384
    pr, cm = R.range_proof_vec.create_rangeproof(c["vals"][:m], c["bl"][:m], nb, 1, nonce=R.Nonce.seeded(c["seed"]), fp=fp)
    assert pr.shape == (1, 1504)
    assert (c["ocm"][:m] == cm).all() and (c["opr"][0] == pr[0]).all()
    assert orc.verify_rangeproof(pr, cm, nb, 32, 7) == (0, True)
    assert R.range_proof_vec.verify_rangeproof(c["opr"][:1], c["ocm"][:m], nb, verifier_seed=b"\x02" * 32, fp=fp)


def test_cfg2_whole_proof_full_size_vs_oracle(R, full_oracle):
    """BASELINE cfg 2 (the headline): d = 25 000, 32-bit, P = 4 -- ALL four chunks and all commitments against the oracle, bit for
    bit (chunks 1-3 used to be covered by round trip / tamper only).  The oracle proves the chunks on four threads
    (once per session: conftest.full_oracle)."""
    c = full_oracle.case("cfg2")
    pr, cm = R.range_proof_vec.create_rangeproof(c["vals"], c["bl"], c["nb"], c["P"], nonce=R.Nonce.seeded(c["seed"]), fp=c["fp"])
    opr, ocm = c["opr"], c["ocm"]
    assert opr.shape == pr.shape == (4, 1440)
    for k in range(4):
        assert (opr[k] == pr[k]).all(), "chunk %d differs from the oracle" % k
    assert (ocm == cm).all()
    assert orc.verify_rangeproof(pr, cm, c["nb"], 32, 7) == (0, True)


def test_cfg4_whole_proof_full_size_vs_oracle(R, full_oracle):
    """BASELINE cfg 4 shape, one client: d = 55 000 (m = 16 384 per chunk, N = 2^19), 32-bit, P = 4 -- all four chunks bit for bit."""
    c = full_oracle.case("cfg4")
    pr, cm = R.range_proof_vec.create_rangeproof(c["vals"], c["bl"], c["nb"], c["P"], nonce=R.Nonce.seeded(c["seed"]), fp=c["fp"])
    opr, ocm = c["opr"], c["ocm"]
    assert opr.shape == pr.shape == (4, 1504)
    for k in range(4):
        assert (opr[k] == pr[k]).all(), "chunk %d differs from the oracle" % k
    assert (ocm == cm).all()


@pytest.mark.parametrize("d", [25000, 55000])
def test_e2e_partition_chunks_vs_oracle(R, d):
    """n_partition = 64 (cifar_large.yml:39-46) at d = 25 000 and d = 55 000: chunks 0, 31 and 63 against the oracle proving exactly that
    chunk from its own position in the client's nonce space (orc.prove_chunk), the last one with its padding (shifted value 0,
    blinding 0: range_proof_vec/mod.rs:45-50)."""
    fp = (32, 7)
    rng = np.random.default_rng(640000 + d)
    nb, P = 32, 64
    vals = _uniform(R, rng, d, nb, fp)
    bl = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); bl[:, 31] &= 0x0F
    seed = b"\x53" * 32
    pr, cm = R.range_proof_vec.create_rangeproof(vals, bl, nb, P, nonce=R.Nonce.seeded(seed), fp=fp)
    dp = 1 << (d - 1).bit_length(); m = dp // P
    assert pr.shape[0] == 64
    vp = np.zeros(dp, np.float32); vp[:d] = vals
    bp = np.zeros((dp, 32), np.uint8); bp[:d] = bl
    # 2^(n-1) B, the shift between the API's commitments C_j and the V_j the proof is about (range_proof_vec/mod.rs:96-99, 155-167)
    off_pt = R.pedersen_ops.commit_vec(np.frombuffer((1 << (nb - 1)).to_bytes(32, "little"), np.uint8).reshape(1, 32), np.zeros((1, 32), np.uint8))[0]
    for c in (0, 31, 63):
        lo, hi = c * m, min((c + 1) * m, d)
        rc, oproof, oV = orc.prove_chunk(vp[c * m:(c + 1) * m], bp[c * m:(c + 1) * m], nb, c, 7, seed, n_real=max(hi - lo, 0))
        assert rc == 0 and (oproof == pr[c]).all(), "chunk %d differs from the oracle" % c
        if hi > lo:      # the oracle's V_j of the real values are the returned commitments shifted up
            assert (R.pedersen_ops.compute_shifted_values_rp(cm[lo:hi], off_pt) == oV[:hi - lo]).all()
    assert R.range_proof_vec.verify_rangeproof(pr, cm, nb, verifier_seed=b"\x01" * 32, fp=fp)


def _sub(seed, tag, *witness):
    from rofl_project_code_amd.params import witness_digest      # XXH3-128 / BLAKE2b over the raw witness bytes
    return hashlib.sha3_256(b"rofl-zk/params/v2" + seed + tag + witness_digest(*witness)).digest()


@pytest.mark.parametrize("d", [25000, 55000])
def test_cfg3_cfg5_l2_composite_vs_oracle(R, d):
    """BASELINE cfg 3 (d = 25 000) / cfg 5 (d = 55 000) per client: EncParamsL2::encrypt = 8-bit L-inf range proofs (value_range 8,
    P = 4) + L2 sum proof (l2_value_range 32) + per-element SquareRandProofs, fp32 / frac7 (cifar_large.yml:41-43, 99-102)."""
    fp = (32, 7)
    rng = np.random.default_rng(d)
    nb, P, l2n = 8, 4, 32
    x = (rng.integers(-3, 4, size=d) / 128.0).astype(np.float32)          # on the quantisation grid, inside the 8-bit range
    bl = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); bl[:, 31] &= 0x0F
    r2 = rng.integers(0, 256, size=(d, 32), dtype=np.uint8); r2[:, 31] &= 0x0F
    seed = b"\x51" * 32
    enc = R.EncParamsL2.encrypt(x, bl, nb, P, l2n, nonce_seed=seed, rand_scalars=r2, fp=fp)
    dp = 1 << (d - 1).bit_length(); m = dp // P
    assert enc.range_proofs.shape == (P, 32 * (9 + 2 * ((nb * m).bit_length() - 1)))
    # 8-bit L-inf leg: chunk 0 bit for bit
    rc, opr, ocm = orc.create_rangeproof(x[:m], bl[:m], nb, 1, 32, 7, seed=_sub(seed, b"range", x, bl, r2))
    assert rc == 0 and (opr[0] == enc.range_proofs[0]).all() and (ocm == enc.enc_values[:m, :32]).all()
    # the L2 sum proof (one 32-bit chunk over sum x^2) bit for bit, with its commitment = sum of the square commitments
    rc, ol2, ol2c = orc.create_rangeproof_l2(x, r2, l2n, P, 32, 7, seed=_sub(seed, b"l2", x, bl, r2))
    assert rc == 0 and (enc.square_range_proof == ol2.reshape(-1)).all()
    assert (R.pedersen_ops.sum_rp_vec(enc.enc_values[:, 64:96]) == ol2c.reshape(-1)).all()
    # sampled square proofs: element i uses nonces 3i..3i+2 of the "sq" stream
    sq_seed = _sub(seed, b"sq", x, bl, r2)
    for i in rng.choice(d, 12, replace=False):
        ns = orc._nonce(seed=sq_seed)
        raw = b""
        for j in range(3):
            out = np.zeros(32, np.uint8)
            orc.lib().orc_nonce_scalar(ctypes.byref(ns), ctypes.c_uint64(3 * int(i) + j), out.ctypes.data_as(ctypes.c_void_p))
            raw += out.tobytes() + bytes(32)
        rc, osq, osqc = orc.sigma_create(1, x[i:i + 1], bl[i:i + 1], r2[i:i + 1], 32, 7, stream=raw, existing=enc.enc_values[i:i + 1, :32].copy())
        assert rc == 0 and (osq[0] == enc.square_proofs[i]).all() and (osqc[0] == enc.enc_values[i]).all()
    # container round trip + verification (params.rs:206-234), tampering in each leg
    back = R.EncParamsL2.deserialize(enc.serialize())
    assert back.verify(verifier_seed=b"\x05" * 32, fp=fp)
    assert orc.verify_rangeproof(back.range_proofs[:1], back.enc_values[:m, :32].copy(), nb, 32, 7) == (0, True)
    assert orc.verify_rangeproof_l2(back.square_range_proof, ol2c, l2n, 32, 7) == (0, True)
    for field, pos in (("range_proofs", (P - 1, 300)), ("square_proofs", (d - 1, 9)), ("square_range_proof", (77,))):
        t = R.EncParamsL2.deserialize(enc.serialize()); getattr(t, field)[pos] ^= 1
        assert not t.verify(fp=fp), field


def test_many_chunk_shapes_fuzz(R):
    """Random shapes with 16 .. 128 chunks of 2^11 .. 2^15 generators (the regime of the reference's e2e runs: 15-bit window tables for
    launches with many problems, device-side Horner, folds down to 64 generators per chunk): three chunks per case against the oracle's
    single-chunk prover, round trip and tamper (tests/gpu_fuzz_many_chunks.py, 30 s here; 75 cases in 300 s when run on its own)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_fuzz_many_chunks.py"), "30", "2026"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "fuzz ok:" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
