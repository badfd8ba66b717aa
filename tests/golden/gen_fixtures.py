"""Generate tests/golden/*.json.  Run in the BUILD container only (needs /opt/conda/lib/libsodium.so):

    python tests/golden/gen_fixtures.py

Sources of truth, per section:
  ristretto / scalar   : libsodium 1.0.18 crypto_core_ristretto255_* (independent RFC 9496 implementation)
  hashes               : hashlib sha3_512 / shake_256
  merlin               : published merlin test vector (transcript "test protocol") + pyref.Transcript,
                         whose Keccak-f is validated against hashlib in pyref.__main__
  generators, pedersen : hashlib SHAKE256 + libsodium from_hash (pins BulletproofGens / PedersenGens)
  conversion           : numpy float32 round-half-even model of fixed 0.3.3 saturating_from_float
  proofs               : the oracle (oracle/liborc.so) under a fixed nonce seed -- pins HIP == oracle == fixture,
                         NOT == the bulletproofs crate ("parity unpinned")
"""
import ctypes
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import pyref as R  # noqa: E402
import orc  # noqa: E402

so = ctypes.CDLL("/opt/conda/lib/libsodium.so")
so.sodium_init()
rng = np.random.default_rng(20260110)


def rb(n):
    return rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()


def buf():
    return ctypes.create_string_buffer(32)


def smul_base(k):
    o = buf(); so.crypto_scalarmult_ristretto255_base(o, k.to_bytes(32, "little")); return o.raw


def smul(k, p):
    o = buf(); assert so.crypto_scalarmult_ristretto255(o, k.to_bytes(32, "little"), p) == 0; return o.raw


def padd(p, q):
    o = buf(); assert so.crypto_core_ristretto255_add(o, p, q) == 0; return o.raw


def from_hash(b):
    o = buf(); so.crypto_core_ristretto255_from_hash(o, b); return o.raw


out = {}
# ---- ristretto
out["base_multiples"] = [smul_base(k).hex() for k in range(1, 17)]
rs = []
for _ in range(32):
    k = int.from_bytes(rb(32), "little") % R.L
    p = from_hash(rb(64))
    rs.append({"k": k.to_bytes(32, "little").hex(), "p": p.hex(), "kp": smul(k, p).hex(), "kB": smul_base(k).hex()})
out["scalarmult"] = rs
out["from_uniform"] = [{"in": (b := rb(64)).hex(), "out": from_hash(b).hex()} for _ in range(16)]
out["add"] = []
for _ in range(8):
    p, q = from_hash(rb(64)), from_hash(rb(64))
    out["add"].append({"p": p.hex(), "q": q.hex(), "sum": padd(p, q).hex()})
# invalid encodings (RFC 9496 A.3 classes): non-canonical, negative, non-square, y = 0 ...
bad = []
cands = [bytes([1] + [0] * 31), (R.P).to_bytes(32, "little"), (R.P + 2).to_bytes(32, "little"), bytes([0xFF] * 32),
         bytes([2] + [0] * 31), bytes([4] + [0] * 31), bytes([6] + [0] * 31), bytes([8] + [0] * 31)]
for i in range(40):
    cands.append(bytes([2 * (i + 5)]) + rb(31)[:30] + bytes([rb(1)[0] & 0x7F]))
for c in cands:
    valid = bool(so.crypto_core_ristretto255_is_valid_point(c))
    assert valid == (R.ristretto_decode(c) is not None)
    bad.append({"enc": c.hex(), "valid": valid})
out["encodings"] = bad
# ---- scalars
sc = []
for _ in range(16):
    w = rb(64)
    a = int.from_bytes(rb(32), "little") % R.L
    b = int.from_bytes(rb(32), "little") % R.L
    sc.append({"wide": w.hex(), "reduced": (int.from_bytes(w, "little") % R.L).to_bytes(32, "little").hex(),
               "a": a.to_bytes(32, "little").hex(), "b": b.to_bytes(32, "little").hex(),
               "mul": (a * b % R.L).to_bytes(32, "little").hex(), "add": ((a + b) % R.L).to_bytes(32, "little").hex(),
               "inv": pow(a, -1, R.L).to_bytes(32, "little").hex()})
out["scalars"] = sc
# ---- hashes
out["hashes"] = [{"in": (m := rb(n)).hex(), "sha3_512": hashlib.sha3_512(m).hexdigest(), "shake256_200": hashlib.shake_256(m).hexdigest(200)}
                 for n in (0, 1, 71, 72, 73, 135, 136, 137, 300)]
# ---- merlin
t = R.Transcript(b"test protocol"); t.append_message(b"some label", b"some data")
mv = t.challenge_bytes(b"challenge", 32).hex()
assert mv == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"   # merlin's published vector
out["merlin_published"] = mv
script = []
T = R.Transcript(b"RangeProof")
for i in range(24):
    m = rb(int(rng.integers(0, 400)))
    T.append_message(b"V", m)
    step = {"append": m.hex()}
    if i % 5 == 0:
        step["challenge64"] = T.challenge_bytes(b"y", 64).hex()
    script.append(step)
out["merlin_script"] = script
# ---- generators / pedersen
Bc = smul_base(1)
out["pedersen"] = {"B": Bc.hex(), "B_blinding": from_hash(hashlib.sha3_512(Bc).digest()).hex()}
gens = {}
for j in (0, 1, 5):
    for which in (b"G", b"H"):
        xof = hashlib.shake_256(b"GeneratorsChain" + which + j.to_bytes(4, "little")).digest(64 * 9)
        gens[f"{which.decode()}{j}"] = [from_hash(xof[64 * i:64 * i + 64]).hex() for i in range(9)]
out["generators"] = gens
# ---- conversion (fixed-point) tables
conv = []
for fb, ff in ((8, 7), (16, 7), (32, 7), (32, 12)):
    vals = [0.0, 0.25, -0.25, 1.25, -1.5, 0.00390625, 0.01171875, -0.01171875, 0.0117, 1.0 / 3.0, -1.0 / 3.0, 127.99, 1e9, -1e9,
            0.5 / (1 << ff), 1.5 / (1 << ff), 2.5 / (1 << ff), -2.5 / (1 << ff), 3.5 / (1 << ff)]
    for v in vals:
        v32 = np.float32(v)
        x = abs(float(v32)) * (1 << ff)
        k = int(np.rint(x)) if x < 2 ** fb else 2 ** fb - 1      # np.rint: ties to even
        k = min(k, 2 ** fb - 1)
        s = k if not (v32 < 0) else (-k) % R.L
        conv.append({"fp_bits": fb, "fp_frac": ff, "v": float(v32), "scalar": s.to_bytes(32, "little").hex()})
out["conversion"] = conv
# ---- reference test values (range_proof_vec/mod.rs:372-375) as Pedersen commitments with zero blinding
refv = {}
for name, vec in (("x", [0.25, 1.25, -1.5]), ("y", [-0.75, 1.25, -2.0]), ("z", [0.5, 1.25, -3.0]), ("sum", [0.0, 3.75, -6.5])):
    refv[name] = []
    for v in vec:
        k = int(round(abs(v) * 128)); s = k if v >= 0 else (-k) % R.L
        refv[name].append(smul_base(s).hex() if s else (bytes(32)).hex())
out["reference_values_fp16_frac7"] = refv
json.dump(out, open(os.path.join(HERE, "primitives.json"), "w"), indent=0)

# ---- full proofs from the oracle under a fixed nonce seed
proofs = []
prng = np.random.default_rng(7)
for (d, nb, P, fb, ff) in ((1, 8, 1, 16, 7), (5, 8, 4, 16, 7), (3, 16, 4, 16, 7), (6, 32, 2, 32, 7), (1, 64, 1, 64, 7)):
    mn, mx = orc.clip_bounds(nb, fb, ff)
    vals = prng.uniform(max(mn, -1000), min(mx, 1000), size=d).astype(np.float32)
    bl = orc.rand_scalars(prng, d)
    seed = bytes(prng.integers(0, 256, size=32, dtype=np.uint8))
    rc, pr, cm = orc.create_rangeproof(vals, bl, nb, P, fb, ff, seed=seed)
    assert rc == 0 and orc.verify_rangeproof(pr, cm, nb, fb, ff) == (0, True)
    proofs.append({"kind": "linf", "d": d, "prove_range": nb, "n_partition": P, "fp_bits": fb, "fp_frac": ff,
                   "values": [float(x) for x in vals], "blindings": bl.tobytes().hex(), "seed": seed.hex(),
                   "proofs": pr.tobytes().hex(), "n_proofs": int(pr.shape[0]), "commits": cm.tobytes().hex()})
for (vals, nb, fb, ff) in (([1.25, 0.5, 0.25], 16, 16, 7), ([7.9], 32, 32, 7), ([0.25, 1.25, -1.5], 16, 16, 7), ([0.0078125] * 40, 8, 32, 7)):
    vals = np.array(vals, dtype=np.float32)
    bl = orc.rand_scalars(prng, len(vals))
    seed = bytes(prng.integers(0, 256, size=32, dtype=np.uint8))
    rc, pr, cm = orc.create_rangeproof_l2(vals, bl, nb, 4, fb, ff, seed=seed)
    assert rc == 0 and orc.verify_rangeproof_l2(pr, cm, nb, fb, ff) == (0, True)
    proofs.append({"kind": "l2", "d": len(vals), "prove_range": nb, "n_partition": 4, "fp_bits": fb, "fp_frac": ff,
                   "values": [float(x) for x in vals], "blindings": bl.tobytes().hex(), "seed": seed.hex(),
                   "proofs": pr.tobytes().hex(), "n_proofs": 1, "commits": cm.tobytes().hex()})
for kind, vals, fb, ff in ((0, [0.25, 1.25, -1.5], 16, 7), (1, [0.25, 1.25, -1.5], 16, 7), (1, [-7.9, 0.0078125], 32, 7)):
    vals = np.array(vals, dtype=np.float32)
    r1, r2 = orc.rand_scalars(prng, len(vals)), orc.rand_scalars(prng, len(vals))
    seed = bytes(prng.integers(0, 256, size=32, dtype=np.uint8))
    rc, pr, cm = orc.sigma_create(kind, vals, r1, r2 if kind else None, fb, ff, seed=seed)
    assert rc == 0 and orc.sigma_verify(kind, pr, cm) == (0, True)
    # independent check of the commitments with libsodium: L = m B + r1 Bb, R = r1 B, c_sq = m^2 B + r2 Bb
    Bb = from_hash(hashlib.sha3_512(smul_base(1)).digest())
    for i, v in enumerate(vals):
        k = int(round(abs(float(v)) * (1 << ff))); m = k if v >= 0 else (-k) % R.L
        r1i = int.from_bytes(r1[i].tobytes(), "little"); r2i = int.from_bytes(r2[i].tobytes(), "little")
        Lx = padd(smul_base(m), smul(r1i, Bb)) if m else smul(r1i, Bb)
        assert cm[i, :32].tobytes() == Lx and cm[i, 32:64].tobytes() == smul_base(r1i)
        if kind:
            assert cm[i, 64:].tobytes() == padd(smul_base(m * m % R.L), smul(r2i, Bb))
    proofs.append({"kind": "rand" if kind == 0 else "sqrand", "d": len(vals), "fp_bits": fb, "fp_frac": ff, "values": [float(x) for x in vals],
                   "r1": r1.tobytes().hex(), "r2": r2.tobytes().hex(), "seed": seed.hex(), "proofs": pr.tobytes().hex(), "commits": cm.tobytes().hex()})
# ---- mode-0 fixtures: EXPLICIT nonce streams (64-byte wide scalars in the upstream draw order).  These are the ones a maintainer can
# replay through bulletproofs' prove_multiple_with_rng / rofl_crypto's Sigma-proof parties (scripts/crosscheck_rust) and that the
# libsodium implementation (tests/golden/sodium_bp.py) reproduces byte for byte.
import sodium_bp as SB  # noqa: E402
srng = np.random.default_rng(20261002)
for (d, nb, P, fb, ff) in ((2, 8, 2, 16, 7), (3, 8, 4, 16, 7), (2, 32, 1, 32, 7), (3, 16, 2, 16, 7)):
    mn, mx = orc.clip_bounds(nb, fb, ff)
    vals = srng.uniform(max(mn, -100), min(mx, 100), size=d).astype(np.float32)
    bl = orc.rand_scalars(srng, d)
    dp = SB.next_pow2(d); m = dp // min(dp, P)
    stream = srng.integers(0, 256, (dp // m) * m * (2 * nb + 4) * 64, dtype=np.uint8).tobytes()
    rc, pr, cm = orc.create_rangeproof(vals, bl, nb, P, fb, ff, stream=stream)
    assert rc == 0 and orc.verify_rangeproof(pr, cm, nb, fb, ff) == (0, True)
    spr, scm = SB.create_rangeproof([float(v) for v in vals], [int.from_bytes(b.tobytes(), "little") for b in bl], nb, P, fb, ff, stream)
    assert b"".join(spr) == pr.tobytes() and b"".join(scm) == cm.tobytes()          # oracle == libsodium implementation
    proofs.append({"kind": "linf", "nonce": "stream", "d": d, "prove_range": nb, "n_partition": P, "fp_bits": fb, "fp_frac": ff,
                   "values": [float(x) for x in vals], "blindings": bl.tobytes().hex(), "stream": stream.hex(),
                   "proofs": pr.tobytes().hex(), "n_proofs": int(pr.shape[0]), "commits": cm.tobytes().hex()})
for (vals, nb, fb, ff) in (([1.25, 0.5, 0.25], 16, 16, 7), ([3.5, -2.25], 32, 32, 7)):
    vals = np.array(vals, dtype=np.float32)
    bl = orc.rand_scalars(srng, len(vals))
    stream = srng.integers(0, 256, (2 * nb + 4) * 64, dtype=np.uint8).tobytes()
    rc, pr, cm = orc.create_rangeproof_l2(vals, bl, nb, 4, fb, ff, stream=stream)
    assert rc == 0 and orc.verify_rangeproof_l2(pr, cm, nb, fb, ff) == (0, True)
    # third implementation: the L2 proof is one (n, m = 1) proof over sum x^2 with blinding sum(blindings), label "L2RangeProof"
    val = sum(SB.f32_to_scalar(float(v), fb, ff) ** 2 for v in vals) % R.L
    spr, sV = SB.prove_single([val & ((1 << fb) - 1)], [sum(int.from_bytes(b.tobytes(), "little") for b in bl) % R.L], nb, SB.StreamRng(stream), label=b"L2RangeProof")
    assert spr == pr.tobytes() and sV[0] == cm.tobytes()
    proofs.append({"kind": "l2", "nonce": "stream", "d": len(vals), "prove_range": nb, "n_partition": 4, "fp_bits": fb, "fp_frac": ff,
                   "values": [float(x) for x in vals], "blindings": bl.tobytes().hex(), "stream": stream.hex(),
                   "proofs": pr.tobytes().hex(), "n_proofs": 1, "commits": cm.tobytes().hex()})
for kind, vals, fb, ff in ((0, [0.25, -1.5], 16, 7), (1, [0.25, 1.25, -1.5], 16, 7)):
    vals = np.array(vals, dtype=np.float32)
    r1, r2 = orc.rand_scalars(srng, len(vals)), orc.rand_scalars(srng, len(vals))
    nn = 3 if kind else 2
    stream = srng.integers(0, 256, nn * len(vals) * 64, dtype=np.uint8).tobytes()
    rc, pr, cm = orc.sigma_create(kind, vals, r1, r2 if kind else None, fb, ff, stream=stream)
    assert rc == 0 and orc.sigma_verify(kind, pr, cm) == (0, True)
    for i, v in enumerate(vals):
        mi = SB.f32_to_scalar(float(v), fb, ff); r1i = int.from_bytes(r1[i].tobytes(), "little"); r2i = int.from_bytes(r2[i].tobytes(), "little")
        rg = SB.StreamRng(stream[nn * 64 * i:nn * 64 * (i + 1)])
        sp, sc_ = SB.create_squarerandproof(mi, r1i, r2i, rg) if kind else SB.create_randproof(mi, r1i, rg)
        assert sp == pr[i].tobytes() and sc_ == cm[i].tobytes()
    proofs.append({"kind": "rand" if kind == 0 else "sqrand", "nonce": "stream", "d": len(vals), "fp_bits": fb, "fp_frac": ff, "values": [float(x) for x in vals],
                   "r1": r1.tobytes().hex(), "r2": r2.tobytes().hex(), "stream": stream.hex(), "proofs": pr.tobytes().hex(), "commits": cm.tobytes().hex()})
# ---- f32 -> fixed ties: values exactly half-way between two grid points.  fixed 0.3.3's saturating_from_float is taken to round
# half to EVEN (the reference pins only |error| <= 2^-(frac+1), conversion32.rs:196-214); a maintainer with the crate checks these
# first (scripts/crosscheck_rust prints them): a different tie rule changes the commitment of every such value.
ties = []
for fb, ff in ((16, 7), (32, 7), (32, 12), (8, 3)):
    for k in (0, 1, 2, 3, 4, 5, 126, 127):
        v = np.float32((k + 0.5) / (1 << ff))
        assert float(v) * (1 << ff) == k + 0.5                      # exactly representable: a true tie
        for sign in (1, -1):
            rc_buf = np.zeros(32, np.uint8)
            assert orc.lib().orc_f32_to_scalar(ctypes.c_float(sign * float(v)), fb, ff, rc_buf.ctypes.data_as(ctypes.c_void_p)) == 0
            want = k + (k & 1)                                        # half to even
            got = int.from_bytes(rc_buf.tobytes(), "little")
            assert got == (want if sign > 0 else (-want) % R.L)
            ties.append({"fp_bits": fb, "fp_frac": ff, "v": sign * float(v), "bits_half_even": want, "bits_half_away": k + 1,
                         "scalar": rc_buf.tobytes().hex(), "assumes": "half-to-even"})
proofs.append({"kind": "tie_cases", "cases": ties})
json.dump(proofs, open(os.path.join(HERE, "proofs.json"), "w"), indent=0)
print("wrote primitives.json, proofs.json")
