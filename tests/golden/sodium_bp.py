"""A THIRD implementation of the Bulletproofs range proof -- verifier AND prover -- that shares no arithmetic with the oracle
(oracle/*.c, 51-bit limbs) or the product (HIP, 25.5-bit limbs): every group operation is libsodium 1.0.18's
crypto_core_ristretto255_* / crypto_scalarmult_ristretto255 (ctypes), scalars are Python integers mod l, the generator chain is
hashlib.shake_256, and the transcript is pyref.Transcript (pure-Python STROBE-128 / Merlin, Keccak-f validated against hashlib,
pinned to merlin's published test vector).

It restates upstream `bulletproofs 4.0.0` (RangeProof::{prove_multiple_with_rng, verify_multiple}, InnerProductProof::{create,
verification_scalars}, the dealer / party transcript protocol) and the thin wrappers of rofl_crypto
(range_proof_vec/mod.rs:16-102, 149-216; l2_range_proof_vec/mod.rs:15-140, 185-253) the naive way: no Pippenger, no batching, one
scalar multiplication per term, the two verification equations checked separately instead of as one random linear combination.
Used by tests/test_sodium_crosscheck.py (CPU, -m "not gpu") on the committed fixtures: a fixture accepted here AND reproduced
byte for byte here from its explicit nonce stream is pinned by three independent code bases.  Test infrastructure only."""
import ctypes
import hashlib
import os

import pyref as R

L = R.L
_CANDIDATES = ("/opt/conda/lib/libsodium.so", "libsodium.so.23", "libsodium.so")


def load_sodium():
    for c in _CANDIDATES:
        try:
            so = ctypes.CDLL(c)
            if so.sodium_init() < 0:
                continue
            return so
        except OSError:
            continue
    return None


so = load_sodium()
IDENTITY = bytes(32)


def _buf():
    return ctypes.create_string_buffer(32)


def smul(k, p):
    """k * p (k any integer, reduced mod l; 0 gives the identity -- libsodium refuses k = 0)"""
    k %= L
    if k == 0 or p == IDENTITY:
        return IDENTITY
    o = _buf()
    if so.crypto_scalarmult_ristretto255(o, k.to_bytes(32, "little"), p) != 0:
        raise ValueError("invalid point")
    return o.raw


def padd(p, q):
    if p == IDENTITY:
        return q
    if q == IDENTITY:
        return p
    o = _buf()
    if so.crypto_core_ristretto255_add(o, p, q) != 0:
        raise ValueError("invalid point")
    return o.raw


def psub(p, q):
    if q == IDENTITY:
        return p
    o = _buf()
    if p == IDENTITY:
        # -q = 0 - q : libsodium has no negation; (l - 1) * q
        return smul(L - 1, q)
    if so.crypto_core_ristretto255_sub(o, p, q) != 0:
        raise ValueError("invalid point")
    return o.raw


def is_valid(p):
    return p == IDENTITY or bool(so.crypto_core_ristretto255_is_valid_point(p))


def from_hash(b64):
    o = _buf(); so.crypto_core_ristretto255_from_hash(o, b64); return o.raw


def msm(scalars, points):
    acc = IDENTITY
    for k, p in zip(scalars, points):
        acc = padd(acc, smul(k, p))
    return acc


B = (lambda: (lambda o: (so.crypto_scalarmult_ristretto255_base(o, (1).to_bytes(32, "little")), o.raw)[1])(_buf()))() if so else None
B_BLINDING = from_hash(hashlib.sha3_512(B).digest()) if so else None          # PedersenGens::default()


def gens(n, m):
    """BulletproofGens::new(n, m): party-major G, H (GeneratorsChain = SHAKE256("GeneratorsChain" || 'G'/'H' || u32le(j)))"""
    G, H = [], []
    for j in range(m):
        for which, dst in ((b"G", G), (b"H", H)):
            xof = hashlib.shake_256(b"GeneratorsChain" + which + j.to_bytes(4, "little")).digest(64 * n)
            dst.extend(from_hash(xof[64 * i:64 * i + 64]) for i in range(n))
    return G, H


def _sc(b):
    return int.from_bytes(b, "little")


def _canonical(b):
    return _sc(b) < L


def _inv(x):
    return pow(x % L, -1, L)


def _transcript_start(label, n, m, V):
    t = R.Transcript(label)
    t.append_message(b"dom-sep", b"rangeproof v1")
    t.append_u64(b"n", n); t.append_u64(b"m", m)
    for v in V:
        t.append_message(b"V", v)
    return t


def verify_single(proof, V, n, label=b"RangeProof"):
    """RangeProof::verify_multiple for ONE aggregated proof over the m = len(V) compressed commitments V.
    Returns True / False (VerificationError); raises ValueError for FormatError."""
    m = len(V)
    if len(proof) % 32 or len(proof) < 7 * 32:
        raise ValueError("FormatError")
    ne = (len(proof) - 7 * 32) // 32
    if ne < 2 or (ne - 2) % 2:
        raise ValueError("FormatError")
    lg = (ne - 2) // 2
    w32 = [proof[32 * i:32 * i + 32] for i in range(len(proof) // 32)]
    A, S, T1, T2 = w32[0:4]
    if not all(_canonical(x) for x in w32[4:7] + w32[-2:]):
        raise ValueError("FormatError")
    t_x, t_x_bl, e_bl = (_sc(x) for x in w32[4:7])
    Ls = [w32[7 + 2 * k] for k in range(lg)]; Rs = [w32[8 + 2 * k] for k in range(lg)]
    a, b = _sc(w32[-2]), _sc(w32[-1])
    N = n * m
    if N != 1 << lg or n not in (8, 16, 32, 64) or m & (m - 1):
        return False
    if any(p == IDENTITY for p in (A, S, T1, T2)) or any(p == IDENTITY for p in Ls + Rs):       # validate_and_append_point
        return False
    if not all(is_valid(p) for p in [A, S, T1, T2] + Ls + Rs + list(V)):
        return False
    t = _transcript_start(label, n, m, V)
    t.append_message(b"A", A); t.append_message(b"S", S)
    y = t.challenge_scalar(b"y"); z = t.challenge_scalar(b"z")
    t.append_message(b"T_1", T1); t.append_message(b"T_2", T2)
    x = t.challenge_scalar(b"x")
    for lab, v in ((b"t_x", t_x), (b"t_x_blinding", t_x_bl), (b"e_blinding", e_bl)):
        t.append_message(lab, v.to_bytes(32, "little"))
    w = t.challenge_scalar(b"w")
    t.append_message(b"dom-sep", b"ipp v1"); t.append_u64(b"n", N)
    us = []
    for k in range(lg):
        t.append_message(b"L", Ls[k]); t.append_message(b"R", Rs[k])
        us.append(t.challenge_scalar(b"u"))
    zz = z * z % L
    # (1) t(x) commitment:  t_x B + t_x_bl Bb == z^2 sum_j z^j V_j + delta(y, z) B + x T_1 + x^2 T_2
    sum_y = sum(pow(y, i, L) for i in range(N)) % L
    sum_2 = (pow(2, n, L) - 1) % L
    sum_z = sum(pow(z, j, L) for j in range(m)) % L
    delta = ((z - zz) * sum_y - zz * z % L * sum_2 % L * sum_z) % L
    lhs = padd(smul(t_x, B), smul(t_x_bl, B_BLINDING))
    rhs = padd(padd(msm([zz * pow(z, j, L) % L for j in range(m)], V), smul(delta, B)), padd(smul(x, T1), smul(x * x % L, T2)))
    if lhs != rhs:
        return False
    # (2) inner-product argument on  P = A + x S - z <1, G> + <z y^i + z^(2+j) 2^(i mod n), H'>  - e_bl Bb + t_x w B,  H'_i = y^-i H_i
    G, H = gens(n, m)
    yinv = _inv(y)
    P = padd(A, smul(x, S))
    P = padd(P, msm([(-z) % L] * N, G))
    hs = [(z + pow(yinv, i, L) * (zz * pow(z, i // n, L) % L) % L * pow(2, i % n, L)) % L for i in range(N)]
    P = padd(P, msm(hs, H))
    P = psub(P, smul(e_bl, B_BLINDING))
    Q = smul(w, B)
    P = padd(P, smul(t_x, Q))
    # fold: P' = P + sum u_k^2 L_k + u_k^-2 R_k ;  s_i = prod_k u_k^(+-1), bit lg-1-k of i
    for k in range(lg):
        P = padd(P, padd(smul(us[k] * us[k] % L, Ls[k]), smul(_inv(us[k] * us[k] % L), Rs[k])))
    s = []
    for i in range(N):
        acc = 1
        for k in range(lg):
            acc = acc * (us[k] if (i >> (lg - 1 - k)) & 1 else _inv(us[k])) % L
        s.append(acc)
    want = padd(msm([a * si % L for si in s], G), msm([b * _inv(si) % L * pow(yinv, i, L) % L for i, si in enumerate(s)], H))
    want = padd(want, smul(a * b % L, Q))
    return want == P


class StreamRng:
    """Scalar::random(rng) = 64 bytes from the RNG, from_bytes_mod_order_wide -- fed from an explicit stream"""

    def __init__(self, stream):
        self.s, self.pos = bytes(stream), 0

    def scalar(self):
        chunk = self.s[self.pos:self.pos + 64]
        if len(chunk) < 64:
            raise ValueError("nonce stream too short")
        self.pos += 64
        return _sc(chunk) % L


def prove_single(values, blindings, n, rng, label=b"RangeProof"):
    """RangeProof::prove_multiple_with_rng: values = m unsigned integers < 2^n, blindings = m scalars.  Returns (proof bytes, [V_j])."""
    m = len(values); N = n * m; lg = N.bit_length() - 1
    assert 1 << lg == N
    G, H = gens(n, m)
    V = [padd(smul(v, B), smul(r, B_BLINDING)) for v, r in zip(values, blindings)]
    t = _transcript_start(label, n, m, V)
    a_bl, s_bl, sL, sR = [], [], [], []
    A = IDENTITY; S = IDENTITY
    for j in range(m):                    # party j: a_blinding, s_blinding, s_L[0..n), s_R[0..n)
        ab = rng.scalar(); sb = rng.scalar()
        sl = [rng.scalar() for _ in range(n)]; sr = [rng.scalar() for _ in range(n)]
        a_bl.append(ab); s_bl.append(sb); sL += sl; sR += sr
        Aj = smul(ab, B_BLINDING)
        for i in range(n):
            Aj = padd(Aj, G[j * n + i]) if (values[j] >> i) & 1 else psub(Aj, H[j * n + i])
        A = padd(A, Aj)
        S = padd(S, padd(smul(sb, B_BLINDING), padd(msm(sl, G[j * n:(j + 1) * n]), msm(sr, H[j * n:(j + 1) * n]))))
    t.append_message(b"A", A); t.append_message(b"S", S)
    y = t.challenge_scalar(b"y"); z = t.challenge_scalar(b"z")
    zz = z * z % L
    l0, l1, r0, r1 = [], sL, [], []
    for k in range(N):
        j, i = divmod(k, n)
        bit = (values[j] >> i) & 1
        yk = pow(y, k, L)
        l0.append((bit - z) % L)
        r0.append((yk * ((bit - 1 + z) % L) + zz * pow(z, j, L) % L * pow(2, i, L)) % L)
        r1.append(yk * sR[k] % L)
    t0 = sum(a * b for a, b in zip(l0, r0)) % L
    t2 = sum(a * b for a, b in zip(l1, r1)) % L
    t1 = (sum((a + c) * (b + d) for a, b, c, d in zip(l0, r0, l1, r1)) - t0 - t2) % L
    t1_bl, t2_bl = [], []
    for j in range(m):                    # party j: t_1_blinding, t_2_blinding
        t1_bl.append(rng.scalar()); t2_bl.append(rng.scalar())
    T1 = padd(smul(t1, B), smul(sum(t1_bl) % L, B_BLINDING)); T2 = padd(smul(t2, B), smul(sum(t2_bl) % L, B_BLINDING))
    t.append_message(b"T_1", T1); t.append_message(b"T_2", T2)
    x = t.challenge_scalar(b"x")
    t_x = (t0 + t1 * x + t2 * x % L * x) % L
    t_x_bl = (sum(zz * pow(z, j, L) % L * blindings[j] for j in range(m)) + x * sum(t1_bl) + x * x % L * sum(t2_bl)) % L
    e_bl = (sum(a_bl) + x * sum(s_bl)) % L
    for lab, v in ((b"t_x", t_x), (b"t_x_blinding", t_x_bl), (b"e_blinding", e_bl)):
        t.append_message(lab, v.to_bytes(32, "little"))
    w = t.challenge_scalar(b"w")
    Q = smul(w, B)
    a = [(p + q * x) % L for p, q in zip(l0, l1)]
    b = [(p + q * x) % L for p, q in zip(r0, r1)]
    yinv = _inv(y)
    Hp = [smul(pow(yinv, i, L), H[i]) for i in range(N)]
    Gc = list(G)
    t.append_message(b"dom-sep", b"ipp v1"); t.append_u64(b"n", N)
    LR = []
    size = N
    while size > 1:
        h = size // 2
        aL, aR, bL, bR, GL, GR, HL, HR = a[:h], a[h:], b[:h], b[h:], Gc[:h], Gc[h:], Hp[:h], Hp[h:]
        cL = sum(p * q for p, q in zip(aL, bR)) % L; cR = sum(p * q for p, q in zip(aR, bL)) % L
        Lp = padd(padd(msm(aL, GR), msm(bR, HL)), smul(cL, Q))
        Rp = padd(padd(msm(aR, GL), msm(bL, HR)), smul(cR, Q))
        LR += [Lp, Rp]
        t.append_message(b"L", Lp); t.append_message(b"R", Rp)
        u = t.challenge_scalar(b"u"); ui = _inv(u)
        a = [(p * u + q * ui) % L for p, q in zip(aL, aR)]
        b = [(p * ui + q * u) % L for p, q in zip(bL, bR)]
        Gc = [padd(smul(ui, p), smul(u, q)) for p, q in zip(GL, GR)]
        Hp = [padd(smul(u, p), smul(ui, q)) for p, q in zip(HL, HR)]
        size = h
    proof = A + S + T1 + T2 + b"".join(v.to_bytes(32, "little") for v in (t_x, t_x_bl, e_bl)) + b"".join(LR) + a[0].to_bytes(32, "little") + b[0].to_bytes(32, "little")
    return proof, V


# ---------------------------------------------------------------- rofl_crypto wrappers
def next_pow2(v):
    return 1 if v <= 1 else 1 << (v - 1).bit_length()


def fix_bits(v, fp_bits, fp_frac):
    """|v| as Fix::saturating_from_float(..).to_bits() (conversion32.rs:11-18); ties: assumed half-to-even (numpy.rint), see
    tests/golden/proofs.json "tie_cases"."""
    import numpy as np
    x = abs(float(np.float32(v))) * (1 << fp_frac)
    k = int(np.rint(x)) if x < 2 ** fp_bits else 2 ** fp_bits - 1
    return min(k, 2 ** fp_bits - 1)


def f32_to_scalar(v, fp_bits, fp_frac):
    import numpy as np
    k = fix_bits(v, fp_bits, fp_frac)
    return (-k) % L if np.float32(v) < 0 else k


def verify_rangeproof(proofs, commits, prove_range):
    """range_proof_vec/mod.rs:149-191: shift by 2^(range-1) B, pad with the identity, chunk by len / proofs.len(), AND."""
    off = smul(1 << (prove_range - 1), B)
    shifted = [padd(c, off) for c in commits]
    dp = next_pow2(len(shifted))
    shifted += [IDENTITY] * (dp - len(shifted))
    chunk = dp // len(proofs)
    ok = True
    for c, pr in enumerate(proofs):
        ok &= verify_single(pr, shifted[c * chunk:(c + 1) * chunk], prove_range)
    return ok


def create_rangeproof(values_f32, blindings, prove_range, n_partition, fp_bits, fp_frac, stream):
    """range_proof_vec/mod.rs:16-102 with the chunks' nonces taken from one explicit stream (chunk c at offset c*m*(2n+4) scalars)."""
    d = len(values_f32); dp = next_pow2(d)
    off = 1 << (prove_range - 1)
    mask = (1 << fp_bits) - 1
    shifted = [((f32_to_scalar(v, fp_bits, fp_frac) + off) % L) & mask for v in values_f32] + [0] * (dp - d)       # read_from_bytes: low fp_bits bits
    bl = list(blindings) + [0] * (dp - d)
    n_chunks = min(dp, n_partition); m = dp // n_chunks
    per = m * (2 * prove_range + 4) * 64
    proofs, commits = [], []
    for c in range(dp // m):
        pr, V = prove_single(shifted[c * m:(c + 1) * m], bl[c * m:(c + 1) * m], prove_range, StreamRng(stream[c * per:(c + 1) * per]))
        proofs.append(pr); commits += V
    negoff = smul(L - off, B)
    return proofs, [padd(v, negoff) for v in commits[:d]]


# ---------------------------------------------------------------- per-element Sigma-proofs (rofl_crypto's own code, in the reference repo)
def _sigma_challenge(label, parts):
    t = R.Transcript(label)
    t.append_message(b"dom-sep", b"randomness proof v1")          # rand_proof/transcript.rs:20-22
    for lab, msg in parts:
        t.append_message(lab, msg)
    return t.challenge_scalar(b"c")


def verify_randproof(proof, pair):
    """RandProof::verify (rand_proof/mod.rs:69-91): proof = C'.L | C'.R | Z_m | Z_r, pair = L | R"""
    Lp, Rp, zm, zr = proof[0:32], proof[32:64], proof[64:96], proof[96:128]
    if not (_canonical(zm) and _canonical(zr)) or not all(is_valid(p) for p in (Lp, Rp, pair[:32], pair[32:])):
        raise ValueError("FormatError")
    c = _sigma_challenge(b"RandProof", [(b"C", pair), (b"C_prime", proof[:64])])
    zm, zr = _sc(zm), _sc(zr)
    okL = padd(smul(zm, B), smul(zr, B_BLINDING)) == padd(Lp, smul(c, pair[:32]))       # eg_gens.commit(Z_m, Z_r).L
    okR = smul(zr, B) == padd(Rp, smul(c, pair[32:]))
    return okL and okR


def verify_squarerandproof(proof, commits):
    """SquareRandProof::verify (square_rand_proof/mod.rs:77-109): proof = C'.L | C'.R | c_sq' | Z_m | Z_r1 | Z_r2, commits = L | R | c_sq"""
    Lp, Rp, Sp = proof[0:32], proof[32:64], proof[64:96]
    zm, zr1, zr2 = proof[96:128], proof[128:160], proof[160:192]
    Lc, Rc, Sc = commits[0:32], commits[32:64], commits[64:96]
    if not all(_canonical(z) for z in (zm, zr1, zr2)) or not all(is_valid(p) for p in (Lp, Rp, Sp, Lc, Rc, Sc)):
        raise ValueError("FormatError")
    c = _sigma_challenge(b"SquareRandProof", [(b"C_eg", commits[:64]), (b"C_ped", Sc), (b"C_prime_eg", proof[:64]), (b"C_prime_ped", Sp)])
    zm, zr1, zr2 = _sc(zm), _sc(zr1), _sc(zr2)
    ok1 = padd(smul(zm, B), smul(zr1, B_BLINDING)) == padd(Lp, smul(c, Lc)) and smul(zr1, B) == padd(Rp, smul(c, Rc))
    ok2 = padd(smul(zm, Lc), smul(zr2, B_BLINDING)) == padd(Sp, smul(c, Sc))
    return ok1 and ok2


def create_squarerandproof(m, r1, r2, rng):
    """SquareRandProof::prove (square_rand_proof/{party.rs:25-62, 153-159, dealer.rs}); nonces m', r1', r2' in this order"""
    Lc = padd(smul(m, B), smul(r1, B_BLINDING)); Rc = smul(r1, B); Sc = padd(smul(m * m % L, B), smul(r2, B_BLINDING))
    mp, r1p, r2p = rng.scalar(), rng.scalar(), rng.scalar()
    Lp = padd(smul(mp, B), smul(r1p, B_BLINDING)); Rp = smul(r1p, B); Sp = padd(smul(mp, Lc), smul(r2p, B_BLINDING))
    c = _sigma_challenge(b"SquareRandProof", [(b"C_eg", Lc + Rc), (b"C_ped", Sc), (b"C_prime_eg", Lp + Rp), (b"C_prime_ped", Sp)])
    z = [(mp + m * c) % L, (r1p + r1 * c) % L, (r2p + (r2 - m * r1) * c) % L]
    return Lp + Rp + Sp + b"".join(v.to_bytes(32, "little") for v in z), Lc + Rc + Sc


def create_randproof(m, r, rng):
    """RandProof::prove (rand_proof/party.rs): nonces m', r'"""
    Lc = padd(smul(m, B), smul(r, B_BLINDING)); Rc = smul(r, B)
    mp, rp = rng.scalar(), rng.scalar()
    Lp = padd(smul(mp, B), smul(rp, B_BLINDING)); Rp = smul(rp, B)
    c = _sigma_challenge(b"RandProof", [(b"C", Lc + Rc), (b"C_prime", Lp + Rp)])
    return Lp + Rp + ((mp + m * c) % L).to_bytes(32, "little") + ((rp + r * c) % L).to_bytes(32, "little"), Lc + Rc
