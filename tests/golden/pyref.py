"""Pure-Python model of the primitives under the hot path (small cases only).

Used ONLY to (a) derive constants, (b) generate golden fixtures that are cross-checked
against independent implementations available in the build container
(libsodium 1.0.18 ristretto255 via ctypes, hashlib sha3/shake).  It is test
infrastructure, never imported by the product package.

Algorithms follow RFC 9496 (ristretto255), FIPS 202 (Keccak), STROBE v1.0.2 / Merlin v1.0
as used by the crates pinned in the reference's Cargo.lock (curve25519-dalek-ng 4.1.1,
merlin 3.0.0, bulletproofs 4.0.0) -- none of which are vendored in /root/reference.
"""
import hashlib

P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493
D = (-121665 * pow(121666, P - 2, P)) % P
SQRT_M1 = pow(2, (P - 1) // 4, P)


def is_neg(x):
    return (x % P) & 1


def fabs(x):
    x %= P
    return P - x if x & 1 else x


def sqrt_ratio_m1(u, v):
    """RFC 9496 4.2 SQRT_RATIO_M1."""
    u %= P
    v %= P
    v3 = v * v % P * v % P
    v7 = v3 * v3 % P * v % P
    r = u * v3 % P * pow(u * v7 % P, (P - 5) // 8, P) % P
    check = v * r % P * r % P
    correct = check == u
    flipped = check == (-u) % P
    flipped_i = check == (-u * SQRT_M1) % P
    if flipped or flipped_i:
        r = r * SQRT_M1 % P
    r = fabs(r)
    return (correct or flipped), r


_ok, INVSQRT_A_MINUS_D = sqrt_ratio_m1(1, (-1 - D) % P)
assert _ok
# sqrt(a*d - 1) with a = -1
_ok, _r = sqrt_ratio_m1((-D - 1) % P, 1)
assert _ok
# RFC 9496 / dalek use the odd ("negative") root:
# 25063068953384623474111414158702152701244531502492656460079210482610430750235
SQRT_AD_MINUS_ONE = P - _r
ONE_MINUS_D_SQ = (1 - D * D) % P
D_MINUS_ONE_SQ = (D - 1) * (D - 1) % P


# ---- Edwards points in extended coordinates (X, Y, Z, T), a = -1 ----
def pt_add(p, q):
    X1, Y1, Z1, T1 = p
    X2, Y2, Z2, T2 = q
    A = (Y1 - X1) * (Y2 - X2) % P
    B = (Y1 + X1) * (Y2 + X2) % P
    C = T1 * 2 * D % P * T2 % P
    Dd = Z1 * 2 * Z2 % P
    E = B - A
    F = Dd - C
    G = Dd + C
    H = B + A
    return (E * F % P, G * H % P, F * G % P, E * H % P)


def pt_neg(p):
    X, Y, Z, T = p
    return ((-X) % P, Y, Z, (-T) % P)


IDENT = (0, 1, 1, 0)


def pt_mul(k, p):
    r = IDENT
    while k:
        if k & 1:
            r = pt_add(r, p)
        p = pt_add(p, p)
        k >>= 1
    return r


_by = 4 * pow(5, P - 2, P) % P
_ok, _bx = sqrt_ratio_m1((_by * _by - 1) % P, (D * _by * _by + 1) % P)
assert _ok
BASE = (_bx, _by, 1, _bx * _by % P)


def ristretto_encode(p):
    X0, Y0, Z0, T0 = p
    u1 = (Z0 + Y0) * (Z0 - Y0) % P
    u2 = X0 * Y0 % P
    _, invsqrt = sqrt_ratio_m1(1, u1 * u2 % P * u2 % P)
    den1 = invsqrt * u1 % P
    den2 = invsqrt * u2 % P
    z_inv = den1 * den2 % P * T0 % P
    ix0 = X0 * SQRT_M1 % P
    iy0 = Y0 * SQRT_M1 % P
    enchanted = den1 * INVSQRT_A_MINUS_D % P
    rotate = is_neg(T0 * z_inv)
    if rotate:
        x, y, den_inv = iy0, ix0, enchanted
    else:
        x, y, den_inv = X0, Y0, den2
    if is_neg(x * z_inv):
        y = (-y) % P
    s = fabs(den_inv * (Z0 - y) % P)
    return s.to_bytes(32, "little")


def ristretto_decode(b):
    s = int.from_bytes(b, "little")
    if s >= P or (s & 1):
        return None
    ss = s * s % P
    u1 = (1 - ss) % P
    u2 = (1 + ss) % P
    u2s = u2 * u2 % P
    v = (-(D * u1 % P * u1) - u2s) % P
    ok, invsqrt = sqrt_ratio_m1(1, v * u2s % P)
    den_x = invsqrt * u2 % P
    den_y = invsqrt * den_x % P * v % P
    x = fabs(2 * s * den_x % P)
    y = u1 * den_y % P
    t = x * y % P
    if (not ok) or is_neg(t) or y == 0:
        return None
    return (x, y, 1, t)


def elligator(r0):
    r = SQRT_M1 * r0 % P * r0 % P
    Ns = (r + 1) * ONE_MINUS_D_SQ % P
    c = P - 1
    Dn = (c - D * r) % P * ((r + D) % P) % P
    ok, s = sqrt_ratio_m1(Ns, Dn)
    s_prime = (-fabs(s * r0 % P)) % P
    if not ok:
        s = s_prime
        c = r
    Nt = (c * ((r - 1) % P) % P * D_MINUS_ONE_SQ - Dn) % P
    ss = s * s % P
    W0 = 2 * s * Dn % P
    W1 = Nt * SQRT_AD_MINUS_ONE % P
    W2 = (1 - ss) % P
    W3 = (1 + ss) % P
    return (W0 * W3 % P, W2 * W1 % P, W1 * W3 % P, W0 * W2 % P)


def from_uniform_bytes(b64):
    r1 = int.from_bytes(b64[:32], "little") & ((1 << 255) - 1)
    r2 = int.from_bytes(b64[32:], "little") & ((1 << 255) - 1)
    return pt_add(elligator(r1 % P), elligator(r2 % P))


# ---- Keccak-f[1600], STROBE-128, Merlin ----
RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
      0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
      0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
      0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
      0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
      0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
ROTC = [1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44]
PILN = [10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1]
M64 = (1 << 64) - 1


def rol(x, n):
    return ((x << n) | (x >> (64 - n))) & M64


def keccak_f(st):
    for rnd in range(24):
        bc = [st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20] for i in range(5)]
        for i in range(5):
            t = bc[(i + 4) % 5] ^ rol(bc[(i + 1) % 5], 1)
            for j in range(0, 25, 5):
                st[j + i] ^= t
        t = st[1]
        for i in range(24):
            j = PILN[i]
            bc0 = st[j]
            st[j] = rol(t, ROTC[i])
            t = bc0
        for j in range(0, 25, 5):
            bc = st[j:j + 5]
            for i in range(5):
                st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5] & M64
        st[0] ^= RC[rnd]
    return st


def keccak_f_bytes(b):
    st = [int.from_bytes(b[8 * i:8 * i + 8], "little") for i in range(25)]
    keccak_f(st)
    return bytearray(b"".join(x.to_bytes(8, "little") for x in st))


def sponge(rate, data, suffix, outlen):
    st = bytearray(200)
    data = bytearray(data)
    data.append(suffix)
    while len(data) % rate:
        data.append(0)
    data[-1] ^= 0x80
    for off in range(0, len(data), rate):
        for i in range(rate):
            st[i] ^= data[off + i]
        st = keccak_f_bytes(st)
    out = bytearray()
    while len(out) < outlen:
        out += st[:rate]
        if len(out) < outlen:
            st = keccak_f_bytes(st)
    return bytes(out[:outlen])


class Strobe128:
    R = 166

    def __init__(self, label):
        st = bytearray(200)
        st[0:6] = bytes([1, self.R + 2, 1, 0, 1, 96])
        st[6:18] = b"STROBEv1.0.2"
        self.st = keccak_f_bytes(st)
        self.pos = 0
        self.pos_begin = 0
        self.cur_flags = 0
        self.meta_ad(label, False)

    def run_f(self):
        self.st[self.pos] ^= self.pos_begin
        self.st[self.pos + 1] ^= 0x04
        self.st[self.R + 1] ^= 0x80
        self.st = keccak_f_bytes(self.st)
        self.pos = 0
        self.pos_begin = 0

    def absorb(self, data):
        for b in data:
            self.st[self.pos] ^= b
            self.pos += 1
            if self.pos == self.R:
                self.run_f()

    def squeeze(self, n):
        out = bytearray()
        for _ in range(n):
            out.append(self.st[self.pos])
            self.st[self.pos] = 0
            self.pos += 1
            if self.pos == self.R:
                self.run_f()
        return bytes(out)

    def begin_op(self, flags, more):
        if more:
            assert self.cur_flags == flags
            return
        old = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self.absorb(bytes([old, flags]))
        if (flags & (4 | 32)) and self.pos != 0:
            self.run_f()

    def meta_ad(self, data, more):
        self.begin_op(16 | 2, more)
        self.absorb(data)

    def ad(self, data, more):
        self.begin_op(2, more)
        self.absorb(data)

    def prf(self, n, more=False):
        self.begin_op(1 | 2 | 4, more)
        return self.squeeze(n)


class Transcript:
    def __init__(self, label):
        self.s = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def append_message(self, label, msg):
        self.s.meta_ad(label, False)
        self.s.meta_ad(len(msg).to_bytes(4, "little"), True)
        self.s.ad(msg, False)

    def append_u64(self, label, x):
        self.append_message(label, x.to_bytes(8, "little"))

    def challenge_bytes(self, label, n):
        self.s.meta_ad(label, False)
        self.s.meta_ad(n.to_bytes(4, "little"), True)
        return self.s.prf(n)

    def challenge_scalar(self, label):
        return int.from_bytes(self.challenge_bytes(label, 64), "little") % L


def generators_chain(label, count):
    """bulletproofs GeneratorsChain: SHAKE256("GeneratorsChain" || label), 64 B per point."""
    xof = hashlib.shake_256(b"GeneratorsChain" + label).digest(64 * count)
    return [from_uniform_bytes(xof[64 * i:64 * i + 64]) for i in range(count)]


def b_blinding():
    return from_uniform_bytes(hashlib.sha3_512(ristretto_encode(BASE)).digest())


if __name__ == "__main__":
    print("D", D)
    print("SQRT_M1", SQRT_M1)
    print("INVSQRT_A_MINUS_D", INVSQRT_A_MINUS_D)
    print("SQRT_AD_MINUS_ONE", SQRT_AD_MINUS_ONE)
    print("B", ristretto_encode(BASE).hex())
    print("B_blinding", ristretto_encode(b_blinding()).hex())
    assert sponge(136, b"abc", 0x06, 32) == hashlib.sha3_256(b"abc").digest()
    assert sponge(136, b"x" * 300, 0x1F, 500) == hashlib.shake_256(b"x" * 300).digest(500)
    assert sponge(72, b"y" * 200, 0x06, 64) == hashlib.sha3_512(b"y" * 200).digest()
    t = Transcript(b"test protocol")
    t.append_message(b"some label", b"some data")
    print("merlin", t.challenge_bytes(b"challenge", 32).hex())
