// Host-only fast curve arithmetic (5 x 51-bit limbs, unsigned __int128) for the strictly sequential tails the
// GPU hands back: the ~256-step Horner chain that combines Pippenger window/bit partial sums, fixed-base
// multiples of B / B_blinding, and the (de)compression of the handful of proof points per round.
// A single GPU lane would need ~0.5 ms for such a chain; a host core needs ~50 us.
#pragma once
#include "fe32.hpp"

namespace rofl {
namespace h51 {

typedef unsigned __int128 u128;
struct fe5 { u64 v[5]; };
struct ge5 { fe5 X, Y, Z, T; };
struct niels5 { fe5 ypx, ymx, t2d; };
static const u64 M51 = 0x7ffffffffffffULL;

inline fe5 from_fe(const fe &a) {   // any representative -> canonical -> 51-bit limbs
    fe c = fe_canon(a);
    u64 t0 = (u64)c.v[0] | ((u64)c.v[1] << 32), t1 = (u64)c.v[2] | ((u64)c.v[3] << 32);
    u64 t2 = (u64)c.v[4] | ((u64)c.v[5] << 32), t3 = (u64)c.v[6] | ((u64)c.v[7] << 32);
    fe5 r;
    r.v[0] = t0 & M51; r.v[1] = ((t0 >> 51) | (t1 << 13)) & M51; r.v[2] = ((t1 >> 38) | (t2 << 26)) & M51;
    r.v[3] = ((t2 >> 25) | (t3 << 39)) & M51; r.v[4] = (t3 >> 12) & M51;
    return r;
}
// the same split without the canonical reduction: any 256-bit representative -> limbs below 2^51 (the top one below 2^52), which
// is all mul / add / sub need.  (The Horner chains convert ~1 000 coordinates per problem; fe_canon was a fifth of their time.)
inline fe5 from_fe_loose(const fe &c) {
    u64 t0 = (u64)c.v[0] | ((u64)c.v[1] << 32), t1 = (u64)c.v[2] | ((u64)c.v[3] << 32);
    u64 t2 = (u64)c.v[4] | ((u64)c.v[5] << 32), t3 = (u64)c.v[6] | ((u64)c.v[7] << 32);
    fe5 r;
    r.v[0] = t0 & M51; r.v[1] = ((t0 >> 51) | (t1 << 13)) & M51; r.v[2] = ((t1 >> 38) | (t2 << 26)) & M51;
    r.v[3] = ((t2 >> 25) | (t3 << 39)) & M51; r.v[4] = t3 >> 12;
    return r;
}
inline void carry(fe5 &h) {
    u64 c;
    c = h.v[0] >> 51; h.v[0] &= M51; h.v[1] += c;
    c = h.v[1] >> 51; h.v[1] &= M51; h.v[2] += c;
    c = h.v[2] >> 51; h.v[2] &= M51; h.v[3] += c;
    c = h.v[3] >> 51; h.v[3] &= M51; h.v[4] += c;
    c = h.v[4] >> 51; h.v[4] &= M51; h.v[0] += c * 19;
}
inline fe to_fe(const fe5 &a) {     // -> canonical 8 x 32
    fe5 t = a; carry(t); carry(t);
    t.v[0] += 19; carry(t);
    t.v[0] += (1ULL << 51) - 19; t.v[1] += (1ULL << 51) - 1; t.v[2] += (1ULL << 51) - 1; t.v[3] += (1ULL << 51) - 1; t.v[4] += (1ULL << 51) - 1;
    u64 c;
    c = t.v[0] >> 51; t.v[0] &= M51; t.v[1] += c;
    c = t.v[1] >> 51; t.v[1] &= M51; t.v[2] += c;
    c = t.v[2] >> 51; t.v[2] &= M51; t.v[3] += c;
    c = t.v[3] >> 51; t.v[3] &= M51; t.v[4] += c;
    t.v[4] &= M51;
    u64 w0 = t.v[0] | (t.v[1] << 51), w1 = (t.v[1] >> 13) | (t.v[2] << 38), w2 = (t.v[2] >> 26) | (t.v[3] << 25), w3 = (t.v[3] >> 39) | (t.v[4] << 12);
    fe r = {{(u32)w0, (u32)(w0 >> 32), (u32)w1, (u32)(w1 >> 32), (u32)w2, (u32)(w2 >> 32), (u32)w3, (u32)(w3 >> 32)}};
    return r;
}
inline fe5 add(const fe5 &f, const fe5 &g) { fe5 h; for (int i = 0; i < 5; i++) h.v[i] = f.v[i] + g.v[i]; carry(h); return h; }
inline fe5 sub(const fe5 &f, const fe5 &g) {
    fe5 h;
    h.v[0] = f.v[0] + 0x1fffffffffffb4ULL - g.v[0];
    for (int i = 1; i < 5; i++) h.v[i] = f.v[i] + 0x1ffffffffffffcULL - g.v[i];
    carry(h); return h;
}
inline fe5 mul(const fe5 &f, const fe5 &g) {
    u128 f0 = f.v[0], f1 = f.v[1], f2 = f.v[2], f3 = f.v[3], f4 = f.v[4];
    u64 g0 = g.v[0], g1 = g.v[1], g2 = g.v[2], g3 = g.v[3], g4 = g.v[4];
    u64 g1_19 = 19 * g1, g2_19 = 19 * g2, g3_19 = 19 * g3, g4_19 = 19 * g4;
    u128 r0 = f0 * g0 + f1 * g4_19 + f2 * g3_19 + f3 * g2_19 + f4 * g1_19;
    u128 r1 = f0 * g1 + f1 * g0 + f2 * g4_19 + f3 * g3_19 + f4 * g2_19;
    u128 r2 = f0 * g2 + f1 * g1 + f2 * g0 + f3 * g4_19 + f4 * g3_19;
    u128 r3 = f0 * g3 + f1 * g2 + f2 * g1 + f3 * g0 + f4 * g4_19;
    u128 r4 = f0 * g4 + f1 * g3 + f2 * g2 + f3 * g1 + f4 * g0;
    fe5 h;
    r1 += (u64)(r0 >> 51); h.v[0] = (u64)r0 & M51;
    r2 += (u64)(r1 >> 51); h.v[1] = (u64)r1 & M51;
    r3 += (u64)(r2 >> 51); h.v[2] = (u64)r2 & M51;
    r4 += (u64)(r3 >> 51); h.v[3] = (u64)r3 & M51;
    u64 c = (u64)(r4 >> 51); h.v[4] = (u64)r4 & M51;
    h.v[0] += c * 19; c = h.v[0] >> 51; h.v[0] &= M51; h.v[1] += c;
    return h;
}
inline fe5 sq(const fe5 &f) {
    u128 f0 = f.v[0], f1 = f.v[1], f2 = f.v[2], f3 = f.v[3], f4 = f.v[4];
    u64 f0_2 = 2 * f.v[0], f1_2 = 2 * f.v[1], f3_19 = 19 * f.v[3], f4_19 = 19 * f.v[4];
    u128 r0 = f0 * f.v[0] + (u128)(2 * f.v[1]) * f4_19 + (u128)(2 * f.v[2]) * f3_19;
    u128 r1 = (u128)f0_2 * f.v[1] + (u128)(2 * f.v[2]) * f4_19 + f3 * f3_19;
    u128 r2 = (u128)f0_2 * f.v[2] + f1 * f.v[1] + (u128)(2 * f.v[3]) * f4_19;
    u128 r3 = (u128)f0_2 * f.v[3] + (u128)f1_2 * f.v[2] + f4 * f4_19;
    u128 r4 = (u128)f0_2 * f.v[4] + (u128)f1_2 * f.v[3] + f2 * f.v[2];
    fe5 h;
    r1 += (u64)(r0 >> 51); h.v[0] = (u64)r0 & M51;
    r2 += (u64)(r1 >> 51); h.v[1] = (u64)r1 & M51;
    r3 += (u64)(r2 >> 51); h.v[2] = (u64)r2 & M51;
    r4 += (u64)(r3 >> 51); h.v[3] = (u64)r3 & M51;
    u64 c = (u64)(r4 >> 51); h.v[4] = (u64)r4 & M51;
    h.v[0] += c * 19; c = h.v[0] >> 51; h.v[0] &= M51; h.v[1] += c;
    return h;
}
inline fe5 zero() { fe5 r = {{0, 0, 0, 0, 0}}; return r; }
inline fe5 one() { fe5 r = {{1, 0, 0, 0, 0}}; return r; }
inline fe5 neg(const fe5 &f) { return sub(zero(), f); }
inline fe5 sqn(fe5 a, int n) { for (int i = 0; i < n; i++) a = sq(a); return a; }
inline fe5 pow_2_250_1(const fe5 &z, fe5 &z11) {
    fe5 z2 = sq(z), z9 = mul(sqn(z2, 2), z); z11 = mul(z9, z2);
    fe5 z_5_0 = mul(sq(z11), z9), z_10_0 = mul(sqn(z_5_0, 5), z_5_0), z_20_0 = mul(sqn(z_10_0, 10), z_10_0);
    fe5 z_40_0 = mul(sqn(z_20_0, 20), z_20_0), z_50_0 = mul(sqn(z_40_0, 10), z_10_0), z_100_0 = mul(sqn(z_50_0, 50), z_50_0);
    fe5 z_200_0 = mul(sqn(z_100_0, 100), z_100_0);
    return mul(sqn(z_200_0, 50), z_50_0);
}
inline fe5 invert(const fe5 &z) { fe5 z11; fe5 t = pow_2_250_1(z, z11); return mul(sqn(t, 5), z11); }
inline fe5 pow22523(const fe5 &z) { fe5 z11; fe5 t = pow_2_250_1(z, z11); return mul(sqn(t, 2), z); }
inline bool isneg(const fe5 &a) { return to_fe(a).v[0] & 1; }
inline bool iszero(const fe5 &a) { fe c = to_fe(a); u32 o = 0; for (int i = 0; i < 8; i++) o |= c.v[i]; return o == 0; }
inline bool eq(const fe5 &a, const fe5 &b) { return iszero(sub(a, b)); }
inline fe5 fabs5(const fe5 &a) { return isneg(a) ? neg(a) : a; }

struct Consts { fe5 d2, sqrtm1, invsqrt_a_minus_d; };
inline const Consts &K() { static Consts k = {from_fe(fe_d2()), from_fe(fe_sqrtm1()), from_fe(fe_invsqrt_a_minus_d())}; return k; }

inline bool sqrt_ratio_i(fe5 &out, const fe5 &u, const fe5 &v) {
    fe5 v3 = mul(sq(v), v), v7 = mul(sq(v3), v);
    fe5 r = mul(mul(u, v3), pow22523(mul(u, v7)));
    fe5 check = mul(v, sq(r)), neg_u = neg(u);
    bool correct = eq(check, u), flipped = eq(check, neg_u), flipped_i = eq(check, mul(neg_u, K().sqrtm1));
    if (flipped || flipped_i) r = mul(r, K().sqrtm1);
    out = fabs5(r);
    return correct || flipped;
}

inline ge5 identity() { ge5 r = {zero(), one(), one(), zero()}; return r; }
inline ge5 from_ge(const ge &p) { ge5 r = {from_fe(p.X), from_fe(p.Y), from_fe(p.Z), from_fe(p.T)}; return r; }
inline ge5 from_ge_loose(const ge &p) { ge5 r = {from_fe_loose(p.X), from_fe_loose(p.Y), from_fe_loose(p.Z), from_fe_loose(p.T)}; return r; }
inline ge to_ge(const ge5 &p) { ge r; r.X = to_fe(p.X); r.Y = to_fe(p.Y); r.Z = to_fe(p.Z); r.T = to_fe(p.T); return r; }
inline niels5 from_niels(const niels &q) { niels5 r = {from_fe(q.ypx), from_fe(q.ymx), from_fe(q.t2d)}; return r; }
inline ge5 gadd(const ge5 &p, const ge5 &q) {
    fe5 A = mul(sub(p.Y, p.X), sub(q.Y, q.X)), B = mul(add(p.Y, p.X), add(q.Y, q.X));
    fe5 C = mul(mul(p.T, q.T), K().d2), D = mul(p.Z, q.Z); D = add(D, D);
    fe5 E = sub(B, A), F = sub(D, C), G = add(D, C), H = add(B, A);
    ge5 r = {mul(E, F), mul(G, H), mul(F, G), mul(E, H)};
    return r;
}
inline ge5 gmadd(const ge5 &p, const niels5 &q, bool negq) {
    fe5 A = mul(sub(p.Y, p.X), negq ? q.ypx : q.ymx), B = mul(add(p.Y, p.X), negq ? q.ymx : q.ypx);
    fe5 C = mul(p.T, q.t2d), D = add(p.Z, p.Z);
    fe5 E = sub(B, A), H = add(B, A), F = negq ? add(D, C) : sub(D, C), G = negq ? sub(D, C) : add(D, C);
    ge5 r = {mul(E, F), mul(G, H), mul(F, G), mul(E, H)};
    return r;
}
inline ge5 gdouble(const ge5 &p) {
    fe5 A = sq(p.X), B = sq(p.Y), C = sq(p.Z); C = add(C, C);
    fe5 E = sub(sub(sq(add(p.X, p.Y)), A), B), G = sub(B, A), F = sub(G, C), H = neg(add(A, B));
    ge5 r = {mul(E, F), mul(G, H), mul(F, G), mul(E, H)};
    return r;
}
inline bool is_identity_ristretto(const ge5 &p) { return iszero(p.X) || iszero(p.Y); }
inline void encode(uint8_t *s, const ge5 &p) {
    fe5 u1 = mul(add(p.Z, p.Y), sub(p.Z, p.Y)), u2 = mul(p.X, p.Y);
    fe5 invsqrt; sqrt_ratio_i(invsqrt, one(), mul(u1, sq(u2)));
    fe5 den1 = mul(invsqrt, u1), den2 = mul(invsqrt, u2), z_inv = mul(mul(den1, den2), p.T);
    fe5 ix0 = mul(p.X, K().sqrtm1), iy0 = mul(p.Y, K().sqrtm1), ench = mul(den1, K().invsqrt_a_minus_d);
    bool rotate = isneg(mul(p.T, z_inv));
    fe5 x = rotate ? iy0 : p.X, y = rotate ? ix0 : p.Y, den_inv = rotate ? ench : den2;
    if (isneg(mul(x, z_inv))) y = neg(y);
    fe_tobytes(s, to_fe(fabs5(mul(den_inv, sub(p.Z, y)))));
}
inline niels5 to_niels5(const ge5 &p) {
    fe5 zi = invert(p.Z), x = mul(p.X, zi), y = mul(p.Y, zi);
    niels5 r = {add(y, x), sub(y, x), mul(mul(x, y), K().d2)};
    return r;
}
inline niels to_niels32(const ge5 &p) { niels5 q = to_niels5(p); niels r; r.ypx = to_fe(q.ypx); r.ymx = to_fe(q.ymx); r.t2d = to_fe(q.t2d); return r; }


// ---------------------------------------------------------------- scalars mod l on the host: 4 x 64-bit Montgomery (R = 2^256, the same
// Montgomery form as the 8 x 32-bit device code, so values pass between the two unchanged).  The per-round challenge inversion
// u -> u^-1 of the inner-product argument sits on the critical path of every host hop: 253 squarings + 63 multiplications here
// (~6 us) instead of ~380 of the portable 8 x 32 routine (~60 us).
struct s4 { u64 v[4]; };
inline const u64 *l64() { static const u64 L[4] = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0ULL, 0x1000000000000000ULL}; return L; }
inline u64 linv64() { static const u64 k = [] { u64 inv = 1; for (int i = 0; i < 7; i++) inv *= 2 - l64()[0] * inv; return (u64)0 - inv; }(); return k; }
inline s4 s4_from(const sc &a) { s4 r; for (int i = 0; i < 4; i++) r.v[i] = (u64)a.v[2 * i] | ((u64)a.v[2 * i + 1] << 32); return r; }
inline sc s4_to(const s4 &a) { sc r; for (int i = 0; i < 4; i++) { r.v[2 * i] = (u32)a.v[i]; r.v[2 * i + 1] = (u32)(a.v[i] >> 32); } return r; }
inline s4 s4_montmul(const s4 &a, const s4 &b) {
    const u64 *L = l64(); const u64 li = linv64();
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u64 carry = 0;
        for (int j = 0; j < 4; j++) { u128 x = (u128)a.v[j] * b.v[i] + t[j] + carry; t[j] = (u64)x; carry = (u64)(x >> 64); }
        u128 y = (u128)t[4] + carry; t[4] = (u64)y; t[5] = (u64)(y >> 64);
        u64 m = t[0] * li;
        u128 x = (u128)m * L[0] + t[0]; carry = (u64)(x >> 64);
        for (int j = 1; j < 4; j++) { x = (u128)m * L[j] + t[j] + carry; t[j - 1] = (u64)x; carry = (u64)(x >> 64); }
        y = (u128)t[4] + carry; t[3] = (u64)y; t[4] = t[5] + (u64)(y >> 64); t[5] = 0;
    }
    bool ge_l = t[4] != 0;
    if (!ge_l) { ge_l = true; for (int i = 3; i >= 0; i--) { if (t[i] > L[i]) break; if (t[i] < L[i]) { ge_l = false; break; } } }
    s4 r;
    if (ge_l) { u64 bw = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)t[i] - L[i] - bw; r.v[i] = (u64)d; bw = (u64)(d >> 64) & 1; } }
    else for (int i = 0; i < 4; i++) r.v[i] = t[i];
    return r;
}
// a^(l-2), Montgomery in / out; fixed 4-bit windows over the public exponent (variable time in the exponent only)
inline sc sc_invert_mont_ladder(const sc &a_mont) {
    static const u64 E[4] = {0x5812631a5cf5d3edULL - 2, 0x14def9dea2f79cd6ULL, 0ULL, 0x1000000000000000ULL};
    s4 tab[16]; tab[1] = s4_from(a_mont); tab[0] = s4_from(sc_one_mont());
    for (int i = 2; i < 16; i++) tab[i] = s4_montmul(tab[i - 1], tab[1]);
    s4 acc = tab[(E[3] >> 60) & 15];
    for (int nib = 62; nib >= 0; nib--) {
        acc = s4_montmul(acc, acc); acc = s4_montmul(acc, acc); acc = s4_montmul(acc, acc); acc = s4_montmul(acc, acc);
        u32 d = (u32)(E[nib >> 4] >> ((nib & 15) * 4)) & 15;
        if (d) acc = s4_montmul(acc, tab[d]);
    }
    return s4_to(acc);
}

// ---- x^-1 mod l by Bernstein-Yang division steps ("safegcd", variable time: the inputs are public challenges).
// The IPP hop inverts one challenge (or one product of four) per round: 252 squarings + ~60 multiplications took 10 us of an 0.2 ms hop.
// Here: batches of 62 division steps on the low 64 bits of (f, g) give a 2 x 2 transition matrix t with [f', g'] = t [f, g] / 2^62; the
// same matrix is applied to (d, e) modulo l (a multiple of l makes the sums divisible by 2^62), keeping d x = f, e x = g (mod l) up to
// the accumulated power of two.  When g = 0, f = +-1 and d = +-x^-1.  Numbers are signed, five limbs of 62 bits.
namespace gcd62 {
typedef __int128 i128;
struct s62 { int64_t v[5]; };
static const int64_t M62 = (int64_t)((1ULL << 62) - 1);
// l = 2^252 + 27742317777372353535851937790883648493 in 62-bit limbs, and l^-1 mod 2^62
inline s62 mod_l() {
    const unsigned __int128 lo = ((unsigned __int128)0x14def9dea2f79cd6ULL << 64) | 0x5812631a5cf5d3edULL;      // low 128 bits of l
    s62 r; r.v[0] = (int64_t)((u64)lo & (u64)M62); r.v[1] = (int64_t)((u64)(lo >> 62) & (u64)M62); r.v[2] = (int64_t)((u64)(lo >> 124));
    r.v[3] = 0; r.v[4] = (int64_t)1 << (252 - 248);      // 2^252 = 2^(4 * 62 + 4)
    return r;
}
inline u64 inv62(u64 a) { u64 x = a; for (int i = 0; i < 6; i++) x *= 2 - a * x; return x & (u64)M62; }      // a odd: a x = 1 (mod 2^62)
struct trans { int64_t u, v, q, r; };
// up to 62 division steps on (f0, g0) = the low bits of f (odd) and g; eta = -delta.  Returns the new eta.
inline int64_t divsteps62(int64_t eta, u64 f0, u64 g0, trans &t) {
    u64 u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
    int i = 62;
    for (;;) {
        int zeros = __builtin_ctzll(g | (~0ULL << i));      // trailing zeros of g, at most i
        g >>= zeros; u <<= zeros; v <<= zeros; eta -= zeros; i -= zeros;
        if (i == 0) break;
        if (eta < 0) { u64 tmp; eta = -eta; tmp = f; f = g; g = 0 - tmp; tmp = u; u = q; q = 0 - tmp; tmp = v; v = r; r = 0 - tmp; }
        g += f; q += u; r += v;                               // f, g odd: g + f is even
    }
    t.u = (int64_t)u; t.v = (int64_t)v; t.q = (int64_t)q; t.r = (int64_t)r;
    return eta;
}
// (a, b) <- t (a, b) / 2^62, exact (f, g)
inline void update_fg(s62 &f, s62 &g, const trans &t) {
    i128 cf = (i128)t.u * f.v[0] + (i128)t.v * g.v[0], cg = (i128)t.q * f.v[0] + (i128)t.r * g.v[0];
    cf >>= 62; cg >>= 62;                                      // the low 62 bits are zero by construction
    for (int i = 1; i < 5; i++) {
        cf += (i128)t.u * f.v[i] + (i128)t.v * g.v[i]; cg += (i128)t.q * f.v[i] + (i128)t.r * g.v[i];
        f.v[i - 1] = (int64_t)cf & M62; cf >>= 62; g.v[i - 1] = (int64_t)cg & M62; cg >>= 62;
    }
    f.v[4] = (int64_t)cf; g.v[4] = (int64_t)cg;
}
// (d, e) <- t (d, e) / 2^62 mod l: md, me multiples of l make the sums divisible by 2^62
inline void update_de(s62 &d, s62 &e, const trans &t, const s62 &L, u64 linv) {
    i128 cd = (i128)t.u * d.v[0] + (i128)t.v * e.v[0], ce = (i128)t.q * d.v[0] + (i128)t.r * e.v[0];
    const int64_t md = (int64_t)((0 - (u64)cd * linv) & (u64)M62), me = (int64_t)((0 - (u64)ce * linv) & (u64)M62);
    cd += (i128)md * L.v[0]; ce += (i128)me * L.v[0];
    cd >>= 62; ce >>= 62;
    for (int i = 1; i < 5; i++) {
        cd += (i128)t.u * d.v[i] + (i128)t.v * e.v[i] + (i128)md * L.v[i]; ce += (i128)t.q * d.v[i] + (i128)t.r * e.v[i] + (i128)me * L.v[i];
        d.v[i - 1] = (int64_t)cd & M62; cd >>= 62; e.v[i - 1] = (int64_t)ce & M62; ce >>= 62;
    }
    d.v[4] = (int64_t)cd; e.v[4] = (int64_t)ce;
}
inline bool is_zero(const s62 &a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3] | a.v[4]) == 0; }
// x (canonical, 0 < x < l) -> x^-1 mod l (canonical); 0 -> 0
inline void invert(u64 out[4], const u64 x[4]) {
    static const s62 L = mod_l();
    static const u64 linv = inv62((u64)L.v[0]);
    s62 f = L, g, d = {{0, 0, 0, 0, 0}}, e = {{1, 0, 0, 0, 0}};
    g.v[0] = (int64_t)(x[0] & (u64)M62); g.v[1] = (int64_t)(((x[0] >> 62) | (x[1] << 2)) & (u64)M62); g.v[2] = (int64_t)(((x[1] >> 60) | (x[2] << 4)) & (u64)M62);
    g.v[3] = (int64_t)(((x[2] >> 58) | (x[3] << 6)) & (u64)M62); g.v[4] = (int64_t)(x[3] >> 56);
    int64_t eta = -1;
    for (int it = 0; it < 16 && !is_zero(g); it++) {          // 12 batches cover the 735 steps the analysis allows for 256-bit inputs
        trans t;
        eta = divsteps62(eta, (u64)f.v[0] | ((u64)f.v[1] << 62), (u64)g.v[0] | ((u64)g.v[1] << 62), t);
        update_de(d, e, t, L, linv);
        update_fg(f, g, t);
    }
    // f = +-1; d = +-x^-1 + k l with a small k: bring it into [0, l)
    const bool neg = f.v[4] < 0 || (f.v[4] == 0 && f.v[3] == 0 && f.v[2] == 0 && f.v[1] == 0 && f.v[0] == 0);      // (f is -1 when its top limb is negative)
    if (f.v[4] < 0) for (int i = 0; i < 5; i++) d.v[i] = -d.v[i];
    (void)neg;
    // normalise the limbs (they may be negative after the negation), then add / subtract l until 0 <= d < l
    for (int pass = 0; pass < 2; pass++) {
        int64_t c = 0;
        for (int i = 0; i < 4; i++) { int64_t w = d.v[i] + c; d.v[i] = w & M62; c = w >> 62; }
        d.v[4] += c;
    }
    auto add_l = [&](int sign) { int64_t c = 0; for (int i = 0; i < 5; i++) { int64_t w = d.v[i] + sign * L.v[i] + c; if (i < 4) { d.v[i] = w & M62; c = w >> 62; } else d.v[i] = w; } };
    auto geq_l = [&]() { for (int i = 4; i >= 0; i--) { if (d.v[i] != L.v[i]) return d.v[i] > L.v[i]; } return true; };
    for (int k = 0; k < 64 && d.v[4] < 0; k++) add_l(+1);
    for (int k = 0; k < 64 && geq_l(); k++) add_l(-1);
    const unsigned __int128 lo = (unsigned __int128)(u64)d.v[0] | ((unsigned __int128)(u64)d.v[1] << 62) | ((unsigned __int128)(u64)d.v[2] << 124);
    out[0] = (u64)lo; out[1] = (u64)(lo >> 64);
    const unsigned __int128 hi = ((unsigned __int128)(u64)d.v[2] >> 4) | ((unsigned __int128)(u64)d.v[3] << 58) | ((unsigned __int128)(u64)d.v[4] << 120);
    out[2] = (u64)hi; out[3] = (u64)(hi >> 64);
}
}  // namespace gcd62

// Montgomery in / out: (a R)^-1 = a^-1 R^-1, times R^2 twice (sc_to_mont multiplies by R)
inline sc sc_invert_mont_fast(const sc &a_mont) {
    u64 x[4], y[4];
    for (int i = 0; i < 4; i++) x[i] = (u64)a_mont.v[2 * i] | ((u64)a_mont.v[2 * i + 1] << 32);
    gcd62::invert(y, x);
    sc r; for (int i = 0; i < 4; i++) { r.v[2 * i] = (u32)y[i]; r.v[2 * i + 1] = (u32)(y[i] >> 32); }
    return sc_to_mont(sc_to_mont(r));
}

}  // namespace h51
}  // namespace rofl
