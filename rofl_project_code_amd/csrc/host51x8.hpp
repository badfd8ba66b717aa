// Host point arithmetic on EIGHT independent problems at a time: AVX-512 IFMA (vpmadd52luq / vpmadd52huq), one problem per 64-bit lane,
// field elements in the same 5 x 51-bit limbs as host51.hpp.
//
// Why: with many chunks per client (n_partition = 64: 128 L / R problems per IPP round) the window combination of an MSM -- a 253-step
// doubling chain per problem -- ran on the device (k_msm_horner): 0.30 ms per round whatever the number of problems, because a chain is as
// deep as its 246 doublings at ~1.2 us each on a quad of lanes.  A host core doubles a point in 79 ns, but 128 chains are 3 ms of scalar CPU
// time per round.  Eight chains per instruction stream bring that to ~0.6 ms of CPU time, i.e. ~40 us on the pool.  The device then only adds
// up each window's bit-sums (k_msm_wsum) and the chains come back to the host.  Used when the CPU has AVX-512 IFMA (run-time check) and a
// launch carries >= 16 problems; k_msm_horner stays as the fallback.
//
// Bounds: every operand of mul() has limbs < 2^52 (IFMA reads 52 bits); add / sub / mul all end with a carry pass that leaves limbs
// < 2^51 + 2^14.  Column sums in mul: <= 5 low halves (< 2^52 each) plus twice <= 5 high halves, < 2^56; after the 19-fold < 2^61.
#pragma once
#include <immintrin.h>
#include <string.h>
#include "host51.hpp"

namespace rofl {
namespace h8 {

#define ROFL_H8 __attribute__((target("avx512f,avx512ifma,avx512dq,avx512vl"))) inline

typedef __m512i V;
struct fe8 { V v[5]; };
struct ge8 { fe8 X, Y, Z, T; };

inline bool available() {
    static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512ifma") && __builtin_cpu_supports("avx512dq");
    return ok;
}

ROFL_H8 V bc(u64 x) { return _mm512_set1_epi64((long long)x); }
ROFL_H8 void carry(fe8 &h) {
    const V M = bc(h51::M51);
    V c;
    c = _mm512_srli_epi64(h.v[0], 51); h.v[0] = _mm512_and_si512(h.v[0], M); h.v[1] = _mm512_add_epi64(h.v[1], c);
    c = _mm512_srli_epi64(h.v[1], 51); h.v[1] = _mm512_and_si512(h.v[1], M); h.v[2] = _mm512_add_epi64(h.v[2], c);
    c = _mm512_srli_epi64(h.v[2], 51); h.v[2] = _mm512_and_si512(h.v[2], M); h.v[3] = _mm512_add_epi64(h.v[3], c);
    c = _mm512_srli_epi64(h.v[3], 51); h.v[3] = _mm512_and_si512(h.v[3], M); h.v[4] = _mm512_add_epi64(h.v[4], c);
    c = _mm512_srli_epi64(h.v[4], 51); h.v[4] = _mm512_and_si512(h.v[4], M);
    // c * 19 = (c << 4) + (c << 1) + c
    V c19 = _mm512_add_epi64(_mm512_add_epi64(_mm512_slli_epi64(c, 4), _mm512_slli_epi64(c, 1)), c);
    h.v[0] = _mm512_add_epi64(h.v[0], c19);
    c = _mm512_srli_epi64(h.v[0], 51); h.v[0] = _mm512_and_si512(h.v[0], M); h.v[1] = _mm512_add_epi64(h.v[1], c);
}
ROFL_H8 fe8 add(const fe8 &a, const fe8 &b) { fe8 r; for (int i = 0; i < 5; i++) r.v[i] = _mm512_add_epi64(a.v[i], b.v[i]); carry(r); return r; }
ROFL_H8 fe8 sub(const fe8 &a, const fe8 &b) {      // a + 4p - b
    fe8 r;
    r.v[0] = _mm512_sub_epi64(_mm512_add_epi64(a.v[0], bc(0x1fffffffffffb4ULL)), b.v[0]);
    for (int i = 1; i < 5; i++) r.v[i] = _mm512_sub_epi64(_mm512_add_epi64(a.v[i], bc(0x1ffffffffffffcULL)), b.v[i]);
    carry(r); return r;
}
ROFL_H8 fe8 mul(const fe8 &a, const fe8 &b) {
    const V Z = _mm512_setzero_si512();
    V lo[9], hi[9];
    for (int k = 0; k < 9; k++) { lo[k] = Z; hi[k] = Z; }
#pragma GCC unroll 5
    for (int i = 0; i < 5; i++)
#pragma GCC unroll 5
        for (int j = 0; j < 5; j++) {
            lo[i + j] = _mm512_madd52lo_epu64(lo[i + j], a.v[i], b.v[j]);
            hi[i + j] = _mm512_madd52hi_epu64(hi[i + j], a.v[i], b.v[j]);
        }
    // column c (weight 2^(51 c)) = lo[c] + 2 hi[c - 1]: the high half of a product sits at 2^52 = 2 * 2^51 of the next column
    V col[10];
    col[0] = lo[0];
    for (int c = 1; c < 9; c++) col[c] = _mm512_add_epi64(lo[c], _mm512_slli_epi64(hi[c - 1], 1));
    col[9] = _mm512_slli_epi64(hi[8], 1);
    fe8 r;
    for (int c = 0; c < 5; c++) {      // 2^255 = 19
        V x = col[c + 5];
        V x19 = _mm512_add_epi64(_mm512_add_epi64(_mm512_slli_epi64(x, 4), _mm512_slli_epi64(x, 1)), x);
        r.v[c] = _mm512_add_epi64(col[c], x19);
    }
    carry(r); return r;
}
// a^2: the ten cross products a_i a_j (i < j) once, doubled in the columns; five squares
ROFL_H8 fe8 sq(const fe8 &a) {
    const V Z = _mm512_setzero_si512();
    V lo[9], hi[9], xl[9], xh[9];
    for (int k = 0; k < 9; k++) { lo[k] = Z; hi[k] = Z; xl[k] = Z; xh[k] = Z; }
#pragma GCC unroll 5
    for (int i = 0; i < 5; i++) {
        lo[2 * i] = _mm512_madd52lo_epu64(lo[2 * i], a.v[i], a.v[i]);
        hi[2 * i] = _mm512_madd52hi_epu64(hi[2 * i], a.v[i], a.v[i]);
#pragma GCC unroll 5
        for (int j = i + 1; j < 5; j++) {
            xl[i + j] = _mm512_madd52lo_epu64(xl[i + j], a.v[i], a.v[j]);
            xh[i + j] = _mm512_madd52hi_epu64(xh[i + j], a.v[i], a.v[j]);
        }
    }
    V col[10];
    for (int c = 0; c < 10; c++) {
        V x = c < 9 ? _mm512_add_epi64(lo[c], _mm512_slli_epi64(xl[c], 1)) : Z;                                     // lo + 2 cross_lo
        if (c > 0) x = _mm512_add_epi64(x, _mm512_add_epi64(_mm512_slli_epi64(hi[c - 1], 1), _mm512_slli_epi64(xh[c - 1], 2)));      // + 2 hi + 4 cross_hi
        col[c] = x;
    }
    fe8 r;
    for (int c = 0; c < 5; c++) {
        V x = col[c + 5];
        V x19 = _mm512_add_epi64(_mm512_add_epi64(_mm512_slli_epi64(x, 4), _mm512_slli_epi64(x, 1)), x);
        r.v[c] = _mm512_add_epi64(col[c], x19);
    }
    carry(r); return r;
}
ROFL_H8 fe8 zero() { fe8 r; for (int i = 0; i < 5; i++) r.v[i] = _mm512_setzero_si512(); return r; }
ROFL_H8 fe8 neg(const fe8 &a) { return sub(zero(), a); }
ROFL_H8 fe8 bcast(const h51::fe5 &a) { fe8 r; for (int i = 0; i < 5; i++) r.v[i] = bc(a.v[i]); return r; }

// the formulas of host51.hpp (extended coordinates, a = -1), operand for operand
ROFL_H8 ge8 gadd(const ge8 &p, const ge8 &q, const fe8 &d2) {
    fe8 A = mul(sub(p.Y, p.X), sub(q.Y, q.X)), B = mul(add(p.Y, p.X), add(q.Y, q.X));
    fe8 C = mul(mul(p.T, q.T), d2), D = mul(p.Z, q.Z); D = add(D, D);
    fe8 E = sub(B, A), F = sub(D, C), G = add(D, C), H = add(B, A);
    ge8 r = {mul(E, F), mul(G, H), mul(F, G), mul(E, H)};
    return r;
}
ROFL_H8 ge8 gdouble(const ge8 &p) {
    fe8 A = sq(p.X), B = sq(p.Y), C = sq(p.Z); C = add(C, C);
    fe8 E = sub(sub(sq(add(p.X, p.Y)), A), B), G = sub(B, A), F = sub(G, C), H = neg(add(A, B));
    ge8 r = {mul(E, F), mul(G, H), mul(F, G), mul(E, H)};
    return r;
}

// lane l <- point pts[l]  (8 x 32-bit saturated coordinates, any representative below 2^256)
ROFL_H8 ge8 load8(const ge *const pts[8]) {
    alignas(64) u64 t[4][5][8];
    for (int l = 0; l < 8; l++) {
        const fe *co[4] = {&pts[l]->X, &pts[l]->Y, &pts[l]->Z, &pts[l]->T};
        for (int c = 0; c < 4; c++) { h51::fe5 f = h51::from_fe_loose(*co[c]); for (int i = 0; i < 5; i++) t[c][i][l] = f.v[i]; }
    }
    ge8 r; fe8 *dst[4] = {&r.X, &r.Y, &r.Z, &r.T};
    for (int c = 0; c < 4; c++) { for (int i = 0; i < 5; i++) dst[c]->v[i] = _mm512_load_si512((const void *)t[c][i]); carry(*dst[c]); }      // loose top limb (< 2^52 + ...) -> tight
    return r;
}
ROFL_H8 void store8(h51::ge5 out[8], const ge8 &p) {
    alignas(64) u64 t[4][5][8];
    const fe8 *src[4] = {&p.X, &p.Y, &p.Z, &p.T};
    for (int c = 0; c < 4; c++) for (int i = 0; i < 5; i++) _mm512_store_si512((void *)t[c][i], src[c]->v[i]);
    for (int l = 0; l < 8; l++) {
        h51::fe5 *dst[4] = {&out[l].X, &out[l].Y, &out[l].Z, &out[l].T};
        for (int c = 0; c < 4; c++) for (int i = 0; i < 5; i++) dst[c]->v[i] = t[c][i][l];
    }
}

// Eight window chains at once: result[l] = sum_w 2^(pos[w]) W[l][w] for W windows whose positions pos[0] = 0 < pos[1] < ... are the same for
// all lanes; wsum(l, w) points at lane l's window sum.  Lanes beyond `lanes` repeat lane 0 (their results are ignored by the caller).
template <class GetPt>
ROFL_H8 void horner8(h51::ge5 out[8], int lanes, int W, const u32 *pos, GetPt wsum) {
    const fe8 d2 = bcast(h51::K().d2);
    const ge *p8[8];
    for (int l = 0; l < 8; l++) p8[l] = wsum(l < lanes ? l : 0, W - 1);
    ge8 acc = load8(p8);
    for (int w = W - 2; w >= 0; w--) {
        for (u32 i = pos[w]; i < pos[w + 1]; i++) acc = gdouble(acc);
        for (int l = 0; l < 8; l++) p8[l] = wsum(l < lanes ? l : 0, w);
        acc = gadd(acc, load8(p8), d2);
    }
    store8(out, acc);
}

// ---- Ristretto encodings eight at a time (h51::encode, operand for operand; the lane-wise choices become mask blends)
typedef __mmask8 M8;
// canonical limbs (value in [0, p), every limb below 2^51): the steps of h51::to_fe
ROFL_H8 fe8 freeze(fe8 t) {
    const V M = bc(h51::M51);
    carry(t); carry(t);
    t.v[0] = _mm512_add_epi64(t.v[0], bc(19)); carry(t);                                // (h mod p) + 19, below 2^255
    t.v[0] = _mm512_add_epi64(t.v[0], bc((1ULL << 51) - 19));                           // + p = (h mod p) + 2^255
    for (int i = 1; i < 5; i++) t.v[i] = _mm512_add_epi64(t.v[i], bc((1ULL << 51) - 1));
    V c;
    c = _mm512_srli_epi64(t.v[0], 51); t.v[0] = _mm512_and_si512(t.v[0], M); t.v[1] = _mm512_add_epi64(t.v[1], c);
    c = _mm512_srli_epi64(t.v[1], 51); t.v[1] = _mm512_and_si512(t.v[1], M); t.v[2] = _mm512_add_epi64(t.v[2], c);
    c = _mm512_srli_epi64(t.v[2], 51); t.v[2] = _mm512_and_si512(t.v[2], M); t.v[3] = _mm512_add_epi64(t.v[3], c);
    c = _mm512_srli_epi64(t.v[3], 51); t.v[3] = _mm512_and_si512(t.v[3], M); t.v[4] = _mm512_add_epi64(t.v[4], c);
    t.v[4] = _mm512_and_si512(t.v[4], M);                                               // drops the 2^255
    return t;
}
ROFL_H8 M8 isneg(const fe8 &a) { fe8 c = freeze(a); return _mm512_test_epi64_mask(c.v[0], bc(1)); }
ROFL_H8 M8 iszero(const fe8 &a) {
    fe8 c = freeze(a);
    V o = _mm512_or_si512(_mm512_or_si512(c.v[0], c.v[1]), _mm512_or_si512(_mm512_or_si512(c.v[2], c.v[3]), c.v[4]));
    return _mm512_testn_epi64_mask(o, o);
}
ROFL_H8 M8 eq(const fe8 &a, const fe8 &b) { return iszero(sub(a, b)); }
ROFL_H8 fe8 blend(M8 pick_b, const fe8 &a, const fe8 &b) { fe8 r; for (int i = 0; i < 5; i++) r.v[i] = _mm512_mask_blend_epi64(pick_b, a.v[i], b.v[i]); return r; }
ROFL_H8 fe8 fabs8(const fe8 &a) { return blend(isneg(a), a, neg(a)); }
ROFL_H8 fe8 sqn(fe8 a, int n) { for (int i = 0; i < n; i++) a = sq(a); return a; }
ROFL_H8 fe8 one() { fe8 r = zero(); r.v[0] = bc(1); return r; }
ROFL_H8 fe8 pow22523(const fe8 &z) {      // z^((p - 5) / 8): h51::pow_2_250_1, then two squarings and z
    fe8 z2 = sq(z), z9 = mul(sqn(z2, 2), z), z11 = mul(z9, z2);
    fe8 z_5_0 = mul(sq(z11), z9), z_10_0 = mul(sqn(z_5_0, 5), z_5_0), z_20_0 = mul(sqn(z_10_0, 10), z_10_0);
    fe8 z_40_0 = mul(sqn(z_20_0, 20), z_20_0), z_50_0 = mul(sqn(z_40_0, 10), z_10_0), z_100_0 = mul(sqn(z_50_0, 50), z_50_0);
    fe8 z_200_0 = mul(sqn(z_100_0, 100), z_100_0);
    fe8 t = mul(sqn(z_200_0, 50), z_50_0);
    return mul(sqn(t, 2), z);
}
// 1 / sqrt(v) as h51::sqrt_ratio_i(out, 1, v) leaves it (the flag is not needed by the encoding)
ROFL_H8 fe8 invsqrt(const fe8 &v, const fe8 &sqrtm1) {
    fe8 v3 = mul(sq(v), v), v7 = mul(sq(v3), v);
    fe8 r = mul(v3, pow22523(v7));
    fe8 check = mul(v, sq(r)), neg_one = neg(one());
    M8 flipped = eq(check, neg_one), flipped_i = eq(check, mul(neg_one, sqrtm1));
    r = blend((M8)(flipped | flipped_i), r, mul(r, sqrtm1));
    return fabs8(r);
}
ROFL_H8 ge8 load8(const h51::ge5 pts[8]) {
    alignas(64) u64 t[4][5][8];
    for (int l = 0; l < 8; l++) {
        const h51::fe5 *co[4] = {&pts[l].X, &pts[l].Y, &pts[l].Z, &pts[l].T};
        for (int c = 0; c < 4; c++) for (int i = 0; i < 5; i++) t[c][i][l] = co[c]->v[i];
    }
    ge8 r; fe8 *dst[4] = {&r.X, &r.Y, &r.Z, &r.T};
    for (int c = 0; c < 4; c++) { for (int i = 0; i < 5; i++) dst[c]->v[i] = _mm512_load_si512((const void *)t[c][i]); carry(*dst[c]); }
    return r;
}
// out[l] = the canonical Ristretto encoding of pts[l] (limbs below 2^52 on entry, as every h51 routine leaves them)
ROFL_H8 void encode8(uint8_t out[8][32], const h51::ge5 pts[8]) {
    const fe8 sqrtm1 = bcast(h51::K().sqrtm1), iamd = bcast(h51::K().invsqrt_a_minus_d);
    ge8 p = load8(pts);
    fe8 u1 = mul(add(p.Z, p.Y), sub(p.Z, p.Y)), u2 = mul(p.X, p.Y);
    fe8 is = invsqrt(mul(u1, sq(u2)), sqrtm1);
    fe8 den1 = mul(is, u1), den2 = mul(is, u2), z_inv = mul(mul(den1, den2), p.T);
    fe8 ix0 = mul(p.X, sqrtm1), iy0 = mul(p.Y, sqrtm1), ench = mul(den1, iamd);
    M8 rotate = isneg(mul(p.T, z_inv));
    fe8 x = blend(rotate, p.X, iy0), y = blend(rotate, p.Y, ix0), den_inv = blend(rotate, den2, ench);
    y = blend(isneg(mul(x, z_inv)), y, neg(y));
    fe8 s = freeze(fabs8(mul(den_inv, sub(p.Z, y))));
    alignas(64) u64 t[5][8];
    for (int i = 0; i < 5; i++) _mm512_store_si512((void *)t[i], s.v[i]);
    for (int l = 0; l < 8; l++) {
        u64 w[4] = {t[0][l] | (t[1][l] << 51), (t[1][l] >> 13) | (t[2][l] << 38), (t[2][l] >> 26) | (t[3][l] << 25), (t[3][l] >> 39) | (t[4][l] << 12)};
        memcpy(out[l], w, 32);
    }
}

}  // namespace h8
}  // namespace rofl
