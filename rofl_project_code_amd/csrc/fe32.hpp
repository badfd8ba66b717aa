// GF(2^255-19), scalars mod l and Edwards/Ristretto255 points for gfx950.
//
// Representation: 8 x 32-bit saturated limbs.  A field element is any value in [0, 2^256)
// congruent to the class mod p = 2^255-19 (2^256 == 38 mod p); canonical reduction happens only at
// encode / compare.  32x32->64 multiply-adds map to v_mad_u64_u32 on CDNA4 -- this path is 255-bit
// integer work, no MFMA.
//
// Everything is __host__ __device__ so the CPU test-suite can exercise the exact same source
// (rofl_dbg_host_* entry points) without a GPU.
//
// Replaces (un-vendored, reference Cargo.lock:363-366) curve25519-dalek-ng 4.1.1
// FieldElement / Scalar / EdwardsPoint / RistrettoPoint as used from
// rofl_crypto/src/range_proof_vec/mod.rs:1-6 and pedersen_ops.rs:1-3.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HD __host__ __device__ __forceinline__
#define HDN __host__ __device__
#else
#define HD inline
#define HDN
#endif

namespace rofl {

typedef uint32_t u32;
typedef uint64_t u64;

struct fe { u32 v[8]; };
struct sc { u32 v[8]; };              // context decides: canonical or Montgomery form
struct ge { fe X, Y, Z, T; };          // extended coordinates, a = -1
struct niels { fe ypx, ymx, t2d; };    // affine (Z = 1): y+x, y-x, 2d*x*y

// ---------------------------------------------------------------- constants
#define FE_CONST(name, a0, a1, a2, a3, a4, a5, a6, a7) \
    HD fe name() { fe r = {{a0, a1, a2, a3, a4, a5, a6, a7}}; return r; }
FE_CONST(fe_d, 0x135978a3u, 0x75eb4dcau, 0x4141d8abu, 0x00700a4du, 0x7779e898u, 0x8cc74079u, 0x2b6ffe73u, 0x52036ceeu)
FE_CONST(fe_d2, 0x26b2f159u, 0xebd69b94u, 0x8283b156u, 0x00e0149au, 0xeef3d130u, 0x198e80f2u, 0x56dffce7u, 0x2406d9dcu)
FE_CONST(fe_sqrtm1, 0x4a0ea0b0u, 0xc4ee1b27u, 0xad2fe478u, 0x2f431806u, 0x3dfbd7a7u, 0x2b4d0099u, 0x4fc1df0bu, 0x2b832480u)
FE_CONST(fe_invsqrt_a_minus_d, 0x805d40eau, 0x99c8fdaau, 0x5a4172beu, 0x9d2f1617u, 0xfe01d840u, 0x16c27b91u, 0xcfaffca2u, 0x786c8905u)
FE_CONST(fe_sqrt_ad_minus_one, 0x497b2e1bu, 0x7e97f6a0u, 0x1b7854bdu, 0xaf9d8e0cu, 0x31f5d1fdu, 0x0f3cfcc9u, 0x2b8348acu, 0x376931bfu)
FE_CONST(fe_one_minus_d_sq, 0x945fc176u, 0xe27c09c1u, 0xcd5e350fu, 0x2c81a138u, 0xbe70dfe4u, 0x9994abddu, 0xb2b3e0d7u, 0x029072a8u)
FE_CONST(fe_d_minus_one_sq, 0x44ed4d20u, 0x31ad5aaau, 0xb01e1999u, 0xd29e4a2cu, 0x529b4eebu, 0x4cdcd32fu, 0xf66c2241u, 0x5968b37au)

HD fe fe_zero() { fe r = {{0, 0, 0, 0, 0, 0, 0, 0}}; return r; }
HD fe fe_one() { fe r = {{1, 0, 0, 0, 0, 0, 0, 0}}; return r; }

// ---------------------------------------------------------------- field
HD fe fe_add(const fe &a, const fe &b) {
    fe r; u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (u64)a.v[i] + b.v[i]; r.v[i] = (u32)c; c >>= 32; }
    // fold the carry: 2^256 == 38
    u64 k = c * 38;
#pragma unroll
    for (int i = 0; i < 8; i++) { k += r.v[i]; r.v[i] = (u32)k; k >>= 32; }
    r.v[0] += (u32)k * 38;   // second wrap only possible when the value is < 38: no further carry
    return r;
}

HD fe fe_sub(const fe &a, const fe &b) {
    fe r; int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (int64_t)a.v[i] - b.v[i]; r.v[i] = (u32)c; c >>= 32; }
    // borrow: value wrapped by +2^256 == +38, so subtract 38
    int64_t k = c * 38;   // c is 0 or -1
#pragma unroll
    for (int i = 0; i < 8; i++) { k += r.v[i]; r.v[i] = (u32)k; k >>= 32; }
    r.v[0] += (u32)((int32_t)k * 38); // second wrap only when value >= 2^256-38: no further borrow
    return r;
}

HD fe fe_neg(const fe &a) { return fe_sub(fe_zero(), a); }

// 16-limb product folded with 2^256 == 38
HD fe fe_reduce16(const u32 t[16]) {
    fe r; u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (u64)t[i] + (u64)t[i + 8] * 38; r.v[i] = (u32)c; c >>= 32; }
    u64 k = c * 38;          // c <= 38
#pragma unroll
    for (int i = 0; i < 8; i++) { k += r.v[i]; r.v[i] = (u32)k; k >>= 32; }
    r.v[0] += (u32)k * 38;
    return r;
}

HD fe fe_mul(const fe &a, const fe &b) {
    u32 t[16];
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 carry = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            u64 x = (u64)a.v[i] * b.v[j] + t[i + j] + carry;
            t[i + j] = (u32)x; carry = (u32)(x >> 32);
        }
        t[i + 8] = carry;
    }
    return fe_reduce16(t);
}

HD fe fe_sq(const fe &a) {
    // off-diagonal products once, doubled, plus the diagonal
    u32 t[16];
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        u32 carry = 0;
#pragma unroll
        for (int j = i + 1; j < 8; j++) {
            u64 x = (u64)a.v[i] * a.v[j] + t[i + j] + carry;
            t[i + j] = (u32)x; carry = (u32)(x >> 32);
        }
        t[i + 8] = carry;
    }
    // double
    u32 top = 0;
#pragma unroll
    for (int i = 1; i < 16; i++) { u32 n = t[i] >> 31; t[i] = (t[i] << 1) | top; top = n; }
    // add squares
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 s = (u64)a.v[i] * a.v[i];
        c += (u64)t[2 * i] + (u32)s; t[2 * i] = (u32)c; c >>= 32;
        c += (u64)t[2 * i + 1] + (u32)(s >> 32); t[2 * i + 1] = (u32)c; c >>= 32;
    }
    return fe_reduce16(t);
}

HD fe fe_mul_small(const fe &a, u32 k) {
    fe r; u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (u64)a.v[i] * k; r.v[i] = (u32)c; c >>= 32; }
    u64 q = c * 38;
#pragma unroll
    for (int i = 0; i < 8; i++) { q += r.v[i]; r.v[i] = (u32)q; q >>= 32; }
    r.v[0] += (u32)q * 38;
    return r;
}

// canonical representative in [0, p)
HD fe fe_canon(const fe &a) {
    fe r = a;
    // fold bit 255: x = hi*2^255 + lo == lo + 19*hi
    u64 c = (u64)(r.v[7] >> 31) * 19; r.v[7] &= 0x7fffffffu;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += r.v[i]; r.v[i] = (u32)c; c >>= 32; }
    // now r < 2^255 + 19; one more fold (only fires when r >= 2^255)
    c = (u64)(r.v[7] >> 31) * 19; r.v[7] &= 0x7fffffffu;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += r.v[i]; r.v[i] = (u32)c; c >>= 32; }
    // r < 2^255: subtract p if r >= p  (r + 19 >= 2^255)
    fe t; c = 19;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += r.v[i]; t.v[i] = (u32)c; c >>= 32; }
    u32 ge_p = t.v[7] >> 31;
    t.v[7] &= 0x7fffffffu;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = ge_p ? t.v[i] : r.v[i];
    return r;
}

HD bool fe_iszero(const fe &a) {
    fe c = fe_canon(a); u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= c.v[i];
    return o == 0;
}
HD bool fe_isneg(const fe &a) { return fe_canon(a).v[0] & 1; }
HD bool fe_eq(const fe &a, const fe &b) { return fe_iszero(fe_sub(a, b)); }
HD fe fe_select(const fe &a, const fe &b, bool pick_b) {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = pick_b ? b.v[i] : a.v[i];
    return r;
}
HD fe fe_abs(const fe &a) { return fe_isneg(a) ? fe_neg(a) : a; }

HD fe fe_frombytes(const uint8_t *s) {   // ignores bit 255 (dalek FieldElement::from_bytes)
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++)
        r.v[i] = (u32)s[4 * i] | ((u32)s[4 * i + 1] << 8) | ((u32)s[4 * i + 2] << 16) | ((u32)s[4 * i + 3] << 24);
    r.v[7] &= 0x7fffffffu;
    return r;
}
HD void fe_tobytes(uint8_t *s, const fe &a) {
    fe c = fe_canon(a);
#pragma unroll
    for (int i = 0; i < 8; i++) { s[4 * i] = (uint8_t)c.v[i]; s[4 * i + 1] = (uint8_t)(c.v[i] >> 8); s[4 * i + 2] = (uint8_t)(c.v[i] >> 16); s[4 * i + 3] = (uint8_t)(c.v[i] >> 24); }
}

HDN inline fe fe_sqn(fe a, int n) { for (int i = 0; i < n; i++) a = fe_sq(a); return a; }

// z^(2^250-1), also returns z^11
HDN inline fe fe_pow_2_250_1(const fe &z, fe &z11) {
    fe z2 = fe_sq(z);
    fe z9 = fe_mul(fe_sqn(z2, 2), z);
    z11 = fe_mul(z9, z2);
    fe z_5_0 = fe_mul(fe_sq(z11), z9);
    fe z_10_0 = fe_mul(fe_sqn(z_5_0, 5), z_5_0);
    fe z_20_0 = fe_mul(fe_sqn(z_10_0, 10), z_10_0);
    fe z_40_0 = fe_mul(fe_sqn(z_20_0, 20), z_20_0);
    fe z_50_0 = fe_mul(fe_sqn(z_40_0, 10), z_10_0);
    fe z_100_0 = fe_mul(fe_sqn(z_50_0, 50), z_50_0);
    fe z_200_0 = fe_mul(fe_sqn(z_100_0, 100), z_100_0);
    return fe_mul(fe_sqn(z_200_0, 50), z_50_0);
}
HDN inline fe fe_invert(const fe &z) { fe z11; fe t = fe_pow_2_250_1(z, z11); return fe_mul(fe_sqn(t, 5), z11); }
HDN inline fe fe_pow22523(const fe &z) { fe z11; fe t = fe_pow_2_250_1(z, z11); return fe_mul(fe_sqn(t, 2), z); }

// RFC 9496 SQRT_RATIO_M1; returns was_square
HDN inline bool fe_sqrt_ratio_i(fe &out, const fe &u, const fe &v) {
    fe v3 = fe_mul(fe_sq(v), v);
    fe v7 = fe_mul(fe_sq(v3), v);
    fe r = fe_mul(fe_mul(u, v3), fe_pow22523(fe_mul(u, v7)));
    fe check = fe_mul(v, fe_sq(r));
    fe neg_u = fe_neg(u);
    bool correct = fe_eq(check, u);
    bool flipped = fe_eq(check, neg_u);
    bool flipped_i = fe_eq(check, fe_mul(neg_u, fe_sqrtm1()));
    fe r_prime = fe_mul(r, fe_sqrtm1());
    r = fe_select(r, r_prime, flipped || flipped_i);
    out = fe_abs(r);
    return correct || flipped;
}

// ---------------------------------------------------------------- scalars mod l (Montgomery, R = 2^256)
#define SC_L0 0x5cf5d3edu
#define SC_L1 0x5812631au
#define SC_L2 0xa2f79cd6u
#define SC_L3 0x14def9deu
#define SC_L7 0x10000000u
#define SC_LINV32 0x12547e1bu
HD u32 sc_l_limb(int i) {
    switch (i) { case 0: return SC_L0; case 1: return SC_L1; case 2: return SC_L2; case 3: return SC_L3; case 7: return SC_L7; default: return 0; }
}
HD sc sc_zero() { sc r = {{0, 0, 0, 0, 0, 0, 0, 0}}; return r; }
HD sc sc_one_plain() { sc r = {{1, 0, 0, 0, 0, 0, 0, 0}}; return r; }
HD sc sc_R2() { sc r = {{0x449c0f01u, 0xa40611e3u, 0x68859347u, 0xd00e1ba7u, 0x17f5be65u, 0xceec73d2u, 0x7c309a3du, 0x0399411bu}}; return r; }
HD sc sc_R3() { sc r = {{0x7b83a2dbu, 0x2a9e4968u, 0xaef7f3ecu, 0x278324e6u, 0x04ec5b65u, 0x8065dc6cu, 0x3599cec7u, 0x0e530b77u}}; return r; }      // 2^768 mod l
HD sc sc_one_mont() { sc r = {{0x8d98951du, 0xd6ec3174u, 0x737dcf70u, 0xc6ef5bf4u, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0x0fffffffu}}; return r; }

HD bool sc_geq_l(const u32 a[8]) {
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        u32 l = sc_l_limb(i);
        if (a[i] > l) return true;
        if (a[i] < l) return false;
    }
    return true;
}
HD void sc_cond_sub_l(u32 a[8], bool doit) {
    int64_t c = 0; u32 t[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (int64_t)a[i] - sc_l_limb(i); t[i] = (u32)c; c >>= 32; }
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = doit ? t[i] : a[i];
}
// a*b*R^-1 mod l ; needs a*b < l*R  (a < 2^256, b < l suffices).  Works for plain and Montgomery forms.
HD sc sc_montmul(const sc &a, const sc &b) {
    u32 t[10];
#pragma unroll
    for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 carry = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            u64 x = (u64)a.v[j] * b.v[i] + t[j] + carry;
            t[j] = (u32)x; carry = (u32)(x >> 32);
        }
        u64 y = (u64)t[8] + carry; t[8] = (u32)y; t[9] = (u32)(y >> 32);
        u32 m = t[0] * SC_LINV32;
        u64 x = (u64)m * SC_L0 + t[0]; carry = (u32)(x >> 32);
#pragma unroll
        for (int j = 1; j < 8; j++) {
            x = (u64)m * sc_l_limb(j) + t[j] + carry;
            t[j - 1] = (u32)x; carry = (u32)(x >> 32);
        }
        y = (u64)t[8] + carry; t[7] = (u32)y; t[8] = t[9] + (u32)(y >> 32); t[9] = 0;
    }
    sc r;
    bool sub = t[8] != 0 || sc_geq_l(t);
    sc_cond_sub_l(t, sub);
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    return r;
}
// Lazy reduction for sums of products: acc (17 limbs) += a * b as a plain 512-bit product; sc_redc_wide turns the sum T into T * R^-1 mod l
// (= the sum of the sc_montmul results, canonical) with ONE Montgomery reduction.  Needs T < 17 * l^2 (sixteen products of operands < l and
// change): the reduction leaves less than T / R + l < 2.1 l, two conditional subtractions.  A product costs 64 multiply-adds here against
// 104 in sc_montmul, and no compare / subtract per term.
HD void sc_mac_wide(u32 acc[17], const sc &a, const sc &b) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 carry = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            u64 x = (u64)a.v[j] * b.v[i] + acc[i + j] + carry;
            acc[i + j] = (u32)x; carry = (u32)(x >> 32);
        }
#pragma unroll
        for (int k = i + 8; k < 17; k++) { u64 y = (u64)acc[k] + carry; acc[k] = (u32)y; carry = (u32)(y >> 32); }
    }
}
HD sc sc_redc_wide(const u32 acc[17]) {
    u32 t[18];
#pragma unroll
    for (int i = 0; i < 17; i++) t[i] = acc[i];
    t[17] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32 m = t[i] * SC_LINV32, carry = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            u32 l = sc_l_limb(j);
            if (l == 0 && j != 0) { u64 y = (u64)t[i + j] + carry; t[i + j] = (u32)y; carry = (u32)(y >> 32); }
            else { u64 x = (u64)m * l + t[i + j] + carry; t[i + j] = (u32)x; carry = (u32)(x >> 32); }
        }
#pragma unroll
        for (int k = i + 8; k < 18; k++) { u64 y = (u64)t[k] + carry; t[k] = (u32)y; carry = (u32)(y >> 32); }
    }
    // t[8 .. 16] = T' < 2.1 l: limbs 16 and up are zero
    u32 r8[8];
#pragma unroll
    for (int i = 0; i < 8; i++) r8[i] = t[8 + i];
    sc_cond_sub_l(r8, sc_geq_l(r8));
    sc_cond_sub_l(r8, sc_geq_l(r8));
    sc r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = r8[i];
    return r;
}
HD sc sc_add(const sc &a, const sc &b) {   // inputs < l
    u32 t[8]; u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (u64)a.v[i] + b.v[i]; t[i] = (u32)c; c >>= 32; }
    sc_cond_sub_l(t, sc_geq_l(t));
    sc r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    return r;
}
HD bool sc_iszero(const sc &a) { u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0; }
HD sc sc_neg(const sc &a) {
    sc r; int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (int64_t)sc_l_limb(i) - a.v[i]; r.v[i] = (u32)c; c >>= 32; }
    bool z = sc_iszero(a);
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = z ? 0 : r.v[i];
    return r;
}
HD sc sc_sub(const sc &a, const sc &b) { return sc_add(a, sc_neg(b)); }
HD sc sc_to_mont(const sc &a) { return sc_montmul(a, sc_R2()); }          // a (< 2^256) -> a*R mod l
HD sc sc_from_mont(const sc &a) { return sc_montmul(a, sc_one_plain()); } // a*R -> a
HD sc sc_mul_plain(const sc &a, const sc &b) { return sc_montmul(sc_montmul(a, b), sc_R2()); }
HD sc sc_frombytes(const uint8_t *s) {
    sc r;
#pragma unroll
    for (int i = 0; i < 8; i++)
        r.v[i] = (u32)s[4 * i] | ((u32)s[4 * i + 1] << 8) | ((u32)s[4 * i + 2] << 16) | ((u32)s[4 * i + 3] << 24);
    return r;
}
HD void sc_tobytes(uint8_t *s, const sc &a) {
#pragma unroll
    for (int i = 0; i < 8; i++) { s[4 * i] = (uint8_t)a.v[i]; s[4 * i + 1] = (uint8_t)(a.v[i] >> 8); s[4 * i + 2] = (uint8_t)(a.v[i] >> 16); s[4 * i + 3] = (uint8_t)(a.v[i] >> 24); }
}
// 64 bytes -> canonical scalar (Scalar::from_bytes_mod_order_wide)
// (lo R + hi R^2) R^-1 = lo + hi 2^256: two plain products under ONE Montgomery reduction (T < 2 * 2^256 * l: the reduction leaves < 3 l)
HD sc sc_from_wide(const sc &lo, const sc &hi) {
    u32 acc[17];
#pragma unroll
    for (int i = 0; i < 17; i++) acc[i] = 0;
    sc_mac_wide(acc, lo, sc_one_mont());
    sc_mac_wide(acc, hi, sc_R2());
    return sc_redc_wide(acc);
}
// the same value in Montgomery form, (lo R^2 + hi R^3) R^-1 = (lo + hi 2^256) R  (for callers that want both: from_mont of this is the canonical one)
HD sc sc_from_wide_mont(const sc &lo, const sc &hi) {
    u32 acc[17];
#pragma unroll
    for (int i = 0; i < 17; i++) acc[i] = 0;
    sc_mac_wide(acc, lo, sc_R2());
    sc_mac_wide(acc, hi, sc_R3());
    return sc_redc_wide(acc);
}
HD sc sc_from_u64(u64 x) { sc r = sc_zero(); r.v[0] = (u32)x; r.v[1] = (u32)(x >> 32); return r; }
HDN inline sc sc_invert_mont(const sc &a) {   // Montgomery in/out: a^(l-2)
    u32 e[8] = {SC_L0 - 2, SC_L1, SC_L2, SC_L3, 0, 0, 0, SC_L7};
    sc acc = sc_one_mont();
    for (int i = 252; i >= 0; i--) {
        acc = sc_montmul(acc, acc);
        if ((e[i >> 5] >> (i & 31)) & 1) acc = sc_montmul(acc, a);
    }
    return acc;
}

// ---------------------------------------------------------------- group
HD ge ge_identity() { ge r; r.X = fe_zero(); r.Y = fe_one(); r.Z = fe_one(); r.T = fe_zero(); return r; }
HD niels niels_identity() { niels r; r.ypx = fe_one(); r.ymx = fe_one(); r.t2d = fe_zero(); return r; }

HD ge ge_add(const ge &p, const ge &q) {
    fe A = fe_mul(fe_sub(p.Y, p.X), fe_sub(q.Y, q.X));
    fe B = fe_mul(fe_add(p.Y, p.X), fe_add(q.Y, q.X));
    fe C = fe_mul(fe_mul(p.T, q.T), fe_d2());
    fe D = fe_mul(p.Z, q.Z); D = fe_add(D, D);
    fe E = fe_sub(B, A), F = fe_sub(D, C), G = fe_add(D, C), H = fe_add(B, A);
    ge r; r.X = fe_mul(E, F); r.Y = fe_mul(G, H); r.T = fe_mul(E, H); r.Z = fe_mul(F, G);
    return r;
}
// p + q (neg = false) or p - q (neg = true), q affine niels: 7 mul
HD ge ge_madd(const ge &p, const niels &q, bool neg) {
    fe a_f = fe_select(q.ymx, q.ypx, neg), b_f = fe_select(q.ypx, q.ymx, neg);
    fe A = fe_mul(fe_sub(p.Y, p.X), a_f);
    fe B = fe_mul(fe_add(p.Y, p.X), b_f);
    fe C = fe_mul(p.T, q.t2d);
    fe D = fe_add(p.Z, p.Z);
    fe E = fe_sub(B, A), H = fe_add(B, A);
    fe Fp = fe_sub(D, C), Gp = fe_add(D, C);
    fe F = fe_select(Fp, Gp, neg), G = fe_select(Gp, Fp, neg);
    ge r; r.X = fe_mul(E, F); r.Y = fe_mul(G, H); r.T = fe_mul(E, H); r.Z = fe_mul(F, G);
    return r;
}
HD ge ge_double(const ge &p) {
    fe A = fe_sq(p.X), B = fe_sq(p.Y), C = fe_sq(p.Z); C = fe_add(C, C);
    fe E = fe_sub(fe_sub(fe_sq(fe_add(p.X, p.Y)), A), B);
    fe G = fe_sub(B, A);               // D + B with D = -A
    fe F = fe_sub(G, C);
    fe H = fe_neg(fe_add(A, B));       // D - B
    ge r; r.X = fe_mul(E, F); r.Y = fe_mul(G, H); r.T = fe_mul(E, H); r.Z = fe_mul(F, G);
    return r;
}
HD ge ge_neg(const ge &p) { ge r; r.X = fe_neg(p.X); r.Y = p.Y; r.Z = p.Z; r.T = fe_neg(p.T); return r; }
HD ge ge_from_niels(const niels &q) { return ge_madd(ge_identity(), q, false); }
HDN inline niels ge_to_niels(const ge &p) {
    fe zi = fe_invert(p.Z);
    fe x = fe_mul(p.X, zi), y = fe_mul(p.Y, zi);
    niels r; r.ypx = fe_add(y, x); r.ymx = fe_sub(y, x); r.t2d = fe_mul(fe_mul(x, y), fe_d2());
    return r;
}
HD niels niels_from_affine(const fe &x, const fe &y) {
    niels r; r.ypx = fe_add(y, x); r.ymx = fe_sub(y, x); r.t2d = fe_mul(fe_mul(x, y), fe_d2()); return r;
}
HD bool ge_is_identity_ristretto(const ge &p) { return fe_iszero(p.X) || fe_iszero(p.Y); }

// RFC 9496 4.3.2 Encode
HDN inline void ristretto_encode(uint8_t *s, const ge &p) {
    fe u1 = fe_mul(fe_add(p.Z, p.Y), fe_sub(p.Z, p.Y));
    fe u2 = fe_mul(p.X, p.Y);
    fe invsqrt; fe_sqrt_ratio_i(invsqrt, fe_one(), fe_mul(u1, fe_sq(u2)));
    fe den1 = fe_mul(invsqrt, u1), den2 = fe_mul(invsqrt, u2);
    fe z_inv = fe_mul(fe_mul(den1, den2), p.T);
    fe ix0 = fe_mul(p.X, fe_sqrtm1()), iy0 = fe_mul(p.Y, fe_sqrtm1());
    fe ench = fe_mul(den1, fe_invsqrt_a_minus_d());
    bool rotate = fe_isneg(fe_mul(p.T, z_inv));
    fe x = fe_select(p.X, iy0, rotate), y = fe_select(p.Y, ix0, rotate);
    fe den_inv = fe_select(den2, ench, rotate);
    if (fe_isneg(fe_mul(x, z_inv))) y = fe_neg(y);
    fe sres = fe_abs(fe_mul(den_inv, fe_sub(p.Z, y)));
    fe_tobytes(s, sres);
}
// RFC 9496 4.3.1 Decode; returns false on invalid encodings.  Output has Z = 1.
HDN inline bool ristretto_decode(ge &p, const uint8_t *sb) {
    fe s = fe_frombytes(sb);
    uint8_t chk[32]; fe_tobytes(chk, s);
    bool canon = true;
    for (int i = 0; i < 32; i++) canon &= (chk[i] == sb[i]);
    if (!canon || (sb[0] & 1)) return false;
    fe ss = fe_sq(s);
    fe u1 = fe_sub(fe_one(), ss), u2 = fe_add(fe_one(), ss);
    fe u2s = fe_sq(u2);
    fe v = fe_sub(fe_neg(fe_mul(fe_d(), fe_sq(u1))), u2s);
    fe invsqrt; bool ok = fe_sqrt_ratio_i(invsqrt, fe_one(), fe_mul(v, u2s));
    fe dx = fe_mul(invsqrt, u2);
    fe dy = fe_mul(fe_mul(invsqrt, dx), v);
    fe x = fe_abs(fe_mul(fe_add(s, s), dx));
    fe y = fe_mul(u1, dy);
    fe t = fe_mul(x, y);
    if (!ok || fe_isneg(t) || fe_iszero(y)) return false;
    p.X = x; p.Y = y; p.Z = fe_one(); p.T = t;
    return true;
}
// dalek RistrettoPoint::elligator_ristretto_flavor
HDN inline ge ristretto_elligator(const fe &r0) {
    fe one = fe_one();
    fe r = fe_mul(fe_sq(r0), fe_sqrtm1());
    fe Ns = fe_mul(fe_add(r, one), fe_one_minus_d_sq());
    fe c = fe_neg(one);
    fe Dn = fe_mul(fe_sub(c, fe_mul(fe_d(), r)), fe_add(r, fe_d()));
    fe s; bool ok = fe_sqrt_ratio_i(s, Ns, Dn);
    fe s_prime = fe_mul(s, r0);
    if (!fe_isneg(s_prime)) s_prime = fe_neg(s_prime);
    s = fe_select(s_prime, s, ok);
    c = fe_select(r, c, ok);
    fe Nt = fe_sub(fe_mul(fe_mul(c, fe_sub(r, one)), fe_d_minus_one_sq()), Dn);
    fe ss = fe_sq(s);
    fe W0 = fe_mul(fe_add(s, s), Dn);
    fe W1 = fe_mul(Nt, fe_sqrt_ad_minus_one());
    fe W2 = fe_sub(one, ss), W3 = fe_add(one, ss);
    ge p; p.X = fe_mul(W0, W3); p.Y = fe_mul(W2, W1); p.Z = fe_mul(W1, W3); p.T = fe_mul(W0, W2);
    return p;
}
HDN inline ge ristretto_from_uniform(const uint8_t *b64) {
    return ge_add(ristretto_elligator(fe_frombytes(b64)), ristretto_elligator(fe_frombytes(b64 + 32)));
}

}  // namespace rofl
