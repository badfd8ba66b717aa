// GF(2^255-19) in radix 2^25.5 (10 unsigned limbs of 26/25 bits) -- the in-register representation of the hot
// kernels.  Measured on gfx950: v_mad_u64_u32, v_add_co/v_addc and the 24-bit multiplies all issue at ~4.5 cycles
// per wave-instruction, plain 32-bit adds at ~2.7; the saturated 8x32 form (fe32.hpp) needs 319 VALU instructions
// per multiplication and ~78 per add/sub because of its carry chains, this form ~165 and 10-20.  Memory layouts stay
// 8 x 32-bit (32 B per element): kernels unpack on load and pack on store.
//
// Limb bounds ("tight": even limbs < 2^26 + 2^19, odd limbs < 2^25 + 2^19):
//   fd_mul / fd_sq / fd_carry / fd_unpack outputs are tight;
//   fd_add: no carry;  fd_sub(a, b) = a + 2p - b needs b tight;
//   fd_mul(f, g) needs f < 2^28 and g < 2^27.75 limb-wise (accumulators stay below 2^64: see DESIGN.md section 3, item 8).
#pragma once
#include "fe32.hpp"
#if defined(ROFL_FD_CHECK_HOST) && !defined(__HIP_DEVICE_COMPILE__)
#include <assert.h>
#define ROFL_FD_CHECK 1
#endif

namespace rofl {

struct fd { u32 v[10]; };
struct gd { fd X, Y, Z, T; };
struct nd { fd ypx, ymx, t2d; };

#define FD_M26 0x3ffffffu
#define FD_M25 0x1ffffffu

HD fd fd_zero() { fd r = {{0, 0, 0, 0, 0, 0, 0, 0, 0, 0}}; return r; }
HD fd fd_one() { fd r = {{1, 0, 0, 0, 0, 0, 0, 0, 0, 0}}; return r; }

HD fd fd_unpack(const fe &a) {
    const u32 *w = a.v;
    fd r;
    r.v[0] = w[0] & FD_M26;
    r.v[1] = ((w[0] >> 26) | (w[1] << 6)) & FD_M25;
    r.v[2] = ((w[1] >> 19) | (w[2] << 13)) & FD_M26;
    r.v[3] = ((w[2] >> 13) | (w[3] << 19)) & FD_M25;
    r.v[4] = (w[3] >> 6) & FD_M26;
    r.v[5] = w[4] & FD_M25;
    r.v[6] = ((w[4] >> 25) | (w[5] << 7)) & FD_M26;
    r.v[7] = ((w[5] >> 19) | (w[6] << 13)) & FD_M25;
    r.v[8] = ((w[6] >> 12) | (w[7] << 20)) & FD_M26;
    r.v[9] = (w[7] >> 6) & FD_M25;
    r.v[0] += 19u * (w[7] >> 31);      // bit 255: 2^255 == 19
    return r;
}
// weak carry: any limbs < 2^32 - 2^26 -> tight
HD fd fd_carry(const fd &a) {
    fd r = a; u32 c;
    c = r.v[0] >> 26; r.v[0] &= FD_M26; r.v[1] += c;
    c = r.v[1] >> 25; r.v[1] &= FD_M25; r.v[2] += c;
    c = r.v[2] >> 26; r.v[2] &= FD_M26; r.v[3] += c;
    c = r.v[3] >> 25; r.v[3] &= FD_M25; r.v[4] += c;
    c = r.v[4] >> 26; r.v[4] &= FD_M26; r.v[5] += c;
    c = r.v[5] >> 25; r.v[5] &= FD_M25; r.v[6] += c;
    c = r.v[6] >> 26; r.v[6] &= FD_M26; r.v[7] += c;
    c = r.v[7] >> 25; r.v[7] &= FD_M25; r.v[8] += c;
    c = r.v[8] >> 26; r.v[8] &= FD_M26; r.v[9] += c;
    c = r.v[9] >> 25; r.v[9] &= FD_M25; r.v[0] += 19u * c;
    c = r.v[0] >> 26; r.v[0] &= FD_M26; r.v[1] += c;
    return r;
}
// -> 8 x 32 words, value < 2^256 (not necessarily canonical)
HD fe fd_pack(const fd &a) {
    fd t = fd_carry(a);
    fe r; u64 acc;
    acc = (u64)t.v[0] + ((u64)t.v[1] << 26);                  // bits 0..
    r.v[0] = (u32)acc; acc >>= 32;
    acc += (u64)t.v[2] << 19;                                 // limb 2 starts at bit 51 = 32 + 19
    r.v[1] = (u32)acc; acc >>= 32;
    acc += (u64)t.v[3] << 13;                                 // 77 = 64 + 13
    r.v[2] = (u32)acc; acc >>= 32;
    acc += (u64)t.v[4] << 6;                                  // 102 = 96 + 6
    r.v[3] = (u32)acc; acc >>= 32;
    acc += (u64)t.v[5];                                       // 128
    acc += (u64)t.v[6] << 25;                                 // 153 = 128 + 25
    r.v[4] = (u32)acc; acc >>= 32;
    acc += (u64)t.v[7] << 19;                                 // 179 = 160 + 19
    r.v[5] = (u32)acc; acc >>= 32;
    acc += (u64)t.v[8] << 12;                                 // 204 = 192 + 12
    r.v[6] = (u32)acc; acc >>= 32;
    acc += (u64)t.v[9] << 6;                                  // 230 = 224 + 6
    r.v[7] = (u32)acc;                                        // value < 2^255 + 2^(230+20): fits in 256 bits
    return r;
}

HD fd fd_add(const fd &a, const fd &b) {
    fd r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}
// a + 2p - b ; b must be tight
HD fd fd_sub(const fd &a, const fd &b) {
    fd r;
    r.v[0] = a.v[0] + 0x7ffffdau - b.v[0];
#pragma unroll
    for (int i = 1; i < 10; i++) r.v[i] = a.v[i] + ((i & 1) ? 0x3fffffeu : 0x7fffffeu) - b.v[i];
    return r;
}
HD fd fd_neg(const fd &a) { return fd_sub(fd_zero(), a); }
HD fd fd_select(const fd &a, const fd &b, bool pick_b) {
    fd r;
#pragma unroll
    for (int i = 0; i < 10; i++) r.v[i] = pick_b ? b.v[i] : a.v[i];
    return r;
}

// carry chain of the ten 64-bit column sums (ref10 order), result tight
HD fd fd_reduce_cols(u64 h0, u64 h1, u64 h2, u64 h3, u64 h4, u64 h5, u64 h6, u64 h7, u64 h8, u64 h9) {
    u64 c;
    c = h0 >> 26; h1 += c; h0 &= FD_M26;
    c = h4 >> 26; h5 += c; h4 &= FD_M26;
    c = h1 >> 25; h2 += c; h1 &= FD_M25;
    c = h5 >> 25; h6 += c; h5 &= FD_M25;
    c = h2 >> 26; h3 += c; h2 &= FD_M26;
    c = h6 >> 26; h7 += c; h6 &= FD_M26;
    c = h3 >> 25; h4 += c; h3 &= FD_M25;
    c = h7 >> 25; h8 += c; h7 &= FD_M25;
    c = h4 >> 26; h5 += c; h4 &= FD_M26;
    c = h8 >> 26; h9 += c; h8 &= FD_M26;
    c = h9 >> 25; h0 += c * 19; h9 &= FD_M25;
    c = h0 >> 26; h1 += c; h0 &= FD_M26;
    fd r = {{(u32)h0, (u32)h1, (u32)h2, (u32)h3, (u32)h4, (u32)h5, (u32)h6, (u32)h7, (u32)h8, (u32)h9}};
    return r;
}

HD fd fd_mul(const fd &f, const fd &g) {
#if defined(ROFL_FD_CHECK)
    for (int i = 0; i < 10; i++) { assert((u64)f.v[i] < (1ULL << 28) + (1ULL << 20)); assert((u64)g.v[i] * 19 < (1ULL << 32)); }
#endif
    u32 f0 = f.v[0], f1 = f.v[1], f2 = f.v[2], f3 = f.v[3], f4 = f.v[4], f5 = f.v[5], f6 = f.v[6], f7 = f.v[7], f8 = f.v[8], f9 = f.v[9];
    u32 g0 = g.v[0], g1 = g.v[1], g2 = g.v[2], g3 = g.v[3], g4 = g.v[4], g5 = g.v[5], g6 = g.v[6], g7 = g.v[7], g8 = g.v[8], g9 = g.v[9];
    u32 g1_19 = 19 * g1, g2_19 = 19 * g2, g3_19 = 19 * g3, g4_19 = 19 * g4, g5_19 = 19 * g5, g6_19 = 19 * g6, g7_19 = 19 * g7, g8_19 = 19 * g8, g9_19 = 19 * g9;
    u32 f1_2 = 2 * f1, f3_2 = 2 * f3, f5_2 = 2 * f5, f7_2 = 2 * f7, f9_2 = 2 * f9;
#define M(a, b) ((u64)(a) * (b))
    u64 h0 = M(f0, g0) + M(f1_2, g9_19) + M(f2, g8_19) + M(f3_2, g7_19) + M(f4, g6_19) + M(f5_2, g5_19) + M(f6, g4_19) + M(f7_2, g3_19) + M(f8, g2_19) + M(f9_2, g1_19);
    u64 h1 = M(f0, g1) + M(f1, g0) + M(f2, g9_19) + M(f3, g8_19) + M(f4, g7_19) + M(f5, g6_19) + M(f6, g5_19) + M(f7, g4_19) + M(f8, g3_19) + M(f9, g2_19);
    u64 h2 = M(f0, g2) + M(f1_2, g1) + M(f2, g0) + M(f3_2, g9_19) + M(f4, g8_19) + M(f5_2, g7_19) + M(f6, g6_19) + M(f7_2, g5_19) + M(f8, g4_19) + M(f9_2, g3_19);
    u64 h3 = M(f0, g3) + M(f1, g2) + M(f2, g1) + M(f3, g0) + M(f4, g9_19) + M(f5, g8_19) + M(f6, g7_19) + M(f7, g6_19) + M(f8, g5_19) + M(f9, g4_19);
    u64 h4 = M(f0, g4) + M(f1_2, g3) + M(f2, g2) + M(f3_2, g1) + M(f4, g0) + M(f5_2, g9_19) + M(f6, g8_19) + M(f7_2, g7_19) + M(f8, g6_19) + M(f9_2, g5_19);
    u64 h5 = M(f0, g5) + M(f1, g4) + M(f2, g3) + M(f3, g2) + M(f4, g1) + M(f5, g0) + M(f6, g9_19) + M(f7, g8_19) + M(f8, g7_19) + M(f9, g6_19);
    u64 h6 = M(f0, g6) + M(f1_2, g5) + M(f2, g4) + M(f3_2, g3) + M(f4, g2) + M(f5_2, g1) + M(f6, g0) + M(f7_2, g9_19) + M(f8, g8_19) + M(f9_2, g7_19);
    u64 h7 = M(f0, g7) + M(f1, g6) + M(f2, g5) + M(f3, g4) + M(f4, g3) + M(f5, g2) + M(f6, g1) + M(f7, g0) + M(f8, g9_19) + M(f9, g8_19);
    u64 h8 = M(f0, g8) + M(f1_2, g7) + M(f2, g6) + M(f3_2, g5) + M(f4, g4) + M(f5_2, g3) + M(f6, g2) + M(f7_2, g1) + M(f8, g0) + M(f9_2, g9_19);
    u64 h9 = M(f0, g9) + M(f1, g8) + M(f2, g7) + M(f3, g6) + M(f4, g5) + M(f5, g4) + M(f6, g3) + M(f7, g2) + M(f8, g1) + M(f9, g0);
    return fd_reduce_cols(h0, h1, h2, h3, h4, h5, h6, h7, h8, h9);
}

HD fd fd_sq(const fd &f) {
#if defined(ROFL_FD_CHECK)
    for (int i = 0; i < 10; i++) assert((u64)f.v[i] * ((i & 1) ? 38 : 19) < (1ULL << 32));
#endif
    u32 f0 = f.v[0], f1 = f.v[1], f2 = f.v[2], f3 = f.v[3], f4 = f.v[4], f5 = f.v[5], f6 = f.v[6], f7 = f.v[7], f8 = f.v[8], f9 = f.v[9];
    u32 f0_2 = 2 * f0, f1_2 = 2 * f1, f2_2 = 2 * f2, f3_2 = 2 * f3, f4_2 = 2 * f4, f5_2 = 2 * f5, f6_2 = 2 * f6, f7_2 = 2 * f7;
    u32 f5_38 = 38 * f5, f6_19 = 19 * f6, f7_38 = 38 * f7, f8_19 = 19 * f8, f9_38 = 38 * f9;
    u64 h0 = M(f0, f0) + M(f1_2, f9_38) + M(f2_2, f8_19) + M(f3_2, f7_38) + M(f4_2, f6_19) + M(f5, f5_38);
    u64 h1 = M(f0_2, f1) + M(f2, f9_38) + M(f3_2, f8_19) + M(f4, f7_38) + M(f5_2, f6_19);
    u64 h2 = M(f0_2, f2) + M(f1_2, f1) + M(f3_2, f9_38) + M(f4_2, f8_19) + M(f5_2, f7_38) + M(f6, f6_19);
    u64 h3 = M(f0_2, f3) + M(f1_2, f2) + M(f4, f9_38) + M(f5_2, f8_19) + M(f6, f7_38);
    u64 h4 = M(f0_2, f4) + M(f1_2, f3_2) + M(f2, f2) + M(f5_2, f9_38) + M(f6_2, f8_19) + M(f7, f7_38);
    u64 h5 = M(f0_2, f5) + M(f1_2, f4) + M(f2_2, f3) + M(f6, f9_38) + M(f7_2, f8_19);
    u64 h6 = M(f0_2, f6) + M(f1_2, f5_2) + M(f2_2, f4) + M(f3_2, f3) + M(f7_2, f9_38) + M(f8, f8_19);
    u64 h7 = M(f0_2, f7) + M(f1_2, f6) + M(f2_2, f5) + M(f3_2, f4) + M(f8, f9_38);
    u64 h8 = M(f0_2, f8) + M(f1_2, f7_2) + M(f2_2, f6) + M(f3_2, f5_2) + M(f4, f4) + M(f9, f9_38);
    u64 h9 = M(f0_2, f9) + M(f1_2, f8) + M(f2_2, f7) + M(f3_2, f6) + M(f4_2, f5);
    return fd_reduce_cols(h0, h1, h2, h3, h4, h5, h6, h7, h8, h9);
}
#undef M

HDN inline fd fd_sqn(fd a, int n) { for (int i = 0; i < n; i++) a = fd_sq(a); return a; }
HDN inline fd fd_invert(const fd &z) {
    fd z2 = fd_sq(z);
    fd z9 = fd_mul(fd_sqn(z2, 2), z);
    fd z11 = fd_mul(z9, z2);
    fd z_5_0 = fd_mul(fd_sq(z11), z9);
    fd z_10_0 = fd_mul(fd_sqn(z_5_0, 5), z_5_0);
    fd z_20_0 = fd_mul(fd_sqn(z_10_0, 10), z_10_0);
    fd z_40_0 = fd_mul(fd_sqn(z_20_0, 20), z_20_0);
    fd z_50_0 = fd_mul(fd_sqn(z_40_0, 10), z_10_0);
    fd z_100_0 = fd_mul(fd_sqn(z_50_0, 50), z_50_0);
    fd z_200_0 = fd_mul(fd_sqn(z_100_0, 100), z_100_0);
    fd z_250_0 = fd_mul(fd_sqn(z_200_0, 50), z_50_0);
    return fd_mul(fd_sqn(z_250_0, 5), z11);
}

// 2d in this radix
HD fd fd_d2() {
    fd r = {{0x2b2f159u, 0x1a6e509u, 0x22add7au, 0x0d4141du, 0x0038052u, 0x0f3d130u, 0x3407977u, 0x19ce331u, 0x1c56dffu, 0x0901b67u}};
    return r;
}

// ---------------------------------------------------------------- points
HD gd gd_identity() { gd r; r.X = fd_zero(); r.Y = fd_one(); r.Z = fd_one(); r.T = fd_zero(); return r; }
HD gd gd_unpack(const ge &p) { gd r; r.X = fd_unpack(p.X); r.Y = fd_unpack(p.Y); r.Z = fd_unpack(p.Z); r.T = fd_unpack(p.T); return r; }
HD ge gd_pack(const gd &p) { ge r; r.X = fd_pack(p.X); r.Y = fd_pack(p.Y); r.Z = fd_pack(p.Z); r.T = fd_pack(p.T); return r; }
HD nd nd_unpack(const niels &q) { nd r; r.ypx = fd_unpack(q.ypx); r.ymx = fd_unpack(q.ymx); r.t2d = fd_unpack(q.t2d); return r; }

// p +- q, q affine niels (tight).  p must have tight coordinates (every output here is tight).
HD gd gd_madd(const gd &p, const nd &q, bool neg) {
    fd a_f = fd_select(q.ymx, q.ypx, neg), b_f = fd_select(q.ypx, q.ymx, neg);
    fd A = fd_mul(fd_sub(p.Y, p.X), a_f);
    fd B = fd_mul(fd_add(p.Y, p.X), b_f);
    fd C = fd_mul(p.T, q.t2d);
    fd D = fd_add(p.Z, p.Z);
    fd E = fd_sub(B, A), H = fd_add(B, A);
    fd DmC = fd_sub(D, C), DpC = fd_add(D, C);      // < 2^28 and < 1.5 * 2^27
    // neg: F = D + C, G = D - C
    gd r;
    fd Fbig = DmC, Gsm = DpC;                        // names by bound: *big* may reach 2^28 (must be the first operand)
    fd XF = fd_select(Fbig, Gsm, neg);               // value of F
    fd XG = fd_select(Gsm, Fbig, neg);               // value of G
    r.X = fd_mul(XF, E);                             // E < 1.5*2^27 is a valid second operand; XF < 2^28
    r.Y = fd_mul(XG, H);
    r.T = fd_mul(E, H);
    r.Z = fd_mul(Fbig, Gsm);                         // F*G either way
    return r;
}
// p + q, both extended with tight coordinates
HD gd gd_add(const gd &p, const gd &q) {
    fd A = fd_mul(fd_sub(p.Y, p.X), fd_sub(q.Y, q.X));
    fd B = fd_mul(fd_add(p.Y, p.X), fd_add(q.Y, q.X));
    fd C = fd_mul(fd_mul(p.T, q.T), fd_d2());
    fd D = fd_mul(p.Z, q.Z); D = fd_add(D, D);
    fd E = fd_sub(B, A), H = fd_add(B, A);
    fd F = fd_sub(D, C), G = fd_add(D, C);
    gd r; r.X = fd_mul(F, E); r.Y = fd_mul(G, H); r.T = fd_mul(E, H); r.Z = fd_mul(F, G);
    return r;
}
HD gd gd_double(const gd &p) {
    fd XX = fd_sq(p.X), YY = fd_sq(p.Y), ZZ = fd_sq(p.Z);
    fd ZZ2 = fd_add(ZZ, ZZ);
    fd S = fd_sq(fd_add(p.X, p.Y));
    fd cY = fd_carry(fd_add(YY, XX));                // YY + XX
    fd cZ = fd_carry(fd_sub(YY, XX));                // YY - XX
    fd cX = fd_sub(S, cY);                           // (X+Y)^2 - YY - XX      < 1.5 * 2^27
    fd cT = fd_sub(ZZ2, cZ);                         // 2 ZZ - (YY - XX)       < 2^28
    gd r; r.X = fd_mul(cT, cX); r.Y = fd_mul(cY, cZ); r.Z = fd_mul(cT, cZ); r.T = fd_mul(cX, cY);
    return r;
}
// doubling inside a chain of doublings: the next doubling reads X, Y, Z only, so the T product is skipped (7 instead of 8
// multiplications); the result's T is stale and must not feed an addition
HD gd gd_double_not(const gd &p) {
    fd XX = fd_sq(p.X), YY = fd_sq(p.Y), ZZ = fd_sq(p.Z);
    fd ZZ2 = fd_add(ZZ, ZZ);
    fd S = fd_sq(fd_add(p.X, p.Y));
    fd cY = fd_carry(fd_add(YY, XX));
    fd cZ = fd_carry(fd_sub(YY, XX));
    fd cX = fd_sub(S, cY);
    fd cT = fd_sub(ZZ2, cZ);
    gd r; r.X = fd_mul(cT, cX); r.Y = fd_mul(cY, cZ); r.Z = fd_mul(cT, cZ); r.T = p.T;
    return r;
}
// affine niels form given 1/Z (batch inversions)
HDN inline niels gd_to_niels_zinv(const gd &p, const fd &zi) {
    fd x = fd_mul(p.X, zi), y = fd_mul(p.Y, zi);
    niels r;
    r.ypx = fd_pack(fd_add(y, x)); r.ymx = fd_pack(fd_sub(y, x)); r.t2d = fd_pack(fd_mul(fd_mul(x, y), fd_d2()));
    return r;
}
HDN inline niels gd_to_niels(const gd &p) {
    fd zi = fd_invert(p.Z);
    fd x = fd_mul(p.X, zi), y = fd_mul(p.Y, zi);
    niels r;
    r.ypx = fd_pack(fd_add(y, x)); r.ymx = fd_pack(fd_sub(y, x)); r.t2d = fd_pack(fd_mul(fd_mul(x, y), fd_d2()));
    return r;
}

// ---------------------------------------------------------------- Ristretto codec in the register radix
// (RFC 9496 4.3.1 / 4.3.2, same steps as fe32.hpp's ristretto_decode / ristretto_encode: the 254-step square-root chain
//  runs on fd_sq -- ~110 VALU instructions instead of ~320 for the saturated form.)  Every fd_sub subtrahend and every
//  multiplication operand below is a product (tight) or a sum of two tight values; comparisons go through the canonical bytes.
HD bool fd_iszero(const fd &a) { return fe_iszero(fd_pack(a)); }
HD bool fd_isneg(const fd &a) { return fe_isneg(fd_pack(a)); }
HD bool fd_eq(const fd &a, const fd &b) { return fe_iszero(fe_sub(fd_pack(a), fd_pack(b))); }
HD fd fd_abs(const fd &a) { fd t = fd_carry(a); return fd_select(t, fd_carry(fd_neg(t)), fd_isneg(t)); }
HDN inline fd fd_pow22523(const fd &z) {      // z^((p-5)/8) = z^(2^252 - 3)
    fd z2 = fd_sq(z);
    fd z9 = fd_mul(fd_sqn(z2, 2), z);
    fd z11 = fd_mul(z9, z2);
    fd z_5_0 = fd_mul(fd_sq(z11), z9);
    fd z_10_0 = fd_mul(fd_sqn(z_5_0, 5), z_5_0);
    fd z_20_0 = fd_mul(fd_sqn(z_10_0, 10), z_10_0);
    fd z_40_0 = fd_mul(fd_sqn(z_20_0, 20), z_20_0);
    fd z_50_0 = fd_mul(fd_sqn(z_40_0, 10), z_10_0);
    fd z_100_0 = fd_mul(fd_sqn(z_50_0, 50), z_50_0);
    fd z_200_0 = fd_mul(fd_sqn(z_100_0, 100), z_100_0);
    fd z_250_0 = fd_mul(fd_sqn(z_200_0, 50), z_50_0);
    return fd_mul(fd_sqn(z_250_0, 2), z);
}
// SQRT_RATIO_M1; u, v tight.  Returns was_square; out = |sqrt(u/v)| or |sqrt(i u/v)|, tight.
HDN inline bool fd_sqrt_ratio_i(fd &out, const fd &u, const fd &v) {
    fd sqrtm1 = fd_unpack(fe_sqrtm1());
    fd v3 = fd_mul(fd_sq(v), v);
    fd v7 = fd_mul(fd_sq(v3), v);
    fd r = fd_mul(fd_mul(u, v3), fd_pow22523(fd_mul(u, v7)));
    fd check = fd_mul(v, fd_sq(r));
    fd neg_u = fd_carry(fd_neg(u));
    bool correct = fd_eq(check, u);
    bool flipped = fd_eq(check, neg_u);
    bool flipped_i = fd_eq(check, fd_mul(neg_u, sqrtm1));
    fd r_prime = fd_mul(r, sqrtm1);
    r = fd_select(r, r_prime, flipped || flipped_i);
    out = fd_abs(r);
    return correct || flipped;
}
HDN inline void gd_ristretto_encode(uint8_t *s, const gd &p) {
    fd sqrtm1 = fd_unpack(fe_sqrtm1());
    fd X = fd_carry(p.X), Y = fd_carry(p.Y), Z = fd_carry(p.Z), T = fd_carry(p.T);
    fd u1 = fd_mul(fd_add(Z, Y), fd_sub(Z, Y));
    fd u2 = fd_mul(X, Y);
    fd invsqrt; fd_sqrt_ratio_i(invsqrt, fd_one(), fd_mul(u1, fd_sq(u2)));
    fd den1 = fd_mul(invsqrt, u1), den2 = fd_mul(invsqrt, u2);
    fd z_inv = fd_mul(fd_mul(den1, den2), T);
    fd ix0 = fd_mul(X, sqrtm1), iy0 = fd_mul(Y, sqrtm1);
    fd ench = fd_mul(den1, fd_unpack(fe_invsqrt_a_minus_d()));
    bool rotate = fd_isneg(fd_mul(T, z_inv));
    fd x = fd_select(X, iy0, rotate), y = fd_select(Y, ix0, rotate);
    fd den_inv = fd_select(den2, ench, rotate);
    if (fd_isneg(fd_mul(x, z_inv))) y = fd_carry(fd_neg(y));
    fd sres = fd_abs(fd_mul(den_inv, fd_sub(Z, y)));
    fe_tobytes(s, fd_pack(sres));
}
// Output has Z = 1 and tight coordinates.
HDN inline bool gd_ristretto_decode(gd &p, const uint8_t *sb) {
    fe s8 = fe_frombytes(sb);
    uint8_t chk[32]; fe_tobytes(chk, s8);
    bool canon = true;
    for (int i = 0; i < 32; i++) canon &= (chk[i] == sb[i]);
    if (!canon || (sb[0] & 1)) return false;
    fd s = fd_unpack(s8);
    fd ss = fd_sq(s);
    fd u1 = fd_carry(fd_sub(fd_one(), ss)), u2 = fd_carry(fd_add(fd_one(), ss));
    fd u2s = fd_sq(u2);
    fd v = fd_carry(fd_sub(fd_carry(fd_neg(fd_mul(fd_unpack(fe_d()), fd_sq(u1)))), u2s));
    fd invsqrt; bool ok = fd_sqrt_ratio_i(invsqrt, fd_one(), fd_mul(v, u2s));
    fd dx = fd_mul(invsqrt, u2);
    fd dy = fd_mul(fd_mul(invsqrt, dx), v);
    fd x = fd_abs(fd_mul(fd_add(s, s), dx));
    fd y = fd_mul(u1, dy);
    fd t = fd_mul(x, y);
    if (!ok || fd_isneg(t) || fd_iszero(y)) return false;
    p.X = x; p.Y = y; p.Z = fd_one(); p.T = t;
    return true;
}

}  // namespace rofl
