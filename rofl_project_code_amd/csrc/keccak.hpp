// Keccak-f[1600] (host + device), SHA3-512 / SHAKE256 helpers, STROBE-128 / Merlin transcript (host).
//
// Replaces the reference's un-vendored crates sha3 0.9.1 / keccak 0.1.0 / merlin 3.0.0
// (Cargo.lock:1419-1422, 761-764, 830-833) as used from rofl_crypto/src/range_proof_vec/mod.rs:124,200
// (Transcript::new(b"RangeProof")) and bulletproofs' generators / transcript protocol.
// Merlin stays on the host: it is a strictly sequential duplex sponge ("host" row).
#pragma once
#include <immintrin.h>
#include <time.h>
#include "fe32.hpp"

namespace rofl {

// (device: a 64-bit rotation as two 32-bit funnel shifts -- v_alignbit_b32 -- instead of two 64-bit shifts and an OR pair; n is a literal everywhere)
HD u64 rotl64(u64 x, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
    u32 lo = (u32)x, hi = (u32)(x >> 32);
    if (n >= 32) { u32 t = lo; lo = hi; hi = t; n -= 32; }
    if (n == 0) return ((u64)hi << 32) | lo;
    const u32 nh = __builtin_amdgcn_alignbit(hi, lo, 32 - n), nl = __builtin_amdgcn_alignbit(lo, hi, 32 - n);
    return ((u64)nh << 32) | nl;
#else
    return (x << n) | (x >> (64 - n));
#endif
}

HDN inline void keccak_f1600(u64 s[25]) {
    const u64 RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
        0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
        0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    for (int rnd = 0; rnd < 24; rnd++) {
        u64 c0 = s[0] ^ s[5] ^ s[10] ^ s[15] ^ s[20];
        u64 c1 = s[1] ^ s[6] ^ s[11] ^ s[16] ^ s[21];
        u64 c2 = s[2] ^ s[7] ^ s[12] ^ s[17] ^ s[22];
        u64 c3 = s[3] ^ s[8] ^ s[13] ^ s[18] ^ s[23];
        u64 c4 = s[4] ^ s[9] ^ s[14] ^ s[19] ^ s[24];
        u64 d0 = c4 ^ rotl64(c1, 1), d1 = c0 ^ rotl64(c2, 1), d2 = c1 ^ rotl64(c3, 1);
        u64 d3 = c2 ^ rotl64(c4, 1), d4 = c3 ^ rotl64(c0, 1);
#if defined(__HIP_DEVICE_COMPILE__) && defined(ROFL_KGROUP) && ROFL_KGROUP == 4
        // (kernel group 4 only -- k_nonce_expand, one permutation per thread and registers to spare; in the Sigma-proof transcript kernels of
        // group 3 the pinned words cost 18 VGPRs and a wave of occupancy)
        // keep the five D words as values: left alone, the compiler XORs c and rol(c) into the 25 lanes separately (100 XORs for 60)
        {   u32 l0 = (u32)d0, h0 = (u32)(d0 >> 32), l1 = (u32)d1, h1 = (u32)(d1 >> 32), l2 = (u32)d2, h2 = (u32)(d2 >> 32), l3 = (u32)d3, h3 = (u32)(d3 >> 32), l4 = (u32)d4, h4 = (u32)(d4 >> 32);
            asm volatile("" : "+v"(l0), "+v"(h0), "+v"(l1), "+v"(h1), "+v"(l2), "+v"(h2), "+v"(l3), "+v"(h3), "+v"(l4), "+v"(h4));
            d0 = ((u64)h0 << 32) | l0; d1 = ((u64)h1 << 32) | l1; d2 = ((u64)h2 << 32) | l2; d3 = ((u64)h3 << 32) | l3; d4 = ((u64)h4 << 32) | l4; }
#endif
#pragma unroll
        for (int j = 0; j < 25; j += 5) { s[j] ^= d0; s[j + 1] ^= d1; s[j + 2] ^= d2; s[j + 3] ^= d3; s[j + 4] ^= d4; }
        // rho + pi
        u64 t = s[1], b;
        b = s[10]; s[10] = rotl64(t, 1); t = b;
        b = s[7]; s[7] = rotl64(t, 3); t = b;
        b = s[11]; s[11] = rotl64(t, 6); t = b;
        b = s[17]; s[17] = rotl64(t, 10); t = b;
        b = s[18]; s[18] = rotl64(t, 15); t = b;
        b = s[3]; s[3] = rotl64(t, 21); t = b;
        b = s[5]; s[5] = rotl64(t, 28); t = b;
        b = s[16]; s[16] = rotl64(t, 36); t = b;
        b = s[8]; s[8] = rotl64(t, 45); t = b;
        b = s[21]; s[21] = rotl64(t, 55); t = b;
        b = s[24]; s[24] = rotl64(t, 2); t = b;
        b = s[4]; s[4] = rotl64(t, 14); t = b;
        b = s[15]; s[15] = rotl64(t, 27); t = b;
        b = s[23]; s[23] = rotl64(t, 41); t = b;
        b = s[19]; s[19] = rotl64(t, 56); t = b;
        b = s[13]; s[13] = rotl64(t, 8); t = b;
        b = s[12]; s[12] = rotl64(t, 25); t = b;
        b = s[2]; s[2] = rotl64(t, 43); t = b;
        b = s[20]; s[20] = rotl64(t, 62); t = b;
        b = s[14]; s[14] = rotl64(t, 18); t = b;
        b = s[22]; s[22] = rotl64(t, 39); t = b;
        b = s[9]; s[9] = rotl64(t, 61); t = b;
        b = s[6]; s[6] = rotl64(t, 20); t = b;
        b = s[1]; s[1] = rotl64(t, 44); t = b;
        // chi
#pragma unroll
        for (int j = 0; j < 25; j += 5) {
            u64 a0 = s[j], a1 = s[j + 1], a2 = s[j + 2], a3 = s[j + 3], a4 = s[j + 4];
            s[j] = a0 ^ (~a1 & a2); s[j + 1] = a1 ^ (~a2 & a3); s[j + 2] = a2 ^ (~a3 & a4);
            s[j + 3] = a3 ^ (~a4 & a0); s[j + 4] = a4 ^ (~a0 & a1);
        }
        s[0] ^= RC[rnd];
    }
}

// SHAKE256 of a short (< 136 byte) message given as whole little-endian u64 words plus tail bytes,
// squeezing exactly one block.  Used for the nonce / verifier-challenge DRBG:
//   SHAKE256(domain16 || seed32 || u64le(index))   (56 bytes = 7 words)
#define ROFL_NONCE_DOM {0x2f6b7a2d6c666f72ULL, 0x32762f65636e6f6eULL}      /* "rofl-zk/" "nonce/v2" */
HD void shake256_seeded_block(u64 st[25], const u64 dom[2], const u64 seed[4], u64 index) {
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = 0;
    st[0] = dom[0]; st[1] = dom[1];
    st[2] = seed[0]; st[3] = seed[1]; st[4] = seed[2]; st[5] = seed[3];
    st[6] = index;
    st[7] ^= 0x1FULL;                    // SHAKE domain suffix right after 56 bytes
    st[16] ^= 0x8000000000000000ULL;     // final bit of the 136-byte rate
    keccak_f1600(st);
}

// ---------------------------------------------------------------- host-only sponge helpers
// Host-only Keccak-f[1600] for the transcripts: two rounds per iteration with the state in locals (clang keeps the in-place form above,
// which the kernels use, ~25 % slower on x86-64 -- and the verifier hashes 8 192 commitments per chunk with the GPU waiting).
inline u64 rol(u64 x, int n) { return (x << n) | (x >> (64 - n)); }
#define ROUND(A, E, rc) \
    { u64 Ca = A##ba ^ A##ga ^ A##ka ^ A##ma ^ A##sa, Ce = A##be ^ A##ge ^ A##ke ^ A##me ^ A##se, Ci = A##bi ^ A##gi ^ A##ki ^ A##mi ^ A##si, \
          Co = A##bo ^ A##go ^ A##ko ^ A##mo ^ A##so, Cu = A##bu ^ A##gu ^ A##ku ^ A##mu ^ A##su; \
      u64 Da = Cu ^ rol(Ce, 1), De = Ca ^ rol(Ci, 1), Di = Ce ^ rol(Co, 1), Do = Ci ^ rol(Cu, 1), Du = Co ^ rol(Ca, 1); \
      u64 B0, B1, B2, B3, B4; \
      B0 = A##ba ^ Da; B1 = rol(A##ge ^ De, 44); B2 = rol(A##ki ^ Di, 43); B3 = rol(A##mo ^ Do, 21); B4 = rol(A##su ^ Du, 14); \
      E##ba = B0 ^ (~B1 & B2) ^ rc; E##be = B1 ^ (~B2 & B3); E##bi = B2 ^ (~B3 & B4); E##bo = B3 ^ (~B4 & B0); E##bu = B4 ^ (~B0 & B1); \
      B0 = rol(A##bo ^ Do, 28); B1 = rol(A##gu ^ Du, 20); B2 = rol(A##ka ^ Da, 3); B3 = rol(A##me ^ De, 45); B4 = rol(A##si ^ Di, 61); \
      E##ga = B0 ^ (~B1 & B2); E##ge = B1 ^ (~B2 & B3); E##gi = B2 ^ (~B3 & B4); E##go = B3 ^ (~B4 & B0); E##gu = B4 ^ (~B0 & B1); \
      B0 = rol(A##be ^ De, 1); B1 = rol(A##gi ^ Di, 6); B2 = rol(A##ko ^ Do, 25); B3 = rol(A##mu ^ Du, 8); B4 = rol(A##sa ^ Da, 18); \
      E##ka = B0 ^ (~B1 & B2); E##ke = B1 ^ (~B2 & B3); E##ki = B2 ^ (~B3 & B4); E##ko = B3 ^ (~B4 & B0); E##ku = B4 ^ (~B0 & B1); \
      B0 = rol(A##bu ^ Du, 27); B1 = rol(A##ga ^ Da, 36); B2 = rol(A##ke ^ De, 10); B3 = rol(A##mi ^ Di, 15); B4 = rol(A##so ^ Do, 56); \
      E##ma = B0 ^ (~B1 & B2); E##me = B1 ^ (~B2 & B3); E##mi = B2 ^ (~B3 & B4); E##mo = B3 ^ (~B4 & B0); E##mu = B4 ^ (~B0 & B1); \
      B0 = rol(A##bi ^ Di, 62); B1 = rol(A##go ^ Do, 55); B2 = rol(A##ku ^ Du, 39); B3 = rol(A##ma ^ Da, 41); B4 = rol(A##se ^ De, 2); \
      E##sa = B0 ^ (~B1 & B2); E##se = B1 ^ (~B2 & B3); E##si = B2 ^ (~B3 & B4); E##so = B3 ^ (~B4 & B0); E##su = B4 ^ (~B0 & B1); }
#define ROFL_KECCAK_BODY \
\
    static const u64 RC[24] = {\
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,\
        0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,\
        0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,\
        0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,\
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,\
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};\
    u64 Aba = s[0], Abe = s[1], Abi = s[2], Abo = s[3], Abu = s[4], Aga = s[5], Age = s[6], Agi = s[7], Ago = s[8], Agu = s[9];\
    u64 Aka = s[10], Ake = s[11], Aki = s[12], Ako = s[13], Aku = s[14], Ama = s[15], Ame = s[16], Ami = s[17], Amo = s[18], Amu = s[19];\
    u64 Asa = s[20], Ase = s[21], Asi = s[22], Aso = s[23], Asu = s[24];\
    u64 Eba, Ebe, Ebi, Ebo, Ebu, Ega, Ege, Egi, Ego, Egu, Eka, Eke, Eki, Eko, Eku, Ema, Eme, Emi, Emo, Emu, Esa, Ese, Esi, Eso, Esu;\
    for (int r = 0; r < 24; r += 2) { ROUND(A, E, RC[r]) ROUND(E, A, RC[r + 1]) }\
    s[0] = Aba; s[1] = Abe; s[2] = Abi; s[3] = Abo; s[4] = Abu; s[5] = Aga; s[6] = Age; s[7] = Agi; s[8] = Ago; s[9] = Agu;\
    s[10] = Aka; s[11] = Ake; s[12] = Aki; s[13] = Ako; s[14] = Aku; s[15] = Ama; s[16] = Ame; s[17] = Ami; s[18] = Amo; s[19] = Amu;\
    s[20] = Asa; s[21] = Ase; s[22] = Asi; s[23] = Aso; s[24] = Asu;
inline void keccak_f1600_plain(u64 s[25]) { ROFL_KECCAK_BODY }
// the same rounds compiled for BMI (andn for chi's ~b & c, rorx): 25 fewer operations a round; chosen once per process
__attribute__((target("bmi,bmi2"))) inline void keccak_f1600_bmi(u64 s[25]) { ROFL_KECCAK_BODY }
inline void keccak_f1600_host(u64 s[25]) {
    static const bool bmi = __builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2");
    if (bmi) keccak_f1600_bmi(s); else keccak_f1600_plain(s);
}
#undef ROFL_KECCAK_BODY
#undef ROUND

// One Keccak-f[1600] state across AVX-512 registers (host only; run-time CPU check).  Why: a transcript is a sequential sponge, and the
// single-client verifier waits for one of 1 579 permutations per chunk with the GPU idle (keccak_x8.hpp covers the case of many transcripts).
// Layout: one register per plane y, lane (x, y) in slot x.  theta and rho are slot-wise; pi becomes one in-register permutation per plane
// that leaves register x' = y holding the lanes (x', y') of the new state in slot y' -- so chi is three-operand logic between registers --
// and a 5 x 5 transposition (ten two-source permutations, two blends) restores the plane layout.  17 shuffles + 20 other operations per
// round against ~150 scalar ones.
#define ROFL_K1 __attribute__((target("avx512f"))) inline
ROFL_K1 void keccak_f1600_zmm(u64 s[25]) {
    typedef __m512i V;
    static const u64 RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
        0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
        0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    const __mmask8 M5 = 0x1F;
    V P0 = _mm512_maskz_loadu_epi64(M5, s), P1 = _mm512_maskz_loadu_epi64(M5, s + 5), P2 = _mm512_maskz_loadu_epi64(M5, s + 10),
      P3 = _mm512_maskz_loadu_epi64(M5, s + 15), P4 = _mm512_maskz_loadu_epi64(M5, s + 20);
    const V rho0 = _mm512_setr_epi64(0, 1, 62, 28, 27, 0, 0, 0), rho1 = _mm512_setr_epi64(36, 44, 6, 55, 20, 0, 0, 0),
            rho2 = _mm512_setr_epi64(3, 10, 43, 25, 39, 0, 0, 0), rho3 = _mm512_setr_epi64(41, 45, 15, 21, 8, 0, 0, 0),
            rho4 = _mm512_setr_epi64(18, 2, 61, 56, 14, 0, 0, 0);
    const V xm1 = _mm512_setr_epi64(4, 0, 1, 2, 3, 5, 6, 7), xp1 = _mm512_setr_epi64(1, 2, 3, 4, 0, 5, 6, 7);
    // pi: slot y' of register p takes lane x = p + 3 y' (mod 5) of plane p
    const V pi0 = _mm512_setr_epi64(0, 3, 1, 4, 2, 5, 6, 7), pi1 = _mm512_setr_epi64(1, 4, 2, 0, 3, 5, 6, 7), pi2 = _mm512_setr_epi64(2, 0, 3, 1, 4, 5, 6, 7),
            pi3 = _mm512_setr_epi64(3, 1, 4, 2, 0, 5, 6, 7), pi4 = _mm512_setr_epi64(4, 2, 0, 3, 1, 5, 6, 7);
    // transposition: pairs of (R0, R1) and (R2, R3) for y = 0..3, fours for y = 0,1 and y = 2,3, then the slot of R4
    const V tpair = _mm512_setr_epi64(0, 8, 1, 9, 2, 10, 3, 11), tq01 = _mm512_setr_epi64(0, 1, 8, 9, 2, 3, 10, 11), tq23 = _mm512_setr_epi64(4, 5, 12, 13, 6, 7, 14, 15);
    const V tlo0 = _mm512_setr_epi64(0, 1, 2, 3, 8, 5, 6, 7), tlo1 = _mm512_setr_epi64(0, 1, 2, 3, 10, 5, 6, 7),
            thi0 = _mm512_setr_epi64(4, 5, 6, 7, 9, 5, 6, 7), thi1 = _mm512_setr_epi64(4, 5, 6, 7, 11, 5, 6, 7);
    const V t4a = _mm512_setr_epi64(4, 12, 0, 0, 0, 0, 0, 0), t4b = _mm512_setr_epi64(0, 0, 4, 12, 0, 0, 0, 0);
    for (int r = 0; r < 24; r++) {
        V C = _mm512_ternarylogic_epi64(_mm512_ternarylogic_epi64(P0, P1, P2, 0x96), P3, P4, 0x96);
        V Cm = _mm512_permutexvar_epi64(xm1, C), Cp = _mm512_rol_epi64(_mm512_permutexvar_epi64(xp1, C), 1);
        P0 = _mm512_rolv_epi64(_mm512_ternarylogic_epi64(P0, Cm, Cp, 0x96), rho0);
        P1 = _mm512_rolv_epi64(_mm512_ternarylogic_epi64(P1, Cm, Cp, 0x96), rho1);
        P2 = _mm512_rolv_epi64(_mm512_ternarylogic_epi64(P2, Cm, Cp, 0x96), rho2);
        P3 = _mm512_rolv_epi64(_mm512_ternarylogic_epi64(P3, Cm, Cp, 0x96), rho3);
        P4 = _mm512_rolv_epi64(_mm512_ternarylogic_epi64(P4, Cm, Cp, 0x96), rho4);
        V Q0 = _mm512_permutexvar_epi64(pi0, P0), Q1 = _mm512_permutexvar_epi64(pi1, P1), Q2 = _mm512_permutexvar_epi64(pi2, P2),
          Q3 = _mm512_permutexvar_epi64(pi3, P3), Q4 = _mm512_permutexvar_epi64(pi4, P4);
        V R0 = _mm512_ternarylogic_epi64(Q0, Q1, Q2, 0xD2), R1 = _mm512_ternarylogic_epi64(Q1, Q2, Q3, 0xD2), R2 = _mm512_ternarylogic_epi64(Q2, Q3, Q4, 0xD2),
          R3 = _mm512_ternarylogic_epi64(Q3, Q4, Q0, 0xD2), R4 = _mm512_ternarylogic_epi64(Q4, Q0, Q1, 0xD2);
        R0 = _mm512_xor_si512(R0, _mm512_zextsi128_si512(_mm_cvtsi64_si128((long long)RC[r])));
        V A = _mm512_permutex2var_epi64(R0, tpair, R1), B = _mm512_permutex2var_epi64(R2, tpair, R3);
        V F01 = _mm512_permutex2var_epi64(A, tq01, B), F23 = _mm512_permutex2var_epi64(A, tq23, B);
        P0 = _mm512_permutex2var_epi64(F01, tlo0, R4); P1 = _mm512_permutex2var_epi64(F01, thi0, R4);
        P2 = _mm512_permutex2var_epi64(F23, tlo1, R4); P3 = _mm512_permutex2var_epi64(F23, thi1, R4);
        V G = _mm512_mask_blend_epi64(0x0C, _mm512_permutex2var_epi64(R0, t4a, R1), _mm512_permutex2var_epi64(R2, t4b, R3));
        P4 = _mm512_mask_blend_epi64(0x10, G, R4);
    }
    _mm512_mask_storeu_epi64(s, M5, P0); _mm512_mask_storeu_epi64(s + 5, M5, P1); _mm512_mask_storeu_epi64(s + 10, M5, P2);
    _mm512_mask_storeu_epi64(s + 15, M5, P3); _mm512_mask_storeu_epi64(s + 20, M5, P4);
}
#undef ROFL_K1
// Which of the two a host runs is measured once (a few dozen microseconds): Intel cores with one-cycle vector logic gain 1.27x
// (267 against 339 ns per permutation on the build container's Xeon); Zen 5 -- two-cycle vector operations, a fast scalar core -- loses
// (252 against 204 ns on the MI355X boxes' EPYC 9575F) and keeps the scalar rounds.  Knob ROFL_KECCAK_ZMM = 0 / 1 overrides (host_rt.hpp).
inline bool keccak_zmm_calibrate() {
    if (!__builtin_cpu_supports("avx512f")) return false;
    auto ns = [](void (*f)(u64 *)) {
        u64 st[25]; for (int i = 0; i < 25; i++) st[i] = 0x0123456789abcdefULL * (u64)(i + 1);
        double best = 1e30;
        for (int t = 0; t < 6; t++) {
            timespec a, b; clock_gettime(CLOCK_MONOTONIC, &a);
            for (int i = 0; i < 48; i++) f(st);
            clock_gettime(CLOCK_MONOTONIC, &b);
            double d = (double)(b.tv_sec - a.tv_sec) * 1e9 + (double)(b.tv_nsec - a.tv_nsec); if (d < best) best = d;
        }
        return best + (double)(st[0] & 1);
    };
    return ns(keccak_f1600_zmm) < 0.92 * ns(keccak_f1600_host);
}
inline bool &keccak_zmm_flag() { static bool on = keccak_zmm_calibrate(); return on; }
inline void keccak_f1600_fast(u64 s[25]) { if (keccak_zmm_flag()) keccak_f1600_zmm(s); else keccak_f1600_host(s); }

struct Sponge {
    u64 st[25]; size_t pos, rate; bool squeezing; uint8_t suffix;
    Sponge(size_t rate_, uint8_t suffix_) : pos(0), rate(rate_), squeezing(false), suffix(suffix_) { memset(st, 0, sizeof st); }
    void xor_byte(size_t p, uint8_t b) { st[p >> 3] ^= (u64)b << (8 * (p & 7)); }
    void absorb(const uint8_t *d, size_t n) {
        for (size_t i = 0; i < n; i++) { xor_byte(pos++, d[i]); if (pos == rate) { keccak_f1600_host(st); pos = 0; } }
    }
    void squeeze(uint8_t *o, size_t n) {
        if (!squeezing) { xor_byte(pos, suffix); xor_byte(rate - 1, 0x80); keccak_f1600(st); pos = 0; squeezing = true; }
        for (size_t i = 0; i < n; i++) { if (pos == rate) { keccak_f1600(st); pos = 0; } o[i] = (uint8_t)(st[pos >> 3] >> (8 * (pos & 7))); pos++; }
    }
};
inline void sha3_512(uint8_t out[64], const uint8_t *in, size_t n) { Sponge s(72, 0x06); s.absorb(in, n); s.squeeze(out, 64); }

// STROBE-128 as restricted by merlin 3.0.0 (strobe.rs): only AD, meta-AD and PRF.
struct Merlin {
    u64 stw[25];                                               // the sponge state; bytes through b() (Keccak-f runs on it in place)
    uint8_t pos, pos_begin, cur_flags;
    static const int R = 166;
    uint8_t *b() { return reinterpret_cast<uint8_t *>(stw); }
    const uint8_t *b() const { return reinterpret_cast<const uint8_t *>(stw); }
    void perm() { keccak_f1600_fast(stw); }
    void run_f() { uint8_t *st = b(); st[pos] ^= pos_begin; st[pos + 1] ^= 0x04; st[R + 1] ^= 0x80; perm(); pos = 0; pos_begin = 0; }
    // runs of bytes up to the end of the rate block, eight at a time (the verifier appends 8 192 commitments per chunk: 41 bytes each)
    void absorb(const uint8_t *d, size_t n) {
        while (n) {
            size_t take = (size_t)R - pos; if (take > n) take = n;
            uint8_t *st = b() + pos;
            size_t i = 0;
            for (; i + 8 <= take; i += 8) { u64 x, y; memcpy(&x, st + i, 8); memcpy(&y, d + i, 8); x ^= y; memcpy(st + i, &x, 8); }
            for (; i < take; i++) st[i] ^= d[i];
            pos = (uint8_t)(pos + take); d += take; n -= take;
            if (pos == R) run_f();
        }
    }
    void squeeze(uint8_t *d, size_t n) { uint8_t *st = b(); for (size_t i = 0; i < n; i++) { d[i] = st[pos]; st[pos] = 0; pos++; if (pos == R) run_f(); } }
    void begin_op(uint8_t flags, bool more) {
        if (more) return;
        uint8_t hdr[2] = {pos_begin, flags};
        pos_begin = (uint8_t)(pos + 1); cur_flags = flags;
        absorb(hdr, 2);
        if ((flags & (4 | 32)) && pos != 0) run_f();
    }
    void meta_ad(const uint8_t *d, size_t n, bool more) { begin_op(16 | 2, more); absorb(d, n); }
    void ad(const uint8_t *d, size_t n, bool more) { begin_op(2, more); absorb(d, n); }
    void prf(uint8_t *d, size_t n) { begin_op(1 | 2 | 4, false); squeeze(d, n); }

    explicit Merlin(const char *label, size_t len) {
        memset(stw, 0, 200); pos = pos_begin = cur_flags = 0;
        const uint8_t hdr[6] = {1, R + 2, 1, 0, 1, 96};
        memcpy(b(), hdr, 6); memcpy(b() + 6, "STROBEv1.0.2", 12);
        perm();
        meta_ad((const uint8_t *)"Merlin v1.0", 11, false);
        append("dom-sep", (const uint8_t *)label, len);
    }
    void append(const char *label, const uint8_t *msg, size_t len) {
        uint8_t le[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
        meta_ad((const uint8_t *)label, strlen(label), false);
        meta_ad(le, 4, true);
        ad(msg, len, false);
    }
    void append_lbl(const uint8_t *label, size_t ll, const uint8_t *msg, size_t len) {     // labels that may contain NUL bytes
        uint8_t le[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
        meta_ad(label, ll, false); meta_ad(le, 4, true); ad(msg, len, false);
    }
    // == count x append("<label>", msg + 32 j, 32) for a one-character label (the m commitments V_j of a chunk: 8 192 appends of 41
    // transcript bytes each: {pos_begin, META_AD} label len {pos_begin', AD} msg).  The record is put together in registers with both STROBE
    // headers computed directly and XORed into the rate block eight bytes at a time.  A record that reaches the end of the block (one in
    // four) is split there: what begin_op / run_f would have done to pos_begin on the way is a function of k, the bytes left in the block --
    // the second header starts at record byte 7, so a block that ends within the first seven bytes (k <= 7) permutes with pos_begin =
    // start + 1 and the second operation begins in the new block with pos_begin 0; otherwise the permutation sees start + 8.
    static void xor_bytes(uint8_t *dst, const uint8_t *src, size_t n) {
        size_t i = 0;
        for (; i + 8 <= n; i += 8) { u64 x, y; memcpy(&x, dst + i, 8); memcpy(&y, src + i, 8); x ^= y; memcpy(dst + i, &x, 8); }
        for (; i < n; i++) dst[i] ^= src[i];
    }
    void append32_run(char label, const uint8_t *msg, size_t count) {
        for (size_t j = 0; j < count; j++, msg += 32) {
            const unsigned k = (unsigned)R - pos;                 // bytes left in the rate block, 1..166
            u64 m0, m1, m2, m3; memcpy(&m0, msg, 8); memcpy(&m1, msg + 8, 8); memcpy(&m2, msg + 16, 8); memcpy(&m3, msg + 24, 8);
            const u64 w0 = (u64)pos_begin | (u64)(16 | 2) << 8 | (u64)(uint8_t)label << 16 | (u64)32 << 24 | (u64)(k <= 7 ? 0 : (uint8_t)(pos + 1)) << 56;
            const u64 w1 = 2 | m0 << 8, w2 = m0 >> 56 | m1 << 8, w3 = m1 >> 56 | m2 << 8, w4 = m2 >> 56 | m3 << 8;
            const uint8_t b40 = (uint8_t)(m3 >> 56);
            uint8_t *st = b() + pos;
            if (k > 41) {
                u64 x;
                memcpy(&x, st, 8); x ^= w0; memcpy(st, &x, 8);
                memcpy(&x, st + 8, 8); x ^= w1; memcpy(st + 8, &x, 8);
                memcpy(&x, st + 16, 8); x ^= w2; memcpy(st + 16, &x, 8);
                memcpy(&x, st + 24, 8); x ^= w3; memcpy(st + 24, &x, 8);
                memcpy(&x, st + 32, 8); x ^= w4; memcpy(st + 32, &x, 8);
                st[40] ^= b40;
                pos_begin = (uint8_t)(pos + 8); pos = (uint8_t)(pos + 41);
            } else {
                uint8_t rec[48];
                memcpy(rec, &w0, 8); memcpy(rec + 8, &w1, 8); memcpy(rec + 16, &w2, 8); memcpy(rec + 24, &w3, 8); memcpy(rec + 32, &w4, 8); rec[40] = b40;
                xor_bytes(st, rec, k);
                uint8_t *s0 = b();
                s0[R] ^= (uint8_t)(pos + (k <= 7 ? 1 : 8)); s0[R + 1] ^= 0x04 ^ 0x80;
                perm();
                xor_bytes(s0, rec + k, 41 - k);
                pos = (uint8_t)(41 - k); pos_begin = (uint8_t)(k <= 7 ? 8 - k : 0);
            }
        }
        if (count) cur_flags = 2;
    }
    void append_u64(const char *label, u64 x) { uint8_t b[8]; memcpy(b, &x, 8); append(label, b, 8); }
    void append_scalar(const char *label, const sc &s) { uint8_t b[32]; sc_tobytes(b, s); append(label, b, 32); }
    void challenge_bytes(const char *label, uint8_t *out, size_t len) {
        uint8_t le[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
        meta_ad((const uint8_t *)label, strlen(label), false);
        meta_ad(le, 4, true);
        prf(out, len);
    }
    // canonical (non-Montgomery) scalar, Scalar::from_bytes_mod_order_wide
    sc challenge_scalar(const char *label) {
        uint8_t b[64]; challenge_bytes(label, b, 64);
        return sc_from_wide(sc_frombytes(b), sc_frombytes(b + 32));
    }
};

}  // namespace rofl
