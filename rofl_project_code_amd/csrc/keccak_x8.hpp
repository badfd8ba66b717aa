// Eight Merlin transcripts per AVX-512 instruction stream (host only).
//
// Why: the verifier hashes the m commitments of every chunk into that chunk's transcript before it can draw y and z -- a sequential duplex
// sponge, 41 transcript bytes per commitment, one Keccak-f[1600] per 166 bytes: 0.53 ms per chunk of 8 192 commitments on a host core, with
// the GPU waiting.  A server that verifies a batch of 48 clients of d = 55 000 (BASELINE cfg 4) has 192 such chunks: 13 ms on sixteen
// cores, more than any kernel of the check.  The chunks are independent and -- having absorbed messages of the same lengths since
// Transcript::new -- their STROBE bookkeeping (pos, pos_begin, cur_flags) is IDENTICAL at every step; only the data differs.  So eight
// transcripts share one instruction stream: the state is held lane-wise (25 x __m512i, one transcript per 64-bit lane), the bytes to absorb
// are collected per transcript in a pending rate block and XORed in -- transposed -- right before each permutation.
// Run-time CPU check (k8::available); the scalar path of keccak.hpp stays for hosts without AVX-512 and for fewer than five transcripts.
#pragma once
#include <immintrin.h>
#include <string.h>
#include "keccak.hpp"

namespace rofl {
namespace k8 {

#define ROFL_K8 __attribute__((target("avx512f"))) inline
typedef __m512i V;

inline bool available() { static const bool ok = __builtin_cpu_supports("avx512f"); return ok; }

ROFL_K8 V x3(V a, V b, V c) { return _mm512_ternarylogic_epi64(a, b, c, 0x96); }       // a ^ b ^ c
ROFL_K8 V chi(V a, V b, V c) { return _mm512_ternarylogic_epi64(a, b, c, 0xD2); }      // a ^ (~b & c)
ROFL_K8 V x2(V a, V b) { return _mm512_xor_si512(a, b); }
#define K8_ROL(x, n) _mm512_rol_epi64((x), (n))

// one round, the same schedule as ROUND in keccak.hpp (state names: row b g k m s, column a e i o u)
#define K8_ROUND(A, E, rc) \
    { V Ca = x3(x3(A##ba, A##ga, A##ka), A##ma, A##sa), Ce = x3(x3(A##be, A##ge, A##ke), A##me, A##se), Ci = x3(x3(A##bi, A##gi, A##ki), A##mi, A##si), \
        Co = x3(x3(A##bo, A##go, A##ko), A##mo, A##so), Cu = x3(x3(A##bu, A##gu, A##ku), A##mu, A##su); \
      V Da = x2(Cu, K8_ROL(Ce, 1)), De = x2(Ca, K8_ROL(Ci, 1)), Di = x2(Ce, K8_ROL(Co, 1)), Do = x2(Ci, K8_ROL(Cu, 1)), Du = x2(Co, K8_ROL(Ca, 1)); \
      V B0, B1, B2, B3, B4; \
      B0 = x2(A##ba, Da); B1 = K8_ROL(x2(A##ge, De), 44); B2 = K8_ROL(x2(A##ki, Di), 43); B3 = K8_ROL(x2(A##mo, Do), 21); B4 = K8_ROL(x2(A##su, Du), 14); \
      E##ba = x2(chi(B0, B1, B2), _mm512_set1_epi64((long long)(rc))); E##be = chi(B1, B2, B3); E##bi = chi(B2, B3, B4); E##bo = chi(B3, B4, B0); E##bu = chi(B4, B0, B1); \
      B0 = K8_ROL(x2(A##bo, Do), 28); B1 = K8_ROL(x2(A##gu, Du), 20); B2 = K8_ROL(x2(A##ka, Da), 3); B3 = K8_ROL(x2(A##me, De), 45); B4 = K8_ROL(x2(A##si, Di), 61); \
      E##ga = chi(B0, B1, B2); E##ge = chi(B1, B2, B3); E##gi = chi(B2, B3, B4); E##go = chi(B3, B4, B0); E##gu = chi(B4, B0, B1); \
      B0 = K8_ROL(x2(A##be, De), 1); B1 = K8_ROL(x2(A##gi, Di), 6); B2 = K8_ROL(x2(A##ko, Do), 25); B3 = K8_ROL(x2(A##mu, Du), 8); B4 = K8_ROL(x2(A##sa, Da), 18); \
      E##ka = chi(B0, B1, B2); E##ke = chi(B1, B2, B3); E##ki = chi(B2, B3, B4); E##ko = chi(B3, B4, B0); E##ku = chi(B4, B0, B1); \
      B0 = K8_ROL(x2(A##bu, Du), 27); B1 = K8_ROL(x2(A##ga, Da), 36); B2 = K8_ROL(x2(A##ke, De), 10); B3 = K8_ROL(x2(A##mi, Di), 15); B4 = K8_ROL(x2(A##so, Do), 56); \
      E##ma = chi(B0, B1, B2); E##me = chi(B1, B2, B3); E##mi = chi(B2, B3, B4); E##mo = chi(B3, B4, B0); E##mu = chi(B4, B0, B1); \
      B0 = K8_ROL(x2(A##bi, Di), 62); B1 = K8_ROL(x2(A##go, Do), 55); B2 = K8_ROL(x2(A##ku, Du), 39); B3 = K8_ROL(x2(A##ma, Da), 41); B4 = K8_ROL(x2(A##se, De), 2); \
      E##sa = chi(B0, B1, B2); E##se = chi(B1, B2, B3); E##si = chi(B2, B3, B4); E##so = chi(B3, B4, B0); E##su = chi(B4, B0, B1); }

ROFL_K8 void keccak_f1600_x8(V s[25]) {
    static const u64 RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
        0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
        0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    V Aba = s[0], Abe = s[1], Abi = s[2], Abo = s[3], Abu = s[4], Aga = s[5], Age = s[6], Agi = s[7], Ago = s[8], Agu = s[9];
    V Aka = s[10], Ake = s[11], Aki = s[12], Ako = s[13], Aku = s[14], Ama = s[15], Ame = s[16], Ami = s[17], Amo = s[18], Amu = s[19];
    V Asa = s[20], Ase = s[21], Asi = s[22], Aso = s[23], Asu = s[24];
    V Eba, Ebe, Ebi, Ebo, Ebu, Ega, Ege, Egi, Ego, Egu, Eka, Eke, Eki, Eko, Eku, Ema, Eme, Emi, Emo, Emu, Esa, Ese, Esi, Eso, Esu;
    for (int r = 0; r < 24; r += 2) { K8_ROUND(A, E, RC[r]) K8_ROUND(E, A, RC[r + 1]) }
    s[0] = Aba; s[1] = Abe; s[2] = Abi; s[3] = Abo; s[4] = Abu; s[5] = Aga; s[6] = Age; s[7] = Agi; s[8] = Ago; s[9] = Agu;
    s[10] = Aka; s[11] = Ake; s[12] = Aki; s[13] = Ako; s[14] = Aku; s[15] = Ama; s[16] = Ame; s[17] = Ami; s[18] = Amo; s[19] = Amu;
    s[20] = Asa; s[21] = Ase; s[22] = Asi; s[23] = Aso; s[24] = Asu;
}
#undef K8_ROUND

// `lanes` (1..8) transcripts in the same STROBE position: count x append("<label>", msg[l] + 32 j, 32) on each -- what
// Merlin::append32_run does for one.  The transcripts are read at the start and written back at the end.
struct Run8 {
    static const int R = Merlin::R;        // 166: STROBE-128 rate minus its two padding bytes
    static const int BW = 22;              // pending block: 21 rate words + 1 (keeps the lanes 16-byte apart for the gathers)
    alignas(64) u64 blk[8][BW];            // bytes still to be XORed into the state, per transcript
    V S[25];
    uint8_t pos, pos_begin;
    int lanes;

    ROFL_K8 void flush_block() {           // blk -> state (transposed), blk = 0
        const V idx = _mm512_setr_epi64(0, BW, 2 * BW, 3 * BW, 4 * BW, 5 * BW, 6 * BW, 7 * BW);
        for (int w = 0; w < 21; w++) S[w] = _mm512_xor_si512(S[w], _mm512_i64gather_epi64(idx, (const long long *)&blk[0][w], 8));
        memset(blk, 0, sizeof blk);
    }
    ROFL_K8 void run_f() {                 // Merlin::run_f on every lane
        for (int l = 0; l < 8; l++) { uint8_t *b = reinterpret_cast<uint8_t *>(blk[l]); b[pos] ^= pos_begin; b[pos + 1] ^= 0x04; b[R + 1] ^= 0x80; }
        flush_block();
        keccak_f1600_x8(S);
        pos = 0; pos_begin = 0;
    }
    ROFL_K8 void load(Merlin *const t[8], int n) {
        lanes = n; pos = t[0]->pos; pos_begin = t[0]->pos_begin;
        alignas(64) u64 tmp[25][8];
        for (int l = 0; l < 8; l++) { const Merlin *m = t[l < n ? l : 0]; for (int w = 0; w < 25; w++) tmp[w][l] = m->stw[w]; }
        for (int w = 0; w < 25; w++) S[w] = _mm512_load_si512((const void *)tmp[w]);
        memset(blk, 0, sizeof blk);
    }
    ROFL_K8 void store(Merlin *const t[8]) {
        flush_block();
        alignas(64) u64 tmp[25][8];
        for (int w = 0; w < 25; w++) _mm512_store_si512((void *)tmp[w], S[w]);
        for (int l = 0; l < lanes; l++) { for (int w = 0; w < 25; w++) t[l]->stw[w] = tmp[w][l]; t[l]->pos = pos; t[l]->pos_begin = pos_begin; t[l]->cur_flags = 2; }
    }
    // one byte per lane (uniform = the same byte on all of them), with the permutation when the block fills up
    ROFL_K8 void put_uniform(uint8_t b) { for (int l = 0; l < 8; l++) reinterpret_cast<uint8_t *>(blk[l])[pos] ^= b; if (++pos == R) run_f(); }
    ROFL_K8 void put_lanes(const uint8_t *const src[8], size_t off) { for (int l = 0; l < 8; l++) reinterpret_cast<uint8_t *>(blk[l])[pos] ^= src[l][off]; if (++pos == R) run_f(); }
    ROFL_K8 void append32(char label, const uint8_t *const msg[8], size_t count) {
        for (size_t j = 0; j < count; j++) {
            const size_t off = 32 * j;
            if ((unsigned)pos + 41 < (unsigned)R) {      // the whole record stays inside the rate block: 9 header bytes + 32 message bytes
                const uint8_t hdr[9] = {pos_begin, 16 | 2, (uint8_t)label, 32, 0, 0, 0, (uint8_t)(pos + 1), 2};
                for (int l = 0; l < 8; l++) {
                    uint8_t *b = reinterpret_cast<uint8_t *>(blk[l]) + pos;
                    for (int i = 0; i < 9; i++) b[i] ^= hdr[i];
                    const uint8_t *m = msg[l] + off;
                    for (int i = 0; i < 32; i += 8) { u64 x, y; memcpy(&x, b + 9 + i, 8); memcpy(&y, m + i, 8); x ^= y; memcpy(b + 9 + i, &x, 8); }
                }
                pos_begin = (uint8_t)(pos + 8); pos = (uint8_t)(pos + 41);
            } else {                                      // byte by byte, exactly as meta_ad(label) + meta_ad(len, more) + ad(msg) would
                // begin_op: header {old pos_begin, flags}; pos_begin becomes (position of the header) + 1 BEFORE the header is absorbed
                // (a permutation that falls inside the header XORs the new value and then resets it, as Merlin::run_f does)
                { uint8_t old = pos_begin; pos_begin = (uint8_t)(pos + 1); put_uniform(old); put_uniform(16 | 2); }
                put_uniform((uint8_t)label);
                put_uniform(32); put_uniform(0); put_uniform(0); put_uniform(0);
                { uint8_t old = pos_begin; pos_begin = (uint8_t)(pos + 1); put_uniform(old); put_uniform(2); }
                for (int i = 0; i < 32; i++) put_lanes(msg, off + i);
            }
        }
    }
};

// Merlin::append32_run for up to eight transcripts at once.  All of them must be in the same position (same label, same appends so far).
ROFL_K8 void append32_run_x8(Merlin *const t[8], int lanes, char label, const uint8_t *const msg[8], size_t count) {
    Run8 r; r.load(t, lanes);
    const uint8_t *m[8]; for (int l = 0; l < 8; l++) m[l] = msg[l < lanes ? l : 0];
    r.append32(label, m, count);
    r.store(t);
}

}  // namespace k8
}  // namespace rofl
