// Keccak-f[1600] (host + device), SHA3-512 / SHAKE256 helpers, STROBE-128 / Merlin transcript (host).
//
// Replaces the reference's un-vendored crates sha3 0.9.1 / keccak 0.1.0 / merlin 3.0.0
// (Cargo.lock:1419-1422, 761-764, 830-833) as used from rofl_crypto/src/range_proof_vec/mod.rs:124,200
// (Transcript::new(b"RangeProof")) and bulletproofs' generators / transcript protocol.
// Merlin stays on the host: it is a strictly sequential duplex sponge ("host" row).
#pragma once
#include "fe32.hpp"

namespace rofl {

HD u64 rotl64(u64 x, int n) { return (x << n) | (x >> (64 - n)); }

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
struct Sponge {
    u64 st[25]; size_t pos, rate; bool squeezing; uint8_t suffix;
    Sponge(size_t rate_, uint8_t suffix_) : pos(0), rate(rate_), squeezing(false), suffix(suffix_) { memset(st, 0, sizeof st); }
    void xor_byte(size_t p, uint8_t b) { st[p >> 3] ^= (u64)b << (8 * (p & 7)); }
    void absorb(const uint8_t *d, size_t n) {
        for (size_t i = 0; i < n; i++) { xor_byte(pos++, d[i]); if (pos == rate) { keccak_f1600(st); pos = 0; } }
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
    void perm() { keccak_f1600(stw); }
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
