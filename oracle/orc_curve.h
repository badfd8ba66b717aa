/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement of the arithmetic under rofl_crypto's range-proof path:
 * GF(2^255-19), the scalar field mod l, Edwards/Ristretto255 group, Keccak-f[1600]
 * (SHA3-512, SHAKE256), STROBE-128 / Merlin.  These live in un-vendored third-party
 * crates of the reference (Cargo.lock: curve25519-dalek-ng 4.1.1, merlin 3.0.0,
 * sha3 0.9.1, keccak 0.1.0); the published algorithms (RFC 9496, FIPS 202,
 * STROBE v1.0.2, Merlin v1.0) are restated here.
 *
 * Representation is deliberately different from the HIP product (5x51-bit limbs with
 * unsigned __int128 here; 8x32-bit saturated limbs there) so that the two are
 * independent implementations.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
 */
#ifndef ORC_CURVE_H
#define ORC_CURVE_H
#include <stdint.h>
#include <stddef.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[5]; } fe;
typedef struct { uint64_t v[4]; } sc;          /* canonical, < l, little-endian limbs */
typedef struct { fe X, Y, Z, T; } ge;          /* extended Edwards, a = -1 */

/* ---- field ---- */
void fe_frombytes(fe *h, const uint8_t s[32]);
void fe_tobytes(uint8_t s[32], const fe *h);
void fe_0(fe *h);
void fe_1(fe *h);
void fe_add(fe *h, const fe *f, const fe *g);
void fe_sub(fe *h, const fe *f, const fe *g);
void fe_neg(fe *h, const fe *f);
void fe_mul(fe *h, const fe *f, const fe *g);
void fe_sq(fe *h, const fe *f);
void fe_invert(fe *out, const fe *z);
int fe_isneg(const fe *f);
int fe_iszero(const fe *f);
int fe_eq(const fe *f, const fe *g);
void fe_cmov(fe *f, const fe *g, int b);
void fe_abs(fe *h, const fe *f);
int fe_sqrt_ratio_i(fe *r, const fe *u, const fe *v);

/* ---- scalars ---- */
extern const sc SC_ZERO, SC_ONE;
void sc_frombytes_wide(sc *r, const uint8_t s[64]);
void sc_frombytes_modorder(sc *r, const uint8_t s[32]);
int sc_frombytes_canonical(sc *r, const uint8_t s[32]); /* 1 ok, 0 not canonical */
void sc_tobytes(uint8_t s[32], const sc *a);
void sc_from_u64(sc *r, uint64_t x);
void sc_add(sc *r, const sc *a, const sc *b);
void sc_sub(sc *r, const sc *a, const sc *b);
void sc_neg(sc *r, const sc *a);
void sc_mul(sc *r, const sc *a, const sc *b);
void sc_invert(sc *r, const sc *a);
int sc_iszero(const sc *a);
int sc_eq(const sc *a, const sc *b);

/* ---- group ---- */
extern ge GE_BASE;          /* Ristretto basepoint B */
extern ge GE_BBLIND;        /* B_blinding = from_uniform_bytes(SHA3-512(compress(B))) */
void orc_init(void);
void ge_identity(ge *p);
void ge_add(ge *r, const ge *p, const ge *q);
void ge_sub(ge *r, const ge *p, const ge *q);
void ge_neg(ge *r, const ge *p);
void ge_double(ge *r, const ge *p);
int ge_is_identity_ristretto(const ge *p);
int ge_eq_ristretto(const ge *p, const ge *q);
void ge_scalarmult(ge *r, const sc *k, const ge *p);
void ge_double_scalarmult(ge *r, const sc *a, const ge *A, const sc *b, const ge *B);
void ge_msm(ge *r, const sc *k, const ge *p, size_t n);
void ristretto_encode(uint8_t s[32], const ge *p);
int ristretto_decode(ge *p, const uint8_t s[32]);  /* 1 ok, 0 invalid */
void ristretto_from_uniform(ge *p, const uint8_t b[64]);

/* ---- hashes ---- */
void keccak_f1600(uint64_t st[25]);
void sha3_512(uint8_t out[64], const uint8_t *in, size_t len);
typedef struct { uint64_t st[25]; size_t pos; int squeezing; } shake256_ctx;
void shake256_init(shake256_ctx *c);
void shake256_absorb(shake256_ctx *c, const uint8_t *in, size_t len);
void shake256_squeeze(shake256_ctx *c, uint8_t *out, size_t len);

/* ---- Merlin ---- */
typedef struct { uint8_t st[200]; uint8_t pos, pos_begin, cur_flags; } merlin_t;
void merlin_init(merlin_t *t, const uint8_t *label, size_t len);
void merlin_append(merlin_t *t, const char *label, const uint8_t *msg, size_t len);
void merlin_append_lbl(merlin_t *t, const uint8_t *label, size_t ll, const uint8_t *msg, size_t len);   /* labels that may contain NUL */
void merlin_append_u64(merlin_t *t, const char *label, uint64_t x);
void merlin_challenge_bytes(merlin_t *t, const char *label, uint8_t *out, size_t len);
void merlin_challenge_scalar(merlin_t *t, const char *label, sc *out);

#endif
