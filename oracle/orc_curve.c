/*
 * ORACLE (test infrastructure, NOT product code) -- see orc_curve.h.
 * Field / scalar / group / hash / transcript primitives.
 */
#include "orc_curve.h"
#include <stdlib.h>

/* =====================================================================
 * GF(2^255-19), 5 x 51-bit limbs
 * ===================================================================== */
static const uint64_t M51 = 0x7ffffffffffffULL;

static uint64_t load64(const uint8_t *p) { uint64_t r; memcpy(&r, p, 8); return r; }
static void store64(uint8_t *p, uint64_t v) { memcpy(p, &v, 8); }

void fe_0(fe *h) { memset(h, 0, sizeof *h); }
void fe_1(fe *h) { memset(h, 0, sizeof *h); h->v[0] = 1; }

void fe_frombytes(fe *h, const uint8_t s[32]) {
    uint64_t t0 = load64(s), t1 = load64(s + 8), t2 = load64(s + 16), t3 = load64(s + 24);
    h->v[0] = t0 & M51;
    h->v[1] = ((t0 >> 51) | (t1 << 13)) & M51;
    h->v[2] = ((t1 >> 38) | (t2 << 26)) & M51;
    h->v[3] = ((t2 >> 25) | (t3 << 39)) & M51;
    h->v[4] = (t3 >> 12) & M51; /* bit 255 is ignored, as dalek FieldElement::from_bytes */
}

static void fe_carry(fe *h) {
    uint64_t c;
    c = h->v[0] >> 51; h->v[0] &= M51; h->v[1] += c;
    c = h->v[1] >> 51; h->v[1] &= M51; h->v[2] += c;
    c = h->v[2] >> 51; h->v[2] &= M51; h->v[3] += c;
    c = h->v[3] >> 51; h->v[3] &= M51; h->v[4] += c;
    c = h->v[4] >> 51; h->v[4] &= M51; h->v[0] += c * 19;
}

void fe_tobytes(uint8_t s[32], const fe *f) {
    fe t = *f;
    fe_carry(&t); fe_carry(&t);
    /* now 0 <= t < 2^255 (+ tiny), fully carried except possibly v[0] */
    t.v[0] += 19;
    fe_carry(&t);
    /* offset by 19; add 2^255 - 19 so that the final masking subtracts p iff t >= p */
    t.v[0] += (1ULL << 51) - 19;
    t.v[1] += (1ULL << 51) - 1;
    t.v[2] += (1ULL << 51) - 1;
    t.v[3] += (1ULL << 51) - 1;
    t.v[4] += (1ULL << 51) - 1;
    uint64_t c;
    c = t.v[0] >> 51; t.v[0] &= M51; t.v[1] += c;
    c = t.v[1] >> 51; t.v[1] &= M51; t.v[2] += c;
    c = t.v[2] >> 51; t.v[2] &= M51; t.v[3] += c;
    c = t.v[3] >> 51; t.v[3] &= M51; t.v[4] += c;
    t.v[4] &= M51;
    store64(s, t.v[0] | (t.v[1] << 51));
    store64(s + 8, (t.v[1] >> 13) | (t.v[2] << 38));
    store64(s + 16, (t.v[2] >> 26) | (t.v[3] << 25));
    store64(s + 24, (t.v[3] >> 39) | (t.v[4] << 12));
}

void fe_add(fe *h, const fe *f, const fe *g) {
    for (int i = 0; i < 5; i++) h->v[i] = f->v[i] + g->v[i];
    fe_carry(h);
}

void fe_sub(fe *h, const fe *f, const fe *g) {
    /* f + 4p - g ; limbs of inputs are < 2^52 */
    h->v[0] = f->v[0] + 0x1fffffffffffb4ULL - g->v[0];
    h->v[1] = f->v[1] + 0x1ffffffffffffcULL - g->v[1];
    h->v[2] = f->v[2] + 0x1ffffffffffffcULL - g->v[2];
    h->v[3] = f->v[3] + 0x1ffffffffffffcULL - g->v[3];
    h->v[4] = f->v[4] + 0x1ffffffffffffcULL - g->v[4];
    fe_carry(h);
}

void fe_neg(fe *h, const fe *f) { fe z; fe_0(&z); fe_sub(h, &z, f); }

void fe_mul(fe *h, const fe *f, const fe *g) {
    u128 f0 = f->v[0], f1 = f->v[1], f2 = f->v[2], f3 = f->v[3], f4 = f->v[4];
    uint64_t g0 = g->v[0], g1 = g->v[1], g2 = g->v[2], g3 = g->v[3], g4 = g->v[4];
    uint64_t g1_19 = 19 * g1, g2_19 = 19 * g2, g3_19 = 19 * g3, g4_19 = 19 * g4;
    u128 r0 = f0 * g0 + f1 * g4_19 + f2 * g3_19 + f3 * g2_19 + f4 * g1_19;
    u128 r1 = f0 * g1 + f1 * g0 + f2 * g4_19 + f3 * g3_19 + f4 * g2_19;
    u128 r2 = f0 * g2 + f1 * g1 + f2 * g0 + f3 * g4_19 + f4 * g3_19;
    u128 r3 = f0 * g3 + f1 * g2 + f2 * g1 + f3 * g0 + f4 * g4_19;
    u128 r4 = f0 * g4 + f1 * g3 + f2 * g2 + f3 * g1 + f4 * g0;
    uint64_t c;
    r1 += (uint64_t)(r0 >> 51); uint64_t h0 = (uint64_t)r0 & M51;
    r2 += (uint64_t)(r1 >> 51); uint64_t h1 = (uint64_t)r1 & M51;
    r3 += (uint64_t)(r2 >> 51); uint64_t h2 = (uint64_t)r2 & M51;
    r4 += (uint64_t)(r3 >> 51); uint64_t h3 = (uint64_t)r3 & M51;
    c = (uint64_t)(r4 >> 51); uint64_t h4 = (uint64_t)r4 & M51;
    h0 += c * 19;
    c = h0 >> 51; h0 &= M51; h1 += c;
    h->v[0] = h0; h->v[1] = h1; h->v[2] = h2; h->v[3] = h3; h->v[4] = h4;
}

void fe_sq(fe *h, const fe *f) {
    u128 f0 = f->v[0], f1 = f->v[1], f2 = f->v[2], f3 = f->v[3], f4 = f->v[4];
    uint64_t f0_2 = 2 * f->v[0], f1_2 = 2 * f->v[1], f3_19 = 19 * f->v[3], f4_19 = 19 * f->v[4];
    u128 r0 = f0 * f->v[0] + (u128)f1_2 * f4_19 + (u128)(2 * f->v[2]) * f3_19;
    u128 r1 = (u128)f0_2 * f->v[1] + (u128)(2 * f->v[2]) * f4_19 + f3 * f3_19;
    u128 r2 = (u128)f0_2 * f->v[2] + f1 * f->v[1] + (u128)(2 * f->v[3]) * f4_19;
    u128 r3 = (u128)f0_2 * f->v[3] + (u128)f1_2 * f->v[2] + f4 * f4_19;
    u128 r4 = (u128)f0_2 * f->v[4] + (u128)f1_2 * f->v[3] + f2 * f->v[2];
    uint64_t c;
    r1 += (uint64_t)(r0 >> 51); uint64_t h0 = (uint64_t)r0 & M51;
    r2 += (uint64_t)(r1 >> 51); uint64_t h1 = (uint64_t)r1 & M51;
    r3 += (uint64_t)(r2 >> 51); uint64_t h2 = (uint64_t)r2 & M51;
    r4 += (uint64_t)(r3 >> 51); uint64_t h3 = (uint64_t)r3 & M51;
    c = (uint64_t)(r4 >> 51); uint64_t h4 = (uint64_t)r4 & M51;
    h0 += c * 19; c = h0 >> 51; h0 &= M51; h1 += c;
    h->v[0] = h0; h->v[1] = h1; h->v[2] = h2; h->v[3] = h3; h->v[4] = h4;
}

static void fe_sqn(fe *h, const fe *f, int n) {
    fe_sq(h, f);
    for (int i = 1; i < n; i++) fe_sq(h, h);
}

/* z^(2^250-1) helper; returns also z^11 */
static void fe_pow_2_250_1(fe *out, fe *z11, const fe *z) {
    fe z2, z9, z_5_0, z_10_0, z_20_0, z_50_0, z_100_0, t;
    fe_sq(&z2, z);
    fe_sqn(&t, &z2, 2);
    fe_mul(&z9, &t, z);
    fe_mul(z11, &z9, &z2);
    fe_sq(&t, z11);
    fe_mul(&z_5_0, &t, &z9);            /* 2^5 - 1 */
    fe_sqn(&t, &z_5_0, 5);
    fe_mul(&z_10_0, &t, &z_5_0);
    fe_sqn(&t, &z_10_0, 10);
    fe_mul(&z_20_0, &t, &z_10_0);
    fe_sqn(&t, &z_20_0, 20);
    fe_mul(&t, &t, &z_20_0);            /* 2^40 - 1 */
    fe_sqn(&t, &t, 10);
    fe_mul(&z_50_0, &t, &z_10_0);
    fe_sqn(&t, &z_50_0, 50);
    fe_mul(&z_100_0, &t, &z_50_0);
    fe_sqn(&t, &z_100_0, 100);
    fe_mul(&t, &t, &z_100_0);           /* 2^200 - 1 */
    fe_sqn(&t, &t, 50);
    fe_mul(out, &t, &z_50_0);           /* 2^250 - 1 */
}

void fe_invert(fe *out, const fe *z) {
    fe t, z11;
    fe_pow_2_250_1(&t, &z11, z);
    fe_sqn(&t, &t, 5);
    fe_mul(out, &t, &z11);              /* 2^255 - 21 */
}

static void fe_pow22523(fe *out, const fe *z) {
    fe t, z11;
    fe_pow_2_250_1(&t, &z11, z);
    fe_sqn(&t, &t, 2);
    fe_mul(out, &t, z);                 /* 2^252 - 3 */
}

int fe_isneg(const fe *f) { uint8_t s[32]; fe_tobytes(s, f); return s[0] & 1; }
int fe_iszero(const fe *f) {
    uint8_t s[32]; fe_tobytes(s, f);
    uint8_t r = 0; for (int i = 0; i < 32; i++) r |= s[i];
    return r == 0;
}
int fe_eq(const fe *f, const fe *g) {
    uint8_t a[32], b[32]; fe_tobytes(a, f); fe_tobytes(b, g);
    return memcmp(a, b, 32) == 0;
}
void fe_cmov(fe *f, const fe *g, int b) { if (b) *f = *g; }
void fe_abs(fe *h, const fe *f) { if (fe_isneg(f)) fe_neg(h, f); else *h = *f; }

static fe FE_D, FE_D2, FE_SQRTM1, FE_INVSQRT_A_MINUS_D, FE_SQRT_AD_MINUS_ONE, FE_ONE_MINUS_D_SQ,
    FE_D_MINUS_ONE_SQ;

static void hex2bytes(uint8_t *out, const char *hex, size_t n) {
    for (size_t i = 0; i < n; i++) {
        unsigned v; char b[3] = {hex[2 * i], hex[2 * i + 1], 0};
        v = (unsigned)strtoul(b, NULL, 16); out[i] = (uint8_t)v;
    }
}
static void fe_fromhex(fe *h, const char *hex) { uint8_t b[32]; hex2bytes(b, hex, 32); fe_frombytes(h, b); }

/* RFC 9496 4.2 SQRT_RATIO_M1 / dalek FieldElement::sqrt_ratio_i */
int fe_sqrt_ratio_i(fe *r_out, const fe *u, const fe *v) {
    fe v3, v7, r, check, t, neg_u, neg_u_i, r_prime;
    fe_sq(&v3, v); fe_mul(&v3, &v3, v);
    fe_sq(&v7, &v3); fe_mul(&v7, &v7, v);
    fe_mul(&t, u, &v7); fe_pow22523(&t, &t);
    fe_mul(&r, u, &v3); fe_mul(&r, &r, &t);
    fe_sq(&check, &r); fe_mul(&check, &check, v);
    fe_neg(&neg_u, u);
    fe_mul(&neg_u_i, &neg_u, &FE_SQRTM1);
    int correct = fe_eq(&check, u);
    int flipped = fe_eq(&check, &neg_u);
    int flipped_i = fe_eq(&check, &neg_u_i);
    fe_mul(&r_prime, &r, &FE_SQRTM1);
    fe_cmov(&r, &r_prime, flipped | flipped_i);
    fe_abs(r_out, &r);
    return correct | flipped;
}

/* =====================================================================
 * Scalars mod l = 2^252 + 27742317777372353535851937790883648493
 * 4 x 64-bit limbs, Montgomery multiplication with R = 2^256
 * ===================================================================== */
static const uint64_t SC_L[4] = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0, 0x1000000000000000ULL};
static const uint64_t SC_LINV = 0xd2b51da312547e1bULL; /* -l^-1 mod 2^64 */
static const sc SC_R2 = {{0xa40611e3449c0f01ULL, 0xd00e1ba768859347ULL, 0xceec73d217f5be65ULL,
                          0x0399411b7c309a3dULL}};
const sc SC_ZERO = {{0, 0, 0, 0}};
const sc SC_ONE = {{1, 0, 0, 0}};

static int sc_geq_l(const uint64_t a[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > SC_L[i]) return 1;
        if (a[i] < SC_L[i]) return 0;
    }
    return 1;
}
static void sc_sub_l(uint64_t a[4]) {
    u128 b = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)a[i] - SC_L[i] - (uint64_t)b;
        a[i] = (uint64_t)t;
        b = (t >> 64) & 1;
    }
}
/* r = a*b*R^-1 mod l;  requires a*b < l*R (a < 2^256, b < l is enough) */
static void sc_montmul(sc *r, const sc *a, const sc *b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            u128 x = (u128)a->v[j] * b->v[i] + t[j] + (uint64_t)c;
            t[j] = (uint64_t)x; c = x >> 64;
        }
        u128 x = (u128)t[4] + (uint64_t)c; t[4] = (uint64_t)x; t[5] = (uint64_t)(x >> 64);
        uint64_t m = t[0] * SC_LINV;
        x = (u128)m * SC_L[0] + t[0]; c = x >> 64;
        for (int j = 1; j < 4; j++) {
            x = (u128)m * SC_L[j] + t[j] + (uint64_t)c;
            t[j - 1] = (uint64_t)x; c = x >> 64;
        }
        x = (u128)t[4] + (uint64_t)c; t[3] = (uint64_t)x;
        t[4] = t[5] + (uint64_t)(x >> 64);
        t[5] = 0;
    }
    if (t[4] || sc_geq_l(t)) sc_sub_l(t);
    memcpy(r->v, t, 32);
}

void sc_mul(sc *r, const sc *a, const sc *b) {
    sc t; sc_montmul(&t, a, b); sc_montmul(r, &t, &SC_R2);
}
void sc_frombytes_modorder(sc *r, const uint8_t s[32]) {
    sc x, t; memcpy(x.v, s, 32);
    sc_montmul(&t, &x, &SC_R2);      /* x*R mod l */
    sc_montmul(r, &t, &SC_ONE);      /* x mod l */
}
void sc_frombytes_wide(sc *r, const uint8_t s[64]) {
    sc lo, hi, t;
    sc_frombytes_modorder(&lo, s);
    memcpy(hi.v, s + 32, 32);
    sc_montmul(&t, &hi, &SC_R2);     /* hi * 2^256 mod l */
    sc_add(r, &lo, &t);
}
int sc_frombytes_canonical(sc *r, const uint8_t s[32]) {
    memcpy(r->v, s, 32);
    return !sc_geq_l(r->v);
}
void sc_tobytes(uint8_t s[32], const sc *a) { memcpy(s, a->v, 32); }
void sc_from_u64(sc *r, uint64_t x) { r->v[0] = x; r->v[1] = r->v[2] = r->v[3] = 0; }
void sc_add(sc *r, const sc *a, const sc *b) {
    uint64_t t[4]; u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a->v[i] + b->v[i]; t[i] = (uint64_t)c; c >>= 64; }
    if (sc_geq_l(t)) sc_sub_l(t);   /* a,b < l < 2^253 so no carry out */
    memcpy(r->v, t, 32);
}
void sc_neg(sc *r, const sc *a) {
    if (sc_iszero(a)) { *r = SC_ZERO; return; }
    uint64_t t[4]; u128 b = 0;
    for (int i = 0; i < 4; i++) {
        u128 x = (u128)SC_L[i] - a->v[i] - (uint64_t)b;
        t[i] = (uint64_t)x; b = (x >> 64) & 1;
    }
    memcpy(r->v, t, 32);
}
void sc_sub(sc *r, const sc *a, const sc *b) { sc nb; sc_neg(&nb, b); sc_add(r, a, &nb); }
int sc_iszero(const sc *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
int sc_eq(const sc *a, const sc *b) { return memcmp(a->v, b->v, 32) == 0; }
void sc_invert(sc *r, const sc *a) {
    /* a^(l-2) */
    uint64_t e[4]; memcpy(e, SC_L, 32); e[0] -= 2;
    sc acc = SC_ONE;
    for (int i = 252; i >= 0; i--) {
        sc_mul(&acc, &acc, &acc);
        if ((e[i >> 6] >> (i & 63)) & 1) sc_mul(&acc, &acc, a);
    }
    *r = acc;
}

/* =====================================================================
 * Edwards / Ristretto255
 * ===================================================================== */
ge GE_BASE, GE_BBLIND;

void ge_identity(ge *p) { fe_0(&p->X); fe_1(&p->Y); fe_1(&p->Z); fe_0(&p->T); }

void ge_add(ge *r, const ge *p, const ge *q) {
    fe A, B, C, Dd, E, F, G, H, t;
    fe_sub(&A, &p->Y, &p->X); fe_sub(&t, &q->Y, &q->X); fe_mul(&A, &A, &t);
    fe_add(&B, &p->Y, &p->X); fe_add(&t, &q->Y, &q->X); fe_mul(&B, &B, &t);
    fe_mul(&C, &p->T, &q->T); fe_mul(&C, &C, &FE_D2);
    fe_mul(&Dd, &p->Z, &q->Z); fe_add(&Dd, &Dd, &Dd);
    fe_sub(&E, &B, &A); fe_sub(&F, &Dd, &C); fe_add(&G, &Dd, &C); fe_add(&H, &B, &A);
    fe_mul(&r->X, &E, &F); fe_mul(&r->Y, &G, &H); fe_mul(&r->T, &E, &H); fe_mul(&r->Z, &F, &G);
}
void ge_neg(ge *r, const ge *p) { fe_neg(&r->X, &p->X); r->Y = p->Y; r->Z = p->Z; fe_neg(&r->T, &p->T); }
void ge_sub(ge *r, const ge *p, const ge *q) { ge n; ge_neg(&n, q); ge_add(r, p, &n); }
void ge_double(ge *r, const ge *p) {
    fe A, B, C, Dd, E, F, G, H;
    fe_sq(&A, &p->X); fe_sq(&B, &p->Y); fe_sq(&C, &p->Z); fe_add(&C, &C, &C);
    fe_neg(&Dd, &A);
    fe_add(&E, &p->X, &p->Y); fe_sq(&E, &E); fe_sub(&E, &E, &A); fe_sub(&E, &E, &B);
    fe_add(&G, &Dd, &B); fe_sub(&F, &G, &C); fe_sub(&H, &Dd, &B);
    fe_mul(&r->X, &E, &F); fe_mul(&r->Y, &G, &H); fe_mul(&r->T, &E, &H); fe_mul(&r->Z, &F, &G);
}
int ge_eq_ristretto(const ge *p, const ge *q) {
    fe a, b; fe_mul(&a, &p->X, &q->Y); fe_mul(&b, &p->Y, &q->X);
    if (fe_eq(&a, &b)) return 1;
    fe_mul(&a, &p->Y, &q->Y); fe_mul(&b, &p->X, &q->X);
    return fe_eq(&a, &b);
}
int ge_is_identity_ristretto(const ge *p) { return fe_iszero(&p->X) || fe_iszero(&p->Y); }

void ristretto_encode(uint8_t s[32], const ge *p) {
    fe u1, u2, t, invsqrt, den1, den2, z_inv, ix0, iy0, ench, x, y, den_inv, sres;
    fe_add(&u1, &p->Z, &p->Y); fe_sub(&t, &p->Z, &p->Y); fe_mul(&u1, &u1, &t);
    fe_mul(&u2, &p->X, &p->Y);
    fe_sq(&t, &u2); fe_mul(&t, &t, &u1);
    fe one; fe_1(&one);
    fe_sqrt_ratio_i(&invsqrt, &one, &t);
    fe_mul(&den1, &invsqrt, &u1); fe_mul(&den2, &invsqrt, &u2);
    fe_mul(&z_inv, &den1, &den2); fe_mul(&z_inv, &z_inv, &p->T);
    fe_mul(&ix0, &p->X, &FE_SQRTM1); fe_mul(&iy0, &p->Y, &FE_SQRTM1);
    fe_mul(&ench, &den1, &FE_INVSQRT_A_MINUS_D);
    fe_mul(&t, &p->T, &z_inv);
    int rotate = fe_isneg(&t);
    x = p->X; y = p->Y; den_inv = den2;
    fe_cmov(&x, &iy0, rotate); fe_cmov(&y, &ix0, rotate); fe_cmov(&den_inv, &ench, rotate);
    fe_mul(&t, &x, &z_inv);
    if (fe_isneg(&t)) fe_neg(&y, &y);
    fe_sub(&t, &p->Z, &y); fe_mul(&sres, &den_inv, &t);
    fe_abs(&sres, &sres);
    fe_tobytes(s, &sres);
}

int ristretto_decode(ge *p, const uint8_t sbytes[32]) {
    fe s, ss, u1, u2, u2s, v, t, invsqrt, dx, dy, x, y, one;
    uint8_t chk[32];
    fe_frombytes(&s, sbytes);
    fe_tobytes(chk, &s);
    if (memcmp(chk, sbytes, 32) != 0) return 0; /* non-canonical (incl. bit 255 set) */
    if (sbytes[0] & 1) return 0;                /* negative */
    fe_1(&one);
    fe_sq(&ss, &s);
    fe_sub(&u1, &one, &ss); fe_add(&u2, &one, &ss);
    fe_sq(&u2s, &u2);
    fe_sq(&t, &u1); fe_mul(&t, &t, &FE_D); fe_neg(&t, &t); fe_sub(&v, &t, &u2s);
    fe_mul(&t, &v, &u2s);
    int ok = fe_sqrt_ratio_i(&invsqrt, &one, &t);
    fe_mul(&dx, &invsqrt, &u2);
    fe_mul(&dy, &invsqrt, &dx); fe_mul(&dy, &dy, &v);
    fe_add(&t, &s, &s); fe_mul(&x, &t, &dx); fe_abs(&x, &x);
    fe_mul(&y, &u1, &dy);
    fe_mul(&t, &x, &y);
    if (!ok || fe_isneg(&t) || fe_iszero(&y)) return 0;
    p->X = x; p->Y = y; fe_1(&p->Z); p->T = t;
    return 1;
}

static void elligator(ge *out, const fe *r0) {
    fe r, Ns, c, Dn, s, s_prime, Nt, ss, t, one, W0, W1, W2, W3;
    fe_1(&one);
    fe_sq(&r, r0); fe_mul(&r, &r, &FE_SQRTM1);
    fe_add(&Ns, &r, &one); fe_mul(&Ns, &Ns, &FE_ONE_MINUS_D_SQ);
    fe_neg(&c, &one);
    fe_mul(&t, &FE_D, &r); fe_sub(&Dn, &c, &t);
    fe_add(&t, &r, &FE_D); fe_mul(&Dn, &Dn, &t);
    int ok = fe_sqrt_ratio_i(&s, &Ns, &Dn);
    fe_mul(&s_prime, &s, r0);
    if (!fe_isneg(&s_prime)) fe_neg(&s_prime, &s_prime);
    if (!ok) { s = s_prime; c = r; }
    fe_sub(&t, &r, &one); fe_mul(&Nt, &c, &t); fe_mul(&Nt, &Nt, &FE_D_MINUS_ONE_SQ);
    fe_sub(&Nt, &Nt, &Dn);
    fe_sq(&ss, &s);
    fe_add(&W0, &s, &s); fe_mul(&W0, &W0, &Dn);
    fe_mul(&W1, &Nt, &FE_SQRT_AD_MINUS_ONE);
    fe_sub(&W2, &one, &ss);
    fe_add(&W3, &one, &ss);
    fe_mul(&out->X, &W0, &W3); fe_mul(&out->Y, &W2, &W1);
    fe_mul(&out->Z, &W1, &W3); fe_mul(&out->T, &W0, &W2);
}

void ristretto_from_uniform(ge *p, const uint8_t b[64]) {
    fe r1, r2; ge p1, p2;
    fe_frombytes(&r1, b); fe_frombytes(&r2, b + 32);
    elligator(&p1, &r1); elligator(&p2, &r2);
    ge_add(p, &p1, &p2);
}

/* width-5 NAF of a canonical scalar; returns number of digits used */
static void sc_wnaf5(int8_t naf[257], const sc *k) {
    uint64_t x[5] = {k->v[0], k->v[1], k->v[2], k->v[3], 0};
    memset(naf, 0, 257);
    int pos = 0;
    while (pos < 257) {
        if (!(x[0] | x[1] | x[2] | x[3] | x[4])) break;
        if (x[0] & 1) {
            int d = (int)(x[0] & 31);
            if (d >= 16) d -= 32;
            naf[pos] = (int8_t)d;
            /* x -= d */
            if (d >= 0) {
                u128 b = (u128)x[0] - (uint64_t)d; x[0] = (uint64_t)b;
                int bo = (int)((b >> 64) & 1);
                for (int i = 1; i < 5 && bo; i++) { bo = (x[i] == 0); x[i] -= 1; }
            } else {
                u128 c = (u128)x[0] + (uint64_t)(-d); x[0] = (uint64_t)c;
                int co = (int)(c >> 64);
                for (int i = 1; i < 5 && co; i++) { x[i] += 1; co = (x[i] == 0); }
            }
        }
        /* x >>= 1 */
        for (int i = 0; i < 4; i++) x[i] = (x[i] >> 1) | (x[i + 1] << 63);
        x[4] >>= 1;
        pos++;
    }
}

static void ge_odd_table(ge tbl[8], const ge *p) {
    ge p2; ge_double(&p2, p);
    tbl[0] = *p;
    for (int i = 1; i < 8; i++) ge_add(&tbl[i], &tbl[i - 1], &p2);
}

void ge_double_scalarmult(ge *r, const sc *a, const ge *A, const sc *b, const ge *B) {
    int8_t na[257], nb[257];
    ge ta[8], tb[8], acc, t;
    sc_wnaf5(na, a); sc_wnaf5(nb, b);
    ge_odd_table(ta, A); ge_odd_table(tb, B);
    int i = 256;
    while (i >= 0 && !na[i] && !nb[i]) i--;
    ge_identity(&acc);
    for (; i >= 0; i--) {
        ge_double(&acc, &acc);
        if (na[i] > 0) ge_add(&acc, &acc, &ta[na[i] >> 1]);
        else if (na[i] < 0) { ge_neg(&t, &ta[(-na[i]) >> 1]); ge_add(&acc, &acc, &t); }
        if (nb[i] > 0) ge_add(&acc, &acc, &tb[nb[i] >> 1]);
        else if (nb[i] < 0) { ge_neg(&t, &tb[(-nb[i]) >> 1]); ge_add(&acc, &acc, &t); }
    }
    *r = acc;
}

void ge_scalarmult(ge *r, const sc *k, const ge *p) {
    int8_t na[257]; ge ta[8], acc, t;
    sc_wnaf5(na, k); ge_odd_table(ta, p);
    int i = 256;
    while (i >= 0 && !na[i]) i--;
    ge_identity(&acc);
    for (; i >= 0; i--) {
        ge_double(&acc, &acc);
        if (na[i] > 0) ge_add(&acc, &acc, &ta[na[i] >> 1]);
        else if (na[i] < 0) { ge_neg(&t, &ta[(-na[i]) >> 1]); ge_add(&acc, &acc, &t); }
    }
    *r = acc;
}

static unsigned sc_window(const sc *k, unsigned bit, unsigned c) {
    if (bit >= 256) return 0;
    unsigned limb = bit >> 6, off = bit & 63;
    uint64_t v = k->v[limb] >> off;
    if (off + c > 64 && limb < 3) v |= k->v[limb + 1] << (64 - off);
    return (unsigned)(v & ((1u << c) - 1));
}

/* variable-time Pippenger (unsigned c-bit windows) */
void ge_msm(ge *r, const sc *k, const ge *p, size_t n) {
    ge acc, t;
    if (n < 16) {
        ge_identity(&acc);
        for (size_t i = 0; i < n; i++) { ge_scalarmult(&t, &k[i], &p[i]); ge_add(&acc, &acc, &t); }
        *r = acc; return;
    }
    unsigned c = 3;
    while (c < 14 && ((size_t)1 << (c + 3)) <= n) c++;
    size_t nb = ((size_t)1 << c) - 1;
    ge *bk = (ge *)malloc(sizeof(ge) * nb);
    uint8_t *used = (uint8_t *)malloc(nb);
    unsigned nw = (253 + c - 1) / c;
    ge_identity(&acc);
    for (int w = (int)nw - 1; w >= 0; w--) {
        for (unsigned i = 0; i < c; i++) ge_double(&acc, &acc);
        memset(used, 0, nb);
        for (size_t i = 0; i < n; i++) {
            unsigned d = sc_window(&k[i], (unsigned)w * c, c);
            if (!d) continue;
            if (used[d - 1]) ge_add(&bk[d - 1], &bk[d - 1], &p[i]);
            else { bk[d - 1] = p[i]; used[d - 1] = 1; }
        }
        ge run, sum; int have_run = 0, have_sum = 0;
        for (size_t b = nb; b >= 1; b--) {
            if (used[b - 1]) {
                if (have_run) ge_add(&run, &run, &bk[b - 1]); else { run = bk[b - 1]; have_run = 1; }
            }
            if (have_run) {
                if (have_sum) ge_add(&sum, &sum, &run); else { sum = run; have_sum = 1; }
            }
        }
        if (have_sum) ge_add(&acc, &acc, &sum);
    }
    free(bk); free(used);
    *r = acc;
}

/* =====================================================================
 * Keccak-f[1600], SHA3-512, SHAKE256
 * ===================================================================== */
static const uint64_t KRC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
    0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
    0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
static const int KPIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
#define ROL64(x, n) (((x) << (n)) | ((x) >> (64 - (n))))

void keccak_f1600(uint64_t st[25]) {
    uint64_t bc[5], t;
    for (int rnd = 0; rnd < 24; rnd++) {
        for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; i++) {
            t = bc[(i + 4) % 5] ^ ROL64(bc[(i + 1) % 5], 1);
            for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
        }
        t = st[1];
        for (int i = 0; i < 24; i++) {
            int j = KPIL[i]; uint64_t b0 = st[j];
            st[j] = ROL64(t, KROT[i]); t = b0;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = st[j + i];
            for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= KRC[rnd];
    }
}

static void sponge_xor(uint64_t st[25], size_t pos, uint8_t b) { st[pos >> 3] ^= (uint64_t)b << (8 * (pos & 7)); }
static uint8_t sponge_get(const uint64_t st[25], size_t pos) { return (uint8_t)(st[pos >> 3] >> (8 * (pos & 7))); }

void sha3_512(uint8_t out[64], const uint8_t *in, size_t len) {
    uint64_t st[25]; memset(st, 0, sizeof st);
    const size_t rate = 72; size_t pos = 0;
    for (size_t i = 0; i < len; i++) {
        sponge_xor(st, pos++, in[i]);
        if (pos == rate) { keccak_f1600(st); pos = 0; }
    }
    sponge_xor(st, pos, 0x06); sponge_xor(st, rate - 1, 0x80);
    keccak_f1600(st);
    for (size_t i = 0; i < 64; i++) out[i] = sponge_get(st, i);
}

void shake256_init(shake256_ctx *c) { memset(c, 0, sizeof *c); }
void shake256_absorb(shake256_ctx *c, const uint8_t *in, size_t len) {
    for (size_t i = 0; i < len; i++) {
        sponge_xor(c->st, c->pos++, in[i]);
        if (c->pos == 136) { keccak_f1600(c->st); c->pos = 0; }
    }
}
void shake256_squeeze(shake256_ctx *c, uint8_t *out, size_t len) {
    if (!c->squeezing) {
        sponge_xor(c->st, c->pos, 0x1F); sponge_xor(c->st, 135, 0x80);
        keccak_f1600(c->st); c->pos = 0; c->squeezing = 1;
    }
    for (size_t i = 0; i < len; i++) {
        if (c->pos == 136) { keccak_f1600(c->st); c->pos = 0; }
        out[i] = sponge_get(c->st, c->pos++);
    }
}

/* =====================================================================
 * STROBE-128 / Merlin v1.0 (merlin 3.0.0 strobe.rs / transcript.rs semantics)
 * ===================================================================== */
#define STROBE_R 166
static void strobe_perm(merlin_t *t) {
    uint64_t w[25]; memcpy(w, t->st, 200); keccak_f1600(w); memcpy(t->st, w, 200);
}
static void strobe_run_f(merlin_t *t) {
    t->st[t->pos] ^= t->pos_begin;
    t->st[t->pos + 1] ^= 0x04;
    t->st[STROBE_R + 1] ^= 0x80;
    strobe_perm(t);
    t->pos = 0; t->pos_begin = 0;
}
static void strobe_absorb(merlin_t *t, const uint8_t *d, size_t n) {
    for (size_t i = 0; i < n; i++) {
        t->st[t->pos++] ^= d[i];
        if (t->pos == STROBE_R) strobe_run_f(t);
    }
}
static void strobe_squeeze(merlin_t *t, uint8_t *d, size_t n) {
    for (size_t i = 0; i < n; i++) {
        d[i] = t->st[t->pos]; t->st[t->pos] = 0; t->pos++;
        if (t->pos == STROBE_R) strobe_run_f(t);
    }
}
static void strobe_begin_op(merlin_t *t, uint8_t flags, int more) {
    if (more) return;
    uint8_t hdr[2] = {t->pos_begin, flags};
    t->pos_begin = (uint8_t)(t->pos + 1);
    t->cur_flags = flags;
    strobe_absorb(t, hdr, 2);
    if ((flags & (4 | 32)) && t->pos != 0) strobe_run_f(t);
}
static void strobe_meta_ad(merlin_t *t, const uint8_t *d, size_t n, int more) { strobe_begin_op(t, 16 | 2, more); strobe_absorb(t, d, n); }
static void strobe_ad(merlin_t *t, const uint8_t *d, size_t n, int more) { strobe_begin_op(t, 2, more); strobe_absorb(t, d, n); }
static void strobe_prf(merlin_t *t, uint8_t *d, size_t n) { strobe_begin_op(t, 1 | 2 | 4, 0); strobe_squeeze(t, d, n); }

static void merlin_append_raw(merlin_t *t, const uint8_t *label, size_t ll, const uint8_t *msg, size_t len) {
    uint8_t le[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
    strobe_meta_ad(t, label, ll, 0);
    strobe_meta_ad(t, le, 4, 1);
    strobe_ad(t, msg, len, 0);
}
void merlin_init(merlin_t *t, const uint8_t *label, size_t len) {
    memset(t, 0, sizeof *t);
    static const uint8_t hdr[6] = {1, STROBE_R + 2, 1, 0, 1, 96};
    memcpy(t->st, hdr, 6); memcpy(t->st + 6, "STROBEv1.0.2", 12);
    strobe_perm(t);
    strobe_meta_ad(t, (const uint8_t *)"Merlin v1.0", 11, 0);
    merlin_append_raw(t, (const uint8_t *)"dom-sep", 7, label, len);
}
void merlin_append(merlin_t *t, const char *label, const uint8_t *msg, size_t len) {
    merlin_append_raw(t, (const uint8_t *)label, strlen(label), msg, len);
}
void merlin_append_lbl(merlin_t *t, const uint8_t *label, size_t ll, const uint8_t *msg, size_t len) { merlin_append_raw(t, label, ll, msg, len); }
void merlin_append_u64(merlin_t *t, const char *label, uint64_t x) {
    uint8_t b[8]; store64(b, x); merlin_append(t, label, b, 8);
}
void merlin_challenge_bytes(merlin_t *t, const char *label, uint8_t *out, size_t len) {
    uint8_t le[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
    strobe_meta_ad(t, (const uint8_t *)label, strlen(label), 0);
    strobe_meta_ad(t, le, 4, 1);
    strobe_prf(t, out, len);
}
void merlin_challenge_scalar(merlin_t *t, const char *label, sc *out) {
    uint8_t b[64]; merlin_challenge_bytes(t, label, b, 64); sc_frombytes_wide(out, b);
}

/* ===================================================================== */
static int orc_inited = 0;
void orc_init(void) {
    if (orc_inited) return;   /* call once from the main thread before any parallel region (all entry points do) */
    fe_fromhex(&FE_D, "a3785913ca4deb75abd841414d0a700098e879777940c78c73fe6f2bee6c0352");
    fe_fromhex(&FE_D2, "59f1b226949bd6eb56b183829a14e00030d1f3eef2808e19e7fcdf56dcd90624");
    fe_fromhex(&FE_SQRTM1, "b0a00e4a271beec478e42fad0618432fa7d7fb3d99004d2b0bdfc14f8024832b");
    fe_fromhex(&FE_INVSQRT_A_MINUS_D, "ea405d80aafdc899be72415a17162f9d40d801fe917bc216a2fcafcf05896c78");
    fe_fromhex(&FE_SQRT_AD_MINUS_ONE, "1b2e7b49a0f6977ebd54781b0c8e9daffdd1f531c9fc3c0fac48832bbf316937");
    fe_fromhex(&FE_ONE_MINUS_D_SQ, "76c15f94c1097ce20f355ecd38a1812ce4df70beddab9499d7e0b3b2a8729002");
    fe_fromhex(&FE_D_MINUS_ONE_SQ, "204ded44aa5aad3199191eb02c4a9ed2eb4e9b522fd3dc4c41226cf67ab36859");
    uint8_t b[32], h[64];
    hex2bytes(b, "e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76", 32);
    ristretto_decode(&GE_BASE, b);
    sha3_512(h, b, 32);
    ristretto_from_uniform(&GE_BBLIND, h);
    orc_inited = 1;
}
