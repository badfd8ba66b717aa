/*
 * ORACLE (test infrastructure, NOT product code) -- see orc_rofl.h for scope and parity status.
 */
#include "orc_rofl.h"
#include "orc_curve.h"
#include <math.h>
#include <stdlib.h>

/* ------------------------------------------------------------------ helpers */
size_t orc_next_pow2(size_t val) {
    /* range_proof_vec/mod.rs:225-235 (val == 0 underflows in the reference; callers reject it) */
    if (val == 1) return 1;
    size_t n = val - 1;
    while ((n & (n - 1)) != 0) n &= n - 1;
    return n << 1;
}
static unsigned lg2(size_t x) { unsigned r = 0; while (((size_t)1 << r) < x) r++; return r; }
static int is_pow2(size_t x) { return x && !(x & (x - 1)); }
size_t orc_proof_size(size_t n_bits, size_t m) { return 32 * (9 + 2 * (size_t)lg2(n_bits * m)); }
size_t orc_nonces_per_chunk(size_t n_bits, size_t m) { return m * (2 * n_bits + 4); }

static void nonce_get(const orc_nonce_t *ns, uint64_t idx, sc *out) {
    uint8_t b[64];
    if (ns->mode == 0) {
        if (idx >= ns->stream_scalars) { memset(b, 0, 64); }
        else memcpy(b, ns->stream + 64 * idx, 64);
    } else {
        /* two scalars per SHAKE256 block: scalar idx is bytes 64 (idx & 1) .. + 64 of SHAKE256("rofl-zk/nonce/v2" || seed || u64le(idx >> 1)) */
        uint8_t blk[128];
        const uint64_t bi = idx >> 1;
        shake256_ctx c; shake256_init(&c);
        shake256_absorb(&c, (const uint8_t *)"rofl-zk/nonce/v2", 16);
        shake256_absorb(&c, ns->seed, 32);
        uint8_t le[8]; for (int i = 0; i < 8; i++) le[i] = (uint8_t)(bi >> (8 * i));
        shake256_absorb(&c, le, 8);
        shake256_squeeze(&c, blk, 128);
        memcpy(b, blk + 64 * (idx & 1), 64);
    }
    sc_frombytes_wide(out, b);
}
void orc_nonce_scalar(const orc_nonce_t *ns, uint64_t idx, uint8_t out[32]) {
    sc s; orc_init(); nonce_get(ns, idx, &s); sc_tobytes(out, &s);
}
static void verifier_c(const uint8_t seed[32], uint64_t idx, sc *out) {
    uint8_t b[64], le[8];
    shake256_ctx c; shake256_init(&c);
    shake256_absorb(&c, (const uint8_t *)"rofl-zk/vrfyc/v1", 16);
    shake256_absorb(&c, seed, 32);
    for (int i = 0; i < 8; i++) le[i] = (uint8_t)(idx >> (8 * i));
    shake256_absorb(&c, le, 8);
    shake256_squeeze(&c, b, 64);
    sc_frombytes_wide(out, b);
}

/* ------------------------------------------------------------------ primitive wrappers */
void orc_ristretto_from_uniform(const uint8_t in[64], uint8_t out[32]) {
    ge p; orc_init(); ristretto_from_uniform(&p, in); ristretto_encode(out, &p);
}
int orc_ristretto_scalarmult(const uint8_t k[32], const uint8_t p[32], uint8_t out[32]) {
    ge P, Rr; sc s; orc_init();
    if (!ristretto_decode(&P, p)) return -1;
    sc_frombytes_modorder(&s, k); ge_scalarmult(&Rr, &s, &P); ristretto_encode(out, &Rr); return 0;
}
void orc_ristretto_scalarmult_base(const uint8_t k[32], uint8_t out[32]) {
    ge Rr; sc s; orc_init(); sc_frombytes_modorder(&s, k); ge_scalarmult(&Rr, &s, &GE_BASE); ristretto_encode(out, &Rr);
}
int orc_ristretto_add(const uint8_t p[32], const uint8_t q[32], uint8_t out[32]) {
    ge P, Q, Rr; orc_init();
    if (!ristretto_decode(&P, p) || !ristretto_decode(&Q, q)) return -1;
    ge_add(&Rr, &P, &Q); ristretto_encode(out, &Rr); return 0;
}
int orc_ristretto_is_valid(const uint8_t p[32]) { ge P; orc_init(); return ristretto_decode(&P, p); }
void orc_sc_reduce_wide(const uint8_t in[64], uint8_t out[32]) { sc s; sc_frombytes_wide(&s, in); sc_tobytes(out, &s); }
void orc_sc_mul(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) {
    sc x, y, r; sc_frombytes_modorder(&x, a); sc_frombytes_modorder(&y, b); sc_mul(&r, &x, &y); sc_tobytes(out, &r);
}
void orc_sc_add(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) {
    sc x, y, r; sc_frombytes_modorder(&x, a); sc_frombytes_modorder(&y, b); sc_add(&r, &x, &y); sc_tobytes(out, &r);
}
void orc_sc_neg(const uint8_t a[32], uint8_t out[32]) { sc x, r; sc_frombytes_modorder(&x, a); sc_neg(&r, &x); sc_tobytes(out, &r); }
void orc_sc_invert(const uint8_t a[32], uint8_t out[32]) { sc x, r; sc_frombytes_modorder(&x, a); sc_invert(&r, &x); sc_tobytes(out, &r); }
void orc_sha3_512(const uint8_t *in, size_t len, uint8_t out[64]) { sha3_512(out, in, len); }
void orc_shake256(const uint8_t *in, size_t len, uint8_t *out, size_t outlen) {
    shake256_ctx c; shake256_init(&c); shake256_absorb(&c, in, len); shake256_squeeze(&c, out, outlen);
}
void orc_keccak_f1600(uint64_t st[25]) { keccak_f1600(st); }
void orc_merlin_init(void *t, const uint8_t *label, size_t len) { merlin_init((merlin_t *)t, label, len); }
void orc_merlin_append(void *t, const char *label, const uint8_t *msg, size_t len) { merlin_append((merlin_t *)t, label, msg, len); }
void orc_merlin_challenge(void *t, const char *label, uint8_t *out, size_t len) { merlin_challenge_bytes((merlin_t *)t, label, out, len); }
void orc_pedersen_gens(uint8_t B[32], uint8_t Bb[32]) { orc_init(); ristretto_encode(B, &GE_BASE); ristretto_encode(Bb, &GE_BBLIND); }

/* bulletproofs generators.rs: GeneratorsChain::new(label) = SHAKE256("GeneratorsChain" || label) */
static void gens_party(size_t j, size_t n, ge *G, ge *H) {
    for (int which = 0; which < 2; which++) {
        shake256_ctx c; shake256_init(&c);
        uint8_t label[5] = {(uint8_t)(which ? 'H' : 'G'), (uint8_t)j, (uint8_t)(j >> 8), (uint8_t)(j >> 16), (uint8_t)(j >> 24)};
        shake256_absorb(&c, (const uint8_t *)"GeneratorsChain", 15);
        shake256_absorb(&c, label, 5);
        for (size_t i = 0; i < n; i++) {
            uint8_t u[64]; shake256_squeeze(&c, u, 64);
            ristretto_from_uniform(which ? &H[i] : &G[i], u);
        }
    }
}
void orc_bp_gens(size_t n, size_t m, uint8_t *G_out, uint8_t *H_out) {
    orc_init();
    ge *G = malloc(sizeof(ge) * n), *H = malloc(sizeof(ge) * n);
    for (size_t j = 0; j < m; j++) {
        gens_party(j, n, G, H);
        for (size_t i = 0; i < n; i++) {
            ristretto_encode(G_out + 32 * (j * n + i), &G[i]);
            ristretto_encode(H_out + 32 * (j * n + i), &H[i]);
        }
    }
    free(G); free(H);
}
void orc_msm(const uint8_t *scalars, const uint8_t *points, size_t n, uint8_t out[32]) {
    orc_init();
    sc *k = malloc(sizeof(sc) * (n ? n : 1)); ge *p = malloc(sizeof(ge) * (n ? n : 1)); ge r;
    for (size_t i = 0; i < n; i++) { sc_frombytes_modorder(&k[i], scalars + 32 * i); ristretto_decode(&p[i], points + 32 * i); }
    ge_msm(&r, k, p, n); ristretto_encode(out, &r);
    free(k); free(p);
}

/* ------------------------------------------------------------------ conversion32.rs / fp.rs */
static uint64_t fix_max_bits(unsigned fp_bits) { return fp_bits >= 64 ? ~0ULL : ((1ULL << fp_bits) - 1); }
/* Fix::saturating_from_float(|v|).to_bits() -- fixed 0.3.3: round to nearest (ties to even), saturate */
static int fix_from_abs_f32(float v, unsigned fp_bits, unsigned fp_frac, uint64_t *out) {
    if (isnan(v)) return ORC_NON_FINITE;
    double x = fabs((double)v) * (double)(1ULL << fp_frac);
    double lim = ldexp(1.0, (int)fp_bits);
    if (isinf(x) || x >= lim) { *out = fix_max_bits(fp_bits); return 0; }
    double k = nearbyint(x);
    if (k >= lim) { *out = fix_max_bits(fp_bits); return 0; }
    *out = (uint64_t)k;
    return 0;
}
/* Fix::from_bits(k).to_float::<f32>() */
static float fix_to_f32(uint64_t k, unsigned fp_frac) {
    volatile float f = (float)k;
    return f / (float)(1ULL << fp_frac);
}
static uint64_t read_from_bytes(const uint8_t s[32], unsigned fp_bits) {
    /* fp.rs:41-108: low N_BITS/8 bytes little-endian */
    uint64_t r = 0;
    for (unsigned i = 0; i < fp_bits / 8; i++) r |= (uint64_t)s[i] << (8 * i);
    return r;
}
static int f32_to_sc(float v, unsigned fp_bits, unsigned fp_frac, sc *out) {
    /* conversion32.rs:11-18 */
    uint64_t k; int rc = fix_from_abs_f32(v, fp_bits, fp_frac, &k);
    if (rc) return rc;
    sc s; sc_from_u64(&s, k);
    if (v < 0.0f) sc_neg(out, &s); else *out = s;
    return 0;
}
static float sc_to_f32(const sc *s, unsigned fp_bits, unsigned fp_frac) {
    /* conversion32.rs:24-35: "negative" iff the top byte is non-zero */
    uint8_t b[32]; sc_tobytes(b, s);
    if (b[31] != 0) {
        sc n; sc_neg(&n, s); sc_tobytes(b, &n);
        return -fix_to_f32(read_from_bytes(b, fp_bits), fp_frac);
    }
    return fix_to_f32(read_from_bytes(b, fp_bits), fp_frac);
}
/* conversion32.rs:66-88 square */
int orc_fp_square(const uint8_t s32[32], unsigned fp_bits, unsigned fp_frac, uint8_t out[32]) {
    uint8_t b[32]; memcpy(b, s32, 32);
    if (s32[31] != 0) orc_sc_neg(s32, b);
    uint64_t v = read_from_bytes(b, fp_bits);
    unsigned __int128 prod = ((unsigned __int128)v * v) >> fp_frac;       /* fixed 0.3.3 checked_mul: wide product, arithmetic shift */
    if (prod > (unsigned __int128)fix_max_bits(fp_bits)) return ORC_OVERFLOW;
    sc r; sc_from_u64(&r, (uint64_t)prod); sc_tobytes(out, &r);
    return ORC_OK;
}
/* conversion32.rs:101-111 precompute_exponentiate */
void orc_scalar_powers(const uint8_t v32[32], size_t count, uint8_t *out) {
    sc v, acc = SC_ONE; sc_frombytes_modorder(&v, v32);
    for (size_t i = 0; i < count; i++) { sc_tobytes(out + 32 * i, &acc); sc t; sc_mul(&t, &acc, &v); acc = t; }
}
/* conversion32.rs:49-54 f32_to_fp_vec (Fix is unsigned: negatives saturate to 0) and :41-47 uint_to_f32 */
int orc_f32_to_fp(float v, unsigned fp_bits, unsigned fp_frac, uint64_t *out) {
    if (v != v) return ORC_NON_FINITE;
    if (v < 0.0f) { *out = 0; return ORC_OK; }
    return fix_from_abs_f32(v, fp_bits, fp_frac, out);
}
float orc_uint_to_f32(uint64_t k, unsigned fp_bits, unsigned fp_frac) { return fix_to_f32(k & fix_max_bits(fp_bits), fp_frac); }

int orc_f32_to_scalar(float v, unsigned fp_bits, unsigned fp_frac, uint8_t out[32]) {
    sc s; int rc = f32_to_sc(v, fp_bits, fp_frac, &s); if (rc) return rc; sc_tobytes(out, &s); return 0;
}
float orc_scalar_to_f32(const uint8_t s[32], unsigned fp_bits, unsigned fp_frac) {
    sc x; sc_frombytes_modorder(&x, s); return sc_to_f32(&x, fp_bits, fp_frac);
}
void orc_get_clip_bounds(unsigned range, unsigned fp_bits, unsigned fp_frac, float *mn, float *mx) {
    /* conversion32.rs:56-60: Fix::from_bits(((1u128 << (range-1)) - 1) as URawFix).to_float() */
    unsigned __int128 v = ((unsigned __int128)1 << (range - 1)) - 1;
    uint64_t k = (uint64_t)v & fix_max_bits(fp_bits);
    *mx = fix_to_f32(k, fp_frac); *mn = -*mx;
}
float orc_get_l2_clip_bounds(unsigned range, unsigned fp_bits, unsigned fp_frac) {
    unsigned __int128 v = ((unsigned __int128)1 << range) - 1;
    return fix_to_f32((uint64_t)v & fix_max_bits(fp_bits), fp_frac);
}
void orc_clip_f32(const float *in, size_t d, unsigned range, unsigned fp_bits, unsigned fp_frac, float *out) {
    /* range_proof_vec/mod.rs:104-111: f32::min(max, f32::max(min, x)) (NaN -> min) */
    float mn, mx; orc_get_clip_bounds(range, fp_bits, fp_frac, &mn, &mx);
    for (size_t i = 0; i < d; i++) { float t = fmaxf(mn, in[i]); out[i] = fminf(mx, t); }
}
static int is_out_of_range(const float *v, size_t d, unsigned range, unsigned fp_bits, unsigned fp_frac) {
    float mn, mx; orc_get_clip_bounds(range, fp_bits, fp_frac, &mn, &mx);
    for (size_t i = 0; i < d; i++) if (mn > v[i] || v[i] > mx) return 1;
    return 0;
}

/* ------------------------------------------------------------------ pedersen_ops.rs */
static void pedersen_commit(ge *out, const sc *v, const sc *r) {
    ge_double_scalarmult(out, v, &GE_BASE, r, &GE_BBLIND);
}
void orc_commit_vec(const uint8_t *values32, const uint8_t *blind32, size_t d, uint8_t *out32) {
    orc_init();
    for (size_t i = 0; i < d; i++) {
        sc v, r; ge p;
        sc_frombytes_modorder(&v, values32 + 32 * i);
        if (blind32) sc_frombytes_modorder(&r, blind32 + 32 * i); else r = SC_ZERO;
        pedersen_commit(&p, &v, &r); ristretto_encode(out32 + 32 * i, &p);
    }
}
int orc_add_points_vec(const uint8_t *a32, const uint8_t *b32, size_t d, uint8_t *out32) {
    orc_init();
    for (size_t i = 0; i < d; i++) {
        ge a, b, r;
        if (!ristretto_decode(&a, a32 + 32 * i) || !ristretto_decode(&b, b32 + 32 * i)) return ORC_FORMAT_ERROR;
        ge_add(&r, &a, &b); ristretto_encode(out32 + 32 * i, &r);
    }
    return 0;
}
int orc_shift_points(const uint8_t *a32, size_t d, const uint8_t offset[32], uint8_t *out32) {
    orc_init();
    ge o; if (!ristretto_decode(&o, offset)) return ORC_FORMAT_ERROR;
    for (size_t i = 0; i < d; i++) {
        ge a, r; if (!ristretto_decode(&a, a32 + 32 * i)) return ORC_FORMAT_ERROR;
        ge_add(&r, &a, &o); ristretto_encode(out32 + 32 * i, &r);
    }
    return 0;
}

/* ------------------------------------------------------------------ bulletproofs 4.0.0 (restated) */
static void append_point(merlin_t *t, const char *label, const ge *p, uint8_t *enc_out) {
    uint8_t e[32]; ristretto_encode(e, p); merlin_append(t, label, e, 32);
    if (enc_out) memcpy(enc_out, e, 32);
}
static void append_scalar(merlin_t *t, const char *label, const sc *s) { uint8_t b[32]; sc_tobytes(b, s); merlin_append(t, label, b, 32); }
static void inner_product(sc *out, const sc *a, const sc *b, size_t n) {
    sc acc = SC_ZERO, t;
    for (size_t i = 0; i < n; i++) { sc_mul(&t, &a[i], &b[i]); sc_add(&acc, &acc, &t); }
    *out = acc;
}

int orc_bp_prove(const uint8_t *label, size_t label_len, size_t gens_capacity, const uint64_t *values,
                 const uint8_t *blind32, size_t m, size_t n, const orc_nonce_t *ns, uint64_t nbase,
                 uint8_t *proof_out, uint8_t *V_out) {
    orc_init();
    if (!(n == 8 || n == 16 || n == 32 || n == 64)) return ORC_INVALID_BITSIZE;
    if (!is_pow2(m)) return ORC_INVALID_AGGREGATION;
    if (gens_capacity < n) return ORC_INVALID_GENS_LENGTH;
    size_t N = n * m; unsigned lgN = lg2(N);
    ge *G = malloc(sizeof(ge) * N), *H = malloc(sizeof(ge) * N);
    sc *sL = malloc(sizeof(sc) * N), *sR = malloc(sizeof(sc) * N), *l = malloc(sizeof(sc) * N), *r = malloc(sizeof(sc) * N);
    sc *vbl = malloc(sizeof(sc) * m), *hfac = malloc(sizeof(sc) * N), *tmp = malloc(sizeof(sc) * N);
    merlin_t tr; merlin_init(&tr, label, label_len);
    /* Dealer::new -> rangeproof_domain_sep */
    merlin_append(&tr, "dom-sep", (const uint8_t *)"rangeproof v1", 13);
    merlin_append_u64(&tr, "n", n); merlin_append_u64(&tr, "m", m);

    sc a_bl = SC_ZERO, s_bl = SC_ZERO, t;
    ge A, S, pt;
    /* Party::new + assign_position: nonce draw order a_blinding, s_blinding, s_L[0..n), s_R[0..n) per party */
    for (size_t j = 0; j < m; j++) {
        gens_party(j, n, G + j * n, H + j * n);
        uint64_t b = nbase + j * (2 * n + 2);
        nonce_get(ns, b, &t); sc_add(&a_bl, &a_bl, &t);
        nonce_get(ns, b + 1, &t); sc_add(&s_bl, &s_bl, &t);
        for (size_t i = 0; i < n; i++) { nonce_get(ns, b + 2 + i, &sL[j * n + i]); nonce_get(ns, b + 2 + n + i, &sR[j * n + i]); }
        sc_frombytes_modorder(&vbl[j], blind32 + 32 * j);
        sc v; sc_from_u64(&v, values[j]);
        pedersen_commit(&pt, &v, &vbl[j]);
        uint8_t e[32]; ristretto_encode(e, &pt); memcpy(V_out + 32 * j, e, 32);
        merlin_append(&tr, "V", e, 32);
    }
    ge_scalarmult(&A, &a_bl, &GE_BBLIND);
    for (size_t j = 0; j < m; j++)
        for (size_t i = 0; i < n; i++) {
            if ((values[j] >> i) & 1) ge_add(&A, &A, &G[j * n + i]);
            else ge_sub(&A, &A, &H[j * n + i]);
        }
    ge_scalarmult(&S, &s_bl, &GE_BBLIND);
    ge_msm(&pt, sL, G, N); ge_add(&S, &S, &pt);
    ge_msm(&pt, sR, H, N); ge_add(&S, &S, &pt);
    uint8_t *o = proof_out;
    append_point(&tr, "A", &A, o); o += 32;
    append_point(&tr, "S", &S, o); o += 32;
    sc y, z, zz, x, w;
    merlin_challenge_scalar(&tr, "y", &y); merlin_challenge_scalar(&tr, "z", &z);
    sc_mul(&zz, &z, &z);
    /* apply_challenge (bit challenge): l0,l1,r0,r1 and t-poly */
    sc t0 = SC_ZERO, t1 = SC_ZERO, t2 = SC_ZERO, exp_y = SC_ONE, offset_zz = zz, one = SC_ONE, two;
    sc_from_u64(&two, 2);
    sc *l0 = l, *r0 = r;                  /* reuse: l/r hold l0/r0 until x is known */
    sc *r1 = tmp;
    for (size_t j = 0; j < m; j++) {
        sc exp_2 = SC_ONE;
        for (size_t i = 0; i < n; i++) {
            size_t k = j * n + i;
            sc aL, aR, u;
            sc_from_u64(&aL, (values[j] >> i) & 1);
            sc_sub(&aR, &aL, &one);
            sc_sub(&l0[k], &aL, &z);
            sc_add(&u, &aR, &z); sc_mul(&u, &u, &exp_y);
            sc_mul(&t, &offset_zz, &exp_2); sc_add(&r0[k], &u, &t);
            sc_mul(&r1[k], &exp_y, &sR[k]);
            sc_mul(&exp_y, &exp_y, &y);
            sc_add(&exp_2, &exp_2, &exp_2);
        }
        sc_mul(&offset_zz, &offset_zz, &z);
    }
    inner_product(&t0, l0, r0, N);
    inner_product(&t2, sL, r1, N);
    inner_product(&t1, l0, r1, N); inner_product(&t, sL, r0, N); sc_add(&t1, &t1, &t);
    sc t1_bl = SC_ZERO, t2_bl = SC_ZERO;
    for (size_t j = 0; j < m; j++) {
        uint64_t b = nbase + m * (2 * n + 2) + 2 * j;
        nonce_get(ns, b, &t); sc_add(&t1_bl, &t1_bl, &t);
        nonce_get(ns, b + 1, &t); sc_add(&t2_bl, &t2_bl, &t);
    }
    ge T1, T2;
    pedersen_commit(&T1, &t1, &t1_bl); pedersen_commit(&T2, &t2, &t2_bl);
    append_point(&tr, "T_1", &T1, o); o += 32;
    append_point(&tr, "T_2", &T2, o); o += 32;
    merlin_challenge_scalar(&tr, "x", &x);
    sc xx; sc_mul(&xx, &x, &x);
    sc t_x, t_x_bl, e_bl;
    sc_mul(&t, &t1, &x); sc_add(&t_x, &t0, &t); sc_mul(&t, &t2, &xx); sc_add(&t_x, &t_x, &t);
    t_x_bl = SC_ZERO; offset_zz = zz;
    for (size_t j = 0; j < m; j++) { sc_mul(&t, &offset_zz, &vbl[j]); sc_add(&t_x_bl, &t_x_bl, &t); sc_mul(&offset_zz, &offset_zz, &z); }
    sc_mul(&t, &t1_bl, &x); sc_add(&t_x_bl, &t_x_bl, &t);
    sc_mul(&t, &t2_bl, &xx); sc_add(&t_x_bl, &t_x_bl, &t);
    sc_mul(&t, &s_bl, &x); sc_add(&e_bl, &a_bl, &t);
    for (size_t k = 0; k < N; k++) {
        sc_mul(&t, &sL[k], &x); sc_add(&l[k], &l0[k], &t);
        sc_mul(&t, &r1[k], &x); sc_add(&r[k], &r0[k], &t);
    }
    append_scalar(&tr, "t_x", &t_x); append_scalar(&tr, "t_x_blinding", &t_x_bl); append_scalar(&tr, "e_blinding", &e_bl);
    sc_tobytes(o, &t_x); o += 32; sc_tobytes(o, &t_x_bl); o += 32; sc_tobytes(o, &e_bl); o += 32;
    merlin_challenge_scalar(&tr, "w", &w);
    ge Q; ge_scalarmult(&Q, &w, &GE_BASE);
    /* H_factors = y^-i */
    sc yinv; sc_invert(&yinv, &y);
    hfac[0] = SC_ONE; for (size_t k = 1; k < N; k++) sc_mul(&hfac[k], &hfac[k - 1], &yinv);

    /* InnerProductProof::create */
    merlin_append(&tr, "dom-sep", (const uint8_t *)"ipp v1", 6);
    merlin_append_u64(&tr, "n", N);
    sc *a = l, *b = r;
    size_t nn = N; int first = 1;
    sc *ks = malloc(sizeof(sc) * (N + 1)); ge *ps = malloc(sizeof(ge) * (N + 1));
    (void)lgN;
    while (nn != 1) {
        nn /= 2;
        sc cL, cR, u, uinv;
        inner_product(&cL, a, b + nn, nn); inner_product(&cR, a + nn, b, nn);
        ge Lp, Rp;
        /* L = <a_L*g_R, G_R> + <b_R*h_L, H_L> + c_L Q */
        for (size_t i = 0; i < nn; i++) {
            ks[i] = a[i]; ps[i] = G[nn + i];
            if (first) sc_mul(&ks[nn + i], &b[nn + i], &hfac[i]); else ks[nn + i] = b[nn + i];
            ps[nn + i] = H[i];
        }
        ks[2 * nn] = cL; ps[2 * nn] = Q;
        ge_msm(&Lp, ks, ps, 2 * nn + 1);
        for (size_t i = 0; i < nn; i++) {
            ks[i] = a[nn + i]; ps[i] = G[i];
            if (first) sc_mul(&ks[nn + i], &b[i], &hfac[nn + i]); else ks[nn + i] = b[i];
            ps[nn + i] = H[nn + i];
        }
        ks[2 * nn] = cR; ps[2 * nn] = Q;
        ge_msm(&Rp, ks, ps, 2 * nn + 1);
        append_point(&tr, "L", &Lp, o); o += 32;
        append_point(&tr, "R", &Rp, o); o += 32;
        merlin_challenge_scalar(&tr, "u", &u); sc_invert(&uinv, &u);
        for (size_t i = 0; i < nn; i++) {
            sc p, q;
            sc_mul(&p, &a[i], &u); sc_mul(&q, &uinv, &a[nn + i]); sc_add(&a[i], &p, &q);
            sc_mul(&p, &b[i], &uinv); sc_mul(&q, &u, &b[nn + i]); sc_add(&b[i], &p, &q);
            sc gl = uinv, gr = u, hl = u, hr = uinv;
            if (first) { sc_mul(&hl, &u, &hfac[i]); sc_mul(&hr, &uinv, &hfac[nn + i]); }
            ge_double_scalarmult(&G[i], &gl, &G[i], &gr, &G[nn + i]);
            ge_double_scalarmult(&H[i], &hl, &H[i], &hr, &H[nn + i]);
        }
        first = 0;
    }
    sc_tobytes(o, &a[0]); o += 32; sc_tobytes(o, &b[0]); o += 32;
    free(G); free(H); free(sL); free(sR); free(l); free(r); free(vbl); free(hfac); free(tmp); free(ks); free(ps);
    return ORC_OK;
}

int orc_bp_verify(const uint8_t *label, size_t label_len, size_t gens_capacity, const uint8_t *proof,
                  size_t proof_len, const uint8_t *V32, size_t m, size_t n, const uint8_t c_seed[32],
                  uint64_t c_index, int *ok) {
    orc_init();
    *ok = 0;
    /* RangeProof::from_bytes + InnerProductProof::from_bytes */
    if (proof_len % 32 != 0 || proof_len < 7 * 32) return ORC_FORMAT_ERROR;
    size_t ne = (proof_len - 7 * 32) / 32;
    if (ne < 2 || (ne - 2) % 2 != 0) return ORC_FORMAT_ERROR;
    size_t lg_n = (ne - 2) / 2;
    if (lg_n >= 32) return ORC_FORMAT_ERROR;
    sc t_x, t_x_bl, e_bl, a, b;
    if (!sc_frombytes_canonical(&t_x, proof + 4 * 32) || !sc_frombytes_canonical(&t_x_bl, proof + 5 * 32) ||
        !sc_frombytes_canonical(&e_bl, proof + 6 * 32))
        return ORC_FORMAT_ERROR;
    const uint8_t *ipp = proof + 7 * 32;
    if (!sc_frombytes_canonical(&a, ipp + 2 * lg_n * 32) || !sc_frombytes_canonical(&b, ipp + 2 * lg_n * 32 + 32))
        return ORC_FORMAT_ERROR;
    /* verify_multiple */
    if (!(n == 8 || n == 16 || n == 32 || n == 64)) return ORC_INVALID_BITSIZE;
    if (gens_capacity < n) return ORC_INVALID_GENS_LENGTH;
    merlin_t tr; merlin_init(&tr, label, label_len);
    merlin_append(&tr, "dom-sep", (const uint8_t *)"rangeproof v1", 13);
    merlin_append_u64(&tr, "n", n); merlin_append_u64(&tr, "m", m);
    for (size_t j = 0; j < m; j++) merlin_append(&tr, "V", V32 + 32 * j, 32);
    static const uint8_t zero32[32] = {0};
    const char *names[4] = {"A", "S", "T_1", "T_2"};
    sc y, z, x, w;
    for (int i = 0; i < 2; i++) { if (!memcmp(proof + 32 * i, zero32, 32)) return ORC_OK; merlin_append(&tr, names[i], proof + 32 * i, 32); }
    merlin_challenge_scalar(&tr, "y", &y); merlin_challenge_scalar(&tr, "z", &z);
    for (int i = 2; i < 4; i++) { if (!memcmp(proof + 32 * i, zero32, 32)) return ORC_OK; merlin_append(&tr, names[i], proof + 32 * i, 32); }
    merlin_challenge_scalar(&tr, "x", &x);
    append_scalar(&tr, "t_x", &t_x); append_scalar(&tr, "t_x_blinding", &t_x_bl); append_scalar(&tr, "e_blinding", &e_bl);
    merlin_challenge_scalar(&tr, "w", &w);
    sc c; verifier_c(c_seed, c_index, &c);
    /* verification_scalars */
    size_t N = n * m;
    if (N != ((size_t)1 << lg_n)) return ORC_OK; /* VerificationError */
    merlin_append(&tr, "dom-sep", (const uint8_t *)"ipp v1", 6);
    merlin_append_u64(&tr, "n", N);
    sc *u = malloc(sizeof(sc) * (lg_n + 1)), *uinv = malloc(sizeof(sc) * (lg_n + 1));
    sc *usq = malloc(sizeof(sc) * (lg_n + 1)), *uinvsq = malloc(sizeof(sc) * (lg_n + 1));
    int rc = ORC_OK; int bad = 0;
    for (size_t k = 0; k < lg_n && !bad; k++) {
        const uint8_t *Lk = ipp + 64 * k, *Rk = Lk + 32;
        if (!memcmp(Lk, zero32, 32)) { bad = 1; break; }
        merlin_append(&tr, "L", Lk, 32);
        if (!memcmp(Rk, zero32, 32)) { bad = 1; break; }
        merlin_append(&tr, "R", Rk, 32);
        merlin_challenge_scalar(&tr, "u", &u[k]);
    }
    size_t npts = 8 + 2 * lg_n + 2 * N + m;
    sc *ks = malloc(sizeof(sc) * npts); ge *ps = malloc(sizeof(ge) * npts);
    sc *s = malloc(sizeof(sc) * N);
    if (!bad) {
        sc allinv = SC_ONE, t;
        for (size_t k = 0; k < lg_n; k++) { sc_invert(&uinv[k], &u[k]); sc_mul(&allinv, &allinv, &uinv[k]); sc_mul(&usq[k], &u[k], &u[k]); sc_mul(&uinvsq[k], &uinv[k], &uinv[k]); }
        s[0] = allinv;
        for (size_t i = 1; i < N; i++) {
            unsigned lg_i = 0; while (((size_t)2 << lg_i) <= i) lg_i++;
            size_t k = (size_t)1 << lg_i;
            sc_mul(&s[i], &s[i - k], &usq[(lg_n - 1) - lg_i]);
        }
        sc zz, minus_z, yinv;
        sc_mul(&zz, &z, &z); sc_neg(&minus_z, &z); sc_invert(&yinv, &y);
        size_t p = 0;
        /* points: A S T1 T2 L* R* B_blinding B G* H* V* */
        int dec_ok = 1;
        ks[p] = SC_ONE; dec_ok &= ristretto_decode(&ps[p], proof); p++;
        ks[p] = x; dec_ok &= ristretto_decode(&ps[p], proof + 32); p++;
        sc_mul(&ks[p], &c, &x); dec_ok &= ristretto_decode(&ps[p], proof + 64); p++;
        sc_mul(&ks[p], &ks[p - 1], &x); dec_ok &= ristretto_decode(&ps[p], proof + 96); p++;
        for (size_t k = 0; k < lg_n; k++) { ks[p] = usq[k]; dec_ok &= ristretto_decode(&ps[p], ipp + 64 * k); p++; }
        for (size_t k = 0; k < lg_n; k++) { ks[p] = uinvsq[k]; dec_ok &= ristretto_decode(&ps[p], ipp + 64 * k + 32); p++; }
        sc_mul(&t, &c, &t_x_bl); sc_add(&t, &t, &e_bl); sc_neg(&ks[p], &t); ps[p] = GE_BBLIND; p++;
        /* delta(n,m,y,z) */
        sc sum_y = SC_ONE, sum_2 = SC_ZERO, sum_z = SC_ZERO, pw, two;
        sc_from_u64(&two, 2);
        sum_y = SC_ZERO; pw = SC_ONE; for (size_t i = 0; i < N; i++) { sc_add(&sum_y, &sum_y, &pw); sc_mul(&pw, &pw, &y); }
        pw = SC_ONE; for (size_t i = 0; i < n; i++) { sc_add(&sum_2, &sum_2, &pw); sc_add(&pw, &pw, &pw); }
        pw = SC_ONE; for (size_t j = 0; j < m; j++) { sc_add(&sum_z, &sum_z, &pw); sc_mul(&pw, &pw, &z); }
        sc delta, q;
        sc_sub(&delta, &z, &zz); sc_mul(&delta, &delta, &sum_y);
        sc_mul(&q, &zz, &z); sc_mul(&q, &q, &sum_2); sc_mul(&q, &q, &sum_z); sc_sub(&delta, &delta, &q);
        sc bp; sc_mul(&t, &a, &b); sc_sub(&t, &t_x, &t); sc_mul(&bp, &w, &t);
        sc_sub(&t, &delta, &t_x); sc_mul(&t, &c, &t); sc_add(&ks[p], &bp, &t); ps[p] = GE_BASE; p++;
        /* g, h */
        ge *G = malloc(sizeof(ge) * N), *H = malloc(sizeof(ge) * N);
        for (size_t j = 0; j < m; j++) gens_party(j, n, G + j * n, H + j * n);
        for (size_t i = 0; i < N; i++) { sc_mul(&t, &a, &s[i]); sc_sub(&ks[p], &minus_z, &t); ps[p] = G[i]; p++; }
        sc exp_yinv = SC_ONE, exp_z = SC_ONE;
        for (size_t j = 0; j < m; j++) {
            sc exp_2 = SC_ONE;
            for (size_t i = 0; i < n; i++) {
                size_t k = j * n + i;
                sc z_and_2, v;
                sc_mul(&z_and_2, &exp_2, &exp_z);
                sc_mul(&v, &zz, &z_and_2); sc_mul(&t, &b, &s[N - 1 - k]); sc_sub(&v, &v, &t);
                sc_mul(&v, &v, &exp_yinv); sc_add(&ks[p], &z, &v); ps[p] = H[k]; p++;
                sc_mul(&exp_yinv, &exp_yinv, &yinv); sc_add(&exp_2, &exp_2, &exp_2);
            }
            sc_mul(&exp_z, &exp_z, &z);
        }
        exp_z = SC_ONE;
        for (size_t j = 0; j < m; j++) {
            sc_mul(&t, &c, &zz); sc_mul(&ks[p], &t, &exp_z);
            dec_ok &= ristretto_decode(&ps[p], V32 + 32 * j); p++;
            sc_mul(&exp_z, &exp_z, &z);
        }
        if (dec_ok) {
            ge res; ge_msm(&res, ks, ps, p);
            *ok = ge_is_identity_ristretto(&res);
        }
        free(G); free(H);
    }
    free(u); free(uinv); free(usq); free(uinvsq); free(ks); free(ps); free(s);
    return rc;
}

/* ------------------------------------------------------------------ range_proof_vec/mod.rs */
int orc_create_rangeproof(const float *values, size_t d, const uint8_t *blind32, size_t d_blind,
                          size_t prove_range, size_t n_partition, unsigned fp_bits, unsigned fp_frac,
                          const orc_nonce_t *ns, uint8_t *proofs_out, size_t *proof_len_out,
                          size_t *n_proofs_out, uint8_t *commits_out) {
    orc_init();
    if (d != d_blind) return ORC_WRONG_NUM_BLINDING;                      /* :22-24 */
    if (d == 0 || n_partition == 0 || prove_range == 0 || prove_range > fp_bits) return ORC_BAD_PARAM;
    if (is_out_of_range(values, d, (unsigned)prove_range, fp_bits, fp_frac)) return ORC_VALUE_OUT_OF_RANGE; /* :27-29 */
    sc offset; sc_from_u64(&offset, 1ULL << (prove_range - 1));          /* :36 */
    size_t dp = orc_next_pow2(d);
    uint64_t *v = calloc(dp, sizeof(uint64_t));
    uint8_t *bl = calloc(dp, 32);
    for (size_t i = 0; i < d; i++) {                                       /* :38-43 */
        sc s; int rc = f32_to_sc(values[i], fp_bits, fp_frac, &s);
        if (rc) { free(v); free(bl); return rc; }
        sc_add(&s, &s, &offset);
        uint8_t b[32]; sc_tobytes(b, &s);
        v[i] = read_from_bytes(b, fp_bits);
    }
    memcpy(bl, blind32, 32 * d);                                           /* :51 (pad = Scalar::zero) */
    size_t n_chunks = dp < n_partition ? dp : n_partition;                 /* :54 */
    size_t chunk = dp / n_chunks;                                          /* :55 */
    size_t n_proofs = (dp + chunk - 1) / chunk;
    /* every chunk must be a power of two for upstream (else the reference panics) */
    if (!is_pow2(chunk) || dp % chunk) { free(v); free(bl); return ORC_INVALID_AGGREGATION; }
    if (!(prove_range == 8 || prove_range == 16 || prove_range == 32 || prove_range == 64)) { free(v); free(bl); return ORC_INVALID_BITSIZE; }
    size_t plen = orc_proof_size(prove_range, chunk);
    uint8_t *V = malloc(32 * dp);
    int rc_any = 0;
    /* :75-78 par_iter over chunks (rayon in the reference; OpenMP here, ORC threads = min(P, OMP_NUM_THREADS)) */
#pragma omp parallel for schedule(dynamic, 1)
    for (long c = 0; c < (long)n_proofs; c++) {
        int rc = orc_bp_prove((const uint8_t *)"RangeProof", 10, prove_range, v + c * chunk, bl + 32 * c * chunk,
                              chunk, prove_range, ns, c * orc_nonces_per_chunk(prove_range, chunk),
                              proofs_out + c * plen, V + 32 * c * chunk);
        if (rc) {
#pragma omp critical
            rc_any = rc;
        }
    }
    if (rc_any) { free(v); free(bl); free(V); return rc_any; }
    /* :96-99 downshift: decompress, add commit(-offset, 0), truncate to d */
    sc noff; sc_neg(&noff, &offset);
    ge inv_off; ge_scalarmult(&inv_off, &noff, &GE_BASE);
    for (size_t i = 0; i < d; i++) {
        ge p, q; ristretto_decode(&p, V + 32 * i); ge_add(&q, &p, &inv_off); ristretto_encode(commits_out + 32 * i, &q);
    }
    *proof_len_out = plen; *n_proofs_out = n_proofs;
    free(v); free(bl); free(V);
    return ORC_OK;
}

int orc_verify_rangeproof(const uint8_t *proofs, size_t proof_len, size_t n_proofs,
                          const uint8_t *commits32, size_t d, size_t prove_range, unsigned fp_bits,
                          unsigned fp_frac, const uint8_t c_seed[32], int *ok) {
    (void)fp_frac;
    orc_init();
    *ok = 0;
    if (d == 0 || n_proofs == 0 || prove_range == 0 || prove_range > fp_bits) return ORC_BAD_PARAM;
    sc offset; sc_from_u64(&offset, 1ULL << (prove_range - 1));          /* :156-158 */
    ge off; ge_scalarmult(&off, &offset, &GE_BASE);
    size_t dp = orc_next_pow2(d);
    uint8_t *V = calloc(dp, 32);                                           /* identity encodes as 32 zero bytes (:163-167) */
    for (size_t i = 0; i < d; i++) {
        ge p, q; if (!ristretto_decode(&p, commits32 + 32 * i)) { free(V); return ORC_FORMAT_ERROR; }
        ge_add(&q, &p, &off); ristretto_encode(V + 32 * i, &q);
    }
    size_t chunk = dp / n_proofs;                                          /* :168 */
    if (chunk == 0) { free(V); return ORC_BAD_PARAM; }                     /* chunks(0) panics */
    size_t n_chunks = (dp + chunk - 1) / chunk;
    size_t nv = n_proofs < n_chunks ? n_proofs : n_chunks;                 /* zip truncates (:173-176) */
    int res = 1; int rc_first = ORC_OK;
    int *rcs = calloc(nv, sizeof(int)), *oks = calloc(nv, sizeof(int));
#pragma omp parallel for schedule(dynamic, 1)
    for (long c = 0; c < (long)nv; c++) {                                  /* :178-181 par_iter */
        size_t mlen = ((size_t)c + 1) * chunk <= dp ? chunk : dp - c * chunk;
        rcs[c] = orc_bp_verify((const uint8_t *)"RangeProof", 10, prove_range, proofs + c * proof_len, proof_len,
                               V + 32 * c * chunk, mlen, prove_range, c_seed, c, &oks[c]);
    }
    for (size_t c = 0; c < nv; c++) { if (rcs[c] && !rc_first) rc_first = rcs[c]; res &= oks[c]; }
    free(rcs); free(oks);
    free(V);
    if (rc_first) return rc_first;
    *ok = res;
    return ORC_OK;
}

/* ------------------------------------------------------------------ l2_range_proof_vec/mod.rs */
int orc_create_rangeproof_l2(const float *values, size_t d, const uint8_t *blind32, size_t d_blind,
                             size_t prove_range, size_t n_partition, unsigned fp_bits,
                             unsigned fp_frac, const orc_nonce_t *ns, uint8_t *proof_out,
                             size_t *proof_len_out, uint8_t commit_out[32]) {
    orc_init();
    if (d != d_blind) return ORC_WRONG_NUM_BLINDING;                      /* :21-23 */
    if (d == 0 || n_partition == 0 || prove_range == 0) return ORC_BAD_PARAM;
    if (is_out_of_range(values, d, (unsigned)prove_range, fp_bits, fp_frac)) return ORC_VALUE_OUT_OF_RANGE; /* :26-28 */
    sc val = SC_ZERO, bsum = SC_ZERO;
    volatile float val_float = 0.0f;
    float shift = (float)(1u << fp_frac);                                  /* i32::pow(2, frac) as f32 (:44) */
    for (size_t i = 0; i < d; i++) {                                       /* :37-50 */
        sc s, sq; int rc = f32_to_sc(values[i], fp_bits, fp_frac, &s);
        if (rc) return rc;
        sc_mul(&sq, &s, &s); sc_add(&val, &val, &sq);
        volatile float q = sc_to_f32(&s, fp_bits, fp_frac);
        volatile float qq = q * q;
        volatile float term = qq * shift;
        if (i == 0) val_float = term; else val_float = val_float + term;
        sc b; sc_frombytes_modorder(&b, blind32 + 32 * i); sc_add(&bsum, &bsum, &b);
    }
    float val_f = sc_to_f32(&val, fp_bits, fp_frac);
    volatile float diff = val_f - val_float;
    if (fabsf(diff) > 1.1920929e-07f) return ORC_OVERFLOW;                  /* :53-58 f32::EPSILON */
    if (val_f > orc_get_l2_clip_bounds((unsigned)prove_range, fp_bits, fp_frac)) return ORC_NORM_OUT_OF_RANGE; /* :60-64 */
    uint8_t vb[32]; sc_tobytes(vb, &val);
    uint64_t v = read_from_bytes(vb, fp_bits);                             /* :69-73 */
    uint8_t bb[32]; sc_tobytes(bb, &bsum);
    uint8_t V[32];
    int rc = orc_bp_prove((const uint8_t *)"L2RangeProof", 12, 64, &v, bb, 1, prove_range, ns, 0, proof_out, V); /* :156-171 */
    if (rc) return rc;
    *proof_len_out = orc_proof_size(prove_range, 1);
    memcpy(commit_out, V, 32);
    return ORC_OK;
}

int orc_verify_rangeproof_l2(const uint8_t *proof, size_t proof_len, const uint8_t commit[32],
                             size_t prove_range, unsigned fp_bits, unsigned fp_frac,
                             const uint8_t c_seed[32], int *ok) {
    (void)fp_bits; (void)fp_frac;
    orc_init();
    *ok = 0;
    ge p; uint8_t V[32];
    if (!ristretto_decode(&p, commit)) return ORC_FORMAT_ERROR;
    ristretto_encode(V, &p);
    int okc = 0;
    int rc = orc_bp_verify((const uint8_t *)"L2RangeProof", 12, 64, proof, proof_len, V, 1, prove_range, c_seed, 0, &okc);
    if (rc) return rc;
    *ok = okc;
    return ORC_OK;
}

/* ------------------------------------------------------------------ Sigma-proofs (rand_proof, square_rand_proof, square_proof) */
/* kind 0 RandProof       : commitments L|R       , proof L'|R'|Z_m|Z_r                    (rand_proof/mod.rs:31-85)
 * kind 1 SquareRandProof : commitments L|R|c_sq  , proof L'|R'|c_sq'|Z_m|Z_r1|Z_r2        (square_rand_proof/mod.rs:41-151)
 * kind 2 SquareProof     : commitments c_l|c_sq  , proof c_l'|c_sq'|Z_m|Z_r1|Z_r2          (square_proof/mod.rs, party.rs:14-151) */
static void eg_bytes(uint8_t out[64], const ge *L, const ge *R) { ristretto_encode(out, L); ristretto_encode(out + 32, R); }
static void sigma_layout(int kind, size_t *plen, size_t *clen, size_t *nn, int *has_R, int *has_sq) {
    *has_R = kind != 2; *has_sq = kind != 0;
    size_t npts = 1 + (size_t)*has_R + (size_t)*has_sq;
    *clen = 32 * npts; *nn = *has_sq ? 3 : 2; *plen = 32 * (npts + *nn);
}
static void sigma_transcript(int kind, merlin_t *tr, const uint8_t *cm, const uint8_t *pf, int has_R) {
    if (kind == 0) {
        merlin_init(tr, (const uint8_t *)"RandProof", 9);
        merlin_append(tr, "dom-sep", (const uint8_t *)"randomness proof v1", 19);
        merlin_append(tr, "C", cm, 64); merlin_append(tr, "C_prime", pf, 64);
        return;
    }
    size_t w = has_R ? 64 : 32;
    if (kind == 1) merlin_init(tr, (const uint8_t *)"SquareRandProof", 15); else merlin_init(tr, (const uint8_t *)"SquareProof", 11);
    merlin_append(tr, "dom-sep", (const uint8_t *)"randomness proof v1", 19);
    merlin_append(tr, "C_eg", cm, w); merlin_append(tr, "C_ped", cm + w, 32);
    merlin_append(tr, "C_prime_eg", pf, w); merlin_append(tr, "C_prime_ped", pf + w, 32);
}

int orc_sigma_create(int kind, const float *values, size_t d, const uint8_t *r1_32, size_t d_r1, const uint8_t *r2_32,
                     const uint8_t *existing32, unsigned fp_bits, unsigned fp_frac, const orc_nonce_t *ns,
                     uint8_t *proofs_out, uint8_t *commits_out) {
    orc_init();
    if (d != d_r1) return ORC_WRONG_NUM_BLINDING;        /* rand_proof_vec/mod.rs:18-20, square_rand_proof_vec/mod.rs:24-26 */
    size_t plen, clen, nn; int has_R, has_sq;
    sigma_layout(kind, &plen, &clen, &nn, &has_R, &has_sq);
    size_t sq_off = has_R ? 64 : 32;
    for (size_t i = 0; i < d; i++) {
        sc m, r1, r2 = SC_ZERO, mp, r1p, r2p = SC_ZERO, c, t;
        int rc = f32_to_sc(values[i], fp_bits, fp_frac, &m);
        if (rc) return rc;
        sc_frombytes_modorder(&r1, r1_32 + 32 * i);
        if (has_sq) sc_frombytes_modorder(&r2, r2_32 + 32 * i);
        ge L, R, Csq, Lp, Rp, Csqp, tmp;
        if (existing32) { if (!ristretto_decode(&L, existing32 + 32 * i)) return ORC_FORMAT_ERROR; }   /* complete_existing / p_base = m_com */
        else pedersen_commit(&L, &m, &r1);
        uint8_t *cm = commits_out + clen * i, *pf = proofs_out + plen * i;
        ristretto_encode(cm, &L);
        if (has_R) { ge_scalarmult(&R, &r1, &GE_BASE); ristretto_encode(cm + 32, &R); }
        if (has_sq) { sc msq; sc_mul(&msq, &m, &m); pedersen_commit(&Csq, &msq, &r2); ristretto_encode(cm + sq_off, &Csq); }
        /* nonce draw order: m', r1' (, r2') -- party.rs:43-45 / rand_proof/party.rs:23-24 */
        nonce_get(ns, nn * i, &mp); nonce_get(ns, nn * i + 1, &r1p); if (has_sq) nonce_get(ns, nn * i + 2, &r2p);
        pedersen_commit(&Lp, &mp, &r1p); ristretto_encode(pf, &Lp);
        if (has_R) { ge_scalarmult(&Rp, &r1p, &GE_BASE); ristretto_encode(pf + 32, &Rp); }
        if (has_sq) {   /* c_sq' = m' * c.L + r2' * B_blinding */
            ge_scalarmult(&Csqp, &mp, &L); ge_scalarmult(&tmp, &r2p, &GE_BBLIND); ge_add(&Csqp, &Csqp, &tmp);
            ristretto_encode(pf + sq_off, &Csqp);
        }
        merlin_t tr; sigma_transcript(kind, &tr, cm, pf, has_R);
        merlin_challenge_scalar(&tr, "c", &c);
        uint8_t *z = pf + clen;
        sc_mul(&t, &m, &c); sc_add(&t, &t, &mp); sc_tobytes(z, &t);                    /* z_m = m' + m c */
        sc_mul(&t, &r1, &c); sc_add(&t, &t, &r1p); sc_tobytes(z + 32, &t);             /* z_r1 = r1' + r1 c */
        if (has_sq) { sc u; sc_mul(&u, &m, &r1); sc_sub(&u, &r2, &u); sc_mul(&u, &u, &c); sc_add(&u, &u, &r2p); sc_tobytes(z + 64, &u); }
    }
    return ORC_OK;
}

int orc_sigma_verify(int kind, const uint8_t *proofs, const uint8_t *commits, size_t d, int *ok) {
    orc_init();
    *ok = 0;
    size_t plen, clen, nn; int has_R, has_sq;
    sigma_layout(kind, &plen, &clen, &nn, &has_R, &has_sq);
    size_t sq_off = has_R ? 64 : 32;
    int all = 1;
    for (size_t i = 0; i < d; i++) {
        const uint8_t *pf = proofs + plen * i, *cm = commits + clen * i, *z = pf + clen;
        ge L, R, Csq, Lp, Rp, Csqp;
        sc zm, zr1, zr2 = SC_ZERO, c;
        if (!ristretto_decode(&L, cm) || !ristretto_decode(&Lp, pf)) return ORC_FORMAT_ERROR;
        if (has_R && (!ristretto_decode(&R, cm + 32) || !ristretto_decode(&Rp, pf + 32))) return ORC_FORMAT_ERROR;
        if (has_sq && (!ristretto_decode(&Csq, cm + sq_off) || !ristretto_decode(&Csqp, pf + sq_off))) return ORC_FORMAT_ERROR;
        if (!sc_frombytes_canonical(&zm, z) || !sc_frombytes_canonical(&zr1, z + 32)) return ORC_FORMAT_ERROR;
        if (has_sq && !sc_frombytes_canonical(&zr2, z + 64)) return ORC_FORMAT_ERROR;
        merlin_t tr; sigma_transcript(kind, &tr, cm, pf, has_R);
        merlin_challenge_scalar(&tr, "c", &c);
        ge lhs, rhs, t;
        pedersen_commit(&lhs, &zm, &zr1); ge_scalarmult(&t, &c, &L); ge_add(&rhs, &Lp, &t);       /* commit(Z_m, Z_r1) == C'.L + c C.L */
        int e1 = ge_eq_ristretto(&lhs, &rhs), e2 = 1, e3 = 1;
        if (has_R) { ge_scalarmult(&lhs, &zr1, &GE_BASE); ge_scalarmult(&t, &c, &R); ge_add(&rhs, &Rp, &t); e2 = ge_eq_ristretto(&lhs, &rhs); }
        if (has_sq) {   /* Z_m * C.L + Z_r2 * B_blinding == c_sq' + c * c_sq */
            ge_scalarmult(&lhs, &zm, &L); ge_scalarmult(&t, &zr2, &GE_BBLIND); ge_add(&lhs, &lhs, &t);
            ge_scalarmult(&t, &c, &Csq); ge_add(&rhs, &Csqp, &t);
            e3 = ge_eq_ristretto(&lhs, &rhs);
        }
        all &= e1 & e2 & e3;
    }
    *ok = all;
    return ORC_OK;
}

/* ------------------------------------------------------------------ compressed_rand_proof (mod.rs:43-102, party.rs:54-100, dealer.rs)
 * ONE proof for all d ElGamal pairs: z_m = m' + sum m_i c^(i+1), z_r likewise; transcript label "CompressedRandProof",
 * pair i appended under the 3-byte label UNIQUE_U8_TRIPLETS[i] = [(3i)%256, (3i+1)%256, (3i+2)%256]
 * (generate_unique_u8_triplets.py:8-13; the 900 000-entry table itself is a missing large blob of the reference). */
static void compressed_transcript(merlin_t *tr, const uint8_t *pairs, size_t d, const uint8_t cprime[64], sc *c) {
    merlin_init(tr, (const uint8_t *)"CompressedRandProof", 19);
    merlin_append(tr, "dom-sep", (const uint8_t *)"randomness proof v1", 19);
    for (size_t i = 0; i < d; i++) {
        uint8_t lbl[3] = {(uint8_t)(3 * i), (uint8_t)(3 * i + 1), (uint8_t)(3 * i + 2)};
        merlin_append_lbl(tr, lbl, 3, pairs + 64 * i, 64);
    }
    merlin_append(tr, "C_prime_eg", cprime, 64);
    merlin_challenge_scalar(tr, "c", c);
}
int orc_compressed_create(const float *values, size_t d, const uint8_t *r32, size_t d_r, const uint8_t *existing32,
                          unsigned fp_bits, unsigned fp_frac, const orc_nonce_t *ns, uint8_t proof_out[128], uint8_t *pairs_out) {
    orc_init();
    if (d != d_r) return ORC_WRONG_NUM_BLINDING;
    if (d >= 900000) return ORC_BAD_PARAM;                      /* label table size */
    sc *m = malloc(sizeof(sc) * (d ? d : 1)), *r = malloc(sizeof(sc) * (d ? d : 1));
    for (size_t i = 0; i < d; i++) {
        int rc = f32_to_sc(values[i], fp_bits, fp_frac, &m[i]);
        if (rc) { free(m); free(r); return rc; }
        sc_frombytes_modorder(&r[i], r32 + 32 * i);
        ge L, R;
        if (existing32) { if (!ristretto_decode(&L, existing32 + 32 * i)) { free(m); free(r); return ORC_FORMAT_ERROR; } }
        else pedersen_commit(&L, &m[i], &r[i]);
        ge_scalarmult(&R, &r[i], &GE_BASE);
        eg_bytes(pairs_out + 64 * i, &L, &R);
    }
    sc mp, rp, c; nonce_get(ns, 0, &mp); nonce_get(ns, 1, &rp);
    ge Lp, Rp; pedersen_commit(&Lp, &mp, &rp); ge_scalarmult(&Rp, &rp, &GE_BASE);
    eg_bytes(proof_out, &Lp, &Rp);
    merlin_t tr; compressed_transcript(&tr, pairs_out, d, proof_out, &c);
    sc zm = mp, zr = rp, pw = c, t;
    for (size_t i = 0; i < d; i++) { sc_mul(&t, &m[i], &pw); sc_add(&zm, &zm, &t); sc_mul(&t, &r[i], &pw); sc_add(&zr, &zr, &t); sc_mul(&pw, &pw, &c); }
    sc_tobytes(proof_out + 64, &zm); sc_tobytes(proof_out + 96, &zr);
    free(m); free(r);
    return ORC_OK;
}
int orc_compressed_verify(const uint8_t proof[128], const uint8_t *pairs, size_t d, int *ok) {
    orc_init();
    *ok = 0;
    ge Lp, Rp; sc zm, zr, c;
    if (!ristretto_decode(&Lp, proof) || !ristretto_decode(&Rp, proof + 32)) return ORC_FORMAT_ERROR;
    if (!sc_frombytes_canonical(&zm, proof + 64) || !sc_frombytes_canonical(&zr, proof + 96)) return ORC_FORMAT_ERROR;
    if (d >= 900000) return ORC_BAD_PARAM;
    size_t n = d ? d : 1;
    ge *Ls = malloc(sizeof(ge) * n), *Rs = malloc(sizeof(ge) * n); sc *pw = malloc(sizeof(sc) * n);
    for (size_t i = 0; i < d; i++)
        if (!ristretto_decode(&Ls[i], pairs + 64 * i) || !ristretto_decode(&Rs[i], pairs + 64 * i + 32)) { free(Ls); free(Rs); free(pw); return ORC_FORMAT_ERROR; }
    merlin_t tr; compressed_transcript(&tr, pairs, d, proof, &c);
    sc p = c; for (size_t i = 0; i < d; i++) { pw[i] = p; sc_mul(&p, &p, &c); }
    ge sL, sR, lhs, rhs;
    ge_msm(&sL, pw, Ls, d); ge_msm(&sR, pw, Rs, d);
    pedersen_commit(&lhs, &zm, &zr); ge_add(&rhs, &Lp, &sL);
    int e1 = ge_eq_ristretto(&lhs, &rhs);
    ge_scalarmult(&lhs, &zr, &GE_BASE); ge_add(&rhs, &Rp, &sR);
    int e2 = ge_eq_ristretto(&lhs, &rhs);
    *ok = e1 & e2;
    free(Ls); free(Rs); free(pw);
    return ORC_OK;
}

/* ------------------------------------------------------------------ bsgs32.rs:14-73 (server-side extraction of the aggregate)
 * table[compress(x B)] = x for x = 0..m ; solve: up to max_it = 2^bsgs_bits / m giant steps of -mG, value (i*m + pow)
 * truncated to bsgs_bits bits (BSGS_URawFix); solve_discrete_log_with_neg tries -M second and negates.
 * Not found in either direction: the reference unwraps None (panic) -> ORC_BAD_PARAM. */
typedef struct { uint8_t key[32]; uint32_t val; } bsgs_ent;
static int bsgs_cmp(const void *a, const void *b) { return memcmp(((const bsgs_ent *)a)->key, ((const bsgs_ent *)b)->key, 32); }
static int bsgs_lookup(const bsgs_ent *tab, size_t n, const ge *p, uint32_t *val) {
    bsgs_ent k; ristretto_encode(k.key, p);
    const bsgs_ent *e = bsearch(&k, tab, n, sizeof(bsgs_ent), bsgs_cmp);
    if (!e) return 0;
    *val = e->val; return 1;
}
int orc_bsgs_solve(const uint8_t *points32, size_t d, size_t m, unsigned bsgs_bits, uint8_t *scalars_out) {
    orc_init();
    if (m == 0 || bsgs_bits == 0 || bsgs_bits > 32) return ORC_BAD_PARAM;
    bsgs_ent *tab = malloc(sizeof(bsgs_ent) * (m + 1));
    ge cur; ge_identity(&cur);
    for (size_t x = 0; x <= m; x++) { ristretto_encode(tab[x].key, &cur); tab[x].val = (uint32_t)(x & ((bsgs_bits >= 32) ? 0xffffffffu : ((1u << bsgs_bits) - 1))); ge_add(&cur, &cur, &GE_BASE); }
    qsort(tab, m + 1, sizeof(bsgs_ent), bsgs_cmp);
    sc msc; sc_from_u64(&msc, (uint64_t)m & ((bsgs_bits >= 32) ? 0xffffffffull : ((1ull << bsgs_bits) - 1)));   /* Scalar::from(m as BSGS_URawFix) */
    ge mG; ge_scalarmult(&mG, &msc, &GE_BASE);
    uint64_t max_it = (1ull << bsgs_bits) / m;
    uint64_t mask = (bsgs_bits >= 32) ? 0xffffffffull : ((1ull << bsgs_bits) - 1);
    int rc = ORC_OK;
    for (size_t i = 0; i < d && !rc; i++) {
        ge M; if (!ristretto_decode(&M, points32 + 32 * i)) { rc = ORC_FORMAT_ERROR; break; }
        int found = 0; uint64_t value = 0;
        for (int neg = 0; neg < 2 && !found; neg++) {
            ge c = M; if (neg) ge_neg(&c, &M);
            for (uint64_t it = 0; it < max_it; it++) {
                uint32_t pw;
                if (bsgs_lookup(tab, m + 1, &c, &pw)) { value = (it * m + pw) & mask; found = 1 + neg; break; }
                ge_sub(&c, &c, &mG);
            }
        }
        if (!found) { rc = ORC_BAD_PARAM; break; }
        sc v; sc_from_u64(&v, value);
        if (found == 2) sc_neg(&v, &v);
        sc_tobytes(scalars_out + 32 * i, &v);
    }
    free(tab);
    return rc;
}
