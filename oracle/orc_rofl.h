/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement of the reference hot path:
 *   rofl_crypto/src/range_proof_vec/mod.rs      (create_rangeproof :16-102, verify_rangeproof :149-191, helpers)
 *   rofl_crypto/src/l2_range_proof_vec/mod.rs   (create_rangeproof_l2 :15-140, verify_rangeproof_l2 :185-253)
 *   rofl_crypto/src/pedersen_ops.rs             (:9-127)
 *   rofl_crypto/src/conversion32.rs             (:11-66), rofl_crypto/src/fp.rs (feature table, runtime here)
 * plus the published algorithm of the un-vendored crate bulletproofs 4.0.0
 * (RangeProof::prove_multiple / verify_multiple, InnerProductProof, BulletproofGens, PedersenGens),
 * anchored on the reference's call sites range_proof_vec/mod.rs:124-135,200-209 and
 * l2_range_proof_vec/mod.rs:162-171,236-246.
 *
 * PARITY STATUS: the reference holds no golden byte vectors for this path and its crates cannot be
 * built here (no Rust toolchain, crates not vendored).  Primitives are pinned against libsodium /
 * hashlib / the Merlin published test vector (tests/golden); proof *bytes* vs the real
 * bulletproofs crate are "parity unpinned".
 *
 * Return codes (shared with the product C ABI, include/rofl_zk.h).
 */
#ifndef ORC_ROFL_H
#define ORC_ROFL_H
#include <stdint.h>
#include <stddef.h>

enum {
    ORC_OK = 0,
    ORC_WRONG_NUM_BLINDING = 1,
    ORC_VALUE_OUT_OF_RANGE = 2,
    ORC_INVALID_BITSIZE = 3,
    ORC_INVALID_AGGREGATION = 4,
    ORC_FORMAT_ERROR = 5,
    ORC_INVALID_GENS_LENGTH = 6,
    ORC_NORM_OUT_OF_RANGE = 7,
    ORC_OVERFLOW = 8,
    ORC_SUM_ERROR = 9,
    ORC_NON_FINITE = 10,      /* reference panics (fixed::saturating_from_float on NaN) */
    ORC_BAD_PARAM = 11,       /* reference panics (d == 0, n_partition == 0, shift overflow...) */
    ORC_NONCE_SHORT = 12      /* explicit nonce stream too short */
};

/* nonce source: mode 0 = explicit stream of 64-byte wide scalars in reference draw order,
 *               mode 1 = 32-byte seed: scalar k = wide-reduce of bytes 64 (k & 1) .. + 64 of SHAKE256("rofl-zk/nonce/v2"||seed||u64le(k >> 1)) */
typedef struct {
    int mode;
    const uint8_t *stream;
    size_t stream_scalars;
    uint8_t seed[32];
} orc_nonce_t;

size_t orc_next_pow2(size_t v);
size_t orc_proof_size(size_t n_bits, size_t m);            /* 32*(9+2*lg(n*m)) */
size_t orc_nonces_per_chunk(size_t n_bits, size_t m);      /* m*(2n+4) */

/* primitives exposed for KAT tests */
void orc_ristretto_from_uniform(const uint8_t in[64], uint8_t out[32]);
int orc_ristretto_scalarmult(const uint8_t k[32], const uint8_t p[32], uint8_t out[32]);
void orc_ristretto_scalarmult_base(const uint8_t k[32], uint8_t out[32]);
int orc_ristretto_add(const uint8_t p[32], const uint8_t q[32], uint8_t out[32]);
int orc_ristretto_is_valid(const uint8_t p[32]);
void orc_sc_reduce_wide(const uint8_t in[64], uint8_t out[32]);
void orc_sc_mul(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]);
void orc_sc_add(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]);
void orc_sc_neg(const uint8_t a[32], uint8_t out[32]);
void orc_sc_invert(const uint8_t a[32], uint8_t out[32]);
void orc_sha3_512(const uint8_t *in, size_t len, uint8_t out[64]);
void orc_shake256(const uint8_t *in, size_t len, uint8_t *out, size_t outlen);
void orc_keccak_f1600(uint64_t st[25]);
void orc_merlin_init(void *t256, const uint8_t *label, size_t len);
void orc_merlin_append(void *t256, const char *label, const uint8_t *msg, size_t len);
void orc_merlin_challenge(void *t256, const char *label, uint8_t *out, size_t len);
void orc_pedersen_gens(uint8_t B[32], uint8_t B_blinding[32]);
void orc_bp_gens(size_t n, size_t m, uint8_t *G_out, uint8_t *H_out); /* party-major, n*m*32 each */
void orc_nonce_scalar(const orc_nonce_t *ns, uint64_t idx, uint8_t out[32]);
void orc_msm(const uint8_t *scalars, const uint8_t *points, size_t n, uint8_t out[32]);

/* conversion32.rs / fp.rs with runtime (fp_bits, fp_frac) */
int orc_f32_to_scalar(float v, unsigned fp_bits, unsigned fp_frac, uint8_t out[32]);
float orc_scalar_to_f32(const uint8_t s[32], unsigned fp_bits, unsigned fp_frac);
void orc_get_clip_bounds(unsigned range, unsigned fp_bits, unsigned fp_frac, float *mn, float *mx);
float orc_get_l2_clip_bounds(unsigned range, unsigned fp_bits, unsigned fp_frac);
int orc_fp_square(const uint8_t s32[32], unsigned fp_bits, unsigned fp_frac, uint8_t out[32]);
void orc_scalar_powers(const uint8_t v32[32], size_t count, uint8_t *out);
int orc_f32_to_fp(float v, unsigned fp_bits, unsigned fp_frac, uint64_t *out);
float orc_uint_to_f32(uint64_t k, unsigned fp_bits, unsigned fp_frac);
void orc_clip_f32(const float *in, size_t d, unsigned range, unsigned fp_bits, unsigned fp_frac, float *out);

/* pedersen_ops.rs */
void orc_commit_vec(const uint8_t *values32, const uint8_t *blind32, size_t d, uint8_t *out32);
int orc_add_points_vec(const uint8_t *a32, const uint8_t *b32, size_t d, uint8_t *out32);
int orc_shift_points(const uint8_t *a32, size_t d, const uint8_t offset[32], uint8_t *out32);

/* upstream bulletproofs level */
int orc_bp_prove(const uint8_t *label, size_t label_len, size_t gens_capacity, const uint64_t *values,
                 const uint8_t *blind32, size_t m, size_t n_bits, const orc_nonce_t *ns,
                 uint64_t nonce_base, uint8_t *proof_out, uint8_t *V_out);
/* returns ORC_OK with *ok = 0/1, or an error code */
int orc_bp_verify(const uint8_t *label, size_t label_len, size_t gens_capacity, const uint8_t *proof,
                  size_t proof_len, const uint8_t *V32, size_t m, size_t n_bits,
                  const uint8_t c_seed[32], uint64_t c_index, int *ok);

/* range_proof_vec */
int orc_create_rangeproof(const float *values, size_t d, const uint8_t *blind32, size_t d_blind,
                          size_t prove_range, size_t n_partition, unsigned fp_bits, unsigned fp_frac,
                          const orc_nonce_t *ns, uint8_t *proofs_out, size_t *proof_len_out,
                          size_t *n_proofs_out, uint8_t *commits_out);
int orc_verify_rangeproof(const uint8_t *proofs, size_t proof_len, size_t n_proofs,
                          const uint8_t *commits32, size_t d, size_t prove_range, unsigned fp_bits,
                          unsigned fp_frac, const uint8_t c_seed[32], int *ok);
/* l2_range_proof_vec */
int orc_create_rangeproof_l2(const float *values, size_t d, const uint8_t *blind32, size_t d_blind,
                             size_t prove_range, size_t n_partition, unsigned fp_bits,
                             unsigned fp_frac, const orc_nonce_t *ns, uint8_t *proof_out,
                             size_t *proof_len_out, uint8_t commit_out[32]);
int orc_verify_rangeproof_l2(const uint8_t *proof, size_t proof_len, const uint8_t commit[32],
                             size_t prove_range, unsigned fp_bits, unsigned fp_frac,
                             const uint8_t c_seed[32], int *ok);

/* ---- per-element Sigma-proofs (rand_proof, square_rand_proof and their _vec wrappers) ----
 * kind 0: RandProof        (rand_proof/mod.rs:31-85, party.rs:14-85): proof 128 B, commitment = ElGamalPair 64 B,
 *         nonces per element m', r' ; transcript label "RandProof"
 * kind 1: SquareRandProof  (square_rand_proof/mod.rs:41-151, party.rs:14-160): proof 192 B, commitments 96 B,
 *         nonces per element m', r1', r2' ; transcript label "SquareRandProof"
 * `existing32` (may be NULL) = value commitments to complete (prove_existing: L = m_com). */
#ifndef ORC_SIGMA_DECL
#define ORC_SIGMA_DECL
int orc_sigma_create(int kind, const float *values, size_t d, const uint8_t *r1_32, size_t d_r1, const uint8_t *r2_32,
                     const uint8_t *existing32, unsigned fp_bits, unsigned fp_frac, const orc_nonce_t *ns,
                     uint8_t *proofs_out, uint8_t *commits_out);
int orc_sigma_verify(int kind, const uint8_t *proofs, const uint8_t *commits, size_t d, int *ok);
/* kind 2: SquareProof (square_proof/mod.rs): commitments c_l|c_sq (64 B), proof 160 B */
/* compressed_rand_proof/mod.rs:43-102: one 128-byte proof for d ElGamal pairs (d*64 B) */
int orc_compressed_create(const float *values, size_t d, const uint8_t *r32, size_t d_r, const uint8_t *existing32,
                          unsigned fp_bits, unsigned fp_frac, const orc_nonce_t *ns, uint8_t proof_out[128], uint8_t *pairs_out);
int orc_compressed_verify(const uint8_t proof[128], const uint8_t *pairs, size_t d, int *ok);
/* bsgs32.rs: BSGSTable::new(m) + solve_discrete_log_with_neg per point (pedersen_ops.rs:27-53 discrete_log_vec*) */
int orc_bsgs_solve(const uint8_t *points32, size_t d, size_t m, unsigned bsgs_bits, uint8_t *scalars_out);
#endif
#endif
