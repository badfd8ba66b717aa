/*
 * librofl_zk.so -- MI355X (gfx950) drop-in C ABI for rofl_crypto's ZK norm-bound hot path.
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference repo,
 * rofl_crypto/src/...).  Data formats at the boundary:
 *   scalars     : 32 bytes little-endian, canonical (curve25519_dalek_ng::scalar::Scalar::to_bytes)
 *   points      : 32 bytes compressed Ristretto (CompressedRistretto)
 *   range proofs: bulletproofs 4.0.0 RangeProof::to_bytes layout, 32*(9 + 2*lg(n*m)) bytes per chunk
 *   values      : IEEE f32
 * (fp_bits, fp_frac) are the reference's compile-time cargo features fp{8,16,32,64} / frac{0..12}
 * (fp.rs:8-139), taken at run time here (mirrors ModelConfig.fp_bits/fp_frac, flservice.proto:54-55).
 *
 * All buffers are caller-owned host memory unless the name ends in _dev (HIP device pointers on the
 * library's current device).  The library never retains caller pointers after return, and it never
 * hands caller memory to the HIP runtime: host buffers may be ordinary (pageable, freshly allocated)
 * memory; transfers of 32 KB and more are staged through the library's own pinned memory, inputs are
 * consumed and outputs complete when the call returns.
 * No C++ exceptions cross this boundary.  Every entry point is thread-safe; concurrent calls run side by side on the lanes (HIP stream +
 * workspace) of their device, ROFL_LANES of them per device.
 *
 * Return codes: 0 ok; 1 WrongNumBlindingFactors; 2 ValueOutOfRangeError; 3 InvalidBitsize;
 * 4 InvalidAggregation; 5 FormatError; 6 InvalidGeneratorsLength; 7 NormOutOfRangeError;
 * 8 OverflowError; 9 SumError; 10 non-finite input (reference panics); 11 bad parameter
 * (reference panics); 12 nonce stream too short; 99 RCCL error (rofl_comm_*); >= 100 HIP runtime error (100 + hipError_t).
 * A failed verification is NOT an error: it is reported through *ok_out = 0 with return code 0
 * (range_proof_vec/mod.rs:210-215 maps VerificationError to Ok(false)).
 */
#ifndef ROFL_ZK_H
#define ROFL_ZK_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum {
    ROFL_OK = 0, ROFL_WRONG_NUM_BLINDING = 1, ROFL_VALUE_OUT_OF_RANGE = 2, ROFL_INVALID_BITSIZE = 3,
    ROFL_INVALID_AGGREGATION = 4, ROFL_FORMAT_ERROR = 5, ROFL_INVALID_GENS_LENGTH = 6,
    ROFL_NORM_OUT_OF_RANGE = 7, ROFL_OVERFLOW = 8, ROFL_SUM_ERROR = 9, ROFL_NON_FINITE = 10,
    ROFL_BAD_PARAM = 11, ROFL_NONCE_SHORT = 12, ROFL_COMM_ERROR = 99, ROFL_HIP_ERROR = 100
};

/* Prover randomness.  The reference draws every nonce from rand::thread_rng() inside
 * bulletproofs::RangeProof::prove_multiple; for reproducible (bit-exact) proofs the stream is an
 * explicit input here.
 *   mode 0: `stream` holds 64-byte wide scalars in the reference draw order
 *           (per chunk c, offset c*m*(2n+4): per party a_blinding, s_blinding, s_L[0..n), s_R[0..n);
 *            then per party t_1_blinding, t_2_blinding), each reduced like Scalar::from_bytes_mod_order_wide.
 *   mode 1: scalar k = wide-reduce(SHAKE256("rofl-zk/nonce/v2" || seed[32] || u64le(k >> 1))[64 (k & 1) .. 64 (k & 1) + 64]):
 *           two wide scalars per block of the XOF (its rate is 136 bytes). */
typedef struct {
    int mode;
    const uint8_t *stream;
    size_t stream_scalars;
    uint8_t seed[32];
} rofl_nonce_t;

/* ---- context / device ----
 * One process drives any number of GPUs (the reference's server is ONE process that hands its clients to a verification pool,
 * rofl_service/src/flserver/server.rs:379-384, 513-521, 656-687).  The device a call runs on is a property of the CALLING THREAD, as with
 * hipSetDevice: rofl_set_device(d) binds the calling thread to device d and brings that device's context up (an unusable device is reported
 * here and leaves the binding unchanged).  The FIRST successful call of the process also makes d the default of threads that never call it
 * -- a one-GPU host sets it once and calls from any thread; later calls bind their own thread only (rofl_set_option("default_device", d)
 * moves the default explicitly).  A multi-GPU server binds each pool thread to its device, or lists the devices in
 * rofl_set_option("devices", mask) and lets rofl_create_rangeproof_batch / rofl_verify_rangeproof_batch spread their clients.
 * Generator tables are cached per device. */
int rofl_set_device(int device);
int rofl_get_device(int *device_out);                  /* the device the calling thread's next call runs on */
/* The binding half of rofl_set_device alone: the calling thread's calls go to `device` from now on (-1: back to the process default), no HIP
 * call, no lane taken -- for the worker threads of a host-side pool that run calls on behalf of a thread that has already brought the device
 * up (rofl_project_code_amd/params.py binds its pool workers to the submitting thread's device this way; a rayon pool would do it in its
 * start handler after one rofl_set_device per device). */
int rofl_bind_device(int device);
int rofl_last_error(char *buf, size_t len);            /* human-readable text of the calling thread's last failure */
/* BulletproofGens::new(n_bits, m) (generators.rs; re-run by the reference on every helper call,
 * range_proof_vec/mod.rs:126,201) -- built once on the device and cached per (n_bits, m).  When this call returns the shape's tables are
 * complete: a server calls it at start-up.  A create / verify call that meets a new shape does not wait for the large fold table (tens of
 * GB at the paper's sizes: its allocation alone can take most of a second): it is served from a compact table at once, and a background
 * thread builds the full one in the first quiet moment (no call in flight for 20 ms; after 3 s at the latest) and swaps it in.
 * If HBM is too short for the full table even after every table nobody reads has been evicted, the shape keeps its compact table (same
 * results; the first generator fold of a proof is ~2.5 ms slower): the call still returns 0, rofl_last_error carries a note,
 * rofl_bp_gens_table_bytes reports the compact size, and a later rofl_bp_gens_prepare tries again. */
int rofl_bp_gens_prepare(size_t n_bits, size_t m);
/* The same for a process that only VERIFIES the shape -- the reference's server (rofl_service/src/flserver/server.rs:656-687): the generators
 * and the window slices of the fixed-base MSM, without the prover's fold table (2.2 GB instead of 104 GB at BASELINE cfg 4).  The verify
 * entry points build exactly this on first use of a shape, so the call is only there to pay it at start-up.  Tables are role-aware either
 * way: nothing on a verify path ever builds, holds or evicts for a fold table; a create call (or rofl_bp_gens_prepare) that later meets the
 * shape adds it. */
int rofl_bp_gens_prepare_verify(size_t n_bits, size_t m);
/* HBM held by the cached tables of (n_bits, m): generators + fold slices (prover role only) + window slices; 0 if they have not been built */
int rofl_bp_gens_table_bytes(size_t n_bits, size_t m, size_t *bytes_out);
/* copy the cached generators back, compressed, party-major: G_out/H_out n_bits*m*32 bytes each */
int rofl_bp_gens_export(size_t n_bits, size_t m, uint8_t *G_out, uint8_t *H_out);

/* ---- sizes ---- */
size_t rofl_next_pow2(size_t v);                       /* range_proof_vec/mod.rs:225-235 */
size_t rofl_rangeproof_chunks(size_t d, size_t n_partition);       /* number of proofs produced */
size_t rofl_rangeproof_size(size_t n_bits, size_t d, size_t n_partition); /* bytes per proof */
size_t rofl_nonces_per_chunk(size_t n_bits, size_t m);

/* ---- range_proof_vec (range_proof_vec/mod.rs) ---- */
/* create_rangeproof(&Vec<f32>, &Vec<Scalar>, prove_range, n_partition) :16-102 */
int rofl_create_rangeproof(const float *values, size_t d, const uint8_t *blindings32, size_t d_blindings,
                           size_t prove_range, size_t n_partition, unsigned fp_bits, unsigned fp_frac,
                           const rofl_nonce_t *nonce, uint8_t *proofs_out, size_t *proof_len_out,
                           size_t *n_proofs_out, uint8_t *commits_out /* d*32 */);
/* ONE client split by chunks (a single client's update over several GPUs).  The reference proves the chunks of a client
 * independently of each other -- par_iter over the chunks, a transcript and a generator set per chunk (range_proof_vec/mod.rs:54-78,
 * create_rangeproof_helper :118-142) -- so a rank or a device can take any run [chunk_first, chunk_first + chunk_count) of them; there is no
 * collective inside.  `values` / `blindings32` are the client's WHOLE vectors (d fixes the chunk length m = next_pow2(d) / chunks); the range
 * check of :27-29 covers the elements the run reads.  proofs_out receives chunk_count proofs, commits_out the commitments of the run's own
 * elements [chunk_first * m, min(d, (chunk_first + chunk_count) * m)) -- *n_commits_out of them, possibly 0 for a run of padding chunks.
 * The nonce index space stays the client's (chunk c draws from c * m * (2n + 4)): the runs' outputs, concatenated in chunk order, are
 * byte for byte what rofl_create_rangeproof returns.
 * In ONE process the split needs no extra call: with rofl_set_option("devices", mask) naming several devices, rofl_create_rangeproof and
 * rofl_verify_rangeproof deal the client's chunks to them in contiguous runs (one internal thread per device; same bytes, same verdict). */
int rofl_create_rangeproof_chunks(const float *values, size_t d, const uint8_t *blindings32, size_t d_blindings,
                                  size_t prove_range, size_t n_partition, unsigned fp_bits, unsigned fp_frac,
                                  const rofl_nonce_t *nonce, size_t chunk_first, size_t chunk_count,
                                  uint8_t *proofs_out, size_t *proof_len_out, uint8_t *commits_out, size_t *n_commits_out);
/* The verdict of one run of a client's proofs (verify_rangeproof_helper per chunk, :178-181, 193-216); the AND over the runs is
 * rofl_verify_rangeproof's bit.  n_proofs and d are the client's (n_proofs must cover the padded vector exactly); `proofs` holds the run's
 * chunk_count proofs, commits32 the run's own commitments (as rofl_create_rangeproof_chunks returned them; not read when the run has none). */
int rofl_verify_rangeproof_chunks(const uint8_t *proofs, size_t proof_len, size_t n_proofs, size_t chunk_first, size_t chunk_count,
                                  const uint8_t *commits32, size_t d, size_t prove_range, unsigned fp_bits, unsigned fp_frac,
                                  const uint8_t verifier_seed[32], int *ok_out);
/* Client-side batch: n_clients independent updates of one shape (d, prove_range, n_partition) proved as ONE launch sequence -- the
 * counterpart of rofl_verify_rangeproof_batch for hosts that run many clients per process (rofl_service's client binary hosts its
 * clients as tasks of one process, client.rs:265-266; the server hands one client per pool thread, server.rs:513-521, 656-687).
 * values[i] / blindings32[i]: d floats / d scalars of client i (host or device memory); nonces[i]: its prover randomness;
 * proofs_out[i] (n_proofs * proof_len bytes) and commits_out[i] (d * 32 bytes): its results; rc_out[i]: its own outcome
 * (0, 2 ValueOutOfRangeError, 10 non-finite, 12 nonce stream too short) -- a client that fails is left out, the others are proved.
 * The return value is non-zero only for errors of the whole call.  Each client's proof is bit-identical to what
 * rofl_create_rangeproof returns for it.  With rofl_set_option("devices", mask) the clients are dealt round-robin to the listed devices
 * and proved there side by side (one internal thread per device); results arrive in the caller's arrays as before. */
int rofl_create_rangeproof_batch(size_t n_clients, const float *const *values, size_t d, const uint8_t *const *blindings32,
                                 size_t prove_range, size_t n_partition, unsigned fp_bits, unsigned fp_frac,
                                 const rofl_nonce_t *nonces /* [n_clients] */, uint8_t *const *proofs_out, size_t *proof_len_out,
                                 size_t *n_proofs_out, uint8_t *const *commits_out, int *rc_out /* [n_clients] */);
/* verify_rangeproof(&Vec<RangeProof>, &Vec<RistrettoPoint>, prove_range) :149-191.
 * verifier_seed[32] derives the batching scalar c that upstream draws from thread_rng. */
int rofl_verify_rangeproof(const uint8_t *proofs, size_t proof_len, size_t n_proofs,
                           const uint8_t *commits32, size_t d, size_t prove_range, unsigned fp_bits,
                           unsigned fp_frac, const uint8_t verifier_seed[32], int *ok_out);
/* server-side batch (server.rs:656-687 hands one client per pool thread): n_clients independent
 * (proofs, commits) sets with identical (d, prove_range, n_proofs); ok_out[n_clients] = each client's own verdict (a malformed member
 * fails alone).  rofl_set_option("verify_batch", 2) checks the whole batch with ONE random-weighted equation -- one generator MSM per
 * batch instead of one per client -- and looks closer only when that fails, so the verdicts are the same as with per-client checks;
 * rofl_set_option("devices", mask) deals the clients round-robin to the listed devices (one internal thread per device, verdicts gathered
 * in ok_out: no collective is needed inside one process). */
int rofl_verify_rangeproof_batch(size_t n_clients, const uint8_t *const *proofs, size_t proof_len,
                                 size_t n_proofs, const uint8_t *const *commits32, size_t d,
                                 size_t prove_range, unsigned fp_bits, unsigned fp_frac,
                                 const uint8_t verifier_seed[32], int *ok_out);
/* The same with the commitments of client i read every `commit_stride` >= 32 bytes from commits32[i]: the L components of ElGamal pairs
 * (stride 64) or SquareRandProofCommitments (stride 96) exactly as they arrive on the wire -- what the server's verify does with
 * `enc_values.iter().map(|x| x.c.L)` (rofl_service/src/flserver/params.rs:197, 215) without a packing pass on the host. */
int rofl_verify_rangeproof_batch_strided(size_t n_clients, const uint8_t *const *proofs, size_t proof_len,
                                         size_t n_proofs, const uint8_t *const *commits32, size_t commit_stride, size_t d,
                                         size_t prove_range, unsigned fp_bits, unsigned fp_frac,
                                         const uint8_t verifier_seed[32], int *ok_out);
/* clip_f32_to_range_vec :104-111 */
int rofl_clip_f32(const float *in, size_t d, size_t prove_range, unsigned fp_bits, unsigned fp_frac, float *out);

/* ---- l2_range_proof_vec (l2_range_proof_vec/mod.rs) ---- */
/* create_rangeproof_l2 :15-140 */
int rofl_create_rangeproof_l2(const float *values, size_t d, const uint8_t *blindings32, size_t d_blindings,
                              size_t prove_range, size_t n_partition, unsigned fp_bits, unsigned fp_frac,
                              const rofl_nonce_t *nonce, uint8_t *proof_out, size_t *proof_len_out,
                              uint8_t commit_out[32]);
/* verify_rangeproof_l2 :185-253 */
int rofl_verify_rangeproof_l2(const uint8_t *proof, size_t proof_len, const uint8_t commit[32],
                              size_t prove_range, unsigned fp_bits, unsigned fp_frac,
                              const uint8_t verifier_seed[32], int *ok_out);

/* server side (params.rs:220-231 for every client of a round, server.rs:656-687): n_clients one-value L2 sum proofs of one length, commitment i
 * = sum of client i's c_sq (commits32: n_clients * 32 bytes).  ok_out[i] = client i's verdict (a malformed member fails alone);
 * rofl_set_option("verify_batch", 2) checks them with one random-weighted equation and looks closer only when that fails. */
int rofl_verify_rangeproof_l2_batch(size_t n_clients, const uint8_t *const *proofs, size_t proof_len, const uint8_t *commits32,
                                    size_t prove_range, unsigned fp_bits, unsigned fp_frac, const uint8_t verifier_seed[32], int *ok_out);

/* ---- per-element Sigma-proofs ----
 * rand_proof_vec/mod.rs:14-118 (create_randproof_vec, create_randproof_vec_existing, verify_randproof_vec):
 *   ElGamal pair (L = m B + r Bb, R = r B) + proof of knowledge; proof 128 B = C'.L|C'.R|Z_m|Z_r, commitment 64 B = L|R.
 * square_rand_proof_vec/mod.rs:18-159 (create_l2rangeproof_vec(_existing), verify_l2rangeproof_vec):
 *   adds c_sq = m^2 B + r2 Bb; proof 192 B = C'.L|C'.R|c_sq'|Z_m|Z_r1|Z_r2, commitments 96 B = L|R|c_sq.
 * `existing32` (may be NULL) are value commitments to complete (prove_existing: L = m_com).
 * Nonce draw order per element i: m', r' (index 2i..) resp. m', r1', r2' (index 3i..), rofl_nonce_t as above. */
/* One vector over several devices / ranks: the elements of a vector are independent of each other (one proof per element on the
 * reference's rayon pool, rand_proof_vec/mod.rs:45-58, square_rand_proof_vec/mod.rs:45-58).  In ONE process, with rofl_set_option("devices", mask)
 * naming several devices, the create_*_vec / verify_*_vec calls below deal contiguous runs of elements to them (same bytes, same verdict).  A rank of
 * a one-process-per-GPU host proves its run [elem_first, elem_first + elem_count) with rofl_create_sigmaproof_vec_range -- kind 0 RandProof, 1
 * SquareRandProof, 2 SquareProof; the arrays are the WHOLE vector's (d elements; r2_32 NULL for kind 0, existing32 may be NULL), the outputs receive the
 * run's elem_count proofs and commitments; element i keeps its place in the vector's nonce index space, so the runs concatenate to the bytes of the
 * unsplit call -- and verifies a run with the verify_*_vec call on the run's own sub-arrays (the vector's verdict is the AND over the runs). */
int rofl_create_sigmaproof_vec_range(int kind, const float *values, size_t d, const uint8_t *r1_32, const uint8_t *r2_32, const uint8_t *existing32,
                                     unsigned fp_bits, unsigned fp_frac, const rofl_nonce_t *nonce, size_t elem_first, size_t elem_count,
                                     uint8_t *proofs_out, uint8_t *commits_out);
int rofl_create_randproof_vec(const float *values, size_t d, const uint8_t *r32, size_t d_r, const uint8_t *existing32,
                              unsigned fp_bits, unsigned fp_frac, const rofl_nonce_t *nonce, uint8_t *proofs_out /* d*128 */,
                              uint8_t *commits_out /* d*64 */);
int rofl_verify_randproof_vec(const uint8_t *proofs, const uint8_t *commits, size_t d, int *ok_out);
int rofl_create_squarerandproof_vec(const float *values, size_t d, const uint8_t *r1_32, size_t d_r1, const uint8_t *r2_32,
                                    const uint8_t *existing32, unsigned fp_bits, unsigned fp_frac, const rofl_nonce_t *nonce,
                                    uint8_t *proofs_out /* d*192 */, uint8_t *commits_out /* d*96 */);
int rofl_verify_squarerandproof_vec(const uint8_t *proofs, const uint8_t *commits, size_t d, int *ok_out);
/* server side: the vectors of n_clients clients (d elements each) in ONE launch sequence -- 55 000 elements alone leave most of the chip idle,
 * a round of clients fills it.  Every client is its own random linear combination (one problem of a multi-problem Pippenger launch), so
 * ok_out[i] is client i's verdict exactly as the per-client call gives it; a member with a non-canonical scalar or an undecodable point
 * gets ok = 0 and the others are still verified.  csq_sum_out32 (may be NULL; n_clients * 32 bytes): sum_i c_sq_i of every client,
 * compressed -- the commitment of the client's L2 sum proof (params.rs:220, 277), a by-product of decoding (zero bytes = the identity for
 * a malformed member).  With rofl_set_option("devices", mask) the clients are dealt round-robin to the listed devices. */
int rofl_verify_randproof_vec_batch(size_t n_clients, const uint8_t *const *proofs, const uint8_t *const *commits, size_t d, int *ok_out);
int rofl_verify_squarerandproof_vec_batch(size_t n_clients, const uint8_t *const *proofs, const uint8_t *const *commits, size_t d,
                                          int *ok_out, uint8_t *csq_sum_out32);
/* square_proof_vec/mod.rs:18-159 (create_l2rangeproof_vec(_existing), verify_l2rangeproof_vec over Pedersen commitments only):
 *   commitments 64 B = c_l|c_sq, proof 160 B = c_l'|c_sq'|Z_m|Z_r1|Z_r2; nonces m', r1', r2' at 3i.. */
int rofl_create_squareproof_vec(const float *values, size_t d, const uint8_t *r1_32, size_t d_r1, const uint8_t *r2_32,
                                const uint8_t *existing32, unsigned fp_bits, unsigned fp_frac, const rofl_nonce_t *nonce,
                                uint8_t *proofs_out /* d*160 */, uint8_t *commits_out /* d*64 */);
int rofl_verify_squareproof_vec(const uint8_t *proofs, const uint8_t *commits, size_t d, int *ok_out);
int rofl_verify_squareproof_vec_batch(size_t n_clients, const uint8_t *const *proofs, const uint8_t *const *commits, size_t d,
                                      int *ok_out, uint8_t *csq_sum_out32);
/* compressed_rand_proof/mod.rs:43-102, 134-160 (helper_prove, helper_prove_existing, helper_verify): ONE 128-byte proof
 * C'.L|C'.R|Z_m|Z_r for all d ElGamal pairs (d*64 B), z = nonce + sum_i x_i c^(i+1); nonces m', r' at index 0, 1.
 * d < 900 000 (size of the reference's label table UNIQUE_U8_TRIPLETS). */
int rofl_create_compressed_randproof(const float *values, size_t d, const uint8_t *r32, size_t d_r, const uint8_t *existing32,
                                     unsigned fp_bits, unsigned fp_frac, const rofl_nonce_t *nonce, uint8_t proof_out[128],
                                     uint8_t *pairs_out /* d*64 */);
int rofl_verify_compressed_randproof(const uint8_t proof[128], const uint8_t *pairs, size_t d, int *ok_out);

/* ---- pedersen_ops (pedersen_ops.rs) ---- */
int rofl_commit_vec(const uint8_t *values32, const uint8_t *blindings32 /* NULL: commit_no_blinding_vec */,
                    size_t d, uint8_t *out32);                                   /* :9-25 */
int rofl_add_points_vec(const uint8_t *a32, const uint8_t *b32, size_t d, uint8_t *out32);   /* :56-59 */
/* sum of d compressed points read every `stride` bytes (stride >= 32): `iter().map(|x| x.c_sq).sum()` of params.rs:220, 277
 * with stride 96 over SquareRandProofCommitments; d = 0 gives the identity. */
int rofl_sum_points(const uint8_t *points, size_t d, size_t stride, uint8_t out32[32]);
int rofl_shift_points(const uint8_t *a32, size_t d, const uint8_t offset32[32], uint8_t *out32); /* :103-108 */
int rofl_f32_to_scalar_vec(const float *in, size_t d, unsigned fp_bits, unsigned fp_frac, uint8_t *out32); /* conversion32.rs:11-22 */
int rofl_scalar_to_f32_vec(const uint8_t *in32, size_t d, unsigned fp_bits, unsigned fp_frac, float *out); /* conversion32.rs:24-39 */
int rofl_get_clip_bounds(size_t range, unsigned fp_bits, unsigned fp_frac, float *min_out, float *max_out); /* conversion32.rs:56-60 */
int rofl_fp_square_vec(const uint8_t *in32, size_t d, unsigned fp_bits, unsigned fp_frac, uint8_t *out32);      /* conversion32.rs:66-88 square; 8 = overflow (the reference panics) */
int rofl_scalar_powers(const uint8_t value32[32], size_t count, uint8_t *out32);                              /* conversion32.rs:101-122 precompute_exponentiate / exponentiate */
int rofl_scalar_add_vec(const uint8_t *a32, const uint8_t *b32, size_t d, int subtract, uint8_t *out32);      /* pedersen_ops.rs:78-94 add_scalar_vec(_vec) */
int rofl_f32_to_fp_vec(const float *in, size_t d, unsigned fp_bits, unsigned fp_frac, uint64_t *out);         /* conversion32.rs:49-54 */
int rofl_uint_to_f32_vec(const uint64_t *in, size_t d, unsigned fp_bits, unsigned fp_frac, float *out);        /* conversion32.rs:41-47 */
int rofl_get_l2_clip_bounds(size_t range, unsigned fp_bits, unsigned fp_frac, float *out);  /* conversion32.rs:62-64 */

/* ---- server-side extraction of the aggregate (bsgs32.rs:14-73, pedersen_ops.rs:27-53 discrete_log_vec_table) ----
 * BSGSTable::new(table_size) is built once per (table_size) on the device and cached; every point goes through
 * solve_discrete_log_with_neg with max_it = 2^bsgs_bits / table_size giant steps; values wrap to bsgs_bits bits like
 * BSGS_URawFix.  bsgs_bits = 8 (fp8) or 16 (fp16/fp32/fp64 builds, fp.rs:42-108).  A point whose log is not found in
 * either direction returns 11 (the reference unwraps None). */
int rofl_discrete_log_vec(const uint8_t *points32, size_t d, size_t table_size, unsigned bsgs_bits, uint8_t *scalars_out32);

/* ---- wire formats of the encrypted update containers ----
 * proto3 messages of rofl_service/proto/roflservice/flservice.proto:75-100, length-delimited as written by
 * EncParamsRange::serialize (params.rs:513-527; EncParamsRangeCompressed :745-759 uses the same message with the 128-byte
 * CompressedRandProof in rand_proof), EncParamsL2::serialize (:648-663) and EncParamsL2Compressed::serialize (:840-859).
 * Payload fields are the to_bytes concatenations the entry points above produce / consume (ElGamalPair 64 B,
 * SquareRandProofCommitments 96 B, RandProof 128 B, SquareRandProof 192 B, SquareProof 160 B, RangeProof per chunk).
 * Fields a message kind does not have are ignored on encode and left empty on decode. */
enum { ROFL_WIRE_ENC_RANGE = 0, ROFL_WIRE_ENC_NORM = 1, ROFL_WIRE_ENC_NORM_COMPRESSED = 2 };
typedef struct {
    int kind;
    const uint8_t *enc_values;          size_t enc_values_len;
    const uint8_t *rand_proof;          size_t rand_proof_len;
    const uint8_t *square_proof;        size_t square_proof_len;
    const uint8_t *range_proofs;        size_t range_proof_len, n_range_proofs;   /* [n][len] contiguous */
    const uint8_t *square_range_proof;  size_t square_range_proof_len;
    int32_t range_bits, l2_range_bits;
    float check_percentage;
} rofl_wire_msg_t;
size_t rofl_wire_encoded_size(const rofl_wire_msg_t *m);
int rofl_wire_encode(const rofl_wire_msg_t *m, uint8_t *out, size_t cap, size_t *len_out);
/* Decode: spans point into `data`; the repeated range_proof entries are gathered into range_proofs_out (capacity in
 * bytes; pass NULL first to learn n_range_proofs / range_proof_len).  Returns 5 (FormatError) on malformed input. */
int rofl_wire_decode(int kind, const uint8_t *data, size_t len, rofl_wire_msg_t *m, uint8_t *range_proofs_out, size_t range_proofs_cap);

/* ---- multi-process exchange: one process per GPU, RCCL over xGMI ----
 * The proof path has no collective inside it (clients and chunks are independent, server.rs:656-687); what a round exchanges is its RESULTS:
 * every rank's [verdict | proof bytes | commitments] to every rank (the server's collection of the client updates, server.rs:379-384), and
 * a MIN over the verdicts (server.rs:474-484: one failing client fails the round).  These entry points give a host that is not Python that
 * exchange, on the same HIP runtime as the proofs: librccl (ROFL_RCCL_LIB, default "librccl.so.1") is loaded with dlopen on first use, the
 * library has no link-time dependency on it, and every call fails with 99 when it cannot be loaded.  One communicator per process, bound
 * to the device of the thread that calls rofl_comm_init.  Payloads are host memory (that is where the ABI returns proofs and commitments);
 * they are staged through pinned buffers of the communicator and all-gathered device to device.
 *   rofl_comm_unique_id : rank 0 draws the 128-byte id (ncclGetUniqueId) and hands it to the other ranks out of band (a file, a socket, the
 *                         launcher's store)
 *   rofl_comm_init      : ncclCommInitRank; collective over all `world` ranks
 *   rofl_comm_allgather : all_out[r * n .. (r + 1) * n) = rank r's `local`; n equal on every rank
 *   rofl_comm_allreduce_f64 : op 0 sum, 1 min, 2 max over `count` <= 4096 doubles, in place (verdict bits, timings, counts)
 *   rofl_comm_barrier   : every rank has arrived
 *   rofl_comm_info      : rank / world of the communicator (-1 / 0 without one), the RCCL version and the path of the loaded library */
int rofl_comm_unique_id(uint8_t id_out[128]);
int rofl_comm_init(const uint8_t id[128], int rank, int world);
int rofl_comm_allgather(const uint8_t *local, size_t n, uint8_t *all_out /* world * n */);
int rofl_comm_allreduce_f64(double *inout, size_t count, int op);
int rofl_comm_barrier(void);
int rofl_comm_info(int *rank_out, int *world_out, int *rccl_version_out, char *lib_path_out, size_t len);
int rofl_comm_destroy(void);

/* Memory spaces.  The per-element INPUT arrays of rofl_create_rangeproof (values, blindings), rofl_verify_rangeproof(_batch)
 * (proofs, commitments) and of the per-element Sigma-proof entry points (values, randomness, existing commitments, proofs,
 * commitments) may live in host memory or in device memory of the library's device (HIP unified addressing): a caller that
 * already holds the update on the GPU passes device pointers and nothing crosses PCIe on the way in.  Outputs are written to
 * host memory. */

/* ---- behaviour options ----
 * Switches that change WHAT a call returns or how it waits are part of the ABI, not of the process environment.  `key` is one of the
 * names below; the environment variable of the same name in upper case with the ROFL_ prefix (ROFL_VERIFY_ZIP_TRUNCATE, ...) only
 * provides the default that is read once, when the first option is touched.  Options are process-wide (a server that drives several
 * devices sets them once) and may be changed between calls (not while calls are in flight).  Returns 11 (bad parameter) for an unknown
 * key or an out-of-range value.
 *   "verify_zip_truncate"  0 (default): a proof set that does not cover every chunk of the padded commitment vector does not verify
 *                          (ok = 0); 1: the reference's behaviour, zip-truncation (range_proof_vec/mod.rs:169-176), bit for bit
 *   "verify_batch"         1 (default): one random-weighted check per client (a client's chunks share one MSM); 0: one check per proof,
 *                          as upstream verify_multiple does; 2: rofl_verify_rangeproof_batch checks all of its clients with one equation
 *                          and, when that fails, groups of ~sqrt(n) clients and then the clients of the failing groups -- per-client
 *                          verdicts as with 1 (the reference's server rejects the round on any failure, server.rs:474-484, so the common
 *                          case is one generator MSM per batch)
 *   "devices"              bit mask of logical devices (bit d = device d); 0 (default): batch calls run on the calling thread's device;
 *                          otherwise rofl_create_rangeproof_batch / rofl_verify_rangeproof_batch deal their clients round-robin to the
 *                          listed devices, and the single-client calls rofl_create_rangeproof / rofl_verify_rangeproof deal the client's
 *                          CHUNKS to them in contiguous runs (same bytes, same verdict), as do the per-element Sigma-proof vector calls
 *                          with runs of ELEMENTS
 *   "sigma_batch"          1 (default): the per-element Sigma-proofs of a vector are verified as one random linear combination; 0: one
 *                          check per element (rand_proof_vec/mod.rs:93-118)
 *   "default_device"       the device of threads that never called rofl_set_device (default: the first device that was set, else 0)
 *   "blocking_sync"        -1 (default): spin while at most three calls are in flight, sleep between polls beyond that; 0: always spin;
 *                          1: always sleep (one host core per waiting call is not burned; ~50 us more latency per wait)
 * rofl_get_option also answers the read-only key "lanes": the calls that can be in flight on the calling thread's device (ROFL_LANES).
 * The remaining ROFL_* environment variables are tuning knobs that never change results (KNOBS.md). */
int rofl_set_option(const char *key, long value);
int rofl_get_option(const char *key, long *value_out);

#ifdef __cplusplus
}
#endif
#endif
