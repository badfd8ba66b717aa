/*
 * Test, self-check and measurement hooks of librofl_zk.so.  NOT part of the operator API (include/rofl_zk.h): nothing a
 * rofl_crypto / rofl_service binding needs lives here.  tests/ and bench.py use them; a deployment can ignore this header.
 */
#ifndef ROFL_ZK_DEBUG_H
#define ROFL_ZK_DEBUG_H
#include "rofl_zk.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- measurement hooks (bench.py) ---- */
/* Time of the kernels of the last create / verify call, from HIP events on the library's stream. */
typedef struct {
    double total_ms;          /* first launch -> last completion, device clock */
    double msm_accumulate_ms; /* sum over launches of k_msm_accumulate */
    uint64_t msm_accumulate_launches;
    uint64_t msm_terms;       /* non-trivial (scalar, point) terms fed to Pippenger */
    double fold_ms;           /* sum over launches of k_fold_gens */
    uint64_t fold_launches;
    uint64_t fold_point_reads;   /* niels points read by k_fold_gens (96 B each) */
    double host_ms;           /* host-side (transcript, Horner, fixed-base) time */
    uint64_t msm_additions;   /* mixed point additions executed by k_msm_accumulate: terms x windows (7 field multiplications each) */
} rofl_timing_t;
int rofl_last_timing(rofl_timing_t *out);
/* Per-kernel table of the last instrumented call of the calling thread: HIP-event time, launches and the ALGORITHMIC work of
 * those launches -- field multiplications (7 per mixed point addition, 8 per doubling, 9 per extended addition) and the bytes
 * a launch has to move at least once (32 B per scalar or point, 4 B per bucket-list entry). */
enum { ROFL_TK_MSM_ACCUMULATE_FB = 0, ROFL_TK_MSM_ACCUMULATE_GEN = 1, ROFL_TK_MSM_SCATTER = 2, ROFL_TK_MSM_REDUCE = 3,
       ROFL_TK_MSM_SMALL = 4, ROFL_TK_FOLD_TAB = 5, ROFL_TK_FOLD = 6, ROFL_TK_OTHER = 7,
       ROFL_TK_SIGMA = 8,            /* k_sigma_prove / k_sigma_vprep / k_sigma_verify: the per-element Sigma-proofs (the L2 composite of cfg 3 / 5) */
       ROFL_TK_VERIFY_SCALARS = 9,   /* k_verify_scalars: the verifier's 2N generator scalars, summed over the proofs of a group (scalar field: no fe_muls) */
       ROFL_TK_CODEC = 10,           /* k_decode / k_commit: Ristretto decoding / commitment + encoding of d points */
       ROFL_TK_COUNT = 11 };
typedef struct { double ms; uint64_t launches, fe_muls, bytes; } rofl_kernel_time_t;
int rofl_last_kernel_times(rofl_kernel_time_t out[ROFL_TK_COUNT]);
/* 0 = off; 1 = HIP events around every instrumented launch (~150 event records per proof: ~0.7 ms of a 25 ms proof); 2 = only around the
 * fixed-base accumulation, the kernel bench.py prices against the roofline (ten records per proof) */
int rofl_set_timing(int enabled);
/* ---- devices on a one-GPU test box ----
 * rofl_dbg_map_device: logical device `logical` (what rofl_set_device and the "devices" option name) is HIP device `physical`; allowed
 * until the logical device is first used (ROFL_DEVICE_MAP="0,0,..." does the same from the environment).  Two logical devices on GPU 0
 * are two full device contexts -- own streams, workspaces and generator tables -- which is how the multi-device paths are tested here.
 * rofl_dbg_bind_device: the thread-binding half of rofl_set_device without touching HIP (-1 = unbind). */
int rofl_dbg_map_device(int logical, int physical);
int rofl_dbg_bind_device(int device);

/* GPU multi-scalar multiplication sum_i k_i * P_i through the production Pippenger pipeline (dalek
 * vartime_multiscalar_mul as used by upstream verify_multiple); test hook for skewed / extreme scalars. */
int rofl_dbg_msm(const uint8_t *scalars32, const uint8_t *points32, size_t n, uint8_t out32[32]);
/* RangeProof::verify_multiple(&BulletproofGens::new(gens_capacity, m), &PedersenGens::default(), &mut Transcript::new(label), commits, n_bits)
 * on one aggregated proof, called the way upstream's own tests call it: any transcript label, the m (a power of two) commitments as they
 * are -- no shift by 2^(n-1) B, no padding.  The operator API fixes the labels ("RangeProof", "L2RangeProof") and the shift as the reference
 * does; upstream's serialized-proof test vectors use another label (b"Deserialize-And-Verify Test", 64 x 8 generators), so this is the entry
 * through which a byte vector of bulletproofs 4.0.0 would reach the HIP verifier.  5 = malformed proof or commitment, 6 = gens_capacity < n_bits. */
int rofl_dbg_verify_labelled(const uint8_t *label, size_t label_len, size_t gens_capacity, const uint8_t *proof, size_t proof_len,
                             const uint8_t *commits32, size_t m, size_t n_bits, const uint8_t verifier_seed[32], int *ok_out);
/* process-wide counters of the MSM driver: out[0] MSMs that finished on their first attempt's variant, out[1] repeats after a bucket-list
 * overflow of the fused small launch, out[2] repeats on the slot path after a coarse bin of the two-level sort overflowed (scalars built to
 * collide), out[3] repeats after the slot path's overflow list ran out.  The server-path tests use them to show which path a scenario took. */
int rofl_dbg_msm_retries(uint64_t out[4]);
/* field-multiply micro-benchmark: returns GF(2^255-19) multiplications per second on the device */
int rofl_bench_femul(unsigned iters, double *fe_mul_per_sec_out);
/* self-test of the quad-parallel point arithmetic (csrc/quad26.hpp): pair i = (P, Q) -> 2^doublings P + Q, one thread per pair and one quad of lanes per pair */
int rofl_dbg_quad_ops(const uint8_t *pairs64, size_t pairs, unsigned doublings, uint8_t *out_serial32, uint8_t *out_quad32);

/* ---- host-side self-test hooks (same source as the device math, compiled for the CPU) ---- */
int rofl_dbg_host_pool_stress(unsigned threads, unsigned jobs);   /* host thread pool: every index of every job runs exactly once */
int rofl_dbg_host_fe_mul(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]);
int rofl_dbg_host_fe_ops(const uint8_t a[32], const uint8_t b[32], uint8_t out_add[32], uint8_t out_sub[32], uint8_t out_sq[32], uint8_t out_inv[32]);
int rofl_dbg_host_sc_invert(const uint8_t a[32], uint8_t out_ref[32], uint8_t out_fast[32], double *ns_ref, double *ns_fast);
int rofl_dbg_host_sc_mul(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]);
int rofl_dbg_host_sc_wide(const uint8_t in[64], uint8_t out[32]);
/* the same through the Montgomery-form reduction the nonce kernel uses ((lo R^2 + hi R^3) R^-1, then out of Montgomery form) */
int rofl_dbg_host_sc_wide_mont(const uint8_t in[64], uint8_t out[32]);
/* sum of `count` <= 16 products a_k * b_k (operands < l) through the lazily reduced accumulator of k_verify_scalars2 (sc_mac_wide + ONE
 * sc_redc_wide) and as a sum of Montgomery products: both are sum a_k b_k / 2^256 mod l, canonical */
int rofl_dbg_host_sc_lazy(const uint8_t *a32, const uint8_t *b32, size_t count, uint8_t out_lazy[32], uint8_t out_ref[32]);
int rofl_dbg_host_from_uniform(const uint8_t in[64], uint8_t out[32]);
int rofl_dbg_host_scalarmult_base(const uint8_t k[32], int use_blinding_base, uint8_t out[32]);
int rofl_dbg_host_fd_codec(const uint8_t in[32], uint8_t out[32]);
int rofl_dbg_host_decode_encode(const uint8_t in[32], uint8_t out[32]);   /* returns 5 if invalid */
/* radix-2^25.5 kernel arithmetic (fe26.hpp) compiled for the host with bound assertions enabled */
int rofl_dbg_host_fd_ops(const uint8_t a[32], const uint8_t b[32], uint8_t out_mul[32], uint8_t out_sq[32], uint8_t out_add[32], uint8_t out_sub[32], uint8_t out_inv[32]);
int rofl_dbg_host_fd_scalarmult(const uint8_t k[32], const uint8_t p[32], uint8_t out[32]);
int rofl_dbg_host_merlin(const uint8_t *label, size_t label_len, const uint8_t *msg, size_t msg_len, uint8_t out[64]);
int rofl_dbg_host_nonce(const uint8_t seed[32], uint64_t idx, uint8_t out[32]);

/* host micro-benchmarks of the code the per-round hops run (nanoseconds per operation on the calling core).
 * what: 0 Keccak-f[1600]; 1 point doubling, 2 point addition, 3 Ristretto encoding (51-bit host arithmetic);
 *       4 fixed-base scalar multiplication; 5 scalar inversion; 6 Merlin transcript prefix of `iters` commitments (ns per commitment);
 *       7 / 8 / 9 one pool hand-off of a hop: 16 tasks of 30 us after a 300 us / 30 us wait of the caller, 128 tasks of 8 us after 300 us */
int rofl_dbg_host_bench(int what, unsigned iters, double *ns_out);
/* csrc/host51x8.hpp (eight window chains per AVX-512 IFMA stream) against the scalar chain of host51.hpp on 8 x W pseudo-random points,
 * windows c bits apart: 0 = all `lanes` results agree, 1 = mismatch, -1 = the CPU has no AVX-512 IFMA (nothing tested) */
int rofl_dbg_host_horner8_selftest(unsigned W, unsigned c, int lanes, double *us_simd, double *us_scalar);
/* h8::encode8 (eight Ristretto encodings per AVX-512 IFMA stream) against the scalar host encoder: 0 = all equal, 1 = mismatch, -1 = no IFMA */
int rofl_dbg_host_encode8_selftest(unsigned batches, double *us_simd, double *us_scalar);
/* csrc/keccak_x8.hpp (eight Merlin transcripts per AVX-512 stream: the verifier's transcript prefixes) against the scalar transcript:
 * `lanes` transcripts, `count` commitments each, `skew` extra prefix bytes (moves the records across the rate block): 0 = states and the
 * next challenge agree, 1 = mismatch, -1 = no AVX-512 on this CPU */
int rofl_dbg_host_merlin8_selftest(int lanes, unsigned count, unsigned skew, double *us_simd, double *us_scalar);
/* Merlin::append32_run (the run of m commitment appends of a chunk, records assembled in registers and split at the end of the rate block)
 * against `count` plain appends after `skew` (<= 400) extra prefix bytes: 0 = state, positions and next challenge equal, 1 = mismatch */
int rofl_dbg_host_merlin_run_selftest(unsigned count, unsigned skew);
/* csrc/keccak.hpp keccak_f1600_zmm (ONE Keccak-f[1600] state across five AVX-512 registers: what every host transcript runs where the CPU has
 * it) against the scalar permutation: `states` pseudo-random states permuted `chain` times each, and the known answer of the zero state.
 * 0 = all equal, 1 = mismatch, -1 = no AVX-512 on this CPU; ns per permutation of both on request */
int rofl_dbg_host_keccak_zmm_selftest(unsigned states, unsigned chain, double *ns_zmm, double *ns_scalar);
/* host share of the hops of the calling thread's last create / verify: out[0] hops, [1] enqueue ms, [2] wait ms, [3] window-combination wall ms,
 * [4] sum of each hop's slowest pool task ms, [5..8] the maxima over the hops of enqueue, wait, combination wall, slowest task */
int rofl_dbg_last_hops(double out[10]);

#ifdef __cplusplus
}
#endif
#endif
