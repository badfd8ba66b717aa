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

/* GPU multi-scalar multiplication sum_i k_i * P_i through the production Pippenger pipeline (dalek
 * vartime_multiscalar_mul as used by upstream verify_multiple); test hook for skewed / extreme scalars. */
int rofl_dbg_msm(const uint8_t *scalars32, const uint8_t *points32, size_t n, uint8_t out32[32]);
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
/* host share of the hops of the calling thread's last create / verify: out[0] hops, [1] enqueue ms, [2] wait ms, [3] window-combination wall ms,
 * [4] sum of each hop's slowest pool task ms, [5..8] the maxima over the hops of enqueue, wait, combination wall, slowest task */
int rofl_dbg_last_hops(double out[10]);

#ifdef __cplusplus
}
#endif
#endif
