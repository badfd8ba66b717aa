// librofl_zk.so: C ABI (include/rofl_zk.h) + host orchestration of the HIP kernels.  One translation unit, in sections:
//   host_rt.hpp (knobs, pool, buffers, lanes, generator cache) | host_msm.hpp (MSM driver) | host_prover.hpp | host_verifier.hpp |
//   this file: conversion32 helpers, create / verify entry logic, the extern "C" block.
//
// Host responsibilities: Merlin transcripts (sequential), final window/bit combination of MSM partial
// sums (a 256-step Horner chain that would serialise a single GPU lane), fixed-base multiples of B and
// B_blinding, proof (de)serialisation.  Everything proportional to d * n_bits runs in kernels.hpp.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types only: the library is loaded with dlopen when rofl_comm_* is first used (no link-time dependency)
#include <dlfcn.h>
#include <immintrin.h>
#include <chrono>
#include <time.h>
#include <sched.h>
#include <unistd.h>
#include <sys/syscall.h>
#include <linux/futex.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/rofl_zk.h"
#include "../../include/rofl_zk_debug.h"
#include "kernels.hpp"
#if ROFL_KGROUP != 0
#include "kernel_protos.hpp"      // kernels live in their own translation units (build.py)
#endif
#include "wire.hpp"
#include "host51.hpp"
#include "host51x8.hpp"
#include "keccak_x8.hpp"

using namespace rofl;

namespace {

#include "host_rt.hpp"
#include "host_msm.hpp"
#include "host_prover.hpp"
#include "host_verifier.hpp"

// ---------------------------------------------------------------- conversion32.rs helpers (host)
u64 fix_max_bits(unsigned fp_bits) { return fp_bits >= 64 ? ~0ULL : ((1ULL << fp_bits) - 1); }
int fix_from_abs_f32(float v, unsigned fp_bits, unsigned fp_frac, u64 *out) {
    if (std::isnan(v)) return ROFL_NON_FINITE;
    double x = std::fabs((double)v) * (double)(1ULL << fp_frac);
    double lim = std::ldexp(1.0, (int)fp_bits);
    if (std::isinf(x) || x >= lim) { *out = fix_max_bits(fp_bits); return 0; }
    double k = std::nearbyint(x);
    *out = (k >= lim) ? fix_max_bits(fp_bits) : (u64)k;
    return 0;
}
float fix_to_f32(u64 k, unsigned fp_frac) { volatile float f = (float)k; return f / (float)(1ULL << fp_frac); }
u64 read_from_bytes(const sc &s, unsigned fp_bits) { u64 r = (u64)s.v[0] | ((u64)s.v[1] << 32); return r & fix_max_bits(fp_bits); }
int f32_to_sc(float v, unsigned fp_bits, unsigned fp_frac, sc *out) {
    u64 k; int rc = fix_from_abs_f32(v, fp_bits, fp_frac, &k); if (rc) return rc;
    sc s = sc_from_u64(k); *out = (v < 0.0f) ? sc_neg(s) : s; return 0;
}
float sc_to_f32(const sc &s, unsigned fp_bits, unsigned fp_frac) {
    if ((s.v[7] >> 24) != 0) return -fix_to_f32(read_from_bytes(sc_neg(s), fp_bits), fp_frac);
    return fix_to_f32(read_from_bytes(s, fp_bits), fp_frac);
}
void clip_bounds(size_t range, unsigned fp_bits, unsigned fp_frac, float *mn, float *mx) {
    unsigned __int128 v = ((unsigned __int128)1 << (range - 1)) - 1;
    *mx = fix_to_f32((u64)v & fix_max_bits(fp_bits), fp_frac); *mn = -*mx;
}
float l2_clip_bound(size_t range, unsigned fp_bits, unsigned fp_frac) {
    unsigned __int128 v = ((unsigned __int128)1 << range) - 1;
    return fix_to_f32((u64)v & fix_max_bits(fp_bits), fp_frac);
}
bool valid_fp(unsigned fp_bits, unsigned fp_frac) { return (fp_bits == 8 || fp_bits == 16 || fp_bits == 32 || fp_bits == 64) && fp_frac <= 12 && fp_frac < fp_bits; }

thread_local rofl_timing_t g_last_timing{};
thread_local double g_last_hops[10] = {0};      // rofl_dbg_last_hops: the HopStats of the calling thread's last range-proof call
thread_local rofl_kernel_time_t g_last_ktimes[ROFL_TK_COUNT]{};
void timing_begin(Ctx &C) {
    C.tm.reset();
    if (C.tm.enabled) { C.tm.first = C.tm.get(); C.tm.last = C.tm.get(); HIPCHK(hipEventRecord(C.tm.first, C.stream)); }
}
void timing_end(Ctx &C) {
    { const auto &h = C.hs; double v[10] = {(double)h.n, h.enqueue, h.sync, h.horner_wall, h.horner_cpu, h.max_enqueue, h.max_sync, h.max_horner, h.max_task, 0}; memcpy(g_last_hops, v, sizeof v); C.hs = Ctx::HopStats(); }
    if (!C.tm.enabled) {
        g_last_timing = C.tm.t; memset(g_last_ktimes, 0, sizeof g_last_ktimes);
        if (C.tm.acc_only && !C.tm.kev.empty()) {      // the stream has been synchronised by the caller's last wait: the few recorded spans are complete
            float ms = 0;
            for (auto &k : C.tm.kev) {
                if (hipEventSynchronize(k.e1) != hipSuccess || hipEventElapsedTime(&ms, k.e0, k.e1) != hipSuccess) continue;
                rofl_kernel_time_t &o = g_last_ktimes[k.kind]; o.ms += ms; o.launches++; o.fe_muls += k.fe_muls; o.bytes += k.bytes;
            }
        }
        return;
    }
    HIPCHK(hipEventRecord(C.tm.last, C.stream));
    HIPCHK(hipEventSynchronize(C.tm.last));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, C.tm.first, C.tm.last)); C.tm.t.total_ms = ms;
    bool trace = knob("ROFL_TRACE") != nullptr;
    for (size_t i = 0; i < C.tm.acc_ev.size(); i++) { auto &e = C.tm.acc_ev[i]; HIPCHK(hipEventElapsedTime(&ms, e.first, e.second)); C.tm.t.msm_accumulate_ms += ms; if (trace) fprintf(stderr, "[rofl] %-40s accumulate %.3f ms\n", C.tm.acc_tag[i].c_str(), ms); }
    for (size_t i = 0; i < C.tm.fold_ev.size(); i++) { auto &e = C.tm.fold_ev[i]; HIPCHK(hipEventElapsedTime(&ms, e.first, e.second)); C.tm.t.fold_ms += ms; if (trace) fprintf(stderr, "[rofl] %-40s %.3f ms\n", C.tm.fold_tag[i].c_str(), ms); }
    for (auto &k : C.tm.kev) {
        HIPCHK(hipEventElapsedTime(&ms, k.e0, k.e1));
        rofl_kernel_time_t &o = C.tm.kt[k.kind]; o.ms += ms; o.launches++; o.fe_muls += k.fe_muls; o.bytes += k.bytes;
    }
    memcpy(g_last_ktimes, C.tm.kt, sizeof g_last_ktimes);
    g_last_timing = C.tm.t;
}


constexpr size_t kMaxBatchMembers = 65535;      // gridDim.y: what a batch entry point accepts in one call (the launches themselves are checked too, ROFL_LAUNCH)

template <class F> int guarded(F f) {
    try { return f(); }
    catch (const HipErr &e) {
        char buf[256]; snprintf(buf, sizeof buf, "HIP error %d (%s) in %s", (int)e.e, hipGetErrorString(e.e), e.what);
        return fail(ROFL_HIP_ERROR + (int)e.e, buf);
    }
    catch (const std::exception &e) { return fail(ROFL_HIP_ERROR, std::string("exception: ") + e.what()); }
}

// create_rangeproof for `nc` clients of one shape (d, prove_range, n_partition) as ONE launch sequence: the clients' chunks are laid
// side by side ([client][chunk]), every kernel of the proof covers all of them (blockIdx.y = chunk), and every IPP round carries the
// L / R problems of all clients -- the host hops, the latency-bound tail and the launch overheads are paid once per batch instead of
// once per client.  rcs[i] = the per-client outcome (ValueOutOfRange, NaN, nonce stream too short); clients that fail are left out,
// the others are proved.  Returns non-zero only for errors that concern the whole call.
int create_impl(Ctx &C, size_t nc, const float *const *values, size_t d, const uint8_t *const *blind, size_t prove_range, size_t n_partition,
                unsigned fp_bits, unsigned fp_frac, const rofl_nonce_t *nonces, uint8_t *const *proofs_out, size_t *plen_out, size_t *np_out,
                uint8_t *const *commits_out, int *rcs, bool single, size_t chunk_first = 0, size_t chunk_count = 0) {
    // `single`: the call is rofl_create_rangeproof (one client: its errors are the call's errors); batch calls -- also their one-client
    // shards when a batch is spread over several devices -- report per client in rcs.
    // [chunk_first, chunk_first + chunk_count) (count 0: all): the chunks of every client that THIS call proves -- the reference proves a
    // client's chunks independently of each other (range_proof_vec/mod.rs:54-78: par_iter over the chunks, a transcript and a generator set
    // per chunk), so a device or a rank can take any run of them.  values / blindings are still the client's whole vectors (the range check
    // of :27-29 is over the slice this call reads; the caller of a split combines the outcomes); proofs_out[i] receives chunk_count proofs,
    // commits_out[i] the commitments of the run's own elements, i.e. it points at element chunk_first * m of the client's array; the nonce
    // index space stays the client's (chunk c draws from c * m * (2n + 4)), so the bytes are those of the unsplit call.
    for (size_t i = 0; i < nc; i++) rcs[i] = ROFL_OK;
    if (!valid_fp(fp_bits, fp_frac) || d == 0 || n_partition == 0 || prove_range == 0 || prove_range > fp_bits || !nonces || nc == 0)
        return fail(ROFL_BAD_PARAM, "bad parameter (the reference panics here)");
    const size_t d_all = d, dp_all = next_pow2(d);
    size_t n_chunks = std::min(dp_all, n_partition), chunk = dp_all / n_chunks;
    const size_t P_all = (dp_all + chunk - 1) / chunk;
    if (chunk_count == 0) { chunk_first = 0; chunk_count = P_all; }
    if (chunk_first >= P_all || chunk_count > P_all - chunk_first) return fail(ROFL_BAD_PARAM, "chunk range outside the client's chunks");
    // from here on d, dp, P describe the run: `d` real elements (possibly none: a run of padding chunks) at the front of dp = P * chunk
    const size_t el0 = chunk_first * chunk;
    const size_t P = chunk_count, dp = P * chunk;
    d = el0 >= d_all ? 0 : std::min(d_all - el0, dp);
    float mn, mx; clip_bounds(prove_range, fp_bits, fp_frac, &mn, &mx);
    C.init();
    C.batch_mode = nc > 1;
    timing_begin(C);
    float *d_vals = C.vals.as<float>(nc * d + 1);
    u64 *vshift = C.vshift.as<u64>(nc * dp);
    sc *d_blind_buf = C.blind.as<sc>(nc * dp);
    HIPCHK(hipMemsetAsync(d_blind_buf, 0, sizeof(sc) * nc * dp, C.stream));
    for (size_t i = 0; i < nc && d; i++) {      // the callers' arrays: host or device memory
        C.up(d_vals + i * d, values[i] + el0, sizeof(float) * d, C.stream);
        C.up(d_blind_buf + i * dp, blind[i] + el0 * 32, 32 * d, C.stream);
    }
    u32 *status = C.status.as<u32>(nc + 4);
    HIPCHK(hipMemsetAsync(status, 0, 4 * (nc + 4), C.stream));
    ROFL_LAUNCH(k_quantize_shift, grid1(dp, (u32)nc), dim3(TPB), 0, C.stream, d_vals, (u32)d, (u32)dp, (u32)prove_range, fp_bits, fp_frac, mn, mx, vshift, status);
    u32 *h_status = C.h_misc.as<u32>(nc + 4);
    HIPCHK(hipMemcpyAsync(h_status, status, 4 * nc, hipMemcpyDeviceToHost, C.stream));
    C.sync();
    // the reference's order of checks (range_proof_vec/mod.rs:22-29, then the upstream errors)
    bool any = false;
    for (size_t i = 0; i < nc; i++) {
        if (h_status[i] & 1) rcs[i] = ROFL_VALUE_OUT_OF_RANGE;
        else if (h_status[i] & 2) rcs[i] = ROFL_NON_FINITE;
        any |= rcs[i] == ROFL_OK;
    }
    if (single && rcs[0] == ROFL_VALUE_OUT_OF_RANGE) return fail(ROFL_VALUE_OUT_OF_RANGE, "ValueOutOfRangeError");
    if (single && rcs[0] == ROFL_NON_FINITE) return fail(ROFL_NON_FINITE, "non-finite value (the reference panics in fixed::saturating_from_float)");
    if (!is_pow2(chunk) || dp % chunk) return fail(ROFL_INVALID_AGGREGATION, "InvalidAggregation (the reference panics)");
    if (!(prove_range == 8 || prove_range == 16 || prove_range == 32 || prove_range == 64)) return fail(ROFL_INVALID_BITSIZE, "InvalidBitsize");
    for (size_t i = 0; i < nc; i++)
        if (rcs[i] == ROFL_OK && nonces[i].mode == 0 && nonces[i].stream_scalars < P_all * chunk * (2 * prove_range + 4)) {
            rcs[i] = ROFL_NONCE_SHORT;
            if (single) return fail(ROFL_NONCE_SHORT, "nonce stream too short");
        }
    size_t plen = 32 * (9 + 2 * (size_t)lg2u(prove_range * chunk));
    *plen_out = plen; *np_out = P;
    if (2 * nc * P > kMaxBatchMembers) return fail(ROFL_BAD_PARAM, "batch too large (split it)");      // the L / R problems of all chunks index gridDim.y
    std::vector<size_t> act;
    for (size_t i = 0; i < nc; i++) if (rcs[i] == ROFL_OK) act.push_back(i);
    if (act.empty()) { timing_end(C); return ROFL_OK; }
    size_t na = act.size();
    if (na != nc)      // close the gaps: the proof kernels index chunks densely
        for (size_t k = 0; k < na; k++) if (act[k] != k) {
            HIPCHK(hipMemcpyAsync(vshift + k * dp, vshift + act[k] * dp, 8 * dp, hipMemcpyDeviceToDevice, C.stream));
            HIPCHK(hipMemcpyAsync(d_blind_buf + k * dp, d_blind_buf + act[k] * dp, 32 * dp, hipMemcpyDeviceToDevice, C.stream));
        }
    // explicit nonce streams go to the device once
    std::vector<ChunkNonce> cn(na * P);
    { const u64 per = (u64)chunk * (2 * prove_range + 4);      // nonces of one chunk; a run reads scalars [chunk_first * per, (chunk_first + P) * per) of the client's stream
      const size_t run_bytes = (size_t)(P * per) * 64, run_off = (size_t)(chunk_first * per) * 64;
      size_t tot = 0; for (size_t k = 0; k < na; k++) if (nonces[act[k]].mode == 0) tot += run_bytes;
      uint8_t *sb = tot ? C.stream_buf.as<uint8_t>(tot + 64) : nullptr; size_t off = 0;
      for (size_t k = 0; k < na; k++) {
          const rofl_nonce_t &nn = nonces[act[k]];
          ChunkNonce base{}; base.mode = nn.mode;
          if (nn.mode == 1) memcpy(base.seed.w, nn.seed, 32);
          else {      // only the run's part of the stream goes up; the kernel indexes the stream by the client's nonce index, so the base address is moved back by the part that stayed behind (never dereferenced there)
              C.up(sb + off, nn.stream + run_off, run_bytes, C.stream);
              base.d_stream = reinterpret_cast<const uint8_t *>(reinterpret_cast<uintptr_t>(sb + off) - run_off); base.stream_scalars = (chunk_first + P) * per; off += run_bytes; }
          for (size_t c = 0; c < P; c++) { cn[k * P + c] = base; cn[k * P + c].base = (chunk_first + c) * per; }
      } }
    // V_j and un-shifted commitments C_j = V_j - 2^(range-1) B   (range_proof_vec/mod.rs:96-99)
    sc negoff = sc_neg(sc_from_u64(1ULL << (prove_range - 1)));
    niels h_shift = h51::to_niels32(h_fixed_mul(C.ht.B5, negoff));
    niels *d_shift = C.tmp_in.as<niels>(1);
    uint8_t *Vb = C.Vbytes.as<uint8_t>(na * dp * 32), *Cb = C.Cbytes.as<uint8_t>(na * dp * 32);
    // The commitment kernel is a latency chain on d threads (two fixed-base multiplications and two encodings each: 0.4 ms on a tenth
    // of the chip) and nothing in the prover reads its output on the device: it runs on the side stream, beside the nonce expansion and the
    // A / S launches.  The host needs the V bytes when it hashes them into the transcripts (ev_v), the caller's commitment arrays when the
    // call returns.
    if (!C.stream2) HIPCHK(hipStreamCreateWithFlags(&C.stream2, hipStreamNonBlocking));
    if (!C.ev_v) { HIPCHK(hipEventCreateWithFlags(&C.ev_v, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&C.ev_fork, hipEventDisableTiming)); }
    GensPin gens_pin = get_gens(C, prove_range, chunk);           // may throw (allocation): before anything is queued on the side stream
    struct Join { hipStream_t s; ~Join() { (void)hipStreamSynchronize(s); } } join{C.stream2};      // also on the error paths: the side stream copies into caller memory
    HIPCHK(hipEventRecord(C.ev_fork, C.stream));                  // inputs quantised (and compacted)
    HIPCHK(hipStreamWaitEvent(C.stream2, C.ev_fork, 0));
    HIPCHK(hipMemcpyAsync(d_shift, &h_shift, sizeof(niels), hipMemcpyHostToDevice, C.stream2));
    { KSpan ks(C.tm, C.stream2, ROFL_TK_CODEC, (uint64_t)na * dp * ((8 + 1 + 32) * 7 + 2 * 265), (uint64_t)na * dp * (8 + 32 + 64));      // 8 + carry + 32 radix-256 windows, two encodings
    ROFL_LAUNCH(k_commit, grid1(na * dp), dim3(TPB), 0, C.stream2, (u32)(na * dp), vshift, (const sc *)nullptr, d_blind_buf, C.d_tabB8, C.d_tabBb8, d_shift, Vb, Cb, (u32)d, (u32)dp); }
    uint8_t *hV = C.h_V.as<uint8_t>(na * dp * 32);
    HIPCHK(hipMemcpyAsync(hV, Vb, na * dp * 32, hipMemcpyDeviceToHost, C.stream2));
    HIPCHK(hipEventRecord(C.ev_v, C.stream2));
    for (size_t k = 0; k < na && d; k++) C.down(commits_out[act[k]], Cb + k * dp * 32, d * 32, C.stream2);
    std::vector<uint8_t *> pout(na * P);
    for (size_t k = 0; k < na; k++) for (size_t c = 0; c < P; c++) pout[k * P + c] = proofs_out[act[k]] + c * plen;
    prove_chunks(C, "RangeProof", na * P, prove_range, chunk, vshift, d_blind_buf, cn, hV, pout.data(), C.ev_v);
    HIPCHK(hipStreamSynchronize(C.stream2));
    timing_end(C);
    return ROFL_OK;
}

// The reference zips `commits.chunks(len / proofs.len())` with the proofs (range_proof_vec/mod.rs:169-176): when the proof count does
// not divide the padded length the zip silently drops the tail, i.e. commitments that NO proof covers are accepted (3 proofs for
// 8 commitments check 6 of them; dp/2 + 1 proofs check half).  The proof count comes off the wire, so that is a soundness hole, not
// a format quirk: here a set whose proofs do not cover every chunk exactly is reported as "does not verify" (ok = 0, return code 0).
// rofl_set_option("verify_zip_truncate", 1) restores the reference's behaviour bit for bit (byte-level comparisons).
// `single`: the call is rofl_verify_rangeproof (a malformed set is the call's FormatError); batch calls -- also their one-client shards
// when a batch is spread over several devices -- give every client its own verdict.  gid[i] (nullptr: i) = client i's index in the
// caller's batch: it keys the client's random weights, so that a shard draws what the whole batch would have drawn for it.
// cstride: distance in bytes between two commitments of a client's array (32 = packed; 64 / 96 = the L component of ElGamal pairs /
// SquareRandProofCommitments as they arrive on the wire, params.rs:197, 215: `enc_values.iter().map(|x| x.c.L)`).
int verify_impl(Ctx &C, size_t n_clients, const uint8_t *const *proofs, size_t proof_len, size_t n_proofs, const uint8_t *const *commits,
                size_t d, size_t prove_range, unsigned fp_bits, unsigned fp_frac, const uint8_t seed[32], int *ok_out, bool single,
                const size_t *gid = nullptr, size_t cstride = 32, size_t chunk_first = 0, size_t chunk_count = 0) {
    // [chunk_first, chunk_first + chunk_count) (count 0: all): the proofs of every client that THIS call checks -- the reference verifies a
    // client's proofs independently of each other and ANDs the bits (range_proof_vec/mod.rs:168-181), so a device or a rank can take any run
    // of them.  n_proofs and d stay the client's (they fix the chunk length); proofs[i] points at the run's first proof, commits[i] at the
    // run's first commitment (element chunk_first * m of the client's array).
    for (size_t i = 0; i < n_clients; i++) ok_out[i] = 0;
    if (!valid_fp(fp_bits, fp_frac) || d == 0 || n_proofs == 0 || prove_range == 0 || prove_range > fp_bits || n_clients == 0 || cstride < 32)
        return fail(ROFL_BAD_PARAM, "bad parameter (the reference panics here)");
    const size_t d_all = d;
    size_t dp = next_pow2(d);
    size_t chunk = dp / n_proofs;
    if (chunk == 0) return fail(ROFL_BAD_PARAM, "more proofs than padded commitments (the reference panics in chunks(0))");
    size_t n_chunks = (dp + chunk - 1) / chunk;
    size_t nv = std::min(n_proofs, n_chunks);            // zip truncates (range_proof_vec/mod.rs:173-176)
    const bool zip_truncate = opts().zip_truncate.load() != 0;      // rofl_set_option("verify_zip_truncate")
    if (n_proofs * chunk != dp) {
        if (!zip_truncate) { g_err = "proof count does not cover the padded commitment vector: not verified"; return ROFL_OK; }
        if (dp % chunk) return fail(ROFL_BAD_PARAM, "ragged chunks are not supported");
    }
    if (chunk_count) {      // a run of the client's verified chunks: from here on d, dp, nv describe the run
        if (chunk_first >= nv || chunk_count > nv - chunk_first) return fail(ROFL_BAD_PARAM, "chunk range outside the client's proofs");
        const size_t el0 = chunk_first * chunk;
        nv = chunk_count; dp = nv * chunk;
        d = el0 >= d_all ? 0 : std::min(d_all - el0, dp);
    } else chunk_first = 0;
    // Everything below allocates per (prove_range, chunk): check the proofs' own shape against it first (RangeProof::from_bytes,
    // then the N == 2^lg test of verify_multiple), so that a forged proof count cannot make the device build tables.
    if (proof_len % 32 != 0 || proof_len < 7 * 32) return fail(ROFL_FORMAT_ERROR, "FormatError: proof length");
    size_t ne = (proof_len - 7 * 32) / 32;
    if (ne < 2 || (ne - 2) % 2 != 0 || (ne - 2) / 2 >= 32) return fail(ROFL_FORMAT_ERROR, "FormatError: proof length");
    size_t lg = (ne - 2) / 2;
    size_t P = n_clients * nv;
    // every (client, chunk) pair is a row of gridDim.y in the verifier's kernels (and, with the closer look of verify_batch = 2, a few padding rows more)
    if (P + 2 * nv * (size_t)std::ceil(std::sqrt((double)n_clients)) > kMaxBatchMembers) return fail(ROFL_BAD_PARAM, "batch too large (split it)");
    std::vector<uint8_t> pf(P * proof_len);
    for (size_t i = 0; i < n_clients; i++) {      // host or device memory; caller memory is not handed to the HIP runtime
        if (is_device_ptr(proofs[i])) HIPCHK(hipMemcpy(&pf[i * nv * proof_len], proofs[i], nv * proof_len, hipMemcpyDeviceToHost));
        else memcpy(&pf[i * nv * proof_len], proofs[i], nv * proof_len);
    }
    // RangeProof::from_bytes rejects non-canonical scalars.  A single set: FormatError, as the reference (the caller cannot even build
    // its Vec<RangeProof>).  In a batch every client has its own verdict (server.rs:656-687 verifies each client on its own): the
    // offender gets ok = 0 and the others are still verified -- its scalars are zeroed in the local copy so that the shared launch
    // sequence stays well-formed (its check then fails on its own; clients never share a check).
    std::vector<char> bad_format(n_clients, 0);
    for (size_t q = 0; q < P; q++) {
        uint8_t *pb = &pf[q * proof_len];
        const size_t offs[5] = {128, 160, 192, 7 * 32 + 64 * lg, 7 * 32 + 64 * lg + 32};
        for (size_t o : offs)
            if (!sc_is_canonical_bytes(pb + o)) {
                if (single) return fail(ROFL_FORMAT_ERROR, "proof rejected before verification (format / bitsize)");
                bad_format[q / nv] = 1; memset(pb + o, 0, 32);
            }
    }
    if (!(prove_range == 8 || prove_range == 16 || prove_range == 32 || prove_range == 64)) return fail(ROFL_INVALID_BITSIZE, "proof rejected before verification (format / bitsize)");
    if (prove_range * chunk != ((size_t)1 << lg)) return ROFL_OK;      // VerificationError for every chunk -> Ok(false)
    C.init();
    C.batch_mode = n_clients > 1;
    timing_begin(C);
    // shift up by 2^(range-1) B, pad with identity, compress (:155-167)
    niels h_shift = h51::to_niels32(h_fixed_mul(C.ht.B5, sc_from_u64(1ULL << (prove_range - 1))));
    niels *d_shift = C.tmp_out.as<niels>(1);
    HIPCHK(hipMemcpyAsync(d_shift, &h_shift, sizeof(niels), hipMemcpyHostToDevice, C.stream));
    size_t tot = n_clients * dp;
    uint8_t *d_in = C.Cbytes.as<uint8_t>(tot * 32);
    uint8_t *d_enc = C.Vbytes.as<uint8_t>(tot * 32);
    niels *d_vn = C.gbuf[0].as<niels>(tot);
    u32 *status = C.status.as<u32>(n_clients + 4);       // one status word per client: an undecodable commitment fails that client only
    HIPCHK(hipMemsetAsync(status, 0, 4 * (n_clients + 4), C.stream));
    uint8_t *hV = C.h_V.as<uint8_t>(tot * 32);            // pinned: the encodings of the shifted commitments, for the transcripts
    u32 *h_st = C.h_misc.as<u32>(n_clients + 4);
    // Commitments in, decoded, shifted encodings out -- in groups of a few clients: the host hashes the encodings of group g into the
    // transcripts (verify_chunks) while the device decodes group g + 1.  Host memory is staged (one parallel copy per group on the pool);
    // device pointers go straight in.
    const size_t GC = std::max<size_t>(1, std::min<size_t>(n_clients, ((size_t)1 << 19) / dp));      // ~2^19 commitments per group
    std::vector<VerifyReady> ready;
    bool used_up = false;
    struct JoinUp { Ctx &c; bool &used; ~JoinUp() { if (used && c.stream_up) (void)hipStreamSynchronize(c.stream_up); } } join_up{C, used_up};      // nothing of the upload stream outlives the call
    bool all_host = true; for (size_t i = 0; i < n_clients; i++) all_host &= !is_device_ptr(commits[i]);
    for (size_t i0 = 0; i0 < n_clients; i0 += GC) {
        const size_t gc = std::min(GC, n_clients - i0);
        if (d == 0) {      // a run of padding chunks: nothing to bring in, the decode kernel writes identities
        } else if (all_host && (d * 32 >= Stage::kMin || cstride != 32)) {
            uint8_t *st = (uint8_t *)C.stg.alloc(gc * d * 32);
            const size_t slices = std::max<size_t>(1, (d * 32) >> 18);      // ~256 KB per task
            C.pool->run(gc * slices, [&](size_t t) { size_t i = t / slices, k = t % slices;
                                                     if (cstride == 32) { size_t lo = d * 32 * k / slices, hi = d * 32 * (k + 1) / slices; stage_copy(st + i * d * 32 + lo, commits[i0 + i] + lo, hi - lo); }
                                                     else { const uint8_t *src = commits[i0 + i]; uint8_t *dst = st + i * d * 32;      // gather: the staging copy is also the packing
                                                            for (size_t e = d * k / slices, e1 = d * (k + 1) / slices; e < e1; e++) memcpy(dst + e * 32, src + e * cstride, 32); } });
            if (n_clients > GC) {
                // several groups: the upload of group g + 1 runs beside the decoding of group g (a stream of its own; the decode kernel waits for its group's bytes)
                if (!C.stream_up) HIPCHK(hipStreamCreateWithFlags(&C.stream_up, hipStreamNonBlocking));
                if (i0 == 0) { HIPCHK(hipEventRecord(C.pool_event(0), C.stream)); HIPCHK(hipStreamWaitEvent(C.stream_up, C.pool_event(0), 0)); }      // (after the shift upload and the status memset)
                HIPCHK(hipMemcpy2DAsync(d_in + i0 * dp * 32, dp * 32, st, d * 32, d * 32, gc, hipMemcpyHostToDevice, C.stream_up));
                HIPCHK(hipEventRecord(C.pool_event(1 + (ready.size() & 1)), C.stream_up));
                HIPCHK(hipStreamWaitEvent(C.stream, C.pool_event(1 + (ready.size() & 1)), 0));
                used_up = true;
            } else      // one group (a single client: the latency case): no second stream, no events
                HIPCHK(hipMemcpy2DAsync(d_in + i0 * dp * 32, dp * 32, st, d * 32, d * 32, gc, hipMemcpyHostToDevice, C.stream));
        } else
            for (size_t i = i0; i < i0 + gc; i++) {
                if (cstride == 32) C.up(d_in + i * dp * 32, commits[i], d * 32, C.stream);
                else if (is_device_ptr(commits[i])) HIPCHK(hipMemcpy2DAsync(d_in + i * dp * 32, 32, commits[i], cstride, 32, d, hipMemcpyDeviceToDevice, C.stream));
                else { uint8_t *st = (uint8_t *)C.stg.alloc(d * 32); for (size_t e = 0; e < d; e++) memcpy(st + e * 32, commits[i] + e * cstride, 32);
                       HIPCHK(hipMemcpyAsync(d_in + i * dp * 32, st, d * 32, hipMemcpyHostToDevice, C.stream)); }
            }
        { KSpan ks(C.tm, C.stream, ROFL_TK_CODEC, (uint64_t)gc * d * (2 * 265 + 7), (uint64_t)gc * d * 64);
          ROFL_LAUNCH(k_decode, grid1(dp, (u32)gc), dim3(TPB), 0, C.stream, (u32)dp, (u32)d, d_in + i0 * dp * 32, d_shift, d_vn + i0 * dp, d_enc + i0 * dp * 32, status + i0); }
        HIPCHK(hipMemcpyAsync(hV + i0 * dp * 32, d_enc + i0 * dp * 32, gc * dp * 32, hipMemcpyDeviceToHost, C.stream));
        HIPCHK(hipMemcpyAsync(h_st + i0, status + i0, 4 * gc, hipMemcpyDeviceToHost, C.stream));
        ready.push_back(VerifyReady{(i0 + gc) * nv, C.pool_event(3 + ready.size())});      // (events 0..2 of the call: the upload stream's)
        HIPCHK(hipEventRecord(ready.back().ev, C.stream));
    }
    // flatten (client, chunk) -> problem list.  A client's verified chunks are a prefix of its dp commitments: when the proofs cover
    // everything (the normal case) the per-client arrays ARE the per-proof arrays; otherwise one strided copy each
    std::vector<u64> cidx(P);
    for (size_t i = 0; i < n_clients; i++) for (size_t c = 0; c < nv; c++) cidx[i * nv + c] = chunk_first + c;      // the chunk's index in its client: keys its random weights
    const uint8_t *Vh = hV; const niels *d_vn2 = d_vn;
    std::vector<uint8_t> Vh_own;
    if (nv * chunk != dp) {
        C.sync();
        Vh_own.resize(P * chunk * 32);
        for (size_t i = 0; i < n_clients; i++) memcpy(&Vh_own[i * nv * chunk * 32], hV + i * dp * 32, nv * chunk * 32);
        Vh = Vh_own.data();
        niels *cp2 = C.gbuf[1].as<niels>(P * chunk);
        HIPCHK(hipMemcpy2DAsync(cp2, nv * chunk * sizeof(niels), d_vn, dp * sizeof(niels), nv * chunk * sizeof(niels), n_clients, hipMemcpyDeviceToDevice, C.stream));
        d_vn2 = cp2;
    }
    std::vector<int> okc(P);
    GensPin gens_pin = get_gens(C, prove_range, chunk, GENS_VERIFY);      // generators + window slices; a verifier never builds the prover's fold table
    const int vbatch = opts().verify_batch.load();            // rofl_set_option("verify_batch")
    size_t grp = vbatch ? nv : 1;
    // verify_batch = 2: one check for the whole batch, a closer look only when it fails (verify_chunks); clients already known to be
    // malformed are kept out of the shared checks
    const bool hier = vbatch == 2 && n_clients > 1;
    std::vector<u64> ridx(P); std::vector<char> skip(P, 0);
    for (size_t q = 0; q < P; q++) { size_t i = q / nv; ridx[q] = ((u64)(gid ? gid[i] : i) << 24) | cidx[q]; skip[q] = hier && bad_format[i]; }
    // all (client, chunk) pairs in one pass; a client's chunks form one batch of the random-weighted check
    sc v_shift = sc_from_u64(1ULL << (prove_range - 1));
    std::vector<u64> v_real(P);
    for (size_t q = 0; q < P; q++) { size_t lo = cidx[q] * chunk; v_real[q] = lo >= d_all ? 0 : std::min(chunk, d_all - lo); }
    // (an undecodable commitment is known only when its group has been decoded: verify_chunks asks for the client's status word then)
    VerifyInputs vin{&ready, h_st, nv};
    int rc = verify_chunks(C, "RangeProof", prove_range, P, prove_range, chunk, pf.data(), proof_len, Vh, d_vn2, seed, cidx.data(), okc.data(), grp, &v_shift, v_real.data(),
                           n_clients > 1 ? ridx.data() : nullptr, hier, skip.data(), &vin);
    C.sync();      // (a no-op after the checks; verify_chunks may have left before its first wait)
    std::vector<char> bad_commit(n_clients, 0);
    for (size_t i = 0; i < n_clients; i++) bad_commit[i] = (h_st[i] & 4u) != 0;
    // a single set with an invalid encoding: the reference cannot even build its Vec<RistrettoPoint> (decompress fails) -> FormatError;
    // in a batch the other clients are still verified and the offender gets ok = 0
    if (single && bad_commit[0]) { timing_end(C); return fail(ROFL_FORMAT_ERROR, "commitment is not a valid Ristretto encoding"); }
    timing_end(C);
    if (rc) return fail(rc, "proof rejected before verification (format / bitsize)");
    for (size_t i = 0; i < n_clients; i++) { int r = (bad_commit[i] || bad_format[i]) ? 0 : 1; for (size_t c = 0; c < nv; c++) r &= okc[i * nv + c]; ok_out[i] = r; }
    return ROFL_OK;
}

// One persistent worker thread per logical device for the sharded batch calls.  A worker serves one job at a time (concurrent sharded calls
// queue on the device's worker: the device is the shared resource anyway); jobs never throw (their body is guarded()).
class ShardWorkers {
    struct W { std::thread th; std::mutex mu; std::condition_variable cv; std::deque<std::pair<uint64_t, std::function<void()>>> q; uint64_t next_id = 1, done_id = 0; bool stop = false; };
    std::mutex mu; std::map<int, std::unique_ptr<W>> ws;
public:
    struct Ticket { bool ok = false; int dev = 0; uint64_t id = 0; };
    // Up to kPerDevice workers per device: the three legs of ONE L2 update (range proof by chunks, square proofs by elements, their verifiers)
    // split over the devices at the same time, and a leg's share of a device must not wait behind another leg's.  A job goes to an idle
    // worker of its device when there is one (created on demand), otherwise to the one with the shortest queue.
    static constexpr int kPerDevice = 3;
    Ticket submit(int dev, std::function<void()> job) {
        W *w = nullptr; int key = dev * kPerDevice;
        {   std::lock_guard<std::mutex> lk(mu);
            size_t best_load = ~(size_t)0;
            for (int sl = 0; sl < kPerDevice; sl++) {
                auto f = ws.find(dev * kPerDevice + sl);
                if (f == ws.end()) { if (best_load) { key = dev * kPerDevice + sl; best_load = 0; } break; }      // a worker that does not exist yet is idle
                size_t load; { std::lock_guard<std::mutex> lk2(f->second->mu); load = (size_t)(f->second->next_id - 1 - f->second->done_id); }
                if (load < best_load) { best_load = load; key = dev * kPerDevice + sl; }
                if (load == 0) break;
            }
            auto it = ws.find(key);
            if (it == ws.end()) {
                std::unique_ptr<W> nw(new W());
                W *raw = nw.get();
                try {
                    raw->th = std::thread([raw] {
                        for (;;) {
                            std::pair<uint64_t, std::function<void()>> j;
                            { std::unique_lock<std::mutex> lk(raw->mu); raw->cv.wait(lk, [&] { return raw->stop || !raw->q.empty(); }); if (raw->q.empty()) return; j = std::move(raw->q.front()); raw->q.pop_front(); }
                            j.second();
                            { std::lock_guard<std::mutex> lk(raw->mu); raw->done_id = j.first; }
                            raw->cv.notify_all();
                        }
                    });
                } catch (...) { return Ticket{}; }      // EAGAIN / bad_alloc: reported by the caller as an error code
                it = ws.emplace(key, std::move(nw)).first;
            }
            w = it->second.get();
        }
        Ticket t; t.ok = true; t.dev = key;
        { std::lock_guard<std::mutex> lk(w->mu); t.id = w->next_id++; w->q.emplace_back(t.id, std::move(job)); }
        w->cv.notify_all();
        return t;
    }
    void wait(const Ticket &t) {
        W *w; { std::lock_guard<std::mutex> lk(mu); w = ws[t.dev].get(); }
        std::unique_lock<std::mutex> lk(w->mu); w->cv.wait(lk, [&] { return w->done_id >= t.id; });
    }
    ~ShardWorkers() { for (auto &kv : ws) { { std::lock_guard<std::mutex> lk(kv.second->mu); kv.second->stop = true; } kv.second->cv.notify_all(); if (kv.second->th.joinable()) kv.second->th.join(); } }
};
ShardWorkers &shard_workers() { static ShardWorkers s; return s; }

// The devices a batch entry point spreads its clients over: rofl_set_option("devices", mask).  Empty = the calling thread's device.
std::vector<int> batch_devices() {
    std::vector<int> v; long m = opts().devices.load();
    for (int i = 0; i < kMaxDevices && m; i++, m >>= 1) if (m & 1) v.push_back(i);
    return v;
}
// Clients round-robin over the listed devices, one internal thread per device (bound to it for the duration), each running the ordinary
// single-device path on its share; verdicts / return codes land in the caller's arrays, in host memory -- in one process there is no
// collective to run.  run(share, device_slot) is the per-device body; it returns the call-level return code of its share.
// run(k) for k < nd, each on the worker of devs[k] (k = 0 on the calling thread), bound to its device; rcs[k] / errs[k] = what it returned.
// The return value is non-zero only when a worker thread could not be started.
template <class F> int run_on_devices(const std::vector<int> &devs, size_t nd, std::vector<int> &rcs, std::vector<std::string> &errs, F run) {
    rcs.assign(nd, ROFL_OK); errs.assign(nd, std::string());
    auto body = [&](size_t k) { DeviceBinding bind(devs[k]); rcs[k] = guarded([&]() -> int { return run(k); }); if (rcs[k]) errs[k] = g_err; };
    // one persistent worker per device (ShardWorkers): no thread is created on the call path, and a thread that cannot be created at
    // start-up is an error code, not std::terminate from a vector of joinable threads
    ShardWorkers &sw = shard_workers();
    std::vector<ShardWorkers::Ticket> tickets;
    for (size_t k = 1; k < nd; k++) {
        ShardWorkers::Ticket t = sw.submit(devs[k], [&body, k] { body(k); });
        if (!t.ok) { for (auto &q : tickets) sw.wait(q); return fail(ROFL_HIP_ERROR, "could not start the worker thread of a device"); }
        tickets.push_back(t);
    }
    body(0);
    for (auto &t : tickets) sw.wait(t);
    return ROFL_OK;
}
template <class F> int shard_over_devices(size_t n_clients, const std::vector<int> &devs, F run) {
    const size_t nd = std::min(devs.size(), n_clients);
    std::vector<std::vector<size_t>> share(nd);
    for (size_t i = 0; i < n_clients; i++) share[i % nd].push_back(i);
    std::vector<int> rcs; std::vector<std::string> errs;
    if (int rc = run_on_devices(devs, nd, rcs, errs, [&](size_t k) -> int { return run(share[k]); })) return rc;
    for (size_t k = 0; k < nd; k++) if (rcs[k]) return fail(rcs[k], errs[k]);
    return ROFL_OK;
}

// ONE client over several devices ("cfg 2/3 at > 1 GPU -> chunks over ranks").  The reference proves and verifies a client's
// chunks in parallel on its rayon pool (range_proof_vec/mod.rs:54-78, 168-181: every chunk is its own Bulletproof with its own transcript);
// here device k takes the k-th contiguous run of the P chunks -- a run's commitments are one span of the caller's array, the cost of a
// chunk does not depend on its data (padding chunks are proved like any other), so contiguous runs balance exactly like a round-robin deal.
// No collective: proofs and commitments land in the caller's host arrays.  The bytes are those of the unsplit call (the nonce index space is
// the client's).  Returns the geometry in *chunk_out / *P_out; 0 runs = the call is not splittable (one chunk, or one device).
std::vector<std::pair<size_t, size_t>> chunk_runs(size_t P, size_t nd) {
    std::vector<std::pair<size_t, size_t>> r;
    nd = std::min(nd, P);
    for (size_t k = 0; k < nd; k++) { size_t a = k * P / nd, b = (k + 1) * P / nd; if (b > a) r.emplace_back(a, b - a); }
    return r;
}
int create_split(const std::vector<int> &devs, const float *values, size_t d, const uint8_t *blind, size_t prove_range, size_t n_partition, unsigned fp_bits,
                 unsigned fp_frac, const rofl_nonce_t *nonce, uint8_t *proofs_out, size_t *plen_out, size_t *np_out, uint8_t *commits_out) {
    const size_t dp = next_pow2(d), nch = std::min(dp, n_partition), chunk = dp / nch, P = (dp + chunk - 1) / chunk;
    const size_t plen = 32 * (9 + 2 * (size_t)lg2u(prove_range * chunk));      // (every run reports the same; needed here to place the runs' proofs)
    auto runs = chunk_runs(P, devs.size());
    std::vector<int> rcs, rc1(runs.size(), ROFL_OK); std::vector<std::string> errs;
    if (int rc = run_on_devices(devs, runs.size(), rcs, errs, [&](size_t k) -> int {
            LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
            uint8_t *po = proofs_out + runs[k].first * plen, *co = commits_out + std::min(runs[k].first * chunk, d) * 32;
            size_t pl = 0, np = 0;
            return create_impl(C, 1, &values, d, &blind, prove_range, n_partition, fp_bits, fp_frac, nonce, &po, &pl, &np, &co, &rc1[k], false, runs[k].first, runs[k].second); }))
        return rc;
    // the outcome of the unsplit call, in the reference's order of checks: parameters, the range check over ALL values (:27-29), the
    // conversion's panic, then the upstream errors
    auto any = [&](const std::vector<int> &v, int code) { return std::find(v.begin(), v.end(), code) != v.end(); };
    if (any(rcs, ROFL_BAD_PARAM)) return fail(ROFL_BAD_PARAM, "bad parameter (the reference panics here)");
    for (size_t k = 0; k < rcs.size(); k++) if (rcs[k] >= ROFL_HIP_ERROR || rcs[k] == ROFL_COMM_ERROR) return fail(rcs[k], errs[k]);
    if (any(rc1, ROFL_VALUE_OUT_OF_RANGE)) return fail(ROFL_VALUE_OUT_OF_RANGE, "ValueOutOfRangeError");
    if (any(rc1, ROFL_NON_FINITE)) return fail(ROFL_NON_FINITE, "non-finite value (the reference panics in fixed::saturating_from_float)");
    for (size_t k = 0; k < rcs.size(); k++) if (rcs[k]) return fail(rcs[k], errs[k]);
    if (any(rc1, ROFL_NONCE_SHORT)) return fail(ROFL_NONCE_SHORT, "nonce stream too short");
    *plen_out = plen; *np_out = P;
    return ROFL_OK;
}
int verify_split(const std::vector<int> &devs, const uint8_t *proofs, size_t proof_len, size_t n_proofs, const uint8_t *commits, size_t d, size_t prove_range,
                 unsigned fp_bits, unsigned fp_frac, const uint8_t seed[32], int *ok_out) {
    const size_t chunk = next_pow2(d) / n_proofs;
    auto runs = chunk_runs(n_proofs, devs.size());
    std::vector<int> rcs, oks(runs.size(), 0); std::vector<std::string> errs;
    if (int rc = run_on_devices(devs, runs.size(), rcs, errs, [&](size_t k) -> int {
            LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
            const uint8_t *pp = proofs + runs[k].first * proof_len, *cc = commits + std::min(runs[k].first * chunk, d) * 32;
            return verify_impl(C, 1, &pp, proof_len, n_proofs, &cc, d, prove_range, fp_bits, fp_frac, seed, &oks[k], true, nullptr, 32, runs[k].first, runs[k].second); }))
        return rc;
    *ok_out = 0;
    for (size_t k = 0; k < rcs.size(); k++) if (rcs[k]) return fail(rcs[k], errs[k]);      // a malformed set is the call's FormatError, whichever run met it
    int ok = 1; for (int o : oks) ok &= o;
    *ok_out = ok;
    return ROFL_OK;
}
// can this (d, n_proofs) set be split?  Only the regular case: the proofs cover the padded vector exactly
bool verify_splittable(size_t d, size_t n_proofs) { if (!d || n_proofs < 2) return false; size_t dp = next_pow2(d); return n_proofs <= dp && (dp / n_proofs) * n_proofs == dp; }
}  // namespace

// ================================================================ C ABI
extern "C" {

// rofl_set_device binds the calling thread to `device` (and makes it the default of threads that have no binding of their own), then
// brings that device's context up so that a missing device shows here and not in the first proof.
int rofl_set_device(int device) {
    if (device < 0 || device >= kMaxDevices) return fail(ROFL_BAD_PARAM, "bad device index");
    const int prev = t_device;
    t_device = device;
    int rc = guarded([&]() -> int { LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c; C.init(); return ROFL_OK; });
    if (rc) {      // a device that cannot be used is not selected -- and leaves no half-built context behind (rofl_dbg_map_device would see it "in use")
        t_device = prev;
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        auto it = g_ctxs.find(device);
        if (it != g_ctxs.end() && !it->second->inited && it->second->active_calls.load() == 0) { delete it->second; g_ctxs.erase(it); }
        return rc;
    }
    // The process default -- what threads without a binding of their own follow -- is set by the FIRST successful call only: in a server whose
    // pool threads each bind their own device, the default of unbound threads must not become whichever thread called last.
    // rofl_set_option("default_device", d) moves it explicitly.
    bool expected = false;
    if (g_default_set.compare_exchange_strong(expected, true)) g_default_device.store(device);
    return rc;
}
int rofl_get_device(int *device_out) { if (!device_out) return fail(ROFL_BAD_PARAM, "bad parameter"); *device_out = current_device(); return ROFL_OK; }
int rofl_bind_device(int device) { if (device < -1 || device >= kMaxDevices) return fail(ROFL_BAD_PARAM, "bad device index"); t_device = device; return ROFL_OK; }
int rofl_dbg_bind_device(int device) { if (device < -1 || device >= kMaxDevices) return ROFL_BAD_PARAM; t_device = device; return ROFL_OK; }
int rofl_dbg_map_device(int logical, int physical) {
    if (logical < 0 || logical >= kMaxDevices || physical < 0) return ROFL_BAD_PARAM;
    devmap_init();
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    if (g_ctxs.count(logical)) return g_devmap[logical] == physical ? ROFL_OK : fail(ROFL_BAD_PARAM, "the device is already in use");      // (asking for the mapping it already has is fine)
    g_devmap[logical] = physical; return ROFL_OK;
}
int rofl_last_error(char *buf, size_t len) { if (!buf || !len) return ROFL_BAD_PARAM; snprintf(buf, len, "%s", g_err.c_str()); return ROFL_OK; }
size_t rofl_next_pow2(size_t v) { return v ? next_pow2(v) : 0; }
size_t rofl_rangeproof_chunks(size_t d, size_t n_partition) {
    if (!d || !n_partition) return 0;
    size_t dp = next_pow2(d), nc = std::min(dp, n_partition), chunk = dp / nc; return (dp + chunk - 1) / chunk;
}
size_t rofl_rangeproof_size(size_t n_bits, size_t d, size_t n_partition) {
    if (!d || !n_partition) return 0;
    size_t dp = next_pow2(d), nc = std::min(dp, n_partition), chunk = dp / nc; return 32 * (9 + 2 * (size_t)lg2u(n_bits * chunk));
}
size_t rofl_nonces_per_chunk(size_t n_bits, size_t m) { return m * (2 * n_bits + 4); }

int rofl_bp_gens_prepare(size_t n_bits, size_t m) {
    return guarded([&]() -> int { LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c; C.init(); if (!n_bits || !m) return fail(ROFL_BAD_PARAM, "bad parameter"); { GensPin pin = get_gens(C, n_bits, m); }
        if (!gens_wait_full(C, n_bits, m)) g_err = "HBM is short: the shape keeps its compact fold table (same results, slower first fold)";      // not an error: see rofl_zk.h
        return ROFL_OK; });
}
int rofl_bp_gens_prepare_verify(size_t n_bits, size_t m) {
    return guarded([&]() -> int { LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c; C.init(); if (!n_bits || !m) return fail(ROFL_BAD_PARAM, "bad parameter");
        { GensPin pin = get_gens(C, n_bits, m, GENS_VERIFY); }
        return ROFL_OK; });
}
int rofl_bp_gens_table_bytes(size_t n_bits, size_t m, size_t *bytes_out) {
    return guarded([&]() -> int {
        if (!bytes_out) return fail(ROFL_BAD_PARAM, "bad parameter");
        Ctx &P = ctx(); { std::lock_guard<std::mutex> g(P.init_mu); P.init(); }
        std::lock_guard<std::mutex> lk(P.gens_mu);
        auto it = P.gens.find(std::make_pair(n_bits, m));
        *bytes_out = it == P.gens.end() ? 0 : it->second->bytes;
        return ROFL_OK;
    });
}
int rofl_bp_gens_export(size_t n_bits, size_t m, uint8_t *G_out, uint8_t *H_out) {
    return guarded([&]() -> int {
        LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c; C.init();
        if (!n_bits || !m) return fail(ROFL_BAD_PARAM, "bad parameter");
        GensPin gens = get_gens(C, n_bits, m, GENS_VERIFY); niels *tbl = gens.tbl(); size_t N = n_bits * m;
        // encode through the commit path: decode-free -- use k_msm-free helper: copy niels back and encode on host
        std::vector<niels> h(2 * N);
        HIPCHK(hipMemcpy(h.data(), tbl, sizeof(niels) * 2 * N, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < 2 * N; i++) { ge p = ge_from_niels(h[i]); ristretto_encode((i < N ? G_out + 32 * i : H_out + 32 * (i - N)), p); }
        return ROFL_OK;
    });
}

int rofl_create_rangeproof(const float *values, size_t d, const uint8_t *blindings32, size_t d_blindings, size_t prove_range, size_t n_partition,
                           unsigned fp_bits, unsigned fp_frac, const rofl_nonce_t *nonce, uint8_t *proofs_out, size_t *proof_len_out,
                           size_t *n_proofs_out, uint8_t *commits_out) {
    return guarded([&]() -> int {
        if (d != d_blindings) return fail(ROFL_WRONG_NUM_BLINDING, "WrongNumBlindingFactors");
        // rofl_set_option("devices", mask) with several devices: the client's chunks are dealt to them (create_split); malformed parameters take the ordinary path and get its diagnostics
        std::vector<int> devs = batch_devices();
        if (devs.size() > 1 && values && blindings32 && nonce && proofs_out && commits_out && proof_len_out && n_proofs_out && d && n_partition && valid_fp(fp_bits, fp_frac) &&
            prove_range && prove_range <= fp_bits && rofl_rangeproof_chunks(d, n_partition) > 1)
            return create_split(devs, values, d, blindings32, prove_range, n_partition, fp_bits, fp_frac, nonce, proofs_out, proof_len_out, n_proofs_out, commits_out);
        std::unique_ptr<DeviceBinding> bind; if (devs.size() == 1) bind.reset(new DeviceBinding(devs[0]));
        LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
        int rc1 = ROFL_OK;
        int rc = create_impl(C, 1, &values, d, &blindings32, prove_range, n_partition, fp_bits, fp_frac, nonce, &proofs_out, proof_len_out, n_proofs_out, &commits_out, &rc1, true);
        return rc ? rc : rc1; });
}
int rofl_create_rangeproof_chunks(const float *values, size_t d, const uint8_t *blindings32, size_t d_blindings, size_t prove_range, size_t n_partition,
                                  unsigned fp_bits, unsigned fp_frac, const rofl_nonce_t *nonce, size_t chunk_first, size_t chunk_count,
                                  uint8_t *proofs_out, size_t *proof_len_out, uint8_t *commits_out, size_t *n_commits_out) {
    return guarded([&]() -> int {
        if (!values || !blindings32 || !nonce || !proofs_out || !proof_len_out || !commits_out || !n_commits_out || chunk_count == 0) return fail(ROFL_BAD_PARAM, "bad parameter");
        if (d != d_blindings) return fail(ROFL_WRONG_NUM_BLINDING, "WrongNumBlindingFactors");
        LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
        int rc1 = ROFL_OK; size_t np = 0;
        int rc = create_impl(C, 1, &values, d, &blindings32, prove_range, n_partition, fp_bits, fp_frac, nonce, &proofs_out, proof_len_out, &np, &commits_out, &rc1, true, chunk_first, chunk_count);
        if (rc || rc1) return rc ? rc : rc1;
        const size_t dp = next_pow2(d), chunk = dp / std::min(dp, n_partition), el0 = chunk_first * chunk;
        *n_commits_out = el0 >= d ? 0 : std::min(d - el0, chunk_count * chunk);
        return ROFL_OK; });
}
int rofl_create_rangeproof_batch(size_t n_clients, const float *const *values, size_t d, const uint8_t *const *blindings32, size_t prove_range,
                                 size_t n_partition, unsigned fp_bits, unsigned fp_frac, const rofl_nonce_t *nonces, uint8_t *const *proofs_out,
                                 size_t *proof_len_out, size_t *n_proofs_out, uint8_t *const *commits_out, int *rc_out) {
    if (!values || !blindings32 || !proofs_out || !commits_out || !rc_out || !proof_len_out || !n_proofs_out || !nonces) return fail(ROFL_BAD_PARAM, "bad parameter");
    const bool single = false;      // per-client outcomes in rc_out, also for a batch of ONE (the return value is for errors of the whole call)
    std::vector<int> devs = batch_devices();
    if (devs.empty() || n_clients < 2)
        return guarded([&]() -> int { std::unique_ptr<DeviceBinding> bind; if (!devs.empty()) bind.reset(new DeviceBinding(devs[0]));
            LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
            return create_impl(C, n_clients, values, d, blindings32, prove_range, n_partition, fp_bits, fp_frac, nonces, proofs_out, proof_len_out, n_proofs_out, commits_out, rc_out, single); });
    std::mutex out_mu;
    return guarded([&]() -> int { return shard_over_devices(n_clients, devs, [&](const std::vector<size_t> &idx) -> int {
        const size_t k = idx.size();
        std::vector<const float *> v(k); std::vector<const uint8_t *> b(k); std::vector<rofl_nonce_t> nn(k); std::vector<uint8_t *> po(k), co(k); std::vector<int> rc(k, ROFL_OK);
        for (size_t j = 0; j < k; j++) { v[j] = values[idx[j]]; b[j] = blindings32[idx[j]]; nn[j] = nonces[idx[j]]; po[j] = proofs_out[idx[j]]; co[j] = commits_out[idx[j]]; }
        size_t plen = 0, np = 0;
        LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
        int r = create_impl(C, k, v.data(), d, b.data(), prove_range, n_partition, fp_bits, fp_frac, nn.data(), po.data(), &plen, &np, co.data(), rc.data(), false);
        for (size_t j = 0; j < k; j++) rc_out[idx[j]] = rc[j];
        if (!r) { std::lock_guard<std::mutex> lk(out_mu); *proof_len_out = plen; *n_proofs_out = np; }
        return r; }); });
}
int rofl_verify_rangeproof(const uint8_t *proofs, size_t proof_len, size_t n_proofs, const uint8_t *commits32, size_t d, size_t prove_range,
                           unsigned fp_bits, unsigned fp_frac, const uint8_t verifier_seed[32], int *ok_out) {
    return guarded([&]() -> int {
        std::vector<int> devs = batch_devices();
        if (devs.size() > 1 && proofs && commits32 && ok_out && verifier_seed && verify_splittable(d, n_proofs))      // the client's proofs over the listed devices (verify_split)
            return verify_split(devs, proofs, proof_len, n_proofs, commits32, d, prove_range, fp_bits, fp_frac, verifier_seed, ok_out);
        std::unique_ptr<DeviceBinding> bind; if (devs.size() == 1) bind.reset(new DeviceBinding(devs[0]));
        LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
        return verify_impl(C, 1, &proofs, proof_len, n_proofs, &commits32, d, prove_range, fp_bits, fp_frac, verifier_seed, ok_out, true); });
}
int rofl_verify_rangeproof_chunks(const uint8_t *proofs, size_t proof_len, size_t n_proofs, size_t chunk_first, size_t chunk_count, const uint8_t *commits32,
                                  size_t d, size_t prove_range, unsigned fp_bits, unsigned fp_frac, const uint8_t verifier_seed[32], int *ok_out) {
    return guarded([&]() -> int {
        if (!proofs || !commits32 || !ok_out || !verifier_seed || chunk_count == 0) return fail(ROFL_BAD_PARAM, "bad parameter");
        { const size_t dp = d ? next_pow2(d) : 0;
          if (!d || !n_proofs || n_proofs > dp || (dp / n_proofs) * n_proofs != dp) return fail(ROFL_BAD_PARAM, "a run of chunks needs a proof count that covers the padded vector exactly"); }
        LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
        return verify_impl(C, 1, &proofs, proof_len, n_proofs, &commits32, d, prove_range, fp_bits, fp_frac, verifier_seed, ok_out, true, nullptr, 32, chunk_first, chunk_count); });
}
int rofl_verify_rangeproof_batch(size_t n_clients, const uint8_t *const *proofs, size_t proof_len, size_t n_proofs, const uint8_t *const *commits32,
                                 size_t d, size_t prove_range, unsigned fp_bits, unsigned fp_frac, const uint8_t verifier_seed[32], int *ok_out) {
    return rofl_verify_rangeproof_batch_strided(n_clients, proofs, proof_len, n_proofs, commits32, 32, d, prove_range, fp_bits, fp_frac, verifier_seed, ok_out);
}
int rofl_verify_rangeproof_batch_strided(size_t n_clients, const uint8_t *const *proofs, size_t proof_len, size_t n_proofs, const uint8_t *const *commits32, size_t commit_stride,
                                         size_t d, size_t prove_range, unsigned fp_bits, unsigned fp_frac, const uint8_t verifier_seed[32], int *ok_out) {
    if (!proofs || !commits32 || !ok_out || !verifier_seed || commit_stride < 32) return fail(ROFL_BAD_PARAM, "bad parameter");
    const bool single = false;      // a batch has per-member verdicts, also a batch of ONE: a malformed member gets ok = 0, not the FormatError of rofl_verify_rangeproof (found by the long batch fuzz)
    std::vector<int> devs = batch_devices();
    if (devs.empty() || n_clients < 2)
        return guarded([&]() -> int { std::unique_ptr<DeviceBinding> bind; if (!devs.empty()) bind.reset(new DeviceBinding(devs[0]));
            LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
            return verify_impl(C, n_clients, proofs, proof_len, n_proofs, commits32, d, prove_range, fp_bits, fp_frac, verifier_seed, ok_out, single, nullptr, commit_stride); });
    for (size_t i = 0; i < n_clients; i++) ok_out[i] = 0;
    return guarded([&]() -> int { return shard_over_devices(n_clients, devs, [&](const std::vector<size_t> &idx) -> int {
        const size_t k = idx.size();
        std::vector<const uint8_t *> p(k), c(k); std::vector<int> ok(k, 0);
        for (size_t j = 0; j < k; j++) { p[j] = proofs[idx[j]]; c[j] = commits32[idx[j]]; }
        LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
        int r = verify_impl(C, k, p.data(), proof_len, n_proofs, c.data(), d, prove_range, fp_bits, fp_frac, verifier_seed, ok.data(), false, idx.data(), commit_stride);
        for (size_t j = 0; j < k; j++) ok_out[idx[j]] = ok[j];
        return r; }); });
}
int rofl_clip_f32(const float *in, size_t d, size_t prove_range, unsigned fp_bits, unsigned fp_frac, float *out) {
    if (!valid_fp(fp_bits, fp_frac) || prove_range == 0) return fail(ROFL_BAD_PARAM, "bad parameter");
    float mn, mx; clip_bounds(prove_range, fp_bits, fp_frac, &mn, &mx);
    for (size_t i = 0; i < d; i++) { float t = fmaxf(mn, in[i]); out[i] = fminf(mx, t); }
    return ROFL_OK;
}

int rofl_create_rangeproof_l2(const float *values, size_t d, const uint8_t *blindings32, size_t d_blindings, size_t prove_range, size_t n_partition,
                              unsigned fp_bits, unsigned fp_frac, const rofl_nonce_t *nonce, uint8_t *proof_out, size_t *proof_len_out, uint8_t commit_out[32]) {
    return guarded([&]() -> int {
        LaneLock lane_lock = acquire_lane(false, true); Ctx &C = *lane_lock.c;
        if (d != d_blindings) return fail(ROFL_WRONG_NUM_BLINDING, "WrongNumBlindingFactors");
        if (!valid_fp(fp_bits, fp_frac) || d == 0 || n_partition == 0 || prove_range == 0 || !nonce) return fail(ROFL_BAD_PARAM, "bad parameter (the reference panics here)");
        float mn, mx; clip_bounds(prove_range, fp_bits, fp_frac, &mn, &mx);
        for (size_t i = 0; i < d; i++) if (mn > values[i] || values[i] > mx) return fail(ROFL_VALUE_OUT_OF_RANGE, "ValueOutOfRangeError");
        // l2_range_proof_vec/mod.rs:37-79: scalar sum of squares, the f32 shadow sum in the reference's
        // serial left-to-right order (its result decides the OverflowError branch), and the blinding sum.
        sc val = sc_zero(), bsum = sc_zero();
        volatile float val_float = 0.0f; float shift = (float)(1u << fp_frac);
        for (size_t i = 0; i < d; i++) {
            sc s; int rc = f32_to_sc(values[i], fp_bits, fp_frac, &s); if (rc) return fail(rc, "non-finite value");
            val = sc_add(val, h_mul(s, s));
            volatile float q = sc_to_f32(s, fp_bits, fp_frac); volatile float qq = q * q; volatile float term = qq * shift;
            val_float = (i == 0) ? term : val_float + term;
            sc bl = sc_frombytes(blindings32 + 32 * i); if (sc_geq_l(bl.v)) bl = sc_from_mont(sc_to_mont(bl));
            bsum = sc_add(bsum, bl);
        }
        float val_f = sc_to_f32(val, fp_bits, fp_frac);
        volatile float diff = val_f - val_float;
        if (std::fabs(diff) > 1.1920929e-07f) return fail(ROFL_OVERFLOW, "OverflowError");
        if (val_f > l2_clip_bound(prove_range, fp_bits, fp_frac)) return fail(ROFL_NORM_OUT_OF_RANGE, "NormOutOfRangeError");
        if (!(prove_range == 8 || prove_range == 16 || prove_range == 32 || prove_range == 64)) return fail(ROFL_INVALID_BITSIZE, "InvalidBitsize");
        if (nonce->mode == 0 && nonce->stream_scalars < 2 * prove_range + 4) return fail(ROFL_NONCE_SHORT, "nonce stream too short");
        C.init();
        timing_begin(C);
        u64 v = read_from_bytes(val, fp_bits);
        u64 *vshift = C.vshift.as<u64>(1); sc *d_bl = C.blind.as<sc>(1);
        HIPCHK(hipMemcpyAsync(vshift, &v, 8, hipMemcpyHostToDevice, C.stream));
        HIPCHK(hipMemcpyAsync(d_bl, &bsum, 32, hipMemcpyHostToDevice, C.stream));
        uint8_t *Vb = C.Vbytes.as<uint8_t>(32);
        ROFL_LAUNCH(k_commit, grid1(1), dim3(TPB), 0, C.stream, 1u, vshift, (const sc *)nullptr, d_bl, C.d_tabB8, C.d_tabBb8, (const niels *)nullptr, Vb, (uint8_t *)nullptr, 0u, 1u);
        uint8_t hV[32];
        HIPCHK(hipMemcpyAsync(hV, Vb, 32, hipMemcpyDeviceToHost, C.stream));
        C.sync();
        // BulletproofGens::new(64, 1), label "L2RangeProof" (l2_range_proof_vec/mod.rs:156-171): the first
        // prove_range generators of party 0 are the same chain prefix.
        ChunkNonce cn{}; cn.mode = nonce->mode;
        if (nonce->mode == 1) memcpy(cn.seed.w, nonce->seed, 32);
        else { uint8_t *sb = C.stream_buf.as<uint8_t>(nonce->stream_scalars * 64 + 64); C.up(sb, nonce->stream, nonce->stream_scalars * 64, C.stream); cn.d_stream = sb; cn.stream_scalars = nonce->stream_scalars; }
        uint8_t *pout = proof_out;
        prove_chunks(C, "L2RangeProof", 1, prove_range, 1, vshift, d_bl, std::vector<ChunkNonce>(1, cn), hV, &pout);
        timing_end(C);
        memcpy(commit_out, hV, 32);
        *proof_len_out = 32 * (9 + 2 * (size_t)lg2u(prove_range));
        return ROFL_OK;
    });
}
int rofl_verify_rangeproof_l2(const uint8_t *proof, size_t proof_len, const uint8_t commit[32], size_t prove_range, unsigned fp_bits, unsigned fp_frac,
                              const uint8_t verifier_seed[32], int *ok_out) {
    return guarded([&]() -> int {
        LaneLock lane_lock = acquire_lane(false, true); Ctx &C = *lane_lock.c;
        *ok_out = 0;
        if (!valid_fp(fp_bits, fp_frac) || prove_range == 0) return fail(ROFL_BAD_PARAM, "bad parameter");
        C.init();
        timing_begin(C);
        uint8_t *d_in = C.Cbytes.as<uint8_t>(32), *d_enc = C.Vbytes.as<uint8_t>(32);
        niels *d_vn = C.gbuf[0].as<niels>(1);
        u32 *status = C.status.as<u32>(4);
        HIPCHK(hipMemsetAsync(status, 0, 16, C.stream));
        HIPCHK(hipMemcpyAsync(d_in, commit, 32, hipMemcpyHostToDevice, C.stream));
        ROFL_LAUNCH(k_decode, grid1(1), dim3(TPB), 0, C.stream, 1u, 1u, d_in, (const niels *)nullptr, d_vn, d_enc, status);
        uint8_t hV[32]; u32 st = 0;
        HIPCHK(hipMemcpyAsync(hV, d_enc, 32, hipMemcpyDeviceToHost, C.stream));
        HIPCHK(hipMemcpyAsync(&st, status, 4, hipMemcpyDeviceToHost, C.stream));
        C.sync();
        if (st & 4u) return fail(ROFL_FORMAT_ERROR, "commitment is not a valid Ristretto encoding");
        u64 cidx = 0; int ok = 0;
        int rc = verify_chunks(C, "L2RangeProof", 64, 1, prove_range, 1, proof, proof_len, hV, d_vn, verifier_seed, &cidx, &ok);
        timing_end(C);
        if (rc) return fail(rc, "proof rejected before verification (format / bitsize)");
        *ok_out = ok;
        return ROFL_OK;
    });
}

// verify_rangeproof_l2 for the clients of a round (params.rs:220-231, server.rs:656-687): n_clients one-value proofs over the (64, 1)
// generators, commitments = each client's sum of c_sq.  The proofs share the generators, so with verify_batch = 2 they are ONE
// random-weighted equation (a closer look only when it fails); with 1 one check per client, all in one launch sequence.  Per-client
// verdicts either way; a malformed member (length is per call; non-canonical scalar, undecodable commitment, identity point) fails alone.
int rofl_verify_rangeproof_l2_batch(size_t n_clients, const uint8_t *const *proofs, size_t proof_len, const uint8_t *commits32, size_t prove_range,
                                    unsigned fp_bits, unsigned fp_frac, const uint8_t verifier_seed[32], int *ok_out) {
    if (!ok_out || !verifier_seed || (n_clients && (!proofs || !commits32))) return fail(ROFL_BAD_PARAM, "bad parameter");
    return guarded([&]() -> int {
        LaneLock lane_lock = acquire_lane(false, true); Ctx &C = *lane_lock.c;
        for (size_t i = 0; i < n_clients; i++) ok_out[i] = 0;
        if (!valid_fp(fp_bits, fp_frac) || prove_range == 0) return fail(ROFL_BAD_PARAM, "bad parameter");
        if (n_clients == 0) return ROFL_OK;
        if (proof_len % 32 != 0 || proof_len < 7 * 32) return fail(ROFL_FORMAT_ERROR, "FormatError: proof length");
        const size_t ne = (proof_len - 7 * 32) / 32;
        if (ne < 2 || (ne - 2) % 2 != 0 || (ne - 2) / 2 >= 32) return fail(ROFL_FORMAT_ERROR, "FormatError: proof length");
        const size_t lg = (ne - 2) / 2, nc = n_clients;
        if (nc > kMaxBatchMembers / 2) return fail(ROFL_BAD_PARAM, "batch too large (split it)");      // a member = a row of gridDim.y
        std::vector<uint8_t> pf(nc * proof_len);
        std::vector<char> skip(nc, 0);
        for (size_t i = 0; i < nc; i++) {
            if (is_device_ptr(proofs[i])) HIPCHK(hipMemcpy(&pf[i * proof_len], proofs[i], proof_len, hipMemcpyDeviceToHost)); else memcpy(&pf[i * proof_len], proofs[i], proof_len);
            uint8_t *pb = &pf[i * proof_len];
            const size_t offs[5] = {128, 160, 192, 7 * 32 + 64 * lg, 7 * 32 + 64 * lg + 32};
            for (size_t o : offs) if (!sc_is_canonical_bytes(pb + o)) { skip[i] = 1; memset(pb + o, 0, 32); }      // its own verdict: false; the others go on
        }
        if (!(prove_range == 8 || prove_range == 16 || prove_range == 32 || prove_range == 64)) return fail(ROFL_INVALID_BITSIZE, "proof rejected before verification (format / bitsize)");
        if (prove_range != ((size_t)1 << lg)) return ROFL_OK;      // VerificationError for every client -> false
        C.init();
        C.batch_mode = nc > 1;
        timing_begin(C);
        uint8_t *d_in = C.Cbytes.as<uint8_t>(nc * 32), *d_enc = C.Vbytes.as<uint8_t>(nc * 32);
        niels *d_vn = C.gbuf[0].as<niels>(nc);
        u32 *status = C.status.as<u32>(nc + 4);
        HIPCHK(hipMemsetAsync(status, 0, 4 * (nc + 4), C.stream));
        C.up(d_in, commits32, nc * 32, C.stream);
        ROFL_LAUNCH(k_decode, grid1(1, (u32)nc), dim3(TPB), 0, C.stream, 1u, 1u, d_in, (const niels *)nullptr, d_vn, d_enc, status);
        uint8_t *hV = C.h_V.as<uint8_t>(nc * 32); u32 *h_st = C.h_misc.as<u32>(nc + 4);
        HIPCHK(hipMemcpyAsync(hV, d_enc, nc * 32, hipMemcpyDeviceToHost, C.stream));
        HIPCHK(hipMemcpyAsync(h_st, status, 4 * nc, hipMemcpyDeviceToHost, C.stream));
        C.sync();
        for (size_t i = 0; i < nc; i++) if (h_st[i] & 4u) skip[i] = 1;
        std::vector<u64> cidx(nc, 0), ridx(nc); std::vector<int> okc(nc, 0);
        for (size_t i = 0; i < nc; i++) ridx[i] = (u64)i << 24;
        const bool hier = opts().verify_batch.load() == 2 && nc > 1;
        int rc = verify_chunks(C, "L2RangeProof", 64, nc, prove_range, 1, pf.data(), proof_len, hV, d_vn, verifier_seed, cidx.data(), okc.data(), 1, nullptr, nullptr,
                               ridx.data(), hier, skip.data());
        timing_end(C);
        if (rc) return fail(rc, "proof rejected before verification (format / bitsize)");
        for (size_t i = 0; i < nc; i++) ok_out[i] = skip[i] ? 0 : okc[i];
        return ROFL_OK;
    });
}

namespace {
DMerlin sigma_init_state(int kind) {
    const char *lbl = kind == 0 ? "RandProof" : (kind == 1 ? "SquareRandProof" : "SquareProof");
    Merlin t(lbl, strlen(lbl));
    // rand_proof_domain_sep (rand_proof/transcript.rs:20-22) -- but the begin_op of the NEXT append depends on pos_begin,
    // so the whole state (bytes, pos, pos_begin) is handed to the kernel
    t.append("dom-sep", (const uint8_t *)"randomness proof v1", 19);
    DMerlin d; memcpy(d.st, t.b(), 200); d.pos = t.pos; d.pos_begin = t.pos_begin;
    return d;
}
// [elem_first, elem_first + d) of a vector of d_all elements (d_all = 0: the whole vector): the elements of a Sigma-proof vector are independent
// of each other (rand_proof_vec/mod.rs:45-58, square_rand_proof_vec/mod.rs:45-58: one proof per element on the rayon pool), so a device or a
// rank can take any run of them -- SURVEY 8(e), the third unit of independence.  The arrays passed in are the RUN's (values, r1, r2, existing
// and both outputs start at the run's first element); the nonce index space stays the vector's (element i draws from nn * i), so the bytes
// are those of the unsplit call.
int sigma_create(int kind, const float *values, size_t d, const uint8_t *r1, size_t d_r1, const uint8_t *r2, const uint8_t *existing,
                 unsigned fp_bits, unsigned fp_frac, const rofl_nonce_t *nonce, uint8_t *proofs_out, uint8_t *commits_out, size_t elem_first = 0, size_t d_all = 0) {
    LaneLock lane_lock = acquire_lane(false, true); Ctx &C = *lane_lock.c;
    if (d != d_r1) return fail(ROFL_WRONG_NUM_BLINDING, "WrongNumBlindingFactors");
    if (!valid_fp(fp_bits, fp_frac) || !nonce) return fail(ROFL_BAD_PARAM, "bad parameter");
    if (d_all == 0) { d_all = d; elem_first = 0; }
    if (elem_first > d_all || d > d_all - elem_first) return fail(ROFL_BAD_PARAM, "element range outside the vector");
    bool has_sq = kind != 0;
    size_t npts = 1 + (kind != 2) + (has_sq ? 1 : 0), nn = has_sq ? 3 : 2, clen = 32 * npts, plen = 32 * (npts + nn);
    if (nonce->mode == 0 && nonce->stream_scalars < nn * d_all) return fail(ROFL_NONCE_SHORT, "nonce stream too short");
    if (d == 0) return ROFL_OK;
    C.init();
    timing_begin(C);
    float *dv = C.vals.as<float>(d); sc *dr1 = C.tmp_in.as<sc>(d); sc *dr2 = has_sq ? C.tmp_in2.as<sc>(d) : nullptr;
    uint8_t *dex = existing ? C.Cbytes.as<uint8_t>(d * 32) : nullptr;
    uint8_t *dp = C.aux_pts.as<uint8_t>(d * plen), *dc = C.aux_scal.as<uint8_t>(d * clen);
    u32 *status = C.status.as<u32>(4);
    HIPCHK(hipMemsetAsync(status, 0, 16, C.stream));
    C.up(dv, values, 4 * d, C.stream);
    C.up(dr1, r1, 32 * d, C.stream);
    if (has_sq) C.up(dr2, r2, 32 * d, C.stream);
    if (dex) C.up(dex, existing, 32 * d, C.stream);
    NonceSeed seed{}; const uint8_t *d_stream = nullptr; u64 ss = 0;
    const u64 nonce_base = (u64)nn * elem_first;
    if (nonce->mode == 1) memcpy(seed.w, nonce->seed, 32);
    else {      // only the run's part of the stream goes up; the kernels index it by the vector's nonce index, so the base address is moved back by what stayed behind
        const size_t run_bytes = nn * d * 64, run_off = (size_t)nonce_base * 64;
        uint8_t *sb = C.stream_buf.as<uint8_t>(run_bytes + 64); C.up(sb, nonce->stream + run_off, run_bytes, C.stream);
        d_stream = reinterpret_cast<const uint8_t *>(reinterpret_cast<uintptr_t>(sb) - run_off); ss = nonce_base + nn * d; }
    // algorithmic work per element AS BUILT: fixed-base multiplications of 32 mixed additions (radix-256 tables) -- L, L' and c_sq, c_sq' two each,
    // R, R' one each -- and 2 * npts encodings (the one-thread-per-element form: 7 multiplications of 64 additions + a variable-base one of
    // ~325 point operations); bytes as SURVEY 8(d): value + randomness in, commitments + proof out
    static const bool split = !(knob("ROFL_SIGMA_SPLIT") && atoi(knob("ROFL_SIGMA_SPLIT")) == 0);
    const uint64_t sg_muls = split ? (uint64_t)(4 + (kind != 2 ? 2 : 0) + (has_sq ? 4 : 0)) * 32 * 7 + 2 * npts * 265 : (uint64_t)7 * 64 * 7 + 325 * 8 + 2 * npts * 265;
    { KSpan ks_sigma(C.tm, C.stream, ROFL_TK_SIGMA, (uint64_t)d * sg_muls, (uint64_t)d * (4 + 32 * (has_sq ? 2 : 1) + clen + plen));
      if (split) {      // one thread per point (blockIdx.y = slot), then transcripts + responses per element
          SgSlots sl{}; auto add = [&](int id) { sl.id[sl.n++] = id; };
          // c_sq' in its fixed-base form (SG_CSQP_F: the prover knows the opening of L); a commitment handed in is compared with m B + r1 Bb first
          // (SG_LCMP) and the elements where it differs -- none, unless the caller's commitments are not the values' -- are redone by
          // k_sigma_point_var exactly as the reference computes them
          uint8_t *marks = nullptr;
          if (has_sq && dex) marks = C.vspart.as<uint8_t>(d);      // (written for every element by the SG_LCMP slot: no clearing)
          if (!dex) add(SG_L); else add(has_sq ? SG_LCMP : SG_LCHK);
          if (has_sq) { add(SG_CSQP_F); add(SG_CSQ); }
          add(SG_LP);
          if (kind != 2) { add(SG_R); add(SG_RP); }
          ROFL_LAUNCH(k_sigma_points, dim3((unsigned)((d + 63) / 64), (unsigned)sl.n), dim3(64), 0, C.stream, kind, sl, (u32)d, dv, fp_bits, fp_frac, dr1, dr2, dex,
                      nonce->mode, seed, d_stream, ss, nonce_base, C.d_tabB8, C.d_tabBb8, dp, dc, status, marks);
          if (marks)
              ROFL_LAUNCH(k_sigma_point_var, dim3((unsigned)std::min<size_t>((d + 63) / 64, 256)), dim3(64), 0, C.stream, kind, (u32)d, dv, fp_bits, fp_frac, dr1, dr2, dex,      // (walks the marks: one block per CU at most)
                          nonce->mode, seed, d_stream, ss, nonce_base, C.d_tabB, C.d_tabBb, dp, dc, status, marks);
          ROFL_LAUNCH(k_sigma_finish, grid1(d), dim3(TPB), 0, C.stream, kind, (u32)d, dv, fp_bits, fp_frac, dr1, dr2, dex, nonce->mode, seed, d_stream, ss, nonce_base,
                      sigma_init_state(kind), dp, dc, status);
      } else
      ROFL_LAUNCH(k_sigma_prove, dim3((unsigned)((d + 63) / 64)), dim3(64), 0, C.stream, kind, (u32)d, dv, fp_bits, fp_frac, dr1, dr2, dex,
                         nonce->mode, seed, d_stream, ss, nonce_base, sigma_init_state(kind), C.d_tabB, C.d_tabBb, dp, dc, status); }
    u32 st = 0;
    C.down(proofs_out, dp, d * plen, C.stream);
    C.down(commits_out, dc, d * clen, C.stream);
    HIPCHK(hipMemcpyAsync(&st, status, 4, hipMemcpyDeviceToHost, C.stream));
    C.sync();
    timing_end(C);
    if (st & 2u) return fail(ROFL_NON_FINITE, "non-finite value (the reference panics in fixed::saturating_from_float)");
    if (st & 4u) return fail(ROFL_FORMAT_ERROR, "invalid Ristretto encoding");
    return ROFL_OK;
}
// Verification of the per-element Sigma-proofs of `nc` vectors of d elements (the clients of a round: the reference's server verifies every
// client's update, server.rs:656-687, each with verify_randproof_vec / verify_l2rangeproof_vec, params.rs:188-189, 208, 262).  One launch
// sequence for all of them: the clients' bytes are staged and uploaded group by group (the pool copies group g + 1 while group g is on its
// way and being decoded), k_sigma_vprep runs over (element, client), and every client is ONE problem of a multi-problem Pippenger launch --
// its own random linear combination, hence its own verdict, with no closer look needed (clients share nothing here, unlike the range
// proofs' generator MSM).  55 000 elements alone are less than one wave per SIMD; 48 clients fill the chip.
// csq_sum_out (kinds 1, 2; may be null): sum_i c_sq_i of every client, compressed (params.rs:220, 277) -- the points are decoded here anyway.
// `single`: the call is one of the rofl_verify_*_vec entry points (a malformed vector is the call's FormatError); batch calls give the
// offender ok = 0 and go on.
int sigma_verify_batch(int kind, size_t nc, const uint8_t *const *proofs, const uint8_t *const *commits, size_t d, int *ok_out, uint8_t *csq_sum_out, bool single) {
    LaneLock lane_lock = acquire_lane(false, nc == 1); Ctx &C = *lane_lock.c;
    for (size_t i = 0; i < nc; i++) ok_out[i] = 0;
    const bool has_R = kind != 2, has_sq = kind != 0;
    if (csq_sum_out) memset(csq_sum_out, 0, 32 * nc);      // the identity (also the sum of an empty vector)
    if (nc == 0) return ROFL_OK;
    if (d == 0) { for (size_t i = 0; i < nc; i++) ok_out[i] = 1; return ROFL_OK; }
    const size_t npts = 1 + (has_R ? 1 : 0) + (has_sq ? 1 : 0), clen = 32 * npts, plen = 32 * (npts + (has_sq ? 3 : 2));
    const size_t nslots = 2 * npts, nblk = (d + TPB - 1) / TPB;
    // (the members of a batch index gridDim.y of the decode / transcript / sum kernels and the problems of the Pippenger launches: 65 535 at most)
    if (nc > kMaxBatchMembers || nc * nslots * d >= ((size_t)1 << 31) || nc * nblk > ((size_t)1 << 22)) return fail(ROFL_BAD_PARAM, "batch too large (split it)");
    C.init();
    C.batch_mode = nc > 1;
    timing_begin(C);
    if (!opts().sigma_batch.load()) {      // rofl_set_option("sigma_batch", 0): one check per element (the reference's form), client by client
        int rc_all = ROFL_OK;
        for (size_t i = 0; i < nc; i++) {
            uint8_t *dp = C.aux_pts.as<uint8_t>(d * plen), *dc = C.aux_scal.as<uint8_t>(d * clen);
            u32 *status = C.status.as<u32>(4);
            HIPCHK(hipMemsetAsync(status, 0, 16, C.stream));
            C.up(dp, proofs[i], d * plen, C.stream); C.up(dc, commits[i], d * clen, C.stream);
            { KSpan ks_sigma(C.tm, C.stream, ROFL_TK_SIGMA, (uint64_t)d * (2 * npts * 265 + (kind ? 3 : 2) * 2 * 325 * 8), (uint64_t)d * (clen + plen));
              ROFL_LAUNCH(k_sigma_verify, dim3((unsigned)((d + 63) / 64)), dim3(64), 0, C.stream, kind, (u32)d, dp, dc, sigma_init_state(kind), C.d_tabB, C.d_tabBb, status + 1, status); }
            u32 *st = C.h_misc.as<u32>(4);
            HIPCHK(hipMemcpyAsync(st, status, 8, hipMemcpyDeviceToHost, C.stream));
            if (csq_sum_out && has_sq) {
                u32 nb2 = (u32)std::min<size_t>(64, nblk); ge *part = C.partial2.as<ge>(nb2);
                ROFL_LAUNCH(k_decode_sum, dim3(nb2), dim3(TPB), TPB * sizeof(ge), C.stream, dc + (has_R ? 64 : 32), (u32)d, (u32)clen, part, status + 2);
                ge *hp = C.h_part.as<ge>(nb2);
                HIPCHK(hipMemcpyAsync(hp, part, sizeof(ge) * nb2, hipMemcpyDeviceToHost, C.stream));
                C.sync();
                ge5 acc = h51::identity(); for (u32 k = 0; k < nb2; k++) acc = h51::gadd(acc, h51::from_ge(hp[k]));
                h51::encode(csq_sum_out + 32 * i, acc);
            } else C.sync();
            if (st[0] & 4u) { if (single) { timing_end(C); return fail(ROFL_FORMAT_ERROR, "FormatError: non-canonical scalar or invalid point"); } if (csq_sum_out) memset(csq_sum_out + 32 * i, 0, 32); continue; }
            ok_out[i] = st[1] == 0;
        }
        timing_end(C);
        return rc_all;
    }
    static const bool strace = knob("ROFL_TRACE") && atoi(knob("ROFL_TRACE")) >= 2;
    double st0 = now_ms(), stl = st0;
    auto smark = [&](const char *what) { if (!strace) return; double t = now_ms(); fprintf(stderr, "[rofl-trace sigma-verify] %-18s +%.3f ms  (t=%.3f)\n", what, t - stl, t - st0); stl = t; };
    uint8_t *dp = C.aux_pts.as<uint8_t>(nc * d * plen), *dc = C.aux_scal.as<uint8_t>(nc * d * clen);
    u32 *status = C.status.as<u32>(nc + 4);
    HIPCHK(hipMemsetAsync(status, 0, 4 * (nc + 4), C.stream));
    NonceSeed ws{};
    { FILE *f = fopen("/dev/urandom", "rb"); bool got = f && fread(ws.w, 1, 32, f) == 32; if (f) fclose(f);
      if (!got) return fail(ROFL_HIP_ERROR, "no randomness for the batched Sigma-proof check"); }
    niels *pts = C.gbuf[0].as<niels>(nc * nslots * d);
    sc *scal = C.SL.as<sc>(nc * nslots * d);
    sc *d_fixed = C.tmp_out.as<sc>(nc * nblk * 2);
    const DMerlin init = sigma_init_state(kind);
    // the weights' size: their top bit must not be the top bit of a window of the layout the MSM will use (see k_sigma_vprep)
    static const u32 wbits = [] {      // (whichever of the generic layouts the MSM driver ends up with, retries included)
        for (u32 cand = 127; cand > 96; cand--) {
            bool hit = false;
            for (u32 c : {4u, 7u, 10u, 13u, 16u}) { MsmPlan mp = msm_plan_c(c); for (u32 w = 0, end = 0; w < mp.W; w++) { end += w + 1 == mp.W ? mp.c + 1 : (w < mp.wide ? mp.c : mp.c - 1); hit |= end == cand; } }
            if (!hit) return cand;
        }
        return 96u; }();
    // groups of clients: ~64 MB of caller bytes each, through two staging buffers
    const size_t per = d * (plen + clen);
    const size_t G = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(nc, 16), ((size_t)64 << 20) / per));      // (at most sixteen: a group is also a row count of gridDim.y and part of one Pippenger launch)
    bool all_host = true; for (size_t i = 0; i < nc; i++) all_host &= !is_device_ptr(proofs[i]) && !is_device_ptr(commits[i]);
    // three staging buffers; the uploads run on a stream of their own (the copy of group g + 1 beside the kernels of group g: on one stream they
    // alternated, and the "staging" time of the first version was the host waiting for that stream)
    constexpr size_t NST = 3;
    uint8_t *stage[NST] = {nullptr, nullptr, nullptr}; bool used[NST] = {false, false, false};
    if (all_host && per >= Stage::kMin) for (size_t b = 0; b < NST && b * G < nc; b++) stage[b] = (uint8_t *)C.stg.alloc(G * per);
    if (stage[0] && !C.stream_up) HIPCHK(hipStreamCreateWithFlags(&C.stream_up, hipStreamNonBlocking));
    struct JoinUp { hipStream_t s; ~JoinUp() { if (s) (void)hipStreamSynchronize(s); } } join_up{stage[0] ? C.stream_up : nullptr};
    // events of the call: 0..2 upload of staging buffer b done, 3 / 4 a launch's clients decoded, 5 / 6 a launch finished
    if (stage[0]) { HIPCHK(hipEventRecord(C.pool_event(7), C.stream)); HIPCHK(hipStreamWaitEvent(C.stream_up, C.pool_event(7), 0)); }      // (after the status memset)
    // Every client is one problem of a multi-problem Pippenger launch; launches of 16 to 31 clients (whole decode groups; the slot array of a launch grows with
    // its problems) go to the SIDE stream as soon as their clients are decoded, so that they run while the host is still staging and the main
    // stream still uploading and decoding the later clients.  Two MSM workspaces alternate: launch k + 2 waits for launch k's results.
    if (!C.stream2) HIPCHK(hipStreamCreateWithFlags(&C.stream2, hipStreamNonBlocking));
    struct Join { hipStream_t s; ~Join() { (void)hipStreamSynchronize(s); } } join{C.stream2};      // nothing of the side stream outlives the call, error paths included
    struct Job { size_t c0, cnt; MsmJob J; MsmAllow al; std::vector<MsmProb> pr; std::vector<ge5> res; bool done = false, redo = false; };
    std::vector<std::unique_ptr<Job>> jobs;
    const size_t SGC = 16;
    auto finish_job = [&](Job &jb, size_t k) {
        C.wait_event(C.pool_event(5 + (k & 1)));
        if (msm_retry(jb.J, jb.al)) jb.redo = true;      // a fixed-size structure overflowed (scalars built to collide): repeated on its own after the pipeline
        else msm_finish(C, jb.J, jb.res, MsmOpt());
        jb.done = true;
    };
    auto launch_job = [&](size_t c0, size_t cnt) {
        const size_t k = jobs.size();
        if (k >= 2 && !jobs[k - 2]->done) finish_job(*jobs[k - 2], k - 2);      // its workspace is taken over
        std::unique_ptr<Job> jb(new Job()); jb->c0 = c0; jb->cnt = cnt; jb->pr.resize(cnt);
        for (size_t q = 0; q < cnt; q++) jb->pr[q] = MsmProb{pts + (c0 + q) * nslots * d, scal + (c0 + q) * nslots * d};
        C.tm.t.msm_terms += cnt * nslots * d;
        HIPCHK(hipEventRecord(C.pool_event(3 + (k & 1)), C.stream));                 // the clients of this launch are decoded, their scalars written
        HIPCHK(hipStreamWaitEvent(C.stream2, C.pool_event(3 + (k & 1)), 0));
        jb->J = msm_enqueue(C, C.mws[k & 1], jb->pr, nslots * d, MsmOpt(), jb->al, C.stream2);
        HIPCHK(hipEventRecord(C.pool_event(5 + (k & 1)), C.stream2));
        jobs.push_back(std::move(jb));
    };
    size_t sg_start = 0;
    for (size_t g0 = 0, gi = 0; g0 < nc; g0 += G, gi++) {
        const size_t gc = std::min(G, nc - g0), b = gi % NST;
        if (stage[0]) {
            if (used[b]) C.wait_event(C.pool_event(b));      // the upload that read this buffer three groups ago
            uint8_t *sp = stage[b], *sq = stage[b] + gc * d * plen;
            const size_t sl_p = std::max<size_t>(1, (d * plen) >> 18), sl_c = std::max<size_t>(1, (d * clen) >> 18);      // ~256 KB per task
            C.pool->run(gc * (sl_p + sl_c), [&](size_t t) {
                size_t i = t / (sl_p + sl_c), k = t % (sl_p + sl_c);
                if (k < sl_p) { size_t lo = d * plen * k / sl_p, hi = d * plen * (k + 1) / sl_p; stage_copy(sp + i * d * plen + lo, proofs[g0 + i] + lo, hi - lo); }
                else { k -= sl_p; size_t lo = d * clen * k / sl_c, hi = d * clen * (k + 1) / sl_c; stage_copy(sq + i * d * clen + lo, commits[g0 + i] + lo, hi - lo); }
            });
            HIPCHK(hipMemcpyAsync(dp + g0 * d * plen, sp, gc * d * plen, hipMemcpyHostToDevice, C.stream_up));
            HIPCHK(hipMemcpyAsync(dc + g0 * d * clen, sq, gc * d * clen, hipMemcpyHostToDevice, C.stream_up));
            HIPCHK(hipEventRecord(C.pool_event(b), C.stream_up)); used[b] = true;
            HIPCHK(hipStreamWaitEvent(C.stream, C.pool_event(b), 0));      // this group's kernels wait for its bytes
        } else
            for (size_t i = g0; i < g0 + gc; i++) { C.up(dp + i * d * plen, proofs[i], d * plen, C.stream); C.up(dc + i * d * clen, commits[i], d * clen, C.stream); }
        {   KSpan ks_sigma(C.tm, C.stream, ROFL_TK_SIGMA, (uint64_t)gc * d * (2 * npts * 265), (uint64_t)gc * d * (clen + plen));      // decoding of 2 npts points per element
            ROFL_LAUNCH(k_sigma_vdecode, dim3((unsigned)((nslots * d + TPB - 1) / TPB), (unsigned)gc), dim3(TPB), 0, C.stream, kind, (u32)d, dp + g0 * d * plen, dc + g0 * d * clen,
                               pts + g0 * nslots * d, status + g0);
            ROFL_LAUNCH(k_sigma_vprep, dim3((unsigned)nblk, (unsigned)gc), dim3(TPB), 0, C.stream, kind, (u32)d, dp + g0 * d * plen, dc + g0 * d * clen, init, ws, (u64)(g0 * d), wbits,
                               scal + g0 * nslots * d, d_fixed + g0 * nblk * 2, status + g0); }
        if (g0 + gc - sg_start >= SGC || g0 + gc == nc) { launch_job(sg_start, g0 + gc - sg_start); sg_start = g0 + gc; }
    }
    smark("staged + enqueued");
    sc *h_fixed = C.h_part.as<sc>(nc * nblk * 2);
    u32 *h_st = C.h_misc.as<u32>(nc + 4);
    HIPCHK(hipMemcpyAsync(h_fixed, d_fixed, sizeof(sc) * nc * nblk * 2, hipMemcpyDeviceToHost, C.stream));
    HIPCHK(hipMemcpyAsync(h_st, status, 4 * nc, hipMemcpyDeviceToHost, C.stream));
    const u32 nb2 = (u32)std::min<size_t>(64, nblk);
    ge *h_csq = nullptr;
    if (csq_sum_out && has_sq) {
        ge *part = C.partial2.as<ge>(nc * nb2);
        ROFL_LAUNCH(k_niels_sum, dim3(nb2, (unsigned)nc), dim3(TPB), TPB * sizeof(ge), C.stream, (const niels *)(pts + (nslots - 2) * d), (u32)d, nslots * d, part);
        h_csq = C.h_misc2.as<ge>(nc * nb2);
        HIPCHK(hipMemcpyAsync(h_csq, part, sizeof(ge) * nc * nb2, hipMemcpyDeviceToHost, C.stream));
    }
    C.sync();
    smark("decoded");
    for (size_t k = 0; k < jobs.size(); k++) if (!jobs[k]->done) finish_job(*jobs[k], k);
    HIPCHK(hipStreamSynchronize(C.stream2));
    for (auto &jb : jobs) if (jb->redo) { jb->res.clear(); msm_run(C, jb->pr, nslots * d, jb->res); }
    std::vector<size_t> good;
    std::vector<char> is_bad(nc, 0);
    for (size_t i = 0; i < nc; i++) {
        if (h_st[i] & 4u) { is_bad[i] = 1; if (single) { timing_end(C); return fail(ROFL_FORMAT_ERROR, "FormatError: non-canonical scalar or invalid point"); } }
        else good.push_back(i);
    }
    for (auto &jb : jobs)
        for (size_t q = 0; q < jb->cnt; q++) {
            const size_t i = jb->c0 + q;
            if (is_bad[i]) continue;      // (its problem ran with the others -- the scalars of a malformed member are still well-formed numbers -- and its result is ignored)
            sc sB = h_canon(sum_partials(h_fixed + i * nblk * 2, nblk, 2, 0)), sBb = h_canon(sum_partials(h_fixed + i * nblk * 2, nblk, 2, 1));
            ge5 tot = h51::gadd(jb->res[q], h51::gadd(h_fixed_mul(C.ht.B5, sB), h_fixed_mul(C.ht.Bb5, sBb)));
            ok_out[i] = h51::is_identity_ristretto(tot) ? 1 : 0;
        }
    smark("msm + verdicts");
    if (h_csq) for (size_t i : good) {
        ge5 acc = h51::identity(); for (u32 k = 0; k < nb2; k++) acc = h51::gadd(acc, h51::from_ge(h_csq[i * nb2 + k]));
        h51::encode(csq_sum_out + 32 * i, acc);
    }
    timing_end(C);
    return ROFL_OK;
}
int sigma_verify(int kind, const uint8_t *proofs, const uint8_t *commits, size_t d, int *ok_out) {
    return sigma_verify_batch(kind, 1, &proofs, &commits, d, ok_out, nullptr, true);
}
// ---- compressed_rand_proof
sc compressed_challenge(const uint8_t *pairs, size_t d, const uint8_t cprime[64]) {
    Merlin t("CompressedRandProof", 19);
    t.append("dom-sep", (const uint8_t *)"randomness proof v1", 19);
    for (size_t i = 0; i < d; i++) {      // label = UNIQUE_U8_TRIPLETS[i] (generate_unique_u8_triplets.py:8-13)
        uint8_t lbl[3] = {(uint8_t)(3 * i), (uint8_t)(3 * i + 1), (uint8_t)(3 * i + 2)};
        t.append_lbl(lbl, 3, pairs + 64 * i, 64);
    }
    t.append("C_prime_eg", cprime, 64);
    return t.challenge_scalar("c");
}
int compressed_create(const float *values, size_t d, const uint8_t *r32, size_t d_r, const uint8_t *existing, unsigned fp_bits, unsigned fp_frac,
                      const rofl_nonce_t *nonce, uint8_t *proof_out, uint8_t *pairs_out) {
    LaneLock lane_lock = acquire_lane(false, true); Ctx &C = *lane_lock.c;
    if (d != d_r) return fail(ROFL_WRONG_NUM_BLINDING, "WrongNumBlindingFactors");
    if (!valid_fp(fp_bits, fp_frac) || !nonce || d >= 900000) return fail(ROFL_BAD_PARAM, "bad parameter");
    if (nonce->mode == 0 && nonce->stream_scalars < 2) return fail(ROFL_NONCE_SHORT, "nonce stream too short");
    C.init();
    timing_begin(C);
    size_t dd = d ? d : 1;
    float *dv = C.vals.as<float>(dd); sc *dr = C.tmp_in.as<sc>(dd);
    uint8_t *dex = existing ? C.Cbytes.as<uint8_t>(dd * 32) : nullptr;
    uint8_t *dpairs = C.aux_scal.as<uint8_t>(dd * 64);
    u32 *status = C.status.as<u32>(4);
    HIPCHK(hipMemsetAsync(status, 0, 16, C.stream));
    const uint8_t *pairs_host = pairs_out;      // where the pairs can be read on the host after the wait (the staging copy of a large transfer)
    if (d) {
        C.up(dv, values, 4 * d, C.stream);
        C.up(dr, r32, 32 * d, C.stream);
        if (dex) C.up(dex, existing, 32 * d, C.stream);
        ROFL_LAUNCH(k_eg_pairs, dim3((unsigned)((d + 63) / 64)), dim3(64), 0, C.stream, (u32)d, dv, fp_bits, fp_frac, dr, dex, C.d_tabB8, C.d_tabBb8, dpairs, status);
        pairs_host = (const uint8_t *)C.down(pairs_out, dpairs, 64 * d, C.stream);
    }
    u32 st = 0;
    HIPCHK(hipMemcpyAsync(&st, status, 4, hipMemcpyDeviceToHost, C.stream));
    C.sync();
    if (st & 2u) return fail(ROFL_NON_FINITE, "non-finite value (the reference panics in fixed::saturating_from_float)");
    if (st & 4u) return fail(ROFL_FORMAT_ERROR, "invalid Ristretto encoding");
    // nonces m', r' (party.rs:66-67), C' = commit(m', r')
    sc nc[2];
    for (int j = 0; j < 2; j++) {
        uint8_t b[32];
        if (nonce->mode == 1) rofl_dbg_host_nonce(nonce->seed, (uint64_t)j, b);
        else { sc w = sc_from_wide(sc_frombytes(nonce->stream + 64 * j), sc_frombytes(nonce->stream + 64 * j + 32)); sc_tobytes(b, w); }
        nc[j] = sc_frombytes(b);
    }
    h51::encode(proof_out, h51::gadd(h_fixed_mul(C.ht.B5, nc[0]), h_fixed_mul(C.ht.Bb5, nc[1])));
    h51::encode(proof_out + 32, h_fixed_mul(C.ht.B5, nc[1]));
    sc c = compressed_challenge(pairs_host, d, proof_out);
    sc zm = nc[0], zr = nc[1];
    if (d) {
        CPow cp; fill_pow2(cp.sq, h_mont(c), MAX_LG);
        u32 nblk = (u32)std::min<size_t>(64, (d + TPB - 1) / TPB);
        sc *part = C.tmp_out.as<sc>(64 * 2);
        ROFL_LAUNCH(k_cpow_dot, dim3(nblk), dim3(TPB), 0, C.stream, (u32)d, dv, fp_bits, fp_frac, dr, cp, part);
        sc hp[128];
        HIPCHK(hipMemcpyAsync(hp, part, sizeof(sc) * nblk * 2, hipMemcpyDeviceToHost, C.stream));
        C.sync();
        zm = sc_add(zm, h_canon(sum_partials(hp, nblk, 2, 0))); zr = sc_add(zr, h_canon(sum_partials(hp, nblk, 2, 1)));
    }
    sc_tobytes(proof_out + 64, zm); sc_tobytes(proof_out + 96, zr);
    timing_end(C);
    return ROFL_OK;
}
int compressed_verify(const uint8_t *proof, const uint8_t *pairs, size_t d, int *ok_out) {
    LaneLock lane_lock = acquire_lane(false, true); Ctx &C = *lane_lock.c;
    *ok_out = 0;
    ge Lp32, Rp32;
    if (!ristretto_decode(Lp32, proof) || !ristretto_decode(Rp32, proof + 32) || !sc_is_canonical_bytes(proof + 64) || !sc_is_canonical_bytes(proof + 96))
        return fail(ROFL_FORMAT_ERROR, "FormatError");
    if (d >= 900000) return fail(ROFL_BAD_PARAM, "bad parameter");
    C.init();
    timing_begin(C);
    ge5 sumL = h51::identity(), sumR = h51::identity();
    sc c = compressed_challenge(pairs, d, proof);
    if (d) {
        uint8_t *dpairs = C.tmp_in.as<uint8_t>(d * 64);
        niels *pts = C.aux_pts.as<niels>(2 * d); sc *scal = C.aux_scal.as<sc>(d);
        u32 *status = C.status.as<u32>(4);
        HIPCHK(hipMemsetAsync(status, 0, 16, C.stream));
        C.up(dpairs, pairs, 64 * d, C.stream);
        ROFL_LAUNCH(k_decode_pairs, grid1(2 * d), dim3(TPB), 0, C.stream, (u32)d, dpairs, pts, pts + d, status);
        CPow cp; fill_pow2(cp.sq, h_mont(c), MAX_LG);
        ROFL_LAUNCH(k_cpow_scalars, grid1(d), dim3(TPB), 0, C.stream, (u32)d, cp, scal);
        u32 st = 0;
        HIPCHK(hipMemcpyAsync(&st, status, 4, hipMemcpyDeviceToHost, C.stream));
        C.sync();
        if (st & 4u) return fail(ROFL_FORMAT_ERROR, "FormatError: invalid ElGamal pair");
        std::vector<MsmProb> pr = {MsmProb{pts, scal}, MsmProb{pts + d, scal}}; std::vector<ge5> res;
        msm_run(C, pr, d, res);          // sum_i c^(i+1) L_i and sum_i c^(i+1) R_i share the scalars
        sumL = res[0]; sumR = res[1];
    }
    sc zm = sc_frombytes(proof + 64), zr = sc_frombytes(proof + 96);
    auto neg5 = [](const ge5 &p) { ge5 r = p; r.X = h51::neg(p.X); r.T = h51::neg(p.T); return r; };
    ge5 e1 = h51::gadd(h51::gadd(h_fixed_mul(C.ht.B5, zm), h_fixed_mul(C.ht.Bb5, zr)), neg5(h51::gadd(h51::from_ge(Lp32), sumL)));
    ge5 e2 = h51::gadd(h_fixed_mul(C.ht.B5, zr), neg5(h51::gadd(h51::from_ge(Rp32), sumR)));
    *ok_out = h51::is_identity_ristretto(e1) && h51::is_identity_ristretto(e2);
    timing_end(C);
    return ROFL_OK;
}
}  // namespace

namespace {
// ONE vector of per-element Sigma-proofs over several devices: contiguous runs of elements (rofl_set_option("devices", mask)), as the single-client
// range-proof calls deal chunks.  Runs shorter than a few thousand elements are not worth a device.
std::vector<std::pair<size_t, size_t>> elem_runs(size_t d, size_t nd) {
    std::vector<std::pair<size_t, size_t>> r;
    nd = std::max<size_t>(1, std::min(nd, d / 2048));
    for (size_t k = 0; k < nd; k++) { size_t a = k * d / nd, b = (k + 1) * d / nd; if (b > a) r.emplace_back(a, b - a); }
    return r;
}
int sigma_create_any(int kind, const float *values, size_t d, const uint8_t *r1, size_t d_r1, const uint8_t *r2, const uint8_t *existing, unsigned fp_bits, unsigned fp_frac,
                     const rofl_nonce_t *nonce, uint8_t *proofs_out, uint8_t *commits_out) {
    std::vector<int> devs = batch_devices();
    auto runs = elem_runs(d, devs.size());
    if (devs.size() < 2 || runs.size() < 2 || d != d_r1 || !values || !r1 || !nonce || !proofs_out || !commits_out || (kind != 0 && !r2)) {
        std::unique_ptr<DeviceBinding> bind; if (devs.size() == 1) bind.reset(new DeviceBinding(devs[0]));
        return sigma_create(kind, values, d, r1, d_r1, r2, existing, fp_bits, fp_frac, nonce, proofs_out, commits_out);
    }
    const bool has_sq = kind != 0;
    const size_t npts = 1 + (kind != 2) + (has_sq ? 1 : 0), nn = has_sq ? 3 : 2, clen = 32 * npts, plen = 32 * (npts + nn);
    std::vector<int> rcs; std::vector<std::string> errs;
    if (int rc = run_on_devices(devs, runs.size(), rcs, errs, [&](size_t k) -> int {
            const size_t e0 = runs[k].first, cnt = runs[k].second;
            return sigma_create(kind, values + e0, cnt, r1 + 32 * e0, cnt, r2 ? r2 + 32 * e0 : nullptr, existing ? existing + 32 * e0 : nullptr, fp_bits, fp_frac, nonce,
                                proofs_out + e0 * plen, commits_out + e0 * clen, e0, d); }))
        return rc;
    // the unsplit call's outcome: the NaN check comes before anything is decoded (k_sigma_finish / k_sigma_prove report 2 before 4 per element; across
    // elements the call reports NON_FINITE first, sigma_create)
    for (size_t k = 0; k < rcs.size(); k++) if (rcs[k] >= ROFL_HIP_ERROR || rcs[k] == ROFL_BAD_PARAM || rcs[k] == ROFL_NONCE_SHORT) return fail(rcs[k], errs[k]);
    for (size_t k = 0; k < rcs.size(); k++) if (rcs[k] == ROFL_NON_FINITE) return fail(rcs[k], errs[k]);
    for (size_t k = 0; k < rcs.size(); k++) if (rcs[k]) return fail(rcs[k], errs[k]);
    return ROFL_OK;
}
int sigma_verify_any(int kind, const uint8_t *proofs, const uint8_t *commits, size_t d, int *ok_out) {
    std::vector<int> devs = batch_devices();
    auto runs = elem_runs(d, devs.size());
    if (devs.size() < 2 || runs.size() < 2 || !proofs || !commits || !ok_out) {
        std::unique_ptr<DeviceBinding> bind; if (devs.size() == 1) bind.reset(new DeviceBinding(devs[0]));
        return sigma_verify_batch(kind, 1, &proofs, &commits, d, ok_out, nullptr, true);
    }
    const bool has_sq = kind != 0;
    const size_t npts = 1 + (kind != 2) + (has_sq ? 1 : 0), nn = has_sq ? 3 : 2, clen = 32 * npts, plen = 32 * (npts + nn);
    std::vector<int> rcs, oks(runs.size(), 0); std::vector<std::string> errs;
    if (int rc = run_on_devices(devs, runs.size(), rcs, errs, [&](size_t k) -> int {
            const uint8_t *pp = proofs + runs[k].first * plen, *cc = commits + runs[k].first * clen;
            return sigma_verify_batch(kind, 1, &pp, &cc, runs[k].second, &oks[k], nullptr, true); }))      // every run is its own random linear combination
        return rc;
    *ok_out = 0;
    for (size_t k = 0; k < rcs.size(); k++) if (rcs[k]) return fail(rcs[k], errs[k]);      // a malformed vector is the call's FormatError, whichever run met it
    int ok = 1; for (int o : oks) ok &= o;
    *ok_out = ok;
    return ROFL_OK;
}
}  // namespace
int rofl_create_sigmaproof_vec_range(int kind, const float *values, size_t d, const uint8_t *r1_32, const uint8_t *r2_32, const uint8_t *existing32, unsigned fp_bits, unsigned fp_frac,
                                     const rofl_nonce_t *nonce, size_t elem_first, size_t elem_count, uint8_t *proofs_out, uint8_t *commits_out) {
    return guarded([&]() -> int {
        if (kind < 0 || kind > 2 || !values || !r1_32 || !nonce || !proofs_out || !commits_out || (kind != 0 && !r2_32) || elem_first > d || elem_count > d - elem_first)
            return fail(ROFL_BAD_PARAM, "bad parameter");
        return sigma_create(kind, values + elem_first, elem_count, r1_32 + 32 * elem_first, elem_count, r2_32 ? r2_32 + 32 * elem_first : nullptr,
                            existing32 ? existing32 + 32 * elem_first : nullptr, fp_bits, fp_frac, nonce, proofs_out, commits_out, elem_first, d); });
}
int rofl_create_squareproof_vec(const float *values, size_t d, const uint8_t *r1_32, size_t d_r1, const uint8_t *r2_32, const uint8_t *existing32,
                                unsigned fp_bits, unsigned fp_frac, const rofl_nonce_t *nonce, uint8_t *proofs_out, uint8_t *commits_out) {
    return guarded([&]() -> int { return sigma_create_any(2, values, d, r1_32, d_r1, r2_32, existing32, fp_bits, fp_frac, nonce, proofs_out, commits_out); });
}
int rofl_verify_squareproof_vec(const uint8_t *proofs, const uint8_t *commits, size_t d, int *ok_out) {
    return guarded([&]() -> int { return sigma_verify_any(2, proofs, commits, d, ok_out); });
}
int rofl_create_compressed_randproof(const float *values, size_t d, const uint8_t *r32, size_t d_r, const uint8_t *existing32, unsigned fp_bits, unsigned fp_frac,
                                     const rofl_nonce_t *nonce, uint8_t proof_out[128], uint8_t *pairs_out) {
    return guarded([&]() -> int { return compressed_create(values, d, r32, d_r, existing32, fp_bits, fp_frac, nonce, proof_out, pairs_out); });
}
int rofl_verify_compressed_randproof(const uint8_t proof[128], const uint8_t *pairs, size_t d, int *ok_out) {
    return guarded([&]() -> int { return compressed_verify(proof, pairs, d, ok_out); });
}
int rofl_create_randproof_vec(const float *values, size_t d, const uint8_t *r32, size_t d_r, const uint8_t *existing32, unsigned fp_bits, unsigned fp_frac,
                              const rofl_nonce_t *nonce, uint8_t *proofs_out, uint8_t *commits_out) {
    return guarded([&]() -> int { return sigma_create_any(0, values, d, r32, d_r, nullptr, existing32, fp_bits, fp_frac, nonce, proofs_out, commits_out); });
}
int rofl_verify_randproof_vec(const uint8_t *proofs, const uint8_t *commits, size_t d, int *ok_out) {
    return guarded([&]() -> int { return sigma_verify_any(0, proofs, commits, d, ok_out); });
}
int rofl_create_squarerandproof_vec(const float *values, size_t d, const uint8_t *r1_32, size_t d_r1, const uint8_t *r2_32, const uint8_t *existing32,
                                    unsigned fp_bits, unsigned fp_frac, const rofl_nonce_t *nonce, uint8_t *proofs_out, uint8_t *commits_out) {
    return guarded([&]() -> int { return sigma_create_any(1, values, d, r1_32, d_r1, r2_32, existing32, fp_bits, fp_frac, nonce, proofs_out, commits_out); });
}
int rofl_verify_squarerandproof_vec(const uint8_t *proofs, const uint8_t *commits, size_t d, int *ok_out) {
    return guarded([&]() -> int { return sigma_verify_any(1, proofs, commits, d, ok_out); });
}
namespace {
int sigma_batch_entry(int kind, size_t n_clients, const uint8_t *const *proofs, const uint8_t *const *commits, size_t d, int *ok_out, uint8_t *csq_sum_out32) {
    if (!ok_out || (n_clients && (!proofs || !commits))) return fail(ROFL_BAD_PARAM, "bad parameter");
    std::vector<int> devs = batch_devices();      // rofl_set_option("devices", mask): the clients are dealt round-robin to the listed devices, as in the range-proof batch calls
    if (devs.empty() || n_clients < 2)
        return guarded([&]() -> int { std::unique_ptr<DeviceBinding> bind; if (!devs.empty()) bind.reset(new DeviceBinding(devs[0]));
            return sigma_verify_batch(kind, n_clients, proofs, commits, d, ok_out, csq_sum_out32, false); });
    for (size_t i = 0; i < n_clients; i++) ok_out[i] = 0;
    return guarded([&]() -> int { return shard_over_devices(n_clients, devs, [&](const std::vector<size_t> &idx) -> int {
        const size_t k = idx.size();
        std::vector<const uint8_t *> p(k), c(k); std::vector<int> ok(k, 0); std::vector<uint8_t> sums(csq_sum_out32 ? 32 * k : 0);
        for (size_t j = 0; j < k; j++) { p[j] = proofs[idx[j]]; c[j] = commits[idx[j]]; }
        int r = sigma_verify_batch(kind, k, p.data(), c.data(), d, ok.data(), csq_sum_out32 ? sums.data() : nullptr, false);
        for (size_t j = 0; j < k; j++) { ok_out[idx[j]] = ok[j]; if (csq_sum_out32) memcpy(csq_sum_out32 + 32 * idx[j], &sums[32 * j], 32); }
        return r; }); });
}
}  // namespace
int rofl_verify_randproof_vec_batch(size_t n_clients, const uint8_t *const *proofs, const uint8_t *const *commits, size_t d, int *ok_out) {
    return sigma_batch_entry(0, n_clients, proofs, commits, d, ok_out, nullptr);
}
int rofl_verify_squarerandproof_vec_batch(size_t n_clients, const uint8_t *const *proofs, const uint8_t *const *commits, size_t d, int *ok_out, uint8_t *csq_sum_out32) {
    return sigma_batch_entry(1, n_clients, proofs, commits, d, ok_out, csq_sum_out32);
}
int rofl_verify_squareproof_vec_batch(size_t n_clients, const uint8_t *const *proofs, const uint8_t *const *commits, size_t d, int *ok_out, uint8_t *csq_sum_out32) {
    return sigma_batch_entry(2, n_clients, proofs, commits, d, ok_out, csq_sum_out32);
}

int rofl_commit_vec(const uint8_t *values32, const uint8_t *blindings32, size_t d, uint8_t *out32) {
    return guarded([&]() -> int {
        LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
        if (d == 0) return ROFL_OK;
        C.init();
        sc *dv = C.tmp_in.as<sc>(d); sc *db = blindings32 ? C.tmp_in2.as<sc>(d) : nullptr;
        uint8_t *o = C.Cbytes.as<uint8_t>(d * 32);
        C.up(dv, values32, 32 * d, C.stream);
        if (db) C.up(db, blindings32, 32 * d, C.stream);
        ROFL_LAUNCH(k_commit, grid1(d), dim3(TPB), 0, C.stream, (u32)d, (const u64 *)nullptr, dv, db, C.d_tabB8, C.d_tabBb8, (const niels *)nullptr, (uint8_t *)nullptr, o, (u32)d, (u32)d);
        C.down(out32, o, 32 * d, C.stream);
        C.sync();
        return ROFL_OK;
    });
}
int rofl_add_points_vec(const uint8_t *a32, const uint8_t *b32, size_t d, uint8_t *out32) {
    return guarded([&]() -> int {
        LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
        if (d == 0) return ROFL_OK;
        C.init();
        uint8_t *da = C.tmp_in.as<uint8_t>(d * 32), *db = C.tmp_in2.as<uint8_t>(d * 32), *o = C.Cbytes.as<uint8_t>(d * 32);
        u32 *status = C.status.as<u32>(4);
        HIPCHK(hipMemsetAsync(status, 0, 16, C.stream));
        C.up(da, a32, 32 * d, C.stream);
        C.up(db, b32, 32 * d, C.stream);
        ROFL_LAUNCH(k_add_points, grid1(d), dim3(TPB), 0, C.stream, (u32)d, da, db, o, status);
        u32 st = 0;
        C.down(out32, o, 32 * d, C.stream);
        HIPCHK(hipMemcpyAsync(&st, status, 4, hipMemcpyDeviceToHost, C.stream));
        C.sync();
        if (st & 4u) return fail(ROFL_FORMAT_ERROR, "invalid Ristretto encoding");
        return ROFL_OK;
    });
}
int rofl_sum_points(const uint8_t *points, size_t d, size_t stride, uint8_t out32[32]) {
    return guarded([&]() -> int {
        LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
        if (stride < 32 || !out32 || (d && !points)) return fail(ROFL_BAD_PARAM, "bad parameter");
        if (d == 0) { memset(out32, 0, 32); return ROFL_OK; }      // empty sum = identity
        C.init();
        uint8_t *da = C.tmp_in.as<uint8_t>(d * stride);
        u32 nblk = (u32)std::min<size_t>(64, (d + TPB - 1) / TPB);
        ge *part = C.partial2.as<ge>(nblk);
        u32 *status = C.status.as<u32>(4);
        HIPCHK(hipMemsetAsync(status, 0, 16, C.stream));
        C.up(da, points, stride * d, C.stream);
        ROFL_LAUNCH(k_decode_sum, dim3(nblk), dim3(TPB), TPB * sizeof(ge), C.stream, da, (u32)d, (u32)stride, part, status);
        std::vector<ge> hp(nblk); u32 st = 0;
        HIPCHK(hipMemcpyAsync(hp.data(), part, sizeof(ge) * nblk, hipMemcpyDeviceToHost, C.stream));
        HIPCHK(hipMemcpyAsync(&st, status, 4, hipMemcpyDeviceToHost, C.stream));
        C.sync();
        if (st & 4u) return fail(ROFL_FORMAT_ERROR, "invalid Ristretto encoding");
        ge5 acc = h51::identity();
        for (u32 i = 0; i < nblk; i++) acc = h51::gadd(acc, h51::from_ge(hp[i]));
        h51::encode(out32, acc);
        return ROFL_OK;
    });
}
int rofl_shift_points(const uint8_t *a32, size_t d, const uint8_t offset32[32], uint8_t *out32) {
    return guarded([&]() -> int {
        LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
        if (d == 0) return ROFL_OK;
        C.init();
        ge off; if (!ristretto_decode(off, offset32)) return fail(ROFL_FORMAT_ERROR, "invalid Ristretto encoding");
        niels hs = ge_to_niels(off);
        niels *ds = C.tmp_in2.as<niels>(1);
        uint8_t *da = C.tmp_in.as<uint8_t>(d * 32), *o = C.Cbytes.as<uint8_t>(d * 32);
        u32 *status = C.status.as<u32>(4);
        HIPCHK(hipMemsetAsync(status, 0, 16, C.stream));
        HIPCHK(hipMemcpyAsync(ds, &hs, sizeof hs, hipMemcpyHostToDevice, C.stream));
        C.up(da, a32, 32 * d, C.stream);
        ROFL_LAUNCH(k_decode, grid1(d), dim3(TPB), 0, C.stream, (u32)d, (u32)d, da, ds, (niels *)nullptr, o, status);
        u32 st = 0;
        C.down(out32, o, 32 * d, C.stream);
        HIPCHK(hipMemcpyAsync(&st, status, 4, hipMemcpyDeviceToHost, C.stream));
        C.sync();
        if (st & 4u) return fail(ROFL_FORMAT_ERROR, "invalid Ristretto encoding");
        return ROFL_OK;
    });
}
int rofl_f32_to_scalar_vec(const float *in, size_t d, unsigned fp_bits, unsigned fp_frac, uint8_t *out32) {
    if (!valid_fp(fp_bits, fp_frac)) return fail(ROFL_BAD_PARAM, "bad parameter");
    for (size_t i = 0; i < d; i++) { sc s; int rc = f32_to_sc(in[i], fp_bits, fp_frac, &s); if (rc) return fail(rc, "non-finite value"); sc_tobytes(out32 + 32 * i, s); }
    return ROFL_OK;
}
int rofl_scalar_to_f32_vec(const uint8_t *in32, size_t d, unsigned fp_bits, unsigned fp_frac, float *out) {
    if (!valid_fp(fp_bits, fp_frac)) return fail(ROFL_BAD_PARAM, "bad parameter");
    for (size_t i = 0; i < d; i++) out[i] = sc_to_f32(sc_frombytes(in32 + 32 * i), fp_bits, fp_frac);
    return ROFL_OK;
}
// conversion32::square (conversion32.rs:66-88): |s| as Fix, checked_mul (fixed 0.3.3: wide product >> frac, truncated), panic on overflow
int rofl_fp_square_vec(const uint8_t *in32, size_t d, unsigned fp_bits, unsigned fp_frac, uint8_t *out32) {
    if (!valid_fp(fp_bits, fp_frac)) return fail(ROFL_BAD_PARAM, "bad parameter");
    for (size_t i = 0; i < d; i++) {
        sc s = sc_frombytes(in32 + 32 * i);
        u64 v = (s.v[7] >> 24) != 0 ? read_from_bytes(sc_neg(s), fp_bits) : read_from_bytes(s, fp_bits);
        unsigned __int128 prod = ((unsigned __int128)v * v) >> fp_frac;
        if (prod > (unsigned __int128)fix_max_bits(fp_bits)) return fail(ROFL_OVERFLOW, "square overflows the fixed-point type (the reference panics)");
        sc_tobytes(out32 + 32 * i, sc_from_u64((u64)prod));
    }
    return ROFL_OK;
}
// conversion32::precompute_exponentiate (conversion32.rs:101-111): 1, v, v^2, ..., v^(count-1); exponentiate(v, e) = element e
int rofl_scalar_powers(const uint8_t value32[32], size_t count, uint8_t *out32) {
    sc v = h_mont(sc_frombytes(value32)), acc = sc_one_mont();
    for (size_t i = 0; i < count; i++) { sc_tobytes(out32 + 32 * i, h_canon(acc)); acc = sc_montmul(acc, v); }
    return ROFL_OK;
}
// pedersen_ops::add_scalar_vec (pedersen_ops.rs:78-81); subtract != 0 gives a - b (generate_cancelling_scalar_vec negates a running sum)
int rofl_scalar_add_vec(const uint8_t *a32, const uint8_t *b32, size_t d, int subtract, uint8_t *out32) {
    for (size_t i = 0; i < d; i++) {
        sc a = sc_frombytes(a32 + 32 * i), b = sc_frombytes(b32 + 32 * i);
        if (sc_geq_l(a.v)) a = sc_from_mont(sc_to_mont(a));
        if (sc_geq_l(b.v)) b = sc_from_mont(sc_to_mont(b));
        sc_tobytes(out32 + 32 * i, subtract ? sc_sub(a, b) : sc_add(a, b));
    }
    return ROFL_OK;
}
// conversion32::f32_to_fp_vec / uint_to_f32_vec (conversion32.rs:41-54): Fix is unsigned, negative inputs saturate to 0
int rofl_f32_to_fp_vec(const float *in, size_t d, unsigned fp_bits, unsigned fp_frac, uint64_t *out) {
    if (!valid_fp(fp_bits, fp_frac)) return fail(ROFL_BAD_PARAM, "bad parameter");
    for (size_t i = 0; i < d; i++) {
        if (std::isnan(in[i])) return fail(ROFL_NON_FINITE, "non-finite value (the reference panics in fixed::saturating_from_float)");
        if (in[i] < 0.0f) { out[i] = 0; continue; }
        sc s; int rc = f32_to_sc(in[i], fp_bits, fp_frac, &s); if (rc) return fail(rc, "non-finite value");
        out[i] = read_from_bytes(s, fp_bits);
    }
    return ROFL_OK;
}
int rofl_uint_to_f32_vec(const uint64_t *in, size_t d, unsigned fp_bits, unsigned fp_frac, float *out) {
    if (!valid_fp(fp_bits, fp_frac)) return fail(ROFL_BAD_PARAM, "bad parameter");
    for (size_t i = 0; i < d; i++) out[i] = fix_to_f32(in[i] & fix_max_bits(fp_bits), fp_frac);
    return ROFL_OK;
}
int rofl_get_clip_bounds(size_t range, unsigned fp_bits, unsigned fp_frac, float *mn, float *mx) {
    if (!valid_fp(fp_bits, fp_frac) || range == 0 || range > 128) return fail(ROFL_BAD_PARAM, "bad parameter");
    clip_bounds(range, fp_bits, fp_frac, mn, mx); return ROFL_OK;
}
int rofl_get_l2_clip_bounds(size_t range, unsigned fp_bits, unsigned fp_frac, float *out) {
    if (!valid_fp(fp_bits, fp_frac) || range == 0 || range > 127) return fail(ROFL_BAD_PARAM, "bad parameter");
    *out = l2_clip_bound(range, fp_bits, fp_frac); return ROFL_OK;
}

int rofl_dbg_msm(const uint8_t *scalars32, const uint8_t *points32, size_t n, uint8_t out32[32]) {
    return guarded([&]() -> int {
        LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
        if (n == 0) return fail(ROFL_BAD_PARAM, "bad parameter");
        C.init();
        std::vector<sc> hs(n);
        for (size_t i = 0; i < n; i++) { hs[i] = sc_frombytes(scalars32 + 32 * i); if (sc_geq_l(hs[i].v)) hs[i] = sc_from_mont(sc_to_mont(hs[i])); }
        uint8_t *dp = C.tmp_in.as<uint8_t>(n * 32); niels *dn = C.aux_pts.as<niels>(n); sc *ds = C.aux_scal.as<sc>(n);
        u32 *status = C.status.as<u32>(4);
        HIPCHK(hipMemsetAsync(status, 0, 16, C.stream));
        C.up(dp, points32, n * 32, C.stream);
        C.up(ds, hs.data(), n * 32, C.stream);
        ROFL_LAUNCH(k_decode, grid1(n), dim3(TPB), 0, C.stream, (u32)n, (u32)n, dp, (const niels *)nullptr, dn, (uint8_t *)nullptr, status);
        u32 st = 0;
        HIPCHK(hipMemcpyAsync(&st, status, 4, hipMemcpyDeviceToHost, C.stream));
        C.sync();
        if (st & 4u) return fail(ROFL_FORMAT_ERROR, "invalid Ristretto encoding");
        std::vector<MsmProb> pr(1, MsmProb{dn, ds}); std::vector<ge5> res;
        msm_run(C, pr, n, res);
        h51::encode(out32, res[0]);
        return ROFL_OK;
    });
}
// bulletproofs' RangeProof::verify_multiple(&bp_gens, &pc_gens, &mut Transcript::new(label), &value_commitments, n) on ONE aggregated proof
// exactly as upstream's own tests call it: any transcript label (upstream's serialized-proof vectors use b"Deserialize-And-Verify Test"),
// the commitments as they are (no shift, no padding: m must be a power of two), generators of capacity >= n.  If byte vectors of
// bulletproofs 4.0.0 ever become reachable, they go straight through the HIP verifier here (the crate-level pin the oracle lacks).
int rofl_dbg_verify_labelled(const uint8_t *label, size_t label_len, size_t gens_capacity, const uint8_t *proof, size_t proof_len,
                             const uint8_t *commits32, size_t m, size_t n_bits, const uint8_t verifier_seed[32], int *ok_out) {
    return guarded([&]() -> int {
        LaneLock lane_lock = acquire_lane(false, true); Ctx &C = *lane_lock.c;
        if (!label || !label_len || label_len > 255 || !proof || !commits32 || !m || !is_pow2(m) || !verifier_seed || !ok_out) return fail(ROFL_BAD_PARAM, "bad parameter");
        *ok_out = 0;
        std::string lbl((const char *)label, label_len);
        if (lbl.find('\0') != std::string::npos) return fail(ROFL_BAD_PARAM, "the label holds a NUL byte");
        C.init();
        timing_begin(C);
        uint8_t *d_in = C.Cbytes.as<uint8_t>(m * 32), *d_enc = C.Vbytes.as<uint8_t>(m * 32);
        niels *d_vn = C.gbuf[0].as<niels>(m);
        u32 *status = C.status.as<u32>(4);
        HIPCHK(hipMemsetAsync(status, 0, 16, C.stream));
        C.up(d_in, commits32, m * 32, C.stream);
        ROFL_LAUNCH(k_decode, grid1(m), dim3(TPB), 0, C.stream, (u32)m, (u32)m, d_in, (const niels *)nullptr, d_vn, d_enc, status);
        std::vector<uint8_t> hV(m * 32); u32 *h_st = C.h_misc.as<u32>(4);
        uint8_t *hp = C.h_V.as<uint8_t>(m * 32);
        HIPCHK(hipMemcpyAsync(hp, d_enc, m * 32, hipMemcpyDeviceToHost, C.stream));
        HIPCHK(hipMemcpyAsync(h_st, status, 4, hipMemcpyDeviceToHost, C.stream));
        C.sync();
        if (h_st[0] & 4u) { timing_end(C); return fail(ROFL_FORMAT_ERROR, "commitment is not a valid Ristretto encoding"); }
        u64 cidx = 0; int ok = 0;
        int rc = verify_chunks(C, lbl.c_str(), gens_capacity, 1, n_bits, m, proof, proof_len, hp, d_vn, verifier_seed, &cidx, &ok);
        timing_end(C);
        if (rc) return fail(rc, "proof rejected before verification (format / bitsize / generator capacity)");
        *ok_out = ok;
        return ROFL_OK;
    });
}
int rofl_dbg_msm_retries(uint64_t out[4]) { if (!out) return ROFL_BAD_PARAM; for (int i = 0; i < 4; i++) out[i] = g_msm_stat[i].load(); return ROFL_OK; }
int rofl_dbg_quad_ops(const uint8_t *pairs64, size_t pairs, unsigned doublings, uint8_t *out_serial32, uint8_t *out_quad32) {
    return guarded([&]() -> int {
        LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
        if (pairs == 0) return fail(ROFL_BAD_PARAM, "bad parameter");
        C.init();
        uint8_t *din = C.tmp_in.as<uint8_t>(pairs * 64), *dout = C.tmp_out.as<uint8_t>(pairs * 64);
        u32 *status = C.status.as<u32>(4);
        HIPCHK(hipMemsetAsync(status, 0, 16, C.stream));
        HIPCHK(hipMemsetAsync(dout, 0, pairs * 64, C.stream));
        C.up(din, pairs64, pairs * 64, C.stream);
        ROFL_LAUNCH(k_dbg_quad, grid1(pairs * 4), dim3(TPB), 0, C.stream, (u32)pairs, doublings, (const uint8_t *)din, dout, dout + pairs * 32, status);
        u32 st = 0;
        HIPCHK(hipMemcpyAsync(&st, status, 4, hipMemcpyDeviceToHost, C.stream));
        C.down(out_serial32, dout, pairs * 32, C.stream);
        C.down(out_quad32, dout + pairs * 32, pairs * 32, C.stream);
        C.sync();
        if (st & 4u) return fail(ROFL_FORMAT_ERROR, "invalid Ristretto encoding");
        return ROFL_OK;
    });
}
int rofl_discrete_log_vec(const uint8_t *points32, size_t d, size_t table_size, unsigned bsgs_bits, uint8_t *scalars_out32) {
    return guarded([&]() -> int {
        LaneLock lane_lock = acquire_lane(true); Ctx &C = *lane_lock.c;
        if (table_size == 0 || table_size >= (1u << 30) || !(bsgs_bits == 8 || bsgs_bits == 16 || bsgs_bits == 32)) return fail(ROFL_BAD_PARAM, "bad parameter");
        if (d == 0) return ROFL_OK;
        C.init();
        timing_begin(C);
        auto it = C.bsgs.find(table_size);
        if (it == C.bsgs.end()) {
            Ctx::Bsgs b; u32 nslots = 1; while (nslots < 2 * (table_size + 1)) nslots <<= 1;
            b.mask = nslots - 1;
            HIPCHK(hipMalloc(&b.keys, 32 * (table_size + 1))); HIPCHK(hipMalloc(&b.slots, sizeof(u32) * nslots));
            HIPCHK(hipMemsetAsync(b.slots, 0, sizeof(u32) * nslots, C.stream));
            ROFL_LAUNCH(k_bsgs_build, dim3((unsigned)((table_size + 1 + 63) / 64)), dim3(64), 0, C.stream, (u32)table_size, C.d_tabB, b.keys, b.slots, b.mask);
            it = C.bsgs.emplace(table_size, b).first;
        }
        const Ctx::Bsgs &B = it->second;
        u64 mask = bsgs_bits >= 32 ? 0xffffffffULL : ((1ULL << bsgs_bits) - 1);
        // mG = B * Scalar::from(m as BSGS_URawFix)  (bsgs32.rs:18): the multiplier wraps to bsgs_bits bits
        niels neg_mG = h51::to_niels32(h_fixed_mul(C.ht.B5, sc_neg(sc_from_u64((u64)table_size & mask))));
        u64 max_it = (1ULL << bsgs_bits) / table_size;
        uint8_t *dp = C.tmp_in.as<uint8_t>(d * 32), *dout = C.Cbytes.as<uint8_t>(d * 32);
        u32 *status = C.status.as<u32>(4);
        HIPCHK(hipMemsetAsync(status, 0, 16, C.stream));
        C.up(dp, points32, 32 * d, C.stream);
        ROFL_LAUNCH(k_bsgs_solve, dim3((unsigned)((d + 63) / 64)), dim3(64), 0, C.stream, (u32)d, dp, (u32)table_size, bsgs_bits, max_it, neg_mG, B.keys, B.slots, B.mask, dout, status);
        u32 st = 0;
        C.down(scalars_out32, dout, 32 * d, C.stream);
        HIPCHK(hipMemcpyAsync(&st, status, 4, hipMemcpyDeviceToHost, C.stream));
        C.sync();
        timing_end(C);
        if (st & 4u) return fail(ROFL_FORMAT_ERROR, "invalid Ristretto encoding");
        if (st & 8u) return fail(ROFL_BAD_PARAM, "discrete log not found (the reference unwraps None)");
        return ROFL_OK;
    });
}
size_t rofl_wire_encoded_size(const rofl_wire_msg_t *m) { return m ? wire::encoded_size(*m) : 0; }
int rofl_wire_encode(const rofl_wire_msg_t *m, uint8_t *out, size_t cap, size_t *len_out) {
    if (!m || !out) return fail(ROFL_BAD_PARAM, "bad parameter");
    int rc = wire::encode(*m, out, cap, len_out);
    return rc ? fail(rc, "wire encode: unknown message kind or output buffer too small") : ROFL_OK;
}
int rofl_wire_decode(int kind, const uint8_t *data, size_t len, rofl_wire_msg_t *m, uint8_t *range_proofs_out, size_t range_proofs_cap) {
    if (!m || !data) return fail(ROFL_BAD_PARAM, "bad parameter");
    int rc = wire::decode(kind, data, len, m, range_proofs_out, range_proofs_cap);
    return rc ? fail(rc, rc == ROFL_FORMAT_ERROR ? "malformed message (prost's decode_length_delimited would return Err; the reference unwraps it)" : "bad parameter") : ROFL_OK;
}
}  // extern "C"

// ================================================================ multi-process exchange (RCCL over xGMI)
// One process per GPU: the exchange steps of a round -- all-gather of [verdict | proof bytes | commitments] of every rank, MIN / MAX / SUM
// reductions of a few numbers -- for hosts that are NOT Python (a C / Rust rofl_service has no torch.distributed), and so that every rank of
// a node runs the proof path AND its collectives on ONE HIP runtime: librccl is loaded here, next to the runtime this library is bound to.
// Payloads live in host memory at the ABI (proofs and commitments are returned to the host; SURVEY 8(e): no collective inside the proof
// path): they are staged through the communicator's pinned buffers, all-gathered device to device, and handed back in host memory.
namespace {
struct Rccl {
    void *h = nullptr; std::string path, err;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl &rccl() {
    static Rccl r; static std::once_flag once;
    std::call_once(once, [] {
        // ROFL_RCCL_LIB: which librccl (default: by soname, i.e. the one of the ROCm installation the process links).  A Python host that has
        // imported torch names /opt/rocm/lib/librccl.so.1 explicitly: "librccl.so.1" alone would resolve to the copy torch bundles.
        const char *e = knob("ROFL_RCCL_LIB");
        r.path = e && *e ? e : "librccl.so.1";
        r.h = dlopen(r.path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!r.h) { const char *m = dlerror(); r.err = m ? m : "dlopen failed"; return; }
        auto sym = [&](const char *n) { void *f = dlsym(r.h, n); if (!f && r.err.empty()) r.err = std::string("missing symbol ") + n; return f; };
        r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        if (!r.err.empty()) { dlclose(r.h); r.h = nullptr; }
    });
    return r;
}
struct NcclErr { ncclResult_t e; const char *what; };
#define NCCLCHK(x) do { ncclResult_t e__ = (x); if (e__ != ncclSuccess) throw NcclErr{e__, #x}; } while (0)
struct CommState {
    std::mutex mu; ncclComm_t comm = nullptr; int rank = 0, world = 0, phys = 0; hipStream_t st = nullptr;
    DevBuf send, recv; PinBuf hsend, hrecv;
};
CommState g_comm;
template <class F> int comm_guarded(F f) {
    return guarded([&]() -> int {
        try { return f(); }
        catch (const NcclErr &e) {
            Rccl &R = rccl();
            return fail(ROFL_COMM_ERROR, std::string("RCCL error ") + std::to_string((int)e.e) + " (" + (R.GetErrorString ? R.GetErrorString(e.e) : "?") + ") in " + e.what);
        }
    });
}
int comm_ready(Rccl *&R) {
    R = &rccl();
    if (!R->h) return fail(ROFL_COMM_ERROR, "librccl could not be loaded (" + R->path + "): " + R->err);
    return ROFL_OK;
}
}  // namespace

extern "C" {
int rofl_comm_unique_id(uint8_t id_out[128]) {
    return comm_guarded([&]() -> int {
        Rccl *R; if (int rc = comm_ready(R)) return rc;
        if (!id_out) return fail(ROFL_BAD_PARAM, "bad parameter");
        static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
        ncclUniqueId id; NCCLCHK(R->GetUniqueId(&id)); memcpy(id_out, &id, 128); return ROFL_OK;
    });
}
int rofl_comm_init(const uint8_t id[128], int rank, int world) {
    return comm_guarded([&]() -> int {
        Rccl *R; if (int rc = comm_ready(R)) return rc;
        if (!id || world < 1 || rank < 0 || rank >= world) return fail(ROFL_BAD_PARAM, "bad parameter");
        std::lock_guard<std::mutex> lk(g_comm.mu);
        if (g_comm.comm) return fail(ROFL_BAD_PARAM, "a communicator already exists in this process (rofl_comm_destroy first)");
        Ctx &P = ctx(); { std::lock_guard<std::mutex> g(P.init_mu); P.init(); }      // the calling thread's device: one process per GPU
        HIPCHK(hipSetDevice(P.phys));
        ncclUniqueId uid; memcpy(&uid, id, 128);
        ncclComm_t c = nullptr;
        NCCLCHK(R->CommInitRank(&c, world, uid, rank));
        g_comm.comm = c; g_comm.rank = rank; g_comm.world = world; g_comm.phys = P.phys;
        HIPCHK(hipStreamCreateWithFlags(&g_comm.st, hipStreamNonBlocking));
        return ROFL_OK;
    });
}
int rofl_comm_info(int *rank_out, int *world_out, int *rccl_version_out, char *lib_path_out, size_t len) {
    Rccl &R = rccl();
    std::lock_guard<std::mutex> lk(g_comm.mu);
    if (rank_out) *rank_out = g_comm.comm ? g_comm.rank : -1;
    if (world_out) *world_out = g_comm.comm ? g_comm.world : 0;
    if (rccl_version_out) { int v = 0; if (R.h && R.GetVersion) (void)R.GetVersion(&v); *rccl_version_out = v; }
    if (lib_path_out && len) {      // the file that was mapped, not the name it was asked for
        std::string real = R.path;
        if (R.h) { Dl_info di; if (dladdr((void *)R.GetVersion, &di) && di.dli_fname) real = di.dli_fname; }
        snprintf(lib_path_out, len, "%s", real.c_str());
    }
    return R.h ? ROFL_OK : fail(ROFL_COMM_ERROR, "librccl could not be loaded (" + R.path + "): " + R.err);
}
int rofl_comm_allgather(const uint8_t *local, size_t n, uint8_t *all_out) {
    return comm_guarded([&]() -> int {
        Rccl *R; if (int rc = comm_ready(R)) return rc;
        std::lock_guard<std::mutex> lk(g_comm.mu);
        if (!g_comm.comm) return fail(ROFL_BAD_PARAM, "no communicator (rofl_comm_init)");
        if (!n) return ROFL_OK;
        if (!local || !all_out) return fail(ROFL_BAD_PARAM, "bad parameter");
        HIPCHK(hipSetDevice(g_comm.phys));
        const size_t W = (size_t)g_comm.world;
        uint8_t *ds = g_comm.send.as<uint8_t>(n), *dr = g_comm.recv.as<uint8_t>(n * W);
        uint8_t *hs = g_comm.hsend.as<uint8_t>(n), *hr = g_comm.hrecv.as<uint8_t>(n * W);
        memcpy(hs, local, n);
        HIPCHK(hipMemcpyAsync(ds, hs, n, hipMemcpyHostToDevice, g_comm.st));
        NCCLCHK(R->AllGather(ds, dr, n, ncclUint8, g_comm.comm, g_comm.st));
        HIPCHK(hipMemcpyAsync(hr, dr, n * W, hipMemcpyDeviceToHost, g_comm.st));
        HIPCHK(hipStreamSynchronize(g_comm.st));
        memcpy(all_out, hr, n * W);
        return ROFL_OK;
    });
}
int rofl_comm_allreduce_f64(double *inout, size_t count, int op) {
    return comm_guarded([&]() -> int {
        Rccl *R; if (int rc = comm_ready(R)) return rc;
        std::lock_guard<std::mutex> lk(g_comm.mu);
        if (!g_comm.comm) return fail(ROFL_BAD_PARAM, "no communicator (rofl_comm_init)");
        if (!inout || !count || count > 4096 || op < 0 || op > 2) return fail(ROFL_BAD_PARAM, "bad parameter");
        HIPCHK(hipSetDevice(g_comm.phys));
        double *d = g_comm.send.as<double>(count), *h = g_comm.hsend.as<double>(count);
        memcpy(h, inout, 8 * count);
        HIPCHK(hipMemcpyAsync(d, h, 8 * count, hipMemcpyHostToDevice, g_comm.st));
        NCCLCHK(R->AllReduce(d, d, count, ncclFloat64, op == 0 ? ncclSum : op == 1 ? ncclMin : ncclMax, g_comm.comm, g_comm.st));
        HIPCHK(hipMemcpyAsync(h, d, 8 * count, hipMemcpyDeviceToHost, g_comm.st));
        HIPCHK(hipStreamSynchronize(g_comm.st));
        memcpy(inout, h, 8 * count);
        return ROFL_OK;
    });
}
int rofl_comm_barrier(void) { double one = 1.0; return rofl_comm_allreduce_f64(&one, 1, 0); }
int rofl_comm_destroy(void) {
    return comm_guarded([&]() -> int {
        std::lock_guard<std::mutex> lk(g_comm.mu);
        if (!g_comm.comm) return ROFL_OK;
        Rccl &R = rccl();
        (void)hipSetDevice(g_comm.phys);
        if (g_comm.st) { (void)hipStreamSynchronize(g_comm.st); (void)hipStreamDestroy(g_comm.st); g_comm.st = nullptr; }
        ncclComm_t c = g_comm.comm; g_comm.comm = nullptr; g_comm.world = 0;
        NCCLCHK(R.CommDestroy(c));
        return ROFL_OK;
    });
}
}  // extern "C"

extern "C" {
namespace {
struct OptSlot { std::atomic<int> *i; std::atomic<long> *l; long lo, hi; };
bool option_slot(const char *key, OptSlot *o) {
    Options &O = opts();
    const struct { const char *k; OptSlot s; } tab[] = {
        {"verify_zip_truncate", {&O.zip_truncate, nullptr, 0, 1}}, {"verify_batch", {&O.verify_batch, nullptr, 0, 2}},
        {"sigma_batch", {&O.sigma_batch, nullptr, 0, 1}}, {"blocking_sync", {&O.blocking_sync, nullptr, -1, 1}},
        {"devices", {nullptr, &O.devices, 0, (long)(~0UL >> 1)}}};
    for (auto &t : tab) if (key && !strcmp(key, t.k)) { *o = t.s; return true; }
    return false;
}
}  // namespace
int rofl_set_option(const char *key, long value) {
    if (key && !strcmp(key, "default_device")) {      // the device of threads that never called rofl_set_device (otherwise: the first device that was set)
        if (value < 0 || value >= kMaxDevices) return fail(ROFL_BAD_PARAM, "unknown option or value out of range");
        g_default_device.store((int)value); g_default_set.store(true); return ROFL_OK;
    }
    OptSlot o;
    if (!option_slot(key, &o) || value < o.lo || value > o.hi) return fail(ROFL_BAD_PARAM, "unknown option or value out of range");
    if (o.i) o.i->store((int)value); else o.l->store(value);
    return ROFL_OK;
}
int rofl_get_option(const char *key, long *value_out) {
    if (!value_out) return fail(ROFL_BAD_PARAM, "bad parameter");
    if (key && !strcmp(key, "lanes")) {      // read-only: the lanes of the calling thread's device (ROFL_LANES after clamping)
        return guarded([&]() -> int { Ctx &P = ctx(); { std::lock_guard<std::mutex> g(P.init_mu); P.init(); } *value_out = P.nlanes; return ROFL_OK; });
    }
    if (key && !strcmp(key, "default_device")) { *value_out = g_default_device.load(); return ROFL_OK; }
    OptSlot o;
    if (!option_slot(key, &o)) return fail(ROFL_BAD_PARAM, "unknown option");
    *value_out = o.i ? (long)o.i->load() : o.l->load();
    return ROFL_OK;
}
int rofl_set_timing(int enabled) {
    return guarded([&]() -> int {
        Ctx &P = ctx(); { std::lock_guard<std::mutex> g(P.init_mu); P.init(); }
        { std::lock_guard<std::mutex> lk(P.mu); P.tm.enabled = enabled == 1; P.tm.acc_only = enabled == 2; }
        for (Ctx *s : P.sibs) { std::lock_guard<std::mutex> lk(s->mu); s->tm.enabled = enabled == 1; s->tm.acc_only = enabled == 2; }
        return ROFL_OK;
    });
}
/* timing of the last instrumented call made by the calling thread */
int rofl_last_timing(rofl_timing_t *out) { if (!out) return ROFL_BAD_PARAM; *out = g_last_timing; return ROFL_OK; }
int rofl_last_kernel_times(rofl_kernel_time_t *out) { if (!out) return ROFL_BAD_PARAM; memcpy(out, g_last_ktimes, sizeof g_last_ktimes); return ROFL_OK; }
int rofl_bench_femul(unsigned iters, double *out) {
    return guarded([&]() -> int {
        LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c; C.init();
        const u32 blocks = 256 * 8, threads = blocks * TPB;
        fe *din, *dout; HIPCHK(hipMalloc(&din, sizeof(fe) * 256)); HIPCHK(hipMalloc(&dout, sizeof(fe) * threads));
        std::vector<fe> h(256); for (int i = 0; i < 256; i++) for (int k = 0; k < 8; k++) h[i].v[k] = 0x9e3779b9u * (i * 8 + k + 1);
        HIPCHK(hipMemcpy(din, h.data(), sizeof(fe) * 256, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
        // ROFL_FEMUL_LDS: dynamic LDS per block, to hold the microbenchmark at the occupancy of a real kernel (40960 -> 4 blocks per CU =
        // 4 waves/SIMD, what k_msm_accumulate's 127 VGPRs allow); default 0 = 8 waves/SIMD
        static const size_t fl = knob("ROFL_FEMUL_LDS") ? (size_t)atol(knob("ROFL_FEMUL_LDS")) : 0;
        if (fl) HIPCHK(hipFuncSetAttribute((const void *)k_bench_femul, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fl));
        // ROFL_FEMUL_MODE = 1 / 2: the 7-multiplication mixed addition instead (entry held in registers / fetched per addition from a 32 KB table)
        static const int mode = knob("ROFL_FEMUL_MODE") ? atoi(knob("ROFL_FEMUL_MODE")) : 0;
        if (mode) {
            // mode 3: ROFL_FEMUL_TABLE entries (power of two, default 2^21 = 256 MB) gathered at random, like the window table is
            static const size_t tab = knob("ROFL_FEMUL_TABLE") ? (size_t)atol(knob("ROFL_FEMUL_TABLE")) : ((size_t)1 << 21);
            size_t entries = mode == 3 ? tab : 256;
            ndm *dt; ge *dg; HIPCHK(hipMalloc(&dt, sizeof(ndm) * entries)); HIPCHK(hipMalloc(&dg, sizeof(ge) * threads));
            HIPCHK(hipMemset(dt, 0x11, sizeof(ndm) * entries));
            std::vector<ndm> ht(256); for (int i = 0; i < 256; i++) for (int k = 0; k < 32; k++) ht[i].v[k] = (0x9e3779b9u * (i * 32 + k + 1)) >> 8;
            HIPCHK(hipMemcpy(dt, ht.data(), sizeof(ndm) * 256, hipMemcpyHostToDevice));
            auto launch = [&](u32 it) {
                if (mode == 3) ROFL_LAUNCH(k_bench_madd_gather, dim3(blocks), dim3(TPB), fl, C.stream, it, (u32)entries, dt, dg);
                else if (mode == 2) ROFL_LAUNCH(k_bench_madd_l1, dim3(blocks), dim3(TPB), fl, C.stream, it, (u32)entries, dt, dg);
                else ROFL_LAUNCH(k_bench_madd_regs, dim3(blocks), dim3(TPB), fl, C.stream, it, (u32)entries, dt, dg);
            };
            launch(8u);
            HIPCHK(hipEventRecord(e0, C.stream));
            launch(iters);
            HIPCHK(hipEventRecord(e1, C.stream));
            HIPCHK(hipEventSynchronize(e1));
            float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            *out = (double)threads * iters * 7.0 / (ms * 1e-3);
            HIPCHK(hipFree(dt)); HIPCHK(hipFree(dg)); HIPCHK(hipFree(din)); HIPCHK(hipFree(dout)); HIPCHK(hipEventDestroy(e0)); HIPCHK(hipEventDestroy(e1));
            return ROFL_OK;
        }
        ROFL_LAUNCH(k_bench_femul, dim3(blocks), dim3(TPB), fl, C.stream, 8u, din, dout);
        HIPCHK(hipEventRecord(e0, C.stream));
        ROFL_LAUNCH(k_bench_femul, dim3(blocks), dim3(TPB), fl, C.stream, iters, din, dout);
        HIPCHK(hipEventRecord(e1, C.stream));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        *out = (double)threads * iters * 4.0 / (ms * 1e-3);
        HIPCHK(hipFree(din)); HIPCHK(hipFree(dout)); HIPCHK(hipEventDestroy(e0)); HIPCHK(hipEventDestroy(e1));
        return ROFL_OK;
    });
}

// ---- host-side self-tests of the shared host/device math (no GPU needed)
int rofl_dbg_host_pool_stress(unsigned threads, unsigned jobs) {
    // many tiny jobs back to back: every index of every job must run exactly once (a lost index would hang run())
    HostPool pool((int)threads);
    for (unsigned it = 0; it < jobs; it++) {
        size_t n = 2 + it % 7;
        std::atomic<int> hits[8];
        for (auto &h : hits) h = 0;
        pool.run(n, [&](size_t i) { hits[i].fetch_add(1); });
        for (size_t i = 0; i < n; i++) if (hits[i].load() != 1) return ROFL_BAD_PARAM;
    }
    return ROFL_OK;
}
int rofl_dbg_host_bench(int what, unsigned iters, double *ns_out) {
    if (!ns_out || !iters) return ROFL_BAD_PARAM;
    static const uint8_t Bc[32] = {0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f,
                                   0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};
    ge b; ristretto_decode(b, Bc);
    ge5 p = h51::from_ge(b), q = h51::gdouble(p);
    volatile u64 sink = 0;
    double t0 = now_ms();
    if (what == 0) { u64 st[25]; memset(st, 1, sizeof st); for (unsigned i = 0; i < iters; i++) keccak_f1600_host(st); sink = st[0]; }
    else if (what == 1) { for (unsigned i = 0; i < iters; i++) p = h51::gdouble(p); sink = p.X.v[0]; }
    else if (what == 2) { for (unsigned i = 0; i < iters; i++) p = h51::gadd(p, q); sink = p.X.v[0]; }
    else if (what == 3) { uint8_t e[32]; for (unsigned i = 0; i < iters; i++) { h51::encode(e, p); p = h51::gadd(p, q); } sink = p.X.v[0]; }
    else if (what == 4) {
        std::vector<niels> T; std::vector<niels5> T5; build_fixed_table(T, b); to_tab5(T5, T);
        sc k = sc_from_u64(0x123456789abcdefULL); t0 = now_ms();
        for (unsigned i = 0; i < iters; i++) { ge5 r = h_fixed_mul(T5, k); k.v[3] ^= (u32)r.X.v[0]; k.v[7] &= 0x0fffffffu; } sink = k.v[3];
    } else if (what == 5) { sc a = h_mont(sc_from_u64(0xdeadbeefcafeULL)); for (unsigned i = 0; i < iters; i++) { a = h51::sc_invert_mont_fast(a); a.v[0] |= 1; } sink = a.v[0]; }
    else if (what == 6) {
        std::vector<uint8_t> V((size_t)iters * 32); for (size_t i = 0; i < V.size(); i++) V[i] = (uint8_t)(i * 131 + 7);
        t0 = now_ms();
        Merlin t("RangeProof", 10);
        t.append32_run('V', V.data(), iters);
        sink = t.stw[0];
    } else if (what >= 7 && what <= 9) {
        // the pool hand-off of a hop: `iters` times { the caller busy-waits 300 us (what = 7, 9) or 30 us (8) as it does for the GPU, then runs 16
        // tasks of ~30 us (7, 8) or 128 tasks of ~8 us (9) }; result = ns per hand-off beyond nothing (ideal: 16 x 30 / threads, at least 30 us)
        int nt = std::min(16, std::max(2, usable_cores())); if (const char *e = knob("ROFL_HOST_THREADS")) nt = atoi(e);
        HostPool pool(nt);
        auto spin = [](double us) { double t = now_ms(); while ((now_ms() - t) * 1e3 < us) __builtin_ia32_pause(); };
        const size_t ntask = what == 9 ? 128 : 16; const double task_us = what == 9 ? 8.0 : 30.0, gap_us = what == 8 ? 30.0 : 300.0;
        double tot = 0;
        for (unsigned i = 0; i < iters; i++) {
            spin(gap_us);
            double a = now_ms();
            pool.run(ntask, [&](size_t) { spin(task_us); });
            tot += now_ms() - a;
        }
        *ns_out = tot * 1e6 / iters; return ROFL_OK;
    } else return ROFL_BAD_PARAM;
    *ns_out = (now_ms() - t0) * 1e6 / iters; (void)sink;
    return ROFL_OK;
}
// host51x8.hpp against host51.hpp: 8 x W pseudo-random window sums through the SIMD chain and through the scalar one.
// Returns 0 when all eight results agree, 1 on a mismatch, -1 when the CPU has no AVX-512 IFMA (nothing tested).
int rofl_dbg_host_horner8_selftest(unsigned W, unsigned c, int lanes, double *us_simd, double *us_scalar) {
    if (!h8::available()) return -1;
    if (W < 2 || W > 64 || c < 2 || c > 16 || lanes < 1 || lanes > 8) return ROFL_BAD_PARAM;
    static const uint8_t Bc[32] = {0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f,
                                   0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};
    ge b; ristretto_decode(b, Bc);
    u32 pos[64]; for (unsigned w = 0; w < W; w++) pos[w] = w * c;
    std::vector<ge> pts(8 * W);
    ge5 cur = h51::from_ge(b);
    for (size_t i = 0; i < pts.size(); i++) {
        cur = h51::gadd(h51::gdouble(cur), h51::from_ge(b)); if (i % 3 == 0) cur = h51::gdouble(cur);
        pts[i] = h51::to_ge(cur);
        if (i % 4 == 1) { u64 cy = 0; const u32 pw[8] = {0xffffffedu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x7fffffffu};      // a loose representative: X + p
                          for (int k = 0; k < 8; k++) { cy += (u64)pts[i].X.v[k] + pw[k]; pts[i].X.v[k] = (u32)cy; cy >>= 32; } }
    }
    auto wsum = [&](int l, int w) { return (const ge *)&pts[(size_t)l * W + w]; };
    uint8_t ref[8][32], got[8][32];
    double t0 = now_ms();
    for (int l = 0; l < lanes; l++) {
        ge5 acc = h51::from_ge_loose(*wsum(l, (int)W - 1));
        for (int w = (int)W - 2; w >= 0; w--) { for (u32 i = pos[w]; i < pos[w + 1]; i++) acc = h51::gdouble(acc); acc = h51::gadd(acc, h51::from_ge_loose(*wsum(l, w))); }
        h51::encode(ref[l], acc);
    }
    double t1 = now_ms();
    ge5 out[8];
    h8::horner8(out, lanes, (int)W, pos, wsum);
    double t2 = now_ms();
    int bad = 0;
    for (int l = 0; l < lanes; l++) { h51::encode(got[l], out[l]); bad |= memcmp(got[l], ref[l], 32) != 0; }
    if (us_simd) *us_simd = (t2 - t1) * 1e3;
    if (us_scalar) *us_scalar = (t1 - t0) * 1e3;
    return bad;
}
// h8::encode8 against h51::encode on `batches` x 8 points of a pseudo-random walk (arbitrary Z), the identity and small multiples included.
// Returns 0 when every encoding agrees, 1 on a mismatch, -1 without AVX-512 IFMA.
int rofl_dbg_host_encode8_selftest(unsigned batches, double *us_simd, double *us_scalar) {
    if (!h8::available()) return -1;
    static const uint8_t Bc[32] = {0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f,
                                   0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};
    ge b; ristretto_decode(b, Bc);
    ge5 b5 = h51::from_ge(b), cur = b5;
    int bad = 0; double ts = 0, tv = 0;
    for (unsigned it = 0; it < batches; it++) {
        ge5 pts[8];
        for (int l = 0; l < 8; l++) {
            cur = h51::gadd(h51::gdouble(cur), b5); if ((it + l) % 3 == 0) cur = h51::gdouble(cur);
            pts[l] = cur;
        }
        if (it == 0) { pts[0] = h51::identity(); pts[1] = b5; pts[2] = h51::gdouble(b5); pts[3] = h51::gadd(b5, h51::identity()); }
        uint8_t ref[8][32], got[8][32];
        double t0 = now_ms();
        for (int l = 0; l < 8; l++) h51::encode(ref[l], pts[l]);
        double t1 = now_ms();
        h8::encode8(got, pts);
        double t2 = now_ms();
        ts += t1 - t0; tv += t2 - t1;
        bad |= memcmp(ref, got, sizeof ref) != 0;
    }
    if (us_simd) *us_simd = tv * 1e3 / (batches ? batches : 1);
    if (us_scalar) *us_scalar = ts * 1e3 / (batches ? batches : 1);
    return bad;
}
// k8::append32_run_x8 against Merlin::append32_run: `lanes` transcripts that have absorbed the same prefix, `count` commitments each, from every
// starting position of the rate block (`skew` extra prefix bytes shift it); the challenge drawn afterwards must agree.  0 = equal, 1 = mismatch,
// -1 = no AVX-512.  Timings: microseconds for all lanes together.
// Merlin::append32_run (records put together in registers, split at the end of the rate block) against `count` plain append("V", msg, 32)
// calls after `skew` extra prefix bytes: state, positions and the next challenge.  0 = equal, 1 = mismatch.
int rofl_dbg_host_merlin_run_selftest(unsigned count, unsigned skew) {
    if (skew > 400) return ROFL_BAD_PARAM;
    std::vector<uint8_t> data((size_t)count * 32 + 1);
    for (size_t i = 0; i < data.size(); i++) data[i] = (uint8_t)((i * 2246822519u) >> 11);
    std::vector<uint8_t> pre(skew, 0xa5);
    Merlin a("RangeProof", 10), b("RangeProof", 10);
    if (skew) { a.append("skew", pre.data(), skew); b.append("skew", pre.data(), skew); }
    a.append32_run('V', data.data(), count);
    for (unsigned j = 0; j < count; j++) b.append("V", data.data() + (size_t)j * 32, 32);
    int bad = a.pos != b.pos || a.pos_begin != b.pos_begin || a.cur_flags != b.cur_flags || memcmp(a.stw, b.stw, 200) != 0;
    uint8_t ca[64], cb[64]; a.challenge_bytes("y", ca, 64); b.challenge_bytes("y", cb, 64); bad |= memcmp(ca, cb, 64) != 0;
    return bad;
}
// keccak_f1600_zmm (one state across AVX-512 registers) against the scalar permutation on `states` pseudo-random states, chained `chain` deep,
// plus the known answer of the all-zero state.  0 = all equal, 1 = mismatch, -1 = no AVX-512 on this CPU.
int rofl_dbg_host_keccak_zmm_selftest(unsigned states, unsigned chain, double *ns_zmm, double *ns_scalar) {
    if (!__builtin_cpu_supports("avx512f")) return -1;
    if (!states || !chain) return ROFL_BAD_PARAM;
    int bad = 0;
    { u64 z[25] = {0}; keccak_f1600_zmm(z); bad |= z[0] != 0xF1258F7940E1DDE7ULL || z[24] != 0xEAF1FF7B5CECA249ULL; }
    u64 x = 0x9e3779b97f4a7c15ULL;
    double tz = 0, ts = 0;
    for (unsigned i = 0; i < states; i++) {
        u64 a[25], b[25];
        for (int k = 0; k < 25; k++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; a[k] = b[k] = (i % 5 == 4 && k % 3 == 0) ? 0 : x; }
        double t0 = now_ms();
        for (unsigned c = 0; c < chain; c++) keccak_f1600_zmm(a);
        double t1 = now_ms();
        for (unsigned c = 0; c < chain; c++) keccak_f1600_host(b);
        double t2 = now_ms();
        tz += t1 - t0; ts += t2 - t1;
        bad |= memcmp(a, b, sizeof a) != 0;
    }
    if (ns_zmm) *ns_zmm = tz * 1e6 / ((double)states * chain);
    if (ns_scalar) *ns_scalar = ts * 1e6 / ((double)states * chain);
    return bad;
}
int rofl_dbg_host_merlin8_selftest(int lanes, unsigned count, unsigned skew, double *us_simd, double *us_scalar) {
    if (!k8::available()) return -1;
    if (lanes < 1 || lanes > 8 || skew > 400) return ROFL_BAD_PARAM;
    std::vector<uint8_t> data((size_t)8 * count * 32);
    for (size_t i = 0; i < data.size(); i++) data[i] = (uint8_t)((i * 2654435761u) >> 13);
    std::vector<uint8_t> pre(skew, 0x5a);
    std::vector<Merlin> a, b;
    for (int l = 0; l < lanes; l++) { a.emplace_back("RangeProof", 10); b.emplace_back("RangeProof", 10); }
    for (int l = 0; l < lanes; l++) { uint8_t id = (uint8_t)l; a[l].append("id", &id, 1); b[l].append("id", &id, 1); if (skew) { a[l].append("skew", pre.data(), skew); b[l].append("skew", pre.data(), skew); } }
    double t0 = now_ms();
    for (int l = 0; l < lanes; l++) a[l].append32_run('V', data.data() + (size_t)l * count * 32, count);
    double t1 = now_ms();
    Merlin *t[8]; const uint8_t *msg[8];
    for (int l = 0; l < lanes; l++) { t[l] = &b[l]; msg[l] = data.data() + (size_t)l * count * 32; }
    k8::append32_run_x8(t, lanes, 'V', msg, count);
    double t2 = now_ms();
    int bad = 0;
    for (int l = 0; l < lanes; l++) {
        bad |= a[l].pos != b[l].pos || a[l].pos_begin != b[l].pos_begin || a[l].cur_flags != b[l].cur_flags || memcmp(a[l].stw, b[l].stw, 200) != 0;
        uint8_t ca[64], cb[64]; a[l].challenge_bytes("y", ca, 64); b[l].challenge_bytes("y", cb, 64); bad |= memcmp(ca, cb, 64) != 0;
    }
    if (us_simd) *us_simd = (t2 - t1) * 1e3;
    if (us_scalar) *us_scalar = (t1 - t0) * 1e3;
    return bad;
}
// the host hops of the calling thread's last create / verify call: out[0..8] = number of MSM hops, enqueue ms, wait ms, window-combination
// wall ms, sum of each hop's slowest task ms, then the maxima over the hops: enqueue, wait, combination wall, slowest task
int rofl_dbg_last_hops(double out[10]) { if (!out) return fail(ROFL_BAD_PARAM, "bad parameter"); memcpy(out, g_last_hops, sizeof g_last_hops); return ROFL_OK; }
int rofl_dbg_host_fe_mul(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) { fe_tobytes(out, fe_mul(fe_frombytes(a), fe_frombytes(b))); return 0; }
int rofl_dbg_host_fe_ops(const uint8_t a[32], const uint8_t b[32], uint8_t oa[32], uint8_t os[32], uint8_t oq[32], uint8_t oi[32]) {
    fe x = fe_frombytes(a), y = fe_frombytes(b);
    // exercise the non-canonical range as well: add 2p-ish slack by doubling through fe_add
    fe_tobytes(oa, fe_add(x, y)); fe_tobytes(os, fe_sub(x, y)); fe_tobytes(oq, fe_sq(x)); fe_tobytes(oi, fe_invert(x)); return 0;
}
// both inversion routines (canonical in / out) and their timings in nanoseconds per call
int rofl_dbg_host_sc_invert(const uint8_t a[32], uint8_t out_ref[32], uint8_t out_fast[32], double *ns_ref, double *ns_fast) {
    sc am = h_mont(sc_frombytes(a));
    sc r1 = sc_invert_mont(am), r2 = h51::sc_invert_mont_fast(am);
    sc_tobytes(out_ref, h_canon(r1)); sc_tobytes(out_fast, h_canon(r2));
    if (ns_ref && ns_fast) {
        const int reps = 200; volatile u32 sink = 0;
        double t0 = now_ms(); for (int i = 0; i < reps; i++) { am.v[0] ^= (u32)i; sink += sc_invert_mont(am).v[0]; } double t1 = now_ms();
        for (int i = 0; i < reps; i++) { am.v[0] ^= (u32)i; sink += h51::sc_invert_mont_fast(am).v[0]; } double t2 = now_ms();
        *ns_ref = (t1 - t0) * 1e6 / reps; *ns_fast = (t2 - t1) * 1e6 / reps; (void)sink;
    }
    return 0;
}
int rofl_dbg_host_sc_mul(const uint8_t a[32], const uint8_t b[32], uint8_t out[32]) { sc_tobytes(out, h_mul(sc_frombytes(a), sc_frombytes(b))); return 0; }
int rofl_dbg_host_sc_lazy(const uint8_t *a32, const uint8_t *b32, size_t count, uint8_t out_lazy[32], uint8_t out_ref[32]) {
    if (count > 16) return ROFL_BAD_PARAM;      // sc_redc_wide's bound: sixteen products of operands < l
    u32 acc[17]; for (int i = 0; i < 17; i++) acc[i] = 0;
    sc ref = sc_zero();
    for (size_t k = 0; k < count; k++) {
        const sc a = sc_frombytes(a32 + 32 * k), b = sc_frombytes(b32 + 32 * k);
        sc_mac_wide(acc, a, b);
        ref = sc_add(ref, sc_montmul(a, b));
    }
    sc_tobytes(out_lazy, sc_redc_wide(acc)); sc_tobytes(out_ref, ref);
    return 0;
}
int rofl_dbg_host_sc_wide_mont(const uint8_t in[64], uint8_t out[32]) { sc_tobytes(out, sc_from_mont(sc_from_wide_mont(sc_frombytes(in), sc_frombytes(in + 32)))); return 0; }
int rofl_dbg_host_sc_wide(const uint8_t in[64], uint8_t out[32]) { sc_tobytes(out, sc_from_wide(sc_frombytes(in), sc_frombytes(in + 32))); return 0; }
int rofl_dbg_host_from_uniform(const uint8_t in[64], uint8_t out[32]) { ristretto_encode(out, ristretto_from_uniform(in)); return 0; }
int rofl_dbg_host_scalarmult_base(const uint8_t k[32], int use_bb, uint8_t out[32]) {
    static HostTables ht; static bool init = false; static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (!init) {
        static const uint8_t Bc[32] = {0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f,
                                       0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};
        ristretto_decode(ht.base, Bc); uint8_t h[64]; sha3_512(h, Bc, 32); ht.bblind = ristretto_from_uniform(h);
        build_fixed_table(ht.B, ht.base); build_fixed_table(ht.Bb, ht.bblind); init = true;
    }
    if (use_bb & 2) { to_tab5(ht.B5, ht.B); to_tab5(ht.Bb5, ht.Bb); h51::encode(out, h_fixed_mul((use_bb & 1) ? ht.Bb5 : ht.B5, sc_frombytes(k))); return 0; }
    ristretto_encode(out, h_fixed_mul32((use_bb & 1) ? ht.Bb : ht.B, sc_frombytes(k))); return 0;
}
int rofl_dbg_host_fd_ops(const uint8_t a[32], const uint8_t b[32], uint8_t om[32], uint8_t oq[32], uint8_t oa[32], uint8_t os[32], uint8_t oi[32]) {
    fe fa = fe_frombytes(a), fb = fe_frombytes(b);
    fa.v[7] |= (u32)(a[31] & 0x80) << 24; fb.v[7] |= (u32)(b[31] & 0x80) << 24;   // keep bit 255 to exercise the unpack fold
    fd x = fd_unpack(fa), y = fd_unpack(fb);
    fe_tobytes(om, fd_pack(fd_mul(fd_sub(fd_add(x, x), x), fd_add(y, fd_zero()))));   // (2x - x) * y with a loose first operand
    fe_tobytes(oq, fd_pack(fd_sq(x))); fe_tobytes(oa, fd_pack(fd_add(x, y))); fe_tobytes(os, fd_pack(fd_sub(x, y)));
    fe_tobytes(oi, fd_pack(fd_invert(x)));
    return 0;
}
// double-and-add ladder on the kernel point formulas: exercises gd_double, gd_madd (+/-), gd_add, gd_to_niels
int rofl_dbg_host_fd_scalarmult(const uint8_t k[32], const uint8_t p[32], uint8_t out[32]) {
    ge P; if (!ristretto_decode(P, p)) return ROFL_FORMAT_ERROR;
    nd q = nd_unpack(ge_to_niels(P));
    sc s = sc_frombytes(k);
    int8_t naf[256]; int top = sc_naf(naf, s);
    gd acc = gd_identity();
    for (int i = top; i >= 0; i--) { acc = gd_double(acc); if (naf[i]) acc = gd_madd(acc, q, naf[i] < 0); }
    gd twice = gd_add(acc, acc);                   // 2kP by the unified addition
    gd back = gd_madd(twice, nd_unpack(gd_to_niels(acc)), true);   // 2kP - kP
    ristretto_encode(out, gd_pack(back));
    return 0;
}
// the register-radix codec on the host (limb bounds asserted): decode, add the identity, re-encode
int rofl_dbg_host_fd_codec(const uint8_t in[32], uint8_t out[32]) {
    gd p; if (!gd_ristretto_decode(p, in)) return ROFL_FORMAT_ERROR;
    gd q = gd_add(gd_double(p), gd_madd(gd_identity(), nd_unpack(gd_to_niels(p)), true));     // 2P - P through the kernel formulas
    gd_ristretto_encode(out, q); return 0;
}
int rofl_dbg_host_decode_encode(const uint8_t in[32], uint8_t out[32]) { ge p; if (!ristretto_decode(p, in)) return ROFL_FORMAT_ERROR; ristretto_encode(out, ge_add(p, ge_identity())); return 0; }
int rofl_dbg_host_merlin(const uint8_t *label, size_t label_len, const uint8_t *msg, size_t msg_len, uint8_t out[64]) {
    Merlin t((const char *)label, label_len); t.append("msg", msg, msg_len); t.challenge_bytes("chal", out, 64); return 0;
}
int rofl_dbg_host_nonce(const uint8_t seed[32], uint64_t idx, uint8_t out[32]) {
    const u64 dom[2] = ROFL_NONCE_DOM;
    u64 sd[4]; memcpy(sd, seed, 32); u64 st[25]; shake256_seeded_block(st, dom, sd, idx >> 1);
    const int h = (int)(idx & 1);
    sc lo, hi; for (int i = 0; i < 4; i++) { lo.v[2 * i] = (u32)st[8 * h + i]; lo.v[2 * i + 1] = (u32)(st[8 * h + i] >> 32); hi.v[2 * i] = (u32)st[8 * h + 4 + i]; hi.v[2 * i + 1] = (u32)(st[8 * h + 4 + i] >> 32); }
    sc_tobytes(out, sc_from_wide(lo, hi)); return 0;
}

}  // extern "C"
