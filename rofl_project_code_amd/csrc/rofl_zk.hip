// librofl_zk.so: C ABI (include/rofl_zk.h) + host orchestration of the HIP kernels.
//
// Host responsibilities: Merlin transcripts (sequential), final window/bit combination of MSM partial
// sums (a 256-step Horner chain that would serialise a single GPU lane), fixed-base multiples of B and
// B_blinding, proof (de)serialisation.  Everything proportional to d * n_bits runs in kernels.hpp.
#include <hip/hip_runtime.h>
#include <chrono>
#include <time.h>
#include <sched.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <atomic>
#include <condition_variable>
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

using namespace rofl;

namespace {

// ---------------------------------------------------------------- tuning knobs
// Every ROFL_* environment variable the library reads, in ONE table (scripts/gen_knob_table.py turns it into the table of DESIGN.md and
// tests/test_host_lib.py checks that no other name is read).  Knobs never change results -- proofs, commitments and verdicts are the same
// for every setting (the behaviour switches of rofl_set_option are the exception and are marked "option") -- they move work between
// variants, and most of them exist because an experiment in DESIGN.md needed them.  Read once per process (or per device context).
struct Knob { const char *name, *dflt, *what; };
static const Knob KNOBS[] = {
    {"ROFL_LANES", "3", "calls that can be in flight on a device (HIP stream + workspace each); 1..8"},
    {"ROFL_HOST_THREADS", "usable cores, in [2, 16]", "threads of the primary lane's host pool (window combination, encodings, transcripts of a round); other lanes: 6"},
    {"ROFL_POOL_SPIN_US", "400", "how long an idle pool worker polls for the next job before it sleeps (0 = sleep at once)"},
    {"ROFL_BLOCKING_SYNC", "-1", "option blocking_sync: -1 spin while <= 3 calls are in flight, 0 always spin, 1 sleep between polls"},
    {"ROFL_VERIFY_ZIP_TRUNCATE", "0", "option verify_zip_truncate: 1 = the reference's zip-truncating verify_rangeproof"},
    {"ROFL_VERIFY_BATCH", "1", "option verify_batch: 0 = one check per proof instead of one per client"},
    {"ROFL_SIGMA_BATCH", "1", "option sigma_batch: 0 = one check per element in the Sigma-proof verifiers"},
    {"ROFL_GENS_BUDGET_MB", "98304", "HBM budget of the generator-table cache (LRU eviction of unpinned entries beyond it)"},
    {"ROFL_FOLD_T1", "3", "IPP rounds before the first generator fold (1..6)"},
    {"ROFL_FOLD_T", "2", "IPP rounds between later folds (1..6)"},
    {"ROFL_FOLD_MIN", "1024", "no fold once fewer generators per chunk would remain (launches with many chunks fold down to 64)"},
    {"ROFL_FOLD_TAB", "1", "0 = first fold without the precomputed odd-multiple slices"},
    {"ROFL_FOLD_PB", "32", "piece width of the fold table in bits (16, 32, 64)"},
    {"ROFL_FOLD_W", "8", "NAF width of the fold table (3..8; narrowed until the table fits ROFL_FOLD_TAB_MB)"},
    {"ROFL_FOLD_TAB_MB", "32768", "HBM budget of one (n, m) fold table"},
    {"ROFL_FOLD_UNIT", "1", "0 = do not keep the common factor s_0 of a fold in gscale / hscale"},
    {"ROFL_FOLD_K", "0", "digit-position segments per fold output (1, 2, 4; 0 = by launch size)"},
    {"ROFL_FOLD_THREADS", "131072", "fold launches with fewer threads split their chains into segments"},
    {"ROFL_FOLD_REGS", "1", "0 = the generic fold kernel instead of the three-sources-in-registers one"},
    {"ROFL_IPP_FUSED", "1", "0 = k_ipp_fold_ab + k_ipp_scalars + k_ipp_inner instead of one k_ipp_round per round"},
    {"ROFL_MSM_LR", "1", "0 = separate L and R scalar arrays (with zeros) instead of the merged layout"},
    {"ROFL_MSM_FB", "1", "0 = no window tables (every MSM generic)"},
    {"ROFL_MSM_FB_MIN", "4096", "generator sets smaller than this get no window table"},
    {"ROFL_MSM_FB_C", "0", "one window width (13, 15, 16) for every window table; 0 = 16-bit tables, plus a 15-bit one for generator sets below 2^17 that launches with many problems and the verifier use"},
    {"ROFL_MSM_FB_THREADS", "524288", "accumulate threads a fixed-base launch aims for (decides the number of bucket sets)"},
    {"ROFL_MSM_TWO_LEVEL", "1", "0 = slot sort instead of the two-level bucket sort in fixed-base launches"},
    {"ROFL_MSM_SLOTS", "1", "0 = count / scan / scatter sort only (no fixed-capacity structures)"},
    {"ROFL_MSM_LDS", "1", "0 = per-item global atomics instead of LDS ranking in the slot sort"},
    {"ROFL_MSM_LDS_MIN", "8192", "MSMs with fewer terms use the per-item slot sort"},
    {"ROFL_MSM_LDS_TILE", "131072", "items one block of the LDS slot sort ranks"},
    {"ROFL_MSM_SMALL_MAX", "8192", "terms per side up to which a generic MSM runs as one fused launch (0 = off)"},
    {"ROFL_MSM_HOST8", "1", "0 = launches with many problems combine their windows on the device (k_msm_horner) even when the host has AVX-512 IFMA"},
    {"ROFL_MSM_DEV_HORNER_MIN", "32", "launches with at least this many problems combine their windows on the device"},
    {"ROFL_MSM_T13", "8192", "generic MSMs from this many terms on use 13-bit windows"},
    {"ROFL_MSM_T10", "512", "generic MSMs from this many terms on use 10-bit windows (7-bit below, 4-bit below 64)"},
    {"ROFL_MSM_C", "0", "window width of every generic MSM (4, 7, 10, 13, 16; 0 = by size)"},
    {"ROFL_MSM_GROUP_REDUCE", "0", "1 = two-launch bucket reduction by groups of 512 (measured slower)"},
    {"ROFL_RED_SPLIT", "0", "1 = four threads per 8-group in k_msm_reduce_level (measured slower)"},
    {"ROFL_RED_FUSED_T", "512", "largest block of k_msm_reduce_fused"},
    {"ROFL_ACC_BALANCE", "1", "0 = accumulate blocks in plain descending-load order instead of equal-work blocks"},
    {"ROFL_TRACE", "0", "1 = one line per MSM on stderr, 2 = per-phase host timeline of every proof / verification"},
    {"ROFL_DBG_IDX_MASK", "0x7fffffff", "timing experiments only (WRONG results): confines the table gathers to a prefix"},
    {"ROFL_DBG_SCATTER", "0", "timing experiments only (WRONG results): 1 = no range reservation, 2 = no slot stores"},
    {"ROFL_DBG_SMALL_TIMELINE", "", "set: per-phase block timings of every fused small-MSM launch on stderr (synchronises; debugging)"},
    {"ROFL_DBG_ACC_TIMELINE", "", "file to append per-wave start / end / placement records of every fixed-base accumulate launch to"},
    {"ROFL_FEMUL_LDS", "0", "rofl_bench_femul: dynamic LDS per block (pins the micro-benchmark's occupancy)"},
    {"ROFL_FEMUL_MODE", "0", "rofl_bench_femul: 0 multiplication chain, 1-3 mixed addition from registers / a 32 KB table / a gathered table"},
    {"ROFL_FEMUL_TABLE", "2097152", "rofl_bench_femul mode 3: table entries (128 B each)"},
};
// getenv restricted to the table above
const char *knob(const char *name) {
#ifndef NDEBUG
    bool known = false; for (const Knob &k : KNOBS) known |= !strcmp(k.name, name);
    if (!known) { fprintf(stderr, "librofl_zk: unregistered knob %s\n", name); abort(); }
#endif
    return getenv(name);
}

// ---------------------------------------------------------------- error plumbing
thread_local std::string g_err;
int fail(int code, const std::string &msg) { g_err = msg; return code; }
struct HipErr { hipError_t e; const char *what; };
#define HIPCHK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) throw HipErr{e__, #x}; } while (0)

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---------------------------------------------------------------- host scalar helpers (canonical <-> Montgomery)
sc h_mont(const sc &canon) { return sc_to_mont(canon); }
sc h_canon(const sc &mont) { return sc_from_mont(mont); }
sc h_mul(const sc &a, const sc &b) { return sc_mul_plain(a, b); }          // canonical * canonical
sc h_inv(const sc &canon) { return h_canon(h51::sc_invert_mont_fast(h_mont(canon))); }
bool sc_is_canonical_bytes(const uint8_t *b) { sc s = sc_frombytes(b); return !sc_geq_l(s.v); }

// width-2 NAF (digits -1,0,1) of a canonical scalar; returns index of the highest non-zero digit (-1 if zero)
int sc_naf(int8_t out[256], const sc &k) {
    u32 x[9]; for (int i = 0; i < 8; i++) x[i] = k.v[i]; x[8] = 0;
    int top = -1;
    for (int pos = 0; pos < 256; pos++) {
        int d = 0;
        if (x[0] & 1) {
            d = 2 - (int)(x[0] & 3);          // 1 -> +1, 3 -> -1
            if (d > 0) { x[0] -= 1; }
            else { u64 c = 1; for (int i = 0; i < 9 && c; i++) { c += x[i]; x[i] = (u32)c; c >>= 32; } }
            top = pos;
        }
        out[pos] = (int8_t)d;
        for (int i = 0; i < 8; i++) x[i] = (x[i] >> 1) | (x[i + 1] << 31);
        x[8] >>= 1;
    }
    return top;
}

// width-4 NAF (digits in +-{1,3,5,7}) of a 64-bit piece; returns the highest non-zero position (-1 if zero)
// width-w NAF of a piece of at most 64 bits: odd digits |d| < 2^(w-1), at most one non-zero among w consecutive positions
int wnaf_u64(int8_t out[FOLD_TAB_DIGITS], u64 piece, unsigned w) {
    unsigned __int128 k = piece; int top = -1;
    const int full = 1 << w, half = 1 << (w - 1);
    for (int pos = 0; pos < FOLD_TAB_DIGITS; pos++) {
        int d = 0;
        if (k & 1) { d = (int)(k & (unsigned)(full - 1)); if (d >= half) d -= full; if (d >= 0) k -= (unsigned)d; else k += (unsigned)(-d); top = pos; }
        out[pos] = (int8_t)d; k >>= 1;
    }
    return top;
}

// ---------------------------------------------------------------- host point helpers
using h51::ge5; using h51::niels5;
struct HostTables { std::vector<niels> B, Bb; std::vector<niels5> B5, Bb5; ge base, bblind; };

void build_fixed_table(std::vector<niels> &tab, ge P) {
    tab.resize(64 * 8);
    for (int w = 0; w < 64; w++) {
        ge acc = P;
        for (int e = 0; e < 8; e++) {
            tab[w * 8 + e] = ge_to_niels(acc);
            acc = ge_add(acc, P);
        }
        for (int k = 0; k < 4; k++) P = ge_double(P);
    }
}
ge5 h_fixed_mul(const std::vector<niels5> &tab, const sc &k_canon) {
    ge5 acc = h51::identity();
    int carry = 0;
    for (int i = 0; i < 64; i++) {
        int v = (int)((k_canon.v[i >> 3] >> ((i & 7) * 4)) & 15) + carry;
        carry = (v + 8) >> 4;
        int d = v - (carry << 4);
        if (d > 0) acc = h51::gmadd(acc, tab[i * 8 + d - 1], false);
        else if (d < 0) acc = h51::gmadd(acc, tab[i * 8 - d - 1], true);
    }
    return acc;
}
void to_tab5(std::vector<niels5> &o, const std::vector<niels> &t) { o.resize(t.size()); for (size_t i = 0; i < t.size(); i++) o[i] = h51::from_niels(t[i]); }
ge h_fixed_mul32(const std::vector<niels> &tab, const sc &k_canon) {
    ge acc = ge_identity();
    int carry = 0;
    for (int i = 0; i < 64; i++) {
        int v = (int)((k_canon.v[i >> 3] >> ((i & 7) * 4)) & 15) + carry;
        carry = (v + 8) >> 4;
        int d = v - (carry << 4);
        if (d > 0) acc = ge_madd(acc, tab[i * 8 + d - 1], false);
        else if (d < 0) acc = ge_madd(acc, tab[i * 8 - d - 1], true);
    }
    return acc;
}

// cores this process may use: the affinity mask, capped by the cgroup v2 CPU quota (the GPU boxes show 256 CPUs under a 16-core quota)
int usable_cores() {
    int n = (int)std::thread::hardware_concurrency(); if (n < 1) n = 1;
    cpu_set_t set; CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0) { int c = CPU_COUNT(&set); if (c > 0 && c < n) n = c; }
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64]; long per = 0;
        if (fscanf(f, "%63s %ld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) { long c = atol(q) / per; if (c >= 1 && c < n) n = (int)c; }
        fclose(f);
    }
    return n;
}
// ---------------------------------------------------------------- small host thread pool
// The per-round host tails (one Horner chain + transcript per chunk) are independent across chunks.
class HostPool {
    std::vector<std::thread> workers; std::mutex mu; std::condition_variable cv;
    std::function<void(size_t)> fn; std::atomic<size_t> count{0}, next{0}, done{0}; std::atomic<int> active{0}; std::atomic<uint64_t> gen{0}; bool stop = false;
    double spin_us = 400.0;      // ROFL_POOL_SPIN_US: how long an idle worker polls for the next job before it sleeps
    const std::atomic<int> *calls_in_flight = nullptr;      // polling is for a call that is alone on the device: with several in flight the pools of the lanes would fight over the cores
    // An index is claimed only after it has been checked against `count` (compare-and-swap, not fetch-add): a worker that wakes
    // late and arrives while run() is resetting the job (count == 0 in that window) must not consume an index of the next job --
    // with a blind fetch-add it could take index 0 between `next = 0` and `count = n` and drop it, and run() would wait forever.
    void work() {
        active.fetch_add(1);
        for (;;) {
            size_t i = next.load();
            if (i >= count.load()) break;
            if (!next.compare_exchange_weak(i, i + 1)) continue;
            fn(i); done.fetch_add(1);
        }
        active.fetch_sub(1);
    }
    // A worker that has just finished a job polls for the next one for a short while before it sleeps: the hops of a proof follow each
    // other at 0.1-0.3 ms, and a sleeping thread has to be put back on a CPU by the scheduler first -- on a busy host (the GPU boxes run
    // at a load average above 20) that wake-up is where multi-millisecond outliers of a 25 ms proof came from.
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            if (spin_us > 0 && (!calls_in_flight || calls_in_flight->load(std::memory_order_relaxed) <= 1)) {
                auto t0 = std::chrono::steady_clock::now();
                while (gen.load(std::memory_order_acquire) == seen) {
                    __builtin_ia32_pause();
                    if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > spin_us) break;
                }
            }
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || gen.load() != seen; }); if (stop) return; seen = gen.load(); }
            work();
        }
    }
public:
    explicit HostPool(int nthreads, const std::atomic<int> *in_flight = nullptr) : calls_in_flight(in_flight) {
        if (const char *e = knob("ROFL_POOL_SPIN_US")) spin_us = atof(e);
        for (int i = 1; i < nthreads; i++) workers.emplace_back([this] { loop(); });
    }
    ~HostPool() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (auto &t : workers) t.join(); }
    void run(size_t n, std::function<void(size_t)> f) {
        if (n <= 1 || workers.empty()) { for (size_t i = 0; i < n; i++) f(i); return; }
        while (active.load() > 0) std::this_thread::yield();     // no straggler of the previous job may still look at fn
        { std::lock_guard<std::mutex> lk(mu); count = 0; fn = std::move(f); done = 0; next = 0; count = n; gen++; }
        cv.notify_all();
        work();
        while (done.load() < n) std::this_thread::yield();
        // workers that woke late see next >= count and go back to sleep; make sure none is still inside work()
        // with a stale fn before the next run() replaces it: done == n implies every claimed index finished.
    }
};

// ---------------------------------------------------------------- device buffers
struct DevBuf {
    void *p = nullptr; size_t cap = 0;
    void *ensure(size_t bytes) {
        if (bytes > cap) {
            if (p) HIPCHK(hipFree(p));
            p = nullptr; cap = 0;
            size_t want = bytes + bytes / 8 + 256;
            HIPCHK(hipMalloc(&p, want)); cap = want;
        }
        return p;
    }
    template <class T> T *as(size_t count) { return reinterpret_cast<T *>(ensure(count * sizeof(T))); }
};
// Pinned host memory that kernels can address directly (mapped, coherent): small per-round results go from the kernels straight
// into it and per-round challenges are read from it -- no hipMemcpyAsync on the hop (each costs 10-15 us of host time).
struct PinBuf {
    void *p = nullptr, *dp = nullptr; size_t cap = 0;
    void *ensure(size_t bytes) {
        if (bytes > cap) {
            if (p) HIPCHK(hipHostFree(p));
            p = nullptr; dp = nullptr; cap = 0;
            HIPCHK(hipHostMalloc(&p, bytes + 256, hipHostMallocMapped)); cap = bytes + 256;
            HIPCHK(hipHostGetDevicePointer(&dp, p, 0));
        }
        return p;
    }
    template <class T> T *as(size_t count) { return reinterpret_cast<T *>(ensure(count * sizeof(T))); }
    template <class T> T *dev(size_t count) { ensure(count * sizeof(T)); return reinterpret_cast<T *>(dp); }      // the same memory, as the device sees it
};

struct Timing {
    bool enabled = false;      // the full instrumentation: per-kind spans, accumulate / fold events, first / last
    bool acc_only = false;     // only the spans of the fixed-base accumulation (rofl_set_timing(2)): ten event records per proof instead of ~150
    rofl_timing_t t{};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> acc_ev, fold_ev;
    std::vector<std::string> acc_tag, fold_tag;
    struct KEv { int kind; hipEvent_t e0, e1; uint64_t fe_muls, bytes; };
    std::vector<KEv> kev;                                  // per-kernel-kind spans (rofl_last_kernel_times)
    rofl_kernel_time_t kt[ROFL_TK_COUNT]{};
    hipEvent_t first = nullptr, last = nullptr;
    std::vector<hipEvent_t> pool; size_t used = 0;
    hipEvent_t get() {
        if (used == pool.size()) { hipEvent_t e; HIPCHK(hipEventCreate(&e)); pool.push_back(e); }
        return pool[used++];
    }
    void reset() { t = rofl_timing_t{}; acc_ev.clear(); fold_ev.clear(); acc_tag.clear(); fold_tag.clear(); kev.clear(); for (auto &k : kt) k = rofl_kernel_time_t{}; used = 0; first = last = nullptr; }
};
// HIP events around the launches of one kernel kind, with the algorithmic work of those launches (field multiplications: 7 per
// mixed addition, 8 per doubling, 9 per extended addition; bytes: the data the launch has to read and write at least once)
struct KSpan {
    Timing *tm = nullptr; hipStream_t s = nullptr; size_t idx = 0;
    KSpan(Timing &t, hipStream_t st, int kind, uint64_t fe_muls, uint64_t bytes) {
        if (!t.enabled && !(t.acc_only && kind == ROFL_TK_MSM_ACCUMULATE_FB)) return;
        tm = &t; s = st; idx = t.kev.size();
        t.kev.push_back(Timing::KEv{kind, t.get(), t.get(), fe_muls, bytes});
        HIPCHK(hipEventRecord(t.kev[idx].e0, s));
    }
    ~KSpan() { if (tm) (void)hipEventRecord(tm->kev[idx].e1, s); }
    KSpan(const KSpan &) = delete; KSpan &operator=(const KSpan &) = delete;
};

// Workspace of one MSM in flight on a lane (a lane holds two: the verifier queues its two MSMs behind one synchronisation)
struct MsmWs {
    DevBuf cnt, off, cur, tail, perm, sorted, ovf, buckets, S[2], Cacc[2], probs;
    PinBuf h_res, h_ovf, h_probs;
    std::vector<MsmProb> probs_on_dev;      // what `probs` holds on the device: an unchanged problem list is not uploaded again
};

// One cached BulletproofGens::new(n, m): [G(N) | H(N)] + fold slices (tbl), the 16 window slices of the fixed-base MSM (wtab)
struct GensEntry { niels *tbl = nullptr; ndm *wtab = nullptr, *wtab_many = nullptr; u32 wc = 16, wc_many = 0; FoldTabCfg fc{}; size_t bytes = 0, n = 0, m = 0; u64 tick = 0; int users = 0; };

// One Ctx = one "lane": a HIP stream with its own workspace, staging buffers, timing and host pool.  The primary
// lane of a device owns the shared read-only state (fixed-base tables, generator cache).  An API call runs on one
// lane; concurrent calls from different host threads (the reference's server verifies clients from a thread pool,
// server.rs:656-687) take different lanes, so the latency-bound phases of one call (small rounds, host Horner,
// transcripts) overlap the throughput-bound phases of another.  ROFL_LANES = size of the pool.
struct Ctx {
    int device = 0;
    bool inited = false;
    Ctx *parent = nullptr;
    std::vector<Ctx *> sibs;      // additional lanes
    int nlanes = 3;      // ROFL_LANES: number of calls that can be in flight on this device
    std::mutex init_mu, gens_mu;      // primary lane only: one-time initialisation; generator-table cache
    std::atomic<int> active_calls{0};  // primary lane only: calls currently holding a lane
    std::atomic<unsigned> rr{0};
    hipStream_t stream = nullptr, stream2 = nullptr;      // stream2: side stream for work that may run beside the main one (created on first use)
    std::mutex mu;
    HostTables ht;
    niels *d_tabB = nullptr, *d_tabBb = nullptr;
    sc *d_two_pow = nullptr;
    std::map<std::pair<size_t, size_t>, std::unique_ptr<GensEntry>> gens;   // (n, m) -> tables; primary lane only, under gens_mu
    u64 gens_tick = 0; size_t gens_budget = (size_t)96 << 30;   // ROFL_GENS_BUDGET_MB: evict least recently used (unpinned) tables beyond this
    u32 fold_pb = 32, fold_w = 8; size_t fold_tab_budget = (size_t)32 << 30;   // widest NAF whose table fits the per-(n, m) budget (ROFL_FOLD_W, ROFL_FOLD_TAB_MB)
    int msm_lds = 1, msm_two_level = 1, msm_group_reduce = 0; size_t msm_lds_min = 8192, msm_lds_tile = 131072;
    size_t msm_fb_threads = (size_t)1 << 19;
    int msm_fb = 1; size_t msm_fb_min = (size_t)1 << 12; int msm_lr = 1;   // window tables for every (n, m) with 2N >= 4096: many small chunks (n_partition = 64) share them
    bool crowded() const { const Ctx *P = parent ? parent : this; return P->active_calls.load() > 1; }   // other calls in flight on this device
    // Waiting for the lane's stream.  hipStreamSynchronize spins (lowest latency: right for a call that is alone on the device); with
    // more than three calls in flight -- or when the host asked for it (ROFL_BLOCKING_SYNC=1) -- the thread sleeps between queries instead, so
    // a server that keeps several clients in flight does not burn one host core per client on busy-waiting (ROFL_BLOCKING_SYNC=0: always spin).
    // behaviour options (rofl_set_option; primary lane only -- the lanes read their parent's): the environment only provides defaults
    int opt_zip_truncate = 0, opt_verify_batch = 1, opt_sigma_batch = 1;
    int blocking_sync = -1; hipEvent_t ev_block = nullptr, ev_v = nullptr, ev_fork = nullptr, ev_a = nullptr, ev_a0 = nullptr, ev_m2 = nullptr, ev_m2j = nullptr; bool batch_mode = false;
    void sync() {
        const Ctx *P = parent ? parent : this;
        // (up to three calls in flight still spin: the three proofs of ONE client's L2 update run side by side -- EncParamsL2::encrypt --
        //  and that is a latency case; a server with more clients in flight is a throughput case)
        bool block = P->blocking_sync == 1 || (P->blocking_sync < 0 && (P->active_calls.load() > 3 || batch_mode));
        if (!block) { HIPCHK(hipStreamSynchronize(stream)); return; }
        // (hipEventSynchronize on a hipEventBlockingSync event still keeps the calling thread runnable on this runtime -- measured: 100 %
        //  of a core either way -- so the wait is a query loop with short sleeps: ~50 us of extra latency per wait, no CPU)
        if (!ev_block) HIPCHK(hipEventCreateWithFlags(&ev_block, hipEventDisableTiming));
        HIPCHK(hipEventRecord(ev_block, stream));
        for (;;) {
            hipError_t q = hipEventQuery(ev_block);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) throw HipErr{q, "hipEventQuery"};
            struct timespec ts = {0, 20000}; nanosleep(&ts, nullptr);
        }
    }
    struct Bsgs { uint8_t *keys; u32 *slots; u32 mask; };
    std::map<size_t, Bsgs> bsgs;                          // table_size -> baby-step table
    std::unique_ptr<HostPool> pool;
    size_t fold_min = 1024;
    size_t msm_small_max = 8192;      // ROFL_MSM_SMALL_MAX: generic MSMs with at most this many terms per problem side run as one fused launch (0 = off)
    size_t msm_dev_horner_min = 32;   // ROFL_MSM_DEV_HORNER_MIN: launches with at least this many problems finish their Horner chains on the device
    bool msm_slots = true;
    int fold_t = 2, fold_t1 = 3, fold_k = 0, fold_tab = 1, fold_unit = 1; long fold_threads = 131072;
    Timing tm;
    struct HopStats { double enqueue = 0, sync = 0, horner_wall = 0, horner_cpu = 0, host_wall = 0, host_cpu = 0; int n = 0; } hs;      // ROFL_TRACE: where the host hops go
    // workspace
    DevBuf cp, sL, sR, party, Scanon, vshift, blind, Vbytes, Cbytes, status, partial, partial2, scpart, a, b, a2, b2, ptab[2], yinv,
        SL, SR, powtabs, foldprobs, naf,
        gbuf[2], aux_pts, aux_scal, vscal, tmp_in, tmp_in2, tmp_out, vals, uni, stream_buf;
    PinBuf h_cp, h_part, h_misc, h_misc2, h_auxc, h_auxs, h_V, h_ip, h_round, h_fdig, h_fprob, h_abfin;
    MsmWs mws[2];

    void init() {
        if (inited) return;
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        // PedersenGens::default(): B = Ristretto basepoint, B_blinding = hash_from_bytes::<Sha3_512>(B)
        static const uint8_t Bc[32] = {0xe2, 0xf2, 0xae, 0x0a, 0x6a, 0xbc, 0x4e, 0x71, 0xa8, 0x84, 0xa9, 0x61, 0xc5, 0x00, 0x51, 0x5f,
                                       0x58, 0xe3, 0x0b, 0x6a, 0xa5, 0x82, 0xdd, 0x8d, 0xb6, 0xa6, 0x59, 0x45, 0xe0, 0x8d, 0x2d, 0x76};
        ristretto_decode(ht.base, Bc);
        uint8_t h[64]; sha3_512(h, Bc, 32);
        ht.bblind = ristretto_from_uniform(h);
        build_fixed_table(ht.B, ht.base);
        build_fixed_table(ht.Bb, ht.bblind);
        to_tab5(ht.B5, ht.B); to_tab5(ht.Bb5, ht.Bb);
        HIPCHK(hipFuncSetAttribute((const void *)k_msm_reduce_fused, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void *)k_msm_small, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));      // + 2.5 KB of static LDS (bucket order)
        HIPCHK(hipFuncSetAttribute((const void *)k_msm_scatter_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        HIPCHK(hipFuncSetAttribute((const void *)k_msm_reduce_groups, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        HIPCHK(hipFuncSetAttribute((const void *)k_msm_bin_l1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIPCHK(hipFuncSetAttribute((const void *)k_msm_bin_l2, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        HIPCHK(hipMalloc(&d_tabB, sizeof(niels) * 512));
        HIPCHK(hipMalloc(&d_tabBb, sizeof(niels) * 512));
        HIPCHK(hipMemcpy(d_tabB, ht.B.data(), sizeof(niels) * 512, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_tabBb, ht.Bb.data(), sizeof(niels) * 512, hipMemcpyHostToDevice));
        sc tp[64]; sc two = h_mont(sc_from_u64(2)); tp[0] = sc_one_mont();
        for (int i = 1; i < 64; i++) tp[i] = sc_montmul(tp[i - 1], two);
        HIPCHK(hipMalloc(&d_two_pow, sizeof(tp)));
        HIPCHK(hipMemcpy(d_two_pow, tp, sizeof(tp), hipMemcpyHostToDevice));
        if (const char *e = knob("ROFL_FOLD_T")) { int v = atoi(e); if (v >= 1 && v <= 6) fold_t = v; }
        if (const char *e = knob("ROFL_MSM_SLOTS")) msm_slots = atoi(e) != 0;
        if (const char *e = knob("ROFL_MSM_FB")) msm_fb = atoi(e);
        if (const char *e = knob("ROFL_MSM_FB_THREADS")) { long v = atol(e); if (v >= 1) msm_fb_threads = (size_t)v; }
        if (const char *e = knob("ROFL_MSM_LDS")) msm_lds = atoi(e);
        if (const char *e = knob("ROFL_MSM_TWO_LEVEL")) msm_two_level = atoi(e);
        if (const char *e = knob("ROFL_MSM_GROUP_REDUCE")) msm_group_reduce = atoi(e);
        if (const char *e = knob("ROFL_MSM_LDS_MIN")) { long v = atol(e); if (v >= 1) msm_lds_min = (size_t)v; }
        if (const char *e = knob("ROFL_MSM_LDS_TILE")) { long v = atol(e); if (v >= 1024) msm_lds_tile = (size_t)v; }
        if (const char *e = knob("ROFL_MSM_LR")) msm_lr = atoi(e);
        if (const char *e = knob("ROFL_MSM_FB_MIN")) { long v = atol(e); if (v >= 1) msm_fb_min = (size_t)v; }
        if (const char *e = knob("ROFL_FOLD_MIN")) { long v = atol(e); if (v >= 1) fold_min = (size_t)v; }
        if (const char *e = knob("ROFL_MSM_DEV_HORNER_MIN")) { long v = atol(e); if (v >= 1) msm_dev_horner_min = (size_t)v; }
        if (const char *e = knob("ROFL_MSM_SMALL_MAX")) { long v = atol(e); if (v >= 0) msm_small_max = (size_t)v; }
        {   // host pool of the primary lane: the per-round tails of many chunks (n_partition = 64: 128 window combinations, 128 encodings,
            // 64 transcripts per round) scale with it -- 8 -> 14 threads took 3 ms off a 35 ms proof.  Default: the cores this process may
            // use (affinity mask capped by the cgroup CPU quota; the calling thread is one of the pool's executors), within [2, 16].
            int nt = std::min(16, std::max(2, usable_cores()));
            if (const char *e = knob("ROFL_HOST_THREADS")) nt = atoi(e);
            if (nt < 1) nt = 1; if (nt > 64) nt = 64;
            pool.reset(new HostPool(nt, &active_calls)); }
        if (const char *e = knob("ROFL_FOLD_T1")) { int v = atoi(e); if (v >= 1 && v <= 6) fold_t1 = v; }
        if (const char *e = knob("ROFL_FOLD_TAB")) fold_tab = atoi(e) != 0;
        if (const char *e = knob("ROFL_GENS_BUDGET_MB")) { long v = atol(e); if (v >= 1) gens_budget = (size_t)v << 20; }
        if (const char *e = knob("ROFL_FOLD_PB")) { int v = atoi(e); if (v == 16 || v == 32 || v == 64) fold_pb = (u32)v; }
        if (const char *e = knob("ROFL_FOLD_W")) { int v = atoi(e); if (v >= 3 && v <= 8) fold_w = (u32)v; }
        if (const char *e = knob("ROFL_FOLD_TAB_MB")) { long v = atol(e); if (v >= 1) fold_tab_budget = (size_t)v << 20; }
        if (const char *e = knob("ROFL_FOLD_UNIT")) fold_unit = atoi(e) != 0;
        if (const char *e = knob("ROFL_FOLD_K")) { int v = atoi(e); if (v == 1 || v == 2 || v == 4) fold_k = v; }
        if (const char *e = knob("ROFL_FOLD_THREADS")) { long v = atol(e); if (v > 0) fold_threads = v; }
        if (const char *e = knob("ROFL_LANES")) { int v = atoi(e); if (v >= 1 && v <= 8) nlanes = v; }
        if (const char *e = knob("ROFL_BLOCKING_SYNC")) blocking_sync = atoi(e) != 0;
        if (const char *e = knob("ROFL_VERIFY_ZIP_TRUNCATE")) opt_zip_truncate = atoi(e) != 0;
        if (const char *e = knob("ROFL_VERIFY_BATCH")) opt_verify_batch = atoi(e) != 0;
        if (const char *e = knob("ROFL_SIGMA_BATCH")) opt_sigma_batch = atoi(e) != 0;
        inited = true;
        for (int i = 1; i < nlanes; i++) { Ctx *s = new Ctx(); s->init_lane(*this); sibs.push_back(s); }
    }
    void init_lane(Ctx &p) {
        parent = &p; device = p.device;
        HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        ht = p.ht; d_tabB = p.d_tabB; d_tabBb = p.d_tabBb; d_two_pow = p.d_two_pow;
        msm_lds = p.msm_lds; msm_two_level = p.msm_two_level; msm_group_reduce = p.msm_group_reduce; msm_lds_min = p.msm_lds_min; msm_lds_tile = p.msm_lds_tile;
        msm_fb_threads = p.msm_fb_threads;
        msm_fb = p.msm_fb; msm_fb_min = p.msm_fb_min; msm_lr = p.msm_lr;
        fold_min = p.fold_min; msm_dev_horner_min = p.msm_dev_horner_min; msm_small_max = p.msm_small_max; msm_slots = p.msm_slots; fold_t = p.fold_t; fold_t1 = p.fold_t1; fold_k = p.fold_k; fold_tab = p.fold_tab;
        fold_unit = p.fold_unit; fold_threads = p.fold_threads; nlanes = 1;
        { int nt = 6; if (const char *e = knob("ROFL_HOST_THREADS")) nt = atoi(e); if (nt < 1) nt = 1; pool.reset(new HostPool(nt, &p.active_calls)); }
        inited = true;
    }
};

// NB the HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and reads the variable once, when
// it initialises; lanes that share a queue serialise each other's kernels.  The library does not touch the process environment:
// a host that wants more than four calls in flight exports GPU_MAX_HW_QUEUES itself before the first HIP call.

std::mutex g_ctx_mu;
std::map<int, Ctx *> g_ctxs;
int g_device = 0;
Ctx &ctx() {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    auto it = g_ctxs.find(g_device);
    if (it == g_ctxs.end()) { Ctx *c = new Ctx(); c->device = g_device; it = g_ctxs.emplace(g_device, c).first; }
    return *it->second;
}

// A lane held for the duration of one API call.
struct LaneLock {
    Ctx *c = nullptr; Ctx *primary = nullptr; std::unique_lock<std::mutex> lk;
    LaneLock() = default;
    LaneLock(LaneLock &&o) noexcept : c(o.c), primary(o.primary), lk(std::move(o.lk)) { o.c = nullptr; o.primary = nullptr; }
    ~LaneLock() { if (primary) primary->active_calls.fetch_sub(1); }
};
LaneLock acquire_lane(bool primary_only = false) {
    Ctx &P = ctx();
    { std::lock_guard<std::mutex> g(P.init_mu); P.init(); }
    HIPCHK(hipSetDevice(P.device));                    // the calling thread may be new to HIP
    LaneLock ll; ll.primary = &P; P.active_calls.fetch_add(1);
    size_t L = primary_only ? 1 : 1 + P.sibs.size();
    for (size_t i = 0; i < L; i++) {
        Ctx *c = i ? P.sibs[i - 1] : &P;
        std::unique_lock<std::mutex> t(c->mu, std::try_to_lock);
        if (t.owns_lock()) { ll.c = c; ll.lk = std::move(t); c->batch_mode = false; return ll; }
    }
    size_t i = primary_only ? 0 : P.rr.fetch_add(1) % L;    // all busy: queue on one of them
    Ctx *c = i ? P.sibs[i - 1] : &P;
    ll.lk = std::unique_lock<std::mutex>(c->mu); ll.c = c; c->batch_mode = false;
    return ll;
}

inline dim3 grid1(size_t n, u32 y = 1) { return dim3((unsigned)((n + TPB - 1) / TPB), y, 1); }
unsigned lg2u(size_t x) { unsigned r = 0; while (((size_t)1 << r) < x) r++; return r; }
bool is_pow2(size_t x) { return x && !(x & (x - 1)); }
size_t next_pow2(size_t val) { if (val == 1) return 1; size_t n = val - 1; while ((n & (n - 1)) != 0) n &= n - 1; return n << 1; }

// ---------------------------------------------------------------- generators
struct MsmPlan { u32 c, W, B, levels, wide; };
MsmPlan msm_plan_c(u32 c) {
    MsmPlan p; p.c = c;
    p.W = (253 - p.c + p.c - 1) / p.c + 1;                // top window [253-c, 254) + ceil((253-c)/c) lower windows
    p.wide = (253 - p.c) - (p.c - 1) * (p.W - 1);          // wide*c + (W-1-wide)*(c-1) = 253 - c
    p.B = 1u << (p.c - 1);
    p.levels = (p.c - 1) / 3;
    return p;
}
// 0 = the default layouts (16-bit windows; small generator sets also get a 15-bit table for launches with many problems)
u32 fb_window_c(size_t gens) {
    static const int force = knob("ROFL_MSM_FB_C") ? atoi(knob("ROFL_MSM_FB_C")) : 0;      // tuning: one layout (13, 15 or 16) for every table and launch
    (void)gens;
    return (force == 13 || force == 15 || force == 16) ? (u32)force : 0u;
}
// Generator-table cache, shared by the lanes of a device.  An entry is pinned (users > 0) for the duration of every call that
// reads it; eviction (LRU, beyond gens_budget, or to make room after a failed hipMalloc) only ever frees unpinned entries, so it is
// safe with any number of calls in flight.  (n, m) reach this point from untrusted wire messages: callers validate the proof
// format against (n, m) BEFORE asking for tables, and an allocation failure degrades (evict, then the compact table layout, then
// no window table) instead of leaving the device full.
void gens_free_entry(GensEntry *e) {
    if (e->wtab) (void)hipFree(e->wtab);
    if (e->wtab_many) (void)hipFree(e->wtab_many);
    e->wtab_many = nullptr;
    if (e->tbl) (void)hipFree(e->tbl);
    e->wtab = nullptr; e->tbl = nullptr;
}
// caller holds gens_mu.  Frees unpinned entries, least recently used first, until `keep_bytes` or less are held.
void gens_evict(Ctx &P0, size_t keep_bytes, const GensEntry *spare) {
    for (;;) {
        size_t total = 0; for (auto &kv : P0.gens) total += kv.second->bytes;
        if (total <= keep_bytes) return;
        auto victim = P0.gens.end();
        for (auto it = P0.gens.begin(); it != P0.gens.end(); ++it)
            if (it->second.get() != spare && it->second->users == 0 && (victim == P0.gens.end() || it->second->tick < victim->second->tick)) victim = it;
        if (victim == P0.gens.end()) return;       // everything left is in use
        gens_free_entry(victim->second.get());
        P0.gens.erase(victim);
    }
}
hipError_t gens_malloc(Ctx &P0, void **p, size_t bytes, const GensEntry *spare) {
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess) return e;
    (void)hipGetLastError();
    gens_evict(P0, 0, spare);                      // drop every table nobody is reading, then try once more
    e = hipMalloc(p, bytes);
    if (e != hipSuccess) (void)hipGetLastError();
    return e;
}
struct GensPin {
    Ctx *P0 = nullptr; GensEntry *e = nullptr;
    GensPin() = default;
    GensPin(Ctx *p, GensEntry *en) : P0(p), e(en) {}
    GensPin(GensPin &&o) noexcept : P0(o.P0), e(o.e) { o.P0 = nullptr; o.e = nullptr; }
    GensPin &operator=(GensPin &&o) noexcept { release(); P0 = o.P0; e = o.e; o.P0 = nullptr; o.e = nullptr; return *this; }
    GensPin(const GensPin &) = delete; GensPin &operator=(const GensPin &) = delete;
    ~GensPin() { release(); }
    void release() { if (e) { std::lock_guard<std::mutex> lk(P0->gens_mu); e->users--; } e = nullptr; }
    niels *tbl() const { return e->tbl; }
    const niels *wtab() const { return reinterpret_cast<const niels *>(e->wtab); }      // opaque to the host: 128-byte ndm records
    u32 wc() const { return e->wc; }                                                    // window width of the window table's layout
    // the table to use for a launch of `problems` bucket-array owners: small generator sets carry a second, 15-bit layout for launches with
    // many problems (n_partition = 64), where 16-bit windows would spread a handful of entries per bucket over millions of buckets
    void fb_for(size_t problems, const niels **tab, u32 *c) const {
        if (e->wtab_many && problems >= 32) { *tab = reinterpret_cast<const niels *>(e->wtab_many); *c = e->wc_many; }
        else { *tab = reinterpret_cast<const niels *>(e->wtab); *c = e->wc; }
    }
    const FoldTabCfg &fc() const { return e->fc; }
};
GensPin get_gens(Ctx &C, size_t n, size_t m) {
    Ctx &P0 = C.parent ? *C.parent : C;
    std::lock_guard<std::mutex> gens_lock(P0.gens_mu);
    auto key = std::make_pair(n, m);
    auto it = P0.gens.find(key);
    if (it != P0.gens.end()) { it->second->tick = ++P0.gens_tick; it->second->users++; return GensPin(&P0, it->second.get()); }
    size_t N = n * m;
    std::unique_ptr<GensEntry> ent(new GensEntry());
    FoldTabCfg fc{P0.fold_pb, P0.fold_w, 256 / P0.fold_pb, 1u << (P0.fold_w - 2)};
    // HBM capacity for VALU work: a width-w NAF needs 2^(w-2) odd multiples per piece and leaves 1/(w+1) of the digits non-zero
    while (fc.w > 6 && sizeof(niels) * 2 * N * fc.np * fc.e > P0.fold_tab_budget) { fc.w--; fc.e = 1u << (fc.w - 2); }
    if (sizeof(niels) * 2 * N * fc.np * fc.e > ((size_t)40 << 30)) fc = FoldTabCfg{64, 4, 4, 4};     // very large tables: the compact layout
    void *tblv = nullptr;
    hipError_t me = gens_malloc(P0, &tblv, sizeof(niels) * 2 * N * fc.np * fc.e, nullptr);      // slice 0 = generators, the rest = fold tables
    if (me != hipSuccess && !(fc.pb == 64 && fc.w == 4)) {      // HBM is short even after eviction: the compact fold-table layout (16 slices)
        fc = FoldTabCfg{64, 4, 4, 4};
        me = gens_malloc(P0, &tblv, sizeof(niels) * 2 * N * fc.np * fc.e, nullptr);
    }
    if (me != hipSuccess) throw HipErr{me, "hipMalloc(generator tables)"};
    niels *tbl = reinterpret_cast<niels *>(tblv);
    ent->tbl = tbl; ent->fc = fc; ent->n = n; ent->m = m;
    ent->bytes = sizeof(niels) * 2 * N * fc.np * fc.e;
    try {
        uint8_t *uni = C.uni.as<uint8_t>(2 * N * 64);
        hipLaunchKernelGGL(k_gens_xof, grid1(2 * m), dim3(TPB), 0, C.stream, (u32)n, (u32)m, uni);
        hipLaunchKernelGGL(k_gens_map, grid1(2 * N), dim3(TPB), 0, C.stream, (u32)(2 * N), uni, tbl);
        hipLaunchKernelGGL(k_gens_tables, grid1(2 * N * fc.np), dim3(TPB), 0, C.stream, (u32)(2 * N), fc, tbl, (size_t)(2 * N));
        // Window table of the fixed-base MSM.  Its window width follows the size of the generator set: 16-bit windows (16 slices, 32 768
        // buckets per set) from 2^17 generators on; 13-bit windows (20 slices, 4 096 buckets) below -- many small chunks (n_partition = 64:
        // 128 L / R problems of 16 384 terms per round) would otherwise spread 8 entries per bucket over 4 M buckets, and the bucket
        // reduction, not the accumulation, was the cost of such a launch.
        const u32 c_force = fb_window_c(2 * N);
        MsmPlan fp = msm_plan_c(c_force ? c_force : 16);
        if (P0.msm_fb && 2 * N >= P0.msm_fb_min && 2 * N * fp.W < ((size_t)1 << 31)) {      // entry index (w * 2N + i) must fit 31 bits
            void *wtv = nullptr;
            if (gens_malloc(P0, &wtv, sizeof(ndm) * 2 * N * fp.W, ent.get()) == hipSuccess) {      // without it the MSMs over these generators run in generic mode
                ndm *wt = reinterpret_cast<ndm *>(wtv);
                hipLaunchKernelGGL(k_gens_wtab, grid1(2 * N), dim3(TPB), 0, C.stream, (u32)(2 * N), MsmWin{fp.c, fp.W, fp.wide}, tbl, wt, (size_t)(2 * N));
                ent->wtab = wt; ent->wc = fp.c; ent->bytes += sizeof(ndm) * 2 * N * fp.W;
                // Small generator sets also get a 15-bit layout (17 slices; 71 MB at 2N = 32 768).  Many small chunks (n_partition = 64: 128 L / R
                // problems of 16 384 terms per round) spread 8 entries per bucket over 4 M buckets at c = 16 and the bucket REDUCTION (0.85 ms
                // per round) rivals the accumulation; 15-bit windows halve the buckets for one more window: 3.2 -> 2.8 ms per round.  (13-bit
                // windows were measured too: their 4 096-bucket arrays fall off the two-level sort and the narrow windows unbalance the
                // lists -- 3.5 ms per round.)  A client with FEW chunks of this size (cfg 1: four chunks of 2 048 8-bit values) keeps c = 16.
                if (!c_force && 2 * N < ((size_t)1 << 17)) {
                    MsmPlan f2 = msm_plan_c(15);
                    void *w2 = nullptr;
                    if (gens_malloc(P0, &w2, sizeof(ndm) * 2 * N * f2.W, ent.get()) == hipSuccess) {
                        hipLaunchKernelGGL(k_gens_wtab, grid1(2 * N), dim3(TPB), 0, C.stream, (u32)(2 * N), MsmWin{f2.c, f2.W, f2.wide}, tbl, reinterpret_cast<ndm *>(w2), (size_t)(2 * N));
                        ent->wtab_many = reinterpret_cast<ndm *>(w2); ent->wc_many = f2.c; ent->bytes += sizeof(ndm) * 2 * N * f2.W;
                    }
                }
            }
        }
        C.sync();
    } catch (...) { gens_free_entry(ent.get()); throw; }
    ent->tick = ++P0.gens_tick; ent->users = 1;
    GensEntry *raw = ent.get();
    P0.gens[key] = std::move(ent);
    gens_evict(P0, P0.gens_budget, raw);            // keep the cache inside its HBM budget (unpinned entries only)
    return GensPin(&P0, raw);
}

// ---------------------------------------------------------------- MSM driver
MsmPlan msm_plan(size_t n) {
    MsmPlan p;
    static const size_t t13 = knob("ROFL_MSM_T13") ? (size_t)atol(knob("ROFL_MSM_T13")) : ((size_t)1 << 13);
    static const size_t t10 = knob("ROFL_MSM_T10") ? (size_t)atol(knob("ROFL_MSM_T10")) : ((size_t)1 << 9);
    if (n >= (1u << 17)) p.c = 16; else if (n >= t13) p.c = 13; else if (n >= t10) p.c = 10; else if (n >= 64) p.c = 7; else p.c = 4;
    if (const char *e = knob("ROFL_MSM_C")) { int v = atoi(e); if (v == 4 || v == 7 || v == 10 || v == 13 || v == 16) p.c = (u32)v; }
    p.W = (254 + p.c - 1) / p.c;
    p.W = (253 - p.c + p.c - 1) / p.c + 1;                // top window [253-c, 254) + ceil((253-c)/c) lower windows
    p.wide = (253 - p.c) - (p.c - 1) * (p.W - 1);          // wide*c + (W-1-wide)*(c-1) = 253 - c
    p.B = 1u << (p.c - 1);
    p.levels = (p.c - 1) / 3;
    return p;
}
// results[p] = sum_i scal[p][i] * pts[p][i]   (all problems have n terms).
// opt.lr_nh != 0: `probs` holds (L, R) pairs that share a merged scalar array (see MsmMap); opt.fb: every problem's points
// are the generator table `opt.fb_gens` (n terms from its start, slice stride opt.fb_stride) for which a window table exists.
// overlap: host work to run while the kernels execute; post(p): runs on the pool thread that finished problem p's window combination,
// right after results[p] is final (the caller's per-problem tail -- encoding, transcript -- without a second pool hand-off on the hop)
struct MsmOpt { u32 lr_nh = 0, lr_ng = 0; const niels *fb_wtab = nullptr; size_t fb_stride = 0; u32 fb_c = 16; std::function<void()> overlap; std::function<void(size_t)> post; };

// An MSM goes through four stages: PLAN (which variant, window layout, bucket sets, capacities) -> SORT (digits into per-bucket lists)
// -> ACCUMULATE (one thread per bucket) -> REDUCE (bit-sum tree; the window combination is left to the host, or to k_msm_horner when a
// launch carries many problems).  msm_enqueue runs plan + the launches of one attempt on the lane's stream and returns an MsmJob;
// after the stream has been synchronised msm_retry says whether the attempt overflowed one of its fixed-size structures (scalars built
// to collide) and which variants are still allowed, and msm_finish turns the partial sums in mapped host memory into results.
// Variants, fastest first: fixed-base (two-level sort, or slot sort) | fused small-MSM launch | generic slot sort | count / scan / scatter.
struct MsmAllow { bool fb = true, small = true, two = true, slots = true; };
enum class MsmKind { FixedBase, Small, Slots, CountSort };
struct MsmJob {
    MsmWs *ws = nullptr; size_t np = 0, nq = 0, n = 0, PW = 0; MsmPlan P{}; MsmKind kind = MsmKind::CountSort; bool lr = false, two = false, dev_horner = false;
    bool host8 = false;      // dev_horner launches whose chains come back to the host, eight per SIMD stream (k_msm_wsum + h8::horner8)
    u32 sets = 0, cap = 0;
    bool fb() const { return kind == MsmKind::FixedBase; }
};
static const u32 MSM_OVF_MAX = 4096;

// ---- plan
bool msm_plan_job(Ctx &C, MsmJob &J, const MsmOpt &opt, const MsmAllow &al, MsmMap &mm, u32 &small_cap, Msm2L &tl) {
    const size_t np = J.np, n = J.n; const bool lr = J.lr; const size_t nq = J.nq;
    const size_t per_side = lr ? n / 2 : n;
    mm = MsmMap{opt.lr_nh, opt.lr_ng, 0, 0, 0};
    J.two = false; J.sets = 0;
    bool want_fb = al.fb && opt.fb_wtab != nullptr && C.msm_slots;
    if (want_fb) {
        J.kind = MsmKind::FixedBase;
        J.P = msm_plan_c(opt.fb_c);
        // Fewer sets = less bucket-reduction work but fewer accumulate threads.  Alone on the device the call wants the
        // threads (latency); with other calls in flight the GPU is full anyway and the work is what counts: the smallest number of
        // sets (a divisor of the window count: every set takes the same number of windows) that gives `want` accumulate threads.
        bool crowded = C.crowded();
        size_t want = crowded ? C.msm_fb_threads / 2 : C.msm_fb_threads;
        u32 sets = J.P.W;
        for (u32 sdiv = 1; sdiv <= J.P.W; sdiv++)
            if (J.P.W % sdiv == 0 && (size_t)nq * (lr ? 2 : 1) * sdiv * J.P.B >= want) { sets = sdiv; break; }      // many problems (n_partition = 64): one set each is plenty
        J.sets = sets;
        mm.fb_sets = sets; mm.fb_wps = J.P.W / sets; mm.fb_stride = (u32)opt.fb_stride;
        J.PW = nq * (lr ? 2 : 1) * sets;
        // the three (c-1)-bit windows of the layout fill only half of the buckets: twice the mean load there
        u32 cap = 16; while (cap < 2048 && (size_t)cap * J.P.B < 3 * per_side * (mm.fb_wps + 1)) cap *= 2;
        if (mm.fb_wps == 1) cap *= 2;
        J.cap = cap;
        // two-level bucket sort (coarse bins through HBM in full lines, then per-bin ranking in LDS)
        bool two = al.two && C.msm_two_level && J.P.W * opt.fb_stride <= ((size_t)1 << 24) && (J.P.B == 32768 || J.P.B == 16384);
        const u32 fb0 = J.P.B == 32768 ? 7u : 6u;      // 256 coarse bins of 128 (64) buckets
        tl = Msm2L{256, fb0, 24, 0, 144};
        if (two) {
            // a coarse bin has to fit one block's LDS in level 2: 256 bins of 128 buckets while that holds (<= 4 windows per array at
            // 2^19 terms), 512 bins of 64 buckets with half the staging row for arrays that take 8 windows (two sets per problem)
            auto size_bins = [&]() { size_t avg = per_side * mm.fb_wps / tl.nbins; tl.cap_bin = (u32)((2 * avg + 256 + 63) / 64 * 64); return (size_t)tl.cap_bin * 4 + 1024 <= 96 * 1024; };
            bool fits = size_bins();
            if (!fits) { tl = Msm2L{512, fb0 - 1, 24, 0, 72}; fits = size_bins(); }
            if (!fits || per_side < 8192) two = false;
        }
        J.two = two;
        if (!two && (size_t)J.PW * J.P.B * J.cap * 4 > ((size_t)8 << 30)) return false;      // slot array too large: next variant
        return true;
    }
    bool slots_mode = al.slots && C.msm_slots;
    J.P = msm_plan(n);
    // up to msm_small_max terms per side the fused small-MSM launch takes the problem: that wants 10-bit windows (512 buckets = one block)
    if (C.msm_small_max && al.small && slots_mode && per_side <= C.msm_small_max && J.P.c > 10 && np * 26 <= 512) J.P = msm_plan_c(10);
    J.PW = np * J.P.W;
    { u32 cap = 16; while (cap < 256 && (size_t)cap * J.P.B < 4 * n) cap *= 2; J.cap = cap; }
    // the IPP tail (a few thousand terms per problem): one launch instead of memset / scatter / scan / accumulate / overflow / reduce
    // (its blocks hold up to 130 KB of LDS, one per CU: with thousands of bucket arrays -- n_partition = 64 -- the general pipeline is faster)
    // ... unless the blocks are small: at c <= 7 a block needs < 24 KB, eight of them share a CU and thousands of arrays go through in a few batches
    small_cap = per_side <= 8 * J.P.B ? (u32)MSM_SMALL_CAP : 72u;      // list entries per bucket: mean load <= 16 / <= 32
    size_t small_lds = std::max((size_t)J.P.B * 4 * (1 + small_cap), std::max(((size_t)(J.P.B / 8) * 4 + (size_t)(J.P.B / 16) * 5 + 1) * sizeof(ge), ((size_t)J.P.B + (size_t)J.P.B * 3 / 4 + 1) * sizeof(ge)));
    bool small = slots_mode && al.small && C.msm_small_max && per_side <= C.msm_small_max && J.P.c <= 10 && per_side <= 16 * J.P.B && (J.PW <= 512 || small_lds <= 24 * 1024);
    J.kind = small ? MsmKind::Small : slots_mode ? MsmKind::Slots : MsmKind::CountSort;
    if (J.kind == MsmKind::Slots && (size_t)J.PW * J.P.B * J.cap * 4 > ((size_t)8 << 30)) return false;
    return true;
}

// ---- sort + accumulate + reduce of one attempt, enqueued on the lane's stream (no synchronisation)
MsmJob msm_enqueue(Ctx &C, MsmWs &W, const std::vector<MsmProb> &probs, size_t n, const MsmOpt &opt, MsmAllow &al, hipStream_t st = nullptr) {
    if (!st) st = C.stream;
    static const u32 acc_balance = knob("ROFL_ACC_BALANCE") ? (atoi(knob("ROFL_ACC_BALANCE")) ? 1u : 0u) : 1u;   // equal-work blocks in k_msm_accumulate (0 = plain descending order)
    static const u32 dbg_mask = knob("ROFL_DBG_IDX_MASK") ? (u32)strtoul(knob("ROFL_DBG_IDX_MASK"), nullptr, 0) : 0x7fffffffu;   // timing experiments only (wrong results): gathers confined to a cache-resident prefix
    static const u32 dbg_scatter = knob("ROFL_DBG_SCATTER") ? (u32)atoi(knob("ROFL_DBG_SCATTER")) : 0u;   // timing experiments only: 1 = no range reservation, 2 = no slot stores
    MsmJob J; J.ws = &W; J.np = probs.size(); J.n = n; J.lr = opt.lr_nh != 0; J.nq = J.lr ? J.np / 2 : J.np;
    const size_t np = J.np, nq = J.nq; const bool lr = J.lr;
    MsmMap mm{}; u32 small_cap = 0; Msm2L tl{};
    while (!msm_plan_job(C, J, opt, al, mm, small_cap, tl)) {      // a variant whose structures would not fit: the next one
        if (al.fb && opt.fb_wtab && C.msm_slots) al.fb = false; else al.slots = false;
    }
    const MsmPlan &P = J.P; const size_t PW = J.PW; const bool fb = J.fb();
    MsmProb *d_probs = W.probs.as<MsmProb>(np);
    MsmProb *h_probs = W.h_probs.as<MsmProb>(np);
    for (size_t i = 0; i < np; i++) h_probs[i] = fb ? MsmProb{opt.fb_wtab, probs[i].scal} : probs[i];
    if (W.probs_on_dev.size() != np || memcmp(W.probs_on_dev.data(), h_probs, sizeof(MsmProb) * np) != 0) {      // an unchanged problem list (constant within a fold level) is not uploaded again
        HIPCHK(hipMemcpyAsync(d_probs, h_probs, sizeof(MsmProb) * np, hipMemcpyHostToDevice, st));
        W.probs_on_dev.assign(h_probs, h_probs + np);
    }
    u32 *cnt = W.cnt.as<u32>(PW * P.B + 4), *off = W.off.as<u32>(PW * P.B), *cur = W.cur.as<u32>(PW * P.B);
    u32 *perm = W.perm.as<u32>(PW * P.B);
    ge *buckets = W.buckets.as<ge>(PW * P.B);
    MsmWin mw{P.c, P.W, P.wide};
    // results and flags go from the kernels straight into mapped host memory (no D2H copies on the hop); with many problems the
    // Horner chains run on the device and only one point per problem comes back
    J.dev_horner = !fb && np >= C.msm_dev_horner_min && P.W <= 64;
    static const bool host8_on = h8::available() && !(knob("ROFL_MSM_HOST8") && atoi(knob("ROFL_MSM_HOST8")) == 0);
    J.host8 = J.dev_horner && host8_on;
    ge *hres_dev = W.h_res.dev<ge>(PW * (size_t)P.c + np);
    u32 *h_flag = W.h_ovf.as<u32>(4), *d_flag = W.h_ovf.dev<u32>(4);
    const u32 Wb = (u32)(PW / nq);                           // bucket arrays per grid problem
    const u32 n_side = (u32)(lr ? n / 2 : n);
    const u32 nb_final = P.c - 1;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    *h_flag = 0;
    if (J.kind == MsmKind::Small) {
        ge *S_fin_s = J.dev_horner ? W.S[0].as<ge>(PW) : hres_dev;
        ge *C_fin_s = J.dev_horner ? W.Cacc[0].as<ge>(PW * (size_t)nb_final) : hres_dev + PW;
        size_t lds_lists = (size_t)P.B * 4 * (1 + small_cap);
        size_t lds_red = std::max(((size_t)(P.B / 8) * 4 + (size_t)(P.B / 16) * 5 + 1) * sizeof(ge), ((size_t)P.B + (size_t)P.B * 3 / 4 + 1) * sizeof(ge));      // fused / binary reduction trees
        uint64_t items = (uint64_t)np * n_side * P.W;
        KSpan ks(C.tm, st, ROFL_TK_MSM_SMALL, items * 7 + (uint64_t)PW * P.B * 10, (uint64_t)np * n_side * (32 + 96));
        static const bool small_tl = knob("ROFL_DBG_SMALL_TIMELINE") != nullptr;
        unsigned long long *tl_dev = nullptr;
        if (small_tl) { HIPCHK(hipMalloc(&tl_dev, PW * 32)); HIPCHK(hipMemsetAsync(tl_dev, 0, PW * 32, st)); }
        hipLaunchKernelGGL(k_msm_small, dim3((unsigned)PW), dim3(P.B < 64 ? 64 : P.B), std::max(lds_lists, lds_red), st, n_side, mw, mm, d_probs, buckets,
                           S_fin_s, C_fin_s, nb_final, d_flag, small_cap, tl_dev);
        if (small_tl) {      // mean phase durations over the blocks of this launch (100 MHz clock)
            std::vector<unsigned long long> hts(PW * 4);
            HIPCHK(hipMemcpyAsync(hts.data(), tl_dev, PW * 32, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st)); HIPCHK(hipFree(tl_dev));
            double ph[3] = {0, 0, 0}; unsigned long long t_lo = ~0ull, t_hi = 0;
            for (size_t b = 0; b < PW; b++) { for (int k = 0; k < 3; k++) ph[k] += (double)(hts[b * 4 + k + 1] - hts[b * 4 + k]); t_lo = std::min(t_lo, hts[b * 4]); t_hi = std::max(t_hi, hts[b * 4 + 3]); }
            fprintf(stderr, "[rofl] k_msm_small PW=%zu n_side=%u c=%u: rank %.1f us, bucket sums %.1f us, reduce %.1f us (block means); first start -> last end %.1f us\n",
                    PW, n_side, P.c, ph[0] / PW * 0.01, ph[1] / PW * 0.01, ph[2] / PW * 0.01, (double)(t_hi - t_lo) * 0.01);
        }
        if (J.host8) hipLaunchKernelGGL(k_msm_wsum, grid1(PW * 4), dim3(TPB), 0, st, (u32)PW, (const ge *)S_fin_s, (const ge *)C_fin_s, nb_final, hres_dev);
        else if (J.dev_horner) hipLaunchKernelGGL(k_msm_horner, dim3((unsigned)np), dim3(256), 0, st, mw, (const ge *)S_fin_s, (const ge *)C_fin_s, nb_final, hres_dev);
        return J;
    }
    // ---- SORT (+ ACCUMULATE: its list format depends on the sort)
    const uint64_t terms = (uint64_t)(lr ? nq : np) * n;
    if (J.two) {
        u32 *bins = W.sorted.as<u32>(PW * tl.nbins * tl.cap_bin);
        u32 *bcur = W.cur.as<u32>(PW * tl.nbins * 2);
        u32 *btail = W.tail.as<u32>(PW * tl.nbins * (size_t)MSM_BIN_TAIL);
        HIPCHK(hipMemsetAsync(bcur, 0, sizeof(u32) * PW * tl.nbins * 2, st));
        u32 iter_pts = (tl.stage >= 144 ? 16384u : 12288u) / mm.fb_wps; if (iter_pts < 1024) iter_pts = 1024;      // ~ 64 (48) new items per bin and iteration against a row of 144 (72)
        u32 tile = iter_pts;
        while (((size_t)((n_side + tile - 1) / tile) * PW > 512 || (n_side + tile - 1) / tile > 48) && tile < n_side) tile *= 2;      // <= 48 tiles per array: their left-overs (< 32 each) fit the bin tails with room for row spills
        dim3 grid((n_side + tile - 1) / tile, (u32)PW);
        { KSpan ks(C.tm, st, ROFL_TK_MSM_SCATTER, 0, terms * 32 + terms * P.W * 4);
          hipLaunchKernelGGL(k_msm_bin_l1, grid, dim3(1024), (size_t)(tl.nbins + tl.nbins * tl.stage) * 4, st, n_side, tile, iter_pts, mw, mm, d_probs, bcur, bins, btail, tl, d_flag);
          hipLaunchKernelGGL(k_msm_bin_l2, dim3(tl.nbins, (u32)PW), dim3(512), (size_t)(2 * 128 + tl.cap_bin) * 4, st, tl, P.B, bcur, bins, (const u32 *)btail, cnt, off, d_flag); }
        hipLaunchKernelGGL(k_msm_scan, dim3((unsigned)PW), dim3(TPB), 0, st, P.B, cnt, off, (u32 *)nullptr, perm);
        if (C.tm.enabled) { e0 = C.tm.get(); e1 = C.tm.get(); HIPCHK(hipEventRecord(e0, st)); }
        {
            KSpan ks_acc(C.tm, st, ROFL_TK_MSM_ACCUMULATE_FB, terms * P.W * 7, terms * 32);
            static const char *timeline = knob("ROFL_DBG_ACC_TIMELINE");      // debugging: per-wave start / end / placement of every launch, appended to this file
            if (timeline) {
                dim3 g = grid1((size_t)Wb * P.B, (u32)nq);
                size_t waves = (size_t)g.x * g.y * (TPB / 64);
                unsigned long long *rec; HIPCHK(hipMalloc(&rec, waves * 32)); HIPCHK(hipMemsetAsync(rec, 0, waves * 32, st));
                hipLaunchKernelGGL(k_msm_accumulate_fb_dbg, g, dim3(TPB), 0, st, (u32)n, P.c, Wb, (u32)(np / nq), d_probs, cnt, off, bins, perm, buckets, MSM_LIST_ABS, dbg_mask, acc_balance, rec);
                std::vector<unsigned long long> h(waves * 4);
                HIPCHK(hipMemcpyAsync(h.data(), rec, waves * 32, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st));
                if (FILE *f = fopen(timeline, "ab")) { unsigned long long hdr[4] = {0x54494d45ull, waves, g.x, g.y}; fwrite(hdr, 8, 4, f); fwrite(h.data(), 8, h.size(), f); fclose(f); }
                HIPCHK(hipFree(rec));
            } else
            hipLaunchKernelGGL(k_msm_accumulate_fb, grid1((size_t)Wb * P.B, (u32)nq), dim3(TPB), 0, st, (u32)n, P.c, Wb, (u32)(np / nq), d_probs, cnt, off, bins, perm, buckets, MSM_LIST_ABS, dbg_mask, acc_balance);
        }
        if (C.tm.enabled) HIPCHK(hipEventRecord(e1, st));
    } else if (J.kind != MsmKind::CountSort) {      // slot sort (fixed-base or generic)
        HIPCHK(hipMemsetAsync(cnt, 0, sizeof(u32) * (PW * P.B + 4), st));
        const u32 cap = J.cap;
        u32 *slots = W.sorted.as<u32>(PW * P.B * cap);
        MsmOvf *ovf = W.ovf.as<MsmOvf>(MSM_OVF_MAX);
        u32 *ovf_count = cnt + PW * P.B;
        if (C.msm_lds && n >= C.msm_lds_min && (size_t)P.B * 4 <= 128 * 1024) {
            u32 per_q = fb ? J.sets : P.W;
            u32 tile = n_side;
            // tile so that a block ranks ~128k items at most, and the launch has a few hundred blocks
            u32 wps = fb ? mm.fb_wps : 1;
            while (tile > 1024 && ((size_t)tile * wps > (size_t)C.msm_lds_tile || (size_t)((n_side + tile - 1) / tile) * nq * (lr ? 2 : 1) * per_q < 256)) tile /= 2;
            dim3 grid((n_side + tile - 1) / tile, (u32)(nq * (lr ? 2 : 1) * per_q));
            uint64_t items = terms * P.W;
            KSpan ks(C.tm, st, ROFL_TK_MSM_SCATTER, 0, terms * 32 + items * 4);
            hipLaunchKernelGGL(k_msm_scatter_lds, grid, dim3(1024), (size_t)P.B * 4, st, n_side, tile, mw, mm, d_probs, cnt, slots, cap, ovf_count, ovf, MSM_OVF_MAX, dbg_scatter);
        } else
            hipLaunchKernelGGL(k_msm_scatter_slots, grid1(n, (u32)(nq * P.W)), dim3(TPB), 0, st, (u32)n, mw, mm, d_probs, cnt, slots, cap, ovf_count, ovf, MSM_OVF_MAX);
        hipLaunchKernelGGL(k_msm_scan, dim3((unsigned)PW), dim3(TPB), 0, st, P.B, cnt, off, (u32 *)nullptr, perm);
        if (C.tm.enabled) { e0 = C.tm.get(); e1 = C.tm.get(); HIPCHK(hipEventRecord(e0, st)); }
        {
            uint64_t acc_adds = terms * P.W;
            KSpan ks_acc(C.tm, st, fb ? ROFL_TK_MSM_ACCUMULATE_FB : ROFL_TK_MSM_ACCUMULATE_GEN, acc_adds * 7, terms * 32);
            if (fb) hipLaunchKernelGGL(k_msm_accumulate_fb, grid1((size_t)Wb * P.B, (u32)nq), dim3(TPB), 0, st, (u32)n, P.c, Wb, (u32)(np / nq), d_probs, cnt, off, slots, perm, buckets, cap, dbg_mask, acc_balance);
            else hipLaunchKernelGGL(k_msm_accumulate_gen, grid1((size_t)Wb * P.B, (u32)nq), dim3(TPB), 0, st, (u32)n, P.c, Wb, (u32)(np / nq), d_probs, cnt, off, slots, perm, buckets, cap, dbg_mask, acc_balance);
        }
        if (C.tm.enabled) HIPCHK(hipEventRecord(e1, st));
        hipLaunchKernelGGL(k_msm_overflow, dim3(1), dim3(64), 0, st, Wb, P.B, (u32)(np / nq), d_probs, ovf_count, ovf, MSM_OVF_MAX, buckets, fb ? 1 : 0);
        HIPCHK(hipMemcpyAsync(h_flag, ovf_count, 4, hipMemcpyDeviceToHost, st));
    } else {      // count / scan / scatter: no fixed-size structure, always sufficient
        HIPCHK(hipMemsetAsync(cnt, 0, sizeof(u32) * (PW * P.B + 4), st));
        u32 *sorted = W.sorted.as<u32>(PW * n * 2);
        hipLaunchKernelGGL(k_msm_count, grid1(n, (u32)(nq * P.W)), dim3(TPB), 0, st, (u32)n, mw, mm, d_probs, cnt);
        hipLaunchKernelGGL(k_msm_scan, dim3((unsigned)PW), dim3(TPB), 0, st, P.B, cnt, off, cur, perm);
        hipLaunchKernelGGL(k_msm_scatter, grid1(n, (u32)(nq * P.W)), dim3(TPB), 0, st, (u32)n, mw, mm, d_probs, cur, sorted);
        if (C.tm.enabled) { e0 = C.tm.get(); e1 = C.tm.get(); HIPCHK(hipEventRecord(e0, st)); }
        hipLaunchKernelGGL(k_msm_accumulate_gen, grid1((size_t)Wb * P.B, (u32)nq), dim3(TPB), 0, st, (u32)n, P.c, Wb, (u32)(np / nq), d_probs, cnt, off, sorted, perm, buckets, 0u, dbg_mask, acc_balance);
        if (C.tm.enabled) HIPCHK(hipEventRecord(e1, st));
    }
    if (C.tm.enabled) {
        C.tm.acc_ev.push_back({e0, e1}); C.tm.t.msm_accumulate_launches++; C.tm.t.msm_additions += terms * P.W;
        char tg[96]; snprintf(tg, sizeof tg, "msm np=%zu n=%zu c=%u cap=%u fb=%u lr=%d", np, n, P.c, J.kind != MsmKind::CountSort ? J.cap : 0u, fb ? J.sets : 0u, (int)lr); C.tm.acc_tag.push_back(tg);
    }
    // ---- REDUCE: bit-sum tree, global 8-ary levels while more than 512 nodes remain, then one fused launch per bucket array
    const ge *S_in = buckets; const ge *C_in = nullptr;
    u32 E = P.B, nb = 0, lv = 0;
    uint64_t red_adds = 0;
    { u32 e = P.B, b = 0; while (e > 512) { red_adds += (uint64_t)(e / 8) * (11 + 7 * b); e /= 8; b += 3; }
      red_adds += (uint64_t)(e / 8) * (16 + 7 * b); e /= 8; b += 3; while (e > 1) { red_adds += (uint64_t)(e / 2) * (1 + b); e /= 2; b++; } }
    ge *S_fin = J.dev_horner ? W.S[1].as<ge>(PW) : hres_dev;
    ge *C_fin = J.dev_horner ? W.Cacc[1].as<ge>(PW * (size_t)nb_final) : hres_dev + PW;
    {
        KSpan ks_red(C.tm, st, ROFL_TK_MSM_REDUCE, red_adds * PW * 9, (uint64_t)PW * P.B * 128);
        if (C.msm_group_reduce && P.B >= 1024) {
            // every run of 512 buckets reduced by its own block, then one block per array combines the groups (two launches, the
            // first at full occupancy, instead of a chain of three whose last one ran on PW blocks)
            u32 G = P.B / 512, gbits = P.c - 1 - 9;
            ge *GS = W.S[0].as<ge>(PW * G);
            ge *GC = W.Cacc[0].as<ge>(PW * (size_t)G * 9);
            size_t lds_a = ((size_t)64 * 4 + (size_t)32 * 5 + 1) * sizeof(ge);
            hipLaunchKernelGGL(k_msm_reduce_fused, dim3((unsigned)(PW * G)), dim3(256), lds_a, st, 512u, 0u, (const ge *)buckets, (const ge *)nullptr, GS, GC, 9u);
            u32 half = G / 2 ? G / 2 : 1, nout = 10 + gbits;
            hipLaunchKernelGGL(k_msm_reduce_groups, dim3((unsigned)PW), dim3(half, nout), (size_t)nout * half * sizeof(ge), st, G, gbits, (const ge *)GS, (const ge *)GC, S_fin, C_fin, nb_final);
        } else {
            while (E > 512) {
                u32 E8 = E / 8;
                ge *S_out = W.S[lv & 1].as<ge>(PW * E8);
                ge *C_out = W.Cacc[lv & 1].as<ge>(PW * (size_t)(nb + 3) * E8);
                static const int red_split = knob("ROFL_RED_SPLIT") ? atoi(knob("ROFL_RED_SPLIT")) : 0;
                hipLaunchKernelGGL(k_msm_reduce_level, grid1((size_t)E8 * ((red_split ? 4 : 1) + nb), (u32)PW), dim3(TPB), 0, st, E, nb, S_in, C_in, S_out, C_out, red_split);
                S_in = S_out; C_in = C_out; E = E8; nb += 3; lv++;
            }
            if (J.dev_horner) { S_fin = W.S[lv & 1].as<ge>(PW); C_fin = W.Cacc[lv & 1].as<ge>(PW * (size_t)nb_final); }
            // block size = first-level work items (small bucket arrays, c = 7: 32 items -- a 256-thread block would idle 7 of its 8
            // waves and, at 163 VGPRs, hold a whole CU: thousands of such blocks (n_partition = 64) ran 18 deep per CU)
            static const u32 red_fused_max = knob("ROFL_RED_FUSED_T") ? (u32)atoi(knob("ROFL_RED_FUSED_T")) : 512u;
            u32 fused_items = (E / 8) * (4 + nb), fused_threads = fused_items > 256 ? 512 : fused_items > 128 ? 256 : fused_items > 64 ? 128 : 64;
            if (fused_threads > red_fused_max) fused_threads = red_fused_max;
            size_t lds = ((size_t)(E / 8) * (1 + nb + 3) + (size_t)(E / 16) * (1 + nb + 4) + 1) * sizeof(ge);
            hipLaunchKernelGGL(k_msm_reduce_fused, dim3((unsigned)PW), dim3(fused_threads), lds, st, E, nb, S_in, C_in, S_fin, C_fin, nb_final);
        }
    }
    if (J.host8)           // many problems, AVX-512 IFMA host: the device adds up each window's bit-sums, the chains across the windows go to the host
        hipLaunchKernelGGL(k_msm_wsum, grid1(PW * 4), dim3(TPB), 0, st, (u32)PW, (const ge *)S_fin, (const ge *)C_fin, nb_final, hres_dev);
    else if (J.dev_horner)      // many problems: their Horner chains run side by side on the device, one point per problem comes back
        hipLaunchKernelGGL(k_msm_horner, dim3((unsigned)np), dim3(256), 0, st, mw, (const ge *)S_fin, (const ge *)C_fin, nb_final, hres_dev);
    return J;
}

// after the stream has been synchronised: did the attempt overflow one of its fixed-size structures?  (then `al` has lost that variant)
bool msm_retry(const MsmJob &J, MsmAllow &al) {
    u32 flag = *J.ws->h_ovf.as<u32>(4);
    if (knob("ROFL_TRACE") && J.kind != MsmKind::CountSort) fprintf(stderr, "[rofl] msm np=%zu n=%zu c=%u cap=%u fb=%u lr=%d overflow=%u\n", J.np, J.n, J.P.c, J.cap, J.fb() ? J.sets : 0u, (int)J.lr, flag);
    if (J.kind == MsmKind::Small) { if (flag) { al.small = false; return true; } return false; }      // a bucket list overflowed: repeat through the general pipeline
    if (J.two) { if (flag) { al.two = false; return true; } return false; }                              // a coarse bin overflowed (skewed scalars): repeat on the slot path
    if (J.kind == MsmKind::CountSort) return false;
    if (flag > MSM_OVF_MAX) { if (J.fb()) al.fb = false; else al.slots = false; return true; }           // pathological input: the next (slower, always sufficient) variant
    return false;
}

// ---- window combination on the host (the device already did it for launches with many problems)
void msm_finish(Ctx &C, const MsmJob &J, std::vector<ge5> &results, const MsmOpt &opt) {
    const size_t np = J.np, PW = J.PW; const MsmPlan &P = J.P; const u32 sets = J.sets;
    u32 nb = P.c - 1;
    ge *h = J.ws->h_res.as<ge>(PW * (size_t)P.c + np);
    double t0 = now_ms();
    results.resize(np);
    std::vector<double> cpu_each(np, 0.0);
    if (J.host8) {
        // h holds one point per (problem, window): eight problems per task run their 253-step chains in the lanes of one AVX-512 stream
        u32 pos[64];
        for (u32 w = 0; w < P.W; w++) pos[w] = w + 1 == P.W ? 253 - P.c : (w < P.wide ? w * P.c : P.wide * P.c + (w - P.wide) * (P.c - 1));      // msm_window's layout
        C.pool->run((np + 7) / 8, [&](size_t b) {
            size_t p0 = b * 8; int lanes = (int)std::min<size_t>(8, np - p0);
            ge5 out[8];
            h8::horner8(out, lanes, (int)P.W, pos, [&](int l, int w) { return (const ge *)&h[(p0 + (size_t)l) * P.W + (size_t)w]; });
            for (int l = 0; l < lanes; l++) { results[p0 + l] = out[l]; if (opt.post) opt.post(p0 + l); }
        });
    } else if (J.dev_horner) {
        if (opt.post) C.pool->run(np, [&](size_t p) { results[p] = h51::from_ge_loose(h[p]); opt.post(p); });
        else for (size_t p = 0; p < np; p++) results[p] = h51::from_ge_loose(h[p]);
    } else if (J.fb()) {
        // sets of a problem carry equal weight: add them up, then one Horner over the c - 1 bit-sums
        C.pool->run(np, [&](size_t p) {
            double tc0 = now_ms();
            size_t base = p * sets;                         // lr: problem 2q+side owns sets [(2q+side)*sets, ...)
            ge5 acc = h51::identity(); bool started = false;
            for (int l = (int)nb - 1; l >= 0; l--) {
                if (started) acc = h51::gdouble(acc);
                for (u32 s = 0; s < sets; s++) { acc = h51::gadd(acc, h51::from_ge_loose(h[PW + (base + s) * nb + l])); started = true; }
                if (l == 0) for (u32 s = 0; s < sets; s++) acc = h51::gadd(acc, h51::from_ge_loose(h[base + s]));
            }
            results[p] = acc; if (opt.post) opt.post(p); cpu_each[p] = now_ms() - tc0;
        });
    } else {
        // One 253-step chain per problem: sum_w 2^(pos_w) (S_w + sum_l 2^l D_(w,l)).  With few problems (the IPP rounds of a client with four
        // chunks: eight) the chain is split over TWO pool threads: the upper windows (about three eighths of them: that part also carries the
        // doublings down to bit 0) and the lower ones; the second to finish adds the halves.  46 -> ~30 us on the hop.
        auto horner_range = [&](size_t p, int w_hi, int w_lo, bool down_to_zero) {      // windows [w_lo, w_hi], result scaled by 2^(pos of w_lo) unless down_to_zero
            ge5 acc = h51::identity(); bool started = false;
            for (int w = w_hi; w >= w_lo; w--) {
                size_t pw = p * P.W + w;
                int width = (u32)w + 1 == P.W ? (int)P.c + 1 : ((u32)w < P.wide ? (int)P.c : (int)P.c - 1);
                for (int l = width - 1; l >= 0; l--) {
                    if (started) acc = h51::gdouble(acc);
                    if (l <= (int)P.c - 2) { acc = h51::gadd(acc, h51::from_ge_loose(h[PW + pw * nb + l])); started = true; }
                    if (l == 0) { acc = h51::gadd(acc, h51::from_ge_loose(h[pw])); started = true; }
                }
            }
            if (down_to_zero && w_lo > 0 && started) {
                u32 pos = (u32)w_lo < P.wide ? (u32)w_lo * P.c : P.wide * P.c + ((u32)w_lo - P.wide) * (P.c - 1);      // msm_window's layout
                for (u32 i = 0; i < pos; i++) acc = h51::gdouble(acc);
            }
            return acc;
        };
        const bool split = np * 2 <= 32 && P.W >= 8;
        if (!split) {
            C.pool->run(np, [&](size_t p) {
                double tc0 = now_ms();
                results[p] = horner_range(p, (int)P.W - 1, 0, false); if (opt.post) opt.post(p); cpu_each[p] = now_ms() - tc0;
            });
        } else {
            const int w_split = (int)P.W - (int)(P.W * 3 / 8);      // windows [w_split, W) on one thread, [0, w_split) on another
            std::vector<ge5> part(2 * np);
            std::unique_ptr<std::atomic<int>[]> half_done(new std::atomic<int>[np]);
            for (size_t p = 0; p < np; p++) half_done[p].store(0);
            C.pool->run(2 * np, [&](size_t t) {
                double tc0 = now_ms();
                size_t p = t >> 1; bool upper = (t & 1) != 0;
                part[t] = upper ? horner_range(p, (int)P.W - 1, w_split, true) : horner_range(p, w_split - 1, 0, false);
                if (half_done[p].fetch_add(1) == 1) { results[p] = h51::gadd(part[2 * p], part[2 * p + 1]); if (opt.post) opt.post(p); }
                double dt = now_ms() - tc0; if (dt > cpu_each[p]) cpu_each[p] = dt;
            });
        }
    }
    C.tm.t.host_ms += now_ms() - t0;
    C.hs.horner_wall += now_ms() - t0; { double mx = 0; for (double v : cpu_each) mx = std::max(mx, v); C.hs.horner_cpu += mx; } C.hs.n++;
}

// one MSM, start to finish: enqueue, wait, repeat through the next variant if a fixed-size structure overflowed, combine.
void msm_run(Ctx &C, const std::vector<MsmProb> &probs, size_t n, std::vector<ge5> &results, const MsmOpt &opt = MsmOpt()) {
    MsmAllow al; bool overlap_done = false;
    for (;;) {
        double t_enter = now_ms();
        MsmJob J = msm_enqueue(C, C.mws[0], probs, n, opt, al);
        if (opt.overlap && !overlap_done) { opt.overlap(); overlap_done = true; }
        double t_sync0 = now_ms(); C.hs.enqueue += t_sync0 - t_enter;
        C.sync();
        C.hs.sync += now_ms() - t_sync0;
        if (msm_retry(J, al)) continue;
        msm_finish(C, J, results, opt);
        return;
    }
}
// two independent MSMs behind ONE synchronisation (the verifier's generator MSM and its proof-point MSM): their launches queue back
// to back on the lane's stream, each with its own workspace; an overflow in either repeats that one on its own.
void msm_run2(Ctx &C, const std::vector<MsmProb> &pa, size_t na, const MsmOpt &oa, std::vector<ge5> &ra,
              const std::vector<MsmProb> &pb, size_t nb, const MsmOpt &ob, std::vector<ge5> &rb) {
    MsmAllow ala, alb;
    double t_enter = now_ms();
    // the second one (small, latency-bound launches) runs on the side stream beside the first one's kernels and joins before the wait
    if (!C.stream2) HIPCHK(hipStreamCreateWithFlags(&C.stream2, hipStreamNonBlocking));
    if (!C.ev_m2) { HIPCHK(hipEventCreateWithFlags(&C.ev_m2, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&C.ev_m2j, hipEventDisableTiming)); }
    struct Join { hipStream_t s; ~Join() { (void)hipStreamSynchronize(s); } } join{C.stream2};      // nothing of the side stream outlives the call, error paths included
    HIPCHK(hipEventRecord(C.ev_m2, C.stream));                    // inputs of both are ready
    HIPCHK(hipStreamWaitEvent(C.stream2, C.ev_m2, 0));
    MsmJob Jb = msm_enqueue(C, C.mws[1], pb, nb, ob, alb, C.stream2);
    HIPCHK(hipEventRecord(C.ev_m2j, C.stream2));
    MsmJob Ja = msm_enqueue(C, C.mws[0], pa, na, oa, ala);
    HIPCHK(hipStreamWaitEvent(C.stream, C.ev_m2j, 0));
    double t_sync0 = now_ms(); C.hs.enqueue += t_sync0 - t_enter;
    C.sync();
    C.hs.sync += now_ms() - t_sync0;
    bool again_a = msm_retry(Ja, ala), again_b = msm_retry(Jb, alb);
    if (!again_a) msm_finish(C, Ja, ra, oa);
    if (!again_b) msm_finish(C, Jb, rb, ob);
    if (again_a) { for (;;) { MsmJob J = msm_enqueue(C, C.mws[0], pa, na, oa, ala); C.sync(); if (msm_retry(J, ala)) continue; msm_finish(C, J, ra, oa); break; } }
    if (again_b) { for (;;) { MsmJob J = msm_enqueue(C, C.mws[1], pb, nb, ob, alb); C.sync(); if (msm_retry(J, alb)) continue; msm_finish(C, J, rb, ob); break; } }
}

// ---------------------------------------------------------------- transcript helpers
void tr_append_point(Merlin &t, const char *label, const ge5 &p, uint8_t *enc_out) {
    uint8_t e[32]; h51::encode(e, p); t.append(label, e, 32); if (enc_out) memcpy(enc_out, e, 32);
}
void fill_pow2(sc *tab, sc base_mont, int count) { tab[0] = base_mont; for (int i = 1; i < count; i++) tab[i] = sc_montmul(tab[i - 1], tab[i - 1]); }

sc sum_partials(const sc *p, size_t count, size_t stride, size_t which) {
    sc acc = sc_zero();
    for (size_t i = 0; i < count; i++) acc = sc_add(acc, p[i * stride + which]);
    return acc;
}

// ================================================================ prover (bulletproofs RangeProof::prove_multiple)
// P chunks of m values each; vshift [P][m] (device), blind_canon [P][m] (device).
// Outputs: proofs (host, P*plen), V bytes (host, P*m*32).
// nonces[c]: where chunk c draws its nonces (a device-resident stream or a seed, and the index of its first nonce);
// proofs_out[c]: where chunk c's proof goes (host).  The chunks may belong to different clients (batched create).
struct ChunkNonce { int mode; NonceSeed seed; const uint8_t *d_stream; u64 stream_scalars, base; };
void prove_chunks(Ctx &C, const char *label, size_t P, size_t n, size_t m, const u64 *d_vshift, const sc *d_blind,
                  const std::vector<ChunkNonce> &nonces, const uint8_t *h_V /* [P][m][32] host */, uint8_t *const *proofs_out,
                  hipEvent_t v_ready = nullptr /* recorded after the copy that fills h_V; nullptr: already complete */) {
    size_t N = n * m; unsigned lgN = lg2u(N);
    static const bool ptrace = knob("ROFL_TRACE") && atoi(knob("ROFL_TRACE")) >= 2;
    double pt0 = now_ms(), ptl = pt0;
    auto mark = [&](const char *what, long a = -1) {
        if (!ptrace) return;
        double t = now_ms(); fprintf(stderr, "[rofl-trace lane=%p] %-14s %6ld  +%.3f ms  (t=%.3f)\n", (void *)&C, what, a, t - ptl, t - pt0); ptl = t;
    };
    size_t plen = 32 * (9 + 2 * (size_t)lgN);
    GensPin gens = get_gens(C, n, m);            // pinned until the proofs are done
    niels *tbl = gens.tbl();
    const niels *wtab = gens.wtab();
    ChunkParams *h_cp = C.h_cp.as<ChunkParams>(P);
    ChunkParams *d_cp = C.cp.as<ChunkParams>(P);
    memset(h_cp, 0, sizeof(ChunkParams) * P);
    u64 per = (u64)m * (2 * n + 4);
    for (size_t c = 0; c < P; c++) {
        h_cp[c].nonce_base = nonces[c].base; h_cp[c].nonce_mode = nonces[c].mode; h_cp[c].nonce_seed = nonces[c].seed;
        h_cp[c].nonce_stream = nonces[c].d_stream; h_cp[c].nonce_stream_scalars = nonces[c].stream_scalars;
    }
    HIPCHK(hipMemcpyAsync(d_cp, h_cp, sizeof(ChunkParams) * P, hipMemcpyHostToDevice, C.stream));
    sc *sL = C.sL.as<sc>(P * N), *sR = C.sR.as<sc>(P * N), *party = C.party.as<sc>(P * 4 * m), *Scanon = C.Scanon.as<sc>(P * 2 * N);
    hipLaunchKernelGGL(k_nonce_expand, grid1(per, (u32)P), dim3(TPB), 0, C.stream, (u32)n, (u32)m, d_cp, sL, sR, party, Scanon);
    // A partials: they depend on the values only and the host reads them after the S MSM -- side stream, beside the nonce expansion
    if (!C.stream2) HIPCHK(hipStreamCreateWithFlags(&C.stream2, hipStreamNonBlocking));
    if (!C.ev_a) { HIPCHK(hipEventCreateWithFlags(&C.ev_a, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&C.ev_a0, hipEventDisableTiming)); }
    HIPCHK(hipEventRecord(C.ev_a0, C.stream));                    // d_vshift is ready (and the previous call's reads of `partial` are done)
    HIPCHK(hipStreamWaitEvent(C.stream2, C.ev_a0, 0));
    ge *partial = C.partial.as<ge>(P * m);
    hipLaunchKernelGGL(k_bitcommit, grid1(m, (u32)P), dim3(TPB), 0, C.stream2, (u32)n, (u32)m, d_vshift, tbl, partial);
    u32 nblkA = (u32)std::min<size_t>(16, (m + TPB - 1) / TPB);
    ge *partial2 = C.partial2.as<ge>(P * nblkA);
    hipLaunchKernelGGL(k_point_sum, dim3(nblkA, (u32)P), dim3(TPB), TPB * sizeof(ge), C.stream2, partial, (u32)m, partial2);
    ge *h_A = C.h_part.as<ge>(P * nblkA);
    HIPCHK(hipMemcpyAsync(h_A, partial2, sizeof(ge) * P * nblkA, hipMemcpyDeviceToHost, C.stream2));
    HIPCHK(hipEventRecord(C.ev_a, C.stream2));
    u32 nblkS = (u32)std::min<size_t>(16, (m + TPB - 1) / TPB);
    sc *scpart = C.scpart.as<sc>(P * 64 * 3);
    PowTabs *d_pt = C.powtabs.as<PowTabs>(P);
    hipLaunchKernelGGL(k_party_sums, dim3(nblkS, (u32)P), dim3(TPB), 0, C.stream, (u32)m, 0, d_cp, (const PowTabs *)d_pt, party, d_blind, scpart);
    sc *h_sc = C.h_misc2.as<sc>(P * 64 * 3);
    HIPCHK(hipMemcpyAsync(h_sc, scpart, sizeof(sc) * P * nblkS * 3, hipMemcpyDeviceToHost, C.stream));
    // S = <sL,G> + <sR,H> + s_bl * Bb
    std::vector<MsmProb> probs(P); std::vector<ge5> res;
    for (size_t c = 0; c < P; c++) probs[c] = MsmProb{tbl, Scanon + c * 2 * N};
    C.tm.t.msm_terms += P * 2 * N;
    mark("setup");
    std::vector<Merlin> tr; tr.reserve(P);
    std::vector<sc> a_bl(P), s_bl(P), y(P), z(P), zz(P), x(P), w(P);
    for (size_t c = 0; c < P; c++) tr.emplace_back(label, strlen(label));
    {
        MsmOpt mo; if (wtab) { gens.fb_for(P, &mo.fb_wtab, &mo.fb_c); mo.fb_stride = 2 * N; }
        mo.overlap = [&]() {      // the transcript prefix (m commitments per chunk) does not depend on S: hash it while the MSM runs
            double t0 = now_ms();
            if (v_ready) HIPCHK(hipEventSynchronize(v_ready));      // first in the stream: long done by the time the S launches are enqueued
            C.pool->run(P, [&](size_t c) {
                Merlin &t = tr[c];
                t.append("dom-sep", (const uint8_t *)"rangeproof v1", 13);
                t.append_u64("n", n); t.append_u64("m", m);
                t.append32_run('V', h_V + c * m * 32, m);
            });
            C.tm.t.host_ms += now_ms() - t0;
        };
        msm_run(C, probs, 2 * N, res, mo);
    }
    mark("msm S");
    HIPCHK(hipEventSynchronize(C.ev_a));      // the A partials (side stream: done long before the S MSM)

    double th = now_ms();
    C.pool->run(P, [&](size_t c) {
        uint8_t *o = proofs_out[c];
        Merlin &t = tr[c];
        a_bl[c] = h_canon(sum_partials(h_sc + c * nblkS * 3, nblkS, 3, 0));
        s_bl[c] = h_canon(sum_partials(h_sc + c * nblkS * 3, nblkS, 3, 1));
        ge5 A = h_fixed_mul(C.ht.Bb5, a_bl[c]);
        for (u32 k = 0; k < nblkA; k++) A = h51::gadd(A, h51::from_ge(h_A[c * nblkA + k]));
        ge5 S = h51::gadd(res[c], h_fixed_mul(C.ht.Bb5, s_bl[c]));
        tr_append_point(t, "A", A, o); tr_append_point(t, "S", S, o + 32);
        y[c] = t.challenge_scalar("y"); z[c] = t.challenge_scalar("z");
        zz[c] = h_mul(z[c], z[c]);
        ChunkParams &cp = h_cp[c];
        cp.y = h_mont(y[c]); cp.z = h_mont(z[c]); cp.zz = h_mont(zz[c]);
        cp.yinv = h_mont(h_inv(y[c]));
        fill_pow2(cp.ypow2, cp.y, MAX_LG); fill_pow2(cp.yinvpow2, cp.yinv, MAX_LG); fill_pow2(cp.zpow2, cp.z, MAX_LG);
    });
    C.tm.t.host_ms += now_ms() - th;
    HIPCHK(hipMemcpyAsync(d_cp, h_cp, sizeof(ChunkParams) * P, hipMemcpyHostToDevice, C.stream));
    u32 nblkT = (u32)std::min<size_t>(64, (N + TPB - 1) / TPB);
    sc *tpart = C.tmp_out.as<sc>(P * 64 * 3);
    hipLaunchKernelGGL(k_pow_tables, dim3((4 * PT_L * PT_E + TPB - 1) / TPB, (u32)P), dim3(TPB), 0, C.stream, d_cp, d_pt, lgN, 0);
    hipLaunchKernelGGL(k_poly_t, dim3(nblkT, (u32)P), dim3(TPB), 0, C.stream, (u32)n, (u32)m, d_cp, (const PowTabs *)d_pt, d_vshift, sL, sR, C.d_two_pow, tpart);
    hipLaunchKernelGGL(k_party_sums, dim3(nblkS, (u32)P), dim3(TPB), 0, C.stream, (u32)m, 1, d_cp, (const PowTabs *)d_pt, party, d_blind, scpart);
    sc *h_t = C.h_part.as<sc>(P * 64 * 3);
    HIPCHK(hipMemcpyAsync(h_t, tpart, sizeof(sc) * P * nblkT * 3, hipMemcpyDeviceToHost, C.stream));
    HIPCHK(hipMemcpyAsync(h_sc, scpart, sizeof(sc) * P * nblkS * 3, hipMemcpyDeviceToHost, C.stream));
    C.sync();
    th = now_ms();
    C.pool->run(P, [&](size_t c) {
        uint8_t *o = proofs_out[c];
        Merlin &t = tr[c];
        sc t0 = h_canon(sum_partials(h_t + c * nblkT * 3, nblkT, 3, 0));
        sc t1 = h_canon(sum_partials(h_t + c * nblkT * 3, nblkT, 3, 1));
        sc t2 = h_canon(sum_partials(h_t + c * nblkT * 3, nblkT, 3, 2));
        sc t1_bl = h_canon(sum_partials(h_sc + c * nblkS * 3, nblkS, 3, 0));
        sc t2_bl = h_canon(sum_partials(h_sc + c * nblkS * 3, nblkS, 3, 1));
        sc zvbl = h_canon(sum_partials(h_sc + c * nblkS * 3, nblkS, 3, 2));
        ge5 T1 = h51::gadd(h_fixed_mul(C.ht.B5, t1), h_fixed_mul(C.ht.Bb5, t1_bl));
        ge5 T2 = h51::gadd(h_fixed_mul(C.ht.B5, t2), h_fixed_mul(C.ht.Bb5, t2_bl));
        tr_append_point(t, "T_1", T1, o + 64); tr_append_point(t, "T_2", T2, o + 96);
        x[c] = t.challenge_scalar("x");
        sc xx = h_mul(x[c], x[c]);
        sc t_x = sc_add(sc_add(t0, h_mul(t1, x[c])), h_mul(t2, xx));
        sc t_x_bl = sc_add(sc_add(zvbl, h_mul(t1_bl, x[c])), h_mul(t2_bl, xx));
        sc e_bl = sc_add(a_bl[c], h_mul(s_bl[c], x[c]));
        t.append_scalar("t_x", t_x); t.append_scalar("t_x_blinding", t_x_bl); t.append_scalar("e_blinding", e_bl);
        sc_tobytes(o + 128, t_x); sc_tobytes(o + 160, t_x_bl); sc_tobytes(o + 192, e_bl);
        w[c] = t.challenge_scalar("w");
        h_cp[c].x = h_mont(x[c]);
        h_cp[c].gscale = sc_one_mont(); h_cp[c].hscale = sc_one_mont();
        // InnerProductProof::create
        t.append("dom-sep", (const uint8_t *)"ipp v1", 6);
        t.append_u64("n", N);
    });
    C.tm.t.host_ms += now_ms() - th;
    HIPCHK(hipMemcpyAsync(d_cp, h_cp, sizeof(ChunkParams) * P, hipMemcpyHostToDevice, C.stream));
    sc *a = C.a.as<sc>(P * N), *b = C.b.as<sc>(P * N), *yinvpow = C.yinv.as<sc>(P * N);
    hipLaunchKernelGGL(k_lr_vec, grid1(N, (u32)P), dim3(TPB), 0, C.stream, (u32)n, (u32)m, d_cp, (const PowTabs *)d_pt, d_vshift, sL, sR, C.d_two_pow, a, b, yinvpow);
    mark("poly/T/x");

    // ---- IPP rounds with lazily folded generators
    // Invariant: true G[j] = gscale * Gc[j], true H[j] = hscale * y^-j * Hc[j] for the materialised arrays Gc, Hc.
    size_t n_g = N; unsigned r = 0;
    std::vector<const niels *> cur(P, tbl);
    std::vector<std::vector<sc>> pu(P), pui(P);    // pending challenges (Montgomery)
    std::vector<sc> gscale(P, sc_one_mont()), hscale(P, sc_one_mont());
    int gsel = 0; bool first_level = true;
    auto stab = [&](size_t c, u32 h, sc &g, sc &hh) {
        g = sc_one_mont(); hh = sc_one_mont();
        for (unsigned q = 0; q < r; q++) {
            bool bit = (h >> (r - 1 - q)) & 1;
            g = sc_montmul(g, bit ? pu[c][q] : pui[c][q]);
            hh = sc_montmul(hh, bit ? pui[c][q] : pu[c][q]);
        }
    };
    sc *h_round = C.h_round.as<sc>(2 * P);
    sc *a2 = C.a2.as<sc>(P * N), *b2 = C.b2.as<sc>(P * N);      // ping-pong partners of a, b (k_ipp_round folds out of place)
    static const bool ipp_fused = !(knob("ROFL_IPP_FUSED") && atoi(knob("ROFL_IPP_FUSED")) == 0);
    static const bool fold_regs = !(knob("ROFL_FOLD_REGS") && atoi(knob("ROFL_FOLD_REGS")) == 0);
    bool just_materialised = false, ab_on_host = false;
    sc *ptab[2] = {C.ptab[0].as<sc>(P * 2 * N), C.ptab[1].as<sc>(P * 2 * N)}; int psel = 0;      // pending-challenge product tables (ping-pong)
    std::unique_ptr<std::atomic<int>[]> lr_done(new std::atomic<int>[P]);
    for (unsigned round = 0; round < lgN; round++) {
        size_t n_k = n_g >> r, nh = n_k / 2;
        sc *SL = C.SL.as<sc>(P * 2 * n_g), *SR = C.SR.as<sc>(P * 2 * n_g);
        bool merged = C.msm_lr != 0;
        bool fused = merged && ipp_fused && round > 0;           // one launch: fold by the previous challenge + this round's scalars + inner products
        u32 nblkI;
        sc *h_ip = C.h_ip.as<sc>(P * 256 * 2);                     // the partial sums land in mapped host memory
        if (fused) {
            nblkI = (u32)std::min<size_t>(256, std::max<size_t>(1, (n_g + TPB - 1) / TPB));      // one slot per thread while the 256 partial-sum rows last (the tail rounds are one 13-multiplication chain deep)
            int use_new = just_materialised ? 0 : 1;
            hipLaunchKernelGGL(k_ipp_round, dim3(nblkI, (u32)P), dim3(TPB), 0, C.stream, (u32)n_g, (u32)n_k, use_new ? r - 1 : 0u, use_new, d_cp,
                               (const sc *)C.h_round.dev<sc>(2 * P), (const sc *)a, (const sc *)b, a2, b2, N, yinvpow, N, SL, C.h_ip.dev<sc>(P * 256 * 2),
                               (const sc *)ptab[psel], ptab[psel ^ 1], N, n_k == 2 ? C.h_abfin.dev<sc>(4 * P) : (sc *)nullptr);
            if (n_k == 2) ab_on_host = true;
            std::swap(a, a2); std::swap(b, b2); psel ^= 1;
        } else {
            hipLaunchKernelGGL(k_ipp_scalars, grid1(n_g, (u32)P), dim3(TPB), 0, C.stream, (u32)n_g, (u32)n_k, r, d_cp, a, b, N, yinvpow, N, SL, SR, merged ? 1 : 0);
            nblkI = (u32)std::min<size_t>(32, (nh + TPB - 1) / TPB);
            hipLaunchKernelGGL(k_ipp_inner, dim3(nblkI, (u32)P), dim3(TPB), 0, C.stream, (u32)nh, a, b, N, C.h_ip.dev<sc>(P * 256 * 2));
        }
        just_materialised = false;
        std::vector<MsmProb> pr(2 * P);
        for (size_t c = 0; c < P; c++) { pr[2 * c] = MsmProb{cur[c], SL + c * 2 * n_g}; pr[2 * c + 1] = MsmProb{cur[c], (merged ? SL : SR) + c * 2 * n_g}; }
        C.tm.t.msm_terms += P * 2 * n_g;
        MsmOpt mo;
        if (merged) { mo.lr_nh = (u32)nh; mo.lr_ng = (u32)n_g; }
        if (first_level && wtab) { gens.fb_for(2 * P, &mo.fb_wtab, &mo.fb_c); mo.fb_stride = 2 * N; }
        // The host tail of the round runs inside the MSM's own pool tasks: the thread that finishes problem 2c (+1) adds c_L w B (c_R w B)
        // and encodes L (R); the second of a chunk's two to get there hashes both into the transcript, draws u and inverts it.  L and R
        // of a chunk are encoded side by side and the hop has one pool hand-off instead of two.
        for (size_t c = 0; c < P; c++) lr_done[c].store(0);
        mo.post = [&](size_t p) {
            size_t c = p >> 1; int side = (int)(p & 1);
            uint8_t *o = proofs_out[c] + 7 * 32 + 64 * round;
            sc cx = h_canon(sum_partials(h_ip + c * nblkI * 2, nblkI, 2, (size_t)side));
            h51::encode(o + 32 * side, h51::gadd(res[p], h_fixed_mul(C.ht.B5, h_mul(cx, w[c]))));
            if (lr_done[c].fetch_add(1) != 1) return;          // the chunk's other point is still on its way
            tr[c].append("L", o, 32); tr[c].append("R", o + 32, 32);
            sc u = tr[c].challenge_scalar("u");
            sc um = h_mont(u), uim = h51::sc_invert_mont_fast(um);
            h_round[2 * c] = um; h_round[2 * c + 1] = uim;          // mapped: k_ipp_fold_ab reads it and records it in the chunk's pending list
            h_cp[c].pend_u[pu[c].size()] = um; h_cp[c].pend_ui[pu[c].size()] = uim;
            pu[c].push_back(um); pui[c].push_back(uim);
        };
        msm_run(C, pr, 2 * n_g, res, mo);
        mark("round msm", (long)(2 * n_g));
        bool last = (round + 1 == lgN);
        // the fold of a, b by this challenge happens inside the next round's k_ipp_round; only the old three-kernel path and the
        // last round (whose result is the proof's final a, b) fold here
        if ((last && !ab_on_host) || !(merged && ipp_fused))
            hipLaunchKernelGGL(k_ipp_fold_ab, grid1(nh, (u32)P), dim3(TPB), 0, C.stream, (u32)nh, d_cp, (const sc *)C.h_round.dev<sc>(2 * P), r, a, b, N);
        r++;
        unsigned t_now = first_level ? (unsigned)C.fold_t1 : (unsigned)C.fold_t;
        // fold_min is a per-chunk size chosen for P = 4 (below it the fold kernel is latency-bound); what matters is the number of
        // outputs in the launch, so many small chunks (n_partition = 64) keep folding down to 64 generators each
        size_t n_after = n_g >> r;
        bool fold_pays = n_after >= C.fold_min || (n_after >= 64 && 2 * P * n_after >= 8 * C.fold_min);
        if (!last && r >= t_now && fold_pays) {
            // materialise: new[i] = sum_h s_h * cur[h*n_new + i]; with fold_unit the common factor s_0 moves into
            // gscale / hscale so that source 0 needs a single addition
            size_t n_new = n_g >> r; u32 nsrc = 1u << r;
            bool use_tab = first_level && C.fold_tab;
            int unit = C.fold_unit;
            FoldTabCfg fc = gens.fc();
            size_t dstride = use_tab ? (size_t)fc.np * FOLD_TAB_DIGITS : 256;
            th = now_ms();
            int8_t *h_dig = C.h_fdig.as<int8_t>(2 * P * nsrc * dstride);      // the fold's own pinned staging: nothing else writes them while its copies are queued
            memset(h_dig, 0, 2 * P * nsrc * dstride);
            FoldProb *h_fp = C.h_fprob.as<FoldProb>(2 * P + 2 * P);
            FoldTabProb *h_ftp = reinterpret_cast<FoldTabProb *>(h_fp + 2 * P);
            niels *gnew = C.gbuf[gsel].as<niels>(P * 2 * n_new);
            // (per chunk and independent: on the pool -- at n_partition = 64 this loop was 1.0-1.4 ms of one thread with the GPU idle, three times per proof)
            std::vector<int> topc(P, 0);
            C.pool->run(P, [&](size_t c) {
                int top = 0;
                sc yn = sc_one_mont();                       // y^-(h*n_new), stepping by y^-n_new
                sc ystep = sc_one_mont();
                { size_t e = n_new; int bidx = 0; while (e) { if (e & 1) ystep = sc_montmul(ystep, h_cp[c].yinvpow2[bidx]); e >>= 1; bidx++; } }
                // s_G(0) = prod uinv and s_G(all ones) = prod u = 1 / s_G(0); for H the roles of u and uinv swap
                sc g0, h0; stab(c, 0, g0, h0);
                sc gall, hall; stab(c, nsrc - 1, gall, hall);
                for (u32 h = 0; h < nsrc; h++) {
                    sc g, hh; stab(c, h, g, hh);
                    hh = sc_montmul(hh, yn);
                    yn = sc_montmul(yn, ystep);
                    if (unit) {
                        if (h == 0) continue;                 // scalar 1: handled by one addition in the kernel
                        g = sc_montmul(g, gall); hh = sc_montmul(hh, hall);
                    }
                    sc gc = h_canon(g), hc = h_canon(hh);
                    if (use_tab) {
                        for (u32 pc = 0; pc < fc.np; pc++) {
                            auto piece = [&](const sc &s) {
                                u32 bit0 = pc * fc.pb; u64 lo = (u64)s.v[bit0 / 32] | ((bit0 / 32 + 1 < 8) ? (u64)s.v[bit0 / 32 + 1] << 32 : 0);
                                lo >>= (bit0 % 32);
                                return fc.pb == 64 ? lo : (lo & (((u64)1 << fc.pb) - 1));
                            };
                            int t1 = wnaf_u64(h_dig + (((2 * c) * nsrc + h) * fc.np + pc) * FOLD_TAB_DIGITS, piece(gc), fc.w);
                            int t2 = wnaf_u64(h_dig + (((2 * c + 1) * nsrc + h) * fc.np + pc) * FOLD_TAB_DIGITS, piece(hc), fc.w);
                            top = std::max(top, std::max(t1, t2));
                        }
                    } else {
                        int t1 = sc_naf(h_dig + ((2 * c) * nsrc + h) * 256, gc);
                        int t2 = sc_naf(h_dig + ((2 * c + 1) * nsrc + h) * 256, hc);
                        top = std::max(top, std::max(t1, t2));
                    }
                }
                if (unit) { gscale[c] = sc_montmul(gscale[c], g0); hscale[c] = sc_montmul(hscale[c], h0); h_cp[c].gscale = gscale[c]; h_cp[c].hscale = hscale[c]; }
                h_fp[2 * c] = FoldProb{cur[c], gnew + c * 2 * n_new};
                h_fp[2 * c + 1] = FoldProb{cur[c] + n_g, gnew + c * 2 * n_new + n_new};
                h_ftp[2 * c] = FoldTabProb{0u, gnew + c * 2 * n_new};
                h_ftp[2 * c + 1] = FoldTabProb{(u32)n_g, gnew + c * 2 * n_new + n_new};
                topc[c] = top;
            });
            int top = 0; for (int t : topc) top = std::max(top, t);
            C.tm.t.host_ms += now_ms() - th;
            int8_t *d_dig = C.naf.as<int8_t>(2 * P * nsrc * dstride);
            HIPCHK(hipMemcpyAsync(d_dig, h_dig, 2 * P * nsrc * dstride, hipMemcpyHostToDevice, C.stream));
            void *d_fpv = C.foldprobs.ensure(2 * P * 16);
            if (use_tab) HIPCHK(hipMemcpyAsync(d_fpv, h_ftp, sizeof(FoldTabProb) * 2 * P, hipMemcpyHostToDevice, C.stream));
            else HIPCHK(hipMemcpyAsync(d_fpv, h_fp, sizeof(FoldProb) * 2 * P, hipMemcpyHostToDevice, C.stream));
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (C.tm.enabled) { e0 = C.tm.get(); e1 = C.tm.get(); HIPCHK(hipEventRecord(e0, C.stream)); }
            {
                // segment the digit positions so that K threads share one output with equal work
                u32 K = 1;
                size_t thr = (size_t)2 * P * n_new;
                // (segments trade extra doublings for parallelism: not worth it while other calls keep the GPU busy)
                while (K < FOLD_MAXSEG && thr * K < (size_t)C.fold_threads / (C.crowded() ? 2 : 1)) K *= 2;
                if (C.fold_k > 0) K = (u32)C.fold_k;
                FoldSeg seg{};
                double eff = (double)(nsrc - (unit ? 1 : 0));
                double cst = 1.0 + (use_tab ? eff * fc.np / (fc.w + 1.0) : eff / 3.0), lo_t = 0, hi_t = (top + 1) * cst + top + 1;
                int bounds[FOLD_MAXSEG + 1];
                for (int it = 0; it < 60; it++) {
                    double T = 0.5 * (lo_t + hi_t), pos = 0;
                    for (u32 k = 0; k < K; k++) { double len = (T - pos) / cst; if (len < 0) len = 0; pos += len; }
                    if (pos >= top + 1) hi_t = T; else lo_t = T;
                }
                { double pos = 0; bounds[0] = 0; for (u32 k = 0; k < K; k++) { double len = (hi_t - pos) / cst; if (len < 0) len = 0; pos += len; bounds[k + 1] = (int)(pos + 0.5); } }
                bounds[K] = top + 1;
                for (u32 k = 1; k <= K; k++) if (bounds[k] < bounds[k - 1]) bounds[k] = bounds[k - 1];
                for (u32 k = 0; k <= FOLD_MAXSEG; k++) seg.lo[k] = bounds[k <= K ? k : K];
                dim3 grid((unsigned)((n_new + 63) / 64), (u32)(2 * P)), block(64, K);
                uint64_t nz = 0;
                if (C.tm.enabled) { size_t tot_d = 2 * P * nsrc * dstride; for (size_t q = 0; q < tot_d; q++) nz += h_dig[q] != 0; }
                // algorithmic work per output: the non-zero digits of its problem (mixed additions) and ONE chain of top+1 doublings
                // (the K-1 redundant chains of a segmented launch buy latency, they are not work)
                uint64_t fold_muls = (nz * 7 / (2 * P) + (uint64_t)(top + 1) * 8 + 7) * (uint64_t)(2 * P * n_new);
                KSpan ks_fold(C.tm, C.stream, use_tab ? ROFL_TK_FOLD_TAB : ROFL_TK_FOLD, fold_muls, (uint64_t)2 * P * n_g * 32 + (uint64_t)2 * P * n_new * 32);
                if (use_tab)
                    hipLaunchKernelGGL(k_fold_gens_tab, grid, block, (K - 1) * 64 * sizeof(ge), C.stream, (u32)n_new, nsrc, seg, fc, tbl, (size_t)(2 * N),
                                       (const FoldTabProb *)d_fpv, d_dig, unit);
                else if (nsrc == 4 && unit && fold_regs)      // three scalar-carrying sources, kept in registers
                    hipLaunchKernelGGL(k_fold_gens4, grid, block, (K - 1) * 64 * sizeof(ge), C.stream, (u32)n_new, seg, (const FoldProb *)d_fpv, d_dig);
                else
                    hipLaunchKernelGGL(k_fold_gens, grid, block, (K - 1) * 64 * sizeof(ge), C.stream, (u32)n_new, nsrc, seg, (const FoldProb *)d_fpv, d_dig, unit);
            }
            if (C.tm.enabled) { HIPCHK(hipEventRecord(e1, C.stream)); C.tm.fold_ev.push_back({e0, e1}); C.tm.t.fold_launches++; { char tg[96]; snprintf(tg, sizeof tg, "fold n_g=%zu nsrc=%u tab=%d", n_g, nsrc, (int)use_tab); C.tm.fold_tag.push_back(tg); } C.tm.t.fold_point_reads += (uint64_t)2 * P * n_g; }
            HIPCHK(hipMemcpyAsync(d_cp, h_cp, sizeof(ChunkParams) * P, hipMemcpyHostToDevice, C.stream));   // gscale / hscale
            // no sync here: the next round's launches queue up behind the fold on the same stream (their enqueue cost hides under it); the
            // pinned digit / problem staging buffers are not written again before the next fold, at least two synchronised rounds away
            if (ptrace) C.sync();      // the phase trace wants the fold's own wall time
            mark("fold", (long)n_new);
            for (size_t c = 0; c < P; c++) { cur[c] = gnew + c * 2 * n_new; pu[c].clear(); pui[c].clear(); }
            n_g = n_new; r = 0; gsel ^= 1; first_level = false; just_materialised = true;
        }
    }
    // a[0], b[0]
    if (ab_on_host) {      // the last round's kernel left a_0, a_1, b_0, b_1 (Montgomery) in mapped host memory, the hop left u, u^-1 in h_round
        const sc *q = C.h_abfin.as<sc>(4 * P);
        for (size_t c = 0; c < P; c++) {
            uint8_t *o = proofs_out[c] + 7 * 32 + 64 * lgN;
            const sc &um = h_round[2 * c], &uim = h_round[2 * c + 1];
            sc_tobytes(o, h_canon(sc_add(sc_montmul(q[4 * c], um), sc_montmul(q[4 * c + 1], uim))));
            sc_tobytes(o + 32, h_canon(sc_add(sc_montmul(q[4 * c + 2], uim), sc_montmul(q[4 * c + 3], um))));
        }
        if (ptrace) {
            fprintf(stderr, "[rofl-hops] %d msm calls: enqueue %.3f ms, sync wait %.3f, horner wall %.3f (max task cpu %.3f), round-host wall %.3f (max task cpu %.3f)\n",
                    C.hs.n, C.hs.enqueue, C.hs.sync, C.hs.horner_wall, C.hs.horner_cpu, C.hs.host_wall, C.hs.host_cpu);
            C.hs = Ctx::HopStats();
        }
        return;
    }
    sc *h_ab = C.h_part.as<sc>(2 * P);
    // element 0 of every chunk: two strided copies (one copy per chunk and vector costs ~7 us of stream time each -- 0.9 ms at n_partition = 64)
    HIPCHK(hipMemcpy2DAsync(h_ab, 2 * sizeof(sc), a, N * sizeof(sc), sizeof(sc), P, hipMemcpyDeviceToHost, C.stream));
    HIPCHK(hipMemcpy2DAsync(h_ab + 1, 2 * sizeof(sc), b, N * sizeof(sc), sizeof(sc), P, hipMemcpyDeviceToHost, C.stream));
    C.sync();
    for (size_t c = 0; c < P; c++) {
        uint8_t *o = proofs_out[c] + 7 * 32 + 64 * lgN;
        sc_tobytes(o, h_canon(h_ab[2 * c])); sc_tobytes(o + 32, h_canon(h_ab[2 * c + 1]));
    }
    if (ptrace) {
        fprintf(stderr, "[rofl-hops] %d msm calls: enqueue %.3f ms, sync wait %.3f, horner wall %.3f (max task cpu %.3f), round-host wall %.3f (max task cpu %.3f)\n",
                C.hs.n, C.hs.enqueue, C.hs.sync, C.hs.horner_wall, C.hs.horner_cpu, C.hs.host_wall, C.hs.host_cpu);
        C.hs = Ctx::HopStats();
    }
}

// ================================================================ verifier (RangeProof::verify_multiple)
// P chunks; proofs host [P][plen]; V bytes host [P][m][32]; V niels device [P][m].
// results: ok[P] (0/1); returns error code (format etc.)
sc verifier_c(const uint8_t seed[32], u64 idx) {
    const u64 dom[2] = {0x2f6b7a2d6c666f72ULL, 0x31762f6379667276ULL};  // "rofl-zk/" "vrfyc/v1"
    u64 sd[4]; memcpy(sd, seed, 32);
    u64 st[25]; shake256_seeded_block(st, dom, sd, idx);
    sc lo, hi;
    for (int i = 0; i < 4; i++) { lo.v[2 * i] = (u32)st[i]; lo.v[2 * i + 1] = (u32)(st[i] >> 32); hi.v[2 * i] = (u32)st[4 + i]; hi.v[2 * i + 1] = (u32)(st[4 + i] >> 32); }
    return sc_from_wide(lo, hi);
}

// d_Vniels holds the UNSHIFTED commitments C_j (k_decode); h_V the encodings of V_j = C_j + v_shift * B for the first v_real[c] values of
// chunk c (identity padding after them): the check needs sum_j s_j V_j, which is the MSM over the C_j plus (sum_{j < v_real} s_j) * v_shift on B.
int verify_chunks(Ctx &C, const char *label, size_t gens_capacity, size_t P, size_t n, size_t m, const uint8_t *proofs, size_t plen,
                  const uint8_t *h_V, const niels *d_Vniels, const uint8_t seed[32], const u64 *c_index, int *ok, size_t group = 1,
                  const sc *v_shift = nullptr, const u64 *v_real = nullptr) {
    // `group` consecutive proofs are checked as one batch: sum_c rho_c * (check_c) == 0 with random weights rho_c, so their
    // generator terms share one MSM.  Every proof of a batch gets the batch's verdict (callers AND them per client anyway).
    for (size_t c = 0; c < P; c++) ok[c] = 0;
    static const bool vtrace = knob("ROFL_TRACE") && atoi(knob("ROFL_TRACE")) >= 2;
    double vt0 = now_ms(), vtl = vt0;
    auto vmark = [&](const char *what) { if (!vtrace) return; double t = now_ms(); fprintf(stderr, "[rofl-trace verify] %-14s +%.3f ms  (t=%.3f)\n", what, t - vtl, t - vt0); vtl = t; };
    if (group == 0 || P % group) group = 1;
    size_t ngroups = P / group;
    // RangeProof::from_bytes / InnerProductProof::from_bytes
    if (plen % 32 != 0 || plen < 7 * 32) return ROFL_FORMAT_ERROR;
    size_t ne = (plen - 7 * 32) / 32;
    if (ne < 2 || (ne - 2) % 2 != 0) return ROFL_FORMAT_ERROR;
    size_t lg = (ne - 2) / 2;
    if (lg >= 32) return ROFL_FORMAT_ERROR;
    for (size_t c = 0; c < P; c++) {
        const uint8_t *p = proofs + c * plen;
        if (!sc_is_canonical_bytes(p + 128) || !sc_is_canonical_bytes(p + 160) || !sc_is_canonical_bytes(p + 192) ||
            !sc_is_canonical_bytes(p + 7 * 32 + 64 * lg) || !sc_is_canonical_bytes(p + 7 * 32 + 64 * lg + 32))
            return ROFL_FORMAT_ERROR;
    }
    if (!(n == 8 || n == 16 || n == 32 || n == 64)) return ROFL_INVALID_BITSIZE;
    if (gens_capacity < n) return ROFL_INVALID_GENS_LENGTH;
    size_t N = n * m;
    std::vector<char> dead(P, 0);
    if (N != ((size_t)1 << lg)) return ROFL_OK;    // VerificationError for every chunk
    GensPin gens = get_gens(C, n, m);
    niels *tbl = gens.tbl();
    const niels *wtab = gens.wtab();
    ChunkParams *h_cp = C.h_cp.as<ChunkParams>(P);
    ChunkParams *d_cp = C.cp.as<ChunkParams>(P);
    memset(h_cp, 0, sizeof(ChunkParams) * P);
    size_t naux = m + 4 + 2 * lg;
    uint8_t *h_auxc = C.h_auxc.as<uint8_t>(P * (4 + 2 * lg) * 32);
    sc *h_auxs = C.h_auxs.as<sc>(P * (4 + 2 * lg));
    std::vector<sc> sB(P), sBb(P);
    static const uint8_t zero32[32] = {0};
    // the proof points A S T1 T2 L* R* of every chunk go to the device and are decoded while the host hashes the transcripts
    for (size_t c = 0; c < P; c++) {
        const uint8_t *p = proofs + c * plen; const uint8_t *ipp = p + 7 * 32;
        uint8_t *ac = h_auxc + c * (4 + 2 * lg) * 32;
        memcpy(ac, p, 128);
        for (size_t k = 0; k < lg; k++) { memcpy(ac + 128 + 32 * k, ipp + 64 * k, 32); memcpy(ac + 128 + 32 * (lg + k), ipp + 64 * k + 32, 32); }
    }
    uint8_t *d_auxc = C.tmp_in.as<uint8_t>(P * (4 + 2 * lg) * 32);
    niels *d_auxn = C.tmp_in2.as<niels>(P * (4 + 2 * lg));
    u32 *status = C.status.as<u32>(4);
    HIPCHK(hipMemsetAsync(status, 0, 16, C.stream));
    HIPCHK(hipMemcpyAsync(d_auxc, h_auxc, P * (4 + 2 * lg) * 32, hipMemcpyHostToDevice, C.stream));
    hipLaunchKernelGGL(k_decode, grid1(P * (4 + 2 * lg)), dim3(TPB), 0, C.stream, (u32)(P * (4 + 2 * lg)), (u32)(P * (4 + 2 * lg)), d_auxc, (const niels *)nullptr, d_auxn, (uint8_t *)nullptr, status);
    double th = now_ms();
    C.pool->run(P, [&](size_t c) {
        const uint8_t *p = proofs + c * plen; const uint8_t *ipp = p + 7 * 32;
        Merlin t(label, strlen(label));
        t.append("dom-sep", (const uint8_t *)"rangeproof v1", 13);
        t.append_u64("n", n); t.append_u64("m", m);
        t.append32_run('V', h_V + c * m * 32, m);
        // validate_and_append_point rejects the identity encoding
        bool bad = false;
        for (int i = 0; i < 4; i++) if (!memcmp(p + 32 * i, zero32, 32)) bad = true;
        for (size_t k = 0; k < 2 * lg; k++) if (!memcmp(ipp + 32 * k, zero32, 32)) bad = true;
        if (bad) { dead[c] = 1; }
        t.append("A", p, 32); t.append("S", p + 32, 32);
        sc y = t.challenge_scalar("y"), z = t.challenge_scalar("z");
        t.append("T_1", p + 64, 32); t.append("T_2", p + 96, 32);
        sc x = t.challenge_scalar("x");
        sc t_x = sc_frombytes(p + 128), t_x_bl = sc_frombytes(p + 160), e_bl = sc_frombytes(p + 192);
        t.append("t_x", p + 128, 32); t.append("t_x_blinding", p + 160, 32); t.append("e_blinding", p + 192, 32);
        sc w = t.challenge_scalar("w");
        sc cc = verifier_c(seed, c_index[c]);
        sc rho = group > 1 ? verifier_c(seed, c_index[c] | (1ULL << 62)) : sc_one_plain();
        t.append("dom-sep", (const uint8_t *)"ipp v1", 6);
        t.append_u64("n", N);
        ChunkParams &cp = h_cp[c];
        sc a = sc_frombytes(ipp + 64 * lg), b = sc_frombytes(ipp + 64 * lg + 32);
        std::vector<sc> u(lg), ui(lg);
        for (size_t k = 0; k < lg; k++) {
            t.append("L", ipp + 64 * k, 32); t.append("R", ipp + 64 * k + 32, 32);
            u[k] = t.challenge_scalar("u");
            cp.u[k] = h_mont(u[k]);
        }
        sc zz = h_mul(z, z);
        cp.y = h_mont(y); cp.z = h_mont(z); cp.zz = h_mont(zz); cp.x = h_mont(x);
        {   // u_0^-1 .. u_(lg-1)^-1 and y^-1 with ONE inversion (10 us each otherwise, on the verifier's critical path): prefix products,
            // invert the last, walk back.  A zero challenge (probability 2^-252) makes every inverse zero, as the single inversions would.
            std::vector<sc> pre(lg + 1);
            sc run = cp.y; pre[0] = run;
            for (size_t k = 0; k < lg; k++) { run = sc_montmul(run, cp.u[k]); pre[k + 1] = run; }
            bool any_zero = sc_iszero(h_canon(run));
            sc inv = any_zero ? sc_zero() : h51::sc_invert_mont_fast(run);
            for (size_t k = lg; k >= 1; k--) { cp.uinv[k - 1] = sc_montmul(inv, pre[k - 1]); inv = sc_montmul(inv, cp.u[k - 1]); }
            cp.yinv = inv;
            if (any_zero) { cp.yinv = h51::sc_invert_mont_fast(cp.y); for (size_t k = 0; k < lg; k++) cp.uinv[k] = h51::sc_invert_mont_fast(cp.u[k]); }
            for (size_t k = 0; k < lg; k++) ui[k] = h_canon(cp.uinv[k]);
        }
        fill_pow2(cp.ypow2, cp.y, MAX_LG); fill_pow2(cp.yinvpow2, cp.yinv, MAX_LG); fill_pow2(cp.zpow2, cp.z, MAX_LG);
        cp.a_fin = h_mont(a); cp.b_fin = h_mont(b);
        cp.c_zz = h_mont(h_mul(rho, h_mul(cc, zz)));
        cp.rz = h_mont(h_mul(rho, z)); cp.ra = h_mont(h_mul(rho, a)); cp.rb = h_mont(h_mul(rho, b)); cp.rzz = h_mont(h_mul(rho, zz));
        // aux points and scalars: A S T1 T2 L* R*
        sc *as = h_auxs + c * (4 + 2 * lg);
        as[0] = rho; as[1] = h_mul(rho, x); as[2] = h_mul(as[1], cc); as[3] = h_mul(as[2], x);
        for (size_t k = 0; k < lg; k++) { as[4 + k] = h_mul(rho, h_mul(u[k], u[k])); as[4 + lg + k] = h_mul(rho, h_mul(ui[k], ui[k])); }
        // B_blinding: -e_bl - c t_x_bl ; B: w (t_x - a b) + c (delta - t_x)
        sBb[c] = h_mul(rho, sc_neg(sc_add(e_bl, h_mul(cc, t_x_bl))));
        // sum_{i<N} y^i = prod_b (1 + y^(2^b)) for N = 2^lg ; likewise for 2^n and z^m
        auto geo = [&](const sc *pow2tab, unsigned bits) { sc acc = sc_one_mont(); for (unsigned q = 0; q < bits; q++) acc = sc_montmul(acc, sc_add(sc_one_mont(), pow2tab[q])); return h_canon(acc); };
        sc sum_y = geo(cp.ypow2, (unsigned)lg);
        sc sum_z = geo(cp.zpow2, lg2u(m));
        sc twom[MAX_LG]; fill_pow2(twom, h_mont(sc_from_u64(2)), 8);
        sc sum_2 = geo(twom, lg2u(n));
        sc delta = sc_sub(h_mul(sc_sub(z, zz), sum_y), h_mul(h_mul(h_mul(zz, z), sum_2), sum_z));
        sB[c] = h_mul(rho, sc_add(h_mul(w, sc_sub(t_x, h_mul(a, b))), h_mul(cc, sc_sub(delta, t_x))));
        if (v_shift && v_real && v_real[c]) {
            // sum_{j < cnt} rho c z^(2+j) = rho c z^2 (z^cnt - 1) / (z - 1)
            sc zm = h_mont(z), zc = sc_one_mont();
            for (u64 e = v_real[c], bidx = 0; e; e >>= 1, bidx++) if (e & 1) zc = sc_montmul(zc, cp.zpow2[bidx]);
            sc num = sc_sub(h_canon(zc), sc_one_plain()), den = sc_sub(z, sc_one_plain());
            sc geo_z = sc_iszero(den) ? sc_from_u64(v_real[c]) : h_mul(num, h_inv(den));
            (void)zm;
            sB[c] = sc_add(sB[c], h_mul(h_mul(h_mul(rho, h_mul(cc, zz)), geo_z), *v_shift));
        }
    });
    C.tm.t.host_ms += now_ms() - th;
    vmark("transcripts");
    HIPCHK(hipMemcpyAsync(d_cp, h_cp, sizeof(ChunkParams) * P, hipMemcpyHostToDevice, C.stream));
    sc *gh = C.SL.as<sc>(ngroups * 2 * N);
    PowTabs *d_pt = C.powtabs.as<PowTabs>(P);
    hipLaunchKernelGGL(k_pow_tables, dim3((4 * PT_L * PT_E + TPB - 1) / TPB, (u32)P), dim3(TPB), 0, C.stream, d_cp, d_pt, (u32)lg, 1);
    hipLaunchKernelGGL(k_verify_scalars, grid1(N, (u32)ngroups), dim3(TPB), 0, C.stream, (u32)n, (u32)m, (u32)lg, (u32)group, d_cp, (const PowTabs *)d_pt, C.d_two_pow, gh);
    // aux arrays
    niels *aux_pts = C.aux_pts.as<niels>(P * naux);
    sc *aux_scal = C.aux_scal.as<sc>(P * naux);
    hipLaunchKernelGGL(k_vscalars, grid1(m, (u32)P), dim3(TPB), 0, C.stream, (u32)m, d_cp, (const PowTabs *)d_pt, aux_scal, naux);
    {   // per chunk: [m commitments | 4 + 2 lg proof points] and the scalars of the latter -- three strided copies for all chunks
        const size_t na2 = 4 + 2 * lg;
        HIPCHK(hipMemcpy2DAsync(aux_pts, naux * sizeof(niels), d_Vniels, m * sizeof(niels), m * sizeof(niels), P, hipMemcpyDeviceToDevice, C.stream));
        HIPCHK(hipMemcpy2DAsync(aux_pts + m, naux * sizeof(niels), d_auxn, na2 * sizeof(niels), na2 * sizeof(niels), P, hipMemcpyDeviceToDevice, C.stream));
        HIPCHK(hipMemcpy2DAsync(aux_scal + m, naux * sizeof(sc), h_auxs, na2 * sizeof(sc), na2 * sizeof(sc), P, hipMemcpyHostToDevice, C.stream));
    }
    u32 *h_stat = C.h_misc2.as<u32>(4);
    HIPCHK(hipMemcpyAsync(h_stat, status, 4, hipMemcpyDeviceToHost, C.stream));      // k_decode's verdict on the proof points
    std::vector<MsmProb> pr(ngroups), prB(ngroups); std::vector<ge5> resA, resB;
    for (size_t g = 0; g < ngroups; g++) pr[g] = MsmProb{tbl, gh + g * 2 * N};
    for (size_t g = 0; g < ngroups; g++) prB[g] = MsmProb{aux_pts + g * group * naux, aux_scal + g * group * naux};
    C.tm.t.msm_terms += ngroups * 2 * N + P * naux;
    // the generator MSM (2N terms per group, fixed-base) and the proof-point MSM (commitments, A, S, T, L, R) queue back to back: one wait
    // (one problem per client with every window in its own bucket set: the 15-bit layout's smaller arrays win here whenever it exists)
    { MsmOpt mo; if (wtab) { gens.fb_for(1000, &mo.fb_wtab, &mo.fb_c); mo.fb_stride = 2 * N; } msm_run2(C, pr, 2 * N, mo, resA, prB, group * naux, MsmOpt(), resB); }
    vmark("msm");
    const u32 h_status = *h_stat;
    th = now_ms();
    for (size_t g = 0; g < ngroups; g++) {
        ge5 tot = h51::gadd(resA[g], resB[g]);
        sc b1 = sc_zero(), b2 = sc_zero(); bool any_dead = false;
        for (size_t c = g * group; c < (g + 1) * group; c++) { b1 = sc_add(b1, sB[c]); b2 = sc_add(b2, sBb[c]); any_dead |= dead[c] != 0; }
        tot = h51::gadd(tot, h_fixed_mul(C.ht.B5, b1));
        tot = h51::gadd(tot, h_fixed_mul(C.ht.Bb5, b2));
        int okg = (!any_dead && h51::is_identity_ristretto(tot)) ? 1 : 0;
        for (size_t c = g * group; c < (g + 1) * group; c++) ok[c] = okg;
    }
    C.tm.t.host_ms += now_ms() - th;
    if (h_status & 4u) {
        // some proof point failed to decompress: upstream returns VerificationError for that proof.
        // Re-check per chunk on the host to attribute the failure.
        for (size_t c = 0; c < P; c++) {
            const uint8_t *ac = h_auxc + c * (4 + 2 * lg) * 32;
            for (size_t k = 0; k < 4 + 2 * lg; k++) { ge tmp; if (!ristretto_decode(tmp, ac + 32 * k)) for (size_t c2 = c / group * group; c2 < (c / group + 1) * group; c2++) ok[c2] = 0; }
        }
    }
    return ROFL_OK;
}

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
thread_local rofl_kernel_time_t g_last_ktimes[ROFL_TK_COUNT]{};
void timing_begin(Ctx &C) {
    C.tm.reset();
    if (C.tm.enabled) { C.tm.first = C.tm.get(); C.tm.last = C.tm.get(); HIPCHK(hipEventRecord(C.tm.first, C.stream)); }
}
void timing_end(Ctx &C) {
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
                uint8_t *const *commits_out, int *rcs) {
    for (size_t i = 0; i < nc; i++) rcs[i] = ROFL_OK;
    if (!valid_fp(fp_bits, fp_frac) || d == 0 || n_partition == 0 || prove_range == 0 || prove_range > fp_bits || !nonces || nc == 0)
        return fail(ROFL_BAD_PARAM, "bad parameter (the reference panics here)");
    size_t dp = next_pow2(d);
    size_t n_chunks = std::min(dp, n_partition), chunk = dp / n_chunks;
    size_t P = (dp + chunk - 1) / chunk;
    float mn, mx; clip_bounds(prove_range, fp_bits, fp_frac, &mn, &mx);
    C.init();
    C.batch_mode = nc > 1;
    timing_begin(C);
    float *d_vals = C.vals.as<float>(nc * d);
    u64 *vshift = C.vshift.as<u64>(nc * dp);
    sc *d_blind_buf = C.blind.as<sc>(nc * dp);
    HIPCHK(hipMemsetAsync(d_blind_buf, 0, sizeof(sc) * nc * dp, C.stream));
    for (size_t i = 0; i < nc; i++) {      // the callers' arrays: host or device memory
        HIPCHK(hipMemcpyAsync(d_vals + i * d, values[i], sizeof(float) * d, hipMemcpyDefault, C.stream));
        HIPCHK(hipMemcpyAsync(d_blind_buf + i * dp, blind[i], 32 * d, hipMemcpyDefault, C.stream));
    }
    u32 *status = C.status.as<u32>(nc + 4);
    HIPCHK(hipMemsetAsync(status, 0, 4 * (nc + 4), C.stream));
    hipLaunchKernelGGL(k_quantize_shift, grid1(dp, (u32)nc), dim3(TPB), 0, C.stream, d_vals, (u32)d, (u32)dp, (u32)prove_range, fp_bits, fp_frac, mn, mx, vshift, status);
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
    if (nc == 1 && rcs[0] == ROFL_VALUE_OUT_OF_RANGE) return fail(ROFL_VALUE_OUT_OF_RANGE, "ValueOutOfRangeError");
    if (nc == 1 && rcs[0] == ROFL_NON_FINITE) return fail(ROFL_NON_FINITE, "non-finite value (the reference panics in fixed::saturating_from_float)");
    if (!is_pow2(chunk) || dp % chunk) return fail(ROFL_INVALID_AGGREGATION, "InvalidAggregation (the reference panics)");
    if (!(prove_range == 8 || prove_range == 16 || prove_range == 32 || prove_range == 64)) return fail(ROFL_INVALID_BITSIZE, "InvalidBitsize");
    for (size_t i = 0; i < nc; i++)
        if (rcs[i] == ROFL_OK && nonces[i].mode == 0 && nonces[i].stream_scalars < P * chunk * (2 * prove_range + 4)) {
            rcs[i] = ROFL_NONCE_SHORT;
            if (nc == 1) return fail(ROFL_NONCE_SHORT, "nonce stream too short");
        }
    size_t plen = 32 * (9 + 2 * (size_t)lg2u(prove_range * chunk));
    *plen_out = plen; *np_out = P;
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
    { size_t tot = 0; for (size_t k = 0; k < na; k++) if (nonces[act[k]].mode == 0) tot += nonces[act[k]].stream_scalars * 64;
      uint8_t *sb = tot ? C.stream_buf.as<uint8_t>(tot + 64) : nullptr; size_t off = 0;
      u64 per = (u64)chunk * (2 * prove_range + 4);
      for (size_t k = 0; k < na; k++) {
          const rofl_nonce_t &nn = nonces[act[k]];
          ChunkNonce base{}; base.mode = nn.mode;
          if (nn.mode == 1) memcpy(base.seed.w, nn.seed, 32);
          else { HIPCHK(hipMemcpyAsync(sb + off, nn.stream, nn.stream_scalars * 64, hipMemcpyHostToDevice, C.stream)); base.d_stream = sb + off; base.stream_scalars = nn.stream_scalars; off += nn.stream_scalars * 64; }
          for (size_t c = 0; c < P; c++) { cn[k * P + c] = base; cn[k * P + c].base = c * per; }
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
    hipLaunchKernelGGL(k_commit, grid1(na * dp), dim3(TPB), 0, C.stream2, (u32)(na * dp), vshift, (const sc *)nullptr, d_blind_buf, C.d_tabB, C.d_tabBb, d_shift, Vb, Cb, (u32)d, (u32)dp);
    uint8_t *hV = C.h_V.as<uint8_t>(na * dp * 32);
    HIPCHK(hipMemcpyAsync(hV, Vb, na * dp * 32, hipMemcpyDeviceToHost, C.stream2));
    HIPCHK(hipEventRecord(C.ev_v, C.stream2));
    for (size_t k = 0; k < na; k++) HIPCHK(hipMemcpyAsync(commits_out[act[k]], Cb + k * dp * 32, d * 32, hipMemcpyDeviceToHost, C.stream2));
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
int verify_impl(Ctx &C, size_t n_clients, const uint8_t *const *proofs, size_t proof_len, size_t n_proofs, const uint8_t *const *commits,
                size_t d, size_t prove_range, unsigned fp_bits, unsigned fp_frac, const uint8_t seed[32], int *ok_out) {
    for (size_t i = 0; i < n_clients; i++) ok_out[i] = 0;
    if (!valid_fp(fp_bits, fp_frac) || d == 0 || n_proofs == 0 || prove_range == 0 || prove_range > fp_bits || n_clients == 0)
        return fail(ROFL_BAD_PARAM, "bad parameter (the reference panics here)");
    size_t dp = next_pow2(d);
    size_t chunk = dp / n_proofs;
    if (chunk == 0) return fail(ROFL_BAD_PARAM, "more proofs than padded commitments (the reference panics in chunks(0))");
    size_t n_chunks = (dp + chunk - 1) / chunk;
    size_t nv = std::min(n_proofs, n_chunks);            // zip truncates (range_proof_vec/mod.rs:173-176)
    const Ctx &P0 = C.parent ? *C.parent : C;
    const bool zip_truncate = P0.opt_zip_truncate != 0;      // rofl_set_option("verify_zip_truncate")
    if (n_proofs * chunk != dp) {
        if (!zip_truncate) { g_err = "proof count does not cover the padded commitment vector: not verified"; return ROFL_OK; }
        if (dp % chunk) return fail(ROFL_BAD_PARAM, "ragged chunks are not supported");
    }
    // Everything below allocates per (prove_range, chunk): check the proofs' own shape against it first (RangeProof::from_bytes,
    // then the N == 2^lg test of verify_multiple), so that a forged proof count cannot make the device build tables.
    if (proof_len % 32 != 0 || proof_len < 7 * 32) return fail(ROFL_FORMAT_ERROR, "FormatError: proof length");
    size_t ne = (proof_len - 7 * 32) / 32;
    if (ne < 2 || (ne - 2) % 2 != 0 || (ne - 2) / 2 >= 32) return fail(ROFL_FORMAT_ERROR, "FormatError: proof length");
    size_t lg = (ne - 2) / 2;
    size_t P = n_clients * nv;
    std::vector<uint8_t> pf(P * proof_len);
    for (size_t i = 0; i < n_clients; i++)
        HIPCHK(hipMemcpy(&pf[i * nv * proof_len], proofs[i], nv * proof_len, hipMemcpyDefault));   // host or device memory
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
                if (n_clients == 1) return fail(ROFL_FORMAT_ERROR, "proof rejected before verification (format / bitsize)");
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
    std::vector<uint8_t> hV(tot * 32);
    for (size_t i = 0; i < n_clients; i++) {
        HIPCHK(hipMemcpyAsync(d_in + i * dp * 32, commits[i], d * 32, hipMemcpyDefault, C.stream));
        hipLaunchKernelGGL(k_decode, grid1(dp), dim3(TPB), 0, C.stream, (u32)dp, (u32)d, d_in + i * dp * 32, d_shift, d_vn + i * dp, d_enc + i * dp * 32, status + i);
    }
    HIPCHK(hipMemcpyAsync(hV.data(), d_enc, tot * 32, hipMemcpyDeviceToHost, C.stream));
    u32 *h_st = C.h_misc.as<u32>(n_clients + 4);
    HIPCHK(hipMemcpyAsync(h_st, status, 4 * n_clients, hipMemcpyDeviceToHost, C.stream));
    C.sync();
    std::vector<char> bad_commit(n_clients, 0);
    for (size_t i = 0; i < n_clients; i++) bad_commit[i] = (h_st[i] & 4u) != 0;
    // a single set with an invalid encoding: the reference cannot even build its Vec<RistrettoPoint> (decompress fails) -> FormatError;
    // in a batch the other clients are still verified and the offender gets ok = 0
    if (n_clients == 1 && bad_commit[0]) return fail(ROFL_FORMAT_ERROR, "commitment is not a valid Ristretto encoding");
    // flatten (client, chunk) -> problem list
    std::vector<uint8_t> Vh(P * chunk * 32);
    std::vector<u64> cidx(P);
    niels *d_vn2 = C.gbuf[1].as<niels>(P * chunk);
    for (size_t i = 0; i < n_clients; i++)
        for (size_t c = 0; c < nv; c++) {
            size_t q = i * nv + c;
            memcpy(&Vh[q * chunk * 32], &hV[(i * dp + c * chunk) * 32], chunk * 32);
            cidx[q] = c;
        }
    // a client's verified chunks are a prefix of its dp commitments: one strided copy (a plain one when the proofs cover everything)
    HIPCHK(hipMemcpy2DAsync(d_vn2, nv * chunk * sizeof(niels), d_vn, dp * sizeof(niels), nv * chunk * sizeof(niels), n_clients, hipMemcpyDeviceToDevice, C.stream));
    std::vector<int> okc(P);
    GensPin gens_pin = get_gens(C, prove_range, chunk);
    C.sync();
    const bool vbatch = P0.opt_verify_batch != 0;             // rofl_set_option("verify_batch")
    size_t grp = vbatch ? nv : 1;
    // all (client, chunk) pairs in one pass; a client's chunks form one batch of the random-weighted check
    sc v_shift = sc_from_u64(1ULL << (prove_range - 1));
    std::vector<u64> v_real(P);
    for (size_t q = 0; q < P; q++) { size_t lo = cidx[q] * chunk; v_real[q] = lo >= d ? 0 : std::min(chunk, d - lo); }
    int rc = verify_chunks(C, "RangeProof", prove_range, P, prove_range, chunk, pf.data(), proof_len, Vh.data(), d_vn2, seed, cidx.data(), okc.data(), grp, &v_shift, v_real.data());
    timing_end(C);
    if (rc) return fail(rc, "proof rejected before verification (format / bitsize)");
    for (size_t i = 0; i < n_clients; i++) { int r = (bad_commit[i] || bad_format[i]) ? 0 : 1; for (size_t c = 0; c < nv; c++) r &= okc[i * nv + c]; ok_out[i] = r; }
    return ROFL_OK;
}

}  // namespace

// ================================================================ C ABI
extern "C" {

int rofl_set_device(int device) { g_device = device; return guarded([&]() -> int { LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c; C.init(); return ROFL_OK; }); }
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
    return guarded([&]() -> int { LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c; C.init(); if (!n_bits || !m) return fail(ROFL_BAD_PARAM, "bad parameter"); get_gens(C, n_bits, m); return ROFL_OK; });
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
        GensPin gens = get_gens(C, n_bits, m); niels *tbl = gens.tbl(); size_t N = n_bits * m;
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
    return guarded([&]() -> int { LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
        if (d != d_blindings) return fail(ROFL_WRONG_NUM_BLINDING, "WrongNumBlindingFactors");
        int rc1 = ROFL_OK;
        int rc = create_impl(C, 1, &values, d, &blindings32, prove_range, n_partition, fp_bits, fp_frac, nonce, &proofs_out, proof_len_out, n_proofs_out, &commits_out, &rc1);
        return rc ? rc : rc1; });
}
int rofl_create_rangeproof_batch(size_t n_clients, const float *const *values, size_t d, const uint8_t *const *blindings32, size_t prove_range,
                                 size_t n_partition, unsigned fp_bits, unsigned fp_frac, const rofl_nonce_t *nonces, uint8_t *const *proofs_out,
                                 size_t *proof_len_out, size_t *n_proofs_out, uint8_t *const *commits_out, int *rc_out) {
    return guarded([&]() -> int { LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
        if (!values || !blindings32 || !proofs_out || !commits_out || !rc_out || !proof_len_out || !n_proofs_out) return fail(ROFL_BAD_PARAM, "bad parameter");
        return create_impl(C, n_clients, values, d, blindings32, prove_range, n_partition, fp_bits, fp_frac, nonces, proofs_out, proof_len_out, n_proofs_out, commits_out, rc_out); });
}
int rofl_verify_rangeproof(const uint8_t *proofs, size_t proof_len, size_t n_proofs, const uint8_t *commits32, size_t d, size_t prove_range,
                           unsigned fp_bits, unsigned fp_frac, const uint8_t verifier_seed[32], int *ok_out) {
    return guarded([&]() -> int { LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
        return verify_impl(C, 1, &proofs, proof_len, n_proofs, &commits32, d, prove_range, fp_bits, fp_frac, verifier_seed, ok_out); });
}
int rofl_verify_rangeproof_batch(size_t n_clients, const uint8_t *const *proofs, size_t proof_len, size_t n_proofs, const uint8_t *const *commits32,
                                 size_t d, size_t prove_range, unsigned fp_bits, unsigned fp_frac, const uint8_t verifier_seed[32], int *ok_out) {
    return guarded([&]() -> int { LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
        return verify_impl(C, n_clients, proofs, proof_len, n_proofs, commits32, d, prove_range, fp_bits, fp_frac, verifier_seed, ok_out); });
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
        LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
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
        hipLaunchKernelGGL(k_commit, grid1(1), dim3(TPB), 0, C.stream, 1u, vshift, (const sc *)nullptr, d_bl, C.d_tabB, C.d_tabBb, (const niels *)nullptr, Vb, (uint8_t *)nullptr, 0u, 1u);
        uint8_t hV[32];
        HIPCHK(hipMemcpyAsync(hV, Vb, 32, hipMemcpyDeviceToHost, C.stream));
        C.sync();
        // BulletproofGens::new(64, 1), label "L2RangeProof" (l2_range_proof_vec/mod.rs:156-171): the first
        // prove_range generators of party 0 are the same chain prefix.
        ChunkNonce cn{}; cn.mode = nonce->mode;
        if (nonce->mode == 1) memcpy(cn.seed.w, nonce->seed, 32);
        else { uint8_t *sb = C.stream_buf.as<uint8_t>(nonce->stream_scalars * 64 + 64); HIPCHK(hipMemcpyAsync(sb, nonce->stream, nonce->stream_scalars * 64, hipMemcpyHostToDevice, C.stream)); cn.d_stream = sb; cn.stream_scalars = nonce->stream_scalars; }
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
        LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
        *ok_out = 0;
        if (!valid_fp(fp_bits, fp_frac) || prove_range == 0) return fail(ROFL_BAD_PARAM, "bad parameter");
        C.init();
        timing_begin(C);
        uint8_t *d_in = C.Cbytes.as<uint8_t>(32), *d_enc = C.Vbytes.as<uint8_t>(32);
        niels *d_vn = C.gbuf[0].as<niels>(1);
        u32 *status = C.status.as<u32>(4);
        HIPCHK(hipMemsetAsync(status, 0, 16, C.stream));
        HIPCHK(hipMemcpyAsync(d_in, commit, 32, hipMemcpyHostToDevice, C.stream));
        hipLaunchKernelGGL(k_decode, grid1(1), dim3(TPB), 0, C.stream, 1u, 1u, d_in, (const niels *)nullptr, d_vn, d_enc, status);
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
int sigma_create(int kind, const float *values, size_t d, const uint8_t *r1, size_t d_r1, const uint8_t *r2, const uint8_t *existing,
                 unsigned fp_bits, unsigned fp_frac, const rofl_nonce_t *nonce, uint8_t *proofs_out, uint8_t *commits_out) {
    LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
    if (d != d_r1) return fail(ROFL_WRONG_NUM_BLINDING, "WrongNumBlindingFactors");
    if (!valid_fp(fp_bits, fp_frac) || !nonce) return fail(ROFL_BAD_PARAM, "bad parameter");
    if (d == 0) return ROFL_OK;
    bool has_sq = kind != 0;
    size_t npts = 1 + (kind != 2) + (has_sq ? 1 : 0), nn = has_sq ? 3 : 2, clen = 32 * npts, plen = 32 * (npts + nn);
    if (nonce->mode == 0 && nonce->stream_scalars < nn * d) return fail(ROFL_NONCE_SHORT, "nonce stream too short");
    C.init();
    timing_begin(C);
    float *dv = C.vals.as<float>(d); sc *dr1 = C.tmp_in.as<sc>(d); sc *dr2 = has_sq ? C.tmp_in2.as<sc>(d) : nullptr;
    uint8_t *dex = existing ? C.Cbytes.as<uint8_t>(d * 32) : nullptr;
    uint8_t *dp = C.aux_pts.as<uint8_t>(d * plen), *dc = C.aux_scal.as<uint8_t>(d * clen);
    u32 *status = C.status.as<u32>(4);
    HIPCHK(hipMemsetAsync(status, 0, 16, C.stream));
    HIPCHK(hipMemcpyAsync(dv, values, 4 * d, hipMemcpyDefault, C.stream));
    HIPCHK(hipMemcpyAsync(dr1, r1, 32 * d, hipMemcpyDefault, C.stream));
    if (has_sq) HIPCHK(hipMemcpyAsync(dr2, r2, 32 * d, hipMemcpyDefault, C.stream));
    if (dex) HIPCHK(hipMemcpyAsync(dex, existing, 32 * d, hipMemcpyDefault, C.stream));
    NonceSeed seed{}; const uint8_t *d_stream = nullptr; u64 ss = 0;
    if (nonce->mode == 1) memcpy(seed.w, nonce->seed, 32);
    else { ss = nonce->stream_scalars; uint8_t *sb = C.stream_buf.as<uint8_t>(ss * 64 + 64); HIPCHK(hipMemcpyAsync(sb, nonce->stream, ss * 64, hipMemcpyHostToDevice, C.stream)); d_stream = sb; }
    hipLaunchKernelGGL(k_sigma_prove, dim3((unsigned)((d + 63) / 64)), dim3(64), 0, C.stream, kind, (u32)d, dv, fp_bits, fp_frac, dr1, dr2, dex,
                       nonce->mode, seed, d_stream, ss, sigma_init_state(kind), C.d_tabB, C.d_tabBb, dp, dc, status);
    u32 st = 0;
    HIPCHK(hipMemcpyAsync(proofs_out, dp, d * plen, hipMemcpyDeviceToHost, C.stream));
    HIPCHK(hipMemcpyAsync(commits_out, dc, d * clen, hipMemcpyDeviceToHost, C.stream));
    HIPCHK(hipMemcpyAsync(&st, status, 4, hipMemcpyDeviceToHost, C.stream));
    C.sync();
    timing_end(C);
    if (st & 2u) return fail(ROFL_NON_FINITE, "non-finite value (the reference panics in fixed::saturating_from_float)");
    if (st & 4u) return fail(ROFL_FORMAT_ERROR, "invalid Ristretto encoding");
    return ROFL_OK;
}
int sigma_verify(int kind, const uint8_t *proofs, const uint8_t *commits, size_t d, int *ok_out) {
    LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
    *ok_out = 0;
    if (d == 0) { *ok_out = 1; return ROFL_OK; }
    size_t npts = 1 + (kind != 2) + (kind != 0 ? 1 : 0), clen = 32 * npts, plen = 32 * (npts + (kind != 0 ? 3 : 2));
    C.init();
    timing_begin(C);
    uint8_t *dp = C.aux_pts.as<uint8_t>(d * plen), *dc = C.aux_scal.as<uint8_t>(d * clen);
    u32 *status = C.status.as<u32>(4);
    HIPCHK(hipMemsetAsync(status, 0, 16, C.stream));
    HIPCHK(hipMemcpyAsync(dp, proofs, d * plen, hipMemcpyDefault, C.stream));
    HIPCHK(hipMemcpyAsync(dc, commits, d * clen, hipMemcpyDefault, C.stream));
    const bool sg_batch = (C.parent ? C.parent : &C)->opt_sigma_batch != 0;      // rofl_set_option("sigma_batch")
    if (sg_batch) {
        // one random linear combination of all elements' equations: decode + transcripts per element on the device, then ONE Pippenger MSM
        // over the 4-6 d points and two fixed-base terms (k_sigma_vprep); the weights come from fresh OS randomness
        size_t nslots = 2 * npts, nblk = (d + TPB - 1) / TPB;
        NonceSeed ws{};
        { FILE *f = fopen("/dev/urandom", "rb"); bool got = f && fread(ws.w, 1, 32, f) == 32; if (f) fclose(f);
          if (!got) return fail(ROFL_HIP_ERROR, "no randomness for the batched Sigma-proof check"); }
        niels *pts = C.gbuf[0].as<niels>(nslots * d);
        sc *scal = C.SL.as<sc>(nslots * d);
        sc *d_fixed = C.tmp_out.as<sc>(nblk * 2);
        hipLaunchKernelGGL(k_sigma_vprep, dim3((unsigned)nblk), dim3(TPB), 0, C.stream, kind, (u32)d, dp, dc, sigma_init_state(kind), ws, pts, scal, d_fixed, status);
        std::vector<sc> h_fixed(nblk * 2);
        u32 st0 = 0;
        HIPCHK(hipMemcpyAsync(h_fixed.data(), d_fixed, sizeof(sc) * nblk * 2, hipMemcpyDeviceToHost, C.stream));
        HIPCHK(hipMemcpyAsync(&st0, status, 4, hipMemcpyDeviceToHost, C.stream));
        C.sync();
        if (st0 & 4u) { timing_end(C); return fail(ROFL_FORMAT_ERROR, "FormatError: non-canonical scalar or invalid point"); }
        std::vector<MsmProb> pr(1, MsmProb{pts, scal}); std::vector<ge5> res;
        msm_run(C, pr, nslots * d, res);
        sc sB = h_canon(sum_partials(h_fixed.data(), nblk, 2, 0)), sBb = h_canon(sum_partials(h_fixed.data(), nblk, 2, 1));
        ge5 tot = h51::gadd(res[0], h51::gadd(h_fixed_mul(C.ht.B5, sB), h_fixed_mul(C.ht.Bb5, sBb)));
        timing_end(C);
        *ok_out = h51::is_identity_ristretto(tot) ? 1 : 0;
        return ROFL_OK;
    }
    hipLaunchKernelGGL(k_sigma_verify, dim3((unsigned)((d + 63) / 64)), dim3(64), 0, C.stream, kind, (u32)d, dp, dc, sigma_init_state(kind), C.d_tabB, C.d_tabBb, status + 1, status);
    u32 st[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(st, status, 8, hipMemcpyDeviceToHost, C.stream));
    C.sync();
    timing_end(C);
    if (st[0] & 4u) return fail(ROFL_FORMAT_ERROR, "FormatError: non-canonical scalar or invalid point");
    *ok_out = st[1] == 0;
    return ROFL_OK;
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
    LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
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
    if (d) {
        HIPCHK(hipMemcpyAsync(dv, values, 4 * d, hipMemcpyHostToDevice, C.stream));
        HIPCHK(hipMemcpyAsync(dr, r32, 32 * d, hipMemcpyHostToDevice, C.stream));
        if (dex) HIPCHK(hipMemcpyAsync(dex, existing, 32 * d, hipMemcpyHostToDevice, C.stream));
        hipLaunchKernelGGL(k_eg_pairs, dim3((unsigned)((d + 63) / 64)), dim3(64), 0, C.stream, (u32)d, dv, fp_bits, fp_frac, dr, dex, C.d_tabB, C.d_tabBb, dpairs, status);
        HIPCHK(hipMemcpyAsync(pairs_out, dpairs, 64 * d, hipMemcpyDeviceToHost, C.stream));
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
    sc c = compressed_challenge(pairs_out, d, proof_out);
    sc zm = nc[0], zr = nc[1];
    if (d) {
        CPow cp; fill_pow2(cp.sq, h_mont(c), MAX_LG);
        u32 nblk = (u32)std::min<size_t>(64, (d + TPB - 1) / TPB);
        sc *part = C.tmp_out.as<sc>(64 * 2);
        hipLaunchKernelGGL(k_cpow_dot, dim3(nblk), dim3(TPB), 0, C.stream, (u32)d, dv, fp_bits, fp_frac, dr, cp, part);
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
    LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
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
        HIPCHK(hipMemcpyAsync(dpairs, pairs, 64 * d, hipMemcpyHostToDevice, C.stream));
        hipLaunchKernelGGL(k_decode_pairs, grid1(2 * d), dim3(TPB), 0, C.stream, (u32)d, dpairs, pts, pts + d, status);
        CPow cp; fill_pow2(cp.sq, h_mont(c), MAX_LG);
        hipLaunchKernelGGL(k_cpow_scalars, grid1(d), dim3(TPB), 0, C.stream, (u32)d, cp, scal);
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

int rofl_create_squareproof_vec(const float *values, size_t d, const uint8_t *r1_32, size_t d_r1, const uint8_t *r2_32, const uint8_t *existing32,
                                unsigned fp_bits, unsigned fp_frac, const rofl_nonce_t *nonce, uint8_t *proofs_out, uint8_t *commits_out) {
    return guarded([&]() -> int { return sigma_create(2, values, d, r1_32, d_r1, r2_32, existing32, fp_bits, fp_frac, nonce, proofs_out, commits_out); });
}
int rofl_verify_squareproof_vec(const uint8_t *proofs, const uint8_t *commits, size_t d, int *ok_out) {
    return guarded([&]() -> int { return sigma_verify(2, proofs, commits, d, ok_out); });
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
    return guarded([&]() -> int { return sigma_create(0, values, d, r32, d_r, nullptr, existing32, fp_bits, fp_frac, nonce, proofs_out, commits_out); });
}
int rofl_verify_randproof_vec(const uint8_t *proofs, const uint8_t *commits, size_t d, int *ok_out) {
    return guarded([&]() -> int { return sigma_verify(0, proofs, commits, d, ok_out); });
}
int rofl_create_squarerandproof_vec(const float *values, size_t d, const uint8_t *r1_32, size_t d_r1, const uint8_t *r2_32, const uint8_t *existing32,
                                    unsigned fp_bits, unsigned fp_frac, const rofl_nonce_t *nonce, uint8_t *proofs_out, uint8_t *commits_out) {
    return guarded([&]() -> int { return sigma_create(1, values, d, r1_32, d_r1, r2_32, existing32, fp_bits, fp_frac, nonce, proofs_out, commits_out); });
}
int rofl_verify_squarerandproof_vec(const uint8_t *proofs, const uint8_t *commits, size_t d, int *ok_out) {
    return guarded([&]() -> int { return sigma_verify(1, proofs, commits, d, ok_out); });
}

int rofl_commit_vec(const uint8_t *values32, const uint8_t *blindings32, size_t d, uint8_t *out32) {
    return guarded([&]() -> int {
        LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
        if (d == 0) return ROFL_OK;
        C.init();
        sc *dv = C.tmp_in.as<sc>(d); sc *db = blindings32 ? C.tmp_in2.as<sc>(d) : nullptr;
        uint8_t *o = C.Cbytes.as<uint8_t>(d * 32);
        HIPCHK(hipMemcpyAsync(dv, values32, 32 * d, hipMemcpyHostToDevice, C.stream));
        if (db) HIPCHK(hipMemcpyAsync(db, blindings32, 32 * d, hipMemcpyHostToDevice, C.stream));
        hipLaunchKernelGGL(k_commit, grid1(d), dim3(TPB), 0, C.stream, (u32)d, (const u64 *)nullptr, dv, db, C.d_tabB, C.d_tabBb, (const niels *)nullptr, (uint8_t *)nullptr, o, (u32)d, (u32)d);
        HIPCHK(hipMemcpyAsync(out32, o, 32 * d, hipMemcpyDeviceToHost, C.stream));
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
        HIPCHK(hipMemcpyAsync(da, a32, 32 * d, hipMemcpyHostToDevice, C.stream));
        HIPCHK(hipMemcpyAsync(db, b32, 32 * d, hipMemcpyHostToDevice, C.stream));
        hipLaunchKernelGGL(k_add_points, grid1(d), dim3(TPB), 0, C.stream, (u32)d, da, db, o, status);
        u32 st = 0;
        HIPCHK(hipMemcpyAsync(out32, o, 32 * d, hipMemcpyDeviceToHost, C.stream));
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
        HIPCHK(hipMemcpyAsync(da, points, stride * d, hipMemcpyHostToDevice, C.stream));
        hipLaunchKernelGGL(k_decode_sum, dim3(nblk), dim3(TPB), TPB * sizeof(ge), C.stream, da, (u32)d, (u32)stride, part, status);
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
        HIPCHK(hipMemcpyAsync(da, a32, 32 * d, hipMemcpyHostToDevice, C.stream));
        hipLaunchKernelGGL(k_decode, grid1(d), dim3(TPB), 0, C.stream, (u32)d, (u32)d, da, ds, (niels *)nullptr, o, status);
        u32 st = 0;
        HIPCHK(hipMemcpyAsync(out32, o, 32 * d, hipMemcpyDeviceToHost, C.stream));
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
        HIPCHK(hipMemcpyAsync(dp, points32, n * 32, hipMemcpyHostToDevice, C.stream));
        HIPCHK(hipMemcpyAsync(ds, hs.data(), n * 32, hipMemcpyHostToDevice, C.stream));
        hipLaunchKernelGGL(k_decode, grid1(n), dim3(TPB), 0, C.stream, (u32)n, (u32)n, dp, (const niels *)nullptr, dn, (uint8_t *)nullptr, status);
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
int rofl_dbg_quad_ops(const uint8_t *pairs64, size_t pairs, unsigned doublings, uint8_t *out_serial32, uint8_t *out_quad32) {
    return guarded([&]() -> int {
        LaneLock lane_lock = acquire_lane(); Ctx &C = *lane_lock.c;
        if (pairs == 0) return fail(ROFL_BAD_PARAM, "bad parameter");
        C.init();
        uint8_t *din = C.tmp_in.as<uint8_t>(pairs * 64), *dout = C.tmp_out.as<uint8_t>(pairs * 64);
        u32 *status = C.status.as<u32>(4);
        HIPCHK(hipMemsetAsync(status, 0, 16, C.stream));
        HIPCHK(hipMemsetAsync(dout, 0, pairs * 64, C.stream));
        HIPCHK(hipMemcpyAsync(din, pairs64, pairs * 64, hipMemcpyHostToDevice, C.stream));
        hipLaunchKernelGGL(k_dbg_quad, grid1(pairs * 4), dim3(TPB), 0, C.stream, (u32)pairs, doublings, (const uint8_t *)din, dout, dout + pairs * 32, status);
        u32 st = 0;
        HIPCHK(hipMemcpyAsync(&st, status, 4, hipMemcpyDeviceToHost, C.stream));
        HIPCHK(hipMemcpyAsync(out_serial32, dout, pairs * 32, hipMemcpyDeviceToHost, C.stream));
        HIPCHK(hipMemcpyAsync(out_quad32, dout + pairs * 32, pairs * 32, hipMemcpyDeviceToHost, C.stream));
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
            hipLaunchKernelGGL(k_bsgs_build, dim3((unsigned)((table_size + 1 + 63) / 64)), dim3(64), 0, C.stream, (u32)table_size, C.d_tabB, b.keys, b.slots, b.mask);
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
        HIPCHK(hipMemcpyAsync(dp, points32, 32 * d, hipMemcpyHostToDevice, C.stream));
        hipLaunchKernelGGL(k_bsgs_solve, dim3((unsigned)((d + 63) / 64)), dim3(64), 0, C.stream, (u32)d, dp, (u32)table_size, bsgs_bits, max_it, neg_mG, B.keys, B.slots, B.mask, dout, status);
        u32 st = 0;
        HIPCHK(hipMemcpyAsync(scalars_out32, dout, 32 * d, hipMemcpyDeviceToHost, C.stream));
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
namespace {
int *option_slot(Ctx &P, const char *key, long *lo, long *hi) {
    struct { const char *k; int Ctx::*f; long lo, hi; } tab[] = {
        {"verify_zip_truncate", &Ctx::opt_zip_truncate, 0, 1}, {"verify_batch", &Ctx::opt_verify_batch, 0, 1},
        {"sigma_batch", &Ctx::opt_sigma_batch, 0, 1}, {"blocking_sync", &Ctx::blocking_sync, -1, 1}};
    for (auto &t : tab) if (key && !strcmp(key, t.k)) { *lo = t.lo; *hi = t.hi; return &(P.*(t.f)); }
    return nullptr;
}
}  // namespace
int rofl_set_option(const char *key, long value) {
    return guarded([&]() -> int {
        Ctx &P = ctx(); { std::lock_guard<std::mutex> g(P.init_mu); P.init(); }
        long lo, hi; int *slot = option_slot(P, key, &lo, &hi);
        if (!slot || value < lo || value > hi) return fail(ROFL_BAD_PARAM, "unknown option or value out of range");
        *slot = (int)value; return ROFL_OK;
    });
}
int rofl_get_option(const char *key, long *value_out) {
    return guarded([&]() -> int {
        Ctx &P = ctx(); { std::lock_guard<std::mutex> g(P.init_mu); P.init(); }
        long lo, hi; int *slot = option_slot(P, key, &lo, &hi);
        if (!slot || !value_out) return fail(ROFL_BAD_PARAM, "unknown option");
        *value_out = *slot; return ROFL_OK;
    });
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
                if (mode == 3) hipLaunchKernelGGL(k_bench_madd_gather, dim3(blocks), dim3(TPB), fl, C.stream, it, (u32)entries, dt, dg);
                else if (mode == 2) hipLaunchKernelGGL(k_bench_madd_l1, dim3(blocks), dim3(TPB), fl, C.stream, it, (u32)entries, dt, dg);
                else hipLaunchKernelGGL(k_bench_madd_regs, dim3(blocks), dim3(TPB), fl, C.stream, it, (u32)entries, dt, dg);
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
        hipLaunchKernelGGL(k_bench_femul, dim3(blocks), dim3(TPB), fl, C.stream, 8u, din, dout);
        HIPCHK(hipEventRecord(e0, C.stream));
        hipLaunchKernelGGL(k_bench_femul, dim3(blocks), dim3(TPB), fl, C.stream, iters, din, dout);
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
    const u64 dom[2] = {0x2f6b7a2d6c666f72ULL, 0x31762f65636e6f6eULL};
    u64 sd[4]; memcpy(sd, seed, 32); u64 st[25]; shake256_seeded_block(st, dom, sd, idx);
    sc lo, hi; for (int i = 0; i < 4; i++) { lo.v[2 * i] = (u32)st[i]; lo.v[2 * i + 1] = (u32)(st[i] >> 32); hi.v[2 * i] = (u32)st[4 + i]; hi.v[2 * i + 1] = (u32)(st[4 + i] >> 32); }
    sc_tobytes(out, sc_from_wide(lo, hi)); return 0;
}

}  // extern "C"
