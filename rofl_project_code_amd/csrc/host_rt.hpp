// Host runtime of librofl_zk.so: tuning-knob registry, error plumbing, host scalar / point helpers, the per-lane thread pool, device and
// pinned buffers, timing, the lane context (Ctx) and the generator-table cache.  Included by rofl_zk.hip inside its anonymous namespace.
#pragma once

// ---------------------------------------------------------------- tuning knobs
// Every ROFL_* environment variable the library reads, in ONE table (scripts/gen_knob_table.py turns it into the table of KNOBS.md and
// tests/test_host_lib.py checks that no other name is read).  Knobs never change results -- proofs, commitments and verdicts are the same
// for every setting (the behaviour switches of rofl_set_option are the exception and are marked "option") -- they move work between
// variants, and most of them exist because an experiment in profiles/r0*_experiments.txt needed them.  Read once per process (or per device context).
struct Knob { const char *name, *dflt, *what; };
static const Knob KNOBS[] = {
    {"ROFL_LANES", "3", "calls that can be in flight on a device (HIP stream + workspace each); 1..16, larger values are clamped (with a note on stderr)"},
    {"ROFL_DEVICES", "0", "option devices: bit mask of the logical devices the batch entry points shard their clients over (0 = the calling thread's device only)"},
    {"ROFL_DEVICE_MAP", "", "logical -> physical HIP device, comma separated (0,0 = two logical devices on GPU 0: how the multi-device paths are tested on a one-GPU box); default: identity"},
    {"ROFL_RCCL_LIB", "librccl.so.1", "the RCCL library rofl_comm_* loads with dlopen (a host that has torch in the process names /opt/rocm/lib/librccl.so.1: the bare soname would resolve to torch's bundled copy)"},
    {"ROFL_HOST_THREADS", "usable cores, in [2, 16]", "threads of the primary lane's host pool (window combination, encodings, transcripts of a round); other lanes: usable cores, in [6, 12]"},
    {"ROFL_POOL_SPIN_US", "400", "how long an idle pool worker polls for the next job before it sleeps (0 = sleep at once)"},
    {"ROFL_BLOCKING_SYNC", "-1", "option blocking_sync: -1 spin while <= 3 calls are in flight, 0 always spin, 1 sleep between polls"},
    {"ROFL_VERIFY_ZIP_TRUNCATE", "0", "option verify_zip_truncate: 1 = the reference's zip-truncating verify_rangeproof"},
    {"ROFL_VERIFY_BATCH", "1", "option verify_batch: 0 = one check per proof, 1 = one per client, 2 = one per batch of clients (bisecting down to per-client checks on failure)"},
    {"ROFL_SIGMA_BATCH", "1", "option sigma_batch: 0 = one check per element in the Sigma-proof verifiers"},
    {"ROFL_STAGE_COHERENT", "0", "1 = the staging arenas are coherent (hipHostMallocDefault) pinned memory as in rounds 3-4 instead of non-coherent (the DMA engine reads 32 instead of 57 GB/s out of it)"},
    {"ROFL_STAGE_KEEP_MB", "256", "pinned staging memory a lane keeps between calls (a call that needed more frees it when it ends)"},
    {"ROFL_MSM_WINDOW_ORDER", "1", "0 = the bucket lists of the fixed-base two-level sort stay in arrival order and the accumulation's blocks in (problem, array) order (default: lists ordered by window slot, blocks array-major, so that the gathers of a launch stay within a few window slices of the table at a time)"},
    {"ROFL_SIGMA_SPLIT", "1", "0 = the per-element Sigma-proof prover runs as one thread per element (k_sigma_prove) instead of one thread per output point + a finishing kernel (k_sigma_points, k_sigma_finish); same bytes"},
    {"ROFL_MSM_BIN_TILE_SEARCH", "1", "0 = the coarse-bin pass of the two-level bucket sort keeps its power-of-two tile also when another tile length fills the CUs in fewer, shorter passes (batched rounds)"},
    {"ROFL_HEAVY_LO", "0", "1 = while several calls share the device, the chip-filling launches (bucket accumulation, generator folds) go to a lowest-priority stream of their lane so that other calls' short kernels are dispatched between their blocks (measured: neutral for rounds of range proofs, slower for rounds of L2 updates -- profiles/r06_experiments.txt item 2; off)"},
    {"ROFL_GENS_LAZY", "1", "0 = the first call of a shape waits for its full fold table (otherwise it is served from the compact table while a background thread builds the full one)"},
    {"ROFL_GENS_LAZY_IDLE_MS", "20", "the background build of a full fold table starts when no call has been in flight for this long (its allocation stalls every HIP call of the process)"},
    {"ROFL_GENS_LAZY_MAX_WAIT_MS", "3000", "... or after this long, whichever comes first (a host that never pauses still gets its full table)"},
    {"ROFL_GENS_RESERVE_MB", "40960", "HBM a table allocation of 4 GB or more leaves free on the device (other contexts / processes / workspaces; the runtime aborts when HBM runs out): the table is narrowed instead"},
    {"ROFL_GENS_BUDGET_MB", "131072", "HBM budget of the generator-table cache (LRU eviction of unpinned entries beyond it)"},
    {"ROFL_FOLD_T1", "3", "IPP rounds before the first generator fold (1..6)"},
    {"ROFL_FOLD_T", "2", "IPP rounds between later folds (1..6)"},
    {"ROFL_FOLD_MIN", "1024", "no fold once fewer generators per chunk would remain (launches with many chunks fold down to 64)"},
    {"ROFL_MSM_FB_FITSETS", "1", "0 = do not add bucket sets to a fixed-base launch whose coarse bins would not fit the two-level sort"},
    {"ROFL_MSM_BIN_SIGMA", "8", "room of a coarse bin of the two-level sort above twice its mean load, in standard deviations of that load (0 = the fixed 256 entries of rounds 2-3)"},
    {"ROFL_POOL_NAP", "1", "0 = pool workers do not nap through a wait whose length the caller announced (they poll for ROFL_POOL_SPIN_US, then sleep until woken)"},
    {"ROFL_SYNC_POLL", "0", "1 = wait for the lane's stream with hipStreamQuery in a pause loop instead of hipStreamSynchronize"},
    {"ROFL_FOLD_TAB", "1", "0 = first fold without the precomputed odd-multiple slices"},
    {"ROFL_FOLD_PB", "32", "piece width of the fold table in bits (16, 32, 64)"},
    {"ROFL_FOLD_W", "10", "NAF width of the fold table (3..10: 2^(w-2) odd multiples per 32-bit piece, 1/(w+1) of the digits non-zero; narrowed until the table fits ROFL_FOLD_TAB_MB)"},
    {"ROFL_FOLD_TAB_MB", "106496", "HBM budget of one (n, m) fold table (cfg 2 at width 10 and cfg 4 at width 9: 102.4 GB of the 288; 57344 = one width less, half the memory, +0.25 ms per cfg-2 proof)"},
    {"ROFL_FOLD_WNAF", "0", "later folds: NAF width over odd multiples of the sources built on a side stream during the preceding rounds (3..6; 0 = plain NAF over the sources alone: the build disturbs the rounds it runs beside by what the shorter fold chains save -- measured neutral, off by default)"},
    {"ROFL_FOLD_TAB_EV", "1", "0 = the first (table) fold scans digit arrays (k_fold_gens_tab) instead of walking an event list with operand prefetch (k_fold_gens_w)"},
    {"ROFL_FOLD_UNIT", "1", "0 = do not keep the common factor s_0 of a fold in gscale / hscale"},
    {"ROFL_FOLD_K", "0", "digit-position segments per fold output (1, 2, 4; 0 = by launch size)"},
    {"ROFL_FOLD_THREADS", "131072", "fold launches with fewer threads split their chains into segments"},
    {"ROFL_FOLD_DEFER", "1", "0 = the first fold converts every output to affine form itself (one inversion per output) instead of leaving that to k_niels_batch on the side stream"},
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
    {"ROFL_MSM_SMALL_GROUP", "4", "windows of a problem per block of the fused launch at c = 7 with more than 512 bucket arrays (1 = one window per one-wave block, 2, 4)"},
    {"ROFL_MERLIN_X8", "1", "0 = the verifier hashes every chunk's transcript prefix on its own even when the host has AVX-512 (eight chunks per instruction stream otherwise, from 32 chunks on)"},
    {"ROFL_MSM_HOST8", "1", "0 = launches with many problems combine their windows on the device (k_msm_horner) even when the host has AVX-512 IFMA"},
    {"ROFL_MSM_HOST8_MIN", "8", "launches with at least this many problems run their window chains eight per AVX-512 IFMA stream on the host (when the CPU has it)"},
    {"ROFL_MSM_FB_HOST8_MIN", "8", "fixed-base launches with at least this many problems finish eight problems per AVX-512 IFMA task (32 = as in rounds 3-4: one scalar chain per pool task below 32 problems)"},
    {"ROFL_KECCAK_ZMM", "", "host transcripts: 1 = Keccak-f[1600] with one state across five AVX-512 registers, 0 = the scalar rounds; unset = whichever a 50 us measurement at first use finds faster (Intel: the registers, 1.27x; Zen 5: scalar)"},
    {"ROFL_LR_FIRST", "1", "0 = l(x), r(x), the first round's MSM scalars and its inner products in three launches (k_lr_vec, k_ipp_scalars, k_ipp_inner) instead of one (k_lr_first)"},
    {"ROFL_HOP_CQ", "1", "0 = the <a,b> w B terms of a round are computed on the hop (after the wait) instead of on the pool while the round's MSM runs"},
    {"ROFL_MSM_DEV_HORNER_MIN", "32", "launches with at least this many problems combine their windows on the device"},
    {"ROFL_MSM_T13", "8192", "generic MSMs from this many terms on use 13-bit windows"},
    {"ROFL_MSM_T10", "512 / 2048", "generic MSMs from this many terms on use 10-bit windows (7-bit below, 4-bit below 64); default 2048 for launches with >= 32 problems"},
    {"ROFL_MSM_C", "0", "window width of every generic MSM (4, 7, 10, 13, 16; 0 = by size)"},
    {"ROFL_MSM_GROUP_REDUCE", "0", "1 = two-launch bucket reduction by groups of 512 (measured slower)"},
    {"ROFL_RED_SPLIT", "0", "1 = four threads per 8-group in k_msm_reduce_level (measured slower)"},
    {"ROFL_RED_FUSED_T", "768", "largest block of k_msm_reduce_fused"},
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
static const bool g_keccak_knob_applied = [] { if (const char *e = knob("ROFL_KECCAK_ZMM")) if (*e) keccak_zmm_flag() = atoi(e) != 0 && __builtin_cpu_supports("avx512f"); return true; }();

// ---------------------------------------------------------------- error plumbing
thread_local std::string g_err;
int fail(int code, const std::string &msg) { g_err = msg; return code; }
struct HipErr { hipError_t e; const char *what; };
#define HIPCHK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) throw HipErr{e__, #x}; } while (0)
// Every kernel launch is checked where it is made: a launch the runtime refuses (a grid dimension past 65 535, too much LDS) would otherwise
// leave its outputs as the previous call left them, and a verifier would read a verdict out of stale data.  hipGetLastError is a thread-local
// read; whether the runtime's last-error slot is sticky or overwritten by the next successful call, the launch is the last call made here.
// (The slot is emptied first: every other runtime call of the library is checked where it is made, so what an earlier one left there --
// a hipErrorNotReady from a query, where the runtime records those -- is not this launch's.)
#define ROFL_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); hipError_t e__ = hipGetLastError(); \
                              if (e__ != hipSuccess && e__ != hipErrorNotReady) throw HipErr{e__, "kernel launch"}; } while (0)

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---------------------------------------------------------------- host scalar helpers (canonical <-> Montgomery)
sc h_mont(const sc &canon) { return sc_to_mont(canon); }
sc h_canon(const sc &mont) { return sc_from_mont(mont); }
sc h_mul(const sc &a, const sc &b) { return sc_mul_plain(a, b); }          // canonical * canonical
sc h_inv(const sc &canon) { return h_canon(h51::sc_invert_mont_fast(h_mont(canon))); }
bool sc_is_canonical_bytes(const uint8_t *b) { sc s = sc_frombytes(b); return !sc_geq_l(s.v); }

// width-w NAF (3 <= w <= 8) of a canonical scalar: odd digits |d| < 2^(w-1), at most one non-zero among w consecutive positions, density
// 1 / (w + 1); returns the index of the highest non-zero digit (-1 if zero)
int sc_wnaf(int8_t out[256], const sc &k, unsigned w) {
    u32 x[9]; for (int i = 0; i < 8; i++) x[i] = k.v[i]; x[8] = 0;
    const int full = 1 << w, half = 1 << (w - 1);
    int top = -1;
    for (int pos = 0; pos < 256; pos++) {
        int d = 0;
        if (x[0] & 1) {
            d = (int)(x[0] & (u32)(full - 1)); if (d >= half) d -= full;
            if (d > 0) { x[0] -= (u32)d; }        // (the low w bits are d: no borrow)
            else { u64 c = (u64)(-d); for (int i = 0; i < 9 && c; i++) { c += x[i]; x[i] = (u32)c; c >>= 32; } }
            top = pos;
        }
        out[pos] = (int8_t)d;
        for (int i = 0; i < 8; i++) x[i] = (x[i] >> 1) | (x[i + 1] << 31);
        x[8] >>= 1;
    }
    return top;
}
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
int wnaf_u64(int16_t out[FOLD_TAB_DIGITS], u64 piece, unsigned w) {
    unsigned __int128 k = piece; int top = -1;
    const int full = 1 << w, half = 1 << (w - 1);
    for (int pos = 0; pos < FOLD_TAB_DIGITS; pos++) {
        int d = 0;
        if (k & 1) { d = (int)(k & (unsigned)(full - 1)); if (d >= half) d -= full; if (d >= 0) k -= (unsigned)d; else k += (unsigned)(-d); top = pos; }
        out[pos] = (int16_t)d; k >>= 1;
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
    std::vector<std::thread> workers; std::mutex mu;
    std::function<void(size_t)> fn; std::atomic<size_t> count{0}, next{0}, done{0}; std::atomic<int> active{0}; std::atomic<uint64_t> gen{0}; std::atomic<bool> stop{false};
    std::unique_ptr<std::atomic<uint32_t>[]> asleep;      // asleep[w] == 1: worker w waits on this word (futex)
    void wake(int w) { if (asleep[w].exchange(0, std::memory_order_seq_cst) == 1) syscall(SYS_futex, reinterpret_cast<uint32_t *>(&asleep[w]), FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0); }
    double spin_us = 400.0;      // ROFL_POOL_SPIN_US: how long an idle worker polls for the next job before it sleeps
    std::atomic<uint64_t> hint_seq{0}; std::atomic<double> gap_us{0.0};      // expect_gap(): the caller's estimate of its coming wait for the device
    const std::atomic<int> *calls_in_flight = nullptr;      // polling is for a call that is alone on the device: with several in flight the pools of the lanes would fight over the cores
    // Only as many workers poll as recent jobs had tasks for (the "hot" set: worker w polls while w < hot); the others sleep on their own
    // futex words and are woken one by one when a job has a task for them.  A four-chunk client never has more than eight tasks in a job -- sixteen pollers burnt
    // 6.5 cores for a proof that keeps three busy.
    std::atomic<int> hot{0}; int recent_max = 0, recent_runs = 0;
    // An index is claimed only after it has been checked against `count` (compare-and-swap, not fetch-add): a worker that wakes
    // late and arrives while run() is resetting the job (count == 0 in that window) must not consume an index of the next job --
    // with a blind fetch-add it could take index 0 between `next = 0` and `count = n` and drop it, and run() would wait forever.
    // A task that throws (the tasks allocate) must not take the process down from a worker thread, nor leave run() waiting for a `done`
    // that never comes: the first exception is kept, the index still counts as done, and run() rethrows on the caller's thread, where
    // guarded() turns it into an error code.
    std::exception_ptr first_error; std::mutex err_mu;
    void work() {
        struct Active { std::atomic<int> &a; explicit Active(std::atomic<int> &x) : a(x) { a.fetch_add(1); } ~Active() { a.fetch_sub(1); } } guard(active);
        for (;;) {
            size_t i = next.load();
            if (i >= count.load()) break;
            if (!next.compare_exchange_weak(i, i + 1)) continue;
            try { fn(i); } catch (...) { std::lock_guard<std::mutex> lk(err_mu); if (!first_error) first_error = std::current_exception(); }
            done.fetch_add(1);
        }
    }
    // A worker that has just finished a job polls for the next one for a short while before it sleeps: the hops of a proof follow each
    // other at 0.1-0.3 ms, and a sleeping thread has to be put back on a CPU by the scheduler first -- on a busy host (the GPU boxes run
    // at a load average above 20) that wake-up is where multi-millisecond outliers of a 25 ms proof came from.
    void loop(int my_index) {
        uint64_t seen = 0, my_hint = 0;
        for (;;) {
            if (spin_us > 0 && my_index < hot.load(std::memory_order_relaxed) && (!calls_in_flight || calls_in_flight->load(std::memory_order_relaxed) <= 1)) {
                auto t0 = std::chrono::steady_clock::now();
                uint64_t g;
                while ((g = gen.load(std::memory_order_acquire)) == seen) {
                    __builtin_ia32_pause();
                    // the caller is about to wait ~gap_us for the device (expect_gap): sleep through most of it and be polling again when the
                    // next job arrives -- no wake-up has to come from the caller's thread (fifteen futex wake-ups on the hop, and on a loaded
                    // host the woken threads landing on the waker's CPU, were the 1.5-2 ms hops of the slow steps)
                    uint64_t hs = hint_seq.load(std::memory_order_acquire);
                    if (hs != my_hint) {
                        my_hint = hs;
                        double gap = gap_us.load(std::memory_order_relaxed);
                        if (gap > 700.0) {
                            double ns = std::min(gap - 350.0, 6000.0) * 1e3;
                            struct timespec ts = {0, (long)ns}; nanosleep(&ts, nullptr);
                            t0 = std::chrono::steady_clock::now();
                            continue;
                        }
                    }
                    if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > spin_us) break;
                }
                // a job seen while polling is taken without the mutex (fifteen pollers queueing for it cost the hop tens of microseconds): run()
                // publishes fn / count / next before it bumps gen, and the acquire load above orders this thread's reads after that
                // (a hint is for the wait it was announced before: one that arrived while this thread was busy or asleep is over by now)
                if (g != seen) { seen = g; my_hint = hint_seq.load(std::memory_order_acquire); work(); continue; }
            }
            // sleep on this worker's own futex word: run() wakes exactly the workers its job has tasks for (w < n), nobody else
            for (;;) {
                asleep[my_index].store(1, std::memory_order_seq_cst);
                if (stop.load(std::memory_order_seq_cst) || gen.load(std::memory_order_seq_cst) != seen) { asleep[my_index].store(0, std::memory_order_relaxed); break; }
                syscall(SYS_futex, reinterpret_cast<uint32_t *>(&asleep[my_index]), FUTEX_WAIT_PRIVATE, 1u, nullptr, nullptr, 0);
                if (asleep[my_index].load(std::memory_order_acquire) == 0) break;      // woken for a job (or for stop); spurious wake-ups go round again
            }
            if (stop.load()) return;
            seen = gen.load(std::memory_order_acquire);
            my_hint = hint_seq.load(std::memory_order_acquire);
            work();
        }
    }
    // The waits of run() last microseconds: pause-spin.  (sched_yield hands the CPU to whatever else is runnable on it -- on a host that other
    // tenants load, that is a full time slice of theirs: milliseconds in the middle of a 0.2 ms hop.)  Only a wait that drags on -- a worker
    // that lost its CPU in mid-task -- starts yielding.
    template <class Pred> static void spin_until(Pred ready) {
        for (unsigned i = 0; !ready(); i++) {
            if (i < 20000) __builtin_ia32_pause(); else std::this_thread::yield();
        }
    }
public:
    explicit HostPool(int nthreads, const std::atomic<int> *in_flight = nullptr) : calls_in_flight(in_flight) {
        if (const char *e = knob("ROFL_POOL_SPIN_US")) spin_us = atof(e);
        if (const char *e = knob("ROFL_POOL_NAP")) nap = atoi(e) != 0;
        asleep.reset(new std::atomic<uint32_t>[nthreads > 0 ? nthreads : 1]);
        for (int i = 0; i < nthreads; i++) asleep[i].store(0);
        for (int i = 1; i < nthreads; i++) workers.emplace_back([this, i] { loop(i); });
    }
    ~HostPool() { stop.store(true, std::memory_order_seq_cst); for (size_t w = 1; w <= workers.size(); w++) wake((int)w); for (auto &t : workers) t.join(); }
    // called right before the caller starts a wait it expects to last `us` microseconds (0 = unknown): polling workers nap through it
    void expect_gap(double us) { if (!nap) return; gap_us.store(us, std::memory_order_relaxed); hint_seq.fetch_add(1, std::memory_order_release); }
    bool nap = true;      // ROFL_POOL_NAP=0: workers only poll / sleep on the condition variable, as before
    void run(size_t n, std::function<void(size_t)> f) {
        if (n <= 1 || workers.empty()) { for (size_t i = 0; i < n; i++) f(i); return; }
        spin_until([&] { return active.load() == 0; });     // no straggler of the previous job may still look at fn
        {   // the hot set follows the largest job of the last 64 (the caller is executor 0: a job of n tasks wants n - 1 workers)
            int want = (int)std::min<size_t>(n, workers.size() + 1);
            recent_max = std::max(recent_max, want);
            if (want > hot.load(std::memory_order_relaxed)) hot.store(want, std::memory_order_relaxed);
            if (++recent_runs >= 64) { hot.store(recent_max, std::memory_order_relaxed); recent_max = 0; recent_runs = 0; }
        }
        { std::lock_guard<std::mutex> lk(mu); count = 0; fn = std::move(f); done = 0; next = 0; first_error = nullptr; count = n; gen.fetch_add(1, std::memory_order_seq_cst); }
        for (size_t w = 1; w < n && w <= workers.size(); w++) wake((int)w);      // (polling and napping workers have asleep == 0: one atomic exchange each)
        work();
        spin_until([&] { return done.load() >= n; });
        if (first_error) { std::exception_ptr e; { std::lock_guard<std::mutex> lk(err_mu); e = first_error; first_error = nullptr; } std::rethrow_exception(e); }
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

// Caller memory never goes to the HIP runtime directly.  A hipMemcpyAsync on ordinary (pageable) host memory makes the runtime pin the
// caller's pages for the DMA -- and when the caller frees such a buffer later, the driver's MMU notifier tears the mapping down inside
// munmap.  Both showed on the path: the first call after a process had released its previous results waited 20-30 ms in a 0.5 ms
// rofl_commit_vec, and a caller that kept its results alive (fresh pages for every output) ran every L2 composite at 40 ms instead of 12
// (profiles/r03_experiments.txt item 26).  Large transfers are staged through the lane's own pinned memory instead: one CPU copy
// (~10 GB/s) on the way in, one on the way out after the call's last synchronisation.  Device pointers pass through untouched.
struct Stage {
    struct Chunk { void *p = nullptr; size_t cap = 0, used = 0; };
    struct Pending { void *user; const void *stage; size_t n; };
    std::vector<Chunk> chunks; std::vector<Pending> pend; bool dirty = false;
    static constexpr size_t kMin = 32 << 10;      // below this the runtime's own bounce buffers do the same job
    // The arena is only ever the source or destination of copies (no kernel reads it): non-coherent pinned memory.  The DMA engine moves
    // 57 GB/s out of it against 32 GB/s out of the default (coherent, fine-grained) flavour on this platform (scripts/stage_bw.hip).
    static unsigned flags() { static const unsigned f = (knob("ROFL_STAGE_COHERENT") && atoi(knob("ROFL_STAGE_COHERENT")) != 0) ? hipHostMallocDefault : hipHostMallocNonCoherent; return f; }
    void *alloc(size_t n) {
        n = (n + 255) & ~(size_t)255; dirty = true;
        for (auto &c : chunks) if (c.cap - c.used >= n) { void *r = (char *)c.p + c.used; c.used += n; return r; }
        Chunk c; c.cap = std::max<size_t>(n, (size_t)4 << 20);      // never moves or frees a chunk that copies in flight may still use
        HIPCHK(hipHostMalloc(&c.p, c.cap, flags()));
        c.used = n; chunks.push_back(c); return c.p;
    }
    // after the call's last synchronisation: hand the results over, recycle the arena (several chunks -> one of their total size next time)
    // The arena is trimmed to ROFL_STAGE_KEEP_MB afterwards: one large batched call (dozens of clients' commitments) must not leave
    // gigabytes of pinned host memory allocated for the life of the process, on every lane.
    void finish(bool deliver) {
        if (deliver) for (auto &q : pend) memcpy(q.user, q.stage, q.n);
        pend.clear();
        static const size_t keep = (knob("ROFL_STAGE_KEEP_MB") ? (size_t)std::max(1L, atol(knob("ROFL_STAGE_KEEP_MB"))) : (size_t)256) << 20;
        size_t tot = 0; for (auto &c : chunks) tot += c.cap;
        if (tot > keep || chunks.size() > 1) {
            for (auto &c : chunks) (void)hipHostFree(c.p);
            chunks.clear();
            if (tot <= keep) { Chunk c; c.cap = tot; if (hipHostMalloc(&c.p, c.cap, flags()) == hipSuccess) chunks.push_back(c); }
        }
        for (auto &c : chunks) c.used = 0;
        dirty = false;
    }
};
// The staging copy of a large input: caller memory -> pinned staging, read once by the DMA engine afterwards.  Ordinary stores would first
// fetch every destination line (read-for-ownership): three units of memory traffic per byte copied instead of two, and the copy of a
// round's 760 MB of Sigma-proofs is bound by exactly that.  Non-temporal stores from 64 KB on (AVX2, checked at run time).
__attribute__((target("avx2"))) inline void stream_copy_avx2(uint8_t *dst, const uint8_t *src, size_t n) {
    size_t head = (32 - ((uintptr_t)dst & 31)) & 31; if (head > n) head = n;
    memcpy(dst, src, head); dst += head; src += head; n -= head;
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        __m256i a = _mm256_loadu_si256((const __m256i *)(src + i)), b = _mm256_loadu_si256((const __m256i *)(src + i + 32));
        __m256i c = _mm256_loadu_si256((const __m256i *)(src + i + 64)), e = _mm256_loadu_si256((const __m256i *)(src + i + 96));
        _mm256_stream_si256((__m256i *)(dst + i), a); _mm256_stream_si256((__m256i *)(dst + i + 32), b);
        _mm256_stream_si256((__m256i *)(dst + i + 64), c); _mm256_stream_si256((__m256i *)(dst + i + 96), e);
    }
    _mm_sfence();
    memcpy(dst + i, src + i, n - i);
}
inline void stage_copy(void *dst, const void *src, size_t n) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && n >= ((size_t)64 << 10)) stream_copy_avx2((uint8_t *)dst, (const uint8_t *)src, n); else memcpy(dst, src, n);
}
inline bool is_device_ptr(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }      // unregistered host memory
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

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
// The window tables are owned by reference count: the fast-start swap hands the SAME tables to the entry that replaces the compact one, and
// calls that still read the retired entry keep them alive -- whoever drops the last reference frees them (an eviction of the new entry while
// a retired reader was in mid-proof used to free them under it).
struct WTabs { ndm *wtab = nullptr, *wtab_many = nullptr; size_t bytes = 0;
               ~WTabs() { if (wtab) (void)hipFree(wtab); if (wtab_many) (void)hipFree(wtab_many); } };
struct GensEntry { niels *tbl = nullptr; std::shared_ptr<WTabs> wt; ndm *wtab = nullptr, *wtab_many = nullptr; u32 wc = 16, wc_many = 0; FoldTabCfg fc{}; size_t bytes = 0, n = 0, m = 0; u64 tick = 0; int users = 0;
                   bool retired = false;      // a replaced fast-start entry: still read by the calls that pinned it, no longer in the cache map
                   int full_state = 0;        // 0 = this is the shape's final table; 1 = the full fold table is still to come; -1 = its build failed (HBM short): compact for good unless prepare retries
                   bool has_fold = true;      // false: built for a VERIFIER -- generators + window slices only (the verifier never folds); a prover that meets it builds the fold table first
                   bool fold_building = false; };      // a prover is adding the fold table to a verifier's entry right now (others wait for the swap)
// Who asks for a shape's tables.  The prover reads everything; the verifier reads the generators (slice 0 of `tbl`) and the window slices of the
// fixed-base MSM, never the fold slices -- 102 GB of the 104 at BASELINE cfg 2 / cfg 4.  The reference's server (rofl_service/src/flserver/server.rs:656-687)
// only ever verifies: its process must not build, hold or evict for a table it never reads.
enum GensRole { GENS_PROVE = 0, GENS_VERIFY = 1 };

// Behaviour options (rofl_set_option): process-wide, so that a server that drives several devices sets them once.  The environment
// only provides the defaults, read when the first option is touched.
struct Options {
    std::atomic<int> zip_truncate{0}, verify_batch{1}, sigma_batch{1}, blocking_sync{-1};
    std::atomic<long> devices{0};      // bit d = logical device d takes a share of the batch entry points' clients
};
Options &opts() {
    static Options o;
    static std::once_flag once;
    std::call_once(once, [] {
        if (const char *e = knob("ROFL_BLOCKING_SYNC")) o.blocking_sync = atoi(e) != 0;
        if (const char *e = knob("ROFL_VERIFY_ZIP_TRUNCATE")) o.zip_truncate = atoi(e) != 0;
        if (const char *e = knob("ROFL_VERIFY_BATCH")) { int v = atoi(e); o.verify_batch = v < 0 ? 0 : v > 2 ? 2 : v; }
        if (const char *e = knob("ROFL_SIGMA_BATCH")) o.sigma_batch = atoi(e) != 0;
        if (const char *e = knob("ROFL_DEVICES")) { long v = strtol(e, nullptr, 0); if (v >= 0) o.devices = v; }
    });
    return o;
}

// Calls in flight in the whole process.  The pools' polling is for a call that is alone: with several in flight -- on one device or, in a
// server that drives N devices, on several -- the pools of the lanes would fight over the host's cores.
std::atomic<int> g_calls_in_flight{0};
std::atomic<bool> g_gens_shutdown{false};      // process exit: background table builders that are still waiting for a quiet moment give up

// One Ctx = one "lane": a HIP stream with its own workspace, staging buffers, timing and host pool.  The primary
// lane of a device owns the shared read-only state (fixed-base tables, generator cache).  An API call runs on one
// lane; concurrent calls from different host threads (the reference's server verifies clients from a thread pool,
// server.rs:656-687) take different lanes, so the latency-bound phases of one call (small rounds, host Horner,
// transcripts) overlap the throughput-bound phases of another.  ROFL_LANES = size of the pool.
struct Ctx {
    int device = 0, phys = 0;      // logical device (the key of the context table) and the HIP device behind it (ROFL_DEVICE_MAP / rofl_dbg_map_device)
    bool inited = false;
    Stage stg;      // staging of caller memory for this lane's current call (see Stage)
    // caller memory (host or device) -> device
    void up(void *dst_dev, const void *src, size_t n, hipStream_t st) {
        if (n < Stage::kMin || is_device_ptr(src)) { HIPCHK(hipMemcpyAsync(dst_dev, src, n, hipMemcpyDefault, st)); return; }
        void *s = stg.alloc(n); memcpy(s, src, n);
        HIPCHK(hipMemcpyAsync(dst_dev, s, n, hipMemcpyHostToDevice, st));
    }
    // device -> caller memory (host): delivered by Stage::finish when the lane is released
    // Returns where the bytes are readable once `st` has been synchronised (the staging copy, or dst_user itself for a small transfer).
    const void *down(void *dst_user, const void *src_dev, size_t n, hipStream_t st) {
        if (n < Stage::kMin) { HIPCHK(hipMemcpyAsync(dst_user, src_dev, n, hipMemcpyDeviceToHost, st)); return dst_user; }
        void *s = stg.alloc(n);
        HIPCHK(hipMemcpyAsync(s, src_dev, n, hipMemcpyDeviceToHost, st));
        stg.pend.push_back({dst_user, s, n});
        return s;
    }
    std::vector<hipEvent_t> ev_pool;      // numbered events of the current call (pipelined input groups)
    hipEvent_t pool_event(size_t k) {
        while (ev_pool.size() <= k) { hipEvent_t e; HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); ev_pool.push_back(e); }
        return ev_pool[k];
    }
    Ctx *parent = nullptr;
    std::vector<Ctx *> sibs;      // additional lanes
    static constexpr int kMaxLanes = 16;
    int nlanes = 3;      // ROFL_LANES: number of calls that can be in flight on this device (1..kMaxLanes)
    std::mutex init_mu, gens_mu;      // primary lane only: one-time initialisation; generator-table cache
    std::atomic<int> active_calls{0};  // primary lane only: calls currently holding a lane
    std::atomic<unsigned> rr{0};
    hipStream_t stream = nullptr, stream2 = nullptr;      // stream2: side stream for work that may run beside the main one (created on first use)
    hipStream_t stream3 = nullptr;                        // the odd-multiple tables of the later folds are built here, beside the rounds that precede the fold
    hipStream_t stream_up = nullptr;                      // uploads of a pipelined batch call: the copy of group g + 1 runs beside the kernels of group g (created on first use)
    hipStream_t stream_lo = nullptr;                      // LOWEST priority: the VALU-saturating launches (bucket accumulation, generator folds) of calls that share the device (HeavyScope)
    std::vector<hipEvent_t> heavy_ev; size_t heavy_k = 0;  // ring of fork / join events of the heavy launches
    std::mutex mu;
    HostTables ht;
    niels *d_tabB = nullptr, *d_tabBb = nullptr, *d_tabB8 = nullptr, *d_tabBb8 = nullptr;      // radix-16 (64 x 8) and radix-256 (32 x 128) fixed-base tables of B and B~
    sc *d_two_pow = nullptr;
    std::map<std::pair<size_t, size_t>, std::unique_ptr<GensEntry>> gens;   // (n, m) -> tables; primary lane only, under gens_mu
    std::vector<std::unique_ptr<GensEntry>> gens_retired;                  // fast-start entries that were replaced while calls still read them (freed when unpinned)
    std::vector<std::thread> gens_upgrades;                                // background builders of the full fold tables (joined at exit)
    std::map<std::pair<size_t, size_t>, int> gens_pending;                 // shapes whose full table is still being built (rofl_bp_gens_prepare waits for them)
    u64 gens_tick = 0; size_t gens_budget = (size_t)128 << 30;   // ROFL_GENS_BUDGET_MB: evict least recently used (unpinned) tables beyond this
    u32 fold_pb = 32, fold_w = 10; size_t fold_tab_budget = (size_t)104 << 30;   // widest NAF whose table fits the per-(n, m) budget (ROFL_FOLD_W, ROFL_FOLD_TAB_MB)
    int msm_lds = 1, msm_two_level = 1, msm_group_reduce = 0; size_t msm_lds_min = 8192, msm_lds_tile = 131072;
    size_t msm_fb_threads = (size_t)1 << 19;
    int msm_fb = 1; size_t msm_fb_min = (size_t)1 << 12; int msm_lr = 1;   // window tables for every (n, m) with 2N >= 4096: many small chunks (n_partition = 64) share them
    bool crowded() const { const Ctx *P = parent ? parent : this; return P->active_calls.load() > 1; }   // other calls in flight on this device
    // Waiting for the lane's stream.  hipStreamSynchronize spins (lowest latency: right for a call that is alone on the device); with
    // more than three calls in flight -- or when the host asked for it (ROFL_BLOCKING_SYNC=1) -- the thread sleeps between queries instead, so
    // a server that keeps several clients in flight does not burn one host core per client on busy-waiting (ROFL_BLOCKING_SYNC=0: always spin).
    hipEvent_t ev_block = nullptr, ev_v = nullptr, ev_fork = nullptr, ev_a = nullptr, ev_a0 = nullptr, ev_m2 = nullptr, ev_m2j = nullptr, ev_norm = nullptr, ev_norm0 = nullptr, ev_ip = nullptr, ev_mult = nullptr, ev_mult0 = nullptr; bool batch_mode = false;
    void sync() {
        const Ctx *P = parent ? parent : this;
        // (up to three calls in flight still spin: the three proofs of ONE client's L2 update run side by side -- EncParamsL2::encrypt --
        //  and that is a latency case; a server with more clients in flight is a throughput case)
        const int bs = opts().blocking_sync.load(std::memory_order_relaxed);
        bool block = bs == 1 || (bs < 0 && (P->active_calls.load() > 3 || batch_mode));
        if (!block) {
            static const bool poll = knob("ROFL_SYNC_POLL") && atoi(knob("ROFL_SYNC_POLL")) != 0;
            if (!poll) { HIPCHK(hipStreamSynchronize(stream)); return; }
            for (;;) {      // hipStreamQuery in a pause loop: never gives the CPU away
                hipError_t q = hipStreamQuery(stream);
                if (q == hipSuccess) return;
                if (q != hipErrorNotReady) throw HipErr{q, "hipStreamQuery"};
                for (int i = 0; i < 16; i++) __builtin_ia32_pause();
            }
        }
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
    // the same policy for an event of this call (the pipelined input groups of the batch verifier): a server that verifies many batches side by
    // side must not spin one host core per waiting call here either
    void wait_event(hipEvent_t ev) {
        const Ctx *P = parent ? parent : this;
        const int bs = opts().blocking_sync.load(std::memory_order_relaxed);
        if (!(bs == 1 || (bs < 0 && (P->active_calls.load() > 3 || batch_mode)))) { HIPCHK(hipEventSynchronize(ev)); return; }
        for (;;) {
            hipError_t q = hipEventQuery(ev);
            if (q == hipSuccess) return;
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
    int fold_t = 2, fold_t1 = 3, fold_k = 0, fold_tab = 1, fold_unit = 1, fold_wnaf = 0; long fold_threads = 131072;
    Timing tm;
    struct HopStats { double enqueue = 0, sync = 0, horner_wall = 0, horner_cpu = 0, host_wall = 0, host_cpu = 0, max_enqueue = 0, max_sync = 0, max_horner = 0, max_task = 0; int n = 0; } hs;      // where the host hops of the current call go (ROFL_TRACE, rofl_dbg_last_hops)
    // workspace
    DevBuf cp, sL, sR, party, Scanon, vshift, blind, Vbytes, Cbytes, status, partial, partial2, scpart, a, b, a2, b2, ptab[2], yinv,
        SL, SR, powtabs, foldprobs, naf,
        gbuf[2], aux_pts, aux_scal, vscal, tmp_in, tmp_in2, tmp_out, vals, uni, stream_buf, vgroups, vtabs, ipdev, qpts, vspart, foldext, fmul_ext, fmul_tab, fold_ev;
    PinBuf h_cp, h_part, h_misc, h_misc2, h_auxc, h_auxs, h_V, h_ip, h_round, h_fdig, h_fprob, h_abfin, h_vgrp, h_q, h_fev;
    MsmWs mws[2];
    std::map<uint64_t, double> wait_ms;      // how long the wait of a tagged hop took the last times (hint for the pool workers' naps)

    void init() {
        if (inited) return;
        HIPCHK(hipSetDevice(phys));
        HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        // The side stream is created WITH the main stream, not on first use: the runtime deals streams to its hardware queues in creation order
        // (GPU_MAX_HW_QUEUES, 4 by default), and a side stream created after the other lanes' main streams landed on its own lane's queue for some
        // lane counts -- the commitment kernel then ran behind the nonce expansion instead of beside it: 0.3-0.6 ms per step of a lone client at every
        // lane count on one box, 0.6-0.9 ms at 2 and 6 lanes on another (profiles/r06_experiments.txt item 17).
        HIPCHK(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&stream_up, hipStreamNonBlocking));      // (the upload stream of the pipelined batch verifiers: same reason)
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
        HIPCHK(hipMalloc(&d_tabB8, sizeof(niels) * 4096));
        HIPCHK(hipMalloc(&d_tabBb8, sizeof(niels) * 4096));
        ROFL_LAUNCH(k_fixed_tab8, dim3(64), dim3(64), 0, 0, (const niels *)d_tabB, d_tabB8);
        ROFL_LAUNCH(k_fixed_tab8, dim3(64), dim3(64), 0, 0, (const niels *)d_tabBb, d_tabBb8);
        HIPCHK(hipStreamSynchronize(nullptr));      // (the null stream only: another context's lanes on this device -- non-blocking streams -- are not waited for)
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
            pool.reset(new HostPool(nt, &g_calls_in_flight)); }
        if (const char *e = knob("ROFL_FOLD_T1")) { int v = atoi(e); if (v >= 1 && v <= 6) fold_t1 = v; }
        if (const char *e = knob("ROFL_FOLD_TAB")) fold_tab = atoi(e) != 0;
        if (const char *e = knob("ROFL_GENS_BUDGET_MB")) { long v = atol(e); if (v >= 1) gens_budget = (size_t)v << 20; }
        if (const char *e = knob("ROFL_FOLD_PB")) { int v = atoi(e); if (v == 16 || v == 32 || v == 64) fold_pb = (u32)v; }
        if (const char *e = knob("ROFL_FOLD_W")) { int v = atoi(e); if (v >= 3 && v <= 10) fold_w = (u32)v; }
        if (const char *e = knob("ROFL_FOLD_TAB_MB")) { long v = atol(e); if (v >= 1) fold_tab_budget = (size_t)v << 20; }
        if (const char *e = knob("ROFL_FOLD_UNIT")) fold_unit = atoi(e) != 0;
        if (const char *e = knob("ROFL_FOLD_WNAF")) { int v = atoi(e); if (v == 0 || (v >= 3 && v <= 6)) fold_wnaf = v; }
        if (const char *e = knob("ROFL_FOLD_K")) { int v = atoi(e); if (v == 1 || v == 2 || v == 4) fold_k = v; }
        if (const char *e = knob("ROFL_FOLD_THREADS")) { long v = atol(e); if (v > 0) fold_threads = v; }
        if (const char *e = knob("ROFL_LANES")) {      // out-of-range values are clamped, not ignored: the caller sized its thread pool by them
            int v = atoi(e), w = v < 1 ? 1 : v > kMaxLanes ? kMaxLanes : v;
            if (w != v) fprintf(stderr, "librofl_zk: ROFL_LANES=%d is outside 1..%d, using %d\n", v, kMaxLanes, w);
            nlanes = w;
        }
        inited = true;
        for (int i = 1; i < nlanes; i++) { Ctx *s = new Ctx(); s->init_lane(*this); sibs.push_back(s); }
    }
    void init_lane(Ctx &p) {
        parent = &p; device = p.device; phys = p.phys;
        HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));      // (with the main stream: see init)
        HIPCHK(hipStreamCreateWithFlags(&stream_up, hipStreamNonBlocking));
        ht = p.ht; d_tabB = p.d_tabB; d_tabBb = p.d_tabBb; d_tabB8 = p.d_tabB8; d_tabBb8 = p.d_tabBb8; d_two_pow = p.d_two_pow;
        msm_lds = p.msm_lds; msm_two_level = p.msm_two_level; msm_group_reduce = p.msm_group_reduce; msm_lds_min = p.msm_lds_min; msm_lds_tile = p.msm_lds_tile;
        msm_fb_threads = p.msm_fb_threads;
        msm_fb = p.msm_fb; msm_fb_min = p.msm_fb_min; msm_lr = p.msm_lr;
        fold_min = p.fold_min; msm_dev_horner_min = p.msm_dev_horner_min; msm_small_max = p.msm_small_max; msm_slots = p.msm_slots; fold_t = p.fold_t; fold_t1 = p.fold_t1; fold_k = p.fold_k; fold_tab = p.fold_tab;
        fold_unit = p.fold_unit; fold_wnaf = p.fold_wnaf; fold_threads = p.fold_threads; nlanes = 1;
        // (the other lanes: up to twelve -- a batched server call that lands on a sibling lane stages hundreds of MB through its pool, and which of two
        //  concurrent batch calls finds the primary lane free is a race; idle workers sleep on their futex words and cost nothing)
        { int nt = std::min(12, std::max(6, usable_cores())); if (const char *e = knob("ROFL_HOST_THREADS")) nt = atoi(e); if (nt < 1) nt = 1; pool.reset(new HostPool(nt, &g_calls_in_flight)); }
        inited = true;
    }
};

// NB the HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and reads the variable once, when
// it initialises; lanes that share a queue serialise each other's kernels.  The library does not touch the process environment:
// a host that wants more than four calls in flight exports GPU_MAX_HW_QUEUES itself before the first HIP call.

// Devices.  One context (primary lane + siblings, generator cache) per LOGICAL device, created on first use.  Which device a call runs on
// is a property of the calling thread: rofl_set_device binds the thread (as hipSetDevice does) and also becomes the process default that
// threads without a binding of their own follow -- so a one-device host keeps its "set once, call from any thread" behaviour, and a
// server process that drives N GPUs (the reference's server is ONE process with a verification pool, server.rs:379-384, 656-687) binds
// each pool thread to its device, or lists the devices in rofl_set_option("devices", mask) and lets the batch entry points shard.
// Logical device i is HIP device g_devmap[i] (identity unless ROFL_DEVICE_MAP / rofl_dbg_map_device say otherwise).
constexpr int kMaxDevices = 64;
std::mutex g_ctx_mu;
std::map<int, Ctx *> g_ctxs;
std::atomic<int> g_default_device{0};
std::atomic<bool> g_default_set{false};      // the first successful rofl_set_device (or option default_device) has chosen the default
thread_local int t_device = -1;
int g_devmap[kMaxDevices];
std::once_flag g_devmap_once;
void devmap_init() {
    std::call_once(g_devmap_once, [] {
        for (int i = 0; i < kMaxDevices; i++) g_devmap[i] = i;
        if (const char *e = knob("ROFL_DEVICE_MAP")) { int i = 0; for (const char *p = e; *p && i < kMaxDevices; i++) { g_devmap[i] = atoi(p); while (*p && *p != ',') p++; if (*p == ',') p++; } }
    });
}
int current_device() { return t_device >= 0 ? t_device : g_default_device.load(std::memory_order_relaxed); }
Ctx &ctx_of(int dev) {
    devmap_init();
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    auto it = g_ctxs.find(dev);
    if (it == g_ctxs.end()) { Ctx *c = new Ctx(); c->device = dev; c->phys = g_devmap[dev]; it = g_ctxs.emplace(dev, c).first; }
    return *it->second;
}
Ctx &ctx() { return ctx_of(current_device()); }
// binds the calling thread for the lifetime of the object (the worker threads of a sharded batch call)
// EXPERIMENT (ROFL_HEAVY_LO=1; off): the launches that fill the chip for milliseconds -- k_msm_accumulate_*, k_fold_gens* -- at the LOWEST
// stream priority while several calls share the device (batched rounds, calls in flight on other lanes).  The idea: at equal priority a foreign
// lane's sort / reduce / scalar kernels (no field arithmetic, a few hundred blocks) get their workgroups dispatched only as the heavy launch
// drains -- profiles/r06_cfg4_timeline_inflight3.txt: k_msm_bin_l2 1.6 ms alone, 23 ms beside another lane's accumulation; k_msm_reduce_level
// 0.7 -> 11.5 ms -- so the short kernels should cut in as blocks retire.  Measured (profiles/r06_experiments.txt item 2): a cfg-4 round 1 368 /
// 1 364 ms against 1 370 / 1 360 (neutral: with three calls in flight the device is saturated either way -- the stretched kernels wait, the
// chip does not), a cfg-5 round 810 + 153 ms against 540 + 64 (the Sigma-proof legs' Pippenger launches wait behind everything).  Kept as a knob.
struct HeavyScope {
    Ctx &C; hipStream_t st, run; bool forked = false;
    static bool enabled() { static const bool on = knob("ROFL_HEAVY_LO") && atoi(knob("ROFL_HEAVY_LO")) != 0; return on; }
    HeavyScope(Ctx &c, hipStream_t s, bool want = true) : C(c), st(s), run(s) {      // want = false: a launch too short to be worth two events
        if (!want || !enabled() || !(C.batch_mode || C.crowded())) return;
        if (!C.stream_lo) {
            int least = 0, greatest = 0;
            if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); return; }
            if (hipStreamCreateWithPriority(&C.stream_lo, hipStreamNonBlocking, least) != hipSuccess) { (void)hipGetLastError(); C.stream_lo = nullptr; return; }
        }
        while (C.heavy_ev.size() < 16) { hipEvent_t e; HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); C.heavy_ev.push_back(e); }
        hipEvent_t e0 = C.heavy_ev[(C.heavy_k++) & 15];
        HIPCHK(hipEventRecord(e0, st)); HIPCHK(hipStreamWaitEvent(C.stream_lo, e0, 0));
        run = C.stream_lo; forked = true;
    }
    void end() {
        if (!forked) return;
        forked = false;
        hipEvent_t e1 = C.heavy_ev[(C.heavy_k++) & 15];
        HIPCHK(hipEventRecord(e1, run)); HIPCHK(hipStreamWaitEvent(st, e1, 0));
    }
    ~HeavyScope() { if (forked) { hipEvent_t e1 = C.heavy_ev[(C.heavy_k++) & 15]; (void)hipEventRecord(e1, run); (void)hipStreamWaitEvent(st, e1, 0); } }
    HeavyScope(const HeavyScope &) = delete; HeavyScope &operator=(const HeavyScope &) = delete;
};

struct DeviceBinding { int saved; explicit DeviceBinding(int dev) : saved(t_device) { t_device = dev; } ~DeviceBinding() { t_device = saved; } };

// A lane held for the duration of one API call.
struct LaneLock {
    Ctx *c = nullptr; Ctx *primary = nullptr; std::unique_lock<std::mutex> lk;
    LaneLock() = default;
    LaneLock(LaneLock &&o) noexcept : c(o.c), primary(o.primary), lk(std::move(o.lk)) { o.c = nullptr; o.primary = nullptr; }
    // the call is over (every API function synchronises its streams before it returns or unwinds): results staged for the caller are
    // copied out unless an exception is in flight, and the staging arena is recycled
    ~LaneLock() {
        if (c && (c->stg.dirty || !c->stg.pend.empty())) {
            // (near no-ops on the normal path; an error path may leave before its waits: nothing may still read or write the arena)
            if (c->stream) (void)hipStreamSynchronize(c->stream);
            if (c->stream2) (void)hipStreamSynchronize(c->stream2);
            c->stg.finish(std::uncaught_exceptions() == 0);
        }
        if (primary) { primary->active_calls.fetch_sub(1); g_calls_in_flight.fetch_sub(1); }
    }
};
// `side`: a call that hardly uses the host pool (one-value sum proofs, the per-element Sigma-proof kernels) looks for a free sibling lane
// first and leaves the primary lane -- the one with the big pool -- to a range proof over a whole update.  The three proofs of an L2
// composite start on three threads at once; which of them found the primary lane free used to be a race, and a range proof that lost it
// ran its 19 hops on a sibling's six threads (one client in six at 28 ms instead of 20).
LaneLock acquire_lane(bool primary_only = false, bool side = false) {
    Ctx &P = ctx();
    { std::lock_guard<std::mutex> g(P.init_mu); P.init(); }
    HIPCHK(hipSetDevice(P.phys));                      // the calling thread may be new to HIP, or last used another device
    LaneLock ll; ll.primary = &P; P.active_calls.fetch_add(1); g_calls_in_flight.fetch_add(1);
    size_t L = primary_only ? 1 : 1 + P.sibs.size();
    for (size_t k = 0; k < L; k++) {
        size_t i = side && L > 1 ? (k + 1) % L : k;      // side calls: siblings first, the primary lane last
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
    e->wt.reset();                                // the window tables go with their last owner (WTabs)
    e->wtab_many = nullptr;
    if (e->tbl) (void)hipFree(e->tbl);
    e->wtab = nullptr; e->tbl = nullptr;
}
// caller holds gens_mu.  Frees unpinned entries, least recently used first, until `keep_bytes` or less are held.
void gens_evict(Ctx &P0, size_t keep_bytes, const GensEntry *spare) {
    for (;;) {
        size_t total = 0; for (auto &kv : P0.gens) total += kv.second->bytes;
        for (auto &r : P0.gens_retired) total += r->bytes;      // (held until their readers return: they count, they cannot be chosen)
        if (total <= keep_bytes) return;
        auto victim = P0.gens.end();
        for (auto it = P0.gens.begin(); it != P0.gens.end(); ++it)
            if (it->second.get() != spare && it->second->users == 0 && (victim == P0.gens.end() || it->second->tick < victim->second->tick)) victim = it;
        if (victim == P0.gens.end()) return;       // everything left is in use
        gens_free_entry(victim->second.get());
        P0.gens.erase(victim);
    }
}
// Big tables never take the device below a reserve (ROFL_GENS_RESERVE_MB): other contexts and processes on the same GPU, the lanes' workspaces
// and the runtime's own queues and scratch allocate later, and the runtime does not fail when HBM is gone -- it ABORTS the process
// (HSA_STATUS_ERROR_OUT_OF_RESOURCES: seen with two contexts holding 100 GB fold tables each and a third process starting).  The caller
// narrows the table and tries again.
hipError_t gens_malloc(Ctx &P0, void **p, size_t bytes, const GensEntry *spare) {
    if (bytes >= ((size_t)4 << 30)) {
        static const size_t reserve = (size_t)(knob("ROFL_GENS_RESERVE_MB") ? atol(knob("ROFL_GENS_RESERVE_MB")) : 40960) << 20;
        auto fits = [&] { size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return true; } return fr >= bytes + reserve; };
        if (!fits()) { gens_evict(P0, 0, spare); if (!fits()) return hipErrorOutOfMemory; }
    }
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
    void release() {
        if (e) {
            std::lock_guard<std::mutex> lk(P0->gens_mu); e->users--;
            for (size_t k = 0; k < P0->gens_retired.size();)      // a replaced fast-start entry goes when its last reader has gone
                if (P0->gens_retired[k]->users == 0) { gens_free_entry(P0->gens_retired[k].get()); P0->gens_retired.erase(P0->gens_retired.begin() + (long)k); } else k++;
        }
        e = nullptr;
    }
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
// The full fold table of a shape that was started on the compact one (fast start).  Runs on a background thread (wait_quiet: after the
// first quiet moment) or, when an earlier build failed for want of HBM, synchronously from rofl_bp_gens_prepare.  The caller has pinned
// `raw` (users) and registered `key` in gens_pending; both are released here.  Returns whether the shape now has its full table.
bool gens_upgrade(Ctx &P0, GensEntry *raw, std::pair<size_t, size_t> key, FoldTabCfg fc_full, size_t N, bool wait_quiet, const FoldTabCfg *then_full = nullptr) {
    // then_full: `fc_full` is only the compact layout (a prover met a verifier's entry and wants to start at once); the entry that is swapped in
    // is registered for a background build of *then_full, exactly like a fresh fast-start entry
    auto give_up = [&](int state) { std::lock_guard<std::mutex> lk(P0.gens_mu); raw->users--; raw->full_state = state; P0.gens_pending.erase(key);
                                    for (size_t k = 0; k < P0.gens_retired.size();)
                                        if (P0.gens_retired[k]->users == 0) { gens_free_entry(P0.gens_retired[k].get()); P0.gens_retired.erase(P0.gens_retired.begin() + (long)k); } else k++; };
    // Wait for a quiet moment: allocating tens of GB holds the runtime's lock for 0.05-0.7 s (every HIP call of the process waits) and the
    // table kernel then fills the device for ~70 ms -- inside the first call that is the whole gain of the compact table gone again.
    // A client proves once per training round and a server verifies once per round: pauses are plentiful.  rofl_bp_gens_prepare hurries it.
    if (wait_quiet) {
        static const double idle_ms = knob("ROFL_GENS_LAZY_IDLE_MS") ? atof(knob("ROFL_GENS_LAZY_IDLE_MS")) : 20.0;
        static const double max_ms = knob("ROFL_GENS_LAZY_MAX_WAIT_MS") ? atof(knob("ROFL_GENS_LAZY_MAX_WAIT_MS")) : 3000.0;
        const double w0 = now_ms(); double quiet_since = -1;
        for (;;) {
            if (g_gens_shutdown.load()) { give_up(1); return false; }
            { std::lock_guard<std::mutex> lk(P0.gens_mu); auto it = P0.gens_pending.find(key); if (it != P0.gens_pending.end() && it->second == 2) break; }      // hurried
            const double t = now_ms();
            if (t - w0 >= max_ms) break;
            if (g_calls_in_flight.load() == 0) { if (quiet_since < 0) quiet_since = t; if (t - quiet_since >= idle_ms) break; } else quiet_since = -1;
            struct timespec ts = {0, 1000000}; nanosleep(&ts, nullptr);
        }
    }
    if (hipSetDevice(P0.phys) != hipSuccess) { (void)hipGetLastError(); give_up(-1); return false; }
    size_t bytes = sizeof(niels) * 2 * N * fc_full.np * fc_full.e;
    void *tv = nullptr; hipStream_t bs = nullptr;
    const bool btrace = knob("ROFL_TRACE") && atoi(knob("ROFL_TRACE")) >= 2; const double bt0 = now_ms();
    {   // the allocation goes through the cache's own allocator: room is made within the budget first (unpinned entries, least recently used),
        // and a failed hipMalloc evicts everything unpinned before it gives up -- the same rules as every other table
        std::lock_guard<std::mutex> lk(P0.gens_mu);
        gens_evict(P0, P0.gens_budget > bytes ? P0.gens_budget - bytes : 0, raw);
        if (gens_malloc(P0, &tv, bytes, raw) != hipSuccess) tv = nullptr;
        while (!tv && fc_full.pb != 64 && fc_full.w > 7) {      // HBM is short: one NAF width less = half the table
            fc_full.w--; fc_full.e = 1u << (fc_full.w - 2);
            bytes = sizeof(niels) * 2 * N * fc_full.np * fc_full.e;
            if (gens_malloc(P0, &tv, bytes, raw) != hipSuccess) tv = nullptr;
        }
    }
    if (!tv) { give_up(-1); return false; }      // HBM is short: the compact table stays; rofl_bp_gens_prepare reports it / tries again
    if (btrace) fprintf(stderr, "[rofl-trace gens-upgrade] hipMalloc of %.1f GB +%.3f ms\n", bytes / 1e9, now_ms() - bt0);
    // the builder's one large launch runs at the LOWEST stream priority: the proofs that are served from the compact table meanwhile keep
    // getting their workgroups dispatched (at equal priority the first create of a fresh process waited ~65 ms behind it)
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    bool ok = hipStreamCreateWithPriority(&bs, hipStreamNonBlocking, prio_least) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); bs = nullptr; ok = hipStreamCreateWithFlags(&bs, hipStreamNonBlocking) == hipSuccess; }
    ok = ok && hipMemcpyAsync(tv, raw->tbl, sizeof(niels) * 2 * N, hipMemcpyDeviceToDevice, bs) == hipSuccess;
    if (ok) { hipLaunchKernelGGL(k_gens_tables, grid1(2 * N * fc_full.np), dim3(TPB), 0, bs, (u32)(2 * N), fc_full, reinterpret_cast<niels *>(tv), (size_t)(2 * N)); ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(bs) == hipSuccess; }
    if (bs) (void)hipStreamDestroy(bs);
    if (btrace) fprintf(stderr, "[rofl-trace gens-upgrade] table built at +%.3f ms\n", now_ms() - bt0);
    std::lock_guard<std::mutex> lk(P0.gens_mu);
    raw->users--; P0.gens_pending.erase(key);
    auto it = P0.gens.find(key);
    if (!ok || it == P0.gens.end() || it->second.get() != raw) {      // (evicted meanwhile, or the build failed)
        (void)hipFree(tv); raw->full_state = ok ? raw->full_state : -1;
        for (size_t k = 0; k < P0.gens_retired.size();)
            if (P0.gens_retired[k]->users == 0) { gens_free_entry(P0.gens_retired[k].get()); P0.gens_retired.erase(P0.gens_retired.begin() + (long)k); } else k++;
        return false;
    }
    std::unique_ptr<GensEntry> full(new GensEntry(*raw));      // shares the window tables (WTabs) with the entry it replaces
    full->tbl = reinterpret_cast<niels *>(tv); full->fc = fc_full; full->users = 0; full->tick = ++P0.gens_tick; full->full_state = 0;
    full->has_fold = true; full->fold_building = false;
    full->bytes = raw->bytes - sizeof(niels) * 2 * N * raw->fc.np * raw->fc.e + bytes;
    std::unique_ptr<GensEntry> old = std::move(it->second);
    it->second = std::move(full);
    old->retired = true; old->bytes = sizeof(niels) * 2 * N * old->fc.np * old->fc.e;      // what it still holds alone: its compact fold table
    if (old->users == 0) gens_free_entry(old.get()); else P0.gens_retired.push_back(std::move(old));
    gens_evict(P0, P0.gens_budget, it->second.get());
    if (then_full) {
        GensEntry *nraw = it->second.get();
        const FoldTabCfg fcf = *then_full;
        nraw->users++; nraw->full_state = 1; P0.gens_pending[key] = 1;
        Ctx *pp = &P0;
        P0.gens_upgrades.emplace_back([pp, nraw, key, fcf, N] { gens_upgrade(*pp, nraw, key, fcf, N, /*wait_quiet=*/true); });
    }
    return true;
}
// the fold-table layout a prover wants for N = n * m generators per side: the widest NAF whose table fits the per-shape budget
FoldTabCfg gens_full_cfg(const Ctx &P0, size_t N) {
    FoldTabCfg fc{P0.fold_pb, P0.fold_w, 256 / P0.fold_pb, 1u << (P0.fold_w - 2)};
    // HBM capacity for VALU work: a width-w NAF needs 2^(w-2) odd multiples per piece and leaves 1/(w+1) of the digits non-zero
    while (fc.w > 6 && sizeof(niels) * 2 * N * fc.np * fc.e > P0.fold_tab_budget) { fc.w--; fc.e = 1u << (fc.w - 2); }
    if (sizeof(niels) * 2 * N * fc.np * fc.e > std::max(P0.fold_tab_budget, (size_t)40 << 30)) fc = FoldTabCfg{64, 4, 4, 4};     // very large tables: the compact layout
    return fc;
}
bool gens_lazy_for(const FoldTabCfg &fc, size_t N) {
    static const bool lazy_on = !(knob("ROFL_GENS_LAZY") && atoi(knob("ROFL_GENS_LAZY")) == 0);
    return lazy_on && !(fc.pb == 64 && fc.w == 4) && sizeof(niels) * 2 * N * fc.np * fc.e >= ((size_t)4 << 30);
}
GensPin get_gens(Ctx &C, size_t n, size_t m, GensRole role = GENS_PROVE) {
    Ctx &P0 = C.parent ? *C.parent : C;
    auto key = std::make_pair(n, m);
    size_t N = n * m;
    for (;;) {      // (a prover that meets a verifier's entry adds the fold table and looks again)
        std::unique_lock<std::mutex> find_lock(P0.gens_mu);
        auto it = P0.gens.find(key);
        if (it == P0.gens.end()) break;      // (the lock is released here; the builder below takes it again -- see there)
        GensEntry *e = it->second.get();
        if (role == GENS_VERIFY || e->has_fold) { e->tick = ++P0.gens_tick; e->users++; return GensPin(&P0, e); }
        if (e->fold_building) { find_lock.unlock(); struct timespec ts = {0, 200000}; nanosleep(&ts, nullptr); continue; }      // another prover is at it
        e->fold_building = true; e->users++; P0.gens_pending[key] = 2;
        find_lock.unlock();
        const FoldTabCfg fc_full = gens_full_cfg(P0, N);
        const bool lazy = gens_lazy_for(fc_full, N);
        const FoldTabCfg fc_first = lazy ? FoldTabCfg{64, 4, 4, 4} : fc_full;
        if (!gens_upgrade(P0, e, key, fc_first, N, /*wait_quiet=*/false, lazy ? &fc_full : nullptr)) {
            { std::lock_guard<std::mutex> lk(P0.gens_mu); auto it2 = P0.gens.find(key); if (it2 != P0.gens.end() && it2->second.get() == e) e->fold_building = false; }
            throw HipErr{hipErrorOutOfMemory, "hipMalloc(fold table of a shape first used by a verifier)"};
        }
    }
    std::lock_guard<std::mutex> gens_lock(P0.gens_mu);
    {   // (between the two locks another thread may have built the shape)
        auto it = P0.gens.find(key);
        if (it != P0.gens.end() && (role == GENS_VERIFY || it->second->has_fold)) { it->second->tick = ++P0.gens_tick; it->second->users++; return GensPin(&P0, it->second.get()); }
        if (it != P0.gens.end()) throw HipErr{hipErrorNotReady, "generator tables changed hands while they were being looked up"};      // (a verifier's entry appeared in the gap: the caller retries the call; never seen)
    }
    std::unique_ptr<GensEntry> ent(new GensEntry());
    FoldTabCfg fc = gens_full_cfg(P0, N);
    // Fast start: allocating a 25 GB table takes 0 - 0.7 s depending on the state of the box, and the first call of a process waited for it.
    // The first calls are served from the compact 16-slice fold table (0.8 GB at cfg 2; the first fold is ~2.5 ms slower) while a background
    // thread allocates and builds the full table, then swaps it in; calls that still read the compact entry keep it alive until they return.
    const FoldTabCfg fc_full = fc;
    const bool lazy = role == GENS_PROVE && gens_lazy_for(fc, N);
    if (lazy) fc = FoldTabCfg{64, 4, 4, 4};
    if (role == GENS_VERIFY) fc = FoldTabCfg{64, 4, 1, 1};      // one "slice": the generators themselves
    void *tblv = nullptr;
    hipError_t me = gens_malloc(P0, &tblv, sizeof(niels) * 2 * N * fc.np * fc.e, nullptr);      // slice 0 = generators, the rest = fold tables
    while (me != hipSuccess && !(fc.pb == 64 && fc.w == 4) && fc.w > 7) {      // HBM is short: one NAF width less = half the table
        fc.w--; fc.e = 1u << (fc.w - 2);
        me = gens_malloc(P0, &tblv, sizeof(niels) * 2 * N * fc.np * fc.e, nullptr);
    }
    if (me != hipSuccess && !(fc.pb == 64 && fc.w == 4)) {      // still short even after eviction: the compact fold-table layout (16 slices)
        fc = FoldTabCfg{64, 4, 4, 4};
        me = gens_malloc(P0, &tblv, sizeof(niels) * 2 * N * fc.np * fc.e, nullptr);
    }
    if (me != hipSuccess) throw HipErr{me, "hipMalloc(generator tables)"};
    niels *tbl = reinterpret_cast<niels *>(tblv);
    ent->tbl = tbl; ent->fc = fc; ent->n = n; ent->m = m; ent->has_fold = role != GENS_VERIFY;
    ent->bytes = sizeof(niels) * 2 * N * fc.np * fc.e;
    try {
        uint8_t *uni = C.uni.as<uint8_t>(2 * N * 64);
        ROFL_LAUNCH(k_gens_xof, grid1(2 * m), dim3(TPB), 0, C.stream, (u32)n, (u32)m, uni);
        ROFL_LAUNCH(k_gens_map, grid1(2 * N), dim3(TPB), 0, C.stream, (u32)(2 * N), uni, tbl);
        if (ent->has_fold) ROFL_LAUNCH(k_gens_tables, grid1(2 * N * fc.np), dim3(TPB), 0, C.stream, (u32)(2 * N), fc, tbl, (size_t)(2 * N));
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
                ent->wt = std::make_shared<WTabs>(); ent->wt->wtab = wt; ent->wt->bytes = sizeof(ndm) * 2 * N * fp.W;
                ROFL_LAUNCH(k_gens_wtab, grid1(2 * N), dim3(TPB), 0, C.stream, (u32)(2 * N), MsmWin{fp.c, fp.W, fp.wide}, tbl, wt, (size_t)(2 * N));
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
                        ROFL_LAUNCH(k_gens_wtab, grid1(2 * N), dim3(TPB), 0, C.stream, (u32)(2 * N), MsmWin{f2.c, f2.W, f2.wide}, tbl, reinterpret_cast<ndm *>(w2), (size_t)(2 * N));
                        ent->wt->wtab_many = reinterpret_cast<ndm *>(w2); ent->wt->bytes += sizeof(ndm) * 2 * N * f2.W;
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
    if (lazy) {
        raw->users++;                                // the builder reads this entry's generators: pinned until it has swapped (or given up)
        raw->full_state = 1;
        P0.gens_pending[key] = 1;
        P0.gens_upgrades.emplace_back([&P0, raw, key, fc_full, N] { gens_upgrade(P0, raw, key, fc_full, N, /*wait_quiet=*/true); });
    }
    return GensPin(&P0, raw);
}
// rofl_bp_gens_prepare: the shape's tables are COMPLETE when it returns (a server calls it at start-up and pays the whole build there).
// A full table whose background build failed for want of HBM is tried once more here, synchronously; false = the shape stays on the
// compact fold table (slower first fold, same results) and the caller is told.
bool gens_wait_full(Ctx &C, size_t n, size_t m) {
    Ctx &P0 = C.parent ? *C.parent : C;
    const auto key = std::make_pair(n, m);
    for (;;) {
        GensEntry *retry = nullptr; FoldTabCfg fc_full{};
        {   std::lock_guard<std::mutex> lk(P0.gens_mu);
            auto pit = P0.gens_pending.find(key);
            if (pit != P0.gens_pending.end()) pit->second = 2;      // 2 = somebody is waiting: build now
            else {
                auto it = P0.gens.find(key);
                if (it == P0.gens.end() || it->second->full_state == 0) return true;
                if (it->second->full_state < 0) {      // the build failed earlier: once more, on this thread
                    retry = it->second.get(); retry->users++; retry->full_state = 1; P0.gens_pending[key] = 2;
                    fc_full = FoldTabCfg{P0.fold_pb, P0.fold_w, 256 / P0.fold_pb, 1u << (P0.fold_w - 2)};
                    while (fc_full.w > 6 && sizeof(niels) * 2 * n * m * fc_full.np * fc_full.e > P0.fold_tab_budget) { fc_full.w--; fc_full.e = 1u << (fc_full.w - 2); }
                }
            }
        }
        if (retry) return gens_upgrade(P0, retry, key, fc_full, n * m, /*wait_quiet=*/false);
        struct timespec ts = {0, 500000}; nanosleep(&ts, nullptr);
    }
}
// the background builders are joined before the process tears the HIP runtime down
struct GensUpgradeJoiner { ~GensUpgradeJoiner() { g_gens_shutdown.store(true); std::vector<std::thread> ts; { std::lock_guard<std::mutex> lk(g_ctx_mu); for (auto &kv : g_ctxs) for (auto &t : kv.second->gens_upgrades) ts.push_back(std::move(t)); }
                                                  for (auto &t : ts) if (t.joinable()) t.join(); } } g_gens_upgrade_joiner;
