// The MSM driver of librofl_zk.so: plan -> enqueue (sort, accumulate, reduce) -> retry -> finish (window combination on the host pool),
// msm_run / msm_run2.  Included by rofl_zk.hip inside its anonymous namespace, after host_rt.hpp.
#pragma once

// ---------------------------------------------------------------- MSM driver
MsmPlan msm_plan(size_t n, size_t np = 1) {
    MsmPlan p;
    static const size_t t13 = knob("ROFL_MSM_T13") ? (size_t)atol(knob("ROFL_MSM_T13")) : ((size_t)1 << 13);
    // 10-bit windows from 512 terms on -- from 2 048 on when the launch carries many problems (n_partition = 64: 128 problems of 1 024 terms
    // run as 4 736 one-wave blocks of 64 buckets rather than 3 328 eight-wave blocks of 512: create 25.1 / 25.9 -> 24.5 ms same-box)
    static const size_t t10_knob = knob("ROFL_MSM_T10") ? (size_t)atol(knob("ROFL_MSM_T10")) : 0;
    const size_t t10 = t10_knob ? t10_knob : (np >= 32 ? (size_t)1 << 11 : (size_t)1 << 9);
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
struct MsmOpt { u32 lr_nh = 0, lr_ng = 0; const niels *fb_wtab = nullptr; size_t fb_stride = 0; u32 fb_c = 16; std::function<void()> overlap; std::function<void(size_t)> post; u32 tag = 0;      /* tag: which hop of its caller this is (0 = unknown): keys the wait-time estimate */
                std::function<void(size_t, int)> post8;
                // inner-product rounds: the launch may add c_side * Q to problem 2c + side itself (the fused small-MSM launch does): partial inner
                // products [chunk][ip_nblk][2] in device memory, Q per chunk; *ip_included tells the caller's finisher whether it happened
                const sc *ip_dev = nullptr; u32 ip_nblk = 0; const niels *qpts = nullptr; bool *ip_included = nullptr;
                hipEvent_t pts_ready = nullptr; };      // the points are still being written on another stream: the first kernel that reads them waits for this      // post8(p0, count): the finisher of problems p0 .. p0 + count - 1 (p0 a multiple of 8) of a host8 task, instead of `count` calls of post

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
    u32 small_group = 1;     // fused small launch: windows of a problem per block (4 at c = 7 with thousands of bucket arrays: n_partition = 64)
    bool fb() const { return kind == MsmKind::FixedBase; }
};
static const u32 MSM_OVF_MAX = 4096;

// ---- plan
// Entries a coarse bin of the two-level sort has room for, given the mean load `avg` of a bin.  The (c-1)-bit windows of the layout reach only
// half of the buckets, so the bins of that half carry 2 x avg; on top of that eight standard deviations of a Poisson load (a fixed margin of
// 256 was 2.8 sigma at avg = 4 096 -- the verifier's 2^20-term generator MSM of a d = 55 000 client overflowed a bin in six calls of ten and
// was repeated on the slot path, 3.0 -> 4.5 ms per verification).
static inline u32 msm_bin_cap(size_t avg) {
    static const int sigmas = knob("ROFL_MSM_BIN_SIGMA") ? atoi(knob("ROFL_MSM_BIN_SIGMA")) : 8;
    size_t hot = 2 * avg, dev = 1;
    if (sigmas <= 0) return (u32)((hot + 256 + 63) / 64 * 64);
    while (dev * dev < hot) dev++;                      // ceil(sqrt(hot))
    return (u32)((hot + (size_t)sigmas * dev + 64 + 63) / 64 * 64);
}

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
        // ... but the two-level sort needs a coarse bin to fit a block's LDS: with many problems of many terms (a batched call of cfg-4 clients:
        // 48 problems of 2^19 terms a side) one set per problem means 16 windows per array and bins of 131 KB -- the launch fell back to the slot
        // sort (a 6.4 GB slot array, 3x the sort time).  More sets per problem until the bins fit (ROFL_MSM_FB_FITSETS=0: as before).
        static const bool fitsets = !(knob("ROFL_MSM_FB_FITSETS") && atoi(knob("ROFL_MSM_FB_FITSETS")) == 0);
        if (fitsets && al.two && C.msm_two_level && (J.P.B == 32768 || J.P.B == 16384) && per_side >= 8192)
            while (sets < J.P.W && (size_t)msm_bin_cap(per_side * (J.P.W / sets) / 512) * 4 + 1024 > 96 * 1024) {
                u32 nx = sets + 1; while (nx < J.P.W && J.P.W % nx) nx++;
                sets = nx;
            }
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
            auto size_bins = [&]() { tl.cap_bin = msm_bin_cap(per_side * mm.fb_wps / tl.nbins); return (size_t)tl.cap_bin * 4 + 1024 <= 96 * 1024; };
            bool fits = size_bins();
            if (!fits) { tl = Msm2L{512, fb0 - 1, 24, 0, 72}; fits = size_bins(); }
            if (!fits || per_side < 8192) two = false;
        }
        J.two = two;
        if (!two && (size_t)J.PW * J.P.B * J.cap * 4 > ((size_t)8 << 30)) return false;      // slot array too large: next variant
        return true;
    }
    bool slots_mode = al.slots && C.msm_slots;
    J.P = msm_plan(n, J.np);
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
    // Thousands of one-wave blocks (c = 7: 64 buckets, one (problem, window) each -- n_partition = 64 runs 128 problems x 37 windows per round)
    // are bound by instruction issue, and a wave works as long as its FULLEST bucket (mean load 2-4, maximum ~14): four windows of a problem
    // share a block of four waves, their 256 buckets are sorted by load so that every wave gets buckets of similar depth (14 + 5 + 3 + 2
    // additions instead of 4 x 14), and the four bit-sum trees run side by side in the same lanes (ROFL_MSM_SMALL_GROUP=1: one window per block).
    J.small_group = 1;
    if (small && J.P.B == 64 && J.PW > 512) {
        static const u32 grp = knob("ROFL_MSM_SMALL_GROUP") ? (u32)std::max(1, std::min(4, atoi(knob("ROFL_MSM_SMALL_GROUP")))) : 4u;
        J.small_group = grp == 3 ? 2 : grp;
    }
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
    // The window chains of a launch (253 doublings per problem) run (a) on the host pool, two half chains per problem -- 16 tasks of ~31 us for
    // the eight L / R problems of a four-chunk client, as fast as it gets with sixteen idle cores, 4x as long with four; (b) with AVX-512 IFMA,
    // eight problems per instruction stream on ONE thread after the device has added up each window's bit-sums (k_msm_wsum): ~50 us for the
    // hop's whole host part (chains, eight encodings, four transcripts, one batched inversion), no pool hand-off, the other cores stay asleep;
    // (c) on the device (k_msm_horner) when there are many problems and no IFMA.  (b) from eight problems on (ROFL_MSM_HOST8_MIN): on a rank
    // with 4 cores -- one of eight ranks of a node -- 24.2 against 25.0 ms per client and 2.5 against 2.8 busy cores; on a loaded 16-core host
    // as fast as (a) at 2.5 against 6.6 busy cores (profiles/r04_experiments.txt item 4).
    static const bool host8_on = h8::available() && !(knob("ROFL_MSM_HOST8") && atoi(knob("ROFL_MSM_HOST8")) == 0);
    static const size_t host8_min = knob("ROFL_MSM_HOST8_MIN") ? (size_t)std::max(1L, atol(knob("ROFL_MSM_HOST8_MIN"))) : 8;
    J.dev_horner = !fb && P.W <= 64 && (np >= C.msm_dev_horner_min || (host8_on && np >= host8_min));
    J.host8 = J.dev_horner && host8_on;
    ge *hres_dev = W.h_res.dev<ge>(PW * (size_t)P.c + np);
    u32 *h_flag = W.h_ovf.as<u32>(4), *d_flag = W.h_ovf.dev<u32>(4);
    const u32 Wb = (u32)(PW / nq);                           // bucket arrays per grid problem
    const u32 n_side = (u32)(lr ? n / 2 : n);
    const u32 nb_final = P.c - 1;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    *h_flag = 0;
    if (opt.ip_included) *opt.ip_included = J.kind == MsmKind::Small && opt.ip_dev && opt.qpts && lr;
    if (J.kind == MsmKind::Small) {
        ge *S_fin_s = J.dev_horner ? W.S[0].as<ge>(PW) : hres_dev;
        ge *C_fin_s = J.dev_horner ? W.Cacc[0].as<ge>(PW * (size_t)nb_final) : hres_dev + PW;
        const u32 G = J.small_group;
        size_t lds_lists = G > 1 ? (size_t)G * P.B * (4 + 2 * (size_t)small_cap)      // 16-bit list entries (k_msm_small_g)
                                 : (size_t)P.B * 4 * (1 + small_cap);
        size_t lds_red = G > 1 ? ((size_t)G * P.B * 3 / 4 + 1) * sizeof(ge)      // in-place trees over G windows
                               : std::max(((size_t)(P.B / 8) * 4 + (size_t)(P.B / 16) * 5 + 1) * sizeof(ge), ((size_t)P.B + (size_t)P.B * 3 / 4 + 1) * sizeof(ge));      // fused / binary reduction trees
        uint64_t items = (uint64_t)np * n_side * P.W;
        KSpan ks(C.tm, st, ROFL_TK_MSM_SMALL, items * 7 + (uint64_t)PW * P.B * 10, (uint64_t)np * n_side * (32 + 96));
        static const bool small_tl = knob("ROFL_DBG_SMALL_TIMELINE") != nullptr;
        unsigned long long *tl_dev = nullptr;
        if (small_tl) { HIPCHK(hipMalloc(&tl_dev, PW * 32)); HIPCHK(hipMemsetAsync(tl_dev, 0, PW * 32, st)); }
        if (opt.pts_ready) HIPCHK(hipStreamWaitEvent(st, opt.pts_ready, 0));
        const u32 Wg = (P.W + G - 1) / G;      // blocks per problem
        if (G > 1)
            ROFL_LAUNCH(k_msm_small_g, dim3((unsigned)(np * Wg)), dim3(G * P.B), std::max(lds_lists, lds_red), st, n_side, mw, mm, d_probs, buckets,
                               S_fin_s, C_fin_s, nb_final, d_flag, small_cap, G, tl_dev, J.host8 ? hres_dev : (ge *)nullptr,
                               lr ? opt.ip_dev : (const sc *)nullptr, opt.ip_nblk, lr ? opt.qpts : (const niels *)nullptr);
        else
            ROFL_LAUNCH(k_msm_small, dim3((unsigned)PW), dim3(P.B < 64 ? 64 : P.B), std::max(lds_lists, lds_red), st, n_side, mw, mm, d_probs, buckets,
                               S_fin_s, C_fin_s, nb_final, d_flag, small_cap, G, tl_dev, J.host8 ? hres_dev : (ge *)nullptr,
                               lr ? opt.ip_dev : (const sc *)nullptr, opt.ip_nblk, lr ? opt.qpts : (const niels *)nullptr);
        if (small_tl) {      // mean phase durations over the blocks of this launch (100 MHz clock)
            std::vector<unsigned long long> hts(PW * 4);
            HIPCHK(hipMemcpyAsync(hts.data(), tl_dev, PW * 32, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st)); HIPCHK(hipFree(tl_dev));
            double ph[3] = {0, 0, 0}; unsigned long long t_lo = ~0ull, t_hi = 0;
            const size_t nblocks = G > 1 ? np * Wg : PW;
            for (size_t b = 0; b < nblocks; b++) { for (int k = 0; k < 3; k++) ph[k] += (double)(hts[b * 4 + k + 1] - hts[b * 4 + k]); t_lo = std::min(t_lo, hts[b * 4]); t_hi = std::max(t_hi, hts[b * 4 + 3]); }
            fprintf(stderr, "[rofl] k_msm_small PW=%zu blocks=%zu n_side=%u c=%u: rank %.1f us, bucket sums %.1f us, reduce %.1f us (block means); first start -> last end %.1f us\n",
                    PW, nblocks, n_side, P.c, ph[0] / nblocks * 0.01, ph[1] / nblocks * 0.01, ph[2] / nblocks * 0.01, (double)(t_hi - t_lo) * 0.01);
        }
        if (J.host8) {}      // (the launch added up each window's bit-sums itself)
        else if (J.dev_horner) ROFL_LAUNCH(k_msm_horner, dim3((unsigned)np), dim3(256), 0, st, mw, (const ge *)S_fin_s, (const ge *)C_fin_s, nb_final, hres_dev);
        return J;
    }
    // ---- SORT (+ ACCUMULATE: its list format depends on the sort)
    const uint64_t terms = (uint64_t)(lr ? nq : np) * n;
    if (J.two) {
        u32 *bins = W.sorted.as<u32>(PW * tl.nbins * tl.cap_bin);
        u32 *bcur = W.cur.as<u32>(PW * tl.nbins * 2);
        u32 *btail = W.tail.as<u32>(PW * tl.nbins * (size_t)MSM_BIN_TAIL);
        HIPCHK(hipMemsetAsync(bcur, 0, sizeof(u32) * PW * tl.nbins * 2, st));
        // window-ordered bucket lists + set-major block order of the accumulation (ROFL_MSM_WINDOW_ORDER; r06_experiments.txt item 14): when the index of
        // an entry says which window slot it belongs to by a shift (stride and sets powers of two) and there is more than one slot per array
        static const bool worder_on = !(knob("ROFL_MSM_WINDOW_ORDER") && atoi(knob("ROFL_MSM_WINDOW_ORDER")) == 0);
        const u64 wdiv = (u64)mm.fb_stride * mm.fb_sets;
        const bool worder = worder_on && mm.fb_wps > 1 && mm.fb_wps <= 16 && wdiv && (wdiv & (wdiv - 1)) == 0 && wdiv < ((u64)1 << tl.ebits);
        const u32 worder_wps = worder ? mm.fb_wps : 1u, worder_shift = worder ? (u32)lg2u((size_t)wdiv) : 0u;
        u32 iter_pts = (tl.stage >= 144 ? 16384u : 12288u) / mm.fb_wps; if (iter_pts < 1024) iter_pts = 1024;      // ~ 64 (48) new items per bin and iteration against a row of 144 (72)
        u32 tile = iter_pts;
        while (((size_t)((n_side + tile - 1) / tile) * PW > 512 || (n_side + tile - 1) / tile > 48) && tile < n_side) tile *= 2;      // <= 48 tiles per array: their left-overs (< 32 each) fit the bin tails with room for row spills
        {   // A block holds a CU (its staging rows are ~148 KB of LDS), so the launch runs in ceil(blocks / 256) passes of one tile each.  With many bucket
            // arrays (a batch of clients: PW = 96) the doubling above lands on e.g. 3 tiles x 96 arrays = 288 blocks -- a full pass and a second one for 32
            // blocks: 5.1 ms where 2.6 were due (profiles/r06_cfg4_timeline_inflight1.txt).  Look for a tile (any multiple of an iteration) whose passes
            // x tile length is shorter; the power-of-two choice stays unless another is clearly better (a lone client's 512 blocks are two exact passes).
            static const bool tile_search = !(knob("ROFL_MSM_BIN_TILE_SEARCH") && atoi(knob("ROFL_MSM_BIN_TILE_SEARCH")) == 0);
            const size_t kCU = 256;
            auto cost = [&](u32 t) { const size_t tiles = (n_side + t - 1) / t; return (double)((tiles * PW + kCU - 1) / kCU) * ((double)std::min<u32>(t, n_side) + 0.25 * iter_pts); };      // (+ a block's fixed part: counters, the flush of its rows)
            if (tile_search) {
                u32 best = tile; double best_cost = cost(tile);
                for (u32 t = iter_pts; t < n_side + iter_pts; t += iter_pts) {
                    const size_t tiles = (n_side + t - 1) / t;
                    if (tiles > 48 || tiles * PW > 4096) continue;
                    const double c = cost(t);
                    if (c < best_cost * 0.999) { best = t; best_cost = c; }
                }
                if (knob("ROFL_TRACE")) fprintf(stderr, "[rofl] bin_l1 tiling: n_side=%u PW=%zu iter=%u: tile %u (%zu blocks, cost %.0f) -> best %u (%zu blocks, cost %.0f)\n", n_side, PW, iter_pts, tile,
                                                (size_t)((n_side + tile - 1) / tile) * PW, cost(tile), best, (size_t)((n_side + best - 1) / best) * PW, best_cost);
                if (best_cost < 0.85 * cost(tile)) tile = best;
            }
        }
        dim3 grid((n_side + tile - 1) / tile, (u32)PW);
        { KSpan ks(C.tm, st, ROFL_TK_MSM_SCATTER, 0, terms * 32 + terms * P.W * 4);
          ROFL_LAUNCH(k_msm_bin_l1, grid, dim3(1024), (size_t)(tl.nbins + tl.nbins * tl.stage) * 4, st, n_side, tile, iter_pts, mw, mm, d_probs, bcur, bins, btail, tl, d_flag);
          ROFL_LAUNCH(k_msm_bin_l2, dim3(tl.nbins, (u32)PW), dim3(512), (size_t)(2 * 128 * worder_wps + tl.cap_bin) * 4, st, tl, P.B, bcur, bins, (const u32 *)btail, cnt, off, d_flag,
                      worder_wps, worder_shift); }
        ROFL_LAUNCH(k_msm_scan, dim3((unsigned)PW), dim3(P.B >= 1024 ? 1024 : 256), 0, st, P.B, cnt, off, (u32 *)nullptr, perm);
        if (C.tm.enabled) { e0 = C.tm.get(); e1 = C.tm.get(); HIPCHK(hipEventRecord(e0, st)); }
        {
            HeavyScope heavy(C, st); hipStream_t hst = heavy.run;      // (the debug timeline variant below stays on st)
            KSpan ks_acc(C.tm, hst, ROFL_TK_MSM_ACCUMULATE_FB, terms * P.W * 7, terms * 32);
            static const char *timeline = knob("ROFL_DBG_ACC_TIMELINE");      // debugging: per-wave start / end / placement of every launch, appended to this file
            if (timeline) {
                dim3 g = grid1((size_t)Wb * P.B, (u32)nq);
                size_t waves = (size_t)g.x * g.y * (TPB / 64);
                unsigned long long *rec; HIPCHK(hipMalloc(&rec, waves * 32)); HIPCHK(hipMemsetAsync(rec, 0, waves * 32, st));
                ROFL_LAUNCH(k_msm_accumulate_fb_dbg, g, dim3(TPB), 0, st, (u32)n, P.c, Wb, (u32)(np / nq), d_probs, cnt, off, bins, perm, buckets, MSM_LIST_ABS, dbg_mask, acc_balance, rec);
                std::vector<unsigned long long> h(waves * 4);
                HIPCHK(hipMemcpyAsync(h.data(), rec, waves * 32, hipMemcpyDeviceToHost, st)); HIPCHK(hipStreamSynchronize(st));
                if (FILE *f = fopen(timeline, "ab")) { unsigned long long hdr[4] = {0x54494d45ull, waves, g.x, g.y}; fwrite(hdr, 8, 4, f); fwrite(h.data(), 8, h.size(), f); fclose(f); }
                HIPCHK(hipFree(rec));
            } else
            ROFL_LAUNCH(k_msm_accumulate_fb, grid1((size_t)Wb * P.B, (u32)nq), dim3(TPB), 0, hst, (u32)n, P.c, Wb, (u32)(np / nq), d_probs, cnt, off, bins, perm, buckets, MSM_LIST_ABS, dbg_mask,
                        acc_balance | ((worder && nq > 1 && Wb > 1) ? 2u : 0u));
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
            ROFL_LAUNCH(k_msm_scatter_lds, grid, dim3(1024), (size_t)P.B * 4, st, n_side, tile, mw, mm, d_probs, cnt, slots, cap, ovf_count, ovf, MSM_OVF_MAX, dbg_scatter);
        } else
            ROFL_LAUNCH(k_msm_scatter_slots, grid1(n, (u32)(nq * P.W)), dim3(TPB), 0, st, (u32)n, mw, mm, d_probs, cnt, slots, cap, ovf_count, ovf, MSM_OVF_MAX);
        ROFL_LAUNCH(k_msm_scan, dim3((unsigned)PW), dim3(P.B >= 1024 ? 1024 : 256), 0, st, P.B, cnt, off, (u32 *)nullptr, perm);
        if (opt.pts_ready) HIPCHK(hipStreamWaitEvent(st, opt.pts_ready, 0));      // (the sort above only read the scalars)
        if (C.tm.enabled) { e0 = C.tm.get(); e1 = C.tm.get(); HIPCHK(hipEventRecord(e0, st)); }
        {
            uint64_t acc_adds = terms * P.W;
            const bool big = acc_adds >= ((uint64_t)1 << 22);      // (a launch of a few hundred microseconds is not worth two events)
            HeavyScope heavy(C, st, big); hipStream_t hst = heavy.run;
            KSpan ks_acc(C.tm, hst, fb ? ROFL_TK_MSM_ACCUMULATE_FB : ROFL_TK_MSM_ACCUMULATE_GEN, acc_adds * 7, terms * 32);
            if (fb) ROFL_LAUNCH(k_msm_accumulate_fb, grid1((size_t)Wb * P.B, (u32)nq), dim3(TPB), 0, hst, (u32)n, P.c, Wb, (u32)(np / nq), d_probs, cnt, off, slots, perm, buckets, cap, dbg_mask, acc_balance);
            else ROFL_LAUNCH(k_msm_accumulate_gen, grid1((size_t)Wb * P.B, (u32)nq), dim3(TPB), 0, hst, (u32)n, P.c, Wb, (u32)(np / nq), d_probs, cnt, off, slots, perm, buckets, cap, dbg_mask, acc_balance);
        }
        if (C.tm.enabled) HIPCHK(hipEventRecord(e1, st));
        ROFL_LAUNCH(k_msm_overflow, dim3(1), dim3(64), 0, st, Wb, P.B, (u32)(np / nq), d_probs, ovf_count, ovf, MSM_OVF_MAX, buckets, fb ? 1 : 0);
        HIPCHK(hipMemcpyAsync(h_flag, ovf_count, 4, hipMemcpyDeviceToHost, st));
    } else {      // count / scan / scatter: no fixed-size structure, always sufficient
        HIPCHK(hipMemsetAsync(cnt, 0, sizeof(u32) * (PW * P.B + 4), st));
        u32 *sorted = W.sorted.as<u32>(PW * n * 2);
        ROFL_LAUNCH(k_msm_count, grid1(n, (u32)(nq * P.W)), dim3(TPB), 0, st, (u32)n, mw, mm, d_probs, cnt);
        ROFL_LAUNCH(k_msm_scan, dim3((unsigned)PW), dim3(P.B >= 1024 ? 1024 : 256), 0, st, P.B, cnt, off, cur, perm);
        ROFL_LAUNCH(k_msm_scatter, grid1(n, (u32)(nq * P.W)), dim3(TPB), 0, st, (u32)n, mw, mm, d_probs, cur, sorted);
        if (opt.pts_ready) HIPCHK(hipStreamWaitEvent(st, opt.pts_ready, 0));
        if (C.tm.enabled) { e0 = C.tm.get(); e1 = C.tm.get(); HIPCHK(hipEventRecord(e0, st)); }
        ROFL_LAUNCH(k_msm_accumulate_gen, grid1((size_t)Wb * P.B, (u32)nq), dim3(TPB), 0, st, (u32)n, P.c, Wb, (u32)(np / nq), d_probs, cnt, off, sorted, perm, buckets, 0u, dbg_mask, acc_balance);
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
            ROFL_LAUNCH(k_msm_reduce_fused, dim3((unsigned)(PW * G)), dim3(256), lds_a, st, 512u, 0u, (const ge *)buckets, (const ge *)nullptr, GS, GC, 9u);
            u32 half = G / 2 ? G / 2 : 1, nout = 10 + gbits;
            ROFL_LAUNCH(k_msm_reduce_groups, dim3((unsigned)PW), dim3(half, nout), (size_t)nout * half * sizeof(ge), st, G, gbits, (const ge *)GS, (const ge *)GC, S_fin, C_fin, nb_final);
        } else {
            while (E > 512) {
                u32 E8 = E / 8;
                ge *S_out = W.S[lv & 1].as<ge>(PW * E8);
                ge *C_out = W.Cacc[lv & 1].as<ge>(PW * (size_t)(nb + 3) * E8);
                static const int red_split = knob("ROFL_RED_SPLIT") ? atoi(knob("ROFL_RED_SPLIT")) : 0;
                ROFL_LAUNCH(k_msm_reduce_level, grid1((size_t)E8 * ((red_split ? 4 : 1) + nb), (u32)PW), dim3(TPB), 0, st, E, nb, S_in, C_in, S_out, C_out, red_split);
                S_in = S_out; C_in = C_out; E = E8; nb += 3; lv++;
            }
            if (J.dev_horner) { S_fin = W.S[lv & 1].as<ge>(PW); C_fin = W.Cacc[lv & 1].as<ge>(PW * (size_t)nb_final); }
            // block size = first-level work items (small bucket arrays, c = 7: 32 items -- a 256-thread block would idle 7 of its 8
            // waves and, at 163 VGPRs, hold a whole CU: thousands of such blocks (n_partition = 64) ran 18 deep per CU)
            static const u32 red_fused_max = knob("ROFL_RED_FUSED_T") ? (u32)atoi(knob("ROFL_RED_FUSED_T")) : 768u;
            // (a fixed-base array enters with E = 512, nb = 6: 640 first-level items -- on 512 threads that was two passes of a seven-addition chain)
            u32 fused_items = (E / 8) * (4 + nb), fused_threads = fused_items > 512 ? (fused_items > 640 ? 768 : 640) : fused_items > 256 ? 512 : fused_items > 128 ? 256 : fused_items > 64 ? 128 : 64;
            if (fused_threads > red_fused_max) fused_threads = red_fused_max;
            size_t lds = ((size_t)(E / 8) * (1 + nb + 3) + (size_t)(E / 16) * (1 + nb + 4) + 1) * sizeof(ge);
            ROFL_LAUNCH(k_msm_reduce_fused, dim3((unsigned)PW), dim3(fused_threads), lds, st, E, nb, S_in, C_in, S_fin, C_fin, nb_final);
        }
    }
    if (J.host8)           // many problems, AVX-512 IFMA host: the device adds up each window's bit-sums, the chains across the windows go to the host
        ROFL_LAUNCH(k_msm_wsum, grid1(PW * 4), dim3(TPB), 0, st, (u32)PW, (const ge *)S_fin, (const ge *)C_fin, nb_final, hres_dev);
    else if (J.dev_horner)      // many problems: their Horner chains run side by side on the device, one point per problem comes back
        ROFL_LAUNCH(k_msm_horner, dim3((unsigned)np), dim3(256), 0, st, mw, (const ge *)S_fin, (const ge *)C_fin, nb_final, hres_dev);
    return J;
}

// after the stream has been synchronised: did the attempt overflow one of its fixed-size structures?  (then `al` has lost that variant)
// rofl_dbg_msm_retries: MSMs finished / repeated after a list overflow of the fused small launch / after a coarse-bin overflow of the two-level
// sort (-> slot path) / after an overflow of the slot path's overflow list (-> the next slower variant).  Process-wide; the tests of the server
// path use them to show that a scenario really took the path it was built for.
std::atomic<uint64_t> g_msm_stat[4];
bool msm_retry_inner(const MsmJob &J, MsmAllow &al);
bool msm_retry(const MsmJob &J, MsmAllow &al) {
    const bool small = J.kind == MsmKind::Small, two = J.two;
    const bool again = msm_retry_inner(J, al);
    g_msm_stat[again ? (small ? 1 : two ? 2 : 3) : 0].fetch_add(1, std::memory_order_relaxed);
    return again;
}
bool msm_retry_inner(const MsmJob &J, MsmAllow &al) {
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
            double tc0 = now_ms();
            size_t p0 = b * 8; int lanes = (int)std::min<size_t>(8, np - p0);
            ge5 out[8];
            h8::horner8(out, lanes, (int)P.W, pos, [&](int l, int w) { return (const ge *)&h[(p0 + (size_t)l) * P.W + (size_t)w]; });
            for (int l = 0; l < lanes; l++) results[p0 + l] = out[l];
            if (opt.post8) opt.post8(p0, lanes);
            else if (opt.post) for (int l = 0; l < lanes; l++) opt.post(p0 + l);
            cpu_each[p0] = now_ms() - tc0;
        });
    } else if (J.dev_horner) {
        if (opt.post) C.pool->run(np, [&](size_t p) { results[p] = h51::from_ge_loose(h[p]); opt.post(p); });
        else for (size_t p = 0; p < np; p++) results[p] = h51::from_ge_loose(h[p]);
    } else if (J.fb()) {
        // sets of a problem carry equal weight: add them up, then one Horner over the c - 1 bit-sums
        auto fb_one = [&](size_t p) {
            size_t base = p * sets;                         // lr: problem 2q+side owns sets [(2q+side)*sets, ...)
            ge5 acc = h51::identity(); bool started = false;
            for (int l = (int)nb - 1; l >= 0; l--) {
                if (started) acc = h51::gdouble(acc);
                for (u32 s = 0; s < sets; s++) { acc = h51::gadd(acc, h51::from_ge_loose(h[PW + (base + s) * nb + l])); started = true; }
                if (l == 0) for (u32 s = 0; s < sets; s++) acc = h51::gadd(acc, h51::from_ge_loose(h[base + s]));
            }
            results[p] = acc;
        };
        static const size_t fb8_min = knob("ROFL_MSM_FB_HOST8_MIN") ? (size_t)std::max(1L, atol(knob("ROFL_MSM_FB_HOST8_MIN"))) : 8;
        if (opt.post8 && np >= fb8_min && h8::available() && sets * (nb + 1) <= 64) {
            // From eight problems on (the L / R problems of a four-chunk client: ONE task, on the calling thread, no pool hand-off; n_partition = 64:
            // sixteen tasks): eight problems per AVX-512 IFMA stream, finished together -- the bit-sum chains (the sets' sums and bit-sums as
            // "windows" that share positions: sum_l 2^l sum_s D[s][l] + sum_s S[s]), eight encodings per stream and the four challenge inversions
            // of the task's chunks behind one inversion, as the generic launches do.  (Rounds 3-4 ran eight scalar chains on eight pool threads
            // here below 32 problems: ~0.1 ms per hop against ~0.05.)
            const int Wn = (int)(sets * (nb + 1));
            u32 pos[64];
            for (u32 s = 0; s < sets; s++) pos[s] = 0;
            for (u32 l = 0; l < nb; l++) for (u32 s = 0; s < sets; s++) pos[sets + l * sets + s] = l;
            C.pool->run((np + 7) / 8, [&](size_t b) {
                double tc0 = now_ms();
                size_t p0 = b * 8; int cnt = (int)std::min<size_t>(8, np - p0);
                ge5 out[8];
                h8::horner8(out, cnt, Wn, pos, [&](int l, int w) {
                    const size_t base = (p0 + (size_t)l) * sets;
                    if ((u32)w < sets) return (const ge *)&h[base + (size_t)w];
                    const u32 lv = ((u32)w - sets) / sets, s = ((u32)w - sets) % sets;
                    return (const ge *)&h[PW + (base + s) * nb + lv];
                });
                for (int l = 0; l < cnt; l++) results[p0 + (size_t)l] = out[l];
                opt.post8(p0, cnt);
                cpu_each[p0] = now_ms() - tc0;
            });
        } else
        C.pool->run(np, [&](size_t p) {
            double tc0 = now_ms();
            fb_one(p); if (opt.post) opt.post(p); cpu_each[p] = now_ms() - tc0;
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
    { double w = now_ms() - t0; C.hs.horner_wall += w; C.hs.max_horner = std::max(C.hs.max_horner, w); double mx = 0; for (double v : cpu_each) mx = std::max(mx, v); C.hs.horner_cpu += mx; C.hs.max_task = std::max(C.hs.max_task, mx); } C.hs.n++;
}

// one MSM, start to finish: enqueue, wait, repeat through the next variant if a fixed-size structure overflowed, combine.
void msm_run(Ctx &C, const std::vector<MsmProb> &probs, size_t n, std::vector<ge5> &results, const MsmOpt &opt = MsmOpt()) {
    MsmAllow al; bool overlap_done = false;
    for (;;) {
        double t_enter = now_ms();
        MsmJob J = msm_enqueue(C, C.mws[0], probs, n, opt, al);
        if (opt.overlap && !overlap_done) { opt.overlap(); overlap_done = true; }
        double t_sync0 = now_ms(); C.hs.enqueue += t_sync0 - t_enter; C.hs.max_enqueue = std::max(C.hs.max_enqueue, t_sync0 - t_enter);
        const uint64_t wkey = ((uint64_t)opt.tag << 40) ^ ((uint64_t)probs.size() << 28) ^ (uint64_t)n;
        if (opt.tag) { auto it = C.wait_ms.find(wkey); C.pool->expect_gap(it == C.wait_ms.end() ? 0.0 : it->second * 1e3); }
        C.sync();
        { double w = now_ms() - t_sync0; C.hs.sync += w; C.hs.max_sync = std::max(C.hs.max_sync, w);
          if (opt.tag) { double &e = C.wait_ms[wkey]; e = e == 0 ? w : std::min(w, 0.5 * (e + w)); } }
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
