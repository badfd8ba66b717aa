// The prover of librofl_zk.so (bulletproofs RangeProof::prove_multiple with lazily folded generators): transcript helpers + prove_chunks.
// Included by rofl_zk.hip inside its anonymous namespace, after host_msm.hpp.
#pragma once

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
    ROFL_LAUNCH(k_nonce_expand, grid1(per / 2 + 1, (u32)P), dim3(TPB), 0, C.stream, (u32)n, (u32)m, d_cp, sL, sR, party, Scanon);
    // A partials: they depend on the values only and the host reads them after the S MSM -- side stream, beside the nonce expansion
    if (!C.stream2) HIPCHK(hipStreamCreateWithFlags(&C.stream2, hipStreamNonBlocking));
    if (!C.ev_a) { HIPCHK(hipEventCreateWithFlags(&C.ev_a, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&C.ev_a0, hipEventDisableTiming)); }
    HIPCHK(hipEventRecord(C.ev_a0, C.stream));                    // d_vshift is ready (and the previous call's reads of `partial` are done)
    HIPCHK(hipStreamWaitEvent(C.stream2, C.ev_a0, 0));
    ge *partial = C.partial.as<ge>(P * m);
    ROFL_LAUNCH(k_bitcommit, grid1(m, (u32)P), dim3(TPB), 0, C.stream2, (u32)n, (u32)m, d_vshift, tbl, partial);
    u32 nblkA = (u32)std::min<size_t>(16, (m + TPB - 1) / TPB);
    ge *partial2 = C.partial2.as<ge>(P * nblkA);
    ROFL_LAUNCH(k_point_sum, dim3(nblkA, (u32)P), dim3(TPB), TPB * sizeof(ge), C.stream2, partial, (u32)m, partial2);
    ge *h_A = C.h_part.as<ge>(P * nblkA);
    HIPCHK(hipMemcpyAsync(h_A, partial2, sizeof(ge) * P * nblkA, hipMemcpyDeviceToHost, C.stream2));
    HIPCHK(hipEventRecord(C.ev_a, C.stream2));
    u32 nblkS = (u32)std::min<size_t>(16, (m + TPB - 1) / TPB);
    sc *scpart = C.scpart.as<sc>(P * 64 * 3);
    PowTabs *d_pt = C.powtabs.as<PowTabs>(P);
    ROFL_LAUNCH(k_party_sums, dim3(nblkS, (u32)P), dim3(TPB), 0, C.stream, (u32)m, 0, d_cp, (const PowTabs *)d_pt, party, d_blind, scpart);
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
        MsmOpt mo; mo.tag = 99; if (wtab) { gens.fb_for(P, &mo.fb_wtab, &mo.fb_c); mo.fb_stride = 2 * N; }
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
    mark("y z host");
    HIPCHK(hipMemcpyAsync(d_cp, h_cp, sizeof(ChunkParams) * P, hipMemcpyHostToDevice, C.stream));
    // (enough blocks for four waves per SIMD when the chunks are few: 64 per chunk ran a four-chunk client's 1 M slots on 65 536 threads)
    u32 nblkT = (u32)std::min<size_t>(std::max<size_t>(64, std::min<size_t>(256, 2048 / P)), (N + TPB - 1) / TPB);
    sc *tpart = C.tmp_out.as<sc>(P * 256 * 3);
    ROFL_LAUNCH(k_pow_tables, dim3((4 * PT_L * PT_E + TPB - 1) / TPB, (u32)P), dim3(TPB), 0, C.stream, d_cp, d_pt, lgN, 0);
    ROFL_LAUNCH(k_poly_t, dim3(nblkT, (u32)P), dim3(TPB), 0, C.stream, (u32)n, (u32)m, d_cp, (const PowTabs *)d_pt, d_vshift, sL, sR, C.d_two_pow, tpart);
    ROFL_LAUNCH(k_party_sums, dim3(nblkS, (u32)P), dim3(TPB), 0, C.stream, (u32)m, 1, d_cp, (const PowTabs *)d_pt, party, d_blind, scpart);
    sc *h_t = C.h_part.as<sc>(P * 256 * 3);
    HIPCHK(hipMemcpyAsync(h_t, tpart, sizeof(sc) * P * nblkT * 3, hipMemcpyDeviceToHost, C.stream));
    HIPCHK(hipMemcpyAsync(h_sc, scpart, sizeof(sc) * P * nblkS * 3, hipMemcpyDeviceToHost, C.stream));
    C.sync();
    mark("t kernels");
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
    mark("T x host");
    HIPCHK(hipMemcpyAsync(d_cp, h_cp, sizeof(ChunkParams) * P, hipMemcpyHostToDevice, C.stream));
    sc *a = C.a.as<sc>(P * N), *b = C.b.as<sc>(P * N), *yinvpow = C.yinv.as<sc>(P * N);
    // l(x), r(x) -- and with them, in the same pass, the first round's MSM scalars and inner products (k_lr_first; ROFL_LR_FIRST=0: three launches)
    static const bool lr_first_on = !(knob("ROFL_LR_FIRST") && atoi(knob("ROFL_LR_FIRST")) == 0);
    const bool lr_first = lr_first_on && C.msm_lr != 0 && N >= 2;
    const u32 lr_first_blocks = (u32)std::min<size_t>(256, (N / 2 + TPB - 1) / TPB);
    if (lr_first)
        ROFL_LAUNCH(k_lr_first, dim3(lr_first_blocks, (u32)P), dim3(TPB), 0, C.stream, (u32)n, (u32)m, d_cp, (const PowTabs *)d_pt, d_vshift, sL, sR, C.d_two_pow, a, b, yinvpow,
                           C.SL.as<sc>(P * 2 * N), C.h_ip.dev<sc>(P * 256 * 2));
    else
        ROFL_LAUNCH(k_lr_vec, grid1(N, (u32)P), dim3(TPB), 0, C.stream, (u32)n, (u32)m, d_cp, (const PowTabs *)d_pt, d_vshift, sL, sR, C.d_two_pow, a, b, yinvpow);
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
    // Q = w B per chunk (upstream's InnerProductProof takes Q): the fused small-MSM launches of the later rounds add <a_L, b_R> Q and
    // <a_R, b_L> Q on the device; the other launches leave that term to the host finisher (a fixed-base multiplication per problem)
    sc *ip_dev = C.ipdev.as<sc>(P * 256 * 2);
    niels *d_q = C.qpts.as<niels>(P);
    {
        niels *h_q = C.h_q.as<niels>(P);
        C.pool->run(P, [&](size_t c) { h_q[c] = h51::to_niels32(h_fixed_mul(C.ht.B5, w[c])); });
        HIPCHK(hipMemcpyAsync(d_q, h_q, sizeof(niels) * P, hipMemcpyHostToDevice, C.stream));
    }
    bool ip_included = false, pts_pending = false;
    // odd multiples of the materialised generators for the next (non-table) fold, built on a stream of their own while the rounds before it run
    const int fold_wnaf = C.fold_unit ? C.fold_wnaf : 0;
    const u32 mult_E = fold_wnaf ? 1u << (fold_wnaf - 2) : 0;
    bool mult_ready = false, mult_used = false; const niels *mult_base = nullptr; size_t mult_count = 0;
    struct JoinMult { Ctx &c; bool &used; ~JoinMult() { if (used && c.stream3) (void)hipStreamSynchronize(c.stream3); } } join_mult{C, mult_used};      // nothing of that stream outlives the call
    std::vector<ge5> cq;      // per round: c_L w B, c_R w B of every chunk, computed while the round's MSM runs
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
            ROFL_LAUNCH(k_ipp_round, dim3(nblkI, (u32)P), dim3(TPB), 0, C.stream, (u32)n_g, (u32)n_k, use_new ? r - 1 : 0u, use_new, d_cp,
                               (const sc *)C.h_round.dev<sc>(2 * P), (const sc *)a, (const sc *)b, a2, b2, N, yinvpow, N, SL, C.h_ip.dev<sc>(P * 256 * 2),
                               (const sc *)ptab[psel], ptab[psel ^ 1], N, n_k == 2 ? C.h_abfin.dev<sc>(4 * P) : (sc *)nullptr, ip_dev);
            if (n_k == 2) ab_on_host = true;
            std::swap(a, a2); std::swap(b, b2); psel ^= 1;
        } else if (round == 0 && lr_first) {
            nblkI = lr_first_blocks;      // scalars and partial inner products were written with l(x), r(x)
        } else {
            ROFL_LAUNCH(k_ipp_scalars, grid1(n_g, (u32)P), dim3(TPB), 0, C.stream, (u32)n_g, (u32)n_k, r, d_cp, a, b, N, yinvpow, N, SL, SR, merged ? 1 : 0);
            nblkI = (u32)std::min<size_t>(32, (nh + TPB - 1) / TPB);
            ROFL_LAUNCH(k_ipp_inner, dim3(nblkI, (u32)P), dim3(TPB), 0, C.stream, (u32)nh, a, b, N, C.h_ip.dev<sc>(P * 256 * 2));
        }
        just_materialised = false;
        // <a_L, b_R> w B and <a_R, b_L> w B of the launches that do not add them on the device (everything but the fused small launch): the partial
        // inner products are in mapped host memory as soon as the round's first kernel has run, long before its MSM has -- the host computes the
        // 2P fixed-base multiples while the device works (they were 6 us each on the hop: eight in a row where one task finishes eight problems)
        if (!C.ev_ip) HIPCHK(hipEventCreateWithFlags(&C.ev_ip, hipEventDisableTiming));
        HIPCHK(hipEventRecord(C.ev_ip, C.stream));
        bool cq_ready = false;
        std::vector<MsmProb> pr(2 * P);
        for (size_t c = 0; c < P; c++) { pr[2 * c] = MsmProb{cur[c], SL + c * 2 * n_g}; pr[2 * c + 1] = MsmProb{cur[c], (merged ? SL : SR) + c * 2 * n_g}; }
        C.tm.t.msm_terms += P * 2 * n_g;
        MsmOpt mo;
        if (merged) { mo.lr_nh = (u32)nh; mo.lr_ng = (u32)n_g; }
        mo.tag = 100 + round;
        if (pts_pending) { mo.pts_ready = C.ev_norm; pts_pending = false; }
        if (fused) { mo.ip_dev = ip_dev; mo.ip_nblk = nblkI; mo.qpts = d_q; mo.ip_included = &ip_included; }
        if (first_level && wtab) { gens.fb_for(2 * P, &mo.fb_wtab, &mo.fb_c); mo.fb_stride = 2 * N; }
        // The host tail of the round runs inside the MSM's own pool tasks: the thread that finishes problem 2c (+1) adds c_L w B (c_R w B)
        // and encodes L (R); the second of a chunk's two to get there hashes both into the transcript, draws u and inverts it.  L and R
        // of a chunk are encoded side by side and the hop has one pool hand-off instead of two.
        for (size_t c = 0; c < P; c++) lr_done[c].store(0);
        mo.overlap = [&]() {      // (after the launches are queued, before the wait: ip_included is known)
            static const bool cq_on = !(knob("ROFL_HOP_CQ") && atoi(knob("ROFL_HOP_CQ")) == 0);
            if (ip_included || !cq_on) return;
            C.wait_event(C.ev_ip);
            cq.resize(2 * P);
            C.pool->run(2 * P, [&](size_t p) { sc cx = h_canon(sum_partials(h_ip + (p >> 1) * nblkI * 2, nblkI, 2, p & 1)); cq[p] = h_fixed_mul(C.ht.B5, h_mul(cx, w[p >> 1])); });
            cq_ready = true;
        };
        mo.post = [&](size_t p) {
            size_t c = p >> 1; int side = (int)(p & 1);
            uint8_t *o = proofs_out[c] + 7 * 32 + 64 * round;
            if (ip_included) h51::encode(o + 32 * side, res[p]);
            else if (cq_ready) h51::encode(o + 32 * side, h51::gadd(res[p], cq[p]));
            else {
                sc cx = h_canon(sum_partials(h_ip + c * nblkI * 2, nblkI, 2, (size_t)side));
                h51::encode(o + 32 * side, h51::gadd(res[p], h_fixed_mul(C.ht.B5, h_mul(cx, w[c]))));
            }
            if (lr_done[c].fetch_add(1) != 1) return;          // the chunk's other point is still on its way
            tr[c].append("L", o, 32); tr[c].append("R", o + 32, 32);
            sc u = tr[c].challenge_scalar("u");
            sc um = h_mont(u), uim = h51::sc_invert_mont_fast(um);
            h_round[2 * c] = um; h_round[2 * c + 1] = uim;          // mapped: k_ipp_fold_ab reads it and records it in the chunk's pending list
            h_cp[c].pend_u[pu[c].size()] = um; h_cp[c].pend_ui[pu[c].size()] = uim;
            pu[c].push_back(um); pui[c].push_back(uim);
        };
        // Launches whose window chains come back eight problems per task (host8: n_partition = 64) finish those eight together: the eight
        // encodings in the lanes of one AVX-512 stream (the 254-step inverse square root is 80 % of an encoding) and the four challenge
        // inversions of the task's four chunks behind ONE inversion (Montgomery's trick).  Problems 2c, 2c + 1 always share a task.
        mo.post8 = [&](size_t p0, int cnt) {
            ge5 pts[8];
            for (int l = 0; l < 8; l++) {
                if (l >= cnt) { pts[l] = h51::identity(); continue; }
                size_t p = p0 + (size_t)l, c = p >> 1;
                if (ip_included) { pts[l] = res[p]; continue; }
                if (cq_ready) { pts[l] = h51::gadd(res[p], cq[p]); continue; }
                sc cx = h_canon(sum_partials(h_ip + c * nblkI * 2, nblkI, 2, p & 1));
                pts[l] = h51::gadd(res[p], h_fixed_mul(C.ht.B5, h_mul(cx, w[c])));
            }
            uint8_t enc[8][32];
            h8::encode8(enc, pts);
            const int nch = cnt / 2;
            sc um[4], pre[4];
            for (int k = 0; k < nch; k++) {
                size_t c = (p0 >> 1) + (size_t)k;
                uint8_t *o = proofs_out[c] + 7 * 32 + 64 * round;
                memcpy(o, enc[2 * k], 32); memcpy(o + 32, enc[2 * k + 1], 32);
                tr[c].append("L", o, 32); tr[c].append("R", o + 32, 32);
                um[k] = h_mont(tr[c].challenge_scalar("u"));
                pre[k] = k ? sc_montmul(pre[k - 1], um[k]) : um[k];
            }
            sc uim[4];
            if (nch && !sc_iszero(h_canon(pre[nch - 1]))) {
                sc inv = h51::sc_invert_mont_fast(pre[nch - 1]);
                for (int k = nch - 1; k >= 1; k--) { uim[k] = sc_montmul(inv, pre[k - 1]); inv = sc_montmul(inv, um[k]); }
                uim[0] = inv;
            } else
                for (int k = 0; k < nch; k++) uim[k] = h51::sc_invert_mont_fast(um[k]);      // a zero challenge (probability 2^-252): one by one, as post does
            for (int k = 0; k < nch; k++) {
                size_t c = (p0 >> 1) + (size_t)k;
                h_round[2 * c] = um[k]; h_round[2 * c + 1] = uim[k];
                h_cp[c].pend_u[pu[c].size()] = um[k]; h_cp[c].pend_ui[pu[c].size()] = uim[k];
                pu[c].push_back(um[k]); pui[c].push_back(uim[k]);
            }
        };
        msm_run(C, pr, 2 * n_g, res, mo);
        mark("round msm", (long)(2 * n_g));
        bool last = (round + 1 == lgN);
        // the fold of a, b by this challenge happens inside the next round's k_ipp_round; only the old three-kernel path and the
        // last round (whose result is the proof's final a, b) fold here
        if ((last && !ab_on_host) || !(merged && ipp_fused))
            ROFL_LAUNCH(k_ipp_fold_ab, grid1(nh, (u32)P), dim3(TPB), 0, C.stream, (u32)nh, d_cp, (const sc *)C.h_round.dev<sc>(2 * P), r, a, b, N);
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
            // width-w NAF over the odd multiples that were built since the previous fold (all chunks' sources are one contiguous array)
            const bool use_w = !use_tab && mult_ready && unit && nsrc <= 64 && mult_base == cur[0] && mult_count == 2 * P * n_g;
            std::vector<std::vector<u32>> evs(use_w ? 2 * P : 0);
            FoldTabCfg fc = gens.fc();
            size_t dstride = use_tab ? (size_t)fc.np * FOLD_TAB_DIGITS : 256;
            th = now_ms();
            const size_t dbytes = use_tab ? 2 : 1;      // table folds: width-9 NAF digits reach +-255 (int16_t); plain NAF: int8_t
            int8_t *h_dig = C.h_fdig.as<int8_t>(2 * P * nsrc * dstride * dbytes);      // the fold's own pinned staging: nothing else writes them while its copies are queued
            int16_t *h_dig16 = reinterpret_cast<int16_t *>(h_dig);
            memset(h_dig, 0, 2 * P * nsrc * dstride * dbytes);
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
                            int t1 = wnaf_u64(h_dig16 + (((2 * c) * nsrc + h) * fc.np + pc) * FOLD_TAB_DIGITS, piece(gc), fc.w);
                            int t2 = wnaf_u64(h_dig16 + (((2 * c + 1) * nsrc + h) * fc.np + pc) * FOLD_TAB_DIGITS, piece(hc), fc.w);
                            top = std::max(top, std::max(t1, t2));
                        }
                    } else if (use_w) {
                        int8_t dg[2][256];
                        int t1 = sc_wnaf(dg[0], gc, (unsigned)fold_wnaf), t2 = sc_wnaf(dg[1], hc, (unsigned)fold_wnaf);
                        for (int sd = 0; sd < 2; sd++)
                            for (int b = 0; b <= (sd ? t2 : t1); b++)
                                if (int d = dg[sd][b]) evs[2 * c + sd].push_back(FOLD_EV(b, h, (u32)((d < 0 ? -d : d) - 1) >> 1, d < 0));
                        top = std::max(top, std::max(t1, t2));
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
            int8_t *d_dig = C.naf.as<int8_t>(2 * P * nsrc * dstride * dbytes);
            HIPCHK(hipMemcpyAsync(d_dig, h_dig, 2 * P * nsrc * dstride * dbytes, hipMemcpyHostToDevice, C.stream));
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
                double cst = 1.0 + (use_tab ? eff * fc.np / (fc.w + 1.0) : use_w ? eff / (fold_wnaf + 1.0) : eff / 3.0), lo_t = 0, hi_t = (top + 1) * cst + top + 1;
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
                if (C.tm.enabled) { size_t tot_d = 2 * P * nsrc * dstride; if (use_w) { for (auto &e : evs) nz += e.size(); } else for (size_t q = 0; q < tot_d; q++) nz += use_tab ? h_dig16[q] != 0 : h_dig[q] != 0; }
                // algorithmic work per output: the non-zero digits of its problem (mixed additions) and ONE chain of top+1 doublings
                // (the K-1 redundant chains of a segmented launch buy latency, they are not work)
                uint64_t fold_muls = (nz * 7 / (2 * P) + (uint64_t)(top + 1) * 8 + 7) * (uint64_t)(2 * P * n_new);
                HeavyScope heavy(C, C.stream, 2 * P * n_new >= ((size_t)1 << 15)); hipStream_t fst = heavy.run;      // big folds at the lowest priority while calls share the device (host_rt.hpp)
                KSpan ks_fold(C.tm, fst, use_tab ? ROFL_TK_FOLD_TAB : ROFL_TK_FOLD, fold_muls, (uint64_t)2 * P * n_g * 32 + (uint64_t)2 * P * n_new * 32);
                // A big first fold leaves its outputs in extended coordinates and k_niels_batch converts them, eight per inversion, on the side
                // stream while the next round's k_ipp_round and sort kernels run; that round's first point-reading kernel waits (pts_ready).
                static const bool defer_on = !(knob("ROFL_FOLD_DEFER") && atoi(knob("ROFL_FOLD_DEFER")) == 0);
                const bool defer = use_tab && defer_on && 2 * P * n_new >= ((size_t)1 << 17);
                ge *ext = defer ? C.foldext.as<ge>(2 * P * n_new) : nullptr;
                static const bool tab_ev = !(knob("ROFL_FOLD_TAB_EV") && atoi(knob("ROFL_FOLD_TAB_EV")) == 0);
                if (use_tab && tab_ev && unit && nsrc <= 64 && (size_t)fc.np * fc.e <= 4095) {
                    // the table fold as an event list (bit, source, slice, sign), highest bit first: no scan over the ~1 800 mostly-zero digit
                    // slots of a chain, and the operand of the next addition is in flight while the current one runs
                    std::vector<std::vector<u32>> tev(2 * P);
                    C.pool->run(2 * P, [&](size_t q) {
                        std::vector<u32> &e = tev[q];
                        for (u32 h = 1; h < nsrc; h++)
                            for (u32 pc = 0; pc < fc.np; pc++) {
                                const int16_t *dd = h_dig16 + ((q * nsrc + h) * fc.np + pc) * FOLD_TAB_DIGITS;
                                for (int b = 0; b < FOLD_TAB_DIGITS; b++) if (int d = dd[b]) e.push_back(FOLD_EV(b, h, pc * fc.e + ((u32)((d < 0 ? -d : d) - 1) >> 1), d < 0));
                            }
                        std::stable_sort(e.begin(), e.end(), [](u32 x, u32 y) { return (x & 511u) > (y & 511u); });
                    });
                    size_t tot_ev = 0; for (auto &e : tev) tot_ev += e.size();
                    u32 *h_ev = C.h_fev.as<u32>(tot_ev + 4 + (2 * P * sizeof(FoldWProb) + 3) / 4);
                    const size_t wp_off = (tot_ev + 2) & ~(size_t)1;
                    FoldWProb *h_wp = reinterpret_cast<FoldWProb *>(h_ev + wp_off);
                    size_t off = 0;
                    for (size_t q = 0; q < 2 * P; q++) {
                        std::vector<u32> &e = tev[q];
                        FoldWProb &w = h_wp[q];
                        w.src = tbl + h_ftp[q].src_off; w.dst = h_ftp[q].dst; w.tab_off = h_ftp[q].src_off; w.ev_off = (u32)off; w.n_ev = (u32)e.size();
                        for (u32 k = 0; k < FOLD_MAXSEG; k++) {
                            const int hi = seg.lo[std::min<u32>(k + 1, FOLD_MAXSEG)] - 1;
                            u32 j = 0; while (j < e.size() && (int)(e[j] & 511u) > hi) j++;
                            w.seg_start[k] = j;
                        }
                        memcpy(h_ev + off, e.data(), e.size() * sizeof(u32)); off += e.size();
                    }
                    u32 *d_ev = C.fold_ev.as<u32>(tot_ev + 4 + (2 * P * sizeof(FoldWProb) + 3) / 4);
                    HIPCHK(hipMemcpyAsync(d_ev, h_ev, wp_off * sizeof(u32) + 2 * P * sizeof(FoldWProb), hipMemcpyHostToDevice, fst));
                    // slice s of the table = tbl + s * stride; slice 0 = the generators themselves (the kernel reads table slices as tab + (s - 1) * stride)
                    ROFL_LAUNCH(k_fold_gens_w, grid, block, (K - 1) * 64 * sizeof(ge), fst, (u32)n_new, seg, reinterpret_cast<const FoldWProb *>(d_ev + wp_off),
                                       (const u32 *)d_ev, (const niels *)(tbl + (size_t)(2 * N)), (size_t)(2 * N), ext);
                } else if (use_tab) {
                    ROFL_LAUNCH(k_fold_gens_tab, grid, block, (K - 1) * 64 * sizeof(ge), fst, (u32)n_new, nsrc, seg, fc, tbl, (size_t)(2 * N),
                                       (const FoldTabProb *)d_fpv, reinterpret_cast<const int16_t *>(d_dig), unit, ext);
                }
                if (use_tab) {
                  heavy.end();      // (C.stream continues behind the fold)
                  if (defer) {
                    if (!C.ev_norm) { HIPCHK(hipEventCreateWithFlags(&C.ev_norm, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&C.ev_norm0, hipEventDisableTiming)); }
                    HIPCHK(hipEventRecord(C.ev_norm0, C.stream));
                    HIPCHK(hipStreamWaitEvent(C.stream2, C.ev_norm0, 0));
                    const size_t tot = 2 * P * n_new;
                    ROFL_LAUNCH(k_niels_batch, grid1((tot + NB_BATCH - 1) / NB_BATCH), dim3(TPB), 0, C.stream2, (u32)tot, (const ge *)ext, gnew);
                    HIPCHK(hipEventRecord(C.ev_norm, C.stream2));
                    pts_pending = true;
                  }
                }
                else if (use_w) {
                    // events of every problem, highest bit first; a segment starts at its first event at or below its top bit
                    size_t tot_ev = 0; for (auto &e : evs) tot_ev += e.size();
                    u32 *h_ev = C.h_fev.as<u32>(tot_ev + 4 + (2 * P * sizeof(FoldWProb) + 3) / 4);
                    FoldWProb *h_wp = reinterpret_cast<FoldWProb *>(h_ev + ((tot_ev + 2) & ~(size_t)1));
                    size_t off = 0;
                    for (size_t q = 0; q < 2 * P; q++) {
                        std::vector<u32> &e = evs[q];
                        std::stable_sort(e.begin(), e.end(), [](u32 x, u32 y) { return (x & 511u) > (y & 511u); });
                        FoldWProb &w = h_wp[q];
                        w.src = h_fp[q].src; w.dst = h_fp[q].dst; w.tab_off = (u32)(h_fp[q].src - mult_base); w.ev_off = (u32)off; w.n_ev = (u32)e.size();
                        for (u32 k = 0; k < FOLD_MAXSEG; k++) {
                            const int hi = seg.lo[std::min<u32>(k + 1, FOLD_MAXSEG)] - 1;
                            u32 j = 0; while (j < e.size() && (int)(e[j] & 511u) > hi) j++;
                            w.seg_start[k] = j;
                        }
                        memcpy(h_ev + off, e.data(), e.size() * sizeof(u32)); off += e.size();
                    }
                    u32 *d_ev = C.fold_ev.as<u32>(tot_ev + 4 + (2 * P * sizeof(FoldWProb) + 3) / 4);
                    const size_t wp_off = (tot_ev + 2) & ~(size_t)1;
                    HIPCHK(hipMemcpyAsync(d_ev, h_ev, (wp_off) * sizeof(u32) + 2 * P * sizeof(FoldWProb), hipMemcpyHostToDevice, fst));
                    HIPCHK(hipStreamWaitEvent(fst, C.ev_mult, 0));      // the table was built beside the last rounds
                    ROFL_LAUNCH(k_fold_gens_w, grid, block, (K - 1) * 64 * sizeof(ge), fst, (u32)n_new, seg, reinterpret_cast<const FoldWProb *>(d_ev + wp_off),
                                       (const u32 *)d_ev, (const niels *)C.fmul_tab.p, mult_count, (ge *)nullptr);
                }
                else if (nsrc == 4 && unit && fold_regs)      // three scalar-carrying sources, kept in registers
                    ROFL_LAUNCH(k_fold_gens4, grid, block, (K - 1) * 64 * sizeof(ge), fst, (u32)n_new, seg, (const FoldProb *)d_fpv, d_dig);
                else
                    ROFL_LAUNCH(k_fold_gens, grid, block, (K - 1) * 64 * sizeof(ge), fst, (u32)n_new, nsrc, seg, (const FoldProb *)d_fpv, d_dig, unit);
            }
            if (C.tm.enabled) { HIPCHK(hipEventRecord(e1, C.stream)); C.tm.fold_ev.push_back({e0, e1}); C.tm.t.fold_launches++; { char tg[96]; snprintf(tg, sizeof tg, "fold n_g=%zu nsrc=%u tab=%d", n_g, nsrc, (int)use_tab); C.tm.fold_tag.push_back(tg); } C.tm.t.fold_point_reads += (uint64_t)2 * P * n_g; }
            HIPCHK(hipMemcpyAsync(d_cp, h_cp, sizeof(ChunkParams) * P, hipMemcpyHostToDevice, C.stream));   // gscale / hscale
            // no sync here: the next round's launches queue up behind the fold on the same stream (their enqueue cost hides under it); the
            // pinned digit / problem staging buffers are not written again before the next fold, at least two synchronised rounds away
            if (ptrace) C.sync();      // the phase trace wants the fold's own wall time
            mark("fold", (long)n_new);
            for (size_t c = 0; c < P; c++) { cur[c] = gnew + c * 2 * n_new; pu[c].clear(); pui[c].clear(); }
            mult_ready = false;
            {   // will there be another fold, fold_t rounds from here?  Then its sources' odd multiples are built now, off the critical path.
                const size_t n_next = n_new >> C.fold_t;
                const bool next_pays = n_next >= C.fold_min || (n_next >= 64 && 2 * P * n_next >= 8 * C.fold_min);
                const size_t cnt = 2 * P * n_new;
                if (mult_E > 1 && (size_t)round + C.fold_t + 1 < lgN && next_pays && ((size_t)1 << C.fold_t) <= 64 && cnt <= ((size_t)1 << 22)) {
                    if (!C.stream3) HIPCHK(hipStreamCreateWithFlags(&C.stream3, hipStreamNonBlocking));
                    if (!C.ev_mult) { HIPCHK(hipEventCreateWithFlags(&C.ev_mult, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&C.ev_mult0, hipEventDisableTiming)); }
                    ge *mext = C.fmul_ext.as<ge>((mult_E - 1) * cnt);
                    niels *mtab = C.fmul_tab.as<niels>((mult_E - 1) * cnt);
                    // the points must be there in affine form: after the side-stream conversion of a deferred first fold, or after the fold kernel itself
                    if (pts_pending) HIPCHK(hipStreamWaitEvent(C.stream3, C.ev_norm, 0));
                    else { HIPCHK(hipEventRecord(C.ev_mult0, C.stream)); HIPCHK(hipStreamWaitEvent(C.stream3, C.ev_mult0, 0)); }
                    ROFL_LAUNCH(k_odd_multiples, grid1(cnt), dim3(TPB), 0, C.stream3, (u32)cnt, mult_E, (const niels *)gnew, mext);
                    const size_t tot = (mult_E - 1) * cnt;
                    ROFL_LAUNCH(k_niels_batch, grid1((tot + NB_BATCH - 1) / NB_BATCH), dim3(TPB), 0, C.stream3, (u32)tot, (const ge *)mext, mtab);
                    HIPCHK(hipEventRecord(C.ev_mult, C.stream3));
                    mult_ready = true; mult_used = true; mult_base = gnew; mult_count = cnt;
                }
            }
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
