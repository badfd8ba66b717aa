// The verifier of librofl_zk.so (bulletproofs RangeProof::verify_multiple, one random-weighted check per client): verify_chunks.
// Included by rofl_zk.hip inside its anonymous namespace, after host_prover.hpp.
#pragma once

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
//
// Inputs that arrive while the transcripts are being hashed (a batch of clients whose commitments are decoded group by group):
// ready[k] = (end proof, event): the V encodings and status words of the proofs below `end` are complete once `ev` has fired.
// h_status[i] & 4: client i (proofs_per_client proofs each) has a commitment that does not decode -- its proofs leave the shared checks.
struct VerifyReady { size_t end; hipEvent_t ev; };
struct VerifyInputs { const std::vector<VerifyReady> *ready; const u32 *h_status; size_t proofs_per_client; };
//
// `group` consecutive proofs form a UNIT (a client's chunks: the reference returns one bool per client) that is checked as one batch:
// sum_c rho_c * (check_c) == 0 with random weights rho_c, so the generator terms of its proofs share one MSM.  Every proof of a unit gets
// the unit's verdict.  `hier` (rofl_set_option("verify_batch", 2), the server role: server.rs:474-484 fails the whole round on one bad
// client, so the common case is "everything verifies") first checks ALL units as one batch -- one generator MSM for the whole call instead
// of one per client -- and only when that fails looks closer: groups of ~sqrt(units) units, then the units of the groups that failed, so
// that every unit still gets its own verdict, identical to the per-unit check's.  rho_index[c] (default c_index[c]) keys proof c's weight:
// proofs that share a check need distinct ones.  skip[c] != 0 takes proof c out of every check (weight 0) and fails its unit -- a batch
// member the caller already knows to be malformed must not make its neighbours pay for the closer look.
int verify_chunks(Ctx &C, const char *label, size_t gens_capacity, size_t P, size_t n, size_t m, const uint8_t *proofs, size_t plen,
                  const uint8_t *h_V, const niels *d_Vniels, const uint8_t seed[32], const u64 *c_index, int *ok, size_t group = 1,
                  const sc *v_shift = nullptr, const u64 *v_real = nullptr, const u64 *rho_index = nullptr, bool hier = false,
                  const char *skip = nullptr, const VerifyInputs *vin = nullptr) {
    for (size_t c = 0; c < P; c++) ok[c] = 0;
    static const bool vtrace = knob("ROFL_TRACE") && atoi(knob("ROFL_TRACE")) >= 2;
    double vt0 = now_ms(), vtl = vt0;
    auto vmark = [&](const char *what) { if (!vtrace) return; double t = now_ms(); fprintf(stderr, "[rofl-trace verify] %-14s +%.3f ms  (t=%.3f)\n", what, t - vtl, t - vt0); vtl = t; };
    if (group == 0 || P % group) group = 1;
    const size_t units = P / group;
    if (units < 2) hier = false;
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
    GensPin gens = get_gens(C, n, m, GENS_VERIFY);
    vmark("gens");
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
    ROFL_LAUNCH(k_decode, grid1(P * (4 + 2 * lg)), dim3(TPB), 0, C.stream, (u32)(P * (4 + 2 * lg)), (u32)(P * (4 + 2 * lg)), d_auxc, (const niels *)nullptr, d_auxn, (uint8_t *)nullptr, status);
    vmark("decode launch");
    double th = now_ms();
    std::atomic<u64> tr_prefix_ns{0}, tr_rest_ns{0};      // slowest task's share (ROFL_TRACE=2)
    // the transcript prefix (the m commitments of a chunk; sequential sponge): with enough proofs eight chunks share one AVX-512
    // instruction stream (keccak_x8.hpp), otherwise one chunk per task
    static const bool x8_on = k8::available() && !(knob("ROFL_MERLIN_X8") && atoi(knob("ROFL_MERLIN_X8")) == 0);
    auto prefix = [&](Merlin *const t[8], const size_t c[8], int cnt) {
        const uint8_t *msg[8];
        for (int l = 0; l < cnt; l++) {
            t[l]->append("dom-sep", (const uint8_t *)"rangeproof v1", 13);
            t[l]->append_u64("n", n); t[l]->append_u64("m", m);
            msg[l] = h_V + c[l] * m * 32;
        }
        if (x8_on && cnt >= 5) k8::append32_run_x8(t, cnt, 'V', msg, m);
        else for (int l = 0; l < cnt; l++) t[l]->append32_run('V', msg[l], m);
    };
    auto transcript = [&](size_t c, Merlin &t) {
        const uint8_t *p = proofs + c * plen; const uint8_t *ipp = p + 7 * 32;
        // validate_and_append_point rejects the identity encoding
        bool bad = false;
        for (int i = 0; i < 4; i++) if (!memcmp(p + 32 * i, zero32, 32)) bad = true;
        for (size_t k = 0; k < 2 * lg; k++) if (!memcmp(ipp + 32 * k, zero32, 32)) bad = true;
        if (bad || (skip && skip[c]) || (hier && vin && (vin->h_status[c / vin->proofs_per_client] & 4u))) { dead[c] = 1; }
        t.append("A", p, 32); t.append("S", p + 32, 32);
        sc y = t.challenge_scalar("y"), z = t.challenge_scalar("z");
        t.append("T_1", p + 64, 32); t.append("T_2", p + 96, 32);
        sc x = t.challenge_scalar("x");
        sc t_x = sc_frombytes(p + 128), t_x_bl = sc_frombytes(p + 160), e_bl = sc_frombytes(p + 192);
        t.append("t_x", p + 128, 32); t.append("t_x_blinding", p + 160, 32); t.append("e_blinding", p + 192, 32);
        sc w = t.challenge_scalar("w");
        sc cc = verifier_c(seed, c_index[c]);
        sc rho = (group > 1 || hier) ? verifier_c(seed, (rho_index ? rho_index[c] : c_index[c]) | (1ULL << 62)) : sc_one_plain();
        if (dead[c]) rho = sc_zero();      // out of every shared check (all of its scalars below become zero); its unit fails through dead[]
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
    };
    // proofs [c0, c1): eight per task when the lanes of the SIMD sponge can be filled and the pool still gets a task per thread
    auto hash_range = [&](size_t c0, size_t c1) {
        if (c1 <= c0) return;
        const size_t cnt = c1 - c0, per = (x8_on && cnt >= 32) ? 8 : 1;
        C.pool->run((cnt + per - 1) / per, [&](size_t b) {
            Merlin tr[8] = {Merlin(label, strlen(label)), Merlin(label, strlen(label)), Merlin(label, strlen(label)), Merlin(label, strlen(label)),
                            Merlin(label, strlen(label)), Merlin(label, strlen(label)), Merlin(label, strlen(label)), Merlin(label, strlen(label))};
            Merlin *t[8]; size_t cs[8]; int k = 0;
            for (size_t c = c0 + b * per; c < std::min(c1, c0 + (b + 1) * per); c++, k++) { t[k] = &tr[k]; cs[k] = c; }
            const double ta = vtrace ? now_ms() : 0;
            prefix(t, cs, k);
            const double tb = vtrace ? now_ms() : 0;
            for (int l = 0; l < k; l++) transcript(cs[l], tr[l]);
            if (vtrace) {
                const double tc = now_ms();
                u64 a = (u64)((tb - ta) * 1e6), r = (u64)((tc - tb) * 1e6), o;
                o = tr_prefix_ns.load(); while (a > o && !tr_prefix_ns.compare_exchange_weak(o, a)) {}
                o = tr_rest_ns.load(); while (r > o && !tr_rest_ns.compare_exchange_weak(o, r)) {}
            }
        });
    };
    if (vin && vin->ready && !vin->ready->empty()) {      // group by group, as the encodings arrive (the device is still decoding the later groups)
        size_t c0 = 0;
        for (const VerifyReady &r : *vin->ready) {
            C.wait_event(r.ev);
            const size_t c1 = std::min(r.end, P);
            hash_range(c0, c1);
            c0 = c1;
        }
    } else hash_range(0, P);
    C.tm.t.host_ms += now_ms() - th;
    vmark("transcripts");
    if (vtrace) fprintf(stderr, "[rofl-trace verify] slowest task: prefix %.3f ms, challenges and scalars %.3f ms\n", tr_prefix_ns.load() * 1e-6, tr_rest_ns.load() * 1e-6);
    HIPCHK(hipMemcpyAsync(d_cp, h_cp, sizeof(ChunkParams) * P, hipMemcpyHostToDevice, C.stream));
    PowTabs *d_pt = C.powtabs.as<PowTabs>(P);
    ROFL_LAUNCH(k_pow_tables, dim3((4 * PT_L * PT_E + TPB - 1) / TPB, (u32)P), dim3(TPB), 0, C.stream, d_cp, d_pt, (u32)lg, 1);
    // the block-structured form of the generator scalars (k_verify_scalars2) takes whole blocks of 512 indices, all of them inside the tables
    const bool vs2 = N >= 512 && lg <= 3 * PT_W && n <= 64;
    VTabs *d_vt = nullptr;
    if (vs2) {
        d_vt = C.vtabs.as<VTabs>(P);
        ROFL_LAUNCH(k_vtabs, dim3((9 * PT_E + 64 + 255) / 256, (u32)P), dim3(256), 0, C.stream, d_cp, (const PowTabs *)d_pt, C.d_two_pow, d_vt, (u32)n, lg2u(n));
    }
    // aux arrays: per proof [m commitments | 4 + 2 lg proof points] and their scalars.  The closer look of `hier` checks runs of units whose
    // last one may be shorter: the arrays end in one group's worth of zero scalars, so every MSM problem of a launch can take the same length.
    const size_t gB = hier ? (size_t)std::ceil(std::sqrt((double)units)) : 1;      // units per group of the middle level
    const size_t pad = hier ? gB * group : 0;
    niels *aux_pts = C.aux_pts.as<niels>((P + pad) * naux);
    sc *aux_scal = C.aux_scal.as<sc>((P + pad) * naux);
    if (pad) HIPCHK(hipMemsetAsync(aux_scal + P * naux, 0, sizeof(sc) * pad * naux, C.stream));
    ROFL_LAUNCH(k_vscalars, grid1(m, (u32)P), dim3(TPB), 0, C.stream, (u32)m, d_cp, (const PowTabs *)d_pt, aux_scal, naux);
    {   // three strided copies for all proofs
        const size_t na2 = 4 + 2 * lg;
        HIPCHK(hipMemcpy2DAsync(aux_pts, naux * sizeof(niels), d_Vniels, m * sizeof(niels), m * sizeof(niels), P, hipMemcpyDeviceToDevice, C.stream));
        HIPCHK(hipMemcpy2DAsync(aux_pts + m, naux * sizeof(niels), d_auxn, na2 * sizeof(niels), na2 * sizeof(niels), P, hipMemcpyDeviceToDevice, C.stream));
        HIPCHK(hipMemcpy2DAsync(aux_scal + m, naux * sizeof(sc), h_auxs, na2 * sizeof(sc), na2 * sizeof(sc), P, hipMemcpyHostToDevice, C.stream));
    }
    u32 *h_stat = C.h_misc2.as<u32>(4);
    HIPCHK(hipMemcpyAsync(h_stat, status, 4, hipMemcpyDeviceToHost, C.stream));      // k_decode's verdict on the proof points
    // One pass over a list of groups (start proof, proof count; equal counts, except that a group ending at P may be shorter): the generator
    // MSM (2N terms per group, fixed-base) and the proof-point MSM (commitments, A, S, T, L, R) of every group queue back to back behind one wait.
    struct Grp { u32 start, count; };
    auto check = [&](const std::vector<Grp> &groups, std::vector<int> &verdict) {
        const size_t ng = groups.size();
        size_t maxc = 0; for (auto &g : groups) maxc = std::max<size_t>(maxc, g.count);
        sc *gh = C.SL.as<sc>(ng * 2 * N);
        Grp *d_grp = C.vgroups.as<Grp>(ng);
        Grp *h_grp = C.h_vgrp.as<Grp>(ng);
        memcpy(h_grp, groups.data(), sizeof(Grp) * ng);
        HIPCHK(hipMemcpyAsync(d_grp, h_grp, sizeof(Grp) * ng, hipMemcpyHostToDevice, C.stream));
        uint64_t vs_proofs = 0; for (auto &g : groups) vs_proofs += g.count;
        { KSpan ks_vs(C.tm, C.stream, ROFL_TK_VERIFY_SCALARS, 0, (uint64_t)ng * 2 * N * 32 + vs_proofs * sizeof(PowTabs));
          if (vs2) {
              // enough blocks for the chip: a group's proofs are split into slices when N / 512 x groups is small
              size_t nsl = std::min<size_t>(maxc, std::max<size_t>(1, 1024 / ((N / 512) * ng)));
              const u32 per = (u32)((maxc + nsl - 1) / nsl); nsl = (maxc + per - 1) / per;
              sc *dst = nsl > 1 ? C.vspart.as<sc>(ng * nsl * 2 * N) : gh;
              ROFL_LAUNCH(k_verify_scalars2, dim3((unsigned)(N / 512), (u32)ng, (u32)nsl), dim3(256), 0, C.stream, (u32)n, lg2u(n), (u32)m, reinterpret_cast<const uint2 *>(d_grp), d_cp, (const VTabs *)d_vt, dst, per);
              if (nsl > 1) ROFL_LAUNCH(k_vs_sum, grid1(2 * N, (u32)ng), dim3(TPB), 0, C.stream, (u32)(2 * N), (u32)nsl, (const sc *)dst, gh);
          }
          else ROFL_LAUNCH(k_verify_scalars, grid1(N, (u32)ng), dim3(TPB), 0, C.stream, (u32)n, (u32)m, (u32)lg, reinterpret_cast<const uint2 *>(d_grp), d_cp, (const PowTabs *)d_pt, C.d_two_pow, gh); }
        std::vector<MsmProb> pr(ng), prB(ng); std::vector<ge5> resA, resB;
        for (size_t g = 0; g < ng; g++) pr[g] = MsmProb{tbl, gh + g * 2 * N};
        for (size_t g = 0; g < ng; g++) prB[g] = MsmProb{aux_pts + (size_t)groups[g].start * naux, aux_scal + (size_t)groups[g].start * naux};
        C.tm.t.msm_terms += ng * 2 * N + ng * maxc * naux;
        // (one problem per group with every window in its own bucket set: the 15-bit layout's smaller arrays win here whenever it exists)
        { MsmOpt mo; if (wtab) { gens.fb_for(1000, &mo.fb_wtab, &mo.fb_c); mo.fb_stride = 2 * N; } msm_run2(C, pr, 2 * N, mo, resA, prB, maxc * naux, MsmOpt(), resB); }
        double th2 = now_ms();
        verdict.assign(ng, 0);
        for (size_t g = 0; g < ng; g++) {
            ge5 tot = h51::gadd(resA[g], resB[g]);
            sc b1 = sc_zero(), b2 = sc_zero();
            for (size_t c = groups[g].start; c < (size_t)groups[g].start + groups[g].count; c++) { b1 = sc_add(b1, sB[c]); b2 = sc_add(b2, sBb[c]); }
            tot = h51::gadd(tot, h_fixed_mul(C.ht.B5, b1));
            tot = h51::gadd(tot, h_fixed_mul(C.ht.Bb5, b2));
            verdict[g] = h51::is_identity_ristretto(tot) ? 1 : 0;
        }
        C.tm.t.host_ms += now_ms() - th2;
    };
    std::vector<int> unit_ok(units, 0), verdict;
    if (!hier) {
        std::vector<Grp> all(units);
        for (size_t u = 0; u < units; u++) all[u] = Grp{(u32)(u * group), (u32)group};
        check(all, verdict);
        unit_ok = verdict;
    } else {
        check({Grp{0u, (u32)P}}, verdict);
        vmark("batch check");
        if (verdict[0]) unit_ok.assign(units, 1);
        else {
            std::vector<size_t> suspects;
            if (units > 3) {
                std::vector<Grp> mid;
                for (size_t u0 = 0; u0 < units; u0 += gB) mid.push_back(Grp{(u32)(u0 * group), (u32)(std::min(gB, units - u0) * group)});
                check(mid, verdict);
                for (size_t g = 0; g < mid.size(); g++)
                    for (size_t u = g * gB; u < std::min(units, (g + 1) * gB); u++) { if (verdict[g]) unit_ok[u] = 1; else suspects.push_back(u); }
            } else for (size_t u = 0; u < units; u++) suspects.push_back(u);
            if (!suspects.empty()) {
                std::vector<Grp> one(suspects.size());
                for (size_t k = 0; k < suspects.size(); k++) one[k] = Grp{(u32)(suspects[k] * group), (u32)group};
                check(one, verdict);
                for (size_t k = 0; k < suspects.size(); k++) unit_ok[suspects[k]] = verdict[k];
            }
        }
    }
    vmark("msm");
    const u32 h_status = *h_stat;
    for (size_t u = 0; u < units; u++) {
        bool any_dead = false;
        for (size_t c = u * group; c < (u + 1) * group; c++) any_dead |= dead[c] != 0;
        for (size_t c = u * group; c < (u + 1) * group; c++) ok[c] = (unit_ok[u] && !any_dead) ? 1 : 0;
    }
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
