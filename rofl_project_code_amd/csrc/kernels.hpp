// HIP kernels for the range-proof hot path (gfx950 only).  See DESIGN.md for the kernel table.
//
// Conventions: blockIdx.y = "problem" (one Bulletproof chunk, or one MSM problem).
// Scalars in device arrays are in Montgomery form unless a name ends with _canon.
#pragma once
#include <type_traits>
#include "fe32.hpp"
#include "fe26.hpp"
#include "quad26.hpp"
#include "keccak.hpp"

namespace rofl {

#define TPB 256

// Kernel groups: the library is built from one translation unit per group (build.py compiles them in parallel) plus the host
// translation unit, which only sees the prototypes (kernel_protos.hpp, generated from this file).  ROFL_KGROUP: 0 = every kernel
// is defined here (single-TU builds), g > 0 = only group g, -1 = none.
#ifndef ROFL_KGROUP
#define ROFL_KGROUP 0
#endif
#define ROFL_KG(g) (ROFL_KGROUP == 0 || ROFL_KGROUP == (g))

// ---------------------------------------------------------------- small helpers
// A pointer read out of a descriptor in memory is "generic" to the compiler and its accesses become flat_*; every
// such pointer here is device memory, and saying so turns them into global_* accesses.
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) v4u g_uint4;
__device__ __forceinline__ niels gload_niels(const niels *p) {
    niels r;
    const g_uint4 *s = (const g_uint4 *)p;
    v4u *d = reinterpret_cast<v4u *>(&r);
#pragma unroll
    for (int i = 0; i < 6; i++) d[i] = s[i];
    return r;
}
__device__ __forceinline__ void gstore_niels(niels *p, const niels &v) {
    g_uint4 *d = (g_uint4 *)p;
    const v4u *s = reinterpret_cast<const v4u *>(&v);
#pragma unroll
    for (int i = 0; i < 6; i++) d[i] = s[i];
}
__device__ __forceinline__ sc gload_sc(const sc *p) {
    sc r;
    const g_uint4 *s = (const g_uint4 *)p;
    v4u *d = reinterpret_cast<v4u *>(&r);
    d[0] = s[0]; d[1] = s[1];
    return r;
}
__device__ __forceinline__ niels load_niels(const niels *p) {
    niels r;
    const uint4 *s = reinterpret_cast<const uint4 *>(p);
    uint4 *d = reinterpret_cast<uint4 *>(&r);
#pragma unroll
    for (int i = 0; i < 6; i++) d[i] = s[i];
    return r;
}
__device__ __forceinline__ void store_niels(niels *p, const niels &v) {
    uint4 *d = reinterpret_cast<uint4 *>(p);
    const uint4 *s = reinterpret_cast<const uint4 *>(&v);
#pragma unroll
    for (int i = 0; i < 6; i++) d[i] = s[i];
}
__device__ __forceinline__ ge load_ge(const ge *p) {
    ge r;
    const uint4 *s = reinterpret_cast<const uint4 *>(p);
    uint4 *d = reinterpret_cast<uint4 *>(&r);
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = s[i];
    return r;
}
__device__ __forceinline__ void store_ge(ge *p, const ge &v) {
    uint4 *d = reinterpret_cast<uint4 *>(p);
    const uint4 *s = reinterpret_cast<const uint4 *>(&v);
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = s[i];
}
__device__ __forceinline__ nd load_nd(const niels *p) { return nd_unpack(load_niels(p)); }
__device__ __forceinline__ nd gload_nd(const niels *p) { return nd_unpack(gload_niels(p)); }
// Window-table entry of the fixed-base MSM: the affine niels triple already in the register radix (3 x 10 limbs + 2 words
// of padding = 128 B, one cache line per gather; the 96-byte packed form straddles two 64-byte sectors just the same and
// costs 66 VALU instructions per addition to unpack).
#ifndef ROFL_ACC_LIST_CHUNK
#define ROFL_ACC_LIST_CHUNK 0
#endif
struct ndm { u32 v[32]; };
__device__ __forceinline__ nd gload_ndm(const ndm *p) {
    v4u w[8];
    const g_uint4 *s = (const g_uint4 *)p;
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = s[i];
    const u32 *f = reinterpret_cast<const u32 *>(w);
    nd r;
#pragma unroll
    for (int i = 0; i < 10; i++) { r.ypx.v[i] = f[i]; r.ymx.v[i] = f[10 + i]; r.t2d.v[i] = f[20 + i]; }
    return r;
}
__device__ __forceinline__ void store_ndm(ndm *p, const niels &q) {
    nd t = nd_unpack(q);
    u32 f[32];
#pragma unroll
    for (int i = 0; i < 10; i++) { f[i] = t.ypx.v[i]; f[10 + i] = t.ymx.v[i]; f[20 + i] = t.t2d.v[i]; }
    f[30] = 0; f[31] = 0;
    uint4 *d = reinterpret_cast<uint4 *>(p);
    const uint4 *s = reinterpret_cast<const uint4 *>(f);
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = s[i];
}
__device__ __forceinline__ gd load_gd(const ge *p) { return gd_unpack(load_ge(p)); }
__device__ __forceinline__ void store_gd(ge *p, const gd &v) { store_ge(p, gd_pack(v)); }
__device__ __forceinline__ sc load_sc(const sc *p) {
    sc r;
    const uint4 *s = reinterpret_cast<const uint4 *>(p);
    uint4 *d = reinterpret_cast<uint4 *>(&r);
    d[0] = s[0]; d[1] = s[1];
    return r;
}
// a scalar handed over the C ABI as raw bytes: reduced mod l if it is not canonical
__device__ __forceinline__ sc load_sc_reduced(const sc *p) { sc r = load_sc(p); if (sc_geq_l(r.v)) r = sc_from_mont(sc_to_mont(r)); return r; }
__device__ __forceinline__ void store_sc(sc *p, const sc &v) {
    uint4 *d = reinterpret_cast<uint4 *>(p);
    const uint4 *s = reinterpret_cast<const uint4 *>(&v);
    d[0] = s[0]; d[1] = s[1];
}

// x^e (Montgomery) from the table sq[b] = x^(2^b)
__device__ __forceinline__ sc sc_pow_tab(const sc *sq, u32 e) {
    sc acc = sc_one_mont();
    for (int b = 0; e; b++, e >>= 1)
        if (e & 1) acc = sc_montmul(acc, sq[b]);
    return acc;
}

// block-wide sum of NS scalars per thread; result valid in thread 0
template <int NS>
__device__ __forceinline__ void block_sum_sc(sc (&v)[NS], sc *lds /* TPB*NS */) {
    int t = threadIdx.x;
#pragma unroll
    for (int k = 0; k < NS; k++) lds[k * TPB + t] = v[k];
    __syncthreads();
    for (int s = TPB / 2; s > 0; s >>= 1) {
        if (t < s) {
#pragma unroll
            for (int k = 0; k < NS; k++) lds[k * TPB + t] = sc_add(lds[k * TPB + t], lds[k * TPB + t + s]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < NS; k++) v[k] = lds[k * TPB];
}

// ---------------------------------------------------------------- per-chunk parameter block
#define MAX_LG 32
struct NonceSeed { u64 w[4]; };
struct ChunkParams {
    sc y, z, zz, x, yinv;            // Montgomery
    sc ypow2[MAX_LG];                // y^(2^b)
    sc yinvpow2[MAX_LG];             // y^-(2^b)
    sc zpow2[MAX_LG];                // z^(2^b)
    sc u[MAX_LG], uinv[MAX_LG];      // IPP challenges (verify: all rounds; prove: current round)
    sc pend_u[MAX_LG], pend_ui[MAX_LG];   // prove: challenges not yet folded into the materialised generators
    sc gscale, hscale;               // prove: common factors kept out of the materialised generators
    sc a_fin, b_fin;                 // verify: ipp a, b
    sc c_zz;                         // verify: rho * c * z^2
    sc rz, ra, rb, rzz;              // verify: rho * z, rho * a, rho * b, rho * z^2 (rho = weight of this proof in its batch)
    u64 nonce_base;                  // index of this chunk's first nonce in its client's nonce space
    NonceSeed nonce_seed;            // mode 1: the client's seed
    const uint8_t *nonce_stream;     // mode 0: the client's explicit stream (device memory), nonce_stream_scalars wide scalars
    u64 nonce_stream_scalars;
    int nonce_mode;
};

// Radix-128 power tables of a chunk's challenges: x^k = T[0][k & 127] * T[1][(k >> 7) & 127] * T[2][(k >> 14) & 127] -- two
// multiplications per index instead of one per set bit (up to 18 at N = 2^18) in every per-slot kernel.  `s` is the verifier's
// IPP table: the product over the rounds of u_q or u_q^-1 selected by the index bits (its inverse = the same table at ~k).
#define PT_W 7
#define PT_E 128
#define PT_L 3
struct PowTabs { sc y[PT_L][PT_E], yinv[PT_L][PT_E], z[PT_L][PT_E], s[PT_L][PT_E]; };     // Montgomery form
#if ROFL_KG(4)
__global__ void __launch_bounds__(TPB) k_pow_tables(const ChunkParams *cp, PowTabs *pt, u32 lgN, int with_s) {
    u32 c = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 kind = t / (PT_L * PT_E), lvl = (t / PT_E) % PT_L, e = t % PT_E;
    if (kind >= 4 || (kind == 3 && !with_s)) return;
    const ChunkParams &P = cp[c];
    sc acc = sc_one_mont();
    for (u32 b = 0; b < PT_W; b++) {
        u32 p = PT_W * lvl + b;              // bit position of the index
        bool bit = (e >> b) & 1;
        if (kind == 3) {
            if (p < lgN) { u32 q = lgN - 1 - p; acc = sc_montmul(acc, bit ? load_sc(&P.u[q]) : load_sc(&P.uinv[q])); }
        } else if (bit && p < MAX_LG) {
            acc = sc_montmul(acc, load_sc(kind == 0 ? &P.ypow2[p] : kind == 1 ? &P.yinvpow2[p] : &P.zpow2[p]));
        }
    }
    sc *dst = kind == 0 ? &pt[c].y[lvl][e] : kind == 1 ? &pt[c].yinv[lvl][e] : kind == 2 ? &pt[c].z[lvl][e] : &pt[c].s[lvl][e];
    store_sc(dst, acc);
}
#endif
// x^e from a chunk's table; index bits above 21 (N > 2^21) fall back to the squarings table
__device__ __forceinline__ sc pt_pow(const sc (*tab)[PT_E], const sc *sq, u32 e) {
    sc acc = load_sc(&tab[0][e & (PT_E - 1)]);
    if (e >> PT_W) acc = sc_montmul(acc, load_sc(&tab[1][(e >> PT_W) & (PT_E - 1)]));
    if (e >> (2 * PT_W)) acc = sc_montmul(acc, load_sc(&tab[2][(e >> (2 * PT_W)) & (PT_E - 1)]));
    e >>= 3 * PT_W;
    for (int b = 3 * PT_W; e; b++, e >>= 1)
        if (e & 1) acc = sc_montmul(acc, sq[b]);
    return acc;
}

// ================================================================ K1: generators
// bulletproofs GeneratorsChain: SHAKE256("GeneratorsChain" || label5), 64 B per generator.
// One thread per (which, party): sequential XOF squeeze, n <= 64 generators.
#if ROFL_KG(2)
__global__ void __launch_bounds__(TPB) k_gens_xof(u32 n, u32 m, uint8_t *uni /* [2][m][n][64] */) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * m) return;
    u32 which = t / m, j = t % m;
    u64 st[25];
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = 0;
    // message: "GeneratorsChain" (15) || 'G'/'H' || u32le(j)   = 20 bytes, then 0x1F pad
    const char lab[16] = {'G', 'e', 'n', 'e', 'r', 'a', 't', 'o', 'r', 's', 'C', 'h', 'a', 'i', 'n', 0};
    u64 w0 = 0, w1 = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) w0 |= (u64)(uint8_t)lab[i] << (8 * i);
#pragma unroll
    for (int i = 0; i < 7; i++) w1 |= (u64)(uint8_t)lab[8 + i] << (8 * i);
    w1 |= (u64)(which ? 'H' : 'G') << 56;
    st[0] = w0; st[1] = w1;
    st[2] = (u64)j | (0x1FULL << 32);
    st[16] ^= 0x8000000000000000ULL;
    keccak_f1600(st);
    u64 *out = reinterpret_cast<u64 *>(uni + ((size_t)which * m + j) * n * 64);
    u32 pos = 0;   // word position in the 17-word rate
    for (u32 i = 0; i < n * 8; i++) {
        if (pos == 17) { keccak_f1600(st); pos = 0; }
        out[i] = st[pos++];
    }
}
#endif
// uniform bytes -> affine niels table [G(N) | H(N)]
#if ROFL_KG(2)
__global__ void __launch_bounds__(TPB) k_gens_map(u32 total, const uint8_t *uni, niels *tbl) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    ge p = ristretto_from_uniform(uni + (size_t)t * 64);
    store_niels(&tbl[t], ge_to_niels(p));
}
#endif

// Fold tables (see k_fold_gens_tab): slice (q * E + e) = (2e + 1) * 2^(PB q) * P for the NP = 256 / PB pieces of a
// scalar and the E = 2^(w-2) odd multiples of a width-w NAF; slice 0 is the plain generator table.
// One thread per (generator, piece); the E conversions to affine share one inversion.
struct FoldTabCfg { u32 pb, w, np, e; };
#define FOLD_TAB_MAXE 16
#if ROFL_KG(2)
__global__ void __launch_bounds__(TPB) k_gens_tables(u32 total, FoldTabCfg cfg, niels *tbl, size_t stride) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 g = t % total, q = t / total;
    if (q >= cfg.np) return;
    gd cur = gd_unpack(ge_from_niels(load_niels(&tbl[g])));
    for (u32 d = 0; d < q * cfg.pb; d++) cur = gd_double(cur);
    gd p2 = gd_double(cur);
    ge mult[FOLD_TAB_MAXE]; fd pref[FOLD_TAB_MAXE];
    gd pk = cur;                      // (2e + 1) * cur, walked in groups of FOLD_TAB_MAXE with one inversion per group
    for (u32 e0 = 0; e0 < cfg.e; e0 += FOLD_TAB_MAXE) {
        u32 cnt = cfg.e - e0 < FOLD_TAB_MAXE ? cfg.e - e0 : FOLD_TAB_MAXE;
        for (u32 j = 0; j < cnt; j++) {
            if (e0 + j) pk = gd_add(pk, p2);
            mult[j] = gd_pack(pk);
            pref[j] = j ? fd_mul(pref[j - 1], pk.Z) : pk.Z;
        }
        fd inv = fd_invert(pref[cnt - 1]);
        for (int j = (int)cnt - 1; j >= 0; j--) {
            gd m = gd_unpack(mult[j]);
            fd zi = j ? fd_mul(inv, pref[j - 1]) : inv;
            inv = fd_mul(inv, m.Z);
            u32 e = e0 + (u32)j;
            if (q || e) store_niels(&tbl[(size_t)(q * cfg.e + e) * stride + g], gd_to_niels_zinv(m, zi));
        }
    }
}
#endif

// ================================================================ nonces
// mode 1: scalar idx = bytes 64 (idx & 1) .. + 64 of SHAKE256("rofl-zk/nonce/v2" || seed || u64le(idx >> 1)) -- two wide scalars per block of
// the XOF, one Keccak-f per thread ; mode 0: explicit 64-byte stream.
// Reference draw order (bulletproofs party.rs): per party j: a_bl, s_bl, s_L[0..n), s_R[0..n);
// then per party: t1_bl, t2_bl.
#if ROFL_KG(4)
__global__ void __launch_bounds__(TPB) k_nonce_expand(u32 n, u32 m, const ChunkParams *cp, sc *sL, sc *sR, sc *party /* [chunk][4][m] */, sc *S_canon /* [chunk][2N] */) {
    u32 c = blockIdx.y;
    const u64 per = (u64)m * (2 * n + 4);
    // every chunk names its own nonce source: the chunks of a launch may belong to different clients (batched create)
    const u64 base = cp[c].nonce_base;
    const u64 blk = (base >> 1) + (u64)blockIdx.x * blockDim.x + threadIdx.x;      // thread = one XOF block = scalars 2 blk, 2 blk + 1
    if (2 * blk >= base + per) return;
    const int mode = cp[c].nonce_mode;
    const uint8_t *stream = cp[c].nonce_stream; const u64 stream_scalars = cp[c].nonce_stream_scalars;
    u64 st[25];
    if (mode == 1) {
        const u64 dom[2] = ROFL_NONCE_DOM;
        u64 sw[4] = {cp[c].nonce_seed.w[0], cp[c].nonce_seed.w[1], cp[c].nonce_seed.w[2], cp[c].nonce_seed.w[3]};
        shake256_seeded_block(st, dom, sw, blk);
    }
    const u64 first = (u64)m * (2 * n + 2);
    const size_t N = (size_t)n * m;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const u64 idx = 2 * blk + h;
        if (idx < base || idx >= base + per) continue;
        const u64 k = idx - base;
        sc lo, hi;
        if (mode == 1) {
#pragma unroll
            for (int i = 0; i < 4; i++) { lo.v[2 * i] = (u32)st[8 * h + i]; lo.v[2 * i + 1] = (u32)(st[8 * h + i] >> 32); hi.v[2 * i] = (u32)st[8 * h + 4 + i]; hi.v[2 * i + 1] = (u32)(st[8 * h + 4 + i] >> 32); }
        } else if (idx < stream_scalars) {
            const u32 *s = reinterpret_cast<const u32 *>(stream + idx * 64);
#pragma unroll
            for (int i = 0; i < 8; i++) { lo.v[i] = s[i]; hi.v[i] = s[8 + i]; }
        } else { lo = sc_zero(); hi = sc_zero(); }
        sc v = sc_from_wide_mont(lo, hi);       // Montgomery form for the polynomial kernels, canonical for the MSM: one wide reduction + one conversion
        sc vc = sc_from_mont(v);
        if (k < first) {
            u32 j = (u32)(k / (2 * n + 2)), r = (u32)(k % (2 * n + 2));
            if (r == 0) store_sc(&party[((size_t)c * 4 + 0) * m + j], v);
            else if (r == 1) store_sc(&party[((size_t)c * 4 + 1) * m + j], v);
            else if (r < 2 + n) { store_sc(&sL[c * N + (size_t)j * n + (r - 2)], v); store_sc(&S_canon[c * 2 * N + (size_t)j * n + (r - 2)], vc); }
            else { store_sc(&sR[c * N + (size_t)j * n + (r - 2 - n)], v); store_sc(&S_canon[c * 2 * N + N + (size_t)j * n + (r - 2 - n)], vc); }
        } else {
            u64 q = k - first; u32 j = (u32)(q / 2);
            store_sc(&party[((size_t)c * 4 + 2 + (q & 1)) * m + j], v);
        }
    }
}
#endif

// ================================================================ K2: quantize + shift
// conversion32.rs:11-18 f32_to_scalar, range_proof_vec/mod.rs:27-43.  status bits: 1 out-of-range, 2 NaN.
#if ROFL_KG(4)
__global__ void __launch_bounds__(TPB) k_quantize_shift(const float *vals, u32 d, u32 dpad, u32 prove_range, u32 fp_bits, u32 fp_frac,
                                 float clip_min, float clip_max, u64 *vshift, u32 *status) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= dpad) return;
    vals += (size_t)blockIdx.y * d; vshift += (size_t)blockIdx.y * dpad; status += blockIdx.y;      // one client per grid row
    if (i >= d) { vshift[i] = 0; return; }
    float v = vals[i];
    u32 st = 0;
    if (clip_min > v || v > clip_max) st |= 1;
    if (v != v) st |= 2;
    if (st) { atomicOr(status, st); vshift[i] = 0; return; }
    double x = fabs((double)v) * (double)(1ULL << fp_frac);
    double lim = ldexp(1.0, (int)fp_bits);
    u64 maxbits = fp_bits >= 64 ? ~0ULL : ((1ULL << fp_bits) - 1);
    u64 k;
    if (x >= lim) k = maxbits;
    else { double r = rint(x); k = (r >= lim) ? maxbits : (u64)r; }
    // (+-k + 2^(range-1)) mod l, then read_from_bytes keeps the low fp_bits bits (fp.rs)
    u64 off = 1ULL << (prove_range - 1);
    u64 lowbits;
    if (v < 0.0f) {
        if (off >= k) lowbits = off - k;
        else {   // l - (k - off): low 64 bits of l minus (k-off)
            u64 l_lo = ((u64)SC_L1 << 32) | SC_L0;
            lowbits = l_lo - (k - off);
        }
    } else lowbits = k + off;   // k, off < 2^63 for fp_bits <= 63; fp64 wraps like the u64 cast chain
    vshift[i] = lowbits & maxbits;
}
#endif

// ================================================================ K3: Pedersen commit (fixed-base)
// tables: radix-256 signed digits, tab[w][e] = (e+1) * 256^w * P, w < 32, e < 128  (affine niels; built by k_fixed_tab8 from the radix-16
// table the per-element chains started with: half the mixed additions, the 393 KB of a base stay in L2)
__device__ __forceinline__ gd fixed_base_mul_acc(gd acc, const niels *tab, const sc &k, int nwin) {
    // k canonical (< 2^253 when nwin == 32 so the carry digit is zero)
    int carry = 0;
    for (int i = 0; i < nwin; i++) {
        int v = (int)((k.v[i >> 2] >> ((i & 3) * 8)) & 255) + carry;
        carry = (v + 128) >> 8;
        int dgt = v - (carry << 8);
        int ad = dgt < 0 ? -dgt : dgt;
        if (ad) acc = gd_madd(acc, load_nd(&tab[i * 128 + ad - 1]), dgt < 0);
    }
    if (carry && nwin < 32) acc = gd_madd(acc, load_nd(&tab[nwin * 128 + 0]), false);
    return acc;
}
// V_j = v_j*B + r_j*Bb (compressed), and optionally C_j = V_j + shift (compressed)
#if ROFL_KG(4)
__global__ void __launch_bounds__(TPB) k_commit(u32 count, const u64 *v64, const sc *v256_canon, const sc *blind_canon /* may be null */,
                         const niels *tabB, const niels *tabBb, const niels *shift /* may be null */,
                         uint8_t *V_out /* may be null */, uint8_t *C_out /* may be null */, u32 c_count, u32 c_period) {
    // C_out is written for the first c_count entries of every c_period (batched create: c_period = padded length of one client)
    u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    gd acc = gd_identity();
    // scalars arrive from the C ABI as 32 raw bytes: anything >= l is reduced (a dalek Scalar is always < l; the windows below
    // assume < 2^253)
    if (v64) { sc v = sc_from_u64(v64[j]); acc = fixed_base_mul_acc(acc, tabB, v, 8); }
    else { sc v = load_sc_reduced(&v256_canon[j]); acc = fixed_base_mul_acc(acc, tabB, v, 32); }
    if (blind_canon) { sc r = load_sc_reduced(&blind_canon[j]); acc = fixed_base_mul_acc(acc, tabBb, r, 32); }
    if (V_out) gd_ristretto_encode(V_out + (size_t)j * 32, acc);
    if (C_out && (j % c_period) < c_count) {
        gd cpt = shift ? gd_madd(acc, load_nd(shift), false) : acc;
        gd_ristretto_encode(C_out + (size_t)j * 32, cpt);
    }
}
#endif

// ================================================================ K4: A = sum (bit ? G : -H)
#if ROFL_KG(4)
__global__ void __launch_bounds__(TPB) k_bitcommit(u32 n, u32 m, const u64 *vshift /* [chunk][m] */, const niels *tbl, ge *partial /* [chunk][m] */) {
    u32 c = blockIdx.y;
    u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    size_t N = (size_t)n * m;
    u64 v = vshift[(size_t)c * m + j];
    gd acc = gd_identity();
    for (u32 i = 0; i < n; i++) {
        bool bit = (v >> i) & 1;
        const niels *p = bit ? &tbl[(size_t)j * n + i] : &tbl[N + (size_t)j * n + i];
        acc = gd_madd(acc, gload_nd(p), !bit);
    }
    store_gd(&partial[(size_t)c * m + j], acc);
}
#endif

// generic point reduction: in [prob][n] -> out [prob][gridDim.x]
#if ROFL_KG(4)
__global__ void __launch_bounds__(TPB) k_point_sum(const ge *in, u32 n, ge *out) {
    extern __shared__ __align__(16) unsigned char smem[];
    ge *lds = reinterpret_cast<ge *>(smem);
    u32 p = blockIdx.y, t = threadIdx.x;
    const ge *src = in + (size_t)p * n;
    ge acc = ge_identity();
    for (u32 i = blockIdx.x * blockDim.x + t; i < n; i += gridDim.x * blockDim.x) acc = ge_add(acc, load_ge(&src[i]));
    lds[t] = acc;
    __syncthreads();
    for (u32 s = blockDim.x / 2; s > 0; s >>= 1) {
        if (t < s) lds[t] = ge_add(lds[t], lds[t + s]);
        __syncthreads();
    }
    if (t == 0) store_ge(&out[(size_t)p * gridDim.x + blockIdx.x], lds[0]);
}
#endif

// sums of decoded points held as affine niels triples: vector y = in[y * stride .. + n) -> out[y][gridDim.x] block partials.  The batched
// Sigma-proof verifier has every c_sq decoded already (k_sigma_vprep); `sum c_sq` per client (params.rs:220) costs one mixed addition each.
#if ROFL_KG(4)
__global__ void __launch_bounds__(TPB) k_niels_sum(const niels *in, u32 n, size_t stride, ge *out) {
    extern __shared__ __align__(16) unsigned char smem[];
    ge *lds = reinterpret_cast<ge *>(smem);
    u32 t = threadIdx.x;
    const niels *src = in + (size_t)blockIdx.y * stride;
    gd acc = gd_identity();
    for (u32 i = blockIdx.x * blockDim.x + t; i < n; i += gridDim.x * blockDim.x) acc = gd_madd(acc, load_nd(&src[i]), false);
    lds[t] = gd_pack(acc);
    __syncthreads();
    for (u32 s = blockDim.x / 2; s > 0; s >>= 1) {
        if (t < s) lds[t] = ge_add(lds[t], lds[t + s]);
        __syncthreads();
    }
    if (t == 0) store_ge(&out[(size_t)blockIdx.y * gridDim.x + blockIdx.x], lds[0]);
}
#endif

// sum of compressed points (params.rs:220, 277: `enc_values.iter().map(|x| x.c_sq).sum()`): decode + grid-stride sum + LDS tree
#if ROFL_KG(4)
__global__ void __launch_bounds__(TPB) k_decode_sum(const uint8_t *in, u32 n, u32 stride, ge *out, u32 *status) {
    extern __shared__ __align__(16) unsigned char smem[];
    ge *lds = reinterpret_cast<ge *>(smem);
    u32 t = threadIdx.x;
    gd gacc = gd_identity();
    for (u32 i = blockIdx.x * blockDim.x + t; i < n; i += gridDim.x * blockDim.x) {
        __align__(16) uint8_t b[32];
        const uint8_t *s = in + (size_t)i * stride;
        for (int q = 0; q < 32; q++) b[q] = s[q];
        gd p;
        if (!gd_ristretto_decode(p, b)) { atomicOr(status, 4u); p = gd_identity(); }
        gacc = gd_add(gacc, p);
    }
    lds[t] = gd_pack(gacc);
    __syncthreads();
    for (u32 s = blockDim.x / 2; s > 0; s >>= 1) {
        if (t < s) lds[t] = ge_add(lds[t], lds[t + s]);
        __syncthreads();
    }
    if (t == 0) store_ge(&out[blockIdx.x], lds[0]);
}
#endif

// ================================================================ party-level scalar sums
// phase 0: sum a_bl, s_bl              -> out[chunk][blk][0..1]
// phase 1: sum t1_bl, t2_bl, zz*z^j*vbl -> out[chunk][blk][0..2]
#if ROFL_KG(4)
__global__ void __launch_bounds__(TPB) k_party_sums(u32 m, int phase, const ChunkParams *cp, const PowTabs *pt, const sc *party, const sc *blind_canon /* [chunk][m] */, sc *out) {
    __shared__ sc lds[TPB * 3];
    u32 c = blockIdx.y;
    sc v[3] = {sc_zero(), sc_zero(), sc_zero()};
    for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += gridDim.x * blockDim.x) {
        if (phase == 0) {
            v[0] = sc_add(v[0], load_sc(&party[((size_t)c * 4 + 0) * m + j]));
            v[1] = sc_add(v[1], load_sc(&party[((size_t)c * 4 + 1) * m + j]));
        } else {
            v[0] = sc_add(v[0], load_sc(&party[((size_t)c * 4 + 2) * m + j]));
            v[1] = sc_add(v[1], load_sc(&party[((size_t)c * 4 + 3) * m + j]));
            sc zj = sc_montmul(cp[c].zz, pt_pow(pt[c].z, cp[c].zpow2, j));
            sc bl = sc_to_mont(load_sc(&blind_canon[(size_t)c * m + j]));
            v[2] = sc_add(v[2], sc_montmul(zj, bl));
        }
    }
    block_sum_sc<3>(v, lds);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) store_sc(&out[((size_t)c * gridDim.x + blockIdx.x) * 3 + k], v[k]);
    }
}
#endif

// ================================================================ K6: polynomial vectors
// bulletproofs party.rs apply_challenge: l0 = aL - z, l1 = sL, r0 = y^k (aR + z) + z^(2+j) 2^i, r1 = y^k sR
__device__ __forceinline__ void slot_vectors(const ChunkParams &P, const PowTabs &T, u32 n, u32 k, u64 v, const sc &sR,
                                             sc &l0, sc &r0, sc &r1, const sc *two_pow) {
    u32 j = k / n, i = k % n;
    bool bit = (v >> i) & 1;
    sc one = sc_one_mont();
    sc aL = bit ? one : sc_zero();
    sc aR = bit ? sc_zero() : sc_neg(one);
    sc yk = pt_pow(T.y, P.ypow2, k);
    l0 = sc_sub(aL, P.z);
    sc zj2 = sc_montmul(sc_montmul(P.zz, pt_pow(T.z, P.zpow2, j)), two_pow[i]);
    r0 = sc_add(sc_montmul(yk, sc_add(aR, P.z)), zj2);
    r1 = sc_montmul(yk, sR);
}
// t0,t1,t2 partial sums -> out[chunk][blk][3]
#if ROFL_KG(4)
__global__ void __launch_bounds__(TPB) k_poly_t(u32 n, u32 m, const ChunkParams *cp, const PowTabs *pt, const u64 *vshift, const sc *sL, const sc *sR,
                         const sc *two_pow, sc *out) {
    __shared__ sc lds[TPB * 3];
    u32 c = blockIdx.y;
    size_t N = (size_t)n * m;
    sc v[3] = {sc_zero(), sc_zero(), sc_zero()};
    for (u32 k = blockIdx.x * blockDim.x + threadIdx.x; k < N; k += gridDim.x * blockDim.x) {
        sc l0, r0, r1;
        sc l1 = load_sc(&sL[c * N + k]);
        slot_vectors(cp[c], pt[c], n, k, vshift[(size_t)c * m + k / n], load_sc(&sR[c * N + k]), l0, r0, r1, two_pow);
        v[0] = sc_add(v[0], sc_montmul(l0, r0));
        v[1] = sc_add(v[1], sc_add(sc_montmul(l0, r1), sc_montmul(l1, r0)));
        v[2] = sc_add(v[2], sc_montmul(l1, r1));
    }
    block_sum_sc<3>(v, lds);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) store_sc(&out[((size_t)c * gridDim.x + blockIdx.x) * 3 + k], v[k]);
    }
}
#endif
// a = l(x), b = r(x); also yinv^k table
#if ROFL_KG(4)
__global__ void __launch_bounds__(TPB) k_lr_vec(u32 n, u32 m, const ChunkParams *cp, const PowTabs *pt, const u64 *vshift, const sc *sL, const sc *sR,
                         const sc *two_pow, sc *a, sc *b, sc *yinvpow) {
    u32 c = blockIdx.y;
    size_t N = (size_t)n * m;
    u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    sc l0, r0, r1;
    sc l1 = load_sc(&sL[c * N + k]);
    slot_vectors(cp[c], pt[c], n, k, vshift[(size_t)c * m + k / n], load_sc(&sR[c * N + k]), l0, r0, r1, two_pow);
    store_sc(&a[c * N + k], sc_add(l0, sc_montmul(l1, cp[c].x)));
    store_sc(&b[c * N + k], sc_add(r0, sc_montmul(r1, cp[c].x)));
    store_sc(&yinvpow[c * N + k], pt_pow(pt[c].yinv, cp[c].yinvpow2, k));
}
#endif

// k_lr_vec + the first round's k_ipp_scalars (merged form) + k_ipp_inner in one pass: a thread owns the pairs (i, N/2 + i) of its chunk and
// writes a, b, y^-j, the round's MSM scalars (no pending challenges yet: s_G = gscale, s_H = hscale y^-j; generator j of the low half takes
// the vectors' high half and vice versa) and the partial inner products <a_L, b_R>, <a_R, b_L> of its block -> ip_out[chunk][block][2].
#if ROFL_KG(4)
__global__ void __launch_bounds__(TPB, 3) k_lr_first(u32 n, u32 m, const ChunkParams *cp, const PowTabs *pt, const u64 *vshift, const sc *sL, const sc *sR,
                           const sc *two_pow, sc *a, sc *b, sc *yinvpow, sc *SL, sc *ip_out) {
    __shared__ sc lds[TPB * 2];
    const u32 c = blockIdx.y;
    const size_t N = (size_t)n * m;
    const u32 nh = (u32)(N / 2);
    const sc sG = load_sc(&cp[c].gscale), sH0 = load_sc(&cp[c].hscale), x = cp[c].x;
    sc *sl = SL + (size_t)c * 2 * N;
    sc v[2] = {sc_zero(), sc_zero()};
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < nh; i += gridDim.x * blockDim.x) {
        // the two elements one after the other in a ROLLED loop (unrolled, the compiler interleaves them: 230 VGPRs and spills), results in named
        // registers (an indexed pair would live in scratch)
        sc a0 = sc_zero(), b0 = a0, y0 = a0, a1 = a0, b1 = a0, y1 = a0;
#pragma unroll 1
        for (int s = 0; s < 2; s++) {
            const u32 k = s ? nh + i : i;
            sc l0, r0, r1;
            sc l1 = load_sc(&sL[c * N + k]);
            slot_vectors(cp[c], pt[c], n, k, vshift[(size_t)c * m + k / n], load_sc(&sR[c * N + k]), l0, r0, r1, two_pow);
            sc ao = sc_add(l0, sc_montmul(l1, x));
            sc bo = sc_add(r0, sc_montmul(r1, x));
            sc yo = pt_pow(pt[c].yinv, cp[c].yinvpow2, k);
            store_sc(&a[c * N + k], ao); store_sc(&b[c * N + k], bo); store_sc(&yinvpow[c * N + k], yo);
            if (s == 0) { a0 = ao; b0 = bo; y0 = yo; } else { a1 = ao; b1 = bo; y1 = yo; }
        }
        store_sc(&sl[i], sc_from_mont(sc_montmul(a1, sG)));
        store_sc(&sl[N + i], sc_from_mont(sc_montmul(b1, sc_montmul(sH0, y0))));
        store_sc(&sl[nh + i], sc_from_mont(sc_montmul(a0, sG)));
        store_sc(&sl[N + nh + i], sc_from_mont(sc_montmul(b0, sc_montmul(sH0, y1))));
        v[0] = sc_add(v[0], sc_montmul(a0, b1));
        v[1] = sc_add(v[1], sc_montmul(a1, b0));
    }
    block_sum_sc<2>(v, lds);
    if (threadIdx.x == 0) {
        store_sc(&ip_out[((size_t)c * gridDim.x + blockIdx.x) * 2 + 0], v[0]);
        store_sc(&ip_out[((size_t)c * gridDim.x + blockIdx.x) * 2 + 1], v[1]);
    }
}
#endif

// ================================================================ K7: inner-product argument
// Lazily folded generators: the materialised arrays Gc/Hc have n_g entries; the logical vectors have
// n_k = n_g >> r entries; true G[i] = sum_h s_G(h) Gc[h*n_k+i], true H[i] = sum_h s_H(h) y^-j Hc[j].
// Writes canonical MSM scalars for L (SL) and R (SR) over [Gc | Hc].
#if ROFL_KG(4)
__global__ void __launch_bounds__(TPB) k_ipp_scalars(u32 n_g, u32 n_k, u32 r, const ChunkParams *cp, const sc *a, const sc *b, size_t ab_stride,
                              const sc *yinvpow, size_t y_stride, sc *SL, sc *SR, int merged) {
    u32 c = blockIdx.y;
    u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_g) return;
    u32 h = j / n_k, i = j % n_k, nh = n_k / 2;
    const sc *ac = a + c * ab_stride, *bc = b + c * ab_stride;
    // s_G(h) = gscale * prod_q (bit_q(h) ? u_q : u_q^-1), s_H(h) with u and u^-1 swapped; challenge q <-> bit r-1-q of h
    sc sG = load_sc(&cp[c].gscale), sH = load_sc(&cp[c].hscale);
    for (u32 q = 0; q < r; q++) {
        bool bit = (h >> (r - 1 - q)) & 1;
        sc up = load_sc(&cp[c].pend_u[q]), ui = load_sc(&cp[c].pend_ui[q]);
        sG = sc_montmul(sG, bit ? up : ui);
        sH = sc_montmul(sH, bit ? ui : up);
    }
    sH = sc_montmul(sH, load_sc(&yinvpow[c * y_stride + j]));
    sc *sl = SL + (size_t)c * 2 * n_g, *sr = SR + (size_t)c * 2 * n_g;
    if (merged) {      // one array, every term non-zero; the side of a term is a function of its index (MsmMap)
        bool lo = i < nh; u32 ii = lo ? nh + i : i - nh;
        store_sc(&sl[j], sc_from_mont(sc_montmul(load_sc(&ac[ii]), sG)));
        store_sc(&sl[n_g + j], sc_from_mont(sc_montmul(load_sc(&bc[ii]), sH)));
        return;
    }
    sc zero = sc_zero();
    if (i < nh) {
        // G_L / H_L halves: R gets a_R * G_L ; L gets b_R * H_L
        store_sc(&sr[j], sc_from_mont(sc_montmul(load_sc(&ac[nh + i]), sG)));
        store_sc(&sl[j], zero);
        store_sc(&sl[n_g + j], sc_from_mont(sc_montmul(load_sc(&bc[nh + i]), sH)));
        store_sc(&sr[n_g + j], zero);
    } else {
        u32 ii = i - nh;
        store_sc(&sl[j], sc_from_mont(sc_montmul(load_sc(&ac[ii]), sG)));
        store_sc(&sr[j], zero);
        store_sc(&sr[n_g + j], sc_from_mont(sc_montmul(load_sc(&bc[ii]), sH)));
        store_sc(&sl[n_g + j], zero);
    }
}
#endif
// c_L = <a_L, b_R>, c_R = <a_R, b_L>  -> out[chunk][blk][2]
#if ROFL_KG(4)
__global__ void __launch_bounds__(TPB) k_ipp_inner(u32 nh, const sc *a, const sc *b, size_t ab_stride, sc *out) {
    __shared__ sc lds[TPB * 2];
    u32 c = blockIdx.y;
    const sc *ac = a + c * ab_stride, *bc = b + c * ab_stride;
    sc v[2] = {sc_zero(), sc_zero()};
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < nh; i += gridDim.x * blockDim.x) {
        v[0] = sc_add(v[0], sc_montmul(load_sc(&ac[i]), load_sc(&bc[nh + i])));
        v[1] = sc_add(v[1], sc_montmul(load_sc(&ac[nh + i]), load_sc(&bc[i])));
    }
    block_sum_sc<2>(v, lds);
    if (threadIdx.x == 0) {
        store_sc(&out[((size_t)c * gridDim.x + blockIdx.x) * 2 + 0], v[0]);
        store_sc(&out[((size_t)c * gridDim.x + blockIdx.x) * 2 + 1], v[1]);
    }
}
#endif
// a_L = a_L u + u^-1 a_R ; b_L = b_L u^-1 + u b_R.  This round's challenge (u, u^-1; Montgomery) is read from `round_ch`
// [chunk][2] -- mapped host memory the host wrote after the transcript step, so no H2D copy sits on the hop -- and recorded as
// pending challenge number `pend_idx` of the chunk (k_ipp_scalars of the next rounds reads the pending list from device memory).
#if ROFL_KG(4)
__global__ void __launch_bounds__(TPB) k_ipp_fold_ab(u32 nh, ChunkParams *cp, const sc *round_ch, u32 pend_idx, sc *a, sc *b, size_t ab_stride) {
    u32 c = blockIdx.y;
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    sc u = load_sc(&round_ch[2 * c]), ui = load_sc(&round_ch[2 * c + 1]);
    if (i == 0) { store_sc(&cp[c].pend_u[pend_idx], u); store_sc(&cp[c].pend_ui[pend_idx], ui); }
    if (i >= nh) return;
    sc *ac = a + c * ab_stride, *bc = b + c * ab_stride;
    store_sc(&ac[i], sc_add(sc_montmul(load_sc(&ac[i]), u), sc_montmul(load_sc(&ac[nh + i]), ui)));
    store_sc(&bc[i], sc_add(sc_montmul(load_sc(&bc[i]), ui), sc_montmul(load_sc(&bc[nh + i]), u)));
}
#endif

// One launch per IPP round (rounds >= 1, merged L/R layout) instead of k_ipp_fold_ab + k_ipp_scalars + k_ipp_inner and an H2D copy
// of the challenge:  (1) fold a, b with the previous round's challenge u -- read from mapped host memory, once per block -- into the
// other ping-pong buffer; (2) the MSM scalars of THIS round over the materialised generators, i.e. the folded values times the
// products of the pending challenges (those already on the device plus, unless the generators were re-materialised in between, u);
// (3) the partial sums of c_L = <a_L, b_R>, c_R = <a_R, b_L> -> ip_out [chunk][gridDim.x][2] (mapped host memory).
// n_k = logical vector length of this round (the inputs have 2 n_k entries).  Every thread folds what it needs itself, so nothing
// in the launch depends on another thread's output.
#if ROFL_KG(4)
__global__ void __launch_bounds__(TPB) k_ipp_round(u32 n_g, u32 n_k, u32 r_prev, int use_new, ChunkParams *cp, const sc *round_ch, const sc *a_in, const sc *b_in,
                                                   sc *a_out, sc *b_out, size_t ab_stride, const sc *yinvpow, size_t y_stride, sc *SL, sc *ip_out,
                                                   const sc *ptab_in, sc *ptab_out, size_t ptab_stride, sc *ab_host, sc *ip_dev) {
    __shared__ sc lds[TPB * 2];
    __shared__ sc s_u[2];
    u32 c = blockIdx.y;
    if (threadIdx.x == 0) { s_u[0] = load_sc(&round_ch[2 * c]); s_u[1] = load_sc(&round_ch[2 * c + 1]); }
    __syncthreads();
    const sc u = s_u[0], ui = s_u[1];
    if (use_new && blockIdx.x == 0 && threadIdx.x == 0) { store_sc(&cp[c].pend_u[r_prev], u); store_sc(&cp[c].pend_ui[r_prev], ui); }   // read by later launches only
    const sc *ai = a_in + c * ab_stride, *bi = b_in + c * ab_stride;
    sc *ao = a_out + c * ab_stride, *bo = b_out + c * ab_stride;
    const u32 nh = n_k / 2;
    // the chunk's scales OUT of Montgomery form: a product with them then leaves the Montgomery domain by itself, and the MSM scalars
    // a' s_G, b' s_H come out canonical without a conversion of their own (two multiplications per element less)
    const sc gs = sc_from_mont(load_sc(&cp[c].gscale)), hs = sc_from_mont(load_sc(&cp[c].hscale));
    sc *sl = SL + (size_t)c * 2 * n_g;
    sc v[2] = {sc_zero(), sc_zero()};
    // x u + y u' as ONE reduction over the sum of the two plain products (fe32.hpp: sc_mac_wide / sc_redc_wide)
    auto fold2 = [](const sc &x, const sc &cu, const sc &y, const sc &cv) {
        u32 w[17];
#pragma unroll
        for (int q = 0; q < 17; q++) w[q] = 0;
        sc_mac_wide(w, x, cu); sc_mac_wide(w, y, cv);
        return sc_redc_wide(w);
    };
    for (u32 j = blockIdx.x * blockDim.x + threadIdx.x; j < n_g; j += gridDim.x * blockDim.x) {
        u32 h = j / n_k, i = j % n_k;
        bool lo = i < nh; u32 ii = lo ? nh + i : i - nh;
        sc af = fold2(load_sc(&ai[ii]), u, load_sc(&ai[n_k + ii]), ui);
        sc bf = fold2(load_sc(&bi[ii]), ui, load_sc(&bi[n_k + ii]), u);
        // product of the pending challenges selected by the bits of h (challenge q <-> bit r-1-q): the table of the previous round
        // (2^r_prev entries per chunk and side, written by that round's launch) extended by the newest challenge -- one multiplication
        // per element instead of r of them (r reaches 11 in the tail)
        u32 hp = use_new ? h >> 1 : h;
        // (the table carries the chunk's scale along, in plain form: gscale / hscale are constant while challenges are pending -- a fold
        //  changes them and empties the table -- so s_G and s_H need no multiplication of their own)
        sc tG = r_prev ? load_sc(&ptab_in[(size_t)c * 2 * ptab_stride + hp]) : gs;
        sc tH = r_prev ? load_sc(&ptab_in[(size_t)c * 2 * ptab_stride + ptab_stride + hp]) : hs;
        if (use_new) { bool bit = h & 1; tG = sc_montmul(tG, bit ? u : ui); tH = sc_montmul(tH, bit ? ui : u); }
        if (i == 0) { store_sc(&ptab_out[(size_t)c * 2 * ptab_stride + h], tG); store_sc(&ptab_out[(size_t)c * 2 * ptab_stride + ptab_stride + h], tH); }
        const sc sG = tG, sH = sc_montmul(tH, load_sc(&yinvpow[c * y_stride + j]));      // plain values (the table's are)
        store_sc(&sl[j], sc_montmul(af, sG));                                  // Montgomery x plain -> canonical
        store_sc(&sl[n_g + j], sc_montmul(bf, sH));
        if (h == 0) {
            store_sc(&ao[ii], af); store_sc(&bo[ii], bf);
            // last round (n_k == 2): the two entries of a and b also go to mapped host memory -- the host applies the final challenge
            // itself (a = a_0 u + a_1 u^-1, b = b_0 u^-1 + b_1 u) instead of one more launch, two copies and a wait at the end of every proof
            if (ab_host && n_k == 2) { store_sc(&ab_host[c * 4 + ii], af); store_sc(&ab_host[c * 4 + 2 + ii], bf); }
            if (lo) {      // this thread holds a'[nh+i], b'[nh+i]; with a'[i], b'[i] it owns one term of each inner product
                sc al = fold2(load_sc(&ai[i]), u, load_sc(&ai[n_k + i]), ui);
                sc bl = fold2(load_sc(&bi[i]), ui, load_sc(&bi[n_k + i]), u);
                v[0] = sc_add(v[0], sc_montmul(al, bf));
                v[1] = sc_add(v[1], sc_montmul(af, bl));
            }
        }
    }
    block_sum_sc<2>(v, lds);
    if (threadIdx.x == 0) {
        store_sc(&ip_out[((size_t)c * gridDim.x + blockIdx.x) * 2 + 0], v[0]);
        store_sc(&ip_out[((size_t)c * gridDim.x + blockIdx.x) * 2 + 1], v[1]);
        if (ip_dev) {      // the same partial sums where the round's MSM launch can read them without crossing PCIe (k_msm_small adds c_L w B itself)
            store_sc(&ip_dev[((size_t)c * gridDim.x + blockIdx.x) * 2 + 0], v[0]);
            store_sc(&ip_dev[((size_t)c * gridDim.x + blockIdx.x) * 2 + 1], v[1]);
        }
    }
}
#endif

// Materialise folded generators: dst[i] = sum_{h < nsrc} s_h * src[h*n_new + i], all outputs of a problem
// share the scalars s_h (given as NAF digits, wave-uniform control flow => no divergence).
struct FoldProb { const niels *src; niels *dst; };
#define FOLD_MAXSRC 64
#define FOLD_MAXSEG 4
// The 254 NAF digit positions are split into K contiguous segments (threadIdx.y); segment k covers bits
// [seg.lo[k], seg.lo[k+1]) and finishes with seg.lo[k] plain doublings, so K threads share one output and the
// launch has K times as many waves in flight (the chain is latency-bound at 2 waves/SIMD otherwise).
struct FoldSeg { int lo[FOLD_MAXSEG + 1]; };
#if ROFL_KG(2)
__global__ void __launch_bounds__(256, 4) k_fold_gens(u32 n_new, u32 nsrc, FoldSeg seg, const FoldProb *probs, const int8_t *naf /* [prob][nsrc][256] */, int unit_first) {
    extern __shared__ __align__(16) unsigned char smem[];
    ge *lds = reinterpret_cast<ge *>(smem);
    u32 q = blockIdx.y;
    u32 i = blockIdx.x * 64 + threadIdx.x;
    u32 k = threadIdx.y, K = blockDim.y;
    bool active = i < n_new;
    const niels *src = probs[q].src;
    const int8_t *dg = naf + (size_t)q * nsrc * 256;
    gd acc = gd_identity();
    if (active) {
        int lo = seg.lo[k], hi = seg.lo[k + 1] - 1;
        for (int bit = hi; bit >= lo; bit--) {
            acc = gd_double(acc);
            for (u32 h = unit_first ? 1 : 0; h < nsrc; h++) {
                int d = dg[h * 256 + bit];
                if (d != 0) acc = gd_madd(acc, gload_nd(&src[(size_t)h * n_new + i]), d < 0);
            }
        }
        for (int t = 0; t < lo; t++) acc = gd_double(acc);
    }
    if (K > 1) {
        if (k > 0) lds[(k - 1) * 64 + threadIdx.x] = gd_pack(acc);
        __syncthreads();
        if (k == 0)
            for (u32 s2 = 1; s2 < K; s2++) acc = gd_add(acc, gd_unpack(lds[(s2 - 1) * 64 + threadIdx.x]));
    }
    if (active && k == 0) {
        if (unit_first) acc = gd_madd(acc, gload_nd(&src[i]), false);   // source 0 has scalar 1 (scale kept aside)
        gstore_niels(&probs[q].dst[i], gd_to_niels(acc));
    }
}
#endif

// The same fold for nsrc = 4 with the unit scalar on source 0 (every fold after the first at the default schedule): the three
// sources that carry a scalar are loaded ONCE into registers (3 x 30 limbs) instead of being gathered again for each of the ~250
// additions of the chain -- the chain is then pure VALU work with no memory latency in it.  Needs ~230 VGPRs: two waves per SIMD,
// which is what the launch offers anyway (the digit positions are split into K segments to have more chains in flight).
#if ROFL_KG(2)
__global__ void __launch_bounds__(256, 2) k_fold_gens4(u32 n_new, FoldSeg seg, const FoldProb *probs, const int8_t *naf /* [prob][4][256] */) {
    extern __shared__ __align__(16) unsigned char smem[];
    ge *lds = reinterpret_cast<ge *>(smem);
    u32 q = blockIdx.y;
    u32 i = blockIdx.x * 64 + threadIdx.x;
    u32 k = threadIdx.y, K = blockDim.y;
    bool active = i < n_new;
    const niels *src = probs[q].src;
    const int8_t *dg = naf + (size_t)q * 4 * 256;
    gd acc = gd_identity();
    if (active) {
        const nd s1 = gload_nd(&src[(size_t)1 * n_new + i]), s2 = gload_nd(&src[(size_t)2 * n_new + i]), s3 = gload_nd(&src[(size_t)3 * n_new + i]);
        int lo = seg.lo[k], hi = seg.lo[k + 1] - 1;
        for (int bit = hi; bit >= lo; bit--) {
            acc = gd_double(acc);
            int d1 = dg[256 + bit], d2 = dg[512 + bit], d3 = dg[768 + bit];      // wave-uniform
            if (d1 != 0) acc = gd_madd(acc, s1, d1 < 0);
            if (d2 != 0) acc = gd_madd(acc, s2, d2 < 0);
            if (d3 != 0) acc = gd_madd(acc, s3, d3 < 0);
        }
        for (int t = 0; t < lo; t++) acc = gd_double(acc);
    }
    if (K > 1) {
        if (k > 0) lds[(k - 1) * 64 + threadIdx.x] = gd_pack(acc);
        __syncthreads();
        if (k == 0)
            for (u32 s2i = 1; s2i < K; s2i++) acc = gd_add(acc, gd_unpack(lds[(s2i - 1) * 64 + threadIdx.x]));
    }
    if (active && k == 0) {
        acc = gd_madd(acc, gload_nd(&src[i]), false);   // source 0 has scalar 1 (scale kept aside)
        gstore_niels(&probs[q].dst[i], gd_to_niels(acc));
    }
}
#endif

// Later folds with odd multiples of their sources.  The sources of a later fold are the previous fold's outputs -- proof-specific, so nothing
// can be precomputed across proofs; but they exist two rounds before the fold needs them.  k_odd_multiples builds (2e+1) P for e = 1 .. E-1 of
// every materialised point on the SIDE stream while those rounds run (extended coordinates; k_niels_batch turns them into affine niels, eight
// per inversion), and the fold walks width-w NAF digits (one non-zero in w + 1 positions instead of one in three): its chain -- the critical
// path of the phase, at two waves per SIMD -- loses a third of its additions for work that was done off the path.
#if ROFL_KG(2)
__global__ void __launch_bounds__(TPB) k_odd_multiples(u32 count, u32 E, const niels *src, ge *ext /* [E-1][count] */) {
    const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const gd p = gd_unpack(ge_from_niels(load_niels(&src[j])));
    const gd p2 = gd_double(p);
    gd cur = p;
    for (u32 e = 1; e < E; e++) { cur = gd_add(cur, p2); store_gd(&ext[(size_t)(e - 1) * count + j], cur); }
}
#endif
// The fold itself: an EVENT list per problem instead of digit arrays -- (bit, source, multiple, sign), highest bit first, wave-uniform -- so
// that the operand of the next addition is fetched while the doublings in front of it run (the table is in HBM: gathered inside the chain,
// as the generic kernel does with its sources, every addition waited ~1.7 us for its operand).  Segments as in k_fold_gens.
struct FoldWProb { const niels *src; niels *dst; u32 tab_off, ev_off, n_ev, seg_start[FOLD_MAXSEG]; };
#define FOLD_EV(bit, h, e, neg) ((u32)(bit) | ((u32)(h) << 9) | ((u32)(e) << 15) | ((u32)(neg) << 27))      /* e: table slice, 0 = the source itself (12 bits) */
#if ROFL_KG(2)
__device__ __forceinline__ void fold_w_body(u32 n_new, const FoldSeg &seg, const FoldWProb *probs, const u32 *events, const niels *tab, size_t tstride, ge *ext_out) {
    extern __shared__ __align__(16) unsigned char smem[];
    ge *lds = reinterpret_cast<ge *>(smem);
    const u32 q = blockIdx.y;
    const u32 i = blockIdx.x * 64 + threadIdx.x;
    const u32 k = threadIdx.y, K = blockDim.y;
    const bool active = i < n_new;
    const niels *src = probs[q].src;
    const u32 *ev = events + probs[q].ev_off;
    gd acc = gd_identity();
    if (active) {
        const niels *tb = tab + probs[q].tab_off;
        auto operand = [&](u32 w) { const u32 h = (w >> 9) & 63u, e = (w >> 15) & 4095u;
                                    const niels *p = e ? tb + (size_t)(e - 1) * tstride : src;
                                    return gload_nd(&p[(size_t)h * n_new + i]); };
        const int lo = seg.lo[k];
        int pos = seg.lo[k + 1];                              // the accumulator's scale: `pos - b` doublings go in front of an addition at bit b
        u32 j = probs[q].seg_start[k];
        const u32 jend = k == 0 ? probs[q].n_ev : probs[q].seg_start[k - 1];      // (segments run from the top down: k - 1 covers the bits below k's)
        if (j < jend) {
            u32 w = ev[j];
            nd nxt = operand(w);
            while (j < jend) {
                const nd cur = nxt; const u32 wc = w;
                j++;
                if (j < jend) { w = ev[j]; nxt = operand(w); }
                const int b = (int)(wc & 511u);
                for (; pos > b; pos--) acc = gd_double(acc);
                acc = gd_madd(acc, cur, (wc >> 27) & 1u);
            }
        }
        for (; pos > lo; pos--) acc = gd_double(acc);
        for (int t = 0; t < lo; t++) acc = gd_double(acc);
    }
    if (K > 1) {
        if (k > 0) lds[(k - 1) * 64 + threadIdx.x] = gd_pack(acc);
        __syncthreads();
        if (k == 0)
            for (u32 s2 = 1; s2 < K; s2++) acc = gd_add(acc, gd_unpack(lds[(s2 - 1) * 64 + threadIdx.x]));
    }
    if (active && k == 0) {
        acc = gd_madd(acc, gload_nd(&src[i]), false);   // source 0 has scalar 1 (scale kept aside)
        if (ext_out) store_gd(&ext_out[(size_t)q * n_new + i], acc);      // (the first fold: k_niels_batch converts on the side stream)
        else gstore_niels(&probs[q].dst[i], gd_to_niels(acc));
    }
}
// (167 VGPRs: three waves per SIMD.  Compiled for four -- 128 VGPRs, 260 bytes of scratch -- the table fold is 0.5 ms slower; held at two by LDS
// padding 1.0 ms slower: the operand gathers want waves to hide behind, but not at the price of spills.)
__global__ void __launch_bounds__(256, 2) k_fold_gens_w(u32 n_new, FoldSeg seg, const FoldWProb *probs, const u32 *events, const niels *tab, size_t tstride, ge *ext_out) {
    fold_w_body(n_new, seg, probs, events, tab, tstride, ext_out);
}
#endif

// First materialisation: the sources are the FIXED generators, for which get_gens precomputed
//   tbl16[(q*4+e)*stride + g] = (2e+1) * 2^(64q) * G_g      (q < 4, e < 4; affine niels)
// so a 253-bit scalar becomes four 64-bit pieces in width-4 NAF: 64 doublings per output instead of 253 and
// ~51 instead of ~84 mixed additions per source.  HBM capacity (16 x 50 MB at N = 262144) traded for VALU work.
#define FOLD_TAB_DIGITS 72
struct FoldTabProb { u32 src_off; niels *dst; };
#if ROFL_KG(2)
__global__ void __launch_bounds__(256, 4) k_fold_gens_tab(u32 n_new, u32 nsrc, FoldSeg seg, FoldTabCfg cfg, const niels *tbl16, size_t stride,
                                                       const FoldTabProb *probs, const int16_t *dig /* [prob][nsrc][np][72] */, int unit_first, ge *ext_out) {
    extern __shared__ __align__(16) unsigned char smem[];
    ge *lds = reinterpret_cast<ge *>(smem);
    u32 q = blockIdx.y;
    u32 i = blockIdx.x * 64 + threadIdx.x;
    u32 k = threadIdx.y, K = blockDim.y;
    bool active = i < n_new;
    const niels *src = tbl16 + probs[q].src_off;
    const int16_t *dg = dig + (size_t)q * nsrc * cfg.np * FOLD_TAB_DIGITS;
    gd acc = gd_identity();
    if (active) {
        int lo = seg.lo[k], hi = seg.lo[k + 1] - 1;
        for (int bit = hi; bit >= lo; bit--) {
            acc = gd_double(acc);
            for (u32 h = unit_first ? 1 : 0; h < nsrc; h++) {
                for (u32 pc = 0; pc < cfg.np; pc++) {
                    int d = dg[(h * cfg.np + pc) * FOLD_TAB_DIGITS + bit];
                    if (d != 0) {
                        u32 e = (u32)((d < 0 ? -d : d) - 1) >> 1;
                        acc = gd_madd(acc, gload_nd(&src[(size_t)(pc * cfg.e + e) * stride + (size_t)h * n_new + i]), d < 0);
                    }
                }
            }
        }
        for (int t = 0; t < lo; t++) acc = gd_double(acc);
    }
    if (K > 1) {
        if (k > 0) lds[(k - 1) * 64 + threadIdx.x] = gd_pack(acc);
        __syncthreads();
        if (k == 0)
            for (u32 s2 = 1; s2 < K; s2++) acc = gd_add(acc, gd_unpack(lds[(s2 - 1) * 64 + threadIdx.x]));
    }
    if (active && k == 0) {
        if (unit_first) acc = gd_madd(acc, gload_nd(&src[i]), false);
        // ext_out: leave the point in extended coordinates; k_niels_batch converts eight of them behind one inversion (the inversion chain
        // was 14 % of this kernel's multiplications) on the side stream, while the next round's scalar and sort kernels run
        if (ext_out) store_gd(&ext_out[(size_t)q * n_new + i], acc);
        else gstore_niels(&probs[q].dst[i], gd_to_niels(acc));
    }
}
#endif
// out[i] = affine niels form of ext[i], NB_BATCH points per thread behind ONE field inversion (Montgomery's trick: prefix products, invert,
// walk back): 36 multiplications per point instead of 265.
#define NB_BATCH 8
#if ROFL_KG(2)
__global__ void __launch_bounds__(TPB) k_niels_batch(u32 count, const ge *ext, niels *out) {
    const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t i0 = (size_t)t * NB_BATCH;
    if (i0 >= count) return;
    const u32 nb = (u32)min((size_t)NB_BATCH, (size_t)count - i0);
    fe pre[NB_BATCH];                                   // prefix products, packed
    fd run = fd_unpack(reinterpret_cast<const fe *>(&ext[i0])[2]);      // Z of point 0 (ge = X | Y | Z | T)
    pre[0] = fd_pack(run);
#pragma unroll
    for (u32 k = 1; k < NB_BATCH; k++) {
        if (k < nb) { run = fd_mul(run, fd_unpack(reinterpret_cast<const fe *>(&ext[i0 + k])[2])); pre[k] = fd_pack(run); }
    }
    fd inv = fd_invert(run);
#pragma unroll
    for (int k = NB_BATCH - 1; k >= 0; k--) {
        if ((u32)k >= nb) continue;
        const gd p = load_gd(&ext[i0 + k]);
        const fd zi = k ? fd_mul(inv, fd_unpack(pre[k - 1])) : inv;
        if (k) inv = fd_mul(inv, p.Z);
        gstore_niels(&out[i0 + k], gd_to_niels_zinv(p, zi));
    }
}
#endif

// ================================================================ K5: Pippenger MSM
// Booth-recoded signed c-bit digit of window w: in [-2^(c-1), 2^(c-1)]
// Window layout over bits [0, 254): the top window is c+1 bits wide, [253-c, 254) -- scalars are < l ~ 2^252, so
// its two top bits are (almost) always zero and its digits fill the same 2^(c-1) buckets as a signed c-bit window;
// below it `wide` windows of c bits and the rest of c-1 bits.  (A short top window would put every term into a
// handful of buckets and serialise the accumulation.)  Digits above 2^(c-1) can only occur for scalars >= 2^252
// and are split into two bucket entries.
struct MsmWin { u32 c, W, wide; };
__device__ __forceinline__ void msm_window(const MsmWin &mw, u32 w, u32 &pos, u32 &width) {
    if (w + 1 == mw.W) { pos = 253 - mw.c; width = mw.c + 1; }
    else if (w < mw.wide) { pos = w * mw.c; width = mw.c; }
    else { pos = mw.wide * mw.c + (w - mw.wide) * (mw.c - 1); width = mw.c - 1; }
}
// window table for the fixed-base MSM: wtab[w][g] = 2^(pos_w) * gens[g] for the W windows of `mw` (affine niels)
#if ROFL_KG(2)
__global__ void __launch_bounds__(TPB) k_gens_wtab(u32 total, MsmWin mw, const niels *gens, ndm *wtab, size_t stride) {
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    niels p0 = load_niels(&gens[g]);
    store_ndm(&wtab[g], p0);
    gd cur = gd_unpack(ge_from_niels(p0));
    u32 at = 0;
    for (u32 w = 1; w < mw.W; w++) {
        u32 pos, wid; msm_window(mw, w, pos, wid);
        for (; at + 1 < pos; at++) cur = gd_double_not(cur);
        for (; at < pos; at++) cur = gd_double(cur);
        store_ndm(&wtab[(size_t)w * stride + g], gd_to_niels(cur));
    }
}
#endif

// Booth-recoded signed digit of the window [pos, pos + width): in [-2^(width-1), 2^(width-1)]
__device__ __forceinline__ int msm_digit(const sc &k, u32 wpos, u32 c) {
    int pos = (int)wpos - 1;              // lowest bit needed (b_{pos-1}), -1 for the first window
    u32 need = c + 1;
    u64 bits;
    if (pos < 0) {
        bits = ((u64)k.v[0] | ((u64)k.v[1] << 32)) << 1;
    } else {
        u32 limb = (u32)pos >> 5, off = (u32)pos & 31;
        u64 lo = limb < 8 ? k.v[limb] : 0, mid = limb + 1 < 8 ? k.v[limb + 1] : 0;
        bits = (lo >> off) | (mid << (32 - off));
    }
    u32 x = (u32)(bits & ((1ULL << need) - 1));
    // x = b_{pos-1} + 2*V, V = the c window bits; digit = V_low(c-1 bits) + b_{pos-1} - top * 2^(c-1)
    u32 V = x >> 1, bm1 = x & 1, top = (V >> (c - 1)) & 1;
    return (int)(V & ((1u << (c - 1)) - 1)) + (int)bm1 - (int)(top << (c - 1));
}
// the same digit with the scalar's 32-bit words read from memory at run-time indices: indexing a register-resident sc with a
// run-time limb number makes the compiler spill it to scratch (36 B/lane in the sort kernels, and PMC showed the spills as HBM writes)
__device__ __forceinline__ int msm_digit_mem(const u32 *kw, u32 wpos, u32 c) {
    int pos = (int)wpos - 1;
    u32 need = c + 1;
    u64 bits;
    if (pos < 0) {
        bits = ((u64)kw[0] | ((u64)kw[1] << 32)) << 1;
    } else {
        u32 limb = (u32)pos >> 5, off = (u32)pos & 31;
        u64 lo = limb < 8 ? kw[limb] : 0, mid = limb + 1 < 8 ? kw[limb + 1] : 0;
        bits = (lo >> off) | (mid << (32 - off));
    }
    u32 x = (u32)(bits & ((1ULL << need) - 1));
    u32 V = x >> 1, bm1 = x & 1, top = (V >> (c - 1)) & 1;
    return (int)(V & ((1u << (c - 1)) - 1)) + (int)bm1 - (int)(top << (c - 1));
}
struct MsmProb { const niels *pts; const sc *scal; };   // per problem: points and canonical scalars
// How the (term, window) grid maps to bucket arrays.
//  * lr_nh != 0: "L/R merged" IPP round.  The grid runs over chunks; chunk q owns problems 2q (L) and 2q+1 (R), which
//    share one scalar array over [Gc | Hc] in which every term is non-zero and belongs to exactly one side
//    (k_ipp_scalars, merged): G-side term j is L iff (j mod n_k) >= nh, H-side the other way round.
//  * fb_sets != 0: fixed-base mode.  pts is the window table T[w][i] = 2^(pos_w) P_i (stride fb_stride; the c = 16
//    window layout of msm_window), and windows [s*wps, (s+1)*wps) of a problem share bucket set s: no doublings are
//    left between windows, and the bucket reduction runs over fb_sets sets instead of 16 windows.  With lr each side
//    has its own sets.
struct MsmMap { u32 lr_nh, lr_ng, fb_sets, fb_wps, fb_stride; };
struct MsmItem { u32 pw, entry, ad; };                  // bucket array index, slot entry (point index | sign), |digit|
__device__ __forceinline__ MsmItem msm_item(u32 n, const MsmWin &mw, const MsmMap &mm, const MsmProb *probs, u32 y, u32 i) {
    u32 q = y / mw.W, w = y % mw.W;
    u32 side = 0, p = q;
    if (mm.lr_nh) { bool isL = i < mm.lr_ng ? (i & mm.lr_nh) != 0 : (i & mm.lr_nh) == 0; side = isL ? 0u : 1u; p = 2 * q + side; }
    sc k = gload_sc(&probs[p].scal[i]);
    MsmItem it;
    u32 wpos, wwid; msm_window(mw, w, wpos, wwid);
    int d = msm_digit(k, wpos, wwid);
    if (mm.fb_sets) {
        u32 sets_tot = mm.fb_sets * (mm.lr_nh ? 2u : 1u);
        it.pw = q * sets_tot + side * mm.fb_sets + w % mm.fb_sets;      // set a = windows a, a + sets, a + 2 sets, ... (see k_msm_bin_l1)
        it.entry = w * mm.fb_stride + i;
    } else {
        it.pw = p * mw.W + w;
        it.entry = i;
    }
    it.ad = (u32)(d < 0 ? -d : d);
    if (d < 0) it.entry |= 0x80000000u;
    return it;
}
// Counting sort of (term, window) pairs by bucket.  blockIdx.y = prob * W + window, so the blocks in flight at
// any time hit one 4*B-byte histogram and one 4*n-byte output region: both stay resident in the XCD L2s
// instead of spraying partial-line writes over the whole [prob][W][n] array.
#if ROFL_KG(1)
__global__ void __launch_bounds__(TPB) k_msm_count(u32 n, MsmWin mw, MsmMap mm, const MsmProb *probs, u32 *cnt /* [prob][W][B] */) {
    u32 B = 1u << (mw.c - 1);
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    MsmItem it = msm_item(n, mw, mm, probs, blockIdx.y, i);
    u32 ad = it.ad;
    if (ad > B) { atomicAdd(&cnt[(size_t)it.pw * B + B - 1], 1u); ad -= B; }
    if (ad) atomicAdd(&cnt[(size_t)it.pw * B + ad - 1], 1u);
}
#endif
// One block per (prob, window): exclusive scan of the histogram (off, cursor) and a bucket permutation sorted by
// descending count (perm), so that the 64 lanes of an accumulate wave own buckets of (nearly) equal size.
#if ROFL_KG(1)
// One block per bucket array: (cursor != nullptr) exclusive prefix sums of the counts -> off, cursor; always: perm = the array's buckets by
// descending count (equal-count buckets in any order).  1024 threads, the counting passes coalesced (the first version walked 128 consecutive
// buckets per thread: a 512-byte stride across the lanes, 55 us per fixed-base launch on 16 blocks).
__global__ void __launch_bounds__(1024) k_msm_scan(u32 B, const u32 *cnt, u32 *off, u32 *cursor, u32 *perm) {
    __shared__ u32 part[1024];
    __shared__ u32 hist[256];
    const size_t base = (size_t)blockIdx.x * B;
    const u32 t = threadIdx.x, T = blockDim.x;
    if (t < 256) hist[t] = 0;
    if (cursor) {      // ordered: thread t owns a contiguous run of buckets
        u32 per = (B + T - 1) / T, lo = t * per, hi = lo + per < B ? lo + per : B, s = 0;
        for (u32 i = lo; i < hi; i++) s += cnt[base + i];
        part[t] = s;
        __syncthreads();
        for (u32 d = 1; d < T; d <<= 1) {
            u32 v = t >= d ? part[t - d] : 0;
            __syncthreads();
            part[t] += v;
            __syncthreads();
        }
        u32 run = t ? part[t - 1] : 0;
        for (u32 i = lo; i < hi; i++) { off[base + i] = run; cursor[base + i] = run; run += cnt[base + i]; }
    }
    __syncthreads();
    for (u32 i = t; i < B; i += T) { u32 cv = cnt[base + i]; atomicAdd(&hist[cv > 255 ? 255 : cv], 1u); }
    __syncthreads();
    // descending-count start offsets: start[b] = #buckets with count bin > b
    if (t < 256) part[t] = hist[255 - t];          // reversed, then inclusive scan
    __syncthreads();
    for (u32 d = 1; d < 256; d <<= 1) {
        u32 v = (t < 256 && t >= d) ? part[t - d] : 0;
        __syncthreads();
        if (t < 256) part[t] += v;
        __syncthreads();
    }
    if (t < 256) hist[255 - t] = t ? part[t - 1] : 0;   // exclusive start of bin (255 - t)
    __syncthreads();
    for (u32 i = t; i < B; i += T) {
        u32 cv = cnt[base + i];
        u32 pos = atomicAdd(&hist[cv > 255 ? 255 : cv], 1u);
        perm[base + pos] = i;
    }
}
#endif
#if ROFL_KG(1)
__global__ void __launch_bounds__(TPB) k_msm_scatter(u32 n, MsmWin mw, MsmMap mm, const MsmProb *probs, u32 *cursor, u32 *sorted /* [prob][W][n] */) {
    u32 B = 1u << (mw.c - 1);
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    MsmItem it = msm_item(n, mw, mm, probs, blockIdx.y, i);
    u32 ad = it.ad, pw = it.pw;
    if (ad > B) { u32 pos = atomicAdd(&cursor[(size_t)pw * B + B - 1], 1u); sorted[(size_t)pw * n * 2 + pos] = it.entry; ad -= B; }
    if (ad) { u32 pos = atomicAdd(&cursor[(size_t)pw * B + ad - 1], 1u); sorted[(size_t)pw * n * 2 + pos] = it.entry; }
}
#endif
// Single-pass variant: every bucket owns `cap` slots (HBM capacity instead of a counting pass); cursor doubles as the
// per-bucket count.  The (astronomically rare for hash-derived scalars) entries beyond `cap` go to an overflow list
// that k_msm_overflow adds afterwards; if even that list overflows the host falls back to the two-pass path.
struct MsmOvf { u32 bucket; u32 entry; };
#if ROFL_KG(1)
__global__ void __launch_bounds__(TPB) k_msm_scatter_slots(u32 n, MsmWin mw, MsmMap mm, const MsmProb *probs, u32 *cursor, u32 *slots /* [prob][W][B][cap] */,
                                                           u32 cap, u32 *ovf_count, MsmOvf *ovf, u32 ovf_max) {
    u32 B = 1u << (mw.c - 1);
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    MsmItem it = msm_item(n, mw, mm, probs, blockIdx.y, i);
    u32 ad = it.ad;
    for (int rep = 0; rep < 2; rep++) {
        u32 a1 = rep == 0 ? (ad > B ? B : 0u) : (ad > B ? ad - B : ad);
        if (!a1) continue;
        u32 bi = it.pw * B + a1 - 1;
        u32 pos = atomicAdd(&cursor[bi], 1u);
        if (pos < cap) slots[(size_t)bi * cap + pos] = it.entry;
        else { u32 o = atomicAdd(ovf_count, 1u); if (o < ovf_max) { ovf[o].bucket = bi; ovf[o].entry = it.entry; } }
    }
}
#endif
// Same slot layout, but without one memory-side atomic per item (random-address device atomics retire at ~20 G/s on
// this chip and were the whole cost of the kernel above).  One 1024-thread block owns a tile of the terms of one
// bucket array (fixed-base: the `wps` windows of one set; otherwise one window): it ranks its items in an LDS
// histogram, reserves a range per touched bucket with ONE global atomic per (tile, bucket) -- issued over consecutive
// buckets, i.e. whole 256-byte wave requests -- and then places the items with LDS atomics only.
// grid: x = tile, y = (grid problem q, side, set-or-window).  In L/R mode a block enumerates only the terms of its side.
__device__ __forceinline__ u32 msm_side_term(const MsmMap &mm, u32 side, u32 k) {
    // k-th term of `side` in the merged array [Gc (lr_ng) | Hc (lr_ng)]; lr_nh is a power of two
    u32 half = mm.lr_ng / 2, hside = k >= half ? 1u : 0u, kk = k - hside * half;
    u32 lowmask = mm.lr_nh - 1;
    u32 j = ((kk & ~lowmask) << 1) | (kk & lowmask);
    bool bit = hside ? side == 1 : side == 0;       // G side: L has the bit set; H side: R has it
    if (bit) j |= mm.lr_nh;
    return hside * mm.lr_ng + j;
}
#if ROFL_KG(1)
__global__ void __launch_bounds__(1024) k_msm_scatter_lds(u32 n_side, u32 tile_pts, MsmWin mw, MsmMap mm, const MsmProb *probs, u32 *cursor, u32 *slots,
                                                          u32 cap, u32 *ovf_count, MsmOvf *ovf, u32 ovf_max, u32 dbg) {
    extern __shared__ u32 lcnt[];
    const u32 B = 1u << (mw.c - 1);
    u32 nside = mm.lr_nh ? 2u : 1u;
    u32 per_q = mm.fb_sets ? mm.fb_sets : mw.W;           // bucket arrays per (q, side)
    u32 y = blockIdx.y, a = y % per_q, side = (y / per_q) % nside, q = y / (per_q * nside);
    u32 p = mm.lr_nh ? 2 * q + side : q;
    u32 pw = p * per_q + a;                                // == msm_item's pw for both layouts
    // fixed-base: set a takes the windows a, a + sets, a + 2 sets, ... ; generic: array a is window a
    const u32 wn = mm.fb_sets ? mm.fb_wps : 1u, wstep = mm.fb_sets ? mm.fb_sets : 1u;
    u32 k0 = blockIdx.x * tile_pts, k1 = k0 + tile_pts < n_side ? k0 + tile_pts : n_side;
    const sc *scal = probs[p].scal;
    for (u32 b = threadIdx.x; b < B; b += 1024) lcnt[b] = 0;
    __syncthreads();
    for (u32 k = k0 + threadIdx.x; k < k1; k += 1024) {
        u32 i = mm.lr_nh ? msm_side_term(mm, side, k) : k;
        const u32 *kw = reinterpret_cast<const u32 *>(&scal[i]);
        for (u32 wk = 0; wk < wn; wk++) {
            u32 w = a + wk * wstep;
            u32 wpos, wwid; msm_window(mw, w, wpos, wwid);
            int d = msm_digit_mem(kw, wpos, wwid);
            u32 ad = (u32)(d < 0 ? -d : d);
            if (ad > B) { atomicAdd(&lcnt[B - 1], 1u); ad -= B; }
            if (ad) atomicAdd(&lcnt[ad - 1], 1u);
        }
    }
    __syncthreads();
    for (u32 b = threadIdx.x; b < B; b += 1024) {
        u32 c = lcnt[b];
        lcnt[b] = (c && !(dbg & 1)) ? atomicAdd(&cursor[(size_t)pw * B + b], c) : 0u;
    }
    __syncthreads();
    for (u32 k = k0 + threadIdx.x; k < k1; k += 1024) {
        u32 i = mm.lr_nh ? msm_side_term(mm, side, k) : k;
        const u32 *kw = reinterpret_cast<const u32 *>(&scal[i]);
        for (u32 wk = 0; wk < wn; wk++) {
            u32 w = a + wk * wstep;
            u32 wpos, wwid; msm_window(mw, w, wpos, wwid);
            int d = msm_digit_mem(kw, wpos, wwid);
            u32 ad = (u32)(d < 0 ? -d : d);
            u32 entry = (mm.fb_sets ? w * mm.fb_stride + i : i) | (d < 0 ? 0x80000000u : 0u);
            for (int rep = 0; rep < 2; rep++) {
                u32 a1 = rep == 0 ? (ad > B ? B : 0u) : (ad > B ? ad - B : ad);
                if (!a1) continue;
                u32 pos = atomicAdd(&lcnt[a1 - 1], 1u);
                u32 bi = pw * B + a1 - 1;
                if (dbg & 2) { if (pos == 0xffffffffu) slots[0] = entry; continue; }
                if (pos < cap) slots[(size_t)bi * cap + pos] = entry;
                else { u32 o = atomicAdd(ovf_count, 1u); if (o < ovf_max) { ovf[o].bucket = bi; ovf[o].entry = entry; } }
            }
        }
    }
}
#endif
// ---- Two-level bucket sort of the fixed-base MSM (round 2).  k_msm_scatter_lds places every (term, window) item straight into its
// bucket's slot list: a tile holds ~4 items per bucket, so each (tile, bucket) pair dirties a 64-byte sector for 16 useful bytes
// (PMC: 600 MB written per launch for 134 MB of entries) and the scalars are decoded twice (count pass, place pass).  Here:
//   level 1 (k_msm_bin_l1): one pass over the scalars; an item goes to the COARSE BIN of its bucket (bucket >> fbits; 256 bins per
//     bucket array) through per-bin staging rows in LDS that are flushed in whole 128-byte lines (one atomic per (iteration, bin)
//     reserves a multiple of 32 entries in the bin's HBM region, so every line is written once; the < 32 left-overs of a tile go
//     to the bin's small tail region).  An item is one u32: entry (window-table index, `ebits` bits) | fine bucket
//     (fbits bits) | sign (bit 31).
//   level 2 (k_msm_bin_l2): one block per (bucket array, bin) loads the bin (~8 K items), ranks it by fine bucket in LDS and writes
//     it back IN PLACE as per-bucket lists, plus the count and the absolute list offset of each of its buckets.
// Every entry is written to HBM twice, in full lines (2 x 134 MB), and k_msm_accumulate reads compact lists (cap == MSM_LIST_ABS).
// A bin that outgrows its region (scalars built to collide) raises *overflow and the host repeats the MSM on the slot path.
struct Msm2L { u32 nbins, fbits, ebits, cap_bin, stage; };      // stage = LDS staging slots per bin
#define MSM_LIST_ABS 0xffffffffu
#define MSM_BIN_TAIL 2048u      /* entries of a coarse bin's tail region: the < 32 left-overs of every tile (<= 48 tiles per array), and staging-row spills */
#if ROFL_KG(1)
__global__ void __launch_bounds__(1024) k_msm_bin_l1(u32 n_side, u32 tile_pts, u32 iter_pts, MsmWin mw, MsmMap mm, const MsmProb *probs, u32 *bin_cursor /* [PW][nbins][2] */,
                                                     u32 *bins /* [PW][nbins][cap_bin] */, u32 *tails /* [PW][nbins][MSM_BIN_TAIL] */, Msm2L L, u32 *overflow) {
    extern __shared__ u32 sm2[];
    u32 *lcnt = sm2, *stage = sm2 + L.nbins;                       // [nbins], [nbins][L.stage]
    const u32 B = 1u << (mw.c - 1);
    u32 nside = mm.lr_nh ? 2u : 1u;
    u32 per_q = mm.fb_sets;                                        // bucket arrays per (q, side)
    u32 y = blockIdx.y, a = y % per_q, side = (y / per_q) % nside, q = y / (per_q * nside);
    u32 p = mm.lr_nh ? 2 * q + side : q;
    u32 pw = p * per_q + a;
    // set a takes the windows a, a + sets, a + 2 sets, ...: the three 15-bit windows of the c = 16 layout (12, 13, 14: their digits fill only the
    // lower half of the buckets, at twice the load) land in three different sets instead of all in the last one, whose lower buckets would
    // carry 112 entries against 16 in the upper half (S launch: 1.92 ms against 1.70 for the others)
    u32 k0 = blockIdx.x * tile_pts, k1 = k0 + tile_pts < n_side ? k0 + tile_pts : n_side;
    const sc *scal = probs[p].scal;
    u32 *cur = bin_cursor + (size_t)pw * L.nbins * 2;              // [bin][0] = entries in the bin's main region, [bin][1] = in its tail
    u32 *reg = bins + (size_t)pw * L.nbins * L.cap_bin;
    u32 *tail = tails + (size_t)pw * L.nbins * MSM_BIN_TAIL;
    const u32 fmask = (1u << L.fbits) - 1;
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    for (u32 b = threadIdx.x; b < L.nbins; b += blockDim.x) lcnt[b] = 0;
    __syncthreads();
    for (u32 base = k0; base < k1; base += iter_pts) {
        u32 kend = base + iter_pts < k1 ? base + iter_pts : k1;
        for (u32 k = base + threadIdx.x; k < kend; k += blockDim.x) {
            u32 i = mm.lr_nh ? msm_side_term(mm, side, k) : k;
            const u32 *kw = reinterpret_cast<const u32 *>(&scal[i]);
            for (u32 wk = 0; wk < mm.fb_wps; wk++) {
                u32 w = a + wk * mm.fb_sets;
                u32 wpos, wwid; msm_window(mw, w, wpos, wwid);
                int d = msm_digit_mem(kw, wpos, wwid);
                u32 ad = (u32)(d < 0 ? -d : d);
                u32 entry = (w * mm.fb_stride + i) | (d < 0 ? 0x80000000u : 0u);
                for (int rep = 0; rep < 2; rep++) {
                    u32 a1 = rep == 0 ? (ad > B ? B : 0u) : (ad > B ? ad - B : ad);
                    if (!a1) continue;
                    u32 bkt = a1 - 1, bin = bkt >> L.fbits;
                    u32 item = entry | ((bkt & fmask) << L.ebits);
                    u32 pos = atomicAdd(&lcnt[bin], 1u);
                    if (pos < L.stage) stage[bin * L.stage + pos] = item;
                    else {                                          // a staging row ran full (skewed digits): one entry straight to the bin's tail
                        u32 g = atomicAdd(&cur[2 * bin + 1], 1u);
                        if (g < MSM_BIN_TAIL) tail[(size_t)bin * MSM_BIN_TAIL + g] = item; else *(volatile u32 *)overflow = 1u;
                    }
                }
            }
        }
        __syncthreads();
        // Flush whole 128-byte lines only: every reservation in the main region is a multiple of 32 entries, so runs start
        // line-aligned and no line is written twice.  What is left (< 32) stays at the front of the row for the next iteration;
        // after the last iteration it goes to the bin's tail.
        bool last_iter = kend == k1;
        // a wave owns the bins wave, wave + nwaves, ...: lane l reserves for the l-th of them, so the global atomics of a flush are
        // one round trip per wave instead of one per bin
        u32 my_bin = wave + lane * nwaves, my_c = 0, my_gm = 0, my_gt = 0;
        if (my_bin < L.nbins) {
            my_c = lcnt[my_bin]; if (my_c > L.stage) my_c = L.stage;
            u32 full = my_c & ~31u, rem = my_c - full;
            if (full) my_gm = atomicAdd(&cur[2 * my_bin], full);
            if (last_iter && rem) my_gt = atomicAdd(&cur[2 * my_bin + 1], rem);
        }
        u32 kbin = 0;
        for (u32 bin = wave; bin < L.nbins; bin += nwaves, kbin++) {
            u32 c = __shfl(my_c, kbin), g = __shfl(my_gm, kbin), gt = __shfl(my_gt, kbin);
            u32 full = c & ~31u, rem = c - full;
            if (full) {
                bool fits = g + full <= L.cap_bin;
                if (!fits && lane == 0) *(volatile u32 *)overflow = 1u;      // mapped host memory: plain, idempotent store
                for (u32 j = lane; j < full && fits; j += 64) reg[(size_t)bin * L.cap_bin + g + j] = stage[bin * L.stage + j];
            }
            u32 carry = (lane < rem) ? stage[bin * L.stage + full + lane] : 0u;      // rem < 32 <= 64 lanes
            if (last_iter) {
                if (rem) {
                    if (gt + rem <= MSM_BIN_TAIL) { if (lane < rem) tail[(size_t)bin * MSM_BIN_TAIL + gt + lane] = carry; }
                    else if (lane == 0) *(volatile u32 *)overflow = 1u;
                }
            } else {
                if (lane < rem) stage[bin * L.stage + lane] = carry;
                if (lane == 0) lcnt[bin] = rem;
            }
        }
        __syncthreads();
    }
}
#endif
#if ROFL_KG(1)
// wps > 1: a bucket's list comes out ordered by WINDOW SLOT (entry >> wshift = which of the wps windows of its bucket array the entry belongs
// to): the accumulation walks the lists in order, so the threads of a launch gather from one or two window slices of the table at a time
// instead of all of them (profiles/r06_experiments.txt item 14) -- the ranking key is (bucket, slot) instead of the bucket, nothing else changes.
__global__ void __launch_bounds__(512) k_msm_bin_l2(Msm2L L, u32 B, const u32 *bin_cursor, u32 *bins, const u32 *tails, u32 *cnt /* [PW][B] */, u32 *off /* [PW][B] */, u32 *overflow,
                                                    u32 wps, u32 wshift) {
    extern __shared__ u32 sm2[];
    const u32 FBK = 1u << L.fbits;                                  // buckets of a coarse bin
    const u32 FB = FBK * wps;                                       // ranking keys: (bucket, window slot)
    u32 *hist = sm2, *ofs = sm2 + FB, *out = sm2 + 2 * FB;          // [FB], [FB], [cap_bin]
    u32 bin = blockIdx.x, pw = blockIdx.y;
    u32 n = bin_cursor[((size_t)pw * L.nbins + bin) * 2], n2 = bin_cursor[((size_t)pw * L.nbins + bin) * 2 + 1];
    size_t rbase = ((size_t)pw * L.nbins + bin) * L.cap_bin;
    // A bin that outgrew its region (scalars built to collide: a proof with a = 0 gives every G term the same scalar) has HOLES: level 1
    // skips a reservation that straddles cap_bin and a tail flush beyond MSM_BIN_TAIL, so parts of [0, n) were never written and hold
    // stale words of an earlier launch (or of a fresh allocation).  Such a bin must not reach the accumulation, which gathers
    // window-table entries by these words before the host has seen the flag: its buckets are reported empty and the MSM is repeated
    // on the slot path anyway.
    if (n > L.cap_bin || n2 > MSM_BIN_TAIL || n + n2 > L.cap_bin) {
        if (threadIdx.x == 0) *(volatile u32 *)overflow = 1u;
        for (u32 f = threadIdx.x; f < (1u << L.fbits); f += blockDim.x) {
            size_t bi = (size_t)pw * B + (size_t)bin * (1u << L.fbits) + f;
            cnt[bi] = 0; off[bi] = (u32)rbase;
        }
        return;
    }
    u32 *reg = bins + rbase;
    const u32 *tl = tails + ((size_t)pw * L.nbins + bin) * MSM_BIN_TAIL;
    const u32 fmask = FBK - 1, keep = ~(fmask << L.ebits), emask = (1u << L.ebits) - 1;
    auto key = [&](u32 v) { u32 f = (v >> L.ebits) & fmask; if (wps <= 1) return f; u32 ws = (v & emask) >> wshift; return f * wps + (ws < wps ? ws : wps - 1); };
    for (u32 f = threadIdx.x; f < FB; f += blockDim.x) hist[f] = 0;
    __syncthreads();
    for (u32 j = threadIdx.x; j < n + n2; j += blockDim.x) atomicAdd(&hist[key(j < n ? reg[j] : tl[j - n])], 1u);
    __syncthreads();
    if (threadIdx.x < 64) {                                          // exclusive scan of FB <= 128 counters by one wave
        u32 per = (FB + 63) / 64, lo = threadIdx.x * per, sum = 0;
        for (u32 f = lo; f < lo + per && f < FB; f++) sum += hist[f];
        u32 incl = sum;
        for (u32 dlt = 1; dlt < 64; dlt <<= 1) { u32 v = __shfl_up(incl, dlt); if (threadIdx.x >= dlt) incl += v; }
        u32 run = incl - sum;
        for (u32 f = lo; f < lo + per && f < FB; f++) { ofs[f] = run; run += hist[f]; }
    }
    __syncthreads();
    for (u32 f = threadIdx.x; f < FBK; f += blockDim.x) {
        size_t bi = (size_t)pw * B + (size_t)bin * FBK + f;
        u32 tot = 0; for (u32 q = 0; q < wps; q++) tot += hist[f * wps + q];
        cnt[bi] = tot;
        off[bi] = (u32)(rbase + ofs[f * wps]);
    }
    __syncthreads();
    for (u32 j = threadIdx.x; j < n + n2; j += blockDim.x) {
        u32 v = j < n ? reg[j] : tl[j - n];
        u32 pos = atomicAdd(&ofs[key(v)], 1u);
        out[pos] = v & keep;
    }
    __syncthreads();
    for (u32 j = threadIdx.x; j < n + n2; j += blockDim.x) reg[j] = out[j];
}
#endif
// one 64-lane wave; lane l owns the overflow entries whose bucket index is l mod 64 (no two lanes share a bucket)
#if ROFL_KG(1)
__global__ void k_msm_overflow(u32 W, u32 B, u32 pstep, const MsmProb *probs, const u32 *ovf_count, const MsmOvf *ovf, u32 ovf_max, ge *buckets, int fb) {
    if (blockIdx.x) return;
    u32 cnt = *ovf_count; if (cnt > ovf_max) cnt = ovf_max;
    for (u32 o = 0; o < cnt; o++) {
        u32 bi = ovf[o].bucket, v = ovf[o].entry, p = bi / (W * B) * pstep;
        if ((bi & 63u) != threadIdx.x) continue;
        gd acc = load_gd(&buckets[bi]);
        u32 idx = v & 0x7fffffffu;
        acc = gd_madd(acc, fb ? gload_ndm(reinterpret_cast<const ndm *>(probs[p].pts) + idx) : gload_nd(&probs[p].pts[idx]), (v >> 31) != 0);
        store_gd(&buckets[bi], acc);
    }
}
#endif
// one thread per bucket: sum its points.  buckets [prob][W][B] extended.
// blockIdx.y = grid problem q owning W bucket arrays; its points are probs[q * pstep].pts (pstep = 2 for merged L/R pairs)
template <bool FB> __device__ __forceinline__ void msm_accumulate_body(u32 n, u32 c, u32 W, u32 pstep, const MsmProb *probs, const u32 *cnt, const u32 *off,
                                 const u32 *sorted, const u32 *perm, ge *buckets, u32 cap, u32 idx_mask, u32 balance) {
    u32 p = blockIdx.y, B = 1u << (c - 1);
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if ((balance & 2u) && B >= TPB) {
        // set-major dispatch: the launch's blocks in the order (bucket array of a problem, problem, block) instead of (problem, array, block), so that
        // the blocks resident at one time belong to one or two arrays of every problem -- with window-ordered lists (k_msm_bin_l2) they gather
        // from a few window slices of the table at a time (the problems of a launch share the table).  gridDim.x = W * B / TPB exactly.
        const u32 nbs = B / TPB, lin = blockIdx.y * gridDim.x + blockIdx.x, per_set = nbs * gridDim.y;
        const u32 ws = lin / per_set, rem = lin - ws * per_set;
        p = rem / nbs;
        t = ws * B + (rem - p * nbs) * TPB + threadIdx.x;
    }
    if (t >= W * B) return;
    u32 w = t / B;
    if ((balance & 1u) && B >= 1024) {
        // perm lists an array's buckets by descending count, so consecutive waves -- and blocks -- get lighter and lighter and the launch
        // ends on a long tail of half-empty CUs.  Give every block the same work instead: block j of an array takes the waves ranked
        // j, nw-1-j, nw/2-1-j and nw/2+j of its nw (two symmetric pairs around the median).  No rotation of the classes over a block's
        // waves: the dispatcher already starts each of the four blocks that share a CU (256 apart in launch order) on a different SIMD
        // (timeline: waves 0..3 -> SIMDs 1,3,0,2 / 3,0,2,1 / 0,2,1,3 / 2,1,3,0), so every SIMD gets one wave of each class; rotating by
        // (block id >> 8) cancels exactly that and costs 20 %.
        u32 tb = t & (B - 1), nw = B / 64, j = tb / TPB, k = (tb / 64) & 3;
        u32 sw = k == 0 ? j : k == 1 ? nw - 1 - j : k == 2 ? nw / 2 - 1 - j : nw / 2 + j;
        t = w * B + sw * 64 + (tb & 63);
    }
    size_t bi = ((size_t)p * W + w) * B + perm[(size_t)p * W * B + t];
    u32 num = cnt[bi];
    const u32 *lst;
    if (cap == MSM_LIST_ABS) lst = sorted + off[bi];                        // two-level sort: compact lists, absolute offsets
    else if (cap) { lst = sorted + bi * cap; if (num > cap) num = cap; }        // slot mode: `sorted` is the slot array
    else lst = sorted + ((size_t)p * W + w) * n * 2 + off[bi];   // stride 2n: a top-window digit may emit two entries
    const niels *pts = probs[p * pstep].pts;
    gd acc = gd_identity();
    // (a software-pipelined variant that keeps the next point load in flight costs 12 VGPRs -> 3 waves/SIMD and was slower;
    //  forcing 5 waves/SIMD -- 96 VGPRs, 136 B/lane of scratch -- takes 1.65x as long)
    // the list entry of the NEXT addition is fetched one iteration ahead (one register): the entry -> point gather chain is two
    // dependent memory latencies per addition otherwise
#if ROFL_ACC_LIST_CHUNK
    // The 64 lanes of a wave stream 64 different lists; with 16 waves per CU that is 128 KB of live list lines against a 32 KB L1 and this
    // CU's share of L2, and a 128-byte line read four bytes at a time is evicted between two of its 32 reads about every other time
    // (PMC, profiles/r04_experiments.txt: 5.8 GB per launch against the 3.8 GB that the table records and list entries amount to; the gather
    // micro-benchmark, which has no lists, fetches exactly its 128 bytes per record).  This variant takes the entries two (ROFL_ACC_LIST_CHUNK
    // = 2) or four (= 4) at a time from the aligned chunk that holds them (a chunk may start before the list and end after it: both lie inside
    // the sort's own arrays) and requests the next chunk when the last entry of the current one has been taken, one addition ahead.
    // MEASURED: four at a time 5.79 -> 4.99 GB per launch and 1.225 -> 1.312 ms (the selects and the branch cost more than the re-fetches);
    // not the default.
    const uintptr_t la = reinterpret_cast<uintptr_t>(lst);
#if ROFL_ACC_LIST_CHUNK == 2
    const uint2 *ch = reinterpret_cast<const uint2 *>(la & ~(uintptr_t)7);
    u32 k = (u32)(la >> 2) & 1u;
    uint2 cur = num ? *ch : make_uint2(0, 0);
    for (u32 e = 0; e < num; e++) {
        const u32 v = k ? cur.y : cur.x;
        if (k) { if (e + 1 < num) cur = *++ch; }
        k ^= 1u;
        u32 idx = v & idx_mask;
        acc = gd_madd(acc, FB ? gload_ndm(reinterpret_cast<const ndm *>(pts) + idx) : gload_nd(&pts[idx]), (v >> 31) != 0);
    }
#else
    const uint4 *ch = reinterpret_cast<const uint4 *>(la & ~(uintptr_t)15);
    u32 k = (u32)(la >> 2) & 3u;
    uint4 cur = num ? *ch : make_uint4(0, 0, 0, 0);
    for (u32 e = 0; e < num; e++) {
        const u32 v = k == 0 ? cur.x : k == 1 ? cur.y : k == 2 ? cur.z : cur.w;
        if (k == 3) { if (e + 1 < num) cur = *++ch; k = 0; } else k++;
        u32 idx = v & idx_mask;
        acc = gd_madd(acc, FB ? gload_ndm(reinterpret_cast<const ndm *>(pts) + idx) : gload_nd(&pts[idx]), (v >> 31) != 0);
    }
#endif
#else
    u32 vnext = num ? lst[0] : 0u;
    for (u32 e = 0; e < num; e++) {
        u32 v = vnext;
        if (e + 1 < num) vnext = lst[e + 1];
        u32 idx = v & idx_mask;
        acc = gd_madd(acc, FB ? gload_ndm(reinterpret_cast<const ndm *>(pts) + idx) : gload_nd(&pts[idx]), (v >> 31) != 0);
    }
#endif
    store_gd(&buckets[bi], acc);
}
#if ROFL_KG(1)
__global__ void __launch_bounds__(TPB) k_msm_accumulate_fb(u32 n, u32 c, u32 W, u32 pstep, const MsmProb *probs, const u32 *cnt, const u32 *off,
                                 const u32 *sorted, const u32 *perm, ge *buckets, u32 cap, u32 idx_mask, u32 balance) {
    msm_accumulate_body<true>(n, c, W, pstep, probs, cnt, off, sorted, perm, buckets, cap, idx_mask, balance);
}
#endif
// debugging aid (ROFL_DBG_ACC_TIMELINE): the same launch with one record per wave -- start / end on the 100 MHz wall clock, HW_ID, XCC_ID
#if ROFL_KG(1)
__global__ void __launch_bounds__(TPB) k_msm_accumulate_fb_dbg(u32 n, u32 c, u32 W, u32 pstep, const MsmProb *probs, const u32 *cnt, const u32 *off,
                                 const u32 *sorted, const u32 *perm, ge *buckets, u32 cap, u32 idx_mask, u32 balance, unsigned long long *rec) {
    unsigned long long t0 = wall_clock64();
    msm_accumulate_body<true>(n, c, W, pstep, probs, cnt, off, sorted, perm, buckets, cap, idx_mask, balance);
    unsigned long long t1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) {
        u32 hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        size_t wv = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (TPB / 64) + threadIdx.x / 64;
        rec[wv * 4 + 0] = t0; rec[wv * 4 + 1] = t1; rec[wv * 4 + 2] = hw; rec[wv * 4 + 3] = xcc;
    }
}
#endif
#if ROFL_KG(1)
__global__ void __launch_bounds__(TPB) k_msm_accumulate_gen(u32 n, u32 c, u32 W, u32 pstep, const MsmProb *probs, const u32 *cnt, const u32 *off,
                                 const u32 *sorted, const u32 *perm, ge *buckets, u32 cap, u32 idx_mask, u32 balance) {
    msm_accumulate_body<false>(n, c, W, pstep, probs, cnt, off, sorted, perm, buckets, cap, idx_mask, balance);
}
#endif
// Bucket reduction without doublings: sum_b (b+1) B_b = S + sum_l 2^l D_l, D_l = sum of buckets whose
// index has bit l set.  One 8-ary tree level per launch:
//   role 0 threads: 8 children of S_in -> S_out, and the three new bit-sums (11 adds)
//   role r>=1     : carried bit-sum r-1: 8 children -> 1 (7 adds)
// in: S_in [PW][E], C_in [PW][nb][E];  out: S_out [PW][E/8], C_out [PW][nb+3][E/8]
// one work item of a tree level; pointers are already offset to this (prob, window)
__device__ __forceinline__ void msm_reduce_item(u32 E, u32 nb, const ge *S_in, const ge *C_in, ge *S_out, ge *C_out, u32 item) {
    u32 E8 = E / 8, role = item / E8, g = item % E8;
    if (role == 0) {
        const ge *s = S_in + (size_t)g * 8;
        gd p0 = load_gd(&s[0]), p1 = load_gd(&s[1]);
        gd q0 = gd_add(p0, p1); gd D0 = p1;
        p0 = load_gd(&s[2]); p1 = load_gd(&s[3]);
        gd q1 = gd_add(p0, p1); D0 = gd_add(D0, p1);
        p0 = load_gd(&s[4]); p1 = load_gd(&s[5]);
        gd q2 = gd_add(p0, p1); D0 = gd_add(D0, p1);
        p0 = load_gd(&s[6]); p1 = load_gd(&s[7]);
        gd q3 = gd_add(p0, p1); D0 = gd_add(D0, p1);
        store_gd(&C_out[(size_t)(nb + 0) * E8 + g], D0);
        gd D1 = gd_add(q1, q3);
        store_gd(&C_out[(size_t)(nb + 1) * E8 + g], D1);
        gd r1 = gd_add(q2, q3);
        store_gd(&C_out[(size_t)(nb + 2) * E8 + g], r1);
        gd S = gd_add(gd_add(q0, q1), r1);
        store_gd(&S_out[g], S);
    } else {
        const ge *s = C_in + (size_t)(role - 1) * E + (size_t)g * 8;
        gd acc = gd_add(load_gd(&s[0]), load_gd(&s[1]));
#pragma unroll 1
        for (int k = 2; k < 8; k++) acc = gd_add(acc, load_gd(&s[k]));
        store_gd(&C_out[(size_t)(role - 1) * E8 + g], acc);
    }
}
// The same tree level with the four outputs of an 8-group (S and the three new bit-sums) on four work items: 7 / 3 / 3 / 3
// additions with two live points each instead of one item doing 11 additions with seven live points (256 VGPRs + AGPR
// spills, 1 wave/SIMD).  item = role * E8 + g; roles 0..3 as described, role r >= 4 = carried bit-sum r - 4.
__device__ __forceinline__ void msm_reduce_item_split(u32 E, u32 nb, const ge *S_in, const ge *C_in, ge *S_out, ge *C_out, u32 item) {
    u32 E8 = E / 8, role = item / E8, g = item % E8;
    if (role >= 4) {      // carried bit-sum role-4: 8 children -> 1
        const ge *cs = C_in + (size_t)(role - 4) * E + (size_t)g * 8;
        gd acc = gd_add(load_gd(&cs[0]), load_gd(&cs[1]));
#pragma unroll 1
        for (int t = 2; t < 8; t++) acc = gd_add(acc, load_gd(&cs[t]));
        store_gd(&C_out[(size_t)(role - 4) * E8 + g], acc);
        return;
    }
    const ge *s = S_in + (size_t)g * 8;
    if (role == 0) {
        gd acc = gd_add(load_gd(&s[0]), load_gd(&s[1]));
#pragma unroll 1
        for (int t = 2; t < 8; t++) acc = gd_add(acc, load_gd(&s[t]));
        store_gd(&S_out[g], acc);
    } else {
        // bit (role-1) of the child index: children {1,3,5,7}, {2,3,6,7}, {4,5,6,7}
        u32 b = role - 1, st = 1u << b;
        u32 i0 = st, i1 = b == 2 ? 5 : 3, i2 = b == 0 ? 5 : 6, i3 = 7;
        gd acc = gd_add(load_gd(&s[i0]), load_gd(&s[i1]));
        acc = gd_add(acc, load_gd(&s[i2]));
        acc = gd_add(acc, load_gd(&s[i3]));
        store_gd(&C_out[(size_t)(nb + b) * E8 + g], acc);
    }
}
#if ROFL_KG(1)
__global__ void __launch_bounds__(TPB) k_msm_reduce_level(u32 E, u32 nb, const ge *S_in, const ge *C_in, ge *S_out, ge *C_out, int split) {
    u32 pw = blockIdx.y, E8 = E / 8;
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (split) {      // the four outputs of an 8-group on four threads: 7 / 3 / 3 / 3 additions deep instead of 11 on one thread
        if (t >= E8 * (4 + nb)) return;
        msm_reduce_item_split(E, nb, S_in + (size_t)pw * E, C_in + (size_t)pw * nb * E, S_out + (size_t)pw * E8, C_out + (size_t)pw * (nb + 3) * E8, t);
        return;
    }
    if (t >= E8 * (1 + nb)) return;
    msm_reduce_item(E, nb, S_in + (size_t)pw * E, C_in + (size_t)pw * nb * E, S_out + (size_t)pw * E8, C_out + (size_t)pw * (nb + 3) * E8, t);
}
#endif
// One binary level of the bit-sum tree inside a block: S'[g] = S[2g] + S[2g+1], new bit-sum nb = S[2g+1], carried bit-sums pairwise.
// While the level has more work items than a quarter of the block, one thread per item; from there on (the last levels, where most of
// the block would idle: E2 (1 + nb) <= blockDim / 4) FOUR lanes per item -- lane q adds coordinate q (quad26.hpp: three
// multiplications deep instead of nine; 7.3 -> ~3.5 us per level in k_msm_small, whose nine levels were half of its 130 us).
__device__ __forceinline__ void msm_binary_level(u32 E, u32 nb, const ge *si, const ge *ci, ge *so, ge *co) {
    const u32 E2 = E / 2, items = E2 * (1 + nb);
    if (items * 4 <= blockDim.x) {
        const u32 item = threadIdx.x >> 2, q = threadIdx.x & 3;
        if (item < items) {
            const u32 role = item / E2, g = item % E2;
            const ge *pa, *pb; ge *dst;
            if (role == 0) { pa = &si[2 * g]; pb = &si[2 * g + 1]; dst = &so[g]; }
            else { const ge *cc = ci + (size_t)(role - 1) * E; pa = &cc[2 * g]; pb = &cc[2 * g + 1]; dst = &co[(size_t)(role - 1) * E2 + g]; }
            const fe fa = reinterpret_cast<const fe *>(pa)[q], fb = reinterpret_cast<const fe *>(pb)[q];
            gq a, b; a.v = fd_unpack(fa); b.v = fd_unpack(fb);
            reinterpret_cast<fe *>(dst)[q] = fd_pack(gq_add(a, b, q).v);
            if (role == 0) reinterpret_cast<fe *>(&co[(size_t)nb * E2 + g])[q] = fb;
        }
        return;
    }
    for (u32 item = threadIdx.x; item < items; item += blockDim.x) {
        u32 role = item / E2, g = item % E2;
        if (role == 0) {
            ge lo = si[2 * g], hi = si[2 * g + 1];
            so[g] = gd_pack(gd_add(gd_unpack(lo), gd_unpack(hi)));
            co[(size_t)nb * E2 + g] = hi;
        } else {
            const ge *cc = ci + (size_t)(role - 1) * E;
            co[(size_t)(role - 1) * E2 + g] = gd_pack(gd_add(gd_unpack(cc[2 * g]), gd_unpack(cc[2 * g + 1])));
        }
    }
}
// All remaining levels (E <= 512) of one (prob, window) in a single block through LDS: one launch instead of three
// or four latency-bound ones.  First level 8-ary (global -> LDS), the rest binary (one addition deep per level):
// S'[g] = S[2g] + S[2g+1], new bit-sum = S[2g+1], carried bit-sums pairwise.  Output: S_fin [PW], C_fin [PW][nb_final].
__device__ __forceinline__ void msm_reduce_fused_body(u32 E, u32 nb, const ge *si, const ge *ci, ge *fin_s, ge *fin_c, unsigned char *smem) {
    ge *buf0 = reinterpret_cast<ge *>(smem);
    // level 1: 8-ary
    u32 E8 = E / 8;
    ge *so = E8 == 1 ? fin_s : buf0, *co = E8 == 1 ? fin_c : buf0 + E8;
    // (latency-bound: the four outputs of an 8-group -- S and the three new bit-sums -- go to four threads, 7 / 3 additions
    //  deep instead of one thread doing all 11)
    for (u32 item = threadIdx.x; item < E8 * (4 + nb); item += blockDim.x) msm_reduce_item_split(E, nb, si, ci, so, co, item);
    __syncthreads();
    if (E8 == 1) return;
    u32 half = E8 * (1 + nb + 3);
    ge *bufs[2] = {buf0, buf0 + half};
    int sel = 1;
    si = so; ci = co; E = E8; nb += 3;
    while (E > 1) {
        u32 E2 = E / 2;
        if (E2 == 1) { so = fin_s; co = fin_c; } else { so = bufs[sel]; co = bufs[sel] + E2; }
        msm_binary_level(E, nb, si, ci, so, co);
        __syncthreads();
        si = so; ci = co; E = E2; nb += 1; sel ^= 1;
    }
}
#if ROFL_KG(1)
__global__ void __launch_bounds__(768) k_msm_reduce_fused(u32 E, u32 nb, const ge *S_in, const ge *C_in, ge *S_fin, ge *C_fin, u32 nb_final) {
    extern __shared__ __align__(16) unsigned char smem[];
    u32 pw = blockIdx.x;
    msm_reduce_fused_body(E, nb, S_in + (size_t)pw * E, C_in + (size_t)pw * nb * E, S_fin + pw, C_fin + (size_t)pw * nb_final, smem);
}
#endif
// Two-launch bucket reduction (round 2) for arrays of B = 512 * G buckets: k_msm_reduce_fused treats every run of 512 buckets as an
// array of its own (E = 512, nb = 0: one block each, 64 x 16 blocks at B = 32768 instead of the 16 blocks that used to finish the
// tree), then this kernel combines the G groups of an array: S = sum_g S_g; bit-sums 0..8 = sum_g D_(l,g); bit-sum 9 + t = sum of
// the S_g of the groups whose index has bit t set (a bucket's 0-based index is g * 512 + low bits).  One block per array, one
// output per threadIdx.y, G / 2 lanes per output + LDS tree.
#if ROFL_KG(1)
__global__ void __launch_bounds__(512) k_msm_reduce_groups(u32 G, u32 gbits, const ge *GS /* [PW][G] */, const ge *GC /* [PW][G][9] */, ge *S_fin, ge *C_fin, u32 nb_final) {
    extern __shared__ __align__(16) unsigned char smem[];
    ge *lds = reinterpret_cast<ge *>(smem);                  // [nout][half]
    u32 pw = blockIdx.x, o = threadIdx.y, t = threadIdx.x, half = blockDim.x;      // half = max(G / 2, 1)
    const ge *gs = GS + (size_t)pw * G, *gc = GC + (size_t)pw * G * 9;
    gd acc = gd_identity();
    for (u32 g = t; g < G; g += half) {
        if (o == 0) acc = gd_add(acc, load_gd(&gs[g]));
        else if (o <= 9) acc = gd_add(acc, load_gd(&gc[(size_t)g * 9 + (o - 1)]));
        else if ((g >> (o - 10)) & 1) acc = gd_add(acc, load_gd(&gs[g]));
    }
    lds[o * half + t] = gd_pack(acc);
    __syncthreads();
    for (u32 s2 = half / 2; s2 >= 1; s2 >>= 1) {
        if (t < s2) lds[o * half + t] = gd_pack(gd_add(gd_unpack(lds[o * half + t]), gd_unpack(lds[o * half + t + s2])));
        __syncthreads();
    }
    if (t == 0) {
        if (o == 0) store_ge(&S_fin[pw], lds[0]);
        else store_ge(&C_fin[(size_t)pw * nb_final + (o - 1)], lds[o * half]);
    }
}
#endif
// The same reduction as msm_reduce_fused_body with binary levels from the start (first level global -> LDS): log2(E) levels of ONE
// addition each instead of a 7-deep 8-ary first level -- for a block that is alone on its CU the depth of the dependency chain is the
// cost (E = 512: 9 additions deep instead of 13).  Level k: S'[g] = S[2g] + S[2g+1], new bit-sum k = S[2g+1], carried bit-sums pairwise.
__device__ __forceinline__ void msm_reduce_binary_body(u32 E, const ge *si_g, ge *fin_s, ge *fin_c, unsigned char *smem) {
    ge *buf0 = reinterpret_cast<ge *>(smem);
    u32 E2 = E / 2;
    ge *bufs[2] = {buf0, buf0 + (size_t)E2 * 2};                   // level-1 output: S[E2] | C[1][E2]; the next one is smaller
    {
        ge *so = bufs[0], *co = bufs[0] + E2;
        for (u32 g = threadIdx.x; g < E2; g += blockDim.x) {
            ge lo = load_ge(&si_g[2 * g]), hi = load_ge(&si_g[2 * g + 1]);
            so[g] = gd_pack(gd_add(gd_unpack(lo), gd_unpack(hi)));
            co[g] = hi;
        }
    }
    __syncthreads();
    const ge *si = bufs[0], *ci = bufs[0] + E2;
    u32 nb = 1; int sel = 1;
    E = E2;
    while (E > 1) {
        E2 = E / 2;
        ge *so, *co;
        if (E2 == 1) { so = fin_s; co = fin_c; } else { so = bufs[sel]; co = bufs[sel] + E2; }
        msm_binary_level(E, nb, si, ci, so, co);
        __syncthreads();
        si = so; ci = co; E = E2; nb += 1; sel ^= 1;
    }
}
// msm_reduce_binary_body for G bucket arrays of E buckets side by side (G * E threads): the arrays are treated as ONE array of G * E entries --
// a pair (2g, 2g + 1) never straddles two trees because E is a power of two -- so the lanes that a single 64-bucket tree leaves idle from its
// second level on carry the other trees (four windows: ~6 + 9 wave-level additions instead of 4 x (3 + 3)).  LDS is what limits the blocks per CU
// here, so (a) the first two levels are taken straight from the bucket arrays in global memory (S2[g] = b[4g] + .. + b[4g+3], bit-sum 0 =
// b[4g+1] + b[4g+3], bit-sum 1 = b[4g+2] + b[4g+3]: the level-1 state is never stored) and (b) every later level works IN PLACE (read, barrier,
// write): 3/4 G E entries instead of 7/4 G E.  The last level (one entry per tree, bit-sum-major) is stored window-major, and only for the
// `gv` <= G arrays that exist (the spare ones read the last real array again: defined values, discarded).
__device__ __forceinline__ void msm_binary_level_inplace(u32 E, u32 nb, ge *buf) {      // S[E] | C[nb][E]  ->  S[E/2] | C[nb + 1][E/2], same buffer
    const u32 E2 = E / 2, items = E2 * (1 + nb);
    const ge *si = buf, *ci = buf + E;
    if (items * 4 <= blockDim.x) {
        const u32 item = threadIdx.x >> 2, q = threadIdx.x & 3;
        const bool on = item < items;
        const u32 role = on ? item / E2 : 0, g = on ? item % E2 : 0;
        fe res, hi;
        if (on) {
            const ge *pa, *pb;
            if (role == 0) { pa = &si[2 * g]; pb = &si[2 * g + 1]; }
            else { const ge *cc = ci + (size_t)(role - 1) * E; pa = &cc[2 * g]; pb = &cc[2 * g + 1]; }
            const fe fa = reinterpret_cast<const fe *>(pa)[q]; hi = reinterpret_cast<const fe *>(pb)[q];
            gq a, b; a.v = fd_unpack(fa); b.v = fd_unpack(hi);
            res = fd_pack(gq_add(a, b, q).v);
        }
        __syncthreads();
        if (on) {
            ge *so = buf, *co = buf + E2;
            ge *dst = role == 0 ? &so[g] : &co[(size_t)(role - 1) * E2 + g];
            reinterpret_cast<fe *>(dst)[q] = res;
            if (role == 0) reinterpret_cast<fe *>(&co[(size_t)nb * E2 + g])[q] = hi;
        }
        __syncthreads();
        return;
    }
    // one item per thread (items <= blockDim for the callers of this function: E <= blockDim / 2 ... checked by the caller's geometry)
    const u32 item = threadIdx.x;
    const bool on = item < items;
    const u32 role = on ? item / E2 : 0, g = on ? item % E2 : 0;
    ge res, hi;
    if (on) {
        const ge *cc = role == 0 ? si : ci + (size_t)(role - 1) * E;
        const ge lo = cc[2 * g]; hi = cc[2 * g + 1];
        res = gd_pack(gd_add(gd_unpack(lo), gd_unpack(hi)));
    }
    __syncthreads();
    if (on) {
        ge *so = buf, *co = buf + E2;
        if (role == 0) { so[g] = res; co[(size_t)nb * E2 + g] = hi; }
        else co[(size_t)(role - 1) * E2 + g] = res;
    }
    __syncthreads();
}
__device__ __forceinline__ void msm_reduce_binary_multi(u32 E, u32 G, u32 gv, const ge *si_g, ge *fin_s, ge *fin_c, u32 nb_final, unsigned char *smem) {
    ge *buf = reinterpret_cast<ge *>(smem);
    const u32 Et = G * E, E4 = Et / 4;                             // (blockDim == Et; E >= 8)
    // levels 1 + 2 from global memory, one addition deep each: A[g] = b[4g] + b[4g+1] (-> S slot), bit-sum 0, bit-sum 1; then S2[g] = A[g] + bit-sum 1
    {
        const u32 item = threadIdx.x;
        if (item < 3 * E4) {
            const u32 role = item / E4, g = item % E4;
            const u32 tree = (4 * g) / E, in_tree = (4 * g) % E, src_tree = tree < gv ? tree : gv - 1;
            const ge *b = si_g + (size_t)src_tree * E + in_tree;
            const ge x = load_ge(&b[role == 0 ? 0 : role == 1 ? 1 : 2]), y = load_ge(&b[role == 0 ? 1 : 3]);
            buf[(size_t)role * E4 + g] = gd_pack(gd_add(gd_unpack(x), gd_unpack(y)));      // [0] A | [1] bit-sum 0 | [2] bit-sum 1
        }
        __syncthreads();
        if (4 * E4 <= blockDim.x) {                                 // S2 = A + bit-sum 1, four lanes per entry
            const u32 g = threadIdx.x >> 2, q = threadIdx.x & 3;
            if (g < E4) {
                gq a, c; a.v = fd_unpack(reinterpret_cast<const fe *>(&buf[g])[q]); c.v = fd_unpack(reinterpret_cast<const fe *>(&buf[(size_t)2 * E4 + g])[q]);
                reinterpret_cast<fe *>(&buf[g])[q] = fd_pack(gq_add(a, c, q).v);      // (entry g is read and written by its own quad only)
            }
        } else {
            for (u32 g = threadIdx.x; g < E4; g += blockDim.x) buf[g] = gd_pack(gd_add(gd_unpack(buf[g]), gd_unpack(buf[(size_t)2 * E4 + g])));
        }
        __syncthreads();
    }
    u32 nb = 2, Ecur = E4;                                         // entries of the concatenated array; Ecur / G per tree
    while (Ecur > G) { msm_binary_level_inplace(Ecur, nb, buf); Ecur /= 2; nb += 1; }
    // buf: S[G] | C[nb][G]  (nb == nb_final == log2 E)
    for (u32 item = threadIdx.x; item < gv * (1 + nb); item += blockDim.x) {
        const u32 g = item / (1 + nb), l = item % (1 + nb);
        if (l == 0) store_ge(&fin_s[g], buf[g]);
        else if (l - 1 < nb_final) store_ge(&fin_c[(size_t)g * nb_final + (l - 1)], buf[(size_t)G + (size_t)(l - 1) * G + g]);
    }
}
// A small MSM (the IPP tail: a few thousand terms per problem) in ONE launch instead of memset / scatter / scan / accumulate /
// overflow / reduce: block = one (problem, window) bucket array (generic window layout, c <= 10).  The window's digits are
// ranked into per-bucket lists in LDS (SMALL_CAP entries each), thread b sums bucket b's points in a uniform loop, the buckets
// go to HBM and the same block runs the bucket reduction of k_msm_reduce_fused over them.  A list overflow (scalars built
// to collide) raises *overflow and the host repeats the MSM through the general pipeline.
#define MSM_SMALL_CAP 64      /* list entries per bucket at a mean load <= 16 (n_side <= 8 B, narrow windows fill half of the buckets): P(overflow) ~ 1e-18 per bucket */
#define MSM_SMALL_CAP_MAX 80  /* upper bound of the run-time `cap` (72 at n_side <= 16 B: mean load <= 32, P(overflow) ~ 1e-12) */
#if ROFL_KG(1)
// L16: the bucket lists hold 16-bit entries (term index < 0x7fff | sign in bit 15) -- half the LDS of the lists; the grouped launch (at most
// 16 x 64 terms a side) uses it to fit four blocks per CU at every list capacity.
template <bool L16>
__device__ __forceinline__ void msm_small_body(unsigned char *smem, u32 n_side, MsmWin mw, MsmMap mm, const MsmProb *probs, ge *buckets, ge *S_fin, ge *C_fin,
                                               u32 nb_final, u32 *overflow, u32 cap, u32 G, unsigned long long *dbg, ge *wsum_out,
                                               const sc *ip_dev, u32 ip_nblk, const niels *qpts) {
    // debugging aid (ROFL_DBG_SMALL_TIMELINE): 100 MHz wall-clock stamps of the block's phases -- start, ranked, buckets summed, reduced
    auto stamp = [&](int k) { if (dbg && threadIdx.x == 0) dbg[(size_t)blockIdx.x * 4 + k] = wall_clock64(); };
    stamp(0);
    const u32 B = 1u << (mw.c - 1);
    // G = 1: block = one (problem, window).  G > 1 (c = 7, thousands of arrays): block = G consecutive windows of a problem, G * B threads;
    // "virtual bucket" vb = g * B + b is bucket b of window w0 + g.  The last block of a problem may hold fewer than G windows: its spare
    // virtual buckets stay empty and nothing of them is stored.
    const u32 Wg = (mw.W + G - 1) / G;
    const u32 p = G > 1 ? blockIdx.x / Wg : blockIdx.x / mw.W, w0 = G > 1 ? (blockIdx.x % Wg) * G : blockIdx.x % mw.W;
    const u32 gv = min(G, mw.W - w0);                                // windows this block really holds
    const u32 pw = p * mw.W + w0;                                    // bucket array of the block's first window
    const u32 VB = G * B;
    u32 side = mm.lr_nh ? (p & 1u) : 0u;
    typedef typename std::conditional<L16, unsigned short, u32>::type lst_t;
    const u32 SIGN = L16 ? 0x8000u : 0x80000000u, IDXM = SIGN - 1;
    u32 *lcnt = reinterpret_cast<u32 *>(smem);                       // [VB]
    lst_t *lst = reinterpret_cast<lst_t *>(lcnt + VB);               // [VB][cap]: term index | sign
    for (u32 b = threadIdx.x; b < VB; b += blockDim.x) lcnt[b] = 0;
    __syncthreads();
    const sc *scal = probs[p].scal;
    auto put = [&](const u32 *kw, u32 g, u32 entry_idx) {            // the digit of window w0 + g of the scalar at kw -> that window's bucket list
        u32 wpos, wwid; msm_window(mw, w0 + g, wpos, wwid);
        int d = msm_digit_mem(kw, wpos, wwid);
        u32 ad = (u32)(d < 0 ? -d : d), entry = entry_idx | (d < 0 ? SIGN : 0u);
        for (int rep = 0; rep < 2; rep++) {
            u32 a1 = rep == 0 ? (ad > B ? B : 0u) : (ad > B ? ad - B : ad);
            if (!a1) continue;
            u32 vb = g * B + a1 - 1;
            u32 pos = atomicAdd(&lcnt[vb], 1u);
            if (pos < cap) lst[vb * cap + pos] = (lst_t)entry;
            else *(volatile u32 *)overflow = 1u;          // mapped host memory: a plain store (idempotent)
        }
    };
    for (u32 k = threadIdx.x; k < n_side; k += blockDim.x) {
        u32 i = mm.lr_nh ? msm_side_term(mm, side, k) : k;
        const u32 *kw = reinterpret_cast<const u32 *>(&scal[i]);
        for (u32 g = 0; g < gv; g++) put(kw, g, i);
    }
    // One more term for the L / R problems of an inner-product round: c_side * Q with Q = w B (upstream's <a_L, b_R> Q) -- c_side is the sum of
    // the partial inner products the round's k_ipp_round launch left in device memory, Q the chunk's point.  The host added this term with a
    // fixed-base multiplication per problem on every hop (6 us each; eight of them in a row where one thread finishes eight problems).
    const u32 QIDX = IDXM;
    if (ip_dev && threadIdx.x == 0) {
        sc acc = sc_zero();
        for (u32 b = 0; b < ip_nblk; b++) acc = sc_add(acc, load_sc(&ip_dev[((size_t)(p >> 1) * ip_nblk + b) * 2 + (p & 1u)]));
        sc cs = sc_from_mont(acc);
        for (u32 g = 0; g < gv; g++) put(reinterpret_cast<const u32 *>(&cs), g, QIDX);
    }
    __syncthreads();
    stamp(1);
    const niels *pts = probs[p].pts;
    const niels *qpt = qpts ? qpts + (p >> 1) : pts;
    auto fetch = [&](u32 v) { const u32 i = v & IDXM; return gload_nd(i == QIDX ? qpt : &pts[i]); };
    // Balance the bucket sums over the block's waves.  A wave runs as long as its fullest bucket (mean 4 entries, ~10 in every wave of 64
    // unsorted buckets); with the buckets sorted by load every wave gets buckets of similar depth.  blockDim == B = 512 (8 waves, two per
    // SIMD): wave w < 4 takes the w-th heaviest group of 64 and its SIMD partner w + 4 the (7 - w)-th -- every SIMD sees ~11 additions' worth
    // of issue instead of ~20.  G windows of 64 buckets (G waves, one per SIMD): each wave takes one of the G groups of 64 of the load order.
    u32 my_b = threadIdx.x;
    const bool sorted = (blockDim.x == 512 && B == 512) || G > 1;
    if (sorted) {
        __shared__ u32 chist[MSM_SMALL_CAP_MAX + 2], cbase[MSM_SMALL_CAP_MAX + 2];
        __shared__ unsigned short order[512];
        if (threadIdx.x < MSM_SMALL_CAP_MAX + 2) chist[threadIdx.x] = 0;
        __syncthreads();
        u32 mycnt = lcnt[threadIdx.x]; if (mycnt > cap) mycnt = cap;      // (blockDim == VB: one virtual bucket per thread)
        u32 rank_in = atomicAdd(&chist[mycnt], 1u);
        __syncthreads();
        if (threadIdx.x == 0) { u32 run = 0; for (int cval = (int)cap; cval >= 0; cval--) { cbase[cval] = run; run += chist[cval]; } }      // descending load
        __syncthreads();
        order[cbase[mycnt] + rank_in] = (unsigned short)threadIdx.x;
        __syncthreads();
        u32 wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
        // (G waves, one per SIMD: the blocks that share a CU rotate the assignment, otherwise wave 0 of every block -- the deepest buckets -- would
        //  land on the same SIMD)
        u32 grp = G > 1 ? (wv + blockIdx.x) % G : (wv < 4 ? wv : 11 - wv);
        my_b = order[grp * 64 + ln];
    }
    for (u32 vb = my_b; vb < VB; vb += blockDim.x) {
        u32 num = lcnt[vb]; if (num > cap) num = cap;
        gd acc = gd_identity();
        // the next point is in flight while the current one is added (the block runs at 2 waves/SIMD: registers are not the limit here,
        // the gather latency in front of every addition was)
        u32 v = num ? lst[vb * cap] : 0u;
        nd nxt = fetch(v);
        for (u32 e = 0; e < num; e++) {
            nd q = nxt; bool ng = (v & SIGN) != 0;
            if (e + 1 < num) { v = lst[vb * cap + e + 1]; nxt = fetch(v); }
            acc = gd_madd(acc, q, ng);
        }
        if (vb < gv * B) store_gd(&buckets[(size_t)pw * B + vb], acc);      // (consecutive windows of a problem are consecutive bucket arrays)
    }
    __threadfence_block();
    __syncthreads();
    stamp(2);
    if (G > 1) msm_reduce_binary_multi(B, G, gv, buckets + (size_t)pw * B, S_fin + pw, C_fin + (size_t)pw * nb_final, nb_final, smem);
    else if (B >= 16) msm_reduce_binary_body(B, buckets + (size_t)pw * B, S_fin + pw, C_fin + (size_t)pw * nb_final, smem);
    else msm_reduce_fused_body(B, 0, buckets + (size_t)pw * B, nullptr, S_fin + pw, C_fin + (size_t)pw * nb_final, smem);
    stamp(3);
    // launches whose window chains go to the host eight per SIMD stream (k_msm_wsum's job, without a launch of its own): one quad per window adds
    // up the window's bit-sums, W = S + sum_l 2^l D_l, and hands the host ONE point per (problem, window)
    if (wsum_out) {
        __threadfence_block();
        __syncthreads();
        if (threadIdx.x < 4 * gv) {
            const u32 q = threadIdx.x & 3, g = threadIdx.x >> 2;
            auto coord = [&](const ge *pt) { gq r; r.v = fd_unpack(reinterpret_cast<const fe *>(pt)[q]); return r; };
            const ge *cf = C_fin + (size_t)(pw + g) * nb_final;
            gq acc = coord(&cf[nb_final - 1]);
#pragma unroll 1
            for (int l = (int)nb_final - 2; l >= 0; l--) acc = gq_add(gq_double(acc, q), coord(&cf[l]), q);
            acc = gq_add(acc, coord(&S_fin[pw + g]), q);
            reinterpret_cast<fe *>(&wsum_out[pw + g])[q] = fd_pack(acc.v);
        }
    }
}
__global__ void __launch_bounds__(512) k_msm_small(u32 n_side, MsmWin mw, MsmMap mm, const MsmProb *probs, ge *buckets, ge *S_fin, ge *C_fin,
                                                   u32 nb_final, u32 *overflow, u32 cap, u32 G, unsigned long long *dbg, ge *wsum_out,
                                                   const sc *ip_dev, u32 ip_nblk, const niels *qpts) {
    extern __shared__ __align__(16) unsigned char smem[];
    msm_small_body<false>(smem, n_side, mw, mm, probs, buckets, S_fin, C_fin, nb_final, overflow, cap, G, dbg, wsum_out, ip_dev, ip_nblk, qpts);
}
// the same body for blocks of G windows x 64 buckets (256 threads): compiled for four waves per SIMD, so that four of these blocks share a CU
// (the 512-thread form above is compiled for two; thousands of small blocks want the occupancy, not the registers)
__global__ void __launch_bounds__(256, 4) k_msm_small_g(u32 n_side, MsmWin mw, MsmMap mm, const MsmProb *probs, ge *buckets, ge *S_fin, ge *C_fin,
                                                        u32 nb_final, u32 *overflow, u32 cap, u32 G, unsigned long long *dbg, ge *wsum_out,
                                                        const sc *ip_dev, u32 ip_nblk, const niels *qpts) {
    extern __shared__ __align__(16) unsigned char smem[];
    msm_small_body<true>(smem, n_side, mw, mm, probs, buckets, S_fin, C_fin, nb_final, overflow, cap, G, dbg, wsum_out, ip_dev, ip_nblk, qpts);
}
#endif

// Finish an MSM on the device when a launch carries many problems (n_partition = 64: 128 L/R problems per IPP round, whose
// 253-step Horner chains would otherwise queue on the host pool).  One block per problem, one QUAD of lanes per window (quad26.hpp:
// lane q holds coordinate q, a doubling is one squaring + one multiplication deep): quad w folds its window's bit-sums
// (S + sum_l 2^l D_l), shifts the result to the window position (pos_w doublings -- the chains of the W windows run side by side, so
// the launch is as deep as ONE chain) and the block adds the W terms through LDS.
#if ROFL_KG(1)
__global__ void __launch_bounds__(256) k_msm_horner(MsmWin mw, const ge *S_fin, const ge *C_fin, u32 nb, ge *out) {
    __shared__ ge sh[64];
    const u32 p = blockIdx.x, w = threadIdx.x >> 2, q = threadIdx.x & 3;
    auto coord = [&](const ge *pt) { gq r; r.v = fd_unpack(reinterpret_cast<const fe *>(pt)[q]); return r; };      // this lane's coordinate of *pt
    gq acc;
    if (w < mw.W) {
        size_t pw = (size_t)p * mw.W + w;
        acc = coord(&C_fin[pw * nb + nb - 1]);
#pragma unroll 1
        for (int l = (int)nb - 2; l >= 0; l--) acc = gq_add(gq_double(acc, q), coord(&C_fin[pw * nb + l]), q);
        acc = gq_add(acc, coord(&S_fin[pw]), q);
        u32 pos, wid; msm_window(mw, w, pos, wid);
#pragma unroll 1
        for (u32 i = 0; i < pos; i++) acc = gq_double(acc, q);
        reinterpret_cast<fe *>(&sh[w])[q] = fd_pack(acc.v);
    }
    __syncthreads();
#pragma unroll 1
    for (u32 s = 32; s >= 1; s >>= 1) {
        if (w < s && w + s < mw.W) {
            acc = gq_add(acc, coord(&sh[w + s]), q);
            reinterpret_cast<fe *>(&sh[w])[q] = fd_pack(acc.v);
        }
        __syncthreads();
    }
    if (w == 0) reinterpret_cast<fe *>(&out[p])[q] = fd_pack(acc.v);
}
#endif

// The first step of k_msm_horner alone: one quad of lanes per (problem, window) adds up the window's bit-sums, W_w = S + sum_l 2^l D_l
// (c - 1 doublings deep), and writes it to `out` [PW] -- mapped host memory: the 253-step chains across the windows run on the host, eight
// problems per AVX-512 IFMA instruction stream (host51x8.hpp).
#if ROFL_KG(1)
__global__ void __launch_bounds__(256) k_msm_wsum(u32 PW, const ge *S_fin, const ge *C_fin, u32 nb, ge *out) {
    const u32 pw = (blockIdx.x * blockDim.x + threadIdx.x) >> 2, q = threadIdx.x & 3;
    if (pw >= PW) return;                               // whole quads
    auto coord = [&](const ge *pt) { gq r; r.v = fd_unpack(reinterpret_cast<const fe *>(pt)[q]); return r; };
    gq acc = coord(&C_fin[(size_t)pw * nb + nb - 1]);
#pragma unroll 1
    for (int l = (int)nb - 2; l >= 0; l--) acc = gq_add(gq_double(acc, q), coord(&C_fin[(size_t)pw * nb + l]), q);
    acc = gq_add(acc, coord(&S_fin[pw]), q);
    reinterpret_cast<fe *>(&out[pw])[q] = fd_pack(acc.v);
}
#endif

// self-test of quad26.hpp: pair i = (P, Q) -> 2^doublings P + Q, once with one thread per pair (gd_*), once with one quad per pair (gq_*)
#if ROFL_KG(4)
__global__ void __launch_bounds__(TPB) k_dbg_quad(u32 pairs, u32 doublings, const uint8_t *in /* [pairs][2][32] */, uint8_t *out_serial, uint8_t *out_quad, u32 *status) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 i = t >> 2, q = t & 3;
    if (i >= pairs) return;
    __align__(16) uint8_t b[64];
    for (int k = 0; k < 64; k++) b[k] = in[(size_t)i * 64 + k];
    gd P, Q;
    if (!gd_ristretto_decode(P, b) || !gd_ristretto_decode(Q, b + 32)) { atomicOr(status, 4u); return; }
    gq a = gq_from_gd(P, q), c = gq_from_gd(Q, q);
    for (u32 k = 0; k < doublings; k++) a = gq_double(a, q);
    a = gq_add(a, c, q);
    gd R = gq_to_gd(a);
    if (q == 0) {
        gd_ristretto_encode(out_quad + (size_t)i * 32, R);
        gd S = P;
        for (u32 k = 0; k < doublings; k++) S = gd_double(S);
        S = gd_add(S, Q);
        gd_ristretto_encode(out_serial + (size_t)i * 32, S);
    }
}
#endif

// ================================================================ K8/K9: verification
// decode compressed points into affine niels (+ validity); optional re-encode of the point shifted by `shift`
// (verify_rangeproof: range_proof_vec/mod.rs:155-167).  out_niels is the UNSHIFTED point.
#if ROFL_KG(4)
__global__ void __launch_bounds__(TPB) k_decode(u32 count, u32 valid_count, const uint8_t *in, const niels *shift, niels *out_niels,
                         uint8_t *out_enc, u32 *status) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    {   // blockIdx.y = vector (a batch of clients): `count` entries each, one status word each
        const size_t y = blockIdx.y;
        in += y * count * 32; status += y;
        if (out_niels) out_niels += y * count;
        if (out_enc) out_enc += y * count * 32;
    }
    if (i >= valid_count) {   // padding: identity
        if (out_niels) store_niels(&out_niels[i], niels_identity());
        if (out_enc) { uint4 z = make_uint4(0, 0, 0, 0); reinterpret_cast<uint4 *>(out_enc + (size_t)i * 32)[0] = z; reinterpret_cast<uint4 *>(out_enc + (size_t)i * 32)[1] = z; }
        return;
    }
    __align__(16) uint8_t b[32];
    const uint4 *s = reinterpret_cast<const uint4 *>(in + (size_t)i * 32);
    reinterpret_cast<uint4 *>(b)[0] = s[0]; reinterpret_cast<uint4 *>(b)[1] = s[1];
    gd p;
    if (!gd_ristretto_decode(p, b)) { atomicOr(status, 4u); p = gd_identity(); }
    // the niels form is that of the DECODED point (Z = 1: no inversion chain); a caller that wants sum_j s_j (P_j + shift) from an MSM
    // over these points adds (sum_j s_j) * shift itself (verify_chunks does: the shift is a multiple of B)
    if (out_niels) { niels r; r.ypx = fd_pack(fd_add(p.Y, p.X)); r.ymx = fd_pack(fd_sub(p.Y, p.X)); r.t2d = fd_pack(fd_mul(p.T, fd_d2())); store_niels(&out_niels[i], r); }
    if (shift) p = gd_madd(p, load_nd(shift), false);
    if (out_enc) gd_ristretto_encode(out_enc + (size_t)i * 32, p);
}
#endif
// out = a + b (compressed in/out): pedersen_ops.rs:56-59 add_rp_vec
#if ROFL_KG(4)
__global__ void __launch_bounds__(TPB) k_add_points(u32 count, const uint8_t *a, const uint8_t *b, uint8_t *out, u32 *status) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    gd p, q;
    if (!gd_ristretto_decode(p, a + (size_t)i * 32)) { atomicOr(status, 4u); p = gd_identity(); }
    if (!gd_ristretto_decode(q, b + (size_t)i * 32)) { atomicOr(status, 4u); q = gd_identity(); }
    gd_ristretto_encode(out + (size_t)i * 32, gd_add(p, q));
}
#endif
// bulletproofs verify_multiple: g_k = -z - a s_k ; h_k = z + y^-k (zz z^j 2^i - b s_k^-1)  -> canonical [g | h]
// One array of 2N scalars per group of consecutive proofs (grp[g] = first proof, proof count): sum_c rho_c * (g_c | h_c) -- the proofs of
// a group share the generators, so their G/H terms collapse into one MSM (rho_c is folded into rz, ra, rb, rzz by the host).
#if ROFL_KG(4)
__global__ void __launch_bounds__(TPB) k_verify_scalars(u32 n, u32 m, u32 lgN, const uint2 *grp, const ChunkParams *cp, const PowTabs *pt, const sc *two_pow, sc *out) {
    u32 gidx = blockIdx.y;
    const u32 first = grp[gidx].x, group = grp[gidx].y;
    size_t N = (size_t)n * m;
    u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    u32 j = k / n, i = k % n;
    sc gacc = sc_zero(), hacc = sc_zero();
    for (u32 cc = 0; cc < group; cc++) {
        const ChunkParams &P = cp[first + cc];
        const PowTabs &T = pt[first + cc];
        // s = prod_q (bit_{lgN-1-q}(k) ? u_q : u_q^-1) (challenge q in creation order), 1/s = the same product at ~k
        u32 kc = ~k;
        sc s = sc_montmul(sc_montmul(load_sc(&T.s[0][k & (PT_E - 1)]), load_sc(&T.s[1][(k >> PT_W) & (PT_E - 1)])), load_sc(&T.s[2][(k >> (2 * PT_W)) & (PT_E - 1)]));
        sc sinv = sc_montmul(sc_montmul(load_sc(&T.s[0][kc & (PT_E - 1)]), load_sc(&T.s[1][(kc >> PT_W) & (PT_E - 1)])), load_sc(&T.s[2][(kc >> (2 * PT_W)) & (PT_E - 1)]));
        for (u32 q = 0; q + 3 * PT_W < lgN; q++) {       // index bits beyond the tables (N > 2^21)
            bool bit = (k >> (lgN - 1 - q)) & 1;
            s = sc_montmul(s, bit ? P.u[q] : P.uinv[q]);
            sinv = sc_montmul(sinv, bit ? P.uinv[q] : P.u[q]);
        }
        sc g = sc_neg(sc_add(P.rz, sc_montmul(P.ra, s)));
        sc zj2 = sc_montmul(sc_montmul(P.rzz, pt_pow(T.z, P.zpow2, j)), two_pow[i]);
        sc h = sc_add(P.rz, sc_montmul(pt_pow(T.yinv, P.yinvpow2, k), sc_sub(zj2, sc_montmul(P.rb, sinv))));
        gacc = sc_add(gacc, g); hacc = sc_add(hacc, h);
    }
    sc *o = out + (size_t)gidx * 2 * N;
    store_sc(&o[k], sc_from_mont(gacc));
    store_sc(&o[N + k], sc_from_mont(hacc));
}
#endif
// The same sums with the index structure taken out of the inner loop (batches of clients: 192 proofs per group at cfg 4 made the loop above
// the verifier's largest kernel, ~13 scalar multiplications per (proof, index) at the VALU's multiply rate).  Everything that depends on
// k only through its high digits is shared by a block:
//   s_c(k)                  = [s1(d1) s2'(d2)] * s0(lo)                       k = 128 d + lo, d = 128 d2 + d1;  s2' carries ra
//   -rb s_c^-1(k) y^-k      = [q1(d1) q2'(d2)] * q0(lo)                       q_l(e) = s_l(~e) yinv^(e 128^l);  q2' carries -rb
//   rzz z^j 2^i y^-k        = [w0 w1 w2'](j) * zi(i),  k = n j + i            w = yinv^n z, zi(i) = (2 yinv)^i;  w2' carries rzz
// A block of 256 threads covers 512 consecutive k (four values of d, 512 / n values of j); per tile of VS_TP proofs it first computes the
// bracketed factors into LDS, then every thread does three multiplications per (proof, index) for its two indices, which share lo and i.
struct VTabs { sc s0[PT_E], s1[PT_E], s2[PT_E], q0[PT_E], q1[PT_E], q2[PT_E], w0[PT_E], w1[PT_E], w2[PT_E], zi[64]; };     // Montgomery form
#define VS_TP 8
#if ROFL_KG(4)
__global__ void __launch_bounds__(256) k_vtabs(const ChunkParams *cp, const PowTabs *pt, const sc *two_pow, VTabs *vt, u32 n, u32 lgn) {
    const u32 c = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 9 * PT_E + 64) return;
    const ChunkParams &P = cp[c]; const PowTabs &T = pt[c]; VTabs &V = vt[c];
    if (t >= 9 * PT_E) { u32 i = t - 9 * PT_E; store_sc(&V.zi[i], i < n ? sc_montmul(load_sc(&two_pow[i]), load_sc(&T.yinv[0][i])) : sc_zero()); return; }
    const u32 tab = t / PT_E, e = t % PT_E, lvl = tab % 3;
    sc r;
    if (tab < 3) { r = load_sc(&T.s[lvl][e]); if (lvl == 2) r = sc_montmul(r, load_sc(&P.ra)); }
    else if (tab < 6) {
        r = sc_montmul(load_sc(&T.s[lvl][~e & (PT_E - 1)]), load_sc(&T.yinv[lvl][e]));
        if (lvl == 2) r = sc_montmul(r, sc_neg(load_sc(&P.rb)));
    } else {      // w^(e 128^lvl), w = yinv^n z
        r = sc_montmul(pt_pow(T.yinv, P.yinvpow2, e << (lgn + PT_W * lvl)), load_sc(&T.z[lvl][e]));
        if (lvl == 2) r = sc_montmul(r, load_sc(&P.rzz));
    }
    sc *dst = tab == 0 ? V.s0 : tab == 1 ? V.s1 : tab == 2 ? V.s2 : tab == 3 ? V.q0 : tab == 4 ? V.q1 : tab == 5 ? V.q2 : tab == 6 ? V.w0 : tab == 7 ? V.w1 : V.w2;
    store_sc(&dst[e], r);
}
// blockIdx.z = slice of the group's proofs (few blocks otherwise: one client with n_partition = 64 has N / 512 = 32): slice z takes `per`
// consecutive proofs and writes its partial sums to out + (group * gridDim.z + z) * 2N; k_vs_sum adds the slices up.
__global__ void __launch_bounds__(256) k_verify_scalars2(u32 n, u32 lgn, u32 m, const uint2 *grp, const ChunkParams *cp, const VTabs *vt, sc *out, u32 per) {
    __shared__ sc sh[VS_TP][8 + 64];
    u32 first = grp[blockIdx.y].x, count = grp[blockIdx.y].y;
    { const u32 lo = min(count, blockIdx.z * per), hi = min(count, lo + per); first += lo; count = hi - lo; }
    const size_t N = (size_t)n * m;
    const u32 base = blockIdx.x * 512, t = threadIdx.x, lo = t & 127, half = t >> 7;
    const u32 nj = 512 >> lgn, i = lo & (n - 1), j0 = base >> lgn;
    u32 kk[2], js[2];
#pragma unroll
    for (int r = 0; r < 2; r++) { kk[r] = base + (half + 2 * r) * 128 + lo; js[r] = (kk[r] >> lgn) - j0; }
    sc gacc[2] = {sc_zero(), sc_zero()}, hacc[2] = {sc_zero(), sc_zero()}, rzsum = sc_zero();
    // the products of a tile of VS_TP proofs are summed unreduced (sc_mac_wide) and reduced once: 2 x VS_TP = 16 products per accumulator at most
    static_assert(VS_TP <= 8, "sc_redc_wide takes the sum of at most sixteen products");
    for (u32 c0 = 0; c0 < count; c0 += VS_TP) {
        const u32 tp = min((u32)VS_TP, count - c0);
        for (u32 task = t; task < tp * (8 + nj); task += 256) {
            const u32 pi = task / (8 + nj), v = task % (8 + nj);
            const VTabs &V = vt[first + c0 + pi];
            sc r;
            if (v < 8) { const u32 d = (base >> 7) + (v & 3); r = v < 4 ? sc_montmul(load_sc(&V.s1[d & 127]), load_sc(&V.s2[d >> 7])) : sc_montmul(load_sc(&V.q1[d & 127]), load_sc(&V.q2[d >> 7])); }
            else { const u32 j = j0 + v - 8; r = sc_montmul(sc_montmul(load_sc(&V.w0[j & 127]), load_sc(&V.w1[(j >> 7) & 127])), load_sc(&V.w2[j >> 14])); }
            sh[pi][v] = r;
        }
        __syncthreads();
        u32 gw[2][17], hw[2][17];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int q = 0; q < 17; q++) { gw[r][q] = 0; hw[r][q] = 0; }
        for (u32 pi = 0; pi < tp; pi++) {
            const VTabs &V = vt[first + c0 + pi];
            const sc s0 = load_sc(&V.s0[lo]), q0 = load_sc(&V.q0[lo]), zi = load_sc(&V.zi[i]);
#pragma unroll
            for (int r = 0; r < 2; r++) {
                sc_mac_wide(gw[r], sh[pi][half + 2 * r], s0);
                sc_mac_wide(hw[r], sh[pi][4 + half + 2 * r], q0);
                sc_mac_wide(hw[r], sh[pi][8 + js[r]], zi);
            }
            rzsum = sc_add(rzsum, load_sc(&cp[first + c0 + pi].rz));
        }
#pragma unroll
        for (int r = 0; r < 2; r++) { gacc[r] = sc_add(gacc[r], sc_redc_wide(gw[r])); hacc[r] = sc_add(hacc[r], sc_redc_wide(hw[r])); }
        __syncthreads();
    }
    sc *o = out + ((size_t)blockIdx.y * gridDim.z + blockIdx.z) * 2 * N;
#pragma unroll
    for (int r = 0; r < 2; r++) {
        store_sc(&o[kk[r]], sc_from_mont(sc_neg(sc_add(rzsum, gacc[r]))));
        store_sc(&o[N + kk[r]], sc_from_mont(sc_add(rzsum, hacc[r])));
    }
}
// out[g][k] = sum_z part[g][z][k] (canonical scalars), k < len
__global__ void __launch_bounds__(TPB) k_vs_sum(u32 len, u32 nsl, const sc *part, sc *out) {
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x, g = blockIdx.y;
    if (k >= len) return;
    sc acc = load_sc(&part[(size_t)g * nsl * len + k]);
    for (u32 z = 1; z < nsl; z++) acc = sc_add(acc, load_sc(&part[((size_t)g * nsl + z) * len + k]));
    store_sc(&out[(size_t)g * len + k], acc);
}
#endif
#if ROFL_KG(4)
__global__ void __launch_bounds__(TPB) k_vscalars(u32 m, const ChunkParams *cp, const PowTabs *pt, sc *out, size_t stride) {
    u32 c = blockIdx.y;
    u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    store_sc(&out[c * stride + j], sc_from_mont(sc_montmul(cp[c].c_zz, pt_pow(pt[c].z, cp[c].zpow2, j))));
}
#endif

// ================================================================ K11: per-element Sigma-proofs
// rand_proof (ElGamal pair + proof of knowledge of (m, r)) and square_rand_proof (adds c_sq = m^2 B + r2 Bb and the
// proof that it commits to the square of the value in c.L); one thread per element, Merlin transcript on the device
// (3-4 Keccak-f per element).  Reference: rand_proof/{mod,party,dealer,transcript}.rs, square_rand_proof/{mod,party,dealer}.rs.
struct DMerlin { u64 st[25]; u32 pos, pos_begin; };
__device__ __forceinline__ void dm_xor(DMerlin &t, u32 p, u32 b) { t.st[p >> 3] ^= (u64)(b & 0xffu) << (8 * (p & 7)); }
__device__ inline void dm_run_f(DMerlin &t) {
    dm_xor(t, t.pos, t.pos_begin); dm_xor(t, t.pos + 1, 0x04); dm_xor(t, 167, 0x80);
    keccak_f1600(t.st); t.pos = 0; t.pos_begin = 0;
}
__device__ inline void dm_absorb(DMerlin &t, const uint8_t *d, u32 n) {
    for (u32 i = 0; i < n; i++) { dm_xor(t, t.pos, d[i]); if (++t.pos == 166) dm_run_f(t); }
}
__device__ inline void dm_begin_op(DMerlin &t, u32 flags) {
    uint8_t hdr[2] = {(uint8_t)t.pos_begin, (uint8_t)flags};
    t.pos_begin = t.pos + 1;
    dm_absorb(t, hdr, 2);
    if ((flags & (4 | 32)) && t.pos != 0) dm_run_f(t);
}
__device__ inline void dm_append(DMerlin &t, const char *label, u32 ll, const uint8_t *msg, u32 len) {
    uint8_t le[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
    dm_begin_op(t, 16 | 2); dm_absorb(t, (const uint8_t *)label, ll);
    dm_absorb(t, le, 4);                               // meta-AD continued ("more")
    dm_begin_op(t, 2); dm_absorb(t, msg, len);
}
__device__ inline sc dm_challenge_scalar(DMerlin &t, const char *label, u32 ll) {
    uint8_t le[4] = {64, 0, 0, 0};
    dm_begin_op(t, 16 | 2); dm_absorb(t, (const uint8_t *)label, ll);
    dm_absorb(t, le, 4);
    dm_begin_op(t, 1 | 2 | 4);
    u32 w[16];
    for (u32 i = 0; i < 64; i++) {
        u32 p = t.pos; u32 sh = 8 * (p & 7);
        u32 b = (u32)(t.st[p >> 3] >> sh) & 0xffu;
        t.st[p >> 3] &= ~(0xffULL << sh);
        if ((i & 3) == 0) w[i >> 2] = 0;
        w[i >> 2] |= b << (8 * (i & 3));
        if (++t.pos == 166) dm_run_f(t);
    }
    sc lo, hi;
#pragma unroll
    for (int i = 0; i < 8; i++) { lo.v[i] = w[i]; hi.v[i] = w[8 + i]; }
    return sc_from_wide(lo, hi);
}
__device__ inline gd sg_fixed_mul(const niels *tab, const sc &k) {     // k canonical
    gd acc = gd_identity(); int carry = 0;
    for (int i = 0; i < 64; i++) {
        int v = (int)((k.v[i >> 3] >> ((i & 7) * 4)) & 15) + carry;
        carry = (v + 8) >> 4;
        int dgt = v - (carry << 4), ad = dgt < 0 ? -dgt : dgt;
        if (ad) acc = gd_madd(acc, load_nd(&tab[i * 8 + ad - 1]), dgt < 0);
    }
    return acc;
}
// the same with radix-256 signed digits: tab8[w][e] = (e+1) * 256^w * P, w < 32, e < 128 (393 KB per base, built once by k_fixed_tab8: gathers
// from L2) -- 32 mixed additions instead of 64.  k canonical (< 2^253: the top digit takes the carry without overflowing)
__device__ inline gd sg_fixed_mul8(const niels *tab8, const sc &k) {
    gd acc = gd_identity(); int carry = 0;
    for (int i = 0; i < 32; i++) {
        int v = (int)((k.v[i >> 2] >> ((i & 3) * 8)) & 255) + carry;
        carry = (v + 128) >> 8;
        int dgt = v - (carry << 8), ad = dgt < 0 ? -dgt : dgt;
        if (ad) acc = gd_madd(acc, load_nd(&tab8[i * 128 + ad - 1]), dgt < 0);
    }
    return acc;
}
#if ROFL_KG(2)
__global__ void __launch_bounds__(64) k_fixed_tab8(const niels *tab4, niels *tab8) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x, w = t >> 7, e = t & 127;
    if (w >= 32) return;
    if (w == 31 && e >= 32) { store_niels(&tab8[t], load_niels(&tab4[0])); return; }      // (never read: the top digit of a canonical scalar is <= 32)
    sc k = sc_zero(); k.v[w >> 2] = (e + 1) << ((w & 3) * 8);
    store_niels(&tab8[t], gd_to_niels(sg_fixed_mul(tab4, k)));
}
#endif
__device__ inline gd gd_neg(const gd &p) { gd r; r.X = fd_neg(p.X); r.Y = p.Y; r.Z = p.Z; r.T = fd_neg(p.T); return r; }
// k * P, signed radix-16 windows, k canonical (< 2^253)
__device__ inline gd sg_var_mul(const sc &k, const gd &P) {
    ge tab[8];                                          // packed 1P..8P (local memory)
    gd cur = P; tab[0] = gd_pack(cur);
    for (int e = 1; e < 8; e++) { cur = gd_add(cur, P); tab[e] = gd_pack(cur); }
    int8_t dg[65]; int carry = 0;
    for (int i = 0; i < 64; i++) { int v = (int)((k.v[i >> 3] >> ((i & 7) * 4)) & 15) + carry; carry = (v + 8) >> 4; dg[i] = (int8_t)(v - (carry << 4)); }
    dg[64] = (int8_t)carry;
    gd acc = gd_identity();
    for (int i = 64; i >= 0; i--) {
        if (i != 64) { acc = gd_double(acc); acc = gd_double(acc); acc = gd_double(acc); acc = gd_double(acc); }
        int d = dg[i], ad = d < 0 ? -d : d;
        if (ad) { gd q = gd_unpack(tab[ad - 1]); if (d < 0) q = gd_neg(q); acc = gd_add(acc, q); }
    }
    return acc;
}
__device__ inline void sg_encode(uint8_t *out, const gd &p) { gd_ristretto_encode(out, p); }
__device__ inline bool sg_decode(gd &p, const uint8_t *in) { bool ok = gd_ristretto_decode(p, in); if (!ok) p = gd_identity(); return ok; }
// niels form of a freshly decoded point (gd_ristretto_decode leaves Z = 1: no inversion)
__device__ inline niels sg_affine_niels(const gd &p) { niels r; r.ypx = fd_pack(fd_add(p.Y, p.X)); r.ymx = fd_pack(fd_sub(p.Y, p.X)); r.t2d = fd_pack(fd_mul(p.T, fd_d2())); return r; }
__device__ inline bool sg_is_identity(const gd &p) { ge t = gd_pack(p); return ge_is_identity_ristretto(t); }

// kind 0 RandProof (L|R ; L'|R'|Zm|Zr), 1 SquareRandProof (L|R|c_sq ; L'|R'|c_sq'|Zm|Zr1|Zr2), 2 SquareProof (c_l|c_sq ; c_l'|c_sq'|Zm|Zr1|Zr2)
__device__ inline void sg_transcript(int kind, DMerlin &t, const uint8_t *cm, const uint8_t *pf, bool has_R) {
    if (kind == 0) { dm_append(t, "C", 1, cm, 64); dm_append(t, "C_prime", 7, pf, 64); return; }
    u32 w = has_R ? 64 : 32;
    dm_append(t, "C_eg", 4, cm, w); dm_append(t, "C_ped", 5, cm + w, 32); dm_append(t, "C_prime_eg", 10, pf, w); dm_append(t, "C_prime_ped", 11, pf + w, 32);
}
__device__ inline sc sg_f32_to_sc(float v, u32 fp_bits, u32 fp_frac) {     // conversion32.rs:11-18
    double x = fabs((double)v) * (double)(1ULL << fp_frac), lim = ldexp(1.0, (int)fp_bits);
    u64 maxbits = fp_bits >= 64 ? ~0ULL : ((1ULL << fp_bits) - 1), kq;
    if (x >= lim) kq = maxbits; else { double r = rint(x); kq = (r >= lim) ? maxbits : (u64)r; }
    sc m = sc_from_u64(kq); if (v < 0.0f) m = sc_neg(m);
    return m;
}
// nonce idx of a Sigma-proof call (same sources as k_nonce_expand).  `cache` keeps the last XOF block: consecutive indices share one Keccak-f.
struct SgNonceCache { u64 blk; u64 w[16]; };
__device__ inline sc sg_nonce(int mode, const NonceSeed &seed, const uint8_t *stream, u64 stream_scalars, u64 idx, SgNonceCache &cache) {
    sc lo, hi;
    if (mode == 1) {
        if (cache.blk != (idx >> 1)) {
            const u64 dom[2] = ROFL_NONCE_DOM;
            u64 st[25]; shake256_seeded_block(st, dom, seed.w, idx >> 1);
#pragma unroll
            for (int q = 0; q < 16; q++) cache.w[q] = st[q];
            cache.blk = idx >> 1;
        }
        const bool h = idx & 1;
#pragma unroll
        for (int q = 0; q < 4; q++) { u64 a = h ? cache.w[8 + q] : cache.w[q], b = h ? cache.w[12 + q] : cache.w[4 + q];
                                      lo.v[2 * q] = (u32)a; lo.v[2 * q + 1] = (u32)(a >> 32); hi.v[2 * q] = (u32)b; hi.v[2 * q + 1] = (u32)(b >> 32); }
    } else if (idx < stream_scalars) {
        const u32 *s = reinterpret_cast<const u32 *>(stream + idx * 64);
#pragma unroll
        for (int q = 0; q < 8; q++) { lo.v[q] = s[q]; hi.v[q] = s[8 + q]; }
    } else { lo = sc_zero(); hi = sc_zero(); }
    return sc_from_wide(lo, hi);
}
#if ROFL_KG(3)
__global__ void __launch_bounds__(64) k_sigma_prove(int kind, u32 d, const float *vals, u32 fp_bits, u32 fp_frac, const sc *r1c, const sc *r2c,
                                                    const uint8_t *existing, int mode, NonceSeed seed, const uint8_t *stream, u64 stream_scalars, u64 nonce_base,
                                                    DMerlin init, const niels *tabB, const niels *tabBb, uint8_t *proofs, uint8_t *commits, u32 *status) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d) return;
    bool has_R = kind != 2, has_sq = kind != 0;
    u32 npts = 1 + (has_R ? 1 : 0) + (has_sq ? 1 : 0), nn = has_sq ? 3 : 2, clen = 32 * npts, plen = 32 * (npts + nn), sq_off = has_R ? 64 : 32;
    float v = vals[i];
    if (v != v) { atomicOr(status, 2u); return; }
    sc m = sg_f32_to_sc(v, fp_bits, fp_frac);
    sc r1 = load_sc_reduced(&r1c[i]), r2 = has_sq ? load_sc_reduced(&r2c[i]) : sc_zero();
    sc nc[3];
    SgNonceCache ncache; ncache.blk = ~0ULL;
    for (u32 j = 0; j < nn; j++) nc[j] = sg_nonce(mode, seed, stream, stream_scalars, nonce_base + (u64)nn * i + j, ncache);   // m', r1' (, r2')
    uint8_t *cm = commits + (size_t)clen * i, *pf = proofs + (size_t)plen * i;
    gd L;
    if (existing) { if (!sg_decode(L, existing + (size_t)32 * i)) { atomicOr(status, 4u); return; } for (int q = 0; q < 32; q++) cm[q] = existing[(size_t)32 * i + q]; }
    else { L = gd_add(sg_fixed_mul(tabB, m), sg_fixed_mul(tabBb, r1)); sg_encode(cm, L); }
    if (has_R) sg_encode(cm + 32, sg_fixed_mul(tabB, r1));
    if (has_sq) { sc msq = sc_mul_plain(m, m); sg_encode(cm + sq_off, gd_add(sg_fixed_mul(tabB, msq), sg_fixed_mul(tabBb, r2))); }
    sg_encode(pf, gd_add(sg_fixed_mul(tabB, nc[0]), sg_fixed_mul(tabBb, nc[1])));
    if (has_R) sg_encode(pf + 32, sg_fixed_mul(tabB, nc[1]));
    if (has_sq) sg_encode(pf + sq_off, gd_add(sg_var_mul(nc[0], L), sg_fixed_mul(tabBb, nc[2])));
    DMerlin t = init;
    sg_transcript(kind, t, cm, pf, has_R);
    sc c = dm_challenge_scalar(t, "c", 1);
    uint8_t *z = pf + clen;
    sc_tobytes(z, sc_add(nc[0], sc_mul_plain(m, c)));
    sc_tobytes(z + 32, sc_add(nc[1], sc_mul_plain(r1, c)));
    if (has_sq) sc_tobytes(z + 64, sc_add(nc[2], sc_mul_plain(sc_sub(r2, sc_mul_plain(m, r1)), c)));
}
#endif

// The same prover with ONE THREAD PER POINT instead of one per element (ROFL_SIGMA_SPLIT, the default): an element's 4-6 output points --
// R, c_sq, L', R', c_sq' (and L when no commitment is handed in) -- are independent of each other; each costs one or two fixed-base
// multiplications and one encoding (a 254-step square-root chain), c_sq' a variable-base multiplication on top.  One thread doing all of them
// is a chain of ~7 300 field multiplications in 256 VGPRs + 60 AGPRs + 1.8 KB of scratch at one wave per SIMD (3.3 ms for a vector of
// 55 000 alone on the chip); blockIdx.y = the slot, so a block's threads run the same formula, a vector is 4-6x as many threads of a
// fifth of the length, and the transcript + responses (scalar arithmetic, k_sigma_finish) read the encoded bytes back.  Same formulas,
// same nonce indices, same bytes.
enum { SG_L = 0, SG_LCHK = 1, SG_R = 2, SG_CSQ = 3, SG_LP = 4, SG_RP = 5, SG_CSQP = 6, SG_CSQP_F = 7, SG_LCMP = 8 };
struct SgSlots { int n; int id[6]; };
// (two kernels: the slots that are fixed-base multiplications + one encoding, and c_sq' with its variable-base multiplication and its 1 KB
//  table in scratch -- in one kernel every slot would be given the registers and the scratch of the largest)
template <bool VAR> __device__ __forceinline__ void sigma_point_elem(u32 i, int kind, int slot, u32 d, const float *vals, u32 fp_bits, u32 fp_frac, const sc *r1c, const sc *r2c,
                                                                    const uint8_t *existing, int mode, const NonceSeed &seed, const uint8_t *stream, u64 stream_scalars, u64 nonce_base,
                                                                    const niels *tabB, const niels *tabBb, uint8_t *proofs, uint8_t *commits, u32 *status, uint8_t *slow_mark) {
    bool has_R = kind != 2, has_sq = kind != 0;
    u32 npts = 1 + (has_R ? 1 : 0) + (has_sq ? 1 : 0), nn = has_sq ? 3 : 2, clen = 32 * npts, plen = 32 * (npts + nn), sq_off = has_R ? 64 : 32;
    float v = vals[i];
    if (v != v) return;                                // (k_sigma_finish reports it)
    uint8_t *cm = commits + (size_t)clen * i, *pf = proofs + (size_t)plen * i;
    SgNonceCache ncache; ncache.blk = ~0ULL;
    auto nonce = [&](u32 j) { return sg_nonce(mode, seed, stream, stream_scalars, nonce_base + (u64)nn * i + j, ncache); };      // m', r1' (, r2')
    if (VAR) {      // SG_CSQP: m' L + r2' Bb
        gd L;
        if (existing) { if (!sg_decode(L, existing + (size_t)32 * i)) { atomicOr(status, 4u); return; } }
        else { sc m = sg_f32_to_sc(v, fp_bits, fp_frac), r1 = load_sc_reduced(&r1c[i]); L = gd_add(sg_fixed_mul(tabB, m), sg_fixed_mul(tabBb, r1)); }
        sc n0 = nonce(0), n2 = nonce(2);
        sg_encode(pf + sq_off, gd_add(sg_var_mul(n0, L), sg_fixed_mul(tabBb, n2)));
        return;
    }
    if (slot == SG_LCHK) { gd L; if (!sg_decode(L, existing + (size_t)32 * i)) atomicOr(status, 4u); return; }
    // every other slot: a B + b Bb (b may be absent), encoded
    // SG_CSQP_F: c_sq' = m' L + r2' Bb WITHOUT the variable-base multiplication -- the prover knows the opening of L = m B + r1 Bb (they are its own
    // inputs), so m' L = (m' m) B + (m' r1) Bb and c_sq' = (m' m) B + (m' r1 + r2') Bb: two fixed-base multiplications (128 mixed additions) instead of
    // a decoding, ~320 point operations on a table in scratch and one fixed-base multiplication.  A commitment that is HANDED IN (prove_existing) is
    // what the reference multiplies (square_rand_proof/party.rs: c_sq' = m' * c.L + ...): SG_LCMP recomputes m B + r1 Bb, compares its encoding with
    // the bytes handed in, and marks the element when they differ -- k_sigma_point_var then redoes exactly those elements the reference's way
    // (an inconsistent commitment gives a proof that does not verify either way, but its BYTES stay the reference's).
    sc a, b; bool two = true; uint8_t *out = nullptr;
    if (slot == SG_CSQP_F) { sc m = sg_f32_to_sc(v, fp_bits, fp_frac), n0 = nonce(0); a = sc_mul_plain(n0, m); b = sc_add(sc_mul_plain(n0, load_sc_reduced(&r1c[i])), nonce(2)); out = pf + sq_off; }
    else if (slot == SG_LCMP) { a = sg_f32_to_sc(v, fp_bits, fp_frac); b = load_sc_reduced(&r1c[i]); }
    else if (slot == SG_L) { a = sg_f32_to_sc(v, fp_bits, fp_frac); b = load_sc_reduced(&r1c[i]); out = cm; }
    else if (slot == SG_R) { a = load_sc_reduced(&r1c[i]); b = sc_zero(); two = false; out = cm + 32; }
    else if (slot == SG_CSQ) { sc m = sg_f32_to_sc(v, fp_bits, fp_frac); a = sc_mul_plain(m, m); b = load_sc_reduced(&r2c[i]); out = cm + sq_off; }
    else if (slot == SG_LP) { a = nonce(0); b = nonce(1); out = pf; }
    else { a = nonce(1); b = sc_zero(); two = false; out = pf + 32; }      // SG_RP
    gd P = sg_fixed_mul8(tabB, a);                      // (k_sigma_points is handed the radix-256 tables, k_sigma_point_var the radix-16 ones)
    if (two) P = gd_add(P, sg_fixed_mul8(tabBb, b));
    if (slot == SG_LCMP) {
        uint8_t tmp[32]; sg_encode(tmp, P);
        bool same = true; for (int q = 0; q < 32; q++) same &= tmp[q] == existing[(size_t)32 * i + q];
        slow_mark[i] = same ? 0 : 1;                    // (every element's mark is written here: the buffer needs no clearing -- a NaN element, which
    } else sg_encode(out, P);                           //  returns above, is skipped by k_sigma_point_var the same way whatever its mark holds)
}
// k_sigma_points: one thread per (element, slot).  k_sigma_point_var: a SMALL grid that walks the marks (an honest caller has none set: the launch
// is a scan of d bytes that finds a place beside other calls' heavy launches at once, not d / 64 blocks queueing for workgroup slots)
template <bool VAR> __device__ __forceinline__ void sigma_point_body(int kind, int slot, u32 d, const float *vals, u32 fp_bits, u32 fp_frac, const sc *r1c, const sc *r2c,
                                                                    const uint8_t *existing, int mode, const NonceSeed &seed, const uint8_t *stream, u64 stream_scalars, u64 nonce_base,
                                                                    const niels *tabB, const niels *tabBb, uint8_t *proofs, uint8_t *commits, u32 *status, uint8_t *slow_mark) {
    if (!VAR) {
        u32 i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i < d) sigma_point_elem<false>(i, kind, slot, d, vals, fp_bits, fp_frac, r1c, r2c, existing, mode, seed, stream, stream_scalars, nonce_base, tabB, tabBb, proofs, commits, status, slow_mark);
        return;
    }
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < d; i += gridDim.x * blockDim.x)
        if (!slow_mark || slow_mark[i]) sigma_point_elem<true>(i, kind, slot, d, vals, fp_bits, fp_frac, r1c, r2c, existing, mode, seed, stream, stream_scalars, nonce_base, tabB, tabBb, proofs, commits, status, slow_mark);
}
#if ROFL_KG(2)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) k_sigma_points(int kind, SgSlots slots, u32 d, const float *vals, u32 fp_bits, u32 fp_frac, const sc *r1c, const sc *r2c,
                                                     const uint8_t *existing, int mode, NonceSeed seed, const uint8_t *stream, u64 stream_scalars, u64 nonce_base,
                                                     const niels *tabB, const niels *tabBb, uint8_t *proofs, uint8_t *commits, u32 *status, uint8_t *slow_mark) {
    sigma_point_body<false>(kind, slots.id[blockIdx.y], d, vals, fp_bits, fp_frac, r1c, r2c, existing, mode, seed, stream, stream_scalars, nonce_base, tabB, tabBb, proofs, commits, status, slow_mark);
}
#endif
#if ROFL_KG(3)
__global__ void __launch_bounds__(64) k_sigma_point_var(int kind, u32 d, const float *vals, u32 fp_bits, u32 fp_frac, const sc *r1c, const sc *r2c,
                                                        const uint8_t *existing, int mode, NonceSeed seed, const uint8_t *stream, u64 stream_scalars, u64 nonce_base,
                                                        const niels *tabB, const niels *tabBb, uint8_t *proofs, uint8_t *commits, u32 *status, uint8_t *slow_mark) {
    sigma_point_body<true>(kind, SG_CSQP, d, vals, fp_bits, fp_frac, r1c, r2c, existing, mode, seed, stream, stream_scalars, nonce_base, tabB, tabBb, proofs, commits, status, slow_mark);
}
#endif
#if ROFL_KG(4)
__global__ void __launch_bounds__(TPB) k_sigma_finish(int kind, u32 d, const float *vals, u32 fp_bits, u32 fp_frac, const sc *r1c, const sc *r2c,
                                                      const uint8_t *existing, int mode, NonceSeed seed, const uint8_t *stream, u64 stream_scalars, u64 nonce_base,
                                                      DMerlin init, uint8_t *proofs, uint8_t *commits, u32 *status) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d) return;
    bool has_R = kind != 2, has_sq = kind != 0;
    u32 npts = 1 + (has_R ? 1 : 0) + (has_sq ? 1 : 0), nn = has_sq ? 3 : 2, clen = 32 * npts, plen = 32 * (npts + nn);
    float v = vals[i];
    if (v != v) { atomicOr(status, 2u); return; }
    sc m = sg_f32_to_sc(v, fp_bits, fp_frac);
    sc r1 = load_sc_reduced(&r1c[i]), r2 = has_sq ? load_sc_reduced(&r2c[i]) : sc_zero();
    sc nc[3];
    SgNonceCache ncache; ncache.blk = ~0ULL;
    for (u32 j = 0; j < nn; j++) nc[j] = sg_nonce(mode, seed, stream, stream_scalars, nonce_base + (u64)nn * i + j, ncache);
    uint8_t *cm = commits + (size_t)clen * i, *pf = proofs + (size_t)plen * i;
    if (existing) for (int q = 0; q < 32; q++) cm[q] = existing[(size_t)32 * i + q];
    DMerlin t = init;
    sg_transcript(kind, t, cm, pf, has_R);
    sc c = dm_challenge_scalar(t, "c", 1);
    uint8_t *z = pf + clen;
    sc_tobytes(z, sc_add(nc[0], sc_mul_plain(m, c)));
    sc_tobytes(z + 32, sc_add(nc[1], sc_mul_plain(r1, c)));
    if (has_sq) sc_tobytes(z + 64, sc_add(nc[2], sc_mul_plain(sc_sub(r2, sc_mul_plain(m, r1)), c)));
}
#endif

#if ROFL_KG(3)
__global__ void __launch_bounds__(64) k_sigma_verify(int kind, u32 d, const uint8_t *proofs, const uint8_t *commits, DMerlin init,
                                                     const niels *tabB, const niels *tabBb, u32 *fail_count, u32 *status) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d) return;
    bool has_R = kind != 2, has_sq = kind != 0;
    u32 npts = 1 + (has_R ? 1 : 0) + (has_sq ? 1 : 0), nn = has_sq ? 3 : 2, clen = 32 * npts, plen = 32 * (npts + nn), sq_off = has_R ? 64 : 32;
    const uint8_t *pf = proofs + (size_t)plen * i, *cm = commits + (size_t)clen * i, *z = pf + clen;
    gd L, R, Csq, Lp, Rp, Csqp;
    bool okd = sg_decode(L, cm) & sg_decode(Lp, pf);
    if (has_R) okd = okd & sg_decode(R, cm + 32) & sg_decode(Rp, pf + 32);
    if (has_sq) okd = okd & sg_decode(Csq, cm + sq_off) & sg_decode(Csqp, pf + sq_off);
    sc zm = sc_frombytes(z), zr1 = sc_frombytes(z + 32), zr2 = has_sq ? sc_frombytes(z + 64) : sc_zero();
    if (!okd || sc_geq_l(zm.v) || sc_geq_l(zr1.v) || sc_geq_l(zr2.v)) { atomicOr(status, 4u); return; }
    DMerlin t = init;
    sg_transcript(kind, t, cm, pf, has_R);
    sc c = dm_challenge_scalar(t, "c", 1);
    sc cneg = sc_neg(c);
    // Z_m B + Z_r1 Bb - c C.L - C'.L == 0 ;  Z_r1 B - c C.R - C'.R == 0
    gd e1 = gd_add(gd_add(sg_fixed_mul(tabB, zm), sg_fixed_mul(tabBb, zr1)), gd_add(sg_var_mul(cneg, L), gd_neg(Lp)));
    bool ok = sg_is_identity(e1);
    if (has_R) { gd e2 = gd_add(sg_fixed_mul(tabB, zr1), gd_add(sg_var_mul(cneg, R), gd_neg(Rp))); ok = ok & sg_is_identity(e2); }
    if (has_sq) {   // Z_m C.L + Z_r2 Bb - c c_sq - c_sq' == 0
        gd e3 = gd_add(gd_add(sg_var_mul(zm, L), sg_fixed_mul(tabBb, zr2)), gd_add(sg_var_mul(cneg, Csq), gd_neg(Csqp)));
        ok = ok & sg_is_identity(e3);
    }
    if (!ok) atomicAdd(fail_count, 1u);
}
#endif

// Batched verification of the per-element Sigma-proofs (round 2).  The reference checks every element on its own (one task per
// element, rand_proof_vec/mod.rs:93-118, square_rand_proof_vec/mod.rs:131-159) and returns ONE bool for the vector; each check costs
// 2-4 variable-base scalar multiplications (~325 point operations each).  Here the 2-3 group equations of all d elements are folded
// into a single random linear combination  sum_i (w1_i e1_i + w2_i e2_i + w3_i e3_i) == 0  -- one Pippenger MSM over the 4-6 d decoded
// points (~16 mixed additions per point) plus two fixed-base terms -- exactly the trick upstream's verify_multiple uses for range
// proofs.  A forged element passes with probability 2^-126 at most (127- or 126-bit weights from fresh OS randomness).  This kernel does the
// per-element part: decode + validity (FormatError), the element's Merlin transcript -> c_i, the weights, the MSM scalars, and the
// block partial sums of the B / B_blinding coefficients.
//   e1: c L + L' - Z_m B - Z_r1 Bb = 0        e2 (kinds 0, 1): c R + R' - Z_r1 B = 0        e3 (kinds 1, 2): c Csq + Csq' - Z_m L - Z_r2 Bb = 0
// point / scalar slot k of element i at index k * d + i;  slots: L, L', [R, R'], [Csq, Csq'].
#if ROFL_KG(3)
// blockIdx.y = vector (the clients of a server-side batch: `d` elements each, laid out one after the other in every array; its own
// status word; weight index widx0 + y * d + i so that no two elements of a batch share a weight).
__global__ void __launch_bounds__(TPB) k_sigma_vprep(int kind, u32 d, const uint8_t *proofs, const uint8_t *commits, DMerlin init, NonceSeed wseed, u64 widx0, u32 wbits,
                                                      sc *scal_canon, sc *fixed_part /* [gridDim.y][gridDim.x][2] Montgomery */, u32 *status /* [gridDim.y] */) {
    __shared__ sc lds[TPB * 2];
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    bool has_R = kind != 2, has_sq = kind != 0;
    u32 npts = 1 + (has_R ? 1 : 0) + (has_sq ? 1 : 0), nn = has_sq ? 3 : 2, clen = 32 * npts, plen = 32 * (npts + nn), sq_off = has_R ? 64 : 32;
    {
        const size_t y = blockIdx.y;
        proofs += y * d * plen; commits += y * d * clen; scal_canon += y * 2 * npts * d;
        fixed_part += y * gridDim.x * 2; status += y; widx0 += y * d;
    }
    sc v[2] = {sc_zero(), sc_zero()};
    if (i < d) {
        const uint8_t *pf = proofs + (size_t)plen * i, *cm = commits + (size_t)clen * i, *z = pf + clen;
        sc zm = sc_frombytes(z), zr1 = sc_frombytes(z + 32), zr2 = has_sq ? sc_frombytes(z + 64) : sc_zero();
        bool okf = !sc_geq_l(zm.v) && !sc_geq_l(zr1.v) && !sc_geq_l(zr2.v);
        sc c;
        { DMerlin t = init; sg_transcript(kind, t, cm, pf, has_R); c = dm_challenge_scalar(t, "c", 1); }      // needs the bytes only
        // weights: the low `wbits` (127 or 126) bits of SHAKE256("rofl-zk/sgbatch" || seed || 3 i + k).  Weights of that size are what batch
        // verification needs (a forged element passes with probability 2^-wbits) and the three primed points then carry short scalars: their
        // upper windows are empty, a quarter of the MSM's additions gone.  The host picks wbits so that the top bit of a weight is NOT the top
        // bit of a window of the MSM's layout: the signed-digit recoding would carry it into the next window as the digit +1 for half of all
        // terms -- one bucket with tens of thousands of entries, summed by one thread.  The combination is taken with the sign that leaves +w on the primed points:
        //   sum_i  w1 (c L + L' - Z_m B - Z_r1 Bb) + w2 (c R + R' - Z_r1 B) + w3 (c Csq + Csq' - Z_m L - Z_r2 Bb) == 0
        sc w[3];
        const u64 dom[2] = {0x2f6b7a2d6c666f72ULL, 0x686374616267732fULL};      // "rofl-zk/" "/sgbatch"
        for (int k = 0; k < 3; k++) {
            u64 st[25]; shake256_seeded_block(st, dom, wseed.w, 3ull * (widx0 + i) + k);
#pragma unroll
            for (int q = 0; q < 2; q++) { w[k].v[2 * q] = (u32)st[q]; w[k].v[2 * q + 1] = (u32)(st[q] >> 32); }
#pragma unroll
            for (int q = 4; q < 8; q++) w[k].v[q] = 0;
            w[k].v[3] &= 0xffffffffu >> (128 - wbits);
        }
        if (!has_R) w[1] = sc_zero();
        if (!has_sq) w[2] = sc_zero();
        // B: -(w1 Z_m + w2 Z_r1) ; Bb: -(w1 Z_r1 + w3 Z_r2)
        v[0] = sc_to_mont(sc_neg(sc_add(sc_mul_plain(w[0], zm), sc_mul_plain(w[1], zr1))));
        v[1] = sc_to_mont(sc_neg(sc_add(sc_mul_plain(w[0], zr1), sc_mul_plain(w[2], zr2))));
        // the scalars of the element's points (the points themselves are decoded by k_sigma_vdecode, one thread per point): slot k of element i at k * d + i
        u32 slot = 0;
        store_sc(&scal_canon[(size_t)slot * d + i], sc_sub(sc_mul_plain(w[0], c), sc_mul_plain(w[2], zm))); slot++;      // L: w1 c - w3 Z_m
        store_sc(&scal_canon[(size_t)slot * d + i], w[0]); slot++;                                                          // L': w1
        if (has_R) {
            store_sc(&scal_canon[(size_t)slot * d + i], sc_mul_plain(w[1], c)); slot++;                                     // R: w2 c
            store_sc(&scal_canon[(size_t)slot * d + i], w[1]); slot++;                                                      // R': w2
        }
        if (has_sq) {
            store_sc(&scal_canon[(size_t)slot * d + i], sc_mul_plain(w[2], c)); slot++;                                     // Csq: w3 c
            store_sc(&scal_canon[(size_t)slot * d + i], w[2]); slot++;                                                      // Csq': w3
        }
        if (!okf) atomicOr(status, 4u);            // FormatError: the host ignores the sum
    }
    block_sum_sc<2>(v, lds);
    if (threadIdx.x == 0) { store_sc(&fixed_part[blockIdx.x * 2], v[0]); store_sc(&fixed_part[blockIdx.x * 2 + 1], v[1]); }
}
#endif
// The points of the batched Sigma-proof check: ONE THREAD PER POINT (slot k of element i of vector y at pts[(y * nslots + k) * d + i]; the
// commitments' points in the even slots, the proof's primed points in the odd ones), decoded into affine niels form.  Kept apart from
// k_sigma_vprep -- whose transcript and weight sponges hold 256 VGPRs + 75 AGPRs at one wave per SIMD -- this is the plain inverse-square-root
// chain at the occupancy and rate of k_decode: the fused kernel took 34.7 ms for a round of 48 clients of d = 55 000, the two together 2x ms.
#if ROFL_KG(3)
__global__ void __launch_bounds__(TPB) k_sigma_vdecode(int kind, u32 d, const uint8_t *proofs, const uint8_t *commits, niels *pts, u32 *status /* [gridDim.y] */) {
    bool has_R = kind != 2, has_sq = kind != 0;
    u32 npts = 1 + (has_R ? 1 : 0) + (has_sq ? 1 : 0), nn = has_sq ? 3 : 2, clen = 32 * npts, plen = 32 * (npts + nn), nslots = 2 * npts;
    const size_t y = blockIdx.y;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)nslots * d) return;
    u32 slot = (u32)(t / d), i = (u32)(t % d);
    const uint8_t *src = (slot & 1) ? proofs + (y * d + i) * plen + 32 * (slot >> 1) : commits + (y * d + i) * clen + 32 * (slot >> 1);
    __align__(16) uint8_t b[32];
    reinterpret_cast<uint4 *>(b)[0] = reinterpret_cast<const uint4 *>(src)[0]; reinterpret_cast<uint4 *>(b)[1] = reinterpret_cast<const uint4 *>(src)[1];
    gd p;
    if (!gd_ristretto_decode(p, b)) { atomicOr(status + y, 4u); p = gd_identity(); }
    store_niels(&pts[(y * nslots + slot) * d + i], sg_affine_niels(p));
}
#endif
// ---- compressed_rand_proof: the d ElGamal pairs, the challenge-power dot products, the verification scalars
#if ROFL_KG(3)
__global__ void __launch_bounds__(64) k_eg_pairs(u32 d, const float *vals, u32 fp_bits, u32 fp_frac, const sc *rc, const uint8_t *existing,
                                                 const niels *tabB, const niels *tabBb, uint8_t *pairs, u32 *status) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d) return;
    float v = vals[i];
    if (v != v) { atomicOr(status, 2u); return; }
    sc m = sg_f32_to_sc(v, fp_bits, fp_frac), r = load_sc_reduced(&rc[i]);
    uint8_t *o = pairs + (size_t)64 * i;
    if (existing) { gd L; if (!sg_decode(L, existing + (size_t)32 * i)) { atomicOr(status, 4u); return; } for (int q = 0; q < 32; q++) o[q] = existing[(size_t)32 * i + q]; }
    else sg_encode(o, gd_add(sg_fixed_mul8(tabB, m), sg_fixed_mul8(tabBb, r)));      // (radix-256 tables)
    sg_encode(o + 32, sg_fixed_mul8(tabB, r));
}
#endif
// partial sums of m_i c^(i+1) and r_i c^(i+1)  -> out[blk][2] (Montgomery);  cpow2[b] = c^(2^b) (Montgomery)
struct CPow { sc sq[MAX_LG]; };
#if ROFL_KG(3)
__global__ void __launch_bounds__(TPB) k_cpow_dot(u32 d, const float *vals, u32 fp_bits, u32 fp_frac, const sc *rc, CPow cp, sc *out) {
    __shared__ sc lds[TPB * 2];
    sc v[2] = {sc_zero(), sc_zero()};
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < d; i += gridDim.x * blockDim.x) {
        sc p = sc_pow_tab(cp.sq, i + 1);
        sc m = sc_to_mont(sg_f32_to_sc(vals[i], fp_bits, fp_frac)), r = sc_to_mont(load_sc(&rc[i]));
        v[0] = sc_add(v[0], sc_montmul(m, p)); v[1] = sc_add(v[1], sc_montmul(r, p));
    }
    block_sum_sc<2>(v, lds);
    if (threadIdx.x == 0) { store_sc(&out[blockIdx.x * 2], v[0]); store_sc(&out[blockIdx.x * 2 + 1], v[1]); }
}
#endif
#if ROFL_KG(3)
__global__ void __launch_bounds__(TPB) k_cpow_scalars(u32 d, CPow cp, sc *out_canon) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d) return;
    store_sc(&out_canon[i], sc_from_mont(sc_pow_tab(cp.sq, i + 1)));
}
#endif
// de-interleave and decode d ElGamal pairs into two niels arrays
#if ROFL_KG(3)
__global__ void __launch_bounds__(TPB) k_decode_pairs(u32 d, const uint8_t *pairs, niels *Ls, niels *Rs, u32 *status) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * d) return;
    u32 i = t >> 1, which = t & 1;
    gd p;
    if (!gd_ristretto_decode(p, pairs + (size_t)64 * i + 32 * which)) { atomicOr(status, 4u); p = gd_identity(); }
    store_niels(which ? &Rs[i] : &Ls[i], gd_to_niels(p));
}
#endif

// ================================================================ BSGS discrete log (bsgs32.rs:14-73, pedersen_ops.rs:27-53)
// Baby-step table: keys[x] = compress(x B), x = 0..m, indexed by an open-addressing hash table (slot = first 8 key
// bytes, linear probing) that lives in HBM; one thread per point walks the giant steps.
__device__ __forceinline__ u64 bsgs_hash(const uint8_t *k) {
    u64 h = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) h |= (u64)k[i] << (8 * i);
    return h * 0x9E3779B97F4A7C15ULL;
}
#if ROFL_KG(3)
__global__ void __launch_bounds__(64) k_bsgs_build(u32 m, const niels *tabB, uint8_t *keys /* [(m+1)][32] */, u32 *slots, u32 slot_mask) {
    u32 x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x > m) return;
    gd p = sg_fixed_mul(tabB, sc_from_u64(x));
    uint8_t *key = keys + (size_t)32 * x;
    sg_encode(key, p);
    u32 s = (u32)(bsgs_hash(key) >> 32) & slot_mask;
    for (;;) { if (atomicCAS(&slots[s], 0u, x + 1) == 0u) break; s = (s + 1) & slot_mask; }
}
#endif
__device__ inline bool bsgs_find(const uint8_t *enc, const uint8_t *keys, const u32 *slots, u32 slot_mask, u32 &val) {
    u32 s = (u32)(bsgs_hash(enc) >> 32) & slot_mask;
    for (;;) {
        u32 e = slots[s];
        if (!e) return false;
        const uint8_t *k = keys + (size_t)32 * (e - 1);
        bool eq = true;
        for (int i = 0; i < 32; i++) eq &= (k[i] == enc[i]);
        if (eq) { val = e - 1; return true; }
        s = (s + 1) & slot_mask;
    }
}
#if ROFL_KG(3)
__global__ void __launch_bounds__(64) k_bsgs_solve(u32 d, const uint8_t *points, u32 m, u32 bsgs_bits, u64 max_it, niels neg_mG, const uint8_t *keys,
                                                   const u32 *slots, u32 slot_mask, uint8_t *out, u32 *status) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d) return;
    gd M;
    if (!sg_decode(M, points + (size_t)32 * i)) { atomicOr(status, 4u); return; }
    nd step = nd_unpack(neg_mG);
    u64 mask = bsgs_bits >= 32 ? 0xffffffffULL : ((1ULL << bsgs_bits) - 1);
    int found = 0; u64 value = 0;
    for (int neg = 0; neg < 2 && !found; neg++) {
        gd c = neg ? gd_neg(M) : M;
        uint8_t enc[32];
        if (!neg) { for (int q = 0; q < 32; q++) enc[q] = points[(size_t)32 * i + q]; } else sg_encode(enc, c);
        for (u64 it = 0; it < max_it; it++) {
            u32 pw;
            if (bsgs_find(enc, keys, slots, slot_mask, pw)) { value = (it * m + (pw & mask)) & mask; found = 1 + neg; break; }
            if (it + 1 < max_it) { c = gd_madd(c, step, false); sg_encode(enc, c); }
        }
    }
    if (!found) { atomicOr(status, 8u); return; }
    sc v = sc_from_u64(value);
    if (found == 2) v = sc_neg(v);
    sc_tobytes(out + (size_t)32 * i, v);
}
#endif

// ================================================================ micro-benchmark: field multiply rate
#if ROFL_KG(4)
__global__ void __launch_bounds__(TPB) k_bench_femul(u32 iters, const fe *in, fe *out) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    fd a = fd_unpack(in[t & 255]), b = fd_unpack(in[(t + 1) & 255]);
    for (u32 i = 0; i < iters; i++) { a = fd_mul(a, b); b = fd_sq(b); a = fd_mul(a, b); b = fd_mul(b, a); }
    out[t] = fd_pack(fd_add(a, b));
}
// the same for the 7-multiplication mixed addition k_msm_accumulate_fb runs, at its shape (4 blocks per CU = 4 waves/SIMD, 128 VGPRs):
// MODE 0 keeps one table entry in registers (ALU only), 1 fetches a cache-resident 128-byte entry per addition (32 KB table), 2 gathers
// uniformly at random from a table of `entries` (power of two) records like the window table is gathered
template <int MODE> __device__ __forceinline__ void bench_madd_body(u32 iters, u32 entries, const ndm *tbl, ge *out) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    gd acc = gd_identity();
    if (MODE == 2) {
        u32 h = t * 2654435761u + 12345u;
        for (u32 i = 0; i < iters; i++) { h = h * 1664525u + 1013904223u; acc = gd_madd(acc, gload_ndm(tbl + ((h >> 7) & (entries - 1))), (h >> 31) != 0); }
    } else if (MODE == 1) {
        for (u32 i = 0; i < iters; i++) acc = gd_madd(acc, gload_ndm(tbl + ((t * 7 + i * 13) & 255)), ((t + i) & 1) != 0);
    } else {
        nd q = gload_ndm(tbl + (t & 255));
        for (u32 i = 0; i < iters; i++) acc = gd_madd(acc, q, ((t + i) & 1) != 0);
    }
    store_gd(&out[t], acc);
}
__global__ void __launch_bounds__(TPB, 4) k_bench_madd_regs(u32 iters, u32 entries, const ndm *tbl, ge *out) { bench_madd_body<0>(iters, entries, tbl, out); }
__global__ void __launch_bounds__(TPB, 4) k_bench_madd_l1(u32 iters, u32 entries, const ndm *tbl, ge *out) { bench_madd_body<1>(iters, entries, tbl, out); }
__global__ void __launch_bounds__(TPB, 4) k_bench_madd_gather(u32 iters, u32 entries, const ndm *tbl, ge *out) { bench_madd_body<2>(iters, entries, tbl, out); }
#endif

}  // namespace rofl
