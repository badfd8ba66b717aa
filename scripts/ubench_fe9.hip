// VERDICT r5 "next 7": one field-arithmetic experiment with a kill criterion.
// GF(2^255-19) in NINE limbs of 29 / 28 bits (sizes 29,28,28 x3: limb i starts at bit ceil(85 i / 3)) against the product's ten limbs of
// 26 / 25 bits (csrc/fe26.hpp): 81 instead of 100 v_mad_u64_u32 per multiplication.  19 * 2^29 does not fit a 32-bit operand, so the
// wrap-around is applied to the HIGH COLUMNS instead of to a pre-multiplied operand: the 17 column sums are formed without any factor 19,
// then column k + 9 is folded into columns k and k + 1 through its two 32-bit halves (which are the accumulator's two registers -- no
// shift): h_k += 19 * lo32, h_{k+1} += 19 * 2^(32 - s_k) * hi32.  97 multiply-adds + 6 doublings + a 9-column carry chain.
// What the narrower slack costs: a 64-bit column holds 9 products of (limb_f * limb_g * 2) only while limb_f * limb_g < 2^59.8, so of the
// six sums / differences a mixed addition feeds into multiplications, three need a carry pass first (the 25.5-bit radix needs none).
//   build: cd scripts && hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ubench_fe9 ubench_fe9.hip
//   ubench_fe9 selftest          host only: f9_mul / g9_madd against fd_mul / gd_madd on random inputs (no GPU call)
//   ubench_fe9                   GPU: multiplications / s and mixed additions / s of both forms at 4 and 8 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include "../rofl_project_code_amd/csrc/fe26.hpp"
using namespace rofl;

struct f9 { u32 v[9]; };
struct g9 { f9 X, Y, Z, T; };
struct n9 { f9 ypx, ymx, t2d; };
#define F9_S(i) (((i) % 3) == 0 ? 29 : 28)
#define F9_P(i) ((85 * (i) + 2) / 3)
#define F9_MASK(i) ((1u << F9_S(i)) - 1u)

HD f9 f9_unpack(const fe &a) {
    f9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int p = F9_P(i), w = p >> 5, o = p & 31;
        u32 x = a.v[w] >> o;
        if (o && w + 1 < 8 && o + F9_S(i) > 32) x |= a.v[w + 1] << (32 - o);
        r.v[i] = x & F9_MASK(i);
    }
    r.v[0] += 19u * (a.v[7] >> 31);
    return r;
}
// any limbs < 2^32 -> tight (limb i < 2^s_i + 2^12)
HD f9 f9_carry(const f9 &a) {
    f9 r = a; u32 c;
#pragma unroll
    for (int i = 0; i < 8; i++) { c = r.v[i] >> F9_S(i); r.v[i] &= F9_MASK(i); r.v[i + 1] += c; }
    c = r.v[8] >> 28; r.v[8] &= F9_MASK(8); r.v[0] += 19u * c;
    c = r.v[0] >> 29; r.v[0] &= F9_MASK(0); r.v[1] += c;
    return r;
}
HD fe f9_pack(const f9 &a) {
    f9 t = f9_carry(a);
    fe r; u64 acc = 0; int have = 0, wi = 0;      // bit-serial packer (not on any hot path of the experiment)
#pragma unroll
    for (int i = 0; i < 9; i++) {
        acc += (u64)t.v[i] << have; have += F9_S(i);      // limb i may carry a few bits into the next limb's range: added, not or-ed
        while (have >= 32 && wi < 8) { r.v[wi++] = (u32)acc; acc >>= 32; have -= 32; }
    }
    if (wi < 8) r.v[wi] = (u32)acc;
    return r;
}
HD f9 f9_add(const f9 &a, const f9 &b) { f9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
    return r; }
// a + 2p - b ; b tight.  2p in this radix: limb 0 = 2^30 - 38, limb i = 2^(s_i + 1) - 2
HD f9 f9_sub(const f9 &a, const f9 &b) { f9 r;
    r.v[0] = a.v[0] + ((1u << 30) - 38u) - b.v[0];
#pragma unroll
    for (int i = 1; i < 9; i++) r.v[i] = a.v[i] + ((1u << (F9_S(i) + 1)) - 2u) - b.v[i];
    return r; }
HD f9 f9_select(const f9 &a, const f9 &b, bool pick_b) { f9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = pick_b ? b.v[i] : a.v[i];
    return r; }

// limb-wise f * g * 2 * 9 must stay below 2^64: f < 2^30.6 with g tight, or both < 2^29.9
HD f9 f9_mul(const f9 &f, const f9 &g) {
    u64 h[17];
#pragma unroll
    for (int k = 0; k < 17; k++) h[k] = 0;
    u32 f2[9], g2[9];
#pragma unroll
    for (int i = 0; i < 9; i++) { f2[i] = 2 * f.v[i]; g2[i] = 2 * g.v[i]; }      // (only i = 1 mod 3 of each is used: six doublings survive)
#pragma unroll
    for (int i = 0; i < 9; i++)
#pragma unroll
        for (int j = 0; j < 9; j++) {
            const int a = i % 3, b = j % 3;
            u32 x = f.v[i], y = g.v[j];
            if (a == 1 && (b == 1 || b == 2)) x = f2[i];
            else if (a == 2 && b == 1) y = g2[j];
            h[i + j] += (u64)x * y;
        }
#pragma unroll
    for (int k = 0; k < 8; k++) {      // column k + 9 -> columns k, k + 1
        const u32 lo = (u32)h[k + 9], hi = (u32)(h[k + 9] >> 32);
        h[k] += (u64)lo * 19u;
        h[k + 1] += (u64)hi * (19u << (32 - F9_S(k)));
    }
    u64 c;
#pragma unroll
    for (int i = 0; i < 8; i++) { c = h[i] >> F9_S(i); h[i] &= F9_MASK(i); h[i + 1] += c; }
    c = h[8] >> 28; h[8] &= F9_MASK(8); h[0] += c * 19;
    c = h[0] >> 29; h[0] &= F9_MASK(0); h[1] += c;
    f9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = (u32)h[i];
    return r;
}
// p +- q, q affine niels (tight), p tight: gd_madd of fe26.hpp with the three carry passes this radix needs
HD g9 g9_madd(const g9 &p, const n9 &q, bool neg) {
    f9 a_f = f9_select(q.ymx, q.ypx, neg), b_f = f9_select(q.ypx, q.ymx, neg);
    f9 A = f9_mul(f9_sub(p.Y, p.X), a_f);           // 2^30.6 x tight
    f9 B = f9_mul(f9_add(p.Y, p.X), b_f);
    f9 C = f9_mul(p.T, q.t2d);
    f9 D = f9_add(p.Z, p.Z);
    f9 E = f9_carry(f9_sub(B, A)), H = f9_add(B, A);
    f9 DmC = f9_carry(f9_sub(D, C)), DpC = f9_carry(f9_add(D, C));
    g9 r;
    f9 XF = f9_select(DmC, DpC, neg), XG = f9_select(DpC, DmC, neg);
    r.X = f9_mul(XF, E); r.Y = f9_mul(H, XG); r.T = f9_mul(H, E); r.Z = f9_mul(DmC, DpC);
    return r;
}

// ---------------------------------------------------------------- GPU kernels
__global__ void __launch_bounds__(256) k_mul10(u32 iters, const fe *in, fe *out) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    fd a = fd_unpack(in[t & 255]), b = fd_unpack(in[(t + 1) & 255]);
    for (u32 i = 0; i < iters; i++) { a = fd_mul(a, b); b = fd_mul(b, a); a = fd_mul(a, b); b = fd_mul(b, a); }
    out[t] = fd_pack(fd_add(a, b));
}
__global__ void __launch_bounds__(256) k_mul9(u32 iters, const fe *in, fe *out) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    f9 a = f9_unpack(in[t & 255]), b = f9_unpack(in[(t + 1) & 255]);
    for (u32 i = 0; i < iters; i++) { a = f9_mul(a, b); b = f9_mul(b, a); a = f9_mul(a, b); b = f9_mul(b, a); }
    out[t] = f9_pack(f9_add(a, b));
}
__global__ void __launch_bounds__(256, 4) k_madd10(u32 iters, const fe *in, fe *out) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    nd q; q.ypx = fd_unpack(in[t & 255]); q.ymx = fd_unpack(in[(t + 3) & 255]); q.t2d = fd_unpack(in[(t + 5) & 255]);
    gd acc = gd_identity();
    for (u32 i = 0; i < iters; i++) acc = gd_madd(acc, q, ((t + i) & 1) != 0);
    out[t] = fd_pack(fd_add(fd_add(acc.X, acc.Y), fd_add(acc.Z, acc.T)));
}
__global__ void __launch_bounds__(256, 4) k_madd9(u32 iters, const fe *in, fe *out) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    n9 q; q.ypx = f9_unpack(in[t & 255]); q.ymx = f9_unpack(in[(t + 3) & 255]); q.t2d = f9_unpack(in[(t + 5) & 255]);
    g9 acc; acc.X = f9_unpack(in[0]); acc.Y = f9_unpack(in[1]); acc.Z = f9_unpack(in[2]); acc.T = f9_unpack(in[3]);
    for (u32 i = 0; i < iters; i++) acc = g9_madd(acc, q, ((t + i) & 1) != 0);
    out[t] = f9_pack(f9_carry(f9_add(f9_carry(f9_add(acc.X, acc.Y)), f9_carry(f9_add(acc.Z, acc.T)))));
}

static fe rnd_fe(uint64_t &s) { fe r; for (int k = 0; k < 8; k++) { s = s * 6364136223846793005ULL + 1442695040888963407ULL; r.v[k] = (u32)(s >> 32); } r.v[7] &= 0x7fffffffu; return r; }
static bool same(const fe &a, const fe &b) { uint8_t x[32], y[32]; fe_tobytes(x, a); fe_tobytes(y, b); return !memcmp(x, y, 32); }

int selftest() {
    uint64_t s = 12345; int bad = 0;
    for (int it = 0; it < 200000 && !bad; it++) {
        fe a = rnd_fe(s), b = rnd_fe(s), c = rnd_fe(s), d = rnd_fe(s);
        if (it < 4) { for (int k = 0; k < 8; k++) a.v[k] = b.v[k] = 0xffffffffu; a.v[7] = b.v[7] = 0x7fffffffu; if (it & 1) a.v[0] = 0xffffffecu; }      // p - 1 ... 2^255 - 1
        if (!same(f9_pack(f9_unpack(a)), fd_pack(fd_unpack(a)))) { printf("pack mismatch\n"); bad = 1; }
        if (!same(f9_pack(f9_mul(f9_unpack(a), f9_unpack(b))), fd_pack(fd_mul(fd_unpack(a), fd_unpack(b))))) { printf("mul mismatch at %d\n", it); bad = 1; }
        // worst-case operand bounds of the mixed addition: (Y - X + 2p) x tight
        f9 ys = f9_sub(f9_unpack(a), f9_unpack(b)); fd yd = fd_sub(fd_unpack(a), fd_unpack(b));
        if (!same(f9_pack(f9_mul(ys, f9_unpack(c))), fd_pack(fd_mul(yd, fd_unpack(c))))) { printf("sub-mul mismatch at %d\n", it); bad = 1; }
        g9 p9; p9.X = f9_unpack(a); p9.Y = f9_unpack(b); p9.Z = f9_unpack(c); p9.T = f9_unpack(d);
        gd p10; p10.X = fd_unpack(a); p10.Y = fd_unpack(b); p10.Z = fd_unpack(c); p10.T = fd_unpack(d);
        n9 q9; q9.ypx = f9_unpack(b); q9.ymx = f9_unpack(d); q9.t2d = f9_unpack(a);
        nd q10; q10.ypx = fd_unpack(b); q10.ymx = fd_unpack(d); q10.t2d = fd_unpack(a);
        for (int neg = 0; neg < 2; neg++) {
            g9 r9 = g9_madd(p9, q9, neg); gd r10 = gd_madd(p10, q10, neg);
            r9 = g9_madd(r9, q9, !neg); r10 = gd_madd(r10, q10, !neg);      // chained: outputs feed inputs
            if (!same(f9_pack(r9.X), fd_pack(r10.X)) || !same(f9_pack(r9.Y), fd_pack(r10.Y)) || !same(f9_pack(r9.Z), fd_pack(r10.Z)) || !same(f9_pack(r9.T), fd_pack(r10.T))) { printf("madd mismatch at %d\n", it); bad = 1; }
        }
    }
    printf(bad ? "selftest FAILED\n" : "selftest ok: f9_mul, g9_madd == fd_mul, gd_madd on 200000 random inputs (+ edge values)\n");
    return bad;
}

template <class K> double timeit(K kern, int blocks, u32 iters, const fe *din, fe *dout, size_t lds) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, 8u, din, dout);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, iters, din, dout);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3;
}
int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "selftest")) return selftest();
    const int blocks = 256 * 8;
    fe *din, *dout; hipMalloc(&din, sizeof(fe) * 256); hipMalloc(&dout, sizeof(fe) * blocks * 256);
    fe h[256]; uint64_t s = 99; for (int i = 0; i < 256; i++) h[i] = rnd_fe(s);
    hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
    // the two forms must agree on the device too (same inputs, same chain)
    fe *o10 = (fe *)malloc(sizeof(fe) * 1024), *o9 = (fe *)malloc(sizeof(fe) * 1024);
    hipLaunchKernelGGL(k_mul10, dim3(4), dim3(256), 0, 0, 5u, din, dout); hipMemcpy(o10, dout, sizeof(fe) * 1024, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k_mul9, dim3(4), dim3(256), 0, 0, 5u, din, dout); hipMemcpy(o9, dout, sizeof(fe) * 1024, hipMemcpyDeviceToHost);
    int diff = 0; for (int i = 0; i < 1024; i++) diff += !same(o10[i], o9[i]);
    printf("device agreement of the multiplication chains: %d of 1024 differ\n", diff);
    const u32 iters = 400;
    for (size_t lds : {(size_t)0, (size_t)40960}) {      // 0: up to 8 waves / SIMD; 40 KB per block: 4 blocks per CU = 4 waves / SIMD (the accumulate kernel's occupancy)
        if (lds) { hipFuncSetAttribute((const void *)k_mul10, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipFuncSetAttribute((const void *)k_mul9, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                   hipFuncSetAttribute((const void *)k_madd10, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipFuncSetAttribute((const void *)k_madd9, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }
        double t10 = timeit(k_mul10, blocks, iters, din, dout, lds), t9 = timeit(k_mul9, blocks, iters, din, dout, lds);
        double m10 = timeit(k_madd10, blocks, iters, din, dout, lds), m9 = timeit(k_madd9, blocks, iters, din, dout, lds);
        const double n = (double)blocks * 256 * iters;
        printf("lds/block %6zu: multiplications/s  10 limbs %.3e   9 limbs %.3e  (x%.3f)   |  mixed additions/s  10 limbs %.3e   9 limbs %.3e  (x%.3f)\n",
               lds, n * 4 / t10, n * 4 / t9, t10 / t9, n / m10, n / m9, m10 / m9);
    }
    return 0;
}
