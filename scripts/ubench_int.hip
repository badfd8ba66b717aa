// Integer VALU instruction-rate probe for gfx950: cycles per wave-instruction at full occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int K> __global__ void __launch_bounds__(256) kern(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 77, a3 = a1 + 99;
    uint64_t b0 = a0, b1 = a1, b2 = a2, b3 = a3;
    uint32_t m = seed | 1;
    for (int i = 0; i < iters; i++) {
        if (K == 0) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(m), "v"(a0) : "vcc");) }
        if (K == 1) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));) }
        if (K == 2) { REP64(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));) }
        if (K == 3) { REP64(asm volatile("v_mad_u32_u24 %0, %0, %4, %0\n v_mad_u32_u24 %1, %1, %4, %1\n v_mad_u32_u24 %2, %2, %4, %2\n v_mad_u32_u24 %3, %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));) }
        if (K == 4) { REP64(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");) }
        if (K == 5) { REP64(asm volatile("v_mul_hi_u32_u24 %0, %0, %4\n v_mul_hi_u32_u24 %1, %1, %4\n v_mul_hi_u32_u24 %2, %2, %4\n v_mul_hi_u32_u24 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));) }
        if (K == 6) { REP64(asm volatile("v_add_u32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));) }
        if (K == 7) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n v_add_u32 %1, %1, %4" : "+v"(b0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(a0) : "vcc");) }
        if (K == 9) { REP64(asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(b0 | 1));) }
        if (K == 10) { REP64(asm volatile("v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshrrev_b64 %2, 1, %2\n v_lshrrev_b64 %3, 1, %3" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));) }
        if (K == 11) { REP64(asm volatile("v_lshl_add_u32 %0, %0, 4, %4\n v_lshl_add_u32 %1, %1, 1, %4\n v_lshl_add_u32 %2, %2, 4, %4\n v_lshl_add_u32 %3, %3, 1, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));) }
        if (K == 12) { REP64(asm volatile("v_and_b32 %0, %0, %4\n v_alignbit_b32 %1, %1, %4, 26\n v_and_b32 %2, %2, %4\n v_alignbit_b32 %3, %3, %4, 26" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));) }
        if (K == 8) { double d0 = a0, d1 = a1, d2 = a2, d3 = a3, dm = m; REP64(asm volatile("v_fma_f64 %0, %0, %4, %0\n v_fma_f64 %1, %1, %4, %1\n v_fma_f64 %2, %2, %4, %2\n v_fma_f64 %3, %3, %4, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dm));) a0 = (uint32_t)d0; a1 = (uint32_t)d1; a2 = (uint32_t)d2; a3 = (uint32_t)d3; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ (uint32_t)b0 ^ (uint32_t)b1 ^ (uint32_t)(b2 >> 32) ^ (uint32_t)b3;
}
template <int K> void run(const char *name, int waves_per_simd) {
    int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = 1 per SIMD per block-per-CU
    uint32_t *out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 200;
    hipLaunchKernelGGL(kern<K>, dim3(blocks), dim3(256), 0, 0, out, 10, 12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern<K>, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double inst_per_wave = (double)iters * 64 * 4;
    double waves_per_simd_total = (double)blocks * 4 / (256.0 * 4);
    double ns_per_inst = ms * 1e6 / (inst_per_wave * waves_per_simd_total);
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.1f cycles @2.4GHz)\n", name, waves_per_simd, ms, ns_per_inst, ns_per_inst * 2.4);
    hipFree(out);
}
int main() {
    for (int w : {1, 4, 8}) {
        run<0>("v_mad_u64_u32", w); run<1>("v_mul_lo_u32", w); run<2>("v_mul_hi_u32", w); run<3>("v_mad_u32_u24", w);
        run<5>("v_mul_hi_u32_u24", w); run<4>("v_add_co/addc", w); run<6>("v_add_u32/xor", w); run<7>("mad64 + 3 adds (per 4)", w); run<8>("v_fma_f64", w);
        run<9>("v_lshl_add_u64", w); run<10>("v_lshrrev_b64", w); run<11>("v_lshl_add_u32", w); run<12>("v_and/alignbit", w);
    }
    return 0;
}
