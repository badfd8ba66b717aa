// Staging-path probe (run on the GPU box): CPU copy bandwidth from ordinary memory into ordinary / pinned host memory of several
// flavours, with 1 and T threads, ordinary and non-temporal stores, and the H2D rate out of each pinned flavour.
// hipcc --offload-arch=gfx950 -O2 -mavx2 scripts/stage_bw.hip -o /tmp/stage_bw -lpthread && /tmp/stage_bw [threads]
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void nt_copy(uint8_t *d, const uint8_t *s, size_t n) {
    for (size_t i = 0; i + 128 <= n; i += 128) {
        __m256i a = _mm256_loadu_si256((const __m256i *)(s + i)), b = _mm256_loadu_si256((const __m256i *)(s + i + 32)), c = _mm256_loadu_si256((const __m256i *)(s + i + 64)), e = _mm256_loadu_si256((const __m256i *)(s + i + 96));
        _mm256_stream_si256((__m256i *)(d + i), a); _mm256_stream_si256((__m256i *)(d + i + 32), b); _mm256_stream_si256((__m256i *)(d + i + 64), c); _mm256_stream_si256((__m256i *)(d + i + 96), e);
    }
    _mm_sfence();
}
static double run(uint8_t *dst, const uint8_t *src, size_t n, int T, bool nt) {
    double best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
        double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back([=] { size_t lo = n * t / T / 128 * 128, hi = n * (t + 1) / T / 128 * 128; if (nt) nt_copy(dst + lo, src + lo, hi - lo); else memcpy(dst + lo, src + lo, hi - lo); });
        for (auto &x : th) x.join();
        best = std::min(best, now() - t0);
    }
    return n / best / 1e9;
}
int main(int argc, char **argv) {
    int T = argc > 1 ? atoi(argv[1]) : 16;
    const size_t n = (size_t)512 << 20;
    uint8_t *src = (uint8_t *)aligned_alloc(4096, n); memset(src, 1, n);
    uint8_t *plain = (uint8_t *)aligned_alloc(4096, n); memset(plain, 2, n);
    void *dev; hipMalloc(&dev, n);
    struct { const char *name; unsigned flags; } kinds[] = {{"hipHostMallocDefault", hipHostMallocDefault}, {"hipHostMallocNonCoherent", hipHostMallocNonCoherent}, {"hipHostMallocWriteCombined", hipHostMallocWriteCombined}, {"hipHostMallocMapped", hipHostMallocMapped}};
    printf("copy into ordinary memory: 1 thread %.1f GB/s, %d threads %.1f GB/s, nt %.1f / %.1f\n", run(plain, src, n, 1, false), T, run(plain, src, n, T, false), run(plain, src, n, 1, true), run(plain, src, n, T, true));
    for (auto &k : kinds) {
        void *p = nullptr;
        if (hipHostMalloc(&p, n, k.flags) != hipSuccess) { printf("%s: allocation failed\n", k.name); continue; }
        memset(p, 3, n);
        double a = run((uint8_t *)p, src, n, 1, false), b = run((uint8_t *)p, src, n, T, false), c = run((uint8_t *)p, src, n, 1, true), e = run((uint8_t *)p, src, n, T, true);
        hipDeviceSynchronize();
        double t0 = now(); for (int r = 0; r < 4; r++) hipMemcpy(dev, p, n, hipMemcpyHostToDevice); double h2d = 4.0 * n / (now() - t0) / 1e9;
        printf("%-28s copy-in 1 thread %.1f GB/s, %d threads %.1f | nt stores %.1f / %.1f | H2D %.1f GB/s\n", k.name, a, T, b, c, e, h2d);
        hipHostFree(p);
    }
    hipHostRegister(plain, n, hipHostRegisterDefault);
    { double t0 = now(); for (int r = 0; r < 4; r++) hipMemcpy(dev, plain, n, hipMemcpyHostToDevice); printf("registered ordinary memory: H2D %.1f GB/s\n", 4.0 * n / (now() - t0) / 1e9); }
    { double t0 = now(); hipHostUnregister(plain); double t1 = now(); hipHostRegister(plain, n, hipHostRegisterDefault); printf("hipHostUnregister %.1f ms, hipHostRegister %.1f ms for 512 MB\n", (t1 - t0) * 1e3, (now() - t1) * 1e3); }
    return 0;
}
