// Hop latency probe: how long from "host enqueues a short kernel chain" to "host has the result", for
//   (a) hipStreamSynchronize, (b) a completion flag in mapped host memory written by the last kernel and polled by the host.
// Build: hipcc --offload-arch=gfx950 -O2 -o hop_probe scripts/hop_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <vector>
#include <algorithm>
#include <atomic>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_spin(unsigned long long cycles, unsigned *out) {
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 1;
}
__global__ void k_spin_flag(unsigned long long cycles, unsigned *counter, volatile unsigned *flag, unsigned seq, unsigned *res) {
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (threadIdx.x == 0) {
        res[blockIdx.x] = seq;                       // "result" in mapped host memory
        __threadfence_system();
        unsigned done = atomicAdd(counter, 1u);
        if (done == gridDim.x - 1) { *counter = 0; __threadfence_system(); *flag = seq; }
    }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned *d_out, *d_counter; CK(hipMalloc(&d_out, 4096)); CK(hipMalloc(&d_counter, 4)); CK(hipMemset(d_counter, 0, 4));
    unsigned *h_flag, *h_res; CK(hipHostMalloc(&h_flag, 64, hipHostMallocMapped)); CK(hipHostMalloc(&h_res, 4096, hipHostMallocMapped));
    unsigned *dflag, *dres; CK(hipHostGetDevicePointer((void **)&dflag, h_flag, 0)); CK(hipHostGetDevicePointer((void **)&dres, h_res, 0));
    *h_flag = 0;
    const int reps = 300;
    for (unsigned long long kus : {20ull, 100ull}) {
        unsigned long long cyc = kus * 100;      // wall_clock64: 100 MHz
        for (int nk : {1, 2}) {
            std::vector<double> a, b;
            for (int r = 0; r < reps; r++) {
                double t0 = now_us();
                for (int k = 0; k < nk; k++) hipLaunchKernelGGL(k_spin, dim3(208), dim3(256), 0, st, cyc, d_out);
                CK(hipStreamSynchronize(st));
                a.push_back(now_us() - t0 - (double)kus * nk);
                // idle a little like the host part of a round
                double w = now_us(); while (now_us() - w < 80) { }
            }
            unsigned seq = 0;
            for (int r = 0; r < reps; r++) {
                seq++;
                double t0 = now_us();
                for (int k = 0; k < nk - 1; k++) hipLaunchKernelGGL(k_spin, dim3(208), dim3(256), 0, st, cyc, d_out);
                hipLaunchKernelGGL(k_spin_flag, dim3(208), dim3(256), 0, st, cyc, d_counter, dflag, seq, dres);
                while (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != seq) { }
                b.push_back(now_us() - t0 - (double)kus * nk);
                double w = now_us(); while (now_us() - w < 80) { }
            }
            CK(hipStreamSynchronize(st));
            std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
            printf("kernel %llu us x %d: overhead beyond kernel time  sync: median %.1f us (p10 %.1f, p90 %.1f)   flag: median %.1f us (p10 %.1f, p90 %.1f)\n",
                   kus, nk, a[reps / 2], a[reps / 10], a[reps * 9 / 10], b[reps / 2], b[reps / 10], b[reps * 9 / 10]);
        }
    }
    return 0;
}
