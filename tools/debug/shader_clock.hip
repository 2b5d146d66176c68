// What shader clock does a short kernel run at when the GPU is otherwise idle (one launch per ~50 us, as F110Env's loop does)?
// clock64() counts shader cycles, wall_clock64() a constant 100 MHz: their ratio over a ~30 us dependent chain, per launch.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/sclk tools/debug/shader_clock.hip && /tmp/sclk
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <immintrin.h>
__global__ void __launch_bounds__(64) k_chain(unsigned long long *out, unsigned long long *word, unsigned long long seq, int n, double x0)
{
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    double x = x0;
    for (int i = 0; i < n; ++i) x = __builtin_fma(x, 1.0000001, 1e-9);   // a dependent chain of n f64 FMAs
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[0] = w1 - w0; out[1] = c1 - c0; out[2] = (unsigned long long)(x > 1e300);
        __threadfence_system();
        __hip_atomic_store(word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// s_sleep 127 = 127 * 64 shader cycles asleep: its wall-clock duration gives the REAL shader clock whatever clock64() counts
__global__ void __launch_bounds__(64) k_sleep(unsigned long long *out, unsigned long long *word, unsigned long long seq, int reps)
{
    const unsigned long long w0 = wall_clock64();
    for (int i = 0; i < reps; ++i) __builtin_amdgcn_s_sleep(127);
    const unsigned long long w1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[0] = w1 - w0;
        __threadfence_system();
        __hip_atomic_store(word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// background load: `blocks` workgroups of 256 lanes spinning on FMAs until *stop != 0
__global__ void __launch_bounds__(256) k_spin(volatile unsigned int *stop, double *sink)
{
    double x = threadIdx.x;
    while (!*stop) {
        for (int i = 0; i < 256; ++i) x = __builtin_fma(x, 1.0000001, 1e-9);
    }
    if (x == 12345.678) sink[0] = x;
}
int main()
{
    hipStream_t st; hipStreamCreate(&st);
    unsigned long long *word, *dword, *out, *dout;
    hipHostMalloc((void **)&word, 64, hipHostMallocDefault); hipHostGetDevicePointer((void **)&dword, word, 0); *word = 0;
    hipHostMalloc((void **)&out, 64, hipHostMallocDefault); hipHostGetDevicePointer((void **)&dout, out, 0);
    unsigned long long seq = 0;
    const int n = 4000;
    unsigned int *stop, *dstop; double *sink;
    hipHostMalloc((void **)&stop, 64, hipHostMallocDefault); hipHostGetDevicePointer((void **)&dstop, stop, 0);
    hipMalloc((void **)&sink, 64);
    hipStream_t bg; hipStreamCreateWithFlags(&bg, hipStreamNonBlocking);
    for (int spin_blocks : {0, 1, 8, 64, 256, 1024}) {
    *stop = 0;
    if (spin_blocks) hipLaunchKernelGGL(k_spin, dim3(spin_blocks), dim3(256), 0, bg, dstop, sink);
    printf("== background load: %d workgroups of 256 lanes spinning on f64 FMAs\n", spin_blocks);
    {
        std::vector<double> f;
        for (int i = 0; i < 600; ++i) {
            ++seq;
            hipLaunchKernelGGL(k_sleep, dim3(1), dim3(64), 0, st, dout, dword, seq, 8);
            while (__atomic_load_n(word, __ATOMIC_ACQUIRE) != seq) _mm_pause();
            if (i >= 100) f.push_back(8.0 * 127 * 64 / (out[0] * 0.01));   // cycles / us = MHz
            const auto te = std::chrono::steady_clock::now() + std::chrono::microseconds(30);
            while (std::chrono::steady_clock::now() < te) _mm_pause();
        }
        std::sort(f.begin(), f.end());
        printf("s_sleep: 8 x 127 x 64 cycles take -> shader clock p10 %.0f  p50 %.0f  p90 %.0f MHz (if s_sleep counts shader cycles)\n", f[f.size() / 10], f[f.size() / 2], f[f.size() * 9 / 10]);
    }
    for (int gap : {0, 50}) {
        std::vector<double> mhz, us, cyc;
        for (int i = 0; i < 1500; ++i) {
            ++seq;
            hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, st, dout, dword, seq, n, 1.0);
            while (__atomic_load_n(word, __ATOMIC_ACQUIRE) != seq) _mm_pause();
            if (i >= 500) { mhz.push_back(100.0 * (double)out[1] / (double)out[0]); us.push_back(out[0] * 0.01); cyc.push_back((double)out[1] / n); }
            const auto te = std::chrono::steady_clock::now() + std::chrono::microseconds(gap);
            while (std::chrono::steady_clock::now() < te) _mm_pause();
        }
        std::sort(mhz.begin(), mhz.end()); std::sort(us.begin(), us.end()); std::sort(cyc.begin(), cyc.end());
        printf("host gap %4d us between launches: loop of %d dependent f64 FMAs takes p50 %.1f us (p10 %.1f, p90 %.1f) = %.1f shader cycles per iteration (FMA + loop branch); clock64 / wall_clock64 -> p10 %.0f  p50 %.0f  p90 %.0f MHz\n", gap, n,
               us[us.size() / 2], us[us.size() / 10], us[us.size() * 9 / 10], cyc[cyc.size() / 2], mhz[mhz.size() / 10], mhz[mhz.size() / 2], mhz[mhz.size() * 9 / 10]);
    }
    *stop = 1;
    hipStreamSynchronize(bg);
    }
    return 0;
}
