#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <immintrin.h>
struct Big { unsigned long long v[200]; };   // 1600 bytes of kernel arguments
__device__ __forceinline__ unsigned long long wall() { return wall_clock64(); }
template <bool BIG>
__global__ void __launch_bounds__(256) k_busy(unsigned long long *word, unsigned long long seq, int ticks, Big big)
{
    const unsigned long long t0 = wall();
    unsigned long long acc = 0;
    if (BIG) for (int i = 0; i < 200; i += 37) acc += big.v[i];
    while (wall() - t0 < (unsigned long long)ticks) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(word, seq + (acc == 12345ull ? 1 : 0), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
static double pct(std::vector<double> &v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; }
int main()
{
    hipStream_t st; hipStreamCreate(&st);
    unsigned long long *word, *dword;
    hipHostMalloc((void **)&word, 64, hipHostMallocDefault); hipHostGetDevicePointer((void **)&dword, word, 0); *word = 0;
    Big big{}; for (int i = 0; i < 200; ++i) big.v[i] = i;
    unsigned long long seq = 0;
    for (int bigarg = 0; bigarg < 2; ++bigarg) for (int wgs : {1, 9}) for (int ticks : {0, 3900}) for (int gap : {0, 12}) {
        std::vector<double> tot;
        for (int i = 0; i < 3200; ++i) {
            ++seq;
            const auto t0 = std::chrono::steady_clock::now();
            if (bigarg) hipLaunchKernelGGL(k_busy<true>, dim3(wgs), dim3(256), 0, st, dword, seq, ticks, big);
            else hipLaunchKernelGGL(k_busy<false>, dim3(wgs), dim3(256), 0, st, dword, seq, ticks, big);
            while (__atomic_load_n(word, __ATOMIC_ACQUIRE) != seq) _mm_pause();
            const auto t2 = std::chrono::steady_clock::now();
            if (i >= 200) tot.push_back(std::chrono::duration<double, std::micro>(t2 - t0).count());
            if (gap) { const auto te = t2 + std::chrono::microseconds(gap); while (std::chrono::steady_clock::now() < te) _mm_pause(); }
        }
        hipStreamSynchronize(st);
        printf("args %s  workgroups %d  kernel busy %4.1f us  host gap %2d us : launch -> word seen p10 %.1f p50 %.1f p90 %.1f us  (overhead p50 %.1f)\n", bigarg ? "1.6 KB used" : "1.6 KB unused",
               wgs, ticks * 0.01, gap, pct(tot, 0.1), pct(tot, 0.5), pct(tot, 0.9), pct(tot, 0.5) - ticks * 0.01);
    }
    return 0;
}
