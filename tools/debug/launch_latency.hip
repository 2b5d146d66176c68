// What one kernel launch costs on this box from the host's side, whatever the kernel does: launch an (almost) empty kernel and wait for it —
// (a) polling a word in page-locked host memory that the kernel stores (system-scope release), (b) polling hipStreamQuery, (c) hipStreamSynchronize.
// The floor under f110_step_host's in-call time for a tiny batch.   hipcc --offload-arch=gfx950 -O3 launch_latency.hip -o launch_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <immintrin.h>

__global__ void k_signal(unsigned long long *word, unsigned long long seq, int spin)
{
    // `spin` dependent steps of device-side work (0 = none)
    unsigned long long v = seq;
    for (int i = 0; i < spin; ++i) v = v * 6364136223846793005ull + 1442695040888963407ull;
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(word, v == 0 ? seq + 1 : seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static double pct(std::vector<double> &v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; }

int main()
{
    hipStream_t st;
    hipStreamCreate(&st);
    unsigned long long *word = nullptr, *dword = nullptr;
    hipHostMalloc(reinterpret_cast<void **>(&word), 64, hipHostMallocDefault);
    hipHostGetDevicePointer(reinterpret_cast<void **>(&dword), word, 0);
    *word = 0;
    const int n = 5000;
    for (int mode = 0; mode < 3; ++mode) {
        for (int spin : {0, 2000}) {
            std::vector<double> enq, tot;
            unsigned long long seq = (unsigned long long)mode * 1000000ull + (spin ? 500000ull : 0ull);
            for (int i = 0; i < n + 200; ++i) {
                ++seq;
                const auto t0 = std::chrono::steady_clock::now();
                hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, st, dword, seq, spin);
                const auto t1 = std::chrono::steady_clock::now();
                if (mode == 0) {
                    while (__atomic_load_n(word, __ATOMIC_ACQUIRE) != seq) _mm_pause();
                } else if (mode == 1) {
                    while (hipStreamQuery(st) == hipErrorNotReady) _mm_pause();
                } else {
                    hipStreamSynchronize(st);
                }
                const auto t2 = std::chrono::steady_clock::now();
                if (i >= 200) {
                    enq.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
                    tot.push_back(std::chrono::duration<double, std::micro>(t2 - t0).count());
                }
            }
            if (mode == 0) hipStreamSynchronize(st);
            const char *names[] = {"poll a host word the kernel stores", "poll hipStreamQuery", "hipStreamSynchronize"};
            printf("%-36s kernel of %4d dependent steps: launch call %.1f us (p50), launch -> host sees completion p10 %.1f p50 %.1f p90 %.1f us\n", names[mode], spin,
                   pct(enq, 0.5), pct(tot, 0.1), pct(tot, 0.5), pct(tot, 0.9));
        }
    }
    return 0;
}
