// Micro-benchmark: cost of one wave-level gather on a gfx950 CU as a function of (a) how many
// distinct 128-byte lines the 64 lanes touch and (b) the access width.  The table is L1/L2
// resident, every CU runs 8 waves/SIMD, and the loop is 1 load + 3 integer VALU ops, so the
// vector-memory path (TA/TCP/TD) is the limiter; the figure printed is CU cycles per
// wave-instruction at saturation.   hipcc --offload-arch=gfx950 -O3 ta_bench.hip -o ta_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

template <typename T>
__global__ void __launch_bounds__(256) k_gather(const char *__restrict__ tbl, uint32_t mask, uint32_t lane_stride,
                                                uint32_t iter_stride, int iters, uint64_t *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t off;
    if (lane_stride & 0x80000000u) {  // random placement inside a window of (lane_stride & 0xffffff) bytes
        uint32_t h = (lane + 1u) * 2654435761u ^ (wave * 40503u);
        h ^= h >> 15;
        h *= 2246822519u;
        h ^= h >> 13;
        off = ((h & ((lane_stride & 0xffffffu) - 1u)) + wave * 4096u) & mask;
    } else {
        off = (lane * lane_stride + wave * 4096u) & mask;
    }
    uint64_t acc = 0;
    for (int i = 0; i < iters; ++i) {
        const T v = *reinterpret_cast<const T *>(tbl + (off & ~(uint32_t)(sizeof(T) - 1)));
        acc += (uint64_t)v;
        off = (off + iter_stride) & mask;
    }
    if (acc == 0x1234567u) out[0] = acc;  // never true; keeps the loads alive
}

template <typename T>
static double run(const char *d_tbl, uint32_t bytes, uint32_t lane_stride, uint64_t *d_out, int cus, double mhz)
{
    const int iters = 4096, blocks = cus * 8, threads = 256;  // 8 blocks x 4 waves = 32 waves per CU
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k_gather<T>, dim3(blocks), dim3(threads), 0, 0, d_tbl, bytes - 1, lane_stride, 8192u + 136u, 64,
                       d_out);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k_gather<T>, dim3(blocks), dim3(threads), 0, 0, d_tbl, bytes - 1, lane_stride, 8192u + 136u, iters,
                       d_out);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double wave_instrs_per_cu = 32.0 * iters;
    return ms * 1e-3 * mhz * 1e6 / wave_instrs_per_cu;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double mhz = p.clockRate / 1000.0;
    printf("%s  CUs %d  clock %.0f MHz\n", p.name, cus, mhz);
    uint64_t *d_out;
    hipMalloc(&d_out, 8);
    for (uint32_t bytes : {16u << 10, 1u << 20}) {
        char *d_tbl;
        hipMalloc(&d_tbl, bytes);
        hipMemset(d_tbl, 1, bytes);
        printf("table %u KiB: cycles per wave-level load (lane stride bytes -> distinct 128 B lines)\n", bytes >> 10);
        printf("%10s %8s %8s %8s %8s\n", "stride", "lines", "u8", "u32", "u64");
        for (uint32_t ls : {0u, 1u, 2u, 4u, 8u, 16u, 32u, 64u, 128u, 136u, 520u}) {
            const uint32_t span = ls * 63u;
            const uint32_t lines = ls >= 128 ? 64 : span / 128 + 1;
            const double c1 = run<uint8_t>(d_tbl, bytes, ls, d_out, cus, mhz);
            const double c4 = run<uint32_t>(d_tbl, bytes, ls, d_out, cus, mhz);
            const double c8 = run<uint64_t>(d_tbl, bytes, ls, d_out, cus, mhz);
            printf("%10u %8u %8.1f %8.1f %8.1f\n", ls, lines, c1, c4, c8);
        }
        printf("%10s %8s %8s %8s   (lanes placed at random inside a window)\n", "window", "u8", "u32", "u64");
        for (uint32_t w : {64u, 128u, 256u, 512u, 1024u, 2048u, 4096u, 8192u, 16384u}) {
            const uint32_t ls = 0x80000000u | w;
            const double c1 = run<uint8_t>(d_tbl, bytes, ls, d_out, cus, mhz);
            const double c4 = run<uint32_t>(d_tbl, bytes, ls, d_out, cus, mhz);
            const double c8 = run<uint64_t>(d_tbl, bytes, ls, d_out, cus, mhz);
            printf("%10u %8.1f %8.1f %8.1f\n", w, c1, c4, c8);
        }
        hipFree(d_tbl);
    }
    return 0;
}
