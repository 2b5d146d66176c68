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

// (c) partially active waves: the same u64 gather with only the lanes of `active` executing it
__global__ void __launch_bounds__(256) k_gather_masked(const char *__restrict__ tbl, uint32_t mask, uint32_t window, uint64_t active,
                                                       int iters, uint64_t *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t h = (lane + 1u) * 2654435761u ^ (wave * 40503u);
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    uint32_t off = ((h & (window - 1u)) + wave * 4096u) & mask;
    uint64_t acc = 0;
    if ((active >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
            const uint64_t v = *reinterpret_cast<const uint64_t *>(tbl + (off & ~7u));
            acc += v;
            off = (off + 8192u + 136u) & mask;
        }
    }
    if (acc == 0x1234567u) out[0] = acc;
}

// (d) the LDS alternative (north_star's "grid in LDS"): one byte code gathered from a window staged in
// LDS, then the dependent 8-byte value from a 256-entry LUT in LDS — per sample two LDS reads instead
// of one global gather.  Cycles per wave-level (code, value) pair at 8 waves/SIMD... as many waves as
// the LDS footprint admits.
template <int WIN_BYTES>
__global__ void __launch_bounds__(256) k_lds_pair(int iters, uint32_t spread, uint64_t *out)
{
    __shared__ uint8_t win[WIN_BYTES];
    __shared__ double lut[256];
    for (int t = threadIdx.x; t < WIN_BYTES; t += blockDim.x) win[t] = (uint8_t)((t * 37u + (t >> 7)) & 0xffu);
    for (int t = threadIdx.x; t < 256; t += blockDim.x) lut[t] = 1.0 + t;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t h = (lane + 1u) * 2654435761u ^ (wave * 40503u + blockIdx.x * 977u);
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    uint32_t off = (h & (spread - 1u)) % WIN_BYTES;
    double acc = 0.;
    for (int i = 0; i < iters; ++i) {
        const uint32_t code = win[off];
        const double v = lut[code];
        acc += v;
        off = (off + 4099u + (uint32_t)v) % WIN_BYTES;   // next address depends on the value, as the march does
    }
    if (acc == 0.123) out[0] = (uint64_t)acc;
}

template <int WIN_BYTES>
static double run_lds(uint32_t spread, uint64_t *d_out, int cus, double mhz, int *waves_per_cu)
{
    const int iters = 4096, threads = 256;
    int per_cu = (160 * 1024) / (WIN_BYTES + 2048);
    if (per_cu > 8) per_cu = 8;
    const int blocks = cus * per_cu;
    *waves_per_cu = per_cu * 4;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k_lds_pair<WIN_BYTES>, dim3(blocks), dim3(threads), 0, 0, 64, spread, d_out);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k_lds_pair<WIN_BYTES>, dim3(blocks), dim3(threads), 0, 0, iters, spread, d_out);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e-3 * mhz * 1e6 / ((double)per_cu * 4.0 * iters);
}

static double run_masked(const char *d_tbl, uint32_t bytes, uint32_t window, uint64_t active, uint64_t *d_out, int cus, double mhz)
{
    const int iters = 4096, blocks = cus * 8, threads = 256;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k_gather_masked, dim3(blocks), dim3(threads), 0, 0, d_tbl, bytes - 1, window, active, 64, d_out);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k_gather_masked, dim3(blocks), dim3(threads), 0, 0, d_tbl, bytes - 1, window, active, iters, d_out);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e-3 * mhz * 1e6 / (32.0 * iters);
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
        if (bytes == (1u << 20)) {
            printf("partially active waves, u64 gather, random inside a 2048-byte window: cycles per wave-level load\n");
            struct { const char *name; uint64_t m; } masks[] = {
                {"all 64 lanes", ~0ull}, {"lanes 0-47", (1ull << 48) - 1}, {"lanes 0-31", (1ull << 32) - 1}, {"lanes 0-15", 0xffffull},
                {"lanes 0-3", 0xfull}, {"every other quad (32)", 0x0f0f0f0f0f0f0f0full}, {"1 lane per quad (16)", 0x1111111111111111ull},
                {"every other lane (32)", 0x5555555555555555ull}, {"scattered 37 lanes", 0x9b5e3d27a4c6f1b3ull}};
            for (auto &mk : masks) printf("%28s %8.1f\n", mk.name, run_masked(d_tbl, bytes, 2048u, mk.m, d_out, cus, mhz));
        }
        hipFree(d_tbl);
    }
    printf("LDS window: cycles per wave-level (1-byte code from the window, dependent 8-byte LUT value) pair\n");
    printf("%10s %10s %10s %10s\n", "window", "spread", "waves/CU", "cycles");
    for (uint32_t spread : {256u, 4096u, 1u << 20}) {
        int w;
        double c = run_lds<16384>(spread, d_out, cus, mhz, &w);
        printf("%10d %10u %10d %10.1f\n", 16384, spread, w, c);
        c = run_lds<36864>(spread, d_out, cus, mhz, &w);
        printf("%10d %10u %10d %10.1f\n", 36864, spread, w, c);
        c = run_lds<65536>(spread, d_out, cus, mhz, &w);
        printf("%10d %10u %10d %10.1f\n", 65536, spread, w, c);
    }
    return 0;
}
