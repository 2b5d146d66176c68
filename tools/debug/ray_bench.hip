// Micro-benchmark (round 3): how fast can ONE long wall-grazing ray be marched?
//   (a) march_padded, the step kernel's loop: one dependent table gather per sample (64 lanes = 64 rays)
//   (b) march_padded_block, one ray per wave: 16 x 16-cell blocks in registers, next block requested ahead
// on a synthetic corridor (a wall along the x axis; the rays run parallel to it at a constant distance, so
// every sample advances `h` cells: the creeping regime of the scan's longest rays).  Prints ns per sample for
// an otherwise idle GPU and with the rest of the chip busy gathering from the same table.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I f1tenth_gym_amd/csrc tools/debug/ray_bench.hip -o tools/debug/ray_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "f110_kernels.hpp"

__global__ void k_lockstep(ScanConst k, double ux0, double uy0, double cux, double cuy, double d0, double *out, int *nl_out, long long *cyc)
{
    const uint32_t lane = threadIdx.x & 63u;
    int hr, hc, nl;
    double r;
    const long long t0 = wall_clock64();
    march_padded<false>(k, ux0 + 0.37 * lane, uy0 + 1e-3 * lane, cux, cuy, d0, r, hr, hc, nl);
    const long long t1 = wall_clock64();
    out[blockIdx.x * 64 + lane] = r;
    if (lane == 0 && blockIdx.x == 0) { *nl_out = nl; *cyc = t1 - t0; }
}

__global__ void k_block(ScanConst k, double ux0, double uy0, double cux, double cuy, double d0, double *out, int *nl_out, long long *cyc)
{
    int nl, nb;
    double r;
    const long long t0 = wall_clock64();
    march_padded_block(k, ux0 + 0.37 * blockIdx.x, uy0, cux, cuy, d0, r, nl, nb);
    const long long t1 = wall_clock64();
    out[blockIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) { *nl_out = nl; *cyc = t1 - t0; }
}

// background load: every other wave slot gathers from the table like the scan does
__global__ void k_noise(const double *tbl, uint32_t mask, int iters, double *sink)
{
    uint32_t h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
    double acc = 0;
    for (int i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        acc += tbl[(h >> 4) & mask];
    }
    if (acc == 1.234) *sink = acc;
}

int main()
{
    const int W = 2692, H = 2692;   // example_map's padded size
    const double res = 0.0625;
    std::vector<double> t((size_t)W * H);
    const int wall_row = 1000;
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) t[(size_t)r * W + c] = res * fabs((double)(r - wall_row));
    double *d_t, *d_out; int *d_nl; long long *d_cyc;
    hipMalloc(&d_t, t.size() * 8); hipMemcpy(d_t, t.data(), t.size() * 8, hipMemcpyHostToDevice);
    hipMalloc(&d_out, 1 << 20); hipMalloc(&d_nl, 4); hipMalloc(&d_cyc, 8);
    ScanConst k{};
    k.pad = d_t; k.pad_width = W; k.pad_height = H; k.pad_row_bytes = W * 8; k.eps = 1e-4; k.max_range = 30.0; k.pad_max_samples = 100000;
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    for (double h : {1.2, 2.5}) {          // cells between the ray and the wall = cells advanced per sample
        for (int busy = 0; busy < 2; ++busy) {
            for (int which = 0; which < 2; ++which) {
                const double ux0 = 600.3, uy0 = wall_row + h + 0.4, cux = 1.0 / res, cuy = 0.0, d0 = res * h;
                if (busy) hipLaunchKernelGGL(k_noise, dim3(256 * 16), dim3(256), 0, s2, d_t, (uint32_t)((1u << 22) - 1), 4000, d_out + 100000);
                hipDeviceSynchronize == nullptr ? (void)0 : (void)0;
                if (which == 0) hipLaunchKernelGGL(k_lockstep, dim3(64), dim3(64), 0, s1, k, ux0, uy0, cux, cuy, d0, d_out, d_nl, d_cyc);
                else hipLaunchKernelGGL(k_block, dim3(256), dim3(64), 0, s1, k, ux0, uy0, cux, cuy, d0, d_out, d_nl, d_cyc);
                hipDeviceSynchronize();
                int nl = 0; long long cyc = 0; double r0 = 0;
                hipMemcpy(&nl, d_nl, 4, hipMemcpyDeviceToHost); hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&r0, d_out, 8, hipMemcpyDeviceToHost);
                printf("h = %.1f cells/sample  %-9s  %-34s  samples %4d  range %.4f  %.1f ns per sample (wall_clock64 at 100 MHz)\n", h, busy ? "chip busy" : "chip idle",
                       which == 0 ? "lock-step march_padded (64 rays)" : "march_padded_block (1 ray/wave)", nl, r0, (double)cyc * 10.0 / nl);
            }
        }
    }
    return 0;
}
