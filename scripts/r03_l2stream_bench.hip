// r03_l2stream_bench.hip -- how fast can every CU pull the SAME L2-resident weight (1 MiB) with coalesced 1 KiB wave loads?
// (the bound of the row-panel GEMMs: every workgroup streams the whole weight).  Modes:
//   0 every workgroup reads the buffer in the same order     1 start rotated per workgroup     2 private 1 MiB per workgroup (HBM)
//   3 private 64 KiB per workgroup read 16 times (L2-resident, no sharing)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int MODE, int INFLIGHT, int THREADS>
__global__ void __launch_bounds__(THREADS) k_stream(const f32x4* __restrict__ buf, long units_per_wg /*16-B units*/, float* __restrict__ sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = THREADS / 64;
    const long per_wave = units_per_wg / NW;          // units (16 B) per wave
    const long steps = per_wave / 64;                 // 1 KiB wave loads
    long base = 0, rot = 0, wrap = steps;
    if (MODE == 2) base = (long)blockIdx.x * units_per_wg;
    if (MODE == 3) { base = (long)blockIdx.x * (units_per_wg / 16); wrap = steps / 16; }
    if (MODE == 1) rot = ((long)(blockIdx.x >> 3) * 37) % steps;
    const f32x4* p = buf + base + (MODE == 3 ? (long)wave * (per_wave / 16) : (long)wave * per_wave) + lane;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (long s = 0; s < steps; s += INFLIGHT) {
        f32x4 v[INFLIGHT];
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) {
            long t = s + u + rot;
            if (MODE == 1 && t >= steps) t -= steps;
            if (MODE == 3) t %= wrap;
            v[u] = p[t * 64];
        }
#pragma unroll
        for (int u = 0; u < INFLIGHT; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

template <int MODE, int INFLIGHT, int THREADS>
static void run(const char* name, const f32x4* buf, float* sink, int grid) {
    const long units = (1 << 20) / 16;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_stream<MODE, INFLIGHT, THREADS>), dim3(grid), dim3(THREADS), 0, 0, buf, units, sink);
    CK(hipEventRecord(e0, 0));
    const int R = 20;
    for (int i = 0; i < R; ++i) hipLaunchKernelGGL((k_stream<MODE, INFLIGHT, THREADS>), dim3(grid), dim3(THREADS), 0, 0, buf, units, sink);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / R, bytes = (double)grid * (1 << 20);
    printf("%-34s grid %3d thr %4d inflight %2d: %7.2f us  %6.2f TB/s  %5.1f B/clk/CU(2.4GHz)\n", name, grid, THREADS, INFLIGHT, us, bytes / us / 1e6,
           (1 << 20) / (us * 2400.0));
}

int main() {
    f32x4* buf; float* sink;
    CK(hipMalloc(&buf, (size_t)256 << 20)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, (size_t)256 << 20));
    run<0, 8, 512>("same order", buf, sink, 256);
    run<0, 16, 512>("same order", buf, sink, 256);
    run<0, 4, 512>("same order", buf, sink, 256);
    run<0, 8, 256>("same order", buf, sink, 256);
    run<0, 8, 1024>("same order", buf, sink, 256);
    run<0, 8, 512>("same order, 225 wgs", buf, sink, 225);
    run<0, 8, 512>("same order, 512 wgs (2/CU)", buf, sink, 512);
    run<1, 8, 512>("rotated start", buf, sink, 256);
    run<1, 16, 512>("rotated start", buf, sink, 256);
    run<2, 8, 512>("private 1 MiB (HBM)", buf, sink, 256);
    run<2, 16, 512>("private 1 MiB (HBM)", buf, sink, 256);
    run<3, 8, 512>("private 64 KiB x16 (L2)", buf, sink, 256);
    run<3, 16, 512>("private 64 KiB x16 (L2)", buf, sink, 256);
    run<3, 8, 1024>("private 64 KiB x16 (L2)", buf, sink, 256);
    return 0;
}
