// r03_store_bench.hip -- HBM write rate of the two store shapes of the row-panel epilogues: (A) a wave instruction writes 16 rows x
// 64 contiguous bytes (rows 4 KiB apart: the MFMA C^T layout stored as it is), (B) a wave instruction writes one 1 KiB row segment.
// 225 workgroups x 512 threads, 32 rows x 1024 floats per workgroup and array, rotating destination buffers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
template <int MODE>
__global__ void __launch_bounds__(512) k_store(float* __restrict__ dst, int ld) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const long m0 = (long)blockIdx.x * 32;
    const f32x4 v = (f32x4){(float)tid, 1.f, 2.f, 3.f};
    if (MODE == 0) {
        for (int c = 0; c < ld / 256; ++c)
            for (int t = 0; t < 2; ++t)
                for (int i = 0; i < 2; ++i) {
                    const int n = 256 * c + 32 * wave + 16 * t + 4 * g;
                    *(f32x4*)(dst + (m0 + 16 * i + l15) * ld + n) = v;
                }
    } else {
        for (int c = 0; c < ld / 256; ++c)
            for (int r = 0; r < 4; ++r) *(f32x4*)(dst + (m0 + 4 * wave + r) * ld + 256 * c + 4 * lane) = v;
    }
}
template <int MODE>
static void run(const char* name, float** bufs, int nb, int ld) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_store<MODE>, dim3(225), dim3(512), 0, 0, bufs[i % nb], ld);
    CK(hipEventRecord(e0, 0));
    const int R = 32;
    for (int i = 0; i < R; ++i) hipLaunchKernelGGL(k_store<MODE>, dim3(225), dim3(512), 0, 0, bufs[i % nb], ld);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / R, bytes = 7200.0 * ld * 4;
    printf("%-40s ld %4d: %6.2f us  %.2f TB/s\n", name, ld, us, bytes / us / 1e6);
}
int main() {
    const int NB = 12;
    float* bufs[NB];
    for (int i = 0; i < NB; ++i) CK(hipMalloc(&bufs[i], (size_t)7200 * 1024 * 4));
    run<0>("16 rows x 64 B per instruction", bufs, NB, 1024);
    run<1>("1 KiB row segment per instruction", bufs, NB, 1024);
    run<0>("16 rows x 64 B per instruction", bufs, NB, 256);
    run<1>("1 KiB row segment per instruction", bufs, NB, 256);
    run<0>("16 rows x 64 B, one warm buffer", bufs, 1, 1024);
    run<1>("1 KiB rows, one warm buffer", bufs, 1, 1024);
    return 0;
}
