// r03_panel_bench.hip -- standalone micro-benchmark of the row-panel GEMM (vss_cffm_amd/csrc/panel_kernels.h) at the block's
// shapes: correctness against fp64 on sampled rows, then back-to-back launch time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/r03_panel_bench.hip -o build/r03_panel_bench
#define CFFM_EXPERIMENTS 1
#include "../vss_cffm_amd/csrc/panel_kernels.h"
#include "r03_panel_experiments.h"
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

template <int MT, int NTW, int NT, int D>
static void run(const char* name, int M, int N, int K, bool nn) {
    if (N != PNL_WAVES * NTW * 16) { printf("%s: bad N\n", name); return; }
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hC((size_t)M * N);
    for (auto& v : hA) v = frand();
    for (auto& v : hW) v = frand() * 0.1f;
    float *dA, *dW, *dC; f32x4* dWf;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&dC, hC.size() * 4)); CK(hipMalloc(&dWf, hW.size() * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    // the weight as stored: NT form W[N][K]; NN form: the kernel computes C = A Wst with Wst [K][N] stored row-major ([contraction][out])
    const int Wrows = nn ? K : N, Wcols = nn ? N : K;
    hipLaunchKernelGGL(k_pnl_pack_weight, dim3((unsigned)(((long)N * K / 8 + 255) / 256)), dim3(256), 0, 0, dW, Wrows, Wcols, nn ? 1 : 0, dWf);
    CK(hipGetLastError());
    auto kern = k_panel_gemm<MT, NTW, NT, D, false, 0>;
    const int lds = PNL_LDS(MT);
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int grid = (M + 16 * MT - 1) / (16 * MT);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(PNL_THREADS), lds, 0, dA, K, M, K, dWf, dC, N, nullptr);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (int s = 0; s < 48; ++s) {
        const int m = (s * 151 + (s % 3 == 0 ? M - 1 - s : 0)) % M;
        for (int n = 0; n < N; ++n) {
            double acc = 0;
            for (int k = 0; k < K; ++k) acc += (double)hA[(size_t)m * K + k] * (nn ? hW[(size_t)k * N + n] : hW[(size_t)n * K + k]);
            maxerr = fmax(maxerr, fabs(acc - hC[(size_t)m * N + n]));
            maxref = fmax(maxref, fabs(acc));
        }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(PNL_THREADS), lds, 0, dA, K, M, K, dWf, dC, N, nullptr);
    CK(hipEventRecord(e0, 0));
    const int R = 50;
    for (int i = 0; i < R; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(PNL_THREADS), lds, 0, dA, K, M, K, dWf, dC, N, nullptr);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / R;
    printf("%-26s M=%5d N=%4d K=%4d MT=%d NTW=%d NT=%d D=%d grid=%d: %.2f us  %.0f TF(fp32-equiv)  rel.err %.2e\n", name, M, N, K, MT, NTW, NT, D, grid, us,
           2.0 * M * N * K / us / 1e6, maxerr / maxref);
    CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(dC)); CK(hipFree(dWf));
}

int main() {
    srand(1);
    printf("PNL_ABLATE=%d\n", PNL_ABLATE);
    run<2, 2, 2, 4>("proj", 7200, 256, 256, false);
    run<2, 8, 2, 4>("fc1 fwd NT2 D4", 7200, 1024, 256, false);
    run<2, 2, 2, 4>("fc2 fwd", 7200, 256, 1024, false);
    run<3, 6, 2, 4>("qkv fwd MT3", 10368, 768, 256, false);
    run<3, 2, 2, 4>("qkv dX MT3 (nn)", 10368, 256, 768, true);
    return 0;
}
