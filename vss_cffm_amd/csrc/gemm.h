// gemm.h -- the plain dense Linear GEMMs of the block (qkv, proj, fc1, fc2 and their gradients), fp32.
// These are library GEMMs (rocBLAS SGEMM on the caller's stream); the hand-written kernels of this
// library are the CFFA / CFM path around them.  Row-major operands are mapped onto rocBLAS'
// column-major interface by swapping operand roles (C^T = B^T A^T), never by copying.
#pragma once
#include "cffm_common.h"

#ifdef CFFM_EMU
// TEST INFRASTRUCTURE (emulator build only): naive host loops standing in for rocBLAS.
static int gemm_nt(const float* x, const float* w, float* y, long M, int N, int K, hipStream_t) {
    for (long m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double acc = 0;
            for (int k = 0; k < K; ++k) acc += (double)x[m * K + k] * w[(long)n * K + k];
            y[m * N + n] = (float)acc;
        }
    return 0;
}
static int gemm_nn(const float* dy, const float* w, float* dx, long M, int N, int K, hipStream_t) {
    for (long m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k) {
            double acc = 0;
            for (int n = 0; n < N; ++n) acc += (double)dy[m * N + n] * w[(long)n * K + k];
            dx[m * K + k] = (float)acc;
        }
    return 0;
}
static int gemm_tn(const float* dy, const float* x, float* dw, long M, int N, int K, hipStream_t) {
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            double acc = 0;
            for (long m = 0; m < M; ++m) acc += (double)dy[m * N + n] * x[m * K + k];
            dw[(long)n * K + k] = (float)acc;
        }
    return 0;
}
#else
#include <rocblas/rocblas.h>
static rocblas_handle g_rocblas = nullptr;
static int gemm_ready(hipStream_t st) {
    if (!g_rocblas) {
        if (rocblas_create_handle(&g_rocblas) != rocblas_status_success) return -1;
        rocblas_set_pointer_mode(g_rocblas, rocblas_pointer_mode_host);
    }
    return rocblas_set_stream(g_rocblas, st) == rocblas_status_success ? 0 : -1;
}
// y[M,N] = x[M,K] w[N,K]^T      (col-major: y^T[N,M] = w_cm^T[N,K] x_cm[K,M])
static int gemm_nt(const float* x, const float* w, float* y, long M, int N, int K, hipStream_t st) {
    if (gemm_ready(st)) return -1;
    const float one = 1.f, zero = 0.f;
    return rocblas_sgemm(g_rocblas, rocblas_operation_transpose, rocblas_operation_none, N, (int)M, K, &one, w, K, x, K, &zero, y, N) ==
                   rocblas_status_success ? 0 : -1;
}
// dx[M,K] = dy[M,N] w[N,K]      (col-major: dx^T[K,M] = w_cm[K,N] dy_cm[N,M])
static int gemm_nn(const float* dy, const float* w, float* dx, long M, int N, int K, hipStream_t st) {
    if (gemm_ready(st)) return -1;
    const float one = 1.f, zero = 0.f;
    return rocblas_sgemm(g_rocblas, rocblas_operation_none, rocblas_operation_none, K, (int)M, N, &one, w, K, dy, N, &zero, dx, K) ==
                   rocblas_status_success ? 0 : -1;
}
// dw[N,K] = dy[M,N]^T x[M,K]    (col-major: dw^T[K,N] = x_cm[K,M] dy_cm^T[M,N])
static int gemm_tn(const float* dy, const float* x, float* dw, long M, int N, int K, hipStream_t st) {
    if (gemm_ready(st)) return -1;
    const float one = 1.f, zero = 0.f;
    return rocblas_sgemm(g_rocblas, rocblas_operation_none, rocblas_operation_transpose, K, N, (int)M, &one, x, K, dy, N, &zero, dw, K) ==
                   rocblas_status_success ? 0 : -1;
}
#endif
