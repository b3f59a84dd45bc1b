// gemm.h -- host side of the dense Linear GEMMs (qkv, proj, fc1, fc2 and their gradients; the head's classifiers and embedding):
// launchers of the hand-written split-bf16 MFMA kernels of gemm_kernels.h (fp32 operands split into bf16 hi + lo, three MFMA
// products per fp32 product, fp32 accumulate).  There is no library GEMM behind this file: rounds 1-3 carried a dlopen-ed
// rocBLAS SGEMM as a cross-check (CFFM_GEMM=lib); it left with the rest of the alternative code paths in round 4.
#pragma once
#include <stdlib.h>
#include "gemm_kernels.h"
#include "dw_kernels.h"
#include "dws_kernels.h"

// ---- hand-written split-bf16 MFMA path (default) -------------------------------------------------------------
float* lib_scratch(size_t nfloats);   // cffm_hip.hip: library-owned device scratch (grows on demand)
__global__ void k_sum_splits(const float* __restrict__ part, int nsplit, long n, float* __restrict__ out);

template <bool A_T, bool B_T, int EPI = 0, bool A_PRE = false, bool B_PRE = false>
static int gemm_split_launch(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int ksplit,
                             hipStream_t st, const float* bias = nullptr, float* aux = nullptr, int prefer_big = 0,
                             float* aux2 = nullptr) {
    int klen = ((K + ksplit - 1) / ksplit + 63) / 64 * 64;
    ksplit = (K + klen - 1) / klen;
    float* out = C;
    const long split_stride = (long)M * ldc;
    if (ksplit > 1) {
        out = lib_scratch((size_t)ksplit * split_stride);
        if (!out) return -1;
    }
    const long b128 = (long)((N + 127) / 128) * ((M + 127) / 128) * ksplit;
#ifdef CFFM_EMU
#define GEMM_BIG_LDS(BM_, BN_, BK_, PF_)
#else   // more than 64 KiB of dynamic LDS has to be granted per kernel, once
#define GEMM_BIG_LDS(BM_, BN_, BK_, PF_)                                                                                      \
    if (GEMM_LDS(BM_, BN_, BK_) > 65536) {                                                                                    \
        static bool granted = false;                                                                                          \
        if (!granted) {                                                                                                       \
            if (hipFuncSetAttribute((const void*)k_gemm_split<BM_, BN_, BK_, A_T, B_T, EPI, PF_, A_PRE, B_PRE>,                              \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS(BM_, BN_, BK_)) != hipSuccess)       \
                return -1;                                                                                                    \
            granted = true;                                                                                                   \
        }                                                                                                                     \
    }
#endif
#define GEMM_GO(BM_, BN_, BK_, PF_) do {                                                                                               \
    GEMM_BIG_LDS(BM_, BN_, BK_, PF_)                                                                                                   \
    CFFM_LAUNCH((k_gemm_split<BM_, BN_, BK_, A_T, B_T, EPI, PF_, A_PRE, B_PRE>), ((unsigned)(((N + BN_ - 1) / BN_) * ((M + BM_ - 1) / BM_) * ksplit)), (256), \
                GEMM_LDS(BM_, BN_, BK_), st, A, \
                B, out, M, N, K, lda, ldb, ldc, klen, split_stride, bias, aux, aux2); } while (0)
    // measured on MI355X (scripts/gemm_bench.py, CFFM-B1 shapes): 128x128 wins when it already gives >= 384 workgroups
    // (qkv / fc1 forward, the 1024-wide input gradient), 64x64 otherwise; prefetch depth beyond the listed one is neutral.
    static int sel = -1;   // tuning aid: CFFM_GEMM_SEL = 1 -> 128x64 tiles, 2 -> 64x128, 3 -> 128x128 for the small cases
    if (sel < 0) { const char* e = cffm_tune("CFFM_GEMM_SEL"); sel = e ? atoi(e) : 0; }
    if (b128 >= 384 || prefer_big) {
        GEMM_GO(128, 128, 32, 1);   // a second K-tile in flight in registers: neutral (k-contiguous forms) or one workgroup per CU (others)
    } else if (sel == 1) GEMM_GO(128, 64, 32, 2);
    else if (sel == 2) GEMM_GO(64, 128, 32, 2);
    else if (sel == 3) GEMM_GO(128, 128, 32, 1);
    else GEMM_GO(64, 64, 32, 3);
#undef GEMM_GO
    if (ksplit > 1) {
        const long n4 = split_stride / 4;
        CFFM_LAUNCH(k_sum_splits, ((unsigned)((n4 + 63) / 64)), (256), 0, st, (const float*)out, ksplit, split_stride, C);
    }
    return 0;
}
// y[M,N] = x[M,K] w[N,K]^T
static int gemm_nt_split(const float* x, const float* w, float* y, long M, int N, int K, hipStream_t st) {
    return gemm_split_launch<false, false>(x, w, y, (int)M, N, K, K, K, N, 1, st);
}
// the same with the weight (and optionally the activation) in split-4 storage
template <bool X_PRE>
static int gemm_nt_split_pre(const float* x, const float* w_s, float* y, long M, int N, int K, hipStream_t st) {
    return gemm_split_launch<false, false, 0, X_PRE, true>(x, w_s, y, (int)M, N, K, K, K, N, 1, st);
}
// hraw[M,N] = x w^T (raw, kept for backward), act = gelu(hraw + b)      (fc1 of the Mlp, cffm_transformer.py:21-22)
static int gemm_nt_gelu_split(const float* x, const float* w, const float* b, float* hraw, float* act, long M, int N, int K, hipStream_t st) {
    return gemm_split_launch<false, false, 1>(x, w, hraw, (int)M, N, K, K, K, N, 1, st, b, act);
}
// x, w and the written act all in split-4 storage
static int gemm_nt_gelu_split_pre(const float* x_s, const float* w_s, const float* b, float* hraw, float* act_s, long M, int N, int K,
                                  hipStream_t st) {
    return gemm_split_launch<false, false, 5, true, true>(x_s, w_s, hraw, (int)M, N, K, K, K, N, 1, st, b, act_s);
}
// out[M,N] = res + x w^T + b                                             (fc2 + residual, cffm_transformer.py:824)
static int gemm_nt_residual_split(const float* x, const float* w, const float* b, const float* res, float* out, long M, int N, int K,
                                  hipStream_t st) {
    return gemm_split_launch<false, false, 2>(x, w, out, (int)M, N, K, K, K, N, 1, st, b, const_cast<float*>(res));
}
static int gemm_nt_residual_split_pre(const float* x_s, const float* w_s, const float* b, const float* res, float* out, long M, int N,
                                      int K, hipStream_t st) {
    return gemm_split_launch<false, false, 2, true, true>(x_s, w_s, out, (int)M, N, K, K, K, N, 1, st, b, const_cast<float*>(res));
}
// qkv16[M,768] (f16) = (x w^T + b) with the q third pre-scaled                (qkv Linear feeding the CFM kernels)
static int gemm_nt_qkv16_split(const float* x, const float* w, const float* b, h16* qkv16, long M, int N, int K, hipStream_t st) {
    return gemm_split_launch<false, false, 3>(x, w, nullptr, (int)M, N, K, K, K, N, 1, st, b, (float*)qkv16);
}
static int gemm_nt_qkv16_split_pre(const float* x_s, const float* w_s, const float* b, h16* qkv16, long M, int N, int K, hipStream_t st) {
    return gemm_split_launch<false, false, 3, true, true>(x_s, w_s, nullptr, (int)M, N, K, K, K, N, 1, st, b, (float*)qkv16);
}
// dx[M,K] = dy[M,N] w[N,K]: output cols = K, contraction = N
static int gemm_nn_split(const float* dy, const float* w, float* dx, long M, int N, int K, hipStream_t st) {
    return gemm_split_launch<false, true>(dy, w, dx, (int)M, K, N, N, K, K, 1, st);
}
// dh[M,K] = (dy[M,N] w[N,K]) * gelu'(hraw + b1), in split-4 storage or plain; part[2 * ceil(M/128) * 2][K] = column-sum
// records of dh (the bias gradient of fc1).  Always 128x128 tiles (the record layout is theirs).
#define GEMM_GELUBWD_RECORDS(M) (4 * (((M) + 127) / 128))
template <bool SPLIT_OUT>
static int gemm_nn_gelubwd_split_pre(const float* dy, const float* w_s, const float* hraw, const float* b1, float* dh, float* part,
                                     long M, int N, int K, hipStream_t st) {
    return gemm_split_launch<false, true, SPLIT_OUT ? 7 : 6, false, true>(dy, w_s, dh, (int)M, K, N, N, K, K, 1, st, b1,
                                                                          const_cast<float*>(hraw), 1, part);
}
template <bool DY_PRE>
static int gemm_nn_split_pre(const float* dy, const float* w_s, float* dx, long M, int N, int K, hipStream_t st) {
    return gemm_split_launch<false, true, 0, DY_PRE, true>(dy, w_s, dx, (int)M, K, N, N, K, K, 1, st);
}
// dw[N,K] = dy[M,N]^T x[M,K]: output N x K, contraction = M (long) split over workgroups
template <bool DY_PRE = false, bool X_PRE = false>
static int gemm_tn_split(const float* dy, const float* x, float* dw, long M, int N, int K, hipStream_t st) {
    // 128x128 tiles halve the operand re-reads through the CU load path (the bound of these kernels: see gemm_kernels.h);
    // enough contraction splits to give every CU a workgroup
    const int big = (N > 96 && K >= 128);   // (N = 124: the head's classifiers -- one ragged 128-row tile beats two 64-row ones: 84 -> ? us)
    const int tiles = big ? ((N + 127) / 128) * ((K + 127) / 128) : ((N + 63) / 64) * ((K + 63) / 64);
    int ksplit = (320 + tiles - 1) / tiles;
    const int maxsplit = (int)((M + 127) / 128);   // at least 128 rows of the contraction per split
    if (ksplit > maxsplit) ksplit = maxsplit;
    if (ksplit < 1) ksplit = 1;
    return gemm_split_launch<true, true, 0, DY_PRE, X_PRE>(dy, x, dw, N, K, (int)M, N, K, K, ksplit, st, nullptr, nullptr, big);
}

// ---- grouped weight gradients: dw_p[N_p,K_p] = dy_p[M_p,N_p]^T x_p[M_p,K_p], p < n <= 4, one launch (k_gemm_group_tt) --------
struct GemmTN { const float* dy; const float* x; float* dw; long M; int N, K; };
struct GemmTNPre { int dy_pre, x_pre; const float* x_bias; };   // operand in split-4 storage (1); x_pre == 2: x = gelu(stored + x_bias) on load
// `scratch_fn`: where the split-K partial slabs live (default: the library scratch of the caller's stream; a group issued on the
// side stream of the block backward passes the side scratch -- the two run concurrently); `target_wgs`: workgroups to aim for
// floats of split-K partial slabs a grouped launch of these problems asks its scratch for (same arithmetic as below)
static size_t gemm_tn_group_partial_floats(const GemmTN* pr, int n, int target_wgs) {
    long units = 0;
    for (int p = 0; p < n; ++p) {
        if (pr[p].N % 128 || pr[p].K % 128 || pr[p].M < 32) return 0;
        units += (long)(pr[p].N / 128) * (pr[p].K / 128) * ((pr[p].M + 31) / 32);
    }
    const char* e = cffm_tune("CFFM_GROUP_WGS");
    const int env = e ? atoi(e) : 0, target = env > 0 ? env : (target_wgs > 0 ? target_wgs : 480);
    long ksteps = (units + target - 1) / target;
    if (ksteps < 4) ksteps = 4;
    const long klen = ksteps * 32;
    size_t part = 0;
    for (int p = 0; p < n; ++p) {
        const long ks = (pr[p].M + klen - 1) / klen;
        if (ks > 1) part += (size_t)ks * pr[p].N * pr[p].K;
    }
    return part;
}
// `after_gemm`: called with the stream right behind the GEMM launch(es), before the partial-slab sum (the block backward records
// an event there: from that point on the operands may be overwritten)
static int gemm_tn_group_split(const GemmTN* pr, int n, hipStream_t st, const GemmTNPre* pre = nullptr,
                               float* (*scratch_fn)(size_t) = lib_scratch, int target_wgs = 0, void (*after_gemm)(hipStream_t) = nullptr) {
    bool groupable = n >= 1 && n <= GEMM_GROUP_MAX;
    for (int p = 0; p < n && groupable; ++p) groupable = pr[p].N % 128 == 0 && pr[p].K % 128 == 0 && pr[p].M >= 32;
    if (!groupable) {
        for (int p = 0; p < n; ++p)
            if (pre && pre[p].x_pre == 2) return -1;   // the on-load GELU exists in the grouped kernel only
        for (int p = 0; p < n; ++p) {
            const int a = pre ? pre[p].dy_pre : 0, b = pre ? pre[p].x_pre : 0;
            int rc;
            if (a && b) rc = gemm_tn_split<true, true>(pr[p].dy, pr[p].x, pr[p].dw, pr[p].M, pr[p].N, pr[p].K, st);
            else if (a) rc = gemm_tn_split<true, false>(pr[p].dy, pr[p].x, pr[p].dw, pr[p].M, pr[p].N, pr[p].K, st);
            else if (b) rc = gemm_tn_split<false, true>(pr[p].dy, pr[p].x, pr[p].dw, pr[p].M, pr[p].N, pr[p].K, st);
            else rc = gemm_tn_split<false, false>(pr[p].dy, pr[p].x, pr[p].dw, pr[p].M, pr[p].N, pr[p].K, st);
            if (rc) return -1;
        }
        if (after_gemm) after_gemm(st);
        return 0;
    }
    // one slice length for every problem: ~480 workgroups (two per CU are co-resident: 80 KB of LDS each)
    long units = 0;
    for (int p = 0; p < n; ++p) units += (long)(pr[p].N / 128) * (pr[p].K / 128) * ((pr[p].M + 31) / 32);
    static int target_env = -1;   // tuning aid: CFFM_GROUP_WGS
    if (target_env < 0) { const char* e = cffm_tune("CFFM_GROUP_WGS"); target_env = e ? atoi(e) : 0; if (target_env < 0) target_env = 0; }
    const int target = target_env ? target_env : (target_wgs > 0 ? target_wgs : 480);
    long ksteps = (units + target - 1) / target;
    if (ksteps < 4) ksteps = 4;
    const int klen = (int)ksteps * 32;
    GemmGroup G;
    SumGroup Sg;
    size_t part_floats = 0;
    int ksplit[GEMM_GROUP_MAX];
    for (int p = 0; p < n; ++p) {
        ksplit[p] = (int)((pr[p].M + klen - 1) / klen);
        if (ksplit[p] > 1) part_floats += (size_t)ksplit[p] * pr[p].N * pr[p].K;
    }
    float* part = part_floats ? scratch_fn(part_floats) : nullptr;
    if (part_floats && !part) return -1;
    int wg = 0, blk = 0, nsum = 0;
    for (int p = 0; p < n; ++p) {
        G.A[p] = pr[p].dy; G.B[p] = pr[p].x;
        G.a_pre[p] = pre ? pre[p].dy_pre : 0; G.b_pre[p] = pre ? pre[p].x_pre : 0; G.b_bias[p] = pre ? pre[p].x_bias : nullptr;
        G.M[p] = pr[p].N; G.N[p] = pr[p].K; G.K[p] = (int)pr[p].M;
        G.C[p] = pr[p].dw;
        if (ksplit[p] > 1) {
            G.C[p] = part;
            Sg.part[nsum] = part; Sg.out[nsum] = pr[p].dw; Sg.n[nsum] = (long)pr[p].N * pr[p].K; Sg.nsplit[nsum] = ksplit[p];
            blk += (int)((Sg.n[nsum] / 4 + 255) / 256);
            Sg.blk_end[nsum] = blk;
            ++nsum;
            part += (size_t)ksplit[p] * pr[p].N * pr[p].K;
        }
        wg += (pr[p].N / 128) * (pr[p].K / 128) * ksplit[p];
        G.wg_end[p] = wg;
    }
    for (int p = n; p < GEMM_GROUP_MAX; ++p) { G.A[p] = G.B[p] = nullptr; G.C[p] = nullptr; G.M[p] = G.N[p] = 128; G.K[p] = 0; G.wg_end[p] = wg; G.a_pre[p] = G.b_pre[p] = 0; G.b_bias[p] = nullptr; }
    G.klen = klen; G.n = n;
#ifndef CFFM_EMU
    static bool granted = false;
    if (!granted) {
        if (hipFuncSetAttribute((const void*)k_gemm_group_tt, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS(128, 128, 32) + 16384) != hipSuccess)
            return -1;
        granted = true;
    }
#endif
    // 8 KB more LDS than the tiles need: ONE workgroup of the group per CU instead of two.  Two fill a CU's LDS exactly, and the CFFA
    // backward that runs beside the group in a block backward (69 KB per workgroup) then finds no slot until the group's workgroups
    // finish; with one per CU the two kernels really share the CUs: the group 62 -> 70 us, k_ln_pool_bwd 85 -> 77, step 0.717-0.730 ->
    // 0.713-0.722 ms (same box, three alternating runs; 256 / 384 / 512 workgroups instead of 480: slower).  CFFM_DW_LDS_PAD=0: two per CU.
    static int lds_pad = -1;
    if (lds_pad < 0) { const char* e = cffm_tune("CFFM_DW_LDS_PAD"); lds_pad = e ? atoi(e) : 8192; if (lds_pad < 0 || lds_pad > 16384) lds_pad = 8192; }
    CFFM_LAUNCH(k_gemm_group_tt, ((unsigned)wg), (256), GEMM_LDS(128, 128, 32) + lds_pad, st, G);
    if (after_gemm) after_gemm(st);
    if (nsum) {
        Sg.cnt = nsum;
        for (int q = nsum; q < 4; ++q) { Sg.part[q] = nullptr; Sg.out[q] = nullptr; Sg.n[q] = 0; Sg.nsplit[q] = 0; Sg.blk_end[q] = blk; }
        CFFM_LAUNCH(k_sum_splits_group, ((unsigned)blk), (256), 0, st, Sg);
    }
    return 0;
}

// ---- weight gradients with both operands in split-4 storage: the LDS-DMA kernel (dw_kernels.h), up to DWD_MAX problems per launch ----
static size_t dw_dma_partial_floats(const GemmTN* pr, int n, int target_wgs, int* klen_out) {
    long units = 0;
    for (int p = 0; p < n; ++p) units += (long)(pr[p].N / 128) * (pr[p].K / 128) * ((pr[p].M + 31) / 32);
    long ksteps = (units + target_wgs - 1) / target_wgs;
    if (ksteps < 4) ksteps = 4;
    const long klen = ksteps * 32;
    size_t part = 0;
    for (int p = 0; p < n; ++p) {
        const long ks = (pr[p].M + klen - 1) / klen;
        if (ks > 1) part += (size_t)ks * pr[p].N * pr[p].K;
    }
    *klen_out = (int)klen;
    return part;
}
// `part`: dw_dma_partial_floats(...) floats of slab scratch (or NULL when that is 0)
static int dw_group_dma(const GemmTN* pr, int n, hipStream_t st, float* part, int target_wgs, void (*after_gemm)(hipStream_t) = nullptr) {
    if (n < 1 || n > DWD_MAX) return -1;
    for (int p = 0; p < n; ++p)
        if (pr[p].N % 128 || pr[p].K % 128 || pr[p].M < 1 || (long)pr[p].M * pr[p].N * 4 >= (1L << 32) || (long)pr[p].M * pr[p].K * 4 >= (1L << 32)) return -1;
    int klen;
    const size_t need = dw_dma_partial_floats(pr, n, target_wgs, &klen);
    if (need && !part) return -1;
    DwGroup G;
    SumGroup Sg;
    int wg = 0, blk = 0, nsum = 0;
    for (int p = 0; p < n; ++p) {
        const int ks = (int)((pr[p].M + klen - 1) / klen);
        G.A[p] = pr[p].dy; G.B[p] = pr[p].x; G.N[p] = pr[p].N; G.K[p] = pr[p].K; G.M[p] = (int)pr[p].M;
        G.C[p] = pr[p].dw;
        if (ks > 1) {
            G.C[p] = part;
            Sg.part[nsum] = part; Sg.out[nsum] = pr[p].dw; Sg.n[nsum] = (long)pr[p].N * pr[p].K; Sg.nsplit[nsum] = ks;
            blk += (int)((Sg.n[nsum] / 4 + 255) / 256);
            Sg.blk_end[nsum] = blk;
            ++nsum;
            part += (size_t)ks * pr[p].N * pr[p].K;
        }
        wg += (pr[p].N / 128) * (pr[p].K / 128) * ks;
        G.wg_end[p] = wg;
    }
    for (int p = n; p < DWD_MAX; ++p) { G.A[p] = G.B[p] = nullptr; G.C[p] = nullptr; G.N[p] = G.K[p] = 128; G.M[p] = 0; G.wg_end[p] = wg; }
    G.klen = klen; G.n = n;
#ifndef CFFM_EMU
    static bool granted = false;
    if (!granted) {
        if (hipFuncSetAttribute((const void*)k_dw_dma, hipFuncAttributeMaxDynamicSharedMemorySize, DWD_LDS) != hipSuccess) return -1;
        granted = true;
    }
#endif
    CFFM_LAUNCH(k_dw_dma, ((unsigned)wg), (256), DWD_LDS, st, G);
    if (after_gemm) after_gemm(st);
    if (nsum) {
        Sg.cnt = nsum;
        for (int q = nsum; q < 4; ++q) { Sg.part[q] = nullptr; Sg.out[q] = nullptr; Sg.n[q] = 0; Sg.nsplit[q] = 0; Sg.blk_end[q] = blk; }
        CFFM_LAUNCH(k_sum_splits_group, ((unsigned)blk), (256), 0, st, Sg);
    }
    return 0;
}

// ---- weight gradients with both operands in T-frag storage: the streaming kernel (dws_kernels.h), up to DWS_MAX problems per launch ----
// The plan of a group: one number of k-steps per wave for all problems such that the launch has at most `target_wgs` workgroups (one per
// CU: a workgroup holds ~350 registers per lane), then evened out inside each problem.
struct DwsPlan { int S[DWS_MAX], kw[DWS_MAX], KS[DWS_MAX], tiles[DWS_MAX]; int wgs; size_t part_floats; };
static bool dw_stream_plan(const GemmTN* pr, int n, int target_wgs, DwsPlan* P) {
    if (n < 1 || n > DWS_MAX) return false;
    long units = 0;
    for (int p = 0; p < n; ++p) {
        if (pr[p].N % DWS_TO || pr[p].K % DWS_TI || pr[p].N < DWS_TO || pr[p].K < DWS_TI || pr[p].M < 1) return false;
        P->KS[p] = (int)((pr[p].M + 31) / 32);
        P->tiles[p] = (pr[p].N / DWS_TO) * (pr[p].K / DWS_TI);
        if ((long)P->KS[p] * 32 * pr[p].N * 4 >= (1L << 32) || (long)P->KS[p] * 32 * pr[p].K * 4 >= (1L << 32)) return false;
        units += (long)P->tiles[p] * P->KS[p];
    }
    if (target_wgs < 1) target_wgs = 256;
    long kw = (units + 4L * target_wgs - 1) / (4L * target_wgs);
    if (kw < 1) kw = 1;
    for (;; ++kw) {
        long wgs = 0;
        for (int p = 0; p < n; ++p) wgs += (long)P->tiles[p] * ((P->KS[p] + 4 * kw - 1) / (4 * kw));
        bool single = true;
        for (int p = 0; p < n; ++p) single = single && P->KS[p] <= 4 * kw;
        if (wgs <= target_wgs || single) break;
    }
    P->wgs = 0;
    P->part_floats = 0;
    for (int p = 0; p < n; ++p) {
        P->S[p] = (int)((P->KS[p] + 4 * kw - 1) / (4 * kw));
        P->kw[p] = (P->KS[p] + 4 * P->S[p] - 1) / (4 * P->S[p]);
        P->wgs += P->tiles[p] * P->S[p];
        if (P->S[p] > 1) P->part_floats += (size_t)P->S[p] * pr[p].N * pr[p].K;
    }
    return true;
}
static int dw_stream_target() {      // tuning aid: CFFM_DWS_WGS
    static int v = -1;
    if (v < 0) { const char* e = cffm_tune("CFFM_DWS_WGS"); v = e ? atoi(e) : 0; if (v < 1) v = 256; }
    return v;
}
// pr[p].dy / .x: T-frag storage of dy [M][N] / x [M][K] (rows past M zero); `part`: plan.part_floats floats of slab scratch (or NULL when 0)
static int dw_group_stream(const GemmTN* pr, int n, hipStream_t st, float* part, int target_wgs, void (*after_gemm)(hipStream_t) = nullptr) {
    DwsPlan P;
    if (!dw_stream_plan(pr, n, target_wgs, &P)) return -1;
    if (P.part_floats && !part) return -1;
    DwsGroup G;
    SumGroup Sg;
    int wg = 0, blk = 0, nsum = 0;
    for (int p = 0; p < n; ++p) {
        G.DY[p] = (const f32x4*)pr[p].dy; G.X[p] = (const f32x4*)pr[p].x; G.N[p] = pr[p].N; G.K[p] = pr[p].K; G.KS[p] = P.KS[p]; G.kw[p] = P.kw[p];
        G.C[p] = pr[p].dw;
        if (P.S[p] > 1) {
            G.C[p] = part;
            Sg.part[nsum] = part; Sg.out[nsum] = pr[p].dw; Sg.n[nsum] = (long)pr[p].N * pr[p].K; Sg.nsplit[nsum] = P.S[p];
            blk += (int)((Sg.n[nsum] / 4 + 255) / 256);
            Sg.blk_end[nsum] = blk;
            ++nsum;
            part += (size_t)P.S[p] * pr[p].N * pr[p].K;
        }
        wg += P.tiles[p] * P.S[p];
        G.wg_end[p] = wg;
    }
    for (int p = n; p < DWS_MAX; ++p) { G.DY[p] = G.X[p] = nullptr; G.C[p] = nullptr; G.N[p] = DWS_TO; G.K[p] = DWS_TI; G.KS[p] = 0; G.kw[p] = 1; G.wg_end[p] = wg; }
    G.n = n;
#ifndef CFFM_EMU
    static bool granted = false;
    if (!granted) {
        if (hipFuncSetAttribute((const void*)k_dw_stream, hipFuncAttributeMaxDynamicSharedMemorySize, DWS_LDS) != hipSuccess) return -1;
        granted = true;
    }
#endif
    CFFM_LAUNCH(k_dw_stream, ((unsigned)wg), (256), DWS_LDS, st, G);
    if (after_gemm) after_gemm(st);
    if (nsum) {
        Sg.cnt = nsum;
        for (int q = nsum; q < 4; ++q) { Sg.part[q] = nullptr; Sg.out[q] = nullptr; Sg.n[q] = 0; Sg.nsplit[q] = 0; Sg.blk_end[q] = blk; }
        CFFM_LAUNCH(k_sum_splits_group, ((unsigned)blk), (256), 0, st, Sg);
    }
    return 0;
}

// ---- the three products of a Linear layer ------------------------------------------------------------------------------
static int gemm_nt(const float* x, const float* w, float* y, long M, int N, int K, hipStream_t st) {
    return gemm_nt_split(x, w, y, M, N, K, st);
}
static int gemm_nn(const float* dy, const float* w, float* dx, long M, int N, int K, hipStream_t st) {
    return gemm_nn_split(dy, w, dx, M, N, K, st);
}
static int gemm_tn(const float* dy, const float* x, float* dw, long M, int N, int K, hipStream_t st) {
    return gemm_tn_split<>(dy, x, dw, M, N, K, st);
}
static int gemm_tn_group(const GemmTN* pr, int n, hipStream_t st, const GemmTNPre* pre = nullptr, float* (*scratch_fn)(size_t) = lib_scratch,
                         int target_wgs = 0, void (*after_gemm)(hipStream_t) = nullptr) {
    return gemm_tn_group_split(pr, n, st, pre, scratch_fn, target_wgs, after_gemm);
}
