// dw_kernels.h -- weight gradients dw[N][K] = dy[M][N]^T x[M][K] (contraction over ~10^4 token rows) with BOTH operands in split-4
// storage in HBM, staged by LDS-DMA (round 5).
//
// Why a second weight-gradient kernel.  k_gemm_group_tt (gemm_kernels.h) stages its operand tiles global -> registers -> bf16 split
// (VALU) -> ds_write -> barrier -> transposed fragment reads -> MFMAs, one wave per SIMD.  Ablations on the block's group
// (scripts/r05_dw_ablate.sh, profiles/r05_dw_ablate.txt: 63.8 us with everything; 52.7 without the K-loop's global loads; 42.2
// without split + LDS stores; 56.9 without the MFMAs) say the chain load -> wait -> split -> store is what the group waits for, not
// the matrix pipe.  Here nothing of that chain runs on the wave:
//  * every operand is ALREADY hi/lo bf16 in memory (split-4 storage: 16 bytes = {bf16 hi x4 | bf16 lo x4} of four consecutive
//    floats of a row; zall, z2, act, dh have been stored that way since round 2, dqkv and a copy of dout since round 5), so a tile
//    needs no arithmetic on its way in;
//  * tiles travel global -> LDS by `buffer_load_dwordx4 ... lds` (dma_ld16): no staging registers, no ds_write instructions, no
//    vmcnt wait in front of VALU work; DWD_NS stages are in flight (three K-tiles ahead of the multiplication);
//  * the LDS image of a tile is the memory image: [32 contraction rows][128 columns x 4 B] = 512 B per row, hi and lo quads
//    interleaved.  The transposed fragment read (ds_read_b64_tr_b16) only needs each lane's four bf16 to be contiguous, which a
//    split-4 quad is: the hi fragment of 16-column tile j reads the 8-byte pieces at row * 512 + 16 * chunk, the lo fragment 8 bytes
//    further.  The DMA writes lane-linearly, so the bank swizzle is applied to the SOURCE: slot sp of row r holds chunk
//    sp ^ (4 (r & 3)) -- the four rows a 16-lane group reads together then fall on four different 64-byte bank groups.
// One barrier per K-tile; waits are counted (vmcnt(8 x stages in flight behind the one being multiplied)).
// Partial tiles of the k-slices go to slabs as in the grouped kernel (k_sum_splits_group adds them): deterministic.
#pragma once
#include "cfm_attn_kernels.h"   // wait_vm
#include "gemm_kernels.h"

#define DWD_NS 4                         // LDS stages (K-tiles in flight + the one being multiplied)
#define DWD_OPB (32 * 512)               // bytes of one operand tile: 32 rows x 128 columns, split-4
#define DWD_STAGE (2 * DWD_OPB)
#define DWD_LDS (DWD_NS * DWD_STAGE)     // 128 KB (the epilogue's transposition tiles reuse it)
#define DWD_MAX 3
struct DwGroup {
    const float* A[DWD_MAX];             // dy [M][N], split-4
    const float* B[DWD_MAX];             // x  [M][K], split-4
    float* C[DWD_MAX];                   // partial outputs [ksplit][N][K] (or dw itself when ksplit == 1)
    int N[DWD_MAX], K[DWD_MAX], M[DWD_MAX];
    int wg_end[DWD_MAX];                 // exclusive prefix of workgroups per problem (after XCD re-numbering)
    int klen, n;                         // contraction rows per k-slice (a multiple of 32), problems
};

// workgroup barrier that orders LDS traffic only (a __syncthreads() also waits vmcnt(0): it would drain the DMA pipeline every K-tile)
__device__ __forceinline__ void dwd_barrier() {
#ifdef CFFM_EMU
    __syncthreads();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#endif
}
// byte offset inside an operand tile of the 8-byte piece {hi | lo} x (k-row, 4-column chunk)
__device__ __forceinline__ int dwd_piece(int krow, int chunk, int lo) { return krow * 512 + ((chunk ^ ((krow & 3) << 2)) << 4) + (lo << 3); }
// MFMA fragment (8 contraction slots of column `col0 + (lane & 15)`) out of an operand tile: two transposed reads
__device__ __forceinline__ bf16x8 dwd_frag(const char* tile, int col0, int lo, int l15, int g) {
    const int krow = 8 * g + (l15 >> 2), chunk = (col0 >> 2) + (l15 & 3);
    const bf16x4 a = lds_tr4_bf16((const bf16*)(tile + dwd_piece(krow, chunk, lo)));
    const bf16x4 b = lds_tr4_bf16((const bf16*)(tile + dwd_piece(krow + 4, chunk, lo)));
    bf16x8 f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[e] = a[e]; f[4 + e] = b[e]; }
    return f;
}

__global__ void __launch_bounds__(256) k_dw_dma(DwGroup G) {
    CFFM_DYN_SMEM(smem);
    int lin = xcd_linear_id(), p = 0;
#pragma unroll
    for (int q = 0; q < DWD_MAX - 1; ++q)
        if (q + 1 < G.n && lin >= G.wg_end[q]) p = q + 1;
    if (p > 0) lin -= G.wg_end[p - 1];
    const int N = G.N[p], K = G.K[p], M = G.M[p], ntk = K / 128, ntn = N / 128;
    const int bx = lin % ntk, by = (lin / ntk) % ntn, bz = lin / (ntk * ntn);
    const int n0 = by * 128, k0 = bx * 128;
    const int mbeg = bz * G.klen, mend = (mbeg + G.klen < M) ? mbeg + G.klen : M;
    const int NKT = (mend - mbeg + 31) / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), l15 = lane & 15, g = lane >> 4;
    // ---- DMA addressing: wave w, instruction q (0..3) fills slots [(4 w + q) 64, + 64) of an operand tile: k-rows 8 w + 2 q, + 1
    const dma_t ra = dma_make(G.A[p], (uint32_t)((long)M * N * 4)), rb = dma_make(G.B[p], (uint32_t)((long)M * K * 4));
    uint32_t va[4], vb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int slot = (4 * wave + q) * 64 + lane, r = slot >> 5, c = (slot & 31) ^ ((r & 3) << 2);
        va[q] = (uint32_t)((r * N + 4 * c) * 4);
        vb[q] = (uint32_t)((r * K + 4 * c) * 4);
    }
    const uint32_t sa0 = (uint32_t)(((long)mbeg * N + n0) * 4), sb0 = (uint32_t)(((long)mbeg * K + k0) * 4);
    const uint32_t dsa = (uint32_t)(32 * N * 4), dsb = (uint32_t)(32 * K * 4);
    auto issue = [&](int t) {
        char* st = smem + (t % DWD_NS) * DWD_STAGE + 4 * wave * 1024;
#pragma unroll
        for (int q = 0; q < 4; ++q) dma_ld16(ra, va[q], sa0 + t * dsa, st + q * 1024);
#pragma unroll
        for (int q = 0; q < 4; ++q) dma_ld16(rb, vb[q], sb0 + t * dsb, st + DWD_OPB + q * 1024);
    };
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < DWD_NS - 1; ++t)
        if (t < NKT) issue(t);
    for (int t = 0; t < NKT; ++t) {
        // stage t has landed when at most the DMAs of the stages issued behind it (8 per thread and stage) are outstanding
        if (t + 2 < NKT) wait_vm<16>();
        else if (t + 1 < NKT) wait_vm<8>();
        else wait_vm0();
        dwd_barrier();            // every wave's share of stage t is in LDS; every wave is done reading stage t - 1
        if (t + DWD_NS - 1 < NKT) issue(t + DWD_NS - 1);       // ... whose buffer the new stage overwrites
        const char* ta = smem + (t % DWD_NS) * DWD_STAGE;
        const char* tb = ta + DWD_OPB;
        bf16x8 bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bh[j] = dwd_frag(tb, wc + 16 * j, 0, l15, g);
            bl[j] = dwd_frag(tb, wc + 16 * j, 1, l15, g);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16x8 ah = dwd_frag(ta, wr + 16 * i, 0, l15, g), al = dwd_frag(ta, wr + 16 * i, 1, l15, g);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = mfma16x16x32_bf16(ah, bl[j], acc[i][j]);
                acc[i][j] = mfma16x16x32_bf16(al, bh[j], acc[i][j]);
                acc[i][j] = mfma16x16x32_bf16(ah, bh[j], acc[i][j]);
            }
        }
    }
    dwd_barrier();                // the operand stages are dead: the transposition tiles below reuse them
    // epilogue as in gemm_tile: acc[i][j][r] = C[n0 + wr + 16 i + 4 g + r][k0 + wc + 16 j + l15]; each wave transposes 32 rows x 64 columns
    // at a time through its own LDS slice and writes whole 256-byte row segments
    float* T = (float*)smem + wave * (32 * GEMM_TLD);
    float* C = G.C[p] + (long)bz * N * K;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        wave_lds_sync();
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) T[(16 * ii + 4 * g + r) * GEMM_TLD + 16 * j + l15] = acc[2 * half + ii][j][r];
        wave_lds_sync();
        const int c4 = 4 * (lane & 15);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int rl = (lane >> 4) + 4 * it;
            *(f32x4*)(C + (long)(n0 + wr + 32 * half + rl) * K + k0 + wc + c4) = *(const f32x4*)(T + rl * GEMM_TLD + c4);
        }
    }
}
