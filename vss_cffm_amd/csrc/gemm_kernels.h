// gemm_kernels.h -- the dense Linear GEMMs of the block (qkv, proj, fc1, fc2; forward, input-gradient and
// weight-gradient forms) on the bf16 matrix cores at (near-)fp32 accuracy.
//
// gfx950 has no TF32; its fp32 MFMA runs at the vector rate (157 TF), 1/16 of the bf16 rate.  Each fp32
// operand x is split on the fly, while it is staged into LDS, into two bf16 numbers hi = bf16(x),
// lo = bf16(x - hi) (16 mantissa bits together, full fp32 exponent range -- gradients of 1e-7 are safe,
// unlike with f16), and a product is three MFMAs: hi*hi + hi*lo + lo*hi, accumulated in fp32.  The dropped
// lo*lo term and the lo rounding are ~2^-17 relative per product, two orders below the f16 rounding of the
// attention operands and three below the 1e-3 contract; the effective peak is 2.5 PF / 3.
//
// One kernel template covers the three forms through the storage order of each operand:
//   C[M,N] (+)= sum_k A(i,k) B(j,k),   A(i,k) = A_T ? A[k*lda+i] : A[i*lda+k],   B likewise
//   forward        y  = x  w^T : A = x  (i=row, k contiguous),  B = w  [N,K]  (k contiguous)        <F,F>
//   input gradient dx = dy w   : A = dy (k = n contiguous),     B(j=k',k=n) = w[n*K+k'] -> B_T       <F,T>
//   weight gradient dw = dy^T x: A(i=n,k=m) = dy[m*N+n] -> A_T, B(j=k',k=m) = x[m*K+k'] -> B_T       <T,T>
// What bounds it (measured, scripts/gemm_bench.py): with fp32 operands a BM x BN tile moves (BM+BN)*4 bytes per
// 2*BM*BN flops through the CU's vector-load path, i.e. BM*BN/(2(BM+BN)) flop/B: 16 for 64x64, 32 for 128x128; at the
// ~6.8 TB/s the 256 CUs pull from L2 / Infinity Cache that is ~110 / ~220 TF -- what the kernels reach (100 / 170 TF;
// rocBLAS SGEMM: 60-100 TF).  K-tile depth (BK 32 vs 64) and prefetch depth are neutral; the small-N layers do not
// have enough output to give 256 CUs 128x128 tiles, hence their lower rate.
// Tiles: BM x BN x 32 (BM, BN in {64,128})
// per 256-thread workgroup (2x2 waves, each (BM/2) x (BN/2) as 16x16x32 MFMA tiles), LDS rows
// padded to 80 B (conflict-free ds_read_b128 fragments), the next PF K-tiles in flight in registers while the
// current one is multiplied.  The weight-gradient form splits its long contraction (M ~ 10^4) over
// gridDim.z: every split writes its own partial output (plain coalesced stores) and a second tiny kernel sums
// them -- deterministic, and row-coalesced fp32 atomics measured 3-4x slower than this on gfx950.
#pragma once
#include "cffm_common.h"

typedef __bf16 bf16;
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma16x16x32_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
#ifdef CFFM_EMU
    struct Frag { float a[8], b[8]; } mine;
    for (int j = 0; j < 8; ++j) { mine.a[j] = (float)a[j]; mine.b[j] = (float)b[j]; }
    int lane = emu::lane_linear() & 63;
    auto s = emu::deposit(&mine, sizeof(mine));
    int col = lane & 15, g = lane >> 4;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * g + r;
        float acc = c[r];
        for (int kg = 0; kg < 4; ++kg) {
            const Frag* fa = reinterpret_cast<const Frag*>(s[row + 16 * kg]);
            const Frag* fb = reinterpret_cast<const Frag*>(s[col + 16 * kg]);
            for (int j = 0; j < 8; ++j) acc += fa->a[j] * fb->b[j];
        }
        c[r] = acc;
    }
    emu::release();
    return c;
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}

__device__ __forceinline__ void split4(f32x4 x, bf16x4& hi, bf16x4& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi[e] = (bf16)x[e];
        lo[e] = (bf16)(x[e] - (float)hi[e]);
    }
}

#define GEMM_LD(BK) ((BK) + 8)  // bf16 per LDS row (BK + 8 pad: 80 / 144 B strides keep ds_read_b128 fragments conflict-free)
#define GEMM_TLD 68 // floats per row of the epilogue transposition tile (64 + 4 pad)

// LDS image of an operand tile (ROWS x 32 k), hi and lo parts:
//   k-contiguous operand (TR = false): [ROWS][BK + 8] bf16, fragment = one ds_read_b128 of 8 consecutive k;
//   row-contiguous operand (TR = true): k-PAIR interleaved dwords [BK/2 k-pairs][ROWS + 4]: a thread that loaded the same
//     4 rows at k and k+1 (two coalesced 16-B loads) packs (k, k+1) per row into one dword and writes 16 B at once;
//     a fragment is 4 x ds_read_b32 (k = 8g + 2jj + {0,1}), 16 lanes reading 16 consecutive dwords (conflict-free,
//     the +4 pad puts the two lane groups of a half-wave on disjoint banks).
#define GEMM_TS(ROWS) ((ROWS) + 4)
template <int ROWS, int BK>
struct TileRegs { f32x4 v[ROWS * BK / 1024]; };

// global -> registers
template <int ROWS, bool TR, int BK>
__device__ __forceinline__ void tile_load(TileRegs<ROWS, BK>& r, const float* __restrict__ P, int ld, int row0, int nrows,
                                          int k0, int kend, int tid) {
    const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!TR) {
#pragma unroll
        for (int it = 0; it < ROWS * BK / 1024; ++it) {
            const int item = tid + 256 * it;
            const int row = row0 + item / (BK / 4), k = k0 + 4 * (item % (BK / 4));
            r.v[it] = (row < nrows && k < kend) ? *(const f32x4*)(P + (long)row * ld + k) : z;   // kend, k multiples of 4
        }
    } else {
#pragma unroll
        for (int it = 0; it < ROWS * BK / 2048; ++it) {
            const int item = tid + 256 * it;
            const int kp = item / (ROWS / 4), row = row0 + 4 * (item % (ROWS / 4));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = k0 + 2 * kp + h;
                f32x4 v = z;
                if (k < kend) {
                    if (row + 3 < nrows) v = *(const f32x4*)(P + (long)k * ld + row);
                    else
                        for (int e = 0; e < 4; ++e)
                            if (row + e < nrows) v[e] = P[(long)k * ld + row + e];
                }
                r.v[2 * it + h] = v;
            }
        }
    }
}

__device__ __forceinline__ uint32_t pack_bf16(bf16 a, bf16 b) {
    unsigned short ua, ub;
    __builtin_memcpy(&ua, &a, 2);
    __builtin_memcpy(&ub, &b, 2);
    return (uint32_t)ua | ((uint32_t)ub << 16);
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// registers -> LDS (split into hi / lo bf16 images)
template <int ROWS, bool TR, int BK>
__device__ __forceinline__ void tile_store(const TileRegs<ROWS, BK>& r, bf16* __restrict__ hi, bf16* __restrict__ lo, int tid) {
    if (!TR) {
#pragma unroll
        for (int it = 0; it < ROWS * BK / 1024; ++it) {
            const int item = tid + 256 * it;
            bf16x4 h, l;
            split4(r.v[it], h, l);
            const int row = item / (BK / 4), kc = 4 * (item % (BK / 4));
            *(bf16x4*)(hi + row * GEMM_LD(BK) + kc) = h;
            *(bf16x4*)(lo + row * GEMM_LD(BK) + kc) = l;
        }
    } else {
#pragma unroll
        for (int it = 0; it < ROWS * BK / 2048; ++it) {
            const int item = tid + 256 * it;
            const int kp = item / (ROWS / 4), row = 4 * (item % (ROWS / 4));
            bf16x4 h0, l0, h1, l1;
            split4(r.v[2 * it], h0, l0);
            split4(r.v[2 * it + 1], h1, l1);
            u32x4 ph, pl;
#pragma unroll
            for (int e = 0; e < 4; ++e) { ph[e] = pack_bf16(h0[e], h1[e]); pl[e] = pack_bf16(l0[e], l1[e]); }
            *(u32x4*)((uint32_t*)hi + kp * GEMM_TS(ROWS) + row) = ph;
            *(u32x4*)((uint32_t*)lo + kp * GEMM_TS(ROWS) + row) = pl;
        }
    }
}

// one MFMA operand fragment (8 k-slots of row `row`) out of an LDS image
template <int ROWS, bool TR, int BK>
__device__ __forceinline__ bf16x8 frag_read(const bf16* __restrict__ img, int row, int g, int kk /* 0 or 32 */) {
    if (!TR) return *(const bf16x8*)(img + row * GEMM_LD(BK) + kk + 8 * g);
    u32x4 d;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) d[jj] = ((const uint32_t*)img)[(kk / 2 + 4 * g + jj) * GEMM_TS(ROWS) + row];
    bf16x8 f;
    __builtin_memcpy(&f, &d, 16);
    return f;
}

#define GEMM_LDS(BM, BN, BK) (((2 * (BM) + 2 * (BN)) * GEMM_LD(BK) * 2) > (4 * 32 * GEMM_TLD * 4) ? ((2 * (BM) + 2 * (BN)) * GEMM_LD(BK) * 2) : (4 * 32 * GEMM_TLD * 4))

// grid (ceil(N/BN), ceil(M/BM), ksplit); BM, BN in {64, 128}; BK in {32, 64}; wave tile (BM/2) x (BN/2).
// K range of split z: [z*klen, min(K, (z+1)*klen)), klen a multiple of BK.
// EPI: 0 plain (+bias) | 1 GELU: C = raw product (saved for backward), aux = gelu(raw + bias) | 2 residual: C = aux + raw + bias
//      3 q|k|v: nothing in C; aux (h16 [M,N]) = f16(raw + bias), the q third (cols < 256) also times 32^-0.5 -- exactly the
//        values the attention kernels used to form from the fp32 product, stored once at half the bytes
template <int BM, int BN, int BK, bool A_T, bool B_T, int EPI, int PF>
__global__ void __launch_bounds__(256) k_gemm_split(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                     int M, int N, int K, int lda, int ldb, int ldc, int klen, long split_stride,
                                                     const float* __restrict__ bias, float* __restrict__ aux) {
    CFFM_DYN_SMEM(smem);
    bf16* Ah = (bf16*)smem;
    bf16* Al = Ah + BM * GEMM_LD(BK);
    bf16* Bh = Al + BM * GEMM_LD(BK);
    bf16* Bl = Bh + BN * GEMM_LD(BK);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int kbeg = blockIdx.z * klen, kend = (kbeg + klen < K) ? kbeg + klen : K;
    constexpr int MT = BM / 32, NT = BN / 32;  // 16x16 tiles per wave along M and N (wave tile = BM/2 x BN/2)
    const int wr = (wave >> 1) * (BM / 2), wc = (wave & 1) * (BN / 2);
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // PF K-tiles are always in flight in registers: a K-step (~0.1-0.4 us of MFMA) is far shorter than the ~1-2 us a
    // tile takes to arrive from HBM / Infinity Cache, and co-resident workgroups run in lockstep, so they cannot hide
    // each other's waits; tile t+PF is requested before tile t is multiplied.
    TileRegs<BM, BK> ra[PF];
    TileRegs<BN, BK> rb[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {   // past kend -> zeros, never used
        tile_load<BM, A_T, BK>(ra[u], A, lda, m0, M, kbeg + u * BK, kend, tid);
        tile_load<BN, B_T, BK>(rb[u], B, ldb, n0, N, kbeg + u * BK, kend, tid);
    }
    for (int k0 = kbeg; k0 < kend; k0 += PF * BK) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int kk = k0 + u * BK;
            if (kk < kend) {   // uniform across the workgroup
                tile_store<BM, A_T, BK>(ra[u], Ah, Al, tid);
                tile_store<BN, B_T, BK>(rb[u], Bh, Bl, tid);
                __syncthreads();
                if (kk + PF * BK < kend) {
                    tile_load<BM, A_T, BK>(ra[u], A, lda, m0, M, kk + PF * BK, kend, tid);
                    tile_load<BN, B_T, BK>(rb[u], B, ldb, n0, N, kk + PF * BK, kend, tid);
                }
#pragma unroll
                for (int ks = 0; ks < BK; ks += 32) {
                    bf16x8 bh[NT], bl[NT];
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        bh[j] = frag_read<BN, B_T, BK>(Bh, wc + 16 * j + l15, g, ks);
                        bl[j] = frag_read<BN, B_T, BK>(Bl, wc + 16 * j + l15, g, ks);
                    }
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const bf16x8 ah = frag_read<BM, A_T, BK>(Ah, wr + 16 * i + l15, g, ks);
                        const bf16x8 al = frag_read<BM, A_T, BK>(Al, wr + 16 * i + l15, g, ks);
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            acc[i][j] = mfma16x16x32_bf16(ah, bl[j], acc[i][j]);
                            acc[i][j] = mfma16x16x32_bf16(al, bh[j], acc[i][j]);
                            acc[i][j] = mfma16x16x32_bf16(ah, bh[j], acc[i][j]);
                        }
                    }
                }
                __syncthreads();
            }
        }
    }
    // epilogue: acc[i][j][r] = C[m0 + wr + 16i + 4g + r][n0 + wc + 16j + l15].  Stored straight from the MFMA layout a
    // store instruction would touch 4 rows x 64 B; instead each wave transposes 32 rows x 64 columns at a time through
    // its own LDS slice (the K-loop's operand images are dead) and writes whole 256-byte row segments as 16-B stores
    float* T = (float*)smem + wave * (32 * GEMM_TLD);
    constexpr int WN = BN / 2;            // columns of a wave tile
    constexpr int LPR = WN / 4;           // lanes per row (16-byte chunks): 16 or 8
#pragma unroll
    for (int half = 0; half < MT / 2; ++half) {
        wave_lds_sync();
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) T[(16 * ii + 4 * g + r) * GEMM_TLD + 16 * j + l15] = acc[2 * half + ii][j][r];
        wave_lds_sync();
        const int c4 = 4 * (lane % LPR), col = n0 + wc + c4;
        f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (bias && blockIdx.z == 0 && col + 3 < N) bv = *(const f32x4*)(bias + col);
#pragma unroll
        for (int it = 0; it < 32 * LPR / 64; ++it) {
            const int rl = lane / LPR + (64 / LPR) * it, row = m0 + wr + 32 * half + rl;
            if (row >= M) continue;
            f32x4 v = *(const f32x4*)(T + rl * GEMM_TLD + c4);
            float* dst = C + (long)blockIdx.z * split_stride + (long)row * ldc + col;
            if (EPI == 1 && col + 3 < N) {          // N % 4 == 0 for every fused use
                *(f32x4*)dst = v;
                f32x4 a;
                for (int e = 0; e < 4; ++e) a[e] = gelu_erf(v[e] + bv[e]);
                *(f32x4*)(aux + (long)row * ldc + col) = a;
                continue;
            }
            if (EPI == 3 && col + 3 < N) {
                const float sc = col < CFFM_C ? 0.17677669529663687f : 1.f;
                typedef h16 h16x4 __attribute__((ext_vector_type(4)));
                h16x4 o;
                for (int e = 0; e < 4; ++e) o[e] = (h16)((v[e] + bv[e]) * sc);
                *(h16x4*)((h16*)aux + (long)row * ldc + col) = o;
                continue;
            }
            if (EPI == 2 && col + 3 < N) {
                *(f32x4*)dst = *(const f32x4*)(aux + (long)row * ldc + col) + v + bv;
                continue;
            }
            v += bv;
            if (col + 3 < N) {
                *(f32x4*)dst = v;
            } else {
                for (int e = 0; e < 4; ++e)
                    if (col + e < N) dst[e] = v[e] + ((bias && blockIdx.z == 0 && col + 3 >= N) ? bias[col + e] : 0.f);
            }
        }
    }
}
