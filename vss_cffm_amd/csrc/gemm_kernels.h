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
// Tiles: BM x 128 x 32 per 256-thread workgroup (2x2 waves, each (BM/2) x 64 as 16x16x32 MFMA tiles), LDS rows
// padded to 80 B (conflict-free ds_read_b128 fragments), next K-tile prefetched into registers while the
// current one is multiplied.  The weight-gradient form splits its long contraction (M ~ 10^4) over
// gridDim.z and accumulates with fp32 atomics into the zeroed output.
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

#define GEMM_BN 128
#define GEMM_BK 32
#define GEMM_LD 40  // bf16 per LDS row (32 + 8 pad)

// items of one operand tile per thread: ROWS*8 f32x4 chunks / 256 threads
template <int ROWS, bool TR>
struct TileRegs { f32x4 v[ROWS / 32]; };

// global -> registers.  Non-transposed: chunk = 4 consecutive k of one row.  Transposed: chunk = 4 consecutive rows of one k.
template <int ROWS, bool TR>
__device__ __forceinline__ void tile_load(TileRegs<ROWS, TR>& r, const float* __restrict__ P, int ld, int row0, int nrows,
                                          int k0, int kend, int tid) {
    const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < ROWS / 32; ++it) {
        const int item = tid + 256 * it;
        if (!TR) {
            const int row = row0 + (item >> 3), k = k0 + 4 * (item & 7);
            r.v[it] = (row < nrows && k < kend) ? *(const f32x4*)(P + (long)row * ld + k) : z;   // kend, k multiples of 4
        } else {
            const int k = k0 + item / (ROWS / 4), row = row0 + 4 * (item % (ROWS / 4));
            f32x4 v = z;
            if (k < kend) {
                if (row + 3 < nrows) v = *(const f32x4*)(P + (long)k * ld + row);
                else
                    for (int e = 0; e < 4; ++e)
                        if (row + e < nrows) v[e] = P[(long)k * ld + row + e];
            }
            r.v[it] = v;
        }
    }
}

// registers -> LDS (split into hi / lo bf16 images [ROWS][GEMM_LD])
template <int ROWS, bool TR>
__device__ __forceinline__ void tile_store(const TileRegs<ROWS, TR>& r, bf16* __restrict__ hi, bf16* __restrict__ lo, int tid) {
#pragma unroll
    for (int it = 0; it < ROWS / 32; ++it) {
        const int item = tid + 256 * it;
        bf16x4 h, l;
        split4(r.v[it], h, l);
        if (!TR) {
            const int row = item >> 3, kc = 4 * (item & 7);
            *(bf16x4*)(hi + row * GEMM_LD + kc) = h;
            *(bf16x4*)(lo + row * GEMM_LD + kc) = l;
        } else {
            const int k = item / (ROWS / 4), row = 4 * (item % (ROWS / 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hi[(row + e) * GEMM_LD + k] = h[e];
                lo[(row + e) * GEMM_LD + k] = l[e];
            }
        }
    }
}

#define GEMM_LDS(BM) ((2 * (BM) + 2 * GEMM_BN) * GEMM_LD * 2)

// grid (ceil(N/128), ceil(M/BM), ksplit).  K range of split z: [z*klen, min(K, (z+1)*klen)), klen multiple of 32.
template <int BM, bool A_T, bool B_T>
__global__ void __launch_bounds__(256) k_gemm_split(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                     int M, int N, int K, int lda, int ldb, int ldc, int klen, int atomic_out,
                                                     const float* __restrict__ bias) {
    CFFM_DYN_SMEM(smem);
    bf16* Ah = (bf16*)smem;
    bf16* Al = Ah + BM * GEMM_LD;
    bf16* Bh = Al + BM * GEMM_LD;
    bf16* Bl = Bh + GEMM_BN * GEMM_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * GEMM_BN, m0 = blockIdx.y * BM;
    const int kbeg = blockIdx.z * klen, kend = (kbeg + klen < K) ? kbeg + klen : K;
    constexpr int MT = BM / 32;  // 16-row tiles per wave along M (wave tile = BM/2 x 64)
    const int wr = (wave >> 1) * (BM / 2), wc = (wave & 1) * 64;
    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    TileRegs<BM, A_T> ra;
    TileRegs<GEMM_BN, B_T> rb;
    tile_load<BM, A_T>(ra, A, lda, m0, M, kbeg, kend, tid);
    tile_load<GEMM_BN, B_T>(rb, B, ldb, n0, N, kbeg, kend, tid);
    for (int k0 = kbeg; k0 < kend; k0 += GEMM_BK) {
        tile_store<BM, A_T>(ra, Ah, Al, tid);
        tile_store<GEMM_BN, B_T>(rb, Bh, Bl, tid);
        __syncthreads();
        if (k0 + GEMM_BK < kend) {  // next K-tile flies while this one is multiplied
            tile_load<BM, A_T>(ra, A, lda, m0, M, k0 + GEMM_BK, kend, tid);
            tile_load<GEMM_BN, B_T>(rb, B, ldb, n0, N, k0 + GEMM_BK, kend, tid);
        }
        bf16x8 bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bh[j] = *(const bf16x8*)(Bh + (wc + 16 * j + l15) * GEMM_LD + 8 * g);
            bl[j] = *(const bf16x8*)(Bl + (wc + 16 * j + l15) * GEMM_LD + 8 * g);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const bf16x8 ah = *(const bf16x8*)(Ah + (wr + 16 * i + l15) * GEMM_LD + 8 * g);
            const bf16x8 al = *(const bf16x8*)(Al + (wr + 16 * i + l15) * GEMM_LD + 8 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = mfma16x16x32_bf16(ah, bl[j], acc[i][j]);
                acc[i][j] = mfma16x16x32_bf16(al, bh[j], acc[i][j]);
                acc[i][j] = mfma16x16x32_bf16(ah, bh[j], acc[i][j]);
            }
        }
        __syncthreads();
    }
    // epilogue: acc[i][j][r] = C[m0 + wr + 16i + 4g + r][n0 + wc + 16j + l15]
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wc + 16 * j + l15;
            if (col >= N) continue;
            const float bv = (bias && blockIdx.z == 0) ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wr + 16 * i + 4 * g + r;
                if (row >= M) continue;
                float* dst = C + (long)row * ldc + col;
                if (atomic_out) atomicAdd(dst, acc[i][j][r] + bv);
                else *dst = acc[i][j][r] + bv;
            }
        }
}
