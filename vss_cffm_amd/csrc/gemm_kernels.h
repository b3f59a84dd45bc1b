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
// per 256-thread workgroup (2x2 waves, each (BM/2) x (BN/2) as 16x16x32 MFMA tiles), 64-byte LDS rows
// with XOR-swizzled 16-byte chunks (conflict-free ds_read_b128 fragments: GEMM_SWZ), the next PF K-tiles in flight in registers while the
// current one is multiplied.  The weight-gradient form splits its long contraction (M ~ 10^4) over
// gridDim.z: every split writes its own partial output (plain coalesced stores) and a second tiny kernel sums
// them -- deterministic, and row-coalesced fp32 atomics measured 3-4x slower than this on gfx950.
#pragma once
#include "cffm_common.h"

// (bf16 types and the split-4 helpers: cffm_common.h)

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

// LDS image of a k-contiguous operand tile: 64-byte rows (32 bf16), NO padding, the four 16-byte chunks of a row stored at
// chunk ^ 2*((row>>3)&1).  A ds_read_b128 is served in four groups of 16 lanes that are not contiguous ({0-3,12-15,20-27},
// {4-11,16-19,28-31}, ...: MI355X_MICROARCH.md, LDS table): a group mixes 8 rows of k-chunk g with 8 rows of chunk g+1, so no
// row padding can separate them (an 80-byte stride measured 45 % of all LDS cycles as bank conflicts: SQ_LDS_BANK_CONFLICT
// / SQ_LDS_IDX_ACTIVE); with the XOR the 16 lanes of every group cover the 16 bank quads exactly once.
#define GEMM_SWZ(row) (2 * (((row) >> 3) & 1))
#define GEMM_IMG(ROWS, BK) (((ROWS) + 4) * (BK))   // bf16 per image: max of the two layouts ([ROWS][BK] | [BK/2][ROWS+4] dwords)
#define GEMM_TLD 68 // floats per row of the epilogue transposition tile (64 + 4 pad)

// LDS image of an operand tile (ROWS x 32 k), hi and lo parts:
//   k-contiguous operand (TR = false): [ROWS][32] bf16 with swizzled chunks (above), fragment = one ds_read_b128 of 8
//     consecutive k;
//   row-contiguous operand (TR = true): kept as it is loaded -- 4 consecutive rows at one k are 8 contiguous bytes -- in
//     [ROWS/16 row blocks][32 k-rows][16 rows] subtiles (1 KiB + 32 B pad each), and the fragment (8 k of row l&15) is read
//     with two LDS transpose reads (lds_tr4_bf16: 4 k-rows x 16 rows each), no k-pair packing when the tile is stored.
//     Inside a subtile k-row k sits at position p(k) = k with bits 2 and 3 swapped: the 32 lanes served together read
//     k in {8g..8g+3} for g = 0, 1 (or 2, 3), whose 32-byte segments must fall on 8 different 32-byte bank groups; the
//     32-byte pad per subtile does the same for the 8 row blocks a ds_write_b64 of one k touches.
#define GEMM_TSUB 528   // bf16 elements per row-block subtile: 32 * 16 + 16 pad
#define GEMM_TPOS(k) ((((k) & 3) | ((((k) >> 3) & 1) << 2) | ((((k) >> 2) & 1) << 3) | ((k) & 16)))
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

// (buf_make / buf_ld16: raw buffer loads, cffm_common.h)  The K-loop's fast path uses them so that a tile row past the
// matrix reads as zeros without a branch and the address is one per-thread VGPR offset (fixed for the whole kernel) plus a
// scalar offset that walks the contraction -- no per-tile vector address arithmetic, no exec-mask branches between the
// MFMAs (which would split the loop body into basic blocks and forbid interleaving the split with the multiplication).
// this thread's fixed byte offset inside an operand tile (whole tiles: the same items as tile_load)
template <int ROWS, bool TR, int BK>
__device__ __forceinline__ uint32_t tile_voff(int ld, int tid) {
    if (!TR) return (uint32_t)(((tid / (BK / 4)) * ld + 4 * (tid % (BK / 4))) * 4);
    return (uint32_t)((2 * (tid / (ROWS / 4)) * ld + 4 * (tid % (ROWS / 4))) * 4);
}
// global -> registers, branch-free; soff = byte offset of the tile origin ((row0*ld + k0)*4, or (k0*ld + row0)*4 when TR)
template <int ROWS, bool TR, int BK>
__device__ __forceinline__ void tile_load_fast(TileRegs<ROWS, BK>& r, buf_t rs, uint32_t voff, uint32_t soff, uint32_t ld4) {
    if (!TR) {
#pragma unroll
        for (int it = 0; it < ROWS * BK / 1024; ++it) r.v[it] = buf_ld16(rs, voff, soff + it * (1024 / BK) * ld4);
    } else {
#pragma unroll
        for (int it = 0; it < ROWS * BK / 2048; ++it)
#pragma unroll
            for (int h = 0; h < 2; ++h) r.v[2 * it + h] = buf_ld16(rs, voff, soff + (it * (2048 / ROWS) + h) * ld4);
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
// `pre`: 0 plain fp32 (split here) | 1 the registers hold split-4 data: no arithmetic | 2 (row-contiguous operands only) the operand
// is gelu(stored + bias): the weight gradient of fc2 reads the saved fc1 product `hraw` and re-applies bias + GELU while it stages the
// tile, so the forward never writes the activation (29.5 MB per block at CFFM-B1 size) -- bit-identical to what it would have stored
template <int ROWS, bool TR, int BK>
__device__ __forceinline__ void tile_store(const TileRegs<ROWS, BK>& r, bf16* __restrict__ hi, bf16* __restrict__ lo, int tid,
                                           int pre = 0, f32x4 bias4 = (f32x4){0.f, 0.f, 0.f, 0.f}) {
    if (!TR) {
#pragma unroll
        for (int it = 0; it < ROWS * BK / 1024; ++it) {
            const int item = tid + 256 * it;
            bf16x4 h, l;
            if (pre) unsplit4(r.v[it], h, l);
            else split4(r.v[it], h, l);
            const int row = item / (BK / 4), kc = 4 * (item % (BK / 4));
            const int o = row * BK + ((((kc >> 3) ^ GEMM_SWZ(row)) << 3) | (kc & 4));
            *(bf16x4*)(hi + o) = h;
            *(bf16x4*)(lo + o) = l;
        }
    } else {
#pragma unroll
        for (int it = 0; it < ROWS * BK / 2048; ++it) {
            const int item = tid + 256 * it;
            const int kp = item / (ROWS / 4), row = 4 * (item % (ROWS / 4));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bf16x4 hh, ll;
                if (pre == 1) unsplit4(r.v[2 * it + h], hh, ll);
                else if (pre == 2) {
                    f32x4 a;
                    for (int e = 0; e < 4; ++e) a[e] = gelu_erf(r.v[2 * it + h][e] + bias4[e]);
                    split4(a, hh, ll);
                } else split4(r.v[2 * it + h], hh, ll);
                const int o = (row >> 4) * GEMM_TSUB + GEMM_TPOS(2 * kp + h) * 16 + (row & 15);
                *(bf16x4*)(hi + o) = hh;
                *(bf16x4*)(lo + o) = ll;
            }
        }
    }
}

// one MFMA operand fragment (8 k-slots of row `row`) out of an LDS image
template <int ROWS, bool TR, int BK>
__device__ __forceinline__ bf16x8 frag_read(const bf16* __restrict__ img, int row, int g, int kk /* 0 or 32 */) {
    if (!TR) return *(const bf16x8*)(img + row * BK + kk + 8 * (g ^ GEMM_SWZ(row)));
    // row = (16-row block base) + (lane & 15): this lane points at k-row 8g + (l15 >> 2) (+4 for the second read), rows
    // 4 (l15 & 3) .. +3 of the block, and receives k = 8g .. 8g+7 of its own row l15 -- the k-slot order of the other layout
    const int l15 = row & 15;
    const bf16* sub = img + (row >> 4) * GEMM_TSUB + 4 * (l15 & 3);
    const int k0 = kk + 8 * g + (l15 >> 2);
    const bf16x4 a = lds_tr4_bf16(sub + GEMM_TPOS(k0) * 16);
    const bf16x4 b = lds_tr4_bf16(sub + GEMM_TPOS(k0 + 4) * 16);
    bf16x8 f;
    for (int e = 0; e < 4; ++e) { f[e] = a[e]; f[4 + e] = b[e]; }
    return f;
}

#define GEMM_LDS(BM, BN, BK) (((2 * GEMM_IMG(BM, BK) + 2 * GEMM_IMG(BN, BK)) * 4) > (4 * 32 * GEMM_TLD * 4) ? ((2 * GEMM_IMG(BM, BK) + 2 * GEMM_IMG(BN, BK)) * 4) : (4 * 32 * GEMM_TLD * 4))

// Scheduling hint for the pipelined K-loop body: issue the split's VALU work / LDS traffic between consecutive MFMAs
// (a wave issues in order: without it the compiler emits the MFMAs back to back and the split as a phase of its own).
#ifdef CFFM_EMU
#define GEMM_INTERLEAVE(NMFMA)
#else
#define GEMM_INTERLEAVE(NMFMA)                                                     \
    _Pragma("unroll") for (int q_ = 0; q_ < (NMFMA); ++q_) {                       \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); /* 1 MFMA */            \
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); /* 3 VALU */            \
        __builtin_amdgcn_sched_group_barrier(0x300, 1, 0); /* 1 DS read/write */   \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); /* 1 VMEM read */       \
    }
#endif

// grid (ceil(N/BN), ceil(M/BM), ksplit); BM, BN in {64, 128}; BK in {32, 64}; wave tile (BM/2) x (BN/2).
// K range of split z: [z*klen, min(K, (z+1)*klen)), klen a multiple of BK.
// EPI: 0 plain (+bias) | 1 GELU: C = raw product (saved for backward), aux = gelu(raw + bias) | 2 residual: C = aux + raw + bias
//      3 q|k|v: nothing in C; aux (h16 [M,N]) = f16(raw + bias), the q third (cols < 256) also times 32^-0.5 -- exactly the
//        values the attention kernels used to form from the fp32 product, stored once at half the bytes
//      5 as 1 with aux written in split-4 storage | 6 GELU backward: C = product * gelu'(aux + bias), aux2 = column-sum records
//        of C (one per 32-row slab of a wave) | 7 as 6 with C in split-4 storage
// XCD-aware placement (speed only): workgroup b runs on XCD b % 8, each XCD with its own 4 MiB L2.  The 1-D grid is
// re-numbered so that every XCD owns a CONTIGUOUS run of (k-slice, row panel, column tile) triples, column tile
// fastest: the tiles that share an A row panel / a k-slice then hit the same L2 instead of eight different ones.
__device__ __forceinline__ int xcd_linear_id() {
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, qq = nwg >> 3, rr = nwg & 7;
    return (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (blockIdx.x >> 3);
}

// One output tile (bx, by) of k-slice bz: the whole body of the GEMM kernels below.
template <int BM, int BN, int BK, bool A_T, bool B_T, int EPI, int PF>
__device__ __forceinline__ void gemm_tile(char* smem, const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                          int M, int N, int K, int lda, int ldb, int ldc, int klen, long split_stride,
                                          const float* __restrict__ bias, float* __restrict__ aux, int bx, int by, int bz,
                                          bool a_pre = false, int b_pre = 0, float* __restrict__ aux2 = nullptr,
                                          const float* __restrict__ b_bias = nullptr /* b_pre == 2: bias of the on-load GELU */) {
    bf16* Ah = (bf16*)smem;
    bf16* Al = Ah + GEMM_IMG(BM, BK);
    bf16* Bh = Al + GEMM_IMG(BM, BK);
    bf16* Bl = Bh + GEMM_IMG(BN, BK);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const int n0 = bx * BN, m0 = by * BM;
    f32x4 bb4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (B_T && b_pre == 2) bb4 = *(const f32x4*)(b_bias + n0 + 4 * (tid % (BN / 4)));   // this thread's four operand rows are fixed
    const int kbeg = bz * klen, kend = (kbeg + klen < K) ? kbeg + klen : K;
    constexpr int MT = BM / 32, NT = BN / 32;  // 16x16 tiles per wave along M and N (wave tile = BM/2 x BN/2)
    const int wr = (wave >> 1) * (BM / 2), wc = (wave & 1) * (BN / 2);
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Software pipeline, one barrier per K-tile: the LDS operand images are double-buffered, so while tile t is
    // multiplied out of buffer t&1 the SAME wave splits tile t+1 (already in registers) into buffer (t+1)&1 -- the split's
    // VALU work and LDS writes issue in the shadow of the MFMAs instead of in a phase of their own -- and tiles
    // t+2 .. t+1+PF are in flight from L2 / HBM into registers (a K-step is far shorter than that latency, and
    // co-resident workgroups run in lockstep, so they cannot hide each other's waits).
    constexpr int IMG = 2 * GEMM_IMG(BM, BK) + 2 * GEMM_IMG(BN, BK);   // bf16 elements of one buffer (Ah | Al | Bh | Bl)
    TileRegs<BM, BK> ra[PF];
    TileRegs<BN, BK> rb[PF];
    const int NKT = (kend - kbeg + BK - 1) / BK;
    tile_load<BM, A_T, BK>(ra[0], A, lda, m0, M, kbeg, kend, tid);
    tile_load<BN, B_T, BK>(rb[0], B, ldb, n0, N, kbeg, kend, tid);
    tile_store<BM, A_T, BK>(ra[0], Ah, Al, tid, a_pre);
    tile_store<BN, B_T, BK>(rb[0], Bh, Bl, tid, b_pre, bb4);
#pragma unroll
    for (int u = 0; u < PF; ++u) {   // tile 1+u -> register set u (past kend -> zeros, never used)
        tile_load<BM, A_T, BK>(ra[u], A, lda, m0, M, kbeg + (1 + u) * BK, kend, tid);
        tile_load<BN, B_T, BK>(rb[u], B, ldb, n0, N, kbeg + (1 + u) * BK, kend, tid);
    }
    __syncthreads();
    int t0 = 0;
    // Fast path (whole K-tiles; a row-contiguous operand also whole row tiles): while at least 2*PF tiles remain every
    // step multiplies, splits and loads unconditionally, so the loop body is ONE basic block and the scheduling hints
    // below can place the split's VALU work / LDS writes / buffer loads between the MFMAs.
    if ((kend - kbeg) % BK == 0 && (!A_T || M % BM == 0) && (!B_T || N % BN == 0)) {
        const buf_t rsa = buf_make(A, (uint32_t)((A_T ? K : M) * (long)lda * 4)), rsb = buf_make(B, (uint32_t)((B_T ? K : N) * (long)ldb * 4));
        const uint32_t va = tile_voff<BM, A_T, BK>(lda, tid), vb = tile_voff<BN, B_T, BK>(ldb, tid);
        const uint32_t lda4 = lda * 4, ldb4 = ldb * 4;
        // byte offset of tile 0 and per-tile advance
        const uint32_t sa0 = A_T ? (uint32_t)(kbeg * lda + m0) * 4 : (uint32_t)(m0 * lda + kbeg) * 4, dsa = A_T ? BK * lda4 : BK * 4;
        const uint32_t sb0 = B_T ? (uint32_t)(kbeg * ldb + n0) * 4 : (uint32_t)(n0 * ldb + kbeg) * 4, dsb = B_T ? BK * ldb4 : BK * 4;
        for (; t0 + 2 * PF < NKT; t0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int t = t0 + u;
                const int cur = (t & 1) * IMG, nxt = IMG - cur;
                bf16x8 bh[NT], bl[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    bh[j] = frag_read<BN, B_T, BK>(Bh + cur, wc + 16 * j + l15, g, 0);
                    bl[j] = frag_read<BN, B_T, BK>(Bl + cur, wc + 16 * j + l15, g, 0);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const bf16x8 ah = frag_read<BM, A_T, BK>(Ah + cur, wr + 16 * i + l15, g, 0);
                    const bf16x8 al = frag_read<BM, A_T, BK>(Al + cur, wr + 16 * i + l15, g, 0);
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        acc[i][j] = mfma16x16x32_bf16(ah, bl[j], acc[i][j]);
                        acc[i][j] = mfma16x16x32_bf16(al, bh[j], acc[i][j]);
                        acc[i][j] = mfma16x16x32_bf16(ah, bh[j], acc[i][j]);
                    }
                }
                tile_store<BM, A_T, BK>(ra[u], Ah + nxt, Al + nxt, tid, a_pre);
                tile_store<BN, B_T, BK>(rb[u], Bh + nxt, Bl + nxt, tid, b_pre, bb4);
                tile_load_fast<BM, A_T, BK>(ra[u], rsa, va, sa0 + (t + 1 + PF) * dsa, lda4);
                tile_load_fast<BN, B_T, BK>(rb[u], rsb, vb, sb0 + (t + 1 + PF) * dsb, ldb4);
                GEMM_INTERLEAVE(3 * MT * NT);
                __syncthreads();
            }
        }
    }
    static_assert(BK == 32, "the pipelined K-loop is written for 32-deep tiles");
    for (; t0 < NKT; t0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int t = t0 + u;
            if (t < NKT) {   // uniform across the workgroup
                const int cur = (t & 1) * IMG, nxt = IMG - cur;
#pragma unroll
                for (int ks = 0; ks < BK; ks += 32) {
                    bf16x8 bh[NT], bl[NT];
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        bh[j] = frag_read<BN, B_T, BK>(Bh + cur, wc + 16 * j + l15, g, ks);
                        bl[j] = frag_read<BN, B_T, BK>(Bl + cur, wc + 16 * j + l15, g, ks);
                    }
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const bf16x8 ah = frag_read<BM, A_T, BK>(Ah + cur, wr + 16 * i + l15, g, ks);
                        const bf16x8 al = frag_read<BM, A_T, BK>(Al + cur, wr + 16 * i + l15, g, ks);
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            acc[i][j] = mfma16x16x32_bf16(ah, bl[j], acc[i][j]);
                            acc[i][j] = mfma16x16x32_bf16(al, bh[j], acc[i][j]);
                            acc[i][j] = mfma16x16x32_bf16(ah, bh[j], acc[i][j]);
                        }
                    }
                }
                if (t + 1 < NKT) {
                    tile_store<BM, A_T, BK>(ra[u], Ah + nxt, Al + nxt, tid, a_pre);
                    tile_store<BN, B_T, BK>(rb[u], Bh + nxt, Bl + nxt, tid, b_pre, bb4);
                }
                if (t + 1 + PF < NKT) {
                    tile_load<BM, A_T, BK>(ra[u], A, lda, m0, M, kbeg + (t + 1 + PF) * BK, kend, tid);
                    tile_load<BN, B_T, BK>(rb[u], B, ldb, n0, N, kbeg + (t + 1 + PF) * BK, kend, tid);
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
        if (bias && bz == 0 && col + 3 < N) bv = *(const f32x4*)(bias + col);
        f32x4 csum = (f32x4){0.f, 0.f, 0.f, 0.f};   // EPI 6 / 7: this lane's column sums of the result over its rows
#pragma unroll
        for (int it = 0; it < 32 * LPR / 64; ++it) {
            const int rl = lane / LPR + (64 / LPR) * it, row = m0 + wr + 32 * half + rl;
            if (row >= M) continue;
            f32x4 v = *(const f32x4*)(T + rl * GEMM_TLD + c4);
            float* dst = C + (long)bz * split_stride + (long)row * ldc + col;
            if ((EPI == 1 || EPI == 5) && col + 3 < N) {          // N % 4 == 0 for every fused use
                *(f32x4*)dst = v;
                f32x4 a;
                for (int e = 0; e < 4; ++e) a[e] = gelu_erf(v[e] + bv[e]);
                *(f32x4*)(aux + (long)row * ldc + col) = (EPI == 5) ? split4_pack(a) : a;   // 5: act in split-4 storage
                continue;
            }
            if ((EPI == 6 || EPI == 7) && col + 3 < N) {          // GELU backward: result = product * gelu'(aux + bias)
                const f32x4 hr = *(const f32x4*)(aux + (long)row * ldc + col) + bv;
                f32x4 o;
                for (int e = 0; e < 4; ++e) o[e] = v[e] * gelu_erf_grad(hr[e]);
                *(f32x4*)dst = (EPI == 7) ? split4_pack(o) : o;      // 7: split-4 storage
                csum += o;
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
                    if (col + e < N) dst[e] = v[e] + ((bias && bz == 0 && col + 3 >= N) ? bias[col + e] : 0.f);
            }
        }
        if (EPI == 6 || EPI == 7) {
            // column sums of this wave's 32 rows (bias gradient of fc1): the LPR-lane groups of the wave hold different rows of
            // the same columns; one record per (row tile, wave row, half): aux2[((by * 2 + (wave >> 1)) * (MT / 2) + half)][N]
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = LPR; m < 64; m <<= 1) csum[e] += __shfl_xor(csum[e], m, 64);
            if (lane < LPR && col + 3 < N)
                *(f32x4*)(aux2 + ((long)(by * 2 + (wave >> 1)) * (MT / 2) + half) * N + col) = csum;
        }
    }
}

template <int BM, int BN, int BK, bool A_T, bool B_T, int EPI, int PF, bool A_PRE = false, bool B_PRE = false>
__global__ void __launch_bounds__(256) k_gemm_split(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                     int M, int N, int K, int lda, int ldb, int ldc, int klen, long split_stride,
                                                     const float* __restrict__ bias, float* __restrict__ aux, float* __restrict__ aux2) {
    CFFM_DYN_SMEM(smem);
    const int lin = xcd_linear_id();
    const int ntn = (N + BN - 1) / BN, ntm = (M + BM - 1) / BM;
    gemm_tile<BM, BN, BK, A_T, B_T, EPI, PF>(smem, A, B, C, M, N, K, lda, ldb, ldc, klen, split_stride, bias, aux, lin % ntn,
                                             (lin / ntn) % ntm, lin / (ntn * ntm), A_PRE, B_PRE, aux2);
}

// Up to GEMM_GROUP_MAX independent weight-gradient GEMMs (dw = dy^T x, <T,T>, 128x128 tiles) in ONE launch.  The four of a
// block (fc2, fc1, proj, qkv) have 4-16 output tiles each and a contraction of ~10^4 rows: launched one by one each needs
// ~20 k-slices to occupy 256 CUs, i.e. 12 K-steps per workgroup between a prologue and a 64 KB partial-tile epilogue, and
// its own partial-sum pass.  None of them feeds the backward chain, so they are deferred and share one grid: ~480
// workgroups with one common slice length (2-3x longer), a third of the partial traffic and one summation launch.
#ifndef GROUP_PF
#define GROUP_PF 1
#endif
#define GEMM_GROUP_MAX 4
struct GemmGroup {
    const float* A[GEMM_GROUP_MAX];
    const float* B[GEMM_GROUP_MAX];
    float* C[GEMM_GROUP_MAX];          // partial outputs [ksplit][M][N] (or the output itself when ksplit == 1)
    int M[GEMM_GROUP_MAX], N[GEMM_GROUP_MAX], K[GEMM_GROUP_MAX];
    int wg_end[GEMM_GROUP_MAX];        // exclusive prefix of workgroups per problem (after XCD re-numbering)
    int klen, n;
    int a_pre[GEMM_GROUP_MAX], b_pre[GEMM_GROUP_MAX];   // operand already in split-4 storage (1); B only: 2 = gelu(stored + b_bias) on load
    const float* b_bias[GEMM_GROUP_MAX];
};
__global__ void __launch_bounds__(256) k_gemm_group_tt(GemmGroup G) {
    CFFM_DYN_SMEM(smem);
    int lin = xcd_linear_id(), p = 0;
#pragma unroll
    for (int q = 0; q < GEMM_GROUP_MAX - 1; ++q)
        if (q + 1 < G.n && lin >= G.wg_end[q]) p = q + 1;
    if (p > 0) lin -= G.wg_end[p - 1];
    const int M = G.M[p], N = G.N[p], ntn = N / 128, ntm = M / 128;
    gemm_tile<128, 128, 32, true, true, 0, GROUP_PF>(smem, G.A[p], G.B[p], G.C[p], M, N, G.K[p], M, N, N, G.klen, (long)M * N, nullptr,
                                              nullptr, lin % ntn, (lin / ntn) % ntm, lin / (ntn * ntm), G.a_pre[p] != 0, G.b_pre[p], nullptr, G.b_bias[p]);
}
