// dws_kernels.h -- weight gradients dw[N][K] = dy[M][N]^T x[M][K] (contraction over ~10^4 token rows) with BOTH operands stored in
// MFMA-fragment order along the contraction ("T-frag" storage), streamed straight into registers (round 5, third session).
//
// Why a third weight-gradient kernel.  k_gemm_group_tt stages row-major tiles global -> registers -> bf16 split -> ds_write -> barrier ->
// transposed fragment reads -> MFMAs with one wave per SIMD; its ablations (profiles/r05_dw_ablate.txt) say that chain, not the matrix
// pipe, is what a K-step costs (~1.7 k cycles against 768 of MFMA), and the LDS-DMA kernel (dw_kernels.h) that removes the VALU part
// of it is bound by the DMA rate (~22 B/clk/CU).  The row-panel kernels (panel_kernels.h) show what does run on this chip: operand
// fragments fetched with fully coalesced 1 KiB buffer loads straight into registers, a ring of k-steps ahead of the MFMAs, no LDS, no
// barrier, ~45 B/clk per CU out of L2.  For a weight gradient both operands would have to be read TRANSPOSED (a lane of an MFMA A / B
// fragment holds 8 consecutive contraction steps = 8 token rows of ONE feature), which row-major storage cannot give without LDS.  But
// every operand of the block's weight gradients passes through a row-panel kernel that has it in LDS as hi / lo bf16 images of 32 (or
// 48) whole token rows: z2, act (k_mlp_fwd writes them), dh, dx1 (k_mlp_bwd writes them), ao, dout (read as panels by the same
// kernels), zall, dqkv (read as panels by the q|k|v panel GEMMs).  Those kernels leave a copy in T-frag order -- two transposed LDS
// reads (ds_read_b64_tr_b16) and one 1 KiB store per (16 features, 32 rows, hi | lo) -- and the kernel below needs nothing else:
//
//   T-frag storage of X[R][C] (C % 16 == 0):  unit (ks, jt, h) = 64 lanes x 16 B at 16-byte word ((ks * C/16 + jt) * 2 + h) * 64;
//   lane (l15, g) of it holds X[32 ks + 8 g + e][16 jt + l15], e = 0..7, as bf16 hi (h = 0) or lo (h = 1); rows >= R are zeros.
//   Same bytes as fp32 / split-4 storage (plus the padding of the last k-step).
//
// A wave owns a 128 (in features, x) x 64 (out features, dy) tile of dw and a run of k-steps: per k-step 16 + 8 KiB-loads (24 KB) for
// 96 MFMAs (32 tiles x 3 passes = 1536 cycles), i.e. 16 B/clk per wave, 64 per CU with one wave per SIMD: the L2 -> CU path is the
// bound (~70 % of the matrix pipe at 45 B/clk), which is 2-3x the 0.28 of the three-pass peak the LDS-staged kernel reaches.  The four
// waves of a workgroup take four k-runs of the SAME tile and add their accumulators through LDS in a fixed order ((w0 + w2) + (w1 + w3)),
// so a workgroup leaves ONE 32 KB partial tile; k-slices across workgroups go to slabs as before (k_sum_splits_group adds them in
// slice order): deterministic, and a third of the slab traffic of the 128 x 128 kernel (7 MB instead of 20.6 at B = 2).
#pragma once
#include "gemm_kernels.h"

#define DWS_MAX 4
#define DWS_TI 128                 // in features (x columns) per tile: 8 A units per k-step
#define DWS_TO 64                  // out features (dy columns) per tile: 4 B units per k-step
#define DWS_TLD 132                // floats per row of a reduction tile (row = out feature): 16-byte LDS stores of 8 lanes hit 32 different banks
#define DWS_LDS (2 * DWS_TO * DWS_TLD * 4)

__device__ __host__ __forceinline__ long tfrag_unit(long ks, int jt, int h, int CT) { return ((ks * CT + jt) * 2 + h) * 64; }

// row-major fp32 (pre == 0) or split-4 storage (pre == 1) -> T-frag storage; one thread per (unit, lane).  A utility (tests, operands
// that no panel kernel sees): the block's operands are written in this order by their producers (panel_kernels.h, pnl_tfrag_store)
__global__ void __launch_bounds__(256) k_tfrag_pack(const float* __restrict__ X, long R, int C, int pre, f32x4* __restrict__ dst, long items) {
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    if (item >= items) return;
    const int lane = (int)(item & 63), l15 = lane & 15, g = lane >> 4, CT = C / 16;
    const long u = item >> 6, ks = u / CT;
    const int jt = (int)(u % CT), col = 16 * jt + l15;
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const long row = 32 * ks + 8 * g + e;
        bf16 h = (bf16)0.f, l = (bf16)0.f;
        if (row < R) {
            if (pre) {
                const bf16* q = (const bf16*)(X + row * C + (col & ~3));
                h = q[col & 3];
                l = q[4 + (col & 3)];
            } else {
                const float x = X[row * C + col];
                h = (bf16)x;
                l = (bf16)(x - (float)h);
            }
        }
        hi[e] = h;
        lo[e] = l;
    }
    dst[tfrag_unit(ks, jt, 0, CT) + lane] = __builtin_bit_cast(f32x4, hi);
    dst[tfrag_unit(ks, jt, 1, CT) + lane] = __builtin_bit_cast(f32x4, lo);
}

struct DwsGroup {
    const f32x4* X[DWS_MAX];       // x  [M][K] in T-frag storage
    const f32x4* DY[DWS_MAX];      // dy [M][N] in T-frag storage
    float* C[DWS_MAX];             // partial outputs [slices][N][K] (or dw itself when there is one slice)
    int N[DWS_MAX], K[DWS_MAX];
    int KS[DWS_MAX];               // k-steps of 32 token rows
    int kw[DWS_MAX];               // k-steps per wave: wave w of slice s multiplies k-steps [(4 s + w) kw, + kw)
    int wg_end[DWS_MAX];           // exclusive prefix of workgroups per problem (after XCD re-numbering)
    int n;
};

// Operand ring of a wave (registers): two k-steps of x units (8 units x {hi, lo} each), refilled IN PLACE -- the two registers of unit i
// are requested for k-step s + 2 as soon as the 12 MFMAs that read them for k-step s have been issued -- and three k-steps of dy units
// (4 units x {hi, lo}; every MFMA of a k-step reads them, so they are refilled at its end, for k-step s + 3): 224 registers hold what a
// plain three-deep ring would need 288 for, and every load has two k-steps (~3-4 k cycles) to arrive.
struct DwsRing { f32x4 a[2][16], b[3][8]; };
__device__ __forceinline__ void dws_load_a(f32x4 (&a)[16], int i, buf_t rsx, uint32_t voff, uint32_t sx) {
    a[2 * i] = buf_ld16(rsx, voff, sx + 2048u * i);
    a[2 * i + 1] = buf_ld16(rsx, voff, sx + 2048u * i + 1024u);
}
__device__ __forceinline__ void dws_load_b(f32x4 (&b)[8], buf_t rsy, uint32_t voff, uint32_t sy) {
#pragma unroll
    for (int u = 0; u < 8; ++u) b[u] = buf_ld16(rsy, voff, sy + 1024u * u);
}
// One MFMA with the accumulator pinned to the AGPR half of the register file and the operands to the VGPR half.  A wave holds 128
// accumulator registers + 2 x 96 of operand ring: left to itself (the builtin) hipcc spreads both over both halves and moves ~100 registers
// between them per k-step (v_accvgpr_read / _write / _mov); with the classes fixed the loop body is loads + MFMAs only.  The hazard
// recogniser does not see inside an asm statement: dws_mma never issues two dependent MFMAs within 32 instructions, and the kernel puts
// explicit wait states between the accumulators' initialisation / the last MFMA and the other instructions that touch them (dws_settle).
__device__ __forceinline__ f32x4 dws_mfma(f32x4 a, f32x4 b, f32x4 c) {
#ifdef CFFM_EMU
    return mfma16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c);
#else
    asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    return c;
#endif
}
__device__ __forceinline__ void dws_settle() {
#ifndef CFFM_EMU
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#endif
}
// x unit i against the four dy units: three passes (hi x lo, lo x hi, hi x hi as in the other kernels); an accumulator is touched again
// four MFMAs later
__device__ __forceinline__ void dws_mma_unit(f32x4 (&acc)[8][4], const f32x4 (&a)[16], const f32x4 (&b)[8], int i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = dws_mfma(a[2 * i], b[2 * j + 1], acc[i][j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = dws_mfma(a[2 * i + 1], b[2 * j], acc[i][j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = dws_mfma(a[2 * i], b[2 * j], acc[i][j]);
}
__device__ __forceinline__ void dws_pin() {
#ifndef CFFM_EMU
    __builtin_amdgcn_sched_barrier(0);
#endif
}
__device__ __forceinline__ void dws_barrier() {      // LDS traffic only (see pnl_lds_barrier)
#ifdef CFFM_EMU
    __syncthreads();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#endif
}

__global__ void __launch_bounds__(256) k_dw_stream(DwsGroup G) {
    CFFM_DYN_SMEM(smem);
    int lin = xcd_linear_id(), p = 0;
#pragma unroll
    for (int q = 0; q < DWS_MAX - 1; ++q)
        if (q + 1 < G.n && lin >= G.wg_end[q]) p = q + 1;
    if (p > 0) lin -= G.wg_end[p - 1];
    const int N = G.N[p], K = G.K[p], KS = G.KS[p], kw = G.kw[p];
    const int nto = N / DWS_TO, ntile = nto * (K / DWS_TI);
    const int tile = lin % ntile, slice = lin / ntile, to = tile % nto, ti = tile / nto;
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), l15 = lane & 15, g = lane >> 4;
    const int CTx = K / 16, CTy = N / 16;
    const buf_t rsx = buf_make(G.X[p], (uint32_t)((long)KS * CTx * 2048)), rsy = buf_make(G.DY[p], (uint32_t)((long)KS * CTy * 2048));
    const uint32_t voff = lane * 16;
    const int ks0 = (4 * slice + wave) * kw, nst = (ks0 + kw <= KS ? kw : KS - ks0);     // (nst <= 0: nothing of the contraction left for this wave)
    const uint32_t dx = (uint32_t)CTx * 2048u, dy = (uint32_t)CTy * 2048u;            // bytes per k-step
    const uint32_t sx0 = (uint32_t)ks0 * dx + (uint32_t)(8 * ti) * 2048u, sy0 = (uint32_t)ks0 * dy + (uint32_t)(4 * to) * 2048u;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // The contraction loop is branch-free: six k-steps per trip (the x ring turns every two, the dy ring every three), ceil(nst / 6) trips;
    // a k-step past the wave's run is requested with an out-of-bounds lane offset (the buffer unit returns zeros without touching
    // memory) and multiplied like any other.  (With the run's end tested by branches hipcc keeps the accumulators in different register
    // classes on the two sides and copies all 128 of them through v_accvgpr_read / _write every trip.)
    DwsRing ring;
#define DWS_OK(ks_) ((ks_) < nst)
#define DWS_VO(ks_) (DWS_OK(ks_) ? voff : BUF_OOB)
#define DWS_SX(ks_) (DWS_OK(ks_) ? sx0 + (uint32_t)(ks_) * dx : 0u)
#define DWS_SY(ks_) (DWS_OK(ks_) ? sy0 + (uint32_t)(ks_) * dy : 0u)
#pragma unroll
    for (int d = 0; d < 2; ++d) {
#pragma unroll
        for (int i = 0; i < 8; ++i) dws_load_a(ring.a[d], i, rsx, DWS_VO(d), DWS_SX(d));
        dws_load_b(ring.b[d], rsy, DWS_VO(d), DWS_SY(d));
    }
    dws_load_b(ring.b[2], rsy, DWS_VO(2), DWS_SY(2));
    dws_settle();
    dws_pin();
    const int nit = (nst + 5) / 6;
    for (int it = 0; it < nit; ++it) {
#pragma unroll
        for (int d = 0; d < 6; ++d) {
            const int ks = it * 6 + d;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                dws_mma_unit(acc, ring.a[d % 2], ring.b[d % 3], i);
                dws_pin();      // (the refill stays BEHIND the MFMAs that read the registers: hoisted above them it needs new ones and copies)
                dws_load_a(ring.a[d % 2], i, rsx, DWS_VO(ks + 2), DWS_SX(ks + 2));
                dws_pin();
            }
            dws_load_b(ring.b[d % 3], rsy, DWS_VO(ks + 3), DWS_SY(ks + 3));
            dws_pin();
        }
    }
#undef DWS_OK
#undef DWS_VO
#undef DWS_SX
#undef DWS_SY
    dws_settle();
    // ---- the four waves' partial tiles -> one: (w0 + w2) + (w1 + w3) through two LDS tiles [64 out][128 in]
    float* T0 = (float*)smem;
    float* T1 = T0 + DWS_TO * DWS_TLD;
    // acc[i][j][r] = dw[out = 16 j + l15][in = 16 i + 4 g + r]
#define DWS_AT(T, i, j) (*(f32x4*)((T) + (16 * (j) + l15) * DWS_TLD + 16 * (i) + 4 * g))
    if (wave >= 2) {
        float* T = wave == 2 ? T0 : T1;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) DWS_AT(T, i, j) = acc[i][j];
    }
    dws_barrier();
    if (wave < 2) {
        float* T = wave == 0 ? T0 : T1;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] += DWS_AT(T, i, j);
#pragma unroll
        for (int i = 0; i < 8; ++i)      // (a wave re-writes exactly the words it has just read: no other wave touches its tile)
#pragma unroll
            for (int j = 0; j < 4; ++j) DWS_AT(T, i, j) = acc[i][j];
    }
#undef DWS_AT
    dws_barrier();
    float* C = G.C[p] + (long)slice * N * K + (long)(to * DWS_TO) * K + ti * DWS_TI;
    const int c4 = 4 * (tid & 31), r0 = tid >> 5;
#pragma unroll
    for (int it = 0; it < DWS_TO / 8; ++it) {
        const int row = r0 + 8 * it;
        *(f32x4*)(C + (long)row * K + c4) = *(const f32x4*)(T0 + row * DWS_TLD + c4) + *(const f32x4*)(T1 + row * DWS_TLD + c4);
    }
}
