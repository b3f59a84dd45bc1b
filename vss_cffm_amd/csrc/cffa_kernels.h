// cffa_kernels.h -- Coarse-to-Fine Feature Assembling (CFFA) kernels + layout transposes.
//
// Reference semantics: CffmTransformerBlock3d3.forward, cffm_transformer.py:709-805
// (LayerNorm on all frames :716, zero padding :721-724, target window pooling :741-776,
// per-reference-frame bilinear resize + window pooling :780-805), restated in SURVEY.md A.1-A.6.
//
// MI355X design: HBM-bound.  One workgroup per (window, frame, clip) reads its 49 pixels x 256
// channels as 1 KiB coalesced rows (one wave per pixel, one f32x4 per lane), LayerNorms them in
// registers and reduces them against the composed [15 x 49] window-local pooling matrix
// (learned pool weights x constant bilinear taps), so the LayerNormed reference frames never
// reach HBM -- only 15 pooled rows per window do, plus the target frame's tokens in window-major
// order (which makes every later per-window access contiguous).
#pragma once
#include "cffm_common.h"

// --------------------------------------------------------------------------- batched transpose
// dst[n][c][r] = src[n][r][c]; src rows x cols, batch strides given (elements).
// NCHW -> NHWC : rows = C, cols = H*W.   NHWC -> NCHW : rows = H*W, cols = C.
// 64 x 64 tiles through LDS; 16-byte global accesses on both sides whenever the contiguous extents allow it (the layer's
// shapes always do: 256 channels, H*W a multiple of 4 for the 60 x 60 grid); otherwise 4-byte accesses, same tile walk.
// addend (may be NULL): laid out like dst; added to image z unless z % add_mod == add_skip (the layer backward's last transpose:
// dx of the three pass-through frames also receives the upstream gradient of those frames, cffm_transformer.py:826).
// copy_dst (may be NULL): laid out like src; image z is ALSO copied there unchanged unless z % add_mod == add_skip (the layer
// forward's first transpose: the three pass-through frames of the reference's output, cffm_transformer.py:826, leave with the read
// that the transpose does anyway -- round 2 ran a separate 44 MB copy kernel beside it).
__device__ __forceinline__ void transpose_body(const float* __restrict__ src, float* __restrict__ dst,
                                               int rows, int cols, long src_bs, long dst_bs,
                                               const float* __restrict__ addend, int add_mod, int add_skip,
                                               float* __restrict__ copy_dst, int bx, int by, int bz) {
    __shared__ float tile[64][65];   // tile[c][r]; odd stride: the scalar LDS accesses below are at most 2-way conflicted
    const int r0 = by * 64, c0 = bx * 64;
    const float* s = src + (long)bz * src_bs;
    float* d = dst + (long)bz * dst_bs;
    const int q = threadIdx.x & 15, p = threadIdx.x >> 4;
    const bool vin = (cols % 4 == 0) && (src_bs % 4 == 0) && (((uintptr_t)src & 15) == 0);
    const bool vout = (rows % 4 == 0) && (dst_bs % 4 == 0) && (((uintptr_t)dst & 15) == 0) && (((uintptr_t)addend & 15) == 0);
    const float* ad = (addend && (bz % add_mod) != add_skip) ? addend + (long)bz * dst_bs : nullptr;
    float* cp = (copy_dst && (bz % add_mod) != add_skip) ? copy_dst + (long)bz * src_bs : nullptr;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + p + 16 * k, c = c0 + 4 * q;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (r < rows) {
            if (vin && c + 3 < cols) {
                v = *(const f32x4*)(s + (long)r * cols + c);
                if (cp) *(f32x4*)(cp + (long)r * cols + c) = v;
            } else
                for (int e = 0; e < 4; ++e)
                    if (c + e < cols) { v[e] = s[(long)r * cols + c + e]; if (cp) cp[(long)r * cols + c + e] = v[e]; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[4 * q + e][p + 16 * k] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int cl = p + 16 * k, c = c0 + cl, r = r0 + 4 * q;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tile[cl][4 * q + e];
        if (c < cols) {
            if (vout && r + 3 < rows) *(f32x4*)(d + (long)c * rows + r) = ad ? v + *(const f32x4*)(ad + (long)c * rows + r) : v;
            else
                for (int e = 0; e < 4; ++e)
                    if (r + e < rows) d[(long)c * rows + r + e] = v[e] + (ad ? ad[(long)c * rows + r + e] : 0.f);
        }
    }
}
__global__ void __launch_bounds__(256) k_transpose(const float* __restrict__ src, float* __restrict__ dst,
                                                    int rows, int cols, long src_bs, long dst_bs,
                                                    const float* __restrict__ addend, int add_mod, int add_skip,
                                                    float* __restrict__ copy_dst = nullptr) {
    transpose_body(src, dst, rows, cols, src_bs, dst_bs, addend, add_mod, add_skip, copy_dst, blockIdx.x, blockIdx.y, blockIdx.z);
}

// dst[z][0..n4) = src[z][0..n4) (16-byte units, batch strides in floats): the pass-through frames of the layer output
__global__ void __launch_bounds__(256) k_copy_batched(const float* __restrict__ src, float* __restrict__ dst, long n4, long src_bs, long dst_bs) {
    const f32x4* s = (const f32x4*)(src + (long)blockIdx.y * src_bs);
    f32x4* d = (f32x4*)(dst + (long)blockIdx.y * dst_bs);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) d[i] = s[i];
}

// --------------------------------------------------------------------------- pooling matrix
// Cell g of a window is a fixed linear functional of the window's 49 LayerNormed pixels
// (SURVEY.md A.6 "window-locality"): g=0 target pool, g=1 frame t-9 (7x7 pool), g=2..5 frame t-6
// (7->6 bilinear, 3x3 pool, 2x2 cells), g=6..14 frame t-3 (7->6 bilinear, 2x2 pool, 3x3 cells).
// Bilinear 7->6, align_corners=False: output r in 0..5 reads source r and r+1 with weights
// (11-2r)/12 and (1+2r)/12 (F.interpolate at cffm_transformer.py:795; scale 7/6 is window periodic).
__device__ __forceinline__ float bil_tap(int r, int p) {
    if (p == r) return (11.f - 2.f * r) * (1.f / 12.f);
    if (p == r + 1) return (1.f + 2.f * r) * (1.f / 12.f);
    return 0.f;
}

// which pool weight vector / geometry a cell uses
__device__ __forceinline__ void cell_geom(int g, int& grp, int& u, int& v, int& wsg) {
    if (g == 0) { grp = 0; u = v = 0; wsg = 7; }
    else if (g == 1) { grp = 1; u = v = 0; wsg = 7; }
    else if (g < 6) { grp = 2; u = (g - 2) >> 1; v = (g - 2) & 1; wsg = 3; }
    else { grp = 3; u = (g - 6) / 3; v = (g - 6) % 3; wsg = 2; }
}

struct PoolW { const float* w[4]; };  // pool_layers.0, pool_layers_clips.{0,1,2} weights
struct PoolWG { float* w[4]; float* b[4]; };   // b: the four pool-bias gradients, or NULL (see k_pool_matrix_bwd: padding of flat gradient buffers)

// entry (cell g, window pixel i) of the pooling matrix
__device__ __forceinline__ float pool_matrix_entry(const PoolW& pw, int g, int i) {
    const int py = i / 7, px = i % 7;
    int grp, u, v, wsg;
    cell_geom(g, grp, u, v, wsg);
    if (grp < 2) return pw.w[grp][i];
    float m = 0.f;
    for (int a = 0; a < wsg; ++a)
        for (int b = 0; b < wsg; ++b)
            m += pw.w[grp][a * wsg + b] * bil_tap(wsg * u + a, py) * bil_tap(wsg * v + b, px);
    return m;
}
__device__ __forceinline__ void pool_matrix_body(const PoolW& pw, float* __restrict__ M) {
    for (int e = threadIdx.x; e < CFFM_NCELL * CFFM_WA; e += 256) M[e] = pool_matrix_entry(pw, e / CFFM_WA, e % CFFM_WA);
}

__global__ void __launch_bounds__(256) k_pool_matrix(PoolW pw, float* __restrict__ M) { pool_matrix_body(pw, M); }

// dW_pool[grp][k] = sum_{g in grp, i} dM[g][i] * dM[g][i]/dw.  One wave per weight (111 weights): lanes over
// the 49 pixels, loop over the group's cells, wave reduction.
__device__ __forceinline__ void pool_matrix_bwd_body(const float* __restrict__ dM, const PoolWG& gw, int t) {
    const int i = threadIdx.x;
    int grp, k;
    if (t < 49) { grp = 0; k = t; }
    else if (t < 98) { grp = 1; k = t - 49; }
    else if (t < 107) { grp = 2; k = t - 98; }
    else { grp = 3; k = t - 107; }
    float acc = 0.f;
    if (grp < 2) {
        if (i == k) acc = dM[grp * CFFM_WA + k];
    } else if (i < CFFM_WA) {
        const int wsg = grp == 2 ? 3 : 2, ncell = grp == 2 ? 2 : 3, g0 = grp == 2 ? 2 : 6;
        const int a = k / wsg, b = k % wsg;
        for (int u = 0; u < ncell; ++u)
            for (int v = 0; v < ncell; ++v)
                acc += dM[(g0 + u * ncell + v) * CFFM_WA + i] * bil_tap(wsg * u + a, i / 7) * bil_tap(wsg * v + b, i % 7);
    }
    acc = wave_sum(acc);
    if (i == 0) gw.w[grp][k] = acc;
    // The seven gradient tensors of a block whose length is not a multiple of 4 floats -- the pooling weights of 49 / 49 / 9 elements
    // and the four scalar pooling biases -- all belong to the pooling Linears.  When the caller keeps every gradient in a slice padded
    // to 16 bytes of one flat buffer (cffm_grad_slices_padded: vss_cffm_amd.ops does, the data-parallel exchange all-reduces that
    // buffer whole), this launch also zeroes the 3 floats behind each of them, so recycled memory never travels through a collective
    // as NaN / Inf bit patterns and no separate fill kernel sits in front of the backward.
    if (t == 0 && i < 3 && gw.b[0]) {
        gw.w[0][49 + i] = 0.f; gw.w[1][49 + i] = 0.f; gw.w[2][9 + i] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) gw.b[q][1 + i] = 0.f;
    }
}
__global__ void __launch_bounds__(64) k_pool_matrix_bwd(const float* __restrict__ dM, PoolWG gw) { pool_matrix_bwd_body(dM, gw, blockIdx.x); }
// the same for up to PMB_MAXD blocks in one launch (grid (111, n)): the end of a layer backward
#define PMB_MAXD 4
struct PoolWGN { const float* dM[PMB_MAXD]; PoolWG gw[PMB_MAXD]; };
__global__ void __launch_bounds__(64) k_pool_matrix_bwd_n(PoolWGN a) { pool_matrix_bwd_body(a.dM[blockIdx.y], a.gw[blockIdx.y], blockIdx.x); }

// --------------------------------------------------------------------------- geometry helpers
struct Geo {
    int B, H0, W0, Hp, Wp, gy, gx, nW, HW, RC;  // RC = 64*nW rows per clip in the token-row space
};
// token-row space of one clip: [0,49nW) target tokens window-major | [49nW,50nW) P0 | [50nW,51nW) f0
// | [51nW,55nW) f1 (2gy x 2gx) | [55nW,64nW) f2 (3gy x 3gx)
__device__ __forceinline__ int cell_row(const Geo& G, int wy, int wx, int g) {
    const int w = wy * G.gx + wx;
    if (g == 0) return 49 * G.nW + w;
    if (g == 1) return 50 * G.nW + w;
    if (g < 6) { int u = (g - 2) >> 1, v = (g - 2) & 1; return 51 * G.nW + (2 * wy + u) * (2 * G.gx) + 2 * wx + v; }
    int u = (g - 6) / 3, v = (g - 6) % 3;
    return 55 * G.nW + (3 * wy + u) * (3 * G.gx) + 3 * wx + v;
}
__device__ __forceinline__ void frame_cells(int frame, int& g0, int& ncell) {
    if (frame == 3) { g0 = 0; ncell = 1; }
    else if (frame == 0) { g0 = 1; ncell = 1; }
    else if (frame == 1) { g0 = 2; ncell = 4; }
    else { g0 = 6; ncell = 9; }
}

struct PoolB { const float* b[4]; };
struct PoolBG { float* b[4]; };
__device__ __forceinline__ int frame_group(int frame) { return frame == 3 ? 0 : frame + 1; }

// --------------------------------------------------------------------------- LN1 + pad + pool (forward)
// grid (nW, 4 frames, B), 256 threads.  x_ref [B,3,HW,C] (batch stride ref_bs), x_tgt [B,HW,C]
// (batch stride tgt_bs), both NHWC.  Writes zall rows (target tokens incl. zero rows of padded
// pixels; pooled rows incl. pool bias) and the per-pixel LayerNorm statistics.
#ifndef LNPF_BATCH
#define LNPF_BATCH 4
#endif
// Waves per workgroup of the two LN + pool kernels.  One workgroup per (window, frame, clip) is 648 workgroups at B = 2; with four
// waves (12-13 pixel rows each, every row a chain of wave reductions) the grid was 2.5 waves per SIMD and the kernels sat at
// s_waitcnt two thirds of the time (SQ_WAIT_ANY / SQ_WAVE_CYCLES = 0.67, profiles/r02_pmc_sq.txt); eight waves halve the chain
// per wave and double the waves in flight.
#ifndef LNP_WAVES
#define LNP_WAVES 8
#endif
#define LNP_THREADS (64 * LNP_WAVES)
#define LNP_PIX ((CFFM_WA + LNP_WAVES - 1) / LNP_WAVES)      // pixel rows per wave
// (six waves per SIMD = three workgroups per CU: all 648 workgroups of a B = 2 launch resident at once instead of 512 + a 136-workgroup
// second round -- 80 VGPRs with one spilled dword; 17.0 -> 15.2 us.  Four per CU would need 64 VGPRs: 87 spills)
__global__ void __launch_bounds__(LNP_THREADS, 6) k_ln_pool_fwd(Geo G, const float* __restrict__ x_ref, long ref_bs,
                                                      const float* __restrict__ x_tgt, long tgt_bs,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ M /* or NULL: built here from pw */, PoolW pw, PoolB pb,
                                                      float* __restrict__ zall, float* __restrict__ mean_out,
                                                      float* __restrict__ rstd_out, int split /* zall rows in split-4 storage */) {
    __shared__ float sM[9 * CFFM_WA];
    __shared__ float red[4][9][CFFM_C];              // (eight waves: the upper four add into the lower four's rows)
    const int w = blockIdx.x, frame = blockIdx.y, b = blockIdx.z;
    const int wy = w / G.gx, wx = w % G.gx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int g0, ncell;
    frame_cells(frame, g0, ncell);
    // (M == NULL: the first block of a layer forward, so that this kernel does not wait for the side-stream parameter prep)
    for (int e = threadIdx.x; e < ncell * CFFM_WA; e += LNP_THREADS) sM[e] = M ? M[g0 * CFFM_WA + e] : pool_matrix_entry(pw, g0 + e / CFFM_WA, e % CFFM_WA);
    __syncthreads();
    const float* xf = (frame == 3) ? x_tgt + (long)b * tgt_bs : x_ref + (long)b * ref_bs + (long)frame * G.HW * CFFM_C;
    const f32x4 gm = *(const f32x4*)(gamma + 4 * lane), bt = *(const f32x4*)(beta + 4 * lane);
    f32x4 acc[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the wave's LNP_PIX pixel rows (1 KiB each) go in batches of LNPF_BATCH, each batch requested before its first row is consumed
#pragma unroll
    for (int k0 = 0; k0 < LNP_PIX; k0 += LNPF_BATCH) {
        f32x4 xr[LNPF_BATCH];
#pragma unroll
        for (int kk = 0; kk < LNPF_BATCH; ++kk) {
            const int i = wave + LNP_WAVES * (k0 + kk);
            const int y = 7 * wy + i / 7, x = 7 * wx + i % 7;
            xr[kk] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (k0 + kk < LNP_PIX && i < CFFM_WA && y < G.H0 && x < G.W0) xr[kk] = *(const f32x4*)(xf + ((long)y * G.W0 + x) * CFFM_C + 4 * lane);
        }
#pragma unroll
        for (int kk = 0; kk < LNPF_BATCH; ++kk) {
            const int i = wave + LNP_WAVES * (k0 + kk);
            if (k0 + kk >= LNP_PIX || i >= CFFM_WA) break;
            const int y = 7 * wy + i / 7, x = 7 * wx + i % 7;
            const bool valid = (y < G.H0) && (x < G.W0);  // wave-uniform
            f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (valid) {
                const long pix = (long)y * G.W0 + x;
                const f32x4 xv = xr[kk];
                const float mu = wave_sum(xv[0] + xv[1] + xv[2] + xv[3]) * (1.f / CFFM_C);
                const f32x4 d = xv - mu;
                const float var = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]) * (1.f / CFFM_C);
                const float rs = 1.f / sqrtf(var + CFFM_LN_EPS);
                z = d * rs * gm + bt;
                if (lane == 0) {
                    mean_out[((long)b * 4 + frame) * G.HW + pix] = mu;
                    rstd_out[((long)b * 4 + frame) * G.HW + pix] = rs;
                }
            }
            if (frame == 3) *(f32x4*)(zall + ((long)b * G.RC + w * CFFM_WA + i) * CFFM_C + 4 * lane) = split ? split4_pack(z) : z;
            if (valid) {
#pragma unroll
                for (int c = 0; c < 9; ++c)
                    if (c < ncell) acc[c] += sM[c * CFFM_WA + i] * z;
            }
        }
    }
    static_assert(LNP_WAVES == 4 || LNP_WAVES == 8, "LNP_WAVES");
    if (wave < 4) {
#pragma unroll
        for (int c = 0; c < 9; ++c)
            if (c < ncell) *(f32x4*)(&red[wave][c][4 * lane]) = acc[c];
    }
    __syncthreads();
    if (LNP_WAVES == 8) {
        if (wave >= 4) {
#pragma unroll
            for (int c = 0; c < 9; ++c)
                if (c < ncell) *(f32x4*)(&red[wave - 4][c][4 * lane]) += acc[c];
        }
        __syncthreads();
    }
    const float pbias = pb.b[frame_group(frame)][0];
    if (!split) {
        for (int c = 0; c < ncell && threadIdx.x < CFFM_C; ++c) {
            const int ch = threadIdx.x;
            const float s = red[0][c][ch] + red[1][c][ch] + red[2][c][ch] + red[3][c][ch] + pbias;
            zall[((long)b * G.RC + cell_row(G, wy, wx, g0 + c)) * CFFM_C + ch] = s;
        }
    } else {   // split-4 storage packs 4 channels into one 16-byte group: one thread per group
        for (int e = threadIdx.x; e < ncell * (CFFM_C / 4); e += LNP_THREADS) {
            const int c = e / (CFFM_C / 4), c4 = 4 * (e % (CFFM_C / 4));
            const f32x4 s = *(const f32x4*)(&red[0][c][c4]) + *(const f32x4*)(&red[1][c][c4]) + *(const f32x4*)(&red[2][c][c4]) +
                            *(const f32x4*)(&red[3][c][c4]) + pbias;
            *(f32x4*)(zall + ((long)b * G.RC + cell_row(G, wy, wx, g0 + c)) * CFFM_C + c4) = split4_pack(s);
        }
    }
}

// --------------------------------------------------------------------------- LN1 + pad + pool (backward)
// Round 5: the CFFA backward is TWO kernels (it was one per block over all four frames: 96 MB per launch, 38 us alone / 81 us beside
// the weight gradients, one workgroup per CU).
//   k_ln_pool_bwd_tgt  per block, on the chain: the target frame only -- dx_tgt for the next block (reads x, dz of the target tokens,
//                      the residual-path gradient; 30 MB at B = 2);
//   k_ln_pool_bwd_ref  ONCE per layer backward (per range of blocks): the three reference frames of EVERY block of the range.
//                      The reference frames pass through the layer unchanged (cffm_transformer.py:826), so every block normalises the
//                      same x with the same statistics (:716) and the LayerNorm backward is linear in its upstream gradient:
//                      dx_ref = LNbwd(x, sum_d gamma_d * dZ_d), with dZ_d = M_d^T dP_d expanded from block d's 14 pooled-cell
//                      gradients per window (:780-805).  One read of x_ref and ONE write of dx_ref per step instead of a read +
//                      read-modify-write per block; the per-block parameter gradients (norm1, pooling matrix, pool biases) come out
//                      of the same pass.
// Inputs: dzall (gradient of every token row of a block: target tokens + pooled cells), dres (residual-path gradient of the target
// frame, may be NULL).  Every workgroup leaves one RECORD of LNP_RSTRIDE floats per block: dgamma[256] | dbeta[256] | tail, the tail
// being dM[cell 0][49] | dpool_bias[0] for a target workgroup and dM[cells 1..14][49] | dpool_bias[1..3] for a reference workgroup;
// k_reduce_records_multi sums them (cffm_hip.hip cffa_reduce) -- deterministic, no atomics.
#define LNP_RSTRIDE 1216
#define LNP_TAIL_TGT (CFFM_WA + 1)
#define LNP_TAIL_REF (14 * CFFM_WA + 3)
// a window's 49 pixels are dealt out to LNB_SPLIT workgroups of LNB_WAVES waves: pixel i belongs to workgroup i % LNB_SPLIT and there
// to wave (i / LNB_SPLIT) % LNB_WAVES -- at most LNB_PIX pixels per wave, ALL of whose rows are requested before the first is consumed
#ifndef LNB_SPLIT
#define LNB_SPLIT 4
#endif
#ifndef LNB_WAVES
#define LNB_WAVES 4
#endif
#define LNB_THREADS (64 * LNB_WAVES)
#define LNB_PIX ((CFFM_WA + LNB_SPLIT * LNB_WAVES - 1) / (LNB_SPLIT * LNB_WAVES))

__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }

// grid (B * nW * LNB_SPLIT), LNB_THREADS.  rec: [B * nW * LNB_SPLIT][LNP_RSTRIDE] (columns [0, 512 + LNP_TAIL_TGT) are written)
__global__ void __launch_bounds__(LNB_THREADS) k_ln_pool_bwd_tgt(Geo G, const float* __restrict__ x_tgt, long tgt_bs,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const float* __restrict__ M, const float* __restrict__ mean_in,
                                                                 const float* __restrict__ rstd_in, const float* __restrict__ dzall,
                                                                 const float* __restrict__ dres, float* __restrict__ dx_tgt, long dtgt_bs,
                                                                 float* __restrict__ rec) {
    __shared__ float red[LNB_WAVES][2][CFFM_C];
    __shared__ float sdM[CFFM_WA];
    __shared__ float sbs;
    const int s = blockIdx.x % LNB_SPLIT, wb = blockIdx.x / LNB_SPLIT;
    const int w = wb % G.nW, b = wb / G.nW;
    const int wy = w / G.gx, wx = w % G.gx;
    const int lane = threadIdx.x & 63, wave = wave_uniform(threadIdx.x >> 6);
    const float* xf = x_tgt + (long)b * tgt_bs;
    float* dxf = dx_tgt + (long)b * dtgt_bs;
    if (threadIdx.x < CFFM_WA) sdM[threadIdx.x] = 0.f;
    // every global read of the wave's pixels is requested here: x rows, LN statistics, the target-token gradients, the residual row
    f32x4 xr[LNB_PIX], dzr[LNB_PIX], addr[LNB_PIX];
    float mur[LNB_PIX], rsr[LNB_PIX], m0[LNB_PIX];
#pragma unroll
    for (int k = 0; k < LNB_PIX; ++k) {
        const int i = s + LNB_SPLIT * (wave + LNB_WAVES * k);
        const int y = 7 * wy + i / 7, x = 7 * wx + i % 7;
        xr[k] = dzr[k] = addr[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mur[k] = rsr[k] = m0[k] = 0.f;
        if (i < CFFM_WA && y < G.H0 && x < G.W0) {
            const long pix = (long)y * G.W0 + x;
            xr[k] = *(const f32x4*)(xf + pix * CFFM_C + 4 * lane);
            mur[k] = mean_in[((long)b * 4 + 3) * G.HW + pix];
            rsr[k] = rstd_in[((long)b * 4 + 3) * G.HW + pix];
            m0[k] = M[i];
            dzr[k] = *(const f32x4*)(dzall + ((long)b * G.RC + w * CFFM_WA + i) * CFFM_C + 4 * lane);
            if (dres) addr[k] = *(const f32x4*)(dres + ((long)b * G.HW + pix) * CFFM_C + 4 * lane);
        }
    }
    const f32x4 dp = *(const f32x4*)(dzall + ((long)b * G.RC + cell_row(G, wy, wx, 0)) * CFFM_C + 4 * lane);
    const f32x4 gm = *(const f32x4*)(gamma + 4 * lane), bt = *(const f32x4*)(beta + 4 * lane);
    if (wave == 0) {
        const float bs = wave_sum(dp[0] + dp[1] + dp[2] + dp[3]);
        if (lane == 0) sbs = (s == 0) ? bs : 0.f;          // the window's pool-bias gradient is counted once
    }
    f32x4 ag = (f32x4){0.f, 0.f, 0.f, 0.f}, ab = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
#pragma unroll
    for (int k = 0; k < LNB_PIX; ++k) {
        const int i = s + LNB_SPLIT * (wave + LNB_WAVES * k);
        if (i >= CFFM_WA) break;
        const int y = 7 * wy + i / 7, x = 7 * wx + i % 7;
        if (!((y < G.H0) && (x < G.W0))) continue;      // padded pixel: z is the constant 0
        const long pix = (long)y * G.W0 + x;
        const float rs = rsr[k];
        const f32x4 xh = (xr[k] - mur[k]) * rs;
        const f32x4 z = xh * gm + bt;
        const f32x4 dz = dzr[k] + m0[k] * dp;
        const float dm = wave_sum_hi(dot4(dp, z));
        if (lane == 63) sdM[i] = dm;                    // pixel i belongs to exactly one wave of one workgroup
        ag += dz * xh;
        ab += dz;
        const f32x4 gz = dz * gm;
        const float m1 = wave_sum(gz[0] + gz[1] + gz[2] + gz[3]) * (1.f / CFFM_C);
        const float m2 = wave_sum(dot4(gz, xh)) * (1.f / CFFM_C);
        const f32x4 dx = (gz - m1 - xh * m2) * rs + addr[k];
        *(f32x4*)(dxf + pix * CFFM_C + 4 * lane) = dx;
    }
    *(f32x4*)(&red[wave][0][4 * lane]) = ag;
    *(f32x4*)(&red[wave][1][4 * lane]) = ab;
    __syncthreads();
    float* r = rec + (long)blockIdx.x * LNP_RSTRIDE;
    for (int e = threadIdx.x; e < 2 * CFFM_C; e += LNB_THREADS) {
        const int which = e / CFFM_C, ch = e % CFFM_C;
        float t = red[0][which][ch];
#pragma unroll
        for (int q = 1; q < LNB_WAVES; ++q) t += red[q][which][ch];
        r[e] = t;
    }
    if (threadIdx.x < CFFM_WA) r[2 * CFFM_C + threadIdx.x] = sdM[threadIdx.x];
    if (threadIdx.x == CFFM_WA) r[2 * CFFM_C + CFFM_WA] = sbs;
}

// The reference frames of up to RB_MAXD blocks in one pass.  grid (B * nW * LNB_SPLIT), LNB_THREADS; dynamic LDS: cffa_ref_lds(D).
#define RB_MAXD 4
struct CffaRefBlocks {
    int n;                                   // blocks in this launch (1..RB_MAXD), any order
    const float* gamma[RB_MAXD];
    const float* beta[RB_MAXD];
    const float* M[RB_MAXD];                 // the block's composed pooling matrix [15][49]
    const float* dzall[RB_MAXD];             // the block's token-row gradient [B * RC][256] (its pooled-cell rows are read)
    float* rec[RB_MAXD];                     // [B * nW * LNB_SPLIT][LNP_RSTRIDE] (columns [0, 512 + LNP_TAIL_REF) are written)
};
// floats of LDS per block: pooled-cell gradients of the 14 reference cells | their pooling-matrix rows | the dM rows of this workgroup
#define RB_LDS_BLOCK (14 * CFFM_C + 2 * 14 * CFFM_WA)
__host__ __device__ constexpr int cffa_ref_lds(int D) {
    return 4 * ((D * RB_LDS_BLOCK > LNB_WAVES * D * 2 * CFFM_C + 14 * D * CFFM_WA ? D * RB_LDS_BLOCK : LNB_WAVES * D * 2 * CFFM_C + 14 * D * CFFM_WA) + 4 * D + 4);
}

// the rows of one reference frame a wave works on: requested as a whole (ln_pool_ref_load), consumed a frame later
struct RefRows { f32x4 x[LNB_PIX], add[LNB_PIX]; float mu[LNB_PIX], rs[LNB_PIX]; };
template <bool ACC>
__device__ __forceinline__ void ln_pool_ref_load(RefRows& R, const Geo& G, const float* __restrict__ xf, const float* __restrict__ dxf,
                                                 const float* __restrict__ mean_f, const float* __restrict__ rstd_f, int s, int wy, int wx,
                                                 int lane, int wave) {
#pragma unroll
    for (int k = 0; k < LNB_PIX; ++k) {
        const int i = s + LNB_SPLIT * (wave + LNB_WAVES * k);
        const int y = 7 * wy + i / 7, x = 7 * wx + i % 7;
        R.x[k] = R.add[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        R.mu[k] = R.rs[k] = 0.f;
        if (i < CFFM_WA && y < G.H0 && x < G.W0) {
            const long pix = (long)y * G.W0 + x;
            R.x[k] = *(const f32x4*)(xf + pix * CFFM_C + 4 * lane);
            R.mu[k] = mean_f[pix];
            R.rs[k] = rstd_f[pix];
            if (ACC) R.add[k] = *(const f32x4*)(dxf + pix * CFFM_C + 4 * lane);
        }
    }
}
// one reference frame (NC cells per window, first cell g0 >= 1) of the workgroup's pixels, for all D blocks
template <int NC, int D>
__device__ __forceinline__ void ln_pool_bwd_ref_frame(const Geo& G, const RefRows& R, float* __restrict__ dxf,
                                                      const float* sP, const float* sMm, float* sdM, const f32x4 (&gm)[D], const f32x4 (&bt)[D],
                                                      f32x4 (&ag)[D], f32x4 (&ab)[D], int g0, int s, int wy, int wx, int lane, int wave) {
#pragma unroll
    for (int k = 0; k < LNB_PIX; ++k) {
        const int i = s + LNB_SPLIT * (wave + LNB_WAVES * k);
        if (i >= CFFM_WA) break;
        const int y = 7 * wy + i / 7, x = 7 * wx + i % 7;
        if (!((y < G.H0) && (x < G.W0))) continue;
        const long pix = (long)y * G.W0 + x;
        const float rs = R.rs[k];
        const f32x4 xh = (R.x[k] - R.mu[k]) * rs;
        f32x4 gz = (f32x4){0.f, 0.f, 0.f, 0.f};
        // The cells that can see pixel (iy, ix): the composed pooling matrix is SPARSE in the strided frames.  A cell (u, v) pools the
        // resized rows wsg u .. wsg u + wsg - 1, a resized row r taps pixel rows r and r + 1 (bil_tap), so pixel row iy reaches the cells
        // u = r / wsg of r in {iy - 1, iy} ^ [0, 5]: one or two per dimension -- on average 1.65 of the 9 cells of frame t-3 and 1.3 of
        // the 4 of frame t-6.  Everywhere else M[c][i] is an exact zero: no share in dz, and the dM entry is one that the pooling-matrix
        // backward multiplies by a zero tap (k_pool_matrix_bwd) -- it stays 0.  (Round 5, second form: all NC cells per pixel cost
        // 14 dot products + wave reductions per block and pixel triple, 48 us for the launch.)
        constexpr int NCD = NC == 9 ? 3 : (NC == 4 ? 2 : 1), WSG = NC == 9 ? 2 : (NC == 4 ? 3 : 7);
        const int iy = i / 7, ix = i % 7;
        int u0 = 0, u1 = 0, v0 = 0, v1 = 0;
        if (NC > 1) {
            u0 = (iy > 0 ? iy - 1 : 0) / WSG; u1 = (iy < 5 ? iy : 5) / WSG;
            v0 = (ix > 0 ? ix - 1 : 0) / WSG; v1 = (ix < 5 ? ix : 5) / WSG;
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const f32x4 z = xh * gm[d] + bt[d];
            f32x4 dz = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int u = u0; u <= u1; ++u)          // (wave-uniform bounds)
                for (int v = v0; v <= v1; ++v) {
                    const int c = d * 14 + g0 - 1 + u * NCD + v;
                    const f32x4 dp = *(const f32x4*)(sP + c * CFFM_C + 4 * lane);
                    dz += sMm[c * CFFM_WA + i] * dp;
                    const float dm = wave_sum_hi(dot4(dp, z));
                    if (lane == 63) sdM[c * CFFM_WA + i] = dm;
                }
            ag[d] += dz * xh;
            ab[d] += dz;
            gz += dz * gm[d];
        }
        const float m1 = wave_sum(gz[0] + gz[1] + gz[2] + gz[3]) * (1.f / CFFM_C);
        const float m2 = wave_sum(dot4(gz, xh)) * (1.f / CFFM_C);
        const f32x4 dx = (gz - m1 - xh * m2) * rs + R.add[k];
        *(f32x4*)(dxf + pix * CFFM_C + 4 * lane) = dx;
    }
}

template <int D, bool ACC>
__global__ void __launch_bounds__(LNB_THREADS) k_ln_pool_bwd_ref(Geo G, const float* __restrict__ x_ref, long ref_bs, CffaRefBlocks Bk,
                                                                 const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                                 float* __restrict__ dx_ref, long dref_bs) {
    CFFM_DYN_SMEM(smem);
    float* sP = (float*)smem;                             // [D][14][256]
    float* sMm = sP + D * 14 * CFFM_C;                    // [D][14][49]
    float* red = (float*)smem;                            // [LNB_WAVES][D][2][256], reuses sP / sMm behind the pixel loops
    constexpr int body = D * (14 * CFFM_C + 14 * CFFM_WA), redn = LNB_WAVES * D * 2 * CFFM_C;
    float* sdM = (float*)smem + (body > redn ? body : redn);   // [D][14][49]
    float* sbs = sdM + D * 14 * CFFM_WA;                  // [D][4]: pool-bias gradients of the three reference groups
    const int s = blockIdx.x % LNB_SPLIT, wb = blockIdx.x / LNB_SPLIT;
    const int w = wb % G.nW, b = wb / G.nW;
    const int wy = w / G.gx, wx = w % G.gx;
    const int lane = threadIdx.x & 63, wave = wave_uniform(threadIdx.x >> 6);
    for (int e = threadIdx.x; e < D * 14 * CFFM_WA; e += LNB_THREADS) {
        const int d = e / (14 * CFFM_WA), r = e % (14 * CFFM_WA);
        sMm[e] = Bk.M[d][CFFM_WA + r];
        sdM[e] = 0.f;
    }
    // the 14 pooled-cell gradient rows of the window, per block (1 KiB rows, one wave per row)
    for (int r = wave; r < D * 14; r += LNB_WAVES) {
        const int d = r / 14, c = r % 14;
        *(f32x4*)(sP + r * CFFM_C + 4 * lane) = *(const f32x4*)(Bk.dzall[d] + ((long)b * G.RC + cell_row(G, wy, wx, 1 + c)) * CFFM_C + 4 * lane);
    }
    f32x4 gm[D], bt[D], ag[D], ab[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        gm[d] = *(const f32x4*)(Bk.gamma[d] + 4 * lane);
        bt[d] = *(const f32x4*)(Bk.beta[d] + 4 * lane);
        ag[d] = ab[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    if (s == 0) {     // pool-bias gradients: sums over the group's cells and channels, counted once per window
        for (int q = wave; q < 3 * D; q += LNB_WAVES) {
            const int d = q / 3, grp = q % 3;
            const int c0 = grp == 0 ? 0 : (grp == 1 ? 1 : 5), nc = grp == 0 ? 1 : (grp == 1 ? 4 : 9);
            float t = 0.f;
            for (int c = 0; c < nc; ++c) {
                const f32x4 v = *(const f32x4*)(sP + (d * 14 + c0 + c) * CFFM_C + 4 * lane);
                t += v[0] + v[1] + v[2] + v[3];
            }
            t = wave_sum(t);
            if (lane == 0) sbs[d * 4 + grp] = t;
        }
    } else if (threadIdx.x < 4 * D) sbs[threadIdx.x] = 0.f;
    const float* xb = x_ref + (long)b * ref_bs;
    float* dxb = dx_ref + (long)b * dref_bs;
    const long fr = (long)G.HW * CFFM_C;
    const float* mb = mean_in + (long)b * 4 * G.HW;
    const float* rb = rstd_in + (long)b * 4 * G.HW;
    // the rows of frame f + 1 are requested before frame f is worked on
    RefRows Ra, Rb;
    ln_pool_ref_load<ACC>(Ra, G, xb, dxb, mb, rb, s, wy, wx, lane, wave);
    ln_pool_ref_load<ACC>(Rb, G, xb + fr, dxb + fr, mb + G.HW, rb + G.HW, s, wy, wx, lane, wave);
    ln_pool_bwd_ref_frame<1, D>(G, Ra, dxb, sP, sMm, sdM, gm, bt, ag, ab, 1, s, wy, wx, lane, wave);
    ln_pool_ref_load<ACC>(Ra, G, xb + 2 * fr, dxb + 2 * fr, mb + 2 * G.HW, rb + 2 * G.HW, s, wy, wx, lane, wave);
    ln_pool_bwd_ref_frame<4, D>(G, Rb, dxb + fr, sP, sMm, sdM, gm, bt, ag, ab, 2, s, wy, wx, lane, wave);
    ln_pool_bwd_ref_frame<9, D>(G, Ra, dxb + 2 * fr, sP, sMm, sdM, gm, bt, ag, ab, 6, s, wy, wx, lane, wave);
    __syncthreads();          // every wave is done with sP / sMm: red may overwrite them
#pragma unroll
    for (int d = 0; d < D; ++d) {
        *(f32x4*)(red + ((wave * D + d) * 2 + 0) * CFFM_C + 4 * lane) = ag[d];
        *(f32x4*)(red + ((wave * D + d) * 2 + 1) * CFFM_C + 4 * lane) = ab[d];
    }
    __syncthreads();
    for (int d = 0; d < D; ++d) {
        float* r = Bk.rec[d] + (long)blockIdx.x * LNP_RSTRIDE;
        for (int e = threadIdx.x; e < 2 * CFFM_C; e += LNB_THREADS) {
            float t = red[(d * 2) * CFFM_C + e];
#pragma unroll
            for (int q = 1; q < LNB_WAVES; ++q) t += red[((q * D + d) * 2) * CFFM_C + e];
            r[e] = t;
        }
        for (int e = threadIdx.x; e < LNP_TAIL_REF; e += LNB_THREADS)
            r[2 * CFFM_C + e] = e < 14 * CFFM_WA ? sdM[d * 14 * CFFM_WA + e] : sbs[d * 4 + (e - 14 * CFFM_WA)];
    }
}
