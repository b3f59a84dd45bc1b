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
__global__ void __launch_bounds__(64) k_pool_matrix_bwd(const float* __restrict__ dM, PoolWG gw) {
    const int t = blockIdx.x, i = threadIdx.x;
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
// Inputs: dzall (gradient of every token row: target tokens + pooled cells), dres (residual-path gradient
// of the target frame, may be NULL).  Outputs: dx of the frame (NHWC; `accum_ref` adds to what is there) and
// one partial record per workgroup, part[blk][LNP_REC] = dgamma[256] | dbeta[256] | dM[15*49] | dpool_bias[4]
// (zeros outside this frame's cells); k_reduce_partials sums the records -- no contended atomics.
#define LNP_REC (2 * CFFM_C + CFFM_NCELL * CFFM_WA + 4)
#ifndef LNPB_BATCH
#define LNPB_BATCH 4
#endif
#ifndef LNPB_ABLATE
#define LNPB_ABLATE 0   // profiling builds only: 1 no dM reductions, 2 no LN reductions, 4 no dx stores
#endif
// body for a frame with NC pooled cells per window (1, 4 or 9: compile-time, so the per-cell loops unroll and their LDS
// reads / wave reductions overlap instead of queueing behind each other)
template <int NC>
__device__ __forceinline__ void ln_pool_bwd_body(const Geo& G, const float* __restrict__ xf, float* __restrict__ dxf, bool accum,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                 const float* __restrict__ M, const float* __restrict__ mean_in,
                                                 const float* __restrict__ rstd_in, const float* __restrict__ dzall,
                                                 const float* __restrict__ dres, float* __restrict__ part, int g0, int w, int frame, int b) {
    __shared__ float sM[NC * CFFM_WA];
    __shared__ float sdP[NC][CFFM_C];
    __shared__ float sdM[NC * CFFM_WA];
    __shared__ float red[LNP_WAVES][2][CFFM_C];
    __shared__ float sbs[LNP_WAVES];
    const int wy = w / G.gx, wx = w % G.gx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < NC * CFFM_WA; e += LNP_THREADS) { sM[e] = M[g0 * CFFM_WA + e]; sdM[e] = 0.f; }
    float bsum = 0.f;
    for (int e = threadIdx.x; e < NC * CFFM_C; e += LNP_THREADS) {
        const int c = e / CFFM_C, ch = e % CFFM_C;
        const float v = dzall[((long)b * G.RC + cell_row(G, wy, wx, g0 + c)) * CFFM_C + ch];
        sdP[c][ch] = v;
        bsum += v;
    }
    bsum = wave_sum(bsum);
    if (lane == 0) sbs[wave] = bsum;
    const f32x4 gm = *(const f32x4*)(gamma + 4 * lane), bt = *(const f32x4*)(beta + 4 * lane);
    f32x4 ag = (f32x4){0.f, 0.f, 0.f, 0.f}, ab = (f32x4){0.f, 0.f, 0.f, 0.f};
    // The wave's LNP_PIX pixels (12-13 with four waves, 6-7 with eight) go in batches of LNPB_BATCH = 4 (measured: 13 at once 47 us, 7: 42, 5: 34, 4: 30.5, 3: 30.8).  ALL global reads of a batch are requested before its first pixel
    // is consumed: x rows, LN statistics, the target-token gradients, and the row that is added to the result (the
    // residual-path gradient of the target frame, or the dx a later block already accumulated for a reference frame).
    // Read inside the pixel loop that last row would sit between stores to the same array, where the compiler cannot hoist
    // it -- one exposed memory round trip per pixel.  Small batches keep the kernel at three workgroups per CU.
    __syncthreads();
#pragma unroll
    for (int k0 = 0; k0 < LNP_PIX; k0 += LNPB_BATCH) {
        f32x4 xr[LNPB_BATCH], dzr[LNPB_BATCH], addr[LNPB_BATCH];
        float mur[LNPB_BATCH], rsr[LNPB_BATCH];
#pragma unroll
        for (int kk = 0; kk < LNPB_BATCH; ++kk) {
            const int i = wave + LNP_WAVES * (k0 + kk);
            const int y = 7 * wy + i / 7, x = 7 * wx + i % 7;
            xr[kk] = dzr[kk] = addr[kk] = (f32x4){0.f, 0.f, 0.f, 0.f};
            mur[kk] = rsr[kk] = 0.f;
            if (k0 + kk < LNP_PIX && i < CFFM_WA && y < G.H0 && x < G.W0) {
                const long pix = (long)y * G.W0 + x;
                xr[kk] = *(const f32x4*)(xf + pix * CFFM_C + 4 * lane);
                mur[kk] = mean_in[((long)b * 4 + frame) * G.HW + pix];
                rsr[kk] = rstd_in[((long)b * 4 + frame) * G.HW + pix];
                if (frame == 3) {
                    dzr[kk] = *(const f32x4*)(dzall + ((long)b * G.RC + w * CFFM_WA + i) * CFFM_C + 4 * lane);
                    if (dres) addr[kk] = *(const f32x4*)(dres + ((long)b * G.HW + pix) * CFFM_C + 4 * lane);
                } else if (accum) {
                    addr[kk] = *(const f32x4*)(dxf + pix * CFFM_C + 4 * lane);
                }
            }
        }
#pragma unroll
        for (int kk = 0; kk < LNPB_BATCH; ++kk) {
            const int i = wave + LNP_WAVES * (k0 + kk);
            if (k0 + kk >= LNP_PIX || i >= CFFM_WA) break;
            const int y = 7 * wy + i / 7, x = 7 * wx + i % 7;
            if (!((y < G.H0) && (x < G.W0))) continue;  // padded pixel: z is the constant 0
            const long pix = (long)y * G.W0 + x;
            const float rs = rsr[kk];
            const f32x4 xh = (xr[kk] - mur[kk]) * rs;
            const f32x4 z = xh * gm + bt;
            f32x4 dz = dzr[kk];
            float dm[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const f32x4 dp = *(const f32x4*)(&sdP[c][4 * lane]);
                dz += sM[c * CFFM_WA + i] * dp;
                dm[c] = dp[0] * z[0] + dp[1] * z[1] + dp[2] * z[2] + dp[3] * z[3];
            }
#if !(LNPB_ABLATE & 1)
#pragma unroll
            for (int c = 0; c < NC; ++c) dm[c] = wave_sum(dm[c]);   // independent chains: they overlap
            if (lane == 0)
#pragma unroll
                for (int c = 0; c < NC; ++c) sdM[c * CFFM_WA + i] = dm[c];   // pixel i belongs to exactly one wave: no conflict
#endif
            ag += dz * xh;
            ab += dz;
            const f32x4 gz = dz * gm;
#if LNPB_ABLATE & 2
            const float m1 = gz[0], m2 = gz[1];
#else
            const float m1 = wave_sum(gz[0] + gz[1] + gz[2] + gz[3]) * (1.f / CFFM_C);
            const float m2 = wave_sum(gz[0] * xh[0] + gz[1] * xh[1] + gz[2] * xh[2] + gz[3] * xh[3]) * (1.f / CFFM_C);
#endif
            const f32x4 dx = (gz - m1 - xh * m2) * rs + addr[kk];
#if LNPB_ABLATE & 4
            ag += dx;
#else
            *(f32x4*)(dxf + pix * CFFM_C + 4 * lane) = dx;
#endif
        }
    }
    *(f32x4*)(&red[wave][0][4 * lane]) = ag;
    *(f32x4*)(&red[wave][1][4 * lane]) = ab;
    __syncthreads();
    float* rec = part + ((long)(b * 4 + frame) * G.nW + w) * LNP_REC;
    for (int e = threadIdx.x; e < 2 * CFFM_C; e += LNP_THREADS) {
        const int which = e / CFFM_C, ch = e % CFFM_C;
        float t = (red[0][which][ch] + red[1][which][ch]) + (red[2][which][ch] + red[3][which][ch]);
        if (LNP_WAVES == 8) t += (red[4][which][ch] + red[5][which][ch]) + (red[6][which][ch] + red[7][which][ch]);
        rec[which * CFFM_C + ch] = t;
    }
    for (int e = threadIdx.x; e < CFFM_NCELL * CFFM_WA + 4; e += LNP_THREADS) {
        float v = 0.f;
        if (e < CFFM_NCELL * CFFM_WA) {
            if (e >= g0 * CFFM_WA && e < (g0 + NC) * CFFM_WA) v = sdM[e - g0 * CFFM_WA];
        } else if (e - CFFM_NCELL * CFFM_WA == frame_group(frame)) {
            v = sbs[0] + sbs[1] + sbs[2] + sbs[3];
            if (LNP_WAVES == 8) v += sbs[4] + sbs[5] + sbs[6] + sbs[7];
        }
        rec[2 * CFFM_C + e] = v;
    }
}
// (one workgroup per CU at 140 VGPRs and 69 KB of LDS -- the three instantiations' arrays add up; sharing one set through a struct
// measured 46 -> 55 us, capping the registers at 128 / 96 spills: 56 / 79 us)
__global__ void __launch_bounds__(LNP_THREADS) k_ln_pool_bwd(Geo G, const float* __restrict__ x_ref, long ref_bs,
                                                      const float* __restrict__ x_tgt, long tgt_bs,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ M, const float* __restrict__ mean_in,
                                                      const float* __restrict__ rstd_in, const float* __restrict__ dzall,
                                                      const float* __restrict__ dres,
                                                      float* __restrict__ dx_ref, long dref_bs, int accum_ref,
                                                      float* __restrict__ dx_tgt, long dtgt_bs, float* __restrict__ part) {
    const int w = blockIdx.x, frame = blockIdx.y, b = blockIdx.z;
    int g0, ncell;
    frame_cells(frame, g0, ncell);
    const float* xf = (frame == 3) ? x_tgt + (long)b * tgt_bs : x_ref + (long)b * ref_bs + (long)frame * G.HW * CFFM_C;
    float* dxf = (frame == 3) ? dx_tgt + (long)b * dtgt_bs : dx_ref + (long)b * dref_bs + (long)frame * G.HW * CFFM_C;
    const bool accum = (frame == 3) ? false : (accum_ref != 0);
    if (ncell == 1) ln_pool_bwd_body<1>(G, xf, dxf, accum, gamma, beta, M, mean_in, rstd_in, dzall, dres, part, g0, w, frame, b);
    else if (ncell == 4) ln_pool_bwd_body<4>(G, xf, dxf, accum, gamma, beta, M, mean_in, rstd_in, dzall, dres, part, g0, w, frame, b);
    else ln_pool_bwd_body<9>(G, xf, dxf, accum, gamma, beta, M, mean_in, rstd_in, dzall, dres, part, g0, w, frame, b);
}

