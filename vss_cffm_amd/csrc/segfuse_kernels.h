// segfuse_kernels.h -- the SegFormer embedding in front of the CFFM hot path, without the 1024-channel concat
// (reference: cffm_head.py:102-119: four `MLP` embeddings, three bilinear resizes to the 1/4 map, torch.cat, the 1x1
// `linear_fuse` conv).  A 1x1 conv over a concat is a sum of per-scale 1x1 convs, and a bilinear resize acts per channel, so
//     conv(cat_i up_i(W_i c_i + b_i)) = sum_i up_i((Wf_i W_i) c_i) + sum_i Wf_i b_i          (resize weights sum to 1)
// i.e. each scale is embedded ONCE at its own resolution with the composed [256 x C_i] matrix (Linear GEMMs of gemm.h on
// token rows), and the only full-resolution work left is the kernel below: one pass that adds the three resized maps and
// the constant to the 1/4-scale embedding, in place.  The [N,1024,H,W] concat (59 MB per frame at 480x480) and the
// 7.5 GFLOP/frame 1024->256 conv over it never exist.
//
// Everything here works on token rows [N*H*W, 256] (NHWC): one wave = one row of 256 channels as 64 x f32x4.
#pragma once
#include "cffm_common.h"
#include "gemm_kernels.h"   // xcd_linear_id

#define SEGF_C 256
#define SEGF_MAX_RATIO 16   // largest resize factor per dimension the adjoint's candidate window is sized for

// F.interpolate(mode='bilinear', align_corners=False) source taps of one output index (ATen UpSample.h
// area_pixel_compute_source_index + the i1 / lambda rule of upsample_bilinear2d): dst in [0,out) -> rows i0, i1 and the
// weight l1 of i1 (l0 = 1 - l1).
__device__ __forceinline__ void segf_taps(int dst, int in, int out, int& i0, int& i1, float& l1) {
    const float scale = (float)in / (float)out;
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i0 = i0 > in - 1 ? in - 1 : i0;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

#define SEGF_MAX_BANDS 32
struct SegfMaps {
    const float* z[3];   // low-resolution embeddings [N, h*w, 256]
    float* dz[3];        // (backward) their gradients
    int h[3], w[3];
    int cnt;             // maps in use (the SegFormer decoder: 3)
    // (backward) work order inside one frame: bands of output rows, inside a band every map's low-resolution rows whose centre
    // falls into it -- so the items that read the same rows of g, whatever their map, are neighbours in the launch order
    int nseg;                          // bands * cnt
    int seg_end[SEGF_MAX_BANDS * 3];   // exclusive prefix of items (workgroups) per (band, map)
    int seg_q0[SEGF_MAX_BANDS * 3];    // first low-resolution row of the segment
    int per_frame;                     // items per frame
};

// weights of two taps (offsets oa, ob in 0..2 from the neighbourhood's first row) as a dense triple -- selects, not indexing
#define SEGF_W3(dst, oa, va, ob, vb)                                                  \
    do {                                                                              \
        dst[0] = ((oa) == 0 ? (va) : 0.f) + ((ob) == 0 ? (vb) : 0.f);                  \
        dst[1] = ((oa) == 1 ? (va) : 0.f) + ((ob) == 1 ? (vb) : 0.f);                  \
        dst[2] = ((oa) == 2 ? (va) : 0.f) + ((ob) == 2 ? (vb) : 0.f);                  \
    } while (0)
// y[n, p, :] += d + sum_m bilinear(z_m)[n, p, :]     (y holds the 1/4-scale embedding on entry)
// One wave owns a 2 x 2 patch of output pixels: their taps of a low-resolution map lie in a 3 x 3 neighbourhood (usually
// 2 x 2), which is loaded once for the four pixels -- ~5 row loads per pixel instead of 13 (one wave per pixel was bound by
// the per-CU L1 rate of those loads: 79 us).  Rows / columns of the neighbourhood no pixel taps are skipped (wave-uniform).
// grid (ceil(N * ceil(H/2) * ceil(W/2) / 4)); every XCD (workgroup b runs on XCD b % 8) owns a contiguous run of patches.
__global__ void __launch_bounds__(256) k_segfuse_fwd(float* __restrict__ y, const float* __restrict__ d, SegfMaps mp, int N, int H, int W) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ph = (H + 1) / 2, pw = (W + 1) / 2;
    const long patch = (long)xcd_linear_id() * 4 + wave;
    if (patch >= (long)N * ph * pw) return;
    const int n = (int)(patch / ((long)ph * pw)), pr = (int)(patch - (long)n * ph * pw), oy = (pr / pw) * 2, ox = (pr - (pr / pw) * pw) * 2;
    const bool vy = oy + 1 < H, vx = ox + 1 < W;        // second row / column of the patch inside the map
    float* yb = y + ((long)n * H * W + (long)oy * W + ox) * SEGF_C;
    const f32x4 dv = ((const f32x4*)d)[lane];
    f32x4 acc[2][2];
    acc[0][0] = ((const f32x4*)yb)[lane] + dv;
    acc[0][1] = vx ? ((const f32x4*)(yb + SEGF_C))[lane] + dv : dv;
    acc[1][0] = vy ? ((const f32x4*)(yb + (long)W * SEGF_C))[lane] + dv : dv;
    acc[1][1] = (vy && vx) ? ((const f32x4*)(yb + (long)(W + 1) * SEGF_C))[lane] + dv : dv;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        if (m >= mp.cnt) continue;
        const int h = mp.h[m], w = mp.w[m];
        // per patch row / column: weights over the neighbourhood rows ya..ya+2 / columns xa..xa+2
        float wy[2][3], wx[2][3];
        int ya, xa;
        {
            int i0, i1, o0, o1;
            float l1;
            segf_taps(oy, h, H, i0, i1, l1);
            ya = i0; o1 = i1 - ya;
            SEGF_W3(wy[0], 0, 1.f - l1, o1, l1);
            segf_taps(vy ? oy + 1 : oy, h, H, i0, i1, l1);
            o0 = i0 - ya; o1 = i1 - ya;
            SEGF_W3(wy[1], o0, 1.f - l1, o1, l1);
            segf_taps(ox, w, W, i0, i1, l1);
            xa = i0; o1 = i1 - xa;
            SEGF_W3(wx[0], 0, 1.f - l1, o1, l1);
            segf_taps(vx ? ox + 1 : ox, w, W, i0, i1, l1);
            o0 = i0 - xa; o1 = i1 - xa;
            SEGF_W3(wx[1], o0, 1.f - l1, o1, l1);
        }
        const float* base = mp.z[m] + (long)n * h * w * SEGF_C;
        f32x4 t[3][3];
        bool use[3][3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                use[j][i] = (wy[0][j] != 0.f || wy[1][j] != 0.f) && (wx[0][i] != 0.f || wx[1][i] != 0.f);
                const int yy = ya + j < h ? ya + j : h - 1, xx = xa + i < w ? xa + i : w - 1;
                t[j][i] = use[j][i] ? ((const f32x4*)(base + ((long)yy * w + xx) * SEGF_C))[lane] : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const f32x4 rowv = t[j][0] * wx[b][0] + t[j][1] * wx[b][1] + t[j][2] * wx[b][2];
                    sum += rowv * wy[a][j];
                }
                acc[a][b] += sum;
            }
    }
    ((f32x4*)yb)[lane] = acc[0][0];
    if (vx) ((f32x4*)(yb + SEGF_C))[lane] = acc[0][1];
    if (vy) ((f32x4*)(yb + (long)W * SEGF_C))[lane] = acc[1][0];
    if (vy && vx) ((f32x4*)(yb + (long)(W + 1) * SEGF_C))[lane] = acc[1][1];
}

// Adjoint of the three resizes: dz_m[n, q, :] = sum over the output pixels p that tap q of weight(p, q) * g[n, p, :].
// Gather form (deterministic, no atomics).  One workgroup owns SEGF_XQ consecutive low-resolution pixels of one row:
// adjacent pixels share half of their output columns, so the union window is read once (g re-read 2.5x per map instead
// of 4x); its four waves take the candidate output rows round-robin and their partial sums meet in LDS -- a wave per item
// left the 8x map's waves with ~700 dependent-latency row loads each while the 2x map's had 56 (152 us, tail-bound).
// The candidate output rows / columns are the inverse image of (q-1, q+1) under the source-index map, widened by one, and
// every candidate's weight is recomputed with the forward's own tap rule (a candidate that does not tap q gets weight 0),
// so forward and adjoint cannot disagree.  grid = N * sum over maps of h*ceil(w/SEGF_XQ).
#define SEGF_XQ 4
#define SEGF_XWIN 96   // >= (SEGF_XQ + 1) * SEGF_MAX_RATIO + 5
__global__ void __launch_bounds__(256) k_segfuse_bwd(const float* __restrict__ g, SegfMaps mp, int N, int H, int W) {
    __shared__ float s_wy[64];
    __shared__ f32x4 s_wx[SEGF_XWIN];
    __shared__ f32x4 s_part[4][SEGF_XQ][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // XCD-aware order: workgroup b runs on XCD b % 8; re-numbered so that every XCD owns a contiguous run of the band-ordered
    // items -- the items that share rows of g then meet in one 4 MB L2 instead of eight, and g comes from HBM about once
    // (per-map raster order: 154 us; that order with contiguous XCD runs: 268 us, the 8x map's heavy items all on one XCD)
    const int lin = xcd_linear_id();
    const int n = lin / mp.per_frame, rem = lin - n * mp.per_frame;
    int sg = 0;
    while (sg + 1 < mp.nseg && rem >= mp.seg_end[sg]) ++sg;
    const int m = sg % mp.cnt, local = rem - (sg ? mp.seg_end[sg - 1] : 0);
    const int h = mp.h[m], w = mp.w[m], wg = (w + SEGF_XQ - 1) / SEGF_XQ;
    const int qy = mp.seg_q0[sg] + local / wg, qx0 = (local % wg) * SEGF_XQ;
    const int qx1 = qx0 + SEGF_XQ - 1 < w - 1 ? qx0 + SEGF_XQ - 1 : w - 1;
    // candidate windows
    const float isy = (float)H / (float)h, isx = (float)W / (float)w;
    int ylo = (int)floorf(((float)qy - 0.5f) * isy - 0.5f) - 1, yhi = (int)ceilf(((float)qy + 1.5f) * isy - 0.5f) + 1;
    int xlo = (int)floorf(((float)qx0 - 0.5f) * isx - 0.5f) - 1, xhi = (int)ceilf(((float)qx1 + 1.5f) * isx - 0.5f) + 1;
    ylo = ylo < 0 ? 0 : ylo; xlo = xlo < 0 ? 0 : xlo;
    yhi = yhi > H - 1 ? H - 1 : yhi; xhi = xhi > W - 1 ? W - 1 : xhi;
    yhi = yhi > ylo + 63 ? ylo + 63 : yhi; xhi = xhi > xlo + SEGF_XWIN - 1 ? xlo + SEGF_XWIN - 1 : xhi;   // (ratio <= 16: never binds)
    {
        int i0, i1;
        float l1;
        if (wave == 0) {
            float wy = 0.f;
            if (ylo + lane <= yhi) {
                segf_taps(ylo + lane, h, H, i0, i1, l1);
                wy = (i0 == qy ? 1.f - l1 : 0.f) + (i1 == qy ? l1 : 0.f);
            }
            s_wy[lane] = wy;
        }
        for (int c = threadIdx.x; c < SEGF_XWIN; c += 256) {
            f32x4 wx = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (xlo + c <= xhi) {
                segf_taps(xlo + c, w, W, i0, i1, l1);
#pragma unroll
                for (int j = 0; j < SEGF_XQ; ++j)
                    wx[j] = (qx0 + j < w) ? (i0 == qx0 + j ? 1.f - l1 : 0.f) + (i1 == qx0 + j ? l1 : 0.f) : 0.f;
            }
            s_wx[c] = wx;
        }
    }
    __syncthreads();
    const float* gb = g + (long)n * H * W * SEGF_C;
    f32x4 acc[SEGF_XQ];
#pragma unroll
    for (int j = 0; j < SEGF_XQ; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int yy = ylo + wave; yy <= yhi; yy += 4) {
        const float wy = s_wy[yy - ylo];
        if (wy == 0.f) continue;
        const float* grow = gb + (long)yy * W * SEGF_C;
        f32x4 part[SEGF_XQ];
#pragma unroll
        for (int j = 0; j < SEGF_XQ; ++j) part[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int xx = xlo;
        for (; xx + 7 <= xhi; xx += 8) {          // eight independent row loads in flight
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ((const f32x4*)(grow + (long)(xx + u) * SEGF_C))[lane];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const f32x4 wx = s_wx[xx + u - xlo];
#pragma unroll
                for (int j = 0; j < SEGF_XQ; ++j) part[j] += v[u] * wx[j];
            }
        }
        for (; xx <= xhi; ++xx) {
            const f32x4 v = ((const f32x4*)(grow + (long)xx * SEGF_C))[lane];
            const f32x4 wx = s_wx[xx - xlo];
#pragma unroll
            for (int j = 0; j < SEGF_XQ; ++j) part[j] += v * wx[j];
        }
#pragma unroll
        for (int j = 0; j < SEGF_XQ; ++j) acc[j] += part[j] * wy;
    }
#pragma unroll
    for (int j = 0; j < SEGF_XQ; ++j) s_part[wave][j][lane] = acc[j];
    __syncthreads();
    // wave j finishes pixel qx0 + j (SEGF_XQ == number of waves)
    if (qx0 + wave < w) {
        const f32x4 t = (s_part[0][wave][lane] + s_part[1][wave][lane]) + (s_part[2][wave][lane] + s_part[3][wave][lane]);
        ((f32x4*)(mp.dz[m] + ((long)n * h * w + (long)qy * w + qx0 + wave) * SEGF_C))[lane] = t;
    }
}

// --------------------------------------------------------------------------- composed embedding weights: the constant and its backward
// d[r] = sum_j fuse_w[r][j * e + c] * b_j[c] (column block j of the [e][k e] fuse weight carries scale k - 1 - j: the reference's cat order
// c4, c3, c2, c1 -- cffm_head.py:112-119): the bias of the four embeddings pushed through linear_fuse.conv.  One wave per output row.
struct FuseBias { const float* b[4]; float* db[4]; };      // b[j] = bias of column block j (i.e. of scale k - 1 - j), e floats each
__global__ void __launch_bounds__(256) k_fuse_const(const float* __restrict__ fw, FuseBias fb, int k, int e, float* __restrict__ d) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= e) return;
    float acc = 0.f;
    for (int j = 0; j < k; ++j)
        for (int c = lane; c < e; c += 64) acc += fw[(long)row * k * e + j * e + c] * fb.b[j][c];
    acc = wave_sum(acc);
    if (lane == 0) d[row] = acc;
}
// backward of the constant: dfw[r][j e + c] += dd[r] * b_j[c] (dfw already holds dA W^T of the composed matrices), db_j[c] = sum_r fw[r][j e + c] dd[r].
// A workgroup owns 64 columns of the [e][k e] matrix; its four waves take every fourth row, the four column sums meet in LDS in a fixed order.
__global__ void __launch_bounds__(256) k_fuse_const_bwd(const float* __restrict__ fw, const float* __restrict__ dd, FuseBias fb, int k, int e,
                                                         float* __restrict__ dfw) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), wv = threadIdx.x >> 6, j = col / e, c = col % e;
    const float b = fb.b[j][c];
    float acc = 0.f;
    for (int r0 = wv; r0 < e; r0 += 32) {          // eight rows in flight per lane (a row at a time: 64 dependent round trips, 22 us)
        float w8[8], d8[8], g8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + 4 * u;
            const long o = (long)r * k * e + col;
            g8[u] = r < e ? dd[r] : 0.f;
            w8[u] = r < e ? fw[o] : 0.f;
            d8[u] = r < e ? dfw[o] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + 4 * u;
            acc += w8[u] * g8[u];
            if (r < e) dfw[(long)r * k * e + col] = d8[u] + g8[u] * b;
        }
    }
    red[wv][threadIdx.x & 63] = acc;
    __syncthreads();
    if (wv == 0) fb.db[j][c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

