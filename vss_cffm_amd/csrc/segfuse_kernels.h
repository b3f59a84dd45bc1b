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

struct SegfMaps {
    const float* z[3];   // low-resolution embeddings [N, h*w, 256]
    float* dz[3];        // (backward) their gradients
    int h[3], w[3];
    int cnt;             // maps in use (the SegFormer decoder: 3)
    int blk_end[3];      // (backward) exclusive prefix of workgroups per map
};

// y[n, p, :] += d + sum_m bilinear(z_m)[n, p, :]     (y holds the 1/4-scale embedding on entry); grid (ceil(N*H*W / 4))
__global__ void __launch_bounds__(256) k_segfuse_fwd(float* __restrict__ y, const float* __restrict__ d, SegfMaps mp, int N, int H, int W) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= (long)N * H * W) return;
    const int n = (int)(row / ((long)H * W)), p = (int)(row - (long)n * H * W), oy = p / W, ox = p - oy * W;
    f32x4 acc = ((const f32x4*)(y + row * SEGF_C))[lane] + ((const f32x4*)d)[lane];
    f32x4 t[3][4];
    float wt[3][4];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        if (m < mp.cnt) {
            int y0, y1, x0, x1;
            float ly, lx;
            segf_taps(oy, mp.h[m], H, y0, y1, ly);
            segf_taps(ox, mp.w[m], W, x0, x1, lx);
            const float* base = mp.z[m] + (long)n * mp.h[m] * mp.w[m] * SEGF_C;
            t[m][0] = ((const f32x4*)(base + ((long)y0 * mp.w[m] + x0) * SEGF_C))[lane];
            t[m][1] = ((const f32x4*)(base + ((long)y0 * mp.w[m] + x1) * SEGF_C))[lane];
            t[m][2] = ((const f32x4*)(base + ((long)y1 * mp.w[m] + x0) * SEGF_C))[lane];
            t[m][3] = ((const f32x4*)(base + ((long)y1 * mp.w[m] + x1) * SEGF_C))[lane];
            wt[m][0] = (1.f - ly) * (1.f - lx);
            wt[m][1] = (1.f - ly) * lx;
            wt[m][2] = ly * (1.f - lx);
            wt[m][3] = ly * lx;
        }
    }
#pragma unroll
    for (int m = 0; m < 3; ++m)
        if (m < mp.cnt) acc += (t[m][0] * wt[m][0] + t[m][1] * wt[m][1]) + (t[m][2] * wt[m][2] + t[m][3] * wt[m][3]);
    ((f32x4*)(y + row * SEGF_C))[lane] = acc;
}

// Adjoint of the three resizes: dz_m[n, q, :] = sum over the output pixels p that tap q of weight(p, q) * g[n, p, :].
// Gather form (deterministic, no atomics): one wave per low-resolution pixel; the candidate output rows / columns are the
// inverse image of (q-1, q+1) under the source-index map, widened by one, and every candidate's weight is recomputed with
// the forward's own tap rule (a candidate that does not tap q gets weight 0), so forward and adjoint cannot disagree.
// Weights of the <= 64 candidates per dimension live in LDS per wave.  grid = total low-res pixels / 4 (per map prefix).
__global__ void __launch_bounds__(256) k_segfuse_bwd(const float* __restrict__ g, SegfMaps mp, int N, int H, int W) {
    __shared__ float s_w[4][2][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int m = 0;
    while (m + 1 < mp.cnt && (int)blockIdx.x >= mp.blk_end[m]) ++m;
    const int blk = (int)blockIdx.x - (m ? mp.blk_end[m - 1] : 0);
    const int h = mp.h[m], w = mp.w[m];
    const long q = (long)blk * 4 + wave, nq = (long)N * h * w;
    const bool live = q < nq;
    const int n = live ? (int)(q / ((long)h * w)) : 0, r = live ? (int)(q - (long)n * h * w) : 0, qy = r / w, qx = r - qy * w;
    // candidate windows
    const float isy = (float)H / (float)h, isx = (float)W / (float)w;
    int ylo = (int)floorf(((float)qy - 0.5f) * isy - 0.5f) - 1, yhi = (int)ceilf(((float)qy + 1.5f) * isy - 0.5f) + 1;
    int xlo = (int)floorf(((float)qx - 0.5f) * isx - 0.5f) - 1, xhi = (int)ceilf(((float)qx + 1.5f) * isx - 0.5f) + 1;
    ylo = ylo < 0 ? 0 : ylo; xlo = xlo < 0 ? 0 : xlo;
    yhi = yhi > H - 1 ? H - 1 : yhi; xhi = xhi > W - 1 ? W - 1 : xhi;
    yhi = yhi > ylo + 63 ? ylo + 63 : yhi; xhi = xhi > xlo + 63 ? xlo + 63 : xhi;   // (ratio <= SEGF_MAX_RATIO: never binds)
    {
        int i0, i1;
        float l1;
        float wy = 0.f, wx = 0.f;
        if (ylo + lane <= yhi) {
            segf_taps(ylo + lane, h, H, i0, i1, l1);
            wy = (i0 == qy ? 1.f - l1 : 0.f) + (i1 == qy ? l1 : 0.f);
        }
        if (xlo + lane <= xhi) {
            segf_taps(xlo + lane, w, W, i0, i1, l1);
            wx = (i0 == qx ? 1.f - l1 : 0.f) + (i1 == qx ? l1 : 0.f);
        }
        s_w[wave][0][lane] = wy;
        s_w[wave][1][lane] = wx;
    }
    __syncthreads();
    if (!live) return;
    const float* gb = g + (long)n * H * W * SEGF_C;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int yy = ylo; yy <= yhi; ++yy) {
        const float wy = s_w[wave][0][yy - ylo];
        if (wy == 0.f) continue;
        const float* grow = gb + (long)yy * W * SEGF_C;
        f32x4 part = (f32x4){0.f, 0.f, 0.f, 0.f};
        int xx = xlo;
        for (; xx + 3 <= xhi; xx += 4) {          // four independent row loads in flight
            f32x4 v0 = ((const f32x4*)(grow + (long)(xx + 0) * SEGF_C))[lane];
            f32x4 v1 = ((const f32x4*)(grow + (long)(xx + 1) * SEGF_C))[lane];
            f32x4 v2 = ((const f32x4*)(grow + (long)(xx + 2) * SEGF_C))[lane];
            f32x4 v3 = ((const f32x4*)(grow + (long)(xx + 3) * SEGF_C))[lane];
            part += (v0 * s_w[wave][1][xx - xlo] + v1 * s_w[wave][1][xx + 1 - xlo]) +
                    (v2 * s_w[wave][1][xx + 2 - xlo] + v3 * s_w[wave][1][xx + 3 - xlo]);
        }
        for (; xx <= xhi; ++xx) part += ((const f32x4*)(grow + (long)xx * SEGF_C))[lane] * s_w[wave][1][xx - xlo];
        acc += part * wy;
    }
    ((f32x4*)(mp.dz[m] + q * SEGF_C))[lane] = acc;
}
