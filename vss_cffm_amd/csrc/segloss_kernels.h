// segloss_kernels.h -- the head's training loss without the full-resolution logits.
// Reference (decode_head.py:744-835 -> losses/cross_entropy_loss.py:9-40): the [M,K,h,w] logits of every frame are resized
// to the label resolution (bilinear, align_corners=False: 120 -> 480, [M,K,H,W] = 114 MB per frame for K = 124), then
// F.cross_entropy(reduction='none', ignore_index) and a mean over ALL pixels; accuracy = top-1 over all pixels.
// Here the resized logits only ever exist in registers: the forward pass interpolates the K logits of an output pixel from
// an LDS tile of the low-resolution map, runs an online softmax over them and keeps one float per pixel (the log-sum-exp);
// the backward pass is the adjoint in gather form (deterministic): a low-resolution pixel sums, over the output pixels that
// tap it, weight * (softmax - onehot), recomputing each interpolated logit from the same LDS tile.
#pragma once
#include "cffm_common.h"
#include "segfuse_kernels.h"   // segf_taps: the bilinear tap rule

#define UPCE_TILE 16            // output pixels per workgroup side (forward)
#define UPCE_QT 4               // low-resolution pixels per workgroup side (backward)
#define UPCE_MAX_RATIO 8

struct UpceGeom {
    int M, K, h, w, H, W;
    int ignore;                 // ignore_index (labels outside [0,K) are treated the same way)
    int rn, cn;                 // rows / columns of the low-resolution LDS tile
    int foot, win;              // (backward) LDS capacity: footprint pixels of a tile, candidate rows / columns of a window
};

// The BACKWARD's LDS tile holds logits * log2(e) and its staged log-sum-exp is scaled the same way, so its exponentials are one
// v_exp_f32 without the multiply of expf (624 -> 595 us).  (The same change made the forward kernel slower, 192 -> 237 us: it keeps
// natural units and expf.)
#define UPCE_LOG2E 1.4426950408889634f
#define UPCE_LN2 0.6931471805599453f
__device__ __forceinline__ float upce_exp2(float x) {
#ifdef CFFM_EMU
    return exp2f(x);
#else
    return __builtin_amdgcn_exp2f(x);
#endif
}
#ifndef UPCE_ABLATE
#define UPCE_ABLATE 0
#endif
#if UPCE_ABLATE & 1          // timing experiments only: no exponential
#define UPCE_EXP(x) (x)
#else
#define UPCE_EXP(x) upce_exp2(x)
#endif
#define UPCE_KP(K) (((K) + 3) & ~3)       // classes padded to whole 16-byte groups (padding logits = -1e30: exp -> 0, never the arg-max)

// stage rows r0.. / columns c0.. (rn x cn, clamped to the map) of all K channels of map m as s_l[cell][KP]: a thread reads
// four classes of a tap with one ds_read_b128 and interpolates them with packed fp32 math
__device__ __forceinline__ void upce_stage(float* s_l, const float* __restrict__ logits, const UpceGeom& G, int m, int r0, int c0, float mul) {
    const int cells = G.rn * G.cn, KP = UPCE_KP(G.K);
    const float* base = logits + (long)m * G.K * G.h * G.w;
    for (int e = threadIdx.x; e < KP * cells; e += 256) {
        const int k = e / cells, rc = e - k * cells, r = rc / G.cn, c = rc - r * G.cn;
        const int rr = r0 + r < G.h ? r0 + r : G.h - 1, cc = c0 + c < G.w ? c0 + c : G.w - 1;
        s_l[rc * KP + k] = k < G.K ? base[((long)k * G.h + rr) * G.w + cc] * mul : -1.0e30f;
    }
}
// s_l[cells * KP + cell] = max over k of s_l[cell][k] (after upce_stage + barrier; needs another barrier)
__device__ __forceinline__ void upce_cell_max(float* s_l, const UpceGeom& G) {
    const int cells = G.rn * G.cn, KP = UPCE_KP(G.K);
    for (int e = threadIdx.x; e < cells; e += 256) {
        float mx = s_l[e * KP];
        for (int k = 1; k < G.K; ++k) mx = fmaxf(mx, s_l[e * KP + k]);
        s_l[cells * KP + e] = mx;
    }
}
// Packed fp32 math on class PAIRS, spelled as instructions: left to itself the compiler pairs the two taps of ONE class (its SLP
// vectoriser follows the expression tree), which costs register shuffles and a cross add per class -- ~14 VALU instructions per
// (pixel, class), and these kernels are VALU-bound (SQ_ACTIVE_INST_VALU x resident waves ~ 100 %, profiles/r01_pmc_sq_rows.txt).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) {
#ifdef CFFM_EMU
    return a * b;
#else
    f32x2 d;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
#endif
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
#ifdef CFFM_EMU
    return a * b + c;
#else
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
#endif
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
#ifdef CFFM_EMU
    return a + b;
#else
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
#endif
}
#define UPCE_LO(v) __builtin_shufflevector(v, v, 0, 1)
#define UPCE_HI(v) __builtin_shufflevector(v, v, 2, 3)
// the four bilinear weights of a pixel, each duplicated into a register pair
struct UpceW { f32x2 w00, w01, w10, w11; };
__device__ __forceinline__ UpceW upce_weights(float hx0, float lx, float hy0, float ly) {
    UpceW w;
    const float a = hy0 * hx0, b = hy0 * lx, c = ly * hx0, d = ly * lx;
    w.w00 = (f32x2){a, a}; w.w01 = (f32x2){b, b}; w.w10 = (f32x2){c, c}; w.w11 = (f32x2){d, d};
    return w;
}
// four classes of one interpolated pixel, v = w00 ta + w01 tb + w10 tc + w11 td (one rounding away from ATen's nested form,
// ~1e-7 relative), plus `off` (minus the bound / the log-sum-exp): two class pairs x (1 mul + 4 fma)
__device__ __forceinline__ void upce_interp4(const float* s_l, int a, int b, int c, int d, int k4, const UpceW& w, f32x2 off, f32x2& lo, f32x2& hi) {
    const f32x4 ta = ((const f32x4*)(s_l + a))[k4];
#if UPCE_ABLATE & 2          // timing experiments only: one LDS read instead of four
    const f32x4 tb = ta, tc = ta, td = ta;
    (void)b; (void)c; (void)d;
#else
    const f32x4 tb = ((const f32x4*)(s_l + b))[k4], tc = ((const f32x4*)(s_l + c))[k4], td = ((const f32x4*)(s_l + d))[k4];
#endif
    lo = pk_fma(UPCE_LO(td), w.w11, pk_fma(UPCE_LO(tc), w.w10, pk_fma(UPCE_LO(tb), w.w01, pk_fma(UPCE_LO(ta), w.w00, off))));
    hi = pk_fma(UPCE_HI(td), w.w11, pk_fma(UPCE_HI(tc), w.w10, pk_fma(UPCE_HI(tb), w.w01, pk_fma(UPCE_HI(ta), w.w00, off))));
}

// grid M * ceil(H/16) * ceil(W/16), 256 threads = 16 x 16 output pixels; dynamic LDS rn*cn*(KP+1) floats.
// lse[m][y][x]; part[block][0] = sum of per-pixel losses, part[block][1] = number of pixels whose argmax is the label.
__global__ void __launch_bounds__(256) k_upce_fwd(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                   float* __restrict__ lse, float* __restrict__ part, UpceGeom G) {
    CFFM_DYN_SMEM(smem);
    float* s_l = (float*)smem;
    __shared__ float s_red[2][4];
    // 1-D grid, re-numbered so that every XCD (workgroup b runs on XCD b % 8) owns a contiguous run of tiles: the tiles that
    // share lines of the low-resolution logits (24-byte segments of 128-byte lines) then fill ONE L2, not eight
    const int gx = (G.W + UPCE_TILE - 1) / UPCE_TILE, gy = (G.H + UPCE_TILE - 1) / UPCE_TILE;
    const int lin = xcd_linear_id(), m = lin / (gx * gy), trem = lin - m * gx * gy;
    const int ty = (trem / gx) * UPCE_TILE, tx = (trem - (trem / gx) * gx) * UPCE_TILE;
    int r0, c0, t1;
    float tl;
    segf_taps(ty, G.h, G.H, r0, t1, tl);
    segf_taps(tx, G.w, G.W, c0, t1, tl);
    upce_stage(s_l, logits, G, m, r0, c0, 1.f);
    __syncthreads();
    upce_cell_max(s_l, G);
    __syncthreads();
    const int oy = ty + (threadIdx.x >> 4), ox = tx + (threadIdx.x & 15);
    float loss = 0.f, hit = 0.f;
    if (oy < G.H && ox < G.W) {
        int y0, y1, x0, x1;
        float ly, lx;
        segf_taps(oy, G.h, G.H, y0, y1, ly);
        segf_taps(ox, G.w, G.W, x0, x1, lx);
        const int KP = UPCE_KP(G.K), cells = G.rn * G.cn;
        const int ca = (y0 - r0) * G.cn + (x0 - c0), cb = (y0 - r0) * G.cn + (x1 - c0), cc = (y1 - r0) * G.cn + (x0 - c0),
                  cd = (y1 - r0) * G.cn + (x1 - c0);
        const int a = ca * KP, b = cb * KP, c = cc * KP, d = cd * KP;
        const float hx0 = 1.f - lx, hy0 = 1.f - ly;
        const long long lab = labels[((long)m * G.H + oy) * G.W + ox];
        const bool counted = lab != G.ignore && lab >= 0 && lab < G.K;
        // an interpolated logit is a convex combination of its four taps, so the largest tap value over all classes bounds every
        // one of them: exponentials relative to that bound need no running rescale
        const float* tm = s_l + cells * KP;
        const float bound = fmaxf(fmaxf(tm[ca], tm[cb]), fmaxf(tm[cc], tm[cd]));
        f32x4 sum4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        float best = -3.0e38f;
        int arg = -1;
        const UpceW wts = upce_weights(hx0, lx, hy0, ly);
        const f32x2 zero2 = (f32x2){0.f, 0.f};
        for (int k4 = 0; k4 < KP / 4; ++k4) {
            f32x2 vlo, vhi;
            upce_interp4(s_l, a, b, c, d, k4, wts, zero2, vlo, vhi);
            const float v[4] = {vlo[0], vlo[1], vhi[0], vhi[1]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sum4[j] += fast_exp(v[j] - bound);
                if (v[j] > best) { best = v[j]; arg = 4 * k4 + j; }
            }
        }
        float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]), mx = bound;
        if (!(sum > 1e-30f)) {      // taps disagreeing by more than ~70 in some class: the bound is too far above; use the maximum
            mx = best; sum = 0.f;
            for (int k4 = 0; k4 < KP / 4; ++k4) {
                f32x2 vlo, vhi;
                upce_interp4(s_l, a, b, c, d, k4, wts, zero2, vlo, vhi);
                sum += (fast_exp(vlo[0] - mx) + fast_exp(vlo[1] - mx)) + (fast_exp(vhi[0] - mx) + fast_exp(vhi[1] - mx));
            }
        }
        float at_label = 0.f;
        if (counted) {
            const int kl = (int)lab;
            at_label = fmaf(s_l[d + kl], wts.w11[0], fmaf(s_l[c + kl], wts.w10[0], fmaf(s_l[b + kl], wts.w01[0], s_l[a + kl] * wts.w00[0])));   // as upce_interp4
        }
        const float l = mx + logf(sum);
        lse[((long)m * G.H + oy) * G.W + ox] = l;
        loss = counted ? l - at_label : 0.f;
        hit = (counted && arg == (int)lab) ? 1.f : 0.f;
    }
    loss = wave_sum(loss);
    hit = wave_sum(hit);
    if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = loss; s_red[1][threadIdx.x >> 6] = hit; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const long blk = lin;
        part[blk * 2 + threadIdx.x] = (s_red[threadIdx.x][0] + s_red[threadIdx.x][1]) + (s_red[threadIdx.x][2] + s_red[threadIdx.x][3]);
    }
}

// dlogits[m][k][qy][qx] = scale * (*gscale) * sum over output pixels p tapping q (labels counted) of w(p,q) * (softmax_k(p) - [k == label_p])
// grid M * ceil(h/4) * ceil(w/4); 256 threads = 16 low-resolution pixels x 16 class lanes, a lane owning the f32x4 class groups
// cl, cl + 16, cl + 32, cl + 48 (K <= 256); dynamic LDS: rn*cn*KP floats (the tile's pixels and one ring around them: every tap of
// every output pixel that taps a tile pixel) + the footprint's lse / labels + the tap tables (16 pixels x 2 x win x 16 B); foot / win are
// sized by the host for the actual resize factor (a worst-case size left three workgroups per CU).
__global__ void __launch_bounds__(256) k_upce_bwd(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                   const float* __restrict__ lse, const float* __restrict__ gscale, float scale,
                                                   float* __restrict__ dlogits, UpceGeom G) {
    CFFM_DYN_SMEM(smem);
    float* s_l = (float*)smem;
    const int KP = UPCE_KP(G.K);
    const int gx = (G.w + UPCE_QT - 1) / UPCE_QT, gy = (G.h + UPCE_QT - 1) / UPCE_QT;      // XCD-contiguous tile order, as in the forward
    const int lin = xcd_linear_id(), m = lin / (gx * gy), trem = lin - m * gx * gy;
    const int q0y = (trem / gx) * UPCE_QT, q0x = (trem - (trem / gx) * gx) * UPCE_QT;
    const int r0 = q0y > 0 ? q0y - 1 : 0, c0 = q0x > 0 ? q0x - 1 : 0;
    upce_stage(s_l, logits, G, m, r0, c0, UPCE_LOG2E);
    // the tile's footprint in the output: log-sum-exp and label of every pixel that can tap a tile pixel, staged once
    const float isy = (float)G.H / (float)G.h, isx = (float)G.W / (float)G.w;
    const int q1y = q0y + UPCE_QT - 1 < G.h - 1 ? q0y + UPCE_QT - 1 : G.h - 1, q1x = q0x + UPCE_QT - 1 < G.w - 1 ? q0x + UPCE_QT - 1 : G.w - 1;
    int fy0 = (int)floorf(((float)q0y - 0.5f) * isy - 0.5f) - 1, fy1 = (int)ceilf(((float)q1y + 1.5f) * isy - 0.5f) + 1;
    int fx0 = (int)floorf(((float)q0x - 0.5f) * isx - 0.5f) - 1, fx1 = (int)ceilf(((float)q1x + 1.5f) * isx - 0.5f) + 1;
    fy0 = fy0 < 0 ? 0 : fy0; fx0 = fx0 < 0 ? 0 : fx0;
    fy1 = fy1 > G.H - 1 ? G.H - 1 : fy1; fx1 = fx1 > G.W - 1 ? G.W - 1 : fx1;
    const int fw = fx1 - fx0 + 1, fn = (fy1 - fy0 + 1) * fw;
    float* s_lse = s_l + KP * G.rn * G.cn;
    int* s_lab = (int*)(s_lse + G.foot);
    for (int e = threadIdx.x; e < fn; e += 256) {
        const int fy = e / fw, fx = e - fy * fw;
        const long pix = ((long)m * G.H + fy0 + fy) * G.W + fx0 + fx;
        const long long lab = labels[pix];
        s_lse[e] = lse[pix] * UPCE_LOG2E;
        s_lab[e] = (lab == G.ignore || lab < 0 || lab >= G.K) ? -1 : (int)lab;
    }
    __syncthreads();
    // one thread per (low-resolution pixel, class lane).  (A wave per pixel with the window's rows split over four lane groups,
    // so that all lanes walk the same columns, was slower: 1000 us vs 754 -- the windows are too short to split.)
    // The tap rule of every candidate row / column of a pixel's window is evaluated ONCE, by the pixel's 16 lanes, into LDS
    // tables {index << 1 | second-tap flag, LDS offset of the first tap, lambda, weight}; the loops below only read them
    // (evaluating the rule per thread and visited pixel, and skipping zero-weight candidates after it, was ~80 % of the kernel).
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4* s_ty = (i32x4*)(s_lab + G.foot);
    i32x4* s_tx = s_ty + 16 * G.win;
    const int q = threadIdx.x >> 4, cl = threadIdx.x & 15, qy = q0y + (q >> 2), qx = q0x + (q & 3);
    const bool live = qy < G.h && qx < G.w;
    int ylo = (int)floorf(((float)qy - 0.5f) * isy - 0.5f) - 1, yhi = (int)ceilf(((float)qy + 1.5f) * isy - 0.5f) + 1;
    int xlo = (int)floorf(((float)qx - 0.5f) * isx - 0.5f) - 1, xhi = (int)ceilf(((float)qx + 1.5f) * isx - 0.5f) + 1;
    ylo = ylo < fy0 ? fy0 : ylo; xlo = xlo < fx0 ? fx0 : xlo;
    yhi = yhi > fy1 ? fy1 : yhi; xhi = xhi > fx1 ? fx1 : xhi;
    yhi = yhi > ylo + G.win - 1 ? ylo + G.win - 1 : yhi; xhi = xhi > xlo + G.win - 1 ? xlo + G.win - 1 : xhi;   // (never binds)
    for (int e = cl; e < G.win; e += 16) {
        i32x4 ty = (i32x4){0, 0, 0, 0}, tx = ty;
        if (live && ylo + e <= yhi) {
            int i0, i1;
            float l1;
            segf_taps(ylo + e, G.h, G.H, i0, i1, l1);
            const float wgt = (i0 == qy ? 1.f - l1 : 0.f) + (i1 == qy ? l1 : 0.f);
            ty = (i32x4){((ylo + e - fy0) << 1) | (i1 != i0), (i0 - r0) * G.cn * KP, __builtin_bit_cast(int, l1), __builtin_bit_cast(int, wgt)};
        }
        if (live && xlo + e <= xhi) {
            int i0, i1;
            float l1;
            segf_taps(xlo + e, G.w, G.W, i0, i1, l1);
            const float wgt = (i0 == qx ? 1.f - l1 : 0.f) + (i1 == qx ? l1 : 0.f);
            tx = (i32x4){((xlo + e - fx0) << 1) | (i1 != i0), (i0 - c0) * KP, __builtin_bit_cast(int, l1), __builtin_bit_cast(int, wgt)};
        }
        s_ty[q * G.win + e] = ty;
        s_tx[q * G.win + e] = tx;
    }
    __syncthreads();
    if (!live) return;
    const int rstep = G.cn * KP;
    // the tapping candidates are a contiguous run of the window: trim the zero-weight ends, so that the four pixels that share a
    // wave run the same number of iterations, all of them useful (their runs start at different candidates)
    int jy0 = 0, jy1 = yhi - ylo + 1, jx0 = 0, jx1 = xhi - xlo + 1;
    while (jy0 < jy1 && s_ty[q * G.win + jy0][3] == 0) ++jy0;
    while (jy1 > jy0 && s_ty[q * G.win + jy1 - 1][3] == 0) --jy1;
    while (jx0 < jx1 && s_tx[q * G.win + jx0][3] == 0) ++jx0;
    while (jx1 > jx0 && s_tx[q * G.win + jx1 - 1][3] == 0) --jx1;
    const int ng = KP / 4;                      // 16-byte class groups; this lane owns groups cl, cl + 16, cl + 32, cl + 48
    f32x2 acc_lo[4], acc_hi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc_lo[i] = (f32x2){0.f, 0.f}; acc_hi[i] = (f32x2){0.f, 0.f}; }
    for (int jy = jy0; jy < jy1; ++jy) {
        const i32x4 ty = s_ty[q * G.win + jy];
        const float wy = __builtin_bit_cast(float, (int)ty[3]);
        if (wy == 0.f) continue;
        const float ly = __builtin_bit_cast(float, (int)ty[2]), hy0 = 1.f - ly;
        const int ra = ty[1], rb = ra + ((ty[0] & 1) ? rstep : 0), fer = (ty[0] >> 1) * fw;
        for (int jx = jx0; jx < jx1; ++jx) {
            const i32x4 tx = s_tx[q * G.win + jx];
            const float wq = wy * __builtin_bit_cast(float, (int)tx[3]);
            if (wq == 0.f) continue;
            const int fe = fer + (tx[0] >> 1);
            const int lab = s_lab[fe];
            if (lab < 0) continue;
            const float l = s_lse[fe];
            const float lx = __builtin_bit_cast(float, (int)tx[2]), hx0 = 1.f - lx;
            const int xa = tx[1], xb = xa + ((tx[0] & 1) ? KP : 0);
            const UpceW wts = upce_weights(hx0, lx, hy0, ly);
            const f32x2 ml = (f32x2){-l, -l}, wq2 = (f32x2){wq, wq};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k4 = cl + 16 * i;
                if (k4 < ng) {
                    f32x2 vlo, vhi;                                       // interpolated logit - lse (log2 units)
                    upce_interp4(s_l, ra + xa, ra + xb, rb + xa, rb + xb, k4, wts, ml, vlo, vhi);
                    const int e = lab - 4 * k4;                          // the label's position in this group, if it is in it
                    const f32x2 plo = (f32x2){UPCE_EXP(vlo[0]) - (e == 0 ? 1.f : 0.f), UPCE_EXP(vlo[1]) - (e == 1 ? 1.f : 0.f)};
                    const f32x2 phi = (f32x2){UPCE_EXP(vhi[0]) - (e == 2 ? 1.f : 0.f), UPCE_EXP(vhi[1]) - (e == 3 ? 1.f : 0.f)};
                    acc_lo[i] = pk_fma(plo, wq2, acc_lo[i]);
                    acc_hi[i] = pk_fma(phi, wq2, acc_hi[i]);
                }
            }
        }
    }
    const float sc = scale * (gscale ? *gscale : 1.f);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * (cl + 16 * i) + j;
            if (k < G.K) dlogits[(((long)m * G.K + k) * G.h + qy) * G.w + qx] = sc * (j < 2 ? acc_lo[i][j & 1] : acc_hi[i][j & 1]);
        }
}


// ---- evaluation: per-class pixel counts of a prediction map against a label map ------------------------------
// Reference: mmseg/core/evaluation/metrics.py:62-119 `intersect_and_union` (numpy on the host): optional reduce_zero_label
// (0 -> 255, the rest minus 1), drop label == ignore_index, then three np.histogram(bins=arange(K+1)) over the matching
// predictions, the predictions and the labels -- a value v counts iff 0 <= v <= K, into bin min(v, K-1) (numpy's last bin is
// closed).  Integer work: per-workgroup LDS histograms, one 64-bit atomic per (workgroup, non-empty bin); results are exact
// and therefore independent of the order.  counts[3][K] (intersect | prediction | label) are ACCUMULATED (total_intersect_and_union).
#define SEGCNT_MAXK 1024
__global__ void __launch_bounds__(256) k_seg_counts(const long long* __restrict__ pred, const long long* __restrict__ label, long n, int K,
                                                     int ignore, int reduce_zero, unsigned long long* __restrict__ counts) {
    __shared__ unsigned int hist[3 * SEGCNT_MAXK];
    for (int e = threadIdx.x; e < 3 * K; e += 256) hist[e] = 0u;
    __syncthreads();
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        long long l = label[i];
        const long long p = pred[i];
        if (reduce_zero) {                       // metrics.py:98-102 (on uint8 label maps: 0 -> 255, others - 1, 254 -> 255)
            if (l == 0) l = 255;
            l -= 1;
            if (l == 254) l = 255;
        }
        if (l == ignore) continue;
        const bool pv = p >= 0 && p <= K, lv = l >= 0 && l <= K;
        const int pb = (int)(p < K ? p : K - 1), lb = (int)(l < K ? l : K - 1);
        if (pv) atomicAdd(&hist[K + pb], 1u);
        if (lv) atomicAdd(&hist[2 * K + lb], 1u);
        if (p == l && pv) atomicAdd(&hist[pb], 1u);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 3 * K; e += 256)
        if (hist[e]) atomicAdd(&counts[e], (unsigned long long)hist[e]);
}


// ---- evaluation: video consistency VC_n -----------------------------------------------------------------------
// Reference: VC_perclip.py:62-78 `get_common`: for every start frame i < F - n, a pixel is "stable" when its label is the same in
// frames i .. i+n-1; acc_i = |stable in the ground truth AND in the prediction| / |stable in the ground truth|.
// counts[i][0] = the numerator, counts[i][1] = the denominator (exact integers, accumulated with one atomic per workgroup).
// grid (ceil(h*w / 256), F - n)
__global__ void __launch_bounds__(256) k_vc_counts(const long long* __restrict__ gt, const long long* __restrict__ pred, long npix, int n,
                                                    unsigned long long* __restrict__ counts) {
    __shared__ unsigned int s_cnt[2][4];
    const int i = blockIdx.y;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    float both = 0.f, stable = 0.f;
    if (p < npix) {
        const long long g0 = gt[(long)i * npix + p], p0 = pred[(long)i * npix + p];
        bool gs = true, ps = true;
        for (int j = 1; j < n; ++j) {
            gs = gs && gt[(long)(i + j) * npix + p] == g0;
            ps = ps && pred[(long)(i + j) * npix + p] == p0;
        }
        stable = gs ? 1.f : 0.f;
        both = (gs && ps) ? 1.f : 0.f;
    }
    both = wave_sum(both);            // <= 64: exact in fp32
    stable = wave_sum(stable);
    if ((threadIdx.x & 63) == 0) { s_cnt[0][threadIdx.x >> 6] = (unsigned int)both; s_cnt[1][threadIdx.x >> 6] = (unsigned int)stable; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const unsigned int v = s_cnt[threadIdx.x][0] + s_cnt[threadIdx.x][1] + s_cnt[threadIdx.x][2] + s_cnt[threadIdx.x][3];
        if (v) atomicAdd(&counts[2 * i + threadIdx.x], (unsigned long long)v);
    }
}
