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
    // Where the logits (and their gradient) live -- round 2: the heads keep them as token rows [.., h, w, K] straight out of the
    // classifier GEMMs, several maps of one clip in one buffer, so the element (map m, class k, cell (r, c)) is at
    //   (m / inner) * ms_outer + (m % inner) * ms_inner + k * ks + (r * w + c) * ps
    // (plain [M,K,h,w]: inner 1, ms_outer K h w, ks h w, ps 1; rows [B, n, h, w, K]: inner n, ms_outer n h w K, ms_inner h w K, ks 1, ps K)
    long ms_outer, ms_inner;
    int inner, ks, ps;
    const int* label_idx;       // per map: the label map it is judged on (NULL: its own index)
    const float* map_scale;     // (backward) per map: factor on its gradient (NULL: 1)
};
__device__ __forceinline__ long upce_map_base(const UpceGeom& G, int m) {
    return (long)(m / G.inner) * G.ms_outer + (long)(m % G.inner) * G.ms_inner;
}
__device__ __forceinline__ long upce_label_map(const UpceGeom& G, int m) { return G.label_idx ? (long)G.label_idx[m] : (long)m; }

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
#define UPCE_EXP(x) upce_exp2(x)
#define UPCE_ROWS_KPER 32        // forward staging of token-row logits: threads per cell (one 128-byte run of classes per pass)
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
__device__ __forceinline__ float upce_max3(float a, float b, float c) {
#ifdef CFFM_EMU
    return fmaxf(fmaxf(a, b), c);
#else
    float d;                    // (fmaxf() costs a canonicalising v_max per operand; the inputs here are never NaN-signalling)
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
#endif
}
// four classes of one interpolated pixel, v = w00 ta + w01 tb + w10 tc + w11 td (one rounding away from ATen's nested form,
// ~1e-7 relative), plus `off` (minus the bound / the log-sum-exp): two class pairs x (1 mul + 4 fma)
__device__ __forceinline__ void upce_interp4(const float* s_l, int a, int b, int c, int d, int k4, const UpceW& w, f32x2 off, f32x2& lo, f32x2& hi) {
    const f32x4 ta = ((const f32x4*)(s_l + a))[k4];
    const f32x4 tb = ((const f32x4*)(s_l + b))[k4], tc = ((const f32x4*)(s_l + c))[k4], td = ((const f32x4*)(s_l + d))[k4];
    lo = pk_fma(UPCE_LO(td), w.w11, pk_fma(UPCE_LO(tc), w.w10, pk_fma(UPCE_LO(tb), w.w01, pk_fma(UPCE_LO(ta), w.w00, off))));
    hi = pk_fma(UPCE_HI(td), w.w11, pk_fma(UPCE_HI(tc), w.w10, pk_fma(UPCE_HI(tb), w.w01, pk_fma(UPCE_HI(ta), w.w00, off))));
}

// grid M * ceil(H/16) * ceil(W/16), 256 threads = 16 x 16 output pixels; dynamic LDS (rn*cn*(KP+1) + 256) floats.
// lse[m][y][x]; part[block][0] = sum of per-pixel losses, part[block][1] = number of pixels whose argmax is the label.
// Round 2: (a) staging walks (cell, class) with the index arithmetic hoisted out of the loop -- thread = (cell, class residue), so the
// LDS writes of a wave spread over the banks (consecutive classes) instead of eight banks (consecutive cells, stride KP), and the
// per-cell class maximum falls out of the staged values (no serial 124-read pass); (b) the class loop keeps the running maximum
// with v_max3 and remembers only the GROUP that raised it (the arg-max is recovered from that one group afterwards, first
// occurrence as before), and folds bound and log2 e into one FMA in front of v_exp_f32: 28 instead of ~40 instructions per
// (pixel, class group); PMC had the kernel VALU-bound with 2.8 k instructions per wave, 60 % of them outside the class loop.
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
    const int KP = UPCE_KP(G.K), cells = G.rn * G.cn;
    float* s_max = s_l + cells * KP;                  // per cell: max over classes
    float* s_pm = s_max + cells;                      // per (cell, class residue): the staging thread's maximum (cells * kper entries)
    {
        // class residues per cell: all 256 threads on one pass of the cells for [K][cell] logits (consecutive threads = consecutive
        // cells of one class: 24-byte runs), 32 for token rows (consecutive threads = consecutive classes of one cell: 128-byte runs)
        const int kper = G.ps != 1 ? UPCE_ROWS_KPER : (cells >= 256 ? 1 : 256 / cells), kofs = threadIdx.x % kper, rstep = 256 / kper;
        const float* base = logits + upce_map_base(G, m);
        for (int rc = threadIdx.x / kper; rc < cells; rc += rstep) {
            const int r = rc / G.cn, c = rc - r * G.cn;
            const int rr = r0 + r < G.h ? r0 + r : G.h - 1, cc = c0 + c < G.w ? c0 + c : G.w - 1;
            const float* src = base + (long)(rr * G.w + cc) * G.ps;
            float* dst = s_l + rc * KP;
            float mx = -3.0e38f;
            for (int k = kofs; k < KP; k += kper) {
                float v = -1.0e30f;
                if (k < G.K) { v = src[(long)k * G.ks]; mx = fmaxf(mx, v); }
                dst[k] = v;
            }
            if (kper > 1) s_pm[rc * kper + kofs] = mx; else s_max[rc] = mx;
        }
        __syncthreads();
        if (kper > 1) {
            for (int e = threadIdx.x; e < cells; e += 256) {
                float mx = s_pm[e * kper];
                for (int i = 1; i < kper; ++i) mx = fmaxf(mx, s_pm[e * kper + i]);
                s_max[e] = mx;
            }
            __syncthreads();
        }
    }
    const int oy = ty + (threadIdx.x >> 4), ox = tx + (threadIdx.x & 15);
    float loss = 0.f, hit = 0.f;
    if (oy < G.H && ox < G.W) {
        int y0, y1, x0, x1;
        float ly, lx;
        segf_taps(oy, G.h, G.H, y0, y1, ly);
        segf_taps(ox, G.w, G.W, x0, x1, lx);
        const int ca = (y0 - r0) * G.cn + (x0 - c0), cb = (y0 - r0) * G.cn + (x1 - c0), cc = (y1 - r0) * G.cn + (x0 - c0),
                  cd = (y1 - r0) * G.cn + (x1 - c0);
        const int a = ca * KP, b = cb * KP, c = cc * KP, d = cd * KP;
        const float hx0 = 1.f - lx, hy0 = 1.f - ly;
        const long long lab = labels[(upce_label_map(G, m) * G.H + oy) * G.W + ox];
        const bool counted = lab != G.ignore && lab >= 0 && lab < G.K;
        // an interpolated logit is a convex combination of its four taps, so the largest tap value over all classes bounds every
        // one of them: exponentials relative to that bound need no running rescale
        const float bound = fmaxf(fmaxf(s_max[ca], s_max[cb]), fmaxf(s_max[cc], s_max[cd]));
        // the class loop works on u = (v - bound) log2 e: the factor rides on the four weights, the offset on the first FMA (2^u is
        // the exponential it needs; u orders the classes as v does)
        const float nb = -bound * UPCE_LOG2E;
        f32x2 slo = (f32x2){0.f, 0.f}, shi = (f32x2){0.f, 0.f};
        float best = -3.0e38f;
        int grp = 0;
        const UpceW wts = upce_weights(hx0, lx, hy0, ly);
        const UpceW wl2 = upce_weights(hx0 * UPCE_LOG2E, lx * UPCE_LOG2E, hy0, ly);
        const f32x2 zero2 = (f32x2){0.f, 0.f}, nb2 = (f32x2){nb, nb};
        // one class group: interpolate, running maximum (+ the first group that reaches it), exponentials relative to the bound
#define UPCE_FWD_GROUP(k4_)                                                                                             \
        {                                                                                                               \
            f32x2 vlo, vhi;                                                                                             \
            upce_interp4(s_l, a, b, c, d, (k4_), wl2, nb2, vlo, vhi);                                                   \
            const float nbest = upce_max3(upce_max3(best, vlo[0], vlo[1]), vhi[0], vhi[1]);                             \
            grp = nbest > best ? (k4_) : grp;                                                                           \
            best = nbest;                                                                                               \
            slo += (f32x2){fast_exp2(vlo[0]), fast_exp2(vlo[1])};                                                       \
            shi += (f32x2){fast_exp2(vhi[0]), fast_exp2(vhi[1])};                                                       \
        }
        int k4 = 0;
        for (; k4 + 1 < KP / 4; k4 += 2) {            // two groups per pass: eight LDS reads in flight
            UPCE_FWD_GROUP(k4)
            UPCE_FWD_GROUP(k4 + 1)
        }
        if (k4 < KP / 4) UPCE_FWD_GROUP(k4)
#undef UPCE_FWD_GROUP
        int arg;
        {   // the arg-max: first class of group grp that equals the maximum (same expression, same bits)
            f32x2 vlo, vhi;
            upce_interp4(s_l, a, b, c, d, grp, wl2, nb2, vlo, vhi);
            arg = 4 * grp + (vlo[0] == best ? 0 : vlo[1] == best ? 1 : vhi[0] == best ? 2 : 3);
        }
        float sum = (slo[0] + slo[1]) + (shi[0] + shi[1]), mx = bound;
        if (!(sum > 1e-30f)) {      // taps disagreeing by more than ~70 in some class: the bound is too far above; use the maximum
            mx = -3.0e38f; sum = 0.f;
            for (int k4 = 0; k4 < KP / 4; ++k4) {
                f32x2 vlo, vhi;
                upce_interp4(s_l, a, b, c, d, k4, wts, zero2, vlo, vhi);
                mx = fmaxf(fmaxf(mx, fmaxf(vlo[0], vlo[1])), fmaxf(vhi[0], vhi[1]));
            }
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
        const long long lab = labels[(upce_label_map(G, m) * G.H + fy0 + fy) * G.W + fx0 + fx];
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
    const float sc = scale * (gscale ? *gscale : 1.f) * (G.map_scale ? G.map_scale[m] : 1.f);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * (cl + 16 * i) + j;
            if (k < G.K) dlogits[(((long)m * G.K + k) * G.h + qy) * G.w + qx] = sc * (j < 2 ? acc_lo[i][j & 1] : acc_hi[i][j & 1]);
        }
}


// ---- backward, block form (round 2; the default) ----------------------------------------------------------------------------
// The gather form above recomputes the soft-max of an output pixel in each of the (up to) four low-resolution pixels it taps: 4 x the
// forward's (pixel, class) evaluations, each with four LDS reads, and the kernel is VALU-bound on them.  Here the work is cut by
// BLOCKS instead: block (r, c) = the output pixels whose bilinear taps are exactly the cells (r, r+1) x (c, c+1) (4 x 4 pixels for 4 x
// upsampling).  A thread owns one block column c and four classes, keeps the four cells' logits in REGISTERS, walks down the block
// rows and evaluates p - onehot once per pixel, adding it into four corner accumulators; the bottom corners of block row r are the top
// corners of block row r + 1, and the right-hand corners travel one lane to the right (the cell's owner) with a 16-lane shuffle.
// A workgroup = 16 block columns (15 owned cell columns + the left ring) x 16 class groups (the grid's fastest index walks the class
// chunks of a tile), over `ty` owned cell rows + the ring row above: (16 / 15) (ty + 1) / ty evaluations per output pixel instead of 4, no LDS traffic for logits, no tap tables.
// Deterministic: every accumulator has one owner and a fixed order.
#define UPCE_BLK_COLS 16
#define UPCE_BLK_THREADS 256
struct UpceBlkGeom { int ty; int fcap; int xcap; int ycap; };   // owned cell rows per workgroup; LDS capacity of the staged footprint / one row / one column
// first output index whose lower tap is >= c (the forward's own tap function decides, so the block edges are exactly its edges)
__device__ __forceinline__ int upce_run_start(int c, int in, int out) {
    if (c <= 0) return 0;
    if (c >= in) return out;
    int f = (int)ceilf(((float)c + 0.5f) * ((float)out / (float)in) - 0.5f), i0, i1;
    float l1;
    f = f < 0 ? 0 : (f > out ? out : f);
    while (f > 0) { segf_taps(f - 1, in, out, i0, i1, l1); if (i0 >= c) --f; else break; }
    while (f < out) { segf_taps(f, in, out, i0, i1, l1); if (i0 < c) ++f; else break; }
    return f;
}
// packed fp32 with one operand broadcast from a register pair's low / high half (VOP3P op_sel: no v_mov to build (w, w) pairs)
#ifdef CFFM_EMU
#define UPCE_PK3(name, expr) __device__ __forceinline__ f32x2 name(f32x2 a, f32x2 b, f32x2 c) { return expr; }
#define UPCE_PK2(name, expr) __device__ __forceinline__ f32x2 name(f32x2 a, f32x2 b) { return expr; }
UPCE_PK3(pk_fma_blo, (a * (f32x2){b[0], b[0]} + c))
UPCE_PK3(pk_fma_bhi, (a * (f32x2){b[1], b[1]} + c))
UPCE_PK3(pk_fma_blo_clo, (a * (f32x2){b[0], b[0]} + (f32x2){c[0], c[0]}))
UPCE_PK2(pk_mul_blo, (a * (f32x2){b[0], b[0]}))
UPCE_PK2(pk_mul_bhi, (a * (f32x2){b[1], b[1]}))
UPCE_PK2(pk_add_ahi, ((f32x2){a[1], a[1]} + b))
__device__ __forceinline__ f32x2 pk_onehot(f32x2 t, f32x2 one) {          // 1 where t == 0, 0 where |t| >= 1
    f32x2 r = one - t * t;
    r[0] = r[0] < 0.f ? 0.f : r[0]; r[1] = r[1] < 0.f ? 0.f : r[1];
    return r;
}
#else
#define UPCE_PK3(name, mods) __device__ __forceinline__ f32x2 name(f32x2 a, f32x2 b, f32x2 c) { \
    f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 " mods : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
#define UPCE_PK2(name, op, mods) __device__ __forceinline__ f32x2 name(f32x2 a, f32x2 b) { \
    f32x2 d; asm(op " %0, %1, %2 " mods : "=v"(d) : "v"(a), "v"(b)); return d; }
UPCE_PK3(pk_fma_blo, "op_sel:[0,0,0] op_sel_hi:[1,0,1]")
UPCE_PK3(pk_fma_bhi, "op_sel:[0,1,0] op_sel_hi:[1,1,1]")
UPCE_PK3(pk_fma_blo_clo, "op_sel:[0,0,0] op_sel_hi:[1,0,0]")
UPCE_PK2(pk_mul_blo, "v_pk_mul_f32", "op_sel:[0,0] op_sel_hi:[1,0]")
UPCE_PK2(pk_mul_bhi, "v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[1,1]")
UPCE_PK2(pk_add_ahi, "v_pk_add_f32", "op_sel:[1,0] op_sel_hi:[1,1]")
__device__ __forceinline__ f32x2 pk_onehot(f32x2 t, f32x2 one) {          // clamp(1 - t t): 1 where t == 0, 0 where |t| >= 1
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %1, %2 neg_lo:[1,0,0] neg_hi:[1,0,0] clamp" : "=v"(d) : "v"(t), "v"(one));
    return d;
}
#endif
struct UpceBlkCells { f32x2 t0l, t0h, t1l, t1h, b0l, b0h, b1l, b1h; };     // four classes of the block's four cells (x log2 e)
struct UpceBlkAcc { f32x2 t0[2], t1[2], b0[2], b1[2]; };                   // ... and of the four corner accumulators
// one output pixel x four classes: interpolate, p - onehot, into the four corners with the pixel's own bilinear weights.
// wt = (w00, w01), wb = (w10, w11); px = (-lse log2 e, label as a float); clo / chi = -(4 k4 + j) for the thread's four classes
__device__ __forceinline__ void upce_blk_eval(const UpceBlkCells& c, f32x2 wt, f32x2 wb, f32x2 px, f32x2 clo, f32x2 chi, f32x2 one, UpceBlkAcc& a) {
    const f32x2 vlo = pk_fma_bhi(c.b1l, wb, pk_fma_blo(c.b0l, wb, pk_fma_bhi(c.t1l, wt, pk_fma_blo_clo(c.t0l, wt, px))));
    const f32x2 vhi = pk_fma_bhi(c.b1h, wb, pk_fma_blo(c.b0h, wb, pk_fma_bhi(c.t1h, wt, pk_fma_blo_clo(c.t0h, wt, px))));
    // (the subtraction is left to the compiler: an inline-asm consumer right behind v_exp_f32 is invisible to its hazard recognizer --
    // gfx950 needs a wait state between a transcendental and the VALU instruction that reads its result)
    const f32x2 plo = (f32x2){UPCE_EXP(vlo[0]), UPCE_EXP(vlo[1])} - pk_onehot(pk_add_ahi(px, clo), one);
    const f32x2 phi = (f32x2){UPCE_EXP(vhi[0]), UPCE_EXP(vhi[1])} - pk_onehot(pk_add_ahi(px, chi), one);
    a.t0[0] = pk_fma_blo(plo, wt, a.t0[0]); a.t0[1] = pk_fma_blo(phi, wt, a.t0[1]);
    a.t1[0] = pk_fma_bhi(plo, wt, a.t1[0]); a.t1[1] = pk_fma_bhi(phi, wt, a.t1[1]);
    a.b0[0] = pk_fma_blo(plo, wb, a.b0[0]); a.b0[1] = pk_fma_blo(phi, wb, a.b0[1]);
    a.b1[0] = pk_fma_bhi(plo, wb, a.b1[0]); a.b1[1] = pk_fma_bhi(phi, wb, a.b1[1]);
}
__global__ void __launch_bounds__(UPCE_BLK_THREADS) k_upce_bwd_blk(const float* __restrict__ logits, const long long* __restrict__ labels,
                                                                    const float* __restrict__ lse, const float* __restrict__ gscale, float scale,
                                                                    float* __restrict__ dlogits, UpceGeom G, UpceBlkGeom Bk) {
    CFFM_DYN_SMEM(smem);
    const int KP = UPCE_KP(G.K), ng = KP / 4, TX = UPCE_BLK_COLS - 1, TY = Bk.ty;
    const int gx = (G.w + TX - 1) / TX, gy = (G.h + TY - 1) / TY;
    const int nch = (ng + UPCE_BLK_THREADS / UPCE_BLK_COLS - 1) / (UPCE_BLK_THREADS / UPCE_BLK_COLS);      // class chunks
    const int lin0 = xcd_linear_id(), chunk = lin0 % nch, lin = lin0 / nch, m = lin / (gx * gy), trem = lin - m * gx * gy;
    const int q0y = (trem / gx) * TY, q0x = (trem - (trem / gx) * gx) * TX;
    // the workgroup's block rows [rlo, rhi] / columns [clo, chi] and their footprint [f0, f1) in the label map
    const int rlo = q0y > 0 ? q0y - 1 : 0, rhi = q0y + TY - 1 < G.h - 1 ? q0y + TY - 1 : G.h - 1;
    const int clo = q0x > 0 ? q0x - 1 : 0, chi = q0x + TX - 1 < G.w - 1 ? q0x + TX - 1 : G.w - 1;
    f32x2* s_px = (f32x2*)smem;                      // per footprint pixel: (-lse log2 e, label as a float)
    f32x2* s_hx = s_px + Bk.fcap;                    // per footprint column: (1 - lambda, lambda)
    f32x2* s_hy = s_hx + Bk.xcap;                    // per footprint row
    int* s_ya = (int*)(s_hy + Bk.ycap);              // first footprint row of block rows rlo .. rhi + 1
    int* s_xa = s_ya + TY + 3;                       // ... column of block columns clo .. chi + 1
    for (int e = threadIdx.x; e <= rhi - rlo + 1; e += UPCE_BLK_THREADS) s_ya[e] = upce_run_start(rlo + e, G.h, G.H);
    for (int e = threadIdx.x; e <= chi - clo + 1; e += UPCE_BLK_THREADS) s_xa[e] = upce_run_start(clo + e, G.w, G.W);
    __syncthreads();
    const int fy0 = s_ya[0], fy1 = s_ya[rhi - rlo + 1], fx0 = s_xa[0], fx1 = s_xa[chi - clo + 1];
    const int fw = fx1 - fx0, fh = fy1 - fy0;
    if (fw > Bk.xcap || fh > Bk.ycap || fw * fh > Bk.fcap) return;          // (cannot happen: the host sizes them from the same run bounds + slack)
    for (int e = threadIdx.x; e < fh * fw; e += UPCE_BLK_THREADS) {
        const int fy = e / fw, fx = e - fy * fw;
        const long pix = ((long)m * G.H + fy0 + fy) * G.W + fx0 + fx;
        const long long lab = labels[(upce_label_map(G, m) * G.H + fy0 + fy) * G.W + fx0 + fx];
        const bool ign = lab == G.ignore || lab < 0 || lab >= G.K;
        // ignored pixel: exp2(v - 1e30) = 0 and no class matches -1
        s_px[e] = ign ? (f32x2){-1.0e30f, -1.f} : (f32x2){-lse[pix] * UPCE_LOG2E, (float)(int)lab};
    }
    for (int e = threadIdx.x; e < fw + fh; e += UPCE_BLK_THREADS) {
        int i0, i1;
        float l1;
        if (e < fw) { segf_taps(fx0 + e, G.w, G.W, i0, i1, l1); s_hx[e] = (f32x2){1.f - l1, l1}; }
        else        { segf_taps(fy0 + e - fw, G.h, G.H, i0, i1, l1); s_hy[e - fw] = (f32x2){1.f - l1, l1}; }
    }
    __syncthreads();
    const int kl = threadIdx.x >> 4, bc = threadIdx.x & (UPCE_BLK_COLS - 1), c = q0x - 1 + bc;
    const int k4 = chunk * (UPCE_BLK_THREADS / UPCE_BLK_COLS) + kl;
    const bool act = c >= 0 && c <= G.w - 1 && k4 < ng;
    const int c1 = c + 1 < G.w ? c + 1 : G.w - 1;
    const int xa = act ? s_xa[c - clo] : 0, xb = act ? s_xa[c - clo + 1] : 0;
    const int ks = G.ks, ps = G.ps;
    const float sc = scale * (gscale ? *gscale : 1.f) * (G.map_scale ? G.map_scale[m] : 1.f);
    const float* base = logits + upce_map_base(G, m) + (long)4 * k4 * ks;
    float* obase = dlogits + upce_map_base(G, m) + (long)4 * k4 * ks;
    const f32x2 one = (f32x2){1.f, 1.f}, cls_lo = (f32x2){-(float)(4 * k4), -(float)(4 * k4 + 1)}, cls_hi = (f32x2){-(float)(4 * k4 + 2), -(float)(4 * k4 + 3)};
    float t0[4], t1[4], b0[4], b1[4], n0[4], n1[4];
    UpceBlkAcc acc;
#pragma unroll
    for (int i = 0; i < 2; ++i) acc.t0[i] = acc.t1[i] = acc.b0[i] = acc.b1[i] = (f32x2){0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) t0[j] = t1[j] = b0[j] = b1[j] = n0[j] = n1[j] = -1.0e30f;
    // cells of low-resolution row r (clamped) into (x0, x1), scaled by log2 e; classes beyond K stay at -1e30 (exp -> 0)
#define UPCE_BLK_LOAD(x0, x1, row)                                                                      \
    if (act) {                                                                                          \
        const int rr_ = (row) < G.h ? (row) : G.h - 1;                                                  \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                   \
            if (4 * k4 + j < G.K) { x0[j] = base[j * ks + (rr_ * G.w + c) * ps] * UPCE_LOG2E; x1[j] = base[j * ks + (rr_ * G.w + c1) * ps] * UPCE_LOG2E; } \
    }
    UPCE_BLK_LOAD(t0, t1, rlo)
    UPCE_BLK_LOAD(b0, b1, rlo + 1)
    for (int r = rlo; r <= rhi; ++r) {
        UPCE_BLK_LOAD(n0, n1, r + 2)                 // the row below the next block row, in flight during this row's evaluations
        const int ya = wave_uniform(s_ya[r - rlo]), yb = wave_uniform(s_ya[r - rlo + 1]);
        if (act) {
            const UpceBlkCells cells = {(f32x2){t0[0], t0[1]}, (f32x2){t0[2], t0[3]}, (f32x2){t1[0], t1[1]}, (f32x2){t1[2], t1[3]},
                                        (f32x2){b0[0], b0[1]}, (f32x2){b0[2], b0[3]}, (f32x2){b1[0], b1[1]}, (f32x2){b1[2], b1[3]}};
            // two footprint rows per pass (independent evaluations to overlap the LDS reads and the quarter-rate exponentials)
            int fy = ya;
            for (; fy + 1 < yb; fy += 2) {
                const f32x2 hya = s_hy[fy - fy0], hyb = s_hy[fy + 1 - fy0];
                const f32x2* row = s_px + (fy - fy0) * fw - fx0;
                for (int fx = xa; fx < xb; ++fx) {
                    const f32x2 hx = s_hx[fx - fx0], pa = row[fx], pb = row[fw + fx];
                    upce_blk_eval(cells, pk_mul_blo(hx, hya), pk_mul_bhi(hx, hya), pa, cls_lo, cls_hi, one, acc);
                    upce_blk_eval(cells, pk_mul_blo(hx, hyb), pk_mul_bhi(hx, hyb), pb, cls_lo, cls_hi, one, acc);
                }
            }
            if (fy < yb) {
                const f32x2 hy = s_hy[fy - fy0];
                const f32x2* row = s_px + (fy - fy0) * fw - fx0;
                for (int fx = xa; fx < xb; ++fx) {
                    const f32x2 hx = s_hx[fx - fx0];
                    upce_blk_eval(cells, pk_mul_blo(hx, hy), pk_mul_bhi(hx, hy), row[fx], cls_lo, cls_hi, one, acc);
                }
            }
        }
        if (r == G.h - 1) {          // last cell row: both taps of its pixels are this row
#pragma unroll
            for (int i = 0; i < 2; ++i) { acc.t0[i] += acc.b0[i]; acc.t1[i] += acc.b1[i]; }
        }
        // cell (r, c) = this thread's left-hand corners + the right-hand corners of the block column to the left
        float out[4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float recv = __shfl(acc.t1[i][j], (bc + UPCE_BLK_COLS - 1) & (UPCE_BLK_COLS - 1), UPCE_BLK_COLS);
                out[2 * i + j] = acc.t0[i][j] + (bc > 0 ? recv : 0.f) + (c1 == c ? acc.t1[i][j] : 0.f);
            }
        if (r >= q0y && act && bc > 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * k4 + j < G.K) obase[j * ks + (r * G.w + c) * ps] = sc * out[j];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) { acc.t0[i] = acc.b0[i]; acc.t1[i] = acc.b1[i]; acc.b0[i] = acc.b1[i] = (f32x2){0.f, 0.f}; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { t0[j] = b0[j]; t1[j] = b1[j]; b0[j] = n0[j]; b1[j] = n1[j]; }
    }
#undef UPCE_BLK_LOAD
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

// The loss scalars on the device (round 6): out[0] = sum_m wl[m] * sum_j part[m][j][0] (the weighted cross-entropy sum), out[1] = sum_m wh[m] *
// sum_j part[m][j][1] (the weighted hit count) from the per-workgroup records of k_upce_fwd, in double, in a fixed order (thread t sums records
// t, t + 256, ... of every map, then a tree over the 256 partial sums): what ~10 torch kernels (view / double / sum / mul / sum / float, twice)
// did after every forward -- 60 us of a replayed head step.  One workgroup.
__global__ void __launch_bounds__(256) k_upce_finalize(const float* __restrict__ part, int M, int per, const double* __restrict__ wl,
                                                        const double* __restrict__ wh, float* __restrict__ out) {
    __shared__ double red[2][256];
    const int t = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int m = 0; m < M; ++m) {
        double sa = 0.0, sb = 0.0;
        for (int j = t; j < per; j += 256) {
            const f32x2 v = *(const f32x2*)(part + 2 * ((long)m * per + j));
            sa += (double)v[0];
            sb += (double)v[1];
        }
        a += sa * wl[m];
        b += sb * wh[m];
    }
    red[0][t] = a;
    red[1][t] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) { red[0][t] += red[0][t + s]; red[1][t] += red[1][t + s]; }
        __syncthreads();
    }
    if (t == 0) { out[0] = (float)red[0][0]; out[1] = (float)red[1][0]; }
}

