// clip_kernels.h -- the clip data path after image decoding, as ONE pass over the clip on the device.
//
// Reference (CPU, numpy / cv2 through mmcv, frame by frame inside the dataloader workers): the `*_clips` transforms of
// local_configs/_base_/datasets/vspw_repeat2.py:8-19 --
//   LoadAnnotations(reduce_zero_label)          label 0 -> 255, the rest minus 1 (254 -> 255)   pipelines/loading.py
//   RandomCrop_clips   transforms.py:1524-1600   one crop box for every frame of the clip, img[y1:y2, x1:x2]
//   RandomFlip_clips   transforms.py:852-910     horizontal mirror of the (cropped) frames and label maps
//   Resize / AlignedResize_clips  transforms.py:475-760 / :236-470  mmcv.imrescale / imresize = cv2.resize: INTER_LINEAR for the frames
//                      (OpenCV's 8-bit path: 11-bit fixed-point weights, two passes), INTER_NEAREST for the label maps -> k_clip_resize
//   PhotoMetricDistortion_clips  transforms.py:2028-2150  per FRAME: random brightness (+beta) and random contrast (*alpha), each
//                      `np.clip(float32(img) * alpha + beta, 0, 255).astype(uint8)` (convert(), :2057-2061); saturation and hue
//                      through cv2's 8-bit BGR <-> HSV conversion (mmcv.bgr2hsv / hsv2bgr): RGB2HSV_b's integer arithmetic and
//                      HSV2RGB_b's float32 arithmetic restated here (round 5).  OpenCV is absent from both boxes, so these two
//                      steps and the resize are checked against oracle/cv_oracle.py only: parity unpinned
//   Normalize_clips    transforms.py:1260-1297   BGR -> RGB, (v - mean) * (1 / std) in float32 (mmcv.imnormalize)
//   Pad_clips          transforms.py:990-1085    bottom / right padding to the crop size: 0 in the (normalised) image, 255 in the labels
//   DefaultFormatBundle_clips  formating.py:261-305   HWC -> CHW, frames stacked: img [T,3,H,W] float32, labels [T,1,H,W] int64
// The random decisions (crop box incl. the cat_max_ratio retries, flip) are drawn on the host exactly as the reference draws them
// (vss_cffm_amd/data.py); this kernel applies them: one thread per output pixel, byte gathers in, three coalesced float planes and
// one int64 plane out.  HBM-bound byte work (2.7 MB in, 11 MB + 7 MB out per 4-frame 480x480 clip).
#pragma once
#include "cffm_common.h"

#define CLIP_MAXT 16
struct ClipFmt {
    int T, H, W;            // input frames [T][H][W][3] uint8 (BGR as decoded), labels [T][H][W] uint8
    int y1, x1, ch, cw;     // crop box: rows y1 .. y1+ch-1, columns x1 .. x1+cw-1 (already clipped to the image)
    int flip;               // mirror the cropped frames horizontally
    int Ho, Wo;             // output size (>= ch, cw: the rest is padding)
    int to_rgb, reduce_zero_label, seg_pad;
    float mean[3], stdinv[3], pad_val;   // in OUTPUT channel order (RGB when to_rgb)
    int photo;              // per-frame brightness / contrast present
    float beta[CLIP_MAXT], alpha[CLIP_MAXT];   // frame t: v = u8(clip(v + beta)) (beta != 0: brightness taken), then v = u8(clip(v * alpha)) (alpha != 1)
    unsigned char has_b[CLIP_MAXT], has_c[CLIP_MAXT];
    // round 5: the two HSV steps, between brightness / early contrast and the late contrast (transforms.py:2121-2137)
    unsigned char c_first[CLIP_MAXT];          // mode == 1: contrast BEFORE saturation / hue
    unsigned char has_s[CLIP_MAXT], has_h[CLIP_MAXT];
    float sat[CLIP_MAXT];                      // S = u8(clip(S * sat))
    int hue[CLIP_MAXT];                        // H = (H + hue) mod 180
};
// ---- OpenCV's 8-bit BGR <-> HSV, hue range 180 (opencv/modules/imgproc/src/color_hsv.cpp: RGB2HSV_b, HSV2RGB_b / HSV2RGB_native) ----
// the two division tables of RGB2HSV_b: cvRound((255 << 12) / i), cvRound((180 << 12) / (6 i)).  Neither quotient is ever a tie
// (2^13 * 255 and 2^14 * 15 have no odd multiple below 256), so round-half-even = floor(x + 1/2) = (2 N + i) / (2 i) in integers.
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
struct HsvDivTab {
    int sdiv[256], hdiv[256];
    constexpr HsvDivTab() : sdiv(), hdiv() {
        for (int i = 1; i < 256; ++i) {
            sdiv[i] = (2 * (255 << 12) + i) / (2 * i);
            hdiv[i] = (2 * ((180 << 12) / 6) + i) / (2 * i);
        }
    }
};
__device__ const HsvDivTab g_hsv_div = HsvDivTab();
__device__ __forceinline__ void bgr2hsv_u8(int b, int g, int r, int& h, int& s, int& v) {
    v = imax(imax(b, g), r);
    const int vmin = imin(imin(b, g), r), diff = v - vmin;
    const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    s = (diff * g_hsv_div.sdiv[v] + (1 << 11)) >> 12;
    h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + (~vg & (r - g + 4 * diff))));
    h = (h * g_hsv_div.hdiv[diff] + (1 << 11)) >> 12;      // (arithmetic shift of a negative h: floor, as in C++)
    h += h < 0 ? 180 : 0;
    h = imin(imax(h, 0), 255);
}
// float32 throughout, every product and difference rounded on its own (no FMA), the result * 255 rounded half to even (cvRound)
__device__ __forceinline__ float f_mul(float a, float b) {
#ifdef CFFM_EMU
    volatile float m = a * b;
    return m;
#else
    return __fmul_rn(a, b);
#endif
}
__device__ __forceinline__ float f_sub(float a, float b) {
#ifdef CFFM_EMU
    volatile float m = a - b;
    return m;
#else
    return __fsub_rn(a, b);
#endif
}
__device__ __forceinline__ int u8_round(float x) {
#ifdef CFFM_EMU
    const int i = (int)nearbyintf(x);
#else
    const int i = __float2int_rn(x);
#endif
    return imin(imax(i, 0), 255);
}
__device__ __forceinline__ void hsv2bgr_u8(int hi, int si, int vi, int& b, int& g, int& r) {
    const float s = f_mul((float)si, 1.0f / 255.0f), v = f_mul((float)vi, 1.0f / 255.0f);
    float fb, fg, fr;
    if (si == 0) {
        fb = fg = fr = v;
    } else {
        float h = f_mul((float)hi, 6.0f / 180.0f);
        h = fmodf(h, 6.f);
        int sector = (int)floorf(h);
        h = f_sub(h, (float)sector);
        if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
        const float t0 = v, t1 = f_mul(v, f_sub(1.f, s)), t2 = f_mul(v, f_sub(1.f, f_mul(s, h))), t3 = f_mul(v, f_sub(1.f, f_mul(s, f_sub(1.f, h))));
        // sector_data: {1,3,0}, {1,0,2}, {3,0,1}, {0,2,1}, {0,1,3}, {2,1,0} = table slots of b, g, r
        switch (sector) {
            case 0: fb = t1; fg = t3; fr = t0; break;
            case 1: fb = t1; fg = t0; fr = t2; break;
            case 2: fb = t3; fg = t0; fr = t1; break;
            case 3: fb = t0; fg = t2; fr = t1; break;
            case 4: fb = t0; fg = t1; fr = t3; break;
            default: fb = t2; fg = t1; fr = t0; break;
        }
    }
    b = u8_round(f_mul(fb, 255.f)); g = u8_round(f_mul(fg, 255.f)); r = u8_round(f_mul(fr, 255.f));
}
// uint8 -> convert(alpha, beta) of the reference: float32 multiply, float32 add (two roundings: no FMA), clip, truncate
__device__ __forceinline__ float clip_convert(float v, float alpha, float beta) {
#ifdef CFFM_EMU
    volatile float m = v * alpha;
    float t = m + beta;
#else
    float t = __fadd_rn(__fmul_rn(v, alpha), beta);
#endif
    t = fminf(fmaxf(t, 0.f), 255.f);
    return (float)(int)t;      // astype(uint8) of a value in [0, 255]: truncation
}
__global__ void __launch_bounds__(256) k_clip_format(const unsigned char* __restrict__ frames, const unsigned char* __restrict__ labels,
                                                      float* __restrict__ out_img, long long* __restrict__ out_lab, ClipFmt P) {
    const long n = (long)P.T * P.Ho * P.Wo;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const int ox = (int)(e % P.Wo), oy = (int)((e / P.Wo) % P.Ho), t = (int)(e / ((long)P.Wo * P.Ho));
        const bool inside = oy < P.ch && ox < P.cw;
        const int sy = P.y1 + oy, sx = P.x1 + (P.flip ? P.cw - 1 - ox : ox);
        const long plane = (long)P.Ho * P.Wo, o = (long)t * 3 * plane + (long)oy * P.Wo + ox;
        if (inside) {
            const unsigned char* px = frames + (((long)t * P.H + sy) * P.W + sx) * 3;
            float v3[3] = {(float)px[0], (float)px[1], (float)px[2]};       // B, G, R as decoded
            if (P.photo) {
                if (P.has_b[t])
                    for (int c = 0; c < 3; ++c) v3[c] = clip_convert(v3[c], 1.f, P.beta[t]);
                if (P.has_c[t] && P.c_first[t])
                    for (int c = 0; c < 3; ++c) v3[c] = clip_convert(v3[c], P.alpha[t], 0.f);
                if (P.has_s[t]) {
                    int h, sv, vv, b, g, r;
                    bgr2hsv_u8((int)v3[0], (int)v3[1], (int)v3[2], h, sv, vv);
                    sv = (int)clip_convert((float)sv, P.sat[t], 0.f);
                    hsv2bgr_u8(h, sv, vv, b, g, r);
                    v3[0] = (float)b; v3[1] = (float)g; v3[2] = (float)r;
                }
                if (P.has_h[t]) {
                    int h, sv, vv, b, g, r;
                    bgr2hsv_u8((int)v3[0], (int)v3[1], (int)v3[2], h, sv, vv);
                    h = (h + P.hue[t]) % 180;
                    if (h < 0) h += 180;                     // Python's % on a negative sum
                    hsv2bgr_u8(h, sv, vv, b, g, r);
                    v3[0] = (float)b; v3[1] = (float)g; v3[2] = (float)r;
                }
                if (P.has_c[t] && !P.c_first[t])
                    for (int c = 0; c < 3; ++c) v3[c] = clip_convert(v3[c], P.alpha[t], 0.f);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) out_img[o + c * plane] = (v3[P.to_rgb ? 2 - c : c] - P.mean[c]) * P.stdinv[c];
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) out_img[o + c * plane] = P.pad_val;
        }
        if (out_lab) {
            int l = P.seg_pad;
            if (inside) {
                l = labels[((long)t * P.H + sy) * P.W + sx];
                if (P.reduce_zero_label) {          // uint8 arithmetic of LoadAnnotations: 0 -> 255, v -> v - 1, 254 -> 255
                    if (l == 0) l = 255;
                    l -= 1;
                    if (l == 254) l = 255;
                }
            }
            out_lab[(long)t * plane + (long)oy * P.Wo + ox] = l;
        }
    }
}

// ---- cv2.resize of a clip: INTER_LINEAR for the frames, INTER_NEAREST for the label maps (opencv/modules/imgproc/src/resize.cpp) ----
// 8-bit INTER_LINEAR = two passes in fixed point: the horizontal one leaves S[sx] a0 + S[sx + 1] a1 with a = cvRound(w * 2048) as
// shorts (w from fx = (float)((dx + 0.5) * scale - 0.5), sx = floor(fx), fx -= sx; sx < 0 -> (0, fx 0), sx >= W - 1 -> (W - 1, fx 0)),
// the vertical one ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2 with the rows sy, sy + 1 clamped to the image and b
// NOT adjusted; a 2x2 -> 1 shrink in both directions is INTER_AREA's box mean (a + b + c + d + 2) >> 2.  INTER_NEAREST: source index
// min(floor(d * (1 / (D / S))), S - 1).  The scales are doubles computed once on the host (the same values in the emulator).
struct ClipResize {
    int T, H, W, Ho, Wo;
    double scale_x, scale_y;      // 1. / (Wo / (double)W), 1. / (Ho / (double)H)
    int half;                     // W == 2 Wo && H == 2 Ho: the box mean
};
__device__ __forceinline__ void resize_axis(int d, double scale, int n, bool clamp_w, int& s0, int& s1, int& w0, int& w1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f = f_sub(f, (float)s);
    if (clamp_w) {                                  // horizontal: the weights follow the border clamp
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= n - 1) { f = 0.f; s = n - 1; }
    }
#ifdef CFFM_EMU
    w0 = (int)nearbyintf(f_mul(f_sub(1.f, f), 2048.f)); w1 = (int)nearbyintf(f_mul(f, 2048.f));
#else
    w0 = __float2int_rn(f_mul(f_sub(1.f, f), 2048.f)); w1 = __float2int_rn(f_mul(f, 2048.f));
#endif
    s0 = imin(imax(s, 0), n - 1); s1 = imin(imax(s + 1, 0), n - 1);
}
__global__ void __launch_bounds__(256) k_clip_resize(const unsigned char* __restrict__ frames, const unsigned char* __restrict__ labels,
                                                      unsigned char* __restrict__ out_frames, unsigned char* __restrict__ out_labels, ClipResize P) {
    const long n = (long)P.T * P.Ho * P.Wo;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const int ox = (int)(e % P.Wo), oy = (int)((e / P.Wo) % P.Ho), t = (int)(e / ((long)P.Wo * P.Ho));
        if (frames) {
            const unsigned char* src = frames + (long)t * P.H * P.W * 3;
            unsigned char* dst = out_frames + e * 3;
            if (P.half) {
                const unsigned char *p0 = src + ((long)(2 * oy) * P.W + 2 * ox) * 3, *p1 = p0 + (long)P.W * 3;
                for (int c = 0; c < 3; ++c) dst[c] = (unsigned char)((p0[c] + p0[3 + c] + p1[c] + p1[3 + c] + 2) >> 2);
            } else {
                int x0, x1, a0, a1, y0, y1, b0, b1;
                resize_axis(ox, P.scale_x, P.W, true, x0, x1, a0, a1);
                resize_axis(oy, P.scale_y, P.H, false, y0, y1, b0, b1);
                const unsigned char *r0 = src + (long)y0 * P.W * 3, *r1 = src + (long)y1 * P.W * 3;
                for (int c = 0; c < 3; ++c) {
                    const int h0 = r0[x0 * 3 + c] * a0 + r0[x1 * 3 + c] * a1, h1 = r1[x0 * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
                    const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                    dst[c] = (unsigned char)imin(imax(v, 0), 255);
                }
            }
        }
        if (labels) {
            const int sx = imin((int)floor((double)ox * P.scale_x), P.W - 1), sy = imin((int)floor((double)oy * P.scale_y), P.H - 1);
            out_labels[e] = labels[((long)t * P.H + sy) * P.W + sx];
        }
    }
}
