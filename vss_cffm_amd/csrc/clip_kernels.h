// clip_kernels.h -- the clip data path after image decoding, as ONE pass over the clip on the device.
//
// Reference (CPU, numpy / cv2 through mmcv, frame by frame inside the dataloader workers): the `*_clips` transforms of
// local_configs/_base_/datasets/vspw_repeat2.py:8-19 --
//   LoadAnnotations(reduce_zero_label)          label 0 -> 255, the rest minus 1 (254 -> 255)   pipelines/loading.py
//   RandomCrop_clips   transforms.py:1524-1600   one crop box for every frame of the clip, img[y1:y2, x1:x2]
//   RandomFlip_clips   transforms.py:852-910     horizontal mirror of the (cropped) frames and label maps
//   PhotoMetricDistortion_clips  transforms.py:2028-2150  per FRAME: random brightness (+beta) and random contrast (*alpha), each
//                      `np.clip(float32(img) * alpha + beta, 0, 255).astype(uint8)` (convert(), :2057-2061); the saturation / hue
//                      branches go through cv2's 8-bit HSV conversion, which this environment cannot pin -- they are drawn (to keep
//                      the random stream in step) and refused or skipped by the host (vss_cffm_amd/data.py), never approximated here
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
};
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
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float v = (float)px[P.to_rgb ? 2 - c : c];
                if (P.photo) {
                    if (P.has_b[t]) v = clip_convert(v, 1.f, P.beta[t]);
                    if (P.has_c[t]) v = clip_convert(v, P.alpha[t], 0.f);
                }
                out_img[o + c * plane] = (v - P.mean[c]) * P.stdinv[c];
            }
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
