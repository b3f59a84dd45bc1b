"""CPU restatement (numpy) of the OpenCV arithmetic behind the reference's resize and HSV steps -- TEST INFRASTRUCTURE.

Only tests/ and __graft_entry__.smoke() may import this module; the product (vss_cffm_amd/) never does.

The reference's clip pipeline (local_configs/_base_/datasets/vspw_repeat2.py:10,27) resizes through ``mmcv.imrescale`` /
``mmcv.imresize`` (mmseg/datasets/pipelines/transforms.py:236-470 ``AlignedResize_clips``, :475-760 ``Resize``) and distorts
saturation / hue through ``mmcv.bgr2hsv`` / ``mmcv.hsv2bgr`` (:2082-2110); mmcv forwards all four to OpenCV (``cv2.resize`` with
``INTER_LINEAR`` / ``INTER_NEAREST``, ``cv2.cvtColor`` with ``COLOR_BGR2HSV`` / ``COLOR_HSV2BGR``) on uint8 arrays.  mmcv and
OpenCV are third-party dependencies that are absent from /root/reference and from both boxes (requirements: mmcv-full 1.x, which
pins opencv-python >= 3), so this file restates OpenCV's PUBLISHED 8-bit algorithms (opencv/modules/imgproc/src/resize.cpp:
``resizeGeneric_`` with ``HResizeLinear<uchar,int,short,2048>`` + ``VResizeLinear<uchar,int,short,FixedPtCast<22>>``, ``resizeNN``,
the 2x2 ``INTER_AREA`` substitution; color_hsv.cpp: ``RGB2HSV_b``, ``HSV2RGB_b`` / ``HSV2RGB_native``), and mmcv's size rules
(mmcv/image/geometric.py ``rescale_size`` / ``_scale_size``).

**Parity unpinned**: there is no cv2 here to produce golden vectors, and the reference's own tests hold none for these steps.  The
checks that exist are internal (tests/test_data_cv.py): the HIP kernels equal this restatement bit for bit; the restatement agrees
with a float64 formulation of bilinear interpolation within 1 LSB and with ``colorsys`` within the 8-bit quantisation; identities
(size-preserving resize, grey pixels, hue shift by 0 after a round trip) hold exactly.
"""
import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def _round_half_even_f32(x):
    """cvRound of float32 values (saturate_cast<short/uchar> of a float): round half to even."""
    return np.rint(np.asarray(x, dtype=np.float32)).astype(np.int64)


def rescale_size(w, h, scale):
    """mmcv.rescale_size((w, h), scale) for a (long, short) tuple or a float factor -> (new_w, new_h)."""
    if isinstance(scale, (float, int)):
        f = float(scale)
    else:
        f = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return int(w * float(f) + 0.5), int(h * float(f) + 0.5)


def _linear_axis(dst, src):
    """offsets and the two 11-bit fixed-point weights of every destination index along one axis (resize.cpp, INTER_LINEAR:
    fx = (float)((dx + 0.5) * scale - 0.5); sx = floor(fx); fx -= sx; border clamps for the HORIZONTAL axis are applied by the caller)."""
    scale = 1.0 / (dst / float(src))                     # scale_x = 1. / inv_scale_x, in double
    f = (((np.arange(dst, dtype=np.float64) + 0.5) * scale) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def resize_linear_u8(img, dw, dh):
    """cv2.resize(img uint8 [H,W] or [H,W,C], (dw, dh), interpolation=cv2.INTER_LINEAR)."""
    img = np.asarray(img)
    assert img.dtype == np.uint8
    sh, sw = img.shape[:2]
    if dw == sw and dh == sh:
        return img.copy()
    src = img.reshape(sh, sw, -1).astype(np.int64)
    if sw == 2 * dw and sh == 2 * dh:
        # "in case of scale_x && scale_y is equal to 2 INTER_AREA (fast) also is equal to INTER_LINEAR": the 2x2 box mean, rounded
        out = (src[0::2, 0::2] + src[0::2, 1::2] + src[1::2, 0::2] + src[1::2, 1::2] + 2) >> 2
        return out.astype(np.uint8).reshape((dh, dw) + img.shape[2:])
    sx, fx = _linear_axis(dw, sw)
    lo = sx < 0
    fx = np.where(lo, np.float32(0), fx); sx = np.where(lo, 0, sx)
    hi = sx >= sw - 1
    fx = np.where(hi, np.float32(0), fx); sx = np.where(hi, sw - 1, sx)
    a0 = np.clip(_round_half_even_f32((np.float32(1) - fx) * np.float32(COEF_SCALE)), -32768, 32767)
    a1 = np.clip(_round_half_even_f32(fx * np.float32(COEF_SCALE)), -32768, 32767)
    sx1 = np.minimum(sx + 1, sw - 1)                       # (weight 0 wherever the clamp bites)
    rows = src[:, sx, :] * a0[None, :, None] + src[:, sx1, :] * a1[None, :, None]          # [sh, dw, C], scaled by 2^11
    sy, fy = _linear_axis(dh, sh)
    b0 = np.clip(_round_half_even_f32((np.float32(1) - fy) * np.float32(COEF_SCALE)), -32768, 32767)
    b1 = np.clip(_round_half_even_f32(fy * np.float32(COEF_SCALE)), -32768, 32767)
    r0, r1 = np.clip(sy, 0, sh - 1), np.clip(sy + 1, 0, sh - 1)
    s0, s1 = rows[r0], rows[r1]
    out = (((b0[:, None, None] * (s0 >> 4)) >> 16) + ((b1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8).reshape((dh, dw) + img.shape[2:])


def resize_nearest(img, dw, dh):
    """cv2.resize(img, (dw, dh), interpolation=cv2.INTER_NEAREST) (resizeNN: sx = min(floor(x * ifx), sw - 1), ifx = 1 / (dw / sw))."""
    img = np.asarray(img)
    sh, sw = img.shape[:2]
    ifx, ify = 1.0 / (dw / float(sw)), 1.0 / (dh / float(sh))
    sx = np.minimum(np.floor(np.arange(dw, dtype=np.float64) * ifx).astype(np.int64), sw - 1)
    sy = np.minimum(np.floor(np.arange(dh, dtype=np.float64) * ify).astype(np.int64), sh - 1)
    return img[sy][:, sx].copy()


def imrescale(img, scale, interpolation='bilinear'):
    """mmcv.imrescale(img, scale, interpolation=...)."""
    h, w = img.shape[:2]
    nw, nh = rescale_size(w, h, scale)
    return resize_linear_u8(img, nw, nh) if interpolation == 'bilinear' else resize_nearest(img, nw, nh)


# ---- 8-bit BGR <-> HSV (hue range 180) ---------------------------------------------------------------------------------------------
HSV_SHIFT = 12
_I = np.arange(1, 256, dtype=np.float64)
SDIV = np.concatenate([[0], np.rint((255 << HSV_SHIFT) / (1.0 * _I)).astype(np.int64)])
HDIV180 = np.concatenate([[0], np.rint((180 << HSV_SHIFT) / (6.0 * _I)).astype(np.int64)])


def bgr2hsv_u8(img):
    """cv2.cvtColor(img uint8 [...,3] BGR, cv2.COLOR_BGR2HSV) -- RGB2HSV_b, integer arithmetic."""
    a = np.asarray(img).astype(np.int64)
    b, g, r = a[..., 0], a[..., 1], a[..., 2]
    v = np.maximum(np.maximum(b, g), r)
    vmin = np.minimum(np.minimum(b, g), r)
    diff = v - vmin
    vr = np.where(v == r, -1, 0)
    vg = np.where(v == g, -1, 0)
    s = (diff * SDIV[v] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT
    h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + (~vg & (r - g + 4 * diff))))
    h = (h * HDIV180[diff] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT
    h = h + np.where(h < 0, 180, 0)
    return np.stack([np.clip(h, 0, 255), s, v], axis=-1).astype(np.uint8)


_SECTOR = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])


def hsv2bgr_u8(img):
    """cv2.cvtColor(img uint8 [...,3] HSV, cv2.COLOR_HSV2BGR) -- HSV2RGB_b: float32 through HSV2RGB_native, * 255, cvRound, saturate."""
    a = np.asarray(img)
    f32 = np.float32
    h = a[..., 0].astype(f32)
    s = a[..., 1].astype(f32) * f32(1.0 / 255.0)
    v = a[..., 2].astype(f32) * f32(1.0 / 255.0)
    hs = h * f32(6.0 / 180.0)
    hs = np.fmod(hs, f32(6.0)).astype(f32)
    sector = np.floor(hs).astype(np.int64)
    frac = (hs - sector.astype(f32)).astype(f32)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    frac = np.where(bad, f32(0), frac)
    one = f32(1)
    tab = np.stack([v, v * (one - s), v * (one - s * frac), v * (one - s * (one - frac))], axis=-1).astype(f32)
    idx = _SECTOR[sector]                                   # [..., 3]: table slots of b, g, r
    bgr = np.take_along_axis(tab, idx, axis=-1)
    grey = (a[..., 1] == 0)[..., None]
    bgr = np.where(grey, v[..., None], bgr).astype(f32)
    return np.clip(_round_half_even_f32(bgr * f32(255.0)), 0, 255).astype(np.uint8)


def convert(img, alpha=1.0, beta=0.0):
    """PhotoMetricDistortion_clips.convert (transforms.py:2057-2061): float32 multiply, add, clip, truncate."""
    out = np.asarray(img).astype(np.float32) * np.float32(alpha) + np.float32(beta)
    return np.clip(out, 0, 255).astype(np.uint8)


def photometric_frame(img, beta=None, alpha=None, contrast_first=False, saturation=None, hue=None):
    """One frame through PhotoMetricDistortion_clips.__call__ (transforms.py:2112-2139) with the random decisions given:
    brightness, [contrast when mode == 1], saturation, hue, [contrast when mode == 0]."""
    if beta is not None:
        img = convert(img, beta=beta)
    if alpha is not None and contrast_first:
        img = convert(img, alpha=alpha)
    if saturation is not None:
        hsv = bgr2hsv_u8(img)
        hsv[..., 1] = convert(hsv[..., 1], alpha=saturation)
        img = hsv2bgr_u8(hsv)
    if hue is not None:
        hsv = bgr2hsv_u8(img)
        hsv[..., 0] = ((hsv[..., 0].astype(int) + int(hue)) % 180).astype(np.uint8)
        img = hsv2bgr_u8(hsv)
    if alpha is not None and not contrast_first:
        img = convert(img, alpha=alpha)
    return img
