"""Clip data path: which frames form a clip, and the `*_clips` augmentation / formatting of a decoded clip on
the device.

Reference: ``CustomDataset_video2`` (mmseg/datasets/custom.py:1959) picks the frames -- training :2256-2267, test time with its
edge cases :2366-2388 -- and runs the pipeline of local_configs/_base_/datasets/vspw_repeat2.py:8-19 on them, frame by frame in
numpy / cv2 inside the dataloader workers.  Here the random DECISIONS are drawn on the host in the reference's order (same numpy /
``random`` calls, so a seeded run picks the same crop and flip), and crop + flip + BGR->RGB + normalisation + padding + CHW stacking
are ONE kernel over the whole clip (``cffm_clip_format_photo``, csrc/clip_kernels.h), including the brightness / contrast steps of
``PhotoMetricDistortion_clips`` (``PhotoMetricDistortionClips``).  Not covered: image decoding, the random rescale and the test-time
``AlignedResize_clips`` (cv2's fixed-point bilinear resize) and the saturation / hue steps of the photometric distortion (cv2's 8-bit
HSV conversion) -- none of them can be pinned without cv2, which neither box has; they stay upstream of ``ClipFormatter`` (frames
arrive decoded, at the scale to crop from).  ``ClipLister`` restates the video / frame lists of ``CustomDataset_video2``.
"""
import ctypes as C
import random as _pyrandom

import numpy as np
import torch

from . import _lib

DILATION = (-9, -6, -3)          # vspw_repeat2.py:48: reference frames t-9, t-6, t-3


def clip_indices_train(n_frames, dilation=DILATION, flip_video=False, np_random=np.random, py_random=_pyrandom):
    """custom.py:2251-2267 (prepare_train_img2): -> (reversed, [i + d for d in dilation] + [i]) or None when the video is too
    short.  The target index i is uniform over the frames that have -dilation[0] predecessors; with ``flip_video`` the frame list
    is reversed with probability 0.5 first (indices then refer to the reversed list)."""
    rev = bool(flip_video and py_random.random() < 0.5)
    n_valid = max(n_frames + dilation[0], 0)            # len(imglist[-dilation[0]:])
    if n_valid < 1:
        return None
    i = int(np_random.choice(list(range(n_valid)))) - dilation[0]
    return rev, [i + d for d in dilation] + [i]


def clip_indices_test(img_index, n_frames, dilation=DILATION):
    """custom.py:2366-2388 (prepare_test_img2): the in-range frames img_index + d, then img_index; for the default dilation the
    frames 3..8 of a video get hand-picked 4-frame clips.  Fewer than 4 frames (img_index < 3) make the head skip CFFM
    (cffm_head.py:127-129)."""
    step = [img_index + d for d in dilation if 0 <= img_index + d < n_frames] + [img_index]
    if list(dilation) == [-9, -6, -3]:
        special = {3: [0, 1, 2, 3], 4: [0, 2, 3, 4], 5: [0, 2, 4, 5], 6: [0, 2, 4, 6], 7: [0, 3, 5, 7], 8: [0, 3, 6, 8]}
        step = special.get(img_index, step)
    return step


def reduce_zero_label(lab):
    """LoadAnnotations(reduce_zero_label=True), pipelines/loading.py:141-145, on uint8 maps."""
    lab = lab.copy()
    lab[lab == 0] = 255
    lab = lab - 1
    lab[lab == 254] = 255
    return lab


class PhotoMetricDistortionClips:
    """The random decisions of ``PhotoMetricDistortion_clips`` (pipelines/transforms.py:2028-2150), drawn per FRAME in the reference's
    order: ``randint(2)`` brightness (+ ``uniform(-delta, delta)``), ``randint(2)`` mode, [mode 1: ``randint(2)`` contrast (+ ``uniform``)],
    ``randint(2)`` saturation (+ ``uniform``), ``randint(2)`` hue (+ ``randint(-delta, delta)``), [mode 0: contrast] -- so a seeded run stays in
    step with the reference's stream.  The clip kernel applies all of them (``cffm_clip_format_hsv``): brightness and contrast as
    ``convert()`` does (:2057-2061, float32 like numpy), saturation and hue through OpenCV's 8-bit BGR <-> HSV arithmetic
    (mmcv.bgr2hsv / hsv2bgr = ``cv2.cvtColor``), restated from OpenCV's published source because neither mmcv nor cv2 exists in the
    boxes this was built and tested in: those two steps are checked against ``oracle/cv_oracle.py`` only (parity unpinned).
    ``on_hsv``: ``'apply'`` (default); ``'skip'`` leaves the two HSV steps out silently, ``'warn'`` with one RuntimeWarning per instance,
    ``'raise'`` refuses a clip that draws them (for runs that accept nothing unpinned).  ``draw()`` returns every decision in all modes.
    (The default was ``'warn'`` -- steps left out -- until round 5: callers of the older package now get the two HSV steps applied.
    TODO(pin): run ``tests/test_data_cv.py::test_oracle_against_cv2_where_it_exists`` on a box that has cv2; SIMD / IPP builds of
    OpenCV may differ from the published scalar arithmetic in the last bit -- INTEGRATION.md section 1 says the same.)"""

    def __init__(self, brightness_delta=32, contrast_range=(0.5, 1.5), saturation_range=(0.5, 1.5), hue_delta=18, on_hsv='apply',
                 np_random=np.random):
        if on_hsv not in ('apply', 'raise', 'skip', 'warn'):
            raise ValueError("on_hsv must be 'apply', 'warn', 'skip' or 'raise'")
        self._warned = False
        self.brightness_delta, self.contrast_range = brightness_delta, tuple(contrast_range)
        self.saturation_range, self.hue_delta, self.on_hsv, self.rng = tuple(saturation_range), hue_delta, on_hsv, np_random

    def draw(self, n_frames):
        """-> dict(beta [T], alpha [T] (NaN = not taken), contrast_first [T] (the reference's mode == 1), saturation [T], hue [T]
        (None = not taken), apply_hsv)."""
        r = self.rng
        beta, alpha, first, sat, hue = [], [], [], [], []
        for _ in range(n_frames):
            b = float(r.uniform(-self.brightness_delta, self.brightness_delta)) if r.randint(2) else float('nan')
            mode = int(r.randint(2))
            a = float('nan')
            if mode == 1 and r.randint(2):
                a = float(r.uniform(*self.contrast_range))
            s_ = float(r.uniform(*self.saturation_range)) if r.randint(2) else None
            h_ = int(r.randint(-self.hue_delta, self.hue_delta)) if r.randint(2) else None
            if mode == 0 and r.randint(2):
                a = float(r.uniform(*self.contrast_range))
            beta.append(b); alpha.append(a); first.append(mode == 1); sat.append(s_); hue.append(h_)
        drew_hsv = any(v is not None for v in sat + hue)
        if self.on_hsv == 'raise' and drew_hsv:
            raise _lib.CffmError('PhotoMetricDistortionClips: a frame drew the saturation / hue distortion, whose OpenCV arithmetic is '
                                 'restated here without a cv2 to pin it against (on_hsv=\'apply\' applies it, \'skip\' leaves the two steps out)')
        if self.on_hsv == 'warn' and not self._warned and drew_hsv:
            import warnings
            warnings.warn('PhotoMetricDistortionClips: the saturation / hue steps (cv2 8-bit HSV arithmetic) are left out; brightness and '
                          'contrast are applied as the reference does (on_hsv=\'apply\' applies them too)', RuntimeWarning, stacklevel=2)
            self._warned = True
        return dict(beta=beta, alpha=alpha, contrast_first=first, saturation=sat, hue=hue, apply_hsv=self.on_hsv == 'apply')


def _imrescale_size(w, h, scale):
    """mmcv.rescale_size((w, h), scale): a (long edge, short edge) bound -> (new_w, new_h), ``int(x * factor + 0.5)``."""
    f = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return int(w * float(f) + 0.5), int(h * float(f) + 0.5)


def resize_clip(frames, labels, size):
    """``cv2.resize`` of a clip on the device: frames [T,H,W,3] uint8 -> [T,h,w,3] (INTER_LINEAR), labels [T,H,W] uint8 -> [T,h,w]
    (INTER_NEAREST); ``size = (w, h)`` as mmcv / cv2 take it; either input may be None."""
    lib = _lib.get()
    ref = frames if frames is not None else labels
    for t, what in ((frames, 'frames'), (labels, 'labels')):
        if t is None:
            continue
        if t.dtype != torch.uint8:
            raise _lib.CffmError('resize_clip: %s must be uint8, got %s' % (what, t.dtype))
        if _lib._override is None and not t.is_cuda:
            raise _lib.CffmError('resize_clip: %s are on %s; the kernel runs only on the GPU (no CPU fallback)' % (what, t.device))
    if frames is not None and (frames.dim() != 4 or frames.shape[3] != 3) or (labels is not None and labels.dim() != 3) or \
            (frames is not None and labels is not None and labels.shape != frames.shape[:3]):
        raise _lib.CffmError('resize_clip: frames [T,H,W,3] and labels [T,H,W] expected')
    t, h, w = ref.shape[:3]
    nw, nh = int(size[0]), int(size[1])
    frames = frames.contiguous() if frames is not None else None
    labels = labels.contiguous() if labels is not None else None
    out_f = torch.empty(t, nh, nw, 3, dtype=torch.uint8, device=ref.device) if frames is not None else None
    out_l = torch.empty(t, nh, nw, dtype=torch.uint8, device=ref.device) if labels is not None else None
    stream = C.c_void_p(torch.cuda.current_stream(ref.device).cuda_stream) if ref.is_cuda else C.c_void_p(0)
    ptr = lambda x: C.c_void_p(x.data_ptr()) if x is not None else C.c_void_p(0)
    _lib.check(lib.cffm_clip_resize(ptr(frames), ptr(labels), t, h, w, ptr(out_f), ptr(out_l), nh, nw, stream), lib)
    return out_f, out_l


class ResizeClips:
    """``Resize(img_scale=(853, 480), ratio_range=(0.5, 2.0), process_clips=True)`` of the training pipeline
    (local_configs/_base_/datasets/vspw_repeat2.py:10; mmseg/datasets/pipelines/transforms.py:475-760): one ratio per clip
    (``np.random.random_sample() * (max - min) + min``, ``scale = (int(853 r), int(480 r))``, :625-637), every frame through
    ``mmcv.imrescale(keep_ratio)`` -- bilinear for the frames, nearest for the label maps.  ``draw()`` makes the random decision in the
    reference's order, ``apply()`` runs the kernel (``cffm_clip_resize``: OpenCV's 8-bit resize arithmetic restated, parity unpinned)."""

    def __init__(self, img_scale=(853, 480), ratio_range=(0.5, 2.0), keep_ratio=True, np_random=np.random):
        if isinstance(img_scale, list):
            if len(img_scale) != 1:
                raise NotImplementedError('ResizeClips: one img_scale (modes 3 / 4 of the reference, several scales, are not used by any CFFM config)')
            img_scale = img_scale[0]
        self.img_scale = tuple(img_scale) if img_scale is not None else None
        self.ratio_range = tuple(ratio_range) if ratio_range is not None else None
        self.keep_ratio, self.rng = keep_ratio, np_random

    def draw(self, shape):
        """-> the ``scale`` tuple of the reference's ``results['scale']`` for a clip of frame shape (H, W)."""
        if self.ratio_range is None:
            if self.img_scale is None:
                raise _lib.CffmError('ResizeClips: neither img_scale nor ratio_range')
            return self.img_scale
        base = self.img_scale if self.img_scale is not None else (shape[1], shape[0])
        lo, hi = self.ratio_range
        ratio = self.rng.random_sample() * (hi - lo) + lo
        return int(base[0] * ratio), int(base[1] * ratio)

    def size_for(self, shape, scale):
        h, w = shape
        return _imrescale_size(w, h, scale) if self.keep_ratio else (int(scale[0]), int(scale[1]))

    def apply(self, frames, labels, scale):
        ref = frames if frames is not None else labels
        return resize_clip(frames, labels, self.size_for(tuple(ref.shape[1:3]), scale))

    def __call__(self, frames, labels=None):
        ref = frames if frames is not None else labels
        scale = self.draw(tuple(ref.shape[1:3]))
        return self.apply(frames, labels, scale) + (scale,)


class AlignedResizeClips(ResizeClips):
    """``AlignedResize_clips(keep_ratio=True, size_divisor=32)`` of the test pipeline (vspw_repeat2.py:27; transforms.py:236-470): the
    frames through ``mmcv.imrescale`` to ``img_scale`` (``MultiScaleFlipAug`` supplies (853, 480)), then ``_align`` (:394-401): a second
    ``mmcv.imresize`` to the next multiples of ``size_divisor`` -- two resizes, as the reference does them (the intermediate image is
    rounded to uint8 in between)."""

    def __init__(self, img_scale=(853, 480), ratio_range=None, keep_ratio=True, size_divisor=32, np_random=np.random):
        super().__init__(img_scale, ratio_range, keep_ratio, np_random)
        self.size_divisor = size_divisor

    def apply(self, frames, labels, scale):
        ref = frames if frames is not None else labels
        d = self.size_divisor
        if not self.keep_ratio:
            nw, nh = int(scale[0]), int(scale[1])
            if nh % d or nw % d:
                raise _lib.CffmError('AlignedResizeClips: img size not align. h:%d w:%d' % (nh, nw))     # the reference's assert (:430)
            return resize_clip(frames, labels, (nw, nh))
        f, l = resize_clip(frames, labels, self.size_for(tuple(ref.shape[1:3]), scale))
        ref = f if f is not None else l
        h, w = ref.shape[1:3]
        ah, aw = -(-h // d) * d, -(-w // d) * d
        return resize_clip(f, l, (aw, ah))       # (same size: the kernel copies, as cv2.resize does)


class ClipFormatter:
    """RandomCrop_clips + RandomFlip_clips + Normalize_clips + Pad_clips + DefaultFormatBundle_clips for one decoded clip.

    ``draw(last_label)`` makes the random decisions exactly as the reference's classes do (transforms.py:1540-1575: one
    ``np.random.randint`` pair per candidate box, up to 10 retries while a single category covers >= cat_max_ratio of the target
    frame's crop; :886-888: one ``np.random.rand`` for the flip) -- labels are judged AFTER reduce_zero_label, as in the pipeline;
    ``apply(frames, labels, params)`` runs the kernel.  ``__call__`` does both."""

    def __init__(self, crop_size=(480, 480), cat_max_ratio=0.75, flip_prob=0.5, mean=(123.675, 116.28, 103.53),
                 std=(58.395, 57.12, 57.375), to_rgb=True, pad_val=0, seg_pad_val=255, ignore_index=255, reduce_zero_label=True,
                 np_random=np.random, photo=None):
        self.crop_size, self.cat_max_ratio, self.flip_prob = tuple(crop_size), cat_max_ratio, flip_prob
        self.mean, self.std, self.to_rgb = tuple(mean), tuple(std), to_rgb
        self.pad_val, self.seg_pad_val, self.ignore_index, self.rzl = pad_val, seg_pad_val, ignore_index, reduce_zero_label
        self.rng = np_random
        self.photo = photo            # a PhotoMetricDistortionClips (drawn after the flip, as the pipeline orders them) or None

    def _bbox(self, h, w):
        margin_h, margin_w = max(h - self.crop_size[0], 0), max(w - self.crop_size[1], 0)
        oy = int(self.rng.randint(0, margin_h + 1))
        ox = int(self.rng.randint(0, margin_w + 1))
        return oy, oy + self.crop_size[0], ox, ox + self.crop_size[1]

    def draw(self, last_label, shape=None, n_frames=4):
        """-> dict(y1, x1, ch, cw, flip[, photo]).  last_label: the TARGET frame's raw uint8 label map [H,W] (or None with `shape`)."""
        h, w = last_label.shape[:2] if last_label is not None else shape
        box = self._bbox(h, w)
        if self.cat_max_ratio < 1. and last_label is not None:
            lab = reduce_zero_label(last_label) if self.rzl else last_label
            for _ in range(10):
                seg = lab[box[0]:box[1], box[2]:box[3]]
                labels, cnt = np.unique(seg, return_counts=True)
                cnt = cnt[labels != self.ignore_index]
                if len(cnt) > 1 and np.max(cnt) / np.sum(cnt) < self.cat_max_ratio:
                    break
                box = self._bbox(h, w)
        flip = bool(self.rng.rand() < self.flip_prob) if self.flip_prob is not None else False
        y1, x1 = box[0], box[2]
        out = dict(y1=y1, x1=x1, ch=min(box[1], h) - y1, cw=min(box[3], w) - x1, flip=flip)
        if self.photo is not None:
            out['photo'] = self.photo.draw(n_frames)
        return out

    def apply(self, frames, labels, params):
        """frames [T,H,W,3] uint8 (BGR), labels [T,H,W] uint8 or None, on the device -> (img [T,3,Ho,Wo] float32,
        gt_semantic_seg [T,1,Ho,Wo] int64 or None)."""
        lib = _lib.get()
        for t, what in ((frames, 'frames'), (labels, 'labels')):
            if t is None:
                continue
            if t.dtype != torch.uint8:
                raise _lib.CffmError('ClipFormatter: %s must be uint8, got %s' % (what, t.dtype))
            if _lib._override is None and not t.is_cuda:
                raise _lib.CffmError('ClipFormatter: %s are on %s; the kernel runs only on the GPU (no CPU fallback)' % (what, t.device))
        if frames.dim() != 4 or frames.shape[3] != 3 or (labels is not None and labels.shape != frames.shape[:3]):
            raise _lib.CffmError('ClipFormatter: frames [T,H,W,3] and labels [T,H,W] expected')
        frames = frames.contiguous()
        labels = labels.contiguous() if labels is not None else None
        t, h, w, _ = frames.shape
        ho, wo = max(self.crop_size[0], params['ch']), max(self.crop_size[1], params['cw'])
        img = torch.empty(t, 3, ho, wo, dtype=torch.float32, device=frames.device)
        lab = torch.empty(t, 1, ho, wo, dtype=torch.int64, device=frames.device) if labels is not None else None
        stream = C.c_void_p(torch.cuda.current_stream(frames.device).cuda_stream) if frames.is_cuda else C.c_void_p(0)
        ptr = lambda x: C.c_void_p(x.data_ptr()) if x is not None else C.c_void_p(0)
        ph = params.get('photo')
        if ph is not None and len(ph['beta']) != t:
            raise _lib.CffmError('ClipFormatter: photometric parameters for %d frames, clip has %d' % (len(ph['beta']), t))
        beta = (C.c_float * t)(*ph['beta']) if ph is not None else None
        alpha = (C.c_float * t)(*ph['alpha']) if ph is not None else None
        first = sat = hue = None
        if ph is not None and ph.get('apply_hsv'):
            nan = float('nan')
            first = (C.c_int * t)(*[int(bool(v)) for v in ph['contrast_first']])
            sat = (C.c_float * t)(*[nan if v is None else float(v) for v in ph['saturation']])
            hue = (C.c_float * t)(*[nan if v is None else float(int(v)) for v in ph['hue']])
        _lib.check(lib.cffm_clip_format_hsv(ptr(frames), ptr(labels), ptr(img), ptr(lab), t, h, w, params['y1'], params['x1'], params['ch'],
                                            params['cw'], int(params['flip']), ho, wo, (C.c_float * 3)(*self.mean),
                                            (C.c_float * 3)(*self.std), int(self.to_rgb), float(self.pad_val), int(self.seg_pad_val),
                                            int(self.rzl), beta, alpha, first, sat, hue, stream), lib)
        return img, lab

    def __call__(self, frames, labels, last_label_host=None):
        """last_label_host: numpy copy of the target frame's raw labels for the crop decision (default: labels[-1] copied back)."""
        if last_label_host is None and labels is not None:
            last_label_host = labels[-1].cpu().numpy()
        params = self.draw(last_label_host, shape=tuple(frames.shape[1:3]), n_frames=int(frames.shape[0]))
        return self.apply(frames, labels, params) + (params,)


def load_clip(img_paths, mask_paths=None, device=None):
    """``LoadImageFromFile`` + ``LoadAnnotations`` (mmseg/datasets/pipelines/loading.py:10-88, :91-155) for the frames of one clip:
    -> (frames [T,H,W,3] uint8 BGR, labels [T,H,W] uint8 or None) on ``device``.

    The reference decodes label maps with Pillow (``imdecode_backend='pillow'``, flag ``'unchanged'``, ``squeeze().astype(uint8)``: :133-135)
    -- done the same way here, bit for bit (PNG is lossless) -- and images with ``cv2.imdecode`` (colour, BGR).  cv2 is not available
    here, so images are decoded with Pillow and flipped to BGR: identical for lossless formats; for JPEG both sit on libjpeg(-turbo)'s
    default decoder, whose output is build-dependent in the last bit for the reference too.  (cv2 also honours an EXIF orientation tag;
    VSPW frames carry none, and a file that does is refused rather than silently left unrotated.)"""
    from PIL import Image
    frames, labels = [], []
    for p_ in img_paths:
        with Image.open(p_) as im:
            if im.getexif().get(0x0112, 1) not in (0, 1):
                raise _lib.CffmError('load_clip: %s carries an EXIF orientation, which cv2.imdecode would apply' % p_)
            frames.append(np.asarray(im.convert('RGB'))[:, :, ::-1])
    for p_ in (mask_paths or []):
        with Image.open(p_) as im:
            labels.append(np.asarray(im).squeeze().astype(np.uint8))
    shapes = {f.shape[:2] for f in frames} | {l.shape[:2] for l in labels}
    if len(shapes) != 1 or (labels and len(labels) != len(frames)) or any(l.ndim != 2 for l in labels):
        raise _lib.CffmError('load_clip: the frames / label maps of a clip must share one size (got %s)' % sorted(shapes))
    f = torch.from_numpy(np.ascontiguousarray(np.stack(frames)))
    l = torch.from_numpy(np.ascontiguousarray(np.stack(labels))) if labels else None
    if device is not None:
        f, l = f.to(device), (l.to(device) if l is not None else None)
    return f, l


class ClipLister:
    """The video / frame lists of ``CustomDataset_video2`` (mmseg/datasets/custom.py:1959-2100) and the clips it serves.

    ``<data_root>/<split>.txt`` names the videos (one per line); split ``'train_val_generate_prototype'`` concatenates train, val
    and test (:2063-2072).  A video's frames are ``sorted(os.listdir(<data_root>/data/<video>/origin))`` (:2079-2084); label maps live
    in ``.../mask/`` under the same name with ``seg_map_suffix``.  Length and indexing follow the reference: a TRAINING item is a
    video (``len(videolists)``, one random clip per visit: ``clip_indices_train``, custom.py:2251-2267), every other split serves one
    clip per frame (``len(img_all)``, ``clip_indices_test``, :2366-2388)."""

    def __init__(self, data_root, split, dilation=DILATION, img_suffix='.jpg', seg_map_suffix='.png', flip_video=True):
        import os
        self.data_root, self.split, self.dilation = data_root, split, list(dilation)
        self.img_suffix, self.seg_map_suffix, self.flip_video = img_suffix, seg_map_suffix, flip_video
        names = ['train', 'val', 'test'] if split == 'train_val_generate_prototype' else [split]
        self.videolists = []
        for n in names:
            with open(os.path.join(data_root, n + '.txt')) as f:
                self.videolists += [line.rstrip('\n') for line in f.readlines()]     # (the reference drops the last character of every line)
        self.imgdic, self.img_all = {}, []
        for video in self.videolists:
            imglist = sorted(os.listdir(os.path.join(data_root, 'data', video, 'origin')))
            self.imgdic[video] = imglist
            self.img_all += [[video, img] for img in imglist]

    @property
    def per_video(self):
        return self.split in ('train', 'train_val_generate_prototype')

    def __len__(self):
        return len(self.videolists) if self.per_video else len(self.img_all)

    def _paths(self, video, names):
        import os
        img_dir, ann_dir = os.path.join(self.data_root, 'data', video, 'origin/'), os.path.join(self.data_root, 'data', video, 'mask/')
        return ([os.path.join(img_dir, n) for n in names], [os.path.join(ann_dir, n.replace(self.img_suffix, self.seg_map_suffix)) for n in names])

    def train_item(self, idx, np_random=np.random, py_random=_pyrandom):
        """-> dict(video, reversed, frames (names, oldest reference frame first, target last), img_paths, mask_paths) or None (video too
        short), with the reference's two random draws (custom.py:2256-2263)."""
        video = self.videolists[idx]
        imglist = self.imgdic[video]
        got = clip_indices_train(len(imglist), self.dilation, self.flip_video, np_random, py_random)
        if got is None:
            return None
        rev, idxs = got
        if rev:
            imglist = imglist[::-1]
        names = [imglist[i] for i in idxs]
        imgs, masks = self._paths(video, names)
        return dict(video=video, reversed=rev, frames=names, img_paths=imgs, mask_paths=masks)

    def test_item(self, idx):
        """-> dict(video, frames, img_paths, mask_paths): the clip of frame idx of ``img_all`` (1 to 4 frames, target last)."""
        video, img_name = self.img_all[idx]
        imglist = self.imgdic[video]
        names = [imglist[i] for i in clip_indices_test(imglist.index(img_name), len(imglist), self.dilation)]
        imgs, masks = self._paths(video, names)
        return dict(video=video, frames=names, img_paths=imgs, mask_paths=masks)
