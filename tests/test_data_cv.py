"""The OpenCV-backed steps of the clip data path (SURVEY.md 8f.3 remainder): cv2.resize behind Resize / AlignedResize_clips and the 8-bit
BGR <-> HSV conversion behind PhotoMetricDistortion_clips' saturation / hue branches.

cv2 / mmcv exist in neither box, so there are no reference-produced goldens for these steps (parity unpinned: oracle/cv_oracle.py,
DESIGN.md 3d).  What is checked: the HIP kernels (emulated here, on the GPU under -m gpu) equal the numpy restatement of OpenCV's
published arithmetic BIT FOR BIT; the restatement itself agrees with independent formulations (float64 bilinear interpolation within
1 LSB, colorsys within the 8-bit quantisation) and satisfies the identities the algorithms imply; the host classes draw their random
numbers in the reference's order and apply mmcv's size rules."""
import colorsys

import numpy as np
import pytest
import torch

from oracle import cv_oracle as CV
from tests import emu
from vss_cffm_amd import _lib, data as D


def _clip(seed, t, h, w):
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, (t, max(h // 8, 1) + 1, max(w // 8, 1) + 1, 3)).astype(np.float64)
    yy, xx = np.linspace(0, base.shape[1] - 1.001, h), np.linspace(0, base.shape[2] - 1.001, w)
    y0, x0 = yy.astype(int), xx.astype(int)
    fy, fx = (yy - y0)[None, :, None, None], (xx - x0)[None, None, :, None]
    sm = (base[:, y0][:, :, x0] * (1 - fy) * (1 - fx) + base[:, y0 + 1][:, :, x0] * fy * (1 - fx) +
          base[:, y0][:, :, x0 + 1] * (1 - fy) * fx + base[:, y0 + 1][:, :, x0 + 1] * fy * fx)
    frames = np.clip(sm + rng.randint(-20, 21, (t, h, w, 3)), 0, 255).astype(np.uint8)
    labels = rng.randint(0, 125, (t, h, w)).astype(np.uint8)
    return frames, labels


# ---- the restatement against independent formulations ---------------------------------------------------------------------------------
def test_oracle_linear_resize_is_bilinear_interpolation_within_one_lsb():
    frames, _ = _clip(0, 1, 37, 53)
    img = frames[0]
    for dw, dh in ((80, 61), (53, 37), (21, 17), (106, 74), (40, 37)):
        got = CV.resize_linear_u8(img, dw, dh).astype(np.float64)
        fx = np.clip((np.arange(dw) + 0.5) * (53 / dw) - 0.5, 0, 52)
        fy = np.clip((np.arange(dh) + 0.5) * (37 / dh) - 0.5, 0, 36)
        x0, y0 = np.minimum(fx.astype(int), 51), np.minimum(fy.astype(int), 35)
        wx, wy = (fx - x0)[None, :, None], (fy - y0)[:, None, None]
        a = img.astype(np.float64)
        want = (a[y0][:, x0] * (1 - wy) * (1 - wx) + a[y0 + 1][:, x0] * wy * (1 - wx) + a[y0][:, x0 + 1] * (1 - wy) * wx +
                a[y0 + 1][:, x0 + 1] * wy * wx)
        assert np.abs(got - want).max() <= 1.0 + 1e-9, (dw, dh)
    assert np.array_equal(CV.resize_linear_u8(img, 53, 37), img)                      # same size: a copy
    even = frames[0][:36, :52]
    box = (even[0::2, 0::2].astype(int) + even[0::2, 1::2] + even[1::2, 0::2] + even[1::2, 1::2] + 2) >> 2
    assert np.array_equal(CV.resize_linear_u8(even, 26, 18), box.astype(np.uint8))    # 2x2 -> 1: INTER_AREA's box mean
    flat = np.full((9, 11, 3), 137, np.uint8)
    assert np.all(CV.resize_linear_u8(flat, 30, 23) == 137) and np.all(CV.resize_linear_u8(flat, 4, 5) == 137)


def test_oracle_nearest_resize_and_size_rules():
    lab = np.arange(7 * 9, dtype=np.uint8).reshape(7, 9)
    out = CV.resize_nearest(lab, 18, 14)
    assert np.array_equal(out, lab[np.arange(14) // 2][:, np.arange(18) // 2])
    out = CV.resize_nearest(lab, 4, 3)
    assert set(out.reshape(-1)) <= set(lab.reshape(-1)) and out[0, 0] == lab[0, 0]
    assert CV.rescale_size(854, 480, (853, 480)) == (853, 479) and CV.rescale_size(854, 480, (1706, 960)) == (1706, 959)
    assert D._imrescale_size(854, 480, (426, 240)) == CV.rescale_size(854, 480, (426, 240))


def test_oracle_hsv_against_colorsys_and_identities():
    rng = np.random.RandomState(5)
    px = rng.randint(0, 256, (4000, 3)).astype(np.uint8)
    hsv = CV.bgr2hsv_u8(px)
    for (b, g, r), (h, s, v) in zip(px[:600].tolist(), hsv[:600].tolist()):
        ch, cs, cv_ = colorsys.rgb_to_hsv(r / 255.0, g / 255.0, b / 255.0)
        assert v == max(b, g, r) and abs(s - cs * 255) <= 0.55     # (the 12-bit division table adds up to ~0.02)
        if cs * cv_ * 255 >= 24:                                   # (the hue of near-grey pixels is ill-conditioned)
            dh = abs(h - ch * 180)
            assert min(dh, 180 - dh) <= 1.5, ((b, g, r), h, ch * 180)
    assert hsv[:, 0].max() < 180
    back = CV.hsv2bgr_u8(hsv).astype(int)
    assert np.abs(back - px.astype(int)).max() <= 5                  # 8-bit round trip: a few LSB, as with cv2
    grey = np.repeat(np.arange(256, dtype=np.uint8)[:, None], 3, 1)
    gh = CV.bgr2hsv_u8(grey)
    assert np.all(gh[:, :2] == 0) and np.array_equal(gh[:, 2], np.arange(256)) and np.array_equal(CV.hsv2bgr_u8(gh), grey)
    # pure colours land on the sector boundaries
    prim = np.array([[0, 0, 255], [0, 255, 0], [255, 0, 0], [0, 255, 255], [255, 255, 0], [255, 0, 255]], np.uint8)    # BGR: red green blue yellow cyan magenta
    assert CV.bgr2hsv_u8(prim)[:, 0].tolist() == [0, 60, 120, 30, 90, 150]
    assert np.array_equal(CV.hsv2bgr_u8(CV.bgr2hsv_u8(prim)), prim)
    tabs = np.array([CV.SDIV[1:], CV.HDIV180[1:]])
    i = np.arange(1, 256)
    assert np.array_equal(tabs[0], (2 * (255 << 12) + i) // (2 * i)) and np.array_equal(tabs[1], (2 * ((180 << 12) // 6) + i) // (2 * i))


def test_oracle_against_cv2_where_it_exists():
    """The pin this repository cannot produce itself: wherever OpenCV is importable (it is in neither box of this build: skipped there), the
    restatement must equal cv2 BIT FOR BIT -- resize (both interpolations, up / down / 2x shrink / identity, 1 and 3 channels) and the two colour
    conversions over all the value ranges.  A maintainer's `pip install opencv-python && pytest tests/test_data_cv.py -k cv2` settles DESIGN 3d's
    'parity unpinned'."""
    # oracle/ref_import.py leaves a stub module named cv2 in sys.modules (the reference's imports need the name): look for the real
    # package on the path, not for the name
    import importlib, importlib.machinery, sys
    if importlib.machinery.PathFinder.find_spec('cv2') is None:
        pytest.skip('OpenCV is not installed')
    stub = sys.modules.pop('cv2', None) if not hasattr(sys.modules.get('cv2'), '__file__') else None
    try:
        cv2 = importlib.import_module('cv2')
    finally:
        if stub is not None:
            sys.modules['cv2'] = stub
    frames, labels = _clip(90, 2, 97, 131)
    for dw, dh in ((200, 150), (131, 97), (64, 48), (262, 194), (33, 97), (131, 20)):
        for img in (frames[0], frames[1][:, :, 0].copy()):
            assert np.array_equal(CV.resize_linear_u8(img, dw, dh), cv2.resize(img, (dw, dh), interpolation=cv2.INTER_LINEAR)), (dw, dh, img.shape)
        assert np.array_equal(CV.resize_nearest(labels[0], dw, dh), cv2.resize(labels[0], (dw, dh), interpolation=cv2.INTER_NEAREST)), (dw, dh)
    even = frames[0][:96, :130]
    assert np.array_equal(CV.resize_linear_u8(even, 65, 48), cv2.resize(even, (65, 48), interpolation=cv2.INTER_LINEAR))
    rng = np.random.RandomState(91)
    px = rng.randint(0, 256, (64, 4096, 3)).astype(np.uint8)
    assert np.array_equal(CV.bgr2hsv_u8(px), cv2.cvtColor(px, cv2.COLOR_BGR2HSV))
    hsv = px.copy(); hsv[..., 0] %= 180
    assert np.array_equal(CV.hsv2bgr_u8(hsv), cv2.cvtColor(hsv, cv2.COLOR_HSV2BGR))
    grid = np.stack(np.meshgrid(np.arange(0, 180, 3), np.arange(0, 256, 5), np.arange(0, 256, 5), indexing='ij'), -1).reshape(1, -1, 3).astype(np.uint8)
    assert np.array_equal(CV.hsv2bgr_u8(grid), cv2.cvtColor(grid, cv2.COLOR_HSV2BGR))


# ---- the kernels against the restatement: bit for bit -----------------------------------------------------------------------------------
RESIZE_CASES = ((1, 4, 37, 53, 80, 61), (2, 2, 48, 64, 24, 32), (3, 4, 30, 41, 30, 41), (4, 1, 9, 7, 40, 33), (5, 3, 50, 70, 17, 23),
                (6, 2, 1, 5, 3, 11), (7, 1, 64, 96, 213, 120), (8, 2, 33, 47, 64, 64))


def run_resize_cases(device):
    for seed, t, h, w, dw, dh in RESIZE_CASES:
        frames, labels = _clip(seed, t, h, w)
        f, l = D.resize_clip(torch.from_numpy(frames).to(device), torch.from_numpy(labels).to(device), (dw, dh))
        want_f = np.stack([CV.resize_linear_u8(x, dw, dh) for x in frames])
        want_l = np.stack([CV.resize_nearest(x, dw, dh) for x in labels])
        assert f.shape == (t, dh, dw, 3) and f.dtype == torch.uint8
        assert np.array_equal(f.cpu().numpy(), want_f), (seed, np.abs(f.cpu().numpy().astype(int) - want_f).max())
        assert np.array_equal(l.cpu().numpy(), want_l), seed
    frames, labels = _clip(11, 2, 40, 30)
    f, l = D.resize_clip(torch.from_numpy(frames).to(device), None, (60, 80))                       # frames only / labels only
    assert l is None and np.array_equal(f.cpu().numpy()[1], CV.resize_linear_u8(frames[1], 60, 80))
    f, l = D.resize_clip(None, torch.from_numpy(labels).to(device), (15, 20))
    assert f is None and np.array_equal(l.cpu().numpy()[0], CV.resize_nearest(labels[0], 15, 20))
    with pytest.raises(_lib.CffmError):
        D.resize_clip(torch.from_numpy(frames).to(device).float(), None, (10, 10))
    # the training transform: one ratio per clip, drawn as the reference draws it (random_sample_ratio, transforms.py:625-637)
    np.random.seed(21)
    ratio = np.random.random_sample() * 1.5 + 0.5
    np.random.seed(21)
    frames, labels = _clip(12, 4, 48, 85)
    rs = D.ResizeClips(img_scale=(85, 48), ratio_range=(0.5, 2.0))
    f, l, scale = rs(torch.from_numpy(frames).to(device), torch.from_numpy(labels).to(device))
    assert scale == (int(85 * ratio), int(48 * ratio))
    nw, nh = CV.rescale_size(85, 48, scale)
    assert f.shape == (4, nh, nw, 3) and np.array_equal(f.cpu().numpy()[3], CV.imrescale(frames[3], scale))
    assert np.array_equal(l.cpu().numpy()[0], CV.imrescale(labels[0], scale, 'nearest'))
    # the test-time transform: rescale, then align both sides to multiples of 32 with a second resize (_align, :394-401)
    al = D.AlignedResizeClips(img_scale=(100, 60), size_divisor=32)
    f, _, scale = al(torch.from_numpy(frames).to(device))
    step1 = CV.imrescale(frames[2], (100, 60))
    ah, aw = -(-step1.shape[0] // 32) * 32, -(-step1.shape[1] // 32) * 32
    assert scale == (100, 60) and f.shape == (4, ah, aw, 3) and np.array_equal(f.cpu().numpy()[2], CV.resize_linear_u8(step1, aw, ah))
    with pytest.raises(_lib.CffmError):
        D.AlignedResizeClips(img_scale=(100, 60), keep_ratio=False)(torch.from_numpy(frames).to(device))       # 60 is no multiple of 32


PHOTO_HSV_CASES = ((31, 4), (32, 4), (33, 3), (34, 4), (35, 1), (36, 4))


def run_photo_hsv_cases(device):
    took = dict(s=0, h=0, both=0, first=0)
    for seed, t in PHOTO_HSV_CASES:
        frames, labels = _clip(seed, t, 40, 56)
        np.random.seed(seed)
        fmt = D.ClipFormatter(crop_size=(32, 48), cat_max_ratio=0.75, flip_prob=0.5, photo=D.PhotoMetricDistortionClips())
        img, gt, params = fmt(torch.from_numpy(frames).to(device), torch.from_numpy(labels).to(device), last_label_host=labels[-1])
        ph = params['photo']
        y1, x1, ch, cw = params['y1'], params['x1'], params['ch'], params['cw']
        mean, stdinv = np.float32(fmt.mean), (1 / np.float64(fmt.std)).astype(np.float32)
        for i in range(t):
            crop = frames[i, y1:y1 + ch, x1:x1 + cw]
            if params['flip']:
                crop = crop[:, ::-1]
            beta = None if ph['beta'][i] != ph['beta'][i] else ph['beta'][i]
            alpha = None if ph['alpha'][i] != ph['alpha'][i] else ph['alpha'][i]
            out = CV.photometric_frame(crop, beta, alpha, ph['contrast_first'][i], ph['saturation'][i], ph['hue'][i])
            want = (out[:, :, ::-1].astype(np.float32) - mean) * stdinv                  # BGR -> RGB, mmcv.imnormalize
            got = img[i, :, :ch, :cw].permute(1, 2, 0).cpu().numpy()
            assert np.array_equal(got, want.astype(np.float32)), (seed, i, np.abs(got - want).max())
            took['s'] += ph['saturation'][i] is not None; took['h'] += ph['hue'][i] is not None
            took['both'] += ph['saturation'][i] is not None and ph['hue'][i] is not None
            took['first'] += bool(ph['contrast_first'][i]) and alpha is not None
    assert took['s'] >= 3 and took['h'] >= 3 and took['both'] >= 1 and took['first'] >= 1, took        # (every branch was exercised)
    # explicit parameters: a hue shift that wraps below zero, a saturation that clips, and the 'skip' policy
    frames, _ = _clip(40, 2, 24, 24)
    fmt = D.ClipFormatter(crop_size=(24, 24), cat_max_ratio=1.0, flip_prob=None, to_rgb=False, mean=(0, 0, 0), std=(1, 1, 1))
    nan = float('nan')
    ph = dict(beta=[nan, 7.5], alpha=[1.25, nan], contrast_first=[False, True], saturation=[1.5, None], hue=[-18, 17], apply_hsv=True)
    img, _ = fmt.apply(torch.from_numpy(frames).to(device), None, dict(y1=0, x1=0, ch=24, cw=24, flip=False, photo=ph))
    for i in range(2):
        beta = None if ph['beta'][i] != ph['beta'][i] else ph['beta'][i]
        alpha = None if ph['alpha'][i] != ph['alpha'][i] else ph['alpha'][i]
        want = CV.photometric_frame(frames[i], beta, alpha, ph['contrast_first'][i], ph['saturation'][i], ph['hue'][i])
        assert np.array_equal(img[i].permute(1, 2, 0).cpu().numpy(), want.astype(np.float32)), i
    ph['apply_hsv'] = False
    img, _ = fmt.apply(torch.from_numpy(frames).to(device), None, dict(y1=0, x1=0, ch=24, cw=24, flip=False, photo=ph))
    assert np.array_equal(img[0].permute(1, 2, 0).cpu().numpy(), CV.convert(frames[0], alpha=1.25).astype(np.float32))
    # the policies that do not apply the two steps
    np.random.seed(3)
    with pytest.raises(_lib.CffmError):
        for _ in range(8):
            D.PhotoMetricDistortionClips(on_hsv='raise').draw(4)
    np.random.seed(3)
    warn = D.PhotoMetricDistortionClips(on_hsv='warn')
    with pytest.warns(RuntimeWarning, match='saturation / hue'):
        for _ in range(8):
            assert not warn.draw(4)['apply_hsv']
    assert D.PhotoMetricDistortionClips().draw(2)['apply_hsv']


def test_load_clip_decodes_like_the_reference_loaders(tmp_path):
    """LoadAnnotations decodes with Pillow ('unchanged', squeeze, uint8): the same call, bit for bit; lossless frames come back exactly, in
    BGR order; clips whose frames differ in size or that carry an EXIF rotation are refused."""
    from PIL import Image
    frames, labels = _clip(60, 3, 20, 28)
    ips, mps = [], []
    for i in range(3):
        ips.append(str(tmp_path / ('%08d.png' % i))); mps.append(str(tmp_path / ('m%08d.png' % i)))
        Image.fromarray(frames[i][:, :, ::-1]).save(ips[-1])          # the file holds RGB; the loader hands back BGR as cv2 would
        Image.fromarray(labels[i], mode='L').save(mps[-1])
    f, l = D.load_clip(ips, mps)
    assert f.dtype == torch.uint8 and np.array_equal(f.numpy(), frames) and np.array_equal(l.numpy(), labels)
    f, l = D.load_clip(ips[:2])
    assert l is None and f.shape == (2, 20, 28, 3)
    pal = Image.fromarray(labels[0], mode='P'); pal.putpalette([i % 256 for i in range(768)]); pal.save(mps[0])     # palette PNGs (VSPW masks): indices, not colours
    assert np.array_equal(D.load_clip(ips[:1], mps[:1])[1].numpy()[0], labels[0])
    jp = str(tmp_path / 'a.jpg')
    Image.fromarray(frames[0][:, :, ::-1]).save(jp, quality=90)
    fj, _ = D.load_clip([jp])
    assert fj.shape == (1, 20, 28, 3) and np.abs(fj.numpy()[0].astype(int) - frames[0]).mean() < 12           # lossy, BGR order kept
    Image.fromarray(frames[0][:10, :, ::-1]).save(ips[1])
    with pytest.raises(_lib.CffmError):
        D.load_clip(ips)
    ex = Image.fromarray(frames[0][:, :, ::-1]); exif = ex.getexif(); exif[0x0112] = 6; ex.save(jp, exif=exif)
    with pytest.raises(_lib.CffmError):
        D.load_clip([jp])


def run_train_pipeline(device, seed, t, h, w, img_scale, crop):
    """The training pipeline of vspw_repeat2.py:8-19 from decoded frames on: Resize(ratio_range) -> RandomCrop_clips -> RandomFlip_clips ->
    PhotoMetricDistortion_clips -> Normalize_clips -> Pad_clips -> DefaultFormatBundle_clips, the random numbers drawn in that order from one
    stream, against the same chain composed from the oracle's pieces."""
    frames, labels = _clip(seed, t, h, w)
    f_t, l_t = torch.from_numpy(frames).to(device), torch.from_numpy(labels).to(device)
    np.random.seed(seed)
    rs = D.ResizeClips(img_scale=img_scale, ratio_range=(0.5, 2.0))
    fmt = D.ClipFormatter(crop_size=crop, cat_max_ratio=0.75, flip_prob=0.5, photo=D.PhotoMetricDistortionClips())
    fr, lr, scale = rs(f_t, l_t)
    img, gt, params = fmt(fr, lr, last_label_host=lr[-1].cpu().numpy())
    # the oracle's chain with the decisions the host classes drew
    nw, nh = CV.rescale_size(w, h, scale)
    rf = np.stack([CV.resize_linear_u8(x, nw, nh) for x in frames])
    rl = np.stack([CV.resize_nearest(x, nw, nh) for x in labels])
    assert np.array_equal(fr.cpu().numpy(), rf) and np.array_equal(lr.cpu().numpy(), rl)
    y1, x1, ch, cw, ph = params['y1'], params['x1'], params['ch'], params['cw'], params['photo']
    mean, stdinv = np.float32(fmt.mean), (1 / np.float64(fmt.std)).astype(np.float32)
    ho, wo = max(crop[0], ch), max(crop[1], cw)
    want_img = np.zeros((t, 3, ho, wo), np.float32)
    want_gt = np.full((t, 1, ho, wo), 255, np.int64)
    for i in range(t):
        c_, l_ = rf[i, y1:y1 + ch, x1:x1 + cw], D.reduce_zero_label(rl[i, y1:y1 + ch, x1:x1 + cw])
        if params['flip']:
            c_, l_ = c_[:, ::-1], l_[:, ::-1]
        beta = None if ph['beta'][i] != ph['beta'][i] else ph['beta'][i]
        alpha = None if ph['alpha'][i] != ph['alpha'][i] else ph['alpha'][i]
        out = CV.photometric_frame(c_, beta, alpha, ph['contrast_first'][i], ph['saturation'][i], ph['hue'][i])
        want_img[i, :, :ch, :cw] = (((out[:, :, ::-1].astype(np.float32) - mean) * stdinv).astype(np.float32)).transpose(2, 0, 1)
        want_gt[i, 0, :ch, :cw] = l_
    assert np.array_equal(img.cpu().numpy(), want_img) and np.array_equal(gt.cpu().numpy(), want_gt), (seed, scale, params['flip'])
    return scale


def test_train_pipeline_emulated():
    with emu.active():
        scales = [run_train_pipeline(torch.device('cpu'), seed, 3, 30, 53, (53, 30), (32, 40)) for seed in (71, 72, 73)]
    assert len(set(scales)) == 3


@pytest.mark.gpu
def test_train_pipeline_gpu():
    """VSPW geometry at a quarter of the linear size (120 x 213 frames, (213, 120) scale, 120 x 120 crops) over several random streams,
    and once at full size (480 x 854, (853, 480), 480 x 480)."""
    for seed in (81, 82, 83, 84):
        run_train_pipeline(torch.device('cuda:0'), seed, 4, 120, 213, (213, 120), (120, 120))
    run_train_pipeline(torch.device('cuda:0'), 85, 4, 480, 854, (853, 480), (480, 480))


def test_clip_resize_emulated():
    with emu.active():
        run_resize_cases(torch.device('cpu'))


def test_photometric_hsv_emulated():
    with emu.active():
        run_photo_hsv_cases(torch.device('cpu'))


@pytest.mark.gpu
def test_clip_resize_gpu():
    run_resize_cases(torch.device('cuda:0'))


@pytest.mark.gpu
def test_photometric_hsv_gpu():
    run_photo_hsv_cases(torch.device('cuda:0'))


@pytest.mark.gpu
def test_clip_resize_full_size_properties_gpu():
    """VSPW-sized frames (480 x 854, 4 frames): the properties that need no oracle run -- size-preserving resize is the identity, a 2x
    shrink is the box mean, labels only take source values, a constant image stays constant under any ratio of the training range."""
    dev = torch.device('cuda:0')
    frames, labels = _clip(50, 4, 480, 854)
    f_t, l_t = torch.from_numpy(frames).to(dev), torch.from_numpy(labels).to(dev)
    f, l = D.resize_clip(f_t, l_t, (854, 480))
    assert torch.equal(f, f_t) and torch.equal(l, l_t)
    f, l = D.resize_clip(f_t, l_t, (427, 240))
    a = f_t.int()
    box = (a[:, 0::2, 0::2] + a[:, 0::2, 1::2] + a[:, 1::2, 0::2] + a[:, 1::2, 1::2] + 2) >> 2
    assert torch.equal(f.int(), box) and torch.equal(l, l_t[:, 0::2, 0::2])
    const = torch.full((1, 480, 854, 3), 93, dtype=torch.uint8, device=dev)
    for ratio in (0.5, 0.77, 1.3, 2.0):
        scale = (int(853 * ratio), int(480 * ratio))
        f, _ = D.ResizeClips().apply(const, None, scale)
        assert f.shape[1:3] == tuple(reversed(CV.rescale_size(854, 480, scale))) and bool((f == 93).all())
    f, l, _ = D.AlignedResizeClips()(f_t, l_t)
    assert f.shape == (4, 480, 864, 3) and l.shape == (4, 480, 864)
    sample = CV.resize_linear_u8(CV.imrescale(frames[1], (853, 480)), 864, 480)
    assert np.array_equal(f[1].cpu().numpy(), sample)
