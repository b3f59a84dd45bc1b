"""Clip data path: frame-index rules and the fused crop / flip / normalise / pad / stack kernel against golden
vectors the REFERENCE's own classes produced (tests/golden/make_golden_clip.py), and against those classes live when
/root/reference is present."""
import random

import numpy as np
import pytest
import torch

from oracle import ref_import as RI
from tests import emu, helpers as H
from tests.golden.make_golden_clip import CLIP_CASES, synth_clip
from vss_cffm_amd import _lib, data as D


def test_test_time_clip_indices_match_reference_golden():
    g = H.load_golden('clip_pipeline')
    for n in (1, 3, 5, 9, 10, 14):
        want = [[int(v) for v in row if v >= 0] for row in g['test_idx/%d' % n]]
        assert [D.clip_indices_test(i, n) for i in range(n)] == want, n
    assert D.clip_indices_test(2, 30) == [2] and D.clip_indices_test(3, 30) == [0, 1, 2, 3] and D.clip_indices_test(8, 30) == [0, 3, 6, 8]
    assert D.clip_indices_test(9, 30) == [0, 3, 6, 9] and D.clip_indices_test(29, 30) == [20, 23, 26, 29]
    assert D.clip_indices_test(5, 30, dilation=(-4, -2)) == [1, 3, 5]      # the hand-picked clips belong to the default dilation only


def test_training_clip_indices_match_reference_golden():
    g = H.load_golden('clip_pipeline')
    for seed, n, fv, *want in g['train_idx'].tolist():
        np.random.seed(seed)
        random.seed(seed)
        got = D.clip_indices_train(n, flip_video=bool(fv))
        if want[0] < 0:
            assert got is None
            continue
        rev, idx = got
        names = list(range(n))[::-1] if rev else list(range(n))
        assert [names[i] for i in idx] == want, (seed, n, fv)
        assert idx[3] - idx[0] == 9 and idx[3] >= 9


def run_clip_cases(device):
    g = H.load_golden('clip_pipeline')
    for seed, crop in CLIP_CASES:
        frames, labels = synth_clip(seed)
        np.random.seed(100 + seed)                    # the same stream the reference classes consumed
        fmt = D.ClipFormatter(crop_size=crop, cat_max_ratio=0.75, flip_prob=0.5)
        img, gt, params = fmt(torch.from_numpy(frames).to(device), torch.from_numpy(labels).to(device), last_label_host=labels[-1])
        assert params['flip'] == bool(g['clip/%d/flip' % seed])
        assert img.shape == g['clip/%d/img' % seed].shape and gt.dtype == torch.int64
        assert torch.equal(gt.cpu(), torch.from_numpy(g['clip/%d/gt' % seed]))                 # integer work: equality
        torch.testing.assert_close(img.cpu(), torch.from_numpy(g['clip/%d/img' % seed]), rtol=0, atol=1e-6)
    # PhotoMetricDistortion_clips between flip and normalisation: brightness / contrast per frame, drawn in the reference's order
    # (goldens: the reference class itself, on seeded streams that never take its cv2-HSV branches)
    from tests.golden.make_golden_clip import PHOTO_CASES
    for clip_seed, seed, t in PHOTO_CASES:
        frames, labels = synth_clip(clip_seed, t=t)
        np.random.seed(seed)
        fmt = D.ClipFormatter(crop_size=(64, 96), cat_max_ratio=0.75, flip_prob=0.5, photo=D.PhotoMetricDistortionClips())
        img, gt, params = fmt(torch.from_numpy(frames).to(device), torch.from_numpy(labels).to(device), last_label_host=labels[-1])
        assert params['flip'] == bool(g['photo/%d/flip' % clip_seed]) and len(params['photo']['beta']) == t
        assert torch.equal(gt.cpu(), torch.from_numpy(g['photo/%d/gt' % clip_seed]))
        torch.testing.assert_close(img.cpu(), torch.from_numpy(g['photo/%d/img' % clip_seed]), rtol=0, atol=1e-6)
        assert any(v == v for v in params['photo']['beta'] + params['photo']['alpha'])       # (something was actually distorted)
    # a stream that draws saturation / hue (applied by default since round 5: tests/test_data_cv.py): the policies that leave the two
    # steps out -- ONE warning per instance with on_hsv='warn', silently with on_hsv='skip', refused with on_hsv='raise'
    np.random.seed(3)
    with pytest.raises(_lib.CffmError):
        for _ in range(8):
            D.PhotoMetricDistortionClips(on_hsv='raise').draw(4)
    np.random.seed(3)
    dflt = D.PhotoMetricDistortionClips(on_hsv='warn')
    with pytest.warns(RuntimeWarning, match='saturation / hue'):
        for _ in range(8):
            dflt.draw(4)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        for _ in range(8):
            dflt.draw(4)                      # (warned once: silent from then on)
    np.random.seed(3)
    ph = D.PhotoMetricDistortionClips(on_hsv='skip').draw(4)
    assert len(ph['saturation']) == 4 and len(ph['hue']) == 4
    # image only (test-time clips have no labels), explicit parameters
    frames, _ = synth_clip(7)
    fmt = D.ClipFormatter(crop_size=(90, 150), cat_max_ratio=1.0, flip_prob=0.5)
    img, gt = fmt.apply(torch.from_numpy(frames).to(device), None, dict(y1=0, x1=0, ch=90, cw=150, flip=True))
    want = ((frames[:, :, ::-1, ::-1].astype(np.float32) - np.float32(fmt.mean)) * (1 / np.float64(fmt.std)).astype(np.float32))
    assert gt is None
    torch.testing.assert_close(img.cpu(), torch.from_numpy(np.ascontiguousarray(want.transpose(0, 3, 1, 2))), rtol=0, atol=1e-6)
    with pytest.raises(_lib.CffmError):
        fmt.apply(torch.from_numpy(frames).to(device), None, dict(y1=80, x1=0, ch=20, cw=150, flip=False))   # box past the frame
    with pytest.raises(_lib.CffmError):
        fmt.apply(torch.from_numpy(frames).to(device).float(), None, dict(y1=0, x1=0, ch=90, cw=150, flip=False))


def test_clip_formatter_emulated():
    with emu.active():
        run_clip_cases(torch.device('cpu'))


@pytest.mark.gpu
def test_clip_formatter_gpu():
    run_clip_cases(torch.device('cuda:0'))


def test_clip_formatter_needs_the_gpu_library():
    if not torch.cuda.is_available():
        frames, labels = synth_clip(1)
        with pytest.raises(_lib.CffmError):
            D.ClipFormatter(crop_size=(64, 64)).apply(torch.from_numpy(frames), None, dict(y1=0, x1=0, ch=64, cw=64, flip=False))


@pytest.mark.skipif(not RI.available(), reason='/root/reference not present')
def test_golden_is_what_the_reference_classes_produce_live():
    from tests.golden.make_golden_clip import reference_clip_pipeline, reference_test_indices, reference_train_indices
    g = H.load_golden('clip_pipeline')
    assert np.array_equal(np.array([s + [-1] * (4 - len(s)) for s in reference_test_indices(14)]), g['test_idx/14'])
    seed, n, fv, *want = g['train_idx'][7].tolist()
    got = reference_train_indices(n, seed, flip_video=bool(fv))
    assert [int(x[:4]) for x in got] == want
    seed, crop = CLIP_CASES[2]
    frames, labels = synth_clip(seed)
    img, gt, flip = reference_clip_pipeline(frames, labels, 100 + seed, crop)
    assert np.array_equal(img, g['clip/%d/img' % seed]) and np.array_equal(gt, g['clip/%d/gt' % seed])


def _video_tree(root, videos):
    import os
    for split, names in (('train', ['v_a', 'v_c']), ('val', ['v_b']), ('test', ['v_d'])):
        with open(os.path.join(root, split + '.txt'), 'w') as f:
            f.write(''.join(n + '\n' for n in names))
    for v, n in videos.items():
        for sub, suf in (('origin', '.jpg'), ('mask', '.png')):
            os.makedirs(os.path.join(root, 'data', v, sub))
            for i in range(n):
                open(os.path.join(root, 'data', v, sub, '%08d%s' % (3 * i + 1, suf)), 'w').close()


def test_clip_lister_lists_and_serves_clips(tmp_path):
    """VSPWDataset2-shaped lister (custom.py:1959-2100) over a directory-tree fixture."""
    root = str(tmp_path)
    _video_tree(root, {'v_a': 14, 'v_b': 5, 'v_c': 9, 'v_d': 12})
    tr = D.ClipLister(root, 'train')
    assert tr.videolists == ['v_a', 'v_c'] and len(tr) == 2 and len(tr.img_all) == 23
    np.random.seed(4); random.seed(4)
    it = tr.train_item(0)
    assert it['video'] == 'v_a' and len(it['frames']) == 4 and it['img_paths'][3].endswith('/data/v_a/origin/' + it['frames'][3])
    assert it['mask_paths'][0].endswith('.png') and '/mask/' in it['mask_paths'][0]
    order = tr.imgdic['v_a'][::-1] if it['reversed'] else tr.imgdic['v_a']
    pos = [order.index(n) for n in it['frames']]
    assert [p - pos[3] for p in pos] == [-9, -6, -3, 0]
    np.random.seed(4); random.seed(4)
    assert tr.train_item(1) is None                                  # 9 frames: no frame has 9 predecessors
    va = D.ClipLister(root, 'val')
    assert len(va) == 5 and [len(va.test_item(i)['frames']) for i in range(5)] == [1, 1, 1, 4, 4]
    assert va.test_item(4)['frames'] == [va.imgdic['v_b'][i] for i in (0, 2, 3, 4)]
    allv = D.ClipLister(root, 'train_val_generate_prototype')
    assert allv.videolists == ['v_a', 'v_c', 'v_b', 'v_d'] and len(allv) == 4


@pytest.mark.skipif(not RI.available(), reason='/root/reference not present')
def test_clip_lister_against_reference_dataset_live(tmp_path):
    """the same tree through the reference's CustomDataset_video2.__init__ (executed where it lies): identical lists and lengths."""
    from tests.golden.make_golden_clip import _reference_modules
    Cu, _, _ = _reference_modules()
    root = str(tmp_path)
    _video_tree(root, {'v_a': 14, 'v_b': 5, 'v_c': 9, 'v_d': 12})
    for split in ('train', 'val', 'train_val_generate_prototype'):
        try:
            ref = Cu.CustomDataset_video2(pipeline=[], img_dir='', split=split, data_root=root, dilation=[-9, -6, -3])
        except Exception as e:   # noqa: BLE001  (the stand-in mmcv may lack what Compose needs)
            pytest.skip('reference dataset not constructible here: %s' % e)
        mine = D.ClipLister(root, split)
        assert mine.videolists == ref.videolists and mine.imgdic == ref.imgdic and mine.img_all == ref.img_all and len(mine) == len(ref)
