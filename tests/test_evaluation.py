"""Evaluation counts on the device: vss_cffm_amd.evaluation / cffm_seg_counts against
  * the REFERENCE's own `intersect_and_union` / `eval_metrics` (mmseg/core/evaluation/metrics.py), imported live when
    /root/reference is present, and against golden counts it produced (tests/golden/eval_counts.npz),
  * the numpy restatement below (np.histogram with bins = arange(K+1): the last bin is closed) at other sizes, class counts,
    reduce_zero_label, label_map, out-of-range values.
Integer work: every count must be EQUAL."""
import numpy as np
import pytest
import torch

from oracle import ref_import as RI
from tests import emu, helpers as H
from vss_cffm_amd import _lib, evaluation as E


def np_intersect_and_union(pred, label, k, ignore, label_map=None, reduce_zero_label=False):
    """metrics.py:62-119 restated (numpy; inputs int64)"""
    label = label.copy()
    if label_map:
        for old_id, new_id in label_map.items():
            label[label == old_id] = new_id
    if reduce_zero_label:
        label[label == 0] = 255
        label = label - 1
        label[label == 254] = 255
    mask = label != ignore
    pred, label = pred[mask], label[mask]
    inter = pred[pred == label]
    bins = np.arange(k + 1)
    ai, ap, al = (np.histogram(a, bins=bins)[0] for a in (inter, pred, label))
    return ai, ap + al - ai, ap, al


def cases():
    rs = np.random.RandomState(7)
    out = []
    for (shape, k, ignore, rz, lm, noise) in [((2, 64, 64), 124, 255, False, None, 0.3), ((1, 37, 53), 19, 255, False, None, 0.5),
                                              ((3, 20, 31), 150, 255, True, None, 0.4), ((1, 16, 16), 5, 255, False, {3: 1, 4: 255}, 0.2),
                                              ((1, 9, 9), 1, 255, False, None, 0.0), ((0, 4, 4), 7, 255, False, None, 0.0)]:
        label = rs.randint(0, k + (1 if rz else 0), size=shape).astype(np.int64)
        label[rs.rand(*shape) < 0.06] = 255
        pred = label.copy()
        flip = rs.rand(*shape) < noise
        pred[flip] = rs.randint(0, k, size=int(flip.sum()))
        pred[pred == 255] = 0
        if k == 19 and pred.size:      # values numpy's histogram treats specially: v == K lands in the last bin, v > K and v < 0 are dropped
            pred.flat[:4] = [19, 20, -1, 18]
            label.flat[:4] = [19, 19, 3, 19]
        out.append((pred, label, k, ignore, rz, lm))
    return out


def run_cases(device):
    for pred, label, k, ignore, rz, lm in cases():
        want = np_intersect_and_union(pred, label, k, ignore, lm, rz)
        got = E.intersect_and_union(torch.from_numpy(pred).to(device), torch.from_numpy(label).to(device), k, ignore, lm, rz)
        for a, b in zip(got, want):
            assert a.dtype == torch.int64 and np.array_equal(a.cpu().numpy(), b), (k, rz)
    # accumulation over a list of images + the ratios
    cs = [c for c in cases() if c[2] == 124] * 3
    preds, labels = [torch.from_numpy(c[0]).to(device) for c in cs], [torch.from_numpy(c[1]).to(device) for c in cs]
    tot = E.total_intersect_and_union(preds, labels, 124, 255)
    one = np_intersect_and_union(cs[0][0], cs[0][1], 124, 255)
    for a, b in zip(tot, one):
        assert np.array_equal(a.cpu().numpy(), 3 * b)
    all_acc, acc, iou, dice = E.eval_metrics(preds, labels, 124, 255, metrics=['mIoU', 'mDice'], nan_to_num=-1)
    ai, au, ap, al = [x.astype(np.float64) for x in one]
    with np.errstate(invalid='ignore', divide='ignore'):
        assert abs(float(all_acc) - ai.sum() / al.sum()) < 1e-12
        np.testing.assert_allclose(iou.cpu().numpy(), np.nan_to_num(ai / au, nan=-1), rtol=1e-12)
        np.testing.assert_allclose(dice.cpu().numpy(), np.nan_to_num(2 * ai / (ap + al), nan=-1), rtol=1e-12)
        np.testing.assert_allclose(acc.cpu().numpy(), np.nan_to_num(ai / al, nan=-1), rtol=1e-12)
    with pytest.raises(KeyError):
        E.eval_metrics(preds, labels, 124, 255, metrics=['mFoo'])
    with pytest.raises(_lib.CffmError):
        E.intersect_and_union(preds[0].int(), labels[0], 124, 255)
    with pytest.raises(_lib.CffmError):
        E.intersect_and_union(preds[0], labels[0][:, :10], 124, 255)
    with pytest.raises(_lib.CffmError):
        E.intersect_and_union(preds[0], labels[0], 5000, 255)


def test_numpy_restatement_against_reference_golden():
    g = H.load_golden('eval_counts')
    for i, (pred, label, k, ignore, rz, lm) in enumerate(cases()):
        if lm is None and pred.size:
            for name, b in zip(('inter', 'union', 'pred', 'label'), np_intersect_and_union(pred, label, k, ignore, lm, rz)):
                assert np.array_equal(g['%d/%s' % (i, name)], b), (i, name)


@pytest.mark.skipif(not RI.available(), reason='/root/reference not present')
def test_numpy_restatement_against_reference_live():
    from tests.golden.make_golden_eval import reference_counts
    for pred, label, k, ignore, rz, lm in cases():
        if pred.size:
            for a, b in zip(reference_counts(pred, label, k, ignore, lm, rz), np_intersect_and_union(pred, label, k, ignore, lm, rz)):
                assert np.array_equal(a, b)


def run_vc(device):
    """k_vc_counts against what the REFERENCE's get_common returned for the same seeded videos (tests/golden/vc_counts.npz,
    written by make_golden_vc.py from the function definition in /root/reference/VC_perclip.py:62-78)."""
    from tests.golden.make_golden_vc import CLIP_NUMS, vc_cases
    gold = H.load_golden('vc_counts')
    for name, gt, pred in vc_cases():
        f, h, w = gt.shape
        for n in CLIP_NUMS + (f, f + 3):
            want = gold['%s/%d' % (name, n)]
            acc, counts = E.video_consistency(torch.from_numpy(gt).to(device), torch.from_numpy(pred).to(device), n)
            assert acc.shape == want.shape and counts.shape == (len(want), 2)
            np.testing.assert_array_equal(acc.cpu().numpy(), want)       # same integer ratio in float64 (nan where 0/0)
    acc, counts = E.video_consistency(torch.zeros(5, 4, 4, dtype=torch.int64, device=device),
                                      torch.ones(5, 4, 4, dtype=torch.int64, device=device), 2)
    assert counts.cpu().tolist() == [[16, 16]] * 3
    with pytest.raises(_lib.CffmError):
        E.video_consistency(torch.zeros(5, 4, 4, dtype=torch.int64, device=device), torch.zeros(5, 4, 5, dtype=torch.int64, device=device), 2)


@pytest.mark.skipif(not RI.available(), reason='/root/reference not present')
def test_vc_golden_is_what_the_reference_function_returns_live():
    from tests.golden.make_golden_vc import CLIP_NUMS, reference_get_common, vc_cases
    get_common = reference_get_common()
    gold = H.load_golden('vc_counts')
    for name, gt, pred in vc_cases():
        f, h, w = gt.shape
        for n in CLIP_NUMS:
            with np.errstate(invalid='ignore', divide='ignore'):
                np.testing.assert_array_equal(np.asarray(get_common(list(gt), list(pred), n, h, w), dtype=np.float64), gold['%s/%d' % (name, n)])


def test_counts_emulated():
    with emu.active():
        run_cases(torch.device('cpu'))


def test_video_consistency_emulated():
    with emu.active():
        run_vc(torch.device('cpu'))


@pytest.mark.gpu
def test_video_consistency_gpu():
    run_vc(torch.device('cuda:0'))
    # a larger video (many workgroups per frame pair) against what the reference's own function returned for it (vc_counts.npz)
    from tests.golden.make_golden_vc import vc_large_case
    gold = H.load_golden('vc_counts')
    name, gt, pred = vc_large_case()
    for n in (8, 16):
        acc, _ = E.video_consistency(torch.from_numpy(gt).cuda(), torch.from_numpy(pred).cuda(), n)
        np.testing.assert_array_equal(acc.cpu().numpy(), gold['%s/%d' % (name, n)])


@pytest.mark.gpu
def test_counts_gpu():
    run_cases(torch.device('cuda:0'))
    # a whole VSPW-sized frame batch: 8 x 480 x 480, 124 classes, against numpy
    rs = np.random.RandomState(8)
    label = rs.randint(0, 124, size=(8, 480, 480)).astype(np.int64)
    label[rs.rand(8, 480, 480) < 0.05] = 255
    pred = np.where(rs.rand(8, 480, 480) < 0.7, label, rs.randint(0, 124, size=(8, 480, 480))).astype(np.int64)
    pred[pred == 255] = 1
    got = E.intersect_and_union(torch.from_numpy(pred).cuda(), torch.from_numpy(label).cuda(), 124, 255)
    for a, b in zip(got, np_intersect_and_union(pred, label, 124, 255)):
        assert np.array_equal(a.cpu().numpy(), b)
