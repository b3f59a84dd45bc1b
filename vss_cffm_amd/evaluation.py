"""Segmentation metrics on the device: the counting part of mmseg/core/evaluation/metrics.py
(`intersect_and_union` :62-119, `total_intersect_and_union` :176-225) in libcffm_hip.so, the ratios of `eval_metrics`
(:296-350) in float64 on whatever device the counts live on.  Exact integers: results equal the reference's numpy histograms."""
import ctypes as C

import torch

from . import _lib
from .ops import _ptr, _require_int64, _stream


def intersect_and_union(pred_label, label, num_classes, ignore_index, label_map=None, reduce_zero_label=False, out=None):
    """(area_intersect, area_union, area_pred_label, area_label), int64 [num_classes] each, for one prediction / label map pair
    (any shape, int64, on the GPU).  `out`: an int64 [3, num_classes] tensor to ACCUMULATE (intersect | pred | label) into --
    what `total_intersect_and_union` does over a list of images."""
    lib = _lib.get()
    pred_label, label = _require_int64(pred_label, 'prediction map'), _require_int64(label, 'label map')
    if pred_label.shape != label.shape:
        raise _lib.CffmError('intersect_and_union: prediction %s vs label %s' % (tuple(pred_label.shape), tuple(label.shape)))
    if label_map:                                    # metrics.py:92-95, applied in place on a copy, in dict order like the reference
        label = label.clone()
        for old_id, new_id in label_map.items():
            label[label == old_id] = new_id
    counts = out if out is not None else torch.zeros(3, num_classes, dtype=torch.int64, device=label.device)
    if counts.shape != (3, num_classes) or counts.dtype != torch.int64 or not counts.is_contiguous():
        raise _lib.CffmError('intersect_and_union: `out` must be a contiguous int64 [3, %d] tensor' % num_classes)
    _lib.check(lib.cffm_seg_counts(_ptr(pred_label), _ptr(label), label.numel(), int(num_classes), int(ignore_index),
                                   int(bool(reduce_zero_label)), _ptr(counts), _stream(label)), lib)
    return counts[0], counts[1] + counts[2] - counts[0], counts[1], counts[2]


def total_intersect_and_union(results, gt_seg_maps, num_classes, ignore_index, label_map=None, reduce_zero_label=False):
    assert len(results) == len(gt_seg_maps)
    dev = gt_seg_maps[0].device if len(gt_seg_maps) else 'cpu'
    counts = torch.zeros(3, num_classes, dtype=torch.int64, device=dev)
    for r, g in zip(results, gt_seg_maps):
        intersect_and_union(r, g, num_classes, ignore_index, label_map, reduce_zero_label, out=counts)
    return counts[0], counts[1] + counts[2] - counts[0], counts[1], counts[2]


def eval_metrics(results, gt_seg_maps, num_classes, ignore_index, metrics=('mIoU',), nan_to_num=None, label_map=None,
                 reduce_zero_label=False):
    """[all_acc, acc per class, then IoU and / or Dice per class] as float64 tensors (metrics.py:296-350)."""
    if isinstance(metrics, str):
        metrics = [metrics]
    if not set(metrics).issubset({'mIoU', 'mDice'}):
        raise KeyError('metrics {} is not supported'.format(metrics))
    inter, union, pred, lab = [t.double() for t in total_intersect_and_union(results, gt_seg_maps, num_classes, ignore_index,
                                                                              label_map, reduce_zero_label)]
    ret = [inter.sum() / lab.sum(), inter / lab]
    for m in metrics:
        ret.append(inter / union if m == 'mIoU' else 2 * inter / (pred + lab))
    if nan_to_num is not None:
        ret = [torch.nan_to_num(r, nan=float(nan_to_num)) for r in ret]
    return ret


def video_consistency(gt_frames, pred_frames, clip_num):
    """VC_perclip.py:62-78 `get_common` for one video: gt / pred [F,h,w] int64 label maps -> float64 [F - clip_num] accuracies
    (|pixels stable over clip_num frames in gt AND pred| / |stable in gt|; 0/0 = nan as in numpy) and the int64 [F - clip_num, 2]
    counts behind them.  The mean over all videos' accuracies is the reported VC_8 / VC_16."""
    lib = _lib.get()
    gt_frames, pred_frames = _require_int64(gt_frames, 'ground-truth frames'), _require_int64(pred_frames, 'predicted frames')
    if gt_frames.dim() != 3 or gt_frames.shape != pred_frames.shape:
        raise _lib.CffmError('video_consistency: [F,h,w] maps expected, got %s / %s' % (tuple(gt_frames.shape), tuple(pred_frames.shape)))
    f, h, w = gt_frames.shape
    m = max(f - int(clip_num), 0)
    counts = torch.zeros(m, 2, dtype=torch.int64, device=gt_frames.device)
    _lib.check(lib.cffm_vc_counts(_ptr(gt_frames), _ptr(pred_frames), f, h * w, int(clip_num), _ptr(counts), _stream(gt_frames)), lib)
    return counts[:, 0].double() / counts[:, 1].double(), counts
