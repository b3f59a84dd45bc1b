#!/usr/bin/env python
"""Golden vectors of the SegFormer embedding in front of the hot path (cffm_head.py:102-119), produced by the REFERENCE
head's own sub-modules (linear_c1..4, mmseg.ops.resize, linear_fuse.conv) imported from /root/reference.
Run in the build container only:  python tests/golden/make_golden_fuse.py
Stores outputs only (pre-BatchNorm map, gradients of the features and of the nine parameters for a fixed upstream
gradient); parameters and inputs are regenerated from oracle/recipe.py by `case_inputs`."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import recipe as R, ref_import as RI  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
B0 = (32, 64, 160, 256)
# name -> (frames, feature sizes c1..c4): a 64x64 frame, and a 52x120 frame (odd sizes, non-integer resize factors)
CASES = {'sq': (2, [(16, 16), (8, 8), (4, 4), (2, 2)]), 'ragged': (1, [(13, 30), (7, 15), (4, 8), (2, 4)])}
PARAMS = ['linear_c%d.proj.%s' % (i, k) for i in (1, 2, 3, 4) for k in ('weight', 'bias')] + ['linear_fuse.conv.weight']


def case_inputs(name, chans=B0):
    n, sizes = CASES[name]
    feats = [R.synth_input('fuse_%s_c%d' % (name, i), (n, c, h, w), seed=51, scale=1.0)
             for i, (c, (h, w)) in enumerate(zip(chans, sizes))]
    gy = R.synth_input('fuse_%s_gy' % name, (n, 256) + tuple(sizes[0]), seed=52, scale=1.0)
    return feats, gy


def reference_fuse(head, feats):
    """the reference's lines 102-119 with its own modules, stopping before linear_fuse.bn"""
    from mmseg.ops import resize
    c1, c2, c3, c4 = feats
    n = c4.shape[0]
    _c4 = head.linear_c4(c4).permute(0, 2, 1).reshape(n, -1, c4.shape[2], c4.shape[3])
    _c4 = resize(_c4, size=c1.size()[2:], mode='bilinear', align_corners=False)
    _c3 = head.linear_c3(c3).permute(0, 2, 1).reshape(n, -1, c3.shape[2], c3.shape[3])
    _c3 = resize(_c3, size=c1.size()[2:], mode='bilinear', align_corners=False)
    _c2 = head.linear_c2(c2).permute(0, 2, 1).reshape(n, -1, c2.shape[2], c2.shape[3])
    _c2 = resize(_c2, size=c1.size()[2:], mode='bilinear', align_corners=False)
    _c1 = head.linear_c1(c1).permute(0, 2, 1).reshape(n, -1, c1.shape[2], c1.shape[3])
    return head.linear_fuse.conv(torch.cat([_c4, _c3, _c2, _c1], dim=1))


def main():
    torch.manual_seed(0)
    head = RI.build_reference_head()
    head.load_state_dict(R.synth_state(head, seed=50), strict=False)
    d = {}
    for name in CASES:
        feats, gy = case_inputs(name)
        fg = [f.clone().requires_grad_(True) for f in feats]
        head.zero_grad()
        y = reference_fuse(head, fg)
        y.backward(gy)
        d[name + '/y'] = y.detach().numpy()
        for i, f in enumerate(fg):
            d['%s/dfeat%d' % (name, i)] = f.grad.numpy()
        sd = dict(head.named_parameters())
        for k in PARAMS:
            d['%s/d.%s' % (name, k)] = sd[k].grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'fuse_b0.npz'), **d)
    for k, v in d.items():
        print(k, v.shape, float(np.abs(v).max()))


if __name__ == '__main__':
    main()
