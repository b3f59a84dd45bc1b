#!/usr/bin/env python
"""Golden vectors of the whole CFFM head, produced by the REFERENCE head built through its own registry
(mmseg.models imported from /root/reference with stand-ins for the absent mmcv/timm: oracle/ref_import.py).
Run in the build container only:  python tests/golden/make_golden_head.py
Stores outputs only; parameters and inputs are regenerated from oracle/recipe.py."""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import recipe as R, ref_import as RI  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
B0 = (32, 64, 160, 256)


def feature_maps(b, t, size, chans=B0, seed=31):
    """Backbone-shaped inputs for an image of `size` px: strides 4/8/16/32, leading dim b*t (frame-major)."""
    return [R.synth_input('c%d' % i, (b * t, c, size // s, size // s), seed=seed, scale=1.0)
            for i, (c, s) in enumerate(zip(chans, (4, 8, 16, 32)))]


def labels(b, t, size, seed=32):
    rs = np.random.RandomState(seed)
    lab = rs.randint(0, 124, size=(b, t, 1, size, size))
    lab[rs.rand(*lab.shape) < 0.05] = 255
    return torch.from_numpy(lab).long()


def main():
    torch.manual_seed(0)
    d = {}
    # ---- CFFM head, B0 shape (BASELINE cfg1: 64x64 frames), dropout p forced to 0 so train mode is deterministic
    head = RI.build_reference_head()          # dropout_ratio 0.1 as configured; p is zeroed below so that
    head.dropout.p = 0.0                      # train mode is deterministic (the reference cannot be built with 0)
    st = R.synth_state(head, seed=30)
    res = head.load_state_dict(st, strict=False)
    assert not res.unexpected_keys
    feats = feature_maps(1, 4, 64)
    head.eval()
    with torch.no_grad():
        d['eval_logits'] = head(feats, 1, 4).numpy()                                   # [1,124,16,16]
        d['eval_t2_logits'] = head(feature_maps(1, 2, 64, seed=33), 1, 2).numpy()      # short-circuit (cffm_head.py:127)
    head.train()
    feats_g = [f.clone().requires_grad_(True) for f in feats]
    out = head(feats_g, 1, 4)                                                          # [1,5,124,16,16]
    d['train_logits'] = out.detach().numpy()
    loss = head.losses(out, labels(1, 4, 64))
    d['loss_seg'] = loss['loss_seg'].detach().numpy()
    d['acc_seg'] = loss['acc_seg'].detach().numpy()
    loss['loss_seg'].backward()
    for i, f in enumerate(feats_g):
        d['dfeat%d' % i] = f.grad.numpy()
    # ---- CFFM++ head (8 prototypes, BASELINE cfg5), eval: x2 + 0.5*x3
    torch.Tensor.cuda = lambda self, *a, **k: self           # cffm_head.py:455 hard-codes .cuda()
    pp = RI.build_reference_head(kind='CFFMHead_clips_resize1_8_finetune_w_prototype3')
    pp.dropout.p = pp.dropout3.p = 0.0
    pp.load_state_dict(R.synth_state(pp, seed=34), strict=False)
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, 'vid0'))
        centers = R.synth_input('centers', (1, 8, 256), seed=35, scale=1.0)
        torch.save(centers, os.path.join(tmp, 'vid0', 'centers.pt'))
        pp.save_path = tmp + '/'
        metas = [{'filename': tmp + '/data/vid0/origin/0001.jpg'}]
        pp.eval()
        with torch.no_grad():
            d['pp_eval_logits'] = pp(feats, 1, 4, None, metas).numpy()
        pp.train()
        d['pp_train_logits'] = pp(feats, 1, 4, None, metas).detach().numpy()
    np.savez_compressed(os.path.join(OUT, 'head_b0_64.npz'), **d)
    import json
    keys = {}
    for kind in ('CFFMHead_clips_resize1_8', 'CFFMHead_clips_resize1_8_gene_prototype',
                 'CFFMHead_clips_resize1_8_finetune_w_prototype3'):
        for name, chans, depths in (('B0', B0, 1), ('B1', (64, 128, 320, 512), 2)):
            m = RI.build_reference_head(kind=kind, in_channels=chans, depths=depths)
            keys['%s/%s' % (kind, name)] = [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()]
    json.dump(keys, open(os.path.join(OUT, 'head_state_dict_keys.json'), 'w'))
    for k, v in d.items():
        print(k, v.shape, float(np.abs(v).max()))


if __name__ == '__main__':
    main()
