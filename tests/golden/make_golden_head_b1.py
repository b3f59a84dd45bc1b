#!/usr/bin/env python
"""Golden vectors of the whole CFFM-B1 head at the BASELINE size (cfg2 / cfg3: 480x480, T = 4, depths = 2), produced by
the REFERENCE head (`CFFMHead_clips_resize1_8`, cffm_head.py:99-157, built through its own registry from /root/reference
with stand-ins for the absent mmcv/timm: oracle/ref_import.py) on 1 clip x 4 frames of backbone-shaped features
(120 / 60 / 30 / 15 px).  The full tensors are large (train logits 5 x 124 x 120 x 120), so the fixture keeps a stride-4
spatial sample plus sum / abs-sum / square-sum of each tensor, the loss / accuracy of `losses()` on seeded 480x480 labels
and the same statistics of the feature gradients.  Run in the build container only (about a minute of CPU):
python tests/golden/make_golden_head_b1.py          (head_b1_480.npz)
python tests/golden/make_golden_head_b1.py c4       (head_b1_512_b2.npz: BASELINE config 4, 512x512, 2 clips, stride-8 sample)
python tests/golden/make_golden_head_b1.py pp       (headpp_b1_480_k8.npz: BASELINE config 5, the CFFM++ head
                                                     `CFFMHead_clips_resize1_8_finetune_w_prototype3`, cffm_head.py:423-535, with 8 prototypes)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import recipe as R, ref_import as RI  # noqa: E402
from tests.golden.make_golden_head import feature_maps, labels  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
B1 = (64, 128, 320, 512)
SIZE, STRIDE = 480, 4


def stats(t):
    t = torch.as_tensor(t).double()
    return np.array([float(t.sum()), float(t.abs().sum()), float(t.square().sum()), float(t.abs().max())])


def sample(t, stride=STRIDE):
    return np.ascontiguousarray(torch.as_tensor(t)[..., ::stride, ::stride].numpy())


DFEAT_STRIDE = (4, 4, 2, 1)


# BASELINE config 4 ("CFFM-B2 512x512, batch 2 per GPU": same C = 256 / depths 2 head, grid 64x64 -> padded 70x70): the second fixture
SIZE_C4, BATCH_C4, STRIDE_C4 = 512, 2, 8
DFEAT_STRIDE_C4 = (16, 8, 4, 2)


def main(size=SIZE, batch=1, stride=STRIDE, dstride=DFEAT_STRIDE, out_name='head_b1_480.npz', seeds=(70, 71, 72)):
    torch.manual_seed(0)
    d = {}
    head = RI.build_reference_head(in_channels=B1, depths=2)
    head.dropout.p = 0.0                      # train mode deterministic (the reference cannot be built with ratio 0)
    res = head.load_state_dict(R.synth_state(head, seed=seeds[0]), strict=False)
    assert not res.unexpected_keys
    feats = feature_maps(batch, 4, size, chans=B1, seed=seeds[1])
    head.eval()
    with torch.no_grad():
        y = head(feats, batch, 4)                                                      # [B,124,size/4,size/4]
    d['eval_logits_s4'], d['eval_logits_stats'] = sample(y, stride), stats(y)
    head.train()
    fg = [f.clone().requires_grad_(True) for f in feats]
    out = head(fg, batch, 4)                                                           # [B,5,124,size/4,size/4]
    d['train_logits_s4'], d['train_logits_stats'] = sample(out.detach(), stride), stats(out.detach())
    loss = head.losses(out, labels(batch, 4, size, seed=seeds[2]))
    d['loss_seg'] = loss['loss_seg'].detach().numpy()
    d['acc_seg'] = loss['acc_seg'].detach().numpy()
    loss['loss_seg'].backward()
    for i, f in enumerate(fg):
        d['dfeat%d_s' % i] = sample(f.grad, dstride[i])
        d['dfeat%d_stats' % i] = stats(f.grad)
    np.savez_compressed(os.path.join(OUT, out_name), **d)
    for k, v in d.items():
        print(k, v.shape, float(np.abs(v).max()))


def main_pp(size=SIZE, stride=STRIDE, out_name='headpp_b1_480_k8.npz', seeds=(90, 91, 92)):
    """BASELINE config 5 at head level: CFFM++-B1, 480x480, T = 4, 8 global-context prototypes read from <save_path>/<video>/centers.pt
    (cffm_head.py:431-457).  Eval logits (x2 + 0.5 x3, :531-533), train logits, and -- after sum(train logits).backward() -- the gradient
    statistics of every parameter the fine-tuning trains (decoder_swin.*, linear_pred3.*)."""
    import tempfile
    torch.manual_seed(0)
    torch.Tensor.cuda = lambda self, *a, **k: self           # cffm_head.py:455 hard-codes .cuda()
    d = {}
    pp = RI.build_reference_head(kind='CFFMHead_clips_resize1_8_finetune_w_prototype3', in_channels=B1, depths=2)
    pp.dropout.p = pp.dropout3.p = 0.0
    res = pp.load_state_dict(R.synth_state(pp, seed=seeds[0]), strict=False)
    assert not res.unexpected_keys
    feats = feature_maps(1, 4, size, chans=B1, seed=seeds[1])
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, 'vid0'))
        torch.save(R.synth_input('centers', (1, 8, 256), seed=seeds[2], scale=1.0), os.path.join(tmp, 'vid0', 'centers.pt'))
        pp.save_path = tmp + '/'
        metas = [{'filename': tmp + '/data/vid0/origin/0001.jpg'}]
        pp.eval()
        with torch.no_grad():
            y = pp(feats, 1, 4, None, metas)
        d['eval_logits_s4'], d['eval_logits_stats'] = sample(y, stride), stats(y)
        pp.train()
        out = pp(feats, 1, 4, None, metas)
        d['train_logits_s4'], d['train_logits_stats'] = sample(out.detach(), stride), stats(out.detach())
        d['train_shape'] = np.array(out.shape)
        (out * torch.as_tensor(np.random.RandomState(seeds[2]).randn(*out.shape[-3:]).astype(np.float32))).sum().backward()
    names = []
    for n, p_ in pp.named_parameters():
        if p_.grad is not None:
            names.append(n)
            d['pg/' + n] = stats(p_.grad)
    d['trained'] = np.array(sorted({n.split('.')[0] for n in names}))
    np.savez_compressed(os.path.join(OUT, out_name), **d)
    for k, v in d.items():
        print(k, v.shape, v if v.dtype.kind in 'US' else float(np.abs(v).max()))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'pp':
        main_pp()
    elif len(sys.argv) > 1 and sys.argv[1] == 'c4':
        main(SIZE_C4, BATCH_C4, STRIDE_C4, DFEAT_STRIDE_C4, 'head_b1_512_b2.npz', (80, 81, 82))
    else:
        main()
