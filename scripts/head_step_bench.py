#!/usr/bin/env python
"""Whole CFFM-B1 decode head, one training step (forward_train + backward; no optimizer) on 2 clips x 4 frames of 480x480:
what the rows next to the hot path buy at the level of the head.  Four settings of the two class switches:
(fuse_impl, loss_impl) in {hip, torch}^2 -- the hot path itself (decoder_focal) is libcffm_hip.so in all of them.
One JSON line: ms per step (HIP events, median of 10) and peak memory for each."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vss_cffm_amd as V  # noqa: E402
from vss_cffm_amd.head import revert_sync_batchnorm  # noqa: E402

B1 = (64, 128, 320, 512)


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    cfg = dict(type='CFFMHead_clips_resize1_8', in_channels=list(B1), in_index=[0, 1, 2, 3], feature_strides=[4, 8, 16, 32],
               channels=128, dropout_ratio=0.1, num_classes=124, norm_cfg=dict(type='SyncBN', requires_grad=True),
               align_corners=False, decoder_params=dict(embed_dim=256, depths=2),
               loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0), num_clips=4)
    head = revert_sync_batchnorm(V.build_head(cfg)).to(dev).train()     # one process: plain BatchNorm statistics
    gen = torch.Generator().manual_seed(1)
    feats = [torch.randn(8, c, 480 // s, 480 // s, generator=gen).to(dev).requires_grad_(True) for c, s in zip(B1, (4, 8, 16, 32))]
    labels = torch.randint(0, 124, (2, 4, 1, 480, 480), generator=gen)
    labels[torch.rand(2, 4, 1, 480, 480, generator=gen) < 0.05] = 255
    labels = labels.to(dev)
    out = {}
    only = os.environ.get('HEAD_STEP_ONLY')       # e.g. hip,hip : one setting (for rocprofv3)
    for fuse in ('hip', 'torch'):
        for loss in ('hip', 'torch'):
            if only and only != '%s,%s' % (fuse, loss):
                continue
            head.fuse_impl, head.loss_impl = fuse, loss

            def step():
                for p in head.parameters():
                    p.grad = None
                for f in feats:
                    f.grad = None
                res = head.forward_train(feats, None, labels, None, 2, 4)
                res['loss_seg'].backward()
                return res
            for _ in range(3):
                res = step()
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats(dev)
            base = torch.cuda.memory_allocated(dev)
            times = []
            for _ in range(10):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                step()
                e1.record()
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1))
            times.sort()
            out['fuse=%s,loss=%s' % (fuse, loss)] = {
                'ms_per_step': round(times[len(times) // 2], 3), 'peak_mb': round((torch.cuda.max_memory_allocated(dev) - base) / 2 ** 20, 1),
                'loss_seg': round(float(res['loss_seg']), 5), 'acc_seg': round(float(res['acc_seg']), 4)}
    out['workload'] = 'CFFM-B1 head, 2 clips x 4 frames of 480x480, forward_train + backward, dropout 0.1, BatchNorm in train mode'
    print(json.dumps(out))


if __name__ == '__main__':
    main()
