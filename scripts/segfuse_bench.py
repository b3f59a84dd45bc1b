#!/usr/bin/env python
"""SegFormer embedding in front of the hot path: ops.segformer_fuse (composed per-scale GEMMs + one
full-resolution pass in libcffm_hip.so) against the reference's op sequence in stock PyTorch on the same GPU
(4 x Linear, 3 x F.interpolate, torch.cat to 1024 channels, 1x1 conv), CFFM-B1 480x480, N = 8 frames (2 clips x 4).
Prints one JSON line: ms per forward+backward (HIP events, median of 20), peak memory of each, per-kernel stage times."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vss_cffm_amd import ops  # noqa: E402

B1, SIZES, N = (64, 128, 320, 512), [(120, 120), (60, 60), (30, 30), (15, 15)], 8


def torch_seq(f, w, b, u):
    maps = []
    for i in (3, 2, 1, 0):
        m = F.linear(f[i].flatten(2).transpose(1, 2), w[i], b[i]).permute(0, 2, 1).reshape(N, 256, *SIZES[i])
        maps.append(m if i == 0 else F.interpolate(m, size=SIZES[0], mode='bilinear', align_corners=False))
    return F.conv2d(torch.cat(maps, 1), u)


def main():
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(0)
    feats = [torch.randn(N, c, *s, generator=gen).to(dev).requires_grad_(True) for c, s in zip(B1, SIZES)]
    lw = [(torch.randn(256, c, generator=gen) * 0.05).to(dev).requires_grad_(True) for c in B1]
    lb = [torch.randn(256, generator=gen).to(dev).requires_grad_(True) for _ in B1]
    fw = (torch.randn(256, 1024, 1, 1, generator=gen) * 0.03).to(dev).requires_grad_(True)
    gy = torch.randn(N, 256, *SIZES[0], generator=gen).to(dev)
    gy_cl = gy.contiguous(memory_format=torch.channels_last)    # what BatchNorm hands back for a channels-last input
    out = {}
    for name, fn, g in (('hip', ops.segformer_fuse, gy_cl), ('torch', torch_seq, gy)):
        def step():
            for t in feats + lw + lb + [fw]:
                t.grad = None
            fn(feats, lw, lb, fw).backward(g)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats(dev)
        base = torch.cuda.memory_allocated(dev)
        times = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            step()
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        times.sort()
        out[name] = {'ms_fwd_bwd': round(times[len(times) // 2], 3), 'min_ms': round(times[0], 3),
                     'peak_mb': round((torch.cuda.max_memory_allocated(dev) - base) / 2 ** 20, 1)}
    out['speedup'] = round(out['torch']['ms_fwd_bwd'] / out['hip']['ms_fwd_bwd'], 2)
    out['workload'] = 'CFFM-B1 480x480, 8 frames: features %s at %s' % (list(B1), SIZES)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
