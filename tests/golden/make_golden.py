#!/usr/bin/env python
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF (imported from /root/reference).

Run in the build container only:  python tests/golden/make_golden.py
Fixtures hold data only -- expected outputs / gradients of the reference modules
(`BasicLayer3d3`, `BasicLayer_cluster`) on inputs and parameters that are regenerated
from oracle/recipe.py (numpy MT19937, machine independent), so no reference source,
bytecode or pickled module is stored.

Loss used for the backward vectors:  L = sum(y[:, -1] * g),  g = recipe('g', seed 2).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import recipe as R, ref_import as RI  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

# name: (B, H, W, depth, full_small_grads)
LAYER_CASES = {
    'layer_b1_8x8_d1': (1, 8, 8, 1, True),       # BASELINE cfg1 hot-path shape (B0 head, 64x64 image)
    'layer_b2_8x8_d2': (2, 8, 8, 2, False),
    'layer_b1_14x21_d2': (1, 14, 21, 2, False),  # no padding, non-square
    'layer_b1_13x30_d1': (1, 13, 30, 1, False),  # padding on both axes
    'layer_b1_60x60_d2': (1, 60, 60, 2, False),  # BASELINE cfg2 hot-path shape (B1, 480x480)
}
GTC_CASES = {
    'gtc_b2_8x8_k8': (2, 8, 8, 8),
    'gtc_b1_13x30_k100': (1, 13, 30, 100),
}
BIG = 4096  # parameters with more elements than this are stored as row/col sums only


def _grad_entries(prefix, named_grads, full_small):
    out = {}
    for k, g in named_grads.items():
        g = g.detach().double()
        out['%s/gnorm/%s' % (prefix, k)] = np.float64(g.norm().item())
        if g.numel() <= BIG or (full_small and g.numel() <= 60000):
            out['%s/g/%s' % (prefix, k)] = g.float().numpy()
        elif g.dim() == 2:
            out['%s/gsum0/%s' % (prefix, k)] = g.sum(0).float().numpy()
            out['%s/gsum1/%s' % (prefix, k)] = g.sum(1).float().numpy()
        else:
            out['%s/gsumlast/%s' % (prefix, k)] = g.sum(-1).float().numpy()
    return out


def make_layer(name, b, h, w, depth, full_small):
    m = RI.build_basic_layer(depth)
    st = R.layer_state(depth, seed=0)
    res = m.load_state_dict(st, strict=False)
    assert not res.unexpected_keys and all(not m.state_dict()[k].dtype.is_floating_point for k in res.missing_keys)
    x = R.synth_input('x', (b, 4, 256, h, w), seed=1).requires_grad_(True)
    g = R.synth_input('g', (b, 256, h, w), seed=2, scale=1.0)
    y = m(x)
    assert torch.equal(y[:, :-1], x[:, :-1])
    (y[:, -1] * g).sum().backward()
    yt = y[:, -1].detach()
    d = {'meta': np.array([b, h, w, depth], dtype=np.int64)}
    stride = 1 if h * w <= 1024 else 3
    d['y_stride'] = np.int64(stride)
    d['y'] = yt[:, :, ::stride, ::stride].contiguous().numpy()
    d['y_stats'] = np.array([yt.double().sum().item(), yt.double().abs().sum().item(),
                             (yt.double() ** 2).sum().item()], dtype=np.float64)
    dx = x.grad.detach()
    d['dx'] = dx[:, :, :, ::stride, ::stride].contiguous().numpy()
    d['dx_stats'] = np.array([dx.double().sum().item(), dx.double().abs().sum().item(),
                              (dx.double() ** 2).sum().item()], dtype=np.float64)
    d.update(_grad_entries('p', {k: p.grad for k, p in m.named_parameters()}, full_small))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **d)
    print(name, 'y', d['y'].shape, 'max|y|', float(yt.abs().max()))


def make_gtc(name, b, h, w, k):
    m = RI.build_cluster_layer(1)
    st = R.gtc_layer_state(1, seed=3)
    res = m.load_state_dict(st, strict=False)
    assert not res.unexpected_keys
    x = R.synth_input('gx', (b, h * w, 256), seed=4).requires_grad_(True)
    c = R.synth_input('gc', (b, k, 256), seed=5).requires_grad_(True)
    g = R.synth_input('gg', (b, h * w, 256), seed=6, scale=1.0)
    y = m(x, h, w, c)[0]
    (y * g).sum().backward()
    d = {'meta': np.array([b, h, w, k], dtype=np.int64), 'y': y.detach().numpy(),
         'dx': x.grad.numpy(), 'dc': c.grad.numpy()}
    grads = {kk: p.grad for kk, p in m.named_parameters() if p.grad is not None}
    d['no_grad_keys'] = np.array(sorted(kk for kk, p in m.named_parameters() if p.grad is None))
    d.update(_grad_entries('p', grads, False))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **d)
    print(name, 'y', d['y'].shape)


if __name__ == '__main__':
    assert RI.available(), 'reference tree not found'
    torch.set_num_threads(8)
    for n, a in LAYER_CASES.items():
        make_layer(n, *a)
    for n, a in GTC_CASES.items():
        make_gtc(n, *a)
