#!/usr/bin/env python
"""Micro-benchmark of the Linear GEMM entry points at the block's shapes (B=2 clips): prints us and TFLOP/s."""
import argparse, ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vss_cffm_amd import _lib
ap = argparse.ArgumentParser(); ap.add_argument('--lib', default=None); a = ap.parse_args()
lib = _lib.bind(os.path.abspath(a.lib)) if a.lib else _lib.get()
dev = torch.device('cuda:0'); P = lambda t: C.c_void_p(t.data_ptr())
shapes = [('qkv', 10368, 768, 256), ('proj', 7200, 256, 256), ('fc1', 7200, 1024, 256), ('fc2', 7200, 256, 1024)]
out = []
for name, M, N, K in shapes:
    x, w, y = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev)
    dy, dx, dw = torch.randn(M, N, device=dev), torch.empty(M, K, device=dev), torch.empty(N, K, device=dev)
    for form, fn, args in (('fwd', lib.cffm_linear_fwd, (P(x), P(w), P(y))), ('dx', lib.cffm_linear_bwd_input, (P(dy), P(w), P(dx))),
                           ('dw', lib.cffm_linear_bwd_weight, (P(dy), P(x), P(dw)))):
        for _ in range(3): fn(*args, M, N, K, None)
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn(*args, M, N, K, None)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        out.append('%s.%s=%.1fus(%.0fTF)' % (name, form, us, 2.0 * M * N * K / us / 1e6))
print((a.lib or 'product') + ': ' + ' '.join(out))
