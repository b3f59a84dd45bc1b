#!/usr/bin/env python
"""Separates the fixed cost (launch, prologue, epilogue) of the Linear GEMM kernels from their per-k-step cost by sweeping the
contraction length at the block's output shapes."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vss_cffm_amd import _lib
lib = _lib.get(); dev = torch.device('cuda:0'); P = lambda t: C.c_void_p(t.data_ptr())
def timeit(fn, args, tail, n=30):
    for _ in range(3): fn(*args, *tail, None)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn(*args, *tail, None)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for name, M, N in (('fwd N=1024', 7200, 1024), ('fwd N=256', 7200, 256), ('fwd N=768', 10368, 768)):
    out = []
    for K in (32, 64, 128, 256, 512, 1024):
        x, w, y = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev)
        out.append('K=%d:%.1f' % (K, timeit(lib.cffm_linear_fwd, (P(x), P(w), P(y)), (M, N, K))))
    print(name, ' '.join(out))
for name, N, K in (('dx out=256 (contract N)', 1024, 256), ('dx out=1024', 256, 1024)):
    out = []
    for NN in (64, 128, 256, 512, 1024):
        M = 7200
        dy, w, dx = torch.randn(M, NN, device=dev), torch.randn(NN, K, device=dev), torch.empty(M, K, device=dev)
        out.append('N=%d:%.1f' % (NN, timeit(lib.cffm_linear_bwd_input, (P(dy), P(w), P(dx)), (M, NN, K))))
    print(name, ' '.join(out))
for name, N, K in (('dw 1024x256', 1024, 256), ('dw 256x256', 256, 256), ('dw 768x256', 768, 256)):
    out = []
    for M in (900, 1800, 3600, 7200, 14400, 28800):
        dy, x, dw = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev), torch.empty(N, K, device=dev)
        out.append('M=%d:%.1f' % (M, timeit(lib.cffm_linear_bwd_weight, (P(dy), P(x), P(dw)), (M, N, K))))
    print(name, ' '.join(out))
