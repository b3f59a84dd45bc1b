#!/usr/bin/env python
"""Experiment: capture one full step (fwd + bwd + AdamW) of the hot path in a HIP graph and compare replay vs eager."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vss_cffm_amd as V
dev = torch.device('cuda:0')
m = V.BasicLayer3d3(dim=256, depth=2, num_heads=8, window_size=7, expand_size=3, pool_method='fc', focal_level=2, focal_window=5).to(dev)
opt = torch.optim.AdamW(m.parameters(), lr=6e-5, weight_decay=0.01, fused=True, capturable=True)
x = torch.randn(2, 4, 256, 60, 60, device=dev) * 1.5
gy = torch.randn(2, 256, 60, 60, device=dev) * 1e-6
def step():
    opt.zero_grad(set_to_none=True)
    (m(x)[:, -1] * gy).sum().backward()
    opt.step()
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print('eager  %.3f ms/step' % timeit(step))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    (m(x)[:, -1] * gy).sum().backward()
    opt.step()
print('graph  %.3f ms/step' % timeit(g.replay))
