#!/usr/bin/env python
"""Profiling helper: per-stage HIP-event times of one fwd+bwd step for a given build of the library
(`--lib path`, default the product library).  Used for ablation builds (-DCFFM_ABLATE=...)."""
import argparse, ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vss_cffm_amd import _lib
ap = argparse.ArgumentParser(); ap.add_argument('--lib', default=None); ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--batch', type=int, default=2); a = ap.parse_args()
if a.lib: _lib._lib = _lib.bind(os.path.abspath(a.lib))
lib = _lib.get()
import vss_cffm_amd as V
dev = torch.device('cuda:0')
m = V.BasicLayer3d3(dim=256, depth=2, num_heads=8, window_size=7, expand_size=3, pool_method='fc', focal_level=2, focal_window=5).to(dev)
x = torch.randn(a.batch, 4, 256, 60, 60, device=dev) * 1.5
gy = torch.randn(a.batch, 256, 60, 60, device=dev) * 1e-6
def step():
    for p in m.parameters(): p.grad = None
    (m(x)[:, -1] * gy).sum().backward()
for _ in range(5): step()
n = lib.cffm_profile_stage_count(); ms, cnt = (C.c_float * n)(), (C.c_int * n)()
lib.cffm_profile_collect(ms, cnt)
torch.cuda.synchronize(); import time; t0 = time.perf_counter()
for _ in range(a.steps): step()
torch.cuda.synchronize(); clean = time.perf_counter() - t0
lib.cffm_profile_enable(-1)
torch.cuda.synchronize(); import time; t0 = time.perf_counter()
for _ in range(a.steps): step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
lib.cffm_profile_enable(0); lib.cffm_profile_collect(ms, cnt)
print('%s: %.3f ms/step (%.3f with stage events)' % (a.lib or 'product', 1e3 * clean / a.steps, 1e3 * dt / a.steps), ' '.join('%s=%.1fus' % (lib.cffm_profile_stage_name(i).decode(), 1e3 * ms[i] / cnt[i]) for i in range(n) if cnt[i]))
