"""k_cfm_attn_fwd alone, back to back, for several builds of the library in ONE process (A/B of kernel variants and ablations).
usage: python scripts/r05_fwd_b2b.py [--b 2] [--rounds 3] lib1.so lib2.so ...   -> us per launch, per library, per round"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vss_cffm_amd import _lib, ops  # noqa: E402

args = sys.argv[1:]
b, rounds = 2, 3
while args and args[0].startswith('--'):
    if args[0] == '--b': b = int(args[1])
    if args[0] == '--rounds': rounds = int(args[1])
    args = args[2:]
dev = torch.device('cuda:0')
GRID = 60
libs = [(p, _lib.bind(os.path.abspath(p))) for p in args]
g = ops.make_geom(libs[0][1], b, GRID, GRID)
key_src, q_dst = ops.device_tables(GRID, GRID, dev)[:2]
gen = torch.Generator().manual_seed(3)
qkv = (torch.randn(b * g.RC, 768, generator=gen) * 0.5).half().to(dev)
biasf = (torch.randn(8 * 4 * 10 * 512, generator=gen) * 0.5).half().to(dev)
ao = torch.empty(b * g.HW, 256, device=dev)
lse = torch.empty(b * g.nW * 8, 64, device=dev)
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
res = {p: [] for p, _ in libs}
ref = None
for r in range(rounds):
    for p, lib in libs:
        run = lambda: lib.cffm_attn_fwd(C.byref(g), P(qkv), P(key_src), P(q_dst), P(biasf), P(ao), P(lse), st)
        for _ in range(20):
            assert run() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            run()
        e1.record()
        torch.cuda.synchronize(dev)
        res[p].append(1e3 * e0.elapsed_time(e1) / 200)
        if r == 0:
            if ref is None: ref = (ao.clone(), lse.clone())
            d = float((ao - ref[0]).abs().max() / ref[0].abs().max()); dl = float((lse - ref[1]).abs().max())
            print('%-34s vs first: ao %.2e lse %.2e' % (p, d, dl))
for p, _ in libs:
    print('%-34s %s us' % (p, ' '.join('%6.2f' % v for v in res[p])))
