"""k_cfm_attn_fwd alone, back to back (200 launches between one pair of events), for a library build; output checked against the
product build on the same inputs.  usage: python scripts/r06_fwd_b2b.py <lib.so> [batch] [grid]   (env switches pass through)"""
import ctypes as C
import os
import sys

import torch

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, R)
from vss_cffm_amd import _lib, ops  # noqa: E402

lib_path = os.path.abspath(sys.argv[1])
b = int(sys.argv[2]) if len(sys.argv) > 2 else 2
grid = int(sys.argv[3]) if len(sys.argv) > 3 else 60
dev = torch.device('cuda:0')
prod = _lib.bind(_lib.LIB_PATH)
lib = _lib.bind(lib_path)
g = ops.make_geom(prod, b, grid, grid)
key_src, q_dst = ops.device_tables(grid, grid, dev)[:2]
gen = torch.Generator().manual_seed(3)
qkv = (torch.randn(b * g.RC, 768, generator=gen) * 0.5).half().to(dev)
biasf = (torch.randn(8 * 4 * 10 * 512, generator=gen) * 0.5).half().to(dev)
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
outs = []
for L in (prod, lib):
    ao = torch.zeros(b * g.HW, 256, device=dev)
    lse = torch.zeros(b * g.nW * 8, 64, device=dev)
    assert L.cffm_attn_fwd(C.byref(g), P(qkv), P(key_src), P(q_dst), P(biasf), P(ao), P(lse), st) == 0
    torch.cuda.synchronize()
    outs.append((ao, lse))
same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
ts = []
for L in (lib,):
    ao, lse = outs[1]
    run = lambda: L.cffm_attn_fwd(C.byref(g), P(qkv), P(key_src), P(q_dst), P(biasf), P(ao), P(lse), st)
    for rep in range(3):
        for _ in range(20):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(1e3 * e0.elapsed_time(e1) / 200)
print('%s B=%d grid=%d slots=%s: %s us  bit-identical to product: %s' % (os.path.basename(lib_path), b, grid, os.environ.get('CFFM_ATTN_FWD_SLOTS', '-'),
                                                                        ' '.join('%.2f' % t for t in ts), same))
