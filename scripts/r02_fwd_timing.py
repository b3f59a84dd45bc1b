"""Shader-clock stamps inside k_cfm_attn_fwd (build with -DFWD_TIMING): per sampled workgroup (every 24th), cycles from its own
start to: 1 gathers landed, 2 LDS stores done, 3 barrier passed, 4 S + max done, 5 PV done, 6 end; and its start relative to
workgroup 0.  usage: python scripts/r02_fwd_timing.py build/fwdt.so"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vss_cffm_amd import _lib
_lib._lib = _lib.bind(os.path.abspath(sys.argv[1]))
lib = _lib.get()
import vss_cffm_amd as V
from vss_cffm_amd import ops
raw = C.CDLL(os.path.abspath(sys.argv[1]))
dev = torch.device('cuda:0')
b, GRID = 2, 60
g = ops.make_geom(lib, b, GRID, GRID)
key_src, q_dst = ops.device_tables(GRID, GRID, dev)[:2]
qkv = (torch.randn(b * g.RC, 768) * 0.5).half().to(dev)
biasf = (torch.randn(8 * 4 * 10 * 512) * 0.5).half().to(dev)      # pair fragments (round 4)
ao = torch.empty(b * g.HW, 256, device=dev); lse = torch.empty(b * g.nW * 8, 64, device=dev)
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for _ in range(5):
    lib.cffm_attn_fwd(C.byref(g), P(qkv), P(key_src), P(q_dst), P(biasf), P(ao), P(lse), st)
torch.cuda.synchronize()
buf = (C.c_longlong * 512)()
assert raw.cffm_debug_fwd_stamps(buf) == 0
t0 = buf[0]
print('wg    start   gathers  ldsstore  barrier   S+max     PV      end   (cycles; start relative to workgroup 0)')
for i in range(54):
    r = [buf[i * 8 + k] for k in range(7)]
    if r[0] == 0: continue
    print('%4d %8d %8d %8d %8d %8d %8d %8d' % (i * 24, r[0] - t0, r[1] - r[0], r[2] - r[0], r[3] - r[0], r[4] - r[0], r[5] - r[0], r[6] - r[0]))
