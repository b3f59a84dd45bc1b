"""Shader-clock stamps inside k_cfm_attn_fwd (build with -DCFFM_EXPERIMENTS -DFWD_TIMING): per sampled workgroup (every 20th) and
wave, cycles from the workgroup's first stamp to: 1 loads arrived, 2 bias products issued, 3 DMA landed, 4 barrier passed, 5 S + max
done, 6 query-48 weights written, 7 PV done, 8 stores issued, 9 last barrier, 10 end; and the workgroup's start relative to workgroup 0.
usage: python scripts/r05_fwd_timing.py build/libcffm_fwdt.so"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vss_cffm_amd import _lib, ops
lib = _lib.bind(os.path.abspath(sys.argv[1]))
raw = C.CDLL(os.path.abspath(sys.argv[1]))
dev = torch.device('cuda:0')
b, GRID = 2, 60
g = ops.make_geom(lib, b, GRID, GRID)
key_src, q_dst = ops.device_tables(GRID, GRID, dev)[:2]
qkv = (torch.randn(b * g.RC, 768) * 0.5).half().to(dev)
biasf = (torch.randn(8 * 4 * 10 * 512) * 0.5).half().to(dev)
ao = torch.empty(b * g.HW, 256, device=dev); lse = torch.empty(b * g.nW * 8, 64, device=dev)
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for _ in range(5):
    lib.cffm_attn_fwd(C.byref(g), P(qkv), P(key_src), P(q_dst), P(biasf), P(ao), P(lse), st)
torch.cuda.synchronize()
N = 64 * 3 * 12
buf = (C.c_longlong * N)()
assert raw.cffm_debug_fwd_stamps(buf) == 0
t0 = min(buf[i * 36] for i in range(64) if buf[i * 36])
print('  wg w    start    loads  biasMM   dmaIn barrier   S+max  q48sm      PV  stores   bar2     end')
for i in range(64):
    for wv in range(3):
        r = [buf[(i * 3 + wv) * 12 + k] for k in range(11)]
        if r[0] == 0: continue
        base = min(buf[(i * 3 + k) * 12] for k in range(3))
        print('%4d %d %8d ' % (i * 20, wv, r[0] - t0) + ' '.join('%7d' % (r[k] - base if r[k] else -1) for k in range(1, 11)))
