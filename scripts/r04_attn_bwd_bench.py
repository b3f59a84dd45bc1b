"""cffm_attn_bwd alone at BASELINE's size (B = 2, 60 x 60): us per call (fused kernel + bias-tile sum + dK / dV gather), n calls between
one pair of events.  Usage: python scripts/r04_attn_bwd_bench.py [lib.so]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vss_cffm_amd import _lib, ops
from tests.test_attn_bwd import bias_buffer, P
lib = _lib.bind(sys.argv[1]) if len(sys.argv) > 1 else _lib.get()
dev = torch.device('cuda')
b, h0, w0 = 2, 60, 60
g = ops.make_geom(lib, b, h0, w0)
nw, rc, hw = g.nW, g.RC, g.HW
ks, qd, ip, ii = ops.device_tables(h0, w0, dev)
torch.manual_seed(0)
qkv = (torch.randn(b * rc, 768, device=dev) * 0.7).half()
bias = torch.randn(8, 64, 304) * 0.5
bb = bias_buffer(bias).to(dev)
ao, dao = torch.randn(b * hw, 256, device=dev), torch.randn(b * hw, 256, device=dev) * 1e-3
lse = torch.full((b * nw * 8, 64), 8.0, device=dev)
dqkv = torch.zeros(b * rc, 768, device=dev)
dbt = torch.zeros(8, 304, 64, device=dev)
ws = torch.zeros(b * nw * (304 * 256 + 8), device=dev)      # f16 partial rows + their scales
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
call = lambda: lib.cffm_attn_bwd(C.byref(g), P(qkv), P(ks), P(qd), P(ip), P(ii), P(bb), P(ao), P(dao), P(lse), P(dqkv), P(dbt), P(ws), st)
for _ in range(5):
    assert call() == 0
torch.cuda.synchronize()
n = 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    call()
e1.record()
torch.cuda.synchronize()
print('attn_bwd stage: %.1f us per call (KS=%s groups=%s)' % (e0.elapsed_time(e1) * 1e3 / n, '-', os.environ.get('CFFM_BWD_GROUPS')))
