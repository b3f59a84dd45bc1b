import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vss_cffm_amd import _lib
_lib._lib = _lib.bind(os.path.abspath(sys.argv[1])); lib = _lib.get()
import vss_cffm_amd as V
dev = torch.device('cuda:0')
m = V.BasicLayer3d3(dim=256, depth=1, num_heads=8, window_size=7, expand_size=3, pool_method='fc', focal_level=2, focal_window=5).to(dev)
x = torch.randn(2, 4, 256, 60, 60, device=dev) * 1.5
for _ in range(3):
    for p in m.parameters(): p.grad = None
    m(x)[:, -1].square().sum().backward()
torch.cuda.synchronize()
buf = (C.c_longlong * 64)(); lib.cffm_debug_bwq.argtypes = [C.c_void_p]; print('rc', lib.cffm_debug_bwq(buf))
t = [list(buf[8 * i: 8 * i + 7]) for i in range(6)]
names = ['kv_store', 'barrier', 'prefetch+Dq', 'loop', 'dq store', 'barrier2']
for i, r in enumerate(t):
    if r[0] == 0: continue
    print('window', i, ' '.join('%s=%d' % (n, r[k + 1] - r[k]) for k, n in enumerate(names)), 'total', r[6] - r[0], 'gap_to_next', (t[i + 1][0] - r[6]) if i + 1 < 6 and t[i + 1][0] else 0)
