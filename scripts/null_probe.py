import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vss_cffm_amd import _lib
if len(sys.argv) > 1: _lib._lib = _lib.bind(os.path.abspath(sys.argv[1]))
import vss_cffm_amd as V
dev = torch.device('cuda:0')
m = V.BasicLayer3d3(dim=256, depth=2, num_heads=8, window_size=7, expand_size=3, pool_method='fc', focal_level=2, focal_window=5).to(dev)
opt = V.optim.AdamW(m.parameters(), lr=6e-5)
x = torch.randn(2, 4, 256, 60, 60, device=dev); gy = torch.zeros(2, 4, 256, 60, 60, device=dev); gy[:, -1].normal_()
def step():
    opt.zero_grad(set_to_none=True); m(x).backward(gy); opt.step()
for _ in range(20): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize(); print('%s: %.3f ms/step' % (sys.argv[1] if len(sys.argv) > 1 else 'product', (time.perf_counter() - t0) / 200 * 1e3))
