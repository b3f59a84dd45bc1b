import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vss_cffm_amd as V
from vss_cffm_amd import ops
dev = torch.device('cuda:0')
m = V.BasicLayer3d3(dim=256, depth=2, num_heads=8, window_size=7, expand_size=3, pool_method='fc', focal_level=2, focal_window=5).to(dev)
x = torch.randn(1, 4, 256, 8, 8, device=dev); gy = torch.randn(1, 4, 256, 8, 8, device=dev)
acc = {'bwd_py': 0.0, 'n': 0}
orig = ops._LayerFn.backward
def timed(ctx, dy):
    t0 = time.perf_counter(); r = orig(ctx, dy); acc['bwd_py'] += time.perf_counter() - t0; acc['n'] += 1; return r
ops._LayerFn.backward = staticmethod(timed)
def fb():
    for p in m.parameters(): p.grad = None
    m(x).backward(gy)
for _ in range(20): fb()
acc.update(bwd_py=0.0, n=0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): fb()
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / 200 * 1e3
print('fwd+bwd %.3f ms; inside _LayerFn.backward (python + C launches) %.3f ms' % (tot, acc['bwd_py'] / acc['n'] * 1e3))
