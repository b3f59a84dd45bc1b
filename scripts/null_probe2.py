import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vss_cffm_amd import _lib, ops
_lib._lib = _lib.bind(os.path.abspath(sys.argv[1]))
import vss_cffm_amd as V
dev = torch.device('cuda:0')
m = V.BasicLayer3d3(dim=256, depth=2, num_heads=8, window_size=7, expand_size=3, pool_method='fc', focal_level=2, focal_window=5).to(dev)
opt = V.optim.AdamW(m.parameters(), lr=6e-5)
x = torch.randn(2, 4, 256, 60, 60, device=dev); gy = torch.zeros(2, 4, 256, 60, 60, device=dev); gy[:, -1].normal_()
acc = {'f': 0.0, 'b': 0.0, 'n': 0}
of, ob = ops._LayerFn.forward, ops._LayerFn.backward
def tf(ctx, *a):
    t0 = time.perf_counter(); r = of(ctx, *a); acc['f'] += time.perf_counter() - t0; return r
def tb(ctx, dy):
    t0 = time.perf_counter(); r = ob(ctx, dy); acc['b'] += time.perf_counter() - t0; acc['n'] += 1; return r
ops._LayerFn.forward = staticmethod(tf); ops._LayerFn.backward = staticmethod(tb)
def T(f, n=300):
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('forward                     %.3f' % T(lambda: m(x)))
def fb(): m(x).backward(gy)
for p in m.parameters(): p.grad = None
acc.update(f=0.0, b=0.0, n=0)
t = T(fb); print('forward+backward (accum)    %.3f   inside fwd fn %.3f  inside bwd fn %.3f' % (t, acc['f'] / (acc['n'] or 1) * 1e3, acc['b'] / (acc['n'] or 1) * 1e3))
def zfb(): opt.zero_grad(set_to_none=True); m(x).backward(gy)
print('zero_grad+forward+backward  %.3f' % T(zfb))
def full(): opt.zero_grad(set_to_none=True); m(x).backward(gy); opt.step()
print('full step                   %.3f' % T(full))
y = m(x)
print('cat only                    %.3f' % T(lambda: torch.cat([x[:, :-1], y[:, -1:]], dim=1)))
