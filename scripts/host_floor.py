"""Host-side cost of one training step: a grid so small (8x8) that the GPU is never the bottleneck."""
import os, sys, time, torch, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vss_cffm_amd as V
dev = torch.device('cuda:0')
m = V.BasicLayer3d3(dim=256, depth=2, num_heads=8, window_size=7, expand_size=3, pool_method='fc', focal_level=2, focal_window=5).to(dev)
opt = V.optim.AdamW(m.parameters(), lr=6e-5)
x = torch.randn(1, 4, 256, 8, 8, device=dev); gy = torch.randn(1, 4, 256, 8, 8, device=dev)
def step():
    opt.zero_grad(set_to_none=True); y = m(x); y.backward(gy); opt.step()
for _ in range(20): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize(); print('host floor: %.3f ms/step' % ((time.perf_counter() - t0) / 200 * 1e3))
def part(f, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('forward only      %.3f' % part(lambda: m(x)))
def fb():
    opt.zero_grad(set_to_none=True); m(x).backward(gy)
print('zero+fwd+bwd      %.3f' % part(fb))
print('optimizer only    %.3f' % part(lambda: opt.step()))
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
