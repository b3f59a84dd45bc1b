"""Feasibility: capture forward + backward of the layer in one torch.cuda.CUDAGraph and time the replay."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vss_cffm_amd as V
dev = torch.device('cuda:0')
m = V.BasicLayer3d3(dim=256, depth=2, num_heads=8, window_size=7, expand_size=3, pool_method='fc', focal_level=2, focal_window=5).to(dev)
x = torch.randn(2, 4, 256, 60, 60, device=dev) * 1.5; gy = torch.zeros(2, 4, 256, 60, 60, device=dev); gy[:, -1].normal_()
def body():
    for p in m.parameters(): p.grad = None
    y = m(x); y.backward(gy)
def T(f, n=100):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(10): body()
print('eager fwd+bwd   %.3f ms' % T(body))
ref = {k: p.grad.clone() for k, p in m.named_parameters()}
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): body()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
g.replay(); torch.cuda.synchronize()
err = max(float((p.grad - ref[k]).abs().max() / (ref[k].abs().max() + 1e-30)) for k, p in m.named_parameters())
print('graph replay    %.3f ms   max rel grad diff vs eager %.2e' % (T(g.replay), err))
