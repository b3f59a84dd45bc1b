import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vss_cffm_amd as V
dev = torch.device('cuda:0')
m = V.BasicLayer3d3(dim=256, depth=2, num_heads=8, window_size=7, expand_size=3, pool_method='fc', focal_level=2, focal_window=5).to(dev)
opt = V.optim.AdamW(m.parameters(), lr=6e-5)
x = torch.randn(2, 4, 256, 60, 60, device=dev); gy = torch.zeros(2, 4, 256, 60, 60, device=dev); gy[:, -1].normal_()
builds = [0]
orig = opt._table
def counted(gi, ps):
    key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in ps)
    if key not in opt._tables.get(gi, {}): builds[0] += 1
    return orig(gi, ps)
opt._table = counted
def step():
    opt.zero_grad(set_to_none=True); m(x).backward(gy); opt.step()
for _ in range(5): step()
torch.cuda.synchronize(); b0 = builds[0]; t0 = time.perf_counter()
for _ in range(40): step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('ms/step %.3f table builds in 40 steps: %d (warmup %d)' % (dt / 40 * 1e3, builds[0] - b0, b0))
# host-only time of one step (no sync): how far ahead the host runs
t0 = time.perf_counter()
for _ in range(40): step()
h = time.perf_counter() - t0; torch.cuda.synchronize()
print('host enqueue time per step %.3f ms' % (h / 40 * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(40): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(28)
