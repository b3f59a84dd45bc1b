import ctypes as C, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vss_cffm_amd as V
from vss_cffm_amd import _lib
lib = _lib.get(); dev = torch.device('cuda:0')
m = V.BasicLayer3d3(dim=256, depth=2, num_heads=8, window_size=7, expand_size=3, pool_method='fc', focal_level=2, focal_window=5).to(dev)
opt = V.optim.AdamW(m.parameters(), lr=6e-5)
x = torch.randn(2, 4, 256, 60, 60, device=dev) * 1.5; gy = torch.zeros(2, 4, 256, 60, 60, device=dev); gy[:, -1].normal_(); gy *= float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
def step():
    opt.zero_grad(set_to_none=True); m(x).backward(gy); opt.step()
def timed(n=60):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(10): step()
names = [lib.cffm_profile_stage_name(i).decode() for i in range(lib.cffm_profile_stage_count())]
n = len(names); ms, cnt = (C.c_float * n)(), (C.c_int * n)()
for rep in range(3):
    lib.cffm_profile_enable(0); a = timed()
    lib.cffm_profile_enable(1 << names.index('cfm_attn_fwd')); b = timed(); lib.cffm_profile_collect(ms, cnt)
    print('mask off %.4f ms   attn_fwd events on %.4f ms' % (a, b))
lib.cffm_profile_enable(0)
