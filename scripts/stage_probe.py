import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vss_cffm_amd import _lib
lib = _lib.get(); dev = torch.device('cuda:0'); P = lambda t: C.c_void_p(t.data_ptr())
n = 7200
xt, y, bp, g, b = (torch.randn(n, 256, device=dev), torch.randn(n, 256, device=dev), torch.randn(256, device=dev), torch.randn(256, device=dev), torch.randn(256, device=dev))
x1, z2, mean, rstd = torch.empty(n, 256, device=dev), torch.empty(n, 256, device=dev), torch.empty(n, device=dev), torch.empty(n, device=dev)
for _ in range(20): lib.cffm_residual_ln(P(xt), n * 256, n, P(y), P(bp), P(g), P(b), P(x1), P(z2), P(mean), P(rstd), n, None)
torch.cuda.synchronize()
