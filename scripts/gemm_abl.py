import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vss_cffm_amd import _lib
dev = torch.device('cuda:0'); P = lambda t: C.c_void_p(t.data_ptr())
for path in sys.argv[1:]:
    lib = _lib.bind(os.path.abspath(path)); out = []
    for (M, N) in ((7200, 1024), (7200, 256)):
        for K in (256, 1024):
            x, w, y = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev)
            for _ in range(3): lib.cffm_linear_fwd(P(x), P(w), P(y), M, N, K, None)
            torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): lib.cffm_linear_fwd(P(x), P(w), P(y), M, N, K, None)
            e1.record(); torch.cuda.synchronize()
            out.append('N=%d,K=%d:%.1f' % (N, K, e0.elapsed_time(e1) * 1e3 / 30))
    print(path, ' '.join(out))
