"""The LDS-DMA weight-gradient kernel (csrc/dw_kernels.h) against the grouped register-staged kernel on the block's three large problems
(B = 2: q|k|v 10368 x 768 x 256, fc1 7200 x 1024 x 256, fc2 7200 x 256 x 1024), each alone and launched back to back."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vss_cffm_amd import _lib  # noqa: E402

if os.environ.get('CFFM_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['CFFM_LIB'])
lib = _lib.get()
dev = torch.device('cuda:0')
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class WGrad(C.Structure):
    _fields_ = [('dy', C.c_void_p), ('x', C.c_void_p), ('dw', C.c_void_p), ('M', C.c_long), ('N', C.c_int), ('K', C.c_int)]


def timed(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


shapes = [(10368, 768, 256), (7200, 1024, 256), (7200, 256, 1024)]
ops = []
for M, N, K in shapes:
    dy, x = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
    dys, xs = torch.empty_like(dy), torch.empty_like(x)
    lib.cffm_split4(P(dy), P(dys), M * N, st); lib.cffm_split4(P(x), P(xs), M * K, st)
    dw1, dw2 = torch.empty(N, K, device=dev), torch.empty(N, K, device=dev)
    ops.append((M, N, K, dy, x, dys, xs, dw1, dw2))
for M, N, K, dy, x, dys, xs, dw1, dw2 in ops:
    prob = (WGrad * 1)(WGrad(dy.data_ptr(), x.data_ptr(), dw1.data_ptr(), M, N, K))
    t_old = timed(lambda: lib.cffm_linear_bwd_weight_group(prob, 1, st))
    t_new = timed(lambda: lib.cffm_linear_bwd_weight_split(P(dys), P(xs), P(dw2), M, N, K, st))
    ref = dy.double().T @ x.double()
    e1, e2 = float((dw1.double() - ref).norm() / ref.norm()), float((dw2.double() - ref).norm() / ref.norm())
    fl = 2.0 * M * N * K
    print('%5d x %4d x %4d  register-staged %.1f us (%.0f TF, err %.1e)   LDS-DMA %.1f us (%.0f TF, err %.1e)' % (M, N, K, t_old, fl / t_old / 1e6, e1, t_new, fl / t_new / 1e6, e2))
prob3 = (WGrad * 3)(*[WGrad(o[3].data_ptr(), o[4].data_ptr(), o[7].data_ptr(), o[0], o[1], o[2]) for o in ops])
t_old = timed(lambda: lib.cffm_linear_bwd_weight_group(prob3, 3, st))
prob3s = (WGrad * 3)(*[WGrad(o[5].data_ptr(), o[6].data_ptr(), o[8].data_ptr(), o[0], o[1], o[2]) for o in ops])
t_new = timed(lambda: lib.cffm_linear_bwd_weight_split_group(prob3s, 3, st))
print('all three as one group (one launch + one slab sum): register-staged %.1f us; LDS-DMA %.1f us' % (t_old, t_new))
for o in ops:
    ref = o[3].double().T @ o[4].double()
    print('  group result err %.1e' % float((o[8].double() - ref).norm() / ref.norm()))
