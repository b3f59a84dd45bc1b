"""The streaming weight-gradient kernel (csrc/dws_kernels.h, operands in T-frag storage) against the grouped register-staged kernel on the
block's four problems (B = 2: q|k|v 10368 x 768 x 256, fc1 7200 x 1024 x 256, fc2 7200 x 256 x 1024, proj 7200 x 256 x 256), each alone and
as one group (one launch + one slab sum)."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vss_cffm_amd import _lib  # noqa: E402

if os.environ.get('CFFM_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['CFFM_LIB'])
lib = _lib.get()
lib.cffm_tfrag_floats.restype = C.c_long
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


shapes = [(10368, 768, 256), (7200, 1024, 256), (7200, 256, 1024), (7200, 256, 256)]
ops = []
for M, N, K in shapes:
    dy, x = torch.randn(M, N, device=dev), torch.randn(M, K, device=dev)
    dyt, xt = torch.empty(lib.cffm_tfrag_floats(M, N), device=dev), torch.empty(lib.cffm_tfrag_floats(M, K), device=dev)
    assert lib.cffm_tfrag_pack(P(dy), P(dyt), M, N, st) == 0 and lib.cffm_tfrag_pack(P(x), P(xt), M, K, st) == 0
    dw1, dw2 = torch.empty(N, K, device=dev), torch.empty(N, K, device=dev)
    ops.append((M, N, K, dy, x, dyt, xt, dw1, dw2))
for M, N, K, dy, x, dyt, xt, dw1, dw2 in ops:
    prob = (WGrad * 1)(WGrad(dy.data_ptr(), x.data_ptr(), dw1.data_ptr(), M, N, K))
    t_old = timed(lambda: lib.cffm_linear_bwd_weight_group(prob, 1, st))
    t_new = timed(lambda: lib.cffm_linear_bwd_weight_tfrag(P(dyt), P(xt), P(dw2), M, N, K, st))
    ref = dy.double().T @ x.double()
    e1, e2 = float((dw1.double() - ref).norm() / ref.norm()), float((dw2.double() - ref).norm() / ref.norm())
    fl = 2.0 * M * N * K
    print('%5d x %4d x %4d  register-staged %.1f us (%.0f TF, err %.1e)   streaming %.1f us (%.0f TF, err %.1e)' % (M, N, K, t_old, fl / t_old / 1e6, e1, t_new, fl / t_new / 1e6, e2))
for sel, name in (((0, 1, 2, 3), 'all four'), ((1, 2, 3), 'fc1 + fc2 + proj')):
    n = len(sel)
    pa = (WGrad * n)(*[WGrad(ops[i][3].data_ptr(), ops[i][4].data_ptr(), ops[i][7].data_ptr(), ops[i][0], ops[i][1], ops[i][2]) for i in sel])
    pb = (WGrad * n)(*[WGrad(ops[i][5].data_ptr(), ops[i][6].data_ptr(), ops[i][8].data_ptr(), ops[i][0], ops[i][1], ops[i][2]) for i in sel])
    t_old = timed(lambda: lib.cffm_linear_bwd_weight_group(pa, n, st))
    t_new = timed(lambda: lib.cffm_linear_bwd_weight_tfrag_group(pb, n, st))
    fl = sum(2.0 * ops[i][0] * ops[i][1] * ops[i][2] for i in sel)
    print('%s as one group (one launch + one slab sum): register-staged %.1f us (%.0f TF); streaming %.1f us (%.0f TF; %.2f of the 833 TF three-pass peak)'
          % (name, t_old, fl / t_old / 1e6, t_new, fl / t_new / 1e6, fl / t_new / 1e6 / 833.))
    for i in sel:
        ref = ops[i][3].double().T @ ops[i][4].double()
        print('  group result err %.1e' % float((ops[i][8].double() - ref).norm() / ref.norm()))
