"""The Linear GEMM entry points of the C ABI (forward, input gradient, weight gradient, grouped weight gradients) against
fp64 matmuls, at ragged shapes (rows not a multiple of the 32-deep K-tile, outputs not a multiple of the 64/128 tiles) and
at shapes that take the whole-tile fast path and several k-slices.  Tolerance: split-bf16 operands (hi+lo, ~2^-17 per
product, fp32 accumulate) -> 2e-5 relative to the result's norm.  CPU: emulator build; GPU: product library."""
import ctypes as C

import pytest
import torch

from tests import emu
from vss_cffm_amd import _lib

TOL = 2e-5


def P(t):
    return C.c_void_p(t.data_ptr())


def rel(a, b):
    return float((a.double().cpu() - b).norm() / b.norm())


def run_linear_checks(lib, device, shapes, group_rows):
    gen = torch.Generator().manual_seed(3)
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream) if device.type == 'cuda' else None
    for (M, N, K) in shapes:
        x, w, dy = (torch.randn(M, K, generator=gen), torch.randn(N, K, generator=gen), torch.randn(M, N, generator=gen))
        xd, wd, dyd = x.to(device), w.to(device), dy.to(device)
        y, dx, dw = (torch.full((M, N), 7., device=device), torch.full((M, K), 7., device=device), torch.full((N, K), 7., device=device))
        assert lib.cffm_linear_fwd(P(xd), P(wd), P(y), M, N, K, stream) == 0
        assert lib.cffm_linear_bwd_input(P(dyd), P(wd), P(dx), M, N, K, stream) == 0
        assert lib.cffm_linear_bwd_weight(P(dyd), P(xd), P(dw), M, N, K, stream) == 0
        assert rel(y, x.double() @ w.double().T) < TOL, (M, N, K)
        assert rel(dx, dy.double() @ w.double()) < TOL, (M, N, K)
        assert rel(dw, dy.double().T @ x.double()) < TOL, (M, N, K)
    # grouped weight gradients: the block's four shapes (N, K), two row counts as in the block (q|k|v rows vs target rows)
    class WGrad(C.Structure):
        _fields_ = [('dy', C.c_void_p), ('x', C.c_void_p), ('dw', C.c_void_p), ('M', C.c_long), ('N', C.c_int), ('K', C.c_int)]
    for rows_qkv, rows_tgt in group_rows:
        prob, keep, want = (WGrad * 4)(), [], []
        for i, (M, N, K) in enumerate([(rows_qkv, 768, 256), (rows_tgt, 1024, 256), (rows_tgt, 256, 1024), (rows_tgt, 256, 256)]):
            dy, x = torch.randn(M, N, generator=gen), torch.randn(M, K, generator=gen)
            dyd, xd, dw = dy.to(device), x.to(device), torch.full((N, K), 7., device=device)
            keep += [dyd, xd, dw]
            want.append(dy.double().T @ x.double())
            prob[i] = WGrad(dyd.data_ptr(), xd.data_ptr(), dw.data_ptr(), M, N, K)
        assert lib.cffm_linear_bwd_weight_group(prob, 4, stream) == 0
        for i in range(4):
            assert rel(keep[3 * i + 2], want[i]) < TOL, (rows_qkv, rows_tgt, i)
    # a problem the grouped kernel cannot tile (N not a multiple of 128) takes the one-by-one path, same results
    prob = (WGrad * 1)()
    dy, x = torch.randn(70, 96, generator=gen), torch.randn(70, 40, generator=gen)
    dyd, xd, dw = dy.to(device), x.to(device), torch.zeros(96, 40, device=device)
    prob[0] = WGrad(dyd.data_ptr(), xd.data_ptr(), dw.data_ptr(), 70, 96, 40)
    assert lib.cffm_linear_bwd_weight_group(prob, 1, stream) == 0
    assert rel(dw, dy.double().T @ x.double()) < TOL
    assert lib.cffm_linear_bwd_weight_group(prob, 5, stream) != 0       # more than 4 problems: error, not a crash


def test_linear_emulated():
    with emu.active():
        # small (the emulator runs one fiber per GPU thread): ragged + one whole-tile multi-slice case
        run_linear_checks(emu.lib(), torch.device('cpu'), [(70, 96, 40), (130, 256, 64), (256, 128, 128)], [(200, 136), (1056, 1024)])


@pytest.mark.gpu
def test_linear_gpu():
    lib = _lib.get()
    shapes = [(70, 96, 40), (1000, 256, 256), (7200, 1024, 256), (7200, 256, 1024), (10368, 768, 256), (6480, 256, 256), (3969, 768, 256)]
    run_linear_checks(lib, torch.device('cuda:0'), shapes, [(10368, 7200), (200, 136), (12960, 12960), (3969, 3600)])


# ---- the head's 1x1 classifiers as a GEMM on token rows (ops.conv1x1; cffm_head.py:121,147,524) ----------------------------
def run_conv1x1_checks(device):
    import torch.nn.functional as F
    from vss_cffm_amd import ops
    gen = torch.Generator().manual_seed(4)
    for (n, c, o, h, w, channels_last, strided_dy) in [(2, 40, 124, 5, 7, True, False), (3, 256, 124, 6, 6, False, True),
                                                       (1, 64, 8, 3, 9, True, True), (0, 32, 124, 4, 4, False, False)]:
        x = torch.randn(n, c, h, w, generator=gen)
        wt, b = torch.randn(o, c, 1, 1, generator=gen) * 0.1, torch.randn(o, generator=gen)
        dy = torch.randn(n, o + 3, h, w, generator=gen)[:, 1:o + 1] if strided_dy else torch.randn(n, o, h, w, generator=gen)
        xr, wr, br = x.double().requires_grad_(True), wt.double().requires_grad_(True), b.double().requires_grad_(True)
        F.conv2d(xr, wr, br).backward(dy.double())
        xd = x.to(device)
        if channels_last:
            xd = xd.contiguous(memory_format=torch.channels_last)
        xd = xd.requires_grad_(True)
        wd, bd = wt.to(device).requires_grad_(True), b.to(device).requires_grad_(True)
        y = ops.conv1x1(xd, wd, bd)
        assert y.shape == (n, o, h, w)
        y.backward(dy.to(device))
        if n:
            assert rel(y.detach(), F.conv2d(x.double(), wt.double(), b.double())) < TOL
            assert rel(xd.grad, xr.grad) < TOL and rel(wd.grad, wr.grad) < TOL and rel(bd.grad, br.grad) < TOL
        else:
            assert float(wd.grad.abs().sum()) == 0.0 and float(bd.grad.abs().sum()) == 0.0


    # the per-clip view [B,T,O,h,w] and a gradient that arrives as B separately placed blocks of token rows (slices of a
    # [B,T+1,h,w,O] buffer: what the head's loss kernel hands back) or as one contiguous / an arbitrary strided tensor
    bsz, t, c, o, h, w = 2, 3, 32, 8, 4, 5
    x = torch.randn(bsz * t, c, h, w, generator=gen)
    wt, b = torch.randn(o, c, 1, 1, generator=gen) * 0.1, torch.randn(o, generator=gen)
    for kind in ('blocks', 'dense', 'strided'):
        if kind == 'blocks':
            buf = torch.randn(bsz, t + 1, h, w, o, generator=gen)
            dy5 = buf.to(device)[:, :t].permute(0, 1, 4, 2, 3)
            dyc = buf[:, :t].permute(0, 1, 4, 2, 3)
        elif kind == 'dense':
            dyc = torch.randn(bsz, t, h, w, o, generator=gen).permute(0, 1, 4, 2, 3)
            dy5 = dyc.to(device)
        else:
            dyc = torch.randn(bsz, t, o, h, w + 2, generator=gen)[..., 1:w + 1]
            dy5 = dyc.to(device)
        xr, wr, br = x.detach().double().requires_grad_(True), wt.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
        F.conv2d(xr, wr, br).backward(dyc.double().reshape(bsz * t, o, h, w))
        xd = x.to(device).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        wd, bd = wt.clone().to(device).requires_grad_(True), b.clone().to(device).requires_grad_(True)
        y = ops.conv1x1(xd, wd, bd, clips=bsz)
        assert y.shape == (bsz, t, o, h, w) and y.permute(0, 1, 3, 4, 2).is_contiguous()
        assert rel(y.detach().reshape(bsz * t, o, h, w), F.conv2d(x.double(), wt.double(), b.double())) < TOL
        y.backward(dy5)
        assert rel(xd.grad, xr.grad) < TOL and rel(wd.grad, wr.grad) < TOL and rel(bd.grad, br.grad) < TOL, kind
    with pytest.raises(_lib.CffmError):
        ops.conv1x1(torch.zeros(3, 32, 2, 2, device=device), torch.zeros(8, 32, 1, 1, device=device), torch.zeros(8, device=device), clips=2)
    with pytest.raises(_lib.CffmError):           # 19 classes: rows of 76 bytes -- the head keeps nn.Conv2d for such sizes
        ops.conv1x1(torch.zeros(1, 32, 2, 2, device=device), torch.zeros(19, 32, 1, 1, device=device), torch.zeros(19, device=device))


def test_conv1x1_emulated():
    with emu.active():
        run_conv1x1_checks(torch.device('cpu'))


@pytest.mark.gpu
def test_conv1x1_gpu():
    run_conv1x1_checks(torch.device('cuda:0'))


def run_dw_split_checks(lib, device, shapes):
    """cffm_linear_bwd_weight_split (the LDS-DMA weight-gradient kernel, csrc/dw_kernels.h): operands in split-4 storage made by
    cffm_split4, against fp64; ragged contraction lengths (last K-tile partly past the array: the DMA's bounds check supplies zeros),
    one K-tile, fewer K-tiles than LDS stages, several k-slices."""
    gen = torch.Generator().manual_seed(9)
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream) if device.type == 'cuda' else None
    for (M, N, K) in shapes:
        dy, x = torch.randn(M, N, generator=gen), torch.randn(M, K, generator=gen)
        dyd, xd = dy.to(device), x.to(device)
        dys, xs, dw = torch.empty_like(dyd), torch.empty_like(xd), torch.full((N, K), 7., device=device)
        assert lib.cffm_split4(P(dyd), P(dys), M * N, stream) == 0 and lib.cffm_split4(P(xd), P(xs), M * K, stream) == 0
        assert lib.cffm_linear_bwd_weight_split(P(dys), P(xs), P(dw), M, N, K, stream) == 0, lib.cffm_last_error()
        assert rel(dw, dy.double().T @ x.double()) < TOL, (M, N, K)
    assert lib.cffm_linear_bwd_weight_split(P(dys), P(xs), P(dw), 10, 96, 128, stream) != 0      # N not a multiple of 128: refused


def test_dw_split_emulated():
    run_dw_split_checks(emu.lib(), torch.device('cpu'), [(7, 128, 128), (32, 128, 256), (100, 256, 128), (333, 128, 128)])


@pytest.mark.gpu
def test_dw_split_gpu():
    run_dw_split_checks(_lib.get(), torch.device('cuda'), [(7, 128, 128), (32, 128, 256), (100, 256, 128), (1000, 128, 128), (10368, 768, 256), (7200, 1024, 256),
                                                           (7200, 256, 1024), (7201, 256, 256)])


def run_dw_tfrag_checks(lib, device, shapes, group=None):
    """cffm_linear_bwd_weight_tfrag (the streaming weight-gradient kernel, csrc/dws_kernels.h): operands in T-frag storage made by
    cffm_tfrag_pack, against fp64; contraction lengths that are not multiples of 32 (the last k-step zero-padded), fewer k-steps than waves
    (waves with nothing to do), several k-slices (slabs), N = 64; then several problems as one group."""
    gen = torch.Generator().manual_seed(11)
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream) if device.type == 'cuda' else None
    lib.cffm_tfrag_floats.restype = C.c_long
    made = []
    for (M, N, K) in shapes:
        dy, x = torch.randn(M, N, generator=gen), torch.randn(M, K, generator=gen)
        dyd, xd = dy.to(device), x.to(device)
        dyt = torch.full((lib.cffm_tfrag_floats(M, N),), float('nan'), device=device)
        xt = torch.full((lib.cffm_tfrag_floats(M, K),), float('nan'), device=device)
        dw = torch.full((N, K), 7., device=device)
        assert lib.cffm_tfrag_pack(P(dyd), P(dyt), M, N, stream) == 0 and lib.cffm_tfrag_pack(P(xd), P(xt), M, K, stream) == 0
        assert lib.cffm_linear_bwd_weight_tfrag(P(dyt), P(xt), P(dw), M, N, K, stream) == 0, lib.cffm_last_error()
        ref = dy.double().T @ x.double()
        assert rel(dw, ref) < TOL, (M, N, K)
        made.append((dyt, xt, M, N, K, ref))
    assert lib.cffm_linear_bwd_weight_tfrag(P(dyt), P(xt), P(dw), 10, 96, 128, stream) != 0      # N not a multiple of 64: refused
    assert lib.cffm_linear_bwd_weight_tfrag(P(dyt), P(xt), P(dw), 10, 128, 64, stream) != 0      # K not a multiple of 128: refused
    if group:
        class WGrad(C.Structure):
            _fields_ = [('dy', C.c_void_p), ('x', C.c_void_p), ('dw', C.c_void_p), ('M', C.c_long), ('N', C.c_int), ('K', C.c_int)]
        sel = [made[i] for i in group]
        outs = [torch.full((m[3], m[4]), 3., device=device) for m in sel]
        prob = (WGrad * len(sel))(*[WGrad(m[0].data_ptr(), m[1].data_ptr(), o.data_ptr(), m[2], m[3], m[4]) for m, o in zip(sel, outs)])
        assert lib.cffm_linear_bwd_weight_tfrag_group(prob, len(sel), stream) == 0, lib.cffm_last_error()
        again = [o.clone() for o in outs]
        assert lib.cffm_linear_bwd_weight_tfrag_group(prob, len(sel), stream) == 0
        for m, o, a in zip(sel, outs, again):
            assert rel(o, m[5]) < TOL and torch.equal(o, a), m[2:5]       # (and bit-identical from call to call: fixed summation order)


def test_dw_tfrag_emulated():
    run_dw_tfrag_checks(emu.lib(), torch.device('cpu'), [(7, 64, 128), (70, 128, 128), (333, 64, 256), (600, 128, 128)], group=(1, 2, 3))


@pytest.mark.gpu
def test_dw_tfrag_gpu():
    run_dw_tfrag_checks(_lib.get(), torch.device('cuda'), [(7, 64, 128), (100, 256, 128), (1000, 128, 128), (7201, 256, 256), (10368, 768, 256),
                                                           (7200, 1024, 256), (7200, 256, 1024), (7200, 256, 256)], group=(4, 5, 6, 7))


def run_gtc_attn_stage_checks(lib, device, cases, tol=2e-5):
    """cffm_gtc_attn_fwd / _bwd (WindowAttention_cluster, pvt/swin_transformer_2d.py:232-257, with only_use_cluster_center_as_context) as stages
    against torch in fp64: every template instance of the matrix-pipe kernels (K <= 32, 64, 96, 128: one to four k-steps of 32 keys, keys
    padded with -inf), the VALU form above 128, token counts that leave the last 16-token tile / the last pair of tiles / a wave's run ragged."""
    gen = torch.Generator().manual_seed(31)
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream) if device.type == 'cuda' else None
    for (B, T, K) in cases:
        q_raw, kv_raw = torch.randn(B * T, 256, generator=gen), torch.randn(B * K, 512, generator=gen)
        bq, bkv = torch.randn(256, generator=gen) * 0.3, torch.randn(512, generator=gen) * 0.3
        dout = torch.randn(B * T, 256, generator=gen)
        # reference
        qd, kvd = q_raw.double().requires_grad_(True), kv_raw.double().requires_grad_(True)
        q = ((qd + bq.double()) * 32 ** -0.5).view(B, T, 8, 32).permute(0, 2, 1, 3)
        kv = (kvd + bkv.double()).view(B, K, 2, 8, 32).permute(2, 0, 3, 1, 4)
        s = q @ kv[0].transpose(-1, -2)
        o_ref = (s.softmax(-1) @ kv[1]).permute(0, 2, 1, 3).reshape(B * T, 256)
        lse_ref = s.logsumexp(-1).permute(0, 2, 1).reshape(B * T, 8)
        o_ref.backward(dout.double())
        d = lambda t: t.to(device)
        o, lse = torch.full((B * T, 256), float('nan'), device=device), torch.full((B * T, 8), float('nan'), device=device)
        assert lib.cffm_gtc_attn_fwd(P(d(q_raw)), P(d(bq)), P(d(kv_raw)), P(d(bkv)), P(o), P(lse), B, T, K, stream) == 0, lib.cffm_last_error()
        assert rel(o, o_ref.detach()) < tol and rel(lse, lse_ref.detach()) < tol, (B, T, K)
        dq, dkv = torch.full((B * T, 256), float('nan'), device=device), torch.full((B * K, 512), float('nan'), device=device)
        qr, kvr = d(q_raw), d(kv_raw)
        assert lib.cffm_gtc_attn_bwd(P(qr), P(d(bq)), P(kvr), P(d(bkv)), P(o), P(d(dout)), P(lse), P(dq), P(dkv), B, T, K, stream) == 0, lib.cffm_last_error()
        if K == 1:      # softmax over one key is the constant 1: nothing flows to q (exact zeros in the reference)
            assert float(qd.grad.abs().max()) == 0.0 and float(dq.abs().max()) < 1e-5 and rel(dkv, kvd.grad) < tol, (B, T, K)
        else:
            assert rel(dq, qd.grad) < tol and rel(dkv, kvd.grad) < tol, (B, T, K, rel(dq, qd.grad), rel(dkv, kvd.grad))


def test_gtc_attn_stages_emulated():
    run_gtc_attn_stage_checks(emu.lib(), torch.device('cpu'), [(1, 35, 8), (2, 17, 33), (1, 70, 64), (1, 33, 65), (1, 16, 96), (1, 20, 128), (1, 40, 130), (1, 3, 1)])


@pytest.mark.gpu
def test_gtc_attn_stages_gpu():
    run_gtc_attn_stage_checks(_lib.get(), torch.device('cuda'), [(1, 35, 8), (2, 17, 33), (2, 700, 64), (1, 333, 65), (2, 3600, 96), (1, 1000, 128), (1, 100, 130),
                                                                 (2, 3600, 100), (1, 1, 1), (1, 15, 32)])
