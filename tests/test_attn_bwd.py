"""The attention backward on its own (cffm_attn_bwd: the fused kernel, the bias-gradient tile sum and the dK / dV gather of
csrc/cfm_attn_kernels.h) against torch autograd of the same windowed cross-attention on the same f16 q|k|v rows: dq of every query,
dk / dv of every token and pooled row, the zeroed q third of the pooled rows and the dense bias gradient.
Reference semantics: cffm_transformer.py:364-606; the key assembly comes from geometry.tables, which
tests/test_geometry.py pins against the oracle's roll / unfold maps."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import emu
from vss_cffm_amd import _lib, geometry, ops


def P(t):
    return C.c_void_p(t.data_ptr())


def bias_buffer(bias):
    """[8, 64, 304] fp32 -> the f16 MFMA fragments cffm_bias_assemble builds"""
    bh = bias.half()
    b320 = torch.zeros(8, 64, 320, dtype=torch.float16)
    b320[:, :, :304] = bh
    # (h, wave, j, pair, g2 (tile of the pair, 8-key half), e) -> (h, wave, pair, g2, j, e): lane 16 g2 + j of pair p
    return b320.view(8, 4, 16, 10, 4, 8).permute(0, 1, 3, 4, 2, 5).contiguous().reshape(-1)


def run_attn_bwd_stage(lib, device, b, h0, w0, grad_scale=1e-3, zero_window=False, raw=False):
    torch.manual_seed(b * 1000 + h0 * 31 + w0)
    g = ops.make_geom(lib, b, h0, w0)
    nw, rc, hw = g.nW, g.RC, g.HW
    ks_t, qd_t, ip_t, ii_t = ops.device_tables(h0, w0, device)
    ks_c, qd_c = ks_t.cpu(), qd_t.cpu()
    qkv = (torch.randn(b * rc, 768) * 0.7).half()
    bias = torch.randn(8, 64, 304) * 0.5
    bias[:, 49:] = 0
    bias[:, :, 289:] = 0
    dao = torch.randn(b * hw, 256) * grad_scale
    dao[: hw // 3] *= 1e-3                       # windows whose gradients are three decades smaller (the common-scale staging)
    if zero_window:
        dao[:] = 0
    # ---- reference: torch autograd on the float values of the same f16 rows
    x = qkv.float().clone().requires_grad_(True)
    xq = x.view(b, rc, 768)
    q = xq[:, :49 * nw, :256].reshape(b, nw, 49, 8, 32).permute(0, 1, 3, 2, 4)
    idx = ks_c[:, :289].long().clamp_min(0)
    valid = ks_c[:, :289] >= 0
    gat = lambda t: (t[:, idx.view(-1)].view(b, nw, 289, 8, 32) * valid.view(1, nw, 289, 1, 1)).permute(0, 1, 3, 2, 4)
    k, v = gat(xq[:, :, 256:512]), gat(xq[:, :, 512:])
    s = q @ k.transpose(-1, -2) + bias.half().float()[None, None, :, :49, :289]
    s = s.masked_fill(~valid.view(1, nw, 1, 1, 289), float('-inf'))
    s.retain_grad()
    lse_ref = torch.logsumexp(s, -1)
    o_rows = (torch.softmax(s, -1) @ v).permute(0, 1, 3, 2, 4).reshape(b, nw * 49, 256)
    qdf = qd_c.view(-1)
    sel = torch.nonzero(qdf >= 0).view(-1)
    (o_rows[:, sel] * dao.view(b, hw, 256)[:, qdf[sel].long()]).sum().backward()
    ao = torch.zeros(b, hw, 256)
    ao[:, qdf[sel].long()] = o_rows[:, sel].detach()
    dref = x.grad.view(b, rc, 768)
    dbias_ref = s.grad.sum((0, 1))
    # ---- library
    lse = torch.zeros(b, nw, 8, 64)
    lse[..., :49] = lse_ref.detach()
    dev = lambda t: t.contiguous().to(device)
    dqkv = torch.full((b * rc, 768), float('nan'), device=device)
    dbias_t = torch.zeros(8, 304, 64, device=device)
    ws = torch.zeros(b * nw * (304 * 256 + 8), device=device)      # f16 partial rows + their scales
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream) if device.type == 'cuda' else None
    args = [dev(t) for t in (qkv, bias_buffer(bias), ao.view(b * hw, 256), dao, lse)]
    _lib.check(lib.cffm_attn_bwd(C.byref(g), P(args[0]), P(ks_t), P(qd_t), P(ip_t), P(ii_t), P(args[1]), P(args[2]), P(args[3]),
                                 P(args[4]), P(dqkv), P(dbias_t), P(ws), stream), lib)
    if raw:
        return dqkv, dbias_t
    got = dqkv.cpu().view(b, rc, 768)
    assert not torch.isnan(got).any()                     # every row of dq | dk | dv is written
    assert float(got[:, 49 * nw:, :256].abs().max()) == 0  # pooled rows have no query
    if zero_window:
        assert float(got.abs().max()) == 0 and float(dbias_t.abs().max()) == 0
        return {}
    rel = lambda a, r: float((a - r).abs().max() / r.abs().max())
    errs = {'dq': rel(got[:, :49 * nw, :256], dref[:, :49 * nw, :256] * 32 ** -0.5),   # d(raw q): the stored q carries 32^-0.5
            'dk': rel(got[:, :, 256:512], dref[:, :, 256:512]), 'dv': rel(got[:, :, 512:], dref[:, :, 512:]),
            'dbias': rel(dbias_t.cpu().transpose(1, 2)[:, :49, :289], dbias_ref)}
    for name, lo, hi in (('tokens', 0, 49 * nw), ('P0', 49 * nw, 50 * nw), ('f0', 50 * nw, 51 * nw), ('f1', 51 * nw, 55 * nw),
                         ('f2', 55 * nw, 64 * nw)):
        errs['dk_' + name] = rel(got[:, lo:hi, 256:512], dref[:, lo:hi, 256:512])
        errs['dv_' + name] = rel(got[:, lo:hi, 512:], dref[:, lo:hi, 512:])
    return errs


STAGE_TOL = 8e-4    # f16 operands (q, k, v exact here; dO, P, dS rounded)
RANGE_TOL = 1.5e-3  # the same per row range, relative to the range's own maximum (a range can hold only small gradients)


def check(errs):
    worst = max(v for k, v in errs.items() if '_' not in k)
    assert worst < STAGE_TOL, errs
    assert max(errs.values()) < RANGE_TOL, errs


# (7, 7): one window that is its own cyclic neighbour eight times over (up to 4 readings of a key by one window); (5, 20): one row of
# windows; (8, 8): 2 x 2 windows (every neighbour direction wraps); (14, 21): no padding, non-square; (13, 30): ragged
# (window groups of the launch: 1 window per group up to 32 windows, then 2, 3, ...: (1, 42, 42) = 36 windows walks 2 per workgroup --
# the shortest run of the kernel's window pipeline that has a 'next' window)
@pytest.mark.parametrize('shape', [(1, 7, 7), (1, 5, 20), (1, 8, 8), (2, 14, 21), (2, 13, 30), (1, 22, 23), (1, 42, 42)])   # noqa: E501
def test_attn_bwd_stage_emulated(shape):
    errs = run_attn_bwd_stage(emu.lib(), torch.device('cpu'), *shape)
    check(errs)


def test_attn_bwd_stage_all_zero_gradient_emulated():
    run_attn_bwd_stage(emu.lib(), torch.device('cpu'), 1, 8, 8, zero_window=True)


def test_attn_bwd_stage_tiny_gradients_emulated():
    """training-size gradients (1e-7) survive the f16 operands: per-(window, head) power-of-two scales"""
    errs = run_attn_bwd_stage(emu.lib(), torch.device('cpu'), 1, 8, 8, grad_scale=1e-7)
    check(errs)


@pytest.mark.gpu
# windows per workgroup: 1, 1, 1, 1, 2, 3, 6, 7, 5 (+ a short last group), 11
@pytest.mark.parametrize('shape', [(1, 7, 7), (2, 8, 8), (2, 14, 21), (2, 13, 30), (1, 42, 42), (1, 56, 63), (2, 60, 60), (2, 64, 64), (1, 60, 108), (4, 60, 60)])
def test_attn_bwd_stage_gpu(shape):
    errs = run_attn_bwd_stage(_lib.get(), torch.device('cuda'), *shape)
    print('attn_bwd stage', shape, {k: '%.2e' % v for k, v in errs.items() if '_' not in k})
    check(errs)


@pytest.mark.gpu
def test_attn_bwd_stage_is_deterministic_gpu():
    a = run_attn_bwd_stage(_lib.get(), torch.device('cuda'), 2, 60, 60, raw=True)
    b = run_attn_bwd_stage(_lib.get(), torch.device('cuda'), 2, 60, 60, raw=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
