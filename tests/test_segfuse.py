"""SegFormer embedding in front of the hot path (cffm_head.py:102-119) without the 1024-channel concat:
ops.segformer_fuse / cffm_segfuse_fwd / cffm_segfuse_bwd against
  * golden vectors the REFERENCE head's own sub-modules produced (tests/golden/make_golden_fuse.py -> fuse_b0.npz),
  * the CPU restatement oracle/cffm_oracle.py::segformer_fuse (itself pinned to those vectors and to F.interpolate here),
  * size-independent properties at the CFFM-B1 480x480 sizes (adjoint identity, determinism) on the GPU.
Tolerance: the Linear GEMMs use split-bf16 operands (hi+lo, ~2^-17 per product, fp32 accumulate) -> 5e-5 max-abs relative
to the tensor's max (measured ~5e-6); the north-star contract is 1e-3."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cffm_oracle as O, recipe as R, ref_import as RI
from tests import emu, helpers as H
from tests.golden.make_golden_fuse import CASES, PARAMS, case_inputs
from vss_cffm_amd import _lib, ops
from vss_cffm_amd.registry import build_head

TOL = 5e-5
B1 = (64, 128, 320, 512)


def _params(seed=50, chans=(32, 64, 160, 256)):
    head = build_head(RI.head_cfg(in_channels=chans))
    head.load_state_dict(R.synth_state(head, seed=seed), strict=False)
    sd = dict(head.named_parameters())
    return [sd[k].detach().clone() for k in PARAMS]


def _split(ps):
    return [ps[0], ps[2], ps[4], ps[6]], [ps[1], ps[3], ps[5], ps[7]], ps[8]


def _run(fn, feats, ps, gy, device):
    fg = [f.to(device).clone().requires_grad_(True) for f in feats]
    pg = [p.to(device).clone().requires_grad_(True) for p in ps]
    lw, lb, fw = _split(pg)
    y = fn(fg, lw, lb, fw)
    y.backward(gy.to(device))
    return y.detach().cpu(), [f.grad.cpu() for f in fg], [p.grad.cpu() for p in pg]


def _check_against_golden(fn, device, tol):
    g = H.load_golden('fuse_b0')
    ps = _params()
    for name in CASES:
        feats, gy = case_inputs(name)
        y, df, dp = _run(fn, feats, ps, gy, device)
        assert y.shape == g[name + '/y'].shape
        assert H.rel_err(y, g[name + '/y']) < tol, name
        for i in range(4):
            assert H.rel_err(df[i], g['%s/dfeat%d' % (name, i)]) < tol, (name, i)
        for k, d in zip(PARAMS, dp):
            assert H.rel_err(d, g['%s/d.%s' % (name, k)]) < tol, (name, k)


# ------------------------------------------------------------------------------------------------ oracle (CPU)
@pytest.mark.parametrize('n_in,n_out', [(8, 16), (4, 16), (2, 16), (7, 13), (15, 30), (4, 30), (2, 13), (15, 120), (5, 5), (1, 9)])
def test_oracle_bilinear_matrix_is_interpolate(n_in, n_out):
    x = torch.randn(2, 3, n_in, 5, generator=torch.Generator().manual_seed(n_in * 100 + n_out))
    want = F.interpolate(x, size=(n_out, 5), mode='bilinear', align_corners=False)
    got = torch.einsum('yh,nchw->ncyw', O.bilinear_matrix(n_in, n_out), x)
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-6)
    assert torch.allclose(O.bilinear_matrix(n_in, n_out).sum(1), torch.ones(n_out), atol=1e-6)   # why the biases fold into a constant


def test_oracle_fuse_against_reference_golden():
    _check_against_golden(O.segformer_fuse, torch.device('cpu'), 2e-6)


@pytest.mark.skipif(not RI.available(), reason='/root/reference not present')
def test_oracle_fuse_against_reference_live():
    from tests.golden.make_golden_fuse import reference_fuse
    head = RI.build_reference_head()
    head.load_state_dict(R.synth_state(head, seed=53), strict=False)
    sd = dict(head.named_parameters())
    ps = [sd[k].detach() for k in PARAMS]
    feats = [R.synth_input('live_c%d' % i, (3, c, h, w), seed=54, scale=1.0)
             for i, (c, (h, w)) in enumerate(zip((32, 64, 160, 256), [(15, 20), (8, 10), (4, 5), (2, 3)]))]
    with torch.no_grad():
        want = reference_fuse(head, feats)
        got = O.segformer_fuse(feats, *_split(ps))
    assert H.rel_err(got, want) < 2e-6


# ------------------------------------------------------------------------------------------------ kernels, emulated (CPU)
def test_segfuse_emulated_against_reference_golden():
    with emu.active():
        _check_against_golden(ops.segformer_fuse, torch.device('cpu'), TOL)


def run_edge_cases(device):
    gen = torch.Generator().manual_seed(5)
    rn = lambda *s: torch.randn(*s, generator=gen)
    # fewer scales than four (the kernel takes 0..3 resized maps), a feature that needs no gradient, an empty batch
    for k, sizes in ((1, [(6, 9)]), (2, [(6, 9), (3, 5)]), (3, [(9, 6), (5, 3), (1, 1)])):
        chans = (24, 40, 64)[:k]
        feats = [rn(2, c, *s) for c, s in zip(chans, sizes)]
        lw, lb, fw = [rn(256, c) * 0.1 for c in chans], [rn(256) for _ in chans], rn(256, 256 * k, 1, 1) * 0.05
        gy = rn(2, 256, *sizes[0])
        # fp64 reference with the oracle's matrices (any number of scales)
        fd = [f.double().requires_grad_(True) for f in feats]
        wd = [t.double().requires_grad_(True) for t in lw + lb + [fw]]
        maps = []
        for i in reversed(range(k)):
            m = F.linear(fd[i].flatten(2).transpose(1, 2), wd[i], wd[k + i]).permute(0, 2, 1).reshape(2, 256, *sizes[i])
            if i:
                m = torch.einsum('yh,nchw,xw->ncyx', O.bilinear_matrix(sizes[i][0], sizes[0][0], torch.float64), m,
                                 O.bilinear_matrix(sizes[i][1], sizes[0][1], torch.float64))
            maps.append(m)
        yw = torch.einsum('oc,nchw->nohw', wd[-1].reshape(256, -1), torch.cat(maps, 1))
        yw.backward(gy.double())
        fg = [f.to(device).clone().requires_grad_(i != 0) for i, f in enumerate(feats)]     # c1 without gradient
        pg = [t.to(device).clone().requires_grad_(True) for t in lw + lb + [fw]]
        y = ops.segformer_fuse(fg, pg[:k], pg[k:2 * k], pg[-1])
        y.backward(gy.to(device))
        assert H.rel_err(y, yw) < TOL, k
        assert fg[0].grad is None
        for i in range(1, k):
            assert H.rel_err(fg[i].grad, fd[i].grad) < TOL, (k, i)
        for a, b in zip(pg, wd):
            assert H.rel_err(a.grad, b.grad) < TOL, k
    empty = [torch.zeros(0, c, *s, device=device) for c, s in zip((24, 40), [(6, 9), (3, 5)])]
    y = ops.segformer_fuse(empty, [rn(256, 24).to(device), rn(256, 40).to(device)], [rn(256).to(device)] * 2,
                           rn(256, 512, 1, 1).to(device))
    assert y.shape == (0, 256, 6, 9)
    # error behaviour: resize factors above 16, operands that do not fit
    with pytest.raises(_lib.CffmError):
        ops.segformer_fuse([rn(1, 8, 40, 40).to(device), rn(1, 8, 2, 2).to(device)], [rn(256, 8).to(device)] * 2,
                           [rn(256).to(device)] * 2, rn(256, 512, 1, 1).to(device))
    with pytest.raises(_lib.CffmError):
        ops.segformer_fuse([rn(1, 8, 4, 4).to(device)], [rn(256, 9).to(device)], [rn(256).to(device)], rn(256, 256, 1, 1).to(device))
    with pytest.raises(_lib.CffmError):
        ops.segformer_fuse([rn(1, 8, 4, 4).to(device)], [rn(256, 8).to(device)], [rn(256).to(device)], rn(256, 512, 1, 1).to(device))


def test_segfuse_edge_cases_emulated():
    with emu.active():
        run_edge_cases(torch.device('cpu'))


def run_adjoint_identity(lib, device, n, H_, W_, sizes):
    """<resize(z), g> == <z, resize^T g> for the kernel pair, through the C ABI (y starts at zero, d = 0)."""
    gen = torch.Generator().manual_seed(9)
    zs = [torch.randn(n * h * w, 256, generator=gen).to(device) for h, w in sizes]
    g = torch.randn(n * H_ * W_, 256, generator=gen).to(device)
    y = torch.zeros(n * H_ * W_, 256, device=device)
    d = torch.zeros(256, device=device)
    dz = [torch.full_like(z, 7.) for z in zs]
    k = len(sizes)
    hs = (C.c_int * 3)(*([h for h, _ in sizes] + [1] * (3 - k)))
    ws = (C.c_int * 3)(*([w for _, w in sizes] + [1] * (3 - k)))
    st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream) if device.type == 'cuda' else None
    zp = (C.c_void_p * 3)(*([z.data_ptr() for z in zs] + [None] * (3 - k)))
    dzp = (C.c_void_p * 3)(*([z.data_ptr() for z in dz] + [None] * (3 - k)))
    assert lib.cffm_segfuse_fwd(C.c_void_p(y.data_ptr()), C.c_void_p(d.data_ptr()), zp, hs, ws, k, n, H_, W_, st) == 0
    assert lib.cffm_segfuse_bwd(C.c_void_p(g.data_ptr()), dzp, hs, ws, k, n, H_, W_, st) == 0
    lhs = float((y.double() * g.double()).sum())
    rhs = float(sum((z.double() * q.double()).sum() for z, q in zip(zs, dz)))
    scale = float(sum((z.double() * q.double()).abs().sum() for z, q in zip(zs, dz)))
    assert abs(lhs - rhs) < 1e-5 * scale, (lhs, rhs, scale)
    return y, dz


def test_segfuse_adjoint_identity_emulated():
    with emu.active():
        run_adjoint_identity(emu.lib(), torch.device('cpu'), 2, 13, 30, [(7, 15), (4, 8), (2, 4)])


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_segfuse_gpu_against_reference_golden():
    _check_against_golden(ops.segformer_fuse, torch.device('cuda:0'), TOL)


@pytest.mark.gpu
def test_segfuse_gpu_edge_cases():
    run_edge_cases(torch.device('cuda:0'))


@pytest.mark.gpu
def test_segfuse_gpu_full_size_properties():
    """CFFM-B1 at 480x480, 2 clips x 4 frames: adjoint identity and determinism of the kernel pair; the operator against the
    reference's op sequence in torch fp64 on the same device; the [N,1024,120,120] concat is never allocated."""
    dev = torch.device('cuda:0')
    lib = _lib.get()
    sizes = [(120, 120), (60, 60), (30, 30), (15, 15)]
    y1, dz1 = run_adjoint_identity(lib, dev, 8, 120, 120, sizes[1:])
    y2, dz2 = run_adjoint_identity(lib, dev, 8, 120, 120, sizes[1:])
    assert torch.equal(y1, y2) and all(torch.equal(a, b) for a, b in zip(dz1, dz2))
    del y1, y2, dz1, dz2
    ps = _params(seed=55, chans=B1)
    feats = [R.synth_input('full_c%d' % i, (8, c, h, w), seed=56, scale=1.0) for i, (c, (h, w)) in enumerate(zip(B1, sizes))]
    gy = R.synth_input('full_gy', (8, 256, 120, 120), seed=57, scale=1.0)
    fg = [f.to(dev).requires_grad_(True) for f in feats]
    pg = [p.to(dev).requires_grad_(True) for p in ps]
    gd = gy.to(dev).contiguous(memory_format=torch.channels_last)   # what BatchNorm hands back for a channels-last input
    torch.cuda.synchronize(dev)
    torch.cuda.reset_peak_memory_stats(dev)
    base = torch.cuda.memory_allocated(dev)
    lw, lb, fw = _split(pg)
    yd = ops.segformer_fuse(fg, lw, lb, fw)
    yd.backward(gd)
    torch.cuda.synchronize(dev)
    peak = torch.cuda.max_memory_allocated(dev) - base
    assert peak < 8 * 1024 * 120 * 120 * 4, peak            # below the size of the concat alone (472 MB); measured 233 MB
    y, df, dp = yd.detach().cpu(), [f.grad.cpu() for f in fg], [p.grad.cpu() for p in pg]
    del yd, fg, pg, gd

    def ref64(f, w, b, u):
        maps = []
        for i in (3, 2, 1, 0):
            m = F.linear(f[i].flatten(2).transpose(1, 2), w[i], b[i]).permute(0, 2, 1).reshape(8, 256, *sizes[i])
            maps.append(m if i == 0 else F.interpolate(m, size=sizes[0], mode='bilinear', align_corners=False))
        return F.conv2d(torch.cat(maps, 1), u)
    yw, dfw, dpw = _run(ref64, [f.double() for f in feats], [p.double() for p in ps], gy.double(), dev)
    assert H.rel_err(y, yw) < TOL
    for i in range(4):
        assert H.rel_err(df[i], dfw[i]) < TOL, i
    for k, a, b in zip(PARAMS, dp, dpw):
        assert H.rel_err(a, b) < TOL, k
