"""The fused row-panel stages of the C ABI (cffm_mlp_fwd / cffm_mlp_bwd, cffm_panel_pack_weight; csrc/panel_kernels.h) against
fp64 torch autograd of the reference's op sequence (cffm_transformer.py:602 proj, :823-824 residual / norm2 / Mlp / residual):
every output, every saved intermediate and every bias / norm gradient, at ragged row counts (rows that are not a multiple of
the 32-row panel, panels that straddle two clips).  Tolerance as for the tiled GEMMs: split-bf16 operands (hi + lo, ~2^-17 per
product) -> 2e-5 of the result's norm.  Also: the LDS swizzle of the panel image is conflict-free for the ds_read_b128 lane
groups of gfx950.  CPU: emulator build; GPU: product library at the CFFM-B1 size."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from tests import emu
from vss_cffm_amd import _lib

TOL = 2e-5


def P(t):
    return C.c_void_p(t.data_ptr())


def rel(a, b):
    b = b.double()
    return float((a.double().cpu() - b.cpu()).norm() / b.norm().clamp_min(1e-30))


def unsplit4(t):
    """split-4 storage ({bf16 hi x4, bf16 lo x4} per 4 floats) -> float64 values."""
    raw = t.detach().cpu().contiguous().view(torch.int16).reshape(-1, 8).to(torch.int32)
    hi = (raw[:, :4] << 16).view(torch.float32).double()
    lo = (raw[:, 4:] << 16).view(torch.float32).double()
    return (hi + lo).reshape(t.shape)


def run_mlp_checks(lib, device, cases):
    gen = torch.Generator().manual_seed(11)
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream) if device.type == 'cuda' else None
    rn = lambda *s, scale=1.0: torch.randn(*s, generator=gen) * scale
    for (B, HW) in cases:
        NP = B * HW
        wp, w1, w2 = rn(256, 256, scale=0.08), rn(1024, 256, scale=0.08), rn(256, 1024, scale=0.05)
        bp, b1, b2, g2, be2 = rn(256, scale=0.1), rn(1024, scale=0.1), rn(256, scale=0.1), 1 + rn(256, scale=0.2), rn(256, scale=0.1)
        ao, stack, dout = rn(NP, 256), rn(B, 4, HW, 256, scale=1.5), rn(NP, 256)
        # fp64 reference through autograd
        d = lambda t: t.double().requires_grad_(True)
        ao_r, xt_r, wp_r, w1_r, w2_r, bp_r, b1_r, b2_r, g2_r, be2_r = map(d, (ao, stack[:, 3].reshape(NP, 256), wp, w1, w2, bp, b1, b2, g2, be2))
        x1_r = xt_r + ao_r @ wp_r.T + bp_r
        z2_r = F.layer_norm(x1_r, (256,), g2_r, be2_r, 1e-5)
        h_r = z2_r @ w1_r.T
        h_r.retain_grad(); x1_r.retain_grad()
        act_r = F.gelu(h_r + b1_r)
        x2_r = x1_r + act_r @ w2_r.T + b2_r
        x2_r.backward(dout.double())
        # device side
        dev = lambda t: t.to(device).contiguous()
        frag = {}
        for name, w in (('wp', wp), ('w1', w1), ('w2', w2)):
            for form in (0, 1):
                out = torch.empty(w.numel(), device=device)
                _lib.check(lib.cffm_panel_pack_weight(P(dev(w)), w.shape[0], w.shape[1], form, P(out), stream), lib)
                frag[name, form] = out
        stack_d, ao_d, dout_d = dev(stack), dev(ao), dev(dout)
        xt_d = stack_d[:, 3]
        pd = {k: dev(v) for k, v in dict(bp=bp, b1=b1, b2=b2, g2=g2, be2=be2).items()}
        mk = lambda *s: torch.full(s, 7., device=device)
        x1, z2s, mean2, rstd2, hraw, acts, x2 = mk(NP, 256), mk(NP, 256), mk(NP), mk(NP), mk(NP, 1024), mk(NP, 1024), mk(NP, 256)
        _lib.check(lib.cffm_mlp_fwd(P(ao_d), P(xt_d), 4 * HW * 256, HW, P(frag['wp', 0]), P(frag['w1', 0]), P(frag['w2', 0]), P(pd['bp']),
                                    P(pd['b1']), P(pd['b2']), P(pd['g2']), P(pd['be2']), P(x1), P(z2s), P(mean2), P(rstd2), P(hraw), P(acts),
                                    P(x2), NP, stream), lib)
        assert rel(x1, x1_r.detach()) < TOL and rel(x2, x2_r.detach()) < TOL, (B, HW)
        assert rel(hraw, h_r.detach()) < TOL and rel(unsplit4(acts), act_r.detach()) < TOL and rel(unsplit4(z2s), z2_r.detach()) < TOL
        mu = x1_r.detach().mean(1)
        assert rel(mean2, mu) < 1e-5 and rel(rstd2, 1 / torch.sqrt(x1_r.detach().var(1, unbiased=False) + 1e-5)) < 1e-5
        dhs, dx1, dao = mk(NP, 1024), mk(NP, 256), mk(NP, 256)
        dg2, dbe2, db1, db2, dbp = mk(256), mk(256), mk(1024), mk(256), mk(256)
        _lib.check(lib.cffm_mlp_bwd(P(dout_d), P(hraw), P(pd['b1']), P(x1), P(mean2), P(rstd2), P(pd['g2']), P(frag['w2', 1]), P(frag['w1', 1]),
                                    P(frag['wp', 1]), P(dhs), P(dx1), P(dao), P(dg2), P(dbe2), P(db1), P(db2), P(dbp), NP, stream), lib)
        assert rel(unsplit4(dhs), h_r.grad) < TOL and rel(dx1, x1_r.grad) < TOL and rel(dao, ao_r.grad) < TOL, (B, HW)
        assert rel(dg2, g2_r.grad) < TOL and rel(dbe2, be2_r.grad) < TOL and rel(db1, b1_r.grad) < TOL
        assert rel(db2, b2_r.grad) < TOL and rel(dbp, bp_r.grad) < TOL
        # repeats are bit-identical (fixed summation order everywhere); the activation is optional (NULL: not stored)
        x2b = mk(NP, 256)
        _lib.check(lib.cffm_mlp_fwd(P(ao_d), P(xt_d), 4 * HW * 256, HW, P(frag['wp', 0]), P(frag['w1', 0]), P(frag['w2', 0]), P(pd['bp']),
                                    P(pd['b1']), P(pd['b2']), P(pd['g2']), P(pd['be2']), P(x1), P(z2s), P(mean2), P(rstd2), P(hraw), None,
                                    P(x2b), NP, stream), lib)
        assert torch.equal(x2, x2b)
    # error behaviour: null pointers and bad sizes are reported, not dereferenced
    assert lib.cffm_mlp_fwd(None, None, 0, 1, *([None] * 15), 5, stream) != 0
    assert b'null' in lib.cffm_last_error()
    assert lib.cffm_mlp_fwd(None, None, 0, 1, *([None] * 15), 0, stream) == 0          # no rows: nothing to do
    assert lib.cffm_panel_pack_weight(None, 256, 256, 0, None, stream) != 0


def test_mlp_stages_emulated():
    with emu.active():
        run_mlp_checks(emu.lib(), torch.device('cpu'), [(2, 41), (1, 32)])


@pytest.mark.gpu
def test_mlp_stages_gpu():
    run_mlp_checks(_lib.get(), torch.device('cuda:0'), [(2, 41), (2, 3600), (1, 4096), (3, 1000)])


def test_panel_image_swizzle_is_conflict_free():
    """pnl_off (panel_kernels.h): a ds_read_b128 of gfx950 is served in four groups of 16 lanes ({0-3,12-15,20-27},
    {4-11,16-19,28-31} and the same + 32: MI355X_MICROARCH.md, LDS table); the 16 lanes of a group must hit 16 different
    16-byte bank quads.  Fragment read of k-step k8: lane (l15, g) reads chunk 4 k8 + g of row 16 i + l15."""
    def off(row, chunk):
        return row * 256 + (((chunk ^ row) & 15) << 3) + ((chunk & 16) << 3)
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in grp] for grp in groups]
    for i in range(3):
        for k8 in range(8):
            for grp in groups:
                quads = {(off(16 * i + (l & 15), 4 * k8 + (l >> 4)) * 2 // 16) % 16 for l in grp}
                assert len(quads) == 16, (i, k8, grp)
    # and the map is a bijection on the 32 chunks of a row
    for row in range(48):
        assert sorted(off(row, c) - row * 256 for c in range(32)) == [8 * c for c in range(32)]
