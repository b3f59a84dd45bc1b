"""RECORD (round 5): the round 1-4 form of the CFFM++ prototype block -- ~13 forward / ~20 backward stage launches sequenced from Python --
kept to time it against the fused entry points (cffm_gtc_block_forward / _backward).  usage: python scripts/r05_gtc_unfused_record.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from vss_cffm_amd import _lib, ops  # noqa: E402
from vss_cffm_amd.ops import _ptr, _stream, _require_device  # noqa: E402


class _GtcBlockFn(torch.autograd.Function):
    """SwinTransformerBlock_cluster.forward (pvt/swin_transformer_2d.py:605-665), shift 0:
    x [B,T,256], centers [B,K,256] -> [B,T,256].  The host sequences the stage-level C entry points."""

    @staticmethod
    def forward(ctx, x, centers, *p):
        lib = _lib.get()
        _require_device(x, 'gtc input')
        _require_device(centers, 'gtc centers')
        n1w, n1b, qw, qb, kvw, kvb, pw, pb, n2w, n2b, w1, b1, w2, b2 = [t.detach().contiguous() for t in p]
        x, centers = x.contiguous(), centers.contiguous()
        b, t, c = x.shape
        k = centers.shape[1]
        nt, nk, st = b * t, b * k, _stream(x)
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=x.device)
        z, mean1, rstd1 = new(nt, c), new(nt), new(nt)
        cn, cmean, crstd = new(nk, c), new(nk), new(nk)
        ck = lambda rc: _lib.check(rc, lib)
        ck(lib.cffm_layernorm_fwd(_ptr(x), _ptr(n1w), _ptr(n1b), _ptr(z), _ptr(mean1), _ptr(rstd1), nt, st))
        ck(lib.cffm_layernorm_fwd(_ptr(centers), _ptr(n1w), _ptr(n1b), _ptr(cn), _ptr(cmean), _ptr(crstd), nk, st))
        qraw, kvraw = new(nt, c), new(nk, 2 * c)
        ck(lib.cffm_linear_fwd(_ptr(z), _ptr(qw), _ptr(qraw), nt, c, c, st))          # q third only (:219-220)
        ck(lib.cffm_linear_fwd(_ptr(cn), _ptr(kvw), _ptr(kvraw), nk, 2 * c, c, st))
        ao, lse = new(nt, c), new(nt, 8)
        ck(lib.cffm_gtc_attn_fwd(_ptr(qraw), _ptr(qb), _ptr(kvraw), _ptr(kvb), _ptr(ao), _ptr(lse), b, t, k, st))
        yraw = new(nt, c)
        ck(lib.cffm_linear_fwd(_ptr(ao), _ptr(pw), _ptr(yraw), nt, c, c, st))
        x1, z2, mean2, rstd2 = new(nt, c), new(nt, c), new(nt), new(nt)
        ck(lib.cffm_residual_ln(_ptr(x), nt * c, nt, _ptr(yraw), _ptr(pb), _ptr(n2w), _ptr(n2b), _ptr(x1), _ptr(z2),
                                _ptr(mean2), _ptr(rstd2), nt, st))
        hraw, act, out = new(nt, 4 * c), new(nt, 4 * c), new(b, t, c)
        ck(lib.cffm_linear_gelu_fwd(_ptr(z2), _ptr(w1), _ptr(b1), _ptr(hraw), _ptr(act), nt, 4 * c, c, st))
        ck(lib.cffm_linear_residual_fwd(_ptr(act), _ptr(w2), _ptr(b2), _ptr(x1), _ptr(out), nt, c, 4 * c, st))
        ctx.save_for_backward(x, centers, z, mean1, rstd1, cn, cmean, crstd, qraw, kvraw, ao, lse, x1, z2, mean2, rstd2,
                              hraw, act, n1w, n1b, qw, qb, kvw, kvb, pw, pb, n2w, n2b, w1, b1, w2, b2)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.get()
        (x, centers, z, mean1, rstd1, cn, cmean, crstd, qraw, kvraw, ao, lse, x1, z2, mean2, rstd2, hraw, act,
         n1w, n1b, qw, qb, kvw, kvb, pw, pb, n2w, n2b, w1, b1, w2, b2) = ctx.saved_tensors
        dout = dout.contiguous()
        b, t, c = x.shape
        k = centers.shape[1]
        nt, nk, st = b * t, b * k, _stream(x)
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=x.device)
        ck = lambda rc: _lib.check(rc, lib)
        g = {n: torch.empty_like(v) for n, v in dict(n1w=n1w, n1b=n1b, qb=qb, kvw=kvw, kvb=kvb, pw=pw, pb=pb, n2w=n2w,
                                                     n2b=n2b, w1=w1, b1=b1, w2=w2, b2=b2).items()}
        gqw = torch.zeros_like(qw)                 # rows 256.. (unused k,v thirds) keep a zero gradient
        gqb = torch.zeros_like(qb)
        ck(lib.cffm_linear_bwd_weight(_ptr(dout), _ptr(act), _ptr(g['w2']), nt, c, 4 * c, st))
        dact = new(nt, 4 * c)
        ck(lib.cffm_linear_bwd_input(_ptr(dout), _ptr(w2), _ptr(dact), nt, c, 4 * c, st))
        ck(lib.cffm_gelu_bwd(_ptr(hraw), _ptr(b1), _ptr(dact), nt, 4 * c, _ptr(g['b1']), st))
        ck(lib.cffm_linear_bwd_weight(_ptr(dact), _ptr(z2), _ptr(g['w1']), nt, 4 * c, c, st))
        dz2, dx1 = new(nt, c), new(nt, c)
        ck(lib.cffm_linear_bwd_input(_ptr(dact), _ptr(w1), _ptr(dz2), nt, 4 * c, c, st))
        ck(lib.cffm_ln_bwd_residual(_ptr(x1), _ptr(mean2), _ptr(rstd2), _ptr(n2w), _ptr(dz2), _ptr(dout), _ptr(dx1),
                                    _ptr(g['n2w']), _ptr(g['n2b']), nt, 1, _ptr(g['b2']), _ptr(g['pb']), st))
        ck(lib.cffm_linear_bwd_weight(_ptr(dx1), _ptr(ao), _ptr(g['pw']), nt, c, c, st))
        dao = new(nt, c)
        ck(lib.cffm_linear_bwd_input(_ptr(dx1), _ptr(pw), _ptr(dao), nt, c, c, st))
        dq, dkv = new(nt, c), new(nk, 2 * c)
        ck(lib.cffm_gtc_attn_bwd(_ptr(qraw), _ptr(qb), _ptr(kvraw), _ptr(kvb), _ptr(ao), _ptr(dao), _ptr(lse), _ptr(dq),
                                 _ptr(dkv), b, t, k, st))
        ck(lib.cffm_colsum(_ptr(dq), nt, c, _ptr(gqb), st))                       # first 256 entries
        ck(lib.cffm_linear_bwd_weight(_ptr(dq), _ptr(z), _ptr(gqw), nt, c, c, st))  # first 256 rows
        ck(lib.cffm_colsum(_ptr(dkv), nk, 2 * c, _ptr(g['kvb']), st))
        ck(lib.cffm_linear_bwd_weight(_ptr(dkv), _ptr(cn), _ptr(g['kvw']), nk, 2 * c, c, st))
        dz, dcn = new(nt, c), new(nk, c)
        ck(lib.cffm_linear_bwd_input(_ptr(dq), _ptr(qw), _ptr(dz), nt, c, c, st))
        ck(lib.cffm_linear_bwd_input(_ptr(dkv), _ptr(kvw), _ptr(dcn), nk, 2 * c, c, st))
        dx, dcenters = new(b, t, c), new(b, k, c)
        ck(lib.cffm_ln_bwd_residual(_ptr(x), _ptr(mean1), _ptr(rstd1), _ptr(n1w), _ptr(dz), _ptr(dx1), _ptr(dx),
                                    _ptr(g['n1w']), _ptr(g['n1b']), nt, 1, None, None, st))
        ck(lib.cffm_ln_bwd_residual(_ptr(centers), _ptr(cmean), _ptr(crstd), _ptr(n1w), _ptr(dcn), None, _ptr(dcenters),
                                    _ptr(g['n1w']), _ptr(g['n1b']), nk, 0, None, None, st))   # same norm1 (:622): accumulate
        return (dx, dcenters, g['n1w'], g['n1b'], gqw, gqb, g['kvw'], g['kvb'], g['pw'], g['pb'], g['n2w'], g['n2b'],
                g['w1'], g['b1'], g['w2'], g['b2'])




def main():
    import vss_cffm_amd as V
    dev = torch.device('cuda:0')
    m = V.BasicLayer_cluster(dim=256, depth=1, num_heads=8, window_size=7).to(dev)
    sd = dict(m.blocks[0].named_parameters())
    params = [sd[k] for k in ops.GTC_PARAM_KEYS]
    for k in (8, 100):
        x = torch.randn(2, 3600, 256, device=dev, requires_grad=True)
        c = torch.randn(2, k, 256, device=dev, requires_grad=True)
        gy = torch.randn(2, 3600, 256, device=dev)
        res = {}
        for name, fn in (('unfused (rounds 1-4)', lambda: _GtcBlockFn.apply(x, c, *params)), ('fused (round 5)', lambda: ops.gtc_block(x, c, params))):
            def step():
                for p in params:
                    p.grad = None
                x.grad = c.grad = None
                fn().backward(gy)
            for _ in range(3):
                step()
            ts = []
            for _ in range(20):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); step(); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts.sort()
            res[name] = (ts[10], x.grad.clone(), [p.grad.clone() for p in params])
            print('K = %3d  %-22s %.3f ms per step (eager, median of 20)' % (k, name, ts[10]))
        a, b_ = res['unfused (rounds 1-4)'], res['fused (round 5)']
        err = max(float((u - v).abs().max() / v.abs().max().clamp_min(1e-30)) for u, v in zip([a[1]] + a[2], [b_[1]] + b_[2]))
        print('K = %3d  largest relative difference between the two forms over dx and the 14 parameter gradients: %.2e' % (k, err))


if __name__ == '__main__':
    main()
