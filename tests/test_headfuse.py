"""The fused row path around the hot path (SURVEY.md 8f.1 remainder): `linear_fuse`'s BatchNorm + ReLU + the 1/4 -> 1/8 resize as
two passes on token rows (cffm_head.py:119,131-135), and the layer called on rows -- against stock PyTorch's batch_norm / relu /
interpolate and against the NCHW layer call, forward and backward.  (The whole head through this path is compared with the
reference head's golden vectors in tests/test_boundary.py.)"""
import pytest
import torch
import torch.nn.functional as F

import vss_cffm_amd as V
from oracle import recipe as R, ref_import as RI
from tests import emu, helpers as H
from vss_cffm_amd import _lib, ops
from vss_cffm_amd import head as Hd
from vss_cffm_amd.registry import build_head


def run_bn_relu_pool(device, n=2, h=6, w=10):
    gen = torch.Generator().manual_seed(3)
    y0 = torch.randn(n, h, w, 256, generator=gen) * 2 + 0.3
    bn = torch.nn.BatchNorm2d(256)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
    ref = torch.nn.BatchNorm2d(256)
    ref.load_state_dict(bn.state_dict())
    bn.to(device)
    gf = torch.randn(n, 256, h, w, generator=gen)
    gs = torch.randn(n, 256, h // 2, w // 2, generator=gen)
    for training in (True, False):
        bn.train(training)
        ref.train(training)
        ya = y0.clone().to(device).permute(0, 3, 1, 2).requires_grad_(True)           # channels-last memory, as the embedding returns it
        fused, stack = ops.bn_relu_pool(ya, bn)
        yb = y0.clone().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        # ReLU with the mask the fused op used: a pre-activation within rounding of zero may switch on one side only (a handful of
        # the 29 M elements of the large case, different ones from run to run with the CPU's reduction order), which moves that
        # element's gradient and the per-channel sums by whole terms; away from the kink the two masks are the same
        pre = ref(yb)
        assert H.rel_err(fused.cpu(), F.relu(pre)) < 1e-5
        f2 = pre * (fused.detach().cpu() > 0).float()
        s2 = F.interpolate(f2, size=(h // 2, w // 2), mode='bilinear', align_corners=False)   # == the 2x2 average for even sides
        assert H.rel_err(fused.cpu(), f2) < 1e-5
        assert H.rel_err(stack.view(n, h // 2, w // 2, 256).permute(0, 3, 1, 2).cpu(), s2) < 1e-5
        for p in list(bn.parameters()) + list(ref.parameters()):
            p.grad = None
        ((fused * gf.to(device)).sum() + (stack.view(n, h // 2, w // 2, 256).permute(0, 3, 1, 2) * gs.to(device)).sum()).backward()
        ((f2 * gf).sum() + (s2 * gs).sum()).backward()
        assert H.rel_err(ya.grad.cpu(), yb.grad) < 2e-5
        assert H.rel_err(bn.weight.grad.cpu(), ref.weight.grad) < 2e-5 and H.rel_err(bn.bias.grad.cpu(), ref.bias.grad) < 2e-5
        assert H.rel_err(bn.running_mean.cpu(), ref.running_mean) < 1e-5 and H.rel_err(bn.running_var.cpu(), ref.running_var) < 1e-5
        assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked)
    # Dropout2d of the first output folded into the pass: whole (frame, channel) planes scaled by 0 or 1/(1-p); the stack is untouched
    bn.train(True); ref.train(True)
    mask = (torch.rand(n, 256, generator=gen) > 0.3).float() / 0.7
    ya = y0.clone().to(device).permute(0, 3, 1, 2).requires_grad_(True)
    fused, stack = ops.bn_relu_pool(ya, bn, drop_mask=mask.to(device))
    yb = y0.clone().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    pre = ref(yb)
    # (the ReLU mask the op used, read off its kept planes -- see above; a dropped plane leaves no trace, the reference's own is used there)
    relu_on = torch.where(mask[:, :, None, None] > 0, fused.detach().cpu() != 0, pre.detach() > 0).float()
    r2 = pre * relu_on
    f2 = r2 * mask[:, :, None, None]
    s2 = F.interpolate(r2, size=(h // 2, w // 2), mode='bilinear', align_corners=False)
    assert H.rel_err(fused.cpu(), f2) < 1e-5 and H.rel_err(stack.view(n, h // 2, w // 2, 256).permute(0, 3, 1, 2).cpu(), s2) < 1e-5
    for p in list(bn.parameters()) + list(ref.parameters()):
        p.grad = None
    ((fused * gf.to(device)).sum() + (stack.view(n, h // 2, w // 2, 256).permute(0, 3, 1, 2) * gs.to(device)).sum()).backward()
    ((f2 * gf).sum() + (s2 * gs).sum()).backward()
    # (inside a DROPPED plane the op's ReLU mask cannot be read off its output, yet it still shapes the stack: a pre-activation within
    #  rounding of zero -- a handful of the 29 M elements of the large case, which ones depends on the CPU's reduction order -- may be on
    #  in one implementation and off in the other; each such element moves its own gradient by a whole term.  At most 8 of them.)
    dgrad = (ya.grad.cpu() - yb.grad).abs()
    assert int((dgrad > 2e-5 * float(yb.grad.abs().max())).sum()) <= 8, float(dgrad.max())
    # (the same handful of elements enters the per-channel sums with whole terms)
    assert H.rel_err(bn.weight.grad.cpu(), ref.weight.grad) < 2e-3 and H.rel_err(bn.bias.grad.cpu(), ref.bias.grad) < 2e-3
    with pytest.raises(_lib.CffmError):
        ops.bn_relu_pool(ya, bn, drop_mask=torch.ones(n, 255, device=device))
    # only one of the two outputs used downstream; stack not requested
    ya = y0.clone().to(device).permute(0, 3, 1, 2).requires_grad_(True)
    fused, stack = ops.bn_relu_pool(ya, bn.train(), want_stack=False)
    assert stack is None
    fused.sum().backward()
    assert torch.isfinite(ya.grad).all()
    with pytest.raises(_lib.CffmError):
        ops.bn_relu_pool(torch.zeros(1, 256, 5, 4, device=device), bn)              # odd side: the resize is not a 2x2 average


def run_bn_momentum_none(device):
    """BatchNorm(momentum=None): torch keeps a CUMULATIVE moving average (factor 1 / num_batches_tracked) -- ADVICE r2."""
    gen = torch.Generator().manual_seed(5)
    bn, ref = torch.nn.BatchNorm2d(256, momentum=None), torch.nn.BatchNorm2d(256, momentum=None)
    bn.to(device).train(); ref.train()
    for _ in range(3):
        y = torch.randn(2, 4, 6, 256, generator=gen) * 1.5 + 0.1
        ops.bn_relu_pool(y.to(device).permute(0, 3, 1, 2), bn)
        ref(y.permute(0, 3, 1, 2).contiguous())
    assert int(bn.num_batches_tracked) == 3
    assert H.rel_err(bn.running_mean.cpu(), ref.running_mean) < 1e-5 and H.rel_err(bn.running_var.cpu(), ref.running_var) < 1e-5


def test_bn_relu_pool_emulated():
    with emu.active():
        run_bn_relu_pool(torch.device('cpu'))
        run_bn_momentum_none(torch.device('cpu'))


@pytest.mark.gpu
def test_bn_relu_pool_gpu():
    run_bn_relu_pool(torch.device('cuda:0'))
    run_bn_relu_pool(torch.device('cuda:0'), n=8, h=120, w=120)


def run_rows_resize(device):
    """ops.rows_resize against F.interpolate(bilinear, align_corners=False), forward and backward, up- and down-sampling, ragged
    factors, and a gradient that arrives as a slice of a larger rows buffer"""
    gen = torch.Generator().manual_seed(5)
    for (n, c, h, w, HH, WW) in [(2, 124, 6, 7, 12, 14), (1, 8, 5, 9, 13, 20), (2, 16, 8, 8, 8, 8), (1, 4, 9, 7, 4, 3), (3, 12, 1, 1, 5, 4), (0, 8, 2, 2, 4, 4)]:
        x = torch.randn(n, h, w, c, generator=gen)
        buf = torch.randn(n, 3, HH, WW, c, generator=gen)            # the gradient: map 2 of every clip
        xa = x.clone().to(device).requires_grad_(True)
        y = ops.rows_resize(xa, (HH, WW))
        y.backward(buf.to(device)[:, 2])
        xr = x.double().permute(0, 3, 1, 2).clone().requires_grad_(True)
        yr = F.interpolate(xr, size=(HH, WW), mode='bilinear', align_corners=False)
        yr.backward(buf[:, 2].double().permute(0, 3, 1, 2))
        if n:
            assert H.rel_err(y.detach().cpu().permute(0, 3, 1, 2), yr.detach()) < 1e-6, (n, c, h, w, HH, WW)
            assert H.rel_err(xa.grad.cpu().permute(0, 3, 1, 2), xr.grad) < 1e-6, (n, c, h, w, HH, WW)
    with pytest.raises(_lib.CffmError):
        ops.rows_resize(torch.zeros(1, 2, 2, 6, device=device), (4, 4))


def test_rows_resize_emulated():
    with emu.active():
        run_rows_resize(torch.device('cpu'))


@pytest.mark.gpu
def test_rows_resize_gpu():
    run_rows_resize(torch.device('cuda:0'))


def run_layer_rows(device, depth=2, b=2, h=8, w=13):
    m = V.BasicLayer3d3(dim=256, depth=depth, num_heads=8, window_size=7, expand_size=3, pool_method='fc', focal_level=2,
                        focal_window=5, focal_l_clips=[1, 2, 3], focal_kernel_clips=[7, 5, 3])
    m.load_state_dict(R.layer_state(depth, seed=51), strict=False)
    m.to(device)
    x = R.synth_input('rows_x', (b, 4, 256, h, w), seed=52).to(device)
    gy = R.synth_input('rows_g', (b, 256, h, w), seed=53, scale=1.0).to(device)
    xa = x.clone().requires_grad_(True)
    (m(xa)[:, -1] * gy).sum().backward()
    want = {k: p.grad.clone() for k, p in m.named_parameters()}
    for p in m.parameters():
        p.grad = None
    xr = x.permute(0, 1, 3, 4, 2).reshape(b, 4, h * w, 256).contiguous().requires_grad_(True)
    yr = m.forward_rows(xr, h, w)
    ya = m(x)[:, -1].permute(0, 2, 3, 1).reshape(b, h * w, 256)
    assert torch.equal(yr, ya)                                                      # the same kernels, the same order
    (yr * gy.permute(0, 2, 3, 1).reshape(b, h * w, 256)).sum().backward()
    assert torch.equal(xr.grad, xa.grad.permute(0, 1, 3, 4, 2).reshape(b, 4, h * w, 256))
    for k, p in m.named_parameters():
        assert torch.equal(p.grad, want[k]), k


def test_layer_on_rows_equals_nchw_call_emulated():
    with emu.active():
        run_layer_rows(torch.device('cpu'))
        run_layer_rows(torch.device('cpu'), depth=1, b=1, h=14, w=7)


@pytest.mark.gpu
def test_layer_on_rows_equals_nchw_call_gpu():
    run_layer_rows(torch.device('cuda:0'), depth=2, b=2, h=60, w=60)


def run_head_ab(device, size=64, chans=(32, 64, 160, 256), depths=1):
    """whole head, training mode (batch statistics), fused row path vs the same head with BatchNorm / ReLU / resize / NCHW layer call in
    stock PyTorch: logits, loss and every gradient agree"""
    from tests.golden.make_golden_head import feature_maps, labels
    head = build_head(RI.head_cfg(in_channels=chans, depths=depths))
    head.load_state_dict(R.synth_state(head, seed=61), strict=False)
    head.dropout.p = 0.0
    Hd.revert_sync_batchnorm(head)
    head.to(device).train()
    feats = [f.to(device) for f in feature_maps(2, 4, size, chans=chans, seed=62)]
    lab = labels(2, 4, size, seed=63).to(device)
    res = {}
    for impl in ('hip', 'torch'):
        head.rows_impl = impl
        for p in head.parameters():
            p.grad = None
        with torch.no_grad():
            head.linear_fuse.bn.running_mean.zero_()
            head.linear_fuse.bn.running_var.fill_(1.0)
        fg = [f.clone().requires_grad_(True) for f in feats]
        out = head(fg, 2, 4)
        loss = head.losses(out, lab)
        loss['loss_seg'].backward()
        res[impl] = (out.detach(), float(loss['loss_seg']), [f.grad for f in fg], {k: p.grad for k, p in head.named_parameters() if p.grad is not None},
                     head.linear_fuse.bn.running_var.clone())
    head.rows_impl = 'hip'
    a, b = res['hip'], res['torch']
    # (on the GPU the hot path rounds its attention operands to f16: 1e-7-level differences in the clip stack move single f16
    #  roundings, which shows as 1e-5..1e-4 in the logits -- well inside the 5e-4 forward tolerance of the parity tests)
    tol = 2e-5 if device.type == 'cpu' else 3e-4
    assert H.rel_err(a[0], b[0]) < tol and abs(a[1] - b[1]) < 10 * tol * abs(b[1])
    for ga, gb in zip(a[2], b[2]):
        assert H.rel_err(ga, gb) < (5e-4 if device.type == 'cpu' else 2e-3)   # (fp32 reassociation; a ReLU input within 1e-6 of zero may flip)
    assert set(a[3]) == set(b[3])
    for k in a[3]:
        if k.startswith('linear_c') and k.endswith('proj.bias'):
            continue      # a constant added in front of a training-mode BatchNorm: its exact gradient is 0, both sides hold rounding noise
        # (CPU: 1e-3 -- the dense ring-bias table's gradient is the sum of a few 1e-6-sized terms; the two evaluation orders measured
        #  4.4e-4 with the round-2 GEMM sequence and 5.1e-4 with the round-3 row-panel kernels, every other tensor <= 2.4e-4)
        assert H.rel_err(a[3][k], b[3][k]) < (1e-3 if device.type == 'cpu' else 3e-3), k
    assert H.rel_err(a[4], b[4]) < 1e-5


def run_headpp_ab(device, size=64, chans=(32, 64, 160, 256), depths=1):
    """The CFFM++ head (cffm_head.py:423-535; frozen embedding, detached CFFM branch, prototype layer + linear_pred3), training and eval
    mode: its row path against the same head on the reference's op sequence -- logits, loss, which parameters train and their gradients."""
    import os, tempfile
    from tests.golden.make_golden_head import feature_maps, labels
    head = build_head(RI.head_cfg(kind='CFFMHead_clips_resize1_8_finetune_w_prototype3', in_channels=chans, depths=depths))
    head.load_state_dict(R.synth_state(head, seed=71), strict=False)
    head.dropout.p = head.dropout3.p = 0.0
    Hd.revert_sync_batchnorm(head)
    head.to(device)
    feats = [f.to(device) for f in feature_maps(2, 4, size, chans=chans, seed=72)]
    lab = labels(2, 4, size, seed=73).to(device)
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        metas = []
        for v in range(2):
            os.makedirs(os.path.join(tmp, 'vid%d' % v))
            torch.save(R.synth_input('centers%d' % v, (1, 8, 256), seed=74 + v, scale=1.0), os.path.join(tmp, 'vid%d' % v, 'centers.pt'))
            metas.append({'filename': tmp + '/data/vid%d/origin/0001.jpg' % v})
        head.save_path = tmp + '/'
        for impl in ('hip', 'torch'):
            head.rows_impl = impl
            for p in head.parameters():
                p.grad = None
            head.eval()
            with torch.no_grad():
                ev = head(feats, 2, 4, None, metas)
            head.train()
            out = head(feats, 2, 4, None, metas)
            loss = head.losses(out, lab)
            loss['loss_seg'].backward()
            res[impl] = (ev, out.detach(), float(loss['loss_seg'].detach()), {k: p.grad for k, p in head.named_parameters() if p.grad is not None})
    head.rows_impl = 'hip'
    a, b = res['hip'], res['torch']
    tol = 2e-5 if device.type == 'cpu' else 3e-4
    assert a[0].shape == b[0].shape and a[1].shape == b[1].shape
    assert H.rel_err(a[0], b[0]) < tol and H.rel_err(a[1], b[1]) < tol and abs(a[2] - b[2]) < 10 * tol * abs(b[2])
    assert set(a[3]) == set(b[3]) and all(k.split('.')[0] in ('linear_pred3', 'decoder_swin') for k in a[3]), sorted(a[3])
    for k in a[3]:
        assert H.rel_err(a[3][k], b[3][k]) < (1e-3 if device.type == 'cpu' else 3e-3), k


def test_headpp_rows_path_equals_torch_glue_emulated():
    with emu.active():
        run_headpp_ab(torch.device('cpu'))


@pytest.mark.gpu
def test_headpp_rows_path_equals_torch_glue_gpu():
    run_headpp_ab(torch.device('cuda:0'), size=128, chans=(64, 128, 320, 512), depths=2)


def test_head_rows_path_equals_torch_glue_emulated():
    with emu.active():
        run_head_ab(torch.device('cpu'))


@pytest.mark.gpu
def test_head_rows_path_equals_torch_glue_gpu():
    run_head_ab(torch.device('cuda:0'), size=128, chans=(64, 128, 320, 512), depths=2)


def run_head_pieces(device):
    """The round-6 pieces of the rows path, each against stock torch under autograd:
    (a) ops.conv1x1(..., clips=B, extra=1) + ops.cat_into == torch.cat([conv(x) per clip, x2], 1) -- and really without copying the frame
        logits (the result shares their memory); gradients of x, weight, bias, x2;
    (b) the composed embedding weights and their constant (cffm_fuse_compose_fwd / _bwd behind ops.segformer_fuse's _ComposeFn): A_i = Wf_i W_i,
        d = sum_i Wf_i b_i, gradients of the nine tensors;
    (c) the loss scalars of ops.head_cross_entropy (cffm_upce_maps_finalize) are covered by tests/test_segloss.py's goldens."""
    from vss_cffm_amd import ops
    gen = torch.Generator().manual_seed(5)
    b, t, c, o, h, w = 2, 3, 16, 12, 5, 6
    x = torch.randn(b * t, c, h, w, generator=gen).to(device).requires_grad_(True)
    wt = torch.randn(o, c, 1, 1, generator=gen).to(device).requires_grad_(True)
    bs = torch.randn(o, generator=gen).to(device).requires_grad_(True)
    x2 = torch.randn(b, 1, o, h, w, generator=gen).to(device).permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3).requires_grad_(True)
    gy = torch.randn(b, t + 1, o, h, w, generator=gen).to(device)
    y = ops.conv1x1(x, wt, bs, clips=b, extra=1)
    assert ops.cat_room(y, 1) and not ops.cat_room(y, 2)
    z = ops.cat_into(y, x2)
    assert z.shape == (b, t + 1, o, h, w) and z.data_ptr() == y.data_ptr()            # the frame logits did not move
    (z * gy).sum().backward()
    xr, wr, br, x2r = [v.detach().double().cpu().requires_grad_(True) for v in (x, wt, bs, x2)]
    zr = torch.cat([torch.nn.functional.conv2d(xr, wr, br).view(b, t, o, h, w), x2r], 1)
    (zr * gy.double().cpu()).sum().backward()
    assert H.rel_err(z.detach(), zr.detach()) < 2e-5
    for got, ref in ((x, xr), (wt, wr), (bs, br), (x2, x2r)):
        assert H.rel_err(got.grad, ref.grad) < 2e-5
    assert not ops.cat_room(ops.conv1x1(x, wt, bs, clips=b), 1)                       # no room asked for: torch.cat's job
    # (a') the same concatenation as ONE node whose backward runs on the library's deferred branch (ops.frame_logits_cat; parameters through
    #      ops.late_params so that autograd accumulates their gradients behind the join): same values, same gradients
    for v in (x, wt, bs, x2):
        v.grad = None
    xc = x.detach().permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).requires_grad_(True)       # channels-last, as the rows path hands it
    lw, lb = ops.late_params(wt, bs)
    z2 = ops.frame_logits_cat(xc, lw, lb, x2, b)
    assert torch.equal(z2.detach(), z.detach())
    (z2 * gy).sum().backward()
    if device.type == 'cuda':
        torch.cuda.synchronize()
    for got, ref in ((xc, xr), (wt, wr), (bs, br), (x2, x2r)):
        assert H.rel_err(got.grad, ref.grad) < 2e-5
    assert not ops._DEFERRED                                                          # joined by the end of the backward pass
    # (b)
    k, e, cins = 3, 256, (8, 20, 64)
    fw = (torch.randn(e, k * e, generator=gen) * 0.05).to(device).requires_grad_(True)
    lw = [(torch.randn(e, ci, generator=gen) * 0.1).to(device).requires_grad_(True) for ci in cins]
    lb = [torch.randn(e, generator=gen).to(device).requires_grad_(True) for _ in cins]
    outs = ops._ComposeFn.apply(fw, *lw, *lb)
    gm = [torch.randn(e, ci, generator=gen).to(device) for ci in cins]
    gd = torch.randn(e, generator=gen).to(device)
    (sum((a * g_).sum() for a, g_ in zip(outs[:k], gm)) + (outs[k] * gd).sum()).backward()
    fr = fw.detach().double().cpu().requires_grad_(True)
    lwr = [v.detach().double().cpu().requires_grad_(True) for v in lw]
    lbr = [v.detach().double().cpu().requires_grad_(True) for v in lb]
    blocks = [fr[:, (k - 1 - i) * e:(k - i) * e] for i in range(k)]                   # cat order c_k .. c_1
    mats = [blocks[i] @ lwr[i] for i in range(k)]
    d = sum(blocks[i] @ lbr[i] for i in range(k))
    (sum((a * g_.double().cpu()).sum() for a, g_ in zip(mats, gm)) + (d * gd.double().cpu()).sum()).backward()
    for a, r in zip(outs[:k], mats):
        assert H.rel_err(a.detach(), r.detach()) < 2e-5
    assert H.rel_err(outs[k].detach(), d.detach()) < 1e-5
    assert H.rel_err(fw.grad, fr.grad) < 2e-5
    for got, ref in zip(lw + lb, lwr + lbr):
        assert H.rel_err(got.grad, ref.grad) < 2e-5


def test_head_pieces_emulated():
    with emu.active():
        run_head_pieces(torch.device('cpu'))


@pytest.mark.gpu
def test_head_pieces_gpu():
    run_head_pieces(torch.device('cuda'))


@pytest.mark.gpu
def test_head_step_replayed_equals_eager_gpu():
    """The whole head's training step (rows path: stage calls on branches of the library's streams, a scratch pool per branch) captured in a HIP
    graph and replayed against the same step launched eagerly: logits, loss, feature gradients and every parameter gradient BIT-identical --
    the branches run in another interleaving under the graph executor, and nothing on them uses atomics, so any difference would be a missing
    dependency (a join taken too early, two branches on one scratch).  Exception as in tests/test_gpu_parity.py: the four Linear weight
    gradients of a CFFM block are split over the contraction differently under capture (rounding-level differences)."""
    from tests.golden.make_golden_head import feature_maps, labels
    device = torch.device('cuda')
    chans, size = (64, 128, 320, 512), 256
    head = build_head(RI.head_cfg(in_channels=chans, depths=2))
    head.load_state_dict(R.synth_state(head, seed=71), strict=False)
    head.dropout.p = 0.0
    Hd.revert_sync_batchnorm(head)
    head.to(device).train()
    feats = [f.to(device).requires_grad_(True) for f in feature_maps(2, 4, size, chans=chans, seed=72)]
    lab = labels(2, 4, size, seed=73).to(device)

    def step():
        for p in head.parameters():
            p.grad = None
        for f in feats:
            f.grad = None
        out = head(feats, 2, 4)
        loss = head.losses(out, lab)
        loss['loss_seg'].backward()
        return out, loss['loss_seg']

    snap = lambda out, loss: (out.detach().clone(), loss.detach().clone(), [f.grad.clone() for f in feats],
                              {k: p.grad.clone() for k, p in head.named_parameters() if p.grad is not None})
    for _ in range(2):
        ref = snap(*step())
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out_g, loss_g = step()
    reps = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        reps.append(snap(out_g, loss_g))
    for r in reps:
        assert torch.equal(r[0], ref[0]) and torch.equal(r[1], ref[1])
        for a, b in zip(r[2], ref[2]):
            assert torch.equal(a, b)
        assert set(r[3]) == set(ref[3])
        for k in ref[3]:
            if 'decoder_focal' in k and k.endswith('.weight') and ref[3][k].dim() == 2 and 'pool' not in k:
                assert float((r[3][k] - ref[3][k]).abs().max()) <= 2e-6 * float(ref[3][k].abs().max()), k
                assert torch.equal(r[3][k], reps[0][3][k]), k         # replay to replay: always bit-identical
            else:
                assert torch.equal(r[3][k], ref[3][k]), k
