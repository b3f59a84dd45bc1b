"""N > 1 path on CPU: two processes, gloo backend, the module wrapped in DistributedDataParallel exactly as
bench.py does on RCCL; kernels run through the emulator library.  Checks that the custom autograd function is
DDP-safe (every parameter gets its gradient in one backward, buckets all-reduce) and that the averaged
gradients equal the mean of the per-rank single-process gradients (clips shard with no data-path collective)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _layer():
    import vss_cffm_amd as V
    from oracle import recipe as R
    m = V.BasicLayer3d3(dim=256, depth=1, num_heads=8, window_size=7, expand_size=3, pool_method='fc', focal_level=2,
                        focal_window=5, focal_l_clips=[1, 2, 3], focal_kernel_clips=[7, 5, 3])
    m.load_state_dict(R.layer_state(1, seed=40), strict=False)
    return m


def _clip(rank):
    from oracle import recipe as R
    return R.synth_input('ddp_x', (1, 4, 256, 8, 8), seed=50 + rank), R.synth_input('ddp_g', (1, 256, 8, 8), seed=60 + rank, scale=1.0)


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), CFFM_EMU_THREADS='2')
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tests import emu
    with emu.active():
        model = torch.nn.parallel.DistributedDataParallel(_layer(), gradient_as_bucket_view=True, broadcast_buffers=False)
        x, g = _clip(rank)
        y = model(x)
        (y[:, -1] * g).sum().backward()
        grads = {k: p.grad.clone() for k, p in model.module.named_parameters()}
        assert all(v is not None for v in grads.values())
        # the step of bench.py: this package's one-launch AdamW on gradients that are views into DDP's buckets
        import vss_cffm_amd as V
        before = {k: p.detach().clone() for k, p in model.module.named_parameters()}
        V.optim.AdamW(model.module.parameters(), lr=1e-3, weight_decay=0.01).step()
        ref = [torch.nn.Parameter(before[k].clone()) for k in before]
        for q, k in zip(ref, before):
            q.grad = grads[k].clone()
        torch.optim.AdamW(ref, lr=1e-3, weight_decay=0.01).step()
        for q, (k, p) in zip(ref, model.module.named_parameters()):
            torch.testing.assert_close(p.detach(), q.detach(), rtol=1e-5, atol=1e-7, msg=k)
    torch.save(grads, os.path.join(out, 'rank%d.pt' % rank))
    torch.save({k: p.detach().clone() for k, p in model.module.named_parameters()}, os.path.join(out, 'param%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_ddp_world_size_2_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g0, g1 = (torch.load(os.path.join(str(tmp_path), 'rank%d.pt' % r)) for r in range(world))
    p0, p1 = (torch.load(os.path.join(str(tmp_path), 'param%d.pt' % r)) for r in range(world))
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k            # all-reduced: identical on both ranks
        assert torch.equal(p0[k], p1[k]), k            # ... and so are the updated parameters
    # single-process reference: mean of the two ranks' own gradients
    from tests import emu, helpers as H
    per_rank = []
    with emu.active():
        for r in range(world):
            m = _layer()
            x, g = _clip(r)
            (m(x)[:, -1] * g).sum().backward()
            per_rank.append({k: p.grad for k, p in m.named_parameters()})
    for k in g0:
        want = (per_rank[0][k] + per_rank[1][k]) / world
        assert H.rel_err(g0[k], want) < 1e-5, k


def _worker_flat(rank, world, port, out):
    """bench.py's default N > 1 path: broadcast_parameters once, one all-reduce of the flat gradient buffer per step."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), CFFM_EMU_THREADS='2')
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tests import emu
    import vss_cffm_amd as V
    with emu.active():
        m = _layer()
        if rank == 1:   # a rank that starts from different values must end up with rank 0's
            with torch.no_grad():
                for p in m.parameters():
                    p.add_(1.0)
        V.distributed.broadcast_parameters(m, 0)
        x, g = _clip(rank)
        (m(x)[:, -1] * g).sum().backward()
        n = V.distributed.allreduce_gradients(list(m.parameters()))
        assert n == 1, n   # every gradient of the layer lives in one buffer -> exactly one collective
        # a gradient that owns its storage next to the layer's: general path, two collectives
        extra = torch.nn.Parameter(torch.ones(5))
        extra.grad = torch.full((5,), float(rank + 1))
        assert V.distributed.allreduce_gradients(list(m.parameters()) + [extra]) == 2
        assert torch.allclose(extra.grad, torch.full((5,), 1.5))
        V.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.01).step()
    torch.save({k: p.grad.clone() for k, p in m.named_parameters()}, os.path.join(out, 'fgrad%d.pt' % rank))
    torch.save({k: p.detach().clone() for k, p in m.named_parameters()}, os.path.join(out, 'fparam%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_flat_allreduce_world_size_2_gloo(tmp_path):
    world = 2
    mp.spawn(_worker_flat, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g0, g1 = (torch.load(os.path.join(str(tmp_path), 'fgrad%d.pt' % r)) for r in range(world))
    p0, p1 = (torch.load(os.path.join(str(tmp_path), 'fparam%d.pt' % r)) for r in range(world))
    from tests import emu, helpers as H
    per_rank = []
    with emu.active():
        for r in range(world):
            m = _layer()
            x, g = _clip(r)
            (m(x)[:, -1] * g).sum().backward()
            per_rank.append({k: p.grad for k, p in m.named_parameters()})
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
        assert torch.equal(p0[k], p1[k]), k
        assert H.rel_err(g0[k], (per_rank[0][k] + per_rank[1][k]) / world) < 1e-5, k


def _layer2():
    import vss_cffm_amd as V
    from oracle import recipe as R
    m = V.BasicLayer3d3(dim=256, depth=2, num_heads=8, window_size=7, expand_size=3, pool_method='fc', focal_level=2,
                        focal_window=5, focal_l_clips=[1, 2, 3], focal_kernel_clips=[7, 5, 3])
    m.load_state_dict(R.layer_state(2, seed=41), strict=False)
    return m


def _worker_blockwise(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), CFFM_EMU_THREADS='2')
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tests import emu
    import vss_cffm_amd as V
    with emu.active():
        m = _layer2()
        V.distributed.broadcast_parameters(m, 0)
        events = []
        red = V.distributed.BlockwiseReducer(on_block=lambda i, t, d: events.append(('block', i, t.numel()))).install()
        try:
            x, g = _clip(rank)
            y = m(x)
            (y[:, -1] * g).sum().backward()
            events.append(('backward returned', len(red.pending)))
            assert red.finish() == 2
        finally:
            red.remove()
        # the exchange of the LAST block is started first (its backward runs first), block 0's after it; both are in
        # flight when backward returns; the two slices tile the flat gradient buffer
        assert [e[:2] for e in events[:2]] == [('block', 1), ('block', 0)] and events[2] == ('backward returned', 2), events
        assert events[0][2] == events[1][2] and events[0][2] * 2 >= sum(p.numel() for p in m.parameters())
        V.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.01).step()
    torch.save({k: p.grad.clone() for k, p in m.named_parameters()}, os.path.join(out, 'bgrad%d.pt' % rank))
    torch.save({k: p.detach().clone() for k, p in m.named_parameters()}, os.path.join(out, 'bparam%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_blockwise_overlapped_allreduce_world_size_4_gloo(tmp_path):
    """The per-block exchange (BlockwiseReducer): 4 ranks, depth 2 -- averaged gradients equal the mean of the ranks' own
    gradients, every rank ends with identical parameters, and the exchange order follows the backward (block 1, then block 0)."""
    world = 4
    mp.spawn(_worker_blockwise, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    gs = [torch.load(os.path.join(str(tmp_path), 'bgrad%d.pt' % r)) for r in range(world)]
    ps = [torch.load(os.path.join(str(tmp_path), 'bparam%d.pt' % r)) for r in range(world)]
    from tests import emu, helpers as H
    per_rank = []
    with emu.active():
        for r in range(world):
            m = _layer2()
            x, g = _clip(r)
            (m(x)[:, -1] * g).sum().backward()
            per_rank.append({k: p.grad for k, p in m.named_parameters()})
    for k in gs[0]:
        for r in range(1, world):
            assert torch.equal(gs[0][k], gs[r][k]), k
            assert torch.equal(ps[0][k], ps[r][k]), k
        assert H.rel_err(gs[0][k], sum(pr[k] for pr in per_rank) / world) < 1e-5, k


def _worker_blockwise_preexisting(rank, world, port, out):
    """with .grad already set (zero_grad(set_to_none=False) / accumulation) autograd ADDS the backward's views into the
    old tensors, so the slices the reducer exchanged are not the gradients: finish(params) must notice and exchange the real ones."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), CFFM_EMU_THREADS='2')
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tests import emu
    import vss_cffm_amd as V
    with emu.active():
        m = _layer()
        V.distributed.broadcast_parameters(m, 0)
        for p in m.parameters():
            p.grad = torch.zeros_like(p)            # what optimizer.zero_grad(set_to_none=False) leaves behind
        # (a) the reducer does not know the parameters until finish(): it exchanges the slices, finish(params) notices that they are
        #     not the gradients -- raising by default, exchanging the real ones on request
        red = V.distributed.BlockwiseReducer().install()
        try:
            x, g = _clip(rank)
            (m(x)[:, -1] * g).sum().backward()
            params = list(m.parameters())
            with pytest.raises(RuntimeError):
                V.distributed.BlockwiseReducer.finish(_Replay(red), params)
            assert red.finish(params, on_mismatch='fallback') == 1 and red.fallbacks == 1
        finally:
            red.remove()
        mine = {k: p.grad.clone() for k, p in m.named_parameters()}
        # (b) installed WITH the parameters it sees the pre-existing gradients at the first block: no slice is exchanged
        #     (no all-reduce racing with autograd's accumulation), finish() exchanges the accumulated gradients once
        for p in m.parameters():
            p.grad = torch.zeros_like(p)
        red = V.distributed.BlockwiseReducer().install(m.parameters())
        try:
            (m(x)[:, -1] * g).sum().backward()
            assert red.accumulating and not red.pending
            assert red.finish() == 0 and red.fallbacks == 1 and not red.accumulating
        finally:
            red.remove()
        for k, p in m.named_parameters():
            assert torch.allclose(p.grad, mine[k], rtol=1e-5, atol=1e-7 * float(mine[k].abs().max())), k
        # (c) ADVICE r4: gradient accumulation after zero_grad(set_to_none=True) -- two backwards, ONE finish().  The first backward's
        #     slices are exchanged block by block and adopted as .grad; the second one must notice them at ITS first block although
        #     exchanges are still pending, complete those, and leave the sum to finish()
        acc_in = dict(m.named_parameters())
        for p in m.parameters():
            p.grad = None
        red = V.distributed.BlockwiseReducer().install(m.parameters())
        try:
            (m(x)[:, -1] * g).sum().backward()
            assert red.pending and not red.accumulating
            x2, g2 = _clip(rank + 7)
            (m(x2)[:, -1] * g2).sum().backward()
            assert red.accumulating and not red.pending and red.drained == 1
            assert red.finish() == 1 and red.fallbacks == 1 and not red.accumulating
        finally:
            red.remove()
        torch.save({k: p.grad.clone() for k, p in acc_in.items()}, os.path.join(out, 'pgrad_acc%d.pt' % rank))
        # (d) ADVICE r5: ONE reducer serving TWO layers in one backward.  The layer that runs its backward first has its slices adopted
        #     as .grad when its autograd node returns; the other layer's first block must not take those for "gradients already there"
        #     (it looks at its own parameters only: ops.current_backward_params) -- every block of both layers is exchanged overlapped,
        #     no fallback.
        m2 = _layer()
        V.distributed.broadcast_parameters(m2, 0)
        for p in list(m.parameters()) + list(m2.parameters()):
            p.grad = None
        red = V.distributed.BlockwiseReducer().install(list(m.parameters()) + list(m2.parameters()))
        try:
            ((m(x)[:, -1] * g).sum() + (m2(x2)[:, -1] * g2).sum()).backward()
            n_blocks = len(red.log)
            assert not red.accumulating and len(red.pending) == n_blocks and n_blocks == len(m.blocks) + len(m2.blocks)
            assert red.finish() == n_blocks and red.fallbacks == 0
        finally:
            red.remove()
        torch.save({k: p.grad.clone() for k, p in m2.named_parameters()}, os.path.join(out, 'pgrad_two%d.pt' % rank))
    torch.save(mine, os.path.join(out, 'pgrad%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


class _Replay:
    """a reducer whose pending list is a copy: lets the 'raise' policy be exercised without consuming the real exchanges"""

    def __init__(self, red):
        self.pending, self.average, self.params, self.accumulating, self.fallbacks = list(red.pending), red.average, None, False, 0
        self.drained = 0

    def _drain(self):
        import vss_cffm_amd as V
        return V.distributed.BlockwiseReducer._drain(self)

    def __setattr__(self, k, v):
        object.__setattr__(self, k, v)


@pytest.mark.timeout(600)
def test_blockwise_reducer_with_preexisting_grads_gloo(tmp_path):
    world = 2
    mp.spawn(_worker_blockwise_preexisting, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g0, g1 = (torch.load(os.path.join(str(tmp_path), 'pgrad%d.pt' % r)) for r in range(world))
    from tests import emu, helpers as H
    per_rank = []
    with emu.active():
        for r in range(world):
            m = _layer()
            x, g = _clip(r)
            (m(x)[:, -1] * g).sum().backward()
            per_rank.append({k: p.grad for k, p in m.named_parameters()})
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
        assert H.rel_err(g0[k], (per_rank[0][k] + per_rank[1][k]) / world) < 1e-5, k
    # (c) two micro-batches per rank, one finish(): the averaged sum of all four gradients
    a0, a1 = (torch.load(os.path.join(str(tmp_path), 'pgrad_acc%d.pt' % r)) for r in range(world))
    second = []
    with emu.active():
        for r in range(world):
            m = _layer()
            x, g = _clip(r + 7)
            (m(x)[:, -1] * g).sum().backward()
            second.append({k: p.grad for k, p in m.named_parameters()})
    for k in a0:
        assert torch.equal(a0[k], a1[k]), k
        want = (per_rank[0][k] + per_rank[1][k] + second[0][k] + second[1][k]) / world
        assert H.rel_err(a0[k], want) < 1e-5, k
    # (d) the second layer of a two-layer backward under one reducer: the ranks' mean of its own gradients (same weights, clip r + 7)
    t0, t1 = (torch.load(os.path.join(str(tmp_path), 'pgrad_two%d.pt' % r)) for r in range(world))
    for k in t0:
        assert torch.equal(t0[k], t1[k]), k
        assert H.rel_err(t0[k], (second[0][k] + second[1][k]) / world) < 1e-5, k


def _bn_case(rank_or_all, world):
    """[N,256,8,6] maps, channels-last, of rank r (or of all ranks concatenated), with the upstream gradients."""
    gen = torch.Generator().manual_seed(77)
    y = torch.randn(2 * world, 256, 8, 6, generator=gen) * 1.3 + 0.2
    df = torch.randn(2 * world, 256, 8, 6, generator=gen)
    ds = torch.randn(2 * world, 12, 256, generator=gen)
    if rank_or_all is None:
        return y, df, ds
    sl = slice(2 * rank_or_all, 2 * rank_or_all + 2)
    return y[sl], df[sl], ds[sl]


def _worker_syncbn(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), CFFM_EMU_THREADS='2')
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from tests import emu
    from vss_cffm_amd import ops
    with emu.active():
        bn = torch.nn.SyncBatchNorm(256)
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, 256))
            bn.bias.copy_(torch.linspace(-0.2, 0.2, 256))
        bn.train()
        y, df, ds = _bn_case(rank, world)
        yd = y.contiguous(memory_format=torch.channels_last).requires_grad_(True)
        fused, stack = ops.bn_relu_pool(yd, bn)
        ((fused * df).sum() + (stack * ds).sum()).backward()
    torch.save(dict(fused=fused.detach(), stack=stack.detach(), dy=yd.grad, dw=bn.weight.grad, db=bn.bias.grad, rm=bn.running_mean,
                    rv=bn.running_var, nbt=bn.num_batches_tracked), os.path.join(out, 'bn%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_syncbn_rows_path_world_size_2_gloo(tmp_path):
    """ops.bn_relu_pool with nn.SyncBatchNorm over two ranks (cffm_head.py:61-66 norm_cfg SyncBN) == single-process BatchNorm over
    the concatenated batch: outputs, input gradients, running buffers; the parameter gradients are the per-rank shares (DDP sums /
    averages them itself), so their SUM over the ranks is the single-process gradient."""
    world = 2
    mp.spawn(_worker_syncbn, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(str(tmp_path), 'bn%d.pt' % r)) for r in range(world)]
    import torch.nn.functional as F
    from tests import helpers as H
    y, df, ds = _bn_case(None, world)
    bn = torch.nn.BatchNorm2d(256).double()
    with torch.no_grad():
        bn.weight.copy_(torch.linspace(0.5, 1.5, 256))
        bn.bias.copy_(torch.linspace(-0.2, 0.2, 256))
    bn.train()
    yr = y.double().requires_grad_(True)
    fused = F.relu(bn(yr))
    stack = F.avg_pool2d(fused, 2).flatten(2).transpose(1, 2)
    ((fused * df.double()).sum() + (stack * ds.double()).sum()).backward()
    for r in range(world):
        sl = slice(2 * r, 2 * r + 2)
        assert H.rel_err(res[r]['fused'], fused[sl].detach()) < 1e-5 and H.rel_err(res[r]['stack'], stack[sl].detach()) < 1e-5
        assert H.rel_err(res[r]['dy'], yr.grad[sl]) < 2e-5
        assert H.rel_err(res[r]['rm'], bn.running_mean) < 1e-5 and H.rel_err(res[r]['rv'], bn.running_var) < 1e-5
        assert int(res[r]['nbt']) == 1
    assert H.rel_err(res[0]['dw'] + res[1]['dw'], bn.weight.grad) < 2e-5
    assert H.rel_err(res[0]['db'] + res[1]['db'], bn.bias.grad) < 2e-5
