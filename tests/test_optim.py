"""AdamW of the hot path (vss_cffm_amd/optim.py -> cffm_adamw_step) against torch.optim.AdamW, the optimizer the
reference's configs name (local_configs/cffm/B1/cffm.b1.480x480.vspw2.160k.py:35).  CPU: the kernel source runs in the
emulator; GPU: the product library."""
import copy

import pytest
import torch

import vss_cffm_amd as V
from vss_cffm_amd import _lib
from tests import emu

SHAPES = [(256,), (768, 256), (3, 5), (2049,), (1024, 256), (1,), (4100,)]   # aligned, ragged and multi-chunk tensors


def run_adamw(device, steps=4):
    gen = torch.Generator().manual_seed(5)
    ref = [torch.nn.Parameter(torch.randn(*s, generator=gen)) for s in SHAPES]
    mine = [torch.nn.Parameter(p.detach().clone().to(device)) for p in ref]
    kw = dict(lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    o_ref, o_mine = torch.optim.AdamW(ref, **kw), V.optim.AdamW(mine, **kw)
    for it in range(steps):
        for p, q in zip(ref, mine):
            g = torch.randn(p.shape, generator=gen) * (10.0 ** (it - 2))
            p.grad = g.clone()
            # a fresh gradient tensor every other step: the chunk table has to follow the new addresses
            if q.grad is None or it % 2 == 0:
                q.grad = g.clone().to(device)
            else:
                q.grad.copy_(g)
        o_ref.step()
        o_mine.step()
    def close(a, b):   # fp32 rounding of a different (but equivalent) operation order; absolute floor from the tensor's scale
        torch.testing.assert_close(a.cpu(), b, rtol=1e-5, atol=1e-6 * float(b.abs().max()))
    for p, q in zip(ref, mine):
        close(q.detach(), p.detach())
        close(o_mine.state[q]['exp_avg'], o_ref.state[p]['exp_avg'])
        close(o_mine.state[q]['exp_avg_sq'], o_ref.state[p]['exp_avg_sq'])


def test_adamw_emulated():
    with emu.active():
        run_adamw(torch.device('cpu'))


def run_adamw_fused_tick_equals_two_launches(device, steps=7):
    """cffm_adamw_step_rows with tickets (the step count advanced inside the update launch, whole chunks prefetched before the row's
    factors are worked out) and without (tick kernel + update kernel): the same parameters and moments, bit for bit, and the same
    device-side step count -- also when a parameter sits out steps (its row must not advance)."""
    gen = torch.Generator().manual_seed(15)
    base = [torch.randn(*s, generator=gen) for s in SHAPES]
    opts, params = [], []
    for fused in (True, False):
        ps = [torch.nn.Parameter(t.clone().to(device)) for t in base]
        o = V.optim.AdamW([{'params': ps[:3], 'lr': 2e-3}, {'params': ps[3:], 'weight_decay': 0.0}], lr=1e-3, weight_decay=0.02)
        o.fused_tick = fused
        o.set_poly_schedule(max_iters=50, power=1.0, min_lr=0.0, warmup_iters=3, warmup_ratio=1e-3)
        opts.append(o)
        params.append(ps)
    for it in range(steps):
        gs = [torch.randn(t.shape, generator=gen) for t in base]
        for o, ps in zip(opts, params):
            for i, (q, g) in enumerate(zip(ps, gs)):
                q.grad = None if (i == 1 and it % 3 == 1) else g.clone().to(device)
            o.step()
    for a, b in zip(*params):
        assert torch.equal(a.detach(), b.detach())
        assert torch.equal(opts[0].state[a]['exp_avg_sq'], opts[1].state[b]['exp_avg_sq'])
    assert opts[0].state_dict()['state'][1]['step'] == opts[1].state_dict()['state'][1]['step'] == steps - 2
    assert opts[0].device_step_count() == opts[1].device_step_count() == steps


def run_adamw_tiny_grids(device):
    """the ticket logic of the one-launch update at grids below its 64 ticket slots: 1, 2 and 65 chunks"""
    for shapes in ([(5,)], [(3,), (2, 2)], [(2048 * 64 + 7,)]):
        gen = torch.Generator().manual_seed(21)
        ref = [torch.nn.Parameter(torch.randn(*s, generator=gen)) for s in shapes]
        mine = [torch.nn.Parameter(p.detach().clone().to(device)) for p in ref]
        o_ref, o_mine = torch.optim.AdamW(ref, lr=1e-2, weight_decay=0.05), V.optim.AdamW(mine, lr=1e-2, weight_decay=0.05)
        for _ in range(4):
            for p, q in zip(ref, mine):
                g = torch.randn(p.shape, generator=gen)
                p.grad, q.grad = g.clone(), g.clone().to(device)
            o_ref.step()
            o_mine.step()
        assert o_mine.device_step_count() == 4
        for p, q in zip(ref, mine):
            torch.testing.assert_close(q.detach().cpu(), p.detach(), rtol=1e-5, atol=1e-6)


def test_adamw_tiny_grids_emulated():
    with emu.active():
        run_adamw_tiny_grids(torch.device('cpu'))


def test_adamw_fused_tick_equals_two_launches_emulated():
    with emu.active():
        run_adamw_fused_tick_equals_two_launches(torch.device('cpu'))


def run_adamw_shared_buffer(device, steps=5):
    """Gradients that alias ONE buffer (what the layer's backward produces): the table holds offsets, the buffer may move."""
    gen = torch.Generator().manual_seed(6)
    ref = [torch.nn.Parameter(torch.randn(*s, generator=gen)) for s in SHAPES]
    mine = [torch.nn.Parameter(p.detach().clone().to(device)) for p in ref]
    kw = dict(lr=2e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    o_ref, o_mine = torch.optim.AdamW(ref, **kw), V.optim.AdamW(mine, **kw)
    sizes = [(p.numel() + 3) // 4 * 4 for p in ref]
    keep = []
    for it in range(steps):
        flat = torch.empty(sum(sizes), device=device)
        keep.append(flat)                                          # kept alive: a different address every step
        for p, q, c in zip(ref, mine, flat.split(sizes)):
            g = torch.randn(p.shape, generator=gen)
            p.grad = g.clone()
            q.grad = c[:p.numel()].view(p.shape).detach()
            q.grad.copy_(g)
        o_ref.step()
        o_mine.step()
        assert len(next(iter(o_mine._devs.values())).tables) == 1          # one table, built once
    assert o_mine.device_step_count() == steps
    for p, q in zip(ref, mine):
        torch.testing.assert_close(q.detach().cpu(), p.detach(), rtol=1e-5, atol=1e-6 * float(p.abs().max()))


def test_adamw_shared_gradient_buffer_emulated():
    with emu.active():
        run_adamw_shared_buffer(torch.device('cpu'))


def run_adamw_resume(device):
    """step, state_dict, more steps, load_state_dict, step -- the cached chunk table must not keep the
    addresses of the replaced moments, and the device-side step count must restart from the loaded one."""
    gen = torch.Generator().manual_seed(9)
    ref = [torch.nn.Parameter(torch.randn(*s, generator=gen)) for s in SHAPES]
    mine = [torch.nn.Parameter(p.detach().clone().to(device)) for p in ref]
    kw = dict(lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    o_ref, o_mine = torch.optim.AdamW(ref, **kw), V.optim.AdamW(mine, **kw)

    def one():
        for p, q in zip(ref, mine):
            g = torch.randn(p.shape, generator=gen)
            p.grad = g.clone()
            if q.grad is None:
                q.grad = g.clone().to(device)
            else:
                q.grad.copy_(g)
        o_ref.step()
        o_mine.step()
    one()
    sd_ref, sd_mine = copy.deepcopy(o_ref.state_dict()), copy.deepcopy(o_mine.state_dict())
    saved = [(p.detach().clone(), q.detach().clone()) for p, q in zip(ref, mine)]
    assert all(s['step'] == 1 for s in sd_mine['state'].values())
    one()
    one()
    with torch.no_grad():
        for (a, b), p, q in zip(saved, ref, mine):
            p.copy_(a)
            q.copy_(b)
    o_ref.load_state_dict(sd_ref)
    o_mine.load_state_dict(sd_mine)
    one()
    assert o_mine.device_step_count() == 2
    for p, q in zip(ref, mine):
        torch.testing.assert_close(q.detach().cpu(), p.detach(), rtol=1e-5, atol=1e-6 * float(p.abs().max()))
        torch.testing.assert_close(o_mine.state[q]['exp_avg'].cpu(), o_ref.state[p]['exp_avg'], rtol=1e-5, atol=1e-7)
    # a state dict written by torch's AdamW (tensor step counts) loads too
    o3 = V.optim.AdamW(mine, **kw)
    o3.load_state_dict(copy.deepcopy(o_ref.state_dict()))
    one_ref = [p.detach().clone() for p in mine]
    for q in mine:
        q.grad.zero_()
    o3.step()
    assert o3.device_step_count() == 3 and all(torch.isfinite(q).all() for q in mine) and len(one_ref) == len(mine)


def test_adamw_resume_emulated():
    with emu.active():
        run_adamw_resume(torch.device('cpu'))


def run_adamw_groups_and_schedule(device):
    """Several parameter groups (paramwise_cfg) in one launch, a learning-rate schedule that changes every step, and a
    parameter that joins late (its own bias correction, as torch's per-parameter step gives it)."""
    gen = torch.Generator().manual_seed(10)
    ref = [torch.nn.Parameter(torch.randn(*s, generator=gen)) for s in SHAPES]
    mine = [torch.nn.Parameter(p.detach().clone().to(device)) for p in ref]

    def groups(ps):
        return [dict(params=ps[:3], lr=1e-3, weight_decay=0.0), dict(params=ps[3:5], lr=1e-2, weight_decay=0.05),
                dict(params=ps[5:], lr=2e-3, betas=(0.8, 0.99))]
    o_ref, o_mine = torch.optim.AdamW(groups(ref), lr=1e-3), V.optim.AdamW(groups(mine), lr=1e-3)
    late = 1                       # SHAPES[1] gets its first gradient at step 3
    for it in range(6):
        for gr, gm in zip(o_ref.param_groups, o_mine.param_groups):
            gr['lr'] = gm['lr'] = gr['lr'] * 0.9          # a schedule
        for i, (p, q) in enumerate(zip(ref, mine)):
            if i == late and it < 2:
                continue
            g = torch.randn(p.shape, generator=gen)
            p.grad = g.clone()
            q.grad = g.clone().to(device)
        o_ref.step()
        o_mine.step()
    dr = next(iter(o_mine._devs.values()))
    assert len(dr.rows) == 4                                # 3 groups + the late cohort
    sd = o_mine.state_dict()
    assert sorted(s['step'] for s in sd['state'].values()) == [4] + [6] * (len(SHAPES) - 1)
    for p, q in zip(ref, mine):
        torch.testing.assert_close(q.detach().cpu(), p.detach(), rtol=2e-5, atol=2e-6 * float(p.abs().max()))


def test_adamw_groups_and_schedule_emulated():
    with emu.active():
        run_adamw_groups_and_schedule(torch.device('cpu'))


def mmcv_poly_lr(base, it, max_iters, power, min_lr, warmup_iters, warmup_ratio):
    """mmcv 1.3 PolyLrUpdaterHook.get_lr + LrUpdaterHook.get_warmup_lr('linear') (by_epoch=False)"""
    lr = (base - min_lr) * (1 - it / max_iters) ** power + min_lr
    if it < warmup_iters:
        lr *= 1 - (1 - it / warmup_iters) * (1 - warmup_ratio)
    return lr


def run_adamw_device_schedule(device, graph=False):
    """set_poly_schedule: the reference's lr schedule (schedule_160k / cffm.b1...160k.py:41-45, scaled down) evaluated on the
    device from the step count, against torch.optim.AdamW driven with the same rates from the host."""
    gen = torch.Generator().manual_seed(14)
    ref = [torch.nn.Parameter(torch.randn(*s, generator=gen)) for s in SHAPES]
    mine = [torch.nn.Parameter(p.detach().clone().to(device)) for p in ref]
    groups = lambda ps: [dict(params=ps[:4], lr=2e-3, weight_decay=0.01), dict(params=ps[4:], lr=2e-2, weight_decay=0.0)]
    o_ref, o_mine = torch.optim.AdamW(groups(ref)), V.optim.AdamW(groups(mine))
    sch = dict(max_iters=40, power=1.0, min_lr=0.0, warmup_iters=6, warmup_ratio=1e-6)
    o_mine.set_poly_schedule(**sch)
    gs = [torch.randn(p.shape, generator=gen) for p in ref]
    for p, q, g in zip(ref, mine, gs):
        p.grad = g.clone()
        q.grad = g.clone().to(device)
    replay = o_mine.step
    if graph:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        base = (2e-3, 2e-2)
        with torch.cuda.stream(s):
            o_mine.step()                         # iteration 0, eager (warm-up of the capture)
        torch.cuda.current_stream().wait_stream(s)
        for grp, b in zip(o_ref.param_groups, base):
            grp['lr'] = mmcv_poly_lr(b, 0, **sch)
        o_ref.step()
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            o_mine.step()
        replay = cg.replay
    for it in range(1 if graph else 0, 12):
        for grp, b in zip(o_ref.param_groups, (2e-3, 2e-2)):
            grp['lr'] = mmcv_poly_lr(b, it, **sch)
        o_ref.step()
        replay()
    if device.type == 'cuda':
        torch.cuda.synchronize()
    assert o_mine.device_step_count() == 12
    for p, q in zip(ref, mine):
        torch.testing.assert_close(q.detach().cpu(), p.detach(), rtol=2e-5, atol=2e-6 * float(p.abs().max()))


def run_adamw_schedule_follows_the_global_iteration(device):
    """mmcv's PolyLrUpdaterHook derives every group's rate from the runner's iteration.  A parameter that sits steps out
    keeps its own step count for the bias correction but must take the rate of the GLOBAL iteration when it takes part again; one
    that first gets a gradient at iteration 5 (the reference trains with find_unused_parameters=True) joins the schedule there
    instead of restarting the linear warm-up.  Against torch.optim.AdamW driven with mmcv's rate of the global iteration."""
    gen = torch.Generator().manual_seed(31)
    init = [torch.randn(200, generator=gen), torch.randn(90, generator=gen), torch.randn(40, generator=gen)]
    mine = [torch.nn.Parameter(t.clone().to(device)) for t in init]
    ref = [torch.nn.Parameter(t.clone().double()) for t in init]
    o_mine = V.optim.AdamW([dict(params=mine, lr=5e-3, weight_decay=0.01)])
    o_ref = torch.optim.AdamW([dict(params=ref, lr=5e-3, weight_decay=0.01)])
    sch = dict(max_iters=30, power=1.0, min_lr=0.0, warmup_iters=8, warmup_ratio=1e-4)
    o_mine.set_poly_schedule(**sch)
    for it in range(12):
        takes_part = [True, it in (0, 1, 4, 9), it >= 5]          # always | intermittent | late
        for p, q, on in zip(mine, ref, takes_part):
            g = torch.randn(p.shape, generator=gen)
            p.grad, q.grad = (g.to(device), g.double()) if on else (None, None)
        o_ref.param_groups[0]['lr'] = mmcv_poly_lr(5e-3, it, **sch)
        o_ref.step()
        o_mine.step()
    for p, q in zip(mine, ref):
        assert float((p.detach().cpu().double() - q.detach()).abs().max()) < 2e-6 * max(1.0, float(q.abs().max()))
    sd = o_mine.state_dict()['state']
    assert [int(sd[i]['step']) for i in range(3)] == [12, 4, 7]       # the bias correction stays per parameter


def test_adamw_schedule_follows_the_global_iteration_emulated():
    with emu.active():
        run_adamw_schedule_follows_the_global_iteration(torch.device('cpu'))


def run_adamw_schedule_survives_resume_and_new_groups(device):
    """the device-side poly schedule counts from the optimizer's GLOBAL iteration.  That count must survive (a) a
    state_dict() / load_state_dict() round trip -- into the same optimizer and into a freshly built one, as a resumed run does; mmcv
    restores runner.iter -- and (b) add_param_group() after set_poly_schedule(); both used to restart the linear warm-up (or, with a
    resolved `first_step`, ran the iteration negative).  Against torch.optim.AdamW driven with mmcv's rate of the global iteration."""
    gen = torch.Generator().manual_seed(77)
    init = [torch.randn(300, generator=gen), torch.randn(64, generator=gen), torch.randn(50, generator=gen)]
    sch = dict(max_iters=40, power=1.0, min_lr=0.0, warmup_iters=9, warmup_ratio=1e-4)
    mk = lambda ts: [torch.nn.Parameter(t.clone().to(device)) for t in ts]
    mine = mk(init)
    ref = [torch.nn.Parameter(t.clone().double()) for t in init]
    o_mine = V.optim.AdamW([dict(params=mine[:2], lr=5e-3, weight_decay=0.01)])
    o_ref = torch.optim.AdamW([dict(params=ref[:2], lr=5e-3, weight_decay=0.01)])
    o_mine.set_poly_schedule(**sch)

    def one(it, o_m, ps, n):
        for p, q in zip(ps[:n], ref[:n]):
            g = torch.randn(p.shape, generator=gen)
            p.grad, q.grad = g.to(device), g.double()
        for grp in o_ref.param_groups:
            grp['lr'] = mmcv_poly_lr(5e-3, it, **sch)
        o_ref.step()
        o_m.step()

    for it in range(4):
        one(it, o_mine, mine, 2)
    sd = copy.deepcopy(o_mine.state_dict())
    assert sd['cffm_global_step'] == 4
    o_mine.load_state_dict(sd)                       # (a) same optimizer
    assert o_mine.global_step() == 4
    for it in range(4, 6):
        one(it, o_mine, mine, 2)
    # (a') a new process: new parameters, new optimizer, schedule set before the first step, then the state loaded
    sd = copy.deepcopy(o_mine.state_dict())
    mine2 = mk([p.detach().cpu() for p in mine])
    o2 = V.optim.AdamW([dict(params=mine2[:2], lr=5e-3, weight_decay=0.01)])
    o2.set_poly_schedule(first_step=0, **sch)
    o2.load_state_dict(sd)
    for it in range(6, 8):
        one(it, o2, mine2, 2)
    # (b) a third tensor joins as a new group at iteration 8 (torch's reference gets the same group)
    o2.add_param_group(dict(params=[mine2[2]], lr=5e-3, weight_decay=0.01))
    o_ref.add_param_group(dict(params=[ref[2]], lr=5e-3, weight_decay=0.01))
    assert o2.global_step() == 8
    for it in range(8, 11):
        one(it, o2, mine2, 3)
    for p, q in zip(mine2, ref):
        assert float((p.detach().cpu().double() - q.detach()).abs().max()) < 2e-6 * max(1.0, float(q.abs().max()))
    # a dict written by torch.optim.AdamW carries no global step: the largest per-parameter count stands in
    o3 = V.optim.AdamW([dict(params=mk(init)[:2], lr=5e-3, weight_decay=0.01)])
    tsd = torch.optim.AdamW([dict(params=ref[:2], lr=5e-3)]).state_dict()
    tsd['state'] = {0: dict(step=torch.tensor(7.), exp_avg=torch.zeros(300), exp_avg_sq=torch.zeros(300)),
                    1: dict(step=torch.tensor(5.), exp_avg=torch.zeros(64), exp_avg_sq=torch.zeros(64))}
    o3.load_state_dict(tsd)
    assert o3.global_step() == 7


def test_adamw_schedule_survives_resume_and_new_groups_emulated():
    with emu.active():
        run_adamw_schedule_survives_resume_and_new_groups(torch.device('cpu'))


@pytest.mark.gpu
def test_adamw_schedule_survives_resume_and_new_groups_gpu():
    run_adamw_schedule_survives_resume_and_new_groups(torch.device('cuda'))


def test_adamw_keeps_captured_tables_alive_when_a_new_row_reallocates_them():
    """a HIP graph captured around step() holds the raw addresses of the optimizer's device tables.  A parameter that gets its
    first gradient later adds a row and REPLACES those tables: the old ones must stay alive (a replay of the stale graph then writes into
    memory the optimizer still owns instead of freed memory) and the caller must be told to capture again."""
    import warnings
    from vss_cffm_amd.optim import AdamW
    with emu.active():
        gen = torch.Generator().manual_seed(5)
        p0, p1 = torch.nn.Parameter(torch.randn(64, generator=gen)), torch.nn.Parameter(torch.randn(32, generator=gen))
        opt = AdamW([p0, p1], lr=1e-3)
        p0.grad = torch.randn(64, generator=gen)
        opt.step()
        dr = next(iter(opt._devs.values()))
        old = (dr.consts, dr.state)
        dr.captured = True                       # what step() records when the current stream is being captured
        p0.grad, p1.grad = torch.randn(64, generator=gen), torch.randn(32, generator=gen)
        with pytest.warns(RuntimeWarning, match='captured again'):
            opt.step()                            # p1 joins with step count 0: a new (group, count) row
        assert len(dr.retired) == 1 and dr.retired[0][0] is old[0] and dr.retired[0][1] is old[1] and not dr.captured
        assert dr.consts is not old[0] and dr.consts.shape[0] == old[0].shape[0] + 1
        with warnings.catch_warnings():
            warnings.simplefilter('error')        # no capture since: further rows replace the tables silently
            p0.grad, p1.grad = torch.randn(64, generator=gen), None
            opt.step()


@pytest.mark.gpu
def test_adamw_schedule_follows_the_global_iteration_gpu():
    run_adamw_schedule_follows_the_global_iteration(torch.device('cuda:0'))


def test_adamw_device_side_poly_schedule_emulated():
    with emu.active():
        run_adamw_device_schedule(torch.device('cpu'))


@pytest.mark.gpu
def test_adamw_device_side_poly_schedule_gpu():
    run_adamw_device_schedule(torch.device('cuda:0'))
    run_adamw_device_schedule(torch.device('cuda:0'), graph=True)


def test_paramwise_groups_follow_mmcv_key_order():
    """mmcv's DefaultOptimizerConstructor: keys sorted by name, then by length descending; the first key that is a substring
    of the full parameter name wins -- so inside `decode_head` the `head` key (lr x10, decay x1) shadows `norm`."""
    ps = {n: torch.nn.Parameter(torch.zeros(1)) for n in (
        'backbone.block1.0.norm1.weight', 'backbone.pos_block.0.weight', 'backbone.patch_embed1.proj.weight',
        'decode_head.decoder_focal.blocks.0.norm1.weight', 'decode_head.linear_pred.weight')}
    gs = V.optim.paramwise_groups(ps.items(), base_lr=6e-5, base_wd=0.01)
    got = {}
    for g in gs:
        for p in g['params']:
            got[[n for n, q in ps.items() if q is p][0]] = (g['lr'], g['weight_decay'])
    assert got['backbone.block1.0.norm1.weight'] == (6e-5, 0.0)
    assert got['backbone.pos_block.0.weight'] == (6e-5, 0.0)
    assert got['backbone.patch_embed1.proj.weight'] == (6e-5, 0.01)
    assert got['decode_head.decoder_focal.blocks.0.norm1.weight'] == (6e-5 * 10, 0.01)
    assert got['decode_head.linear_pred.weight'] == (6e-5 * 10, 0.01)


def test_adamw_skips_params_without_grad_and_rejects_cpu():
    with emu.active():
        a, b = torch.nn.Parameter(torch.ones(10)), torch.nn.Parameter(torch.ones(10))
        opt = V.optim.AdamW([a, b], lr=0.1, weight_decay=0.0)
        a.grad = torch.ones(10)
        opt.step()
        assert torch.all(b == 1) and torch.all(a < 1)
    if not torch.cuda.is_available():   # product mode on a CPU tensor: loud failure, never a fallback
        a = torch.nn.Parameter(torch.ones(10))
        a.grad = torch.ones(10)
        with pytest.raises(_lib.CffmError):
            V.optim.AdamW([a]).step()


def run_adamw_intermittent(device):
    """a parameter that gets a gradient only in some steps keeps torch's per-parameter step count (bias correction, and
    with the device-side schedule its iteration): the row's count advances only in the steps in which the row owns a chunk.
    Both parameters join at step 1 (one row); the second then skips steps -- and gets its own row the first time the counts differ."""
    gen = torch.Generator().manual_seed(9)
    a0, b0 = torch.randn(300, generator=gen), torch.randn(70, generator=gen)
    a, b = torch.nn.Parameter(a0.clone().to(device)), torch.nn.Parameter(b0.clone().to(device))
    ra, rb = torch.nn.Parameter(a0.clone().double()), torch.nn.Parameter(b0.clone().double())
    opt = V.optim.AdamW([a, b], lr=1e-2, weight_decay=0.01)
    ref = torch.optim.AdamW([ra, rb], lr=1e-2, weight_decay=0.01)
    for step in range(8):
        ga, gb = torch.randn(300, generator=gen), torch.randn(70, generator=gen)
        a.grad, ra.grad = ga.to(device), ga.double()
        if step in (0, 3, 6):
            b.grad, rb.grad = gb.to(device), gb.double()
        else:
            b.grad, rb.grad = None, None
        opt.step()
        ref.step()
    assert float((a.detach().cpu().double() - ra.detach()).abs().max()) < 1e-5
    assert float((b.detach().cpu().double() - rb.detach()).abs().max()) < 1e-5
    sd = opt.state_dict()['state']
    assert int(sd[0]['step']) == 8 and int(sd[1]['step']) == 3


def test_adamw_intermittent_gradients_emulated():
    with emu.active():
        run_adamw_intermittent(torch.device('cpu'))


@pytest.mark.gpu
def test_adamw_gpu():
    run_adamw_intermittent(torch.device('cuda:0'))
    run_adamw(torch.device('cuda:0'), steps=6)
    run_adamw_shared_buffer(torch.device('cuda:0'), steps=6)
    run_adamw_resume(torch.device('cuda:0'))
    run_adamw_groups_and_schedule(torch.device('cuda:0'))
    run_adamw_fused_tick_equals_two_launches(torch.device('cuda:0'))
    run_adamw_tiny_grids(torch.device('cuda:0'))


@pytest.mark.gpu
@pytest.mark.parametrize('depth', [1, 3])
def test_training_step_captured_in_a_graph_matches_eager(depth):
    """forward + backward + AdamW captured once in a torch.cuda.CUDAGraph: three replays == three eager steps
    (the optimizer's step count lives on the device, so the bias correction advances inside the replays).  depth 3: the blocks of the
    backward alternate between the two scratch sets, their parameter-gradient tails are launched behind the next block's first kernel
    and the side work runs on the library's four capture streams -- the replayed graph must still compute what the eager launches do."""
    from oracle import recipe as R
    dev = torch.device('cuda:0')
    st = R.layer_state(depth, seed=31)
    x = R.synth_input('gx', (1, 4, 256, 14, 14), seed=32).to(dev)
    gy = torch.zeros(1, 4, 256, 14, 14, device=dev)
    gy[:, -1] = R.synth_input('gg', (1, 256, 14, 14), seed=33, scale=1e-3).to(dev)

    def make():
        m = V.BasicLayer3d3(dim=256, depth=depth, num_heads=8, window_size=7, expand_size=3, pool_method='fc', focal_level=2,
                            focal_window=5, focal_l_clips=[1, 2, 3], focal_kernel_clips=[7, 5, 3])
        m.load_state_dict(st, strict=False)
        m.to(dev)
        return m, V.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.01)

    def body(m, o):
        o.zero_grad(set_to_none=True)
        m(x).backward(gy)
        o.step()

    me, oe = make()
    for _ in range(5):
        body(me, oe)
    mg, og = make()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            body(mg, og)                      # warm-up steps 1, 2 (eager, on the side stream)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(mg, og)                          # capture only: nothing runs, step 3 happens at the first replay
    for _ in range(3):
        g.replay()                            # steps 3, 4, 5
    torch.cuda.synchronize()
    assert og.device_step_count() == 5
    for (k, a), (_, b) in zip(me.named_parameters(), mg.named_parameters()):
        torch.testing.assert_close(b.detach(), a.detach(), rtol=1e-5, atol=1e-7, msg=k)


@pytest.mark.gpu
def test_lr_schedule_reaches_a_replayed_graph():
    """lr / weight decay are not baked into the captured step -- the capture holds a copy node from the
    optimizer's pinned host mirror, so `param_groups[i]['lr'] = ...; refresh_hyper()` between replays is a real schedule;
    state_dict() reports the device-side step count."""
    dev = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(12)
    ref = [torch.nn.Parameter(torch.randn(*s, generator=gen)) for s in SHAPES]
    mine = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref]
    kw = dict(lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02)
    o_ref, o_mine = torch.optim.AdamW(ref, **kw), V.optim.AdamW(mine, **kw)
    gs = [torch.randn(p.shape, generator=gen) for p in ref]
    for p, q, g in zip(ref, mine, gs):
        p.grad = g.clone()
        q.grad = g.clone().to(dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        o_mine.step()
    torch.cuda.current_stream().wait_stream(s)
    o_ref.step()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        o_mine.step()
    for it in range(4):
        lr = 1e-2 * (1 - it / 5.0)
        for grp in o_ref.param_groups + o_mine.param_groups:
            grp['lr'] = lr
            grp['weight_decay'] = 0.02 * (it + 1)
        torch.cuda.synchronize()        # the previous replay has read the pinned mirror (its copy node runs at replay time)
        o_mine.refresh_hyper()
        g.replay()
        o_ref.step()
    torch.cuda.synchronize()
    assert o_mine.device_step_count() == 5
    assert all(st['step'] == 5 for st in o_mine.state_dict()['state'].values())
    for p, q in zip(ref, mine):
        torch.testing.assert_close(q.detach().cpu(), p.detach(), rtol=2e-5, atol=2e-6 * float(p.abs().max()))


@pytest.mark.gpu
def test_stage_events_captured_into_a_graph_are_readable_after_each_replay():
    """cffm_profile_collect_graph: a stage enabled while the stream is being captured leaves event-record NODES in the
    graph; after a replay the pair of every captured launch is readable (bench.py's live roofline timing under replay)."""
    import ctypes as C
    from oracle import recipe as R
    from vss_cffm_amd import _lib
    lib = _lib.get()
    dev = torch.device('cuda:0')
    depth = 2
    m = V.BasicLayer3d3(dim=256, depth=depth, num_heads=8, window_size=7, expand_size=3, pool_method='fc', focal_level=2,
                        focal_window=5, focal_l_clips=[1, 2, 3], focal_kernel_clips=[7, 5, 3])
    m.load_state_dict(R.layer_state(depth, seed=41), strict=False)
    m.to(dev)
    x = R.synth_input('gx', (1, 4, 256, 14, 14), seed=42).to(dev)
    nst = lib.cffm_profile_stage_count()
    names = [lib.cffm_profile_stage_name(i).decode() for i in range(nst)]
    ia = names.index('cfm_attn_fwd')
    ms, n = (C.c_float * nst)(), (C.c_int * nst)()
    with torch.no_grad():
        want = m(x).clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            m(x)
        torch.cuda.current_stream().wait_stream(s)
        assert lib.cffm_profile_collect_graph(ms, n, 1) == 0
        lib.cffm_profile_enable(1 << ia)
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                y = m(x)
        finally:
            lib.cffm_profile_enable(0)
            lib.cffm_profile_collect(ms, n)
    for _ in range(2):
        g.replay()
        torch.cuda.synchronize()
        assert lib.cffm_profile_collect_graph(ms, n, 0) == 0, lib.cffm_last_error().decode()
        assert n[ia] == depth and sum(n) == depth
        assert 0.0 < ms[ia] < 50.0
    torch.testing.assert_close(y, want, rtol=0, atol=0)
    assert lib.cffm_profile_collect_graph(ms, n, 1) == 0
    assert lib.cffm_profile_collect_graph(ms, n, 0) == 0 and sum(n) == 0


def run_adamw_host_step_entry(lib, device):
    """cffm_adamw_step: the entry point with the step count on the host (absolute addresses in the table)."""
    import ctypes as C
    import numpy as np
    gen = torch.Generator().manual_seed(8)
    p0 = torch.randn(5000, generator=gen)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05)
    p, m, v = p0.clone().to(device), torch.zeros(5000, device=device), torch.zeros(5000, device=device)
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream) if device.type == 'cuda' else None
    for t in (1, 2, 3):
        g = torch.randn(5000, generator=gen)
        ref.grad = g.clone()
        opt.step()
        gd = g.to(device)
        off = np.arange(0, 5000, 2048, dtype=np.int64)
        tab = np.stack([p.data_ptr() + 4 * off, gd.data_ptr() + 4 * off, m.data_ptr() + 4 * off, v.data_ptr() + 4 * off,
                        np.minimum(2048, 5000 - off)], axis=1)
        tabd = torch.from_numpy(tab).to(device)
        assert lib.cffm_adamw_step(C.c_void_p(tabd.data_ptr()), tab.shape[0], 1e-2, 0.9, 0.99, 1e-8, 0.05, t, stream) == 0
        if device.type == 'cuda':
            torch.cuda.synchronize()
    torch.testing.assert_close(p.cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
    assert lib.cffm_adamw_step(C.c_void_p(tabd.data_ptr()), tab.shape[0], 1e-2, 0.9, 0.99, 1e-8, 0.05, 0, stream) != 0   # t >= 1


def test_adamw_host_step_entry_emulated():
    with emu.active():
        run_adamw_host_step_entry(emu.lib(), torch.device('cpu'))


@pytest.mark.gpu
def test_adamw_host_step_entry_gpu():
    run_adamw_host_step_entry(_lib.get(), torch.device('cuda:0'))
