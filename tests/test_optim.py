"""AdamW of the hot path (vss_cffm_amd/optim.py -> cffm_adamw_step) against torch.optim.AdamW, the optimizer the
reference's configs name (local_configs/cffm/B1/cffm.b1.480x480.vspw2.160k.py:35).  CPU: the kernel source runs in the
emulator; GPU: the product library."""
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
        assert len(o_mine._tables[0]) == 1          # one table, built once
    assert o_mine.device_step_count() == steps
    for p, q in zip(ref, mine):
        torch.testing.assert_close(q.detach().cpu(), p.detach(), rtol=1e-5, atol=1e-6 * float(p.abs().max()))


def test_adamw_shared_gradient_buffer_emulated():
    with emu.active():
        run_adamw_shared_buffer(torch.device('cpu'))


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


@pytest.mark.gpu
def test_adamw_gpu():
    run_adamw(torch.device('cuda:0'), steps=6)
    run_adamw_shared_buffer(torch.device('cuda:0'), steps=6)


@pytest.mark.gpu
def test_training_step_captured_in_a_graph_matches_eager():
    """forward + backward + AdamW captured once in a torch.cuda.CUDAGraph: three replays == three eager steps
    (the optimizer's step count lives on the device, so the bias correction advances inside the replays)."""
    from oracle import recipe as R
    dev = torch.device('cuda:0')
    st = R.layer_state(1, seed=31)
    x = R.synth_input('gx', (1, 4, 256, 14, 14), seed=32).to(dev)
    gy = torch.zeros(1, 4, 256, 14, 14, device=dev)
    gy[:, -1] = R.synth_input('gg', (1, 256, 14, 14), seed=33, scale=1e-3).to(dev)

    def make():
        m = V.BasicLayer3d3(dim=256, depth=1, num_heads=8, window_size=7, expand_size=3, pool_method='fc', focal_level=2,
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
