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
