"""AdamW for the parameters of the CFFM hot path: one HIP launch per step, capturable in a HIP graph.

The reference trains the head with ``torch.optim.AdamW`` (lr 6e-5, betas (0.9, 0.999), weight decay 0.01:
local_configs/cffm/B1/cffm.b1.480x480.vspw2.160k.py:35).  The update rule here is that optimizer's (decoupled weight
decay, bias-corrected moments, amsgrad off); what differs is the launch shape: every parameter tensor of every block is
cut into 2048-element chunks listed in one device table, and a single kernel (``cffm_adamw_step_dev``,
include/cffm_hip.h) walks the table with one workgroup per chunk.

* The step count lives on the device (a 4-float state tensor per parameter group, advanced by a 1-thread kernel in front
  of the update), so ``torch.cuda.graph`` can capture forward + backward + this step and every replay uses the right
  bias correction.  ``state[p]['step']`` is the host-side mirror (it does not advance during graph replays).
* When all gradients of a group alias one buffer -- what ``_LayerFn.backward`` produces -- the table stores their byte
  offsets inside it and the buffer's address is a kernel argument: the table is built once, wherever the allocator puts
  the gradients.  Otherwise (DistributedDataParallel bucket views, foreign gradients) it stores absolute addresses and
  is rebuilt only when one of them changes.

No CPU path: the moments and the table live on the parameters' device and the step fails loudly without the HIP library.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

CHUNK = 2048   # CFFM_ADAMW_CHUNK


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=6e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError('invalid AdamW hyper-parameters')
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._tables = {}      # group index -> {key: device table}; a few alternating address sets are kept
        self._dev_state = {}   # group index -> float32[4] on the device: step count + bias-correction factors

    def _moments(self, p):
        st = self.state[p]
        if not st:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.CffmError('AdamW: parameters must be contiguous float32')
            st['step'] = 0
            st['exp_avg'] = torch.zeros_like(p)
            st['exp_avg_sq'] = torch.zeros_like(p)
        return st

    def _table(self, gi, ps):
        """-> (device table, gradient base address or 0)."""
        grads = [p.grad for p in ps]
        for p, g in zip(ps, grads):
            if g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device or g.is_sparse:
                raise _lib.CffmError('AdamW: gradients must be dense contiguous float32 on the parameter device')
        base = grads[0].untyped_storage().data_ptr()
        shared = all(g.untyped_storage().data_ptr() == base for g in grads)
        gaddr = [g.data_ptr() - base for g in grads] if shared else [g.data_ptr() for g in grads]
        key = (shared,) + tuple(p.data_ptr() for p in ps) + tuple(gaddr)
        cache = self._tables.setdefault(gi, {})
        tab = cache.get(key)
        if tab is None:
            rows = []
            for p, ga in zip(ps, gaddr):
                st = self.state[p]
                n = p.numel()
                off = np.arange(0, n, CHUNK, dtype=np.int64)
                r = np.empty((off.size, 5), dtype=np.int64)
                r[:, 0] = p.data_ptr() + 4 * off
                r[:, 1] = ga + 4 * off
                r[:, 2] = st['exp_avg'].data_ptr() + 4 * off
                r[:, 3] = st['exp_avg_sq'].data_ptr() + 4 * off
                r[:, 4] = np.minimum(CHUNK, n - off)
                rows.append(r)
            tab = torch.from_numpy(np.concatenate(rows)).to(ps[0].device)
            if len(cache) >= 4:
                cache.clear()
            cache[key] = tab
        return tab, (base if shared else 0)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.get()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group['params'] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            if _lib._override is None and dev.type != 'cuda':
                raise _lib.CffmError('AdamW: parameters are on %s; the update kernel runs only on the GPU (no CPU fallback)' % dev)
            steps = set()
            state = self.state
            for p in ps:
                st = state.get(p)
                if not st:
                    if p.device != dev:
                        raise _lib.CffmError('AdamW: one device per parameter group')
                    st = self._moments(p)
                st['step'] += 1
                steps.add(st['step'])
            if len(steps) != 1:   # a parameter joined late: its bias correction would differ
                raise _lib.CffmError('AdamW: parameters of a group must have taken the same number of steps')
            ds = self._dev_state.get(gi)
            if ds is None:        # first step of the group: the device-side count starts where the host-side one is
                ds = torch.zeros(4, dtype=torch.float32, device=dev)
                ds[0] = float(steps.pop() - 1)
                self._dev_state[gi] = ds
            tab, gbase = self._table(gi, ps)
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == 'cuda' else C.c_void_p(0)
            b1, b2 = group['betas']
            _lib.check(lib.cffm_adamw_step_dev(C.c_void_p(tab.data_ptr()), tab.shape[0], C.c_void_p(gbase), group['lr'], b1, b2,
                                               group['eps'], group['weight_decay'], C.c_void_p(ds.data_ptr()), stream), lib)
        return loss

    def device_step_count(self, group=0):
        """The step count the update kernel has reached (differs from state[p]['step'] after graph replays)."""
        ds = self._dev_state.get(group)
        return 0 if ds is None else int(ds[0].item())
