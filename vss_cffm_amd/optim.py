"""AdamW for the parameters of the CFFM hot path: one HIP launch per step.

The reference trains the head with ``torch.optim.AdamW`` (lr 6e-5, betas (0.9, 0.999), weight decay 0.01:
local_configs/cffm/B1/cffm.b1.480x480.vspw2.160k.py:35).  The update rule here is that optimizer's (decoupled weight
decay, bias-corrected moments, amsgrad off); what differs is the launch shape: every parameter tensor of every block is
cut into 2048-element chunks listed in one device table, and a single kernel (``cffm_adamw_step``,
include/cffm_hip.h) walks the table with one workgroup per chunk.  The table is rebuilt only when a parameter's or a
gradient's address changes -- under DistributedDataParallel with ``gradient_as_bucket_view`` the gradients live in the
reducer's buckets and never move; without it the caching allocator hands the backward pass the same blocks every step.

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
        self._tables = {}   # group index -> {address key: device table}; a few alternating address sets are kept

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
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in ps)
        cache = self._tables.setdefault(gi, {})
        tab = cache.get(key)
        if tab is None:
            rows = []
            for p in ps:
                g, st = p.grad, self.state[p]
                if g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device or g.is_sparse:
                    raise _lib.CffmError('AdamW: gradients must be dense contiguous float32 on the parameter device')
                n = p.numel()
                off = np.arange(0, n, CHUNK, dtype=np.int64)
                r = np.empty((off.size, 5), dtype=np.int64)
                for c, t in enumerate((p, g, st['exp_avg'], st['exp_avg_sq'])):
                    r[:, c] = t.data_ptr() + 4 * off
                r[:, 4] = np.minimum(CHUNK, n - off)
                rows.append(r)
            host = torch.from_numpy(np.concatenate(rows))
            tab = host.to(ps[0].device)
            if len(cache) >= 4:
                cache.clear()
            cache[key] = tab
        return tab

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
            if len(steps) != 1:   # a parameter joined late: its bias correction differs -> one launch per step count
                raise _lib.CffmError('AdamW: parameters of a group must have taken the same number of steps')
            tab = self._table(gi, ps)
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == 'cuda' else C.c_void_p(0)
            b1, b2 = group['betas']
            _lib.check(lib.cffm_adamw_step(C.c_void_p(tab.data_ptr()), tab.shape[0], group['lr'], b1, b2, group['eps'],
                                           group['weight_decay'], steps.pop(), stream), lib)
        return loss
