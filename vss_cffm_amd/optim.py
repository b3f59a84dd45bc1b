"""AdamW for the parameters of the CFFM hot path: one HIP launch per step for EVERY parameter group, capturable in a HIP graph.

The reference trains the head with ``torch.optim.AdamW`` (lr 6e-5, betas (0.9, 0.999), weight decay 0.01) under a
``paramwise_cfg`` (`head` lr_mult 10, `norm` / `pos_block` decay_mult 0) and a poly schedule with linear warm-up
(local_configs/cffm/B1/cffm.b1.480x480.vspw2.160k.py:35-45).  The update rule here is that optimizer's (decoupled weight
decay, bias-corrected moments, amsgrad off); what differs is the launch shape: every parameter tensor of every group is cut
into 2048-element chunks listed in one device table, and a single kernel (``cffm_adamw_step_rows``, include/cffm_hip.h)
walks the table with one workgroup per chunk.  A chunk names the ROW of three small device tables it is updated with:

* ``consts`` (beta1, beta2, eps) -- written when the row is made;
* ``sched`` (lr, weight decay) -- a pinned host mirror that ``step()`` fills from ``param_groups``.  Eager steps copy it to
  the device on the current stream; a step captured in a HIP graph reads the mirror itself when a replay executes it
  (``zero_copy_sched``; False: a memcpy node of the graph): a host-side LR scheduler changes ``param_groups[i]['lr']`` and
  calls ``refresh_hyper()`` (a host write, no launch) between replays; the reference's own schedule (poly decay + linear
  warm-up) can instead be evaluated on the device from the step count (``set_poly_schedule``): no host write, nothing to order;
* ``state`` (step count t and the factors derived from it) -- advanced on the device inside the update launch (``fused_tick``:
  the last workgroup to finish stores it; False: a one-wave kernel in front of the update), so every replay of a captured step
  uses the right bias correction.  ``state[p]['step']`` is the host-side
  mirror; it does not advance during graph replays and is re-read from the device by ``state_dict()``.

A row is a (parameter group, step count) pair: parameters that first receive a gradient later than the rest of their group
(un-freezing, conditionally used branches) get a row -- and a bias correction -- of their own, as torch's per-parameter
step does; a row advances its step count only in the steps in which one of its parameters has a gradient (the chunk table
carries a flag per row), so intermittently used parameters keep torch's bias correction and schedule iteration.  When all gradients alias one buffer -- what ``_LayerFn.backward`` produces -- the table stores their byte
offsets inside it and the buffer's address is a kernel argument: the table is built once, wherever the allocator puts the
gradients.  Otherwise (DistributedDataParallel bucket views, foreign gradients) it stores absolute addresses and is
rebuilt only when one of them changes.  The cache key covers every address the table holds (parameters, gradients, both
moments), and ``load_state_dict`` / ``add_param_group`` drop every cached table and the device rows.

No CPU path: the moments and the tables live on the parameters' device and the step fails loudly without the HIP library.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib

CHUNK = 2048   # CFFM_ADAMW_CHUNK
NCONST = 12    # doubles per row of the consts table (include/cffm_hip.h)


class _DeviceRows:
    """The hyper-parameter rows of one device: (group index, step count) cohorts and their three tables."""

    def __init__(self, device):
        self.device = device
        self.rows = []            # [group index, host-side step count]
        self.consts = self.sched = self.state = self.sched_host = None
        self.sched_sent = None    # what the device copy of `sched` holds (None: unknown)
        self.tables = {}          # key -> device chunk table
        self.schedule = None      # device-side schedule fields (consts[3:10]) given to rows made later
        self.inflight = []        # pinned staging buffers of eager-mode `sched` copies that may still be queued
        self.members = {}         # row -> ids of the parameters updated with it
        self.gstep = 0            # host mirror of the global iteration count (consts[:, 10]; the device's runs ahead after graph replays)
        self.captured = False     # a step was recorded into a HIP graph: that graph holds the addresses of the current tables
        self.retired = []         # tables a captured graph may still point at, kept alive (see _retire)
        self.ticket = torch.zeros(65, dtype=torch.int32, device=device)   # workgroup tickets of the fused tick + update launch (CFFM_ADAMW_TICKETS)

    def _retire(self):
        """Called before the tables are replaced.  A HIP graph captured earlier holds their raw addresses: it must be captured again to
        see the new rows.  Replaying the OLD graph anyway would write through dangling pointers, so the superseded tables are kept
        alive -- such a replay then updates stale rows (wrong, but confined to memory this optimizer owns) -- and a RuntimeWarning says so."""
        if self.captured and self.state is not None:
            import warnings
            self.retired.append((self.consts, self.state, self.sched, self.sched_host, dict(self.tables)))
            warnings.warn('vss_cffm_amd.optim.AdamW: the device tables of a step that was captured in a HIP graph are being reallocated '
                          '(new parameter row); graphs captured before this point must be captured again', RuntimeWarning, stacklevel=3)
            self.captured = False

    def row_for(self, gi, step, group):
        for r, (g, t) in enumerate(self.rows):
            if g == gi and t == step:
                return r
        self._retire()
        self.sync_host()
        old_state = self.state
        self.rows.append([gi, step])
        n = len(self.rows)
        dev = self.device
        b1, b2 = group['betas']
        consts = torch.zeros(n, NCONST, dtype=torch.float64)
        state = torch.zeros(n, 4, dtype=torch.float32)
        if old_state is not None:
            consts[:n - 1] = self.consts.cpu()
            state[:n - 1] = old_state.cpu()
        consts[n - 1, 0], consts[n - 1, 1], consts[n - 1, 2] = b1, b2, group['eps']
        if self.schedule is not None:
            consts[n - 1, 3:10] = torch.tensor(list(self.schedule), dtype=torch.float64)
        consts[n - 1, 10] = float(self.gstep)     # the optimizer's iteration, not the row's: a row made late joins the schedule where it is
        state[n - 1, 0] = float(step)
        self.consts, self.state = consts.to(dev), state.to(dev)
        self.sched = torch.zeros(n, 2, dtype=torch.float32, device=dev)
        self.sched_host = torch.zeros(n, 2, dtype=torch.float32, pin_memory=(dev.type == 'cuda'))
        self.sched_sent = None
        self.tables.clear()
        return n - 1

    def split_row(self, r):
        """A copy of row r (same group, same step count, same constants), made ON THE DEVICE (no host sync: after graph replays
        only the device knows the count): the parameters of r that sit out a step move there before r advances."""
        self._retire()
        self.rows.append(list(self.rows[r]))
        dev = self.device
        self.consts = torch.cat([self.consts, self.consts[r:r + 1]])
        self.state = torch.cat([self.state, self.state[r:r + 1]])
        n = len(self.rows)
        self.sched = torch.zeros(n, 2, dtype=torch.float32, device=dev)
        self.sched_host = torch.zeros(n, 2, dtype=torch.float32, pin_memory=(dev.type == 'cuda'))
        self.sched_sent = None
        self.tables.clear()
        return n - 1

    def sync_host(self):
        """host-side step counts <- device (they differ after graph replays)."""
        if self.state is not None:
            t = self.state[:, 0].cpu()
            for r in range(len(self.rows)):
                self.rows[r][1] = int(round(float(t[r])))
            self.gstep = int(round(float(self.consts[0, 10].item())))


class AdamW(torch.optim.Optimizer):
    # step count advanced inside the update launch (include/cffm_hip.h, cffm_adamw_step_rows `ticket`); CFFM_ADAMW_FUSED=0: two launches
    fused_tick = os.environ.get('CFFM_ADAMW_FUSED', '1') != '0'
    # captured steps read (lr, weight decay) from the pinned host mirror directly instead of through a memcpy node (CFFM_ADAMW_ZEROCOPY=0)
    zero_copy_sched = os.environ.get('CFFM_ADAMW_ZEROCOPY', '1') != '0'

    def __init__(self, params, lr=6e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError('invalid AdamW hyper-parameters')
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._pending_schedule = None
        self._reset_device_side()

    # ------------------------------------------------------------------ cache control
    def _reset_device_side(self):
        """Drops the device rows and tables.  The optimizer's GLOBAL iteration (slot 10 of the constants, what the device-side
        schedule counts from) survives: the rows made afterwards start from it, as mmcv's runner.iter survives a resume."""
        carry = getattr(self, '_gstep0', 0)
        for dr in getattr(self, '_devs', {}).values():
            dr.sync_host()
            carry = max(carry, dr.gstep)
        self._gstep0 = carry
        self._devs = {}       # device -> _DeviceRows
        self._row_of = {}     # id(parameter) -> (device, row)

    def global_step(self):
        """The number of step() calls / replays so far (device rows consulted: it runs ahead of the host after graph replays)."""
        g = getattr(self, '_gstep0', 0)
        for dr in self._devs.values():
            dr.sync_host()
            g = max(g, dr.gstep)
        return g

    def load_state_dict(self, state_dict):
        """The loaded moments are new tensors and the loaded step counts seed new device rows: every cached chunk table
        (it holds raw addresses of the OLD moments) and the device-side step counts are dropped.  The global iteration is restored
        from the dict's `cffm_global_step` (written by state_dict()); a dict from torch.optim.AdamW has none: the largest per-parameter
        step count stands in (equal to it unless parameters sat steps out)."""
        super().load_state_dict(state_dict)
        self._devs = {}
        self._gstep0 = 0
        self._reset_device_side()
        g = state_dict.get('cffm_global_step')
        if g is None:
            g = 0
            for st in self.state.values():
                t = st.get('step', 0)
                g = max(g, int(t.item()) if torch.is_tensor(t) else int(t))
        self._gstep0 = int(g)

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if hasattr(self, '_devs'):
            self._sync_steps()
            self._reset_device_side()

    def __setstate__(self, state):
        super().__setstate__(state)
        self._reset_device_side()

    def _sync_steps(self):
        """state[p]['step'] <- the device-side step count of p's row."""
        for dr in self._devs.values():
            dr.sync_host()
        for group in self.param_groups:
            for p in group['params']:
                loc = self._row_of.get(id(p))
                if loc is not None and self.state.get(p):
                    self.state[p]['step'] = self._devs[loc[0]].rows[loc[1]][1]

    def state_dict(self):
        self._sync_steps()
        sd = super().state_dict()
        sd['cffm_global_step'] = self.global_step()     # (torch's own load_state_dict ignores unknown top-level keys)
        return sd

    # ------------------------------------------------------------------ helpers
    def _moments(self, p):
        st = self.state[p]
        if not st:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.CffmError('AdamW: parameters must be contiguous float32')
            st['step'] = 0
            st['exp_avg'] = torch.zeros_like(p)
            st['exp_avg_sq'] = torch.zeros_like(p)
        else:
            if torch.is_tensor(st['step']):      # a state dict written by torch.optim.AdamW keeps the step as a tensor
                st['step'] = int(st['step'].item())
            for k in ('exp_avg', 'exp_avg_sq'):
                m = st[k]
                if m.dtype != torch.float32 or m.device != p.device or not m.is_contiguous() or m.shape != p.shape:
                    st[k] = m.to(device=p.device, dtype=torch.float32).contiguous().view_as(p).clone()
        return st

    def _table(self, dr, ps, rows):
        """-> (device chunk table, device flags of the rows it references, gradient base address or 0)."""
        grads = [p.grad for p in ps]
        for p, g in zip(ps, grads):
            if g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device or g.is_sparse:
                raise _lib.CffmError('AdamW: gradients must be dense contiguous float32 on the parameter device')
        base = grads[0].untyped_storage().data_ptr()
        shared = all(g.untyped_storage().data_ptr() == base for g in grads)
        gaddr = [g.data_ptr() - base for g in grads] if shared else [g.data_ptr() for g in grads]
        ms = [self.state[p]['exp_avg'] for p in ps]
        vs = [self.state[p]['exp_avg_sq'] for p in ps]
        key = (shared,) + tuple(p.data_ptr() for p in ps) + tuple(gaddr) + tuple(m.data_ptr() for m in ms) + \
            tuple(v.data_ptr() for v in vs) + tuple(rows)
        tab = dr.tables.get(key)
        if tab is None:
            out = []
            for p, ga, m, v, row in zip(ps, gaddr, ms, vs, rows):
                n = p.numel()
                off = np.arange(0, n, CHUNK, dtype=np.int64)
                r = np.empty((off.size, 5), dtype=np.int64)
                r[:, 0] = p.data_ptr() + 4 * off
                r[:, 1] = ga + 4 * off
                r[:, 2] = m.data_ptr() + 4 * off
                r[:, 3] = v.data_ptr() + 4 * off
                r[:, 4] = np.minimum(CHUNK, n - off) | (np.int64(row) << 32)     # {int n; int row;}
                out.append(r)
            act = np.zeros(len(dr.rows), dtype=np.int32)
            act[list(set(rows))] = 1              # only the rows of THIS table advance their step count (k_adamw_tick_rows)
            tab = (torch.from_numpy(np.concatenate(out)).to(dr.device), torch.from_numpy(act).to(dr.device))
            if len(dr.tables) >= 4:
                dr.tables.clear()
            dr.tables[key] = tab
        return tab[0], tab[1], (base if shared else 0)

    def refresh_hyper(self):
        """Write the groups' current lr / weight decay into the pinned host mirrors (no launch, no sync).  A captured
        training step copies the mirror to the device as its first node, so an arbitrary host-side LR scheduler only has to
        call this between replays -- AFTER the previous replay has consumed the mirror (e.g. behind an event recorded after
        it): the copy node reads the mirror when it executes, not when the replay is launched.  HARD PRECONDITION with the default
        zero-copy form (`zero_copy_sched`): the update launch's ~800 workgroups each read their row of the mirror at their own moment
        of the launch, so the mirror must not change while a replay that contains the step is in flight -- call this only after an
        event recorded behind the previous replay has completed, or set `AdamW.zero_copy_sched = False` (one memcpy node reads the
        mirror at a single point).  The reference's own schedule needs none of this: see set_poly_schedule (nothing on the host
        changes between replays).  Eager steps call it themselves."""
        for dr in self._devs.values():
            if dr.sched_host is not None:
                for r, (gi, _) in enumerate(dr.rows):
                    g = self.param_groups[gi]
                    dr.sched_host[r, 0], dr.sched_host[r, 1] = g['lr'], g['weight_decay']

    def set_poly_schedule(self, max_iters, power=1.0, min_lr=0.0, warmup_iters=0, warmup_ratio=1.0, first_step=None):
        """Evaluate the reference's learning-rate schedule on the device (mmcv ``PolyLrUpdaterHook`` with ``warmup='linear'``:
        local_configs/cffm/B1/cffm.b1.480x480.vspw2.160k.py:41-45) from the device-side step count: with iteration
        ``it = step - first_step``, ``lr = (base - min_lr) (1 - it / max_iters)^power + min_lr``, scaled by
        ``1 - (1 - it / warmup_iters)(1 - warmup_ratio)`` during warm-up; ``param_groups[i]['lr']`` is then the BASE rate.
        Replayed HIP graphs follow the schedule with no host write at all.  ``first_step``: the global step count at which iteration 0
        happens (default: the optimizer's next step).  ``set_poly_schedule(None)`` goes back to host-provided rates.
        Rows must exist (take one step first) -- or call it before the first step and it applies to the rows as they are made."""
        kind = 0.0 if max_iters is None else 1.0
        gstep = self.global_step()
        first = float(gstep if first_step is None else first_step)       # resolved NOW, to the optimizer's global step
        for dr in self._devs.values():
            if dr.consts is None:
                continue
            c = dr.consts.cpu()
            for r in range(len(dr.rows)):
                c[r, 3:10] = torch.tensor([kind, float(max_iters or 0), power, min_lr, float(warmup_iters), warmup_ratio, first], dtype=torch.float64)
            dr.consts.copy_(c)
        self._pending_schedule = None if max_iters is None else (kind, float(max_iters), power, min_lr, float(warmup_iters), warmup_ratio, first)
        for dr in self._devs.values():
            dr.schedule = self._pending_schedule

    # ------------------------------------------------------------------ the step
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.get()
        per_dev = {}
        for gi, group in enumerate(self.param_groups):
            for p in group['params']:
                if p.grad is None:
                    continue
                dev = p.device
                if _lib._override is None and dev.type != 'cuda':
                    raise _lib.CffmError('AdamW: parameters are on %s; the update kernel runs only on the GPU (no CPU fallback)' % dev)
                per_dev.setdefault(dev, []).append((gi, group, p))
        for dev, items in per_dev.items():
            dr = self._devs.get(dev)
            if dr is None:
                dr = self._devs[dev] = _DeviceRows(dev)
                dr.schedule = getattr(self, '_pending_schedule', None)
                dr.gstep = getattr(self, '_gstep0', 0)      # rows made after a reset / resume continue the global iteration
            fresh = [(gi, group, p) for gi, group, p in items if id(p) not in self._row_of]
            for gi, group, p in fresh:       # (rows are made before any table is looked up: a new row rebuilds the tables)
                st = self._moments(p)
                r = dr.row_for(gi, st['step'], group)
                self._row_of[id(p)] = (dev, r)
                dr.members.setdefault(r, set()).add(id(p))
            ps = [p for _, _, p in items]
            rows = [self._row_of[id(p)][1] for p in ps]
            # parameters that sit this step out must not advance with their row: they move to a copy of it first
            present = {}
            for p, r in zip(ps, rows):
                present.setdefault(r, set()).add(id(p))
            for r, here in present.items():
                absent = dr.members.get(r, set()) - here
                if absent:
                    nr = dr.split_row(r)
                    dr.members[r] = set(here)
                    dr.members[nr] = set(absent)
                    for pid in absent:
                        self._row_of[pid] = (dev, nr)
            for p in ps:
                self.state[p]['step'] += 1
            for r in set(rows):
                dr.rows[r][1] += 1
            dr.gstep += 1
            self.refresh_hyper()
            capturing = dev.type == 'cuda' and torch.cuda.is_current_stream_capturing()
            dr.captured = dr.captured or capturing
            now = dr.sched_host.clone() if not capturing else None
            sched_ptr = None
            if capturing and self.zero_copy_sched:
                # the update kernel reads the pinned mirror itself when a replay executes it (what the memcpy node below did, without
                # the node: 4 us of copy + two dependent-launch gaps on the tail of every step)
                sched_ptr = dr.sched_host.data_ptr()
                dr.sched_sent = None
            elif capturing:
                dr.sched.copy_(dr.sched_host, non_blocking=True)     # a memcpy node: replays re-read the mirror
                dr.sched_sent = None
            elif dr.sched_sent is None or not torch.equal(now, dr.sched_sent):
                # eager: the host runs ahead of the stream, so the copy reads a buffer of its OWN (the mirror is rewritten by the
                # next step's refresh_hyper while this copy may still be queued); buffers are dropped once their copy has run
                src = now.pin_memory() if dev.type == 'cuda' else now
                dr.sched.copy_(src, non_blocking=True)
                if dev.type == 'cuda':
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(dev))
                    dr.inflight = [(b, e) for b, e in dr.inflight if not e.query()] + [(src, ev)]
                dr.sched_sent = now
            tab, active, gbase = self._table(dr, ps, rows)
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == 'cuda' else C.c_void_p(0)
            _lib.check(lib.cffm_adamw_step_rows(C.c_void_p(tab.data_ptr()), tab.shape[0], C.c_void_p(gbase),
                                                C.c_void_p(dr.state.data_ptr()), C.c_void_p(sched_ptr or dr.sched.data_ptr()),
                                                C.c_void_p(dr.consts.data_ptr()), len(dr.rows), C.c_void_p(active.data_ptr()),
                                                C.c_void_p(dr.ticket.data_ptr()) if self.fused_tick else None, stream), lib)
        return loss

    def device_step_count(self, group=0):
        """The step count the update kernel has reached for (the oldest row of) a parameter group; differs from the host
        mirror state[p]['step'] after graph replays."""
        for dr in self._devs.values():
            for r, (gi, _) in enumerate(dr.rows):
                if gi == group:
                    return int(round(float(dr.state[r, 0].item())))
        return 0


def paramwise_groups(named_params, base_lr=6e-5, base_wd=0.01, custom_keys=None):
    """The parameter groups mmcv's DefaultOptimizerConstructor builds from the reference's ``paramwise_cfg``
    (local_configs/cffm/B1/cffm.b1.480x480.vspw2.160k.py:35-39): the FIRST key of `custom_keys` (sorted by name, then by
    length descending, as mmcv does) that is a substring of the parameter's name sets its lr_mult / decay_mult.
    Default keys are the reference's: pos_block decay 0, norm decay 0, head lr x10.  Parameters with equal (lr, wd) share a
    group.  `named_params`: iterable of (full name, parameter), e.g. ``segmentor.named_parameters()``."""
    if custom_keys is None:
        custom_keys = {'pos_block': dict(decay_mult=0.), 'norm': dict(decay_mult=0.), 'head': dict(lr_mult=10.)}
    keys = sorted(sorted(custom_keys.keys()), key=len, reverse=True)
    groups = {}
    for name, p in named_params:
        if not p.requires_grad:
            continue
        lr, wd = base_lr, base_wd
        for k in keys:
            if k in name:
                lr = base_lr * custom_keys[k].get('lr_mult', 1.)
                wd = base_wd * custom_keys[k].get('decay_mult', 1.)
                break
        groups.setdefault((lr, wd), []).append(p)
    return [dict(params=ps, lr=lr, weight_decay=wd) for (lr, wd), ps in groups.items()]
