"""Data-parallel glue for the hot path: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm).

The reference trains under MMDistributedDataParallel (mmseg/apis/train.py:57-65): parameters broadcast from rank 0 once,
gradients averaged over the ranks every step, the exchange overlapped with the backward pass by the bucketed reducer.
`_LayerFn.backward` produces every parameter gradient of the layer in ONE buffer laid out block by block (the .grad tensors
are views of it) and runs the blocks last-to-first, so the exchange here is one all-reduce PER BLOCK, started as soon as that
block's kernels are enqueued and travelling over xGMI while the previous block's backward runs (`BlockwiseReducer`) -- no
per-parameter reducer hooks, no 52 copies into a bucket, no gradient-ready bookkeeping.  `allreduce_gradients` is the
one-collective form (after the backward).  Clips shard over the ranks; the forward / backward kernels themselves have no
collective.
"""
import torch
import torch.distributed as dist


def broadcast_parameters(module, src=0):
    """Make every rank start from rank `src`'s parameters and buffers (what DistributedDataParallel does at construction)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src)


def allreduce_gradients(params, average=True, grads=None):
    """Average (or sum) the gradients of `params` over the ranks, in place: one collective per distinct underlying
    gradient buffer (a layer's gradients share one), plus one per gradient that owns its storage.
    `grads`: reduce these tensors instead of the parameters' current `.grad` -- for a step replayed from a HIP graph the
    gradients live in the buffers the CAPTURE allocated, whatever `.grad` points to by now."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    grads = [g for g in grads if g is not None] if grads is not None else [p.grad for p in params if p.grad is not None]
    if not grads:
        return 0
    bases = None
    # fast path: every gradient is a dense slice of the FIRST one's storage (autograd hands the layer's gradient views over as
    # aliases of the one buffer `_LayerFn.backward` allocated) -> one flat span of that storage, found with address
    # arithmetic only
    g0 = grads[0]
    st = g0.untyped_storage()
    s_lo, s_hi = st.data_ptr(), st.data_ptr() + st.nbytes()
    lo, hi, ok = s_hi, s_lo, True
    for g in grads:
        a = g.data_ptr()
        e = a + 4 * g.numel()
        if g.dtype != torch.float32 or not g.is_contiguous() or a < s_lo or e > s_hi:
            ok = False
            break
        lo, hi = min(lo, a), max(hi, e)
    if ok:
        bases = [torch.empty(0, dtype=torch.float32, device=g0.device).set_(st, (lo - s_lo) // 4, ((hi - lo) // 4,))]
    else:   # general case: per storage, merge only ADJACENT or overlapping gradient ranges (a gap of <= 3 elements counts as
        #         adjacent: 16-byte alignment padding, which `_LayerFn.backward` zero-fills) and reduce every merged range on its
        #         own -- other data living between two gradient views of one storage is never touched
        spans, bases = {}, []
        for g in grads:
            if not g.is_contiguous():
                bases.append(g)
                continue
            gst = g.untyped_storage()
            spans.setdefault((gst.data_ptr(), g.dtype), [gst, g, []])[2].append((g.storage_offset(), g.storage_offset() + g.numel()))
        for gst, g, ranges in spans.values():
            ranges.sort()
            lo, hi = ranges[0]
            merged = []
            for l, h in ranges[1:]:
                if l <= hi + 3:
                    hi = max(hi, h)
                else:
                    merged.append((lo, hi))
                    lo, hi = l, h
            merged.append((lo, hi))
            for l, h in merged:
                bases.append(torch.empty(0, dtype=g.dtype, device=g.device).set_(gst, l, (h - l,)))
    avg_op = getattr(dist.ReduceOp, 'AVG', None) if (average and dist.get_backend() == 'nccl') else None
    for t in bases:
        if avg_op is not None:
            dist.all_reduce(t, op=avg_op)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            if average:
                t.div_(world)
    return len(bases)


class BlockwiseReducer:
    """Gradient exchange overlapped with the backward pass, block by block.

    ``_LayerFn.backward`` runs the layer's blocks last-to-first and lays every block's parameter gradients out as one
    contiguous slice of the flat gradient buffer.  With this reducer installed it hands each slice over as soon as that block's
    kernels are enqueued; the reducer starts an asynchronous all-reduce of the slice (RCCL runs it on its own stream, ordered
    after the compute stream's position at that moment), so block i's exchange travels over xGMI while block i - 1's backward
    kernels run -- what the reference gets from DistributedDataParallel's bucketed reducer (mmseg/apis/train.py:57-65), without
    per-parameter hooks or bucket copies.  ``finish()`` (before the optimizer step) makes the compute stream wait for every
    pending exchange and divides by the world size where the backend has no averaging reduction.

        red = BlockwiseReducer(); red.install(layer.parameters())
        loss.backward(); red.finish(); optimizer.step()
    """

    def __init__(self, average=True, on_block=None, single_rank_too=False, params=None):
        # single_rank_too: issue the collectives even in a one-rank group (exercises RCCL itself on a one-GPU box)
        # params: the parameters whose gradients the layer's backward produces (see install)
        self.average, self.on_block, self.single_rank_too = average, on_block, single_rank_too
        self.params = list(params) if params is not None else None
        self.pending, self.log = [], []
        self.accumulating = False      # this backward adds into pre-existing .grad tensors: nothing is exchanged block by block
        self.fallbacks = 0
        self.drained = 0               # exchanges completed early (a second backward arrived before finish())

    def install(self, params=None):
        """``params``: the layer's parameters.  With them the reducer notices by itself -- at the first block of every backward --
        that ``.grad`` tensors already exist (``zero_grad(set_to_none=False)``, gradient accumulation), and ``finish()`` always checks
        that what was exchanged really is the parameters' gradients (the check used to be opt-in)."""
        from . import ops
        if params is not None:
            self.params = list(params)
        ops.block_grad_hook = self._hook
        return self

    def remove(self):
        from . import ops
        if ops.block_grad_hook == self._hook:
            ops.block_grad_hook = None

    def start(self, block, flat_slice, depth=0):
        """Start the exchange of one block's gradient slice (what the installed hook does; callable directly by a loop that
        drives the backward pieces itself, e.g. from HIP graphs)."""
        self._hook(block, flat_slice, depth)

    def _hook(self, block, flat_slice, depth):
        self.log.append(block)
        if self.on_block is not None:
            self.on_block(block, flat_slice, depth)
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not self.single_rank_too):
            return
        # First block of a backward: blocks come last-to-first, so it is block depth - 1 (callers of start() that do not pass the depth
        # are recognised by the empty queue, as before).  The hook runs INSIDE the layer's autograd node, so .grad still shows the state
        # before this backward: with gradients already there autograd will ADD the flat views into them once the node returns --
        # nothing is exchanged block by block then, finish() exchanges the real .grad tensors once.  That also covers the SECOND
        # backward of a gradient-accumulation step after zero_grad(set_to_none=True): the first one's slices were adopted as .grad and
        # may still be mid all-reduce, so their exchange is completed here, before autograd adds into that memory (the
        # detection used to be keyed to `not self.pending` and skipped exactly this case).  Averaged-then-accumulated gradients stay
        # correct under finish()'s second average: avg_r(avg(g1) + g2_r) = avg(g1) + avg(g2).
        # "Already there" is asked of the parameters of the layer whose backward is starting (ops.current_backward_params) -- with one
        # reducer serving two CFFM layers the second layer's first block otherwise saw the gradients the FIRST layer's backward had just
        # adopted and fell back to the non-overlapped exchange on every step; direct callers of start() fall back to every
        # installed parameter.
        first = (block == depth - 1) if depth else (not self.pending)
        if self.params is not None and first and not self.accumulating:
            from . import ops
            scope = ops.current_backward_params if ops.current_backward_params is not None else self.params
            self.accumulating = any(p.grad is not None for p in scope)
            if self.accumulating and self.pending:
                self._drain()
        if self.accumulating:
            return
        avg = self.average and dist.get_backend() == 'nccl' and hasattr(dist.ReduceOp, 'AVG')
        work = dist.all_reduce(flat_slice, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, async_op=True)
        self.pending.append((work, flat_slice, self.average and not avg))

    def _drain(self):
        """Complete every pending exchange (stream-side wait for RCCL); -> the address spans that were exchanged."""
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        spans = []
        for work, t, divide in self.pending:
            work.wait()
            if divide:
                t.div_(world)
            spans.append((t.data_ptr(), t.data_ptr() + 4 * t.numel()))
        self.drained += len(self.pending)
        self.pending = []
        return spans

    def finish(self, params=None, on_mismatch='raise'):
        """Wait (stream-side for RCCL) for every exchange started since the last call; returns how many there were.

        The slices that were exchanged are pieces of the backward's own flat buffer.  They ARE the parameters' gradients only when
        autograd adopted them as ``.grad`` -- i.e. when ``.grad`` was None before the backward (``zero_grad(set_to_none=True)``, the
        torch default).  With the parameters known (``install(params)`` or the argument) this is checked on every call: a backward
        that found gradients in place exchanged nothing block by block and its real ``.grad`` tensors are exchanged here
        (`allreduce_gradients`: correct, not overlapped); a ``.grad`` that does not lie inside an exchanged slice although slices were
        exchanged raises (``on_mismatch='fallback'``: exchange those gradients now instead).  Without parameters nothing can be
        checked -- pass them."""
        params = list(params) if params is not None else self.params
        spans = self._drain()
        n, self.drained = self.drained, 0
        if self.accumulating:
            self.accumulating = False
            self.fallbacks += 1
            allreduce_gradients([p for p in params if p.grad is not None], average=self.average)
            return n
        if params is not None and n:
            stray = [p for p in params if p.grad is not None and
                     not any(lo <= p.grad.data_ptr() and p.grad.data_ptr() + 4 * p.grad.numel() <= hi for lo, hi in spans)]
            if stray:
                if on_mismatch == 'raise':
                    raise RuntimeError('BlockwiseReducer: %d gradients do not alias the exchanged buffer (the parameters had .grad set '
                                       'before the backward: use zero_grad(set_to_none=True), or install(params) so that the reducer '
                                       'sees it coming)' % len(stray))
                self.fallbacks += 1
                allreduce_gradients(stray, average=self.average)
        return n
