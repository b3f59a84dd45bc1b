"""Data-parallel glue for the hot path: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm).

The reference trains under MMDistributedDataParallel (mmseg/apis/train.py:57-65): parameters broadcast from rank 0 once,
gradients averaged over the ranks every step.  `_LayerFn.backward` produces every parameter gradient of the layer in ONE
buffer (the .grad tensors are views of it), so the exchange step is ONE all-reduce of that buffer -- no per-parameter
reducer hooks, no 52 copies into a bucket, no gradient-ready bookkeeping -- issued right after backward (the layer's
backward is a single autograd node: there is nothing to overlap it with).  Clips shard over the ranks; the forward /
backward kernels themselves have no collective.
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


def allreduce_gradients(params, average=True):
    """Average (or sum) the gradients of `params` over the ranks, in place: one collective per distinct underlying
    gradient buffer (a layer's gradients share one), plus one per gradient that owns its storage."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    # gradients that share a storage (autograd hands the layer's gradient views over as aliases of the one buffer
    # `_LayerFn.backward` allocated) are reduced as one flat span of that storage
    spans, singles = {}, []
    for p in params:
        g = p.grad
        if g is None:
            continue
        if not g.is_contiguous():
            singles.append(g)
            continue
        st = g.untyped_storage()
        lo, hi = g.storage_offset(), g.storage_offset() + g.numel()
        key = (st.data_ptr(), g.dtype)
        if key in spans:
            spans[key][1] = min(spans[key][1], lo)
            spans[key][2] = max(spans[key][2], hi)
        else:
            spans[key] = [g, lo, hi]
    bases = list(singles)
    for g, lo, hi in spans.values():
        bases.append(torch.empty(0, dtype=g.dtype, device=g.device).set_(g.untyped_storage(), lo, (hi - lo,)))
    avg_op = getattr(dist.ReduceOp, 'AVG', None) if (average and dist.get_backend() == 'nccl') else None
    for t in bases:
        if avg_op is not None:
            dist.all_reduce(t, op=avg_op)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            if average:
                t.div_(world)
    return len(bases)
