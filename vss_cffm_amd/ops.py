"""Torch-facing operators of the CFFM hot path: thin plumbing over the C ABI (include/cffm_hip.h).

PyTorch supplies device memory, the current HIP stream and autograd bookkeeping; all arithmetic of
the hot path happens inside libcffm_hip.so.  Tensors must be fp32, contiguous and on the GPU; a CPU
tensor raises (there is no CPU fallback -- the reference's CPU path is the oracle under oracle/,
which this package never imports).
"""
import contextlib
import ctypes as C

import numpy as np
import torch

from . import _lib, geometry

# field order of one block's parameters as the autograd function receives them (state_dict keys of
# CffmTransformerBlock3d3) -> (struct field, index inside an array field)
BLOCK_PARAM_KEYS = (
    ('norm1.weight', 'norm1_w', None), ('norm1.bias', 'norm1_b', None),
    ('pool_layers.0.weight', 'pool_w', 0), ('pool_layers.0.bias', 'pool_b', 0),
    ('pool_layers_clips.0.weight', 'pool_w', 1), ('pool_layers_clips.0.bias', 'pool_b', 1),
    ('pool_layers_clips.1.weight', 'pool_w', 2), ('pool_layers_clips.1.bias', 'pool_b', 2),
    ('pool_layers_clips.2.weight', 'pool_w', 3), ('pool_layers_clips.2.bias', 'pool_b', 3),
    ('attn.relative_position_bias_table', 'rpb_own', None),
    ('attn.relative_position_bias_table_to_neighbors', 'rpb_ring', None),
    ('attn.relative_position_bias_table_to_windows.0', 'rpb_pool', 0),
    ('attn.relative_position_bias_table_to_windows_clips.0', 'rpb_pool', 1),
    ('attn.relative_position_bias_table_to_windows_clips.1', 'rpb_pool', 2),
    ('attn.relative_position_bias_table_to_windows_clips.2', 'rpb_pool', 3),
    ('attn.qkv.weight', 'qkv_w', None), ('attn.qkv.bias', 'qkv_b', None),
    ('attn.proj.weight', 'proj_w', None), ('attn.proj.bias', 'proj_b', None),
    ('norm2.weight', 'norm2_w', None), ('norm2.bias', 'norm2_b', None),
    ('mlp.fc1.weight', 'fc1_w', None), ('mlp.fc1.bias', 'fc1_b', None),
    ('mlp.fc2.weight', 'fc2_w', None), ('mlp.fc2.bias', 'fc2_b', None),
)
NPB = len(BLOCK_PARAM_KEYS)   # tensors per block


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream(t):
    if t.is_cuda:
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return C.c_void_p(0)


def _branch(lib, st, i):
    """library stream i (0..3) ordered behind everything issued on `st` so far (cffm_branch_begin); everything a branch touches must stay
    referenced until _join(lib, st)"""
    return C.c_void_p(lib.cffm_branch_begin(st, i))


def _mark(lib, st):
    """remember this point of `st` for _take: the caller launches its own (longest) chain first, the branches afterwards (under stream
    capture the first-launched dependant of a node keeps the node's hardware queue)"""
    lib.cffm_branch_mark(st)


def _take(lib, st, i):
    return C.c_void_p(lib.cffm_branch_take(st, i))


def _join(lib, st):
    lib.cffm_branch_join(st)


class _WGrad(C.Structure):
    """cffm_wgrad"""
    _fields_ = [('dy', C.c_void_p), ('x', C.c_void_p), ('dw', C.c_void_p), ('M', C.c_long), ('N', C.c_int), ('K', C.c_int)]


def _require_device(t, what):
    if _lib._override is None and not t.is_cuda:
        raise _lib.CffmError('%s: the CFFM hot path runs only on the GPU (got a %s tensor); there is no CPU '
                             'fallback' % (what, t.device))
    if t.dtype != torch.float32:
        raise _lib.CffmError('%s: fp32 expected, got %s' % (what, t.dtype))


def _require_int64(t, what):
    if _lib._override is None and not t.is_cuda:
        raise _lib.CffmError('%s: runs only on the GPU (got a %s tensor); there is no CPU fallback' % (what, t.device))
    if t.dtype != torch.int64:
        raise _lib.CffmError('%s: int64 expected, got %s' % (what, t.dtype))
    return t.contiguous()


_geom_cache = {}


@contextlib.contextmanager
def _padded_grad_slices(lib):
    """The gradients handed to the library inside this block are 16-byte-aligned slices of ONE flat buffer: the library zeroes the
    padding behind the odd-sized ones itself (include/cffm_hip.h cffm_grad_slices_padded).  Switched on only around these calls: a
    caller of the C ABI with gradient tensors of their own (tests/test_emu_kernels.py block API) must not have 3 floats written
    behind them."""
    lib.cffm_grad_slices_padded(1)
    try:
        yield
    finally:
        lib.cffm_grad_slices_padded(0)


def _checked_padded(lib, fn, *args):
    with _padded_grad_slices(lib):
        _lib.check(fn(*args), lib)


def make_geom(lib, b, h0, w0):
    key = (id(lib), b, h0, w0)
    g = _geom_cache.get(key)
    if g is None:
        g = _lib.Geom()
        _lib.check(lib.cffm_geom_init(C.byref(g), b, h0, w0), lib)
        _geom_cache[key] = g
    return g


_table_cache = {}


def device_tables(h0, w0, device):
    key = (h0, w0, str(device))
    if key not in _table_cache:
        ks, qd = geometry.tables(h0, w0)
        ip, ii = geometry.inverse_tables(h0, w0)
        _table_cache[key] = tuple(torch.from_numpy(np.array(a)).to(device) for a in (ks, qd, ip, ii))
    return _table_cache[key]


def block_ptrs(tensors):
    """26 tensors in BLOCK_PARAM_KEYS order -> a cffm_block_params / cffm_block_grads struct."""
    s = _lib.BlockPtrs()
    for t, (_, field, idx) in zip(tensors, BLOCK_PARAM_KEYS):
        if idx is None:
            setattr(s, field, t.data_ptr())
        else:
            getattr(s, field)[idx] = t.data_ptr()
    return s


_struct_cache = {}


def block_structs(tensors, depth):
    """(BlockPtrs * depth) for a flat tensor list; cached on the addresses (parameters do not move between steps)."""
    key = tuple(t.data_ptr() for t in tensors)
    st = _struct_cache.get(key)
    if st is None:
        if len(_struct_cache) > 64:
            _struct_cache.clear()
        st = (_lib.BlockPtrs * depth)(*[block_ptrs(tensors[i * NPB:(i + 1) * NPB]) for i in range(depth)])
        _struct_cache[key] = st
    return st


def block_ws_layout(lib, g):
    w = _lib.BlockWs()
    _lib.check(lib.cffm_block_ws_layout(C.byref(g), C.byref(w)), lib)
    return w


# Optional callable(block_index, flat_gradient_slice, depth), invoked by the layer's backward right after block `block_index`'s
# kernels have been enqueued on the current stream (blocks come last-to-first).  None: the whole backward is one library call.
block_grad_hook = None
# While a layer's backward hands its blocks to the hook: the parameter tensors of THAT layer (26 * depth, block-major).  A hook that
# serves several layers (one BlockwiseReducer for two CFFM layers) looks here to tell "this layer's .grad tensors exist already"
# (gradient accumulation: nothing may be exchanged block by block) from "another layer's backward has just adopted its gradients".
current_backward_params = None


class _LayerFn(torch.autograd.Function):
    """BasicLayer3d3.forward (cffm_transformer.py:917-927) as one custom op: x [B,T,256,H,W] and the
    26*depth block parameters -> the new target frame [B,256,H,W]."""

    @staticmethod
    def forward(ctx, x, depth, *params):
        lib = _lib.get()
        _require_device(x, 'cffm layer input')
        if x.dim() != 5 or x.shape[2] != 256:
            raise _lib.CffmError('expected x [B,T,256,H,W], got %s' % (tuple(x.shape),))
        if x.shape[1] != 4:
            # the reference indexes reference frames 0..2 and the target [-1] (cffm_transformer.py:780-792)
            raise IndexError('CFFM block needs T == 4 frames (3 reference + target), got T=%d' % x.shape[1])
        assert len(params) == NPB * depth
        b, _, _, h0, w0 = x.shape
        x = x.contiguous()
        for p in params:
            if not p.is_contiguous() or p.dtype != torch.float32:
                raise _lib.CffmError('cffm layer parameters must be contiguous float32')
        g = make_geom(lib, b, h0, w0)
        key_src, q_dst, inv_ptr, inv_idx = device_tables(h0, w0, x.device)
        saved = torch.empty(lib.cffm_layer_saved_floats(C.byref(g), depth), dtype=torch.float32, device=x.device)
        scratch = torch.empty(lib.cffm_layer_scratch_floats(C.byref(g)), dtype=torch.float32, device=x.device)
        y = torch.empty(b, 256, h0, w0, dtype=torch.float32, device=x.device)
        pstructs = block_structs(params, depth)
        _lib.check(lib.cffm_layer_forward(C.byref(g), depth, pstructs, _ptr(x), _ptr(y), _ptr(key_src), _ptr(q_dst),
                                          _ptr(saved), _ptr(scratch), _stream(x)), lib)
        ctx.depth, ctx.geom_args = depth, (b, h0, w0)
        ctx.save_for_backward(saved, key_src, q_dst, inv_ptr, inv_idx, *params)
        ctx.scratch = scratch
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.get()
        saved, key_src, q_dst, inv_ptr, inv_idx, *params = ctx.saved_tensors
        depth = ctx.depth
        b, h0, w0 = ctx.geom_args
        g = make_geom(lib, b, h0, w0)
        # dy is normally the last-frame slice of the [B,4,256,H,W] upstream gradient (the torch.cat in cffm_layer): dense inside
        # a clip, 4 images apart between clips -- handed over with its batch stride instead of being copied
        img = 256 * h0 * w0
        if dy.dim() == 4 and dy.stride()[1:] == (h0 * w0, w0, 1) and (b == 1 or dy.stride(0) >= img):
            dy_bs = dy.stride(0) if b > 1 else img
        else:
            dy, dy_bs = dy.contiguous(), img
        # one allocation for every parameter gradient (16-byte aligned slices), returned as views
        sizes = [(p.numel() + 3) // 4 * 4 for p in params]
        # (the <= 12-byte padding gaps between slices travel through the gradient all-reduce with them and must not carry NaN / Inf
        # bit patterns of recycled memory: the library zeroes them inside the backward of the pooling Linears -- the only tensors
        # with a gap behind them -- see cffm_grad_slices_padded in include/cffm_hip.h; a zero-fill of the whole 6.8 MB buffer was a
        # 7 us kernel on the chain in front of the backward, an index_fill_ of the gaps still a 4.7 us one)
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=dy.device)
        grads = [c[:p.numel()].view(p.shape) if c.numel() != p.numel() else c.view(p.shape)
                 for c, p in zip(flat.split(sizes), params)]
        dx = torch.empty(b, 4, 256, h0, w0, dtype=torch.float32, device=dy.device)
        pstructs = block_structs(params, depth)
        gstructs = block_structs(grads, depth)
        hook = block_grad_hook
        if hook is None:
            _checked_padded(lib, lib.cffm_layer_backward, C.byref(g), depth, pstructs, gstructs, _ptr(dy), dy_bs, _ptr(dx), _ptr(key_src),
                                               _ptr(q_dst), _ptr(inv_ptr), _ptr(inv_idx), _ptr(saved), _ptr(ctx.scratch),
                                               _stream(dy))
        else:
            # block by block (last block first, as the chain runs): once a block's kernels are enqueued its slice of the
            # flat gradient buffer is handed to the hook -- data-parallel training starts that block's all-reduce there,
            # overlapping it with the backward of the blocks still to come (vss_cffm_amd.distributed.BlockwiseReducer)
            per = sum(sizes[:NPB])
            global current_backward_params
            current_backward_params = params
            try:
                for i in range(depth - 1, -1, -1):
                    _checked_padded(lib, lib.cffm_layer_backward_range, C.byref(g), depth, pstructs, gstructs, _ptr(dy), dy_bs, _ptr(dx),
                                                             _ptr(key_src), _ptr(q_dst), _ptr(inv_ptr), _ptr(inv_idx), _ptr(saved),
                                                             _ptr(ctx.scratch), i, i, _stream(dy))
                    hook(i, flat[i * per:(i + 1) * per], depth)
            finally:
                current_backward_params = None
        return (dx, None) + tuple(grads)


class LayerPieces:
    """The layer's forward and backward as explicit pieces over STATIC buffers, without autograd: what a data-parallel training
    loop replays from HIP graphs when it wants the gradient exchange of block i to overlap the backward of block i - 1 --
    graph(forward + backward of the last blocks) | all-reduce(those blocks' gradient slices, asynchronous) | graph(backward of
    block 0) | all-reduce(block 0's slice) | graph(optimizer)  (bench.py, N > 1).  The same library calls, in the same order,
    as ``_LayerFn``; ``grads`` are views of one flat buffer (``flat``), ``block_slice(i)`` is block i's part of it."""

    def __init__(self, x, depth, params):
        lib = _lib.get()
        _require_device(x, 'cffm layer input')
        if x.dim() != 5 or x.shape[1] != 4 or x.shape[2] != 256:
            raise _lib.CffmError('expected x [B,4,256,H,W], got %s' % (tuple(x.shape),))
        self.lib, self.x, self.depth, self.params = lib, x.contiguous(), depth, list(params)
        b, _, _, h0, w0 = x.shape
        self.g = make_geom(lib, b, h0, w0)
        self.tables = device_tables(h0, w0, x.device)
        dev = x.device
        self.saved = torch.empty(lib.cffm_layer_saved_floats(C.byref(self.g), depth), dtype=torch.float32, device=dev)
        self.scratch = torch.empty(lib.cffm_layer_scratch_floats(C.byref(self.g)), dtype=torch.float32, device=dev)
        self.y = torch.empty(b, 256, h0, w0, dtype=torch.float32, device=dev)
        self.dx = torch.empty(b, 4, 256, h0, w0, dtype=torch.float32, device=dev)
        self.sizes = [(p.numel() + 3) // 4 * 4 for p in self.params]
        self.flat = torch.zeros(sum(self.sizes), dtype=torch.float32, device=dev)
        self.grads = [c[:p.numel()].view(p.shape) for c, p in zip(self.flat.split(self.sizes), self.params)]
        self.pstructs = block_structs(self.params, depth)
        self.gstructs = block_structs(self.grads, depth)
        self.per_block = sum(self.sizes[:NPB])

    def block_slice(self, i):
        return self.flat[i * self.per_block:(i + 1) * self.per_block]

    def forward(self):
        key_src, q_dst = self.tables[:2]
        _lib.check(self.lib.cffm_layer_forward(C.byref(self.g), self.depth, self.pstructs, _ptr(self.x), _ptr(self.y), _ptr(key_src),
                                               _ptr(q_dst), _ptr(self.saved), _ptr(self.scratch), _stream(self.x)), self.lib)
        return self.y

    def backward(self, dy, first_block, last_block):
        """blocks first_block .. last_block (descending) of the backward from dy [B,256,H,W] (a dense tensor or the last-frame
        slice of a [B,4,256,H,W] gradient)."""
        key_src, q_dst, inv_ptr, inv_idx = self.tables
        b, h0, w0 = self.x.shape[0], self.x.shape[3], self.x.shape[4]
        img = 256 * h0 * w0
        if not (dy.dim() == 4 and dy.stride()[1:] == (h0 * w0, w0, 1) and (b == 1 or dy.stride(0) >= img)):
            raise _lib.CffmError('LayerPieces.backward: dy must be dense inside a clip')
        dy_bs = dy.stride(0) if b > 1 else img
        _checked_padded(self.lib, self.lib.cffm_layer_backward_range, C.byref(self.g), self.depth, self.pstructs, self.gstructs, _ptr(dy), dy_bs,
                                                      _ptr(self.dx), _ptr(key_src), _ptr(q_dst), _ptr(inv_ptr), _ptr(inv_idx),
                                                      _ptr(self.saved), _ptr(self.scratch), first_block, last_block,
                                                      _stream(self.x))


class _LayerFullFn(torch.autograd.Function):
    """BasicLayer3d3.forward with the reference's whole output: x [B,4,256,H,W] -> [B,4,256,H,W] whose frames 0..2 are the input
    frames (cffm_transformer.py:826) and frame 3 is new.  The library writes the pass-through frames itself (side stream) and adds
    their upstream gradient into dx in its final layout pass, so neither torch.cat nor its backward (a zero-fill, a copy and an
    add over the 29 MB stack) runs around the call."""

    @staticmethod
    def forward(ctx, x, depth, *params):
        lib = _lib.get()
        _require_device(x, 'cffm layer input')
        if x.dim() != 5 or x.shape[2] != 256:
            raise _lib.CffmError('expected x [B,T,256,H,W], got %s' % (tuple(x.shape),))
        if x.shape[1] != 4:
            raise IndexError('CFFM block needs T == 4 frames (3 reference + target), got T=%d' % x.shape[1])
        assert len(params) == NPB * depth
        b, _, _, h0, w0 = x.shape
        x = x.contiguous()
        for p in params:
            if not p.is_contiguous() or p.dtype != torch.float32:
                raise _lib.CffmError('cffm layer parameters must be contiguous float32')
        g = make_geom(lib, b, h0, w0)
        key_src, q_dst, inv_ptr, inv_idx = device_tables(h0, w0, x.device)
        saved = torch.empty(lib.cffm_layer_saved_floats(C.byref(g), depth), dtype=torch.float32, device=x.device)
        scratch = torch.empty(lib.cffm_layer_scratch_floats(C.byref(g)), dtype=torch.float32, device=x.device)
        y = torch.empty(b, 4, 256, h0, w0, dtype=torch.float32, device=x.device)
        _lib.check(lib.cffm_layer_forward_full(C.byref(g), depth, block_structs(params, depth), _ptr(x), _ptr(y), _ptr(key_src), _ptr(q_dst),
                                               _ptr(saved), _ptr(scratch), _stream(x)), lib)
        ctx.depth, ctx.geom_args = depth, (b, h0, w0)
        ctx.save_for_backward(saved, key_src, q_dst, inv_ptr, inv_idx, *params)
        ctx.scratch = scratch
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.get()
        saved, key_src, q_dst, inv_ptr, inv_idx, *params = ctx.saved_tensors
        depth = ctx.depth
        b, h0, w0 = ctx.geom_args
        g = make_geom(lib, b, h0, w0)
        dy = dy.contiguous()
        sizes = [(p.numel() + 3) // 4 * 4 for p in params]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=dy.device)       # (the library zeroes the alignment gaps: see _LayerFn.backward)
        grads = [c[:p.numel()].view(p.shape) if c.numel() != p.numel() else c.view(p.shape)
                 for c, p in zip(flat.split(sizes), params)]
        dx = torch.empty(b, 4, 256, h0, w0, dtype=torch.float32, device=dy.device)
        pstructs, gstructs = block_structs(params, depth), block_structs(grads, depth)
        hook = block_grad_hook
        per = sum(sizes[:NPB])
        global current_backward_params
        current_backward_params = params
        try:
            for first, last in ([(depth - 1, 0)] if hook is None else [(i, i) for i in range(depth - 1, -1, -1)]):
                _checked_padded(lib, lib.cffm_layer_backward_full, C.byref(g), depth, pstructs, gstructs, _ptr(dy), _ptr(dx), _ptr(key_src), _ptr(q_dst),
                                                        _ptr(inv_ptr), _ptr(inv_idx), _ptr(saved), _ptr(ctx.scratch), first, last,
                                                        _stream(dy))
                if hook is not None:
                    hook(first, flat[first * per:(first + 1) * per], depth)
        finally:
            current_backward_params = None
        return (dx, None) + tuple(grads)


def cffm_layer(x, depth, params):
    """x [B,4,256,H,W]; params: flat list of 26*depth tensors (BLOCK_PARAM_KEYS order per block).
    Returns the reference's output [B,4,256,H,W]: frames 0..2 are the input, frame 3 is new."""
    return _LayerFullFn.apply(x, depth, *params)


def cffm_layer_target(x, depth, params):
    """the new target frame only, [B,256,H,W] (callers that use nothing else: cffm_head.py:145)"""
    return _LayerFn.apply(x, depth, *params)


# ---------------------------------------------------------------------------------------------- NCHW <-> token rows
def _to_rows(lib, x):
    """[N,C,H,W] (any strides) -> contiguous token rows [N,H,W,C].  Plain NCHW goes through the library's tiled transpose,
    channels-last memory is already rows.  (The operators RETURN channels-last views: handing BatchNorm / ReLU / dropout plain
    NCHW instead made the whole head step slower, 5.3 -> 6.4 ms.)"""
    n, c, h, w = x.shape
    if x.is_contiguous() and x.numel() and c > 1 and h * w > 1:
        rows = torch.empty(n, h, w, c, dtype=torch.float32, device=x.device)
        _lib.check(lib.cffm_transpose(_ptr(x), _ptr(rows), n, c, h * w, c * h * w, c * h * w, _stream(x)), lib)
        return rows
    return x.permute(0, 2, 3, 1).contiguous()


# ---------------------------------------------------------------------------------------------- SegFormer embedding
class _SegFuseFn(torch.autograd.Function):
    """sum_i resize_i(A_i c_i) + d on token rows: features c_i [N,C_i,h_i,w_i] (c_0 at the output resolution), composed
    matrices A_i [256,C_i], constant d [256] -> [N,256,H,W] (channels-last memory).  See csrc/segfuse_kernels.h."""

    @staticmethod
    def forward(ctx, d, *ca):
        lib = _lib.get()
        k = len(ca) // 2
        feats, mats = ca[:k], [a.contiguous() for a in ca[k:]]
        if not 1 <= k <= 4:
            raise _lib.CffmError('segformer_fuse: 1..4 feature maps expected, got %d' % k)
        for c in feats + tuple(mats) + (d,):
            _require_device(c, 'segformer_fuse operand')
        n, _, H, W = feats[0].shape
        st, dev = _stream(feats[0]), feats[0].device
        toks, zs, keep = [], [], []
        for c, a in zip(feats, mats):
            if c.dim() != 4 or c.shape[0] != n or a.shape != (256, c.shape[1]):
                raise _lib.CffmError('segformer_fuse: feature %s does not fit matrix %s' % (tuple(c.shape), tuple(a.shape)))
        # every scale is an independent chain NCHW -> token rows -> Linear: the 1/4-scale one (three quarters of the bytes) on the caller's
        # stream, the others on branches beside it (one stream: 107 us of a replayed head step for 64 us of the largest chain)
        _mark(lib, st)
        for i in range(k):
            c, a = feats[i].contiguous(), mats[i]
            ci, p = c.shape[1], c.shape[2] * c.shape[3]
            t = torch.empty(n * p, ci, dtype=torch.float32, device=dev)        # token rows [N*h*w, C_i]
            z = torch.empty(n * p, 256, dtype=torch.float32, device=dev)
            if n * p:
                s = _take(lib, st, i) if i else st
                _lib.check(lib.cffm_transpose(_ptr(c), _ptr(t), n, ci, p, ci * p, ci * p, s), lib)
                _lib.check(lib.cffm_linear_fwd(_ptr(t), _ptr(a), _ptr(z), n * p, 256, ci, s), lib)
            toks.append(t)
            zs.append(z)
            keep.append(c)
        _join(lib, st)
        hs = (C.c_int * 3)(*([c.shape[2] for c in feats[1:]] + [1] * (4 - k)))
        ws = (C.c_int * 3)(*([c.shape[3] for c in feats[1:]] + [1] * (4 - k)))
        zp = (C.c_void_p * 3)(*([z.data_ptr() for z in zs[1:]] + [None] * (4 - k)))
        y = zs[0]                                                             # updated in place: nothing else needs it
        if n * H * W:
            _lib.check(lib.cffm_segfuse_fwd(_ptr(y), _ptr(d.contiguous()), zp, hs, ws, k - 1, n, H, W, st), lib)
        ctx.save_for_backward(*toks, *mats)
        ctx.shapes = [tuple(c.shape) for c in feats]
        # channels-last memory: torch's BatchNorm runs faster on it than on plain NCHW (head step 4.6 vs 5.8 ms)
        return y.view(n, H, W, 256).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.get()
        shapes = ctx.shapes
        k = len(shapes)
        toks, mats = ctx.saved_tensors[:k], ctx.saved_tensors[k:]
        n, _, H, W = shapes[0]
        g = _to_rows(lib, g)                                                  # [N,H,W,256] rows
        st, dev = _stream(g), g.device
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        dd = new(256)
        rows = n * H * W
        dzs = [g.view(rows, 256)] + [new(n * s_[2] * s_[3], 256) for s_ in shapes[1:]]
        if k > 1 and rows:
            hs = (C.c_int * 3)(*([s_[2] for s_ in shapes[1:]] + [1] * (4 - k)))
            ws = (C.c_int * 3)(*([s_[3] for s_ in shapes[1:]] + [1] * (4 - k)))
            zp = (C.c_void_p * 3)(*([z.data_ptr() for z in dzs[1:]] + [None] * (4 - k)))
            _lib.check(lib.cffm_segfuse_bwd(_ptr(g), zp, hs, ws, k - 1, n, H, W, st), lib)
        # Behind the adjoint of the resizes everything is independent: branch 1 = the k weight gradients as ONE grouped launch and the
        # column sum of g (both use the library's scratch: one after the other on one branch); the input gradients (Linear + back to NCHW)
        # of the 1/4-scale map on the caller's stream, of the small maps on branches 2 / 3.  (One stream: 213 + 34 us of a replayed step.)
        dmats = [new(256, s_[1]) for s_ in shapes]
        live = [i for i, s_ in enumerate(shapes) if n * s_[2] * s_[3]]
        for i in range(k):
            if i not in live:
                dmats[i].zero_()
        _mark(lib, st)
        dfeats, keep = [None] * k, []

        def dx_chain(i, sx):
            s_, a, dz = shapes[i], mats[i], dzs[i]
            ci, p = s_[1], s_[2] * s_[3]
            if ctx.needs_input_grad[1 + i]:
                dt, dc = new(n * p, ci), new(*s_)
                if n * p:
                    _lib.check(lib.cffm_linear_bwd_input(_ptr(dz), _ptr(a), _ptr(dt), n * p, 256, ci, sx), lib)
                    _lib.check(lib.cffm_transpose(_ptr(dt), _ptr(dc), n, p, ci, ci * p, ci * p, sx), lib)
                dfeats[i] = dc
                keep.append(dt)
        dx_chain(0, st)                                            # the 1/4-scale chain first: it keeps the caller's stream (and its queue)
        if rows:
            # branch 1: the 1/4-scale weight gradient (three quarters of the rows); branch 2: the other scales' (one grouped call: the
            # library groups what its grouped kernel takes and runs the rest one by one) and the column sum of g; branch 3: the small
            # scales' input gradients.  (Weight gradients and column sums take their scratch from the branch's own pool.)
            def wgrads(sel, sx):
                sel = [i for i in sel if i in live]
                if sel:
                    pr = (_WGrad * len(sel))(*[_WGrad(dzs[i].data_ptr(), toks[i].data_ptr(), dmats[i].data_ptr(), n * shapes[i][2] * shapes[i][3], 256,
                                                      shapes[i][1]) for i in sel])
                    _lib.check(lib.cffm_linear_bwd_weight_group(pr, len(sel), sx), lib)
            wgrads([0], _take(lib, st, 1))
            s2 = _take(lib, st, 2)
            _lib.check(lib.cffm_colsum(_ptr(g), rows, 256, _ptr(dd), s2), lib)
            wgrads(range(1, k), s2)
        else:
            dd.zero_()
        if k > 1:
            s3 = _take(lib, st, 3)
            for i in range(1, k):
                dx_chain(i, s3)
        _join(lib, st)
        return (dd,) + tuple(dfeats) + tuple(dmats)


def _ptr_array(ts):
    """host array of the tensors' device pointers (NULL for None)"""
    return (C.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])


class _ComposeFn(torch.autograd.Function):
    """(A_1 .. A_k, d): A_i = Wf_i W_i for every scale (Wf_i = the i-th [256,256] input-channel block of the fuse weight, cat order
    c_k..c_1, read in place through its leading dimension) and d = sum_i Wf_i b_i, as ONE library call per direction
    (cffm_fuse_compose_fwd / _bwd: the k products side by side on branches of the library's streams).  (As torch matmuls these
    70-MFLOP products and their backward took 0.4 ms per step; round 5 ran them as 4 + 8 stage GEMMs one behind the other with two
    layout copies of the fuse weight and torch's cat / gemv / ger for the constant: 97 + 120 us of a replayed head step.)"""

    @staticmethod
    def forward(ctx, fuse_w2d, *wb):
        lib = _lib.get()
        k, e = len(wb) // 2, fuse_w2d.shape[0]
        fuse_w2d = fuse_w2d.contiguous()
        lin_w, lin_b = [w.contiguous() for w in wb[:k]], [b_.contiguous() for b_ in wb[k:]]
        dev = fuse_w2d.device
        mats = [torch.empty(e, w.shape[1], dtype=torch.float32, device=dev) for w in lin_w]
        d = torch.empty(e, dtype=torch.float32, device=dev)
        cin = (C.c_int * k)(*[w.shape[1] for w in lin_w])
        _lib.check(lib.cffm_fuse_compose_fwd(_ptr(fuse_w2d), _ptr_array(lin_w), _ptr_array(lin_b), cin, k, e, _ptr_array(mats), _ptr(d),
                                             _stream(fuse_w2d)), lib)
        ctx.save_for_backward(fuse_w2d, *lin_w, *lin_b)
        return tuple(mats) + (d,)

    @staticmethod
    def backward(ctx, *grads):
        lib = _lib.get()
        fuse_w2d = ctx.saved_tensors[0]
        k = (len(ctx.saved_tensors) - 1) // 2
        lin_w, lin_b = ctx.saved_tensors[1:1 + k], ctx.saved_tensors[1 + k:]
        e, dev = fuse_w2d.shape[0], fuse_w2d.device
        dmats = [torch.zeros(e, w.shape[1], dtype=torch.float32, device=dev) if g is None else g.contiguous() for g, w in zip(grads[:k], lin_w)]
        dd = torch.zeros(e, dtype=torch.float32, device=dev) if grads[k] is None else grads[k].contiguous()
        dfw = torch.empty_like(fuse_w2d)
        dws, dbs = [torch.empty_like(w) for w in lin_w], [torch.empty_like(b_) for b_ in lin_b]
        cin = (C.c_int * k)(*[w.shape[1] for w in lin_w])
        _lib.check(lib.cffm_fuse_compose_bwd(_ptr(fuse_w2d), _ptr_array(lin_w), _ptr_array(lin_b), cin, k, e, _ptr_array(dmats), _ptr(dd), _ptr(dfw),
                                             _ptr_array(dws), _ptr_array(dbs), _stream(fuse_w2d)), lib)
        return (dfw,) + tuple(dws) + tuple(dbs)


def segformer_fuse(feats, lin_w, lin_b, fuse_w):
    """The SegFormer embedding of the CFFM heads without the 1024-channel concat (cffm_head.py:102-119):
    conv1x1(cat([resize(linear_c4(c4)), resize(linear_c3(c3)), resize(linear_c2(c2)), linear_c1(c1)]), fuse_w).

    feats: [c1, c2, c3, c4] NCHW (c1 = the 1/4-scale map the others are resized to); lin_w / lin_b: the four `MLP.proj`
    weights [256,C_i] / biases in the same order; fuse_w: `linear_fuse.conv.weight` [256, 4*256, 1, 1] whose input-channel
    blocks are ordered c4, c3, c2, c1 (the reference's cat order).  Returns the pre-BatchNorm map [N,256,H,W].
    The composed matrices Wf_i W_i and the constant sum_i Wf_i b_i come from one library call (_ComposeFn); autograd returns the
    gradients of the nine original tensors."""
    k = len(feats)
    e = fuse_w.shape[0]
    if e != 256 or fuse_w.shape[1] != k * e:
        raise _lib.CffmError('segformer_fuse: fuse weight %s does not fit %d embeddings of 256' % (tuple(fuse_w.shape), k))
    for w in lin_w:
        _require_device(w, 'segformer_fuse weight')
        if w.shape[0] != e or w.shape[1] % 4:
            raise _lib.CffmError('segformer_fuse: embedding weight %s (rows of 16-byte multiples expected)' % (tuple(w.shape),))
    for b_ in lin_b:
        _require_device(b_, 'segformer_fuse bias')
        if b_.shape != (e,):
            raise _lib.CffmError('segformer_fuse: embedding bias %s, [%d] expected' % (tuple(b_.shape), e))
    *mats, d = _ComposeFn.apply(fuse_w.reshape(e, k * e), *lin_w, *lin_b)
    return _SegFuseFn.apply(d, *feats, *mats)


# ---------------------------------------------------------------------------------------------- 1x1 classifiers
class _Conv1x1Fn(torch.autograd.Function):
    """A 1x1 convolution as a Linear GEMM on channels-last token rows: x [N,C,H,W] (any strides; channels-last costs no copy),
    weight [O,C,1,1], bias [O] -> [N,O,H,W] in channels-last memory -- or, with `clips` = B, the same rows viewed as
    [B, N/B, O, H, W] (the head's per-frame logits): the gradient of that view may then arrive as B separately-placed blocks of
    rows (slices of a [B, T+1, H, W, O] buffer, the loss kernel's output) and is consumed block by block, not copied together."""

    @staticmethod
    def forward(ctx, x, weight, bias, clips, extra=0):
        lib = _lib.get()
        for t in (x, weight, bias):
            _require_device(t, 'conv1x1 operand')
        if x.dim() != 4 or weight.dim() != 4 or tuple(weight.shape[1:]) != (x.shape[1], 1, 1) or bias.shape != (weight.shape[0],):
            raise _lib.CffmError('conv1x1: x %s, weight %s, bias %s do not fit a 1x1 convolution'
                                 % (tuple(x.shape), tuple(weight.shape), tuple(bias.shape)))
        n, c, h, w = x.shape
        o = weight.shape[0]
        if c % 4 or o % 4:
            raise _lib.CffmError('conv1x1: channel counts must be multiples of 4 (16-byte token rows), got %d -> %d' % (c, o))
        if clips and n % clips:
            raise _lib.CffmError('conv1x1: %d maps do not split into %d clips' % (n, clips))
        rows = _to_rows(lib, x)                                   # [N,H,W,C] token rows
        wm, b = weight.reshape(o, c).contiguous(), bias.contiguous()
        ctx.save_for_backward(rows, wm)
        ctx.x_plain = x.is_contiguous()            # hand the input gradient back in the input's own memory layout
        ctx.clips = clips
        if clips and extra:
            # room for `extra` more maps per clip behind the n / clips this call writes (cat_into fills them): the head's [B, T+1, K, h, w]
            # logits are then assembled where they lie -- torch.cat copied the 57 MB of frame logits once more (30 us of a replayed step)
            t = n // clips
            buf = torch.empty(clips, t + extra, h, w, o, dtype=torch.float32, device=x.device)
            per = t * h * w
            for i in range(clips):
                if per:
                    _lib.check(lib.cffm_linear_bias_fwd(C.c_void_p(rows.data_ptr() + 4 * i * per * c), _ptr(wm), _ptr(b), _ptr(buf[i]), per, o, c, _stream(x)), lib)
            ctx.room = buf
            return buf[:, :t].permute(0, 1, 4, 2, 3)
        y = torch.empty(n, h, w, o, dtype=torch.float32, device=x.device)
        _lib.check(lib.cffm_linear_bias_fwd(_ptr(rows), _ptr(wm), _ptr(b), _ptr(y), n * h * w, o, c, _stream(x)), lib)
        if clips:
            return y.view(clips, n // clips, h, w, o).permute(0, 1, 4, 2, 3)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.get()
        rows, wm = ctx.saved_tensors
        n, h, w, c = rows.shape
        o = wm.shape[0]
        m, st = n * h * w, _stream(rows)
        # the gradient as blocks of token rows: [(rows tensor, first row, row count)]
        blocks = None
        if ctx.clips:
            d5 = dy.permute(0, 1, 3, 4, 2)                                   # [B,T,H,W,O]
            if d5.is_contiguous():
                blocks = [(d5, 0, m)]
            elif all(d5[i].is_contiguous() for i in range(ctx.clips)):
                per = m // ctx.clips
                blocks = [(d5[i], i * per, per) for i in range(ctx.clips)]
            else:
                dy = dy.reshape(n, o, h, w)
        if blocks is None:
            blocks = [(_to_rows(lib, dy.reshape(n, o, h, w) if dy.dim() == 5 else dy), 0, m)]
        dx = dwm = db = None
        live = [(blk, r0, nr) for blk, r0, nr in blocks if nr]
        # weight + bias gradient on a branch (ONE grouped launch over the blocks, then their column sums: both use the library's scratch, so
        # one after the other), the input gradient beside them on the caller's stream
        wparts, bparts = [], []
        _mark(lib, st)
        if ctx.needs_input_grad[0]:
            dx = torch.empty(n, h, w, c, dtype=torch.float32, device=dy.device)
            for blk, r0, nr in live:
                _lib.check(lib.cffm_linear_bwd_input(_ptr(blk), _ptr(wm), C.c_void_p(dx.data_ptr() + 4 * r0 * c), nr, o, c, st), lib)
            if ctx.x_plain and m:
                dxp = torch.empty(n, c, h, w, dtype=torch.float32, device=dy.device)
                _lib.check(lib.cffm_transpose(_ptr(dx), _ptr(dxp), n, h * w, c, c * h * w, c * h * w, st), lib)
                dx = dxp
            else:
                dx = dx.permute(0, 3, 1, 2)
        if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and live:
            # weight gradient of block j on branch 1 + j % 2, the column sums on branch 3 (each with the scratch pool of its branch)
            if ctx.needs_input_grad[1]:
                wparts = [torch.empty(o, c, dtype=torch.float32, device=dy.device) for _ in live]
                sw = [_take(lib, st, 1), _take(lib, st, 2)] if len(live) > 1 else [_take(lib, st, 1)]
                for j, ((blk, r0, nr), t) in enumerate(zip(live, wparts)):
                    _lib.check(lib.cffm_linear_bwd_weight(_ptr(blk), C.c_void_p(rows.data_ptr() + 4 * r0 * c), _ptr(t), nr, o, c, sw[j % len(sw)]), lib)
            if ctx.needs_input_grad[2]:
                bparts = [torch.empty(o, dtype=torch.float32, device=dy.device) for _ in live]
                sb = _take(lib, st, 3)
                for (blk, r0, nr), t in zip(live, bparts):
                    _lib.check(lib.cffm_colsum(_ptr(blk), nr, o, _ptr(t), sb), lib)
        _join(lib, st)
        if ctx.needs_input_grad[1]:
            dwm = wparts[0] if wparts else torch.zeros(o, c, dtype=torch.float32, device=dy.device)
            for t in wparts[1:]:
                dwm += t
            dwm = dwm.view(o, c, 1, 1)
        if ctx.needs_input_grad[2]:
            db = bparts[0] if bparts else torch.zeros(o, dtype=torch.float32, device=dy.device)
            for t in bparts[1:]:
                db += t
        return dx, dwm, db, None, None


def conv1x1(x, weight, bias, clips=0, extra=0):
    """nn.Conv2d(C, O, kernel_size=1)(x) (the head's classifiers `linear_pred*`, cffm_head.py:121,147,524) as a split-bf16 MFMA
    GEMM on token rows; returns [N,O,H,W] in channels-last memory, or [clips, N/clips, O, H, W] (same memory) when `clips` is given.
    `extra` (with `clips`): the rows are written into a [clips, N/clips + extra, H, W, O] buffer, so that `cat_into` can append the clip-level
    maps (cffm_head.py:150 `torch.cat([x, x2], 1)`) without copying these."""
    return _Conv1x1Fn.apply(x, weight, bias, int(clips), int(extra))


class _CatIntoFn(torch.autograd.Function):
    """torch.cat([x, x2], 1) of logits maps that live as token rows, where x = buf[:, :T] is the front of a [B, T+e, h, w, K] buffer
    (conv1x1(..., extra=e)): only x2 is copied; the result is the whole buffer viewed as [B, T+e, K, h, w].  Backward: two views."""

    @staticmethod
    def forward(ctx, x, x2):
        t, e = x.shape[1], x2.shape[1]
        rows = x.permute(0, 1, 3, 4, 2)                                   # [B, T, h, w, K]: contiguous slices of the buffer
        buf = torch.as_strided(rows, (x.shape[0], t + e, rows.shape[2], rows.shape[3], rows.shape[4]), rows.stride(), rows.storage_offset())
        buf[:, t:].copy_(x2.permute(0, 1, 3, 4, 2))
        ctx.t = t
        return buf.permute(0, 1, 4, 2, 3)

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.t], g[:, ctx.t:]


_DEFERRED = []         # tensors the deferred branch (cffm_defer_begin) still touches; cleared by _join_deferred


def _join_deferred(t=None):
    """the caller's stream continues behind the deferred branch (no-op when nothing is pending): called by whoever consumes deferred results
    (bn_relu_pool's backward) and once more at the end of every backward pass that deferred something"""
    if _DEFERRED:
        lib = _lib.get()
        lib.cffm_defer_join(_stream(_DEFERRED[0][0]))
        del _DEFERRED[:]


class _LateGradFn(torch.autograd.Function):
    """Identity on parameters whose gradients will come from the deferred branch.  Autograd accumulates a parameter gradient (and may CLONE
    it) the moment the producing node returns -- for a gradient still being computed on another stream that is a race (found by
    tests/test_headfuse.py::test_head_step_replayed_equals_eager_gpu: linear_pred.bias.grad was a copy of stale memory).  Applied EARLY in the
    forward, this node runs LATE in the backward (the engine takes ready nodes in reverse creation order); it joins the deferred branch and
    hands the gradients on, so whatever autograd does with them happens behind the join."""

    @staticmethod
    def forward(ctx, *ps):
        return tuple(p.view_as(p) for p in ps)

    @staticmethod
    def backward(ctx, *gs):
        _join_deferred()
        return gs


def late_params(*ps):
    return _LateGradFn.apply(*ps)


class _FrameLogitsCatFn(torch.autograd.Function):
    """torch.cat([linear_pred(fused) viewed [B,T,K,h,w], x2], 1) (cffm_head.py:121 + :150) as ONE node, created LAST in the forward -- so it
    is the FIRST node of the backward, and the classifier's backward (input gradient: needed by linear_fuse's BatchNorm backward; weight /
    bias gradients: needed by the optimizer) goes to the library's deferred branch and runs BESIDE the clip-level path's and the whole CFFM
    layer's backward instead of behind them (110 us of a replayed head step).  The returned input gradient must not be read on the caller's
    stream before _join_deferred(): ops.bn_relu_pool's backward -- its only consumer on the heads' rows path -- joins first, and the end of
    the backward pass joins in any case.  `weight` / `bias` must come through ops.late_params (their gradients are deferred too)."""

    @staticmethod
    def forward(ctx, fused, weight, bias, x2, clips):
        lib = _lib.get()
        for t in (fused, weight, bias, x2):
            _require_device(t, 'frame_logits_cat operand')
        n, c, h, w = fused.shape
        o = weight.shape[0]
        if weight.dim() != 4 or tuple(weight.shape[1:]) != (c, 1, 1) or bias.shape != (o,) or c % 4 or o % 4 or n % clips:
            raise _lib.CffmError('frame_logits_cat: fused %s, weight %s, bias %s, %d clips do not fit' % (tuple(fused.shape), tuple(weight.shape), tuple(bias.shape), clips))
        t = n // clips
        e = x2.shape[1]
        if x2.shape != (clips, e, o, h, w):
            raise _lib.CffmError('frame_logits_cat: clip-level maps %s, [%d, e, %d, %d, %d] expected' % (tuple(x2.shape), clips, o, h, w))
        rows = _to_rows(lib, fused)
        wm, b = weight.reshape(o, c).contiguous(), bias.contiguous()
        buf = torch.empty(clips, t + e, h, w, o, dtype=torch.float32, device=fused.device)
        per, st = t * h * w, _stream(fused)
        for i in range(clips):
            if per:
                _lib.check(lib.cffm_linear_bias_fwd(C.c_void_p(rows.data_ptr() + 4 * i * per * c), _ptr(wm), _ptr(b), _ptr(buf[i]), per, o, c, st), lib)
        buf[:, t:].copy_(x2.permute(0, 1, 3, 4, 2))
        ctx.save_for_backward(rows, wm)
        ctx.dims = (clips, t, e)
        ctx.x_plain = fused.is_contiguous()
        return buf.permute(0, 1, 4, 2, 3)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.get()
        rows, wm = ctx.saved_tensors
        clips, t, e = ctx.dims
        n, h, w, c = rows.shape
        o = wm.shape[0]
        g5 = g.permute(0, 1, 3, 4, 2)                                      # [B, T+e, h, w, K] as the loss kernel wrote it
        if not all(g5[i].is_contiguous() for i in range(clips)):
            g5 = g5.contiguous()
        per, st, dev = t * h * w, _stream(rows), rows.device
        dx = torch.empty(n, h, w, c, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        dwm = torch.zeros(o, c, dtype=torch.float32, device=dev) if (ctx.needs_input_grad[1] and not per) else (torch.empty(o, c, dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None)
        db = torch.zeros(o, dtype=torch.float32, device=dev) if (ctx.needs_input_grad[2] and not per) else (torch.empty(o, dtype=torch.float32, device=dev) if ctx.needs_input_grad[2] else None)
        keep = [rows, wm, g5, dx, dwm, db]
        if per:
            sd = C.c_void_p(lib.cffm_defer_begin(st))
            blk = [g5[i, :t] for i in range(clips)]
            if dx is not None:
                for i in range(clips):
                    _lib.check(lib.cffm_linear_bwd_input(_ptr(blk[i]), _ptr(wm), C.c_void_p(dx.data_ptr() + 4 * i * per * c), per, o, c, sd), lib)
            for i in range(clips):
                xi = C.c_void_p(rows.data_ptr() + 4 * i * per * c)
                if dwm is not None:
                    tw = dwm if i == 0 else torch.empty_like(dwm)
                    _lib.check(lib.cffm_linear_bwd_weight(_ptr(blk[i]), xi, _ptr(tw), per, o, c, sd), lib)
                    if i:
                        _lib.check(lib.cffm_add_inplace(_ptr(dwm), _ptr(tw), o * c, sd), lib)
                        keep.append(tw)
                if db is not None:
                    tb = db if i == 0 else torch.empty_like(db)
                    _lib.check(lib.cffm_colsum(_ptr(blk[i]), per, o, _ptr(tb), sd), lib)
                    if i:
                        _lib.check(lib.cffm_add_inplace(_ptr(db), _ptr(tb), o, sd), lib)
                        keep.append(tb)
            if sd.value != st.value:
                if not _DEFERRED:
                    torch.autograd.Variable._execution_engine.queue_callback(_join_deferred)
                _DEFERRED.append(keep)
        dfused = None
        if dx is not None:
            if ctx.x_plain and per:                     # (plain NCHW input: the heads' rows path hands channels-last memory, this is the general case)
                _join_deferred()
                dxp = torch.empty(n, c, h, w, dtype=torch.float32, device=dev)
                _lib.check(lib.cffm_transpose(_ptr(dx), _ptr(dxp), n, h * w, c, c * h * w, c * h * w, st), lib)
                dfused = dxp
            else:
                dfused = dx.permute(0, 3, 1, 2)
        return dfused, (dwm.view(o, c, 1, 1) if dwm is not None else None), db, g[:, t:], None


def frame_logits_cat(fused, weight, bias, x2, clips):
    """[B, T+e, K, h, w] = cat([conv1x1(fused) per clip, x2], 1): see _FrameLogitsCatFn (o % 4 == 0 and c % 4 == 0 as for conv1x1)"""
    return _FrameLogitsCatFn.apply(fused, weight, bias, x2, int(clips))


def cat_room(x, extra):
    """True when x [B,T,K,h,w] is the front of a buffer with room for `extra` more maps per clip (conv1x1(..., extra=...))"""
    if x.dim() != 5:
        return False
    b, t, k, h, w = x.shape
    per = h * w * k
    try:
        ok = x.stride() == ((t + extra) * per, per, 1, w * k, k) and x.untyped_storage().nbytes() >= 4 * (x.storage_offset() + b * (t + extra) * per)
    except Exception:       # noqa: BLE001
        ok = False
    return bool(ok)


def cat_into(x, x2):
    return _CatIntoFn.apply(x, x2)


# ---------------------------------------------------------------------------------------------- the layer on token rows
class _LayerRowsFn(torch.autograd.Function):
    """BasicLayer3d3 on channels-last token rows: x_rows [B,4,H*W,256] -> the new target frame [B,H*W,256].  Same kernels as
    _LayerFn without the four NCHW <-> NHWC transposes (the heads' neighbours of the hot path work on rows too)."""

    @staticmethod
    def forward(ctx, x_rows, h0, w0, depth, *params):
        lib = _lib.get()
        _require_device(x_rows, 'cffm layer input')
        if x_rows.dim() != 4 or x_rows.shape[1] != 4 or x_rows.shape[2] != h0 * w0 or x_rows.shape[3] != 256:
            raise _lib.CffmError('expected x_rows [B,4,%d,256], got %s' % (h0 * w0, tuple(x_rows.shape)))
        assert len(params) == NPB * depth
        x_rows = x_rows.contiguous()
        b = x_rows.shape[0]
        g = make_geom(lib, b, h0, w0)
        key_src, q_dst, inv_ptr, inv_idx = device_tables(h0, w0, x_rows.device)
        saved = torch.empty(lib.cffm_layer_saved_floats(C.byref(g), depth), dtype=torch.float32, device=x_rows.device)
        scratch = torch.empty(lib.cffm_layer_scratch_floats(C.byref(g)), dtype=torch.float32, device=x_rows.device)
        y = torch.empty(b, h0 * w0, 256, dtype=torch.float32, device=x_rows.device)
        _lib.check(lib.cffm_layer_forward_rows(C.byref(g), depth, block_structs(params, depth), _ptr(x_rows), _ptr(y), _ptr(key_src),
                                               _ptr(q_dst), _ptr(saved), _ptr(scratch), _stream(x_rows)), lib)
        ctx.depth, ctx.geom_args = depth, (b, h0, w0)
        ctx.save_for_backward(x_rows, saved, key_src, q_dst, inv_ptr, inv_idx, *params)
        ctx.scratch = scratch
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.get()
        x_rows, saved, key_src, q_dst, inv_ptr, inv_idx, *params = ctx.saved_tensors
        depth = ctx.depth
        b, h0, w0 = ctx.geom_args
        g = make_geom(lib, b, h0, w0)
        dy = dy.contiguous()
        sizes = [(p.numel() + 3) // 4 * 4 for p in params]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=dy.device)       # (the library zeroes the alignment gaps: see _LayerFn.backward)
        grads = [c[:p.numel()].view(p.shape) for c, p in zip(flat.split(sizes), params)]
        dx = torch.empty(b, 4, h0 * w0, 256, dtype=torch.float32, device=dy.device)
        _checked_padded(lib, lib.cffm_layer_backward_rows, C.byref(g), depth, block_structs(params, depth), block_structs(grads, depth), _ptr(x_rows),
                                                _ptr(dy), _ptr(dx), _ptr(key_src), _ptr(q_dst), _ptr(inv_ptr), _ptr(inv_idx), _ptr(saved),
                                                _ptr(ctx.scratch), _stream(dy))
        return (dx, None, None, None) + tuple(grads)


def cffm_layer_rows(x_rows, h0, w0, depth, params):
    """x_rows [B,4,H*W,256] (frame-major clip stack, channels-last) -> [B,H*W,256]: the new target frame of the reference's
    BasicLayer3d3 output (frames 0..2 of that output are the input frames)."""
    return _LayerRowsFn.apply(x_rows, h0, w0, depth, *params)


# ---------------------------------------------------------------------------------------------- BN + ReLU + 1/8 clip stack
class _BnReluPoolFn(torch.autograd.Function):
    """y [N,256,H,W] (channels-last memory) -> (fused = ReLU(BatchNorm(y)) [N,256,H,W] channels-last, stack = 2x2 average of
    fused as rows [N, H/2*W/2, 256]) -- `linear_fuse`'s norm + activation and the 1/4 -> 1/8 resize of cffm_head.py:119,131-135
    in two passes over the map (statistics, apply) instead of torch's batch_norm + relu + interpolate (+ layout conversions).
    Training mode uses batch statistics (summed over the ranks of `group` when given: SyncBN) and updates the running buffers
    as torch does; eval mode uses the running statistics."""

    @staticmethod
    def forward(ctx, y, weight, bias, running_mean, running_var, training, momentum, eps, want_stack, group, mask):
        lib = _lib.get()
        _require_device(y, 'bn_relu_pool input')
        n, c, h, w = y.shape
        if c != 256 or h % 2 or w % 2:
            raise _lib.CffmError('bn_relu_pool: [N,256,even,even] expected, got %s' % (tuple(y.shape),))
        if mask is not None and (tuple(mask.shape) != (n, 256) or mask.dtype != torch.float32):
            raise _lib.CffmError('bn_relu_pool: the dropout mask is a float32 [N,256] table, got %s' % (tuple(mask.shape),))
        rows = _to_rows(lib, y) if not y.permute(0, 2, 3, 1).is_contiguous() else y.permute(0, 2, 3, 1)
        r, st, dev = n * h * w, _stream(y), y.device
        count = float(r)
        mask = mask.contiguous() if mask is not None else None
        coef = torch.empty(4, 256, dtype=torch.float32, device=dev)     # scale | shift | rstd | -mean * rstd
        if training and group is not None:
            # SyncBN: the statistics cross the process group between the two passes, so the per-channel arithmetic stays in torch
            import torch.distributed as dist
            part = torch.empty(lib.cffm_colstats_records(r), 512, dtype=torch.float32, device=dev)
            _lib.check(lib.cffm_colstats(_ptr(rows), r, _ptr(part), st), lib)
            packed = torch.cat([part.double().sum(0), torch.full((1,), count, dtype=torch.float64, device=dev)])
            dist.all_reduce(packed, group=group if group is not True else None)
            # the global element count stays ON THE DEVICE (a 0-d tensor): no host synchronisation between the two passes, so a
            # head step under SyncBN neither stalls the stream nor breaks a stream capture (`.item()` here did both)
            sums, count = packed[:512], packed[512]
            mean = sums[:256] / count
            var = (sums[256:] / count - mean * mean).clamp_min(0.)
            if running_mean is not None:
                with torch.no_grad():
                    running_mean.mul_(1 - momentum).add_(mean.float() * momentum)
                    running_var.mul_(1 - momentum).add_((var * (count / (count - 1).clamp_min(1.))).float() * momentum)
            rstd = (var + eps).rsqrt()
            coef[0], coef[1] = (weight.double() * rstd).float(), (bias.double() - mean * weight.double() * rstd).float()
            coef[2], coef[3] = rstd.float(), (-mean * rstd).float()
        else:
            # statistics -> scale / shift / running buffers in ONE launch (stock torch: ~25 kernels on 256-element vectors)
            part = None
            if training:
                part = torch.empty(lib.cffm_colstats_records(r), 512, dtype=torch.float32, device=dev)
                _lib.check(lib.cffm_colstats(_ptr(rows), r, _ptr(part), st), lib)
            w_, b_ = weight.detach().contiguous(), bias.detach().contiguous()
            _lib.check(lib.cffm_bn_finalize_fwd(_ptr(part) if training else None, part.shape[0] if training else 0, count, _ptr(w_), _ptr(b_),
                                                _ptr(running_mean) if running_mean is not None else None,
                                                _ptr(running_var) if running_var is not None else None,
                                                float(momentum), float(eps), _ptr(coef), st), lib)
        fused = torch.empty(n, h, w, 256, dtype=torch.float32, device=dev)
        stack = torch.empty(n, (h // 2) * (w // 2), 256, dtype=torch.float32, device=dev) if want_stack else None
        _lib.check(lib.cffm_bn_relu_pool_fwd(_ptr(rows), _ptr(coef[0]), _ptr(coef[1]), _ptr(mask) if mask is not None else None, _ptr(fused),
                                             _ptr(stack) if want_stack else None, n, h, w, st), lib)
        ctx.save_for_backward(rows, coef, weight, mask if mask is not None else torch.empty(0, device=dev))
        ctx.dims, ctx.training, ctx.count, ctx.group = (n, h, w), training, count, group
        out_stack = stack if want_stack else torch.empty(0, device=dev)
        return fused.permute(0, 3, 1, 2), out_stack

    @staticmethod
    def backward(ctx, dfused, dstack):
        lib = _lib.get()
        _join_deferred()          # dfused may come from the deferred branch (frame_logits_cat's backward)
        rows, coef, weight, mask = ctx.saved_tensors
        mask = mask if mask.numel() else None
        n, h, w = ctx.dims
        r, st, dev = n * h * w, _stream(rows), rows.device
        df = None
        if dfused is not None:
            df = dfused.permute(0, 2, 3, 1)
            df = df if df.is_contiguous() else _to_rows(lib, dfused.contiguous())
        ds = dstack.contiguous() if (dstack is not None and dstack.numel()) else None
        g = torch.empty(n, h, w, 256, dtype=torch.float32, device=dev)
        part = torch.empty(lib.cffm_bn_relu_pool_records(n, h, w), 512, dtype=torch.float32, device=dev)
        _lib.check(lib.cffm_bn_relu_pool_bwd1(_ptr(rows), _ptr(coef[0]), _ptr(coef[1]), _ptr(coef[2]), _ptr(coef[3]),
                                              _ptr(mask) if mask is not None else None, _ptr(df) if df is not None else None,
                                              _ptr(ds) if ds is not None else None, _ptr(g), _ptr(part), n, h, w, st), lib)
        out = torch.empty(5, 256, dtype=torch.float32, device=dev)       # dbias | dweight | mean g | mean g*xhat | gamma*rstd
        w_ = weight.detach().contiguous()
        if ctx.training and ctx.group is not None:
            import torch.distributed as dist
            sums = part.double().sum(0)
            out[0], out[1] = sums[:256].float(), sums[256:].float()       # of THIS rank (DDP averages parameter gradients itself)
            sums = sums.clone()
            dist.all_reduce(sums, group=ctx.group if ctx.group is not True else None)
            out[2], out[3] = (sums[:256] / ctx.count).float(), (sums[256:] / ctx.count).float()
            out[4] = w_ * coef[2]
        else:
            _lib.check(lib.cffm_bn_finalize_bwd(_ptr(part), part.shape[0], ctx.count, _ptr(w_), _ptr(coef[2]), 1 if ctx.training else 0,
                                                _ptr(out), st), lib)
        _lib.check(lib.cffm_bn_bwd2(_ptr(g), _ptr(rows), _ptr(coef[2]), _ptr(coef[3]), _ptr(out[4]), _ptr(out[2]), _ptr(out[3]), r, st), lib)
        return g.permute(0, 3, 1, 2), out[1], out[0], None, None, None, None, None, None, None, None


def bn_relu_pool(y, bn, want_stack=True, drop_mask=None):
    """ReLU(bn(y)) and its 2x2-average clip-stack rows; `bn` is the head's BatchNorm2d / SyncBatchNorm module (its parameters,
    running buffers, momentum, eps and training flag are honoured; SyncBatchNorm exchanges the batch statistics over the default
    process group when one is initialised).  `drop_mask` [N,256] (factors 0 or 1/(1-p)): Dropout2d of the first output folded into
    the same pass -- the stack is taken before it, as cffm_head.py:119-131 does."""
    group = None
    if isinstance(bn, torch.nn.SyncBatchNorm) and bn.training:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            group = bn.process_group if bn.process_group is not None else True
    training = bn.training or bn.running_mean is None
    if bn.training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    if bn.momentum is not None:
        momentum = bn.momentum
    elif bn.training and bn.num_batches_tracked is not None:
        momentum = 1.0 / float(bn.num_batches_tracked)      # torch: momentum=None means a cumulative moving average (one host read, as torch does)
    else:
        momentum = 0.0
    fused, stack = _BnReluPoolFn.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, momentum, bn.eps, want_stack, group,
                                       drop_mask)
    return fused, (stack if want_stack else None)


# ---------------------------------------------------------------------------------------------- bilinear resize of token rows
class _RowsResizeFn(torch.autograd.Function):
    """F.interpolate(mode='bilinear', align_corners=False) on token rows: [N,h,w,C] -> [N,H,W,C] (C % 4 == 0).  The gradient is read
    where it lies (maps may be slices of a larger rows buffer), the adjoint is a deterministic gather."""

    @staticmethod
    def forward(ctx, rows, H, W):
        lib = _lib.get()
        _require_device(rows, 'rows_resize input')
        if rows.dim() != 4 or rows.dtype != torch.float32 or rows.shape[3] % 4:
            raise _lib.CffmError('rows_resize: fp32 [N,h,w,C] with C %% 4 == 0 expected, got %s %s' % (tuple(rows.shape), rows.dtype))
        rows = rows.contiguous()
        n, h, w, c = rows.shape
        out = torch.empty(n, H, W, c, dtype=torch.float32, device=rows.device)
        _lib.check(lib.cffm_rows_resize_fwd(_ptr(rows), h * w * c, _ptr(out), H * W * c, n, h, w, H, W, c, _stream(rows)), lib)
        ctx.dims = (n, h, w, H, W, c)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.get()
        n, h, w, H, W, c = ctx.dims
        if not (n == 0 or (dout.stride()[1:] == (W * c, c, 1) and dout.stride(0) >= H * W * c and dout.stride(0) % 4 == 0)):
            dout = dout.contiguous()
        din = torch.empty(n, h, w, c, dtype=torch.float32, device=dout.device)
        _lib.check(lib.cffm_rows_resize_bwd(_ptr(dout), dout.stride(0) if n else H * W * c, _ptr(din), h * w * c, n, h, w, H, W, c,
                                            _stream(dout)), lib)
        return din, None, None


def rows_resize(rows, size):
    """bilinear resize (align_corners=False) of token rows [N,h,w,C] to [N,H,W,C]"""
    return _RowsResizeFn.apply(rows, int(size[0]), int(size[1]))


# ---------------------------------------------------------------------------------------------- resize + cross entropy
class _UpceFn(torch.autograd.Function):
    """sum over pixels of CE(resize(logits)[pixel], label[pixel]) and the number of pixels whose arg-max is the label,
    without the [M,K,H,W] resized logits (csrc/segloss_kernels.h)."""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        lib = _lib.get()
        _require_device(logits, 'resize_cross_entropy logits')
        if logits.dim() != 4 or labels.dim() != 3 or labels.shape[0] != logits.shape[0] or labels.dtype != torch.int64:
            raise _lib.CffmError('resize_cross_entropy: logits [M,K,h,w] fp32 and labels [M,H,W] int64 expected, got %s %s / %s %s'
                                 % (tuple(logits.shape), logits.dtype, tuple(labels.shape), labels.dtype))
        if labels.device != logits.device:
            raise _lib.CffmError('resize_cross_entropy: labels on %s, logits on %s' % (labels.device, logits.device))
        logits, labels = logits.contiguous(), labels.contiguous()
        m, k, h, w = logits.shape
        H, W = labels.shape[1:]
        lse = torch.empty(m, H, W, dtype=torch.float32, device=logits.device)
        part = torch.empty(lib.cffm_upce_blocks(m, H, W), 2, dtype=torch.float32, device=logits.device)
        _lib.check(lib.cffm_upce_fwd(_ptr(logits), _ptr(labels), _ptr(lse), _ptr(part), m, k, h, w, H, W, int(ignore_index),
                                     _stream(logits)), lib)
        sums = part.double().sum(0).float()          # [loss sum, hits]: a plain deterministic reduction of the workgroup records
        ctx.save_for_backward(logits, labels, lse)
        ctx.ignore_index = int(ignore_index)
        hits = sums[1].clone()
        ctx.mark_non_differentiable(hits)
        return sums[0].clone(), hits

    @staticmethod
    def backward(ctx, gloss, _ghits):
        lib = _lib.get()
        logits, labels, lse = ctx.saved_tensors
        m, k, h, w = logits.shape
        H, W = labels.shape[1:]
        gs = gloss.to(torch.float32).contiguous()    # device scalar: read by the kernel, never by the host
        dlogits = torch.empty_like(logits)
        _lib.check(lib.cffm_upce_bwd(_ptr(logits), _ptr(labels), _ptr(lse), _ptr(gs), 1.0, _ptr(dlogits), m, k, h, w, H, W,
                                     ctx.ignore_index, _stream(logits)), lib)
        return dlogits, None, None


def resize_cross_entropy(logits, labels, ignore_index=255):
    """(sum of per-pixel cross entropies, number of correctly classified pixels) of `logits` [M,K,h,w] resized bilinearly
    (align_corners=False) to the labels' [M,H,W] resolution -- what decode_head.py:744-835 computes through a materialised
    [M,K,H,W] tensor (resize -> F.cross_entropy(reduction='none', ignore_index) -> sum; accuracy's arg-max == label count).
    Ignored pixels add 0 to both, as in the reference; divide by labels.numel() for its `mean` / percentage."""
    return _UpceFn.apply(logits, labels, ignore_index)


_MAPS_TABLES = {}


def _maps_tables(dev, b, t, label_idx, loss_w, hits_w):
    """device copies of the per-map tables of head_cross_entropy, made once per configuration (a host -> device copy per step would
    also keep the step from being captured in a graph)"""
    key = (str(dev), b, t, tuple(label_idx), tuple(loss_w), tuple(hits_w))
    hit = _MAPS_TABLES.get(key)
    if hit is None:
        if len(_MAPS_TABLES) > 64:
            _MAPS_TABLES.clear()
        hit = (torch.tensor([bi * t + int(i) for bi in range(b) for i in label_idx], dtype=torch.int32, device=dev),
               torch.tensor([float(x) for x in loss_w] * b, dtype=torch.float64, device=dev),
               torch.tensor([float(x) for x in hits_w] * b, dtype=torch.float64, device=dev),
               torch.tensor([float(x) for x in loss_w] * b, dtype=torch.float32, device=dev))      # (the backward's per-map scale)
        _MAPS_TABLES[key] = hit
    return hit


class _UpceMapsFn(torch.autograd.Function):
    """The head's whole loss in one pair of kernels: logits [B,n,K,h,w] (plain memory or token rows [B,n,h,w,K] viewed that way --
    no layout copy either way), labels [B,t,H,W]; map (b, i) is judged on label map (b, label_idx[i]) and enters the loss / the
    hit count with loss_w[i] / hits_w[i] (decode_head.py:744-835: the frame maps at half weight on their own frames, the
    clip-level maps on the last frame, accuracy over the frame maps)."""

    @staticmethod
    def forward(ctx, logits, labels, label_idx, loss_w, hits_w, ignore_index):
        lib = _lib.get()
        _require_device(logits, 'head_cross_entropy logits')
        if logits.dim() != 5 or labels.dim() != 4 or labels.shape[0] != logits.shape[0] or labels.dtype != torch.int64 or \
                logits.dtype != torch.float32 or labels.device != logits.device:
            raise _lib.CffmError('head_cross_entropy: logits [B,n,K,h,w] fp32 and labels [B,t,H,W] int64 on one device expected, got %s %s / %s %s'
                                 % (tuple(logits.shape), logits.dtype, tuple(labels.shape), labels.dtype))
        b, n, k, h, w = logits.shape
        t, H, W = labels.shape[1:]
        if len(label_idx) != n or len(loss_w) != n or len(hits_w) != n or any(not 0 <= int(i) < t for i in label_idx):
            raise _lib.CffmError('head_cross_entropy: one label index (< %d) and two weights per logits map expected' % t)
        st = logits.stride()
        rows = k > 1 and st[2] == 1 and st[4] == k and st[3] == w * k            # token rows [.., h, w, K]
        plain = st[4] == 1 and st[3] == w and st[2] == h * w
        if not (rows or plain) or st[1] < k * h * w or st[0] < n * st[1]:
            logits = logits.contiguous()
            st, rows, plain = logits.stride(), False, True
        labels = labels.contiguous()
        dev, m = logits.device, b * n
        lidx, wl, wh, wl32 = _maps_tables(dev, b, t, label_idx, loss_w, hits_w)
        lse = torch.empty(m, H, W, dtype=torch.float32, device=dev)
        nblk = lib.cffm_upce_blocks(m, H, W)
        part = torch.empty(nblk, 2, dtype=torch.float32, device=dev)
        ks, ps = (1, k) if rows else (h * w, 1)
        _lib.check(lib.cffm_upce_maps_fwd(_ptr(logits), _ptr(labels), _ptr(lidx), _ptr(lse), _ptr(part), m, k, h, w, H, W, int(ignore_index),
                                          n, st[0], st[1], ks, ps, _stream(logits)), lib)
        out = torch.zeros(2, dtype=torch.float32, device=dev) if not m else torch.empty(2, dtype=torch.float32, device=dev)
        if m:       # deterministic record sums and the per-map weighting in one launch (it was ~10 torch kernels: 60 us of a replayed step)
            _lib.check(lib.cffm_upce_maps_finalize(_ptr(part), m, nblk // m, _ptr(wl), _ptr(wh), _ptr(out), _stream(logits)), lib)
        loss, hits = out[0], out[1]
        ctx.save_for_backward(logits, labels, lse, lidx, wl32)
        ctx.geom = (n, ks, ps, int(ignore_index))
        ctx.mark_non_differentiable(hits)
        return loss, hits

    @staticmethod
    def backward(ctx, gloss, _ghits):
        lib = _lib.get()
        logits, labels, lse, lidx, wl = ctx.saved_tensors
        n, ks, ps, ignore_index = ctx.geom
        b, _, k, h, w = logits.shape
        H, W = labels.shape[2:]
        st = logits.stride()
        gs = gloss.to(torch.float32).contiguous()    # device scalar: read by the kernel, never by the host
        dlogits = torch.empty_strided(logits.shape, st, dtype=torch.float32, device=logits.device)
        if b * n and (st[1] != k * h * w or st[0] != n * st[1]):
            dlogits.zero_()                          # (gaps between the maps of a padded buffer)
        _lib.check(lib.cffm_upce_maps_bwd(_ptr(logits), _ptr(labels), _ptr(lidx), _ptr(lse), _ptr(gs), _ptr(wl), 1.0, _ptr(dlogits), b * n, k,
                                          h, w, H, W, ignore_index, n, st[0], st[1], ks, ps, _stream(logits)), lib)
        return dlogits, None, None, None, None, None


def head_cross_entropy(logits, labels, label_idx, loss_w, hits_w, ignore_index=255):
    """(sum_i loss_w[i] * CE-sum of map i, sum_i hits_w[i] * correctly classified pixels of map i) over the n logits maps of every
    clip, each resized bilinearly (align_corners=False) to its label map -- the two `resize -> F.cross_entropy` branches and the
    accuracy of decode_head.py:744-835 in ONE forward and ONE backward kernel, on the logits wherever they lie (plain
    [B,n,K,h,w] or the classifiers' token rows viewed as such): no resized logits, no layout conversion, no gradient assembly."""
    return _UpceMapsFn.apply(logits, labels, tuple(label_idx), tuple(loss_w), tuple(hits_w), ignore_index)


# ---------------------------------------------------------------------------------------------- GTC
GTC_PARAM_KEYS = ('norm1.weight', 'norm1.bias', 'attn.qkv.weight', 'attn.qkv.bias', 'attn.qkv_cluster.weight',
                  'attn.qkv_cluster.bias', 'attn.proj_cluster.weight', 'attn.proj_cluster.bias', 'norm2.weight',
                  'norm2.bias', 'mlp.fc1.weight', 'mlp.fc1.bias', 'mlp.fc2.weight', 'mlp.fc2.bias')


# expected shapes of the 14 parameters (dim 256, mlp_ratio 4: the only CFFM++ prototype block the library is built for)
GTC_PARAM_SHAPES = ((256,), (256,), (768, 256), (768,), (512, 256), (512,), (256, 256), (256,), (256,), (256,), (1024, 256), (1024,),
                    (256, 1024), (256,))
_WS_FILL = None      # tests only: fill value for the block's (otherwise uninitialised) workspace, e.g. NaN


def _gtc_check_params(p, dev):
    """The library takes raw pointers: a mis-ordered, non-fp32 or off-device parameter list would be out-of-bounds device accesses, so
    it is refused here (the per-stage path of rounds 1-4 checked each tensor at its stage call)."""
    if len(p) != len(GTC_PARAM_KEYS):
        raise _lib.CffmError('gtc_block: %d parameters expected (GTC_PARAM_KEYS order), got %d' % (len(GTC_PARAM_KEYS), len(p)))
    for key, shape, t in zip(GTC_PARAM_KEYS, GTC_PARAM_SHAPES, p):
        if tuple(t.shape) != shape or t.dtype != torch.float32 or t.device != dev:
            raise _lib.CffmError('gtc_block: parameter %s must be float32 %s on %s, got %s %s on %s'
                                 % (key, shape, dev, t.dtype, tuple(t.shape), t.device))


def _gtc_ptrs(ts):
    """cffm_gtc_params / cffm_gtc_grads from 14 tensors in GTC_PARAM_KEYS order."""
    s = _lib.GtcPtrs()
    for (name, _), t in zip(_lib.GtcPtrs._fields_, ts):
        setattr(s, name, t.data_ptr())
    return s


class _GtcBlockFn(torch.autograd.Function):
    """SwinTransformerBlock_cluster.forward (pvt/swin_transformer_2d.py:605-665), shift 0:
    x [B,T,256], centers [B,K,256] -> [B,T,256].  One library call per direction (cffm_gtc_block_forward / _backward, round 5: the q
    projection as a row-panel GEMM, proj_cluster + residual + norm2 + Mlp as the base block's fused launch, one grouped
    weight-gradient launch); rounds 1-4 sequenced ~33 stage launches here."""

    @staticmethod
    def forward(ctx, x, centers, *p):
        lib = _lib.get()
        _require_device(x, 'gtc input')
        _require_device(centers, 'gtc centers')
        p = [t.detach().contiguous() for t in p]
        _gtc_check_params(p, x.device)
        x, centers = x.contiguous(), centers.contiguous()
        b, t, c = x.shape
        k = centers.shape[1]
        if c != 256 or centers.shape[0] != b or centers.shape[2] != 256 or x.dtype != torch.float32 or centers.dtype != torch.float32:
            raise _lib.CffmError('gtc_block: x [B,T,256] and centers [B,K,256] float32 expected, got %s and %s' % (tuple(x.shape), tuple(centers.shape)))
        n = lib.cffm_gtc_ws_floats(b, t, k)
        if n <= 0:
            raise _lib.CffmError('gtc_block: bad sizes B=%d T=%d K=%d' % (b, t, k))
        ws = torch.empty(n, dtype=torch.float32, device=x.device)
        if _WS_FILL is not None:
            ws.fill_(_WS_FILL)
        out = torch.empty(b, t, c, dtype=torch.float32, device=x.device)
        ps = _gtc_ptrs(p)
        _lib.check(lib.cffm_gtc_block_forward(C.byref(ps), _ptr(x), _ptr(centers), _ptr(out), _ptr(ws), b, t, k, _stream(x)), lib)
        ctx.save_for_backward(x, centers, ws, *p)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.get()
        x, centers, ws = ctx.saved_tensors[:3]
        p = list(ctx.saved_tensors[3:])
        if dout.shape != x.shape or dout.dtype != torch.float32 or dout.device != x.device:
            raise _lib.CffmError('gtc_block backward: float32 gradient of shape %s on %s expected, got %s %s on %s'
                                 % (tuple(x.shape), x.device, dout.dtype, tuple(dout.shape), dout.device))
        dout = dout.contiguous()
        b, t, c = x.shape
        k = centers.shape[1]
        grads = [torch.empty_like(v) for v in p]
        dx, dcenters = torch.empty_like(x), torch.empty_like(centers)
        ps, gs = _gtc_ptrs(p), _gtc_ptrs(grads)
        _lib.check(lib.cffm_gtc_block_backward(C.byref(ps), C.byref(gs), _ptr(x), _ptr(centers), _ptr(dout), _ptr(dx), _ptr(dcenters), _ptr(ws),
                                               b, t, k, _stream(x)), lib)
        return (dx, dcenters) + tuple(grads)


def gtc_block(x, centers, params):
    """params: 14 tensors in GTC_PARAM_KEYS order."""
    return _GtcBlockFn.apply(x, centers, *params)
