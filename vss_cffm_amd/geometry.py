"""Host-side geometry of the CFFM hot path: the gather tables the HIP kernels walk.

The reference expresses the 289-key context of a 7x7 query window through ``torch.roll`` x4,
``window_partition``, ``nn.Unfold`` x4, ``cat`` and index buffers
(mmseg/models/decode_heads/cffm_module/cffm_transformer.py:389-418, :426-518).  Here all of that
is one int32 table built once per (H0, W0): ``key_src[w, n]`` = row of key ``n`` of window ``w`` in
the per-clip *token-row space*, or -1 where ``nn.Unfold`` reads its zero padding (the reference's
additive -100 mask, :445, :490).

Token-row space of one clip (``RC = 64 * nW`` rows of 256 channels; this is also the row order of
the q/k/v GEMM): ``[0, 49 nW)`` target-frame tokens, window-major (w * 49 + 7 i + j);
``[49 nW, 50 nW)`` pooled target windows (gy x gx); ``[50 nW, 51 nW)`` pooled frame t-9 (gy x gx);
``[51 nW, 55 nW)`` pooled frame t-6 (2gy x 2gx); ``[55 nW, 64 nW)`` pooled frame t-3 (3gy x 3gx).
"""
import functools

import numpy as np

WS = 7
EXPAND = 3
NKEY = 289
NKEY_PAD = 304
# (rows offset in units of nW, stride, kernel, padding) of the four pooled key groups, in key order
POOLED_GROUPS = ((49, 1, 5, 2), (50, 1, 7, 3), (51, 2, 5, 2), (55, 3, 3, 1))


def padded(n):
    return (n + WS - 1) // WS * WS


@functools.lru_cache(maxsize=32)
def tables(h0, w0):
    """-> (key_src int32 [nW, 304], q_dst int32 [nW, 49]) as numpy arrays."""
    hp, wp = padded(h0), padded(w0)
    gy, gx = hp // WS, wp // WS
    nw = gy * gx
    key_src = np.full((nw, NKEY_PAD), -1, dtype=np.int32)
    q_dst = np.full((nw, WS * WS), -1, dtype=np.int32)
    ii, jj = np.meshgrid(np.arange(WS), np.arange(WS), indexing='ij')
    ii, jj = ii.reshape(-1), jj.reshape(-1)
    # the four rolled copies keep an L-shaped band of 33 tokens each (valid_ind_rolled, :280-285)
    keep = (((ii >= WS - EXPAND) | (jj >= WS - EXPAND), +EXPAND, +EXPAND),
            ((ii >= WS - EXPAND) | (jj < EXPAND), +EXPAND, -EXPAND),
            ((ii < EXPAND) | (jj >= WS - EXPAND), -EXPAND, +EXPAND),
            ((ii < EXPAND) | (jj < EXPAND), -EXPAND, -EXPAND))
    for wy in range(gy):
        for wx in range(gx):
            w = wy * gx + wx
            y, x = WS * wy + ii, WS * wx + jj
            q_dst[w] = np.where((y < h0) & (x < w0), y * w0 + x, -1)
            key_src[w, :49] = w * 49 + np.arange(49)
            n = 49
            for sel, oy, ox in keep:
                yy, xx = (y[sel] + oy) % hp, (x[sel] + ox) % wp          # cyclic, no border mask
                key_src[w, n:n + yy.size] = ((yy // WS) * gx + xx // WS) * 49 + (yy % WS) * WS + xx % WS
                n += yy.size
            for off, s, kk, pad in POOLED_GROUPS:
                a, b = np.meshgrid(np.arange(kk), np.arange(kk), indexing='ij')
                u, v = s * wy - pad + a.reshape(-1), s * wx - pad + b.reshape(-1)
                ok = (u >= 0) & (u < gy * s) & (v >= 0) & (v < gx * s)
                key_src[w, n:n + kk * kk] = np.where(ok, off * nw + u * (gx * s) + v, -1)
                n += kk * kk
            assert n == NKEY
    key_src.setflags(write=False)
    q_dst.setflags(write=False)
    return key_src, q_dst


@functools.lru_cache(maxsize=32)
def inverse_tables(h0, w0):
    """CSR inverse of ``key_src``: for every token row of a clip, the ``window * 304 + slot`` pairs whose key is
    that row (ring / pooled keys are read by up to 49 windows; 12 ring positions twice by the same window).
    -> (inv_ptr int32 [64 nW + 1], inv_idx int32 [nnz]); the dK/dV gather pass of the backward walks it."""
    key_src, _ = tables(h0, w0)
    nw = key_src.shape[0]
    flat = key_src.reshape(-1)
    slots = np.nonzero(flat >= 0)[0].astype(np.int32)
    rows = flat[slots]
    order = np.argsort(rows, kind='stable')
    inv_idx = np.ascontiguousarray(slots[order])
    counts = np.bincount(rows, minlength=64 * nw)
    inv_ptr = np.zeros(64 * nw + 1, dtype=np.int32)
    inv_ptr[1:] = np.cumsum(counts)
    return inv_ptr, inv_idx
