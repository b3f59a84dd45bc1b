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


# ---- key-owner tables of the attention backward (k_cfm_attn_bwd, key-owner role) ---------------------------------------------
# dK / dV of a key row are sums over every (window, query) that read it: autograd of the roll / unfold / cat assembly
# (cffm_transformer.py:389-518).  The key-owner workgroups of the backward own up to 64 key rows (4 MFMA
# tiles of 16) and walk the windows that read them ("passes"); everything they need to know is derived here from ``key_src``:
#   * main units, one per window: its 49 tokens + its 4 + 9 pooled cells of frames t-6 / t-3; readers = the 9 cyclic
#     neighbours.  The rows are ordered by the 3x3 region of the window they lie in (tile A: top-left 16 tokens, B: top-right,
#     C: bottom-left, D: bottom-right, pooled cells in the free slots) so that most (neighbour, tile) pairs are empty;
#   * pooled units: 16 consecutive cells of the pooled-target / frame t-9 grids (read by up to 25 / 49 windows each).
# A pass = (reader window, half of its queries: 32 of the 64 padded query rows).
# ko_unit int32: header [NU, off_units, off_pass, off_rows, n_slots, 0, 0, 0] | units [NU][12] = {rows_off, ntile, p0..p8 (wave v of
#   the 8 walks passes [p_v, p_v+1)), cost} | passes [NP][4] = {reader window, layers per tile (4 bits each), slot offset, query half} |
#   rows [NU][64] (row of the token-row space or -1).
# ko_slot int16: per pass L layers of [16 keys][4 tiles]: key slot (0..288) under which the reader sees the key, or -1; layer l
#   holds the (l+1)-th occurrence (12 ring positions are read twice by the same window; more on grids narrower than 3 windows).
KO_TILES = 4
KO_WAVES = 8
KO_UNIT_REC = 12


def _main_unit_rows(wy, wx, gy, gx):
    nw = gy * gx
    w = wy * gx + wx
    reg = lambda i: 0 if i < 3 else (1 if i == 3 else 2)
    tiles = {0: [], 1: [], 2: [], 3: []}
    for i in range(7):
        for j in range(7):
            ri, rj = reg(i), reg(j)
            t = 3 if (ri == 2 and rj == 2) else 2 if ri == 2 else 1 if rj == 2 else 0
            tiles[t].append(w * 49 + 7 * i + j)
    f1 = lambda ky, kx: 51 * nw + (2 * wy + ky) * (2 * gx) + 2 * wx + kx
    f2 = lambda ky, kx: 55 * nw + (3 * wy + ky) * (3 * gx) + 3 * wx + kx
    tiles[1] += [f1(0, 0), f1(0, 1), f1(1, 0), f2(0, 0)]
    tiles[2] += [f2(0, 1), f2(1, 0), f2(1, 1)]
    tiles[3] += [f2(2, 2), f2(0, 2), f2(1, 2), f2(2, 0), f2(2, 1), f1(1, 1)]
    rows = []
    for t in range(4):
        assert len(tiles[t]) <= 16
        rows += tiles[t] + [-1] * (16 - len(tiles[t]))
    return rows


@functools.lru_cache(maxsize=32)
def ko_tables(h0, w0):
    """-> (ko_unit int32 [..], ko_slot int16 [..]) as numpy arrays (layout above)."""
    key_src, _ = tables(h0, w0)
    hp, wp = padded(h0), padded(w0)
    gy, gx = hp // WS, wp // WS
    nw = gy * gx
    readers_of = {}                       # row -> [(window, slot), ...] in (window, slot) order
    for w in range(nw):
        for n in range(NKEY):
            r = int(key_src[w, n])
            if r >= 0:
                readers_of.setdefault(r, []).append((w, n))
    unit_rows = [_main_unit_rows(wy, wx, gy, gx) for wy in range(gy) for wx in range(gx)]
    for off in (49, 50):
        for c0 in range(0, nw, 16):
            unit_rows.append([off * nw + c for c in range(c0, min(c0 + 16, nw))])
    units = []
    for rows in unit_rows:
        ntile = (len(rows) + 15) // 16
        rows = rows + [-1] * (64 - len(rows))
        per_reader = {}
        for k, r in enumerate(rows):
            for (w, n) in readers_of.get(r, ()) if r >= 0 else ():
                per_reader.setdefault(w, {}).setdefault(k, []).append(n)
        passes = []
        for w in sorted(per_reader):
            occ = per_reader[w]
            nl = [max([len(occ.get(16 * t + k, ())) for k in range(16)]) for t in range(KO_TILES)]
            L = max(nl)
            slots = np.full((L, 16, KO_TILES), -1, dtype=np.int16)
            for k, ns in occ.items():
                for l, n in enumerate(ns):
                    slots[l, k % 16, k // 16] = n
            for qp in (0, 1):                     # both query halves walk the same slots
                passes.append((w, nl, slots, 1 + sum(nl), qp))
        # longest-processing-time assignment of the passes to the 8 waves
        load = [0] * KO_WAVES
        mine = [[] for _ in range(KO_WAVES)]
        for p in sorted(passes, key=lambda p: -p[3]):
            v = load.index(min(load))
            load[v] += p[3]
            mine[v].append(p)
        units.append((rows, ntile, [sorted(m, key=lambda p: (p[0], p[4])) for m in mine], max(load)))
    units.sort(key=lambda u: -u[3])       # long units first
    nu = len(units)
    npass = sum(len(m) for u in units for m in u[2])
    off_units, off_pass = 8, 8 + KO_UNIT_REC * nu
    off_rows = off_pass + 4 * npass
    ko = np.zeros(off_rows + 64 * nu, dtype=np.int32)
    slot_chunks = []
    slot_at = {}
    ns = 0
    pi = 0
    for ui, (rows, ntile, mine, cost) in enumerate(units):
        rec = ko[off_units + KO_UNIT_REC * ui: off_units + KO_UNIT_REC * (ui + 1)]
        rec[0], rec[1], rec[11] = off_rows + 64 * ui, ntile, cost
        ko[off_rows + 64 * ui: off_rows + 64 * ui + 64] = rows
        for v in range(KO_WAVES):
            rec[2 + v] = pi
            for (w, nl, slots, _, qp) in mine[v]:
                if (ui, w) not in slot_at:        # the two halves of a reader share one slot block
                    slot_at[(ui, w)] = ns
                    slot_chunks.append(slots.reshape(-1))
                    ns += slots.size
                ko[off_pass + 4 * pi: off_pass + 4 * pi + 4] = (w, sum(nl[t] << (4 * t) for t in range(KO_TILES)), slot_at[(ui, w)], qp)
                pi += 1
        rec[2 + KO_WAVES] = pi
    ko[:5] = (nu, off_units, off_pass, off_rows, ns)
    ko_slot = np.concatenate(slot_chunks) if slot_chunks else np.zeros(0, np.int16)
    ko.setflags(write=False)
    ko_slot.setflags(write=False)
    return ko, ko_slot
