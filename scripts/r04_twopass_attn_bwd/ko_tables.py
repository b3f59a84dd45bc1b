"""geometry.ko_tables of the two-pass experiment (see README.md); imports the product geometry for key_src."""
import functools

import numpy as np

from vss_cffm_amd.geometry import NKEY, WS, padded, tables


# ---- key-owner tables of the attention backward (k_cfm_attn_bwd_k) -------------------------------------------------------------
# dK / dV of a key row are sums over every (window, query) that read it: autograd of the roll / unfold / cat assembly
# (cffm_transformer.py:389-518).  A key-owner workgroup (4 waves) owns the key rows of one UNIT and walks the
# windows that read them; everything it needs to know is derived here from ``key_src``:
#   * main units (kind 0), one per window: its 49 tokens + its 9 pooled cells of frame t-3, 4 tiles of 16 rows ordered by the 3x3
#     region of the window they lie in (tile A: top-left 16 tokens, B: top-right, C: bottom-left, D: bottom-right, the pooled cells
#     in the free slots of the tile whose readers they share), wave v owns tile v.  Readers = the 9 cyclic neighbours, every tile is
#     read by 4 of them;
#   * pooled units (kind 1), 1 tile: 16 consecutive cells of the pooled-target / frame t-9 grids (read by up to 25 / 49 windows each)
#     or a 4 x 4 block of cells of the frame t-6 grid (read by up to 16 windows).
# The readers of a unit are walked in INTERVALS of up to two windows staged together.  Main units pair readers that touch disjoint
# tiles (own window | right + left | below + above | two corners | two corners: 5 intervals), wave v multiplies its tile against the
# reader that touches it (both query halves); pooled units take their readers two at a time, wave v multiplies the tile against query
# half v & 1 of reader v >> 1, the four partial sums are added at the end.
# ko_unit int32: header [NU, off_units, off_int, off_rows, n_slots, 0, 0, 0] | units [NU][8] = {rows_off, ntile, kind, first interval,
#   end interval, slot base, 0, 0} | intervals [NI][16] = {reader 0, reader 1 (-1: none), 0, 0, then per wave {flags = reader slot |
#   query halves << 2 | layers << 4 (0: idle), 0, 0}} | rows [NU][64] (row of the token-row space or -1).
# ko_slot int16: per unit [interval][wave][16 keys][2 layers]: key slot (0..288) under which the wave's reader sees each key of the
#   wave's tile, -1 = not read; layer 1 = a second reading of the key by the same window (12 ring positions are read twice).
KO_TILES = 4
KO_WAVES = 4
KO_INT_REC = 16


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
    f2 = lambda ky, kx: 55 * nw + (3 * wy + ky) * (3 * gx) + 3 * wx + kx
    tiles[1] += [f2(0, 0), f2(0, 1), f2(1, 0), f2(1, 1)]                       # read by the window itself only
    tiles[3] += [f2(2, 2), f2(0, 2), f2(1, 2), f2(2, 0), f2(2, 1)]             # read by the right / lower / lower-right neighbour
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
    unit_rows = [(0, _main_unit_rows(wy, wx, gy, gx)) for wy in range(gy) for wx in range(gx)]
    for off in (49, 50):
        for c0 in range(0, nw, 16):
            unit_rows.append((1, [off * nw + c for c in range(c0, min(c0 + 16, nw))]))
    for by in range(0, 2 * gy, 4):
        for bx in range(0, 2 * gx, 4):
            unit_rows.append((1, [51 * nw + cy * (2 * gx) + cx for cy in range(by, min(by + 4, 2 * gy))
                                  for cx in range(bx, min(bx + 4, 2 * gx))]))
    units = []
    for kind, rows in unit_rows:
        ntile = (len(rows) + 15) // 16
        rows = rows + [-1] * (64 - len(rows))
        per_reader = {}
        for k, r in enumerate(rows):
            for (w, n) in readers_of.get(r, ()) if r >= 0 else ():
                per_reader.setdefault(w, {}).setdefault(k, []).append(n)
        readers = []                      # (window, {tile: slots [L][16]})
        for w in sorted(per_reader):
            occ = per_reader[w]
            tl = {}
            for t in range(KO_TILES):
                nl = max([len(occ.get(16 * t + k, ())) for k in range(16)])
                if nl:
                    sl = np.full((nl, 16), -1, dtype=np.int16)
                    for k in range(16):
                        for l, n in enumerate(occ.get(16 * t + k, ())):
                            sl[l, k] = n
                    tl[t] = sl
            readers.append((w, tl))
        intervals = []                    # (r0, r1, [job per wave: (reader slot, query halves, slots [L][16]) or None])
        if kind == 0:
            todo = sorted(readers, key=lambda r: -len(r[1]))
            while todo:
                a = todo.pop(0)
                b = None
                for cand in todo:             # a partner that touches none of a's tiles, the one with the most tiles
                    if not (set(cand[1]) & set(a[1])) and (b is None or len(cand[1]) > len(b[1])):
                        b = cand
                if b is not None:
                    todo.remove(b)
                jobs = []
                for v in range(KO_WAVES):
                    if v in a[1]:
                        jobs.append((0, 3, a[1][v]))
                    elif b is not None and v in b[1]:
                        jobs.append((1, 3, b[1][v]))
                    else:
                        jobs.append(None)
                intervals.append((a[0], b[0] if b is not None else -1, jobs))
        else:
            for i in range(0, len(readers), 2):
                pair = readers[i:i + 2]
                jobs = [(v >> 1, 1 << (v & 1), pair[v >> 1][1][0]) if (v >> 1) < len(pair) else None for v in range(KO_WAVES)]
                intervals.append((pair[0][0], pair[1][0] if len(pair) > 1 else -1, jobs))
        # a job multiplies at most two layers (the second: the 12 ring positions a window reads twice); a reader that sees a key more
        # often (grids narrower than 3 windows) gets further intervals that stage it again for the remaining layers
        more = []
        for (r0, r1, jobs) in intervals:
            depth = max([j[2].shape[0] for j in jobs if j is not None] + [0])
            for l0 in range(2, depth, 2):
                more.append((r0, r1, [(j[0], j[1], j[2][l0:l0 + 2]) if (j is not None and j[2].shape[0] > l0) else None for j in jobs]))
        intervals = [(r0, r1, [(j[0], j[1], j[2][:2]) if j is not None else None for j in jobs]) for (r0, r1, jobs) in intervals] + more
        units.append((kind, rows, ntile, intervals))
    units.sort(key=lambda u: -len(u[3]))          # long units first
    nu = len(units)
    ni = sum(len(u[3]) for u in units)
    off_units, off_int = 8, 8 + 8 * nu
    off_rows = off_int + KO_INT_REC * ni
    ko = np.zeros(off_rows + 64 * nu, dtype=np.int32)
    chunks = []
    ns = 0
    ii = 0
    for ui, (kind, rows, ntile, intervals) in enumerate(units):
        sl_u = np.full((len(intervals), KO_WAVES, 16, 2), -1, dtype=np.int16)
        rec = ko[off_units + 8 * ui: off_units + 8 * ui + 8]
        rec[:6] = (off_rows + 64 * ui, ntile, kind, ii, ii + len(intervals), ns)
        ko[off_rows + 64 * ui: off_rows + 64 * ui + 64] = rows
        for li, (r0, r1, jobs) in enumerate(intervals):
            irec = ko[off_int + KO_INT_REC * ii: off_int + KO_INT_REC * (ii + 1)]
            irec[0], irec[1] = r0, r1
            for v, job in enumerate(jobs):
                if job is None:
                    continue
                rs, qpm, sl = job
                sl_u[li, v, :, :sl.shape[0]] = sl.T
                irec[4 + 3 * v] = rs | (qpm << 2) | (sl.shape[0] << 4)
            ii += 1
        chunks.append(sl_u.reshape(-1))
        ns += sl_u.size
    ko[:5] = (nu, off_units, off_int, off_rows, ns)
    ko_slot = np.concatenate(chunks) if chunks else np.zeros(0, np.int16)
    assert ko_slot.size == ns
    ko.setflags(write=False)
    ko_slot.setflags(write=False)
    return ko, ko_slot


def ko_unit_count(h0, w0):
    """units of a geometry: what the library's grid sizing assumes (csrc/cffm_hip.hip attn_bwd_split)"""
    gy, gx = padded(h0) // WS, padded(w0) // WS
    nw = gy * gx
    return nw + 2 * ((nw + 15) // 16) + ((2 * gy + 3) // 4) * ((2 * gx + 3) // 4)
