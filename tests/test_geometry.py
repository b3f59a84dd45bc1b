"""Host logic: the product's gather tables (vss_cffm_amd/geometry.py) against the oracle's index maps."""
import numpy as np
import pytest

from oracle import cffm_oracle as O
from vss_cffm_amd import geometry as G


@pytest.mark.parametrize('h0,w0', [(8, 8), (14, 21), (13, 30), (60, 60), (64, 64), (60, 108), (7, 7), (1, 1)])
def test_tables_match_oracle_maps(h0, w0):
    key_src, q_dst = G.tables(h0, w0)
    hp, wp = O.padded_size(h0, w0)
    gy, gx = hp // 7, wp // 7
    nw = gy * gx
    assert key_src.shape == (nw, 304) and q_dst.shape == (nw, 49)
    win, ring = O.window_pixels(hp, wp), O.ring_pixels(hp, wp)
    # token row -> padded pixel, to compare with the oracle's pixel-indexed maps
    row_pix = np.zeros(nw * 49, dtype=np.int64)
    row_pix[np.arange(nw * 49)] = win.reshape(-1)
    assert np.array_equal(row_pix[key_src[:, :49]], win)
    assert np.array_equal(row_pix[key_src[:, 49:181]], ring)
    n = 181
    for (off, s, kk, pad) in G.POOLED_GROUPS:
        cells = O.unfold_cells(gy, gx, s, kk, pad)
        got = key_src[:, n:n + kk * kk]
        assert np.array_equal(got >= 0, cells >= 0)
        assert np.array_equal(np.where(got >= 0, got - off * nw, -1), cells)
        n += kk * kk
    assert n == 289 and (key_src[:, 289:] == -1).all()
    y, x = win // wp, win % wp
    assert np.array_equal(q_dst, np.where((y < h0) & (x < w0), y * w0 + x, -1))
    # every unpadded pixel is the destination of exactly one query
    d = q_dst[q_dst >= 0]
    assert d.size == h0 * w0 and np.unique(d).size == d.size


def test_row_space_is_64_rows_per_window():
    key_src, _ = G.tables(60, 60)
    assert key_src.max() < 64 * 81


@pytest.mark.parametrize('h0,w0', [(8, 8), (13, 30), (60, 60)])
def test_inverse_tables_are_the_inverse(h0, w0):
    key_src, _ = G.tables(h0, w0)
    inv_ptr, inv_idx = G.inverse_tables(h0, w0)
    nw = key_src.shape[0]
    assert inv_ptr.shape == (64 * nw + 1,) and inv_ptr[-1] == (key_src >= 0).sum() == inv_idx.size
    flat = key_src.reshape(-1)
    for row in (0, 48, 49 * nw, 50 * nw + nw // 2, 64 * nw - 1):
        got = sorted(inv_idx[inv_ptr[row]:inv_ptr[row + 1]].tolist())
        assert got == sorted(np.nonzero(flat == row)[0].tolist())
    # some pooled cells are read by nobody (e.g. the last row of the stride-3 grid, off-centre unfold): their
    # gradient is zero, the gather pass writes zeros for them
    assert (np.diff(inv_ptr) >= 0).all() and (np.diff(inv_ptr)[:49 * nw] >= 1).all()


def test_lds_row_swizzle_is_conflict_free_for_both_read_patterns():
    """ATT_ROW (csrc/cfm_attn_kernels.h): chunk' = chunk ^ f(row), f = [0,2,3,1][(row>>2)&3], on 64-byte rows over 64 four-byte
    banks.  Checked here against the two access patterns of MI355X_MICROARCH.md's LDS table: ds_read_b128 fragment reads (four
    non-contiguous 16-lane groups) and ds_read_b64_tr_b16 transposed reads (two 32-lane groups)."""
    swz = lambda row: (0x78 >> ((row >> 1) & 6)) & 3
    assert [swz(4 * q) for q in range(4)] == [0, 2, 3, 1]
    dword = lambda row, chunk, off_b: (row * 64 + 16 * (chunk ^ swz(row)) + off_b) // 4
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
              list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
    for r0 in (0, 16, 48, 288):
        for grp in groups:                                   # MFMA fragment read: lane (l15, g) -> 16 bytes, row r0 + l15, chunk g
            banks = []
            for lane in grp:
                l15, g = lane & 15, lane >> 4
                banks += [(dword(r0 + l15, g, 0) + k) % 64 for k in range(4)]
            assert len(set(banks)) == 64, (r0, grp)
    for r0 in (0, 32, 288):
        for c0 in (0, 16):                                   # att_tr_frag: lane -> 8 bytes at row r0 + 4 g + (i >> 2), column c0 + 4 (i & 3)
            for half in (0, 32):
                banks = []
                for lane in range(half, half + 32):
                    i, g = lane & 15, lane >> 4
                    row, col = r0 + 4 * g + (i >> 2), c0 + 4 * (i & 3)
                    d = dword(row, col >> 3, 2 * (col & 7))
                    banks += [d % 64, (d + 1) % 64]
                assert len(set(banks)) == 64, (r0, c0, half)
