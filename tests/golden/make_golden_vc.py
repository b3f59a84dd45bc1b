#!/usr/bin/env python
"""Golden VC_n values of the REFERENCE's `get_common` (/root/reference/VC_perclip.py:62-78).

That file is a script (it reads its dataset at import), so it cannot be imported; this generator parses it with `ast`,
takes ONLY the `get_common` function definition, executes that definition where it lies (nothing is copied into the repo)
and stores what it returns for the seeded label / prediction videos of `vc_cases()`.  Run in the build container only:
python tests/golden/make_golden_vc.py"""
import ast
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_import as RI  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def reference_get_common():
    path = os.path.join(RI.REF_ROOT, 'VC_perclip.py')
    tree = ast.parse(open(path).read(), filename=path)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'get_common']
    assert len(fn) == 1
    ns = {'np': np}
    exec(compile(ast.Module(body=fn, type_ignores=[]), path, 'exec'), ns)
    return ns['get_common']


def vc_cases():
    """(name, gt [F,h,w] int64, pred [F,h,w] int64): seeded videos whose labels drift slowly over time."""
    out = []
    for name, seed, f, h, w, k, drift, noise in (('a', 9, 20, 24, 31, 6, 0.04, 0.10), ('b', 10, 33, 17, 40, 124, 0.01, 0.02),
                                                 ('c', 11, 18, 8, 8, 2, 0.10, 0.30)):
        rs = np.random.RandomState(seed)
        base = rs.randint(0, k, size=(h, w))
        gt = np.stack([np.where(rs.rand(h, w) < drift * t, rs.randint(0, k, size=(h, w)), base) for t in range(f)]).astype(np.int64)
        pred = np.where(rs.rand(f, h, w) < noise, rs.randint(0, k, size=(f, h, w)), gt).astype(np.int64)
        out.append((name, gt, pred))
    return out


def vc_large_case():
    """one larger video (many workgroups per frame pair in k_vc_counts): 40 frames of 120 x 160, 124 classes"""
    rs = np.random.RandomState(12)
    f, h, w = 40, 120, 160
    base = rs.randint(0, 124, size=(h, w))
    gt = np.stack([np.where(rs.rand(h, w) < 0.02 * t, rs.randint(0, 124, size=(h, w)), base) for t in range(f)]).astype(np.int64)
    pred = np.where(rs.rand(f, h, w) < 0.1, rs.randint(0, 124, size=(f, h, w)), gt).astype(np.int64)
    return 'large', gt, pred


CLIP_NUMS = (1, 2, 8, 16)    # VC_8 / VC_16 are the reference's two calls (VC_perclip.py:128-129)


def main():
    get_common = reference_get_common()
    d = {}
    for name, gt, pred in vc_cases():
        f, h, w = gt.shape
        for n in CLIP_NUMS + (f, f + 3):
            with np.errstate(invalid='ignore', divide='ignore'):
                d['%s/%d' % (name, n)] = np.asarray(get_common(list(gt), list(pred), n, h, w), dtype=np.float64)
    name, gt, pred = vc_large_case()
    for n in (8, 16):
        with np.errstate(invalid='ignore', divide='ignore'):
            d['%s/%d' % (name, n)] = np.asarray(get_common(list(gt), list(pred), n, gt.shape[1], gt.shape[2]), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'vc_counts.npz'), **d)
    for k, v in d.items():
        print(k, v.shape, float(np.nanmean(v)) if v.size else None)


if __name__ == '__main__':
    main()
