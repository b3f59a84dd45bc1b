#!/usr/bin/env python
"""Golden counts of the REFERENCE's `intersect_and_union` (mmseg/core/evaluation/metrics.py:62-119, imported from
/root/reference) for the cases of tests/test_evaluation.py.  Run in the build container only:
python tests/golden/make_golden_eval.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_import as RI  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def reference_counts(pred, label, k, ignore, label_map=None, reduce_zero_label=False):
    RI.import_mmseg_models()
    from mmseg.core.evaluation.metrics import intersect_and_union
    return intersect_and_union(pred.copy(), label.copy(), k, ignore, label_map or dict(), reduce_zero_label)


def main():
    from tests.test_evaluation import cases
    d = {}
    for i, (pred, label, k, ignore, rz, lm) in enumerate(cases()):
        if lm is None and pred.size:
            for name, a in zip(('inter', 'union', 'pred', 'label'), reference_counts(pred, label, k, ignore, lm, rz)):
                d['%d/%s' % (i, name)] = np.asarray(a)
    np.savez_compressed(os.path.join(OUT, 'eval_counts.npz'), **d)
    for k, v in d.items():
        print(k, v.shape, int(v.sum()))


if __name__ == '__main__':
    main()
