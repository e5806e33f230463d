"""Helpers shared by the oracle (CPU) and HIP (GPU) parity tests."""
import json
import os

import numpy as np

from tests.golden import make_golden

GOLDEN_PATH = os.path.join(os.path.dirname(os.path.abspath(make_golden.__file__)),
                           'gs_head_golden.npz')


class Golden(object):
    def __init__(self):
        self.z = np.load(GOLDEN_PATH)
        self.cases = json.loads(bytes(self.z['__cases__']).decode())

    def names(self):
        return [c['name'] for c in self.cases]

    def case(self, name):
        return [c for c in self.cases if c['name'] == name][0]

    def get(self, name, key):
        return self.z[name + '/' + key]

    def has(self, name, key):
        return (name + '/' + key) in self.z.files


_G = None


def golden():
    global _G
    if _G is None:
        _G = Golden()
    return _G


def case_names():
    return golden().names()


def case_setup(name):
    """Regenerates tables + inputs of a golden case (no reference needed)."""
    g = golden()
    case = g.case(name)
    counts, l2b, ps, split = make_golden.case_tables(case)
    batch = make_golden.case_inputs(case, l2b, ps)
    from balancedgroupsoftmax_amd import gs_tables
    thr = tuple(case.get('thresholds', (10, 100, 1000)))
    fg_splits = [split[k] for k in gs_tables.split_keys(thr)]
    cls_weights = gs_tables.bin_class_weights(counts, l2b) if case.get('reweight') else None
    return case, l2b, ps, fg_splits, cls_weights, batch
