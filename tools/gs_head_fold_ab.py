#!/usr/bin/env python
"""A/B (round 6): the head step's reduction inside the main launch (bgs_gs_head_fold 1) against the separate
gs_head_reduce_kernel launch (0) — bench.py's gs_head figure (hipGraph replay of bgs_gs_head_step + backward), interleaved.
python tools/gs_head_fold_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from balancedgroupsoftmax_amd import capi  # noqa: E402

lib = capi.load()
for n in (1024, 512, 2048):
    inp = bench.make_inputs(n, 1000, 'cuda:0')
    row = []
    for fold in (1, 0, 1, 0):
        lib.bgs_gs_head_fold(fold)
        m = bench.gs_head_metric(inp, n)
        row.append('fold=%d %.2f us (any upstream %.2f)' % (fold, m['us_per_step'], m['us_per_step_any_upstream']))
    print('N %5d | %s' % (n, ' | '.join(row)), flush=True)
lib.bgs_gs_head_fold(-1)
