#!/usr/bin/env python
"""Generates the BAGS intermediate files from an LVIS-format annotation json — the job of the
reference's ``tools/lvis_analyse.py`` (get_cate_gs :11-60, get_split :62-98, get_bin_weight
:449-484), without the lvis / pycocotools dependency (only ``categories[*].id`` and
``categories[*].instance_count`` are read):

    python tools/make_group_tables.py --ann data/lvis/lvis_v0.5_train.json --out data/lvis

writes ``label2binlabel.pt`` (int64 [5, C]), ``pred_slice_with0.pt`` (int64 [5, 2]),
``valsplit.pkl`` (dict of numpy int arrays incl. 'normal' / 'background' / 'all') and
``bins_cls_weight.pkl`` (the re-weighting variant's per-bin class weights).
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balancedgroupsoftmax_amd import gs_tables  # noqa: E402


def counts_from_annotation(path, num_classes=None):
    with open(path) as f:
        cats = json.load(f)['categories']
    max_id = max(int(c['id']) for c in cats)
    C = (max_id + 1) if num_classes is None else num_classes
    counts = np.zeros(C, dtype=np.int64)
    for c in cats:
        counts[int(c['id'])] = int(c['instance_count'])
    return counts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ann', required=True, help='LVIS train annotation json')
    ap.add_argument('--out', required=True, help='output directory (the configs use ./data/lvis)')
    ap.add_argument('--thresholds', default='10,100,1000')
    ap.add_argument('--num-classes', type=int, default=None, help='incl. background (LVIS v0.5: 1231)')
    a = ap.parse_args()
    thr = tuple(int(t) for t in a.thresholds.split(','))
    counts = counts_from_annotation(a.ann, a.num_classes)
    l2b, ps, split = gs_tables.build_group_tables(counts, thr)
    C = counts.shape[0]
    split = dict(split)
    split['normal'] = np.arange(1, C)
    split['background'] = np.zeros((1,), dtype=np.int64)
    split['all'] = np.arange(C)
    paths = gs_tables.save_group_tables(a.out, l2b, ps, split,
                                        bin_cls_weight=gs_tables.bin_class_weights(counts, l2b))
    print('classes (incl. bg): %d; bin widths: %s' % (C, ps[:, 1].tolist()))
    for k, v in paths.items():
        print('%-16s %s' % (k, v))


if __name__ == '__main__':
    main()
