#!/usr/bin/env python
"""Generates ``tests/golden/multiclass_nms_golden.npz`` by EXECUTING THE REFERENCE's
``multiclass_nms`` (mmdet/core/post_processing/bbox_nms.py) on CPU, with its NMS op bound to
the reference's own ``nms_cpu.cpp`` compiled from source (``oracle/build_ref.py``).

Run in the authoring container only (needs ``/root/reference``):

    python tests/golden/make_golden_det.py

Inputs are regenerated from the seed by ``oracle.det_oracle.make_multiclass_case``; stored per
case: the reference's ``det_bboxes`` / ``det_labels``.  CPU NMS suppresses on IoU >= thr
(nms_cpu.cpp:55), so the HIP parity test runs ``iou_mode=1`` against these vectors.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import build_ref, det_oracle, ref_import  # noqa: E402

OUT = os.path.join(HERE, 'multiclass_nms_golden.npz')

CASES = [
    dict(name='c31_cut', n=200, C=31, seed=101, score_thr=0.01, iou_thr=0.5, max_num=100),
    dict(name='c11_agnostic_all', n=50, C=11, seed=102, agnostic=True, score_thr=0.0, iou_thr=0.5,
         max_num=2000),
    dict(name='c1231_lvis', n=1000, C=1231, seed=103, score_thr=0.0, iou_thr=0.5, max_num=300),
    dict(name='c1231_thr', n=1000, C=1231, seed=104, score_thr=0.05, iou_thr=0.5, max_num=300),
    dict(name='c21_empty', n=64, C=21, seed=105, score_thr=1.5, iou_thr=0.5, max_num=100),
    dict(name='c5_nocap', n=300, C=5, seed=106, score_thr=0.02, iou_thr=0.3, max_num=-1,
         clusters=3),
]


def case_inputs(case):
    return det_oracle.make_multiclass_case(case['n'], case['C'], case['seed'],
                                           agnostic=case.get('agnostic', False),
                                           clusters=case.get('clusters', 12))


def main():
    ref_import.install_stubs()
    sys.modules['mmdet.ops.nms.nms_cpu'] = build_ref.load_nms_cpu()
    from mmdet.core.post_processing.bbox_nms import multiclass_nms
    out = {'__cases__': np.frombuffer(json.dumps(CASES).encode(), dtype=np.uint8)}
    for case in CASES:
        boxes, scores = case_inputs(case)
        max_num = case['max_num']
        db, dl = multiclass_nms(torch.from_numpy(boxes), torch.from_numpy(scores.copy()),
                                case['score_thr'], dict(type='nms', iou_thr=case['iou_thr']),
                                max_num if max_num >= 0 else 10 ** 9)
        out[case['name'] + '/det_bboxes'] = db.numpy().astype(np.float32)
        out[case['name'] + '/det_labels'] = dl.numpy().astype(np.int64)
        print(case['name'], tuple(db.shape))
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()
