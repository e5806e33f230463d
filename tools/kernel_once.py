#!/usr/bin/env python
"""Launches the HBM-bound helper kernels of the cfg[1] step on the operands of a real iteration, a few times each,
so that a rocprofv3 pass (kernel trace / PMC) sees them alone:  python tools/kernel_once.py [iters] [names...]
names: roi_align roi_align_bwd merge1000 merge65536 iou_assign   (default: all).  Used by tools/pmc_hbm_kernels.sh."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
argv = sys.argv[1:]
sys.argv = sys.argv[:1]
import bench  # noqa: E402
from balancedgroupsoftmax_amd import functional as BF, gs_tables  # noqa: E402


def main():
    iters = int(argv[0]) if argv else 10
    names = argv[1:] or ['roi_align', 'roi_align_bwd', 'merge1000', 'merge65536', 'iou_assign']
    dev = torch.device('cuda', 0)
    cap = bench.capture_head_inputs(dev)
    feats, rois = cap['feats'], cap['rois']
    torch.cuda.synchronize()
    if 'roi_align' in names:
        for _ in range(iters):
            out = BF.roi_align_nhwc(feats, rois, cap['strides'], cap['out_size'], cap['sample_num'],
                                    cap['finest_scale'])
        torch.cuda.synchronize()
    if 'roi_align_bwd' in names:
        out = BF.roi_align_nhwc(feats, rois, cap['strides'], cap['out_size'], cap['sample_num'], cap['finest_scale'])
        dout = torch.randn_like(out)
        dfeats = [torch.zeros_like(f) for f in feats]
        for _ in range(iters):
            BF.roi_align_nhwc_bwd(dout, rois, dfeats, cap['strides'], cap['sample_num'], cap['finest_scale'])
        torch.cuda.synchronize()
        del dfeats, dout
    counts = gs_tables.synthetic_instance_counts(bench.NUM_CLASSES, seed=0)
    l2b, ps, _ = gs_tables.build_group_tables(counts)
    c2c = gs_tables.class_to_column(l2b, ps).to(dev)
    W = int(ps[:, 1].sum())
    for name, R in (('merge1000', 1000), ('merge65536', 65536)):
        if name in names:
            z = torch.randn(R, W, device=dev)
            for _ in range(iters):
                BF.gs_merge_score(z, ps, c2c, bench.NUM_CLASSES)
            torch.cuda.synchronize()
            del z
    if 'iou_assign' in names:
        pos, neg, minpos = cap['assign_thr']
        for _ in range(iters):
            BF.iou_assign(cap['anchors'], cap['gt_cat'], cap['gt_offs'], pos, neg, minpos, valid=cap['inside'],
                          shared_boxes=True)
        torch.cuda.synchronize()
    print('kernel_once done:', names, 'x', iters, 'K =', int(rois.shape[0]))


if __name__ == '__main__':
    main()
