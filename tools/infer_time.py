#!/usr/bin/env python
"""Test-time latency of cfg[1] (simple_test, one 800x1344 image, 1000 proposals, 1230-class
batched NMS, max 300 dets) — informational, not the headline metric.

    python tools/infer_time.py [iters]
"""
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import balancedgroupsoftmax_amd as bgs  # noqa: E402
from balancedgroupsoftmax_amd.config import to_config_dict  # noqa: E402
from bench import detector_cfg  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    model_cfg, train_cfg = detector_cfg(tempfile.mkdtemp(prefix='bgs_tables_'))
    test_cfg = dict(rpn=dict(nms_across_levels=False, nms_pre=1000, nms_post=1000, max_num=1000,
                             nms_thr=0.7, min_bbox_size=0),
                    rcnn=dict(score_thr=0.0, nms=dict(type='nms', iou_thr=0.5), max_per_img=300))
    model = bgs.build_detector(to_config_dict(model_cfg), train_cfg=None,
                               test_cfg=to_config_dict(test_cfg)).to(dev).eval()
    with torch.no_grad():
        model.bbox_head.fc_cls.weight.mul_(30.0)
    img = torch.randn(1, 3, 800, 1344, device=dev)
    metas = [dict(img_shape=(800, 1333, 3), pad_shape=(800, 1344, 3), ori_shape=(800, 1333, 3),
                  scale_factor=1.0, flip=False)]
    for _ in range(3):
        res = model(img, metas, return_loss=False, rescale=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        res = model(img, metas, return_loss=False, rescale=True)      # ends in a D2H copy
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    # the same loop with the NEXT image's trunk launched ahead on its own streams (train.TrunkPipeline(inference=True)):
    # same detections, the trunk of image i + 1 beside image i's RPN / NMS / RoI head / 1230-class NMS / D2H copy
    from balancedgroupsoftmax_amd import train
    import numpy as np
    piped = {}
    for depth in (2, 3):
        pipe = train.TrunkPipeline(model, depth=depth, inference=True)
        for _ in range(pipe.depth - 1):
            pipe.push(img)
        for _ in range(3):
            feats = pipe.take()
            pipe.push(img)
            res2 = model(img, metas, return_loss=False, rescale=True, feats=feats)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            feats = pipe.take()
            pipe.push(img)
            res2 = model(img, metas, return_loss=False, rescale=True, feats=feats)
        torch.cuda.synchronize()
        piped[depth] = (time.perf_counter() - t0) / iters * 1e3
        pipe.drain()
        torch.cuda.synchronize()
        assert len(res2) == len(res) and all(np.array_equal(a, b) for a, b in zip(res, res2)), 'pipelined detections differ'
    # device-only portion of the post-processing
    from balancedgroupsoftmax_amd.post_processing import multiclass_nms
    with torch.no_grad():
        x = model.extract_feat(img)
        props = model.simple_test_rpn(x, metas, model.test_cfg.rpn)
        _, _, scores = model.simple_test_bboxes(x, metas, props, model.test_cfg.rcnn)
        boxes = torch.rand(1000, 4 * 1231, device=dev) * 500
        boxes = torch.cat([boxes.view(1000, 1231, 4)[..., :2],
                           boxes.view(1000, 1231, 4)[..., :2] + 60], -1).view(1000, -1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        multiclass_nms(boxes, scores, 0.0, dict(type='nms', iou_thr=0.5), 300)
    torch.cuda.synchronize()
    ms_nms = (time.perf_counter() - t0) / iters * 1e3
    best = min(piped.values())
    print('{"simple_test_ms_per_img": %.3f, "img_per_s": %.2f, "multiclass_nms_1230x1000_ms": %.3f, '
          '"dets": %d, "pipelined_ms_per_img": %.3f, "pipelined_img_per_s": %.2f, "pipelined_by_depth": %s, '
          '"pipelined_note": "train.TrunkPipeline(inference=True): the next image\'s trunk on its own streams beside this '
          'image\'s RPN / NMS / RoI head / multiclass NMS / D2H copy; identical detections (asserted)"}'
          % (ms, 1e3 / ms, ms_nms, sum(r.shape[0] for r in res), best, 1e3 / best,
             str({k: round(v, 3) for k, v in piped.items()}).replace("'", '"').replace('2:', '"2":').replace('3:', '"3":')))


if __name__ == '__main__':
    main()
