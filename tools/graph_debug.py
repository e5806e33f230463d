#!/usr/bin/env python
"""Finds which part of the training iteration breaks hipGraph capture (run on the GPU box)."""
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from balancedgroupsoftmax_amd import train  # noqa: E402


def try_capture(name, fn):
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        print('[OK ]', name, flush=True)
        return True
    except Exception:
        print('[FAIL]', name, flush=True)
        traceback.print_exc(limit=12)
        torch.cuda.synchronize()
        return False


def main():
    dev = torch.device('cuda:0')
    st = bench.DetectorStep(dev, 0, 1, 2)
    m = st.model
    state = {}

    def feats():
        with torch.no_grad():
            state['x'] = m.extract_feat(st.img)
    try_capture('backbone+neck', feats)

    def rpn_fwd():
        with torch.no_grad():
            state['rpn'] = m.rpn_head(state['x'])
    try_capture('rpn forward', rpn_fwd)

    def rpn_loss():
        state['rl'] = m.rpn_head.loss(state['rpn'][0], state['rpn'][1], st.gt_bboxes, st.metas,
                                      m.train_cfg.rpn)
    try_capture('rpn loss', rpn_loss)

    def props():
        state['pl'] = m.rpn_head.get_bboxes(state['rpn'][0], state['rpn'][1], st.metas,
                                            m.train_cfg.rpn_proposal)
    try_capture('rpn get_bboxes (topk + NMS)', props)

    def assign():
        state['samples'] = [m._assign_and_sample(state['pl'][i][0], state['pl'][i][1],
                                                 st.gt_bboxes[i], st.gt_labels[i]) for i in range(2)]
    try_capture('rcnn assign+sample', assign)

    def head():
        s = state['samples']
        rois = torch.cat([torch.cat([x['bboxes'].new_full((512, 1), i), x['bboxes']], 1)
                          for i, x in enumerate(s)], 0)
        f = m.bbox_roi_extractor(state['x'][:4], rois)
        cls, reg = m.bbox_head(f, nhwc=True)
        t = m._bbox_targets(s)
        state['hl'] = m.bbox_head.loss(cls, reg, *t)
    try_capture('roi align + head + GS loss (forward)', head)

    def head_bwd():
        head()
        loss, _ = train.parse_losses(state['hl'])
        for p in st.params:
            p.grad = None
        loss.backward()
    try_capture('head forward + backward', head_bwd)

    def full():
        st()
    try_capture('full step (incl. clip + SGD)', full)


if __name__ == '__main__':
    main()
