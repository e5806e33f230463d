#!/usr/bin/env python
"""Generates ``tests/golden/rpn_assign_fullsize_golden.npz`` by EXECUTING THE REFERENCE'S RPN TARGET + LOSS PATH on CPU
at the BASELINE size: ``MaxIoUAssigner.assign`` (mmdet/core/bbox/assigners/max_iou_assigner.py:52-180),
``RandomSampler`` (samplers/random_sampler.py:19-55, base_sampler.py:35-78), ``anchor_target``
(mmdet/core/anchor/anchor_target.py:7-174) and ``AnchorHead.loss`` / ``RPNHead.loss``
(anchor_heads/anchor_head.py:130-207, rpn_head.py:37-53) on the 268,569 anchors of a 800 x 1344 image x 2 images x 20 GT
each (40 GT), with the shipped ``train_cfg.rpn`` (pos 0.7 / neg 0.3 / min_pos 0.3, 256 samples, pos_fraction 0.5,
allowed_border 0) and seeded head outputs.

Why: the GPU tests of ``bgs_iou_assign`` / ``bgs_rpn_loss`` compared the HIP kernels with
oracle/tensor_forms.py (builder-authored), which CPU tests pin to the reference classes — a two-hop chain.  This fixture
holds the EXECUTED reference's own answers for the kernels' inputs, so tests/test_gpu_rpn_assign.py compares kernel and
reference directly:

* ``assigned{i}``  int8 [268569]: ``AssignResult.gt_inds`` of image i in FULL anchor numbering (-1 where the reference
  never assigns: anchors outside the image, anchor_target.py:100-107; 0 = negative, k = 1-based GT index);
* ``max_overlaps{i}`` float16 is NOT stored (the ints are what the loss consumes);
* ``pos{i}`` / ``neg{i}`` int32: the anchors the reference's numpy sampler drew (recorded, replayed in the test);
* ``loss_rpn_cls`` / ``loss_rpn_bbox`` float32 [5]: the per-level losses of ``RPNHead.loss``;
* ``grad_sum{l}`` / ``grad_abs{l}`` float64 and ``grad_probe{l}`` float32: sum, sum of |.| and every 1009th element of
  d(sum of the ten losses) / d(head output of level l) in the kernels' [N, H, W, A + 4A] layout;
* RoI stage (``rcnn_inputs()``: 2000 proposals + 20 GT per image): ``rcnn_assigned{i}`` int8 [2000] = the assigner on the
  proposals (0.5 / 0.5 / 0.5), ``rcnn_pos{i}`` / ``rcnn_neg{i}`` = the 512 indices the numpy RandomSampler drew (numbering
  of cat([gt, proposals]): ``add_gt_as_proposals``), ``rcnn_rois`` / ``rcnn_labels`` / ``rcnn_label_weights`` /
  ``rcnn_bbox_targets`` / ``rcnn_bbox_weights`` = ``bbox_target`` (bbox_target.py:7-61) on them, [1024, .].

Inputs are regenerated from seeds by ``inputs()`` on both sides (numpy RandomState: platform-stable).

    python tests/golden/make_golden_rpn_assign.py          # authoring container only (~1 min)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

OUT = os.path.join(HERE, 'rpn_assign_fullsize_golden.npz')
SEED = 4242
H, W, IMGS, NGT = 800, 1344, 2, 20
STRIDES = [4, 8, 16, 32, 64]
A = 3
PROBE = 1009

RPN_HEAD = dict(in_channels=256, feat_channels=256, anchor_scales=[8], anchor_ratios=[0.5, 1.0, 2.0],
                anchor_strides=STRIDES, target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0],
                loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0))
RPN_TRAIN = dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3,
                               ignore_iof_thr=-1),
                 sampler=dict(type='RandomSampler', num=256, pos_fraction=0.5, neg_pos_ub=-1,
                              add_gt_as_proposals=False),
                 allowed_border=0, pos_weight=-1, debug=False)


def featmap_sizes():
    return [((H + s - 1) // s, (W + s - 1) // s) for s in STRIDES]


def img_meta():
    return [dict(img_shape=(800, 1333, 3), pad_shape=(H, W, 3), ori_shape=(800, 1333, 3), scale_factor=1.0,
                 flip=False) for _ in range(IMGS)]


def inputs():
    """-> (gt boxes per image [20, 4] float32, head outputs per level: cls [N, A, h, w], reg [N, 4A, h, w] float32).
    GT sides exp(U(log 16, log 400)) as in SURVEY 8(d); a third of them snapped onto anchor geometry (a GT that IS an
    anchor has IoU 1 with it and near-threshold IoUs with its neighbours: the tie / threshold cases)."""
    rs = np.random.RandomState(SEED)
    boxes = []
    for _ in range(IMGS):
        wh = np.exp(rs.uniform(np.log(16), np.log(400), size=(NGT, 2)))
        xy = rs.uniform(0, 1, size=(NGT, 2)) * np.maximum(np.array([1333., 800.]) - wh - 1, 1)
        b = np.concatenate([xy, xy + wh], 1)
        snap = rs.rand(NGT) < 0.33
        b[snap] = np.round(b[snap] / 8.0) * 8.0
        boxes.append(b.astype(np.float32))
    cls, reg = [], []
    for (h, w) in featmap_sizes():
        cls.append((rs.standard_normal((IMGS, A, h, w)) * 2.0 - 1.0).astype(np.float32))
        reg.append((rs.standard_normal((IMGS, 4 * A, h, w)) * 0.3).astype(np.float32))
    return boxes, cls, reg


RCNN_TRAIN = dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, ignore_iof_thr=-1),
                  sampler=dict(type='RandomSampler', num=512, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True),
                  pos_weight=-1, debug=False)
RCNN_MEANS, RCNN_STDS = [0., 0., 0., 0.], [0.1, 0.1, 0.2, 0.2]
NPROP = 2000


def rcnn_inputs():
    """-> (gt boxes, gt labels, proposals [2000, 5]) per image for the RoI-stage part of the fixture: the GT boxes of
    ``inputs()``, labels 1..1230, 2000 proposals of which 400 are jittered copies of GT boxes (IoUs on both sides of the
    0.5 threshold) and the rest uniform; scores descending as the RPN hands them over."""
    boxes, _, _ = inputs()
    rs = np.random.RandomState(SEED + 1)
    labels, props = [], []
    for b in boxes:
        labels.append(rs.randint(1, 1231, size=NGT).astype(np.int64))
        wh = np.exp(rs.uniform(np.log(8), np.log(500), size=(NPROP, 2)))
        xy = rs.uniform(0, 1, size=(NPROP, 2)) * np.maximum(np.array([1333., 800.]) - wh - 1, 1)
        p = np.concatenate([xy, xy + wh], 1)
        src = b[np.arange(400) % NGT]
        scale = (src[:, 2:] - src[:, :2]).repeat(2).reshape(-1, 4)[:, [0, 1, 0, 1]]
        p[:400] = src + rs.standard_normal((400, 4)) * scale * rs.choice([0.02, 0.1, 0.25], size=(400, 1))
        p[:, 2:] = np.maximum(p[:, 2:], p[:, :2] + 1)
        score = np.sort(rs.rand(NPROP))[::-1]
        props.append(np.concatenate([p, score[:, None]], 1).astype(np.float32))
    return boxes, labels, props


def rcnn_part(rec):
    """The RoI stage on the executed reference: MaxIoUAssigner (0.5 / 0.5 / 0.5) on the proposals, RandomSampler
    (512, 0.25, add_gt_as_proposals) with its numpy draws recorded, bbox_target (mmdet/core/bbox/bbox_target.py:7-61)."""
    from mmdet.core.bbox.assigners.max_iou_assigner import MaxIoUAssigner
    from mmdet.core.bbox.samplers.random_sampler import RandomSampler
    from mmdet.core.bbox.bbox_target import bbox_target
    from balancedgroupsoftmax_amd.config import to_config_dict
    boxes, labels, props = rcnn_inputs()
    ac, sc = dict(RCNN_TRAIN['assigner']), dict(RCNN_TRAIN['sampler'])
    ac.pop('type')
    sc.pop('type')
    assigner, sampler = MaxIoUAssigner(**ac), RandomSampler(**sc)
    res = []
    for i in range(IMGS):
        gtb, gtl = torch.from_numpy(boxes[i]), torch.from_numpy(labels[i])
        pr = torch.from_numpy(props[i][:, :4])
        ar = assigner.assign(pr, gtb, None, gtl)
        rec['rcnn_assigned%d' % i] = ar.gt_inds.numpy().astype(np.int8)          # proposals only (before add_gt_)
        sr = sampler.sample(ar, pr, gtb, gtl)
        rec['rcnn_pos%d' % i] = sr.pos_inds.numpy().astype(np.int32)             # numbering of cat([gt, proposals])
        rec['rcnn_neg%d' % i] = sr.neg_inds.numpy().astype(np.int32)
        res.append(sr)
        assert len(sr.pos_inds) + len(sr.neg_inds) == 512
    lab, lw, bt, bw = bbox_target([r.pos_bboxes for r in res], [r.neg_bboxes for r in res],
                                  [r.pos_gt_bboxes for r in res], [r.pos_gt_labels for r in res],
                                  to_config_dict(RCNN_TRAIN), 1, RCNN_MEANS, RCNN_STDS)
    rec['rcnn_labels'] = lab.numpy().astype(np.int16)
    rec['rcnn_label_weights'] = lw.numpy().astype(np.float32)
    rec['rcnn_bbox_targets'] = bt.numpy().astype(np.float32)
    rec['rcnn_bbox_weights'] = bw.numpy().astype(np.float32)
    rec['rcnn_rois'] = torch.cat([torch.cat([r.pos_bboxes, r.neg_bboxes]) for r in res]).numpy().astype(np.float32)
    print('rcnn: positives per image %s, labels > 0: %d of %d'
          % ([len(r.pos_inds) for r in res], int((lab > 0).sum()), lab.numel()))


def main():
    from oracle import ref_import
    from balancedgroupsoftmax_amd.config import to_config_dict
    ref_import.install_stubs()
    import importlib
    AT = importlib.import_module('mmdet.core.anchor.anchor_target')
    from mmdet.core.bbox.assigners.max_iou_assigner import MaxIoUAssigner
    from mmdet.models.anchor_heads.rpn_head import RPNHead
    rec = {}

    # ---- recorders (pure observers)
    inside_calls, assign_calls, single_calls = [], [], []
    inside_flags = AT.anchor_inside_flags

    def inside_rec(*a, **k):
        out = inside_flags(*a, **k)
        inside_calls.append(out.clone())
        return out
    AT.anchor_inside_flags = inside_rec

    assign = MaxIoUAssigner.assign

    def assign_rec(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        res = assign(self, bboxes, gt_bboxes, gt_bboxes_ignore, gt_labels)
        assign_calls.append(res.gt_inds.clone())
        return res
    MaxIoUAssigner.assign = assign_rec

    at_single = AT.anchor_target_single

    def at_single_rec(*a, **k):
        out = at_single(*a, **k)
        labels, label_weights = out[0], out[1]
        i = len(single_calls)
        rec['pos%d' % i] = torch.nonzero(labels == 1).view(-1).numpy().astype(np.int32)
        rec['neg%d' % i] = torch.nonzero((label_weights > 0) & (labels == 0)).view(-1).numpy().astype(np.int32)
        single_calls.append(i)
        return out
    AT.anchor_target_single = at_single_rec
    # rpn_head / anchor_head imported `anchor_target` by name: it resolves anchor_target_single through the module
    # globals of mmdet.core.anchor.anchor_target at call time, so the patch above is seen

    np.random.seed(SEED)                      # the reference sampler draws with the global numpy RNG
    head = RPNHead(**RPN_HEAD)
    boxes, cls, reg = inputs()
    cls_t = [torch.from_numpy(c).requires_grad_(True) for c in cls]
    reg_t = [torch.from_numpy(r).requires_grad_(True) for r in reg]
    losses = head.loss(cls_t, reg_t, [torch.from_numpy(b) for b in boxes], img_meta(), to_config_dict(RPN_TRAIN))
    assert len(assign_calls) == IMGS and len(inside_calls) == IMGS and len(single_calls) == IMGS
    total = sum(losses['loss_rpn_cls']) + sum(losses['loss_rpn_bbox'])
    total.backward()
    rec['loss_rpn_cls'] = np.array([float(t) for t in losses['loss_rpn_cls']], np.float32)
    rec['loss_rpn_bbox'] = np.array([float(t) for t in losses['loss_rpn_bbox']], np.float32)
    n_anchors = sum(h * w * A for (h, w) in featmap_sizes())
    assert n_anchors == 268569
    for i in range(IMGS):
        inside = inside_calls[i].bool().numpy()
        assert inside.shape[0] == n_anchors
        full = np.full(n_anchors, -1, np.int8)
        gi = assign_calls[i].numpy()
        assert gi.shape[0] == int(inside.sum()) and gi.max() <= NGT and gi.min() >= -1
        full[inside] = gi.astype(np.int8)
        rec['assigned%d' % i] = full
        assert (full[rec['pos%d' % i]] > 0).all() and (full[rec['neg%d' % i]] == 0).all()
    for l in range(len(STRIDES)):
        g = torch.cat([cls_t[l].grad, reg_t[l].grad], 1).permute(0, 2, 3, 1).contiguous().view(-1).numpy()
        rec['grad_sum%d' % l] = np.array([g.astype(np.float64).sum()])
        rec['grad_abs%d' % l] = np.array([np.abs(g.astype(np.float64)).sum()])
        rec['grad_probe%d' % l] = g[::PROBE].copy()
    rcnn_part(rec)
    np.savez_compressed(OUT, **rec)
    for i in range(IMGS):
        a = rec['assigned%d' % i]
        print('image %d: %d inside anchors, %d positive, %d negative, %d ignored; drew %d pos + %d neg'
              % (i, int((a >= 0).sum() + ((a == -1) & inside_calls[i].bool().numpy()).sum()), int((a > 0).sum()),
                 int((a == 0).sum()), int(((a == -1) & inside_calls[i].bool().numpy()).sum()),
                 len(rec['pos%d' % i]), len(rec['neg%d' % i])))
    print('loss_rpn_cls', rec['loss_rpn_cls'], 'loss_rpn_bbox', rec['loss_rpn_bbox'])
    print('wrote %s (%.1f KB)' % (OUT, os.path.getsize(OUT) / 1024.0))


if __name__ == '__main__':
    main()
