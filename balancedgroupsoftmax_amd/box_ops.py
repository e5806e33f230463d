"""Box coding / RoI target helpers on the BAGS path.

Semantics follow the reference's legacy (+1 width) box convention:
* ``bbox2delta``  mmdet/core/bbox/transforms.py:6-31
* ``delta2bbox``  mmdet/core/bbox/transforms.py:34-111 (known-answer doctest :64-77)
* ``bbox2roi``    mmdet/core/bbox/transforms.py:149-168
* ``bbox_target`` mmdet/core/bbox/bbox_target.py:7-61
"""
import math

import torch


def _centre_size(b):
    w = b[..., 2] - b[..., 0] + 1.0
    h = b[..., 3] - b[..., 1] + 1.0
    cx = (b[..., 0] + b[..., 2]) * 0.5
    cy = (b[..., 1] + b[..., 3]) * 0.5
    return cx, cy, w, h


def bbox2delta(proposals, gt, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.)):
    """Encode ``gt`` relative to ``proposals``: (dx, dy, log dw, log dh), then (x - mean)/std."""
    assert proposals.size() == gt.size()
    px, py, pw, ph = _centre_size(proposals.float())
    gx, gy, gw, gh = _centre_size(gt.float())
    comps = [(gx - px) / pw, (gy - py) / ph, torch.log(gw / pw), torch.log(gh / ph)]
    # (x - mean) / std with Python scalars: no host->device constant upload (hipGraph-safe)
    return torch.stack([(c - float(m)) / float(s) for c, m, s in zip(comps, means, stds)], dim=-1)


def delta2bbox(rois, deltas, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.), max_shape=None,
               wh_ratio_clip=16 / 1000):
    """Decode ``deltas [N, 4k]`` against ``rois [N, 4]`` -> boxes ``[N, 4k]`` (x1, y1, x2, y2).

    Known answers (reference doctest): rois [[0,0,1,1]]*3 + [[5,5,5,5]], deltas
    [[0,0,0,0],[1,1,1,1],[0,0,2,-1],[.7,-1.9,-.5,.3]], max_shape (32, 32) ->
    [[0,0,1,1],[0.2817,0.2817,4.7183,4.7183],[0,0.6321,7.3891,0.3679],[5.8967,2.9251,5.5033,3.2749]].
    """
    n, k4 = deltas.shape
    d = deltas.view(n, k4 // 4, 4)
    max_ratio = abs(math.log(wh_ratio_clip))
    dx = d[..., 0] * float(stds[0]) + float(means[0])
    dy = d[..., 1] * float(stds[1]) + float(means[1])
    dw = (d[..., 2] * float(stds[2]) + float(means[2])).clamp(min=-max_ratio, max=max_ratio)
    dh = (d[..., 3] * float(stds[3]) + float(means[3])).clamp(min=-max_ratio, max=max_ratio)
    px, py, pw, ph = (t.unsqueeze(1) for t in _centre_size(rois))
    gw, gh = pw * dw.exp(), ph * dh.exp()
    gx, gy = px + pw * dx, py + ph * dy
    x1, y1 = gx - gw * 0.5 + 0.5, gy - gh * 0.5 + 0.5
    x2, y2 = gx + gw * 0.5 - 0.5, gy + gh * 0.5 - 0.5
    if max_shape is not None:
        x1 = x1.clamp(min=0, max=max_shape[1] - 1)
        y1 = y1.clamp(min=0, max=max_shape[0] - 1)
        x2 = x2.clamp(min=0, max=max_shape[1] - 1)
        y2 = y2.clamp(min=0, max=max_shape[0] - 1)
    return torch.stack([x1, y1, x2, y2], dim=-1).view(n, k4)


def bbox2roi(bbox_list):
    """list of per-image ``[n_i, >=4]`` boxes -> ``[sum n_i, 5]`` (batch_ind, x1, y1, x2, y2)."""
    parts = []
    for img_id, b in enumerate(bbox_list):
        if b.size(0) > 0:
            parts.append(torch.cat([b.new_full((b.size(0), 1), img_id), b[:, :4]], dim=-1))
        else:
            parts.append(b.new_zeros((0, 5)))
    return torch.cat(parts, 0)


def bbox_target_single(pos_bboxes, neg_bboxes, pos_gt_bboxes, pos_gt_labels, cfg, reg_classes=1,
                       target_means=(0., 0., 0., 0.), target_stds=(1., 1., 1., 1.)):
    """Positives first, then negatives (label 0); label weight 1 (or cfg.pos_weight) everywhere,
    box weights 1 on positives only."""
    num_pos, num_neg = pos_bboxes.size(0), neg_bboxes.size(0)
    n = num_pos + num_neg
    labels = pos_bboxes.new_zeros(n, dtype=torch.long)
    label_weights = pos_bboxes.new_zeros(n)
    bbox_targets = pos_bboxes.new_zeros(n, 4)
    bbox_weights = pos_bboxes.new_zeros(n, 4)
    if num_pos > 0:
        labels[:num_pos] = pos_gt_labels
        label_weights[:num_pos] = 1.0 if cfg.pos_weight <= 0 else cfg.pos_weight
        bbox_targets[:num_pos] = bbox2delta(pos_bboxes, pos_gt_bboxes, target_means, target_stds)
        bbox_weights[:num_pos] = 1
    if num_neg > 0:
        label_weights[-num_neg:] = 1.0
    return labels, label_weights, bbox_targets, bbox_weights


def bbox_target(pos_bboxes_list, neg_bboxes_list, pos_gt_bboxes_list, pos_gt_labels_list, cfg,
                reg_classes=1, target_means=(0., 0., 0., 0.), target_stds=(1., 1., 1., 1.),
                concat=True):
    outs = [bbox_target_single(p, n, g, l, cfg, reg_classes, target_means, target_stds)
            for p, n, g, l in zip(pos_bboxes_list, neg_bboxes_list, pos_gt_bboxes_list,
                                  pos_gt_labels_list)]
    cols = list(zip(*outs))
    if concat:
        cols = [torch.cat(c, 0) for c in cols]
    return tuple(cols)
