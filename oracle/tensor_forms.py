"""TEST INFRASTRUCTURE ONLY — never imported by the product package.

Tensor-op ("tensor form") restatements of the host-side logic of the BAGS path: the middle hop
between the reference's own classes and the fused HIP kernels.

* pinned to the executed reference classes on CPU (tests/test_detector_host_cpu.py:
  ``bbox_overlaps`` == mmdet/core/bbox/geometry.py, ``max_iou_assign`` == ``MaxIoUAssigner``,
  ``rpn_anchor_targets`` == ``anchor_target``, the samplers' counts == ``RandomSampler``);
* the fused kernels (csrc/det_targets.hip, csrc/sampler.hip, csrc/topk.hip) are then checked
  against these forms on the GPU (tests/test_gpu_detector.py), and whole iterations against the
  executed reference detector (tests/test_gpu_e2e.py).

They used to live inside the package (assign.py, RPNHead.anchor_targets / the tensor-form loss,
TwoStageDetector._assign_and_sample / _bbox_targets, FusedSemanticHead._forward_torch, the torch
loss formulas of losses.py); the product now has ONE path (the HIP kernels) and raises on CPU
tensors.  Functions that need a module take it as their first argument.
"""
import torch
import torch.nn.functional as F

from balancedgroupsoftmax_amd.box_ops import bbox2delta, delta2bbox
from balancedgroupsoftmax_amd.losses import reduce_weighted

_KEY_MAX = 1 << 62


def bbox_overlaps(bboxes1, bboxes2):
    """IoU matrix ``[rows, cols]`` with the legacy ``+1`` widths (geometry.py:36-63)."""
    rows, cols = bboxes1.size(0), bboxes2.size(0)
    if rows * cols == 0:
        return bboxes1.new_zeros((rows, cols))
    lt = torch.max(bboxes1[:, None, :2], bboxes2[None, :, :2])
    rb = torch.min(bboxes1[:, None, 2:4], bboxes2[None, :, 2:4])
    wh = (rb - lt + 1).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1]
    area1 = (bboxes1[:, 2] - bboxes1[:, 0] + 1) * (bboxes1[:, 3] - bboxes1[:, 1] + 1)
    area2 = (bboxes2[:, 2] - bboxes2[:, 0] + 1) * (bboxes2[:, 3] - bboxes2[:, 1] + 1)
    return overlap / (area1[:, None] + area2[None, :] - overlap)


def max_iou_assign(overlaps, pos_iou_thr, neg_iou_thr, min_pos_iou=0.0, gt_max_assign_all=True,
                   valid=None):
    """``assign_wrt_overlaps`` (max_iou_assigner.py:120-180).

    overlaps ``[G, A]`` (gt x boxes); ``valid [A]`` bool: boxes that take part at all (anchors
    inside the image).  Returns ``assigned_gt_inds [A]`` int64: -1 ignore, 0 negative,
    i+1 positive for gt i, and ``max_overlaps [A]``.
    """
    G, A = overlaps.shape
    if G == 0 or A == 0:
        raise ValueError('No gt or proposals')          # max_iou_assigner.py:76-77
    if valid is not None:
        overlaps = torch.where(valid[None, :], overlaps, overlaps.new_full((), -1.0))
    assigned = overlaps.new_full((A,), -1, dtype=torch.long)
    max_overlaps, argmax_overlaps = overlaps.max(dim=0)
    gt_max_overlaps, gt_argmax = overlaps.max(dim=1)
    if isinstance(neg_iou_thr, (tuple, list)):
        lo, hi = neg_iou_thr
    else:
        lo, hi = 0.0, neg_iou_thr
    neg = (max_overlaps >= lo) & (max_overlaps < hi)
    assigned = torch.where(neg, torch.zeros_like(assigned), assigned)
    pos = max_overlaps >= pos_iou_thr
    assigned = torch.where(pos, argmax_overlaps + 1, assigned)
    # step 4: for i in range(G): if gt_max[i] >= min_pos_iou: assigned[ov[i] == gt_max[i]] = i+1
    claim = gt_max_overlaps >= min_pos_iou                       # [G]
    if gt_max_assign_all:
        hit = (overlaps == gt_max_overlaps[:, None]) & claim[:, None]
    else:
        hit = torch.zeros_like(overlaps, dtype=torch.bool)
        hit[torch.arange(G, device=overlaps.device), gt_argmax] = claim
    ids = torch.arange(1, G + 1, device=overlaps.device)[:, None]
    winner = (hit.long() * ids).max(dim=0).values                # last gt (largest i) wins
    assigned = torch.where(winner > 0, winner, assigned)
    if valid is not None:
        assigned = torch.where(valid, assigned, assigned.new_full((), -1))
    return assigned, max_overlaps


def _random_keys(n, device, generator=None):
    """62-bit random integer keys.  On the GPU without an explicit generator: the counter-based
    device RNG of the C ABI (``bgs_random_keys``) — one launch, hipGraph-replayable; with a
    generator (reproducible tests) or on the CPU: ``torch.randint``."""
    if generator is None and torch.device(device).type == 'cuda':
        from . import functional as BF
        return BF.random_keys(n, device)
    return torch.randint(0, _KEY_MAX, (n,), device=device, dtype=torch.int64,
                         generator=generator)


def sample_pos_neg_masks(assigned, num, pos_fraction, neg_pos_ub=-1, generator=None):
    """Dense form (used for the RPN, where only per-anchor weights are needed):
    returns boolean masks ``pos_sampled, neg_sampled [A]`` with exactly
    ``min(int(num*pos_fraction), n_pos)`` positives and ``min(num - n_pos_sampled, n_neg)``
    negatives (base_sampler.py:56-73), all on the device."""
    A = assigned.numel()
    dev = assigned.device
    is_pos, is_neg = assigned > 0, assigned == 0
    keys = _random_keys(A, dev, generator)
    big = torch.full_like(keys, _KEY_MAX)
    n_exp_pos = int(num * pos_fraction)
    kp = min(n_exp_pos, A)
    pos_keys = torch.where(is_pos, keys, big)
    thr_pos = torch.topk(pos_keys, kp, largest=False, sorted=True).values[kp - 1]
    pos_s = is_pos & (pos_keys <= thr_pos)        # thr == big  <=>  fewer positives than asked
    n_pos = pos_s.sum()
    n_exp_neg = num - n_pos                                         # device scalar
    if neg_pos_ub >= 0:
        ub = (neg_pos_ub * n_pos.clamp(min=1)).long()
        n_exp_neg = torch.minimum(n_exp_neg, ub)
    kn = min(num, A)
    neg_keys = torch.where(is_neg, keys, big)
    srt = torch.topk(neg_keys, kn, largest=False, sorted=True).values
    idx = (n_exp_neg - 1).clamp(min=0, max=kn - 1)
    thr_neg = srt.gather(0, idx.view(1))[0]     # (srt[idx] would call .item(): a host sync)
    neg_s = is_neg & (neg_keys <= thr_neg) & (n_exp_neg > 0)
    return pos_s, neg_s


def sample_fixed(assigned, num, pos_fraction, generator=None):
    """Index form (used for the RoI head): ``inds [num]`` into the candidate list with the
    sampled positives first, then the sampled negatives; ``is_pos [num]``, ``valid [num]``
    (False only when fewer than ``num`` candidates exist)."""
    A = assigned.numel()
    dev = assigned.device
    is_pos, is_neg = assigned > 0, assigned == 0
    keys = _random_keys(A, dev, generator)
    big = torch.full_like(keys, _KEY_MAX)
    n_exp_pos = min(int(num * pos_fraction), A)
    pos_keys = torch.where(is_pos, keys, big)
    thr_pos = torch.topk(pos_keys, n_exp_pos, largest=False, sorted=True).values[n_exp_pos - 1]
    pos_s = is_pos & (pos_keys <= thr_pos)
    # composite key: sampled positives in [0, 2^62), negatives in [2^62, 2^63), the rest excluded
    comp = torch.where(pos_s, keys, torch.where(is_neg, keys + _KEY_MAX, torch.full_like(
        keys, torch.iinfo(torch.int64).max)))
    k = min(num, A)
    vals, inds = torch.topk(comp, k, largest=False, sorted=True)
    valid = vals < torch.iinfo(torch.int64).max
    if k < num:   # static shortfall: pad by repeating the first index, flagged invalid
        pad = num - k
        inds = torch.cat([inds, inds[:1].expand(pad)])
        vals = torch.cat([vals, vals.new_full((pad,), torch.iinfo(torch.int64).max)])
        valid = torch.cat([valid, valid.new_zeros(pad)])
    return inds, (vals < _KEY_MAX), valid


# ----------------------------------------------------------------------------------------
# loss formulas (mmdet/models/losses/{cross_entropy_loss,smooth_l1_loss}.py)
# ----------------------------------------------------------------------------------------
def softmax_ce(pred, label, weight=None, reduction='mean', avg_factor=None):
    per_row = F.cross_entropy(pred, label, reduction='none')
    w = None if weight is None else weight.float()
    return reduce_weighted(per_row, w, reduction, avg_factor)


def sigmoid_bce(pred, label, weight=None, reduction='mean', avg_factor=None):
    """RPN objectness mode.  Integer class labels are expanded to one-hot over
    ``pred.size(-1)`` channels with label c>=1 -> channel c-1 (cross_entropy_loss.py:22-32)."""
    if pred.dim() != label.dim():
        # label c >= 1 -> channel c-1 (no nonzero(): that would be a host sync)
        chan = torch.arange(1, pred.size(-1) + 1, device=label.device, dtype=label.dtype)
        label = (label.view(-1, 1) == chan.view(1, -1)).to(label.dtype)
        if weight is not None:
            weight = weight.view(-1, 1).expand(weight.size(0), pred.size(-1))
    w = None if weight is None else weight.float()
    el = F.binary_cross_entropy_with_logits(pred, label.float(), w, reduction='none')
    return reduce_weighted(el, None, reduction, avg_factor)


def mask_bce(pred, target, label, reduction='mean', avg_factor=None):
    """Mask head mode: BCE on the GT-class channel only (cross_entropy_loss.py:54-61)."""
    assert reduction == 'mean' and avg_factor is None
    rows = torch.arange(pred.size(0), dtype=torch.long, device=pred.device)
    chosen = pred[rows, label].squeeze(1)
    return F.binary_cross_entropy_with_logits(chosen, target, reduction='mean')[None]


def smooth_l1(pred, target, weight=None, beta=1.0, reduction='mean', avg_factor=None):
    """0.5 d^2 / beta for |d| < beta else |d| - beta/2 (smooth_l1_loss.py:9-15)."""
    assert beta > 0
    assert pred.size() == target.size() and target.numel() > 0
    d = (pred - target).abs()
    el = torch.where(d < beta, d * d * (0.5 / beta), d - 0.5 * beta)
    return reduce_weighted(el, weight, reduction, avg_factor)




# ----------------------------------------------------------------------------------------
# sampler hooks: deterministic (generator-driven) stand-ins for the device samplers, so that the
# fused and the tensor-form paths can be compared draw for draw
# ----------------------------------------------------------------------------------------
def sampler_hooks(generator):
    """-> dict(rpn=fn(assigned_i, num, pos_fraction, neg_pos_ub) -> (pos, neg) bool masks,
    rcnn=fn(assigned, num, pos_fraction) -> (inds, is_pos, valid)) for the package's ``samplers``
    test hook."""
    return dict(
        rpn=lambda a, num, pf, ub: sample_pos_neg_masks(a, num, pf, ub, generator),
        rcnn=lambda a, num, pf: sample_fixed(a, num, pf, generator))


# ----------------------------------------------------------------------------------------
# RPN targets / loss / proposals (anchor_target.py:94-159, anchor_head.py:163-207,
# rpn_head.py:55-104) on an ``RPNHead`` module
# ----------------------------------------------------------------------------------------
def rpn_anchor_targets(head, anchors, valid, gt_bboxes, img_shape, cfg, generator=None):
    """``anchor_target_single`` for one image, dense outputs over ALL anchors: labels,
    label_weights ``[A]``, bbox_targets, bbox_weights ``[A,4]`` and the scalars n_pos, n_neg."""
    ab = cfg.allowed_border
    inside = valid
    if ab >= 0:
        img_h, img_w = img_shape[:2]
        inside = valid & (anchors[:, 0] >= -ab) & (anchors[:, 1] >= -ab) & \
            (anchors[:, 2] < img_w + ab) & (anchors[:, 3] < img_h + ab)
    ac = cfg.assigner
    overlaps = bbox_overlaps(gt_bboxes, anchors)
    assigned, _ = max_iou_assign(overlaps, ac.pos_iou_thr, ac.neg_iou_thr,
                                 ac.get('min_pos_iou', 0.0),
                                 ac.get('gt_max_assign_all', True), valid=inside)
    sc = cfg.sampler
    pos, neg = sample_pos_neg_masks(assigned, sc.num, sc.pos_fraction,
                                    sc.get('neg_pos_ub', -1), generator)
    gt_of = gt_bboxes[(assigned - 1).clamp(min=0)]
    deltas = bbox2delta(anchors, gt_of, head.target_means, head.target_stds)
    posf = pos.to(anchors.dtype)
    bbox_targets = torch.where(pos[:, None], deltas, torch.zeros_like(deltas))
    bbox_weights = posf[:, None].expand(-1, 4)
    labels = pos.long()
    pw = 1.0 if cfg.pos_weight <= 0 else cfg.pos_weight
    label_weights = posf * pw + neg.to(anchors.dtype)
    return labels, label_weights, bbox_targets, bbox_weights, pos.sum(), neg.sum()


def rpn_loss(head, cls_scores, bbox_preds, gt_bboxes, img_metas, cfg, generator=None):
    """Keys ``loss_rpn_cls`` / ``loss_rpn_bbox``: lists with one scalar per level
    (rpn_head.py:37-53, anchor_head.py:163-207); NHWC head outputs as the package produces."""
    featmap_sizes = [tuple(c.shape[1:3]) for c in cls_scores]
    dev = cls_scores[0].device
    anchor_list, valid_flag_list = head.get_anchors(featmap_sizes, img_metas, dev)
    n_lvl = [a.size(0) for a in anchor_list[0]]
    per_img = []
    n_pos_tot = n_neg_tot = 0
    for i, meta in enumerate(img_metas):
        anchors = torch.cat(anchor_list[i])
        valid = torch.cat(valid_flag_list[i])
        t = rpn_anchor_targets(head, anchors, valid, gt_bboxes[i], meta['img_shape'], cfg, generator)
        per_img.append(t[:4])
        n_pos_tot = n_pos_tot + t[4].clamp(min=1)      # max(inds.numel(), 1) per image
        n_neg_tot = n_neg_tot + t[5].clamp(min=1)
    num_total_samples = (n_pos_tot + n_neg_tot).to(torch.float32)
    stacked = [torch.stack([p[k] for p in per_img]) for k in range(4)]   # [N, A(,4)]
    losses_cls, losses_bbox = [], []
    start = 0
    lc, lb = head.loss_cls, head.loss_bbox
    for lvl, n in enumerate(n_lvl):
        sl = slice(start, start + n)
        start += n
        cs = cls_scores[lvl].reshape(-1, head.cls_out_channels).float()
        bp = bbox_preds[lvl].reshape(-1, 4).float()
        labels = stacked[0][:, sl].reshape(-1)
        lw = stacked[1][:, sl].reshape(-1)
        bt = stacked[2][:, sl].reshape(-1, 4)
        bw = stacked[3][:, sl].reshape(-1, 4)
        losses_cls.append(lc.loss_weight * sigmoid_bce(cs, labels, lw, reduction='mean',
                                                       avg_factor=num_total_samples))
        losses_bbox.append(lb.loss_weight * smooth_l1(bp, bt, bw, beta=lb.beta, reduction='mean',
                                                      avg_factor=num_total_samples))
    return dict(loss_rpn_cls=losses_cls, loss_rpn_bbox=losses_bbox)


def rpn_topk_decode(head, cls_scores, bbox_preds, img_metas, cfg):
    """The pre-NMS part of ``get_bboxes_single`` (rpn_head.py:62-85) with tensor ops: per level the
    ``nms_pre`` highest sigmoid scores, their decoded + clipped boxes -> ``[N, L, nms_pre, 5]`` and
    the per-level counts."""
    featmap_sizes = [tuple(c.shape[1:3]) for c in cls_scores]
    dev = cls_scores[0].device
    mlvl_anchors = head._level_anchors(featmap_sizes, dev)
    N, L, nmax = cls_scores[0].shape[0], len(cls_scores), cfg.nms_pre
    boxes = torch.zeros((N, L, nmax, 5), dtype=torch.float32, device=dev)
    counts = []
    for lvl in range(L):
        scores = cls_scores[lvl].reshape(N, -1).float().sigmoid()
        deltas = bbox_preds[lvl].reshape(N, -1, 4).float()
        k = min(scores.shape[1], nmax)
        top_s, top_i = scores.topk(k, dim=1)                    # sorted, descending
        anchors = mlvl_anchors[lvl][top_i]                      # [N,k,4]
        d = torch.gather(deltas, 1, top_i[..., None].expand(-1, -1, 4))
        for i in range(N):
            boxes[i, lvl, :k, :4] = delta2bbox(anchors[i], d[i], head.target_means,
                                               head.target_stds, img_metas[i]['img_shape'])
        boxes[:, lvl, :k, 4] = top_s
        counts.append(k)
    return boxes, counts


# ----------------------------------------------------------------------------------------
# RoI assignment + sampling + targets (two_stage.py:192-210, bbox_target.py:7-61) on a detector
# ----------------------------------------------------------------------------------------
def assign_and_sample(det, proposals, prop_valid, gt_bboxes, gt_labels, generator=None):
    """One image.  Returns dict of fixed-size tensors for ``num`` sampled RoIs:
    ``bboxes [num,4]``, ``is_pos``, ``valid``, ``labels``, ``gt_bboxes`` (of positives)."""
    rc = det.train_cfg.rcnn
    ac, sc = rc.assigner, rc.sampler
    boxes = proposals[:, :4]
    overlaps = bbox_overlaps(gt_bboxes, boxes)
    assigned, _ = max_iou_assign(overlaps, ac.pos_iou_thr, ac.neg_iou_thr,
                                 ac.get('min_pos_iou', 0.0),
                                 ac.get('gt_max_assign_all', True), valid=prop_valid)
    G = gt_bboxes.size(0)
    if sc.get('add_gt_as_proposals', True):
        # base_sampler.py:49-53 + AssignResult.add_gt_: GTs are prepended and own themselves
        boxes = torch.cat([gt_bboxes, boxes], 0)
        assigned = torch.cat([torch.arange(1, G + 1, device=boxes.device), assigned])
    inds, is_pos, valid = sample_fixed(assigned, sc.num, sc.pos_fraction, generator)
    a = assigned[inds]
    gi = (a - 1).clamp(min=0)
    labels = torch.where(is_pos, gt_labels[gi], torch.zeros_like(gt_labels[gi]))
    return dict(bboxes=boxes[inds], is_pos=is_pos, valid=valid, labels=labels,
                gt_bboxes=gt_bboxes[gi])


def bbox_targets(det, samples):
    """``bbox_target`` (mmdet/core/bbox/bbox_target.py:7-61) on the fixed-size samples."""
    rc = det.train_cfg.rcnn
    head = det.bbox_head
    labels, lw, bt, bw = [], [], [], []
    for s in samples:
        pos = s['is_pos'] & s['valid']
        posf = pos.float()
        d = bbox2delta(s['bboxes'], s['gt_bboxes'], head.target_means, head.target_stds)
        labels.append(torch.where(pos, s['labels'], torch.zeros_like(s['labels'])))
        pw = 1.0 if rc.pos_weight <= 0 else rc.pos_weight
        lw.append(posf * pw + (s['valid'] & ~s['is_pos']).float())
        bt.append(torch.where(pos[:, None], d, torch.zeros_like(d)))   # (not d*0: NaN-safe)
        bw.append(posf[:, None].expand(-1, 4).contiguous())
    return torch.cat(labels), torch.cat(lw), torch.cat(bt), torch.cat(bw)


# ----------------------------------------------------------------------------------------
# heads as plain torch modules (NCHW): the state-dict / shape / value checks against the reference
# ----------------------------------------------------------------------------------------
def semantic_forward(head, feats):
    """``FusedSemanticHead.forward`` (fused_semantic_head.py:86-100), NCHW."""
    lvl = head.fusion_level
    x = F.relu(head.lateral_convs[lvl].conv(feats[lvl]))
    size = tuple(x.shape[-2:])
    for i, feat in enumerate(feats):
        if i != lvl:
            feat = F.interpolate(feat, size=size, mode='bilinear', align_corners=True)
            x = x + F.relu(head.lateral_convs[i].conv(feat))
    for m in head.convs:
        x = F.relu(m.conv(x))
    return head.conv_logits(x), F.relu(head.conv_embedding.conv(x))


def semantic_loss(head, mask_pred, labels):
    """fused_semantic_head.py:102-106."""
    labels = labels.squeeze(1).long()
    return F.cross_entropy(mask_pred, labels, ignore_index=head.ignore_label) * head.loss_weight


def fcn_mask_forward(head, x, labels=None):
    """``FCNMaskHead.forward`` (fcn_mask_head.py:94-104), NCHW; ``labels``: own channel only."""
    for m in head.convs:
        x = F.relu(m.conv(x))
    pred = head.conv_logits(F.relu(head.upsample(x)))
    if labels is None:
        return pred
    ch = torch.zeros_like(labels) if head.class_agnostic else labels
    return pred[torch.arange(pred.size(0)), ch]


def htc_mask_forward(head, x, res_feat=None, return_logits=True, return_feat=True, labels=None):
    """``HTCMaskHead.forward`` (htc_mask_head.py:18-38), NCHW."""
    if res_feat is not None:
        x = x + F.relu(head.conv_res.conv(res_feat))
    for m in head.convs:
        x = F.relu(m.conv(x))
    res_out = x
    outs = []
    if return_logits:
        pred = head.conv_logits(F.relu(head.upsample(x)))
        if labels is not None:
            ch = torch.zeros_like(labels) if head.class_agnostic else labels
            pred = pred[torch.arange(pred.size(0)), ch]
        outs.append(pred)
    if return_feat:
        outs.append(res_out)
    return outs if len(outs) > 1 else outs[0]


def convfc_bbox_forward(head, x):
    """``ConvFCBBoxHead.forward`` (convfc_bbox_head.py:132-168) for the shared-FC heads, NCHW."""
    if head.with_avg_pool:
        x = head.avg_pool(x)
    x = x.reshape(x.size(0), -1)
    for fc in head.shared_fcs:
        x = F.relu(fc(x))
    cls_score = head.fc_cls(x) if head.with_cls else None
    bbox_pred = head.fc_reg(x) if head.with_reg else None
    return cls_score, bbox_pred
