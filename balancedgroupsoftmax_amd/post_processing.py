"""Test-time post-processing: multi-class NMS for 1230 LVIS classes in ONE batched launch.

Mirrors ``multiclass_nms`` (mmdet/core/post_processing/bbox_nms.py:6-66) and ``bbox2result``
(mmdet/core/bbox/transforms.py:181-199).  The reference loops over the classes in Python —
for LVIS that is up to 1230 boolean-mask / cat / NMS-kernel launches and as many host
synchronisations per image (``cls_inds.any()``, the ``nonzero`` inside the NMS wrapper).  Here
the classes are the batch dimension of the NMS kernel pair in ``csrc/nms.hip``:

1. one sort of the ``[C-1, n]`` score matrix (descending, sub-threshold entries pushed last)
   gives every class its candidate list and count;
2. one gather builds the ``[C-1, n, 5]`` problem array;
3. ``bgs_nms_batched`` suppresses all classes at once (IoU ``>`` thr as nms_kernel.cu:60 does;
   ``iou_mode=1`` gives the ``>=`` of nms_cpu.cpp:55 for parity with a CPU run of the reference);
4. one top-k over the survivors applies ``max_per_img``.

The only host synchronisation is the final size of the result (the output is dynamic-shaped
by contract).
"""
import numpy as np
import torch

from . import functional as BF


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None,
                   iou_mode=0):
    """``multi_bboxes [n, 4*C]`` or ``[n, 4]``, ``multi_scores [n, C]`` (column 0 = background,
    ignored).  Returns ``(det_bboxes [k, 5], det_labels [k])`` with 0-based labels; order as the
    reference: class-major (original row order inside a class) when nothing is cut, by descending score when ``max_num`` cuts."""
    cfg = dict(nms_cfg)
    nms_type = cfg.pop('type', 'nms')
    if nms_type != 'nms':
        raise NotImplementedError('only type="nms" is used by the BAGS configs (got %r)' % nms_type)
    iou_thr = float(cfg.pop('iou_thr'))
    n, C = multi_scores.shape
    dev = multi_scores.device
    P = C - 1
    if n == 0 or P <= 0:
        return multi_bboxes.new_zeros((0, 5)), multi_bboxes.new_zeros((0,), dtype=torch.long)
    scores = multi_scores[:, 1:].t().float()                        # [P, n]
    if score_factors is not None:
        scores = scores * score_factors.view(1, n).float()
        live = multi_scores[:, 1:].t() > score_thr                  # threshold is on the raw score
    else:
        live = scores > score_thr
    counts = live.sum(dim=1).to(torch.int32)
    order_key = torch.where(live, scores, scores.new_full((), -float('inf')))
    srt, idx = torch.sort(order_key, dim=1, descending=True, stable=True)   # [P, n]
    if multi_bboxes.shape[1] == 4:
        boxes = multi_bboxes.float()[idx]                            # [P, n, 4]
    else:
        per_cls = multi_bboxes.float().view(n, C, 4)[:, 1:].permute(1, 0, 2)   # [P, n, 4] view
        boxes = torch.gather(per_cls, 1, idx[..., None].expand(-1, -1, 4))
    dets = torch.cat([boxes, torch.gather(scores, 1, idx)[..., None]], dim=2).contiguous()
    keep, keep_n = BF.nms_batched(dets, counts, iou_thr, iou_mode=iou_mode, max_keep=n)
    slot_ok = torch.arange(n, device=dev).view(1, n) < keep_n.view(P, 1)
    kept = torch.gather(dets, 1, keep.long().clamp(min=0, max=n - 1)[..., None].expand(-1, -1, 5))
    total = int(keep_n.sum())                       # the one sync: the result is dynamic-shaped
    if total == 0:
        return multi_bboxes.new_zeros((0, 5)), multi_bboxes.new_zeros((0,), dtype=torch.long)
    labels_all = torch.arange(P, device=dev).view(P, 1).expand(P, n)
    if max_num < 0 or total <= max_num:
        # class-major; inside a class the survivors keep their ORIGINAL row order (both
        # nms_cpu.cpp:58 and nms_kernel.cu:127-130 return ascending input indices)
        orig = torch.gather(idx, 1, keep.long().clamp(min=0, max=n - 1))
        key = (labels_all * n + orig)[slot_ok]
        perm = torch.argsort(key)
        return kept[slot_ok][perm], labels_all[slot_ok][perm]
    flat_scores = torch.where(slot_ok, kept[..., 4], kept.new_full((), -float('inf'))).reshape(-1)
    _, top = flat_scores.topk(max_num)
    return kept.view(-1, 5)[top], labels_all.reshape(-1)[top]


def bbox2result(bboxes, labels, num_classes):
    """``[k,5]`` + ``[k]`` -> list of ``num_classes - 1`` float32 arrays (transforms.py:181-199)."""
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes - 1)]
    b = bboxes.detach().cpu().numpy()
    lab = labels.detach().cpu().numpy()
    return [b[lab == i, :] for i in range(num_classes - 1)]
