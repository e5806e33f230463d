"""Target assignment and sampling on the device, without host synchronisation.

Restates, in fixed-shape vectorised tensor code (no ``nonzero``, no ``.item()``, no CPU
fallback, no Python loop over ground-truth boxes):

* ``bbox_overlaps``      mmdet/core/bbox/geometry.py:4-63 (legacy +1 box sizes)
* ``MaxIoUAssigner``     mmdet/core/bbox/assigners/max_iou_assigner.py:47-180 — the reference
  moves everything to the CPU when an image has more than 50 GT boxes (:76-89) and loops
  over the GTs in Python (:162-168); here step 4 ("each gt claims its best boxes, later gts
  override earlier ones") is one masked max over the GT axis.
* ``RandomSampler``      mmdet/core/bbox/samplers/{base,random}_sampler.py — the reference
  shuffles index lists with numpy on the host (4 syncs per image).  Here sampling is a top-k
  over 62-bit random integer keys: exactly ``min(n_expected, n_available)`` positives and
  ``num - n_pos_sampled`` negatives, uniformly without replacement, emitted as a FIXED-SIZE
  index list (positives first, as ``SamplingResult.bboxes`` orders them) plus a validity
  mask for the rare case that fewer than ``num`` candidates exist.

Sampled sets are random in both implementations (different RNGs): parity is distributional;
the assignment itself is deterministic and tested exactly against the reference class.
"""
import torch

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
