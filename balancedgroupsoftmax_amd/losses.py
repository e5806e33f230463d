"""Loss modules behind the reference's ``LOSSES`` registry keys.

Registry keys / ctor kwargs / call signature follow the reference
(``CrossEntropyLoss``: mmdet/models/losses/cross_entropy_loss.py:64-103,
``SmoothL1Loss``: mmdet/models/losses/smooth_l1_loss.py:18-45); the reduction rule is the
reference's ``weight_reduce_loss`` (mmdet/models/losses/utils.py:26-52):

    weighted = elementwise * weight
    avg_factor given:  'mean' -> weighted.sum() / avg_factor ; 'none' -> weighted ;
                       'sum'  -> ValueError
    avg_factor None :  plain none / mean / sum

Inside ``GSBBoxHeadWith0.loss`` these modules are *configuration carriers* (loss_weight,
beta): the arithmetic runs in the fused HIP kernels (functional.py).  Called directly
(e.g. a plain ``BBoxHead`` or the RPN's sigmoid mode) they evaluate the same formulas
with tensor ops.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .registry import LOSSES

_REDUCTIONS = (None, 'none', 'mean', 'sum')


def reduce_weighted(elementwise, weight=None, reduction='mean', avg_factor=None):
    """The reference's ``weight_reduce_loss`` contract (see module docstring).

    Known answers (reference doctest, losses/utils.py:67-83) for |pred-target| with
    pred=[0,2,3], target=[1,1,1]: mean -> 1.3333; weight [1,0,1] -> 1.0;
    'none' -> [1,1,2]; weight + avg_factor=2 -> 1.5.
    """
    out = elementwise if weight is None else elementwise * weight
    if avg_factor is not None:
        if reduction == 'mean':
            return out.sum() / avg_factor
        if reduction == 'none':
            return out
        raise ValueError('avg_factor can not be used with reduction="sum"')
    if reduction == 'mean':
        return out.mean()
    if reduction == 'sum':
        return out.sum()
    if reduction == 'none':
        return out
    raise ValueError('unknown reduction %r' % (reduction,))


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


@LOSSES.register_module
class CrossEntropyLoss(nn.Module):

    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', loss_weight=1.0):
        super().__init__()
        if use_sigmoid and use_mask:
            raise AssertionError('use_sigmoid and use_mask are mutually exclusive')
        self.use_sigmoid, self.use_mask = use_sigmoid, use_mask
        self.reduction, self.loss_weight = reduction, loss_weight
        self.cls_criterion = (sigmoid_bce if use_sigmoid else
                              mask_bce if use_mask else softmax_ce)

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None,
                **kwargs):
        assert reduction_override in _REDUCTIONS
        red = reduction_override or self.reduction
        return self.loss_weight * self.cls_criterion(cls_score, label, weight, reduction=red,
                                                     avg_factor=avg_factor, **kwargs)


def smooth_l1(pred, target, weight=None, beta=1.0, reduction='mean', avg_factor=None):
    """0.5 d^2 / beta for |d| < beta else |d| - beta/2 (smooth_l1_loss.py:9-15)."""
    assert beta > 0
    assert pred.size() == target.size() and target.numel() > 0
    d = (pred - target).abs()
    el = torch.where(d < beta, d * d * (0.5 / beta), d - 0.5 * beta)
    return reduce_weighted(el, weight, reduction, avg_factor)


@LOSSES.register_module
class SmoothL1Loss(nn.Module):

    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None,
                **kwargs):
        assert reduction_override in _REDUCTIONS
        red = reduction_override or self.reduction
        return self.loss_weight * smooth_l1(pred, target, weight, beta=self.beta, reduction=red,
                                            avg_factor=avg_factor, **kwargs)


def accuracy(pred, target, topk=1):
    """Top-k accuracy in percent (mmdet/models/losses/accuracy.py); ``acc`` key of BBoxHead.loss."""
    ks = (topk,) if isinstance(topk, int) else tuple(topk)
    top = pred.topk(max(ks), dim=1).indices                     # [N, maxk]
    hit = top.eq(target.view(-1, 1))                            # [N, maxk]
    res = [hit[:, :k].any(dim=1).float().sum().mul(100.0 / pred.size(0)).view(1) for k in ks]
    return res[0] if isinstance(topk, int) else res
