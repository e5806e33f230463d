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
beta): the arithmetic runs in the fused HIP kernels (functional.py).  Called directly (a plain
``BBoxHead``) they run HIP kernels as well — there is no tensor-op / CPU evaluation in the
product; the torch formulas that pin the kernels live in oracle/tensor_forms.py.
"""
import torch
import torch.nn as nn

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


def _avg_and_scale(n_rows, reduction, avg_factor):
    """(avg passed to the kernel, scalar to divide the kernel's result by) for the
    ``weight_reduce_loss`` rules above; 'none' has no fused form."""
    if reduction == 'none':
        raise NotImplementedError("reduction='none': the fused HIP losses return reduced scalars")
    if avg_factor is not None:
        if reduction != 'mean':
            raise ValueError('avg_factor can not be used with reduction="sum"')
        if isinstance(avg_factor, torch.Tensor):
            return 1.0, avg_factor.to(torch.float32)
        return float(avg_factor), None
    if reduction == 'mean':
        return float(n_rows), None
    if reduction == 'sum':
        return 1.0, None
    raise ValueError('unknown reduction %r' % (reduction,))


@LOSSES.register_module
class CrossEntropyLoss(nn.Module):
    """Registry key + ctor kwargs of the reference (cross_entropy_loss.py:64-103).  Inside the BAGS
    heads / RPN / mask heads the module is a configuration carrier (``loss_weight``,
    ``use_sigmoid``): the arithmetic runs in the fused kernels (``bgs_gs_loss_fwd_bwd``,
    ``bgs_rpn_loss``, ``bgs_mask_bce``).  Called directly it runs on the GPU too:

    * softmax mode (plain ``BBoxHead``): ONE bin of the GroupSoftmax row kernel spanning all
      ``K`` columns — ``sum_r w_r (lse_r - z_r[label_r]) / avg``;
    * sigmoid / mask mode: fused with their producers (``RPNHead.loss``,
      ``FCNMaskHead.loss_from_features``) — a direct call raises, pointing there.

    There is no CPU / tensor-op path (the torch formulas are in oracle/tensor_forms.py)."""

    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', loss_weight=1.0):
        super().__init__()
        if use_sigmoid and use_mask:
            raise AssertionError('use_sigmoid and use_mask are mutually exclusive')
        self.use_sigmoid, self.use_mask = use_sigmoid, use_mask
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None,
                **kwargs):
        from . import functional as BF
        assert reduction_override in _REDUCTIONS
        red = reduction_override or self.reduction
        if self.use_sigmoid:
            raise NotImplementedError('CrossEntropyLoss(use_sigmoid=True) is evaluated inside '
                                      'RPNHead.loss (bgs_rpn_loss: targets + BCE + SmoothL1 fused)')
        if self.use_mask:
            raise NotImplementedError('CrossEntropyLoss(use_mask=True) is evaluated inside '
                                      'FCNMaskHead.loss_from_features (bgs_mask_bce: GT-channel '
                                      'logits + BCE fused)')
        BF._require_cuda(cls_score, label, weight)
        n, k = cls_score.shape
        avg, div = _avg_and_scale(n, red, avg_factor)
        wts = (torch.ones((1, n), dtype=torch.float32, device=cls_score.device) if weight is None
               else weight.to(torch.float32).reshape(1, n).contiguous())
        bl = label.to(torch.int32).reshape(1, n).contiguous()
        avg_t = torch.full((1,), avg, dtype=torch.float32, device=cls_score.device)
        val = BF.group_softmax_loss(cls_score, bl, [[0, k]], wts, avg_t)[0]
        if div is not None:
            val = val / div
        return self.loss_weight * val


@LOSSES.register_module
class SmoothL1Loss(nn.Module):
    """Registry key + ctor kwargs of the reference (smooth_l1_loss.py:18-45): ``0.5 d^2 / beta`` for
    ``|d| < beta`` else ``|d| - beta / 2``.  The box heads read ``beta`` / ``loss_weight`` and run
    ``bgs_bbox_smooth_l1_fwd_bwd`` on the class-indexed ``[N, 4C]`` predictions; a direct call on
    ``pred / target / weight [P, 4]`` runs the same kernel in its class-agnostic form."""

    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None,
                **kwargs):
        from . import functional as BF
        assert reduction_override in _REDUCTIONS
        red = reduction_override or self.reduction
        BF._require_cuda(pred, target, weight)
        assert pred.dim() == 2 and pred.shape[1] == 4 and pred.size() == target.size() and \
            target.numel() > 0, (tuple(pred.shape), tuple(target.shape))
        avg, div = _avg_and_scale(pred.numel(), red, avg_factor)
        w = torch.ones_like(pred, dtype=torch.float32) if weight is None else weight
        rows = torch.ones((pred.shape[0],), dtype=torch.int64, device=pred.device)
        val = BF.bbox_smooth_l1_loss(pred, rows, target, w, 1, beta=self.beta, avg_factor=avg,
                                     loss_weight=self.loss_weight)
        return val if div is None else val / div


def accuracy(pred, target, topk=1):
    """Top-k accuracy in percent (mmdet/models/losses/accuracy.py); ``acc`` key of BBoxHead.loss."""
    ks = (topk,) if isinstance(topk, int) else tuple(topk)
    top = pred.topk(max(ks), dim=1).indices                     # [N, maxk]
    hit = top.eq(target.view(-1, 1))                            # [N, maxk]
    res = [hit[:, :k].any(dim=1).float().sum().mul(100.0 / pred.size(0)).view(1) for k in ks]
    return res[0] if isinstance(topk, int) else res
