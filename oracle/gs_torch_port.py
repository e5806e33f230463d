"""TEST / BASELINE INFRASTRUCTURE ONLY — torch-CPU port of the reference's BAGS loss path.

Restates ``GSBBoxHeadWith0.loss`` + ``.backward()`` with the same torch ops the reference
issues (``mapping[labels]``, ``narrow``, ``F.cross_entropy(reduction='none')``,
``loss * weight``, ``.sum() / avg_factor``; mmdet/models/bbox_heads/gs_bbox_head_with0.py:63-186,
mmdet/models/losses/cross_entropy_loss.py:9-19, mmdet/models/losses/utils.py:26-52), so that
its wall time on the GPU box's host cores is the reference's own CPU path (``cpu_baseline``
kind "port" in bench.py).  Pinned against the executed-reference fixtures by
tests/test_oracle_golden.py.  Never imported by the product package.
"""
import numpy as np
import torch
import torch.nn.functional as F


def sample_others(label, ratio):
    """gs_bbox_head_with0.py:63-89 (numpy global RNG, host round trips included)."""
    fg = torch.where(label > 0, torch.ones_like(label), torch.zeros_like(label))
    fg_idx = fg.nonzero(as_tuple=True)[0]
    fg_num = fg_idx.shape[0]
    if fg_num == 0:
        return torch.zeros_like(label)
    bg = 1 - fg
    bg_idx = bg.nonzero(as_tuple=True)[0]
    bg_num = bg_idx.shape[0]
    k = int(fg_num * ratio)
    if k >= bg_num:
        return torch.ones_like(label)
    sample_idx = torch.from_numpy(np.random.choice(bg_idx.cpu().numpy(), (k,), replace=False))
    fg[sample_idx] = 1
    return fg


def gs_loss(cls_score, labels, label2binlabel, pred_slice, ratio):
    """Returns the list of per-bin losses (gs_bbox_head_with0.py:91-112,134-171)."""
    B = label2binlabel.shape[0]
    out = []
    for i in range(B):
        bl = label2binlabel[i][labels]
        w = torch.ones_like(bl) if i < 1 else sample_others(bl, ratio)
        avg = max(torch.sum(w).float().item(), 1.)
        pred = cls_score.narrow(1, int(pred_slice[i, 0]), int(pred_slice[i, 1]))
        loss = F.cross_entropy(pred, bl, reduction='none') * w.float()
        out.append(loss.sum() / avg)
    return out


def gs_loss_fwd_bwd(cls_score, labels, label2binlabel, pred_slice, ratio):
    """One reference-equivalent training step of the classification loss; returns
    (losses [B] tensor, grad [N, W])."""
    z = cls_score.detach().requires_grad_(True)
    losses = gs_loss(z, labels, label2binlabel, pred_slice, ratio)
    sum(losses).backward()
    return torch.stack([l.detach() for l in losses]), z.grad
