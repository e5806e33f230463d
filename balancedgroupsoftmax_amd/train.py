"""Training-step glue of the BAGS hot path (what the reference spreads over tools/train.py,
mmdet/apis/train.py and mmdet/core/utils/dist_utils.py — mmcv's Runner itself is not in the
reference tree).

* ``parse_losses``            mmdet/apis/train.py:17-34 — every key containing 'loss' is summed
  (lists are summed over their entries); no ``.item()`` per key here.
* ``select_training_param``   tools/train.py:49-91 — ``selectp``: 0 all, 1 ``bbox_head.fc_cls``
  only (every shipped BAGS config), 2 the whole bbox head, 3 the cascade stages' ``fc_cls``.
* ``allreduce_grads``         mmdet/core/utils/dist_utils.py:9-41 — ONE flat fp32 buffer, SUM
  all-reduce (RCCL over xGMI), divide by world size.  Launched asynchronously right after
  backward on the communication stream; with ``selectp=1`` the payload is 5.07 MB.
* ``DistOptimizerStep``       dist_utils.py:51-58 — zero_grad, backward, all-reduce, clip
  (max_norm=35, L2), SGD step.
"""
from collections import OrderedDict

import torch
import torch.distributed as dist


def parse_losses(losses):
    log_vars = OrderedDict()
    for name, value in losses.items():
        if isinstance(value, torch.Tensor):
            log_vars[name] = value.mean()
        elif isinstance(value, (list, tuple)):
            log_vars[name] = sum(v.mean() for v in value)
        else:
            raise TypeError('{} is not a tensor or list of tensors'.format(name))
    loss = sum(v for k, v in log_vars.items() if 'loss' in k)
    log_vars['loss'] = loss
    return loss, log_vars


def select_training_param(model, selectp):
    """Returns the list of parameters left trainable."""
    if selectp == 0:
        return [p for p in model.parameters() if p.requires_grad]
    for p in model.parameters():
        p.requires_grad = False
    if selectp == 1:
        chosen = [model.bbox_head.fc_cls]
    elif selectp == 2:
        chosen = [model.bbox_head]
    elif selectp == 3:
        chosen = [h.fc_cls for h in model.bbox_head]
    else:
        raise ValueError('selectp must be 0, 1, 2 or 3')
    out = []
    for m in chosen:
        for p in m.parameters():
            p.requires_grad = True
            out.append(p)
    return out


def build_optimizer(params, cfg):
    """``optimizer = dict(type='SGD', lr, momentum, weight_decay)`` (mmdet/apis/train.py:63-140,
    plain branch without paramwise options)."""
    cfg = dict(cfg)
    typ = cfg.pop('type')
    return getattr(torch.optim, typ)(params, **cfg)


def allreduce_grads(params, world_size, async_op=False):
    """Flatten -> one SUM all-reduce -> /world_size -> unflatten (dist_utils.py:9-41)."""
    grads = [p.grad for p in params if p.requires_grad and p.grad is not None]
    if world_size == 1 or not grads:
        return None
    flat = torch.cat([g.reshape(-1) for g in grads])
    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op)

    def finish():
        if work is not None and async_op:
            work.wait()
        flat.div_(world_size)
        off = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n

    if async_op:
        return finish
    finish()
    return None


class DistOptimizerStep(object):
    """One optimizer step with the reference's hook order (dist_utils.py:51-58)."""

    def __init__(self, params, optimizer, grad_clip=None, world_size=1):
        self.params = list(params)
        self.optimizer = optimizer
        self.grad_clip = grad_clip
        self.world_size = world_size

    def exchange_and_update(self):
        """all-reduce -> clip -> step, for gradients already produced by backward."""
        allreduce_grads(self.params, self.world_size)
        if self.grad_clip is not None:
            torch.nn.utils.clip_grad_norm_(self.params, self.grad_clip['max_norm'],
                                           self.grad_clip.get('norm_type', 2))
        self.optimizer.step()

    def __call__(self, loss):
        self.optimizer.zero_grad(set_to_none=False)
        loss.backward()
        self.exchange_and_update()
