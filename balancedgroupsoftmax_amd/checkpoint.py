"""Checkpoint I/O in the reference's on-disk format (mmcv.runner.checkpoint as used by
mmdet/apis/train.py:201-204 and tools/train.py:133-139): a ``torch.save`` dict
``{'meta': {...}, 'state_dict': OrderedDict, 'optimizer': {...}}``; parameter names are the
reference's (see the state-dict layout tests), optionally prefixed ``module.`` by the DDP wrapper.
"""
import collections
import time

import torch


def _strip_module(state_dict):
    if all(k.startswith('module.') for k in state_dict.keys()):
        return collections.OrderedDict((k[7:], v) for k, v in state_dict.items())
    return state_dict


def load_state_dict(module, state_dict, strict=False, logger=None):
    """mmcv semantics: copy what matches, collect what does not (a size mismatch — e.g. a
    1231-row ``fc_cls`` checkpoint into the 1236-row BAGS head — is reported, not fatal)."""
    own = module.state_dict()
    unexpected, mismatched = [], []
    for name, param in state_dict.items():
        if name not in own:
            unexpected.append(name)
            continue
        if tuple(param.shape) != tuple(own[name].shape):
            mismatched.append((name, tuple(own[name].shape), tuple(param.shape)))
            continue
        own[name].copy_(param)
    missing = sorted(set(own.keys()) - set(state_dict.keys()))
    msgs = []
    if unexpected:
        msgs.append('unexpected key in source state_dict: ' + ', '.join(unexpected))
    if missing:
        msgs.append('missing keys in source state_dict: ' + ', '.join(missing))
    for name, a, b in mismatched:
        msgs.append('size mismatch for %s: model %s vs checkpoint %s' % (name, a, b))
    if msgs and strict:
        raise RuntimeError('\n'.join(msgs))
    if msgs and logger is not None:
        logger.warning('\n'.join(msgs))
    return dict(missing=missing, unexpected=unexpected, mismatched=mismatched)


def load_checkpoint(model, filename, map_location='cpu', strict=False, logger=None):
    """Returns the checkpoint dict (``meta`` etc.); accepts a bare state_dict file as well."""
    ckpt = torch.load(filename, map_location=map_location)
    if isinstance(ckpt, dict) and 'state_dict' in ckpt:
        state_dict = ckpt['state_dict']
    elif isinstance(ckpt, (dict, collections.OrderedDict)):
        state_dict, ckpt = ckpt, dict(state_dict=ckpt)
    else:
        raise RuntimeError('No state_dict found in checkpoint file %s' % filename)
    with torch.no_grad():
        ckpt['load_report'] = load_state_dict(model, _strip_module(state_dict), strict, logger)
    return ckpt


def save_checkpoint(model, filename, optimizer=None, meta=None):
    meta = dict(meta or {})
    meta.setdefault('time', time.asctime())
    meta.setdefault('framework', 'balancedgroupsoftmax_amd')
    sd = collections.OrderedDict((k, v.detach().cpu()) for k, v in model.state_dict().items())
    ckpt = dict(meta=meta, state_dict=sd)
    if optimizer is not None:
        ckpt['optimizer'] = optimizer.state_dict()
    torch.save(ckpt, filename)
    return filename
