"""``auto_fp16`` / ``force_fp32`` contract of the reference (mmdet/core/fp16/decorators.py:9-160, the cast rule of
mmdet/core/fp16/utils.py:7-23): a no-op unless ``module.fp16_enabled``; then the selected arguments (``apply_to``, default
all NAMED positional parameters incl. ``self`` — a module is not a tensor and passes through) are cast and, with
``out_fp32`` / ``out_fp16``, the result is cast back — used on the hot path by ``SingleRoIExtractor.forward``
(single_level.py:89: ``@force_fp32(apply_to=('feats', ), out_fp16=True)``) and ``FPN.forward`` (fpn.py:101); the GS loss
is ``@force_fp32`` (gs_bbox_head_with0.py:147), i.e. it always computes in fp32.

The cast rule is the reference's, literally: EVERY tensor found in a selected argument — through mappings and iterables —
becomes the target dtype whatever it was before (``inputs.to(dst_type)``; the ``src_type`` parameter of
``cast_tensor_type`` is unused there), strings and numpy arrays pass through.  Positional arguments beyond the named
parameters (``*args``) are DROPPED by the reference's loop over ``args_info.args[:len(args)]``; so they are here.
``tests/test_formats_cpu.py`` runs both implementations on the same module and compares dtypes and values."""
import functools
from collections import abc
from inspect import getfullargspec

import numpy as np
import torch


def cast_tensor_type(inputs, src_type, dst_type):
    if isinstance(inputs, torch.Tensor):
        return inputs.to(dst_type)
    if isinstance(inputs, (str, np.ndarray)):
        return inputs
    if isinstance(inputs, abc.Mapping):
        return type(inputs)({k: cast_tensor_type(v, src_type, dst_type) for k, v in inputs.items()})
    if isinstance(inputs, abc.Iterable):
        return type(inputs)(cast_tensor_type(v, src_type, dst_type) for v in inputs)
    return inputs


def _make(name, src, dst, apply_to, cast_out):
    def wrapper(old_func):
        info = getfullargspec(old_func)

        @functools.wraps(old_func)
        def new_func(*args, **kwargs):
            if not isinstance(args[0], torch.nn.Module):
                raise TypeError('@%s can only be used to decorate the method of nn.Module' % name)
            if not getattr(args[0], 'fp16_enabled', False):
                return old_func(*args, **kwargs)
            sel = info.args if apply_to is None else apply_to
            new_args = [cast_tensor_type(a, src, dst) if n in sel else a
                        for n, a in zip(info.args[:len(args)], args)]
            new_kwargs = {k: (cast_tensor_type(v, src, dst) if k in sel else v) for k, v in kwargs.items()}
            output = old_func(*new_args, **new_kwargs)
            if cast_out:
                output = cast_tensor_type(output, dst, src)
            return output

        return new_func

    return wrapper


def auto_fp16(apply_to=None, out_fp32=False):
    """fp32 -> fp16 on the way in (decorators.py:9-82); ``out_fp32``: the result back to fp32 (:77-79)."""
    return _make('auto_fp16', torch.float, torch.half, apply_to, out_fp32)


def force_fp32(apply_to=None, out_fp16=False):
    """fp16 -> fp32 on the way in (decorators.py:86-160); ``out_fp16``: the result back to fp16 (:154-156).
    (``apply_to=('cls_score')`` — a plain string, as the reference writes in places — selects by substring there
    (``arg_name in 'cls_score'``); kept.)"""
    return _make('force_fp32', torch.half, torch.float, apply_to, out_fp16)
