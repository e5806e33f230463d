"""``auto_fp16`` / ``force_fp32`` contract of the reference (mmdet/core/fp16/decorators.py:9-160):
when ``module.fp16_enabled`` the decorated method's tensor args are cast; the GS loss is
``@force_fp32`` (gs_bbox_head_with0.py:147), i.e. it always computes in fp32."""
import functools
import inspect

import torch


def _cast(x, src, dst):
    if isinstance(x, torch.Tensor):
        return x.to(dst) if x.dtype in src else x
    if isinstance(x, (list, tuple)):
        return type(x)(_cast(v, src, dst) for v in x)
    if isinstance(x, dict):
        return type(x)((k, _cast(v, src, dst)) for k, v in x.items())
    return x


def _make(src, dst, apply_to):
    def wrapper(old_func):
        names = list(inspect.signature(old_func).parameters)[1:]

        @functools.wraps(old_func)
        def new_func(self, *args, **kwargs):
            if not getattr(self, 'fp16_enabled', False):
                return old_func(self, *args, **kwargs)
            sel = names if apply_to is None else apply_to
            new_args = [(_cast(a, src, dst) if (i < len(names) and names[i] in sel) else a)
                        for i, a in enumerate(args)]
            new_kwargs = {k: (_cast(v, src, dst) if k in sel else v) for k, v in kwargs.items()}
            return old_func(self, *new_args, **new_kwargs)

        return new_func

    return wrapper


def auto_fp16(apply_to=None, out_fp32=False):
    return _make((torch.float32,), torch.half, apply_to)


def force_fp32(apply_to=None, out_fp16=False):
    if isinstance(apply_to, str):  # the reference writes apply_to=('cls_score') in places
        apply_to = (apply_to,)
    return _make((torch.half, torch.bfloat16), torch.float32, apply_to)
