"""``build_*`` helpers (API of mmdet/models/builder.py:8-43): a list of configs becomes an
``nn.Sequential``; detectors receive ``train_cfg`` / ``test_cfg`` as default args."""
from torch import nn

from . import registry as R


def build(cfg, registry, default_args=None):
    if isinstance(cfg, list):
        return nn.Sequential(*[R.build_from_cfg(c, registry, default_args) for c in cfg])
    return R.build_from_cfg(cfg, registry, default_args)


def _builder(reg):
    def fn(cfg):
        return build(cfg, reg)
    return fn


build_backbone = _builder(R.BACKBONES)
build_neck = _builder(R.NECKS)
build_roi_extractor = _builder(R.ROI_EXTRACTORS)
build_shared_head = _builder(R.SHARED_HEADS)
build_head = _builder(R.HEADS)
build_loss = _builder(R.LOSSES)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return build(cfg, R.DETECTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))
