"""balancedgroupsoftmax_amd — MI355X-native Balanced Group Softmax (BAGS) detection hot path.

Hand-written gfx950 HIP kernels behind a C ABI (``include/bgs.h`` -> ``libbgs.so``), driven
from PyTorch-ROCm through the reference's own registry keys and config schema.
"""
from . import (backbone, bbox_heads, checkpoint, config, detectors, gs_tables, losses,  # noqa: F401
               mask_heads, roi_extractor, rpn, semantic_head, train)  # (imports populate the registries)
from .builder import (build_backbone, build_detector, build_head, build_loss, build_neck,
                      build_roi_extractor, build_shared_head)
from .config import Config, ConfigDict
from .registry import (BACKBONES, DETECTORS, HEADS, LOSSES, NECKS, ROI_EXTRACTORS, SHARED_HEADS,
                       Registry, build_from_cfg)

__version__ = '0.1.0'

__all__ = ['BACKBONES', 'DETECTORS', 'HEADS', 'LOSSES', 'NECKS', 'ROI_EXTRACTORS',
           'SHARED_HEADS', 'Registry', 'build_from_cfg', 'build_backbone', 'build_detector',
           'build_head', 'build_loss', 'build_neck', 'build_roi_extractor', 'build_shared_head',
           'Config', 'ConfigDict']
