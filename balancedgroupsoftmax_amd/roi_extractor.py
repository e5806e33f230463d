"""``SingleRoIExtractor`` registry key (mmdet/models/roi_extractors/single_level.py:10-107).

The reference maps RoIs to levels with tensor ops, then per level: boolean mask,
``inds.any()`` (host sync), ``RoIAlign`` launch, ``roi_feats[inds] = ...`` index_put.  Here
one kernel does the level mapping and the pooling for all RoIs (csrc/roi_align.hip) on NHWC
feature maps.  Output ``[K, out, out, C]`` (bin-major) — the consuming FC permutes its weight
columns once (bbox_heads.py), results are identical to flattening the reference's
``[K, C, out, out]``.
"""
import torch.nn as nn

from . import functional as BF
from .registry import ROI_EXTRACTORS


@ROI_EXTRACTORS.register_module
class SingleRoIExtractor(nn.Module):

    def __init__(self, roi_layer, out_channels, featmap_strides, finest_scale=56):
        super().__init__()
        cfg = dict(roi_layer)
        layer_type = cfg.pop('type')
        if layer_type != 'RoIAlign':
            raise NotImplementedError('roi_layer.type=%s (the BAGS configs use RoIAlign)' % layer_type)
        self.out_size = cfg.get('out_size', 7)
        self.sample_num = cfg.get('sample_num', 2)
        self.out_channels = out_channels
        self.featmap_strides = featmap_strides
        self.finest_scale = finest_scale
        self.fp16_enabled = False

    @property
    def num_inputs(self):
        return len(self.featmap_strides)

    def init_weights(self):
        pass

    def forward(self, feats, rois, roi_scale_factor=None, out_size=None, pool=1, add_to=None):
        """``out_size`` / ``pool`` / ``add_to`` (HTC semantic fusion): pool the ``out_size * pool``
        RoIAlign grid down to ``out_size`` inside the kernel and add the result INTO ``add_to``."""
        if roi_scale_factor is not None:
            raise NotImplementedError('roi_scale_factor is not used by the BAGS configs')
        return BF.roi_align_nhwc_autograd(list(feats[:self.num_inputs]), rois,
                                          self.featmap_strides,
                                          out_size=self.out_size if out_size is None else out_size,
                                          sample_num=self.sample_num,
                                          finest_scale=self.finest_scale, pool=pool, add_to=add_to)
