"""``FusedSemanticHead`` registry key (mmdet/models/mask_heads/fused_semantic_head.py:10-106): the
semantic-segmentation branch of Hybrid Task Cascade (configs/bags/gs_htc_x101_64x4d_fpn_20e_16gpu_lvis.py).

    in_i -> [bilinear resize to the fusion level, align_corners] -> 1x1 conv + ReLU --sum-->
         -> 4 x (3x3 conv + ReLU) -+-> 1x1 conv            (183-class segmentation logits)
                                   +-> 1x1 conv + ReLU     (the embedded feature the RoI heads pool)

Parameter names / shapes are the reference's (``lateral_convs.i.conv.weight`` ...,
``conv_embedding.conv.weight``, ``conv_logits.weight``).  On the GPU everything stays NHWC:

* the resize is ``bgs_resize_bilinear_nhwc_f32`` (csrc/resize_bilinear.hip), the convs the fp32
  MFMA implicit-GEMM kernel with bias + ReLU in its epilogue;
* the loss — ``nn.CrossEntropyLoss(ignore_index=255)`` over ``[N,183,H/8,W/8]`` times 0.2 — is the
  fused row-softmax kernel of the BAGS head (csrc/gs_loss.hip) run with ONE bin of 183 columns:
  ignored pixels get weight 0, the normaliser is the number of counted pixels.  The logits are
  already ``[pixels, classes]`` rows in NHWC, so no transpose is needed and the gradient comes out
  of the same launch.
"""
import torch
import torch.nn as nn

from . import functional as BF
from .backbone import ConvModule, _fold_conv_bn, cached_fold
from .registry import HEADS


@HEADS.register_module
class FusedSemanticHead(nn.Module):

    def __init__(self, num_ins, fusion_level, num_convs=4, in_channels=256, conv_out_channels=256,
                 num_classes=183, ignore_label=255, loss_weight=0.2, conv_cfg=None, norm_cfg=None):
        super().__init__()
        if conv_cfg is not None or norm_cfg is not None:
            raise NotImplementedError('FusedSemanticHead with conv_cfg / norm_cfg is outside the '
                                      'BAGS configs')
        self.num_ins, self.fusion_level, self.num_convs = num_ins, fusion_level, num_convs
        self.in_channels, self.conv_out_channels = in_channels, conv_out_channels
        self.num_classes, self.ignore_label, self.loss_weight = num_classes, ignore_label, loss_weight
        self.fp16_enabled = False
        self.lateral_convs = nn.ModuleList(ConvModule(in_channels, in_channels, 1)
                                           for _ in range(num_ins))
        self.convs = nn.ModuleList(
            ConvModule(in_channels if i == 0 else conv_out_channels, conv_out_channels, 3, padding=1)
            for i in range(num_convs))
        self.conv_embedding = ConvModule(conv_out_channels, conv_out_channels, 1)
        self.conv_logits = nn.Conv2d(conv_out_channels, num_classes, 1)

    def init_weights(self):
        """fused_semantic_head.py:83-84 (ConvModules keep their kaiming default)."""
        for m in list(self.lateral_convs) + list(self.convs) + [self.conv_embedding]:
            nn.init.kaiming_normal_(m.conv.weight, mode='fan_out', nonlinearity='relu')
            nn.init.constant_(m.conv.bias, 0)
        nn.init.kaiming_normal_(self.conv_logits.weight, mode='fan_out', nonlinearity='relu')
        nn.init.constant_(self.conv_logits.bias, 0)

    def forward(self, feats):
        """feats: the FPN outputs, NHWC maps on the GPU -> ``(mask_pred [N,h,w,num_classes],
        semantic_feat [N,h,w,C])`` NHWC.  (The NCHW torch restatement used for the state-dict /
        value checks against the reference: oracle/tensor_forms.semantic_forward.)"""
        BF._require_cuda(*feats)
        lvl = self.fusion_level
        w, b = cached_fold(self.lateral_convs[lvl].conv)
        x = BF.conv2d_autograd(feats[lvl], w, b, relu=True)
        size = (x.shape[1], x.shape[2])
        for i, feat in enumerate(feats):
            if i == lvl:
                continue
            w, b = cached_fold(self.lateral_convs[i].conv)
            x = x + BF.conv2d_autograd(BF.resize_bilinear_nhwc_autograd(feat, size), w, b, relu=True)
        first = True
        for m in self.convs:
            w, b = cached_fold(m.conv)
            x = BF.conv2d_autograd(x, w, b, pad=1, relu='consumers', mask_input=not first)
            first = False
        gate = self.num_convs > 0          # x is then a relu='consumers' output
        w, b = cached_fold(self.conv_logits)
        mask_pred = BF.conv2d_autograd(x, w, b, mask_input=gate)
        w, b = cached_fold(self.conv_embedding.conv)
        semantic_feat = BF.conv2d_autograd(x, w, b, relu=True, mask_input=gate)
        return mask_pred, semantic_feat

    def loss(self, mask_pred, labels):
        """``labels``: ``[N,1,h,w]`` (or ``[N,h,w]``) integer map, ``ignore_label`` = not counted.
        ``mask_pred`` NHWC as returned by ``forward``: cross entropy with ``ignore_index`` as ONE
        bin of the GroupSoftmax row kernel."""
        BF._require_cuda(mask_pred, labels)
        n, h, w, k = mask_pred.shape
        lab = labels.reshape(-1).to(torch.int32)
        assert lab.numel() == n * h * w, (tuple(labels.shape), tuple(mask_pred.shape))
        counted = lab != self.ignore_label
        wts = counted.to(torch.float32).view(1, -1)
        bl = torch.where(counted, lab, torch.zeros_like(lab)).view(1, -1).contiguous()
        # mean over the counted pixels (torch returns NaN when there are none; here 0)
        avg = wts.sum(dim=1).clamp(min=1.0)
        val = BF.group_softmax_loss(mask_pred.reshape(n * h * w, k), bl, [[0, k]], wts, avg)
        return val[0] * self.loss_weight
