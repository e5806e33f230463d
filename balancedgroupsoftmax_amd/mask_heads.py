"""Mask head of the BAGS Mask R-CNN behind the reference's ``HEADS`` key ``FCNMaskHead``
(mmdet/models/mask_heads/fcn_mask_head.py:14-123; cfg 4 = configs/bags/gs_mask_rcnn_r50_fpn_1x_lvis.py).

Parameter names / shapes are the reference's (``convs.i.conv.weight``, ``upsample.weight``
``[in, out, 2, 2]``, ``conv_logits.weight`` ``[num_classes, 256, 1, 1]``).  On the GPU:

* the four 3x3 convs (+ReLU) run in the fp32-MFMA implicit-GEMM kernel on NHWC RoI features;
* the 2x2 / stride-2 deconv is ONE 1x1 conv to ``4 * C`` channels (each input pixel owns its 2x2
  output block) followed by a pixel shuffle, bias + ReLU fused;
* ``conv_logits`` is evaluated only for the channel the loss (or the detection) reads —
  ``bgs_mask_bce`` / ``bgs_mask_gt_logits`` — instead of producing ``[P, 1231, 28, 28]`` (988 MB
  for 256 RoIs) and gathering one channel of it;
* mask targets are cut from bitmaps resident in HBM by ``bgs_mask_target`` (the reference goes
  through the host and OpenCV per RoI).
"""
import torch
import torch.nn as nn

from . import functional as BF
from .backbone import ConvModule, _FoldCache, _fold_conv_bn, cached_fold
from .builder import build_loss
from .registry import HEADS


@HEADS.register_module
class FCNMaskHead(nn.Module):

    def __init__(self, num_convs=4, roi_feat_size=14, in_channels=256, conv_kernel_size=3,
                 conv_out_channels=256, upsample_method='deconv', upsample_ratio=2, num_classes=81,
                 class_agnostic=False, conv_cfg=None, norm_cfg=None,
                 loss_mask=dict(type='CrossEntropyLoss', use_mask=True, loss_weight=1.0)):
        super().__init__()
        if upsample_method != 'deconv' or upsample_ratio != 2 or norm_cfg is not None or \
                conv_cfg is not None:
            raise NotImplementedError('FCNMaskHead variant outside the BAGS configs (deconv x2, no '
                                      'norm)')
        self.num_convs, self.in_channels = num_convs, in_channels
        self.conv_kernel_size, self.conv_out_channels = conv_kernel_size, conv_out_channels
        self.upsample_method, self.upsample_ratio = upsample_method, upsample_ratio
        self.num_classes, self.class_agnostic = num_classes, class_agnostic
        self.fp16_enabled = False
        self.loss_mask = build_loss(loss_mask)
        pad = (conv_kernel_size - 1) // 2
        self.convs = nn.ModuleList(
            ConvModule(in_channels if i == 0 else conv_out_channels, conv_out_channels,
                       conv_kernel_size, padding=pad) for i in range(num_convs))
        up_in = conv_out_channels if num_convs > 0 else in_channels
        self.upsample = nn.ConvTranspose2d(up_in, conv_out_channels, 2, stride=2)
        self.conv_logits = nn.Conv2d(conv_out_channels, 1 if class_agnostic else num_classes, 1)
        self.relu = nn.ReLU(inplace=True)

    def init_weights(self):
        """fcn_mask_head.py:86-92 (convs keep the ConvModule default: kaiming)."""
        for m in self.convs:
            nn.init.kaiming_normal_(m.conv.weight, mode='fan_out', nonlinearity='relu')
            nn.init.constant_(m.conv.bias, 0)
        for m in (self.upsample, self.conv_logits):
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            nn.init.constant_(m.bias, 0)

    # -- features -----------------------------------------------------------------------------
    def _deconv_as_conv(self):
        """ConvTranspose2d(k=2, s=2) weight ``[Cin, Cout, 2, 2]`` -> 1x1 conv ``[(a,b,co), 1, 1, Cin]``
        (differentiable when trained)."""
        def build():
            wt = self.upsample.weight.float()
            cin, cout = wt.shape[0], wt.shape[1]
            w = wt.permute(2, 3, 1, 0).reshape(4 * cout, 1, 1, cin).contiguous()
            b = self.upsample.bias.float().repeat(4)
            return w, b
        # frozen: re-laid-out (and, downstream, split into bf16 planes) once per parameter version
        cache = self.__dict__.get('_deconv_cache')
        if cache is None:
            cache = self.__dict__['_deconv_cache'] = _FoldCache()
        return cache.get(self.upsample, build)

    def conv_features(self, x):
        """The ``convs`` stack on NHWC RoI features; the result is a ``relu='consumers'`` output
        (every consumer gates its own data gradient: ``mask_input=True``) unless ``num_convs == 0``.
        INTERNAL contract (the output is tagged; ``BF.conv2d_autograd`` refuses it without
        ``mask_input=True``); hand it to foreign code through ``BF.relu_gate`` only."""
        first = True
        for m in self.convs:
            w, b = cached_fold(m.conv)
            x = BF.conv2d_autograd(x, w, b, pad=m.padding, relu='consumers',
                                   mask_input=not first)
            first = False
        return x

    def features(self, x, nhwc=True):
        """RoI features ``[P, h, w, C]`` (or NCHW with ``nhwc=False``) -> ``[P, 2h, 2w, C']``:
        convs + deconv + ReLU, i.e. ``forward`` up to (not including) ``conv_logits``."""
        if not nhwc:
            x = x.permute(0, 2, 3, 1).contiguous()
        return self.upsample_features(self.conv_features(x))

    def upsample_features(self, x):
        """deconv (as a 1x1 conv to 4C + pixel shuffle) + ReLU on the ``conv_features`` output."""
        first = self.num_convs == 0
        w, b = self._deconv_as_conv()
        if not (torch.is_grad_enabled() and self.upsample.weight.requires_grad):
            w, b = w.detach(), b.detach()
        y = BF.conv2d_autograd(x, w, b, relu=True, mask_input=not first)
        P, h, wd, _ = y.shape
        c = self.conv_out_channels
        return y.view(P, h, wd, 2, 2, c).permute(0, 1, 3, 2, 4, 5).reshape(P, 2 * h, 2 * wd, c)

    def forward(self, x, labels=None, nhwc=False):
        """Reference signature (``labels=None``): ``[P, num_classes, 2h, 2w]`` logits (NCHW).  With
        ``labels [P]``: only ``mask_pred[i, labels[i]]`` -> ``[P, 2h, 2w]`` (what the loss and
        ``get_seg_masks`` read)."""
        BF._require_cuda(x)          # (torch restatement: oracle/tensor_forms.fcn_mask_forward)
        f = self.features(x, nhwc=nhwc)
        P, H, W, C = f.shape
        wl = self.conv_logits.weight.view(-1, C)
        if labels is not None:
            z = BF.mask_gt_logits(f.reshape(P, H * W, C), wl, self.conv_logits.bias,
                                  self._channel(labels))
            return z.view(P, H, W)
        y = BF.conv2d_autograd(f, wl.view(-1, 1, 1, C).contiguous(), self.conv_logits.bias)
        return y.permute(0, 3, 1, 2)

    def _channel(self, labels):
        return torch.zeros_like(labels) if self.class_agnostic else labels

    # -- targets ------------------------------------------------------------------------------
    def get_target_fixed(self, rois, gt_inds, valid, gt_masks, rcnn_train_cfg):
        """``mask_target`` (mmdet/core/mask/mask_target.py:7-38) for fixed-shape positives:
        ``rois [P,5]``, ``gt_inds [P]`` int32 (index into the image's ``gt_masks[n]`` ``[G,H,W]``
        uint8 device bitmaps), ``valid [P]`` -> ``[P, S, S]`` float."""
        size = rcnn_train_cfg.mask_size
        size = size if isinstance(size, int) else size[0]
        return BF.mask_target(gt_masks, rois, gt_inds, valid, size)

    # -- loss ---------------------------------------------------------------------------------
    def loss(self, mask_pred, mask_targets, labels):
        """Reference signature on full ``[P, K, S, S]`` logits (fcn_mask_head.py:113-123).  Not
        evaluated here: the product never materialises the K-channel logits (988 MB at P = 256);
        use ``loss_from_features`` (GT-channel ``conv_logits`` + BCE fused, ``bgs_mask_bce``)."""
        raise NotImplementedError('FCNMaskHead.loss on full [P, K, S, S] logits: call '
                                  'loss_from_features(features(x), mask_targets, labels) — the '
                                  'GT-channel logits and the BCE run fused (bgs_mask_bce)')

    def loss_from_features(self, feats, mask_targets, labels, valid=None):
        """Same value from the ``features()`` output ``[P, S, S, C]``: the single-channel
        ``conv_logits`` and the BCE run fused (``bgs_mask_bce``); ``valid`` masks fixed-shape
        padding slots out of the mean."""
        P, H, W, C = feats.shape
        wl = self.conv_logits.weight.view(-1, C)
        val = BF.mask_bce(feats.reshape(P, H * W, C), wl, self.conv_logits.bias,
                          self._channel(labels), mask_targets.reshape(P, H * W), valid)
        return dict(loss_mask=val * self.loss_mask.loss_weight)

    def get_seg_masks_dense(self, mask_pred, det_bboxes, det_labels, rcnn_test_cfg, ori_shape, scale_factor,
                            rescale):
        """The reference's ``get_seg_masks`` (fcn_mask_head.py:125-181) up to the RLE encoding, on the device:
        ``uint8 [n, img_h, img_w]`` with every detection's thresholded mask pasted at its (rescaled, int-truncated)
        box.  ``mask_pred``: ``[n, S, S]`` probabilities of each detection's class (``get_mask_probs``) or the
        reference's ``[n, num_classes, S, S]`` logits (the detection's channel is gathered and squashed here)."""
        import numpy as np
        if mask_pred.dim() == 4:
            ch = self._channel(det_labels + 1)
            mask_pred = torch.sigmoid(mask_pred[torch.arange(mask_pred.size(0), device=mask_pred.device), ch])
        if rescale:
            img_h, img_w = int(ori_shape[0]), int(ori_shape[1])
        else:           # (:158-161: np.round = half to even; the boxes are already in the network's scale)
            img_h = int(np.round(ori_shape[0] * scale_factor).astype(np.int32))
            img_w = int(np.round(ori_shape[1] * scale_factor).astype(np.int32))
            scale_factor = 1.0
        return BF.mask_paste(mask_pred.float(), det_bboxes[:, :4], float(scale_factor),
                             float(rcnn_test_cfg.mask_thr_binary), img_h, img_w)

    def get_seg_masks(self, mask_pred, det_bboxes, det_labels, rcnn_test_cfg, ori_shape, scale_factor, rescale,
                      encode=None):
        """Reference signature and return structure (fcn_mask_head.py:125-181): ``cls_segms[label]`` = the masks of
        that class in detection order.  The resize / threshold / paste runs on the device
        (``get_seg_masks_dense``); ``encode`` turns one dense ``uint8 [img_h, img_w]`` numpy mask into what the
        caller stores — pass ``lambda m: pycocotools.mask.encode(np.asfortranarray(m[:, :, None]))[0]`` for the
        reference's RLEs (pycocotools is evaluation tooling and not a dependency of this package); the default
        keeps the dense masks (views of one device tensor)."""
        dense = self.get_seg_masks_dense(mask_pred, det_bboxes, det_labels, rcnn_test_cfg, ori_shape,
                                         scale_factor, rescale)
        cls_segms = [[] for _ in range(self.num_classes - 1)]
        labels = det_labels.cpu().tolist()
        host = dense.cpu().numpy() if encode is not None else None
        for i, lab in enumerate(labels):
            cls_segms[int(lab)].append(encode(host[i]) if encode is not None else dense[i])
        return cls_segms

    def get_mask_probs(self, feats, det_labels):
        """Test time: sigmoid of the detection's own class channel, ``[n, S, S]`` (the input of
        ``get_seg_masks``' per-detection resize; RLE encoding needs pycocotools: out of scope)."""
        P, H, W, C = feats.shape
        wl = self.conv_logits.weight.view(-1, C)
        z = BF.mask_gt_logits(feats.reshape(P, H * W, C), wl, self.conv_logits.bias,
                              self._channel(det_labels + (0 if self.class_agnostic else 1)))
        return torch.sigmoid(z).view(P, H, W)


@HEADS.register_module
class HTCMaskHead(FCNMaskHead):
    """mmdet/models/mask_heads/htc_mask_head.py:6-38: ``FCNMaskHead`` + ``conv_res`` (1x1 conv +
    ReLU) that injects the previous stage's mask feature (HTC's mask information flow).

    ``forward`` keeps the reference signature.  The detector uses the split form — ``res_features``
    (``conv_res`` + add + ``convs``: what ``return_logits=False`` computes) and
    ``upsample_features`` + the fused single-channel logits / BCE of the base class — so the
    ``[P, 1231, 28, 28]`` logit tensor is never produced."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.conv_res = ConvModule(self.conv_out_channels, self.conv_out_channels, 1)

    def init_weights(self):
        super().init_weights()
        nn.init.kaiming_normal_(self.conv_res.conv.weight, mode='fan_out', nonlinearity='relu')
        nn.init.constant_(self.conv_res.conv.bias, 0)

    def res_features(self, x, res_feat=None):
        """NHWC ``x [P,h,w,C]`` (+ previous stage's ``res_feat``) -> this stage's ``res_feat``
        (htc_mask_head.py:23-28)."""
        if res_feat is not None:
            w, b = cached_fold(self.conv_res.conv)
            x = x + BF.conv2d_autograd(res_feat, w, b, relu=True, mask_input=self.num_convs > 0)
        return self.conv_features(x)

    def forward(self, x, res_feat=None, return_logits=True, return_feat=True, labels=None,
                nhwc=False):
        """Reference signature (NCHW in / out).  ``labels``: only each RoI's own channel."""
        BF._require_cuda(x)          # (torch restatement: oracle/tensor_forms.htc_mask_forward)
        if not nhwc:
            x = x.permute(0, 2, 3, 1).contiguous()
            res_feat = None if res_feat is None else res_feat.permute(0, 2, 3, 1).contiguous()
        res_out = self.res_features(x, res_feat)
        outs = []
        if return_logits:
            f = self.upsample_features(res_out)
            P, H, W, C = f.shape
            wl = self.conv_logits.weight.view(-1, C)
            if labels is not None:
                outs.append(BF.mask_gt_logits(f.reshape(P, H * W, C), wl, self.conv_logits.bias,
                                              self._channel(labels)).view(P, H, W))
            else:
                y = BF.conv2d_autograd(f, wl.view(-1, 1, 1, C).contiguous(), self.conv_logits.bias)
                outs.append(y.permute(0, 3, 1, 2))
        if return_feat:
            # public boundary: the caller may be anything, so the returned feature carries its own
            # ReLU backward (inside this module the consumers gate it: relu='consumers')
            pub = BF.relu_gate(res_out)
            outs.append(pub if nhwc else pub.permute(0, 3, 1, 2))
        return outs if len(outs) > 1 else outs[0]
