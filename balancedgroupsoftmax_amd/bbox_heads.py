"""RoI box heads behind the reference's ``HEADS`` registry keys.

Keys / ctor kwargs / method signatures / state-dict names follow the reference so that
``configs/bags/*.py`` and reference checkpoints drop in:

* ``BBoxHead``            mmdet/models/bbox_heads/bbox_head.py:14-239
* ``ConvFCBBoxHead``      mmdet/models/bbox_heads/convfc_bbox_head.py:8-168
* ``SharedFCBBoxHead``    mmdet/models/bbox_heads/convfc_bbox_head.py:171-185
* ``GSBBoxHeadWith0``     mmdet/models/bbox_heads/gs_bbox_head_with0.py:15-380
* ``GSBBoxHeadWith0Reweight``  mmdet/models/bbox_heads/gs_bbox_head_with0_reweight.py
* ``GSBBoxHead``          alias of ``GSBBoxHeadWith0`` (no source in the reference tree, only
                          ``type=`` strings in two ablation configs — parity unpinned)

What differs is *where the arithmetic runs*: ``GSBBoxHeadWith0.loss`` issues 2 HIP launches
for all B bins (+2 for the box loss) instead of the reference's ~60 tiny kernels and >=17
host syncs, ``_merge_score`` is one launch, and nothing in ``loss`` touches the host.
"""
import pickle

import numpy as np
import torch
import torch.nn as nn

from . import functional as BF
from . import gs_tables
from .builder import build_loss
from .fp16_utils import auto_fp16, force_fp32
from .losses import accuracy
from .registry import HEADS


def _pair(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


@HEADS.register_module
class BBoxHead(nn.Module):
    """Two sibling FCs on pooled RoI features (reference: bbox_head.py:14-78)."""

    def __init__(self, with_avg_pool=False, with_cls=True, with_reg=True, roi_feat_size=7,
                 in_channels=256, num_classes=81, target_means=[0., 0., 0., 0.],
                 target_stds=[0.1, 0.1, 0.2, 0.2], reg_class_agnostic=False,
                 loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                 loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0)):
        super().__init__()
        assert with_cls or with_reg
        self.with_avg_pool, self.with_cls, self.with_reg = with_avg_pool, with_cls, with_reg
        self.roi_feat_size = _pair(roi_feat_size)
        self.roi_feat_area = self.roi_feat_size[0] * self.roi_feat_size[1]
        self.in_channels, self.num_classes = in_channels, num_classes
        self.target_means, self.target_stds = target_means, target_stds
        self.reg_class_agnostic = reg_class_agnostic
        self.fp16_enabled = False
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(loss_bbox)
        flat = in_channels if with_avg_pool else in_channels * self.roi_feat_area
        if with_avg_pool:
            self.avg_pool = nn.AvgPool2d(self.roi_feat_size)
        if with_cls:
            self.fc_cls = nn.Linear(flat, num_classes)
        if with_reg:
            self.fc_reg = nn.Linear(flat, self.num_reg_outputs)
        self.debug_imgs = None

    @property
    def num_reg_classes(self):
        return 1 if self.reg_class_agnostic else self.num_classes

    @property
    def num_reg_outputs(self):
        return 4 * self.num_reg_classes

    def init_weights(self):
        # bbox_head.py:63-69
        if self.with_cls:
            nn.init.normal_(self.fc_cls.weight, 0, 0.01)
            nn.init.constant_(self.fc_cls.bias, 0)
        if self.with_reg:
            nn.init.normal_(self.fc_reg.weight, 0, 0.001)
            nn.init.constant_(self.fc_reg.bias, 0)

    @auto_fp16()
    def forward(self, x):
        if self.with_avg_pool:
            x = self.avg_pool(x)
        x = x.view(x.size(0), -1)
        cls_score = self.fc_cls(x) if self.with_cls else None
        bbox_pred = self.fc_reg(x) if self.with_reg else None
        return cls_score, bbox_pred

    def get_target(self, sampling_results, gt_bboxes, gt_labels, rcnn_train_cfg):
        """bbox_head.py:80-96 -> mmdet/core/bbox/bbox_target.py:7-61."""
        from .box_ops import bbox_target
        return bbox_target([r.pos_bboxes for r in sampling_results],
                           [r.neg_bboxes for r in sampling_results],
                           [r.pos_gt_bboxes for r in sampling_results],
                           [r.pos_gt_labels for r in sampling_results],
                           rcnn_train_cfg, self.num_reg_classes,
                           target_means=self.target_means, target_stds=self.target_stds)

    def _loss_bbox(self, bbox_pred, labels, bbox_targets, bbox_weights, reduction_override,
                   label_weights=None, n_real=None):
        """Box branch shared by every head (bbox_head.py:117-129, gs_bbox_head_with0.py:173-185):
        HIP gather + SmoothL1 + dense-gradient kernel, no boolean-mask indexing, no sync.

        The normaliser is the reference's ``bbox_targets.size(0)`` = the number of sampled RoIs.
        In a fixed-shape batch that is the number of REAL rows: padding slots (``label_weights``
        == 0; the reference's sampler returns fewer RoIs instead) do not count.

        Deviation kept on purpose: an all-background batch returns 0 instead of tripping the
        reference's ``target.numel() > 0`` assertion (smooth_l1_loss.py:11), which would need a
        device->host sync to detect."""
        if reduction_override not in (None, 'mean'):
            raise NotImplementedError('loss_bbox: only the mean/avg_factor reduction used by the '
                                      'detectors is implemented in the HIP path')
        lb = self.loss_bbox
        if type(lb).__name__ != 'SmoothL1Loss':
            raise NotImplementedError('HIP box loss implements SmoothL1Loss only')
        val = BF.bbox_smooth_l1_loss(bbox_pred, labels, bbox_targets, bbox_weights,
                                     self.num_reg_classes, beta=lb.beta,
                                     avg_factor=bbox_targets.size(0), loss_weight=lb.loss_weight)
        if label_weights is not None and label_weights.is_cuda:
            if n_real is None:      # (the GS heads pass bin 0's avg factor = max(#real rows, 1))
                n_real = (label_weights > 0).sum().to(torch.float32).clamp(min=1.0)
            val = val * (float(bbox_targets.size(0)) / n_real)
        return val

    @force_fp32(apply_to=('cls_score', 'bbox_pred'))
    def loss(self, cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights,
             reduction_override=None):
        losses = dict()
        if cls_score is not None:
            avg_factor = torch.clamp((label_weights > 0).sum().float(), min=1.)
            losses['loss_cls'] = self.loss_cls(cls_score, labels, label_weights,
                                               avg_factor=avg_factor,
                                               reduction_override=reduction_override)
            losses['acc'] = accuracy(cls_score, labels)
        if bbox_pred is not None:
            losses['loss_bbox'] = self._loss_bbox(bbox_pred, labels, bbox_targets, bbox_weights,
                                                  reduction_override, label_weights)
        return losses

    def _scores(self, cls_score):
        return torch.softmax(cls_score, dim=1) if cls_score is not None else None

    @force_fp32(apply_to=('cls_score', 'bbox_pred'))
    def get_det_bboxes(self, rois, cls_score, bbox_pred, img_shape, scale_factor, rescale=False,
                       cfg=None):
        """bbox_head.py:132-167 / gs_bbox_head_with0.py:343-380 (scores differ per head)."""
        from .box_ops import delta2bbox
        if isinstance(cls_score, list):
            cls_score = sum(cls_score) / float(len(cls_score))
        scores = self._scores(cls_score)
        if bbox_pred is not None:
            bboxes = delta2bbox(rois[:, 1:], bbox_pred, self.target_means, self.target_stds,
                                img_shape)
        else:
            bboxes = rois[:, 1:].clone()
            if img_shape is not None:
                bboxes[:, [0, 2]] = bboxes[:, [0, 2]].clamp(min=0, max=img_shape[1] - 1)
                bboxes[:, [1, 3]] = bboxes[:, [1, 3]].clamp(min=0, max=img_shape[0] - 1)
        if rescale:
            if isinstance(scale_factor, float):
                bboxes = bboxes / scale_factor
            else:
                bboxes = bboxes / torch.as_tensor(scale_factor).to(bboxes.device)
        if cfg is None:
            return bboxes, scores
        from .post_processing import multiclass_nms
        return multiclass_nms(bboxes, scores, cfg.score_thr, cfg.nms, cfg.max_per_img)

    @force_fp32(apply_to=('bbox_preds', ))
    def refine_bboxes(self, rois, labels, bbox_preds, pos_is_gts, img_metas):
        """Cascade stage hand-over (bbox_head.py:169-208): regress every RoI with its own
        class' deltas, drop the RoIs that were GT boxes."""
        img_ids = rois[:, 0].long().unique(sorted=True)
        assert img_ids.numel() == len(img_metas)
        out = []
        for i, meta in enumerate(img_metas):
            inds = torch.nonzero(rois[:, 0] == i).squeeze(1)
            boxes = self.regress_by_class(rois[inds, 1:], labels[inds], bbox_preds[inds], meta)
            keep = torch.ones(inds.numel(), dtype=torch.bool, device=rois.device)
            gt_flags = pos_is_gts[i].to(torch.bool)
            keep[:gt_flags.numel()] = ~gt_flags
            out.append(boxes[keep])
        return out

    @force_fp32(apply_to=('bbox_pred', ))
    def regress_by_class(self, rois, label, bbox_pred, img_meta):
        """bbox_head.py:210-239."""
        from .box_ops import delta2bbox
        assert rois.size(1) in (4, 5)
        if not self.reg_class_agnostic:
            cols = (label * 4).view(-1, 1) + torch.arange(4, device=label.device).view(1, 4)
            bbox_pred = torch.gather(bbox_pred, 1, cols)
        assert bbox_pred.size(1) == 4
        if rois.size(1) == 4:
            return delta2bbox(rois, bbox_pred, self.target_means, self.target_stds,
                              img_meta['img_shape'])
        boxes = delta2bbox(rois[:, 1:], bbox_pred, self.target_means, self.target_stds,
                           img_meta['img_shape'])
        return torch.cat((rois[:, [0]], boxes), dim=1)


@HEADS.register_module
class ConvFCBBoxHead(BBoxHead):
    """shared convs -> shared fcs -> {cls convs -> cls fcs -> fc_cls, reg convs -> reg fcs -> fc_reg}
    (convfc_bbox_head.py:8-168).  Parameter names are the reference's: ``shared_fcs.i``,
    ``cls_fcs.i``, ``reg_fcs.i``, ``fc_cls``, ``fc_reg``.  The BAGS configs use FC branches
    only; conv branches are not built here."""

    def __init__(self, num_shared_convs=0, num_shared_fcs=0, num_cls_convs=0, num_cls_fcs=0,
                 num_reg_convs=0, num_reg_fcs=0, conv_out_channels=256, fc_out_channels=1024,
                 conv_cfg=None, norm_cfg=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        counts = (num_shared_convs, num_shared_fcs, num_cls_convs, num_cls_fcs, num_reg_convs,
                  num_reg_fcs)
        assert sum(counts) > 0
        if num_shared_convs or num_cls_convs or num_reg_convs:
            raise NotImplementedError('conv branches of ConvFCBBoxHead are outside the BAGS hot '
                                      'path (no shipped BAGS config uses them)')
        if not self.with_cls:
            assert num_cls_fcs == 0
        if not self.with_reg:
            assert num_reg_fcs == 0
        (self.num_shared_convs, self.num_shared_fcs, self.num_cls_convs, self.num_cls_fcs,
         self.num_reg_convs, self.num_reg_fcs) = counts
        self.conv_out_channels, self.fc_out_channels = conv_out_channels, fc_out_channels
        self.conv_cfg, self.norm_cfg = conv_cfg, norm_cfg

        flat_in = self.in_channels * (1 if self.with_avg_pool else self.roi_feat_area)
        self.shared_convs = nn.ModuleList()
        self.cls_convs = nn.ModuleList()
        self.reg_convs = nn.ModuleList()
        self.shared_fcs, shared_dim = self._fc_stack(num_shared_fcs, flat_in)
        self.shared_out_channels = shared_dim if num_shared_fcs else self.in_channels
        branch_in = shared_dim if num_shared_fcs else flat_in
        self.cls_fcs, self.cls_last_dim = self._fc_stack(num_cls_fcs, branch_in)
        self.reg_fcs, self.reg_last_dim = self._fc_stack(num_reg_fcs, branch_in)
        self.relu = nn.ReLU(inplace=True)
        if self.with_cls:
            self.fc_cls = nn.Linear(self.cls_last_dim, self.num_classes)
        if self.with_reg:
            self.fc_reg = nn.Linear(self.reg_last_dim, self.num_reg_outputs)

    def _fc_stack(self, n, in_dim):
        fcs = nn.ModuleList()
        d = in_dim
        for _ in range(n):
            fcs.append(nn.Linear(d, self.fc_out_channels))
            d = self.fc_out_channels
        return fcs, d

    def init_weights(self):
        super().init_weights()
        for stack in (self.shared_fcs, self.cls_fcs, self.reg_fcs):
            for m in stack:
                nn.init.xavier_uniform_(m.weight)
                nn.init.constant_(m.bias, 0)

    def _fc1_weight(self, fc, nhwc):
        """First FC weight as seen by a bin-major ``[K, h, w, C]`` RoI feature (our RoIAlign
        layout): column ``(h*w_ + w)*C + c`` <- reference column ``c*h_*w_ + h*w_ + w``.
        Cached per parameter version."""
        if not nhwc:
            return fc.weight
        if fc.weight.requires_grad and torch.is_grad_enabled():
            # trained (selectp 0 / 2): the permutation stays on the autograd tape
            O = fc.weight.shape[0]
            return fc.weight.view(O, self.in_channels, self.roi_feat_area).permute(0, 2, 1) \
                .reshape(O, -1)
        key = (id(fc.weight), fc.weight._version, fc.weight.device)
        if getattr(self, '_fc1_key', None) != key:
            O = fc.weight.shape[0]
            C = self.in_channels
            w = fc.weight.detach().view(O, C, self.roi_feat_area).permute(0, 2, 1)
            self._fc1_perm = w.reshape(O, -1).contiguous()
            self._fc1_key = key
        return self._fc1_perm

    def forward(self, x, nhwc=False):
        """``x``: RoI features ``[K, C, h, w]`` (reference layout) or, with ``nhwc=True``,
        ``[K, h, w, C]`` as produced by our RoIAlign.  Every FC runs in the MFMA GEMM kernel (bias +
        ReLU fused).  (torch restatement for the CPU checks: oracle/tensor_forms.convfc_bbox_forward.)"""
        BF._require_cuda(x)
        if self.with_avg_pool:
            if nhwc:
                raise NotImplementedError('with_avg_pool on NHWC RoI features')
            x = self.avg_pool(x)
        x = x.reshape(x.size(0), -1)

        def fc_apply(fc, t, relu, first=False, t_is_relu=False):
            w = self._fc1_weight(fc, nhwc) if first else fc.weight
            # every consumer of a hidden FC output is another FC of this head: the ReLU backward
            # rides in the consumers' dgrad epilogue
            return BF.linear_autograd(t, w, fc.bias, relu='consumers' if relu else False,
                                      mask_input=t_is_relu)

        first = True
        hidden = False          # is the running activation the ReLU output of a hidden FC?
        for fc in self.shared_fcs:
            x = fc_apply(fc, x, True, first, hidden)
            first, hidden = False, True
        if nhwc and self.num_shared_fcs == 0:
            raise NotImplementedError('NHWC RoI features need a shared first FC')
        x_cls = x_reg = x
        h_cls = h_reg = hidden
        for fc in self.cls_fcs:
            x_cls = fc_apply(fc, x_cls, True, False, h_cls)
            h_cls = True
        for fc in self.reg_fcs:
            x_reg = fc_apply(fc, x_reg, True, False, h_reg)
            h_reg = True
        # fc_cls (80 workgroups x 4 K slices) and fc_reg (312) both read the last hidden activation: a frozen fc_reg
        # with no reg branch of its own runs beside fc_cls on the side stream (functional.forked)
        fk = None
        if self.with_cls and self.with_reg and not self.reg_fcs and BF.shortcut_fork_enabled() and \
                not (torch.is_grad_enabled() and (x_reg.requires_grad or self.fc_reg.weight.requires_grad)):
            with BF.forked(x_reg.device) as fk:
                bbox_pred = fc_apply(self.fc_reg, x_reg, False, False, h_reg)
        cls_score = fc_apply(self.fc_cls, x_cls, False, False, h_cls) if self.with_cls else None
        if fk is not None:
            fk.join()
        else:
            bbox_pred = fc_apply(self.fc_reg, x_reg, False, False, h_reg) if self.with_reg else None
        return cls_score, bbox_pred


@HEADS.register_module
class SharedFCBBoxHead(ConvFCBBoxHead):

    def __init__(self, num_fcs=2, fc_out_channels=1024, *args, **kwargs):
        assert num_fcs >= 1
        super().__init__(num_shared_convs=0, num_shared_fcs=num_fcs, num_cls_convs=0,
                         num_cls_fcs=0, num_reg_convs=0, num_reg_fcs=0,
                         fc_out_channels=fc_out_channels, *args, **kwargs)


@HEADS.register_module
class GSBBoxHeadWith0(SharedFCBBoxHead):
    """Balanced Group Softmax box head (gs_bbox_head_with0.py:15-380).

    ``fc_cls`` is widened to ``num_classes + num_bins`` outputs (:27-29).  ``gs_config`` keys
    (attribute access, same as the reference): ``label2binlabel, pred_slice, fg_split,
    others_sample_ratio, loss_bg (unused), num_bins, loss_bin[, bin_cls_weight]``; one extra
    optional key ``sampler`` = ``'device'`` (default: counter-based RNG inside the HIP
    prepare kernel, no host sync) or ``'numpy'`` (the reference's host-side
    ``np.random.choice`` draw, bit-identical weights for parity runs, costs a D2H sync).
    """

    fused_loss_scale = True       # loss(..., loss_scale=w) folds a stage weight into the kernel's weights

    def __init__(self, num_fcs=2, fc_out_channels=1024, gs_config=None, *args, **kwargs):
        super().__init__(num_fcs=num_fcs, fc_out_channels=fc_out_channels, *args, **kwargs)
        gs = gs_config
        self.num_bins = gs.num_bins
        self.fc_cls = nn.Linear(self.cls_last_dim, self.num_classes + gs.num_bins)
        self.loss_bins = [build_loss(gs.loss_bin) for _ in range(gs.num_bins)]
        for lb in self.loss_bins:
            if getattr(lb, 'use_sigmoid', False) or getattr(lb, 'use_mask', False):
                raise NotImplementedError('group softmax bins use softmax cross entropy')
        l2b, ps, fg_splits = gs_tables.load_group_tables(gs.label2binlabel, gs.pred_slice,
                                                         gs.fg_split)
        if l2b.shape[0] != gs.num_bins or ps.shape[0] != gs.num_bins:
            raise ValueError('gs_config.num_bins=%d but the tables hold %d bins'
                             % (gs.num_bins, l2b.shape[0]))
        if int(ps[:, 1].sum()) != self.num_classes + gs.num_bins:
            raise ValueError('pred_slice covers %d logits, fc_cls has %d'
                             % (int(ps[:, 1].sum()), self.num_classes + gs.num_bins))
        # plain attributes in the reference (not in its state_dict) -> non-persistent buffers
        self.register_buffer('label2binlabel', l2b, persistent=False)
        self.register_buffer('pred_slice', ps, persistent=False)
        self.pred_slice_host = ps.numpy().copy()    # static metadata: passed by value to kernels
        self.label2binlabel_host = l2b.numpy().copy()
        self.fg_splits = fg_splits
        self.register_buffer('cls2col', self._class_columns(ps, fg_splits), persistent=False)
        self.register_buffer('bin_loss_weight', torch.tensor(
            [float(lb.loss_weight) for lb in self.loss_bins], dtype=torch.float32),
            persistent=False)
        self._bin_loss_weight_host = [float(lb.loss_weight) for lb in self.loss_bins]
        self.others_sample_ratio = gs.others_sample_ratio
        self.sampler = gs.get('sampler', 'device') if hasattr(gs, 'get') else 'device'
        self.cls_weights = None        # Reweight variant: list of per-bin arrays
        # device-side draw counter: a fresh 'others' sample every call, also under hipGraph replay
        self.register_buffer('_draw', torch.zeros(1, dtype=torch.int64), persistent=False)
        self._seed = None
        self.register_buffer('cls_weight_table', None, persistent=False)

    def _class_columns(self, ps, fg_splits):
        """Column of the widened logits that scores class c (inverse of the ``fg_splits``
        scatter, gs_bbox_head_with0.py:258-259); -1 for classes in no split."""
        col = torch.full((self.num_classes,), -1, dtype=torch.int32)
        col[0] = int(ps[0, 0])
        for i, split in enumerate(fg_splits):
            k = torch.arange(1, split.numel() + 1, dtype=torch.int32)
            col[split.long()] = int(ps[i + 1, 0]) + k
        return col

    # -- label remap + sampling -----------------------------------------------------------
    def _sample_others_numpy(self, bin_label, cls_weight=None):
        """The reference's host-side draw, kept verbatim in behaviour for parity runs
        (gs_bbox_head_with0.py:63-89): all in-bin fg rows + ``int(n_fg * ratio)`` of the rest
        via ``np.random.choice(replace=False)`` on numpy's global RNG."""
        fg = bin_label > 0
        n_fg = int(fg.sum())
        if n_fg == 0:
            return np.zeros(bin_label.shape[0], dtype=np.float32)
        others = np.flatnonzero(~fg)
        k = int(n_fg * self.others_sample_ratio)
        w = np.ones(bin_label.shape[0], dtype=np.float64)
        if k < others.shape[0]:
            w = fg.astype(np.float64)
            w[np.random.choice(others, (k,), replace=False)] = 1.0
        if cls_weight is not None:
            w = w * np.asarray(cls_weight, dtype=np.float64)[bin_label]
        return w

    def _remap_labels(self, labels, label_weights=None):
        """gs_bbox_head_with0.py:91-112 returned Python lists + floats via .item(); here the
        three results are device tensors: bin labels ``[B, N]`` i32, sample weights ``[B, N]``
        f32 and avg factors ``[B]`` f32.

        ``label_weights``: rows with weight 0 are padding slots of a fixed-shape batch (the
        reference's sampler returns fewer RoIs instead of padding, so its head never sees such
        rows and ignores the argument): they are left out of every count, draw and loss."""
        if self.sampler == 'device':
            if self._seed is None:
                self._seed = (torch.initial_seed() * 0x9E3779B1) & 0xFFFFFFFFFFFFFFFF
            self._draw += 1
            return BF.gs_prepare(labels, self.label2binlabel, self.others_sample_ratio,
                                 seed=self._seed, seed_offset=self._draw,
                                 cls_weight=self.cls_weight_table, row_weights=label_weights)
        if self.sampler != 'numpy':
            raise ValueError('gs_config.sampler must be "device" or "numpy"')
        l2b = self.label2binlabel_host
        lab = labels.detach().cpu().numpy()
        B = l2b.shape[0]
        bl = l2b[:, lab]
        w = np.ones((B, lab.shape[0]), dtype=np.float64)
        real = None
        if label_weights is not None:
            real = label_weights.detach().cpu().numpy() > 0
        for i in range(1, B):
            cw = None if self.cls_weights is None else self.cls_weights[i - 1]
            if real is None or real.all():
                w[i] = self._sample_others_numpy(bl[i], cw)
            else:
                w[i, real] = self._sample_others_numpy(bl[i][real], cw)
        if real is not None:
            w[:, ~real] = 0.0
        avg = np.maximum(w.sum(axis=1).astype(np.float32), np.float32(1.0))
        dev = labels.device
        return (torch.from_numpy(np.ascontiguousarray(bl, dtype=np.int32)).to(dev),
                torch.from_numpy(w.astype(np.float32)).to(dev),
                torch.from_numpy(avg.astype(np.float32)).to(dev))

    def _slice_preds(self, cls_score):
        """Column views per bin (gs_bbox_head_with0.py:134-145); not used by ``loss`` (the
        kernel reads ``pred_slice`` itself) but kept for API parity / debugging."""
        ps = self.pred_slice_host.tolist()
        return [cls_score.narrow(1, s, n) for s, n in ps]

    @force_fp32(apply_to=('cls_score', 'bbox_pred'))
    def loss(self, cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights,
             reduction_override=None, loss_scale=1.0):
        """Keys: ``loss_cls_bin0..B-1`` and ``loss_bbox`` (no ``acc``), gs_bbox_head_with0.py:147-186.
        ``loss_scale``: a factor on every term (the cascade's ``stage_loss_weights``,
        cascade_rcnn.py:283-286) folded into the fused kernel's per-bin / box weights instead of one
        multiply launch per term."""
        if reduction_override not in (None, 'mean'):
            if reduction_override == 'sum':
                raise ValueError('avg_factor can not be used with reduction="sum"')
            raise NotImplementedError('reduction_override="none" is not produced by the fused '
                                      'kernel (no detector on the BAGS path requests it)')
        losses = dict()
        n_real = None
        fused = cls_score is not None and self.sampler == 'device' and \
            self.cls_weight_table is None and \
            0 < cls_score.shape[0] <= BF.GS_FUSED_MAX_ROWS and self.num_bins <= 15 and \
            8 * cls_score.shape[1] + 2 * cls_score.shape[0] + 2 * self.num_classes < 60000   # rows + flags + class bits in the LDS window
        if fused and (bbox_pred is None or type(self.loss_bbox).__name__ == 'SmoothL1Loss'):
            # TWO launches for the whole loss(): label remap + "others" sampling + per-bin losses
            # (x their loss weights) + gradient + the box branch in the streaming kernel, then the
            # fixed-order reduce, which also advances the draw counter
            if self._seed is None:
                self._seed = (torch.initial_seed() * 0x9E3779B1) & 0xFFFFFFFFFFFFFFFF
            terms, _total, _avg = BF.gs_head_step(
                cls_score, labels, self.label2binlabel, self.pred_slice_host,
                self.others_sample_ratio, self._seed, draw_counter=self._draw,
                row_weights=label_weights,
                bin_loss_weight=[w * loss_scale for w in self._bin_loss_weight_host]
                if loss_scale != 1.0 else self._bin_loss_weight_host,
                bbox_pred=bbox_pred, bbox_targets=bbox_targets, bbox_weights=bbox_weights,
                num_reg_classes=self.num_reg_classes,
                beta=self.loss_bbox.beta if bbox_pred is not None else 1.0,
                box_loss_weight=(self.loss_bbox.loss_weight if bbox_pred is not None else 1.0) * loss_scale)
            parts = BF.unbind_terms(terms)   # ONE autograd node for all terms; unit gradients pass by identity
            for i in range(self.num_bins):
                losses['loss_cls_bin{}'.format(i)] = parts[i]
            if bbox_pred is not None:
                losses['loss_bbox'] = parts[self.num_bins]
            return losses
        if cls_score is not None:
            bin_labels, weights, avg = self._remap_labels(labels, label_weights)
            per_bin = BF.group_softmax_loss(cls_score, bin_labels, self.pred_slice_host,
                                            weights, avg)
            # bin 0 weighs every real row 1 (gs_bbox_head_with0.py:100-102; the reweight variant's
            # tables start at bin 1): its avg factor is max(#real rows, 1)
            n_real = avg[0]
            per_bin = per_bin * self.bin_loss_weight
            for i in range(self.num_bins):
                losses['loss_cls_bin{}'.format(i)] = per_bin[i]
        if bbox_pred is not None:
            losses['loss_bbox'] = self._loss_bbox(bbox_pred, labels, bbox_targets, bbox_weights,
                                                  reduction_override, label_weights, n_real)
        if loss_scale != 1.0:
            losses = {k: v * loss_scale for k, v in losses.items()}
        return losses

    @force_fp32(apply_to=('cls_score', ))
    def _merge_score(self, cls_score):
        """gs_bbox_head_with0.py:239-273, one HIP launch."""
        return BF.gs_merge_score(cls_score, self.pred_slice_host, self.cls2col, self.num_classes)

    def _scores(self, cls_score):
        return self._merge_score(cls_score)


@HEADS.register_module
class GSBBoxHeadWith0Reweight(GSBBoxHeadWith0):
    """Same head with per-bin class weights multiplied into the sample weights
    (gs_bbox_head_with0_reweight.py:51-54,57-87; table from tools/lvis_analyse.py:449-484)."""

    def __init__(self, *args, gs_config=None, **kwargs):
        super().__init__(*args, gs_config=gs_config, **kwargs)
        with open(gs_config.bin_cls_weight, 'rb') as f:
            weights = pickle.load(f)
        self.cls_weights = [np.asarray(w, dtype=np.float64) for w in weights]
        stride = max(w.shape[0] for w in self.cls_weights)
        table = torch.ones((len(self.cls_weights), stride), dtype=torch.float32)
        for i, w in enumerate(self.cls_weights):
            table[i, :w.shape[0]] = torch.from_numpy(w).float()
        self.register_buffer('cls_weight_table', table, persistent=False)


# ``type='GSBBoxHead'`` appears in configs/ablations/gs_faster_rcnn_r50_fpn_1x_lvis.py:35 but the
# class itself is not in the reference tree; the closest surviving semantics is the With0 head.
HEADS.register_alias('GSBBoxHead', GSBBoxHeadWith0)
