"""Region proposal network behind the reference's ``RPNHead`` registry key.

* ``AnchorGenerator``  mmdet/core/anchor/anchor_generator.py:4-83 (known-answer doctest :7-14)
* ``RPNHead``          mmdet/models/anchor_heads/rpn_head.py:12-104 on top of
  ``AnchorHead`` mmdet/models/anchor_heads/anchor_head.py:15-276 and ``anchor_target``
  mmdet/core/anchor/anchor_target.py:7-159

Differences in *how*, not *what*:
* the shared 3x3 conv (+ReLU) and the two 1x1 heads run in the fp32-MFMA conv kernel; the
  1x1 ``rpn_cls`` / ``rpn_reg`` are evaluated as ONE conv with concatenated output channels;
* outputs stay NHWC ``[N,H,W,A]`` / ``[N,H,W,4A]`` — exactly the ``permute(0,2,3,1)`` order the
  reference flattens to (anchor_head.py:150-158, rpn_head.py:69-77);
* targets: all anchors of a batch are assigned, sampled and turned into losses by fused kernels
  (csrc/det_targets.hip, csrc/sampler.hip): no CPU fallback for >50 GTs, no numpy shuffles, no
  ``nonzero``; ONE path — the tensor-op restatement that pins the kernels to the reference
  classes is test infrastructure (oracle/tensor_forms.py);
* proposals: the 5 levels x N images go through ONE batched on-device NMS launch pair
  (csrc/nms.hip) instead of 10 ``nms_cuda`` calls with a D2H copy each (rpn_head.py:92).
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import functional as BF
from .backbone import _fold_conv_bn, _FoldCache
from .builder import build_loss
from .registry import HEADS


class AnchorGenerator(object):
    """Known answer (reference doctest): ``AnchorGenerator(9, [1.], [1.]).grid_anchors((2, 2),
    stride=16)`` -> [[0,0,8,8],[16,0,24,8],[0,16,8,24],[16,16,24,24]]."""

    def __init__(self, base_size, scales, ratios, scale_major=True, ctr=None):
        self.base_size = base_size
        self.scales = torch.tensor(scales, dtype=torch.float32)
        self.ratios = torch.tensor(ratios, dtype=torch.float32)
        self.scale_major = scale_major
        self.ctr = ctr
        self.base_anchors = self.gen_base_anchors()

    @property
    def num_base_anchors(self):
        return self.base_anchors.size(0)

    def gen_base_anchors(self):
        w = h = float(self.base_size)
        x_ctr, y_ctr = (0.5 * (w - 1), 0.5 * (h - 1)) if self.ctr is None else self.ctr
        h_ratios = torch.sqrt(self.ratios)
        w_ratios = 1 / h_ratios
        if self.scale_major:
            ws = (w * w_ratios[:, None] * self.scales[None, :]).view(-1)
            hs = (h * h_ratios[:, None] * self.scales[None, :]).view(-1)
        else:
            ws = (w * self.scales[:, None] * w_ratios[None, :]).view(-1)
            hs = (h * self.scales[:, None] * h_ratios[None, :]).view(-1)
        return torch.stack([x_ctr - 0.5 * (ws - 1), y_ctr - 0.5 * (hs - 1),
                            x_ctr + 0.5 * (ws - 1), y_ctr + 0.5 * (hs - 1)], dim=-1).round()

    def grid_anchors(self, featmap_size, stride=16, device='cuda'):
        """``[H*W*A, 4]`` ordered (y, x, anchor)."""
        base = self.base_anchors.to(device)
        fh, fw = featmap_size
        sx = torch.arange(0, fw, device=device, dtype=torch.float32) * stride
        sy = torch.arange(0, fh, device=device, dtype=torch.float32) * stride
        yy, xx = torch.meshgrid(sy, sx, indexing='ij')
        shifts = torch.stack([xx, yy, xx, yy], dim=-1).view(-1, 1, 4)
        return (base[None] + shifts).view(-1, 4)

    def valid_flags(self, featmap_size, valid_size, device='cuda'):
        fh, fw = featmap_size
        vh, vw = valid_size
        assert vh <= fh and vw <= fw
        vx = torch.arange(fw, device=device) < vw
        vy = torch.arange(fh, device=device) < vh
        valid = (vy[:, None] & vx[None, :]).view(-1)
        return valid[:, None].expand(valid.size(0), self.num_base_anchors).reshape(-1)


class ProposalList(list):
    """``[(props_i [P, C], valid_i [P]) ...]`` — the per-image list the reference passes around — that also
    carries the batch tensors ``(props [N, P, C], valid [N, P] bool)`` its entries are views of, so the
    fixed-shape consumers (`_sample_rois_fused`) read them without re-stacking (two cat launches and a cast
    per stage)."""

    def __init__(self, props, valid):
        super().__init__([(props[i], valid[i]) for i in range(props.shape[0])])
        self.batched = (props, valid)


@HEADS.register_module
class RPNHead(nn.Module):

    def __init__(self, in_channels, feat_channels=256, anchor_scales=[8, 16, 32],
                 anchor_ratios=[0.5, 1.0, 2.0], anchor_strides=[4, 8, 16, 32, 64],
                 anchor_base_sizes=None, target_means=(.0, .0, .0, .0),
                 target_stds=(1.0, 1.0, 1.0, 1.0),
                 loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                 loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0)):
        super().__init__()
        self.in_channels, self.feat_channels = in_channels, feat_channels
        self.num_classes = 2
        self.anchor_scales, self.anchor_ratios = anchor_scales, anchor_ratios
        self.anchor_strides = anchor_strides
        self.anchor_base_sizes = list(anchor_strides) if anchor_base_sizes is None \
            else anchor_base_sizes
        self.target_means, self.target_stds = target_means, target_stds
        self.use_sigmoid_cls = loss_cls.get('use_sigmoid', False)
        if not self.use_sigmoid_cls:
            raise NotImplementedError('RPN objectness: sigmoid mode only (all BAGS configs)')
        self.sampling = True
        self.cls_out_channels = 1
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(loss_bbox)
        self.fp16_enabled = False
        self.anchor_generators = [AnchorGenerator(b, anchor_scales, anchor_ratios)
                                  for b in self.anchor_base_sizes]
        self.num_anchors = len(anchor_ratios) * len(anchor_scales)
        self.rpn_conv = nn.Conv2d(in_channels, feat_channels, 3, padding=1)
        self.rpn_cls = nn.Conv2d(feat_channels, self.num_anchors * self.cls_out_channels, 1)
        self.rpn_reg = nn.Conv2d(feat_channels, self.num_anchors * 4, 1)
        self._cache = _FoldCache()
        self._anchor_cache = {}

    def init_weights(self):
        for m in (self.rpn_conv, self.rpn_cls, self.rpn_reg):
            nn.init.normal_(m.weight, 0, 0.01)
            nn.init.constant_(m.bias, 0)

    # -- forward ------------------------------------------------------------------------
    def _build_fold(self):
        wc, bc = _fold_conv_bn(self.rpn_cls, None)
        wr, br = _fold_conv_bn(self.rpn_reg, None)
        return dict(conv=_fold_conv_bn(self.rpn_conv, None),
                    head=(torch.cat([wc, wr], 0).contiguous(), torch.cat([bc, br], 0).contiguous()))

    def forward(self, feats):
        """NHWC feature maps -> (cls_scores, bbox_preds): per level ``[N,H,W,A]``, ``[N,H,W,4A]``."""
        f = self._cache.get(self, self._build_fold)
        cls_scores, bbox_preds = [], []
        na = self.num_anchors * self.cls_out_channels

        def level(x):
            h = BF.conv2d_autograd(x, f['conv'][0], f['conv'][1], pad=1, relu='consumers')
            return BF.conv2d_autograd(h, f['head'][0], f['head'][1], mask_input=True)

        if len(feats) > 1 and feats[0].is_cuda and BF.level_fork_enabled():
            # the small levels next to the P2-level launches (functional.forked)
            BF.presplit(f['conv'][0], f['head'][0])      # shared by both streams: cached before the fork
            with BF.forked(feats[0].device) as fk:
                small = [level(x) for x in feats[1:]]
            fused = [level(feats[0])] + small
            fk.join()
        else:
            fused = [level(x) for x in feats]
        for o in fused:
            cls_scores.append(o[..., :na])
            bbox_preds.append(o[..., na:])
        self._fused = fused      # [N,H,W,A+4A] per level: what the fused loss / decode kernels read
        return cls_scores, bbox_preds

    # -- anchors ------------------------------------------------------------------------
    def _level_anchors(self, featmap_sizes, device):
        key = (tuple(featmap_sizes), str(device))
        if key not in self._anchor_cache:
            self._anchor_cache[key] = [
                g.grid_anchors(s, st, device=device)
                for g, s, st in zip(self.anchor_generators, featmap_sizes, self.anchor_strides)]
        return self._anchor_cache[key]

    def get_anchors(self, featmap_sizes, img_metas, device='cuda'):
        """anchor_head.py:101-140: (anchors per image, valid flags per image), per level."""
        mlvl = self._level_anchors([tuple(s) for s in featmap_sizes], device)
        anchor_list = [mlvl for _ in img_metas]
        valid_flag_list = []
        for meta in img_metas:
            flags = []
            h, w = meta['pad_shape'][:2]
            for i, (fh, fw) in enumerate(featmap_sizes):
                st = self.anchor_strides[i]
                vh = min(int(np.ceil(h / st)), fh)
                vw = min(int(np.ceil(w / st)), fw)
                flags.append(self.anchor_generators[i].valid_flags((fh, fw), (vh, vw), device))
            valid_flag_list.append(flags)
        return anchor_list, valid_flag_list

    # -- loss ---------------------------------------------------------------------------
    def loss(self, cls_scores, bbox_preds, gt_bboxes, img_metas, cfg, gt_bboxes_ignore=None,
             samplers=None):
        """Keys ``loss_rpn_cls`` / ``loss_rpn_bbox``: lists with one scalar per level
        (rpn_head.py:37-53, anchor_head.py:163-207).  ONE path: the fused kernels
        (``anchor_target`` + BCE + SmoothL1 in csrc/det_targets.hip / csrc/sampler.hip), which read
        this head's own output buffers.  (The tensor-op restatement that pins them to the
        reference classes lives in oracle/tensor_forms.py: test infrastructure.)
        ``samplers``: test hook, ``dict(rpn=fn)`` replaces the device RandomSampler by a
        caller-supplied draw so that two paths can be compared sample for sample."""
        if not self._use_fused(cls_scores):
            raise RuntimeError("RPNHead.loss reads the head's own GPU outputs: call forward() on "
                               "CUDA tensors first and pass its results unchanged (there is no "
                               "CPU / tensor-op path in the product)")
        featmap_sizes = [tuple(c.shape[1:3]) for c in cls_scores]
        return self._loss_fused(featmap_sizes, gt_bboxes, img_metas, cfg, samplers)

    # -- fused HIP path (csrc/det_targets.hip) -------------------------------------------------
    def _use_fused(self, cls_scores):
        """The fused kernels read the head's own output buffers (``bgs_rpn_loss`` is
        differentiable w.r.t. them: ``bgs_rpn_loss_grad``); they apply whenever the scores passed
        in are the ones this head just produced on the GPU."""
        fused = getattr(self, '_fused', None)
        if fused is None or not cls_scores[0].is_cuda or len(fused) != len(cls_scores):
            return False
        return not any(c.data_ptr() != o.data_ptr() for c, o in zip(cls_scores, fused))

    def _all_anchors(self, featmap_sizes, dev):
        key = ('all', tuple(featmap_sizes), str(dev))
        if key not in self._anchor_cache:
            self._anchor_cache[key] = torch.cat(self._level_anchors(featmap_sizes, dev)).contiguous()
        return self._anchor_cache[key]

    def _inside_flags(self, featmap_sizes, img_metas, allowed_border, dev):
        """anchor_inside_flags (anchor_target.py:162-174) for every image, ``[N,A]`` uint8; depends
        only on the image geometry, so it is cached."""
        key = ('inside', tuple(featmap_sizes), allowed_border, str(dev),
               tuple((tuple(m['pad_shape'][:2]), tuple(m['img_shape'][:2])) for m in img_metas))
        if key not in self._anchor_cache:
            anchors = self._all_anchors(featmap_sizes, dev)
            _, flags = self.get_anchors(featmap_sizes, img_metas, dev)
            rows = []
            for meta, fl in zip(img_metas, flags):
                inside = torch.cat(fl)
                if allowed_border >= 0:
                    h, w = meta['img_shape'][:2]
                    inside = inside & (anchors[:, 0] >= -allowed_border) & \
                        (anchors[:, 1] >= -allowed_border) & (anchors[:, 2] < w + allowed_border) & \
                        (anchors[:, 3] < h + allowed_border)
                rows.append(inside)
            self._anchor_cache[key] = torch.stack(rows).to(torch.uint8).contiguous()
        return self._anchor_cache[key]

    def _loss_fused(self, featmap_sizes, gt_bboxes, img_metas, cfg, samplers=None):
        dev = self._fused[0].device
        anchors = self._all_anchors(featmap_sizes, dev)
        inside = self._inside_flags(featmap_sizes, img_metas, cfg.allowed_border, dev)
        gt_cat = torch.cat([g[:, :4] for g in gt_bboxes]).float().contiguous()
        offs = [0]
        for g in gt_bboxes:
            offs.append(offs[-1] + int(g.size(0)))
        ac, sc = cfg.assigner, cfg.sampler
        if not ac.get('gt_max_assign_all', True):
            raise NotImplementedError('gt_max_assign_all=False')
        assigned = BF.iou_assign(anchors, gt_cat, offs, ac.pos_iou_thr, ac.neg_iou_thr,
                                 ac.get('min_pos_iou', 0.0), valid=inside, shared_boxes=True)
        if not samplers or samplers.get('rpn') is None:      # one launch for the batch (csrc/sampler.hip)
            pos_m, neg_m = BF.sample_pos_neg(assigned, sc.num, sc.pos_fraction,
                                             sc.get('neg_pos_ub', -1))
        else:                     # test hook: caller-supplied draw (oracle/tensor_forms.sampler_hooks)
            pos, neg = [], []
            for i in range(len(img_metas)):
                p, n = samplers['rpn'](assigned[i], sc.num, sc.pos_fraction, sc.get('neg_pos_ub', -1))
                pos.append(p)
                neg.append(n)
            pos_m, neg_m = torch.stack(pos).to(torch.uint8), torch.stack(neg).to(torch.uint8)
        lc, lb, _ = BF.rpn_loss(self._fused, self.num_anchors, anchors, assigned, pos_m, neg_m,
                                gt_cat, offs, self.target_means, self.target_stds,
                                self.loss_bbox.beta, cfg.pos_weight, self.loss_cls.loss_weight,
                                self.loss_bbox.loss_weight)
        L = len(self._fused)
        return dict(loss_rpn_cls=[lc[i] for i in range(L)], loss_rpn_bbox=[lb[i] for i in range(L)])

    # -- proposals ----------------------------------------------------------------------
    @torch.no_grad()
    def get_bboxes(self, cls_scores, bbox_preds, img_metas, cfg, rescale=False):
        """Per image ``[max_num, 5]`` proposals sorted by score + ``valid [max_num]`` mask
        (rpn_head.py:55-104).  Everything stays on the device."""
        if cfg.nms_across_levels or cfg.min_bbox_size > 0:
            raise NotImplementedError('nms_across_levels / min_bbox_size > 0 are not used by the '
                                      'BAGS configs')
        featmap_sizes = [tuple(c.shape[1:3]) for c in cls_scores]
        dev = cls_scores[0].device
        N = cls_scores[0].shape[0]
        L = len(cls_scores)
        nmax = cfg.nms_pre
        if not self._use_fused(cls_scores):
            raise RuntimeError("RPNHead.get_bboxes reads the head's own GPU outputs: call forward() "
                               "on CUDA tensors first (no CPU / tensor-op path in the product)")
        if N * L > 64 or nmax > 4096:
            raise NotImplementedError('bgs_topk_sorted handles <= 64 (image, level) rows of <= 4096 '
                                      'selected entries (got %d rows, nms_pre=%d)' % (N * L, nmax))
        na = self.num_anchors * self.cls_out_channels
        counts = [min(int(c[0].numel()), nmax) for c in cls_scores]
        # sigmoid is monotone: top-k on the logits — all levels in one launch set, read in place from
        # the fused head output (first `na` of the 5 * na channels per pixel)
        top_l, top_i = BF.topk_sorted(self._fused, counts, nmax, inner=na)
        boxes = BF.decode_proposals(self._fused, counts, self.num_anchors,
                                    self._all_anchors(featmap_sizes, dev), top_i, top_l,
                                    [m['img_shape'][:2] for m in img_metas],
                                    self.target_means, self.target_stds)
        return self._nms_and_select(boxes, counts, cfg, N, L, nmax, dev)

    def _nms_and_select(self, boxes, counts, cfg, N, L, nmax, dev):
        ckey = ('cnt', tuple(counts), N, str(dev))
        if ckey not in self._anchor_cache:        # uploaded once (no H2D copy per iteration)
            self._anchor_cache[ckey] = torch.tensor(counts * N, dtype=torch.int32, device=dev)
        cnt = self._anchor_cache[ckey]
        keep, keep_n = BF.nms_batched(boxes.view(N * L, nmax, 5), cnt, cfg.nms_thr, iou_mode=0,
                                      max_keep=cfg.nms_post)
        num = min(cfg.max_num, L * nmax)
        if os.environ.get('BGS_PROPOSAL_TAIL', 'merge') == 'merge' and L <= 16 and L * nmax <= 16384:
            # every level's kept boxes are already in descending score order: the per-image top `max_num` over
            # the levels is an L-way merge — ONE launch (csrc/nms.hip nms_merge_select_kernel) instead of the gather
            # + radix select + sort + gather below (11 launches, 83 us of the cfg[1] step); same boxes, same order
            props, valid = BF.nms_merge_select(boxes.view(N * L, nmax, 5), keep, keep_n, N, num)
            return ProposalList(props, valid)
        # (A/B arm, BGS_PROPOSAL_TAIL=topk) kept boxes into fixed-shape rows (padding slots score -1), then the
        # per-image top `max_num` over the levels: two gathers, one launch each (csrc/nms.hip)
        kept, kept_scores = BF.nms_gather(boxes.view(N * L, nmax, 5), keep, keep_n)
        flat = kept.view(N, L * nmax, 5)
        flat_s = kept_scores.view(N, L * nmax)
        if N > 64 or num > 4096:
            raise NotImplementedError('bgs_topk_sorted: <= 64 rows / <= 4096 selected (got %d, %d)'
                                      % (N, num))
        top_s, top_i = BF.topk_sorted([flat_s], [num], num)
        props, valid = BF.gather_boxes(flat, top_i.view(N, num), top_s.view(N, num))
        return ProposalList(props, valid)
