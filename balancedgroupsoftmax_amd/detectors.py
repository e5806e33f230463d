"""Two-stage detectors behind the reference's ``DETECTORS`` registry keys.

* ``TwoStageDetector``  mmdet/models/detectors/two_stage.py:12-290 (forward_train :134-265)
* ``FasterRCNN``        mmdet/models/detectors/faster_rcnn.py
* ``GroupSoftmax``      mmdet/models/detectors/group_softmax.py:7-29 (an empty subclass)

``forward_train`` returns the reference's loss dict (``loss_rpn_cls``/``loss_rpn_bbox``: lists
over the 5 levels, ``loss_cls_bin0..B-1``, ``loss_bbox``).  The orchestration differs where the
reference synchronises with the host: proposals, RoI assignment and sampling are fixed-shape
device tensors produced by fused kernels (csrc/det_targets.hip, csrc/sampler.hip), so one
training iteration issues no ``.item()`` / ``nonzero()`` / D2H copy at all.  There is ONE path:
CPU tensors raise (the tensor-op restatements used to pin the kernels are test infrastructure,
oracle/tensor_forms.py).
"""
import torch
import torch.nn as nn

from . import builder
from .registry import DETECTORS


_GT_ROWS = {}


def _gt_index_row(G, device):
    """``[1 .. G]`` int32 on ``device`` (the assignment of the GT boxes `add_gt_as_proposals` puts in front
    of the candidates, AssignResult.add_gt_): cached per (G, device) instead of an arange launch per
    image, stage and iteration."""
    key = (int(G), str(device))
    t = _GT_ROWS.get(key)
    if t is None:
        if len(_GT_ROWS) > 4096:
            _GT_ROWS.clear()
        t = _GT_ROWS[key] = torch.arange(1, int(G) + 1, device=device, dtype=torch.int32)
    return t


@DETECTORS.register_module
class TwoStageDetector(nn.Module):

    def __init__(self, backbone, neck=None, shared_head=None, rpn_head=None,
                 bbox_roi_extractor=None, bbox_head=None, mask_roi_extractor=None, mask_head=None,
                 train_cfg=None, test_cfg=None, pretrained=None):
        super().__init__()
        if shared_head is not None:
            raise NotImplementedError('shared_head (C4 heads) is not on the BAGS path')
        self.backbone = builder.build_backbone(backbone)
        self.neck = builder.build_neck(neck) if neck is not None else None
        self.rpn_head = builder.build_head(rpn_head) if rpn_head is not None else None
        self.bbox_roi_extractor = builder.build_roi_extractor(bbox_roi_extractor) \
            if bbox_head is not None else None
        self.bbox_head = builder.build_head(bbox_head) if bbox_head is not None else None
        self.mask_head = None
        if mask_head is not None:
            if mask_roi_extractor is not None:
                self.mask_roi_extractor = builder.build_roi_extractor(mask_roi_extractor)
                self.share_roi_extractor = False
            else:
                self.share_roi_extractor = True
                self.mask_roi_extractor = self.bbox_roi_extractor
            self.mask_head = builder.build_head(mask_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.fp16_enabled = False
        self.init_weights(pretrained=pretrained)

    with_neck = property(lambda self: self.neck is not None)
    with_rpn = property(lambda self: self.rpn_head is not None)
    with_bbox = property(lambda self: self.bbox_head is not None)
    with_mask = property(lambda self: self.mask_head is not None)
    with_shared_head = property(lambda self: False)

    def init_weights(self, pretrained=None):
        # local checkpoint paths are loaded into the backbone, model-zoo URLs warn loudly
        # (backbone.ResNet.init_weights; reference: two_stage.py:66-68 -> resnet.py:496-499)
        self.backbone.init_weights(pretrained=pretrained)
        if self.with_neck:
            self.neck.init_weights()
        if self.with_rpn:
            self.rpn_head.init_weights()
        if self.with_bbox:
            self.bbox_roi_extractor.init_weights()
            self.bbox_head.init_weights()
        if self.with_mask:
            self.mask_head.init_weights()
            if not self.share_roi_extractor:
                self.mask_roi_extractor.init_weights()

    def extract_feat(self, img):
        x = self.backbone(img)
        return self.neck(x) if self.with_neck else x

    # -- RoI assignment + sampling (two_stage.py:192-210), fixed shape ---------------------
    def _sample_rois_fused(self, proposal_list, gt_bboxes, gt_labels, samplers=None, rc=None,
                           head=None):
        """two_stage.py:192-222: one batched assignment launch pair, the fixed-shape sampler, and
        one kernel that emits rois / labels / box targets for all images.
        ``rc`` / ``head``: the stage's rcnn config and bbox head (cascade); defaults: the
        detector's own.  ``samplers``: test hook (``dict(rcnn=fn)``: a caller-supplied draw
        instead of the device RandomSampler; oracle/tensor_forms.sampler_hooks)."""
        from . import functional as BF
        rc = self.train_cfg.rcnn if rc is None else rc
        ac, sc = rc.assigner, rc.sampler
        N = len(proposal_list)
        if hasattr(proposal_list, 'batched'):       # rpn.ProposalList: the batch tensors themselves
            props, pv = proposal_list.batched
            props = props.contiguous()
            pvalid = (pv.view(torch.uint8) if pv.dtype == torch.bool else pv.to(torch.uint8)).contiguous()
        else:
            props = torch.stack([p for p, _ in proposal_list]).contiguous()            # [N, P, 5]
            pvalid = torch.stack([v for _, v in proposal_list]).to(torch.uint8).contiguous()
        gt_cat = torch.cat([g[:, :4] for g in gt_bboxes]).float().contiguous()
        offs = [0]
        for g in gt_bboxes:
            offs.append(offs[-1] + int(g.size(0)))
        assigned = BF.iou_assign(props, gt_cat, offs, ac.pos_iou_thr, ac.neg_iou_thr,
                                 ac.get('min_pos_iou', 0.0), valid=pvalid, shared_boxes=False)
        add_gt = sc.get('add_gt_as_proposals', True)
        rcnn_hook = samplers.get('rcnn') if samplers else None
        boxes_l, assigned_l, inds_l, valid_l = [], [], [], []
        for i in range(N):
            b, a = props[i, :, :4], assigned[i]
            if add_gt:      # base_sampler.py:49-53 + AssignResult.add_gt_
                G = gt_bboxes[i].size(0)
                b = torch.cat([gt_bboxes[i][:, :4].float(), b], 0)
                a = torch.cat([_gt_index_row(G, a.device), a])
            boxes_l.append(b.contiguous())
            assigned_l.append(a.contiguous())
            if rcnn_hook is not None:      # test hook: caller-supplied draw
                inds, _, valid = rcnn_hook(a, sc.num, sc.pos_fraction)
                inds_l.append(inds.contiguous())
                valid_l.append(valid)
        if rcnn_hook is None:
            if any(a.numel() > 4096 for a in assigned_l):
                raise NotImplementedError('bgs_sample_rois sorts <= 4096 candidates per image in LDS '
                                          '(rpn_proposal.max_num + GT boxes)')
            # one launch for the batch (csrc/sampler.hip: sort of (class, random key) composites)
            need_extra = self.with_mask or isinstance(self.bbox_head, nn.ModuleList)
            res = BF.sample_rois(assigned_l, sc.num, sc.pos_fraction,
                                 gt_counts=[int(g.size(0)) if add_gt else 0 for g in gt_bboxes]
                                 if need_extra else None)
            inds_all, valid_all = res[0], res[2].view(torch.bool)
            inds_l = [inds_all[i] for i in range(N)]
            valid_l = [valid_all[i] for i in range(N)]
            if need_extra:
                # what the mask branch and the cascade refinement read from the SamplingResult come out
                # of the sampling launch itself (bgs_sample_rois_ex) instead of ~13 gather / compare /
                # stack launches per stage
                self._sampled_gt_inds = res[3]
                self._sampled_valid = valid_all
                self._sampled_is_gt = res[4].view(torch.bool)
        if rcnn_hook is not None and (self.with_mask or isinstance(self.bbox_head, nn.ModuleList)):
            # (only the mask branch and the cascade refinement read these: skipped for the plain
            #  box detectors, ~10 small launches)
            # gt index of every sampled RoI (pos_assigned_gt_inds for the mask targets), -1 = none
            self._sampled_gt_inds = torch.stack(
                [assigned_l[i].gather(0, inds_l[i]).to(torch.int32) - 1 for i in range(N)])
            # which sampled RoIs are GT boxes added as proposals (SamplingResult.pos_is_gt), and
            # which slots are real (fewer candidates than `num` leaves padding slots)
            self._sampled_valid = torch.stack(valid_l)
            self._sampled_is_gt = torch.stack(
                [(inds_l[i] < gt_bboxes[i].size(0)) if add_gt else torch.zeros_like(valid_l[i])
                 for i in range(N)])
        head = self.bbox_head if head is None else head
        rois, labels, lw, bt, bw = BF.rcnn_targets(
            boxes_l, assigned_l, inds_l, valid_l, [g.contiguous() for g in gt_labels], gt_cat, offs,
            sc.num, head.target_means, head.target_stds, rc.pos_weight)
        return rois, (labels, lw, bt, bw)

    # -- RPN part of a training iteration ------------------------------------------------------
    def _rpn_forward_train(self, x, img_meta, gt_bboxes, proposals, samplers, losses, fork_loss=False):
        """RPN losses + the fixed-shape proposal list (two_stage.py:157-176).

        (Measured and dropped: launching the loss branch — anchor assignment, sampler, BCE +
        SmoothL1 sums, ~0.45 ms of small latency-bound kernels that only read the RPN outputs — on
        a second HIP stream next to the proposal -> RoI-head chain.  Correct and stable over 400
        graph replays, but 12.00 vs 12.05 ms: the replayed graph does not run the two branches
        concurrently, and eager launches are host-bound.  profiles/r2w_*.)"""
        if not self.with_rpn:
            return [(p, torch.ones(p.size(0), dtype=torch.bool, device=p.device)) for p in proposals]
        cls_scores, bbox_preds = self.rpn_head(x)
        from . import functional as BF
        self._rpn_loss_fork = None
        if fork_loss and cls_scores[0].is_cuda and BF.rpn_loss_fork_enabled() and not samplers and \
                not (torch.is_grad_enabled() and cls_scores[0].requires_grad):
            # the RPN loss chain (anchor assignment, sampler, BCE + SmoothL1 sums: ~12 short launches that only
            # read the RPN outputs) beside the proposal -> RoI-head chain; joined in forward_train before the
            # losses are returned.  (Round 2 measured no gain: the step was 560 launches and eager launching was
            # host-bound; at 150 launches it is not.)
            with BF.forked(cls_scores[0].device, lane=1) as fk:
                losses.update(self.rpn_head.loss(cls_scores, bbox_preds, gt_bboxes, img_meta,
                                                 self.train_cfg.rpn, samplers=samplers))
            # the chain reads the RPN outputs (main / lane-0 pools) long after this function has dropped its
            # references (``_fused = None`` below, the return): held until ``_join_rpn_loss`` so that the caching
            # allocator cannot hand their blocks to the RoI stage while ``rpn_loss_kernel`` is still reading them
            fk.hold(cls_scores, bbox_preds, self.rpn_head._fused, gt_bboxes)
            self._rpn_loss_fork = fk
        else:
            losses.update(self.rpn_head.loss(cls_scores, bbox_preds, gt_bboxes, img_meta,
                                             self.train_cfg.rpn, samplers=samplers))
        proposal_cfg = self.train_cfg.get('rpn_proposal', None)
        if proposal_cfg is None:
            proposal_cfg = self.test_cfg.rpn
        proposal_list = self.rpn_head.get_bboxes(cls_scores, bbox_preds, img_meta, proposal_cfg)
        if samplers is not None and 'proposals' in samplers:
            # test hook (tests/test_gpu_e2e.py, shipped-sampler golden): the RoI stage runs on a
            # caller-supplied proposal list so that recorded sampler indices name the same boxes on
            # both sides; the detector's own proposals stay inspectable
            self._own_proposals = proposal_list
            proposal_list = samplers['proposals'](proposal_list)
        # the head's stash of its own outputs carries this iteration's autograd graph when the
        # trunk trains (selectp=0): drop it, or the graph (and its AccumulateGrad nodes, bound to
        # the stream they were created on) would outlive the iteration
        self.rpn_head._fused = None
        return proposal_list

    def _join_rpn_loss(self):
        """Make the current stream wait for the RPN loss chain launched beside it (see _rpn_forward_train)."""
        fk = getattr(self, '_rpn_loss_fork', None)
        if fk is not None:
            fk.join()
            self._rpn_loss_fork = None

    def trunk_is_frozen(self):
        """True when no parameter of the backbone / neck trains (the shipped ``selectp = 1`` / ``3``): their features
        are a pure function of the image, so a training loop may compute the NEXT batch's features while this batch's
        heads, losses, backward and optimizer step run (``train.TrunkPipeline``)."""
        mods = [self.backbone] + ([self.neck] if self.with_neck else [])
        return not any(p.requires_grad for m in mods for p in m.parameters())

    def forward_train(self, img, img_meta, gt_bboxes, gt_labels, gt_bboxes_ignore=None,
                      gt_masks=None, proposals=None, samplers=None, feats=None):
        # ``feats``: the output of ``extract_feat(img)`` computed ahead of this call (train.TrunkPipeline)
        x = self.extract_feat(img) if feats is None else feats
        losses = dict()
        proposal_list = self._rpn_forward_train(x, img_meta, gt_bboxes, proposals, samplers, losses,
                                                fork_loss=True)      # (joined below)
        if self.with_bbox:
            if not self.train_cfg.rcnn.assigner.get('gt_max_assign_all', True):
                raise NotImplementedError('gt_max_assign_all=False')
            rois, targets = self._sample_rois_fused(proposal_list, gt_bboxes, gt_labels, samplers)
            from . import functional as BF
            mask_fk = None
            if self.with_mask and rois.is_cuda and BF.rpn_loss_fork_enabled() and not (
                    torch.is_grad_enabled() and any(p.requires_grad for p in self.mask_head.parameters())):
                # frozen mask branch (mask RoIAlign, four convs, deconv, targets, BCE): independent of the box
                # head — beside it on its own stream when the step is launched eagerly
                with BF.forked(rois.device, lane=2) as mask_fk:
                    mask_losses = self._mask_forward_train(x, rois, targets[0], gt_masks, img.size(0))
                mask_fk.hold(x, rois, targets, gt_masks)
            bbox_feats = self.bbox_roi_extractor(x[:self.bbox_roi_extractor.num_inputs], rois)
            cls_score, bbox_pred = self.bbox_head(bbox_feats, nhwc=True)
            losses.update(self.bbox_head.loss(cls_score, bbox_pred, *targets))
            if mask_fk is not None:
                mask_fk.join()
                losses.update(mask_losses)
            elif self.with_mask:
                losses.update(self._mask_forward_train(x, rois, targets[0], gt_masks, img.size(0)))
        elif self.with_mask:
            losses.update(self._mask_forward_train(x, rois, targets[0], gt_masks, img.size(0)))
        self._join_rpn_loss()
        return losses

    def _mask_forward_train(self, x, rois, labels, gt_masks, num_imgs):
        """two_stage.py:228-263 with a FIXED number of mask RoIs: the sampler puts the positives
        first (at most ``int(num * pos_fraction)`` per image), so the first ``max_pos`` slots of
        every image are the candidates and ``labels > 0`` says which of them are real; padding
        slots get zero weight in the mean instead of being filtered out on the host."""
        if gt_masks is None:
            raise ValueError('the mask branch needs gt_masks (per image a uint8 [G, H, W] tensor)')
        if not rois.is_cuda:
            raise NotImplementedError('the mask branch runs on the GPU path only')
        sc = self.train_cfg.rcnn.sampler
        max_pos = int(sc.num * sc.pos_fraction)
        num = sc.num
        sel = (torch.arange(num_imgs, device=rois.device).view(-1, 1) * num +
               torch.arange(max_pos, device=rois.device).view(1, -1)).reshape(-1)
        pos_rois = rois[sel].contiguous()
        pos_labels = labels[sel].contiguous()
        valid = pos_labels > 0
        gt_inds = self._sampled_gt_inds[:, :max_pos].reshape(-1).contiguous()
        masks = [torch.as_tensor(m).to(device=rois.device, dtype=torch.uint8).contiguous()
                 for m in gt_masks]
        if self.share_roi_extractor:
            raise NotImplementedError('share_roi_extractor=True (7x7 features for the mask head) '
                                      'is not used by the BAGS configs')
        mask_feats = self.mask_roi_extractor(x[:self.mask_roi_extractor.num_inputs], pos_rois)
        feats = self.mask_head.features(mask_feats, nhwc=True)
        mask_targets = self.mask_head.get_target_fixed(pos_rois, gt_inds, valid, masks,
                                                       self.train_cfg.rcnn)
        return self.mask_head.loss_from_features(feats, mask_targets, pos_labels, valid)

    # ------------------------------------------------------------------ test-time path
    def simple_test_rpn(self, x, img_meta, rpn_test_cfg):
        """test_mixins.py:8-12; proposals stay fixed-shape ``([max_num,5], valid)`` per image."""
        cls_scores, bbox_preds = self.rpn_head(x)
        return self.rpn_head.get_bboxes(cls_scores, bbox_preds, img_meta, rpn_test_cfg)

    def simple_test_bboxes(self, x, img_meta, proposals, rcnn_test_cfg, rescale=False):
        """test_mixins.py:39-67 for ONE image (the reference tests with imgs_per_gpu=1):
        RoIAlign -> head -> merged scores + decoded boxes -> one batched 1230-class NMS."""
        from .post_processing import multiclass_nms
        props, valid = proposals[0] if isinstance(proposals[0], tuple) else (proposals[0], None)
        rois = torch.cat([props.new_zeros((props.size(0), 1)), props[:, :4]], dim=1)
        feats = self.bbox_roi_extractor(x[:self.bbox_roi_extractor.num_inputs], rois)
        cls_score, bbox_pred = self.bbox_head(feats, nhwc=True)
        bboxes, scores = self.bbox_head.get_det_bboxes(
            rois, cls_score, bbox_pred, img_meta[0]['img_shape'], img_meta[0]['scale_factor'],
            rescale=rescale, cfg=None)
        if valid is not None:        # padding rows of the fixed-shape proposal list never survive
            scores = torch.where(valid[:, None], scores, scores.new_full((), -1.0))
        det_bboxes, det_labels = multiclass_nms(bboxes, scores, rcnn_test_cfg.score_thr,
                                                rcnn_test_cfg.nms, rcnn_test_cfg.max_per_img)
        return det_bboxes, det_labels, scores

    def simple_test(self, img, img_meta, proposals=None, rescale=False, feats=None):
        """two_stage.py:267-289 (bbox branch): list of ``num_classes-1`` ``[k_c, 5]`` arrays.
        ``feats``: ``extract_feat(img)`` computed ahead of this call (``train.TrunkPipeline(inference=True)``)."""
        from .post_processing import bbox2result
        assert self.with_bbox, 'Bbox head must be implemented.'
        x = self.extract_feat(img) if feats is None else feats
        proposal_list = (self.simple_test_rpn(x, img_meta, self.test_cfg.rpn)
                         if proposals is None else proposals)
        det_bboxes, det_labels, _ = self.simple_test_bboxes(x, img_meta, proposal_list,
                                                            self.test_cfg.rcnn, rescale=rescale)
        bbox_results = bbox2result(det_bboxes, det_labels, self.bbox_head.num_classes)
        if not self.with_mask:
            return bbox_results
        return bbox_results, self.simple_test_mask(x, img_meta, det_bboxes, det_labels,
                                                   rescale=rescale)

    def simple_test_mask(self, x, img_meta, det_bboxes, det_labels, rescale=False, paste=False, encode=None):
        """test_mixins.py:153-180.  Default: the per-detection mask probabilities ``[k, 28, 28]`` of each
        detection's own class (device tensor).  ``paste=True``: the reference's return value — ``cls_segms`` of
        ``FCNMaskHead.get_seg_masks`` (per class the masks pasted into the ``ori_shape`` image, resized /
        thresholded on the device; dense ``uint8`` unless ``encode`` produces RLEs, see ``get_seg_masks``)."""
        if det_bboxes.shape[0] == 0:
            if paste:
                return [[] for _ in range(self.mask_head.num_classes - 1)]
            return det_bboxes.new_zeros((0, 28, 28))
        boxes = det_bboxes[:, :4] * img_meta[0]['scale_factor'] if rescale else det_bboxes[:, :4]
        rois = torch.cat([boxes.new_zeros((boxes.size(0), 1)), boxes], dim=1)
        feats = self.mask_roi_extractor(x[:self.mask_roi_extractor.num_inputs], rois)
        probs = self.mask_head.get_mask_probs(self.mask_head.features(feats, nhwc=True), det_labels)
        if not paste:
            return probs
        return self.mask_head.get_seg_masks(probs, boxes, det_labels, self.test_cfg.rcnn,
                                            img_meta[0]['ori_shape'], img_meta[0]['scale_factor'], rescale,
                                            encode=encode)

    def forward_test(self, imgs, img_metas, **kwargs):
        """base.py forward_test: one scale only (aug_test / TTA is not on the BAGS path)."""
        if isinstance(imgs, (list, tuple)):
            if len(imgs) != 1:
                raise NotImplementedError('multi-scale aug_test is not part of the BAGS path')
            imgs, img_metas = imgs[0], img_metas[0]
        with torch.no_grad():
            return self.simple_test(imgs, img_metas, **kwargs)

    def forward(self, img, img_meta, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_meta, **kwargs)
        return self.forward_test(img, img_meta, **kwargs)


@DETECTORS.register_module
class FasterRCNN(TwoStageDetector):
    pass


@DETECTORS.register_module
class MaskRCNN(TwoStageDetector):
    """mmdet/models/detectors/mask_rcnn.py: the two-stage detector with the mask branch
    (configs/bags/gs_mask_rcnn_r50_fpn_1x_lvis.py = cfg 4)."""
    pass


@DETECTORS.register_module
class CascadeRCNN(TwoStageDetector):
    """mmdet/models/detectors/cascade_rcnn.py:15-420 (bbox branch; cfg 5 =
    configs/bags/gs_cascade_rcnn_x101_64x4d_fpn_1x_lvis.py): ``num_stages`` RoI heads trained at
    increasing IoU thresholds, each stage re-sampling from the boxes the previous stage refined.

    Fixed-shape differences: stage ``i+1``'s proposals are ALL ``sampler.num`` refined RoIs of
    stage ``i`` with a validity mask (``pos_is_gt`` rows and padding slots masked out) instead of
    a filtered list, so the stage loop issues no host synchronisation either."""

    def __init__(self, num_stages, backbone, neck=None, shared_head=None, rpn_head=None,
                 bbox_roi_extractor=None, bbox_head=None, mask_roi_extractor=None, mask_head=None,
                 train_cfg=None, test_cfg=None, pretrained=None):
        assert bbox_roi_extractor is not None and bbox_head is not None
        if shared_head is not None:
            raise NotImplementedError('shared_head (C4 heads) is not on the BAGS path')
        nn.Module.__init__(self)
        self.num_stages = num_stages
        self.backbone = builder.build_backbone(backbone)
        self.neck = builder.build_neck(neck) if neck is not None else None
        self.rpn_head = builder.build_head(rpn_head) if rpn_head is not None else None

        def per_stage(cfg):
            cfg = list(cfg) if isinstance(cfg, (list, tuple)) else [cfg] * num_stages
            assert len(cfg) == num_stages
            return cfg
        self.bbox_roi_extractor = nn.ModuleList(builder.build_roi_extractor(r)
                                                for r in per_stage(bbox_roi_extractor))
        self.bbox_head = nn.ModuleList(builder.build_head(h) for h in per_stage(bbox_head))
        self.mask_head = None
        if mask_head is not None:       # cascade_rcnn.py:67-92 (used by HybridTaskCascade)
            self.mask_head = nn.ModuleList(builder.build_head(h) for h in per_stage(mask_head))
            if mask_roi_extractor is not None:
                self.share_roi_extractor = False
                self.mask_roi_extractor = nn.ModuleList(builder.build_roi_extractor(r)
                                                        for r in per_stage(mask_roi_extractor))
            else:
                self.share_roi_extractor = True
                self.mask_roi_extractor = self.bbox_roi_extractor
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.fp16_enabled = False
        self._build_extra()
        self.init_weights(pretrained=pretrained)

    def _build_extra(self):
        if self.mask_head is not None:
            raise NotImplementedError('Cascade Mask R-CNN is not among the BAGS configs; the '
                                      'cascade mask branch is built as HybridTaskCascade')

    def init_weights(self, pretrained=None):
        self.backbone.init_weights(pretrained=pretrained)
        if self.with_neck:
            self.neck.init_weights()
        if self.with_rpn:
            self.rpn_head.init_weights()
        for ext, head in zip(self.bbox_roi_extractor, self.bbox_head):
            ext.init_weights()
            head.init_weights()
        if self.mask_head is not None:
            for head in self.mask_head:
                head.init_weights()

    def _refined_proposals(self, head, rois, labels, bbox_pred, img_meta, num):
        """``refine_bboxes`` (bbox_head.py:169-208) in fixed shape: every sampled RoI re-regressed
        with its target class; GT rows and padding slots are masked instead of removed."""
        from . import functional as BF
        n_img = len(img_meta)
        with torch.no_grad():
            # regress_by_class + delta2bbox for every image in ONE launch (csrc/det_targets.hip:
            # refine_boxes_kernel; the tensor form is ~35 element-wise launches per image and stage)
            boxes = BF.refine_boxes(rois.contiguous(), labels.contiguous(), bbox_pred.detach().contiguous(),
                                    [m['img_shape'] for m in img_meta], head.target_means,
                                    head.target_stds)
            keep = self._sampled_valid & ~self._sampled_is_gt
            from .rpn import ProposalList
            return ProposalList(boxes.view(n_img, num, 4), keep)

    def forward_train(self, img, img_meta, gt_bboxes, gt_labels, gt_bboxes_ignore=None,
                      gt_masks=None, proposals=None, samplers=None, feats=None):
        if not img.is_cuda:
            raise NotImplementedError('CascadeRCNN.forward_train runs on the GPU path only')
        x = self.extract_feat(img) if feats is None else feats
        losses = dict()
        proposal_list = self._rpn_forward_train(x, img_meta, gt_bboxes, proposals, samplers, losses, fork_loss=True)
        for i in range(self.num_stages):
            rc = self.train_cfg.rcnn[i]
            lw = self.train_cfg.stage_loss_weights[i]
            head, ext = self.bbox_head[i], self.bbox_roi_extractor[i]
            rois, targets = self._sample_rois_fused(proposal_list, gt_bboxes, gt_labels, samplers,
                                                    rc=rc, head=head)
            feats = ext(x[:ext.num_inputs], rois)
            cls_score, bbox_pred = head(feats, nhwc=True)
            if getattr(head, 'fused_loss_scale', False):      # GS heads: the stage weight rides in the kernel
                for name, value in head.loss(cls_score, bbox_pred, *targets, loss_scale=lw).items():
                    losses['s{}.{}'.format(i, name)] = value
            else:
                for name, value in head.loss(cls_score, bbox_pred, *targets).items():
                    losses['s{}.{}'.format(i, name)] = value * lw if 'loss' in name else value
            if i < self.num_stages - 1:       # refine (cascade_rcnn.py:291-296), fixed shape
                proposal_list = self._refined_proposals(head, rois, targets[0], bbox_pred, img_meta,
                                                        rc.sampler.num)
        self._join_rpn_loss()
        return losses

    def simple_test(self, img, img_meta, proposals=None, rescale=False, feats=None):
        """cascade_rcnn.py:300-393 (bbox branch, ensemble result): every stage re-regresses the
        1000 RoIs with its arg-max class, the class logits are averaged over the stages."""
        from .post_processing import bbox2result, multiclass_nms
        x = self.extract_feat(img) if feats is None else feats      # (ahead: train.TrunkPipeline(inference=True))
        proposal_list = (self.simple_test_rpn(x, img_meta, self.test_cfg.rpn)
                         if proposals is None else proposals)
        props, valid = proposal_list[0] if isinstance(proposal_list[0], tuple) \
            else (proposal_list[0], None)
        rois = torch.cat([props.new_zeros((props.size(0), 1)), props[:, :4]], dim=1)
        ms_scores = []
        for i in range(self.num_stages):
            head, ext = self.bbox_head[i], self.bbox_roi_extractor[i]
            cls_score, bbox_pred = head(ext(x[:ext.num_inputs], rois), nhwc=True)
            ms_scores.append(cls_score)
            if i < self.num_stages - 1:
                bbox_label = cls_score.argmax(dim=1)
                rois = head.regress_by_class(rois, bbox_label, bbox_pred, img_meta[0])
        cls_score = sum(ms_scores) / float(self.num_stages)
        bboxes, scores = self.bbox_head[-1].get_det_bboxes(
            rois, cls_score, bbox_pred, img_meta[0]['img_shape'], img_meta[0]['scale_factor'],
            rescale=rescale, cfg=None)
        if valid is not None:
            scores = torch.where(valid[:, None], scores, scores.new_full((), -1.0))
        cfg = self.test_cfg.rcnn
        det_bboxes, det_labels = multiclass_nms(bboxes, scores, cfg.score_thr, cfg.nms,
                                                cfg.max_per_img)
        return bbox2result(det_bboxes, det_labels, self.bbox_head[-1].num_classes)


@DETECTORS.register_module
class HybridTaskCascade(CascadeRCNN):
    """mmdet/models/detectors/htc.py:12-561 (configs/bags/gs_htc_x101_64x4d_fpn_20e_16gpu_lvis.py):
    the cascade with (a) interleaved execution — stage ``i``'s mask head trains on the boxes stage
    ``i``'s box head just refined, (b) mask information flow — ``HTCMaskHead.conv_res`` feeds the
    previous stage's mask feature forward, (c) a semantic-segmentation branch whose embedded
    feature is RoI-pooled and added to the box and mask RoI features.

    GPU specifics: the semantic fusion of the box branch (RoIAlign 14x14 on the stride-8 map ->
    ``adaptive_avg_pool2d`` to 7x7 -> ``bbox_feats +=``) is ONE kernel launch that accumulates into
    the box features (``bgs_roi_align_nhwc_fwd_ex`` with ``pool=2``); mask logits exist only for
    each RoI's own class channel; all sampling is fixed shape (no host synchronisation)."""

    def __init__(self, num_stages, backbone, semantic_roi_extractor=None, semantic_head=None,
                 semantic_fusion=('bbox', 'mask'), interleaved=True, mask_info_flow=True, **kwargs):
        nn.Module.__init__(self)
        self._semantic_cfg = (semantic_roi_extractor, semantic_head)
        self.semantic_fusion = tuple(semantic_fusion)
        self.interleaved = interleaved
        self.mask_info_flow = mask_info_flow
        super().__init__(num_stages, backbone, **kwargs)
        assert self.with_bbox and self.with_mask

    def _build_extra(self):
        ext, head = self._semantic_cfg
        self.semantic_head = None
        if head is not None:
            self.semantic_roi_extractor = builder.build_roi_extractor(ext)
            self.semantic_head = builder.build_head(head)
        if self.mask_head is None or self.share_roi_extractor:
            raise NotImplementedError('HTC without its own mask_roi_extractor / mask_head is '
                                      'outside the BAGS configs')

    with_semantic = property(lambda self: self.semantic_head is not None)

    def init_weights(self, pretrained=None):
        super().init_weights(pretrained=pretrained)
        if self.with_semantic:
            self.semantic_head.init_weights()

    # -- RoI features with the semantic fusion ---------------------------------------------
    def _fused_roi_feats(self, ext, x, rois, semantic_feat, branch):
        feats = ext(x[:ext.num_inputs], rois)
        if semantic_feat is not None and branch in self.semantic_fusion:
            sext = self.semantic_roi_extractor
            pool, rem = divmod(sext.out_size, ext.out_size)
            if rem != 0 or pool not in (1, 2):
                raise NotImplementedError('semantic RoI size %d vs %d' % (sext.out_size, ext.out_size))
            # htc.py:57-64 / 88-96: RoIAlign (+ adaptive_avg_pool2d) + in-place add, one launch
            feats = sext([semantic_feat], rois, out_size=ext.out_size, pool=pool, add_to=feats)
        return feats

    def _mask_features(self, stage, mask_feats, upto_logits=True):
        """Mask information flow (htc.py:98-107): heads ``0..stage-1`` contribute their conv
        features through ``conv_res``; returns stage ``stage``'s pre-logit features."""
        head = self.mask_head[stage]
        if not self.mask_info_flow:
            return head.upsample_features(head.conv_features(mask_feats))
        last = None
        for i in range(stage):
            last = self.mask_head[i].res_features(mask_feats, last)
        return head.upsample_features(head.res_features(mask_feats, last))

    def _htc_mask_forward_train(self, stage, x, rois, labels, gt_masks, rc, semantic_feat):
        """htc.py:75-112 on the fixed-shape positives (the first ``int(num * pos_fraction)``
        sampler slots of every image; ``labels > 0`` marks the real ones)."""
        sc = rc.sampler
        n_img = len(gt_masks)
        max_pos = int(sc.num * sc.pos_fraction)
        sel = (torch.arange(n_img, device=rois.device).view(-1, 1) * sc.num +
               torch.arange(max_pos, device=rois.device).view(1, -1)).reshape(-1)
        pos_rois = rois[sel].contiguous()
        pos_labels = labels[sel].contiguous()
        valid = pos_labels > 0
        gt_inds = self._sampled_gt_inds[:, :max_pos].reshape(-1).contiguous()
        masks = [torch.as_tensor(m).to(device=rois.device, dtype=torch.uint8).contiguous()
                 for m in gt_masks]
        ext, head = self.mask_roi_extractor[stage], self.mask_head[stage]
        mask_feats = self._fused_roi_feats(ext, x, pos_rois, semantic_feat, 'mask')
        feats = self._mask_features(stage, mask_feats)
        mask_targets = head.get_target_fixed(pos_rois, gt_inds, valid, masks, rc)
        return head.loss_from_features(feats, mask_targets, pos_labels, valid)

    def forward_train(self, img, img_meta, gt_bboxes, gt_labels, gt_bboxes_ignore=None,
                      gt_masks=None, gt_semantic_seg=None, proposals=None, samplers=None, feats=None):
        if not img.is_cuda:
            raise NotImplementedError('HybridTaskCascade.forward_train runs on the GPU path only')
        if gt_masks is None:
            raise ValueError('HTC needs gt_masks (per image a uint8 [G, H, W] tensor)')
        x = self.extract_feat(img) if feats is None else feats
        losses = dict()
        proposal_list = self._rpn_forward_train(x, img_meta, gt_bboxes, proposals, samplers, losses, fork_loss=True)
        semantic_feat = None
        if self.with_semantic:
            if gt_semantic_seg is None:
                raise ValueError('the semantic branch needs gt_semantic_seg [N, 1, H/8, W/8]')
            semantic_pred, semantic_feat = self.semantic_head(x)
            losses['loss_semantic_seg'] = self.semantic_head.loss(semantic_pred, gt_semantic_seg)
        for i in range(self.num_stages):
            rc = self.train_cfg.rcnn[i]
            lw = self.train_cfg.stage_loss_weights[i]
            head, ext = self.bbox_head[i], self.bbox_roi_extractor[i]
            num = rc.sampler.num
            rois, targets = self._sample_rois_fused(proposal_list, gt_bboxes, gt_labels, samplers,
                                                    rc=rc, head=head)
            feats = self._fused_roi_feats(ext, x, rois, semantic_feat, 'bbox')
            cls_score, bbox_pred = head(feats, nhwc=True)
            if getattr(head, 'fused_loss_scale', False):      # GS heads: the stage weight rides in the kernel
                for name, value in head.loss(cls_score, bbox_pred, *targets, loss_scale=lw).items():
                    losses['s{}.{}'.format(i, name)] = value
            else:
                for name, value in head.loss(cls_score, bbox_pred, *targets).items():
                    losses['s{}.{}'.format(i, name)] = value * lw if 'loss' in name else value
            refined = None
            if self.interleaved or i < self.num_stages - 1:
                refined = self._refined_proposals(head, rois, targets[0], bbox_pred, img_meta, num)
            mask_rois, mask_labels = rois, targets[0]
            if self.interleaved:
                # htc.py:264-283: the mask branch trains on RoIs re-assigned and re-sampled from
                # the boxes this stage's box head just refined
                with torch.no_grad():
                    mask_rois, mt = self._sample_rois_fused(refined, gt_bboxes, gt_labels, samplers,
                                                            rc=rc, head=head)
                    mask_labels = mt[0]
            loss_mask = self._htc_mask_forward_train(i, x, mask_rois, mask_labels, gt_masks, rc,
                                                     semantic_feat)
            for name, value in loss_mask.items():
                losses['s{}.{}'.format(i, name)] = value * lw if 'loss' in name else value
            if refined is not None:
                proposal_list = refined
        self._join_rpn_loss()
        return losses

    # -- test time -----------------------------------------------------------------------------
    def simple_test(self, img, img_meta, proposals=None, rescale=False, feats=None):
        """htc.py:313-432 with ``keep_all_stages=False``: the ensemble boxes (stage-averaged class
        logits) and, per detection, the mean over stages of its class's mask probability
        ``[k, 28, 28]`` (``merge_aug_masks`` without weights; pasting / RLE is evaluation tooling)."""
        from .post_processing import bbox2result
        det_bboxes, det_labels, masks = self.simple_test_dets(img, img_meta, proposals, rescale, feats=feats)
        return bbox2result(det_bboxes, det_labels, self.bbox_head[-1].num_classes), masks

    def simple_test_dets(self, img, img_meta, proposals=None, rescale=False, feats=None):
        """-> ``(det_bboxes [k,5], det_labels [k], mask_probs [k,28,28])`` device tensors."""
        from .post_processing import multiclass_nms
        if self.test_cfg.get('keep_all_stages', False):
            raise NotImplementedError('keep_all_stages=True (per-stage results) is not built')
        x = self.extract_feat(img) if feats is None else feats      # (ahead: train.TrunkPipeline(inference=True))
        proposal_list = (self.simple_test_rpn(x, img_meta, self.test_cfg.rpn)
                         if proposals is None else proposals)
        semantic_feat = self.semantic_head(x)[1] if self.with_semantic else None
        props, valid = proposal_list[0] if isinstance(proposal_list[0], tuple) \
            else (proposal_list[0], None)
        rois = torch.cat([props.new_zeros((props.size(0), 1)), props[:, :4]], dim=1)
        ms_scores = []
        for i in range(self.num_stages):
            head, ext = self.bbox_head[i], self.bbox_roi_extractor[i]
            cls_score, bbox_pred = head(self._fused_roi_feats(ext, x, rois, semantic_feat, 'bbox'),
                                        nhwc=True)
            ms_scores.append(cls_score)
            if i < self.num_stages - 1:
                rois = head.regress_by_class(rois, cls_score.argmax(dim=1), bbox_pred, img_meta[0])
        cls_score = sum(ms_scores) / float(len(ms_scores))
        scale_factor = img_meta[0]['scale_factor']
        bboxes, scores = self.bbox_head[-1].get_det_bboxes(
            rois, cls_score, bbox_pred, img_meta[0]['img_shape'], scale_factor, rescale=rescale,
            cfg=None)
        if valid is not None:
            scores = torch.where(valid[:, None], scores, scores.new_full((), -1.0))
        cfg = self.test_cfg.rcnn
        det_bboxes, det_labels = multiclass_nms(bboxes, scores, cfg.score_thr, cfg.nms,
                                                cfg.max_per_img)
        if det_bboxes.shape[0] == 0:
            return det_bboxes, det_labels, det_bboxes.new_zeros((0, 28, 28))
        boxes = det_bboxes[:, :4] * scale_factor if rescale else det_bboxes[:, :4]
        mask_rois = torch.cat([boxes.new_zeros((boxes.size(0), 1)), boxes], dim=1)
        return det_bboxes, det_labels, self._ensemble_masks(x, mask_rois, det_labels, semantic_feat)

    def _ensemble_masks(self, x, mask_rois, det_labels, semantic_feat):
        """htc.py:379-405: every stage's mask head on the final boxes' features (mask information
        flow through ``conv_res``), mean of the per-class probabilities."""
        mask_feats = self._fused_roi_feats(self.mask_roi_extractor[-1], x, mask_rois, semantic_feat,
                                           'mask')
        probs, last = [], None
        for i in range(self.num_stages):
            head = self.mask_head[i]
            last = head.res_features(mask_feats, last) if self.mask_info_flow \
                else head.conv_features(mask_feats)
            probs.append(head.get_mask_probs(head.upsample_features(last), det_labels))
        return sum(probs) / float(len(probs))


@DETECTORS.register_module
class GroupSoftmax(TwoStageDetector):
    """group_softmax.py:7-29: identical orchestration; the BAGS logic lives in the bbox head."""
    pass
