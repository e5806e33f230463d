"""bench.py's workloads: the synthetic inputs and step objects of BASELINE.json's configurations (cfg[1] detector
iteration, its Mask R-CNN / Cascade X101 / HTC variants, the GroupSoftmax head step).  Split out of bench.py in round 6
(VERDICT r5 item 8); bench.py re-exports every name, the printed line is unchanged."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from balancedgroupsoftmax_amd import capi  # noqa: E402,F401
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402,F401
from balancedgroupsoftmax_amd import gs_tables  # noqa: E402,F401


NUM_CLASSES = 1231


HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling 6290


def make_inputs(n, seed, dev):
    rs = np.random.RandomState(seed)
    counts = gs_tables.synthetic_instance_counts(NUM_CLASSES, seed=0)
    l2b, ps, _ = gs_tables.build_group_tables(counts)
    W = int(ps[:, 1].sum())
    labels = np.zeros(n, dtype=np.int64)
    nfg = n // 4
    labels[:nfg] = rs.randint(1, NUM_CLASSES, size=nfg)     # positives first (bbox_target_single)
    d = dict(
        logits=torch.from_numpy(rs.standard_normal((n, W)).astype(np.float32)).to(dev),
        bbox_pred=torch.from_numpy(rs.standard_normal((n, 4 * NUM_CLASSES)).astype(np.float32)).to(dev),
        labels=torch.from_numpy(labels).to(dev),
        bbox_targets=torch.from_numpy(rs.standard_normal((n, 4)).astype(np.float32)).to(dev),
        bbox_weights=torch.from_numpy(np.repeat((labels > 0)[:, None], 4, 1).astype(np.float32)).to(dev),
        l2b=torch.from_numpy(l2b).to(dev), ps=torch.from_numpy(ps).to(dev))
    d['l2b_np'], d['ps_np'], d['W'] = l2b, ps, W
    return d


class GsHeadStep(object):
    """One step of the BAGS RoI-head loss: everything the reference's
    GSBBoxHeadWith0.loss() + backward() does for a 1024-RoI batch (selectp=1: the box branch
    contributes its loss value only; cls_score gets its full gradient) — ``bgs_gs_head_step``:
    main kernel (label remap, "others" draw, per-bin losses, gradient, box branch) + reduce (the
    6 loss terms, their sum, the draw counter), then the autograd edge (one scaling launch)."""

    def __init__(self, inp, unit_root=True):
        self.inp = inp
        self.logits = inp['logits'].clone().requires_grad_(True)
        # device-side draw counter, advanced by the reduce kernel: a new sample every step, also under graph replay
        self.draw = torch.zeros(1, dtype=torch.int64, device=inp['logits'].device)
        # root gradient: a persistent tensor, not a fill per step.  unit_root: the library's constant
        # (functional.unit_gradient) — the head's backward recognises it and launches nothing; otherwise some
        # other ones tensor — the scaling launch runs and finds out on the device that every factor is 1
        dev = inp['logits'].device
        self.one = BF.unit_gradient(dev) if unit_root else torch.ones(1, dtype=torch.float32, device=dev)

    def __call__(self):
        i = self.inp
        self.logits.grad = None
        _terms, total, _avg = BF.gs_head_step(self.logits, i['labels'], i['l2b'], i['ps_np'], 8.0, 12345,
                                              draw_counter=self.draw, bbox_pred=i['bbox_pred'],
                                              bbox_targets=i['bbox_targets'],
                                              bbox_weights=i['bbox_weights'],
                                              num_reg_classes=NUM_CLASSES, beta=1.0, box_loss_weight=1.0)
        total.backward(self.one)
        return total


# ---------------------------------------------------------------------------------------------
# detector workload: BASELINE.json configs[1]
# ---------------------------------------------------------------------------------------------
def detector_cfg(table_dir, mask=False, cascade=False, htc=False):
    """gs_faster_rcnn_r50_fpn_1x_lvis_with0_bg8 (reference configs/bags/...), with the three
    absent data files replaced by synthetic tables built with the same rule.  ``mask`` /
    ``cascade`` / ``htc``: the gs_mask_rcnn_r50, gs_cascade_rcnn_x101_64x4d and
    gs_htc_x101_64x4d_fpn_20e_16gpu configs of the same directory."""
    paths = gs_tables.save_group_tables(table_dir, *gs_tables.synthetic_group_tables())
    ce = dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)
    model = dict(
        type='GroupSoftmax', pretrained=None,
        backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3),
                      frozen_stages=1, style='pytorch'),
        neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=5),
        rpn_head=dict(type='RPNHead', in_channels=256, feat_channels=256, anchor_scales=[8],
                      anchor_ratios=[0.5, 1.0, 2.0], anchor_strides=[4, 8, 16, 32, 64],
                      target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0],
                      loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                      loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0)),
        bbox_roi_extractor=dict(type='SingleRoIExtractor',
                                roi_layer=dict(type='RoIAlign', out_size=7, sample_num=2),
                                out_channels=256, featmap_strides=[4, 8, 16, 32]),
        bbox_head=dict(type='GSBBoxHeadWith0', num_fcs=2, in_channels=256, fc_out_channels=1024,
                       gs_config=dict(label2binlabel=paths['label2binlabel'],
                                      pred_slice=paths['pred_slice'], fg_split=paths['fg_split'],
                                      others_sample_ratio=8.0, loss_bg=dict(ce), num_bins=5,
                                      loss_bin=dict(ce)),
                       roi_feat_size=7, num_classes=NUM_CLASSES, target_means=[0., 0., 0., 0.],
                       target_stds=[0.1, 0.1, 0.2, 0.2], reg_class_agnostic=False,
                       loss_cls=dict(ce),
                       loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0)))
    train_cfg = dict(
        rpn=dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.7, neg_iou_thr=0.3,
                               min_pos_iou=0.3, ignore_iof_thr=-1),
                 sampler=dict(type='RandomSampler', num=256, pos_fraction=0.5, neg_pos_ub=-1,
                              add_gt_as_proposals=False),
                 allowed_border=0, pos_weight=-1, debug=False),
        rpn_proposal=dict(nms_across_levels=False, nms_pre=2000, nms_post=2000, max_num=2000,
                          nms_thr=0.7, min_bbox_size=0),
        rcnn=dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.5,
                                min_pos_iou=0.5, ignore_iof_thr=-1),
                  sampler=dict(type='RandomSampler', num=512, pos_fraction=0.25, neg_pos_ub=-1,
                               add_gt_as_proposals=True),
                  pos_weight=-1, debug=False))
    return _cfg_variant(model, train_cfg, mask, cascade, htc)


def _cfg_variant(model_cfg, train_cfg, mask, cascade, htc):
    mask_ext = dict(type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', out_size=14, sample_num=2),
                    out_channels=256, featmap_strides=[4, 8, 16, 32])
    mask_head = dict(type='FCNMaskHead', num_convs=4, in_channels=256, conv_out_channels=256,
                     num_classes=NUM_CLASSES,
                     loss_mask=dict(type='CrossEntropyLoss', use_mask=True, loss_weight=1.0))
    if mask:        # cfg[3] = configs/bags/gs_mask_rcnn_r50_fpn_1x_lvis.py
        model_cfg['type'] = 'MaskRCNN'
        model_cfg['mask_roi_extractor'] = mask_ext
        model_cfg['mask_head'] = mask_head
        train_cfg['rcnn']['mask_size'] = 28
    if cascade or htc:     # cfg[4] = configs/bags/gs_cascade_rcnn_x101_64x4d_fpn_1x_lvis.py (fp32 here)
        model_cfg['type'] = 'CascadeRCNN'
        model_cfg['num_stages'] = 3
        model_cfg['backbone'] = dict(type='ResNeXt', depth=101, groups=64, base_width=4,
                                     num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                                     style='pytorch')
        base = model_cfg['bbox_head']
        heads = []
        for stds in ([0.1, 0.1, 0.2, 0.2], [0.05, 0.05, 0.1, 0.1], [0.033, 0.033, 0.067, 0.067]):
            h = dict(base, reg_class_agnostic=True, target_stds=stds)
            h['gs_config'] = dict(base['gs_config'])
            heads.append(h)
        model_cfg['bbox_head'] = heads
        rc = train_cfg['rcnn']
        train_cfg['rcnn'] = [dict(rc, assigner=dict(rc['assigner'], pos_iou_thr=t, neg_iou_thr=t,
                                                    min_pos_iou=t)) for t in (0.5, 0.6, 0.7)]
        train_cfg['stage_loss_weights'] = [1, 0.5, 0.25]
    if htc:         # configs/bags/gs_htc_x101_64x4d_fpn_20e_16gpu_lvis.py (fp32 here)
        model_cfg['type'] = 'HybridTaskCascade'
        model_cfg['interleaved'] = True
        model_cfg['mask_info_flow'] = True
        model_cfg['mask_roi_extractor'] = mask_ext
        model_cfg['mask_head'] = dict(mask_head, type='HTCMaskHead')
        model_cfg['semantic_roi_extractor'] = dict(
            type='SingleRoIExtractor', roi_layer=dict(type='RoIAlign', out_size=14, sample_num=2),
            out_channels=256, featmap_strides=[8])
        model_cfg['semantic_head'] = dict(
            type='FusedSemanticHead', num_ins=5, fusion_level=1, num_convs=4, in_channels=256,
            conv_out_channels=256, num_classes=183, ignore_label=255, loss_weight=0.2)
        for rc in train_cfg['rcnn']:
            rc['mask_size'] = 28
    return model_cfg, train_cfg


class DetectorStep(object):
    """One training iteration of cfg[1] as shipped (selectp=1: full forward, backward through
    fc_cls, gradient all-reduce, clip, SGD) on synthetic 800x1344 inputs, 512 RoIs/img."""

    def __init__(self, dev, rank, world, imgs, selectp=1, mask=False, cascade=False, htc=False,
                 conv_math='bf16x6'):
        import tempfile
        import balancedgroupsoftmax_amd as bgs
        from balancedgroupsoftmax_amd import train
        from balancedgroupsoftmax_amd.config import to_config_dict
        self.train = train
        torch.manual_seed(0)                      # identical weights on every rank
        tmp = tempfile.mkdtemp(prefix='bgs_tables_')
        model_cfg, train_cfg = detector_cfg(tmp, mask=mask, cascade=cascade, htc=htc)
        self.mask = mask = mask or htc
        self.model = bgs.build_detector(to_config_dict(model_cfg),
                                        train_cfg=to_config_dict(train_cfg), test_cfg=None).to(dev)
        self.selectp = selectp
        self.params = train.select_training_param(self.model, selectp)
        self.model.train()
        opt = train.build_optimizer(self.params, dict(type='SGD', lr=0.01, momentum=0.9,
                                                      weight_decay=0.0001))
        if conv_math == 'bf16':     # mmdet/core/fp16/hooks.py: wrap_fp16_model + Fp16OptimizerHook
            train.wrap_fp16_model(self.model, 'bf16')
            self.step_fn = train.Fp16OptimizerStep(self.params, opt, dict(max_norm=35, norm_type=2),
                                                   world_size=world, loss_scale=512.0)
        else:
            self.step_fn = train.DistOptimizerStep(self.params, opt, dict(max_norm=35, norm_type=2),
                                                   world_size=world)
        self.loss_scale = getattr(self.step_fn, 'loss_scale', 1.0)
        g = torch.Generator().manual_seed(1000 + rank)          # different data per rank
        H, W = 800, 1344                                        # 1333 padded to /32 (Pad(size_divisor=32))
        self.img = torch.randn(imgs, 3, H, W, generator=g).to(dev)
        self.metas = [dict(img_shape=(800, 1333, 3), pad_shape=(H, W, 3), ori_shape=(800, 1333, 3),
                           scale_factor=1.0, flip=False) for _ in range(imgs)]
        self.gt_bboxes, self.gt_labels = [], []
        for _ in range(imgs):                                   # G = 20 boxes / image
            wh = torch.exp(torch.rand(20, 2, generator=g) * (np.log(400) - np.log(16)) + np.log(16))
            xy = torch.rand(20, 2, generator=g) * (torch.tensor([1333., 800.]) - wh).clamp(min=1)
            self.gt_bboxes.append(torch.cat([xy, (xy + wh)], 1).to(dev))
            self.gt_labels.append(torch.randint(1, NUM_CLASSES, (20,), generator=g).to(dev))
        self.gt_masks = None
        if mask:          # an axis-aligned ellipse inside every GT box (SURVEY.md §8d cfg 4)
            self.gt_masks = []
            yy = torch.arange(H, device=dev).view(1, H, 1).float()
            xx = torch.arange(W, device=dev).view(1, 1, W).float()
            for b in self.gt_bboxes:
                cx, cy = ((b[:, 0] + b[:, 2]) / 2).view(-1, 1, 1), ((b[:, 1] + b[:, 3]) / 2).view(-1, 1, 1)
                rx = ((b[:, 2] - b[:, 0]) / 2).clamp(min=1).view(-1, 1, 1)
                ry = ((b[:, 3] - b[:, 1]) / 2).clamp(min=1).view(-1, 1, 1)
                self.gt_masks.append(((((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2) <= 1.0)
                                     .to(torch.uint8).contiguous())
        self.extra = {}
        if htc:           # [N, 1, H/8, W/8] stuff-class map with 20 % ignored pixels
            seg = torch.randint(0, 183, (imgs, 1, H // 8, W // 8), generator=g)
            seg[torch.rand(seg.shape, generator=g) < 0.2] = 255
            self.extra['gt_semantic_seg'] = seg.to(dev)
        self.last = None

    def compute(self, feats=None):
        """forward + losses + backward: free of host synchronisation -> hipGraph-capturable."""
        kw = dict(self.extra)
        if feats is not None:
            kw['feats'] = feats
        losses = self.model(self.img, self.metas, return_loss=True, gt_bboxes=self.gt_bboxes,
                            gt_labels=self.gt_labels, gt_masks=self.gt_masks, **kw)
        loss, log_vars = self.train.parse_losses(losses)
        # grads set to None: backward then STORES each gradient (AccumulateGrad takes the tensor)
        # instead of a zero fill + an add per parameter — 2 x 160 launches of the selectp=0 step.
        # Same values as the reference's zero_grad() + accumulation into zeros.
        self.step_fn.optimizer.zero_grad(set_to_none=os.environ.get('BGS_ZERO_GRAD_FILL') != '1')
        # root gradient = the library's cached unit gradient (train.backward_unit): the same ones, no fill launch, and
        # the fused GroupSoftmax head receives it by identity through parse_losses -> its backward launches nothing
        self.train.backward_unit(loss * self.loss_scale if self.loss_scale != 1.0 else loss)
        # detached copies only: holding the loss would keep the autograd graph (and its
        # AccumulateGrad nodes) alive across iterations
        self.last = {k: v.detach() for k, v in log_vars.items()}

    def apply(self):
        """gradient all-reduce (RCCL), clip, SGD — launched eagerly after the captured part
        (a handful of launches; keeps the collective out of the graph)."""
        self.step_fn.exchange_and_update()

    def __call__(self):
        self.compute()
        self.apply()

    # -- two-stage software pipeline (train.TrunkPipeline): frozen trunk only ---------------------------------------
    def can_pipeline(self):
        return self.selectp in (1, 3) and self.model.trunk_is_frozen()

    def pipelined(self, depth=None):
        """-> a step function in which the frozen trunk of the batches AHEAD is launched — in ``depth - 1`` pieces, each
        on its own stream (train.TrunkPipeline) — before this batch's heads / losses / backward / exchange / optimizer
        step: every call still issues one pass of every piece of the trunk and one head pass; the first calls consume
        the features launched here (untimed prologue)."""
        if depth is None:
            depth = int(os.environ.get('BGS_BENCH_PIPELINE_DEPTH', '4'))
        pipe = self.train.TrunkPipeline(self.model, depth=depth)
        for _ in range(pipe.depth - 1):
            pipe.push(self.img)

        def step():
            feats = pipe.take()
            pipe.push(self.img)                # (the synthetic loader hands out the same batch: the work is a later batch's)
            self.compute(feats)
            self.apply()

        step.drain = pipe.drain
        step.depth = pipe.depth
        return step


CONV_MATH_NOTE = {
    'bf16x6': 'fp32 tensors in HBM, fp32 accumulate, fp32 results; each fp32 product is formed on the '
              'bf16 matrix cores from the exact three-way bf16 split of both operands (six partial '
              'products, dropped terms <= 2^-25 |ab|): error vs fp64 not above the fp32 MFMA kernel\'s '
              '(tests/test_gpu_det_ops.py::test_bfx_error_not_above_f32_mfma)',
    'f32': 'v_mfma_f32_32x32x2_f32: fp32 in / fp32 accumulate, bit-exact fma chain',
    'bf16': 'REDUCED PRECISION (cfg[4] only): conv / linear operands rounded to bf16 for '
            'v_mfma_f32_32x32x16_bf16, fp32 accumulate, fp32 master weights, fp32 '
            'GroupSoftmax / box / mask losses (force_fp32), loss scale 512 (Fp16OptimizerHook); '
            'the activations of the frozen trunk (ResNe(X)t layer1-4) are STORED in bf16 '
            '(csrc/conv_bf16s.hip; the pyramid, RoI features and heads stay fp32) unless '
            'BGS_BF16_STORAGE=0, which keeps fp32 tensors and rounds inside the kernels',
}


def try_graph(step):
    """Capture one step into a hipGraph (the loop is launch-bound: ~10 short kernels)."""
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        from balancedgroupsoftmax_amd import functional as BF
        BF.reset_workspaces()      # scratch buffers of an earlier capture belong to ITS pool
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        torch.cuda.synchronize()
        return g
    except Exception as e:  # pragma: no cover
        import traceback
        traceback.print_exc()
        sys.stderr.write('hipGraph capture failed (%s); timing eager launches\n' % (e,))
        torch.cuda.synchronize()
        return None
