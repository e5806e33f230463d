#!/usr/bin/env python
"""Benchmark of the MI355X-native Balanced Group Softmax hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload gs_head]

Contract: W untimed warm-up steps, then exactly K timed steps bracketed by barrier +
``torch.cuda.synchronize()``; the MAX over ranks is taken and rank 0 prints ONE JSON line.

Workload ``gs_head`` (this round): the RoI-head loss of BASELINE.json config[1]
(Faster R-CNN R50-FPN + BAGS, 2 img/GPU x 512 RoI = 1024 RoIs x 1236 logits, 25 % fg):
one step = ``_remap_labels`` (device sampling) + fused GroupSoftmax loss forward+backward
(+ the reduce / scale kernels) + the box-regression loss, inputs resident in HBM.
The JSON line carries ``roofline`` for the dominant kernel (the fused streaming kernel,
HIP-event timed on its own stream) and ``cpu_baseline`` = the torch-CPU port of the
reference's loss()+backward() timed on this host (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from balancedgroupsoftmax_amd import capi  # noqa: E402
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402
from balancedgroupsoftmax_amd import gs_tables  # noqa: E402
import bench_dist  # noqa: E402
from bench_dist import barrier, timed_loop  # noqa: E402,F401  (the timed-region contract lives there: one copy for N = 1 and N > 1)

NUM_CLASSES = 1231
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling 6290


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--workload', default='detector', choices=['detector', 'gs_head'])
    ap.add_argument('--imgs', type=int, default=2, help='images per GPU per step (cfg: imgs_per_gpu=2)')
    ap.add_argument('--rois', type=int, default=1024, help='RoIs per GPU per step (2 img x 512)')
    ap.add_argument('--cascade', action='store_true',
                    help='cfg[4]: Cascade R-CNN X101-64x4d-FPN + BAGS, 3 stages (fp32)')
    ap.add_argument('--htc', action='store_true',
                    help='Hybrid Task Cascade X101-64x4d-FPN + BAGS (gs_htc_x101_64x4d_fpn_20e_16gpu_'
                         'lvis): cascade + 3 HTCMaskHeads + semantic branch (fp32)')
    ap.add_argument('--selectp', type=int, default=1, choices=[0, 1, 3],
                    help='1 (as shipped): train bbox_head.fc_cls only; 0: train everything '
                         '(tools/train.py:49-57)')
    ap.add_argument('--mask', action='store_true',
                    help='cfg[3]: add the Mask R-CNN branch (gs_mask_rcnn_r50_fpn_1x_lvis)')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the secondary measurements (selectp=0, Mask R-CNN) of the N=1 run')
    ap.add_argument('--child', action='store_true',
                    help='(internal) measurement sub-process of the single-GPU run')
    ap.add_argument('--no-roofline', action='store_true',
                    help='skip the per-kernel roofline timings (used by the extras sub-runs)')
    ap.add_argument('--no-graph', action='store_true', help='time eager launches, not hipGraph replay')
    ap.add_argument('--launch', choices=('auto', 'graph', 'eager', 'pipelined'), default='auto',
                    help='one GPU: auto = calibrate hipGraph replay, eager launches and the trunk pipeline untimed and '
                         'time the fastest (default) | graph | eager | pipelined (train.TrunkPipeline without '
                         'calibration, depth BGS_BENCH_PIPELINE_DEPTH or 5: for traces)')
    ap.add_argument('--dist-graph', action='store_true',
                    help='N > 1 over RCCL: capture the WHOLE step, gradient all-reduce included, into '
                         'one hipGraph per rank (default for N > 1: eager launches)')
    ap.add_argument('--conv-math', default=None, choices=['bf16x6', 'f32', 'bf16'],
                    help="conv / linear arithmetic: 'bf16x6' (default; bf16 MFMA on exactly split fp32 "
                         "operands, fp32-faithful), 'f32' (v_mfma_f32_32x32x2_f32) or 'bf16' (operands "
                         "rounded to bf16, fp32 accumulate: the reduced-precision mode of cfg[4])")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    a = ap.parse_args()
    if a.conv_math is not None:
        BF.set_conv_math(a.conv_math)
    a.conv_math = BF.conv_math()
    if a.steps is None:
        a.steps = 30 if a.workload == 'detector' else 200
    if a.warmup is None:
        a.warmup = 5 if a.workload == 'detector' else 20
    return a


def init_dist(args):
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # (test hooks: BGS_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and BGS_DIST_BACKEND=gloo
    #  swaps the collective backend, so the multi-process code path can be exercised on a
    #  single-GPU box; the real runs use one GPU per rank over RCCL)
    if os.environ.get('BGS_BENCH_ONE_DEVICE'):
        local = 0
        # two PROCESSES time-sharing one GPU, each with side streams: the device scheduler thrashes between the
        # processes' queues at every event edge (185 ms per step instead of 15: profiles/r8q_dist2_onegpu*.json).
        # One rank per GPU — the real configuration — has no second process on the device; the hook turns the
        # side-stream forks off for itself.
        os.environ.setdefault('BGS_LEVEL_FORK', '0')
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('BGS_DIST_BACKEND', 'nccl')                   # nccl = RCCL on ROCm
        import datetime
        # a collective that one rank never joins raises after this long instead of holding the job for torch's default
        # 10 minutes (the watchdogs of bench_dist.run print the line earlier still)
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=int(os.environ.get('BGS_BENCH_COLL_TIMEOUT', '150'))))
    elif os.environ.get('BGS_BENCH_SELF_GROUP'):
        # test hook: a 1-rank RCCL group whose all-reduce really runs, so that the N > 1 launch
        # policies (eager exchange, --dist-graph) can be exercised on a single-GPU box
        import socket
        import torch.distributed as dist
        from balancedgroupsoftmax_amd import train as BT
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        dist.init_process_group(backend='nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0,
                                world_size=1)
        BT.exchange_at_world_size_one(True)
    return rank, local, world


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


def conv_roofline(dev, math, iters=20, wide=0):
    """Dominant kernel of the detector step: the 3x3 halo convolution.  Timed on the largest single
    layer (FPN output conv on P2: 2x200x336 pixels, 3x3, 256->256 = 158.5 algorithmic GFLOP) with
    HIP events on the launch stream.  `achieved` = ALGORITHMIC flops / time.  Peaks
    (MI355X_MICROARCH.md): fp32 matrix 157.3 TFLOP/s; bf16 matrix 2500 TFLOP/s dense — the bf16x6
    kernel spends SIX bf16 MFMA passes per algorithmic fp32 multiply-add, so its matrix-pipe
    ceiling in algorithmic flops is 2500 / 6 = 416.7 TFLOP/s (frac = matrix-pipe busy fraction)."""
    prev = BF.set_conv_math(math)
    prev_wide = BF.set_halo_wide(1) if wide else None
    used = {}
    try:
        x = torch.randn(2, 200, 336, 256, device=dev)
        w = torch.randn(256, 3, 3, 256, device=dev) * 0.02
        b = torch.randn(256, device=dev)
        out = torch.empty(2, 200, 336, 256, device=dev)
        # warm-up: the first ~10 launches after an idle period run 8-10 % slower (clock ramp: 0.81 vs
        # 0.74 ms on the P2 layer); inside the step the kernel runs warm (rocprofv3 average 0.743 ms,
        # profiles/r3k_detector_prof_summary.md), and that is the state a roofline should describe
        for _ in range(25):
            BF.conv2d_nhwc(x, w, b, pad=1, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            BF.conv2d_nhwc(x, w, b, pad=1, out=out)
        e1.record()
        torch.cuda.synchronize()
        used = BF.conv_bfx_last_launch() if math != 'f32' else {}
    finally:
        BF.set_conv_math(prev)
        if wide:
            BF.set_halo_wide(prev_wide)
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * 2 * 200 * 336 * 256 * 256 * 9
    tf = flops / (ms * 1e-3) / 1e12
    if math == 'bf16x6':
        kname = 'conv3x3_halo_bfx4_kernel<2>'
        kdesc = kname + (' (halo-resident A operand split to 3 bf16 planes in LDS, filter slices by '
                         'LDS-DMA, v_mfma_f32_32x32x16_bf16 x 6)')
        peak, passes = 2500.0 / 6.0, 6
        if wide and used.get('halo_wide_units'):
            # the two-launch schedule the trunk pipeline switches on (bgs_conv3x3_halo_bfx_wide(1)): whole rounds of
            # 16 x 16-pixel units (two workgroups per CU) + the left-over rows on the 8 x 16-pixel kernel
            kname = 'conv3x3_halo_bfx7_kernel<3>+conv3x3_halo_bfx4_kernel<2>'
            kdesc = ('conv3x3_halo_bfx7_kernel<3> on %d units of 16 x 16 pixels x 128 channels (two workgroups per CU, '
                     'whole rounds of 512) + conv3x3_halo_bfx4_kernel<2> on the %d left-over 8 x 16-pixel units: two '
                     'launches per layer, bit-identical to the one-launch form (tests/test_gpu_det_ops.py); '
                     'ms_per_launch is the PAIR' % (used['halo_wide_units'], used['halo_tail_units']))
    elif math == 'bf16':
        kname = 'conv3x3_halo_bfx3_kernel<2,1>'
        kdesc = kname + ' (operands rounded to bf16, v_mfma_f32_32x32x16_bf16 x 1)'
        peak, passes = 2500.0, 1
    else:
        halo = BF._use_halo_kernel(2 * 200 * 336, 256)
        kname = 'conv3x3_halo_f32_kernel' if halo else 'conv_igemm_f32_kernel<2,2,16,1>'
        kdesc = kname + ' (v_mfma_f32_32x32x2_f32)'
        peak, passes = 157.3, 1
    traffic = src = None
    ent = {}
    try:      # HBM-side bytes per launch from the committed PMC passes (separate rocprofv3 --pmc runs)
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles',
                               'pmc_traffic.json')) as f:
            ent = json.load(f)[kname]['fpn_p2_out_2x200x336_3x3_256_256']
        traffic, src = ent['traffic_bytes_per_launch'], ent['source']
    except Exception:
        pass
    r = dict(bound='mfma', achieved=round(tf, 2), peak=round(peak, 1), unit='TFLOP/s',
             frac=round(tf / peak, 4), traffic=traffic,
             traffic_source=('committed PMC measurement of this kernel on this layer, not collected '
                             'in this run: %s' % src) if src else None,
             algorithmic_bytes=277610496, kernel=kdesc, ms_per_launch=round(ms, 4),
             flops_per_launch=flops,
             layer='FPN output conv P2: N=2, 200x336, 3x3, 256->256 (M=134400, K=2304)',
             timing='hipEvent over %d back-to-back launches after 25 warm-up launches' % iters)
    if ent.get('matrix_pipe_busy'):
        # committed PMC pass of the same kernel on the same layer (not collected in this run): the
        # fraction of cycles the matrix pipe was busy, and the clock the chip sustained under it —
        # `frac` is priced against the 2.4 GHz data-sheet peak
        r.update(matrix_pipe_busy_pmc=ent['matrix_pipe_busy'],
                 effective_clock_ghz_pmc=ent['effective_clock_ghz'],
                 pmc_note='frac x 2.4 / %.1f ~ matrix_pipe_busy_pmc (%s)'
                          % (ent['effective_clock_ghz'], ent.get('counters', '')))
    if passes > 1:
        r.update(mfma_dtype='bf16', mfma_passes_per_flop=passes,
                 matrix_pipe_tflops=round(tf * passes, 1), peak_bf16_dense=2500.0,
                 peak_note='algorithmic-flop ceiling of the bf16x6 kernel = 2500 (bf16 dense MFMA) / 6 '
                           'passes; against the fp32-MFMA peak (157.3) the same launch is %.2fx'
                           % (tf / 157.3))
    return r


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


def _event_time_us(launch, iters, warm=20, settle=0):
    """HIP-event time per launch over `iters` back-to-back launches.  ``settle`` > 0: the batch is repeated (at most
    `settle` times) until two consecutive batches agree within 1 % and the last one is returned — a streaming
    kernel's first ~20 ms after an idle or compute-bound phase run 10 - 15 % slower (rowwave kernel at N = 65,536
    on fresh inputs: 141, 126, 123, 121, 121 us for five consecutive batches of 50 launches; the memory-side clocks
    ramp), and the roofline is a steady-state figure."""
    for _ in range(warm):
        launch()
    torch.cuda.synchronize()
    prev = None
    for _ in range(max(1, settle)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            launch()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        if prev is not None and abs(us - prev) <= 0.01 * prev:
            break
        prev = us
    return us


def _pmc_traffic(kernel, n):
    """HBM bytes per launch from the PMC counters: collected in separate rocprofv3 passes
    (tools/pmc_traffic.sh), corrected as the microarch guide prescribes, and committed under
    profiles/ — bench.py itself cannot run the profiler."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
            ent = json.load(f)[kernel][str(n)]
        return ent['traffic_bytes_per_launch'], ent.get('source')
    except Exception:
        return None, None


def kernel_roofline(inp, n, iters=300, kernel='fused'):
    """HIP-event timing of ONE GroupSoftmax kernel alone, back to back on the current stream.
    ``kernel='fused'``: the kernel the detector step and the gs_head step actually launch for
    N <= 4096 (``gs_head_multi_kernel`` for N <= 2048 — 4 or 2 rows per workgroup behind one shared
    prologue —, ``gs_head_fused_kernel`` beyond; ``bgs_gs_head_variant_used``: main launch of
    bgs_gs_head_step with loss_out = NULL — label remap + "others" draw + loss + gradient + box branch).  ``kernel='rowwave'``: the plain loss
    kernel (main launch of bgs_gs_loss_fwd_bwd; the path for N > 4096 / reweighted heads).
    Algorithmic bytes per RoI: W*4 read + W*4 written + 8 (label) + B*4."""
    lib = capi.load()
    W, B = inp['W'], inp['ps_np'].shape[0]
    dev = inp['logits'].device
    ps_keep, ps_ptr = capi.host_i64(inp['ps_np'])
    dl = torch.empty_like(inp['logits'])
    ws = torch.empty(lib.bgs_gs_loss_workspace_bytes(n, B), dtype=torch.uint8, device=dev)
    st = capi.current_stream(dev)
    if kernel == 'fused':
        avg = torch.empty(B, dtype=torch.float32, device=dev)
        cbits = BF.gs_class_bin_mask(inp['l2b'])
        variant = lib.bgs_gs_head_variant_used(n)
        kname = {0: 'gs_head_fused_kernel<4,true,true,0>', 1: 'gs_head_fused_kernel<4,true,true,1>',
                 2: 'gs_head_multi_kernel<4,true,true,2>', 3: 'gs_head_multi_kernel<4,true,true,4>',
                 4: 'gs_head_multi_kernel<4,true,true,2,direct>',
                 5: 'gs_head_multi_kernel<4,true,true,4,direct>'}.get(variant, 'gs_head kernel variant %d' % variant)

        def launch():
            rc = lib.bgs_gs_head_step(capi.ptr(inp['logits']), capi.ptr(inp['labels']), capi.ptr(inp['l2b']),
                                      capi.ptr(cbits), None, ps_ptr, None, n, NUM_CLASSES, B, W, 8.0,
                                      12345, None, capi.ptr(inp['bbox_pred']),
                                      capi.ptr(inp['bbox_targets']), capi.ptr(inp['bbox_weights']),
                                      NUM_CLASSES, 1.0, 1.0, None, None, capi.ptr(dl), None,
                                      capi.ptr(avg), None, None, capi.ptr(ws), st)
            capi.check('bgs_gs_head_step', rc)
    else:
        bl, w, avg = BF.gs_prepare(inp['labels'], inp['l2b'], 8.0, seed=1)
        kname = 'gs_loss_rowwave_kernel<4,true>'

        def launch():
            rc = lib.bgs_gs_loss_fwd_bwd(capi.ptr(inp['logits']), capi.ptr(bl), ps_ptr, capi.ptr(w),
                                         capi.ptr(avg), n, B, W, None, capi.ptr(dl), capi.ptr(ws), st)
            capi.check('bgs_gs_loss_fwd_bwd', rc)

    us = _event_time_us(launch, iters, settle=8 if n >= 16384 else 0)
    bytes_per_roi = W * 4 + W * 4 + 8 + B * 4
    achieved = bytes_per_roi * n / (us * 1e-6) / 1e9
    traffic, src = _pmc_traffic(kname, n)
    return dict(bound='hbm', achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                traffic_source=('committed PMC measurement, not collected in this run: %s' % src) if src else None,
                kernel=kname, us_per_launch=round(us, 3),
                launched_by=('the detector step and the gs_head step (N <= 4096)' if kernel == 'fused'
                             else 'heads with N > 4096 rows or per-class reweighting (after gs_prepare)'),
                algorithmic_bytes_per_roi=bytes_per_roi, rois_per_launch=n,
                timing='hipEvent over %d back-to-back launches (includes the ~1.5 us '
                       'inter-kernel boundary)%s' % (iters, '; batches repeated until two agree within 1 % '
                                                     '(steady state, see _event_time_us)' if n >= 16384 else ''))


def capture_head_inputs(dev, conv_math='bf16x6'):
    """One eager cfg[1] iteration with spies on the RoI extractor and the RPN assigner: the REAL operands of the
    HBM-bound helper kernels (sampled RoIs + the NHWC pyramid; anchors, inside flags, gts) for their standalone
    rooflines and for tools/kernel_once.py (the PMC passes)."""
    step = DetectorStep(dev, 0, 1, 2, 1, conv_math=conv_math)
    cap = {}
    ext = step.model.bbox_roi_extractor
    orig = ext.forward

    def spy(feats, rois, *a, **k):
        cap['feats'] = [f.detach() for f in feats[:ext.num_inputs]]
        cap['rois'] = rois.detach().clone()
        cap['strides'] = list(ext.featmap_strides)
        cap['out_size'], cap['sample_num'], cap['finest_scale'] = ext.out_size, ext.sample_num, ext.finest_scale
        return orig(feats, rois, *a, **k)

    ext.forward = spy
    orig_assign = BF.iou_assign

    def spy_assign(boxes, gt_cat, offs, pos, neg, minpos=0.0, valid=None, shared_boxes=False, **k):
        if shared_boxes and 'anchors' not in cap:
            cap['anchors'], cap['gt_cat'], cap['gt_offs'] = boxes, gt_cat.clone(), list(offs)
            cap['assign_thr'] = (pos, neg, minpos)
            cap['inside'] = valid
        return orig_assign(boxes, gt_cat, offs, pos, neg, minpos, valid=valid, shared_boxes=shared_boxes, **k)

    BF.iou_assign = spy_assign
    try:
        os.environ['BGS_RPN_LOSS_FORK'] = '0'
        step()
        torch.cuda.synchronize()
    finally:
        BF.iou_assign = orig_assign
        ext.forward = orig
        os.environ.pop('BGS_RPN_LOSS_FORK', None)
    del step
    return cap


def roi_footprint_bytes(rois, shapes, strides, C, out_size=7, sample_num=2, finest_scale=56.0):
    """the unique input footprint of every RoI, exactly, from the RoIs themselves — the set of
    feature-map pixels its out x out x sample_num^2 bilinear sample points touch (level map single_level.py:69-72,
    sample geometry and the clamping / out-of-bounds rules of roi_align_kernel.cu:16-61,86-118), x C x 4 bytes.
    Returns (sum over RoIs of per-RoI footprints, bytes of the UNION over all RoIs, per-level RoI counts)."""
    r = rois.detach().cpu().numpy().astype(np.float32)
    L = len(strides)
    f32 = np.float32
    scale = np.sqrt((r[:, 3] - r[:, 1] + f32(1)) * (r[:, 4] - r[:, 2] + f32(1)))
    lvl = np.clip(np.floor(np.log2(scale / f32(finest_scale) + f32(1e-6))), 0, L - 1).astype(np.int64)
    per_roi = 0
    union = [dict() for _ in range(L)]
    g = (np.arange(out_size * sample_num, dtype=np.float32) + f32(0.5)) / f32(sample_num)   # sample offsets in bins
    for k in range(r.shape[0]):
        l = int(lvl[k])
        n = int(r[k, 0])
        H, W = shapes[l]
        ss = f32(1.0 / strides[l])
        x1, y1 = r[k, 1] * ss, r[k, 2] * ss
        rw = max((r[k, 3] + f32(1)) * ss - x1, f32(0))
        rh = max((r[k, 4] + f32(1)) * ss - y1, f32(0))
        ys = y1 + g * (rh / f32(out_size))
        xs = x1 + g * (rw / f32(out_size))

        def axis(v, S):
            ok = (v >= -1.0) & (v <= S)
            v = np.maximum(v, 0)
            lo = np.minimum(v.astype(np.int64), S - 1)
            hi = np.minimum(lo + 1, S - 1)
            return ok, lo, hi

        oky, ylo, yhi = axis(ys, H)
        okx, xlo, xhi = axis(xs, W)
        rows = np.unique(np.concatenate([ylo[oky], yhi[oky]]))
        cols = np.unique(np.concatenate([xlo[okx], xhi[okx]]))
        per_roi += rows.size * cols.size
        u = union[l].setdefault(n, np.zeros((H, W), dtype=bool))
        if rows.size and cols.size:
            u[np.ix_(rows, cols)] = True
    union_px = sum(int(m.sum()) for d in union for m in d.values())
    counts = [int((lvl == l).sum()) for l in range(L)]
    return per_roi * C * 4, union_px * C * 4, counts


def hbm_kernel_rooflines(dev, conv_math='bf16x6'):
    """SURVEY.md 8(d)'s other HBM-bound kernels on the operands of a real cfg[1] iteration: RoIAlign forward,
    `_merge_score` (R = 1000 as at test time, and R = 65,536 where the roofline applies), IoU / assignment of the
    RPN's 268,569 anchors x 2 images.  hipEvent time of back-to-back launches; `traffic` from the committed PMC
    passes (tools/pmc_hbm_kernels.sh -> profiles/pmc_traffic.json)."""
    res = {}
    cap = capture_head_inputs(dev, conv_math)
    feats, rois = cap['feats'], cap['rois']
    K, C = int(rois.shape[0]), int(feats[0].shape[3])
    shapes = [(int(f.shape[1]), int(f.shape[2])) for f in feats]
    out_bytes = K * cap['out_size'] ** 2 * C * 4
    pyramid = sum(int(f.numel()) * 4 for f in feats)
    fp_sum, fp_union, counts = roi_footprint_bytes(rois, shapes, cap['strides'], C, cap['out_size'],
                                                   cap['sample_num'], cap['finest_scale'])
    us = _event_time_us(lambda: BF.roi_align_nhwc(feats, rois, cap['strides'], cap['out_size'], cap['sample_num'],
                                                  cap['finest_scale']), 50, settle=4)
    alg = out_bytes + min(pyramid, fp_sum)
    kname = 'roi_align_fwd_grid_kernel<1,false>'      # (round 5: every distinct pixel of a bin loaded once)
    tr, src = _pmc_traffic(kname, K)
    res['roofline_roi_align'] = dict(
        bound='hbm', achieved=round(alg / (us * 1e-6) / 1e9, 1), peak=HBM_PEAK_GBS, unit='GB/s',
        frac=round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), traffic=tr,
        traffic_source=('committed PMC measurement, not collected in this run: %s' % src) if src else None,
        kernel=kname, us_per_launch=round(us, 2), rois_per_launch=K, rois_per_level=counts,
        algorithmic_bytes=alg, output_bytes=out_bytes, pyramid_bytes=pyramid,
        sum_of_per_roi_footprints=fp_sum, union_of_footprints=fp_union,
        bytes_issued_by_the_taps=K * cap['out_size'] ** 2 * cap['sample_num'] ** 2 * 4 * C * 4,
        frac_with_union_footprint=round((out_bytes + fp_union) / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
        note='SURVEY 8(d): bytes = output + min(pyramid, sum of per-RoI unique footprints), footprints computed '
             'exactly from the sampled RoIs of a real iteration (roi_footprint_bytes); `union_of_footprints` is '
             'what a perfect cache would fetch once')
    del feats, cap['feats']
    # _merge_score: read W*4 + write C*4 per RoI
    tdir = __import__('tempfile').mkdtemp(prefix='bgs_tables_')
    counts_t = gs_tables.synthetic_instance_counts(NUM_CLASSES, seed=0)
    l2b, ps, _ = gs_tables.build_group_tables(counts_t)
    c2c = gs_tables.class_to_column(l2b, ps).to(dev)
    W = int(ps[:, 1].sum())
    for R, iters in ((1000, 200), (65536, 30)):
        z = torch.randn(R, W, device=dev)
        us = _event_time_us(lambda: BF.gs_merge_score(z, ps, c2c, NUM_CLASSES), iters, settle=6 if R > 4096 else 0)
        alg = R * (W * 4 + NUM_CLASSES * 4)
        kname = 'gs_merge_rowwave_kernel'
        tr, src = _pmc_traffic(kname, R)
        res['roofline_merge_score' + ('' if R == 1000 else '_n%d' % R)] = dict(
            bound='hbm', achieved=round(alg / (us * 1e-6) / 1e9, 1), peak=HBM_PEAK_GBS, unit='GB/s',
            frac=round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), traffic=tr,
            traffic_source=('committed PMC measurement, not collected in this run: %s' % src) if src else None,
            kernel=kname, us_per_launch=round(us, 2), rois_per_launch=R,
            algorithmic_bytes_per_roi=W * 4 + NUM_CLASSES * 4,
            note='includes the [R, 1231] output allocation of the wrapper (no launch)')
        del z
    # IoU + MaxIoUAssigner of the RPN: per image A anchors x (16 B box + 1 B inside flag) read, 4 B written
    if 'anchors' in cap:
        A = int(cap['anchors'].shape[0])
        N = len(cap['gt_offs']) - 1
        pos, neg, minpos = cap['assign_thr']
        us = _event_time_us(lambda: BF.iou_assign(cap['anchors'], cap['gt_cat'], cap['gt_offs'], pos, neg, minpos,
                                                  valid=cap['inside'], shared_boxes=True), 100)
        alg = N * A * (16 + 1 + 4) + int(cap['gt_cat'].numel()) * 4
        tr, src = _pmc_traffic('iou_gtmax_kernel+iou_assign_kernel', A)
        res['roofline_iou_assign'] = dict(
            bound='hbm', achieved=round(alg / (us * 1e-6) / 1e9, 1), peak=HBM_PEAK_GBS, unit='GB/s',
            frac=round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), traffic=tr,
            traffic_source=('committed PMC measurement, not collected in this run: %s' % src) if src else None,
            kernel='fill_i32_kernel + iou_gtmax_kernel + iou_assign_kernel (bgs_iou_assign: 3 launches)',
            us_per_call=round(us, 2), anchors=A, images=N, gts=int(cap['gt_cat'].shape[0]),
            algorithmic_bytes=alg,
            note='HBM-bound by class (SURVEY 8d) but 11 MB per call: three dependent launches of ~5 us each are '
                 'launch / latency bound, the figure to read is us_per_call')
    return res


def _cpu_time_threads(fn, n, seconds, cores):
    """median time of fn() per thread count; best of 1 / 8 / 32 / min(cores, 64)."""
    best, tried = None, {}
    counts = [nt for nt in sorted(set([1, 8, 32, min(cores, 64)])) if nt <= cores]
    for nt in counts:
        torch.set_num_threads(nt)
        for _ in range(3):
            fn(0)
        times = []
        t_end = time.perf_counter() + seconds / max(len(counts), 1)
        while time.perf_counter() < t_end and len(times) < 500:
            t0 = time.perf_counter()
            fn(len(times))
            times.append(time.perf_counter() - t0)
        med = float(np.median(times))
        tried[str(nt)] = round(med * 1e6 / n, 4)
        if best is None or med < best[0]:
            best = (med, nt, len(times))
    return best, tried


def cpu_baseline(n, seconds):
    """The reference's CPU path on THIS host's cores: the reference's own
    ``GSBBoxHeadWith0.loss()`` + ``backward()`` (gs_bbox_head_with0.py:147-186), imported from the
    head closure that oracle/build_ref.py stages under oracle/_ref/ (``kind: "reference"``); when
    that is absent, the torch-CPU port of it (oracle/gs_torch_port.py, ``kind: "port"``)."""
    import tempfile
    from oracle import build_ref, gs_oracle, gs_torch_port
    counts = gs_tables.synthetic_instance_counts(NUM_CLASSES, seed=0)
    l2b, ps, _ = gs_tables.build_group_tables(counts)
    batch = gs_oracle.make_roi_batch(n, int(ps[:, 1].sum()), NUM_CLASSES, seed=0)
    z, lab = torch.from_numpy(batch['logits']), torch.from_numpy(batch['labels'])
    l2b_t, ps_t = torch.from_numpy(l2b), torch.from_numpy(ps)
    cores = os.cpu_count() or 1

    def port(i):
        np.random.seed(i)
        gs_torch_port.gs_loss_fwd_bwd(z, lab, l2b_t, ps_t, 8.0)

    out = None
    root = build_ref.reference_python_root()
    if root is not None:
        try:
            from oracle import ref_import
            ref_import.install_stubs(root=root)
            tmp = tempfile.mkdtemp(prefix='bgs_ref_tables_')
            gs_tables.save_group_tables(tmp, *gs_tables.synthetic_group_tables())
            head = ref_import.build_reference_head(tmp)
            zr = z.clone().requires_grad_(True)

            def ref(i):
                np.random.seed(i)
                zr.grad = None
                losses = head.loss(zr, None, lab, None, None, None)
                sum(losses.values()).backward()

            (med, nt, cnt), tried = _cpu_time_threads(ref, n, seconds * 0.7, cores)
            out = dict(value=round(med * 1e6 / n, 4), unit='us/RoI', cores=nt, kind='reference',
                       host_cores=cores, threads_tried=tried,
                       sample='%d x (loss()+backward()) of the reference class GSBBoxHeadWith0 itself '
                              '(mmdet/models/bbox_heads/gs_bbox_head_with0.py, imported from %s under the '
                              'dependency stubs of oracle/ref_import.py) on N=%d RoIs x 1236 logits (cls '
                              'branch, numpy sampling incl.), median; best of 1/8/32/64 threads = %d; '
                              'torch %s' % (cnt, 'the reference tree' if root == build_ref.REF else
                                            'oracle/_ref/reference_py (staged by oracle/build_ref.py)',
                                            n, nt, torch.__version__))
            seconds *= 0.3
        except Exception as e:  # pragma: no cover
            sys.stderr.write('reference-class cpu_baseline failed (%r); timing the port\n' % (e,))
            out = None
    (med, nt, cnt), tried = _cpu_time_threads(port, n, seconds, cores)
    pd = dict(value=round(med * 1e6 / n, 4), unit='us/RoI', cores=nt, kind='port',
              host_cores=cores, threads_tried=tried,
              sample='%d x (loss+backward) of the torch-CPU port of GSBBoxHeadWith0.loss '
                     '(oracle/gs_torch_port.py) on N=%d RoIs x 1236 logits, median; best thread count = %d'
                     % (cnt, n, nt))
    if out is None:
        return pd
    out['port'] = pd
    return out


def cpu_baseline_detector(live=True):
    """The EXECUTED reference detector's whole training iteration on CPU (cfg[1]: the same shapes, GT count and
    sampler sizes as this bench, ``selectp=1``).  LIVE on this host's cores when the reference closure staged by
    oracle/build_ref.py travels with the tree (``oracle/_ref/reference_py`` + the host-built ops of oracle/_ref):
    tools/ref_cpu_detector_time.py in a child process (the import stubs patch ``torch.Tensor.cuda``: kept out of
    this process), one warm-up + one timed iteration per thread count, best of 16 / 64 threads — a bounded sample
    (~25 s).  The record measured in the authoring container (8 cores) is quoted next to it, or alone when the
    live leg cannot run."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, 'profiles', 'r2v_reference_detector_cpu_time.jsonl')
    out = None
    try:
        recs = [json.loads(ln) for ln in open(path) if ln.strip().startswith('{')]
        out = dict(kind='reference', where='authoring container (not this host)', records=recs,
                   source='profiles/r2v_reference_detector_cpu_time.jsonl')
    except Exception:  # pragma: no cover
        out = None
    if not live:
        return out
    try:
        from oracle import build_ref
        root = build_ref.reference_python_root()
        if root is None:
            return out
        cores = os.cpu_count() or 1
        best = None
        tried = []
        for nt in sorted({min(16, cores), min(64, cores)}):
            env = dict(os.environ, BGS_REFERENCE_ROOT=root, CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES='',
                       OMP_NUM_THREADS=str(nt))
            r = subprocess.run([sys.executable, os.path.join(here, 'tools', 'ref_cpu_detector_time.py'), '--iters', '1',
                                '--selectp', '1', '--threads', str(nt)], env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.DEVNULL, timeout=240)
            lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith('{')]
            if r.returncode != 0 or not lines:
                continue
            rec = json.loads(lines[-1])
            tried.append((nt, rec['s_per_iter']))
            if best is None or rec['s_per_iter'] < best['s_per_iter']:
                best = rec
        if best is None:
            return out
        live_rec = dict(kind='reference', where='this host', value=best['img_per_s'], unit='img/s',
                        s_per_iter=best['s_per_iter'], cores=best['threads'], host_cores=cores,
                        threads_tried=tried,
                        sample='1 training iteration (2 x 3x800x1344, 20 GT / image, shipped samplers, selectp=1: '
                               'forward + losses + backward) of the executed reference detector after one warm-up '
                               'iteration, per thread count; %s; ops = the reference\'s nms_cpu.cpp / RoIAlign kernels '
                               'built for the host (oracle/_ref); torch %s'
                               % ('the reference tree' if root == build_ref.REF else
                                  'oracle/_ref/reference_py (staged by oracle/build_ref.py)', best['torch']))
        if out is not None:
            live_rec['authoring_container_record'] = out
        return live_rec
    except Exception as e:  # pragma: no cover
        sys.stderr.write('live cpu_baseline_detector failed (%r); quoting the committed record\n' % (e,))
        return out



def extras(dev, args):
    """Secondary single-GPU measurements of the same step in the other §8 configurations (not the
    headline value): selectp=0 (train everything but the frozen stem+layer1), the Mask R-CNN
    config (cfg[3]) and the X101-64x4d cascade (cfg[4], fp32).  Each runs in its OWN process
    (this script with --no-extras), so a failure there can never take the headline line down."""
    import subprocess
    res = {}
    runs = (('f32_mfma_math_selectp1', ['--conv-math', 'f32']),
            ('selectp0', ['--selectp', '0']),
            ('mask_rcnn_selectp1', ['--mask']),
            ('mask_rcnn_selectp0', ['--mask', '--selectp', '0']),
            ('cascade_x101_64x4d_selectp3_fp32', ['--cascade', '--selectp', '3']),
            ('cascade_x101_64x4d_selectp3_bf16', ['--cascade', '--selectp', '3', '--conv-math', 'bf16']),
            ('cascade_x101_64x4d_selectp3_bf16_fp32storage',
             ['--cascade', '--selectp', '3', '--conv-math', 'bf16'], {'BGS_BF16_STORAGE': '0'}),
            ('htc_x101_64x4d_selectp3_fp32', ['--htc', '--selectp', '3']),
            ('htc_x101_64x4d_selectp3_bf16', ['--htc', '--selectp', '3', '--conv-math', 'bf16']))
    for run in runs:
        key, flags = run[0], run[1]
        env = dict(os.environ, **run[2]) if len(run) > 2 else None
        cmd = [sys.executable, os.path.abspath(__file__), '--workload', 'detector', '--steps', '10',
               '--warmup', '3', '--imgs', str(args.imgs), '--no-extras', '--no-cpu-baseline',
               '--no-roofline'] + (['--conv-math', args.conv_math] if '--conv-math' not in flags else []) \
            + flags + (['--no-graph'] if args.no_graph else []) + ['--launch', args.launch]
        try:
            out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420, env=env)
            line = [ln for ln in out.stdout.decode().splitlines() if ln.startswith('{')]
            if out.returncode != 0 or not line:
                res[key] = {'error': 'rc=%d %s' % (out.returncode, out.stderr.decode()[-160:])}
                continue
            d = json.loads(line[-1])
            res[key] = {'img_per_s': d['value'], 'ms_per_step': d['ms_per_step'],
                        'conv_math': d['config'].get('conv_math_mode'),
                        'trainable_params': d['config']['trainable_params'],
                        'loss': d['last_losses']['loss'], 'launch': d['config']['launch']}
        except Exception as e:  # pragma: no cover
            res[key] = {'error': repr(e)[:200]}
    # test time (informational): simple_test on one 800 x 1344 image, 1000 proposals, score_thr = 0, 1230 classes
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'infer_time.py'), '20'],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        line = [ln for ln in out.stdout.decode().splitlines() if ln.startswith('{')]
        res['inference_simple_test'] = json.loads(line[-1]) if out.returncode == 0 and line else \
            {'error': 'rc=%d %s' % (out.returncode, out.stderr.decode()[-160:])}
    except Exception as e:  # pragma: no cover
        res['inference_simple_test'] = {'error': repr(e)[:200]}
    return res


def run_graph_child(args):
    """Single-GPU headline measurement (hipGraph replay of the whole step) in a child process: a
    GPU fault inside a graph replay cannot be caught in-process, so the parent keeps the ability
    to fall back to eager launches and still print its line.  The timed region (barrier + sync
    around exactly K steps) lives entirely in the child."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--workload', 'detector', '--child',
           '--gpus', '1', '--steps', str(args.steps), '--warmup', str(args.warmup),
           '--imgs', str(args.imgs), '--selectp', str(args.selectp), '--no-extras',
           '--no-cpu-baseline', '--no-roofline', '--conv-math', args.conv_math]
    cmd += (['--mask'] if args.mask else []) + (['--cascade'] if args.cascade else [])
    cmd += ['--htc'] if args.htc else []
    cmd += ['--dist-graph'] if args.dist_graph else []
    cmd += ['--launch', args.launch]
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        line = [ln for ln in out.stdout.decode().splitlines() if ln.startswith('{')]
        if out.returncode == 0 and line:
            return json.loads(line[-1])
        sys.stderr.write('graph-replay child failed (rc=%d): %s\n'
                         % (out.returncode, out.stderr.decode()[-300:]))
    except Exception as e:  # pragma: no cover
        sys.stderr.write('graph-replay child failed: %r\n' % (e,))
    return None


def gs_head_metric(inp, n, steps=300, warmup=20):
    """The BASELINE metric's second half, 'GroupSoftmax us/RoI', as the whole head-loss step
    (GSBBoxHeadWith0.loss() + backward(): label remap, 'others' sampling, per-bin loss forward and
    backward, box loss, the sums) on a 1024-RoI batch resident in HBM — what `--workload gs_head`
    reports as its `value`, here as a field of the default line."""
    step = GsHeadStep(inp)
    graph = try_graph(step)
    fn = graph.replay if graph is not None else step
    dt = timed_loop(fn, steps, warmup, 1)
    # the same step under an arbitrary upstream gradient (the scaling launch of the autograd edge runs)
    step_g = GsHeadStep(inp, unit_root=False)
    graph_g = try_graph(step_g)
    dt_g = timed_loop(graph_g.replay if graph_g is not None else step_g, steps, warmup, 1)
    return dict(value=round(dt * 1e6 / (steps * n), 6), unit='us/RoI', us_per_step=round(dt * 1e6 / steps, 2),
                us_per_step_any_upstream=round(dt_g * 1e6 / steps, 2),
                rois_per_step=n, steps=steps,
                launch='hipGraph replay' if graph is not None else 'eager launches',
                what='bgs_gs_head_step + total.backward(unit_gradient): main kernel (label remap + others '
                     'sampling + per-bin loss fwd + bwd + box branch) + reduce (6 terms, total, draw counter); '
                     'the gradient the forward wrote is the answer when the root gradient is the library\'s '
                     'constant 1 (no launch on the autograd edge); us_per_step_any_upstream = the same step '
                     'under any other upstream gradient (one scaling launch more); launch-latency bound')


STEP_GFLOP = {
    # algorithmic flops of one 2-image step (~212 GMAC = 424 GFLOP forward per
    # image; selectp=1 adds dW_cls only, selectp=0 ~3x minus the frozen stem + layer1)
    1: 2 * 426.0, 0: 2 * 1200.0,
}


def step_layer_floor(imgs, conv_math):
    """Per-layer floor of the selectp = 1 cfg[1] step: sum over the conv / linear layers of
    max(flops / matrix-pipe peak, algorithmic bytes / 8 TB/s) — a K = 64 layer of ResNet layer1 is priced by the
    bytes it must move (input + output + residual + filter), not by its MFMAs.  Layer shapes: resnet.py:220-266,
    522-533 (stem 7x7 / s2 + max-pool, stages 3-4-6-3), fpn.py:101-141, rpn_head.py:30-35, convfc_bbox_head.py:
    132-168, at 800 x 1344.  Returns (floor_ms, mfma_part_ms, hbm_bound_ms, n_layers, hbm_bound_layers)."""
    peak = {'bf16x6': 2500.0 / 6.0, 'f32': 157.3, 'bf16': 2500.0}[conv_math] * 1e12
    hbm = HBM_PEAK_GBS * 1e9
    N = imgs
    layers = []     # (name, M = output pixels, K, Cout, input bytes, extra bytes (residual), count)

    def conv(name, H, W, Cin, Cout, R, stride, count=1, residual=False):
        Ho, Wo = H // stride, W // stride
        M = N * Ho * Wo
        inb = N * H * W * Cin * 4
        layers.append((name, M, R * R * Cin, Cout, inb, M * Cout * 4 if residual else 0, count))

    conv('stem', 800, 1344, 3, 64, 7, 2)            # (+ max-pool: its 34 MB output is what leaves)
    H1, W1 = 200, 336
    conv('l1.c1(64)', H1, W1, 64, 64, 1, 1)
    conv('l1.c1(256)', H1, W1, 256, 64, 1, 1, 2)
    conv('l1.c2', H1, W1, 64, 64, 3, 1, 3)
    conv('l1.c3', H1, W1, 64, 256, 1, 1, 3, residual=True)
    conv('l1.ds', H1, W1, 64, 256, 1, 1)
    for pl, (hi, wi), nb in ((128, (200, 336), 4), (256, (100, 168), 6), (512, (50, 84), 3)):
        ho, wo = hi // 2, wi // 2
        conv('c1', hi, wi, pl * 2, pl, 1, 1)
        conv('c2s2', hi, wi, pl, pl, 3, 2)
        conv('ds', hi, wi, pl * 2, pl * 4, 1, 2)
        conv('c1', ho, wo, pl * 4, pl, 1, 1, nb - 1)
        conv('c2', ho, wo, pl, pl, 3, 1, nb - 1)
        conv('c3', ho, wo, pl, pl * 4, 1, 1, nb, residual=True)
    for (h, w, c) in ((200, 336, 256), (100, 168, 512), (50, 84, 1024), (25, 42, 2048)):
        conv('fpn.lat', h, w, c, 256, 1, 1, residual=(h != 25))     # (top-down add fused in the lateral's epilogue)
        conv('fpn.out', h, w, 256, 256, 3, 1)
    for (h, w) in ((200, 336), (100, 168), (50, 84), (25, 42), (13, 21)):
        conv('rpn.conv', h, w, 256, 256, 3, 1)
        conv('rpn.head', h, w, 256, 15, 1, 1)
    R = 512 * N
    for name, K, Cout in (('fc1', 12544, 1024), ('fc2', 1024, 1024), ('fc_cls', 1024, 1236), ('fc_reg', 1024, 4924)):
        layers.append((name, R, K, Cout, R * K * 4, 0, 1))
    layers.append(('fc_cls.dW', 1236, R, 1024, R * (1236 + 1024) * 4, 0, 1))
    floor = mfma_ms = hbm_ms = 0.0
    nl = nh = 0
    for name, M, K, Cout, inb, extra, count in layers:
        flops = 2.0 * M * K * Cout
        outb = M * Cout * 4 if name != 'stem' else M * Cout          # (the stem's map is pooled 4:1 before it leaves)
        byts = inb + outb + extra + K * Cout * 4
        t_m, t_b = flops / peak, byts / hbm
        floor += count * max(t_m, t_b)
        mfma_ms += count * t_m
        nl += count
        if t_b > t_m:
            nh += count
            hbm_ms += count * t_b
    return floor * 1e3, mfma_ms * 1e3, hbm_ms * 1e3, nl, nh


def roofline_step(out, args):
    """The WHOLE step against the matrix-pipe ceiling (the line's `roofline` describes the best
    layer of the dominant kernel only): algorithmic GFLOP per step / ms_per_step / ceiling, plus the
    per-family kernel time of the last committed rocprofv3 trace (profiles/step_families.json)."""
    if args.mask or args.cascade or args.htc or args.selectp not in STEP_GFLOP:
        return None
    gf = STEP_GFLOP[args.selectp] * args.imgs / 2.0
    peak = {'bf16x6': 2500.0 / 6.0, 'f32': 157.3, 'bf16': 2500.0}[args.conv_math]
    tf = gf / out['ms_per_step']          # GFLOP / ms = TFLOP/s
    r = dict(bound='mfma', achieved=round(tf, 1), peak=round(peak, 1), unit='TFLOP/s',
             frac=round(tf / peak, 4), gflop_per_step=gf, ms_per_step=out['ms_per_step'],
             note='algorithmic flops of the whole iteration (conv + FC; SURVEY.md 8d) per GPU / wall '
                  'time per step / the arithmetic mode\'s matrix-pipe ceiling; the step also holds '
                  'HBM- and latency-bound kernels (targets, NMS, RoIAlign, losses, optimizer)')
    if args.selectp == 1:
        fl, mm, hb, nl, nh = step_layer_floor(args.imgs, args.conv_math)
        r['per_layer_floor'] = dict(
            floor_ms=round(fl, 3), frac=round(fl / out['ms_per_step'], 4), mfma_only_ms=round(mm, 3),
            hbm_bound_layers=nh, layers=nl, hbm_bound_ms=round(hb, 3),
            note='sum over the %d conv / linear launches of max(flops / %.1f TFLOP/s, algorithmic bytes / 8 TB/s); '
                 '%d of them (ResNet layer1, the stem, the RPN heads, fc_cls dW) are priced by their bytes; frac = '
                 'floor / ms_per_step' % (nl, peak, nh))
    try:
        with open(os.path.join(ROOT, 'profiles', 'step_families.json')) as f:
            fam = json.load(f)
        r['families_ms'] = fam['families_ms']
        r['families_source'] = fam['source']
    except Exception:
        pass
    return r


def finish_line(out, args, dev, world):
    """Secondary measurements + per-kernel rooflines + CPU baseline, then the ONE JSON line."""
    # (per-kernel rooflines first, the other configurations after them; every measurement here is standalone)
    run_extras = world == 1 and not args.no_extras and args.selectp == 1 and not args.mask \
        and not args.cascade and not args.htc
    if not args.no_roofline:
        out['roofline'] = conv_roofline(dev, args.conv_math)
        if args.conv_math != 'f32':
            out['roofline_f32_mfma_kernel'] = conv_roofline(dev, 'f32')
        if args.conv_math == 'bf16x6' and (out.get('launch_calibration') or {}).get('chosen') == 'eager_pipelined':
            # the timed steps ran through train.TrunkPipeline, which switches the halo kernel's wide schedule on:
            # the same layer under THAT schedule (`roofline` above is the one-launch form the rocprofv3 summary holds)
            out['roofline_wide_schedule'] = conv_roofline(dev, args.conv_math, wide=1)
        rs = roofline_step(out, args)
        if rs:
            out['roofline_step'] = rs
        gs_inp = make_inputs(1024, seed=1000, dev=dev)
        out['roofline_gs_loss'] = kernel_roofline(gs_inp, 1024, kernel='fused')
        out['roofline_gs_loss_rowwave'] = kernel_roofline(gs_inp, 1024, kernel='rowwave')
        big = make_inputs(65536, seed=7, dev=dev)
        out['roofline_gs_loss_n65536'] = kernel_roofline(big, 65536, iters=30, kernel='rowwave')
        del big
        if world == 1:
            out['gs_head'] = gs_head_metric(gs_inp, 1024)
        del gs_inp
        torch.cuda.empty_cache()
        if world == 1 and not args.mask and not args.cascade and not args.htc:
            try:
                out.update(hbm_kernel_rooflines(dev, args.conv_math))
            except Exception as e:  # pragma: no cover  (never lose the line to a secondary measurement)
                out['roofline_hbm_kernels_error'] = repr(e)[:300]
            torch.cuda.empty_cache()
    if run_extras:
        out['also_measured'] = extras(dev, args)
    if world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline(1024, args.cpu_seconds)
        cb['note'] = ('GroupSoftmax loss()+backward() only: the reference cannot run the whole '
                      'detector on CPU as shipped (its RoIAlign has no CPU path, roi_align.py:27-28); '
                      'see cpu_baseline_detector for the whole iteration with its ops built for the host')
        out['cpu_baseline'] = cb
        out['cpu_baseline_1thread'] = {'value': cb['threads_tried'].get('1'), 'unit': 'us/RoI',
                                       'cores': 1, 'kind': cb['kind']}
        cbd = cpu_baseline_detector()
        if cbd:
            out['cpu_baseline_detector'] = cbd
    print(json.dumps(out), flush=True)


def run_dist_graph_children(args, rank, local, world):
    """N > 1: the whole-step hipGraph policy (RCCL all-reduce captured with the rest of the step)
    measured in CHILD processes — one per rank, forming their own process group on another port —
    exactly as the 1-GPU path isolates its graph replay: a fault or a hang inside a replay cannot
    take the parent's eager measurement (and its JSON line) down.  Runs BEFORE the parent builds its
    model, so the GPU is the child's alone.  Returns a dict on rank 0 (None elsewhere)."""
    import socket
    import subprocess
    import torch.distributed as dist
    port = torch.zeros(1, dtype=torch.int64, device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
    if rank == 0:
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port[0] = sk.getsockname()[1]
    dist.broadcast(port, 0)
    env = dict(os.environ, MASTER_PORT=str(int(port.item())), RANK=str(rank), WORLD_SIZE=str(world),
               LOCAL_RANK=str(local))
    env.setdefault('MASTER_ADDR', '127.0.0.1')
    for k in list(env):
        if k.startswith('TORCHELASTIC_'):
            env.pop(k)
    cmd = [sys.executable, os.path.abspath(__file__), '--workload', 'detector', '--child', '--dist-graph',
           '--gpus', str(world), '--steps', str(min(args.steps, 10)), '--warmup', '3',
           '--imgs', str(args.imgs), '--selectp', str(args.selectp), '--no-extras', '--no-cpu-baseline',
           '--no-roofline', '--conv-math', args.conv_math]
    res = dict(ok=False)
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        try:
            so, se = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            p.kill()                      # exactly the child this rank started
            so, se = p.communicate()
            res['error'] = 'timeout'
        line = [ln for ln in so.decode().splitlines() if ln.startswith('{')]
        if p.returncode == 0 and line:
            d = json.loads(line[-1])
            res = dict(ok='hipGraph' in d['config']['launch'], ms_per_step=d['ms_per_step'],
                       img_per_s=d['value'], launch=d['config']['launch'], steps=d['steps'])
        elif 'error' not in res:
            res['error'] = 'rc=%s %s' % (p.returncode, se.decode()[-200:])
    except Exception as e:  # pragma: no cover
        res['error'] = repr(e)[:200]
    barrier(world)
    return res if rank == 0 else None


def detector_line(args, step, world, imgs_per_s, ms_per_step, graph, pipelined, depth, self_group=False):
    """The fields of the detector JSON line that describe WHAT was timed (shared by the N = 1 and N > 1 paths)."""
    cfg_name = 'gs_faster_rcnn_r50_fpn_1x_lvis_with0_bg8 (cfg[1])'
    if args.htc:
        cfg_name = 'gs_htc_x101_64x4d_fpn_20e_16gpu_lvis (cfg[4], HTC)'
    elif args.cascade:
        cfg_name = 'gs_cascade_rcnn_x101_64x4d_fpn_1x_lvis (cfg[4])'
    elif args.mask:
        cfg_name = 'gs_mask_rcnn_r50_fpn_1x_lvis (cfg[3])'
    lv = {k: round(float(v), 5) for k, v in step.last.items()}
    return {
        'metric': 'img/s fwd+bwd R50-FPN+BAGS 1333x800, 512 RoI (BASELINE metric: img/s/GPU '
                  'fwd+bwd R50-FPN+BAGS 1333x800, 512 RoI; GroupSoftmax us/RoI)',
        'value': round(imgs_per_s, 3), 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': {'f32': 'f32', 'bf16': 'bf16 (conv/linear operands rounded to bf16, fp32 accumulate; %s; '
                                               'fp32 master weights and losses)'
                                               % ('bf16 storage of the frozen trunk activations'
                                                  if os.environ.get('BGS_BF16_STORAGE', '1') != '0'
                                                  else 'fp32 storage'),
                  'bf16x6': 'f32 (conv/linear products on the bf16 MFMA from exact 3-way bf16 '
                            'splits, fp32 accumulate: fp32-faithful)'}[args.conv_math],
        'data': 'synthetic',
        'config': {'conv_math_mode': args.conv_math, 'conv_math': CONV_MATH_NOTE[args.conv_math],
                   'workload': cfg_name + ' training '
                               'iteration %s, grad all-reduce, clip 35, SGD): '
                               '%d img/GPU, 3x800x1344 (1333x800 padded /32), 20 GT/img, '
                               '512 RoI/img, 1231 classes, 5 bins; random-init weights'
                               % (('as shipped (selectp=1: full forward incl. RPN '
                                   'losses/proposals/NMS/assign/sample/RoIAlign/FC heads/'
                                   'GroupSoftmax loss, backward through fc_cls')
                                  if args.selectp == 1 else
                                  ('with selectp=0 (train everything but the frozen stem + '
                                   'layer1: full forward and full backward through heads, '
                                   'RoIAlign, RPN, FPN, ResNet layer2-4'), args.imgs),
                   'selectp': args.selectp, 'mask_branch': bool(args.mask),
                   'cascade_x101': bool(args.cascade), 'htc_x101': bool(args.htc),
                   'trainable_params': int(sum(p.numel() for p in step.params)),
                   'imgs_per_gpu': args.imgs, 'rois_per_img': 512,
                   'launch': ('hipGraph replay of the whole step (forward+losses+backward+'
                              + ('RCCL all-reduce+' if (world > 1 or self_group) else '') +
                              'clip+SGD)') if graph else
                   ('eager launches, %d-stage software pipeline (train.TrunkPipeline): the frozen trunk of the '
                    'batches ahead, in %d piece(s) on their own streams (depth 3: backbone(i+2) | FPN(i+1)), beside '
                    'batch i\'s RPN / proposal chain / RoI heads / losses / backward / optimizer step; one pass of '
                    'every piece + one head pass per timed step, bit-identical training (tests/test_gpu_e2e.py)'
                    % (depth, depth - 1) if pipelined else 'eager launches'),
                   'parallelism': 'dp%d (one process per GPU; flat fp32 all-reduce of the '
                                  '%d trainable grads over RCCL)'
                                  % (world, sum(p.numel() for p in step.params)),
                   'ranks': world,
                   'collective_backend': (__import__('torch.distributed').distributed.get_backend()
                                          if world > 1 else None)},
        'img_per_s_per_gpu': round(imgs_per_s / world, 3),
        'last_losses': lv,
    }


def main_detector_dist(args, rank, local, world, dev, step, dist_graph):
    """N > 1 (one process per GPU, launched by torch.distributed.run): calibration, the K timed steps and the
    diagnostics run through bench_dist.run — headline first, every diagnostic guarded and bounded (its docstring)."""
    import torch.distributed as dist

    def make_one_rank_step():
        return DetectorStep(dev, 0, 1, args.imgs, args.selectp, args.mask, args.cascade, args.htc,
                            conv_math=args.conv_math)

    def emit(f):
        if 'ms_per_step' not in f:              # the watchdog fired before the timed region finished
            print(json.dumps({'metric': 'img/s fwd+bwd R50-FPN+BAGS 1333x800, 512 RoI', 'value': None, 'unit': 'img/s',
                              'n_gpus': world, 'error': f.get('error', 'no measurement')}), flush=True)
            return
        depth = f['pipeline_depth']
        out = detector_line(args, step, world, args.imgs * world * args.steps / f['dt'], f['ms_per_step'], None,
                            bool(depth), depth)
        out['rccl_ranks'] = world if dist.get_backend() == 'nccl' else 0
        out['launch_policy'] = 'eager launches (the RCCL all-reduce between backward and the optimizer)'
        out['ms_per_step_by_rank'] = dict(min=min(f['rank_ms']), max=max(f['rank_ms']), all=f['rank_ms'])
        calib = f.get('launch_calibration')
        if calib is not None:
            out['launch_calibration'] = calib
            if depth and calib.get('eager_forks_on_ms') is not None:
                out['ms_per_step_eager'] = calib['eager_forks_on_ms']
        for k, v in (f.get('diagnostics') or {}).items():
            out[k] = v
        if dist_graph is not None:
            out['dist_graph_policy'] = dist_graph
        if f.get('watchdog'):
            # printed from the watchdog thread while the main thread may be stuck in a collective: no GPU work here
            out['watchdog'] = f['watchdog']
            print(json.dumps(out), flush=True)
            return
        finish_line(out, args, dev, world)

    bench_dist.run(step, args, rank, world, make_one_rank_step, emit,
                   launch_auto=args.launch == 'auto' and not os.environ.get('BGS_BENCH_NO_DIST_CALIB'))


def main_detector(args, rank, local, world, dev):
    fallback_note = None
    if world == 1 and not args.no_graph and not args.child:
        out = run_graph_child(args)
        if out is not None:
            finish_line(out, args, dev, world)
            return
        fallback_note = 'hipGraph replay failed in the measurement child; eager launches instead'
        args.no_graph = True
    if args.child and os.environ.get('BGS_BENCH_CHILD_FAIL'):     # test hook for the fallback path
        os._exit(134)
    dist_graph = None
    if world > 1 and not args.child and not args.no_graph and not args.dist_graph \
            and os.environ.get('BGS_BENCH_DIST_GRAPH_CHILD'):
        # opt-in (round 6): the whole-step hipGraph with the RCCL all-reduce captured, in child processes (up to 180 s)
        dist_graph = run_dist_graph_children(args, rank, local, world)
    step = DetectorStep(dev, rank, world, args.imgs, args.selectp, args.mask, args.cascade,
                        args.htc, conv_math=args.conv_math)
    if world > 1 and not args.child and not args.dist_graph:
        return main_detector_dist(args, rank, local, world, dev, step, dist_graph)
    # Launch policy.  The iteration is free of host synchronisation, so on ONE GPU the whole
    # step (forward, losses, backward, clip, SGD: ~560 launches) is captured into a single
    # hipGraph and replayed.  The graph must own the ENTIRE step: on ROCm 7.2 a large graph whose
    # replays are interleaved with eager launches (an eager optimizer step, or the host-side
    # philox bookkeeping of torch.randint inside a captured region) faults after a few dozen
    # replays (tools/two_graphs_repro.py reproduces it: "variants plain" vs "variants whole";
    # DESIGN.md §5).  With N > 1 the gradient all-reduce (RCCL) sits between backward and the
    # optimizer, so multi-GPU runs launch eagerly (main_detector_dist above) — the step is GPU-bound and eager launches
    # cost nothing measurable (10.99 vs 10.96 ms); what is left here for N > 1 is `--dist-graph` (the whole step incl.
    # the collective in one hipGraph per rank) and its measurement children.
    graph = None
    self_group = bool(os.environ.get('BGS_BENCH_SELF_GROUP'))
    rccl = world > 1 and __import__('torch.distributed').distributed.get_backend() == 'nccl'
    # Launch policy on one GPU, `--launch auto` (default): the step forks independent launches onto a side stream
    # (functional.forked); replayed from a hipGraph those forks cost event edges, launched eagerly they run as real
    # concurrent streams but pay the host's launch path — which of the two is faster depends on the box and its
    # host (6.30 vs 6.20 ms on one, 6.40 vs 6.42 on another).  Both are calibrated UNTIMED (8 steps each; the eager
    # one before the capture, so that no replay ever follows an eager step that followed a replay: DESIGN 5) and
    # the official K steps are then timed under the faster policy.
    calib = None
    can_graph = not args.no_graph and ((world == 1 and not self_group) or
                                       (args.dist_graph and (rccl or self_group)))
    auto = can_graph and world == 1 and not self_group and args.launch == 'auto'
    pipe_fn = None
    if auto:
        calib = dict(eager_ms=round(timed_loop(step, 8, 6, 1) * 1e3 / 8, 3))      # (6 warm-up steps: lazy folds / splits / caches)
        if step.can_pipeline() and not os.environ.get('BGS_BENCH_NO_PIPELINE'):
            # third policy (round 5): eager launches with the NEXT batch's frozen trunk on its own stream beside this
            # batch's heads / losses / backward / optimizer step (train.TrunkPipeline: bit-identical training)
            depths = [int(os.environ['BGS_BENCH_PIPELINE_DEPTH'])] if os.environ.get('BGS_BENCH_PIPELINE_DEPTH') \
                else [3, 4, 5]
            best_depth = None
            for dpt in depths:
                pipe_fn = step.pipelined(depth=dpt)
                ms = round(timed_loop(pipe_fn, 8, 3, 1) * 1e3 / 8, 3)
                pipe_fn.drain()
                torch.cuda.synchronize()
                calib['eager_pipelined_depth%d_ms' % dpt] = ms
                if best_depth is None or ms < calib['eager_pipelined_ms']:
                    best_depth, calib['eager_pipelined_ms'] = dpt, ms
            calib['pipeline_depth'] = best_depth
    if can_graph and args.launch not in ('eager', 'pipelined'):
        # --dist-graph: the RCCL all-reduce is captured with the rest of the step (the communicator
        # is set up by the eager warm-up iterations inside try_graph)
        graph = try_graph(step)
    fn = graph.replay if graph is not None else step
    if auto and graph is not None:
        calib['graph_ms'] = round(timed_loop(graph.replay, 8, 3, 1) * 1e3 / 8, 3)
        if calib['eager_ms'] < 0.99 * calib['graph_ms']:
            fn = step
        calib['chosen'] = 'eager' if fn is step else 'graph'
    pipelined = False
    if auto and pipe_fn is not None:
        best = min(calib['eager_ms'], calib.get('graph_ms', 1e9))
        if calib['eager_pipelined_ms'] < 0.99 * best:
            fn = step.pipelined(depth=calib['pipeline_depth'])      # (a fresh pipeline: its first features are launched here, untimed)
            pipelined = True
            calib['chosen'] = 'eager_pipelined'
    if args.launch == 'pipelined' and world == 1 and step.can_pipeline():
        fn, pipelined = step.pipelined(depth=int(os.environ.get('BGS_BENCH_PIPELINE_DEPTH', 5))), True
    dt = timed_loop(fn, args.steps, args.warmup, world)
    if pipelined:
        fn.drain()                             # (the features launched by the last timed call: consumed by nobody)
        torch.cuda.synchronize()
    ms_per_step = dt * 1e3 / args.steps
    imgs_per_s = args.imgs * world * args.steps / dt
    rank_ms = None
    if world > 1:       # every rank's own wall time of the timed region (the line reports the max)
        rank_ms = [round(t, 3) for t in bench_dist.gather_scalar(timed_loop.last_local_dt * 1e3 / args.steps, world)]
    ms_eager = None
    ms_graph = None
    if graph is not None and (fn is step or pipelined):      # eager was the timed policy: the graph figure from the calibration
        ms_graph = calib['graph_ms']
        graph = None                           # (the line's `launch` describes what was timed)
    if pipelined:
        ms_eager = (calib or {}).get('eager_ms')
    elif graph is not None:     # every rank: the same step launched eagerly, for the graph-vs-eager figure
        ms_eager = round(timed_loop(step, 5, 2, world) * 1e3 / 5, 3)
    if rank == 0:
        out = detector_line(args, step, world, imgs_per_s, ms_per_step, graph, pipelined,
                            fn.depth if pipelined else 0, self_group)
        if ms_eager is not None:
            out['ms_per_step_eager'] = ms_eager
        if ms_graph is not None:
            out['ms_per_step_graph'] = ms_graph
        if calib is not None:
            out['launch_calibration'] = dict(calib, note='untimed calibration of both launch policies (8 steps each) '
                                                    'before the timed region; the K timed steps ran under `chosen`')
        if world > 1:
            import torch.distributed as dist
            out['rccl_ranks'] = world if dist.get_backend() == 'nccl' else 0
            out['launch_policy'] = 'hipGraph replay incl. the RCCL all-reduce' if graph else \
                'eager launches (the RCCL all-reduce between backward and the optimizer)'
            out['ms_per_step_by_rank'] = dict(min=min(rank_ms), max=max(rank_ms), all=rank_ms)
        if self_group:
            out['config']['collective_backend'] = 'nccl (1-rank group: test hook BGS_BENCH_SELF_GROUP)'
        if fallback_note:
            out['config']['launch_note'] = fallback_note
        finish_line(out, args, dev, world)
    barrier(world)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run (one
    process per GPU, the reference's tools/dist_train.sh:8-9 layout) and relay its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node',
           str(args.gpus), '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the product has no CPU path)')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(spawn_ranks(args))
    rank, local, world = init_dist(args)
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d: the line would report the wrong '
                         'n_gpus' % (args.gpus, world))
    if world > 1:
        import torch.distributed as dist
        assert dist.get_world_size() == args.gpus
    dev = torch.device('cuda', local)
    if args.workload == 'detector':
        return main_detector(args, rank, local, world, dev)
    n = args.rois
    inp = make_inputs(n, seed=1000 + rank, dev=dev)
    step = GsHeadStep(inp)
    graph = None if args.no_graph else try_graph(step)
    fn = graph.replay if graph is not None else step
    dt = timed_loop(fn, args.steps, args.warmup, world)
    ms_per_step = dt * 1e3 / args.steps
    us_per_roi = dt * 1e6 / (args.steps * n * world)

    if rank == 0:
        # eager (python launch overhead included) for reference
        dt_eager = None
        if graph is not None and world == 1:
            dt_eager = timed_loop(step, min(args.steps, 100), 5, 1) * 1e3 / min(args.steps, 100)
        out = {
            'metric': 'GroupSoftmax us/RoI (loss fwd+bwd; BASELINE metric: img/s/GPU fwd+bwd '
                      'R50-FPN+BAGS 1333x800, 512 RoI; GroupSoftmax us/RoI)',
            'value': round(us_per_roi, 6), 'unit': 'us/RoI', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 5),
            'higher_is_better': False, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'gs_head: RoI-head loss of gs_faster_rcnn_r50_fpn_1x_lvis_'
                                   'with0_bg8 (cfg[1]): %d RoIs/GPU (2 img x 512) x 1236 logits, '
                                   '1231 classes, 5 bins, 25%% fg; _remap_labels + fused '
                                   'GroupSoftmax loss fwd+bwd + loss_bbox' % n,
                       'rois_per_gpu': n, 'launch': 'hipGraph replay' if graph else 'eager',
                       'parallelism': 'dp%d (independent RoI batches, no data-path collective)'
                                      % world},
            'rois_per_s': round(n * world * args.steps / dt, 1),
        }
        if dt_eager is not None:
            out['ms_per_step_eager'] = round(dt_eager, 5)
        out['roofline'] = kernel_roofline(inp, n, kernel='fused' if n <= BF.GS_FUSED_MAX_ROWS else 'rowwave')
        out['roofline_rowwave'] = kernel_roofline(inp, n, kernel='rowwave')
        big = make_inputs(65536, seed=7, dev=dev)
        out['roofline_n65536'] = kernel_roofline(big, 65536, iters=30, kernel='rowwave')
        del big
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(n, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    barrier(world)


if __name__ == '__main__':
    main()
