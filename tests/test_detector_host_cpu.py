"""CPU: host-side detector logic (assignment, sampling, anchors, module/state-dict layout)
checked against the REFERENCE's own classes executed on CPU through the import stubs
(oracle/ref_import.py).  Skipped where the reference tree is absent (GPU box)."""
import os

import numpy as np
import pytest
import torch

import balancedgroupsoftmax_amd as bgs
from oracle import tensor_forms as A
from balancedgroupsoftmax_amd import rpn as R
from balancedgroupsoftmax_amd.config import to_config_dict
from oracle import ref_import

needs_ref = pytest.mark.skipif(not ref_import.reference_available(),
                               reason='reference tree not present')


def rand_boxes(n, seed, size=400):
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand(n, 2, generator=g) * size
    wh = torch.rand(n, 2, generator=g) * 120 + 4
    return torch.cat([xy, xy + wh], 1)


def test_anchor_generator_known_answer():
    """reference doctest mmdet/core/anchor/anchor_generator.py:7-14."""
    g = R.AnchorGenerator(9, [1.], [1.])
    out = g.grid_anchors((2, 2), stride=16, device='cpu')
    assert out.tolist() == [[0., 0., 8., 8.], [16., 0., 24., 8.], [0., 16., 8., 24.],
                            [16., 16., 24., 24.]]


@needs_ref
def test_anchor_generator_vs_reference():
    ref_import.install_stubs()
    from mmdet.core.anchor.anchor_generator import AnchorGenerator as Ref
    for base, fs, st in [(4, (7, 9), 4), (32, (5, 3), 32), (64, (13, 21), 64)]:
        mine, ref = R.AnchorGenerator(base, [8], [0.5, 1.0, 2.0]), Ref(base, [8], [0.5, 1.0, 2.0])
        assert torch.equal(mine.base_anchors, ref.base_anchors)
        assert torch.equal(mine.grid_anchors(fs, st, 'cpu'), ref.grid_anchors(fs, st, 'cpu'))
        v = mine.valid_flags(fs, (fs[0] - 1, fs[1] - 2), 'cpu')
        assert torch.equal(v, ref.valid_flags(fs, (fs[0] - 1, fs[1] - 2), 'cpu').bool())


@needs_ref
def test_bbox_overlaps_vs_reference():
    ref_import.install_stubs()
    from mmdet.core.bbox.geometry import bbox_overlaps as ref
    a, b = rand_boxes(37, 0), rand_boxes(211, 1)
    assert torch.equal(A.bbox_overlaps(a, b), ref(a, b))


@needs_ref
@pytest.mark.parametrize('thr', [(0.7, 0.3, 0.3), (0.5, 0.5, 0.5)])
def test_max_iou_assigner_vs_reference(thr):
    ref_import.install_stubs()
    from mmdet.core.bbox.assigners.max_iou_assigner import MaxIoUAssigner
    pos, neg, minpos = thr
    ref = MaxIoUAssigner(pos_iou_thr=pos, neg_iou_thr=neg, min_pos_iou=minpos, ignore_iof_thr=-1)
    for seed in range(5):
        gts, boxes = rand_boxes(23, seed), rand_boxes(3000, 100 + seed)
        boxes[:40] = gts[torch.arange(40) % 23] + torch.randn(40, 4, generator=torch.Generator(
        ).manual_seed(seed)) * 2            # near-duplicates: real positives
        boxes[40:46] = boxes[:6]            # exact ties between boxes for the same gt (step 4)
        ov = A.bbox_overlaps(gts, boxes)
        exp = ref.assign_wrt_overlaps(ov.clone())
        got, mx = A.max_iou_assign(ov, pos, neg, minpos)
        assert torch.equal(got, exp.gt_inds)
        assert torch.equal(mx, exp.max_overlaps)
        assert (got > 0).sum() > 20 and (got == 0).sum() > 100


def test_sampling_counts_and_order():
    g = torch.Generator().manual_seed(0)
    assigned = torch.full((5000,), -1, dtype=torch.long)
    assigned[:300] = torch.randint(1, 9, (300,), generator=g)      # 300 positives
    assigned[300:4000] = 0                                          # 3700 negatives
    inds, is_pos, valid = A.sample_fixed(assigned, 512, 0.25, g)
    assert inds.shape == (512,) and valid.all()
    assert int(is_pos.sum()) == 128 and is_pos[:128].all() and not is_pos[128:].any()
    assert (assigned[inds[:128]] > 0).all() and (assigned[inds[128:]] == 0).all()
    assert inds.unique().numel() == 512                             # without replacement
    # fewer positives than asked: negatives fill up (base_sampler.py:62)
    assigned[20:300] = 0
    inds, is_pos, valid = A.sample_fixed(assigned, 512, 0.25, g)
    assert int(is_pos.sum()) == 20 and valid.all() and (assigned[inds[20:]] == 0).all()
    # dense form used by the RPN
    p, n = A.sample_pos_neg_masks(assigned, 256, 0.5, generator=g)
    assert int(p.sum()) == 20 and int(n.sum()) == 236 and not (p & n).any()
    assigned[:300] = torch.randint(1, 9, (300,), generator=g)
    p, n = A.sample_pos_neg_masks(assigned, 256, 0.5, generator=g)
    assert int(p.sum()) == 128 and int(n.sum()) == 128
    # uniformity of the positive draw
    hits = torch.zeros(300)
    for _ in range(300):
        p, _ = A.sample_pos_neg_masks(assigned, 256, 0.5, generator=g)
        hits += p[:300].float()
    f = hits / 300
    assert abs(float(f.mean()) - 128 / 300) < 1e-6 and float((f - 128 / 300).abs().max()) < 0.2


def _rpn_cfg(num):
    return to_config_dict(dict(
        assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3,
                      ignore_iof_thr=-1),
        sampler=dict(type='RandomSampler', num=num, pos_fraction=0.5, neg_pos_ub=-1,
                     add_gt_as_proposals=False),
        allowed_border=0, pos_weight=-1, debug=False))


@needs_ref
def test_rpn_anchor_targets_vs_reference():
    """With a sampler budget larger than the anchor count nothing is dropped, so the dense
    targets are deterministic and must equal the reference's anchor_target_single."""
    ref_import.install_stubs()
    from mmdet.core.anchor.anchor_target import anchor_target_single
    head = bgs.build_head(dict(type='RPNHead', in_channels=16, feat_channels=16,
                               anchor_scales=[8], anchor_ratios=[0.5, 1.0, 2.0],
                               anchor_strides=[4, 8, 16, 32, 64]))
    sizes = [(50, 84), (25, 42), (13, 21), (7, 11), (4, 6)]
    meta = dict(img_shape=(200, 333, 3), pad_shape=(200, 336, 3))
    anchor_list, flag_list = head.get_anchors(sizes, [meta], 'cpu')
    anchors, valid = torch.cat(anchor_list[0]), torch.cat(flag_list[0])
    gts = torch.tensor([[20., 30., 120., 140.], [150., 10., 320., 190.], [60., 60., 75., 80.],
                        [5., 100., 40., 180.]])
    cfg = _rpn_cfg(num=10 ** 7)
    labels, lw, bt, bw, npos, nneg = A.rpn_anchor_targets(head, anchors, valid, gts,
                                                          meta['img_shape'], cfg)
    np.random.seed(0)
    exp = anchor_target_single(anchors, valid.to(torch.uint8), gts, None, None, meta,
                               head.target_means, head.target_stds, cfg, sampling=True)
    assert torch.equal(labels, exp[0]) and torch.equal(lw, exp[1])
    assert torch.allclose(bt, exp[2], atol=1e-6) and torch.equal(bw, exp[3])
    assert int(npos) == exp[4].numel() and int(nneg) == exp[5].numel() and int(npos) > 4


REF_KEYS = {}


@needs_ref
def test_state_dict_layout_matches_reference_modules():
    """Checkpoint compatibility surface: parameter/buffer names AND shapes of
    ResNet-50, FPN and RPNHead equal those of the reference's own modules."""
    ref_import.install_stubs()
    from mmdet.models.anchor_heads.rpn_head import RPNHead as RefRPN
    from mmdet.models.backbones.resnet import ResNet as RefResNet
    from mmdet.models.necks.fpn import FPN as RefFPN

    def sig(m):
        return {k: tuple(v.shape) for k, v in m.state_dict().items()}

    mine = bgs.build_backbone(dict(type='ResNet', depth=50, num_stages=4,
                                   out_indices=(0, 1, 2, 3), frozen_stages=1, style='pytorch'))
    ref = RefResNet(depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                    style='pytorch')
    assert sig(mine) == sig(ref)
    assert [n for n, p in mine.named_parameters() if not p.requires_grad] == \
        [n for n, p in ref.named_parameters() if not p.requires_grad]      # frozen_stages=1
    neck = dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=5)
    assert sig(bgs.build_neck(neck)) == sig(RefFPN(in_channels=[256, 512, 1024, 2048],
                                                   out_channels=256, num_outs=5))
    rpn = dict(in_channels=256, feat_channels=256, anchor_scales=[8],
               anchor_ratios=[0.5, 1.0, 2.0], anchor_strides=[4, 8, 16, 32, 64],
               target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0],
               loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
               loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0))
    assert sig(bgs.build_head(dict(type='RPNHead', **rpn))) == sig(RefRPN(**rpn))


@needs_ref
def test_full_detector_builds_from_reference_config(tmp_path):
    """configs/bags/gs_faster_rcnn_r50_fpn_1x_lvis_with0_bg8.py drops in: only the three data
    file paths (absent from the repo, README.md:83-87) are redirected to synthetic tables."""
    from balancedgroupsoftmax_amd import gs_tables, train
    cfg = bgs.Config.fromfile(os.path.join(
        ref_import.REFERENCE_ROOT, 'configs/bags/gs_faster_rcnn_r50_fpn_1x_lvis_with0_bg8.py'))
    paths = gs_tables.save_group_tables(str(tmp_path), *gs_tables.synthetic_group_tables())
    gs = cfg.model.bbox_head.gs_config
    gs.label2binlabel, gs.pred_slice, gs.fg_split = (paths['label2binlabel'], paths['pred_slice'],
                                                     paths['fg_split'])
    model = bgs.build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    assert type(model).__name__ == 'GroupSoftmax'
    keys = set(model.state_dict().keys())
    for k in ['backbone.layer4.2.conv3.weight', 'backbone.layer1.0.downsample.1.running_mean',
              'neck.lateral_convs.3.conv.weight', 'neck.fpn_convs.0.conv.bias',
              'rpn_head.rpn_conv.weight', 'rpn_head.rpn_cls.bias', 'rpn_head.rpn_reg.weight',
              'bbox_head.shared_fcs.0.weight', 'bbox_head.fc_cls.weight', 'bbox_head.fc_reg.bias']:
        assert k in keys, k
    assert model.state_dict()['bbox_head.fc_cls.weight'].shape == (1236, 1024)
    params = train.select_training_param(model, cfg.selectp)           # selectp = 1
    assert sorted(n for n, p in model.named_parameters() if p.requires_grad) == \
        ['bbox_head.fc_cls.bias', 'bbox_head.fc_cls.weight'] and len(params) == 2
    assert sum(p.numel() for p in params) == 1266900                   # 5.07 MB all-reduce payload
    loss, log_vars = train.parse_losses(dict(loss_a=torch.tensor(1.0), acc=torch.tensor(50.0),
                                             loss_b=[torch.tensor(2.0), torch.tensor(3.0)]))
    assert float(loss) == 6.0 and float(log_vars['loss_b']) == 5.0


def test_parse_losses_equals_the_reference_formula_with_gradients():
    """``train.parse_losses`` (single stack + sum reductions) == mmdet/apis/train.py:15-27 (means,
    chained ``sum``) in value and in the gradient every loss tensor receives — on the loss dict
    shapes the detectors produce (per-level lists, 0-dim scalars, a non-scalar entry, an ``acc``)."""
    from collections import OrderedDict
    from balancedgroupsoftmax_amd import train
    g = torch.Generator().manual_seed(0)

    def make():
        d = OrderedDict()
        d['loss_rpn_cls'] = [torch.rand((), generator=g).requires_grad_(True) for _ in range(5)]
        d['loss_rpn_bbox'] = [torch.rand((), generator=g).requires_grad_(True) for _ in range(5)]
        for i in range(5):
            d['s0.loss_cls_bin%d' % i] = torch.rand((), generator=g).requires_grad_(True)
        d['loss_elementwise'] = torch.rand(7, generator=g).requires_grad_(True)     # mean() applies
        d['acc'] = torch.tensor(50.0)                                               # not a loss
        return d

    d = make()
    loss, log_vars = train.parse_losses(d)
    # the reference's arithmetic (its .item() conversion of log_vars aside)
    ref_vars = OrderedDict()
    for k, v in d.items():
        ref_vars[k] = v.mean() if isinstance(v, torch.Tensor) else sum(t.mean() for t in v)
    ref = sum(v for k, v in ref_vars.items() if 'loss' in k)
    assert abs(float(loss.detach()) - float(ref.detach())) < 1e-6
    for k in ref_vars:
        assert abs(float(log_vars[k].detach()) - float(ref_vars[k].detach())) < 1e-6
    assert 'loss' in log_vars and 'acc' in log_vars
    leaves = [t for v in d.values() for t in (v if isinstance(v, list) else [v]) if t.requires_grad]
    got = torch.autograd.grad(loss, leaves)
    exp = torch.autograd.grad(ref, leaves)
    for a, b in zip(got, exp):
        assert torch.allclose(a, b)
    with pytest.raises(TypeError):
        train.parse_losses(dict(loss_x=3.0))
