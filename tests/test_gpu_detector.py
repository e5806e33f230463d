"""GPU: the detector assembled from the HIP kernels — backbone/FPN/RPN/RoI head forward
against a torch-CPU restatement of the reference modules' arithmetic (same parameters), and
one full training iteration of the BAGS Faster R-CNN."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import balancedgroupsoftmax_amd as bgs
from balancedgroupsoftmax_amd import gs_tables, train
from balancedgroupsoftmax_amd.config import to_config_dict

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def randomize_bn(model, seed=0):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = torch.rand(m.weight.shape, generator=g) + 0.5
            m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.1
            m.running_mean = torch.randn(m.running_mean.shape, generator=g) * 0.1
            m.running_var = torch.rand(m.running_var.shape, generator=g) + 0.5


def ref_bottleneck(b, x):
    """mmdet/models/backbones/resnet.py:220-266 (pytorch style), eval-mode BN, NCHW, CPU."""
    def cbn(conv, bn, t, relu):
        t = F.conv2d(t, conv.weight, None, conv.stride, conv.padding)
        t = F.batch_norm(t, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0., bn.eps)
        return F.relu(t) if relu else t
    idt = x if b.downsample is None else cbn(b.downsample[0], b.downsample[1], x, False)
    out = cbn(b.conv1, b.bn1, x, True)
    out = cbn(b.conv2, b.bn2, out, True)
    out = cbn(b.conv3, b.bn3, out, False)
    return F.relu(out + idt)


def ref_resnet(m, img):
    x = F.conv2d(img, m.conv1.weight, None, 2, 3)
    x = F.relu(F.batch_norm(x, m.bn1.running_mean, m.bn1.running_var, m.bn1.weight, m.bn1.bias,
                            False, 0., m.bn1.eps))
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    for name in m.res_layers:
        for blk in getattr(m, name):
            x = ref_bottleneck(blk, x)
        outs.append(x)
    return outs


def ref_fpn(m, inputs):
    """mmdet/models/necks/fpn.py:101-141."""
    lat = [F.conv2d(inputs[i], c.conv.weight, c.conv.bias) for i, c in enumerate(m.lateral_convs)]
    for i in range(len(lat) - 1, 0, -1):
        lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], scale_factor=2, mode='nearest')
    outs = [F.conv2d(lat[i], c.conv.weight, c.conv.bias, padding=1)
            for i, c in enumerate(m.fpn_convs)]
    outs.append(F.max_pool2d(outs[-1], 1, stride=2))
    return outs


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def rel_err(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-6))


def test_resnet50_fpn_rpn_forward_vs_torch_cpu():
    torch.manual_seed(0)
    backbone = bgs.build_backbone(dict(type='ResNet', depth=50, num_stages=4,
                                       out_indices=(0, 1, 2, 3), frozen_stages=1, style='pytorch'))
    neck = bgs.build_neck(dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256,
                               num_outs=5))
    rpn = bgs.build_head(dict(type='RPNHead', in_channels=256, feat_channels=256,
                              anchor_scales=[8], anchor_ratios=[0.5, 1.0, 2.0],
                              anchor_strides=[4, 8, 16, 32, 64]))
    backbone.init_weights()
    neck.init_weights()
    rpn.init_weights()
    for b in backbone.modules():     # zero_init_residual would silence every block: undo it
        if hasattr(b, 'bn3'):
            torch.nn.init.constant_(b.bn3.weight, 0.5)
    randomize_bn(backbone)
    for p in list(backbone.parameters()) + list(neck.parameters()) + list(rpn.parameters()):
        p.requires_grad = False
    img = torch.randn(2, 3, 128, 192)
    with torch.no_grad():
        exp_c = ref_resnet(backbone, img)
        exp_p = ref_fpn(neck, exp_c)
        exp_rpn = []
        for x in exp_p:
            h = F.relu(F.conv2d(x, rpn.rpn_conv.weight, rpn.rpn_conv.bias, padding=1))
            exp_rpn.append((F.conv2d(h, rpn.rpn_cls.weight, rpn.rpn_cls.bias),
                            F.conv2d(h, rpn.rpn_reg.weight, rpn.rpn_reg.bias)))
    backbone.to(DEV), neck.to(DEV), rpn.to(DEV)
    got_c = backbone(img.to(DEV))
    for g, e in zip(got_c, exp_c):
        assert tuple(g.shape) == (e.shape[0], e.shape[2], e.shape[3], e.shape[1])
        assert rel_err(nchw(g), e) < 1e-4
    got_p = neck(got_c)
    assert len(got_p) == 5
    for g, e in zip(got_p, exp_p):
        assert rel_err(nchw(g), e) < 1e-4
    cls, reg = rpn(got_p)
    for c, r, (ec, er) in zip(cls, reg, exp_rpn):
        assert rel_err(nchw(c), ec) < 1e-4 and rel_err(nchw(r), er) < 1e-4


def _detector(tmp_path, small=False):
    paths = gs_tables.save_group_tables(str(tmp_path), *gs_tables.synthetic_group_tables())
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
                       roi_feat_size=7, num_classes=1231, target_means=[0., 0., 0., 0.],
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
    return bgs.build_detector(to_config_dict(model), train_cfg=to_config_dict(train_cfg),
                              test_cfg=None)


def test_roi_head_nhwc_equals_reference_nchw_flatten(tmp_path):
    model = _detector(tmp_path)
    head = model.bbox_head
    head.init_weights()
    torch.manual_seed(1)
    feats = torch.randn(37, 7, 7, 256)
    with torch.no_grad():
        exp_cls, exp_reg = head(feats.permute(0, 3, 1, 2).contiguous())     # CPU, nn.Linear
        head.to(DEV)
        cls, reg = head(feats.to(DEV), nhwc=True)
    assert rel_err(cls.cpu(), exp_cls) < 1e-4 and rel_err(reg.cpu(), exp_reg) < 1e-4


def test_full_training_iteration(tmp_path):
    """One iteration of gs_faster_rcnn_r50_fpn (reduced image size): loss dict keys, finite
    values, gradients only on fc_cls and equal to torch autograd of the last layer."""
    torch.manual_seed(0)
    model = _detector(tmp_path).to(DEV)
    params = train.select_training_param(model, 1)
    model.train()
    H, W = 320, 480
    img = torch.randn(2, 3, H, W, device=DEV)
    metas = [dict(img_shape=(H, W - 5, 3), pad_shape=(H, W, 3), ori_shape=(H, W - 5, 3),
                  scale_factor=1.0, flip=False)] * 2
    g = torch.Generator().manual_seed(3)
    gt_bboxes, gt_labels = [], []
    for _ in range(2):
        xy = torch.rand(12, 2, generator=g) * torch.tensor([W - 120., H - 120.])
        wh = torch.rand(12, 2, generator=g) * 100 + 16
        gt_bboxes.append(torch.cat([xy, xy + wh], 1).to(DEV))
        gt_labels.append(torch.randint(1, 1231, (12,), generator=g).to(DEV))
    losses = model(img, metas, return_loss=True, gt_bboxes=gt_bboxes, gt_labels=gt_labels)
    keys = sorted(losses.keys())
    assert keys == sorted(['loss_rpn_cls', 'loss_rpn_bbox', 'loss_bbox'] +
                          ['loss_cls_bin%d' % i for i in range(5)])
    assert len(losses['loss_rpn_cls']) == 5 and len(losses['loss_rpn_bbox']) == 5
    loss, log_vars = train.parse_losses(losses)
    assert torch.isfinite(loss)
    opt = train.build_optimizer(params, dict(type='SGD', lr=0.01, momentum=0.9, weight_decay=1e-4))
    step = train.DistOptimizerStep(params, opt, dict(max_norm=35, norm_type=2), world_size=1)
    w0 = model.bbox_head.fc_cls.weight.detach().clone()
    step(loss)
    gw = model.bbox_head.fc_cls.weight.grad
    assert gw is not None and torch.isfinite(gw).all() and float(gw.abs().sum()) > 0
    assert model.bbox_head.fc_reg.weight.grad is None
    assert model.backbone.conv1.weight.grad is None
    assert not torch.equal(w0, model.bbox_head.fc_cls.weight)
    # log-loss of 5 bins at random init: bin0 ~ log 2, fg bins ~ log(n_b)
    assert 0.3 < float(log_vars['loss_cls_bin0']) < 1.5
    assert 3.0 < float(log_vars['loss_cls_bin4']) < 9.0
