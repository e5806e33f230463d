"""GPU: ResNeXt grouped conv + Cascade R-CNN (cfg 5 pieces) through the C ABI."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import balancedgroupsoftmax_amd as bgs
from balancedgroupsoftmax_amd import functional as BF
from balancedgroupsoftmax_amd import gs_tables, train
from balancedgroupsoftmax_amd.config import to_config_dict

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('cg,stride', [(4, 1), (8, 2), (16, 1), (32, 2), (32, 1)])
def test_grouped_conv3x3_vs_torch_cpu(cg, stride):
    rs = np.random.RandomState(cg + stride)
    groups = 8
    C = cg * groups
    x = rs.randn(2, 13, 18, C).astype(np.float32)
    w = (rs.randn(C, cg, 3, 3) * 0.2).astype(np.float32)
    b = rs.randn(C).astype(np.float32)
    exp = F.relu(F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w),
                          torch.from_numpy(b), stride=stride, padding=1, groups=groups))
    wk = torch.from_numpy(np.ascontiguousarray(w.transpose(0, 2, 3, 1))).to(DEV)
    got = BF.grouped_conv3x3_nhwc(torch.from_numpy(x).to(DEV), wk, torch.from_numpy(b).to(DEV),
                                  groups, stride=stride, relu=True)
    assert tuple(got.shape) == (2, exp.shape[2], exp.shape[3], C)
    assert float((got.permute(0, 3, 1, 2).cpu() - exp).abs().max()) < 1e-4 * float(exp.abs().max())


@pytest.mark.parametrize('cg,hw', [(4, (13, 18)), (8, (21, 37)), (16, (8, 16)), (32, (25, 42)), (16, (50, 84))])
def test_grouped_conv3x3_lds_resident_kernel_vs_torch_cpu(cg, hw):
    """Stride-1 grouped conv with C % 64 == 0 (every stride-1 conv2 of ResNeXt-101 64x4d): the
    LDS-resident-patch kernel — ragged tiles on both axes, one exact tile, two images, every
    channels-per-group instantiation — against torch-CPU ``F.conv2d(groups=...)``."""
    rs = np.random.RandomState(cg * 7 + hw[0])
    groups = 16
    C = cg * groups
    x = rs.randn(2, hw[0], hw[1], C).astype(np.float32)
    w = (rs.randn(C, cg, 3, 3) * 0.2).astype(np.float32)
    b = rs.randn(C).astype(np.float32)
    exp = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w), torch.from_numpy(b),
                   padding=1, groups=groups)
    wk = torch.from_numpy(np.ascontiguousarray(w.transpose(0, 2, 3, 1))).to(DEV)
    got = BF.grouped_conv3x3_nhwc(torch.from_numpy(x).to(DEV), wk, torch.from_numpy(b).to(DEV), groups,
                                  stride=1, relu=False)
    assert tuple(got.shape) == (2, hw[0], hw[1], C)
    assert float((got.permute(0, 3, 1, 2).cpu() - exp).abs().max()) < 1e-4 * float(exp.abs().max())


@pytest.mark.parametrize('cg,hw', [(4, (13, 18)), (8, (21, 37)), (16, (50, 84)), (32, (25, 42))])
def test_grouped_conv3x3_bf16_mode_is_bf16_rounded_operands(cg, hw):
    """conv_math = 'bf16' (cfg[4]): the stride-1 grouped conv rounds both operands to bf16 and
    accumulates in fp32 — checked against an fp64 grouped convolution of the ROUNDED operands."""
    rs = np.random.RandomState(cg * 3 + hw[1])
    groups = 16
    C = cg * groups
    x = rs.randn(2, hw[0], hw[1], C).astype(np.float32)
    w = (rs.randn(C, cg, 3, 3) * 0.2).astype(np.float32)
    b = rs.randn(C).astype(np.float32)
    xr = torch.from_numpy(x).bfloat16().double()
    wr = torch.from_numpy(w).bfloat16().double()
    exp = F.relu(F.conv2d(xr.permute(0, 3, 1, 2), wr, torch.from_numpy(b).double(), padding=1,
                          groups=groups))
    scale = float(F.conv2d(xr.abs().permute(0, 3, 1, 2), wr.abs(), padding=1, groups=groups).max())
    wk = torch.from_numpy(np.ascontiguousarray(w.transpose(0, 2, 3, 1))).to(DEV)
    prev = BF.set_conv_math('bf16')
    try:
        got = BF.grouped_conv3x3_nhwc(torch.from_numpy(x).to(DEV), wk, torch.from_numpy(b).to(DEV),
                                      groups, stride=1, relu=True)
    finally:
        BF.set_conv_math(prev)
    assert float((got.permute(0, 3, 1, 2).cpu().double() - exp).abs().max()) <= 2e-6 * scale
    # and it IS the rounded arithmetic: the fp32 kernel's result differs by the bf16 rounding error
    ref32 = BF.grouped_conv3x3_nhwc(torch.from_numpy(x).to(DEV), wk, torch.from_numpy(b).to(DEV), groups,
                                    stride=1, relu=True)
    assert float((got - ref32).abs().max()) > 1e-4 * scale


@pytest.mark.parametrize('cg,stride,hw', [(4, 1, (13, 18)), (8, 2, (13, 18)), (16, 1, (12, 17)),
                                          (32, 2, (14, 20)), (32, 1, (9, 11)), (4, 2, (16, 16))])
def test_grouped_conv3x3_backward_vs_torch_autograd(cg, stride, hw):
    """dx / dw / db of the grouped 3x3 conv + ReLU (ResNeXt conv2 under selectp = 0,
    resnext.py:47-57) against torch-CPU fp64 autograd of ``F.conv2d(groups=...)``."""
    rs = np.random.RandomState(cg * 10 + stride)
    groups = 8
    C = cg * groups
    H, W = hw
    x = rs.randn(2, H, W, C).astype(np.float32)
    w = (rs.randn(C, cg, 3, 3) * 0.2).astype(np.float32)
    b = rs.randn(C).astype(np.float32)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).double().requires_grad_(True)
    wt = torch.from_numpy(w).double().requires_grad_(True)
    bt = torch.from_numpy(b).double().requires_grad_(True)
    yt = F.relu(F.conv2d(xt, wt, bt, stride=stride, padding=1, groups=groups))
    gy = rs.randn(*yt.shape).astype(np.float32)
    yt.backward(torch.from_numpy(gy).double())
    xd = torch.from_numpy(x).to(DEV).requires_grad_(True)
    wd = torch.from_numpy(np.ascontiguousarray(w.transpose(0, 2, 3, 1))).to(DEV).requires_grad_(True)
    bd = torch.from_numpy(b).to(DEV).requires_grad_(True)
    yd = BF.grouped_conv3x3_nhwc(xd, wd, bd, groups, stride=stride, relu=True)
    assert float((yd.permute(0, 3, 1, 2).detach().cpu().double() - yt.detach()).abs().max()) < 1e-4
    yd.backward(torch.from_numpy(np.ascontiguousarray(gy.transpose(0, 2, 3, 1))).to(DEV))
    for got, exp in ((xd.grad.permute(0, 3, 1, 2), xt.grad), (wd.grad.permute(0, 3, 1, 2), wt.grad),
                     (bd.grad, bt.grad)):
        assert float((got.cpu().double() - exp).abs().max()) < 2e-5 * float(exp.abs().max())
    # weight gradient is bitwise reproducible (fixed-order chunk reduction)
    xd2 = torch.from_numpy(x).to(DEV)
    wd2 = wd.detach().clone().requires_grad_(True)
    BF.grouped_conv3x3_nhwc(xd2, wd2, None, groups, stride=stride, relu=False).sum().backward()
    wd3 = wd.detach().clone().requires_grad_(True)
    BF.grouped_conv3x3_nhwc(xd2, wd3, None, groups, stride=stride, relu=False).sum().backward()
    assert torch.equal(wd2.grad, wd3.grad)


def test_maxpool_backward_and_trainable_stem_vs_torch():
    """``frozen_stages = 0`` (resnet.py:483-494): max-pool backward (first maximum of a window, the
    element torch routes to — ties included) and the stem conv's weight / BN gradients."""
    rs = np.random.RandomState(4)
    x = rs.randn(2, 17, 23, 8).astype(np.float32)
    x[0, 3:6, 4:9] = 1.5                                   # plateaus: ties inside windows
    x = np.round(x * 4) / 4                                # many exact ties
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).double().requires_grad_(True)
    yt = F.max_pool2d(xt, 3, 2, 1)
    gy = rs.randn(*yt.shape).astype(np.float32)
    yt.backward(torch.from_numpy(gy).double())
    xd = torch.from_numpy(x).to(DEV).requires_grad_(True)
    yd = BF.maxpool3x3s2_nhwc(xd)
    np.testing.assert_array_equal(yd.detach().permute(0, 3, 1, 2).cpu().numpy(), yt.detach().float().numpy())
    yd.backward(torch.from_numpy(np.ascontiguousarray(gy.transpose(0, 2, 3, 1))).to(DEV))
    np.testing.assert_allclose(xd.grad.permute(0, 3, 1, 2).cpu().numpy(), xt.grad.float().numpy(),
                               rtol=0, atol=1e-6)
    # whole stem + layer1 with frozen_stages = 0 vs torch-CPU autograd of the reference arithmetic
    torch.manual_seed(1)
    m = bgs.build_backbone(dict(type='ResNet', depth=50, num_stages=4, out_indices=(0,),
                                frozen_stages=-1, style='pytorch'))
    m.init_weights()
    from tests.test_gpu_detector import randomize_bn
    randomize_bn(m)
    for blk in m.layer1:
        torch.nn.init.constant_(blk.bn3.weight, 0.5)
    m.train()
    img = torch.randn(1, 3, 64, 96)
    ref = {k: v.detach().clone().double().requires_grad_(True) for k, v in m.named_parameters()}
    bufs = {k: v.detach().clone().double() for k, v in m.named_buffers()}

    def cbn(t, pre, bnp, stride, pad, relu):
        t = F.conv2d(t, ref[pre + '.weight'], None, stride, pad)
        t = F.batch_norm(t, bufs[bnp + '.running_mean'], bufs[bnp + '.running_var'],
                         ref[bnp + '.weight'], ref[bnp + '.bias'], False, 0., 1e-5)
        return F.relu(t) if relu else t
    t = cbn(img.double(), 'conv1', 'bn1', 2, 3, True)
    t = F.max_pool2d(t, 3, 2, 1)
    for i in range(3):
        p = 'layer1.%d.' % i
        idt = t if i else cbn(t, p + 'downsample.0', p + 'downsample.1', 1, 0, False)
        o = cbn(t, p + 'conv1', p + 'bn1', 1, 0, True)
        o = cbn(o, p + 'conv2', p + 'bn2', 1, 1, True)
        o = cbn(o, p + 'conv3', p + 'bn3', 1, 0, False)
        t = F.relu(o + idt)
    g = torch.randn(t.shape, dtype=torch.float64)
    t.backward(g)
    m.to(DEV)
    (out,) = m(img.to(DEV))
    assert float((out.permute(0, 3, 1, 2).detach().cpu().double() - t.detach()).abs().max()) < \
        1e-4 * float(t.detach().abs().max())
    out.backward(g.permute(0, 2, 3, 1).contiguous().float().to(DEV))
    params = dict(m.named_parameters())
    for k in ('conv1.weight', 'bn1.weight', 'bn1.bias', 'layer1.0.conv2.weight', 'layer1.2.bn3.bias'):
        got, exp = params[k].grad.cpu().double(), ref[k].grad
        assert got is not None
        rel = float((got - exp).abs().max() / exp.abs().max().clamp(min=1e-12))
        assert rel < 2e-3, (k, rel)      # a ReLU flip at fp32 noise moves single entries


def test_resnext_stage_forward_vs_torch_cpu():
    from tests.test_gpu_detector import randomize_bn, ref_bottleneck, nchw, rel_err
    torch.manual_seed(0)
    m = bgs.build_backbone(dict(type='ResNeXt', depth=50, groups=64, base_width=4, num_stages=4,
                                out_indices=(0, 1, 2, 3), frozen_stages=4, style='pytorch'))
    m.init_weights()
    for b in m.modules():
        if hasattr(b, 'bn3'):
            torch.nn.init.constant_(b.bn3.weight, 0.5)
    randomize_bn(m)
    assert m.layer1[0].conv2.groups == 64 and tuple(m.layer1[0].conv2.weight.shape) == (256, 4, 3, 3)
    img = torch.randn(1, 3, 96, 128)
    with torch.no_grad():
        x = F.conv2d(img, m.conv1.weight, None, 2, 3)
        x = F.relu(F.batch_norm(x, m.bn1.running_mean, m.bn1.running_var, m.bn1.weight, m.bn1.bias,
                                False, 0., m.bn1.eps))
        x = F.max_pool2d(x, 3, 2, 1)
        exp = []
        for name in m.res_layers:
            for blk in getattr(m, name):
                def cbn(conv, bn, t, relu):
                    t = F.conv2d(t, conv.weight, None, conv.stride, conv.padding, 1, conv.groups)
                    t = F.batch_norm(t, bn.running_mean, bn.running_var, bn.weight, bn.bias, False,
                                     0., bn.eps)
                    return F.relu(t) if relu else t
                idt = x if blk.downsample is None else cbn(blk.downsample[0], blk.downsample[1], x, False)
                o = cbn(blk.conv1, blk.bn1, x, True)
                o = cbn(blk.conv2, blk.bn2, o, True)
                o = cbn(blk.conv3, blk.bn3, o, False)
                x = F.relu(o + idt)
            exp.append(x)
        m.to(DEV)
        got = m(img.to(DEV))
    for g, e in zip(got, exp):
        assert rel_err(nchw(g), e) < 1e-4


def _cascade(tmp_path, depth=50):
    paths = gs_tables.save_group_tables(str(tmp_path), *gs_tables.synthetic_group_tables())
    from tests.test_gpu_detector import _detector_cfg
    model, train_cfg = _detector_cfg(paths)
    head = model['bbox_head']
    head['reg_class_agnostic'] = True
    heads = []
    for stds in ([0.1, 0.1, 0.2, 0.2], [0.05, 0.05, 0.1, 0.1], [0.033, 0.033, 0.067, 0.067]):
        h = dict(head)
        h['gs_config'] = dict(head['gs_config'])
        h['target_stds'] = stds
        heads.append(h)
    model.update(type='CascadeRCNN', num_stages=3, bbox_head=heads)
    rcnn = train_cfg['rcnn']
    stages = []
    for iou in (0.5, 0.6, 0.7):
        r = dict(rcnn)
        r['assigner'] = dict(rcnn['assigner'], pos_iou_thr=iou, neg_iou_thr=iou, min_pos_iou=iou)
        stages.append(r)
    train_cfg['rcnn'] = stages
    train_cfg['stage_loss_weights'] = [1, 0.5, 0.25]
    test_cfg = dict(rpn=dict(nms_across_levels=False, nms_pre=1000, nms_post=1000, max_num=1000,
                             nms_thr=0.7, min_bbox_size=0),
                    rcnn=dict(score_thr=0.0, nms=dict(type='nms', iou_thr=0.5), max_per_img=300),
                    keep_all_stages=False)
    return bgs.build_detector(to_config_dict(model), train_cfg=to_config_dict(train_cfg),
                              test_cfg=to_config_dict(test_cfg))


def test_cascade_rcnn_training_iteration_and_test(tmp_path):
    torch.manual_seed(0)
    model = _cascade(tmp_path).to(DEV)
    params = train.select_training_param(model, 3)               # the three fc_cls (cfg 5: selectp=3)
    assert len(params) == 6
    model.train()
    H, W = 320, 480
    img = torch.randn(2, 3, H, W, device=DEV)
    metas = [dict(img_shape=(H, W - 5, 3), pad_shape=(H, W, 3), ori_shape=(H, W - 5, 3),
                  scale_factor=1.0, flip=False)] * 2
    g = torch.Generator().manual_seed(3)
    gtb, gtl = [], []
    for _ in range(2):
        xy = torch.rand(10, 2, generator=g) * torch.tensor([W - 160., H - 160.])
        wh = torch.rand(10, 2, generator=g) * 120 + 30
        gtb.append(torch.cat([xy, xy + wh], 1).to(DEV))
        gtl.append(torch.randint(1, 1231, (10,), generator=g).to(DEV))
    losses = model(img, metas, return_loss=True, gt_bboxes=gtb, gt_labels=gtl)
    keys = set(losses.keys())
    for i in range(3):
        assert {'s%d.loss_cls_bin%d' % (i, b) for b in range(5)} <= keys
        assert 's%d.loss_bbox' % i in keys
    assert {'loss_rpn_cls', 'loss_rpn_bbox'} <= keys and len(keys) == 2 + 3 * 6
    # stage loss weights 1 / 0.5 / 0.25 (cascade_rcnn.py:248-250): bg bins at init ~ w * log 2-ish
    b0 = [float(losses['s%d.loss_cls_bin0' % i]) for i in range(3)]
    assert b0[0] > b0[1] > b0[2] > 0 and 0.2 < b0[1] / b0[0] < 0.8
    loss, _ = train.parse_losses(losses)
    assert torch.isfinite(loss)
    loss.backward()
    for h in model.bbox_head:
        assert h.fc_cls.weight.grad is not None and float(h.fc_cls.weight.grad.abs().sum()) > 0
        assert h.fc_reg.weight.grad is None
    # stage 2 / 3 re-sample from the refined boxes: the kept refined proposals exclude GT rows
    assert model._sampled_valid.shape == (2, 512) and model._sampled_is_gt.shape == (2, 512)
    model.eval()
    with torch.no_grad():
        for h in model.bbox_head:
            h.fc_cls.weight.mul_(30.0)
    res = model(img[:1], metas[:1], return_loss=False, rescale=False)
    assert len(res) == 1230 and sum(r.shape[0] for r in res) == 300
    with torch.no_grad():       # features handed in (train.TrunkPipeline(inference=True)): the same detections
        ahead = model.extract_feat(img[:1])
    res2 = model(img[:1], metas[:1], return_loss=False, rescale=False, feats=ahead)
    assert all(np.array_equal(a_, b_) for a_, b_ in zip(res, res2))


@pytest.mark.parametrize('agnostic', [True, False])
def test_refine_boxes_kernel_equals_regress_by_class(agnostic):
    """bgs_refine_boxes (one launch for all images) == BBoxHead.regress_by_class -> delta2bbox per image
    (bbox_head.py:210-239, transforms.py:34-111): class gather, decode, clip to each image's own shape —
    bit for bit (same operation order, separately rounded)."""
    from balancedgroupsoftmax_amd.box_ops import delta2bbox
    rs = np.random.RandomState(5 + agnostic)
    K, C = 300, 7
    shapes = [(97, 143, 3), (120, 101, 3)]
    img = rs.randint(0, 2, size=K).astype(np.float32)
    x1 = rs.uniform(-5, 120, K); y1 = rs.uniform(-5, 100, K)
    boxes = np.stack([x1, y1, x1 + rs.uniform(0, 60, K), y1 + rs.uniform(0, 60, K)], 1).astype(np.float32)
    rois = torch.from_numpy(np.concatenate([img[:, None], boxes], 1)).to(DEV)
    labels = torch.from_numpy(rs.randint(0, C, size=K).astype(np.int64)).to(DEV)
    pred = torch.from_numpy((rs.standard_normal((K, 4 if agnostic else 4 * C)) * 2).astype(np.float32)).to(DEV)
    means, stds = (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2)
    got = BF.refine_boxes(rois, labels, pred, shapes, means, stds)
    exp = torch.empty_like(got)
    for j, shp in enumerate(shapes):
        m = rois[:, 0] == j
        d = pred[m]
        if not agnostic:
            cols = (labels[m] * 4).view(-1, 1) + torch.arange(4, device=DEV).view(1, 4)
            d = torch.gather(d, 1, cols)
        exp[m] = delta2bbox(rois[m, 1:], d, means, stds, shp)
    assert torch.equal(got, exp)
