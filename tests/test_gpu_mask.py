"""GPU: mask branch through the C ABI — FCNMaskHead against the executed reference's golden
vectors, bgs_mask_target against the numpy restatement (bit-exact), and a Mask R-CNN iteration."""
import json
import os

import numpy as np
import pytest
import torch

import balancedgroupsoftmax_amd as bgs
from balancedgroupsoftmax_amd import functional as BF
from balancedgroupsoftmax_amd import gs_tables, train
from balancedgroupsoftmax_amd.config import to_config_dict
from oracle import mask_oracle
from tests.golden import make_golden_mask

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
GOLD = os.path.join(os.path.dirname(make_golden_mask.__file__), 'mask_head_golden.npz')


def _head(C):
    return bgs.build_head(dict(type='FCNMaskHead', num_convs=4, in_channels=256,
                               conv_out_channels=256, num_classes=C,
                               loss_mask=dict(type='CrossEntropyLoss', use_mask=True,
                                              loss_weight=1.0)))


@pytest.mark.parametrize('name', ['p6_c1231', 'p3_c11'])
def test_fcn_mask_head_vs_executed_reference_golden(name):
    """forward (GT channel only), loss and gradients == the reference FCNMaskHead run on CPU with
    the same seeded parameters (full 1231-channel conv_logits + gather + BCE mean)."""
    z = np.load(GOLD)
    case = [c for c in json.loads(bytes(z['__cases__']).decode()) if c['name'] == name][0]
    head = _head(case['C'])
    with torch.no_grad():
        mask_oracle.fill_mask_head(head.state_dict(), case['seed'] + 1000)
    head.to(DEV)
    feats, labels, targets = make_golden_mask.case_inputs(case)
    x = torch.from_numpy(feats).permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    lab = torch.from_numpy(labels).to(DEV)
    f = head.features(x, nhwc=True)
    assert tuple(f.shape) == (case['P'], 28, 28, 256)
    logits = head(x.detach(), labels=lab, nhwc=True)
    exp = z[name + '/gt_logits']
    assert np.abs(logits.cpu().numpy() - exp).max() < 1e-4 * max(1.0, np.abs(exp).max())
    loss = head.loss_from_features(f, torch.from_numpy(targets).to(DEV), lab)['loss_mask']
    assert abs(float(loss) - float(z[name + '/loss'][0])) < 1e-5
    loss.sum().backward()
    dx = x.grad.permute(0, 3, 1, 2)[:, :, ::5, ::3].cpu().numpy()

    def close(a, b, tol=2e-4):
        # gradients pass through 5 ReLUs: a pre-activation within fp32 noise of zero takes the
        # other branch than in the torch-CPU run, which moves the 3x3 neighbourhoods behind it
        # (measured: 1.8 % of dx entries off by up to 0.3 %, everything else < 1e-6); a wrong
        # kernel would be off by O(1)
        rel = np.abs(a - b) / max(np.abs(b).max(), 1e-12)
        return (rel < tol).mean() > 0.9 and rel.max() < 1e-2
    assert close(dx, z[name + '/dx'])
    dw = head.conv_logits.weight.grad.view(case['C'], 256)[lab].cpu().numpy()
    assert close(dw, z[name + '/dw_rows'].reshape(case['P'], 256))
    assert close(head.convs[0].conv.weight.grad[::16, ::16].cpu().numpy(), z[name + '/dconv0_w'])
    assert close(head.upsample.bias.grad.cpu().numpy(), z[name + '/dup_b'])
    # rows of conv_logits that no RoI uses receive exactly zero
    untouched = np.setdiff1d(np.arange(case['C']), labels)
    assert float(head.conv_logits.weight.grad.view(case['C'], 256)[untouched].abs().max()) == 0.0


def test_full_logits_path_agrees_with_gt_channel_path():
    head = _head(21).to(DEV)
    torch.manual_seed(0)
    head.init_weights()
    x = torch.randn(5, 14, 14, 256, device=DEV)
    lab = torch.tensor([1, 20, 7, 7, 3], device=DEV)
    with torch.no_grad():
        full = head(x, nhwc=True)                                  # [P, 21, 28, 28]
        one = head(x, labels=lab, nhwc=True)
    assert tuple(full.shape) == (5, 21, 28, 28)
    assert torch.allclose(full[torch.arange(5), lab], one, rtol=1e-4, atol=1e-5)
    # mask_cross_entropy on the full logits (cross_entropy_loss.py:54-61, torch restatement) == fused
    from oracle import tensor_forms as TF
    tgt = (torch.rand(5, 28, 28, device=DEV) > 0.5).float()
    a = TF.mask_bce(full, tgt, lab)
    b = head.loss_from_features(head.features(x), tgt, lab)['loss_mask']
    assert torch.allclose(a.reshape(()), b.reshape(()), rtol=1e-5)
    # padding slots are excluded from the mean
    valid = torch.tensor([1, 1, 0, 1, 0], device=DEV, dtype=torch.bool)
    c = head.loss_from_features(head.features(x), tgt, lab, valid)['loss_mask']
    d = TF.mask_bce(full[valid], tgt[valid], lab[valid])
    assert torch.allclose(c.reshape(()), d.reshape(()), rtol=1e-5)
    with pytest.raises(NotImplementedError, match='loss_from_features'):
        head.loss(full, tgt, lab)          # the K-channel entry point is refused, not emulated


def test_mask_target_kernel_bit_exact_vs_oracle():
    rs = np.random.RandomState(4)
    H, W = 96, 128
    boxes = [np.array([[10, 12, 90, 70], [40, 5, 120, 90], [0, 0, 27, 27]], np.float32),
             np.array([[60, 30, 100, 80]], np.float32)]
    masks = [mask_oracle.make_gt_masks(3, H, W, boxes[0], 1), mask_oracle.make_gt_masks(1, H, W, boxes[1], 2)]
    masks[0][2, :28, :28] = (rs.rand(28, 28) > 0.5)
    P = 64
    img = rs.randint(0, 2, P)
    gt = np.array([rs.randint(0, 3) if i == 0 else 0 for i in img], np.int32)
    ctr = rs.uniform(0, 1, (P, 2)) * [W, H]
    size = np.exp(rs.uniform(np.log(1.5), np.log(150), (P, 2)))
    b = np.concatenate([ctr - size / 2, ctr + size / 2], 1)
    b[:, [0, 2]] = np.clip(b[:, [0, 2]], 0, W - 1)
    b[:, [1, 3]] = np.clip(b[:, [1, 3]], 0, H - 1)
    b[0] = [0, 0, 27.9, 27.4]                     # same-size crop: cv2 copies
    img[0], gt[0] = 0, 2
    b[1] = [100.3, 80.2, 140.0, 120.0]            # beyond the bitmap: numpy slicing truncates
    b[2] = [5.2, 5.9, 5.4, 5.95]                  # 1x1 crop
    rois = np.concatenate([img[:, None], b], 1).astype(np.float32)
    valid = np.ones(P, bool)
    valid[5] = False
    got = BF.mask_target([torch.from_numpy(m).to(DEV) for m in masks], torch.from_numpy(rois).to(DEV),
                         torch.from_numpy(gt).to(DEV), torch.from_numpy(valid).to(DEV), 28).cpu().numpy()
    for n in range(2):
        idx = np.nonzero((img == n) & valid)[0]
        exp = mask_oracle.mask_target_single(rois[idx, 1:], gt[idx], masks[n], 28)
        assert np.array_equal(got[idx], exp), n
    assert got[5].max() == 0.0
    assert set(np.unique(got)) <= {0.0, 1.0} and 0.05 < got.mean() < 0.95


def test_mask_target_kernel_is_bounded_by_an_independent_float_bilinear_resize():
    """A second, INDEPENDENT implementation bounds ``bgs_mask_target`` (and the fixed-point restatement of OpenCV it
    is bit-exact with, oracle/mask_oracle.py — "parity unpinned": cv2 itself is not installed): the same crops
    resized by torch's float ``F.interpolate(bilinear, align_corners=False)`` — the half-pixel mapping of
    cv2.INTER_LINEAR, mmdet/core/mask/mask_target.py:31.  OpenCV's 8-bit path computes
    ``(floor(4 b0 r0) + floor(4 b1 r1) + 2) >> 2`` on a 0/1 bitmap (r = the horizontally interpolated rows, b their
    weights: resize.cpp's FixedPtCast<int, uchar, 22> after two truncating shifts), so with v the exact bilinear
    value it yields 0 whenever v < 1/2, 1 whenever v >= 3/4, and either in between — NOT round(v).  The float
    resize must agree outside that band (a margin of 4 / 2048 for the 11-bit coefficients), on ellipses and on
    noise; the band itself is reported."""
    import torch.nn.functional as F
    rs = np.random.RandomState(12)
    H, W, P = 160, 224, 96
    boxes = np.array([[10, 12, 150, 130], [40, 5, 200, 150], [0, 0, 60, 60], [100, 40, 220, 158]], np.float32)
    masks = mask_oracle.make_gt_masks(4, H, W, boxes, 3)
    masks[2, :60, :60] = (rs.rand(60, 60) > 0.5)               # noise: the worst case for an interpolation
    gt = rs.randint(0, 4, P).astype(np.int32)
    ctr = rs.uniform(0, 1, (P, 2)) * [W, H]
    size = np.exp(rs.uniform(np.log(3), np.log(200), (P, 2)))
    b = np.concatenate([ctr - size / 2, ctr + size / 2], 1)
    b[:, [0, 2]] = np.clip(b[:, [0, 2]], 0, W - 1)
    b[:, [1, 3]] = np.clip(b[:, [1, 3]], 0, H - 1)
    rois = np.concatenate([np.zeros((P, 1)), b], 1).astype(np.float32)
    got = BF.mask_target([torch.from_numpy(masks).to(DEV)], torch.from_numpy(rois).to(DEV),
                         torch.from_numpy(gt).to(DEV), torch.ones(P, dtype=torch.bool, device=DEV), 28).cpu().numpy()
    eps = 4.0 / 2048
    band = total = ones_in_band = 0
    for i in range(P):
        x1, y1, x2, y2 = rois[i, 1:].astype(np.int32)
        w, h = max(x2 - x1 + 1, 1), max(y2 - y1 + 1, 1)
        crop = masks[gt[i]][y1:y1 + h, x1:x1 + w]
        if crop.shape == (28, 28):
            assert np.array_equal(got[i], crop.astype(np.float32))          # same size: cv2 copies
            continue
        ref = F.interpolate(torch.from_numpy(crop.astype(np.float64))[None, None], size=(28, 28), mode='bilinear',
                            align_corners=False)[0, 0].numpy()
        assert (got[i][ref < 0.5 - eps] == 0.0).all(), i
        assert (got[i][ref >= 0.75 + eps] == 1.0).all(), i
        amb = (ref >= 0.5 - eps) & (ref < 0.75 + eps)
        band += int(amb.sum())
        ones_in_band += int(got[i][amb].sum())
        total += ref.size
    print('float-bilinear bound: every pixel outside [1/2, 3/4) agrees; %d of %d pixels lie in the band (%d of them 1)'
          % (band, total, ones_in_band))
    assert band < 0.2 * total and 0 < ones_in_band < band


def _mask_rcnn(tmp_path):
    paths = gs_tables.save_group_tables(str(tmp_path), *gs_tables.synthetic_group_tables())
    from tests.test_gpu_detector import _detector_cfg
    model, train_cfg = _detector_cfg(paths)
    model['type'] = 'MaskRCNN'
    model['mask_roi_extractor'] = dict(type='SingleRoIExtractor',
                                       roi_layer=dict(type='RoIAlign', out_size=14, sample_num=2),
                                       out_channels=256, featmap_strides=[4, 8, 16, 32])
    model['mask_head'] = dict(type='FCNMaskHead', num_convs=4, in_channels=256,
                              conv_out_channels=256, num_classes=1231,
                              loss_mask=dict(type='CrossEntropyLoss', use_mask=True, loss_weight=1.0))
    train_cfg['rcnn']['mask_size'] = 28
    test_cfg = dict(rpn=dict(nms_across_levels=False, nms_pre=1000, nms_post=1000, max_num=1000,
                             nms_thr=0.7, min_bbox_size=0),
                    rcnn=dict(score_thr=0.0, nms=dict(type='nms', iou_thr=0.5), max_per_img=100,
                              mask_thr_binary=0.5))
    return bgs.build_detector(to_config_dict(model), train_cfg=to_config_dict(train_cfg),
                              test_cfg=to_config_dict(test_cfg))


@pytest.mark.parametrize('case', [
    # (K, S, img_h, img_w, scale_factor): total bytes not a multiple of 4, odd widths, a scale that moves the boxes
    (9, 28, 97, 131, 1.37), (5, 28, 800, 1333, 1.0), (3, 14, 61, 67, 0.5), (1, 28, 33, 35, 2.0)],
    ids=lambda c: 'x'.join(str(v) for v in c))
def test_mask_paste_kernel_equals_the_oracle_of_get_seg_masks(case):
    """``bgs_mask_paste_u8`` (FCNMaskHead.get_seg_masks, fcn_mask_head.py:156-176, without the RLE step: int
    truncation of box / scale_factor, cv2's float32 INTER_LINEAR resize of the 28 x 28 probabilities to the box,
    > thr, paste into a zero image) == ``oracle.mask_oracle.seg_masks_dense`` — pinned against the EXECUTED
    reference method in tests/test_mask_cpu.py — byte for byte: the same float32 operations in the same order;
    boxes that leave the image are clipped, degenerate boxes become 1 x 1, every output byte is written."""
    K, S, ih, iw, sf = case
    rs = np.random.RandomState(K * 7 + iw)
    probs = (1.0 / (1.0 + np.exp(-rs.standard_normal((K, S, S)) * 2))).astype(np.float32)
    boxes = np.zeros((K, 5), np.float32)
    for i in range(K):
        x1, y1 = rs.rand() * iw * sf * 0.8, rs.rand() * ih * sf * 0.8
        boxes[i] = [x1, y1, x1 + 1 + rs.rand() * iw * sf * 0.5, y1 + 1 + rs.rand() * ih * sf * 0.5, rs.rand()]
    boxes[0, :4] = [0, 0, (iw - 0.1) * sf, (ih - 0.1) * sf]              # the whole image
    if K > 1:
        boxes[1, :4] = [10.2 * sf, 7.7 * sf, 10.3 * sf, 7.9 * sf]         # 1 x 1 after truncation
    if K > 2:
        boxes[2, :4] = [(iw - 9) * sf, (ih - 5) * sf, (iw + 14) * sf, (ih + 8) * sf]     # leaves the image: clipped
    if K > 3:
        boxes[3, :4] = [20 * sf, 30 * sf, (20 + S - 1) * sf + 0.2, (30 + S - 1) * sf + 0.2]   # S x S: no resize
    if K > 4:
        boxes[4, :4] = [5 * sf, 5 * sf, 3 * sf, 4 * sf]                   # x2 < x1: w = h = 1
    out = torch.full((K, ih, iw), 7, dtype=torch.uint8, device=DEV)
    got = BF.mask_paste(torch.from_numpy(probs).to(DEV), torch.from_numpy(boxes).to(DEV), sf, 0.5, ih, iw)
    exp, margin = mask_oracle.seg_masks_dense(probs, boxes, sf, 0.5, ih, iw, return_margin=True)
    g = got.cpu().numpy()
    assert g.shape == exp.shape and g.dtype == np.uint8 and set(np.unique(g).tolist()) <= {0, 1}
    diff = g != exp
    assert not diff.any(), (int(diff.sum()), float(margin[diff].max()))
    assert exp[0].sum() > 0 and exp.sum() < exp.size
    del out


def test_fcn_mask_head_get_seg_masks_structure_and_4d_logits():
    """The reference signature: ``[n, num_classes, S, S]`` logits in, ``cls_segms`` (per class, detection order)
    out; == the oracle on the sigmoid of each detection's channel; ``encode`` receives host numpy masks."""
    C, n, S = 6, 8, 28
    head = _head(C).to(DEV)
    rs = np.random.RandomState(5)
    logits = torch.from_numpy((rs.standard_normal((n, C, S, S)) * 2).astype(np.float32)).to(DEV)
    labels = torch.from_numpy(rs.randint(0, C - 1, n).astype(np.int64)).to(DEV)
    boxes = np.zeros((n, 5), np.float32)
    for i in range(n):
        x1, y1 = rs.rand() * 150, rs.rand() * 100
        boxes[i] = [x1, y1, x1 + 3 + rs.rand() * 80, y1 + 3 + rs.rand() * 60, rs.rand()]
    cfg = to_config_dict(dict(mask_thr_binary=0.5))
    ori_shape, sf = (120, 180, 3), 1.5
    probs = torch.sigmoid(logits)[torch.arange(n), labels + 1].cpu().numpy()
    for rescale in (True, False):
        segms = head.get_seg_masks(logits, torch.from_numpy(boxes).to(DEV), labels, cfg, ori_shape, sf, rescale)
        ih, iw, s = (120, 180, sf) if rescale else (int(np.round(120 * sf)), int(np.round(180 * sf)), 1.0)
        exp = mask_oracle.seg_masks_dense(probs, boxes, s, 0.5, ih, iw)
        assert len(segms) == C - 1 and sum(len(c) for c in segms) == n
        seen = [0] * (C - 1)
        for i, lab in enumerate(labels.cpu().tolist()):
            m = segms[lab][seen[lab]]
            seen[lab] += 1
            assert np.array_equal(m.cpu().numpy(), exp[i]), (rescale, i)
    enc = head.get_seg_masks(logits, torch.from_numpy(boxes).to(DEV), labels, cfg, ori_shape, sf, True,
                             encode=lambda m: (type(m).__name__, int(m.sum())))
    assert all(e[0] == 'ndarray' for c in enc for e in c)


@pytest.mark.parametrize('selectp', [1, 0])
def test_mask_rcnn_training_iteration(tmp_path, selectp):
    torch.manual_seed(0)
    model = _mask_rcnn(tmp_path).to(DEV)
    params = train.select_training_param(model, selectp)
    model.train()
    H, W = 320, 480
    img = torch.randn(2, 3, H, W, device=DEV)
    metas = [dict(img_shape=(H, W - 5, 3), pad_shape=(H, W, 3), ori_shape=(H, W - 5, 3),
                  scale_factor=1.0, flip=False)] * 2
    g = torch.Generator().manual_seed(3)
    gt_bboxes, gt_labels, gt_masks = [], [], []
    for i in range(2):
        xy = torch.rand(8, 2, generator=g) * torch.tensor([W - 160., H - 160.])
        wh = torch.rand(8, 2, generator=g) * 120 + 30
        bb = torch.cat([xy, xy + wh], 1)
        gt_bboxes.append(bb.to(DEV))
        gt_labels.append(torch.randint(1, 1231, (8,), generator=g).to(DEV))
        gt_masks.append(torch.from_numpy(mask_oracle.make_gt_masks(8, H, W, bb.numpy(), 10 + i)).to(DEV))
    losses = model(img, metas, return_loss=True, gt_bboxes=gt_bboxes, gt_labels=gt_labels,
                   gt_masks=gt_masks)
    assert 'loss_mask' in losses and torch.isfinite(losses['loss_mask']).all()
    assert 0.3 < float(losses['loss_mask']) < 3.0             # ~log 2 at init
    loss, _ = train.parse_losses(losses)
    opt = train.build_optimizer(params, dict(type='SGD', lr=0.01, momentum=0.9, weight_decay=1e-4))
    train.DistOptimizerStep(params, opt, dict(max_norm=35, norm_type=2), world_size=1)(loss)
    if selectp == 0:
        for n in ('mask_head.convs.0.conv.weight', 'mask_head.upsample.weight',
                  'mask_head.conv_logits.weight', 'neck.fpn_convs.0.conv.weight'):
            gr = dict(model.named_parameters())[n].grad
            assert gr is not None and torch.isfinite(gr).all() and float(gr.abs().sum()) > 0, n
    else:
        assert model.mask_head.conv_logits.weight.grad is None
    # test-time: boxes + per-detection mask probabilities
    model.eval()
    with torch.no_grad():
        model.bbox_head.fc_cls.weight.mul_(30.0)
    bbox_results, mask_probs = model(img[:1], metas[:1], return_loss=False, rescale=False)
    k = sum(r.shape[0] for r in bbox_results)
    assert len(bbox_results) == 1230 and tuple(mask_probs.shape) == (k, 28, 28)
    assert k == 100
    assert float(mask_probs.min()) >= 0.0 and float(mask_probs.max()) <= 1.0
