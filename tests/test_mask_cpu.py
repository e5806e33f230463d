"""CPU: mask-branch oracle (oracle/mask_oracle.py) against the executed reference head's golden
vectors, resize properties, and the host-side mirror (registry keys, state-dict layout, config)."""
import json
import os

import numpy as np
import pytest
import torch

import balancedgroupsoftmax_amd as bgs
from oracle import mask_oracle, ref_import
from tests.golden import make_golden_mask

GOLD = os.path.join(os.path.dirname(make_golden_mask.__file__), 'mask_head_golden.npz')


def golden():
    z = np.load(GOLD)
    return z, json.loads(bytes(z['__cases__']).decode())


@pytest.mark.parametrize('name', ['p6_c1231', 'p3_c11'])
def test_mask_cross_entropy_restatement_vs_executed_reference(name):
    z, cases = golden()
    case = [c for c in cases if c['name'] == name][0]
    _, _, targets = make_golden_mask.case_inputs(case)
    got = mask_oracle.mask_cross_entropy(z[name + '/gt_logits'], targets)
    assert abs(got - float(z[name + '/loss'][0])) < 2e-6


def test_resize_linear_u8_properties():
    rs = np.random.RandomState(0)
    img = (rs.rand(28, 28) > 0.5).astype(np.uint8)
    assert np.array_equal(mask_oracle.resize_linear_u8(img, (28, 28)), img)       # same size: copy
    for shape in [(5, 9), (60, 33), (1, 1), (1, 40)]:
        ones = np.ones(shape, np.uint8)
        assert mask_oracle.resize_linear_u8(ones, (28, 28)).min() == 1            # constants stay
        assert mask_oracle.resize_linear_u8(ones * 0, (28, 28)).max() == 0
    # exact 2x upsample of a vertical step edge: the edge lands between output columns 13 | 14
    step = np.zeros((14, 14), np.uint8)
    step[:, 7:] = 1
    up = mask_oracle.resize_linear_u8(step, (28, 28))
    assert up[:, :13].max() == 0 and up[:, 15:].min() == 1 and set(np.unique(up)) <= {0, 1}
    # output is always binary for binary input
    big = (rs.rand(200, 131) > 0.7).astype(np.uint8)
    assert set(np.unique(mask_oracle.resize_linear_u8(big, (28, 28)))) <= {0, 1}


def test_mask_target_single_crops_like_numpy_slicing():
    H, W = 64, 80
    gt = np.zeros((2, H, W), np.uint8)
    gt[0, 10:40, 20:60] = 1
    gt[1, :, :] = 1
    props = np.array([[20.7, 10.2, 59.9, 39.9],        # exactly the rectangle (int truncation)
                      [70.0, 50.0, 90.0, 70.0],        # sticks out of the bitmap: slicing truncates
                      [5.0, 5.0, 5.4, 5.2]], np.float32)    # 1x1 crop
    t = mask_oracle.mask_target_single(props, [0, 1, 0], gt, 28)
    assert t.shape == (3, 28, 28) and t.dtype == np.float32
    assert t[0].min() == 1.0 and t[1].min() == 1.0 and t[2].max() == 0.0


def test_fcn_mask_head_state_dict_matches_reference_module():
    if not ref_import.reference_available():
        pytest.skip('reference tree absent')
    ref_import.install_stubs()
    from mmdet.models.mask_heads.fcn_mask_head import FCNMaskHead as RefHead
    ref = RefHead(num_convs=4, in_channels=256, conv_out_channels=256, num_classes=1231)
    mine = bgs.build_head(dict(type='FCNMaskHead', num_convs=4, in_channels=256,
                               conv_out_channels=256, num_classes=1231,
                               loss_mask=dict(type='CrossEntropyLoss', use_mask=True,
                                              loss_weight=1.0)))
    a = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    b = {k: tuple(v.shape) for k, v in mine.state_dict().items()}
    assert a == b
    # the torch restatement over this module's parameter containers (oracle/tensor_forms.py) equals
    # the reference with the same parameters; the module itself refuses CPU tensors
    from oracle import tensor_forms
    with torch.no_grad():
        mask_oracle.fill_mask_head(ref.state_dict(), 7)
        mine.load_state_dict(ref.state_dict())
        x = torch.randn(2, 256, 14, 14)
        assert torch.allclose(tensor_forms.fcn_mask_forward(mine, x), ref(x), atol=1e-5)
        lab = torch.tensor([3, 1200])
        assert torch.allclose(tensor_forms.fcn_mask_forward(mine, x, labels=lab),
                              ref(x)[torch.arange(2), lab], atol=1e-5)
        with pytest.raises(RuntimeError, match='no CPU fallback'):
            mine(x)


def test_mask_rcnn_builds_from_reference_config(tmp_path):
    if not ref_import.reference_available():
        pytest.skip('reference tree absent')
    from balancedgroupsoftmax_amd import gs_tables
    cfg = bgs.Config.fromfile(os.path.join(ref_import.REFERENCE_ROOT,
                                           'configs/bags/gs_mask_rcnn_r50_fpn_1x_lvis.py'))
    counts = gs_tables.synthetic_instance_counts(1231, seed=0)
    paths = gs_tables.save_group_tables(str(tmp_path), *gs_tables.build_group_tables(counts))
    gs = cfg.model.bbox_head.gs_config
    gs.label2binlabel, gs.pred_slice, gs.fg_split = (paths['label2binlabel'], paths['pred_slice'],
                                                     paths['fg_split'])
    model = bgs.build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    assert type(model).__name__ == 'MaskRCNN' and model.with_mask
    assert model.mask_roi_extractor.out_size == 14 and not model.share_roi_extractor
    assert tuple(model.mask_head.conv_logits.weight.shape) == (1231, 256, 1, 1)
    assert cfg.train_cfg.rcnn.mask_size == 28 and cfg.test_cfg.rcnn.mask_thr_binary == 0.5


def test_resize_linear_f32_properties():
    """cv2.resize(float32, INTER_LINEAR) restatement: identity size returns the source, a constant image stays
    constant to within one rounding, upsampling a ramp is monotone, borders replicate."""
    rs = np.random.RandomState(3)
    src = rs.rand(28, 28).astype(np.float32)
    assert np.array_equal(mask_oracle.resize_linear_f32(src, (28, 28)), src)
    c = np.full((28, 28), 0.37, np.float32)
    for size in ((5, 9), (61, 17), (14, 14), (300, 211)):
        out = mask_oracle.resize_linear_f32(c, size)
        assert out.shape == (size[1], size[0]) and np.abs(out - np.float32(0.37)).max() <= 6e-8
    ramp = np.tile(np.linspace(0, 1, 28, dtype=np.float32), (28, 1))
    up = mask_oracle.resize_linear_f32(ramp, (113, 40))
    assert (np.diff(up, axis=1) >= -1e-7).all() and up[:, 0].max() == 0.0 and up[:, -1].min() == 1.0
    out = mask_oracle.resize_linear_f32(src, (56, 56))
    assert out[0, 0] == src[0, 0] and out[-1, -1] == src[-1, -1]


def test_seg_masks_dense_equals_the_executed_reference_get_seg_masks():
    """``oracle.mask_oracle.seg_masks_dense`` (the checker of ``bgs_mask_paste_u8``) against the EXECUTED
    ``FCNMaskHead.get_seg_masks`` (fcn_mask_head.py:125-181) with ``mmcv.imresize`` := the oracle's float32 resize
    and ``mask_util.encode`` := identity: box truncation, w / h, the class channel, threshold, placement and the
    per-class bucketing are the reference's own code; only the cv2 resize is the restatement."""
    if not ref_import.reference_available():
        pytest.skip('reference tree absent')
    ref_import.install_stubs()
    import mmdet.models.mask_heads.fcn_mask_head as ref_mod
    from balancedgroupsoftmax_amd.config import to_config_dict
    rs = np.random.RandomState(11)
    n, C, S = 7, 6, 28
    ref = ref_mod.FCNMaskHead(num_convs=1, in_channels=8, conv_out_channels=8, num_classes=C)
    logits = torch.from_numpy((rs.standard_normal((n, C, S, S)) * 2).astype(np.float32))
    ori_shape, scale = (97, 131, 3), 1.37
    boxes = np.zeros((n, 5), np.float32)
    for i in range(n):
        x1, y1 = rs.rand() * 100 * scale, rs.rand() * 70 * scale
        boxes[i] = [x1, y1, x1 + 2 + rs.rand() * 60, y1 + 2 + rs.rand() * 50, rs.rand()]
    boxes[0, :4] = [0, 0, 130.9 * scale, 96.9 * scale]            # the whole image
    boxes[1, :4] = [50.2, 40.7, 50.9, 41.1]                        # a 1 x 1 box
    labels = torch.from_numpy(rs.randint(0, C - 1, n).astype(np.int64))
    cfg = to_config_dict(dict(mask_thr_binary=0.5))
    saved = (ref_mod.mmcv.imresize, ref_mod.mask_util.encode)
    ref_mod.mmcv.imresize = lambda img, size: mask_oracle.resize_linear_f32(img, size)
    ref_mod.mask_util.encode = lambda arr: [np.array(arr[:, :, 0])]
    try:
        for rescale in (True, False):
            # (rescale: boxes live in the network's scale and the masks go to ori_shape; else both in the network's)
            bx = boxes.copy()
            if not rescale:
                bx[:, :4] = np.minimum(bx[:, :4], [[round(131 * scale) - 1, round(97 * scale) - 1] * 2])
            else:
                bx[:, :4] = np.minimum(bx[:, :4], [[131 * scale - 0.01, 97 * scale - 0.01] * 2])
            segms = ref.get_seg_masks(logits, torch.from_numpy(bx), labels, cfg, ori_shape, scale, rescale)
            probs = torch.sigmoid(logits)[torch.arange(n), labels + 1].numpy()
            if rescale:
                ih, iw, sf = 97, 131, scale
            else:
                ih, iw, sf = int(np.round(97 * scale)), int(np.round(131 * scale)), 1.0
            dense = mask_oracle.seg_masks_dense(probs, bx, sf, 0.5, ih, iw)
            seen = [0] * (C - 1)
            for i in range(n):
                lab = int(labels[i])
                m = segms[lab][seen[lab]]
                seen[lab] += 1
                assert m.shape == (ih, iw) and np.array_equal(m, dense[i]), (rescale, i)
            assert dense.sum() > 0
    finally:
        ref_mod.mmcv.imresize, ref_mod.mask_util.encode = saved
