"""CPU: cascade / ResNeXt host mirror against the reference's own classes and config."""
import os

import pytest
import torch

import balancedgroupsoftmax_amd as bgs
from balancedgroupsoftmax_amd import gs_tables
from oracle import ref_import

needs_ref = pytest.mark.skipif(not ref_import.reference_available(), reason='reference tree absent')


@needs_ref
def test_cascade_x101_builds_from_reference_config_with_reference_state_dict_layout(tmp_path):
    cfg = bgs.Config.fromfile(os.path.join(ref_import.REFERENCE_ROOT,
                                           'configs/bags/gs_cascade_rcnn_x101_64x4d_fpn_1x_lvis.py'))
    paths = gs_tables.save_group_tables(str(tmp_path), *gs_tables.synthetic_group_tables())
    for h in cfg.model.bbox_head:
        h.gs_config.label2binlabel, h.gs_config.pred_slice, h.gs_config.fg_split = (
            paths['label2binlabel'], paths['pred_slice'], paths['fg_split'])
    model = bgs.build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    assert type(model).__name__ == 'CascadeRCNN' and model.num_stages == 3
    assert type(model.backbone).__name__ == 'ResNeXt' and model.backbone.groups == 64
    assert [tuple(h.target_stds) for h in model.bbox_head] == \
        [(0.1, 0.1, 0.2, 0.2), (0.05, 0.05, 0.1, 0.1), (0.033, 0.033, 0.067, 0.067)]
    assert all(h.reg_class_agnostic and tuple(h.fc_reg.weight.shape) == (4, 1024)
               for h in model.bbox_head)
    assert cfg.train_cfg.stage_loss_weights == [1, 0.5, 0.25] and cfg.selectp == 3
    from balancedgroupsoftmax_amd import train
    params = train.select_training_param(model, cfg.selectp)        # cascade: the three fc_cls
    assert len(params) == 6 and sum(p.numel() for p in params) == 3 * (1236 * 1024 + 1236)
    # parameter names / shapes of the ResNeXt trunk == the reference's module
    ref_import.install_stubs()
    from mmdet.models.backbones.resnext import ResNeXt as RefResNeXt
    ref = RefResNeXt(depth=101, groups=64, base_width=4, num_stages=4, out_indices=(0, 1, 2, 3),
                     frozen_stages=1, style='pytorch')
    a = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    b = {k: tuple(v.shape) for k, v in model.backbone.state_dict().items()}
    assert a == b


@needs_ref
def test_refine_bboxes_equals_reference_head():
    """BBoxHead.refine_bboxes / regress_by_class (bbox_head.py:169-239) on CPU tensors."""
    ref_import.install_stubs()
    from mmdet.models.bbox_heads.bbox_head import BBoxHead as RefHead
    kw = dict(with_avg_pool=False, with_cls=True, with_reg=True, roi_feat_size=7, in_channels=16,
              num_classes=5, target_means=[0., 0., 0., 0.], target_stds=[0.1, 0.1, 0.2, 0.2])
    for agnostic in (True, False):
        ref = RefHead(reg_class_agnostic=agnostic, **kw)
        mine = bgs.build_head(dict(type='BBoxHead', reg_class_agnostic=agnostic, **kw))
        g = torch.Generator().manual_seed(0)
        n = 12
        xy = torch.rand(n, 2, generator=g) * 100
        rois = torch.cat([torch.tensor([0.] * 7 + [1.] * 5)[:, None], xy,
                          xy + torch.rand(n, 2, generator=g) * 80 + 4], 1)
        labels = torch.randint(0, 5, (n,), generator=g)
        bbox_pred = torch.randn(n, 4 if agnostic else 20, generator=g) * 0.5
        pos_is_gts = [torch.tensor([1, 0, 0], dtype=torch.uint8), torch.tensor([1, 1], dtype=torch.uint8)]
        metas = [dict(img_shape=(120, 160, 3)), dict(img_shape=(100, 150, 3))]
        exp = ref.refine_bboxes(rois, labels, bbox_pred, pos_is_gts, metas)
        got = mine.refine_bboxes(rois, labels, bbox_pred, pos_is_gts, metas)
        assert len(exp) == len(got) == 2
        for e, gt_ in zip(exp, got):
            assert torch.allclose(e, gt_, atol=1e-5)


@needs_ref
@pytest.mark.parametrize('name', ['gs_faster_rcnn_r50_fpn_1x_lvis_with0_bg8.py',
                                  'gs_faster_rcnn_x101_64x4d_fpn_1x_lvis.py',
                                  'gs_mask_rcnn_r50_fpn_1x_lvis.py',
                                  'gs_cascade_rcnn_x101_64x4d_fpn_1x_lvis.py'])
def test_every_non_htc_bags_config_builds_unmodified(tmp_path, name):
    """configs/bags/*.py load with Config.fromfile and build through the registries; only the
    three data-file paths are redirected to synthetic tables (the files are not in the repo).
    The HTC configs are covered by tests/test_htc_cpu.py."""
    cfg = bgs.Config.fromfile(os.path.join(ref_import.REFERENCE_ROOT, 'configs/bags', name))
    paths = gs_tables.save_group_tables(str(tmp_path), *gs_tables.synthetic_group_tables())
    heads = cfg.model.bbox_head if isinstance(cfg.model.bbox_head, list) else [cfg.model.bbox_head]
    for h in heads:
        h.gs_config.label2binlabel, h.gs_config.pred_slice, h.gs_config.fg_split = (
            paths['label2binlabel'], paths['pred_slice'], paths['fg_split'])
    model = bgs.build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    n_params = sum(p.numel() for p in model.parameters())
    assert n_params > 4e7
    from balancedgroupsoftmax_amd import train
    params = train.select_training_param(model, cfg.selectp)
    assert len(params) > 0
