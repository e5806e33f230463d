"""CPU: the on-disk formats either side of the path: the intermediate-file
generator against the reference's rule, and checkpoint I/O in the mmcv/mmdet layout."""
import collections
import json
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch

import balancedgroupsoftmax_amd as bgs
from balancedgroupsoftmax_amd import checkpoint, gs_tables
from oracle import ref_import

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_lvis_json(path, C=1231, seed=0):
    counts = gs_tables.synthetic_instance_counts(C, seed=seed)
    cats = [dict(id=i, name='c%d' % i, instance_count=int(counts[i]), image_count=1)
            for i in range(1, C)]
    with open(path, 'w') as f:
        json.dump(dict(categories=cats, images=[], annotations=[]), f)
    return counts


def test_make_group_tables_tool_follows_the_reference_rule(tmp_path):
    ann = str(tmp_path / 'lvis_train.json')
    counts = _fake_lvis_json(ann)
    out = str(tmp_path / 'lvis')
    subprocess.run([sys.executable, os.path.join(ROOT, 'tools/make_group_tables.py'), '--ann', ann,
                    '--out', out], check=True, stdout=subprocess.PIPE)
    l2b = torch.load(os.path.join(out, 'label2binlabel.pt')).numpy()
    ps = torch.load(os.path.join(out, 'pred_slice_with0.pt')).numpy()
    with open(os.path.join(out, 'valsplit.pkl'), 'rb') as f:
        split = pickle.load(f)
    # literal restatement of tools/lvis_analyse.py:17-54 / 76-91
    exp = np.zeros((5, 1231), dtype=np.int64)
    cnt = [1, 1, 1, 1, 1]
    exp[0, 1:] = 1
    cnt[0] += 1
    bins = [[], [], [], []]
    for cid in range(1, 1231):
        c = counts[cid]
        b = 1 if c < 10 else 2 if c < 100 else 3 if c < 1000 else 4
        exp[b, cid] = cnt[b]
        cnt[b] += 1
        bins[b - 1].append(cid)
    assert l2b.dtype == np.int64 and np.array_equal(l2b, exp)
    assert np.array_equal(ps[:, 1], np.array(cnt)) and ps[:, 1].sum() == 1236
    assert np.array_equal(ps[:, 0], np.concatenate([[0], np.cumsum(cnt)[:-1]]))
    for key, b in zip(['(0, 10)', '[10, 100)', '[100, 1000)', '[1000, ~)'], bins):
        assert np.array_equal(np.asarray(split[key]), np.array(b))
    assert np.array_equal(split['normal'], np.arange(1, 1231)) and split['all'].shape == (1231,)
    with open(os.path.join(out, 'bins_cls_weight.pkl'), 'rb') as f:
        w = pickle.load(f)
    assert len(w) == 4 and all(x[0] == 1.0 and x.min() >= 0.1 and x.max() <= 5.0 for x in w)
    # and the head loads them
    from balancedgroupsoftmax_amd.config import to_config_dict
    head = bgs.build_head(to_config_dict(dict(
        type='GSBBoxHeadWith0', num_fcs=2, in_channels=256, fc_out_channels=64, roi_feat_size=7,
        num_classes=1231, target_means=[0., 0., 0., 0.], target_stds=[0.1, 0.1, 0.2, 0.2],
        gs_config=dict(label2binlabel=os.path.join(out, 'label2binlabel.pt'),
                       pred_slice=os.path.join(out, 'pred_slice_with0.pt'),
                       fg_split=os.path.join(out, 'valsplit.pkl'), others_sample_ratio=8.0,
                       loss_bg=dict(type='CrossEntropyLoss'), num_bins=5,
                       loss_bin=dict(type='CrossEntropyLoss')),
        loss_cls=dict(type='CrossEntropyLoss'), loss_bbox=dict(type='SmoothL1Loss'))))
    assert tuple(head.fc_cls.weight.shape) == (1236, 64)


def test_checkpoint_roundtrip_and_mmcv_layout(tmp_path):
    model = bgs.build_backbone(dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3),
                                    frozen_stages=1, style='pytorch'))
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.1, momentum=0.9)
    path = str(tmp_path / 'epoch_1.pth')
    checkpoint.save_checkpoint(model, path, optimizer=opt, meta=dict(epoch=1, iter=100))
    raw = torch.load(path)
    assert set(raw.keys()) == {'meta', 'state_dict', 'optimizer'} and raw['meta']['epoch'] == 1
    assert isinstance(raw['state_dict'], collections.OrderedDict)
    other = bgs.build_backbone(dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3),
                                    frozen_stages=1, style='pytorch'))
    rep = checkpoint.load_checkpoint(other, path)['load_report']
    assert rep == dict(missing=[], unexpected=[], mismatched=[])
    for (k, a), (_, b) in zip(model.state_dict().items(), other.state_dict().items()):
        assert torch.equal(a, b), k
    # DDP-style 'module.' prefix, an extra key, a size mismatch: reported, not fatal (mmcv semantics)
    sd = collections.OrderedDict(('module.' + k, v) for k, v in model.state_dict().items())
    sd['module.fc.weight'] = torch.zeros(1000, 2048)
    sd['module.conv1.weight'] = torch.zeros(64, 3, 3, 3)
    torch.save(dict(state_dict=sd, meta={}), path)
    rep = checkpoint.load_checkpoint(other, path)['load_report']
    assert rep['unexpected'] == ['fc.weight'] and rep['missing'] == []
    assert rep['mismatched'] == [('conv1.weight', (64, 3, 7, 7), (64, 3, 3, 3))]
    with pytest.raises(RuntimeError):
        checkpoint.load_checkpoint(other, path, strict=True)


@pytest.mark.skipif(not ref_import.reference_available(), reason='reference tree absent')
def test_reference_module_checkpoint_loads_into_the_detector(tmp_path):
    """A checkpoint written from the REFERENCE's own modules (ResNet-50 + FPN + RPNHead, the
    'pretrained Faster R-CNN' BAGS starts from) loads key-for-key; only the 1231-row fc_cls of a
    plain head mismatches the 1236-row BAGS head — what the reference re-initialises, too."""
    ref_import.install_stubs()
    from mmdet.models.anchor_heads.rpn_head import RPNHead as RefRPN
    from mmdet.models.backbones.resnet import ResNet as RefResNet
    from mmdet.models.necks.fpn import FPN as RefFPN
    sd = collections.OrderedDict()
    for prefix, m in (('backbone', RefResNet(depth=50, num_stages=4, out_indices=(0, 1, 2, 3),
                                            frozen_stages=1, style='pytorch')),
                      ('neck', RefFPN(in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=5)),
                      ('rpn_head', RefRPN(in_channels=256, feat_channels=256, anchor_scales=[8],
                                          anchor_ratios=[0.5, 1.0, 2.0],
                                          anchor_strides=[4, 8, 16, 32, 64]))):
        for k, v in m.state_dict().items():
            sd['%s.%s' % (prefix, k)] = v
    sd['bbox_head.fc_cls.weight'] = torch.zeros(1231, 1024)      # plain Faster R-CNN classifier
    sd['bbox_head.fc_cls.bias'] = torch.zeros(1231)
    path = str(tmp_path / 'faster_rcnn_r50_fpn_1x_lvis.pth')
    torch.save(dict(meta=dict(mmdet_version='1.0.rc0'), state_dict=sd), path)
    cfg = bgs.Config.fromfile(os.path.join(ref_import.REFERENCE_ROOT,
                                           'configs/bags/gs_faster_rcnn_r50_fpn_1x_lvis_with0_bg8.py'))
    paths = gs_tables.save_group_tables(str(tmp_path), *gs_tables.synthetic_group_tables())
    gs = cfg.model.bbox_head.gs_config
    gs.label2binlabel, gs.pred_slice, gs.fg_split = (paths['label2binlabel'], paths['pred_slice'],
                                                     paths['fg_split'])
    model = bgs.build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    rep = checkpoint.load_checkpoint(model, path)['load_report']
    assert rep['unexpected'] == []
    assert [m[0] for m in rep['mismatched']] == ['bbox_head.fc_cls.weight', 'bbox_head.fc_cls.bias']
    assert all(k.startswith('bbox_head.') for k in rep['missing'])
    assert torch.equal(model.backbone.layer3[2].conv2.weight, sd['backbone.layer3.2.conv2.weight'])


def test_halo_kernel_dispatch_policy(monkeypatch):
    """``functional._use_halo_kernel``: the halo-resident 3x3 kernel is chosen for the large-M
    layers whose Cout fills its 128-wide tile; BGS_CONV_HALO=0|1 overrides (tuning / A-B runs)."""
    from balancedgroupsoftmax_amd import functional as BF
    monkeypatch.delenv('BGS_CONV_HALO', raising=False)
    assert BF._use_halo_kernel(2 * 200 * 336, 256)          # FPN output conv / RPN conv on P2
    assert not BF._use_halo_kernel(2 * 100 * 168, 256)      # P3: the general kernel is faster
    assert not BF._use_halo_kernel(2 * 200 * 336, 64)       # layer1 conv2: half the tile would idle
    monkeypatch.setenv('BGS_CONV_HALO', '0')
    assert not BF._use_halo_kernel(2 * 200 * 336, 256)
    monkeypatch.setenv('BGS_CONV_HALO', '1')
    assert BF._use_halo_kernel(10, 15)


def _tiny_model():
    import torch.nn as nn

    class Blk(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(3, 4, 3)
            self.bn1 = nn.BatchNorm2d(4)
            self.gn = nn.GroupNorm(2, 4)
            self.fc_cls = nn.Linear(4, 5)
            self.fc_frozen = nn.Linear(4, 2)
            for p in self.fc_frozen.parameters():
                p.requires_grad = False
    return Blk()


def test_build_optimizer_plain_and_paramwise_rules():
    """mmdet/apis/train.py:63-140: plain branch = the trainable parameters under the global settings;
    ``paramwise_options`` = one group per parameter with the norm / bias multipliers."""
    from balancedgroupsoftmax_amd import train
    m = _tiny_model()
    cfg = dict(type='SGD', lr=0.02, momentum=0.9, weight_decay=1e-4)
    opt = train.build_optimizer(m, cfg)
    assert len(opt.param_groups) == 1
    assert [id(p) for p in opt.param_groups[0]['params']] == [id(p) for p in m.parameters() if p.requires_grad]
    assert cfg['type'] == 'SGD'                                          # the caller's dict is not consumed
    opt = train.build_optimizer(m, dict(cfg, paramwise_options=dict(bias_lr_mult=2., bias_decay_mult=0.,
                                                                    norm_decay_mult=0.5)))
    by_name = {n: g for (n, _), g in zip(m.named_parameters(), opt.param_groups)}
    assert len(opt.param_groups) == len(list(m.named_parameters()))
    assert by_name['conv1.weight']['lr'] == 0.02 and by_name['conv1.weight']['weight_decay'] == 1e-4
    assert by_name['conv1.bias']['lr'] == 0.04 and by_name['conv1.bias']['weight_decay'] == 0.
    assert by_name['bn1.weight']['weight_decay'] == 5e-5 and by_name['bn1.bias']['weight_decay'] == 5e-5
    assert by_name['bn1.bias']['lr'] == 0.02                              # norm rule wins over the bias rule
    assert by_name['gn.weight']['weight_decay'] == 5e-5
    assert by_name['fc_frozen.bias']['lr'] == 0.02 and by_name['fc_frozen.bias']['weight_decay'] == 1e-4
    with pytest.raises(TypeError):
        train.build_optimizer(list(m.parameters()), dict(cfg, paramwise_options=dict(bias_lr_mult=2.)))
    with pytest.raises(AssertionError):
        train.build_optimizer(m, dict(type='SGD', lr=0.1, paramwise_options=dict(norm_decay_mult=0.)))


def test_build_optimizer_paramwise_equals_the_executed_reference():
    from oracle import ref_import
    if not ref_import.reference_available():
        pytest.skip('reference tree not present')
    ref_import.install_stubs()
    try:
        from mmdet.apis.train import build_optimizer as ref_build
    except Exception as e:                                                # the apis module pulls mmcv.runner etc.
        pytest.skip('reference apis.train not importable under the stubs: %r' % (e,))
    from balancedgroupsoftmax_amd import train
    from balancedgroupsoftmax_amd.config import to_config_dict
    m = _tiny_model()
    for pw in (None, dict(bias_lr_mult=2., bias_decay_mult=0., norm_decay_mult=0.5), dict(bias_lr_mult=3.)):
        cfg = dict(type='SGD', lr=0.02, momentum=0.9, weight_decay=1e-4)
        if pw is not None:
            cfg['paramwise_options'] = pw
        mine, ref = train.build_optimizer(m, cfg), ref_build(m, to_config_dict(cfg))
        assert len(mine.param_groups) == len(ref.param_groups)
        for a, b in zip(mine.param_groups, ref.param_groups):
            assert [id(p) for p in a['params']] == [id(p) for p in b['params']]
            for k in ('lr', 'momentum', 'weight_decay', 'dampening', 'nesterov'):
                assert a[k] == b[k], k


# ---------------------------------------------------------------------------------------
# fp16 decorators (mmdet/core/fp16/decorators.py)
# ---------------------------------------------------------------------------------------
def _decorated_module(auto_fp16, force_fp32):
    import torch

    class M(torch.nn.Module):
        def __init__(self):
            super(M, self).__init__()
            self.fp16_enabled = False

        @auto_fp16()
        def fwd_all(self, x, y=None):
            return x, y

        @auto_fp16(apply_to=('x', ), out_fp32=True)
        def fwd_x_out32(self, x, y):
            return dict(x=x * 2, y=y, n=3)

        @force_fp32(apply_to=('feats', ), out_fp16=True)          # single_level.py:89
        def extract(self, feats, rois, roi_scale_factor=None):
            return [f + 1 for f in feats], rois

        @force_fp32(apply_to=('cls_score', 'bbox_pred'))            # gs_bbox_head_with0.py:147
        def loss(self, cls_score, bbox_pred, labels):
            return cls_score.sum() + bbox_pred.sum(), labels

    return M()


def _dtypes(o):
    import torch
    if isinstance(o, torch.Tensor):
        return (str(o.dtype), o.double().sum().item())
    if isinstance(o, dict):
        return {k: _dtypes(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(_dtypes(v) for v in o)
    return o


def _fp16_calls(m):
    import torch
    f32 = torch.arange(6, dtype=torch.float32).view(2, 3)
    f16 = f32.half()
    i64 = torch.arange(4)
    out = []
    for enabled in (False, True):
        m.fp16_enabled = enabled
        out.append(_dtypes(m.fwd_all(f32, y=[f32, i64, 'tag'])))
        out.append(_dtypes(m.fwd_all(f16)))
        out.append(_dtypes(m.fwd_x_out32(f32, f32)))
        out.append(_dtypes(m.fwd_x_out32(x=f32, y=f16)))
        out.append(_dtypes(m.extract((f16, f32), f32)))
        out.append(_dtypes(m.extract(feats=[f16], rois=f16, roi_scale_factor=2.0)))
        out.append(_dtypes(m.loss(f16, f16, i64)))
        out.append(_dtypes(m.loss(f32, bbox_pred=f16, labels=i64)))
    return out


def test_fp16_decorators_incl_out_fp16_and_out_fp32():
    """``auto_fp16(out_fp32=)`` / ``force_fp32(out_fp16=)`` (VERDICT r5 'missing' 2: accepted and ignored before): off
    unless ``fp16_enabled``; then every tensor of a selected argument is cast (whatever its dtype: the reference's
    ``inputs.to(dst)``), nested containers included, and the OUTPUT is cast back when asked — the extractor's
    ``out_fp16=True`` (single_level.py:89) returns half features."""
    import torch
    from balancedgroupsoftmax_amd.fp16_utils import auto_fp16, force_fp32
    m = _decorated_module(auto_fp16, force_fp32)
    f32 = torch.ones(2, 3)
    m.fp16_enabled = True
    feats, rois = m.extract([f32.half()], f32)
    assert feats[0].dtype == torch.half and rois.dtype == torch.half        # out_fp16 casts the whole output
    o = m.fwd_x_out32(f32, f32)
    assert o['x'].dtype == torch.float32 and o['y'].dtype == torch.float32 and o['n'] == 3
    assert m.fwd_all(f32)[0].dtype == torch.half
    loss, labels = m.loss(f32.half(), f32.half(), torch.arange(3))
    assert loss.dtype == torch.float32 and labels.dtype == torch.int64
    m.fp16_enabled = False
    assert m.extract([f32.half()], f32)[0][0].dtype == torch.half and m.fwd_all(f32)[0].dtype == torch.float32
    with pytest.raises(TypeError):
        auto_fp16()(lambda self, x: x)(object(), f32)


@pytest.mark.skipif(not ref_import.reference_available(), reason='reference tree not present')
def test_fp16_decorators_equal_the_executed_reference_decorators():
    ref_import.install_stubs()
    from mmdet.core.fp16.decorators import auto_fp16 as ref_auto, force_fp32 as ref_force
    from balancedgroupsoftmax_amd.fp16_utils import auto_fp16, force_fp32
    assert _fp16_calls(_decorated_module(auto_fp16, force_fp32)) == _fp16_calls(_decorated_module(ref_auto, ref_force))
