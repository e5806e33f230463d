"""CPU: registry / config / group tables / box coding — the host-side mirror of the
reference interface (no kernels involved)."""
import os
import pickle

import numpy as np
import pytest
import torch

import balancedgroupsoftmax_amd as bgs
from balancedgroupsoftmax_amd import box_ops, gs_tables, losses
from balancedgroupsoftmax_amd.registry import Registry, build_from_cfg

REF_CFG = '/root/reference/configs/bags/gs_faster_rcnn_r50_fpn_1x_lvis_with0_bg8.py'


def test_registry_keys_present():
    for k in ['GSBBoxHeadWith0', 'GSBBoxHeadWith0Reweight', 'GSBBoxHead', 'SharedFCBBoxHead',
              'ConvFCBBoxHead', 'BBoxHead']:
        assert k in bgs.HEADS, k
    for k in ['CrossEntropyLoss', 'SmoothL1Loss']:
        assert k in bgs.LOSSES, k
    assert bgs.HEADS.get('GSBBoxHead') is bgs.HEADS.get('GSBBoxHeadWith0')


def test_registry_error_behaviour():
    r = Registry('thing')

    @r.register_module
    class A(object):
        def __init__(self, x=1, y=2):
            self.x, self.y = x, y

    with pytest.raises(KeyError):
        r.register_module(A)
    with pytest.raises(TypeError):
        r.register_module(3)
    with pytest.raises(KeyError):
        build_from_cfg(dict(type='Nope'), r)
    a = build_from_cfg(dict(type='A', x=5), r, default_args=dict(x=7, y=9))
    assert (a.x, a.y) == (5, 9)           # default_args never override the config
    assert build_from_cfg(dict(type=A), r).x == 1


def test_weight_reduce_known_answers():
    """reference doctest mmdet/models/losses/utils.py:67-83."""
    p, t, w = torch.tensor([0., 2, 3]), torch.tensor([1., 1, 1]), torch.tensor([1., 0, 1])
    el = (p - t).abs()
    assert losses.reduce_weighted(el).item() == pytest.approx(1.3333, abs=1e-4)
    assert losses.reduce_weighted(el, w).item() == pytest.approx(1.0)
    assert losses.reduce_weighted(el, reduction='none').tolist() == [1., 1., 2.]
    assert losses.reduce_weighted(el, w, avg_factor=2).item() == pytest.approx(1.5)
    with pytest.raises(ValueError):
        losses.reduce_weighted(el, w, reduction='sum', avg_factor=2)


def test_delta2bbox_known_answers():
    """reference doctest mmdet/core/bbox/transforms.py:64-77."""
    rois = torch.tensor([[0., 0., 1., 1.], [0., 0., 1., 1.], [0., 0., 1., 1.], [5., 5., 5., 5.]])
    deltas = torch.tensor([[0., 0., 0., 0.], [1., 1., 1., 1.], [0., 0., 2., -1.],
                           [0.7, -1.9, -0.5, 0.3]])
    out = box_ops.delta2bbox(rois, deltas, max_shape=(32, 32))
    exp = torch.tensor([[0.0000, 0.0000, 1.0000, 1.0000], [0.2817, 0.2817, 4.7183, 4.7183],
                        [0.0000, 0.6321, 7.3891, 0.3679], [5.8967, 2.9251, 5.5033, 3.2749]])
    assert torch.allclose(out, exp, atol=1e-4)


def test_bbox2delta_roundtrip():
    g = torch.Generator().manual_seed(0)
    xy = torch.rand(50, 2, generator=g) * 500
    wh = torch.rand(50, 2, generator=g) * 200 + 4
    props = torch.cat([xy, xy + wh], 1)
    gts = props + torch.randn(50, 4, generator=g) * 3
    stds = (0.1, 0.1, 0.2, 0.2)
    d = box_ops.bbox2delta(props, gts, stds=stds)
    back = box_ops.delta2bbox(props, d, stds=stds)
    assert torch.allclose(back, gts, atol=2e-3)


def test_group_tables_rule_and_roundtrip(tmp_path):
    counts = gs_tables.synthetic_instance_counts(1231, seed=0)
    l2b, ps, split = gs_tables.build_group_tables(counts)
    assert l2b.shape == (5, 1231) and ps.shape == (5, 2)
    assert l2b[0, 0] == 0 and (l2b[0, 1:] == 1).all()
    assert int(ps[:, 1].sum()) == 1231 + 5 and ps[0].tolist() == [0, 2]
    thr = [10, 100, 1000]
    for c in range(1, 1231):
        b = 1 + int(np.searchsorted(thr, counts[c], side='right'))
        assert l2b[b, c] > 0 and (np.delete(l2b[1:, c], b - 1) == 0).all()
    # valsplit[key][k-1] == c  <=>  label2binlabel[b, c] == k   (tools/lvis_analyse.py:76-91)
    for b, key in enumerate(gs_tables.FG_SPLIT_KEYS_5, start=1):
        ids = split[key]
        assert (l2b[b, ids] == np.arange(1, len(ids) + 1)).all()
    paths = gs_tables.save_group_tables(str(tmp_path), l2b, ps, split)
    l2b_t, ps_t, fg = gs_tables.load_group_tables(paths['label2binlabel'], paths['pred_slice'],
                                                  paths['fg_split'])
    assert l2b_t.dtype == torch.int64 and (l2b_t.numpy() == l2b).all()
    assert [f.numel() for f in fg] == (ps[1:, 1] - 1).tolist()
    col = gs_tables.class_to_column(l2b_t, ps_t)
    assert col[0] == 0 and col.min() >= 0 and col.max() < 1236
    assert len(set(col.tolist())) == 1231   # injective: every class has its own column
    # 3-bin and 9-bin variants (tools/lvis_analyse.py:487-526, 564-621)
    assert gs_tables.build_group_tables(counts, (100,))[1][:, 1].sum() == 1231 + 3
    assert gs_tables.build_group_tables(counts, (5, 10, 50, 100, 500, 1000, 5000))[1][:, 1].sum() \
        == 1231 + 9


def _gs_head_cfg(tmp_path, **over):
    counts = gs_tables.synthetic_instance_counts(1231, seed=0)
    l2b, ps, split = gs_tables.build_group_tables(counts)
    bcw = gs_tables.bin_class_weights(counts, l2b)
    paths = gs_tables.save_group_tables(str(tmp_path), l2b, ps, split, bcw)
    ce = dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)
    cfg = dict(type='GSBBoxHeadWith0', num_fcs=2, in_channels=256, fc_out_channels=1024,
               gs_config=dict(label2binlabel=paths['label2binlabel'],
                              pred_slice=paths['pred_slice'], fg_split=paths['fg_split'],
                              others_sample_ratio=8.0, loss_bg=dict(ce), num_bins=5,
                              loss_bin=dict(ce)),
               roi_feat_size=7, num_classes=1231, target_means=[0., 0., 0., 0.],
               target_stds=[0.1, 0.1, 0.2, 0.2], reg_class_agnostic=False, loss_cls=dict(ce),
               loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0))
    cfg.update(over)
    return bgs.config.to_config_dict(cfg), paths


def test_gs_head_construction_and_state_dict_keys(tmp_path):
    cfg, paths = _gs_head_cfg(tmp_path)
    head = bgs.build_head(cfg)
    sd = {k: tuple(v.shape) for k, v in head.state_dict().items()}
    # the checkpoint compatibility surface: names AND shapes
    assert sd == {'fc_cls.weight': (1236, 1024), 'fc_cls.bias': (1236,),
                  'fc_reg.weight': (4924, 1024), 'fc_reg.bias': (4924,),
                  'shared_fcs.0.weight': (1024, 12544), 'shared_fcs.0.bias': (1024,),
                  'shared_fcs.1.weight': (1024, 1024), 'shared_fcs.1.bias': (1024,)}
    assert head.num_classes == 1231 and head.reg_class_agnostic is False
    assert head.fp16_enabled is False and isinstance(head.fc_cls, torch.nn.Linear)
    head.init_weights()
    # ONE path in the product: forward and loss refuse CPU tensors instead of computing on the host
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        head(torch.randn(3, 256, 7, 7))
    from oracle import tensor_forms
    cls, reg = tensor_forms.convfc_bbox_forward(head, torch.randn(3, 256, 7, 7))   # torch restatement
    assert cls.shape == (3, 1236) and reg.shape == (3, 4924)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        head.loss(cls, reg, torch.zeros(3, dtype=torch.long), torch.ones(3), torch.zeros(3, 4),
                  torch.zeros(3, 4))
    # class -> column table equals the table-derived one
    col = gs_tables.class_to_column(head.label2binlabel, head.pred_slice)
    assert torch.equal(col, head.cls2col)


def test_gs_head_reweight_and_alias(tmp_path):
    cfg, paths = _gs_head_cfg(tmp_path, type='GSBBoxHeadWith0Reweight')
    cfg['gs_config']['bin_cls_weight'] = paths['bin_cls_weight']
    head = bgs.build_head(cfg)
    assert head.cls_weight_table.shape[0] == 4
    with open(paths['bin_cls_weight'], 'rb') as f:
        w = pickle.load(f)
    assert torch.allclose(head.cls_weight_table[1, :len(w[1])], torch.tensor(w[1]).float())
    cfg2, _ = _gs_head_cfg(tmp_path, type='GSBBoxHead')
    assert type(bgs.build_head(cfg2)).__name__ == 'GSBBoxHeadWith0'
    bad, _ = _gs_head_cfg(tmp_path)
    bad['gs_config']['num_bins'] = 4
    with pytest.raises(ValueError):
        bgs.build_head(bad)


@pytest.mark.skipif(not os.path.exists(REF_CFG), reason='reference tree not present')
def test_reference_config_loads_unmodified():
    cfg = bgs.Config.fromfile(REF_CFG)
    assert cfg.model.type == 'GroupSoftmax'
    assert cfg.model.bbox_head.type == 'GSBBoxHeadWith0'
    assert cfg.model.bbox_head.gs_config.num_bins == 5          # attribute access
    assert cfg.model.bbox_head.gs_config.others_sample_ratio == 8.0
    assert cfg.selectp == 1 and cfg.data.imgs_per_gpu == 2
    assert cfg.train_cfg.rcnn.sampler.num == 512
    assert cfg.optimizer_config.grad_clip.max_norm == 35


def test_conv_math_env_is_validated_and_scope_restores():
    """a typo in BGS_CONV_MATH must raise at import instead of silently selecting the
    reduced-precision kernels; conv_math_scope restores the mode on exit, also on error."""
    import subprocess
    import sys
    env = dict(os.environ, BGS_CONV_MATH='fp32')
    r = subprocess.run([sys.executable, '-c', 'import balancedgroupsoftmax_amd.functional'],
                       env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b'unknown conv math' in r.stderr
    from balancedgroupsoftmax_amd import functional as BF
    before = BF.conv_math()
    with pytest.raises(ValueError):
        BF.set_conv_math('BF16X6')
    try:
        with BF.conv_math_scope('f32'):
            assert BF.conv_math() == 'f32'
            raise KeyError('x')
    except KeyError:
        pass
    assert BF.conv_math() == before


def test_wrap_fp16_model_is_scoped_to_the_model():
    """the bf16 mode lives in forward hooks of the wrapped model, not in a process global."""
    import torch
    from balancedgroupsoftmax_amd import functional as BF
    from balancedgroupsoftmax_amd import train
    seen = []

    class M(torch.nn.Module):
        def forward(self, x):
            seen.append(BF.conv_math())
            return x

    m, other = M(), M()
    outside = train.wrap_fp16_model(m, 'bf16')
    assert outside == BF.conv_math() != 'bf16'
    m(torch.zeros(1))
    other(torch.zeros(1))
    assert seen == ['bf16', outside] and BF.conv_math() == outside
    # direct method / sub-module calls bypass the __call__ hooks; fp16_scope covers them
    m.forward(torch.zeros(1))
    assert seen[-1] == outside
    with train.fp16_scope(m):
        m.forward(torch.zeros(1))
        assert seen[-1] == 'bf16'
        with train.fp16_scope(other):          # not wrapped: leaves the mode alone
            assert BF.conv_math() == 'bf16'
    assert BF.conv_math() == outside
    try:
        with train.fp16_scope(m):
            raise KeyError('x')
    except KeyError:
        pass
    assert BF.conv_math() == outside
    train.unwrap_fp16_model(m)
    m(torch.zeros(1))
    assert seen[-1] == outside


def test_parse_losses_batched_segment_sums_equal_the_per_key_formula_with_gradients():
    """``train.parse_losses`` computes every per-key sum and the total from ONE stack and one 0/1 segment
    product; values and gradients equal mmdet/apis/train.py:24-47 (per key: mean of a tensor / sum of the
    means of a list; total over the keys containing 'loss')."""
    import torch
    from balancedgroupsoftmax_amd import train
    g = torch.Generator().manual_seed(0)
    leaves = [torch.randn((), generator=g, requires_grad=True) for _ in range(14)]
    d = dict(loss_rpn_cls=leaves[0:5], loss_rpn_bbox=leaves[5:10], loss_cls_bin0=leaves[10],
             acc=leaves[11], loss_bbox=leaves[12], loss_vec=torch.stack([leaves[13], leaves[13] * 3]))
    loss, log_vars = train.parse_losses(d)
    exp = dict(loss_rpn_cls=sum(leaves[0:5]), loss_rpn_bbox=sum(leaves[5:10]), loss_cls_bin0=leaves[10],
               acc=leaves[11], loss_bbox=leaves[12], loss_vec=leaves[13] * 2)
    for k, v in exp.items():
        assert abs(float(log_vars[k]) - float(v)) < 1e-6, k
    total = sum(v for k, v in exp.items() if 'loss' in k)
    assert abs(float(loss) - float(total)) < 1e-6 and abs(float(log_vars['loss']) - float(total)) < 1e-6
    loss.backward()
    for i, t in enumerate(leaves):
        if i == 11:                     # `acc` is not part of the total: no gradient reaches it (as in the reference)
            assert t.grad is None or float(t.grad) == 0.0
            continue
        want = 2.0 if i == 13 else 1.0
        assert abs(float(t.grad) - want) < 1e-6, i
    assert len(train._SEGMENT_MATS) >= 1          # cached per structure: a second call builds nothing new
    n = len(train._SEGMENT_MATS)
    train.parse_losses({k: ([t.detach() for t in v] if isinstance(v, list) else v.detach()) for k, v in d.items()})
    assert len(train._SEGMENT_MATS) == n


def test_fused_clip_sgd_is_chosen_only_where_it_is_the_same_computation():
    """``DistOptimizerStep`` takes the fused clip + SGD kernels only for what they implement (one param
    group of plain torch.optim.SGD, L2 clipping, fp32 CUDA parameters); everything else — and CPU
    parameters, as here — keeps the torch path."""
    import torch
    from balancedgroupsoftmax_amd import train
    ps = [torch.nn.Parameter(torch.randn(4, 3))]
    opt = torch.optim.SGD(ps, lr=0.1, momentum=0.9, weight_decay=1e-4)
    assert not train._fused_sgd_eligible(opt, ps, dict(max_norm=35, norm_type=2))       # CPU tensors
    step = train.DistOptimizerStep(ps, opt, dict(max_norm=35, norm_type=2))
    assert step.fused is None
    ps[0].grad = torch.ones_like(ps[0])
    before = ps[0].detach().clone()
    step.exchange_and_update()                                   # the torch path still steps
    assert not torch.equal(ps[0].detach(), before)

    class FakeCuda(torch.nn.Parameter):                          # eligibility rules without a GPU
        is_cuda = True
    fp = [FakeCuda(torch.randn(2, 2))]
    for kwargs, clip, ok in ((dict(), dict(max_norm=35, norm_type=2), True),
                             (dict(nesterov=True), dict(max_norm=35, norm_type=2), False),
                             (dict(dampening=0.1), dict(max_norm=35, norm_type=2), False),
                             (dict(), dict(max_norm=35, norm_type=1), False),
                             (dict(), None, True)):
        o = torch.optim.SGD(fp, lr=0.1, momentum=0.9, **kwargs)
        assert train._fused_sgd_eligible(o, fp, clip) == ok, (kwargs, clip)
    two = torch.optim.SGD([dict(params=fp), dict(params=[FakeCuda(torch.randn(2))], lr=0.5)], lr=0.1, momentum=0.9)
    assert not train._fused_sgd_eligible(two, fp, None)
    assert not train._fused_sgd_eligible(torch.optim.Adam(fp), fp, None)


def test_proposal_list_is_the_reference_list_plus_its_batch_tensors():
    import torch
    from balancedgroupsoftmax_amd.rpn import ProposalList
    props, valid = torch.randn(2, 7, 5), torch.rand(2, 7) > 0.5
    pl = ProposalList(props, valid)
    assert len(pl) == 2 and isinstance(pl, list)
    for i, (p, v) in enumerate(pl):
        assert p.data_ptr() == props[i].data_ptr() and torch.equal(v, valid[i])
    assert pl.batched[0] is props and pl.batched[1] is valid


def test_unit_gradient_is_one_cached_tensor_per_device_and_recognised_by_storage():
    """functional.unit_gradient: the constant root gradient the GroupSoftmax head step recognises (no
    scaling launch); a different ones tensor is NOT it, a view of the same storage is — and a buffer somebody
    wrote to in place is no longer taken for ones (version counter), a fresh one is handed out instead."""
    import torch
    from balancedgroupsoftmax_amd import functional as BF
    u = BF.unit_gradient('cpu')
    assert u.data_ptr() == BF.unit_gradient(torch.device('cpu')).data_ptr()
    assert u.shape == (1,) and u.dtype == torch.float32 and float(u) == 1.0
    assert BF._is_unit_gradient(u) and BF._is_unit_gradient(u.detach()) and BF._is_unit_gradient(u.reshape(()))
    u6 = BF.unit_gradient('cpu', 6)
    assert u6.shape == (6,) and u6.data_ptr() == u.data_ptr() and BF._is_unit_gradient(u6)
    assert not BF._is_unit_gradient(torch.ones(1))
    assert not BF._is_unit_gradient(u.double())
    assert not BF._is_unit_gradient(BF.unit_gradient('cpu', 8)[2:5])          # not the head of the buffer
    u.mul_(2.0)                                                               # a caller breaks the contract ..
    assert not BF._is_unit_gradient(u)                                        # .. and the general path runs
    fresh = BF.unit_gradient('cpu')
    assert float(fresh) == 1.0 and fresh.data_ptr() != u.data_ptr() and BF._is_unit_gradient(fresh)


def test_unit_gradient_reaches_the_head_terms_by_identity_through_parse_losses():
    """``train.backward_unit(parse_losses(losses)[0])``: the root gradient is the cached unit gradient, every
    'loss' scalar receives THAT tensor (no arithmetic), and the fused head's term split turns six of them back into
    the cached ones vector — the chain the GPU head relies on to skip its gradient-scaling launch.  With a factor
    anywhere on the way (a weighted total) the general path runs and the values are the weights."""
    import torch
    from balancedgroupsoftmax_amd import functional as BF
    from balancedgroupsoftmax_amd import train

    seen = []

    class Head(torch.autograd.Function):            # stands in for _GsHeadStepFn: records what backward receives
        @staticmethod
        def forward(ctx, x):
            return x * 2.0

        @staticmethod
        def backward(ctx, g):
            seen.append((BF._is_unit_gradient(g), g.detach().clone()))
            return g * 2.0

    x = torch.arange(6, dtype=torch.float32, requires_grad=True)
    parts = BF.unbind_terms(Head.apply(x))
    other = torch.tensor(3.0, requires_grad=True)
    losses = dict(loss_cls_bin0=parts[0], loss_cls_bin1=parts[1], loss_cls_bin2=parts[2], loss_cls_bin3=parts[3],
                  loss_cls_bin4=parts[4], loss_bbox=parts[5], loss_rpn_cls=[other * 1.0, other * 2.0], acc=other * 5.0)
    loss, log_vars = train.parse_losses(losses)
    assert abs(float(loss) - (2 * 15 + 9)) < 1e-6 and abs(float(log_vars['loss_rpn_cls']) - 9.0) < 1e-6
    # (CPU tensors: backward_unit's CUDA fast path is taken on the GPU; hand the unit gradient over explicitly)
    loss.backward(BF.unit_gradient('cpu').reshape(()))
    assert len(seen) == 1 and seen[0][0] and seen[0][1].tolist() == [1.0] * 6
    assert x.grad.tolist() == [2.0] * 6 and float(other.grad) == 3.0
    # a weighted total: not the unit gradient any more
    seen.clear()
    x.grad = None
    parts = BF.unbind_terms(Head.apply(x))
    loss, _ = train.parse_losses(dict(loss_a=parts[0], loss_b=parts[1], loss_c=[parts[2], parts[3]], loss_d=parts[4],
                                      acc=parts[5]))
    (loss * 0.5).backward()
    assert len(seen) == 1 and not seen[0][0] and seen[0][1].tolist() == [0.5] * 5 + [0.0]


def test_parse_losses_contains_a_non_finite_term_to_its_own_key_and_the_total():
    """A NaN loss term makes ITS key and the total NaN; an infinite metric (`acc` is not part of the total) touches
    neither the other keys nor the total — the reference's per-key sums behave like this, a 0/1-matrix product
    would not (0 * NaN = NaN)."""
    import math
    import torch
    from balancedgroupsoftmax_amd import train
    d = dict(loss_a=[torch.tensor(1.0), torch.tensor(2.0)], loss_b=torch.tensor(float('nan')),
             acc=torch.tensor(float('inf')), loss_c=torch.tensor(3.0))
    loss, lv = train.parse_losses(d)
    assert float(lv['loss_a']) == 3.0 and float(lv['loss_c']) == 3.0 and math.isinf(float(lv['acc']))
    assert math.isnan(float(lv['loss_b'])) and math.isnan(float(loss))
    d.pop('loss_b')
    loss, lv = train.parse_losses(d)
    assert float(loss) == 6.0