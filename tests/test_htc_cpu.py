"""CPU: Hybrid Task Cascade host mirror — registry keys, state-dict layout and the plain-torch
forward of ``HTCMaskHead`` / ``FusedSemanticHead`` against the reference's own classes and the
committed golden vectors of the executed reference (tests/golden/make_golden_htc.py)."""
import os

import numpy as np
import pytest
import torch

import balancedgroupsoftmax_amd as bgs
from balancedgroupsoftmax_amd import gs_tables, train
from oracle import mask_oracle, ref_import, tensor_forms
from tests.golden import make_golden_htc as G

needs_ref = pytest.mark.skipif(not ref_import.reference_available(), reason='reference tree absent')
GOLD = os.path.join(os.path.dirname(G.__file__), 'htc_heads_golden.npz')


def _semantic_head():
    return bgs.build_head(dict(type='FusedSemanticHead', **G.semantic_head_cfg()))


def _mask_head():
    return bgs.build_head(dict(type='HTCMaskHead', **G.mask_head_cfg()))


def test_fused_semantic_head_torch_path_vs_executed_reference_golden():
    """The torch restatement (oracle/tensor_forms.py) over the mirror's parameter containers == the
    executed reference: pins the state-dict layout and the oracle the GPU test compares with."""
    z = np.load(GOLD)
    head = _semantic_head()
    with torch.no_grad():
        mask_oracle.fill_mask_head(head.state_dict(), G.SEM['seed'] + 1000)
    feats, labels = G.semantic_inputs()
    pred, emb = tensor_forms.semantic_forward(head, [torch.from_numpy(f) for f in feats])
    assert np.abs(pred.detach().numpy() - z['sem/pred']).max() < 1e-4
    assert np.abs(emb.detach().numpy()[:, ::2] - z['sem/feat']).max() < 1e-4
    loss = tensor_forms.semantic_loss(head, pred, torch.from_numpy(labels))
    assert abs(float(loss.detach()) - float(z['sem/loss'][0])) < 1e-5
    with pytest.raises(RuntimeError, match='no CPU fallback'):     # the product has ONE path
        head([torch.from_numpy(f) for f in feats])


def test_htc_mask_head_torch_path_vs_executed_reference_golden():
    z = np.load(GOLD)
    h0, h1 = _mask_head(), _mask_head()
    with torch.no_grad():
        mask_oracle.fill_mask_head(h0.state_dict(), G.MSK['seed'] + 1000)
        mask_oracle.fill_mask_head(h1.state_dict(), G.MSK['seed'] + 2000)
    feats, labels, targets = G.mask_inputs()
    x, lab = torch.from_numpy(feats), torch.from_numpy(labels)
    with torch.no_grad():
        fwd = tensor_forms.htc_mask_forward
        last = fwd(h0, x, None, return_logits=False)
        assert np.abs(last.numpy()[:, ::4] - z['msk/res_feat0']).max() < 1e-4
        pred0, feat0 = fwd(h0, x, None)
        assert torch.equal(feat0, last) and pred0.shape == (G.MSK['P'], G.MSK['C'], 28, 28)
        z0 = fwd(h0, x, None, return_feat=False, labels=lab)
        assert np.abs(z0.numpy() - z['msk/gt_logits0']).max() < 1e-4
        z1 = fwd(h1, x, last, return_feat=False, labels=lab)
        assert np.abs(z1.numpy() - z['msk/gt_logits1']).max() < 1e-4
        got = mask_oracle.mask_cross_entropy(z1.numpy(), targets)
        assert abs(got - float(z['msk/loss'][0])) < 2e-6


@needs_ref
def test_htc_heads_state_dict_matches_reference_modules():
    ref_import.install_stubs()
    from mmdet.models.mask_heads.fused_semantic_head import FusedSemanticHead as RefSem
    from mmdet.models.mask_heads.htc_mask_head import HTCMaskHead as RefMask
    for ref, mine in ((RefSem(**G.semantic_head_cfg()), _semantic_head()),
                      (RefMask(**G.mask_head_cfg()), _mask_head())):
        a = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
        b = {k: tuple(v.shape) for k, v in mine.state_dict().items()}
        assert a == b


def _load_htc_cfg(tmp_path, name='gs_htc_x101_64x4d_fpn_20e_16gpu_lvis.py', depth=None):
    cfg = bgs.Config.fromfile(os.path.join(ref_import.REFERENCE_ROOT, 'configs/bags', name))
    paths = gs_tables.save_group_tables(str(tmp_path), *gs_tables.synthetic_group_tables())
    for h in cfg.model.bbox_head:
        h.gs_config.label2binlabel, h.gs_config.pred_slice, h.gs_config.fg_split = (
            paths['label2binlabel'], paths['pred_slice'], paths['fg_split'])
    if depth is not None:
        cfg.model.backbone.depth = depth
    cfg.model.pretrained = None
    return cfg


@needs_ref
def test_htc_builds_from_reference_config_with_the_reference_detectors_state_dict(tmp_path):
    """configs/bags/gs_htc_x101_64x4d_fpn_20e_16gpu_lvis.py builds unmodified (tables redirected)
    and its parameter names / shapes equal those of the reference's own ``HybridTaskCascade`` built
    from the same config (trunk depth cut to 50 to keep the test quick)."""
    cfg = _load_htc_cfg(tmp_path, depth=50)
    mine = bgs.build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)
    assert type(mine).__name__ == 'HybridTaskCascade' and mine.num_stages == 3
    assert mine.interleaved and mine.mask_info_flow and mine.with_semantic
    assert mine.semantic_fusion == ('bbox', 'mask')
    assert [type(h).__name__ for h in mine.mask_head] == ['HTCMaskHead'] * 3
    assert mine.semantic_roi_extractor.featmap_strides == [8]
    assert mine.semantic_roi_extractor.out_size == 14 and mine.bbox_roi_extractor[0].out_size == 7
    assert cfg.selectp == 3
    params = train.select_training_param(mine, cfg.selectp)
    assert len(params) == 6 and sum(p.numel() for p in params) == 3 * (1236 * 1024 + 1236)
    ref_import.install_stubs()
    from mmdet.models import build_detector as ref_build
    cfg2 = _load_htc_cfg(tmp_path, depth=50)
    ref = ref_build(cfg2.model, train_cfg=cfg2.train_cfg, test_cfg=cfg2.test_cfg)
    a = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    b = {k: tuple(v.shape) for k, v in mine.state_dict().items()}
    assert a == b and len(a) > 400
    mine.load_state_dict(ref.state_dict())          # a reference checkpoint drops in


@needs_ref
def test_htc_dconv_config_is_refused_loudly(tmp_path):
    """The second HTC config adds deformable convolutions (dcn c3-c5) and multi-scale training:
    deformable conv has no kernel here, and the constructor says so instead of building a model
    that silently uses plain convs."""
    cfg = _load_htc_cfg(tmp_path, 'gs_htc_dconv_c3-c5_mstrain_400_1400_x101_64x4d_fpn_20e_lvis.py')
    with pytest.raises(NotImplementedError, match='dcn'):
        bgs.build_detector(cfg.model, train_cfg=cfg.train_cfg, test_cfg=cfg.test_cfg)


def test_cascade_mask_rcnn_is_refused_loudly(tmp_path):
    """Cascade Mask R-CNN (mask heads without HTC's flow) is not a BAGS config."""
    from bench import detector_cfg
    from balancedgroupsoftmax_amd.config import to_config_dict
    model_cfg, train_cfg = detector_cfg(str(tmp_path), htc=True)
    model_cfg['backbone'] = dict(model_cfg['backbone'], depth=50)
    mine = bgs.build_detector(to_config_dict(model_cfg), train_cfg=to_config_dict(train_cfg),
                              test_cfg=None)
    assert type(mine).__name__ == 'HybridTaskCascade' and len(mine.mask_head) == 3
    for k in ('semantic_roi_extractor', 'semantic_head', 'interleaved', 'mask_info_flow'):
        model_cfg.pop(k)
    model_cfg['type'] = 'CascadeRCNN'
    with pytest.raises(NotImplementedError, match='HybridTaskCascade'):
        bgs.build_detector(to_config_dict(model_cfg), train_cfg=to_config_dict(train_cfg),
                           test_cfg=None)


def test_e2e_golden_fixture_and_seeded_weights_are_reproducible(tmp_path):
    """tests/golden/e2e_inference_golden.npz (executed reference detectors) is present and the
    seeded filler writes the same values into this package's modules on every call."""
    from oracle import det_oracle
    from tests.golden import make_golden_e2e as E
    z = np.load(os.path.join(os.path.dirname(E.__file__), 'e2e_inference_golden.npz'))
    assert z['frcnn/det_bboxes'].shape == (50, 5) and z['htc/mask_probs'].shape == (50, 28, 28)
    assert z['frcnn/proposals'].shape[1] == 5 and z['htc/cls_score2'].shape[1] == 1236
    from balancedgroupsoftmax_amd.config import to_config_dict
    a = bgs.build_detector(to_config_dict(E.configs(str(tmp_path), 'frcnn')), train_cfg=None,
                           test_cfg=to_config_dict(E.TEST_CFG))
    b = bgs.build_detector(to_config_dict(E.configs(str(tmp_path), 'frcnn')), train_cfg=None,
                           test_cfg=to_config_dict(E.TEST_CFG))
    with torch.no_grad():
        det_oracle.fill_detector(a.state_dict(), E.FRCNN_SEED)
        det_oracle.fill_detector(b.state_dict(), E.FRCNN_SEED)
    sa, sb = a.state_dict(), b.state_dict()
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
    assert float(sa['backbone.layer1.0.bn1.running_var'].min()) >= 0.5
    assert E.image().shape == (1, 3, E.H, E.W)
