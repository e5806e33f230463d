"""GPU: whole test-time path against the EXECUTED REFERENCE DETECTORS (tests/golden/
make_golden_e2e.py: the reference's GroupSoftmax Faster R-CNN R50-FPN and HybridTaskCascade run on
CPU with their ops bound to the reference's own nms_cpu.cpp / roi_align_kernel.cu built for the
host).  Same seeded state_dict, same seeded image; every stage is compared: FPN maps, proposals,
RoI features, head outputs, merged scores, final detections, ensemble masks.

Tolerances: the GPU convs sum in a different order than torch's CPU kernels; through ~60 layers
the maps agree to ~1e-5 of their range (asserted: 2e-4).  Discrete stages (top-k, NMS) are compared
as sets with a small allowance for decisions that sit within that noise of a threshold."""
import os
import tempfile

import numpy as np
import pytest
import torch

import balancedgroupsoftmax_amd as bgs
from balancedgroupsoftmax_amd.config import to_config_dict
from oracle import det_oracle
from tests.golden import make_golden_e2e as G

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
GOLD = os.path.join(os.path.dirname(G.__file__), 'e2e_inference_golden.npz')


def _build(which, seed):
    tmp = tempfile.mkdtemp(prefix='bgs_e2e_')
    model = bgs.build_detector(to_config_dict(G.configs(tmp, which)), train_cfg=None,
                               test_cfg=to_config_dict(G.TEST_CFG))
    with torch.no_grad():
        det_oracle.fill_detector(model.state_dict(), seed)
    return model.to(DEV).eval()


def nchw(t):
    return t.permute(0, 3, 1, 2)


def relmax(got, exp):
    return float(np.abs(got - exp).max() / max(np.abs(exp).max(), 1e-12))


def match_boxes(got, exp, tol_px=0.05, tol_score=2e-4):
    """fraction of rows of ``exp [n,5]`` that have a row of ``got`` within tol (box px, score)."""
    if len(exp) == 0:
        return 1.0
    d = np.abs(got[None, :, :4] - exp[:, None, :4]).max(axis=2)
    s = np.abs(got[None, :, 4] - exp[:, None, 4])
    return float(((d < tol_px) & (s < tol_score)).any(axis=1).mean())


def test_faster_rcnn_r50_bags_vs_executed_reference_detector():
    z = np.load(GOLD)
    model = _build('frcnn', G.FRCNN_SEED)
    img = torch.from_numpy(G.image()).to(DEV)
    meta = G.img_meta()
    with torch.no_grad():
        x = model.extract_feat(img)
        for i, f in enumerate(x):
            assert relmax(nchw(f)[:, ::16].cpu().numpy(), z['frcnn/p%d' % i]) < 2e-4, i
        # RPN: own proposals vs the reference's (top-k + per-level NMS + top-k)
        props, valid = model.simple_test_rpn(x, meta, model.test_cfg.rpn)[0]
        mine = props[valid].cpu().numpy()
        ref = z['frcnn/proposals']
        assert abs(len(mine) - len(ref)) <= 3
        assert match_boxes(mine, ref, tol_px=0.02, tol_score=1e-5) > 0.97
        assert match_boxes(ref, mine, tol_px=0.02, tol_score=1e-5) > 0.97
        # RoI head on the REFERENCE's proposals (decouples the head from proposal ordering)
        rp = torch.from_numpy(ref).to(DEV)
        rois = torch.cat([rp.new_zeros((rp.size(0), 1)), rp[:, :4]], dim=1)
        feats = model.bbox_roi_extractor(x[:4], rois)
        assert relmax(nchw(feats)[::10, ::16].cpu().numpy(), z['frcnn/roi_feats']) < 2e-4
        cls_score, bbox_pred = model.bbox_head(feats, nhwc=True)
        assert relmax(cls_score[::8].cpu().numpy(), z['frcnn/cls_score']) < 2e-4
        assert relmax(bbox_pred[::4, ::41].cpu().numpy(), z['frcnn/bbox_pred']) < 2e-4
        db, dl, scores = model.simple_test_bboxes(x, meta, [rp], model.test_cfg.rcnn)
        assert np.abs(scores[::4, ::7].cpu().numpy() - z['frcnn/scores']).max() < 2e-5
        # final detections: same (label, box, score) set
        got = np.concatenate([db.cpu().numpy(), dl.cpu().numpy()[:, None].astype(np.float32)], 1)
        exp = np.concatenate([z['frcnn/det_bboxes'], z['frcnn/det_labels'][:, None].astype(np.float32)], 1)
        assert got.shape == exp.shape == (50, 6)
        hit = 0
        for e in exp:
            same = got[got[:, 5] == e[5]]
            hit += bool(len(same) and (np.abs(same[:, :4] - e[:4]).max(axis=1) < 0.05).any()
                        and np.abs(same[:, 4] - e[4]).min() < 2e-5)
        assert hit >= 48, hit
        # and end to end with its own proposals: the same detections again
        res = model.simple_test(img, meta)
        assert len(res) == 1230 and sum(r.shape[0] for r in res) == 50
        own = np.concatenate([np.concatenate([r, np.full((r.shape[0], 1), c, np.float32)], 1)
                              for c, r in enumerate(res) if r.shape[0]])
        hit = sum(bool((np.abs(own[own[:, 5] == e[5] - 0][:, :4] - e[:4]).max(axis=1) < 0.05).any())
                  for e in exp if (own[:, 5] == e[5]).any())
        assert hit >= 46, hit


def test_htc_vs_executed_reference_detector():
    z = np.load(GOLD)
    model = _build('htc', G.HTC_SEED)
    img = torch.from_numpy(G.image()).to(DEV)
    meta = G.img_meta()
    with torch.no_grad():
        x = model.extract_feat(img)
        _, sem = model.semantic_head(x)
        assert relmax(nchw(sem)[:, ::16].cpu().numpy(), z['htc/semantic_feat']) < 2e-4
        props, valid = model.simple_test_rpn(x, meta, model.test_cfg.rpn)[0]
        mine = props[valid].cpu().numpy()
        ref = z['htc/proposals']
        assert abs(len(mine) - len(ref)) <= 3 and match_boxes(mine, ref, 0.02, 1e-5) > 0.97
        # the three box stages on the reference's proposals (htc.py:331-365)
        rp = torch.from_numpy(ref).to(DEV)
        rois = torch.cat([rp.new_zeros((rp.size(0), 1)), rp[:, :4]], dim=1)
        for i in range(3):
            head, ext = model.bbox_head[i], model.bbox_roi_extractor[i]
            cls_score, bbox_pred = head(model._fused_roi_feats(ext, x, rois, sem, 'bbox'), nhwc=True)
            assert relmax(cls_score[::16].cpu().numpy(), z['htc/cls_score%d' % i]) < 5e-4, i
            if i < 2:
                rois = head.regress_by_class(rois, cls_score.argmax(dim=1), bbox_pred, meta[0])
        # a RoI whose two best classes are within noise regresses with the other class' delta
        # (class-agnostic heads here: regress_by_class ignores the label) -> rois must agree
        assert np.abs(rois.cpu().numpy() - z['htc/final_rois']).max() < 0.05
        db, dl, masks = model.simple_test_dets(img, meta, proposals=[rp])
        got = np.concatenate([db.cpu().numpy(), dl.cpu().numpy()[:, None].astype(np.float32)], 1)
        exp = np.concatenate([z['htc/det_bboxes'], z['htc/det_labels'][:, None].astype(np.float32)], 1)
        assert got.shape == exp.shape == (50, 6) and tuple(masks.shape) == (50, 28, 28)
        m = masks.cpu().numpy()
        hit, worst = 0, 0.0
        for k, e in enumerate(exp):
            j = np.nonzero((got[:, 5] == e[5]) & (np.abs(got[:, :4] - e[:4]).max(axis=1) < 0.05)
                           & (np.abs(got[:, 4] - e[4]) < 2e-5))[0]
            if len(j):
                hit += 1
                worst = max(worst, float(np.abs(m[j[0]] - z['htc/mask_probs'][k]).max()))
        assert hit >= 48, hit
        assert worst < 2e-3, worst          # ensemble mask probability of the matched detections
