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

# The bf16 mode of cfg[4] (operands rounded to bf16, fp32 accumulate, fp32 losses: the reference's fp16 contract,
# mmdet/core/fp16/hooks.py:40-94) against the fp32 goldens of the executed reference, round 5: every loss term within
# BF16_TERM_REL of ITSELF (+ BF16_TERM_ABS), the HIP RPN's proposals reproduced by IoU, and the fc_cls / fc_reg
# gradients within bf16_grad_bound() in relative L2.
#   Measured (profiles/r9f_bf16_parity_measurements.md): worst term ratio 0.17 - 0.43 of the budget; proposals —
#   bf16 operands move the RPN's regression deltas by up to a few per cent of the anchor size, so a fixed pixel / score
#   tolerance is the wrong ruler (0.09 - 0.93 of the reference's proposals within 1 - 2 px, depending on the box
#   sizes of the configuration): the pin is by IoU — 0.865 - 0.94 of them have a HIP proposal with IoU >= 0.7 (0.21 -
#   0.74 at IoU >= 0.9); gradients: stage-0 fc_cls 3e-3 - 9e-3, fc_reg 1.0e-2 - 1.9e-2, and the LATER cascade stages
#   (whose RoIs are the previous stage's bf16-regressed boxes) 1e-2 - 5.5e-2.
BF16_TERM_REL, BF16_TERM_ABS = 5e-2, 1e-3
BF16_PROP_IOU, BF16_MIN_FRAC = 0.7, 0.85
BF16_GRAD_L2_STAGE0, BF16_GRAD_L2_LATER = 2e-2, 8e-2


def bf16_grad_bound(name):
    """relative-L2 bound of a head gradient in the bf16 mode against the fp32 golden (see above)."""
    later = any(t in name for t in ('bbox_head.1.', 'bbox_head.2.', 'mask_head.1.', 'mask_head.2.'))
    if 'fc_reg' in name:
        return 3e-2 if not later else BF16_GRAD_L2_LATER
    return BF16_GRAD_L2_LATER if later else BF16_GRAD_L2_STAGE0
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


def _iou_legacy(a, b):
    a, b = a[:, None, :4].astype(np.float64), b[None, :, :4].astype(np.float64)
    iw = np.clip(np.minimum(a[..., 2], b[..., 2]) - np.maximum(a[..., 0], b[..., 0]) + 1, 0, None)
    ih = np.clip(np.minimum(a[..., 3], b[..., 3]) - np.maximum(a[..., 1], b[..., 1]) + 1, 0, None)
    inter = iw * ih
    area = lambda t: (t[..., 2] - t[..., 0] + 1) * (t[..., 3] - t[..., 1] + 1)      # noqa: E731
    return inter / (area(a) + area(b) - inter)


def proposal_miss_taxonomy(got, exp, tol_px=0.05, tol_score=2e-4, nms_thr=0.7):
    """Why a reference proposal has no HIP proposal within tolerance (printed when the reproduced fraction is below
    1): ``loose`` = a HIP row within 10x the tolerance (arithmetic noise on a coordinate or score), ``nms_tie`` = a kept
    HIP box overlaps it with an IoU within 2e-3 of the NMS threshold (the suppression decision sits on the threshold),
    ``topk_tie`` = its score is within the score tolerance of the lowest reference score (the cut of nms_pre / max_num),
    ``other`` = none of these — a real difference."""
    d = np.abs(got[None, :, :4] - exp[:, None, :4]).max(axis=2)
    sc = np.abs(got[None, :, 4] - exp[:, None, 4])
    hit = ((d < tol_px) & (sc < tol_score)).any(axis=1)
    out = dict(missing=int((~hit).sum()), loose=0, nms_tie=0, topk_tie=0, other=0)
    if hit.all():
        return out
    miss = np.where(~hit)[0]
    iou = _iou_legacy(exp[miss], got)
    lo = float(exp[:, 4].min())
    for k, m in enumerate(miss):
        if ((d[m] < 10 * tol_px) & (sc[m] < 10 * tol_score)).any():
            out['loose'] += 1
        elif (np.abs(iou[k] - nms_thr) < 2e-3).any():
            out['nms_tie'] += 1
        elif abs(float(exp[m, 4]) - lo) < tol_score:
            out['topk_tie'] += 1
        else:
            out['other'] += 1
    return out


def assert_proposals_reproduced(got, exp, image, min_frac=0.999):
    """The HIP RPN reproduces the executed reference's proposals: every full-size run so far logged 1.0000, so the bound is
    0.999 (2 of 2000 — a top-k / NMS tie may flip; 0.97 would have let a real regression of 3 % through), and a miss
    prints its taxonomy."""
    frac = match_boxes(got, exp, tol_px=0.05, tol_score=2e-4)
    print('image %d: %.4f of the reference proposals reproduced by the HIP RPN' % (image, frac))
    if frac < 1.0:
        print('image %d: misses by cause: %s' % (image, proposal_miss_taxonomy(got, exp)))
    assert frac >= min_frac, (frac, proposal_miss_taxonomy(got, exp))
    return frac


def match_iou(got, exp, thr=0.9):
    """fraction of rows of ``exp [n,>=4]`` that overlap a row of ``got`` with IoU >= thr (legacy +1 areas): the
    size-relative match for arithmetic modes that move box coordinates by more than a fixed pixel tolerance."""
    if len(exp) == 0:
        return 1.0
    a, b = exp[:, None, :4].astype(np.float64), got[None, :, :4].astype(np.float64)
    iw = np.clip(np.minimum(a[..., 2], b[..., 2]) - np.maximum(a[..., 0], b[..., 0]) + 1, 0, None)
    ih = np.clip(np.minimum(a[..., 3], b[..., 3]) - np.maximum(a[..., 1], b[..., 1]) + 1, 0, None)
    inter = iw * ih
    area = lambda t: (t[..., 2] - t[..., 0] + 1) * (t[..., 3] - t[..., 1] + 1)      # noqa: E731
    iou = inter / (area(a) + area(b) - inter)
    return float((iou >= thr).any(axis=1).mean())


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


GRAD_L2_SEEN = []      # (rel-L2 of every gradient tensor compared in this process: printed by the tests under -s)


def grad_close(a, b, tol=2e-4, frac=0.8, worst=3e-2, l2tol=3e-3):
    """Gradients travel through up to ~50 ReLU layers: a pre-activation within fp32 noise of zero
    takes the other branch than in the torch-CPU run (a handful of entries move by up to ~1 % of
    the largest one); a wrong kernel is off by O(1) everywhere.  Relative L2 per tensor: <= 3e-3 in the small-size
    tests (round 5; 1e-2 before), <= 1e-3 where the callers at the full size ask for it."""
    rel = np.abs(a - b) / max(np.abs(b).max(), 1e-20)
    l2 = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-20))
    GRAD_L2_SEEN.append(l2)
    ok = (rel < tol).mean() > frac and rel.max() < worst and l2 < l2tol
    if not ok:
        print('grad_close: %.4f within tol, worst %.3e, rel-L2 %.3e' % ((rel < tol).mean(), rel.max(), l2))
    return ok


def test_training_iteration_vs_executed_reference_detector():
    """Full ``forward_train`` + ``backward`` of the BASELINE config against the executed reference
    (tests/golden/make_golden_train.py): every loss term and gradients from the head down to
    layer2 of the trunk.  The samplers are configured to take every candidate (no random draw on
    either side); everything else is the shipped configuration."""
    from balancedgroupsoftmax_amd import train
    from tests.golden import make_golden_train as T
    z = np.load(os.path.join(os.path.dirname(T.__file__), 'e2e_train_golden.npz'))
    tmp = tempfile.mkdtemp(prefix='bgs_e2e_')
    model_cfg, train_cfg = T.configs(tmp)
    model = bgs.build_detector(to_config_dict(model_cfg), train_cfg=to_config_dict(train_cfg),
                               test_cfg=to_config_dict(G.TEST_CFG))
    with torch.no_grad():
        det_oracle.fill_detector(model.state_dict(), T.SEED)
    model.to(DEV)
    train.select_training_param(model, 0)
    model.train()
    boxes, labels = T.gt()
    losses = model(torch.from_numpy(G.image()).to(DEV), G.img_meta(), return_loss=True,
                   gt_bboxes=[torch.from_numpy(boxes).to(DEV)],
                   gt_labels=[torch.from_numpy(labels).to(DEV)])
    for k in ('loss_rpn_cls', 'loss_rpn_bbox'):
        got = np.array([float(t.detach().sum()) for t in losses[k]], np.float32)
        assert np.abs(got - z['loss/' + k]).max() < 1e-4 * max(1.0, np.abs(z['loss/' + k]).max()), \
            (k, got, z['loss/' + k])
    for k in ['loss_cls_bin%d' % i for i in range(5)] + ['loss_bbox']:
        got, exp = float(losses[k].detach().sum()), float(z['loss/' + k][0])
        assert abs(got - exp) < 1e-4 * max(1.0, abs(exp)), (k, got, exp)
    loss, _ = train.parse_losses(losses)
    assert abs(float(loss.detach()) - float(z['loss/total'][0])) < 1e-4 * float(z['loss/total'][0])
    loss.backward()
    params = dict(model.named_parameters())
    assert params['backbone.layer1.0.conv1.weight'].grad is None
    bad = []
    for name, idx in T.GRADS:
        g = params[name].grad
        assert g is not None, name
        if not grad_close(g[idx].cpu().numpy(), z['grad/' + name]):
            bad.append(name)
    assert not bad, bad


def test_side_stream_forks_do_not_change_the_iteration(monkeypatch):
    """functional.forked (small pyramid levels, projection shortcuts, fc_reg, the RPN loss chain on a side stream):
    the same kernels on the same inputs — every loss term of a frozen-trunk iteration (selectp = 1, the shipped
    regime, where all forks are live) is BIT-IDENTICAL with every fork off, with the default set, and with the RPN
    loss chain forced onto the side stream; three repetitions per arm (a missing join or a reuse race would show as
    run-to-run differences)."""
    from balancedgroupsoftmax_amd import train, functional as BF
    from tests.golden import make_golden_train as T
    tmp = tempfile.mkdtemp(prefix='bgs_e2e_')
    model_cfg, train_cfg = T.configs(tmp)
    model = bgs.build_detector(to_config_dict(model_cfg), train_cfg=to_config_dict(train_cfg),
                               test_cfg=to_config_dict(G.TEST_CFG))
    with torch.no_grad():
        det_oracle.fill_detector(model.state_dict(), T.SEED)
    model.to(DEV)
    train.select_training_param(model, 1)
    model.train()
    boxes, labels = T.gt()
    img = torch.from_numpy(G.image()).to(DEV)
    gtb, gtl = [torch.from_numpy(boxes).to(DEV)], [torch.from_numpy(labels).to(DEV)]

    def run():
        for c in BF._KEY_COUNTERS.values():        # every repetition replays the same sampler draws
            c.zero_()
        model.bbox_head._draw.zero_()
        losses = model(img, G.img_meta(), return_loss=True, gt_bboxes=gtb, gt_labels=gtl)
        loss, _ = train.parse_losses(losses)
        loss.backward()
        out = {}
        for k, v in losses.items():
            vs = v if isinstance(v, (list, tuple)) else [v]
            out[k] = torch.stack([t.detach().float().reshape(-1).sum() for t in vs]).cpu()
        out['fc_cls.grad'] = model.bbox_head.fc_cls.weight.grad.detach().clone().cpu()
        model.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        return out

    arms = {'off': dict(BGS_LEVEL_FORK='0'), 'default': {}, 'rpn_loss': dict(BGS_RPN_LOSS_FORK='1'),
            'no_shortcut': dict(BGS_SHORTCUT_FORK='0')}
    ref = None
    for name, env in arms.items():
        for k in ('BGS_LEVEL_FORK', 'BGS_RPN_LOSS_FORK', 'BGS_SHORTCUT_FORK'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for rep in range(3):
            got = run()
            if ref is None:
                ref = got
            assert got.keys() == ref.keys()
            for k in ref:
                assert torch.equal(got[k], ref[k]), (name, rep, k)


def test_trunk_pipeline_soak_at_full_size_without_host_synchronisation():
    """tools/pipe_soak.py: 60 optimizer steps of the bench's full-size cfg[1] step over two alternating batches with NO
    host synchronisation inside the loop — sequential loop vs ``train.TrunkPipeline`` at depths 4, 5, 3, 5 — end in
    bit-identical ``fc_cls`` parameters and last-step losses (the six-step test below reads the losses back every step,
    which also orders the streams; this one leaves the caching allocator and the piece streams to themselves)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'pipe_soak.py'), '60'], capture_output=True, text=True,
                       timeout=900, cwd=root)
    assert r.returncode == 0 and 'SOAK OK' in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
    assert r.stdout.count('identical parameters and last losses: True') == 4


def test_trunk_pipeline_reproduces_the_sequential_training_loop():
    """``train.TrunkPipeline`` (round 5): with a frozen trunk (the shipped selectp = 1) the features of batch i + 1 are
    computed on their own streams (depth 2: the whole trunk one batch ahead; 3: backbone two ahead | FPN one ahead; 4:
    the backbone in two pieces) while batch i's heads / losses / backward / optimizer step run.  Six optimizer steps
    over two different batches, sequential vs pipelined from the same initial state and sampler counters: every loss
    term of every step and the final ``fc_cls`` parameters are BIT-IDENTICAL (a missing join, a reuse race on the
    feature maps or a trunk that saw an updated parameter would all show); a trainable trunk is refused."""
    from balancedgroupsoftmax_amd import train, functional as BF
    from tests.golden import make_golden_train as T
    tmp = tempfile.mkdtemp(prefix='bgs_e2e_')
    model_cfg, train_cfg = T.configs(tmp)
    model = bgs.build_detector(to_config_dict(model_cfg), train_cfg=to_config_dict(train_cfg),
                               test_cfg=to_config_dict(G.TEST_CFG))
    with torch.no_grad():
        det_oracle.fill_detector(model.state_dict(), T.SEED)
    model.to(DEV)
    params = train.select_training_param(model, 1)
    model.train()
    assert model.trunk_is_frozen()
    init = [p.detach().clone() for p in params]
    boxes, labels = T.gt()
    img_a = torch.from_numpy(G.image()).to(DEV)
    img_b = torch.flip(img_a, dims=[3]).contiguous() * 0.9 + 0.05
    gtb, gtl = [torch.from_numpy(boxes).to(DEV)], [torch.from_numpy(labels).to(DEV)]
    batches = [img_a, img_b, img_a, img_b, img_b, img_a]

    def run(pipelined):
        with torch.no_grad():
            for p, v in zip(params, init):
                p.copy_(v)
        for c in BF._KEY_COUNTERS.values():
            c.zero_()
        model.bbox_head._draw.zero_()
        opt = train.build_optimizer(params, dict(type='SGD', lr=0.02, momentum=0.9, weight_decay=1e-4))
        step = train.DistOptimizerStep(params, opt, dict(max_norm=35, norm_type=2), world_size=1)
        pipe = train.TrunkPipeline(model, depth=pipelined) if pipelined else None
        if pipe:
            assert pipe.depth == pipelined
            for k in range(pipe.depth - 1):
                pipe.push(batches[k])
        out = []
        for i, img in enumerate(batches):
            feats = None
            if pipe:
                feats = pipe.take()
                nxt = i + pipe.depth - 1
                pipe.push(batches[nxt] if nxt < len(batches) else None)      # (None: the loader has run dry — flush)
            losses = model(img, G.img_meta(), return_loss=True, gt_bboxes=gtb, gt_labels=gtl, feats=feats)
            loss, _ = train.parse_losses(losses)
            step(loss)
            rec = {}
            for k, v in losses.items():
                vs = v if isinstance(v, (list, tuple)) else [v]
                rec[k] = torch.stack([t.detach().float().reshape(-1).sum() for t in vs]).cpu()
            out.append(rec)
        torch.cuda.synchronize()
        return out, [p.detach().clone().cpu() for p in params]

    seq, wseq = run(0)
    for depth in (2, 3, 4, 3):
        pip, wpip = run(depth)
        rep = depth
        for i, (a, b) in enumerate(zip(seq, pip)):
            assert a.keys() == b.keys()
            for k in a:
                assert torch.equal(a[k], b[k]), (rep, i, k)
        for a, b in zip(wseq, wpip):
            assert torch.equal(a, b)
    assert not torch.equal(wseq[0], init[0].cpu())                 # the steps did update fc_cls
    assert not torch.equal(seq[0]['loss_bbox'], seq[1]['loss_bbox'])          # and the two batches differ
    # the process-wide switches belong to the pipeline that set them: another instance (or its __del__) leaves them alone
    p1 = train.TrunkPipeline(model, depth=2)
    p1.push(img_a)
    p2 = train.TrunkPipeline(model, depth=2)
    assert BF._PIPELINE_ACTIVE[0] == id(p1) and not BF.level_fork_enabled()
    p2._deactivate()
    del p2
    assert BF._PIPELINE_ACTIVE[0] == id(p1)
    p1.drain()
    assert BF._PIPELINE_ACTIVE[0] == 0
    torch.cuda.synchronize()
    next(model.backbone.layer4.parameters()).requires_grad = True       # a trunk that trains: refused
    assert not model.trunk_is_frozen()
    with pytest.raises(ValueError):
        train.TrunkPipeline(model)


@pytest.mark.parametrize('mode', ['bf16x6', 'bf16x6-nohalo', 'f32', 'f32-nohalo'])
def test_fullsize_training_iteration_vs_executed_reference(mode, monkeypatch):
    """BASELINE cfg[1] AT ITS REAL SIZE (2 x 3x800x1344, 20 GT/img) against the executed reference
    (tests/golden/make_golden_fullsize.py): the kernels and tile instantiations the size-based
    dispatch only reaches here (128x128 / 128x64 tiles, halo kernels, > 2000-workgroup XCD-banded
    grids, 268,569 anchors per image) produce the reference's 8 loss terms to 1e-4 and its
    gradients from ``fc_cls`` down to ResNet layer2 — under both conv arithmetic modes, with and
    without the halo kernels.  Samplers take every candidate on both sides (no random draw)."""
    from balancedgroupsoftmax_amd import functional as BF
    from balancedgroupsoftmax_amd import train
    from tests.golden import make_golden_fullsize as T
    z = np.load(os.path.join(os.path.dirname(T.__file__), 'e2e_train_fullsize_golden.npz'))
    math, _, nohalo = mode.partition('-')
    if nohalo:
        monkeypatch.setenv('BGS_CONV_HALO', '0')
    prev = BF.set_conv_math(math)
    try:
        tmp = tempfile.mkdtemp(prefix='bgs_full_')
        model_cfg, train_cfg = T.configs(tmp)
        model = bgs.build_detector(to_config_dict(model_cfg), train_cfg=to_config_dict(train_cfg),
                                   test_cfg=to_config_dict(G.TEST_CFG))
        with torch.no_grad():
            det_oracle.fill_detector(model.state_dict(), T.SEED)
        model.to(DEV)
        train.select_training_param(model, 0)
        model.train()
        boxes, labels = T.gt()
        losses = model(T.image().to(DEV), T.img_meta(), return_loss=True,
                       gt_bboxes=[torch.from_numpy(b).to(DEV) for b in boxes],
                       gt_labels=[torch.from_numpy(l).to(DEV) for l in labels])
        bad = []
        for k in ['loss_rpn_cls', 'loss_rpn_bbox'] + ['loss_cls_bin%d' % i for i in range(5)] + ['loss_bbox']:
            v = losses[k]
            got = np.array([float(t.detach().sum()) for t in (v if isinstance(v, list) else [v])], np.float32)
            exp = z['loss/' + k]
            if np.abs(got - exp).max() > 1e-4 * np.abs(exp).max() + 1e-7:
                bad.append((k, got.tolist(), exp.tolist()))
        assert not bad, bad
        loss, _ = train.parse_losses(losses)
        assert abs(float(loss.detach()) - float(z['loss/total'][0])) < 1e-4 * float(z['loss/total'][0])
        loss.backward()
        params = dict(model.named_parameters())
        bad = []
        for name, idx in T.GRADS:
            g = params[name].grad
            assert g is not None, name
            if not grad_close(g[idx].cpu().numpy(), z['grad/' + name], l2tol=1e-3):    # full size: rel-L2 <= 1e-3 per tensor
                bad.append(name)
        print('full-size gradients: worst rel-L2 %.2e' % max(GRAD_L2_SEEN[-len(T.GRADS):]))
        assert not bad, bad
    finally:
        BF.set_conv_math(prev)
        del model
        torch.cuda.empty_cache()


def test_cascade_bf16_mode_vs_fp32_golden_and_fp16_optimizer_step():
    """BASELINE cfg[4] "bf16": ``train.wrap_fp16_model`` (conv / linear operands rounded to bf16 in
    the MFMA kernels, fp32 accumulate and storage) on the 3-stage Cascade R-CNN against the
    EXECUTED fp32 reference iteration (the golden of the fp32 test above): every loss term within
    the bf16 budget (2 % of the total; discrete NMS / assignment decisions may move), and
    ``Fp16OptimizerStep`` (loss scale 512, mmdet/core/fp16/hooks.py:58-83) produces the same update
    as the unscaled step."""
    from balancedgroupsoftmax_amd import functional as BF
    from balancedgroupsoftmax_amd import train
    from tests.golden import make_golden_train as T
    z = np.load(os.path.join(os.path.dirname(T.__file__), 'e2e_train_golden.npz'))
    tmp = tempfile.mkdtemp(prefix='bgs_e2e_')
    model_cfg, train_cfg = T.configs(tmp, cascade=True)

    def build():
        m = bgs.build_detector(to_config_dict(model_cfg), train_cfg=to_config_dict(train_cfg),
                               test_cfg=to_config_dict(G.TEST_CFG))
        with torch.no_grad():
            det_oracle.fill_detector(m.state_dict(), T.CASCADE_SEED)
        m.to(DEV)
        m.train()
        return m

    boxes, labels = T.gt()

    def run(m):
        return m(torch.from_numpy(G.image()).to(DEV), G.img_meta(), return_loss=True,
                 gt_bboxes=[torch.from_numpy(boxes).to(DEV)],
                 gt_labels=[torch.from_numpy(labels).to(DEV)])

    model = build()
    params = train.select_training_param(model, 3)
    outside = train.wrap_fp16_model(model, 'bf16')
    try:
        # the mode is scoped to the wrapped model (as model.half() is): unchanged outside its forward
        assert BF.conv_math() == outside != 'bf16' and model._conv_math == 'bf16'
        seen = []
        probe = model.backbone.register_forward_pre_hook(lambda m, a: seen.append(BF.conv_math()))
        losses = run(model)
        probe.remove()
        assert seen == ['bf16'] and BF.conv_math() == outside
        total_exp = float(z['cascade/loss/total'][0])
        worst = 0.0
        for k, v in losses.items():
            got = np.array([float(t.detach().sum()) for t in (v if isinstance(v, list) else [v])])
            exp = z['cascade/loss/' + k]
            worst = max(worst, float(np.abs(got - exp).max()))
        loss, _ = train.parse_losses(losses)
        print('bf16 cascade: total %.5f vs fp32 reference %.5f, worst term diff %.2e'
              % (float(loss.detach()), total_exp, worst))
        assert abs(float(loss.detach()) - total_exp) < 2e-2 * total_exp
        assert worst < 2e-2 * total_exp
        # scaled vs unscaled optimizer step from the same state: identical update (512 = 2^9)
        opt_a = torch.optim.SGD(params, lr=0.01, momentum=0.9, weight_decay=1e-4)
        w0 = [p.detach().clone() for p in params]
        train.Fp16OptimizerStep(params, opt_a, grad_clip=dict(max_norm=35, norm_type=2),
                                loss_scale=512.0)(loss)
        wa = [p.detach().clone() for p in params]
        with torch.no_grad():
            for p, w in zip(params, w0):
                p.copy_(w)
        opt_b = torch.optim.SGD(params, lr=0.01, momentum=0.9, weight_decay=1e-4)
        loss_b, _ = train.parse_losses(run(model))
        train.DistOptimizerStep(params, opt_b, grad_clip=dict(max_norm=35, norm_type=2))(loss_b)
        for a, p, w in zip(wa, params, w0):
            step = (p.detach() - w).abs().max()
            assert float(step) > 0
            assert float((a - p.detach()).abs().max()) <= 1e-5 * float(step) + 1e-9
    finally:
        train.unwrap_fp16_model(model)
        assert BF.conv_math() == outside


def test_htc_training_iteration_vs_executed_reference_detector():
    """``HybridTaskCascade.forward_train`` + ``backward`` (htc.py:196-311: three box stages with
    semantic fusion, interleaved re-sampling from the refined boxes, mask information flow, semantic
    loss) against the executed reference on a ResNet-50 trunk: all 27 loss terms and gradients of
    the box / mask / semantic heads, FPN, RPN and trunk.  Deterministic sampling as above; the mask
    targets' bitmap resize on the reference side is oracle/mask_oracle.py's restatement of
    OpenCV (OpenCV itself is not installed: that one step is unpinned)."""
    from balancedgroupsoftmax_amd import train
    from tests.golden import make_golden_train as T
    z = np.load(os.path.join(os.path.dirname(T.__file__), 'e2e_train_golden.npz'))
    tmp = tempfile.mkdtemp(prefix='bgs_e2e_')
    model_cfg, train_cfg = T.configs(tmp, htc=True)
    model = bgs.build_detector(to_config_dict(model_cfg), train_cfg=to_config_dict(train_cfg),
                               test_cfg=to_config_dict(G.TEST_CFG))
    with torch.no_grad():
        det_oracle.fill_detector(model.state_dict(), T.HTC_SEED)
    model.to(DEV)
    train.select_training_param(model, 0)
    model.train()
    boxes, labels = T.gt()
    losses = model(torch.from_numpy(G.image()).to(DEV), G.img_meta(), return_loss=True,
                   gt_bboxes=[torch.from_numpy(boxes).to(DEV)],
                   gt_labels=[torch.from_numpy(labels).to(DEV)],
                   gt_masks=[torch.from_numpy(T.gt_masks(boxes)).to(DEV)],
                   gt_semantic_seg=torch.from_numpy(T.gt_semantic_seg()).to(DEV))
    keys = [k[len('htc/loss/'):] for k in z.files if k.startswith('htc/loss/') and not k.endswith('total')]
    assert set(keys) == set(losses.keys())
    bad = []
    for k in keys:
        v = losses[k]
        got = np.array([float(t.detach().sum()) for t in (v if isinstance(v, list) else [v])], np.float32)
        exp = z['htc/loss/' + k]
        if np.abs(got - exp).max() > 2e-4 * max(1.0, np.abs(exp).max()):
            bad.append((k, got.tolist(), exp.tolist()))
    assert not bad, bad
    loss, _ = train.parse_losses(losses)
    loss.backward()
    params = dict(model.named_parameters())
    bad = []
    for name, idx in T.GRADS_HTC:
        g = params[name].grad
        assert g is not None, name
        if not grad_close(g[idx].cpu().numpy(), z['htc/grad/' + name]):
            bad.append(name)
    assert not bad, bad


def test_mask_rcnn_training_iteration_vs_executed_reference_detector():
    """cfg[3] (gs_mask_rcnn_r50_fpn_1x_lvis): ``MaskRCNN.forward_train`` + ``backward`` against the
    executed reference — the 18 loss terms incl. ``loss_mask`` and gradients of the mask head
    (the single-channel logits / BCE kernels vs the reference's 1231-channel conv + gather), box
    head, FPN and trunk."""
    from balancedgroupsoftmax_amd import train
    from tests.golden import make_golden_train as T
    z = np.load(os.path.join(os.path.dirname(T.__file__), 'e2e_train_golden.npz'))
    tmp = tempfile.mkdtemp(prefix='bgs_e2e_')
    model_cfg, train_cfg = T.configs(tmp, mask=True)
    model = bgs.build_detector(to_config_dict(model_cfg), train_cfg=to_config_dict(train_cfg),
                               test_cfg=to_config_dict(G.TEST_CFG))
    with torch.no_grad():
        det_oracle.fill_detector(model.state_dict(), T.MASK_SEED)
    model.to(DEV)
    train.select_training_param(model, 0)
    model.train()
    boxes, labels = T.gt()
    losses = model(torch.from_numpy(G.image()).to(DEV), G.img_meta(), return_loss=True,
                   gt_bboxes=[torch.from_numpy(boxes).to(DEV)],
                   gt_labels=[torch.from_numpy(labels).to(DEV)],
                   gt_masks=[torch.from_numpy(T.gt_masks(boxes)).to(DEV)])
    keys = [k[len('mask/loss/'):] for k in z.files
            if k.startswith('mask/loss/') and not k.endswith('total')]
    assert set(keys) == set(losses.keys())
    bad = []
    for k in keys:
        v = losses[k]
        got = np.array([float(t.detach().sum()) for t in (v if isinstance(v, list) else [v])], np.float32)
        exp = z['mask/loss/' + k]
        if np.abs(got - exp).max() > 2e-4 * max(1.0, np.abs(exp).max()):
            bad.append((k, got.tolist(), exp.tolist()))
    assert not bad, bad
    loss, _ = train.parse_losses(losses)
    loss.backward()
    params = dict(model.named_parameters())
    bad = []
    for name, idx in T.GRADS_MASK:
        g = params[name].grad
        assert g is not None, name
        if not grad_close(g[idx].cpu().numpy(), z['mask/grad/' + name]):
            bad.append(name)
    assert not bad, bad


def test_cascade_rcnn_training_iteration_vs_executed_reference_detector():
    """cfg[4] orchestration (cascade_rcnn.py:152-298: three box stages at IoU 0.5 / 0.6 / 0.7, each
    re-sampling from the boxes the previous stage refined, stage loss weights 1 / 0.5 / 0.25) on
    a ResNet-50 trunk against the executed reference: 28 loss terms + gradients."""
    from balancedgroupsoftmax_amd import train
    from tests.golden import make_golden_train as T
    z = np.load(os.path.join(os.path.dirname(T.__file__), 'e2e_train_golden.npz'))
    tmp = tempfile.mkdtemp(prefix='bgs_e2e_')
    model_cfg, train_cfg = T.configs(tmp, cascade=True)
    model = bgs.build_detector(to_config_dict(model_cfg), train_cfg=to_config_dict(train_cfg),
                               test_cfg=to_config_dict(G.TEST_CFG))
    with torch.no_grad():
        det_oracle.fill_detector(model.state_dict(), T.CASCADE_SEED)
    model.to(DEV)
    train.select_training_param(model, 0)
    model.train()
    boxes, labels = T.gt()
    losses = model(torch.from_numpy(G.image()).to(DEV), G.img_meta(), return_loss=True,
                   gt_bboxes=[torch.from_numpy(boxes).to(DEV)],
                   gt_labels=[torch.from_numpy(labels).to(DEV)])
    keys = [k[len('cascade/loss/'):] for k in z.files
            if k.startswith('cascade/loss/') and not k.endswith('total')]
    assert set(keys) == set(losses.keys())
    bad = []
    for k in keys:
        v = losses[k]
        got = np.array([float(t.detach().sum()) for t in (v if isinstance(v, list) else [v])], np.float32)
        exp = z['cascade/loss/' + k]
        if np.abs(got - exp).max() > 2e-4 * max(1.0, np.abs(exp).max()):
            bad.append((k, got.tolist(), exp.tolist()))
    assert not bad, bad
    loss, _ = train.parse_losses(losses)
    loss.backward()
    params = dict(model.named_parameters())
    bad = [name for name, idx in T.GRADS_CASCADE
           if not grad_close(params[name].grad[idx].cpu().numpy(), z['cascade/grad/' + name])]
    assert not bad, bad


def test_mask_rcnn_test_pass_vs_executed_reference_detector():
    """cfg[3] at test time: detections and each detection's class mask probability
    (test_mixins.py:153-180 up to the sigmoid of the predicted class' channel)."""
    z = np.load(GOLD)
    model = _build('mask', G.MASK_SEED)
    img = torch.from_numpy(G.image()).to(DEV)
    meta = G.img_meta()
    with torch.no_grad():
        x = model.extract_feat(img)
        rp = torch.from_numpy(z['mask/proposals']).to(DEV)
        db, dl, _ = model.simple_test_bboxes(x, meta, [rp], model.test_cfg.rcnn)
        masks = model.simple_test_mask(x, meta, db, dl).cpu().numpy()
    got = np.concatenate([db.cpu().numpy(), dl.cpu().numpy()[:, None].astype(np.float32)], 1)
    exp = np.concatenate([z['mask/det_bboxes'], z['mask/det_labels'][:, None].astype(np.float32)], 1)
    assert got.shape == exp.shape == (50, 6) and masks.shape == (50, 28, 28)
    hit, worst = 0, 0.0
    for k, e in enumerate(exp):
        j = np.nonzero((got[:, 5] == e[5]) & (np.abs(got[:, :4] - e[:4]).max(axis=1) < 0.05)
                       & (np.abs(got[:, 4] - e[4]) < 2e-5))[0]
        if len(j):
            hit += 1
            worst = max(worst, float(np.abs(masks[j[0]] - z['mask/mask_probs'][k]).max()))
    assert hit >= 48, hit
    assert worst < 2e-3, worst


@pytest.mark.parametrize('math', ['bf16x6', 'bf16x6-stemchain', 'f32'])
def test_fullsize_iteration_with_the_shipped_sampler_sizes_and_the_references_recorded_draws(math, monkeypatch):
    """The configuration ``bench.py`` TIMES — cfg[1] at 2 x 3x800x1344 with the shipped sampler
    sizes (RPN 256 of 268,569 anchors at 50 % positives, 512 RoI / image at 25 % positives, "others"
    ratio 8) — against the executed reference (tests/golden/make_golden_shipped.py).  The
    reference's host-side numpy draws (random_sampler.py:19-53, base_sampler.py:31-78,
    gs_bbox_head_with0.py:63-89) were RECORDED while it ran as shipped and are injected through the
    package's sampler hooks; everything else runs on the HIP path.  8 loss terms to 1e-4; the
    ``fc_cls`` / ``fc_reg`` gradients (no ReLU between them and the loss) to 1e-5 of their largest
    entry; trunk gradients with the ReLU-flip-tolerant criterion."""
    from balancedgroupsoftmax_amd import functional as BF
    from balancedgroupsoftmax_amd import train
    from tests.golden import make_golden_fullsize as F
    from tests.golden import make_golden_shipped as T
    z = np.load(os.path.join(os.path.dirname(T.__file__), 'e2e_train_shipped_samplers_golden.npz'))
    # Trunk-gradient criterion.  The executed reference's stem (torch CPU) and the three-launch stem chain accumulate a
    # conv output in the same k order, so the chain reproduces the reference's stem almost bit for bit and the trunk
    # behind it sees no perturbation at all; the fused stem kernel (round 5, the default) sums in another order — the
    # SAME error against fp64 (5e-7 of the scale max, 6e-8 rms: tools/_stem_error.py) — and a few more of the ~10^8
    # pre-activations then sit on the other side of a ReLU: the per-tensor relative L2 of the trunk gradients moves from
    # < 1e-3 to 1.2 - 1.5e-3.  Both arms run: the chain with the 1e-3 bound, the default path with 3e-3.
    stem_chain = math.endswith('-stemchain')
    math = math.split('-')[0]
    if stem_chain:
        monkeypatch.setenv('BGS_STEM_FUSED', '0')
    trunk_l2tol, trunk_frac = (1e-3, 0.8) if (stem_chain or math == 'f32') else (3e-3, 0.5)
    prev = BF.set_conv_math(math)
    model = None
    try:
        tmp = tempfile.mkdtemp(prefix='bgs_shipped_')
        model_cfg, train_cfg = T.configs(tmp)
        assert train_cfg['rpn']['sampler']['num'] == 256 and train_cfg['rcnn']['sampler']['num'] == 512
        assert model_cfg['bbox_head']['gs_config']['others_sample_ratio'] == 8.0
        model_cfg['bbox_head']['gs_config']['sampler'] = 'numpy'      # the reference's draw, replayed below
        model = bgs.build_detector(to_config_dict(model_cfg), train_cfg=to_config_dict(train_cfg),
                                   test_cfg=to_config_dict(G.TEST_CFG))
        with torch.no_grad():
            det_oracle.fill_detector(model.state_dict(), T.SEED)
        model.to(DEV)
        train.select_training_param(model, 0)
        model.train()
        calls = dict(rpn=0, rcnn=0, gs=0)

        def rpn_hook(assigned, num, pos_fraction, neg_pos_ub):
            i = calls['rpn']
            calls['rpn'] += 1
            pos = torch.zeros(assigned.numel(), dtype=torch.bool, device=DEV)
            neg = torch.zeros_like(pos)
            pi = torch.from_numpy(z['rpn/pos%d' % i].astype(np.int64)).to(DEV)
            ni = torch.from_numpy(z['rpn/neg%d' % i].astype(np.int64)).to(DEV)
            # the recorded draws are subsets of THIS path's positive / negative anchors
            assert bool((assigned[pi] > 0).all()) and bool((assigned[ni] == 0).all())
            assert len(pi) + len(ni) == num
            pos[pi] = True
            neg[ni] = True
            return pos, neg

        def rcnn_hook(assigned, num, pos_fraction):
            i = calls['rcnn']
            calls['rcnn'] += 1
            pi = torch.from_numpy(z['rcnn/pos%d' % i].astype(np.int64)).to(DEV)
            ni = torch.from_numpy(z['rcnn/neg%d' % i].astype(np.int64)).to(DEV)
            assert bool((assigned[pi] > 0).all()) and bool((assigned[ni] == 0).all())
            inds = torch.cat([pi, ni])
            assert inds.numel() == num
            is_pos = torch.cat([torch.ones_like(pi), torch.zeros_like(ni)]).bool()
            return inds, is_pos, torch.ones(num, dtype=torch.bool, device=DEV)

        def proposals_hook(own):
            out = []
            for i, (p, v) in enumerate(own):
                ref = torch.from_numpy(z['proposals%d' % i]).to(DEV)
                assert tuple(ref.shape) == tuple(p.shape), (ref.shape, p.shape)
                assert_proposals_reproduced(p[v].cpu().numpy(), z['proposals%d' % i], i)
                out.append((ref.contiguous(), torch.ones(ref.shape[0], dtype=torch.bool, device=DEV)))
            return out

        def choice_replay(a, size=None, replace=True, p=None):
            j = calls['gs']
            calls['gs'] += 1
            assert len(a) == int(z['gs/cand%d' % j][0]) and not replace
            draw = z['gs/draw%d' % j].astype(np.int64)
            assert tuple(np.atleast_1d(size)) == (len(draw),) and np.isin(draw, a).all()
            return draw
        monkeypatch.setattr(np.random, 'choice', choice_replay)

        boxes, labels = F.gt()
        g = torch.Generator().manual_seed(T.SEED)
        img = torch.randn(F.IMGS, 3, F.H, F.W, generator=g)
        losses = model(img.to(DEV), F.img_meta(), return_loss=True,
                       gt_bboxes=[torch.from_numpy(b).to(DEV) for b in boxes],
                       gt_labels=[torch.from_numpy(l).to(DEV) for l in labels],
                       samplers=dict(rpn=rpn_hook, rcnn=rcnn_hook, proposals=proposals_hook))
        assert calls == dict(rpn=2, rcnn=2, gs=4), calls
        bad = []
        for k in ['loss_rpn_cls', 'loss_rpn_bbox'] + ['loss_cls_bin%d' % i for i in range(5)] + ['loss_bbox']:
            v = losses[k]
            got = np.array([float(t.detach().sum()) for t in (v if isinstance(v, list) else [v])], np.float32)
            exp = z['loss/' + k]
            if np.abs(got - exp).max() > 1e-4 * np.abs(exp).max() + 1e-7:
                bad.append((k, got.tolist(), exp.tolist()))
        assert not bad, bad
        loss, _ = train.parse_losses(losses)
        assert abs(float(loss.detach()) - float(z['loss/total'][0])) < 1e-4 * float(z['loss/total'][0])
        loss.backward()
        params = dict(model.named_parameters())
        bad = []
        for name, idx in T.GRADS:
            g = params[name].grad
            assert g is not None, name
            a, b = g[idx].cpu().numpy(), z['grad/' + name]
            if name.startswith(('bbox_head.fc_cls', 'bbox_head.fc_reg')):
                rel = float(np.abs(a - b).max() / np.abs(b).max())
                print('%s: max |diff| / max |g| = %.2e' % (name, rel))
                if rel > 1e-5:
                    bad.append((name, rel))
            elif not grad_close(a, b, l2tol=trunk_l2tol, frac=trunk_frac):     # ReLU-flip allowance + relative L2 per tensor
                bad.append(name)
        assert not bad, bad
    finally:
        BF.set_conv_math(prev)
        del model
        torch.cuda.empty_cache()


def test_mask_rcnn_fullsize_iteration_with_the_shipped_samplers_vs_executed_reference(monkeypatch):
    """BASELINE cfg[3] AT THE SIZE ITS BENCH ROW TIMES: ``gs_mask_rcnn_r50_fpn_1x_lvis`` on 2 x 3x800x1344 with the
    SHIPPED sampler sizes (RPN 256 @ 0.5, RCNN 512 @ 0.25 + GT, "others" ratio 8) and elliptic instance bitmaps,
    against the executed reference (tests/golden/make_golden_mask_htc_fullsize.py: two_stage.py:134-265,
    fcn_mask_head.py:94-123, mask_target.py:7-38).  The reference's recorded numpy draws are replayed through the
    sampler hooks exactly as in the cfg[1] test; the mask branch then sees the same positive RoIs.  9 loss terms
    (``loss_mask`` included) to 1e-4 / 2e-4, head gradients to 1e-5 of their largest entry, trunk gradients with the
    ReLU-flip-tolerant criterion + a relative-L2 bound."""
    from balancedgroupsoftmax_amd import functional as BF
    from balancedgroupsoftmax_amd import train
    from tests.golden import make_golden_mask_htc_fullsize as T
    z = np.load(T.OUT_MASK)
    prev = BF.set_conv_math('bf16x6')
    model = None
    try:
        tmp = tempfile.mkdtemp(prefix='bgs_maskfull_')
        model_cfg, train_cfg = T.mask_configs(tmp)
        assert train_cfg['rpn']['sampler']['num'] == 256 and train_cfg['rcnn']['sampler']['num'] == 512
        model_cfg['bbox_head']['gs_config']['sampler'] = 'numpy'
        model = bgs.build_detector(to_config_dict(model_cfg), train_cfg=to_config_dict(train_cfg),
                                   test_cfg=to_config_dict(G.TEST_CFG))
        with torch.no_grad():
            det_oracle.fill_detector(model.state_dict(), T.MASK_SEED)
        model.to(DEV)
        train.select_training_param(model, 0)
        model.train()
        calls = dict(rpn=0, rcnn=0, gs=0)

        def rpn_hook(assigned, num, pos_fraction, neg_pos_ub):
            i = calls['rpn']
            calls['rpn'] += 1
            pos = torch.zeros(assigned.numel(), dtype=torch.bool, device=DEV)
            neg = torch.zeros_like(pos)
            pi = torch.from_numpy(z['rpn/pos%d' % i].astype(np.int64)).to(DEV)
            ni = torch.from_numpy(z['rpn/neg%d' % i].astype(np.int64)).to(DEV)
            assert bool((assigned[pi] > 0).all()) and bool((assigned[ni] == 0).all())
            assert len(pi) + len(ni) == num
            pos[pi] = True
            neg[ni] = True
            return pos, neg

        def rcnn_hook(assigned, num, pos_fraction):
            i = calls['rcnn']
            calls['rcnn'] += 1
            pi = torch.from_numpy(z['rcnn/pos%d' % i].astype(np.int64)).to(DEV)
            ni = torch.from_numpy(z['rcnn/neg%d' % i].astype(np.int64)).to(DEV)
            assert bool((assigned[pi] > 0).all()) and bool((assigned[ni] == 0).all())
            inds = torch.cat([pi, ni])
            assert inds.numel() == num
            is_pos = torch.cat([torch.ones_like(pi), torch.zeros_like(ni)]).bool()
            return inds, is_pos, torch.ones(num, dtype=torch.bool, device=DEV)

        def proposals_hook(own):
            out = []
            for i, (p, v) in enumerate(own):
                ref = torch.from_numpy(z['proposals%d' % i]).to(DEV)
                assert tuple(ref.shape) == tuple(p.shape), (ref.shape, p.shape)
                assert_proposals_reproduced(p[v].cpu().numpy(), z['proposals%d' % i], i)
                out.append((ref.contiguous(), torch.ones(ref.shape[0], dtype=torch.bool, device=DEV)))
            return out

        def choice_replay(a, size=None, replace=True, p=None):
            j = calls['gs']
            calls['gs'] += 1
            assert len(a) == int(z['gs/cand%d' % j][0]) and not replace
            draw = z['gs/draw%d' % j].astype(np.int64)
            assert tuple(np.atleast_1d(size)) == (len(draw),) and np.isin(draw, a).all()
            return draw
        monkeypatch.setattr(np.random, 'choice', choice_replay)

        n = 2
        boxes, labels = T.gt(T.MASK_SEED, n)
        losses = model(T.image(T.MASK_SEED, n).to(DEV), T.img_meta(n), return_loss=True,
                       gt_bboxes=[torch.from_numpy(b).to(DEV) for b in boxes],
                       gt_labels=[torch.from_numpy(l).to(DEV) for l in labels],
                       gt_masks=[torch.from_numpy(T.gt_masks(b)).to(DEV) for b in boxes],
                       samplers=dict(rpn=rpn_hook, rcnn=rcnn_hook, proposals=proposals_hook))
        assert calls['rpn'] == 2 and calls['rcnn'] == 2, calls
        keys = [k[len('loss/'):] for k in z.files if k.startswith('loss/') and not k.endswith('total')]
        assert set(keys) == set(losses.keys()), (sorted(keys), sorted(losses.keys()))
        bad = []
        for k in keys:
            v = losses[k]
            got = np.array([float(t.detach().sum()) for t in (v if isinstance(v, list) else [v])], np.float32)
            exp = z['loss/' + k]
            tol = (2e-4 if k == 'loss_mask' else 1e-4) * max(float(np.abs(exp).max()), 1e-3) + 1e-7
            if np.abs(got - exp).max() > tol:
                bad.append((k, got.tolist(), exp.tolist()))
        assert not bad, bad
        loss, _ = train.parse_losses(losses)
        assert abs(float(loss.detach()) - float(z['loss/total'][0])) < 1e-4 * float(z['loss/total'][0])
        train.backward_unit(loss)
        params = dict(model.named_parameters())
        bad = []
        for name, idx in T.GRADS_MASK:
            g = params[name].grad
            assert g is not None, name
            a, b = g[idx].cpu().numpy(), z['grad/' + name]
            rel = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
            l2 = float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))
            print('%s: max |diff| / max |g| = %.2e, relative L2 = %.2e' % (name, rel, l2))
            if name.startswith(('bbox_head.fc_cls', 'bbox_head.fc_reg')):
                if rel > 1e-5:
                    bad.append((name, rel))
            elif not grad_close(a, b, l2tol=1e-3):         # + relative L2 <= 1e-3 per tensor (VERDICT r3 #3c)
                bad.append((name, rel, l2))
        assert not bad, bad
    finally:
        BF.set_conv_math(prev)
        del model
        torch.cuda.empty_cache()


@pytest.mark.parametrize('math', ['bf16x6', 'bf16'])
def test_htc_x101_fullsize_iteration_vs_executed_reference(math):
    """The HTC row of BASELINE cfg[4] AT ITS OWN TRUNK AND SIZE: ``HybridTaskCascade.forward_train``
    (htc.py:197-308: three box stages with semantic fusion, interleaved mask branches with information flow, the
    semantic head) on ResNeXt-101-64x4d at 1 x 3x800x1344 with instance bitmaps and a semantic map, against the
    executed reference (tests/golden/make_golden_mask_htc_fullsize.py; RPN calibrated to non-saturating scores,
    every candidate taken: no draw).  ``bf16x6``: >= 97 % of the reference's proposals reproduced by the HIP RPN,
    then all 27 loss terms to 2e-4 and the box / mask / semantic head gradients; ``bf16`` (what BASELINE names for
    this config): the 3e-2 budget per term of the cascade test.  The launch census asserts the kernels the HTC
    bench row's time comes from ran (halo kernel; the LDS-resident grouped 3x3 kernel in bf16x6, the bf16-STORAGE
    kernels of the frozen trunk in bf16)."""
    from balancedgroupsoftmax_amd import functional as BF
    from balancedgroupsoftmax_amd import train
    from tests.golden import make_golden_mask_htc_fullsize as T
    z = np.load(T.OUT_HTC)
    assert int(z['saturated_scores0'][0]) == 0
    prev = BF.set_conv_math(math)
    prev_storage = BF.set_bf16_storage(math == 'bf16')       # bf16: the frozen trunk on bf16 tensors, as benched
    model = None
    try:
        tmp = tempfile.mkdtemp(prefix='bgs_htcfull_')
        model_cfg, train_cfg = T.htc_configs(tmp)
        model = bgs.build_detector(to_config_dict(model_cfg), train_cfg=to_config_dict(train_cfg),
                                   test_cfg=to_config_dict(G.TEST_CFG))
        with torch.no_grad():
            det_oracle.fill_detector(model.state_dict(), T.HTC_SEED)
        T.apply_rpn_scale(model.state_dict(), float(z['rpn_cls_scale'][0]))
        model.to(DEV)
        for nme, p_ in model.named_parameters():
            p_.requires_grad = nme.startswith(('bbox_head.', 'mask_head.', 'semantic_head.'))
        model.train()
        boxes, labels = T.gt(T.HTC_SEED, 1)
        BF.launch_census(reset=True)

        def proposals_hook(own):
            out = []
            for i, (p, v) in enumerate(own):
                ref = torch.from_numpy(z['proposals%d' % i]).to(DEV)
                frac = match_boxes(p[v].cpu().numpy(), z['proposals%d' % i], tol_px=0.05, tol_score=2e-4)
                print('image %d: %.4f of the reference proposals reproduced by the HIP RPN' % (i, frac))
                if math == 'bf16x6':
                    if frac < 1.0:
                        print('image %d: misses by cause: %s'
                              % (i, proposal_miss_taxonomy(p[v].cpu().numpy(), z['proposals%d' % i])))
                    assert frac >= 0.999, (frac, proposal_miss_taxonomy(p[v].cpu().numpy(), z['proposals%d' % i]))
                else:        # bf16 operands move boxes / scores by more than that tolerance: a widened match
                    wide = match_iou(p[v].cpu().numpy(), z['proposals%d' % i], BF16_PROP_IOU)
                    print('image %d (bf16): %.4f of the reference proposals have a HIP proposal with IoU >= %.1f'
                          % (i, wide, BF16_PROP_IOU))
                    assert wide >= BF16_MIN_FRAC, wide
                nn_ = min(ref.shape[0], p.shape[0])
                pad = torch.zeros_like(p)
                pad[:nn_] = ref[:nn_]
                ok = torch.zeros(p.shape[0], dtype=torch.bool, device=DEV)
                ok[:nn_] = True
                out.append((pad.contiguous(), ok))
            return out

        losses = model(T.image(T.HTC_SEED, 1).to(DEV), T.img_meta(1), return_loss=True,
                       gt_bboxes=[torch.from_numpy(b).to(DEV) for b in boxes],
                       gt_labels=[torch.from_numpy(l).to(DEV) for l in labels],
                       gt_masks=[torch.from_numpy(T.gt_masks(b)).to(DEV) for b in boxes],
                       gt_semantic_seg=torch.from_numpy(T.gt_semantic_seg(T.HTC_SEED)).to(DEV),
                       samplers=dict(proposals=proposals_hook))
        census = BF.launch_census()
        assert census['halo_bfx4'] + census['planes_3x3'] >= 5, census       # (3x3 / stride 1: halo or 8 x 8-pixel planes kernel)
        if math == 'bf16':       # 33 grouped convs + the 1x1 convs of the frozen trunk ran on bf16 tensors
            assert census['grouped_bf16s'] >= 30 and census['bf16s'] >= 70 and census['grouped_lds'] == 0, census
        else:
            assert census['grouped_lds'] >= 30 and census['bf16s'] == 0, census
        keys = [k[len('loss/'):] for k in z.files if k.startswith('loss/') and not k.endswith('total')]
        assert set(keys) == set(losses.keys()), (sorted(keys), sorted(losses.keys()))
        total_exp = float(z['loss/total'][0])
        bad, worst, worst_ratio = [], 0.0, 0.0
        for k in keys:
            v = losses[k]
            got = np.array([float(t.detach().sum()) for t in (v if isinstance(v, list) else [v])], np.float32)
            exp = z['loss/' + k]
            d = float(np.abs(got - exp).max())
            worst = max(worst, d)
            # bf16 mode: a PER-TERM budget (round 5; before: 3e-2 of the TOTAL for every term, under which a small
            # term could be several times off)
            tol = 2e-4 * max(float(np.abs(exp).max()), 1.0) if math == 'bf16x6' else \
                float((BF16_TERM_REL * np.abs(exp) + BF16_TERM_ABS).max())
            if math != 'bf16x6':
                ratio = float((np.abs(got - exp) / (BF16_TERM_REL * np.abs(exp) + BF16_TERM_ABS)).max())
                worst_ratio = max(worst_ratio, ratio)
                if ratio > 1.0:
                    bad.append((k, got.tolist(), exp.tolist()))
            elif d > tol:
                bad.append((k, got.tolist(), exp.tolist()))
        loss, _ = train.parse_losses(losses)
        print('%s HTC X101 @800x1344: total %.5f vs executed reference %.5f, worst term diff %.2e%s'
              % (math, float(loss.detach()), total_exp, worst,
                 '' if math == 'bf16x6' else ', worst |diff| / (%.0e |term| + %.0e) = %.3f'
                 % (BF16_TERM_REL, BF16_TERM_ABS, worst_ratio)))
        assert not bad, bad
        assert abs(float(loss.detach()) - total_exp) < (2e-4 if math == 'bf16x6' else 2e-2) * total_exp
        loss.backward()
        params = dict(model.named_parameters())
        gbad = []
        for name, idx in T.GRADS_HTC:
            g = params[name].grad
            assert g is not None, name
            a, b = g[idx].cpu().numpy(), z['grad/' + name]
            rel = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-20))
            l2 = float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-20))
            print('%s: max |diff| / max |g| = %.2e, rel-L2 = %.2e' % (name, rel, l2))
            if math == 'bf16x6':
                if rel > (1e-4 if 'fc_cls' in name or 'fc_reg' in name else 2e-3):
                    gbad.append((name, rel))
            elif ('fc_cls' in name or 'fc_reg' in name) and l2 > bf16_grad_bound(name):     # vs the fp32 golden
                gbad.append((name, l2))
        assert not gbad, gbad
    finally:
        BF.set_conv_math(prev)
        BF.set_bf16_storage(prev_storage)
        del model
        torch.cuda.empty_cache()


@pytest.mark.parametrize('math', ['bf16x6', 'bf16', 'bf16-fp32storage'])
def test_cascade_x101_fullsize_iteration_vs_executed_reference(math):
    _cascade_x101_vs_executed_reference(math, two_images=False)


@pytest.mark.parametrize('math', ['bf16x6', 'bf16'])
def test_cascade_x101_two_images_non_saturating_rpn_vs_executed_reference(math):
    """The X101 bench shape (2 x 3x800x1344) with an RPN whose objectness scores do not saturate
    (tests/golden/make_golden_cascade_x101_v2.py: ``rpn_cls`` calibrated to a top logit of 8 after the seeded
    fill, the factor stored in the golden and applied here the same way): in ``bf16x6`` the HIP RPN must reproduce
    >= 97 % of the executed reference's 480 proposals of EACH image (what the cfg[1] full-size golden demands)
    before the three RoI stages are compared on them; every loss term, the total and the head gradients as in the
    one-image test."""
    _cascade_x101_vs_executed_reference(math, two_images=True)


def _cascade_x101_vs_executed_reference(math, two_images):
    """BASELINE cfg[4] AT ITS OWN TRUNK AND SIZE: ``gs_cascade_rcnn_x101_64x4d`` (ResNeXt-101-64x4d,
    three GroupSoftmax stages, class-agnostic regression, stage weights 1 / 0.5 / 0.25) on
    1 x 3x800x1344 against the executed reference (tests/golden/make_golden_cascade_x101.py;
    resnext.py:12-91, cascade_rcnn.py:152-298).  ``bf16x6`` (fp32-faithful): every loss term to
    1e-4 (2e-4 for the later stages, whose RoIs are the previous stage's regressed boxes) and the
    three ``fc_cls`` gradients.  ``bf16`` (the mode BASELINE names for this config: operands rounded
    to bf16, fp32 accumulate — the reference's fp16 autocast contract): the documented budget of
    3e-2 of the total per term, 2e-2 on the total.  The launch census ASSERTS that the kernels the
    X101 bench numbers come from ran inside this iteration: the LDS-resident grouped 3x3 conv on
    the large maps and, in the bf16 mode, the bf16-STORAGE kernels of the frozen trunk
    (csrc/conv_bf16s.hip: bf16 activations in HBM; ``bf16-fp32storage`` = the same arithmetic on fp32
    tensors, the 8-wave 128x128 operand ring)."""
    from balancedgroupsoftmax_amd import functional as BF
    from balancedgroupsoftmax_amd import train
    if two_images:
        from tests.golden import make_golden_cascade_x101_v2 as T
        z = np.load(os.path.join(os.path.dirname(T.__file__), 'e2e_cascade_x101_2img_golden.npz'))
        assert int(z['saturated_scores0'][0]) == 0 and int(z['saturated_scores1'][0]) == 0
    else:
        from tests.golden import make_golden_cascade_x101 as T
        z = np.load(os.path.join(os.path.dirname(T.__file__), 'e2e_cascade_x101_fullsize_golden.npz'))
    min_frac = 0.97 if two_images else 0.5
    model = None
    storage = math == 'bf16'
    label, math = math, math.split('-')[0]
    prev = BF.set_conv_math(math)
    prev_storage = BF.set_bf16_storage(storage)
    try:
        tmp = tempfile.mkdtemp(prefix='bgs_x101_')
        model_cfg, train_cfg = T.configs(tmp)
        model = bgs.build_detector(to_config_dict(model_cfg), train_cfg=to_config_dict(train_cfg),
                                   test_cfg=to_config_dict(G.TEST_CFG))
        with torch.no_grad():
            det_oracle.fill_detector(model.state_dict(), T.SEED)
        if two_images:
            T.apply_rpn_scale(model.state_dict(), float(z['rpn_cls_scale'][0]))
        model.to(DEV)
        train.select_training_param(model, 2)          # the three box heads train (fc_cls, fc_reg, shared FCs)
        model.train()
        boxes, labels = T.gt()
        BF.launch_census(reset=True)

        def proposals_hook(own):
            # the reference's proposals (its saturated RPN scores tie at exactly 1.0; see the generator)
            out = []
            for i, (p, v) in enumerate(own):
                ref = torch.from_numpy(z['proposals%d' % i]).to(DEV)
                frac = match_boxes(p[v].cpu().numpy(), z['proposals%d' % i], tol_px=0.05, tol_score=2e-4)
                print('image %d: %.4f of the reference proposals reproduced by the HIP RPN (%d reference '
                      'scores saturated at 1.0)' % (i, frac, int(z['saturated_scores%d' % i][0])))
                if math == 'bf16x6':       # (bf16 operands move scores by more than the match tolerance)
                    assert frac >= min_frac, frac
                elif two_images:           # (the 1-image golden's RPN scores saturate: only the 2-image one is pinned)
                    wide = match_iou(p[v].cpu().numpy(), z['proposals%d' % i], BF16_PROP_IOU)
                    print('image %d (bf16): %.4f of the reference proposals have a HIP proposal with IoU >= %.1f'
                          % (i, wide, BF16_PROP_IOU))
                    assert wide >= BF16_MIN_FRAC, wide
                n = min(ref.shape[0], p.shape[0])
                pad = torch.zeros_like(p)
                pad[:n] = ref[:n]
                ok = torch.zeros(p.shape[0], dtype=torch.bool, device=DEV)
                ok[:n] = True
                out.append((pad.contiguous(), ok))
            return out

        losses = model(T.image().to(DEV), T.img_meta(), return_loss=True,
                       gt_bboxes=[torch.from_numpy(b).to(DEV) for b in boxes],
                       gt_labels=[torch.from_numpy(l).to(DEV) for l in labels],
                       samplers=dict(proposals=proposals_hook))
        census = BF.launch_census()
        assert census['halo_bfx4'] + census['planes_3x3'] >= 5, census       # (3x3 / stride 1: halo or 8 x 8-pixel planes kernel)
        if math == 'bf16' and storage:
            # the frozen trunk ran on bf16 tensors: 33 grouped convs, 66 + 4 1x1 convs, 4 FPN laterals
            assert census['grouped_bf16s'] == 33 and census['bf16s'] >= 74, census
            assert census['grouped_lds'] == 0 and census['bf16_ring8'] == 0, census
        else:
            assert census['grouped_lds'] >= 30, census   # 30 of the 33 grouped convs of an X101 forward
            assert census['bf16s'] == 0 and census['grouped_bf16s'] == 0, census
            if math == 'bf16':
                assert census['bf16_ring8'] >= 40, census
        total_exp = float(z['loss/total'][0])
        bad, worst, worst_ratio = [], 0.0, 0.0
        for k, v in losses.items():
            if 'loss' not in k:
                continue
            got = np.array([float(t.detach().sum()) for t in (v if isinstance(v, list) else [v])], np.float32)
            exp = z['loss/' + k]
            d = float(np.abs(got - exp).max())
            worst = max(worst, d)
            if math == 'bf16x6':
                tol = (1e-4 if (k.startswith('s0.') or 'rpn' in k) else 2e-4) * max(float(np.abs(exp).max()), 1.0)
                if d > tol:
                    bad.append((k, got.tolist(), exp.tolist()))
            else:           # bf16 mode: a PER-TERM budget (round 5; before: 3e-2 of the TOTAL for every term)
                ratio = float((np.abs(got - exp) / (BF16_TERM_REL * np.abs(exp) + BF16_TERM_ABS)).max())
                worst_ratio = max(worst_ratio, ratio)
                if ratio > 1.0:
                    bad.append((k, got.tolist(), exp.tolist()))
        loss, _ = train.parse_losses(losses)
        print('%s cascade X101 @800x1344: total %.5f vs executed reference %.5f, worst term diff %.2e%s'
              % (label, float(loss.detach()), total_exp, worst,
                 '' if math == 'bf16x6' else ', worst |diff| / (%.0e |term| + %.0e) = %.3f'
                 % (BF16_TERM_REL, BF16_TERM_ABS, worst_ratio)))
        assert not bad, bad
        assert abs(float(loss.detach()) - total_exp) < (2e-4 if math == 'bf16x6' else 2e-2) * total_exp
        loss.backward()
        params = dict(model.named_parameters())
        gbad = []
        for name, idx in T.GRADS:
            g = params[name].grad
            assert g is not None, name
            a, b = g[idx].cpu().numpy(), z['grad/' + name]
            rel = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-20))
            l2 = float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-20))
            print('%s: max |diff| / max |g| = %.2e, rel-L2 = %.2e' % (name, rel, l2))
            if math == 'bf16x6':
                if rel > (1e-4 if 'fc_cls' in name or 'fc_reg' in name else 2e-3):
                    gbad.append((name, rel))
            elif ('fc_cls' in name or 'fc_reg' in name) and l2 > bf16_grad_bound(name):     # vs the fp32 golden
                gbad.append((name, l2))
        assert not gbad, gbad
    finally:
        BF.set_conv_math(prev)
        BF.set_bf16_storage(prev_storage)
        del model
        torch.cuda.empty_cache()
