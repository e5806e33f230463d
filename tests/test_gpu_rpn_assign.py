"""``bgs_iou_assign`` / ``bgs_rpn_loss`` against the EXECUTED reference at the baseline size, directly:
tests/golden/rpn_assign_fullsize_golden.npz holds what the reference's ``MaxIoUAssigner.assign``
(max_iou_assigner.py:52-180), its numpy ``RandomSampler``, ``anchor_target`` (anchor_target.py:7-174) and
``RPNHead.loss`` (anchor_head.py:130-207, rpn_head.py:37-53) produced on 268,569 anchors x 2 images x 20 GT; the inputs
are regenerated from the generator's seeds.  No builder-authored tensor form in between: the assignment must equal the
reference's bit for bit, the losses to 1e-5 relative, the gradient by probes and checksums."""
import os

import numpy as np
import pytest
import torch

import balancedgroupsoftmax_amd as bgs
from balancedgroupsoftmax_amd import functional as BF
from balancedgroupsoftmax_amd.config import to_config_dict
from tests.golden import make_golden_rpn_assign as G

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
GOLD = os.path.join(os.path.dirname(G.__file__), 'rpn_assign_fullsize_golden.npz')


def _head_and_geometry():
    head = bgs.build_head(to_config_dict(dict(type='RPNHead', **G.RPN_HEAD))).to(DEV)
    sizes = G.featmap_sizes()
    dev = torch.device(DEV)
    anchors = head._all_anchors(sizes, dev)
    inside = head._inside_flags(sizes, G.img_meta(), G.RPN_TRAIN['allowed_border'], dev)
    return head, anchors, inside


def _gts(boxes):
    gt_cat = torch.from_numpy(np.concatenate(boxes)).to(DEV)
    offs = [0]
    for b in boxes:
        offs.append(offs[-1] + len(b))
    return gt_cat, offs


def test_iou_assign_kernel_equals_the_executed_reference_assigner_at_268569_anchors():
    z = np.load(GOLD)
    boxes, _, _ = G.inputs()
    head, anchors, inside = _head_and_geometry()
    assert anchors.shape[0] == 268569
    gt_cat, offs = _gts(boxes)
    ac = G.RPN_TRAIN['assigner']
    assigned = BF.iou_assign(anchors, gt_cat, offs, ac['pos_iou_thr'], ac['neg_iou_thr'], ac['min_pos_iou'],
                             valid=inside, shared_boxes=True).cpu().numpy()
    for i in range(G.IMGS):
        ref = z['assigned%d' % i].astype(np.int32)
        ins = inside[i].cpu().numpy().astype(bool)
        # outside anchors: the reference never hands them to the assigner (anchor_target.py:100-107); the kernel marks
        # them -1 — and the fixture stores -1 there, so the whole row compares
        assert (assigned[i][~ins] == -1).all()
        bad = np.nonzero(assigned[i] != ref)[0]
        assert bad.size == 0, ('image %d: %d of %d anchors differ, first %s: kernel %s reference %s'
                               % (i, bad.size, ref.size, bad[:5], assigned[i][bad[:5]], ref[bad[:5]]))
        assert int((ref > 0).sum()) >= 90          # the case is not degenerate: positives, negatives and ignored exist
        assert int(((ref == -1) & ins).sum()) > 1000


def test_rpn_loss_kernel_equals_the_executed_reference_loss_on_the_recorded_samples():
    z = np.load(GOLD)
    boxes, cls, reg = G.inputs()
    head, anchors, inside = _head_and_geometry()
    gt_cat, offs = _gts(boxes)
    A = anchors.shape[0]
    assigned = torch.from_numpy(np.stack([z['assigned%d' % i].astype(np.int32) for i in range(G.IMGS)])).to(DEV)
    pos = torch.zeros(G.IMGS, A, dtype=torch.uint8, device=DEV)
    neg = torch.zeros(G.IMGS, A, dtype=torch.uint8, device=DEV)
    for i in range(G.IMGS):
        pos[i, torch.from_numpy(z['pos%d' % i].astype(np.int64)).to(DEV)] = 1
        neg[i, torch.from_numpy(z['neg%d' % i].astype(np.int64)).to(DEV)] = 1
    # the kernels read the head's own buffers: [N, H, W, A + 4A] per level (cls channels first)
    outs = [torch.from_numpy(np.concatenate([c, r], 1)).permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
            for c, r in zip(cls, reg)]
    lc, lb, _ = BF.rpn_loss(outs, head.num_anchors, anchors, assigned, pos, neg, gt_cat, offs, head.target_means,
                            head.target_stds, head.loss_bbox.beta, G.RPN_TRAIN['pos_weight'], 1.0, 1.0)
    (lc.sum() + lb.sum()).backward()
    np.testing.assert_allclose(lc.detach().cpu().numpy(), z['loss_rpn_cls'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(lb.detach().cpu().numpy(), z['loss_rpn_bbox'], rtol=1e-5, atol=1e-7)
    for l, o in enumerate(outs):
        g = o.grad.reshape(-1).cpu().numpy()
        np.testing.assert_allclose(g[::G.PROBE], z['grad_probe%d' % l], rtol=1e-5, atol=1e-9)
        np.testing.assert_allclose(g.astype(np.float64).sum(), z['grad_sum%d' % l][0], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(np.abs(g.astype(np.float64)).sum(), z['grad_abs%d' % l][0], rtol=1e-5, atol=1e-9)


def test_rcnn_assign_and_targets_kernels_equal_the_executed_reference_roi_stage():
    """RoI stage (a8 / a14), directly: ``bgs_iou_assign`` on the 2000 proposals == the executed ``MaxIoUAssigner``
    (0.5 / 0.5 / 0.5) bit for bit, and ``bgs_rcnn_targets`` on the 512 RoIs per image the reference's numpy
    ``RandomSampler`` drew (recorded; ``add_gt_as_proposals`` numbering) == the executed ``bbox_target``
    (mmdet/core/bbox/bbox_target.py:7-61): RoIs, labels and both weight tensors exactly, deltas to 1e-5."""
    z = np.load(GOLD)
    boxes, labels, props = G.rcnn_inputs()
    gt_cat, offs = _gts(boxes)
    ac = G.RCNN_TRAIN['assigner']
    pr = torch.from_numpy(np.stack([p[:, :4] for p in props])).to(DEV).contiguous()          # [N, 2000, 4]
    assigned = BF.iou_assign(pr, gt_cat, offs, ac['pos_iou_thr'], ac['neg_iou_thr'], ac['min_pos_iou'],
                             shared_boxes=False).cpu().numpy()
    for i in range(G.IMGS):
        ref = z['rcnn_assigned%d' % i].astype(np.int32)
        bad = np.nonzero(assigned[i] != ref)[0]
        assert bad.size == 0, (i, bad[:5], assigned[i][bad[:5]], ref[bad[:5]])
        assert 50 < int((ref > 0).sum()) < 1500
    # candidates = [gt boxes; proposals] (add_gt_as_proposals: a GT row is assigned to itself, assign_result.add_gt_)
    cand, asg, inds, gtl = [], [], [], []
    for i in range(G.IMGS):
        cand.append(torch.from_numpy(np.concatenate([boxes[i], props[i][:, :4]])).to(DEV).contiguous())
        full = np.concatenate([np.arange(1, G.NGT + 1, dtype=np.int32), z['rcnn_assigned%d' % i].astype(np.int32)])
        asg.append(torch.from_numpy(full).to(DEV))
        sel = np.concatenate([z['rcnn_pos%d' % i], z['rcnn_neg%d' % i]]).astype(np.int64)
        assert sel.size == 512
        inds.append(torch.from_numpy(sel).to(DEV))
        gtl.append(torch.from_numpy(labels[i]).to(DEV))
    rois, lab, lw, bt, bw = BF.rcnn_targets(cand, asg, inds, [None] * G.IMGS, gtl, gt_cat, offs, 512, G.RCNN_MEANS,
                                            G.RCNN_STDS, G.RCNN_TRAIN['pos_weight'])
    np.testing.assert_array_equal(rois[:, 1:].cpu().numpy(), z['rcnn_rois'])
    np.testing.assert_array_equal(rois[:, 0].cpu().numpy(), np.repeat(np.arange(G.IMGS, dtype=np.float32), 512))
    np.testing.assert_array_equal(lab.cpu().numpy(), z['rcnn_labels'].astype(np.int64))
    np.testing.assert_array_equal(lw.cpu().numpy(), z['rcnn_label_weights'])
    np.testing.assert_array_equal(bw.cpu().numpy(), z['rcnn_bbox_weights'])
    np.testing.assert_allclose(bt.cpu().numpy(), z['rcnn_bbox_targets'], rtol=1e-5, atol=1e-6)
    assert int((lab > 0).sum()) == 256
