#!/usr/bin/env python
"""Generates ``tests/golden/e2e_inference_golden.npz`` by EXECUTING THE REFERENCE DETECTORS end to
end on CPU (test-time path, deterministic):

* ``frcnn``: configs/bags/gs_faster_rcnn_r50_fpn_1x_lvis_with0_bg8.py (the BASELINE config) —
  ``GroupSoftmax`` detector = ResNet-50 + FPN + RPNHead + SingleRoIExtractor + GSBBoxHeadWith0:
  ``extract_feat`` -> ``simple_test_rpn`` -> ``simple_test_bboxes`` (two_stage.py:267-289,
  test_mixins.py:8-12,39-67);
* ``htc``: configs/bags/gs_htc_x101_64x4d_fpn_20e_16gpu_lvis.py with the trunk cut to
  ResNeXt-50-64x4d — ``HybridTaskCascade.simple_test`` (htc.py:313-432): three box stages with
  semantic fusion, stage-averaged logits, ensemble masks with mask information flow.

The reference's compiled ops are bound to THE REFERENCE'S OWN SOURCES built for the host
(oracle/build_ref.py): ``nms_cpu.cpp`` and the ``ROIAlignForward`` kernel of
``roi_align_kernel.cu``.  Weights: ``oracle.det_oracle.fill_detector(seed)`` on the reference's
state_dict (this package's modules take the same state_dict).  Inputs: seeded normal image.

    python tests/golden/make_golden_e2e.py          # authoring container only
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import build_ref, det_oracle, ref_import  # noqa: E402

OUT = os.path.join(HERE, 'e2e_inference_golden.npz')
H, W = 192, 256
IMG_SEED, FRCNN_SEED, HTC_SEED, MASK_SEED = 901, 902, 903, 904
TEST_CFG = dict(rpn=dict(nms_across_levels=False, nms_pre=1000, nms_post=1000, max_num=300,
                         nms_thr=0.7, min_bbox_size=0),
                rcnn=dict(score_thr=0.0, nms=dict(type='nms', iou_thr=0.5), max_per_img=50,
                          mask_thr_binary=0.5),
                keep_all_stages=False)


def image():
    rs = np.random.RandomState(IMG_SEED)
    return rs.standard_normal((1, 3, H, W)).astype(np.float32)


def img_meta():
    return [dict(img_shape=(H, W - 3, 3), pad_shape=(H, W, 3), ori_shape=(H, W - 3, 3),
                 scale_factor=1.0, flip=False)]


def configs(table_dir, which):
    """The model dicts of bench.detector_cfg (same keys / values as the reference config files;
    the three table files are synthetic)."""
    from bench import detector_cfg
    model, _ = detector_cfg(table_dir, htc=(which == 'htc'), mask=(which == 'mask'))
    if which == 'htc':
        model['backbone'] = dict(model['backbone'], depth=50)
    return model


def _bind_reference_ops():
    ref_import.install_stubs()
    sys.modules['mmdet.ops.nms.nms_cpu'] = build_ref.load_nms_cpu()
    import mmdet.ops  # noqa: F401  (mmdet.ops.nms is rebound to the function by its __init__)
    sys.modules['mmdet.ops.nms.nms_wrapper'].nms_cpu = sys.modules['mmdet.ops.nms.nms_cpu']
    ra = sys.modules['mmdet.ops.roi_align.roi_align']

    def roi_align(features, rois, out_size, spatial_scale, sample_num=0):
        out = build_ref.roi_align_reference(features.detach().numpy(), rois.detach().numpy(),
                                            spatial_scale, out_size[0], sample_num)
        return torch.from_numpy(out)
    ra.roi_align = roi_align


def main():
    from balancedgroupsoftmax_amd.config import to_config_dict
    _bind_reference_ops()
    from mmdet.core import bbox2roi
    from mmdet.models import build_detector
    out = {}
    img = torch.from_numpy(image())
    meta = img_meta()
    tcfg = to_config_dict(TEST_CFG)
    tmp = tempfile.mkdtemp(prefix='bgs_e2e_')

    # ---------------------------------------------------------------- Faster R-CNN R50 + BAGS
    model = build_detector(to_config_dict(configs(tmp, 'frcnn')), train_cfg=None, test_cfg=tcfg)
    with torch.no_grad():
        det_oracle.fill_detector(model.state_dict(), FRCNN_SEED)
        model.eval()
        x = model.extract_feat(img)
        props = model.simple_test_rpn(x, meta, tcfg.rpn)
        rois = bbox2roi(props)
        roi_feats = model.bbox_roi_extractor(x[:4], rois)
        cls_score, bbox_pred = model.bbox_head(roi_feats)
        db, dl, scores = model.simple_test_bboxes(x, meta, props, tcfg.rcnn, rescale=False)
    for i, f in enumerate(x):
        out['frcnn/p%d' % i] = f[:, ::16].contiguous().numpy()
    out['frcnn/proposals'] = props[0].numpy()
    out['frcnn/roi_feats'] = roi_feats[::10, ::16].contiguous().numpy()
    out['frcnn/cls_score'] = cls_score[::8].contiguous().numpy()
    out['frcnn/bbox_pred'] = bbox_pred[::4, ::41].contiguous().numpy()
    out['frcnn/scores'] = scores[::4, ::7].contiguous().numpy()
    out['frcnn/det_bboxes'] = db.numpy()
    out['frcnn/det_labels'] = dl.numpy()
    print('frcnn: proposals', tuple(props[0].shape), 'dets', tuple(db.shape),
          'score range', float(db[:, 4].min()), float(db[:, 4].max()))

    # ---------------------------------------------------------------- Mask R-CNN R50 + BAGS (cfg[3])
    model = build_detector(to_config_dict(configs(tmp, 'mask')), train_cfg=None, test_cfg=tcfg)
    with torch.no_grad():
        det_oracle.fill_detector(model.state_dict(), MASK_SEED)
        model.eval()
        x = model.extract_feat(img)
        props = model.simple_test_rpn(x, meta, tcfg.rpn)
        db, dl, _ = model.simple_test_bboxes(x, meta, props, tcfg.rcnn, rescale=False)
        # test_mixins.py:153-180 (simple_test_mask) up to the per-detection class channel
        mask_rois = bbox2roi([db[:, :4]])
        mask_feats = model.mask_roi_extractor(x[:len(model.mask_roi_extractor.featmap_strides)],
                                              mask_rois)
        mask_pred = model.mask_head(mask_feats)
        probs = mask_pred[torch.arange(db.size(0)), dl + 1].sigmoid()
    out['mask/proposals'] = props[0].numpy()
    out['mask/det_bboxes'] = db.numpy()
    out['mask/det_labels'] = dl.numpy()
    out['mask/mask_probs'] = probs.numpy()
    print('mask: dets', tuple(db.shape), 'mask', tuple(probs.shape))

    # ---------------------------------------------------------------- HTC X50-64x4d + BAGS
    model = build_detector(to_config_dict(configs(tmp, 'htc')), train_cfg=None, test_cfg=tcfg)
    with torch.no_grad():
        det_oracle.fill_detector(model.state_dict(), HTC_SEED)
        model.eval()
        x = model.extract_feat(img)
        props = model.simple_test_rpn(x, meta, tcfg.rpn)
        _, semantic_feat = model.semantic_head(x)
        # htc.py:313-376 (keep_all_stages=False), calling the reference's own methods
        rois = bbox2roi(props)
        ms_scores = []
        for i in range(model.num_stages):
            cls_score, bbox_pred = model._bbox_forward_test(i, x, rois, semantic_feat=semantic_feat)
            ms_scores.append(cls_score)
            if i < model.num_stages - 1:
                rois = model.bbox_head[i].regress_by_class(rois, cls_score.argmax(dim=1), bbox_pred,
                                                           meta[0])
        cls_avg = sum(ms_scores) / float(len(ms_scores))
        db, dl = model.bbox_head[-1].get_det_bboxes(rois, cls_avg, bbox_pred, meta[0]['img_shape'],
                                                    1.0, rescale=False, cfg=tcfg.rcnn)
        # htc.py:379-405: ensemble masks of the final detections
        mask_rois = bbox2roi([db[:, :4]])
        ext = model.mask_roi_extractor[-1]
        mask_feats = ext(x[:len(ext.featmap_strides)], mask_rois)
        mask_feats = mask_feats + model.semantic_roi_extractor([semantic_feat], mask_rois)
        last, probs = None, []
        idx = torch.arange(db.size(0))
        for i in range(model.num_stages):
            mask_pred, last = model.mask_head[i](mask_feats, last)
            probs.append(mask_pred[idx, dl + 1].sigmoid())
        merged = sum(probs) / float(len(probs))
    out['htc/semantic_feat'] = semantic_feat[:, ::16].contiguous().numpy()
    out['htc/proposals'] = props[0].numpy()
    for i, s in enumerate(ms_scores):
        out['htc/cls_score%d' % i] = s[::16].contiguous().numpy()
    out['htc/final_rois'] = rois.numpy()
    out['htc/det_bboxes'] = db.numpy()
    out['htc/det_labels'] = dl.numpy()
    out['htc/mask_probs'] = merged.numpy()
    out['htc/mask_probs_stage0'] = probs[0][::5].contiguous().numpy()
    print('htc: proposals', tuple(props[0].shape), 'dets', tuple(db.shape), 'mask', tuple(merged.shape))
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT))


if __name__ == '__main__':
    main()
