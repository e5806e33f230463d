#!/usr/bin/env python
"""Generates ``tests/golden/e2e_cascade_x101_2img_golden.npz`` by EXECUTING THE REFERENCE'S cfg[4] DETECTOR
(``CascadeRCNN.forward_train``, mmdet/models/detectors/cascade_rcnn.py:152-298, of
configs/bags/gs_cascade_rcnn_x101_64x4d_fpn_1x_lvis.py: ResNeXt-101-64x4d, FPN, RPN, three GroupSoftmax box
stages) on CPU at the shape the X101 bench rows time — 2 x 3x800x1344, 20 GT per image — with an RPN whose
objectness scores DO NOT SATURATE.

make_golden_cascade_x101.py (one image) had to substitute the reference's proposals without a tight check: a
seeded X101 drives thousands of RPN logits past 17, their sigmoids round to exactly 1.0 in fp32 (rpn_head.py:66)
and the order among them is torch's tie handling.  Here the classification branch of the RPN head is CALIBRATED
after the seeded fill: one forward of the trunk + RPN on the golden's own images, then ``rpn_cls.weight`` and
``rpn_cls.bias`` are multiplied by ``s = 8 / max |logit|`` (stored in the file, applied by the test the same way):
the largest objectness logit is 8 (sigmoid 0.99966), scores are distinct, and the HIP RPN has to reproduce the
reference's 480 proposals per image on its own (>= 97 %, as the cfg[1] full-size golden demands) before the RoI
stages are compared on them.

Samplers take every candidate on both sides as in make_golden_fullsize.py (a random draw raises).  Stored: the
scale, the proposals, all 3 x 6 stage loss terms + the RPN terms, the total, and gradients of the three stages'
``fc_cls`` (cfg[4] trains them: selectp=3) and of stage 3's ``fc_reg`` / ``shared_fcs``.

    python tests/golden/make_golden_cascade_x101_v2.py     # authoring container only (~10 min on 8 cores)
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

OUT = os.path.join(HERE, 'e2e_cascade_x101_2img_golden.npz')
SEED = 1733
H, W, IMGS, NGT = 800, 1344, 2, 20
TOP_LOGIT = 8.0
from tests.golden.make_golden_cascade_x101 import GRADS, configs  # noqa: E402,F401  (same parameters, same cfg)


def image():
    g = torch.Generator().manual_seed(SEED)
    return torch.randn(IMGS, 3, H, W, generator=g)


def img_meta():
    return [dict(img_shape=(800, 1333, 3), pad_shape=(H, W, 3), ori_shape=(800, 1333, 3),
                 scale_factor=1.0, flip=False) for _ in range(IMGS)]


def gt():
    rs = np.random.RandomState(SEED)
    boxes, labels = [], []
    for _ in range(IMGS):
        wh = np.exp(rs.uniform(np.log(16), np.log(400), size=(NGT, 2)))
        xy = rs.uniform(0, 1, size=(NGT, 2)) * np.maximum(np.array([1333., 800.]) - wh - 1, 1)
        boxes.append(np.concatenate([xy, xy + wh], 1).astype(np.float32))
        labels.append(rs.randint(1, 1231, size=NGT).astype(np.int64))
    return boxes, labels


def apply_rpn_scale(state_dict, scale):
    """The calibration of this golden, on either implementation's state dict."""
    with torch.no_grad():
        state_dict['rpn_head.rpn_cls.weight'].mul_(float(scale))
        state_dict['rpn_head.rpn_cls.bias'].mul_(float(scale))


def main():
    from balancedgroupsoftmax_amd.config import to_config_dict
    from oracle import det_oracle
    from tests.golden import make_golden_e2e as E
    from tests.golden import make_golden_train as T
    T._bind_reference_ops()
    from mmdet.models import build_detector
    tmp = tempfile.mkdtemp(prefix='bgs_x101b_')
    model_cfg, train_cfg = configs(tmp)
    model = build_detector(to_config_dict(model_cfg), train_cfg=to_config_dict(train_cfg),
                           test_cfg=to_config_dict(E.TEST_CFG))
    assert type(model.backbone).__name__ == 'ResNeXt' and model.backbone.depth == 101
    with torch.no_grad():
        det_oracle.fill_detector(model.state_dict(), SEED)
    model.train()
    for n, p in model.named_parameters():
        p.requires_grad = n.startswith('bbox_head.')
    img = image()
    # ---- calibration pass: the trunk's features are kept and handed to forward_train below (the trunk is frozen
    #      and in eval-mode BN: the same tensors either way)
    with torch.no_grad():
        feats = model.extract_feat(img)
        cls_scores, _ = model.rpn_head(feats)
        top = max(float(c.abs().max()) for c in cls_scores)
    scale = TOP_LOGIT / top
    print('largest |objectness logit| of the seeded RPN: %.3f -> rpn_cls scaled by %.6g' % (top, scale))
    apply_rpn_scale(model.state_dict(), scale)
    model.extract_feat = lambda _img: feats
    boxes, labels = gt()
    rec = {}
    get_bboxes = model.rpn_head.get_bboxes

    def get_bboxes_rec(*a, **k):
        props = get_bboxes(*a, **k)
        for i, p in enumerate(props):
            rec['proposals%d' % i] = p.detach().numpy().astype(np.float32)
            s = np.sort(p[:, 4].detach().numpy())
            rec['saturated_scores%d' % i] = np.array([int((s >= 1.0).sum())], np.int32)
            rec['tied_scores%d' % i] = np.array([int((np.diff(s) == 0).sum())], np.int32)
        return props
    model.rpn_head.get_bboxes = get_bboxes_rec
    losses = model.forward_train(img, img_meta(), [torch.from_numpy(b) for b in boxes],
                                 [torch.from_numpy(l) for l in labels])
    out = dict(rec)
    out['rpn_cls_scale'] = np.array([scale], np.float64)
    total = 0
    for k, v in losses.items():
        vals = v if isinstance(v, list) else [v]
        out['loss/' + k] = np.array([float(t.detach().sum()) for t in vals], np.float32)
        if 'loss' in k:
            total = total + sum(t.sum() for t in vals)
    total.backward()
    out['loss/total'] = np.array([float(total.detach())], np.float32)
    params = dict(model.named_parameters())
    for name, idx in GRADS:
        out['grad/' + name] = params[name].grad[idx].contiguous().numpy()
    for k in sorted(out):
        if 'loss/' in k or 'saturated' in k or 'tied' in k or 'scale' in k:
            print(k, out[k])
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT))


if __name__ == '__main__':
    main()
