#!/usr/bin/env python
"""Generates ``tests/golden/e2e_cascade_x101_fullsize_golden.npz`` by EXECUTING THE REFERENCE'S
cfg[4] DETECTOR AT ITS OWN SIZE AND TRUNK on CPU: ``CascadeRCNN.forward_train``
(mmdet/models/detectors/cascade_rcnn.py:152-298) of
configs/bags/gs_cascade_rcnn_x101_64x4d_fpn_1x_lvis.py — ResNeXt-101-64x4d
(mmdet/models/backbones/resnext.py:12-91), FPN, RPN, three GroupSoftmax box stages with
``reg_class_agnostic=True`` and stage loss weights 1 / 0.5 / 0.25 — on 1 x 3x800x1344 with 20 GT.

The small cascade golden (make_golden_train.py) uses an R50 trunk at 192x256; at THIS size the HIP
path runs the kernels the X101 numbers of the bench line come from (``conv_igemm_bf16_ring8_kernel``
at M >= 2048, ``grouped_conv3x3_lds_kernel`` on 200x336 maps), which no executed-reference
comparison reached before.  Samplers take every candidate on both sides as in
make_golden_fullsize.py (RPN ``num=600000``; ``rpn_proposal.max_num=480`` + 20 GT <= 512;
``others_sample_ratio=1e6``; a random draw raises).  Stored: all 3 x 6 stage loss terms + the RPN
terms, the total, and gradients of the three stages' ``fc_cls`` (the parameters cfg[4] trains:
selectp=3) and of stage 3's ``fc_reg`` / ``shared_fcs``.

    python tests/golden/make_golden_cascade_x101.py     # authoring container only (~2-4 min)
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

OUT = os.path.join(HERE, 'e2e_cascade_x101_fullsize_golden.npz')
SEED = 991
H, W, IMGS, NGT = 800, 1344, 1, 20
GRADS = [
    ('bbox_head.0.fc_cls.weight', (slice(None, None, 8), slice(None, None, 16))),
    ('bbox_head.1.fc_cls.weight', (slice(None, None, 8), slice(None, None, 16))),
    ('bbox_head.2.fc_cls.weight', (slice(None, None, 8), slice(None, None, 16))),
    ('bbox_head.0.fc_cls.bias', (slice(None),)),
    ('bbox_head.2.fc_cls.bias', (slice(None),)),
    ('bbox_head.2.fc_reg.weight', (slice(None), slice(None, None, 16))),
    ('bbox_head.2.shared_fcs.1.weight', (slice(None, None, 16), slice(None, None, 16))),
]


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


def configs(table_dir):
    from bench import detector_cfg
    model, train_cfg = detector_cfg(table_dir, cascade=True)      # X101-64x4d trunk, 3 stages
    for h in model['bbox_head']:
        h['gs_config']['others_sample_ratio'] = 1e6
    train_cfg['rpn']['sampler']['num'] = 600000
    train_cfg['rpn_proposal'].update(nms_post=480, max_num=480)
    return model, train_cfg


def main():
    from balancedgroupsoftmax_amd.config import to_config_dict
    from oracle import det_oracle
    from tests.golden import make_golden_e2e as E
    from tests.golden import make_golden_train as T
    T._bind_reference_ops()
    from mmdet.models import build_detector
    tmp = tempfile.mkdtemp(prefix='bgs_x101_')
    model_cfg, train_cfg = configs(tmp)
    model = build_detector(to_config_dict(model_cfg), train_cfg=to_config_dict(train_cfg),
                           test_cfg=to_config_dict(E.TEST_CFG))
    assert type(model.backbone).__name__ == 'ResNeXt' and model.backbone.depth == 101
    with torch.no_grad():
        det_oracle.fill_detector(model.state_dict(), SEED)
    model.train()
    # cfg[4] trains the three fc_cls (selectp=3, tools/train.py:49-91); the extra heads' weights get
    # gradients too so that a wider arm can be compared; the trunk is left out (CPU time)
    for n, p in model.named_parameters():
        p.requires_grad = n.startswith('bbox_head.')
    boxes, labels = gt()
    # The RPN of a randomly initialised X101 saturates: many objectness scores round to exactly 1.0
    # in fp32 after the sigmoid (rpn_head.py:66), so the top-k / NMS order among them is decided by
    # torch's tie handling, which no other implementation reproduces.  The proposals are therefore
    # recorded and the GPU test runs the three RoI stages on THESE boxes (its own RPN is compared
    # with them as a set, reported but not asserted tightly); RPN losses, trunk and FPN are compared
    # directly.
    rec = {}
    get_bboxes = model.rpn_head.get_bboxes

    def get_bboxes_rec(*a, **k):
        props = get_bboxes(*a, **k)
        for i, p in enumerate(props):
            rec['proposals%d' % i] = p.detach().numpy().astype(np.float32)
            rec['saturated_scores%d' % i] = np.array([int((p[:, 4] >= 1.0).sum())], np.int32)
        return props
    model.rpn_head.get_bboxes = get_bboxes_rec
    losses = model.forward_train(image(), img_meta(), [torch.from_numpy(b) for b in boxes],
                                 [torch.from_numpy(l) for l in labels])
    out = dict(rec)
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
        if 'loss/' in k or 'saturated' in k:
            print(k, out[k])
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT))


if __name__ == '__main__':
    main()
