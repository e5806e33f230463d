#!/usr/bin/env python
"""Generates ``tests/golden/e2e_train_fullsize_golden.npz`` by EXECUTING THE REFERENCE DETECTOR'S
TRAINING ITERATION on CPU AT THE BASELINE SIZE: cfg[1] = ``GroupSoftmax``
(``TwoStageDetector.forward_train``, mmdet/models/detectors/two_stage.py:134-265) of
configs/bags/gs_faster_rcnn_r50_fpn_1x_lvis_with0_bg8.py on 2 x 3x800x1344 with 20 GT per image.

At this size the HIP path dispatches kernels / tile instantiations the 192x256 goldens never
reach (128x128 / 128x64 tiles, the halo kernel, XCD-banded grids of > 2000 workgroups, 268,569
anchors per image).  As in make_golden_train.py the three sampling steps are configured to take
EVERY candidate (the reference draws with numpy on the host, which nothing can reproduce draw for
draw): RPN sampler ``num=600000`` (> 268,569 anchors), ``rpn_proposal.max_num=480`` (+20 GT <=
``num=512`` RoIs/img, fewer than 128 positives), ``others_sample_ratio=1e6``;
``RandomSampler.random_choice`` / ``np.random.choice`` raise.  Everything else is the shipped
config.  Stored: the 8 loss terms (RPN terms per level), the total, and gradient slices from
``fc_cls`` down to ResNet layer2.

    python tests/golden/make_golden_fullsize.py          # authoring container only (~1-2 min)
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

OUT = os.path.join(HERE, 'e2e_train_fullsize_golden.npz')
SEED = 977
H, W, IMGS, NGT = 800, 1344, 2, 20
GRADS = [
    ('bbox_head.fc_cls.weight', (slice(None, None, 8), slice(None, None, 16))),
    ('bbox_head.fc_cls.bias', (slice(None),)),
    ('bbox_head.fc_reg.weight', (slice(None, None, 64), slice(None, None, 16))),
    ('bbox_head.shared_fcs.0.weight', (slice(None, None, 16), slice(None, None, 256))),
    ('rpn_head.rpn_conv.weight', (slice(None, None, 16), slice(None, None, 16))),
    ('rpn_head.rpn_cls.weight', (slice(None),)),
    ('neck.lateral_convs.0.conv.weight', (slice(None, None, 8), slice(None, None, 8))),
    ('neck.fpn_convs.0.conv.weight', (slice(None, None, 16), slice(None, None, 16))),
    ('backbone.layer2.0.conv1.weight', (slice(None, None, 4), slice(None, None, 8))),
    ('backbone.layer2.3.conv2.weight', (slice(None, None, 8), slice(None, None, 8))),
    ('backbone.layer3.5.conv2.weight', (slice(None, None, 16), slice(None, None, 16))),
    ('backbone.layer4.2.bn3.bias', (slice(None, None, 8),)),
]


def image():
    g = torch.Generator().manual_seed(SEED)
    return torch.randn(IMGS, 3, H, W, generator=g)


def img_meta():
    return [dict(img_shape=(800, 1333, 3), pad_shape=(H, W, 3), ori_shape=(800, 1333, 3),
                 scale_factor=1.0, flip=False) for _ in range(IMGS)]


def gt():
    """SURVEY.md §8(d) cfg 2: 20 boxes per image, sides exp(U(log 16, log 400)), labels 1..1230."""
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
    model, train_cfg = detector_cfg(table_dir)
    model['bbox_head']['gs_config']['others_sample_ratio'] = 1e6
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
    tmp = tempfile.mkdtemp(prefix='bgs_full_')
    model_cfg, train_cfg = configs(tmp)
    model = build_detector(to_config_dict(model_cfg), train_cfg=to_config_dict(train_cfg),
                           test_cfg=to_config_dict(E.TEST_CFG))
    with torch.no_grad():
        det_oracle.fill_detector(model.state_dict(), SEED)
    model.train()
    boxes, labels = gt()
    losses = model.forward_train(image(), img_meta(), [torch.from_numpy(b) for b in boxes],
                                 [torch.from_numpy(l) for l in labels])
    out = {}
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
        if 'loss/' in k:
            print(k, out[k])
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT))


if __name__ == '__main__':
    main()
