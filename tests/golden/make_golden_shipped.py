#!/usr/bin/env python
"""Generates ``tests/golden/e2e_train_shipped_samplers_golden.npz``: the reference detector's cfg[1]
training iteration (``TwoStageDetector.forward_train`` + ``backward``,
mmdet/models/detectors/two_stage.py:134-265) EXECUTED on CPU at the BASELINE size (2 x 3x800x1344,
20 GT / image) WITH THE SHIPPED SAMPLER SIZES — the configuration ``bench.py`` times:

* RPN ``RandomSampler(num=256, pos_fraction=0.5)`` over the 268,569 anchors of an image
  (mmdet/core/bbox/samplers/random_sampler.py:19-53, base_sampler.py:31-78),
* RCNN ``RandomSampler(num=512, pos_fraction=0.25, add_gt_as_proposals=True)`` over <= 2000
  proposals + 20 GT boxes,
* ``others_sample_ratio=8`` in the GroupSoftmax head (gs_bbox_head_with0.py:63-89).

The reference draws with numpy on the host; no other implementation can reproduce those draws, so
they are RECORDED here (nothing is replaced: the samplers run as shipped under ``np.random.seed``)
and the GPU test injects them through the package's sampler hooks (``rpn.py`` ``samplers['rpn']``,
``detectors.py`` ``samplers['rcnn']`` / ``samplers['proposals']``, ``gs_config.sampler='numpy'``):

* ``rpn/pos{i}``, ``rpn/neg{i}``: sampled anchor indices of image i (full anchor numbering),
* ``proposals{i}``: the reference's RPN proposals ``[<=2000, 5]`` of image i (the HIP proposals are
  compared with them as a set by the test; the RoI stage then runs on THESE boxes so that the
  recorded indices name the same boxes on both sides),
* ``rcnn/pos{i}``, ``rcnn/neg{i}``: sampled indices into ``[GT boxes; proposals]`` (ascending, as
  ``SamplingResult`` holds them),
* ``gs/draw{j}``: the j-th ``np.random.choice`` result of ``_sample_others`` (row indices of the
  1024-row batch) with ``gs/cand{j}`` = the number of candidates it was drawn from,
* the 8 loss terms, the total, full ``fc_cls`` / ``fc_reg`` gradients (strided) and gradient slices
  down to ResNet layer2.

    python tests/golden/make_golden_shipped.py          # authoring container only (~1-2 min)
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests.golden import make_golden_fullsize as F  # noqa: E402

OUT = os.path.join(HERE, 'e2e_train_shipped_samplers_golden.npz')
SEED = 983
NP_SEED = 20260926
GRADS = [
    ('bbox_head.fc_cls.weight', (slice(None, None, 4), slice(None, None, 8))),
    ('bbox_head.fc_cls.bias', (slice(None),)),
    ('bbox_head.fc_reg.weight', (slice(None, None, 16), slice(None, None, 8))),
    ('bbox_head.fc_reg.bias', (slice(None),)),
    ('bbox_head.shared_fcs.0.weight', (slice(None, None, 16), slice(None, None, 256))),
    ('bbox_head.shared_fcs.1.weight', (slice(None, None, 16), slice(None, None, 16))),
    ('rpn_head.rpn_conv.weight', (slice(None, None, 16), slice(None, None, 16))),
    ('rpn_head.rpn_cls.weight', (slice(None),)),
    ('rpn_head.rpn_reg.bias', (slice(None),)),
    ('neck.lateral_convs.0.conv.weight', (slice(None, None, 8), slice(None, None, 8))),
    ('neck.fpn_convs.0.conv.weight', (slice(None, None, 16), slice(None, None, 16))),
    ('backbone.layer2.0.conv1.weight', (slice(None, None, 4), slice(None, None, 8))),
    ('backbone.layer3.5.conv2.weight', (slice(None, None, 16), slice(None, None, 16))),
    ('backbone.layer4.2.bn3.bias', (slice(None, None, 8),)),
]


def configs(table_dir):
    """The shipped config, untouched (bench.detector_cfg == configs/bags/gs_faster_rcnn_r50_fpn_1x_
    lvis_with0_bg8.py with synthetic group tables)."""
    from bench import detector_cfg
    return detector_cfg(table_dir)


def main():
    from balancedgroupsoftmax_amd.config import to_config_dict
    from oracle import det_oracle
    from tests.golden import make_golden_e2e as E
    from tests.golden import make_golden_train as T
    T._bind_reference_ops(forbid_draws=False)         # the samplers draw as shipped
    import importlib
    AT = importlib.import_module("mmdet.core.anchor.anchor_target")
    from mmdet.core.bbox.samplers.base_sampler import BaseSampler
    from mmdet.models import build_detector
    rec = {}

    # ---- recorders (pure observers: every wrapped function returns what the original returned)
    at_single = AT.anchor_target_single
    rpn_calls = []

    def at_single_rec(*a, **k):
        out = at_single(*a, **k)
        labels, label_weights = out[0], out[1]
        i = len(rpn_calls)
        rec['rpn/pos%d' % i] = torch.nonzero(labels == 1).view(-1).numpy().astype(np.int32)
        rec['rpn/neg%d' % i] = torch.nonzero((label_weights > 0) & (labels == 0)).view(-1) \
            .numpy().astype(np.int32)
        rpn_calls.append(i)
        return out
    AT.anchor_target_single = at_single_rec

    base_sample = BaseSampler.sample
    rcnn_calls = []

    def sample_rec(self, assign_result, bboxes, gt_bboxes, gt_labels=None, **kw):
        res = base_sample(self, assign_result, bboxes, gt_bboxes, gt_labels, **kw)
        if self.add_gt_as_proposals:                  # the RoI-stage sampler (the RPN's has False)
            i = len(rcnn_calls)
            rec['rcnn/pos%d' % i] = res.pos_inds.numpy().astype(np.int32)
            rec['rcnn/neg%d' % i] = res.neg_inds.numpy().astype(np.int32)
            rcnn_calls.append(i)
        return res
    BaseSampler.sample = sample_rec

    np_choice = np.random.choice
    gs_calls = []

    def choice_rec(a, size=None, replace=True, p=None):
        out = np_choice(a, size, replace=replace, p=p)
        j = len(gs_calls)
        rec['gs/draw%d' % j] = np.asarray(out).astype(np.int32)
        rec['gs/cand%d' % j] = np.array([len(a)], np.int32)
        gs_calls.append(j)
        return out
    np.random.choice = choice_rec

    tmp = tempfile.mkdtemp(prefix='bgs_shipped_')
    model_cfg, train_cfg = configs(tmp)
    model = build_detector(to_config_dict(model_cfg), train_cfg=to_config_dict(train_cfg),
                           test_cfg=to_config_dict(E.TEST_CFG))
    with torch.no_grad():
        det_oracle.fill_detector(model.state_dict(), SEED)
    model.train()
    get_bboxes = model.rpn_head.get_bboxes

    def get_bboxes_rec(*a, **k):
        props = get_bboxes(*a, **k)
        for i, p in enumerate(props):
            rec['proposals%d' % i] = p.detach().numpy().astype(np.float32)
        return props
    model.rpn_head.get_bboxes = get_bboxes_rec

    g = torch.Generator().manual_seed(SEED)
    img = torch.randn(F.IMGS, 3, F.H, F.W, generator=g)
    boxes, labels = F.gt()
    np.random.seed(NP_SEED)
    losses = model.forward_train(img, F.img_meta(), [torch.from_numpy(b) for b in boxes],
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
    out['meta/seed'] = np.array([SEED, NP_SEED], np.int64)
    assert len(rpn_calls) == F.IMGS and len(rcnn_calls) == F.IMGS, (rpn_calls, rcnn_calls)
    for i in range(F.IMGS):
        print('img %d: rpn pos %d neg %d | proposals %d | rcnn pos %d neg %d' % (
            i, len(out['rpn/pos%d' % i]), len(out['rpn/neg%d' % i]), len(out['proposals%d' % i]),
            len(out['rcnn/pos%d' % i]), len(out['rcnn/neg%d' % i])))
    print('gs draws:', [(int(out['gs/cand%d' % j][0]), len(out['gs/draw%d' % j])) for j in gs_calls])
    for k in sorted(out):
        if 'loss/' in k:
            print(k, out[k])
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT))


if __name__ == '__main__':
    main()
