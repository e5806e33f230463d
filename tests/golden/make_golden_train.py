#!/usr/bin/env python
"""Generates ``tests/golden/e2e_train_golden.npz`` by EXECUTING THE REFERENCE DETECTOR'S TRAINING
ITERATION on CPU: ``GroupSoftmax`` (= ``TwoStageDetector.forward_train``, two_stage.py:134-265) of
configs/bags/gs_faster_rcnn_r50_fpn_1x_lvis_with0_bg8.py -> the full loss dict -> ``backward``.

The reference draws its random samples with numpy on the host (random_sampler.py:19-33,
gs_bbox_head_with0.py:83), which no other implementation can reproduce draw for draw.  The three
sampling steps are therefore configured so that they take EVERY candidate (then no random number is
consumed and the iteration is a deterministic function of the weights and inputs):

* RPN sampler ``num=16384`` (> the 12,276 anchors of the 192x256 input),
* RCNN sampler ``num=512`` with ``rpn_proposal.max_num=300`` (<= 300 proposals + 6 GT, fewer than
  128 positives),
* ``others_sample_ratio=1e6`` (``int(n_fg * ratio) >= n_bg`` -> all weights 1, :80-81).

``RandomSampler.random_choice`` and ``np.random.choice`` are replaced by functions that raise, so a
draw would abort the generation.  Everything else is the shipped config.  The reference's compiled
ops are its own sources built for the host (oracle/build_ref.py): ``nms_cpu.cpp`` and both
``ROIAlignForward`` / ``ROIAlignBackward`` kernels of ``roi_align_kernel.cu``.

    python tests/golden/make_golden_train.py          # authoring container only
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
from tests.golden import make_golden_e2e as E  # noqa: E402

OUT = os.path.join(HERE, 'e2e_train_golden.npz')
SEED = 911
# (parameter name, index expression) of the stored gradient slices
GRADS = [
    ('bbox_head.fc_cls.weight', (slice(None, None, 16), slice(None, None, 16))),
    ('bbox_head.fc_cls.bias', (slice(None),)),
    ('bbox_head.fc_reg.weight', (slice(None, None, 64), slice(None, None, 16))),
    ('bbox_head.shared_fcs.0.weight', (slice(None, None, 16), slice(None, None, 256))),
    ('rpn_head.rpn_conv.weight', (slice(None, None, 16), slice(None, None, 16))),
    ('rpn_head.rpn_cls.weight', (slice(None),)),
    ('rpn_head.rpn_reg.bias', (slice(None),)),
    ('neck.lateral_convs.0.conv.weight', (slice(None, None, 8), slice(None, None, 8))),
    ('neck.fpn_convs.2.conv.weight', (slice(None, None, 16), slice(None, None, 16))),
    ('backbone.layer2.0.conv1.weight', (slice(None, None, 4), slice(None, None, 8))),
    ('backbone.layer2.0.bn1.weight', (slice(None),)),
    ('backbone.layer2.0.downsample.0.weight', (slice(None, None, 16), slice(None, None, 16))),
    ('backbone.layer3.5.conv2.weight', (slice(None, None, 16), slice(None, None, 16))),
    ('backbone.layer4.2.bn3.bias', (slice(None, None, 8),)),
]


GRADS_HTC = [
    ('bbox_head.0.fc_cls.weight', (slice(None, None, 16), slice(None, None, 16))),
    ('bbox_head.1.fc_cls.weight', (slice(None, None, 16), slice(None, None, 16))),
    ('bbox_head.2.fc_cls.weight', (slice(None, None, 16), slice(None, None, 16))),
    ('bbox_head.1.shared_fcs.0.weight', (slice(None, None, 16), slice(None, None, 256))),
    ('bbox_head.2.fc_reg.bias', (slice(None),)),
    ('mask_head.0.convs.0.conv.weight', (slice(None, None, 16), slice(None, None, 16))),
    ('mask_head.1.conv_res.conv.weight', (slice(None, None, 4), slice(None, None, 4))),
    ('mask_head.2.conv_logits.weight', (slice(None, None, 16),)),
    ('mask_head.2.upsample.bias', (slice(None),)),
    ('semantic_head.conv_logits.weight', (slice(None, None, 4), slice(None, None, 4))),
    ('semantic_head.lateral_convs.3.conv.weight', (slice(None, None, 8), slice(None, None, 8))),
    ('semantic_head.conv_embedding.conv.bias', (slice(None),)),
    ('neck.fpn_convs.1.conv.weight', (slice(None, None, 16), slice(None, None, 16))),
    ('rpn_head.rpn_conv.weight', (slice(None, None, 16), slice(None, None, 16))),
    ('backbone.layer3.0.conv1.weight', (slice(None, None, 8), slice(None, None, 8))),
    ('backbone.layer2.1.bn2.weight', (slice(None),)),
]
HTC_SEED = 921
GRADS_MASK = [
    ('bbox_head.fc_cls.weight', (slice(None, None, 16), slice(None, None, 16))),
    ('mask_head.convs.0.conv.weight', (slice(None, None, 16), slice(None, None, 16))),
    ('mask_head.convs.3.conv.bias', (slice(None),)),
    ('mask_head.upsample.weight', (slice(None, None, 8), slice(None, None, 8))),
    ('mask_head.conv_logits.weight', (slice(None, None, 16),)),
    ('mask_head.conv_logits.bias', (slice(None),)),
    ('neck.fpn_convs.0.conv.weight', (slice(None, None, 16), slice(None, None, 16))),
    ('backbone.layer2.3.conv3.weight', (slice(None, None, 16), slice(None, None, 8))),
]
MASK_SEED = 931
GRADS_CASCADE = [
    ('bbox_head.0.fc_cls.weight', (slice(None, None, 16), slice(None, None, 16))),
    ('bbox_head.1.fc_cls.bias', (slice(None),)),
    ('bbox_head.2.fc_reg.weight', (slice(None), slice(None, None, 16))),
    ('bbox_head.2.shared_fcs.1.weight', (slice(None, None, 16), slice(None, None, 16))),
    ('neck.lateral_convs.2.conv.weight', (slice(None, None, 8), slice(None, None, 16))),
    ('backbone.layer4.0.conv2.weight', (slice(None, None, 16), slice(None, None, 16))),
]
CASCADE_SEED = 941


def gt():
    rs = np.random.RandomState(SEED)
    n = 6
    wh = rs.uniform(24, 110, size=(n, 2))
    xy = rs.uniform(0, 1, size=(n, 2)) * (np.array([E.W - 3, E.H]) - wh - 1)
    boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
    labels = rs.randint(1, 1231, size=n).astype(np.int64)
    return boxes, labels


def gt_masks(boxes):
    """One bitmap per GT: an axis-aligned ellipse inscribed in its box, ``[G, H, W]`` uint8."""
    yy = np.arange(E.H, dtype=np.float32)[:, None]
    xx = np.arange(E.W, dtype=np.float32)[None, :]
    out = []
    for x1, y1, x2, y2 in boxes:
        cx, cy, rx, ry = (x1 + x2) / 2, (y1 + y2) / 2, max((x2 - x1) / 2, 1), max((y2 - y1) / 2, 1)
        out.append(((((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2) <= 1.0).astype(np.uint8))
    return np.stack(out)


def gt_semantic_seg():
    rs = np.random.RandomState(SEED + 5)
    seg = rs.randint(0, 183, size=(1, 1, E.H // 8, E.W // 8)).astype(np.int64)
    seg[rs.rand(*seg.shape) < 0.2] = 255
    return seg


def configs(table_dir, htc=False, mask=False, cascade=False):
    from bench import detector_cfg
    model, train_cfg = detector_cfg(table_dir, htc=htc, mask=mask, cascade=cascade)
    heads = model['bbox_head'] if (htc or cascade) else [model['bbox_head']]
    for h in heads:
        h['gs_config']['others_sample_ratio'] = 1e6
    if htc or cascade:         # plain ResNet-50 trunk: full backward on both sides
        model['backbone'] = dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3),
                                 frozen_stages=1, style='pytorch')
    train_cfg['rpn']['sampler']['num'] = 16384
    train_cfg['rpn_proposal'].update(nms_post=300, max_num=300)
    return model, train_cfg


def _bind_reference_ops(forbid_draws=True):
    """``forbid_draws=False``: leave the reference's numpy samplers as shipped (used by
    tools/ref_cpu_detector_time.py, which times the real iteration)."""
    E._bind_reference_ops()
    ra = sys.modules['mmdet.ops.roi_align.roi_align']

    class _RoIAlignRef(torch.autograd.Function):
        """RoIAlignFunction (roi_align.py:9-53) with the compiled reference kernels."""

        @staticmethod
        def forward(ctx, features, rois, out_size, spatial_scale, sample_num):
            ctx.save_for_backward(rois)
            ctx.cfg = (tuple(features.shape), spatial_scale, sample_num)
            return torch.from_numpy(build_ref.roi_align_reference(
                features.detach().numpy(), rois.detach().numpy(), spatial_scale, out_size[0],
                sample_num))

        @staticmethod
        def backward(ctx, grad_output):
            rois, = ctx.saved_tensors
            shape, scale, sample_num = ctx.cfg
            g = build_ref.roi_align_reference_backward(grad_output.contiguous().numpy(),
                                                       rois.numpy(), scale, shape, sample_num)
            return torch.from_numpy(g), None, None, None, None
    ra.roi_align = lambda f, r, o, s, n=0: _RoIAlignRef.apply(f, r, o, s, n)

    def no_draw(*a, **k):
        raise AssertionError('a random draw was requested: the golden must be deterministic')
    if forbid_draws:
        from mmdet.core.bbox.samplers.random_sampler import RandomSampler
        RandomSampler.random_choice = staticmethod(no_draw)
        np.random.choice = no_draw
    # mask_target.py:31 resizes the cropped bitmap with mmcv.imresize = cv2.resize(INTER_LINEAR);
    # neither is installed here: oracle/mask_oracle.py restates OpenCV's fixed-point path
    # ("parity unpinned" for that one step, see its header)
    from oracle import mask_oracle
    sys.modules['mmcv'].imresize = lambda img, size: mask_oracle.resize_linear_u8(img, size)


def main():
    from balancedgroupsoftmax_amd.config import to_config_dict
    _bind_reference_ops()
    from mmdet.models import build_detector
    tmp = tempfile.mkdtemp(prefix='bgs_e2e_')
    model_cfg, train_cfg = configs(tmp)
    model = build_detector(to_config_dict(model_cfg), train_cfg=to_config_dict(train_cfg),
                           test_cfg=to_config_dict(E.TEST_CFG))
    with torch.no_grad():
        det_oracle.fill_detector(model.state_dict(), SEED)
    model.train()                       # norm_eval=True keeps BN in eval; stem + layer1 frozen
    boxes, labels = gt()
    losses = model.forward_train(torch.from_numpy(E.image()), E.img_meta(),
                                 [torch.from_numpy(boxes)], [torch.from_numpy(labels)])
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
    assert params['backbone.layer1.0.conv1.weight'].grad is None      # frozen_stages=1

    # ------------------------------------------------------------ HTC (R50 trunk) training step
    model_cfg, train_cfg = configs(tmp, htc=True)
    model = build_detector(to_config_dict(model_cfg), train_cfg=to_config_dict(train_cfg),
                           test_cfg=to_config_dict(E.TEST_CFG))
    with torch.no_grad():
        det_oracle.fill_detector(model.state_dict(), HTC_SEED)
    model.train()

    class _ReluCopy(torch.nn.Module):       # see make_golden_htc.py: in-place += on a ReLU output
        def forward(self, t):
            return torch.relu(t) * 1.0
    model.semantic_head.lateral_convs[model.semantic_head.fusion_level].activate = _ReluCopy()
    losses = model.forward_train(torch.from_numpy(E.image()), E.img_meta(),
                                 [torch.from_numpy(boxes)], [torch.from_numpy(labels)],
                                 gt_masks=[gt_masks(boxes)],
                                 gt_semantic_seg=torch.from_numpy(gt_semantic_seg()))
    total = 0
    for k, v in losses.items():
        vals = v if isinstance(v, list) else [v]
        out['htc/loss/' + k] = np.array([float(t.detach().sum()) for t in vals], np.float32)
        if 'loss' in k:
            total = total + sum(t.sum() for t in vals)
    total.backward()
    out['htc/loss/total'] = np.array([float(total.detach())], np.float32)
    params = dict(model.named_parameters())
    for name, idx in GRADS_HTC:
        out['htc/grad/' + name] = params[name].grad[idx].contiguous().numpy()

    # ------------------------------------------------------------ Mask R-CNN R50 (cfg[3])
    model_cfg, train_cfg = configs(tmp, mask=True)
    model = build_detector(to_config_dict(model_cfg), train_cfg=to_config_dict(train_cfg),
                           test_cfg=to_config_dict(E.TEST_CFG))
    with torch.no_grad():
        det_oracle.fill_detector(model.state_dict(), MASK_SEED)
    model.train()
    losses = model.forward_train(torch.from_numpy(E.image()), E.img_meta(),
                                 [torch.from_numpy(boxes)], [torch.from_numpy(labels)],
                                 gt_masks=[gt_masks(boxes)])
    total = 0
    for k, v in losses.items():
        vals = v if isinstance(v, list) else [v]
        out['mask/loss/' + k] = np.array([float(t.detach().sum()) for t in vals], np.float32)
        if 'loss' in k:
            total = total + sum(t.sum() for t in vals)
    total.backward()
    out['mask/loss/total'] = np.array([float(total.detach())], np.float32)
    params = dict(model.named_parameters())
    for name, idx in GRADS_MASK:
        out['mask/grad/' + name] = params[name].grad[idx].contiguous().numpy()

    # ------------------------------------------------------------ Cascade R-CNN (cfg[4], R50 trunk)
    model_cfg, train_cfg = configs(tmp, cascade=True)
    model = build_detector(to_config_dict(model_cfg), train_cfg=to_config_dict(train_cfg),
                           test_cfg=to_config_dict(E.TEST_CFG))
    with torch.no_grad():
        det_oracle.fill_detector(model.state_dict(), CASCADE_SEED)
    model.train()
    losses = model.forward_train(torch.from_numpy(E.image()), E.img_meta(),
                                 [torch.from_numpy(boxes)], [torch.from_numpy(labels)])
    total = 0
    for k, v in losses.items():
        vals = v if isinstance(v, list) else [v]
        out['cascade/loss/' + k] = np.array([float(t.detach().sum()) for t in vals], np.float32)
        if 'loss' in k:
            total = total + sum(t.sum() for t in vals)
    total.backward()
    out['cascade/loss/total'] = np.array([float(total.detach())], np.float32)
    params = dict(model.named_parameters())
    for name, idx in GRADS_CASCADE:
        out['cascade/grad/' + name] = params[name].grad[idx].contiguous().numpy()
    for k in sorted(out):
        if 'loss/' in k:
            print(k, out[k])
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT))


if __name__ == '__main__':
    main()
