#!/usr/bin/env python
"""Generates ``tests/golden/htc_heads_golden.npz`` by EXECUTING the reference's HTC head classes
on CPU (stubs: oracle/ref_import.py):

* ``FusedSemanticHead`` (mmdet/models/mask_heads/fused_semantic_head.py): forward + loss +
  backward on five small FPN-shaped levels;
* ``HTCMaskHead`` (mmdet/models/mask_heads/htc_mask_head.py): the mask-information-flow chain of
  two heads exactly as ``HybridTaskCascade._mask_forward_train`` runs it for stage 1
  (htc.py:98-107), then ``loss`` + backward.

    python tests/golden/make_golden_htc.py          # authoring container only

Parameters come from ``oracle.mask_oracle.fill_mask_head(seed)`` (any state_dict), inputs from the
case seeds below.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import mask_oracle, ref_import  # noqa: E402

OUT = os.path.join(HERE, 'htc_heads_golden.npz')
SEM = dict(seed=411, sizes=[(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)], fusion_level=1,
           num_classes=183, ignore=255)
MSK = dict(seed=421, P=5, C=37)


def semantic_inputs():
    rs = np.random.RandomState(SEM['seed'])
    feats = [rs.standard_normal((2, 256, h, w)).astype(np.float32) for h, w in SEM['sizes']]
    h, w = SEM['sizes'][SEM['fusion_level']]
    labels = rs.randint(0, SEM['num_classes'], size=(2, 1, h, w)).astype(np.int64)
    labels[rs.rand(2, 1, h, w) < 0.2] = SEM['ignore']
    return feats, labels


def mask_inputs():
    rs = np.random.RandomState(MSK['seed'])
    P = MSK['P']
    feats = rs.standard_normal((P, 256, 14, 14)).astype(np.float32)
    labels = rs.randint(1, MSK['C'], size=P).astype(np.int64)
    targets = (rs.rand(P, 28, 28) > 0.5).astype(np.float32)
    return feats, labels, targets


def semantic_head_cfg():
    return dict(num_ins=5, fusion_level=SEM['fusion_level'], num_convs=4, in_channels=256,
                conv_out_channels=256, num_classes=SEM['num_classes'], ignore_label=SEM['ignore'],
                loss_weight=0.2)


def mask_head_cfg():
    return dict(num_convs=4, in_channels=256, conv_out_channels=256, num_classes=MSK['C'],
                loss_mask=dict(type='CrossEntropyLoss', use_mask=True, loss_weight=1.0))


def main():
    ref_import.install_stubs()
    from mmdet.models.mask_heads.fused_semantic_head import FusedSemanticHead
    from mmdet.models.mask_heads.htc_mask_head import HTCMaskHead
    out = {}

    head = FusedSemanticHead(**semantic_head_cfg())
    with torch.no_grad():
        mask_oracle.fill_mask_head(head.state_dict(), SEM['seed'] + 1000)
    feats, labels = semantic_inputs()
    xs = [torch.from_numpy(f).requires_grad_(True) for f in feats]

    class _ReluCopy(torch.nn.Module):
        """The reference accumulates ``x += ...`` in place into the ReLU output of the fusion
        level's lateral conv (fused_semantic_head.py:87,93); torch >= 1.5 saves that output for the
        ReLU backward and refuses the in-place update.  Same values through a fresh tensor."""

        def forward(self, t):
            return torch.relu(t) * 1.0
    head.lateral_convs[SEM['fusion_level']].activate = _ReluCopy()
    pred, emb = head(xs)
    loss = head.loss(pred, torch.from_numpy(labels))
    # a second scalar through the embedded feature so that its branch has a gradient too
    rs = np.random.RandomState(SEM['seed'] + 1)
    proj = torch.from_numpy(rs.standard_normal(tuple(emb.shape)).astype(np.float32))
    (loss + (emb * proj).sum() * 1e-3).backward()
    out['sem/pred'] = pred.detach().numpy()
    out['sem/feat'] = emb.detach()[:, ::2].contiguous().numpy()
    out['sem/loss'] = np.array([float(loss)], np.float32)
    out['sem/proj'] = proj.numpy()
    for i, x in enumerate(xs):
        out['sem/dx%d' % i] = x.grad[:, ::4].contiguous().numpy()
    out['sem/dlat0_w'] = head.lateral_convs[0].conv.weight.grad[::2, ::2].contiguous().numpy()
    out['sem/dconv1_w'] = head.convs[1].conv.weight.grad[::8, ::8].contiguous().numpy()
    out['sem/dlogits_b'] = head.conv_logits.bias.grad.numpy()
    out['sem/demb_b'] = head.conv_embedding.conv.bias.grad.numpy()
    print('semantic loss', float(loss))

    h0, h1 = HTCMaskHead(**mask_head_cfg()), HTCMaskHead(**mask_head_cfg())
    with torch.no_grad():
        mask_oracle.fill_mask_head(h0.state_dict(), MSK['seed'] + 1000)
        mask_oracle.fill_mask_head(h1.state_dict(), MSK['seed'] + 2000)
    feats, labels, targets = mask_inputs()
    x = torch.from_numpy(feats).requires_grad_(True)
    lab = torch.from_numpy(labels)
    idx = torch.arange(MSK['P'])
    last = h0(x, None, return_logits=False)                       # htc.py:101-104
    pred1 = h1(x, last, return_feat=False)                        # htc.py:105
    loss = h1.loss(pred1, torch.from_numpy(targets), lab)['loss_mask']
    loss.sum().backward()
    pred0, feat0 = h0(x, None)                                    # test-time form, htc.py:146-148
    out['msk/res_feat0'] = feat0.detach()[:, ::4].contiguous().numpy()
    out['msk/gt_logits0'] = pred0[idx, lab].detach().numpy()
    out['msk/gt_logits1'] = pred1[idx, lab].detach().numpy()
    out['msk/loss'] = loss.detach().numpy().astype(np.float32).reshape(-1)
    out['msk/dx'] = x.grad[:, :, ::5, ::3].contiguous().numpy()
    out['msk/dres_w'] = h1.conv_res.conv.weight.grad[::2, ::2].contiguous().numpy()
    out['msk/dh0_conv0_w'] = h0.convs[0].conv.weight.grad[::16, ::16].contiguous().numpy()
    out['msk/dh1_up_b'] = h1.upsample.bias.grad.numpy()
    print('mask loss', float(loss))
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT))


if __name__ == '__main__':
    main()
