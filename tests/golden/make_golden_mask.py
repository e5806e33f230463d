#!/usr/bin/env python
"""Generates ``tests/golden/mask_head_golden.npz`` by EXECUTING the reference's ``FCNMaskHead``
(mmdet/models/mask_heads/fcn_mask_head.py) forward + loss on CPU (stubs: oracle/ref_import.py).

    python tests/golden/make_golden_mask.py          # authoring container only

Parameters come from ``oracle.mask_oracle.fill_mask_head(seed)``, inputs from the case seed; stored:
the loss, the GT-channel logits ``mask_pred[i, label_i]`` and, for the loss gradient, the
gradient w.r.t. the RoI features (row subset) and the touched rows of ``conv_logits.weight.grad``.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import mask_oracle, ref_import  # noqa: E402

OUT = os.path.join(HERE, 'mask_head_golden.npz')
CASES = [dict(name='p6_c1231', P=6, C=1231, seed=301), dict(name='p3_c11', P=3, C=11, seed=302)]


def case_inputs(case):
    rs = np.random.RandomState(case['seed'])
    P = case['P']
    feats = rs.standard_normal((P, 256, 14, 14)).astype(np.float32)           # NCHW (reference)
    labels = rs.randint(1, case['C'], size=P).astype(np.int64)
    targets = (rs.rand(P, 28, 28) > 0.5).astype(np.float32)
    return feats, labels, targets


def main():
    ref_import.install_stubs()
    from mmdet.models.mask_heads.fcn_mask_head import FCNMaskHead
    out = {'__cases__': np.frombuffer(json.dumps(CASES).encode(), dtype=np.uint8)}
    for case in CASES:
        head = FCNMaskHead(num_convs=4, in_channels=256, conv_out_channels=256,
                           num_classes=case['C'],
                           loss_mask=dict(type='CrossEntropyLoss', use_mask=True, loss_weight=1.0))
        with torch.no_grad():
            mask_oracle.fill_mask_head(head.state_dict(), case['seed'] + 1000)
        feats, labels, targets = case_inputs(case)
        x = torch.from_numpy(feats).requires_grad_(True)
        pred = head(x)
        loss = head.loss(pred, torch.from_numpy(targets), torch.from_numpy(labels))['loss_mask']
        loss.sum().backward()
        n = case['name']
        idx = torch.arange(case['P'])
        out[n + '/loss'] = loss.detach().numpy().astype(np.float32)
        out[n + '/gt_logits'] = pred[idx, torch.from_numpy(labels)].detach().numpy()
        out[n + '/dx'] = x.grad[:, :, ::5, ::3].contiguous().numpy()
        out[n + '/dw_rows'] = head.conv_logits.weight.grad[torch.from_numpy(labels)].numpy()
        out[n + '/dconv0_w'] = head.convs[0].conv.weight.grad[::16, ::16].contiguous().numpy()
        out[n + '/dup_b'] = head.upsample.bias.grad.numpy()
        print(n, float(loss))
    np.savez_compressed(OUT, **out)
    print('wrote', OUT, os.path.getsize(OUT))


if __name__ == '__main__':
    main()
