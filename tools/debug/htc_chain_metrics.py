"""Gradient-agreement metrics of tests/test_gpu_htc.py::test_htc_mask_head_chain... under both conv
arithmetic modes (are the bf16x6 ReLU-flip statistics any different from the fp32 MFMA ones?)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import balancedgroupsoftmax_amd as bgs
from balancedgroupsoftmax_amd import functional as BF
from oracle import mask_oracle
from tests.golden import make_golden_htc as G
DEV = 'cuda:0'
z = np.load(os.path.join(os.path.dirname(G.__file__), 'htc_heads_golden.npz'))


def metrics(a, b, tol=2e-4):
    rel = np.abs(a - b) / max(np.abs(b).max(), 1e-12)
    return '%.4f within tol, worst %.3e, rel-L2 %.3e' % ((rel < tol).mean(), rel.max(),
                                                        np.linalg.norm(a - b) / np.linalg.norm(b))


for math in ('f32', 'bf16x6'):
    BF.set_conv_math(math)
    h0 = bgs.build_head(dict(type='HTCMaskHead', **G.mask_head_cfg()))
    h1 = bgs.build_head(dict(type='HTCMaskHead', **G.mask_head_cfg()))
    with torch.no_grad():
        mask_oracle.fill_mask_head(h0.state_dict(), G.MSK['seed'] + 1000)
        mask_oracle.fill_mask_head(h1.state_dict(), G.MSK['seed'] + 2000)
    h0.to(DEV), h1.to(DEV)
    feats, labels, targets = G.mask_inputs()
    x = torch.from_numpy(feats).permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True)
    lab = torch.from_numpy(labels).to(DEV)
    last = h0.res_features(x, None)
    print(math, 'res_feat0 max abs err', np.abs(last.permute(0, 3, 1, 2).detach().cpu().numpy()[:, ::4] - z['msk/res_feat0']).max())
    f1 = h1.upsample_features(h1.res_features(x, last))
    loss = h1.loss_from_features(f1, torch.from_numpy(targets).to(DEV), lab)['loss_mask']
    print(math, 'loss', float(loss.detach()), float(z['msk/loss'][0]))
    loss.sum().backward()
    print(math, 'dx      ', metrics(x.grad.permute(0, 3, 1, 2)[:, :, ::5, ::3].cpu().numpy(), z['msk/dx']))
    print(math, 'dres_w  ', metrics(h1.conv_res.conv.weight.grad[::2, ::2].cpu().numpy(), z['msk/dres_w']))
    print(math, 'dh0_c0_w', metrics(h0.convs[0].conv.weight.grad[::16, ::16].cpu().numpy(), z['msk/dh0_conv0_w']))
