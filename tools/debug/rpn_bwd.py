import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from balancedgroupsoftmax_amd import functional as BF
torch.manual_seed(0)
dev = 'cuda:0'
for (h, w) in [(32, 48), (16, 24), (8, 12), (4, 6), (2, 3)]:
    x = torch.randn(2, 256, h, w)
    w1 = (torch.randn(256, 256, 3, 3) * 0.01).requires_grad_(True)
    b1 = torch.zeros(256, requires_grad=True)
    w2 = (torch.randn(15, 256, 1, 1) * 0.01).requires_grad_(True)
    b2 = torch.zeros(15, requires_grad=True)
    xr = x.clone().requires_grad_(True)
    hh = F.relu(F.conv2d(xr, w1, b1, padding=1))
    o = F.conv2d(hh, w2, b2)
    cot = torch.randn(o.shape)
    (o * cot).sum().backward()
    exp = [xr.grad, w1.grad, b1.grad, w2.grad, b2.grad]
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    w1g = w1.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    b1g = b1.detach().to(dev).requires_grad_(True)
    w2g = w2.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    b2g = b2.detach().to(dev).requires_grad_(True)
    hg = BF.conv2d_autograd(xg, w1g, b1g, pad=1, relu=True)
    og = BF.conv2d_autograd(hg, w2g, b2g)
    (og * cot.permute(0, 2, 3, 1).contiguous().to(dev)).sum().backward()
    got = [xg.grad.permute(0, 3, 1, 2), w1g.grad.permute(0, 3, 1, 2), b1g.grad, w2g.grad.permute(0, 3, 1, 2), b2g.grad]
    errs = [float((g.cpu() - e).abs().max() / e.abs().max()) for g, e in zip(got, exp)]
    print((h, w), ['%.2e' % e for e in errs], 'fwd', float((og.detach().cpu().permute(0,3,1,2) - o.detach()).abs().max()))
