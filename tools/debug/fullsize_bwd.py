"""Full-size (cfg[1]) conv fwd / dgrad / wgrad of every trainable layer vs torch-GPU autograd."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from balancedgroupsoftmax_amd import functional as BF
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from conv_sweep import L, N
dev = 'cuda:0'
torch.manual_seed(0)
worst = 0
for name, H, W, Cin, Cout, R, stride, cnt in L:
    if name.startswith('stem') or name.startswith('l1.'):
        continue
    pad = R // 2
    x = torch.randn(N, H, W, Cin, device=dev)
    w = torch.randn(Cout, R, R, Cin, device=dev) / (R * R * Cin) ** 0.5
    b = torch.randn(Cout, device=dev)
    xr = x.permute(0, 3, 1, 2).detach().requires_grad_(True)
    wr = w.permute(0, 3, 1, 2).detach().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, br, stride=stride, padding=pad)
    cot = torch.randn_like(y)
    y.backward(cot)
    xg = x.clone().requires_grad_(True); wg = w.clone().requires_grad_(True); bg = b.clone().requires_grad_(True)
    yg = BF.conv2d_autograd(xg, wg, bg, stride=stride, pad=pad)
    yg.backward(cot.permute(0, 2, 3, 1).contiguous())
    def rel(a, e):
        return float((a - e).abs().max() / e.abs().max())
    errs = (rel(yg.detach().permute(0, 3, 1, 2), y.detach()), rel(xg.grad.permute(0, 3, 1, 2), xr.grad),
            rel(wg.grad.permute(0, 3, 1, 2), wr.grad), rel(bg.grad, br.grad))
    worst = max(worst, max(errs))
    flag = '  <<<<<<' if max(errs) > 1e-3 else ''
    print('%-12s M=%7d K=%5d Cout=%5d  fwd %.1e dx %.1e dw %.1e db %.1e%s' % (
        name, N * ((H + 2 * pad - R) // stride + 1) * ((W + 2 * pad - R) // stride + 1), R * R * Cin, Cout, *errs, flag), flush=True)
    del x, w, xr, wr, y, cot, xg, wg, yg
print('worst', worst)
