"""Split-M factor of the bf16x6 weight-gradient kernel per layer (BGS_WGRAD_SPLITS): the plan was
tuned on the fp32-MFMA kernel; layers with few output tiles (1x1 convs with small Cout x K) run one
workgroup per CU at 128 slices."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import functional as BF
from conv_sweep import L as LAYERS, FC, N as NIMG, bench
dev = 'cuda:0'
layers = [l for l in LAYERS if not (l[0].startswith('stem') or l[0].startswith('l1.') or 'head' in l[0])]
layers += [(n, 1, 1, K, Cout, 1, 1, 1) for (n, M, K, Cout) in FC]
seen = set()
for name, H, W, Cin, Cout, R, stride, cnt in layers:
    key = (H, W, Cin, Cout, R, stride)
    if key in seen:
        continue
    seen.add(key)
    Nn = 1024 if name.startswith('fc') else NIMG
    pad = R // 2
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    M, K = Nn * Ho * Wo, R * R * Cin
    tiles = ((Cout + 127) // 128) * ((K + 127) // 128)
    x = torch.randn(Nn, H, W, Cin, device=dev); dy = torch.randn(Nn, Ho, Wo, Cout, device=dev)
    row = []
    for sp in ('auto', 64, 128, 256, 512, 1024):
        if sp != 'auto' and (sp * tiles > 16384 or M // sp < 64):
            continue
        if sp == 'auto':
            os.environ.pop('BGS_WGRAD_SPLITS', None)
        else:
            os.environ['BGS_WGRAD_SPLITS'] = str(sp)
        f = lambda: BF.conv2d_wgrad_nhwc(x, dy, R, stride=stride, pad=pad, bias=True)
        f()
        row.append('%s %.4f' % (sp, bench(f, iters=10)))
    os.environ.pop('BGS_WGRAD_SPLITS', None)
    print('%-12s M %6d K %5d Cout %4d tiles %3d | %s' % (name, M, K, Cout, tiles, ' | '.join(row)), flush=True)
