"""A/B of the bf16x6 weight-gradient kernel against the fp32-MFMA one on the trainable layers of a
cfg[1] `selectp=0` backward (layer2-4, FPN, RPN conv, FC heads), interleaved in one process."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import capi, functional as BF
from conv_sweep import L as LAYERS, FC, N as NIMG, bench
lib = capi.load()
dev = 'cuda:0'
layers = [l for l in LAYERS if not (l[0].startswith('stem') or l[0].startswith('l1.') or 'head' in l[0])]
layers += [(n, 1, 1, K, Cout, 1, 1, 1) for (n, M, K, Cout) in FC]
tot = {'f32': 0.0, 'bfx': 0.0}
for name, H, W, Cin, Cout, R, stride, cnt in layers:
    Nn = 1024 if name.startswith('fc') else NIMG
    pad = R // 2
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    x = torch.randn(Nn, H, W, Cin, device=dev); dy = torch.randn(Nn, Ho, Wo, Cout, device=dev)
    r = {}
    for rnd in range(2):
        for key, on in (('f32', 0), ('bfx', 1)):
            lib.bgs_conv2d_wgrad_bfx_enable(on)
            f = lambda: BF.conv2d_wgrad_nhwc(x, dy, R, stride=stride, pad=pad, bias=True)
            f()
            r[key] = min(r.get(key, 1e9), bench(f, iters=10))
    gf = 2.0 * Nn * Ho * Wo * R * R * Cin * Cout / 1e9
    print('%-12s M %6d K %5d Cout %4d | f32 %.4f ms (%.0f TF) | bfx %.4f ms (%.0f TF)  x%d' % (
        name, Nn * Ho * Wo, R * R * Cin, Cout, r['f32'], gf / r['f32'], r['bfx'], gf / r['bfx'], cnt), flush=True)
    for k in tot:
        tot[k] += r[k] * cnt
lib.bgs_conv2d_wgrad_bfx_enable(1)
print('total wgrad per backward: fp32 MFMA %.3f ms  bf16x6 %.3f ms' % (tot['f32'], tot['bfx']))
