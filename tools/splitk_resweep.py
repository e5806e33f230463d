#!/usr/bin/env python
"""Re-tune of the split-K plans of the bf16x6 kernels (64x64 ring: bfx_plan; halo: halo_bfx_plan) on
cfg[1]'s small-grid layers:  python tools/splitk_resweep.py [--out gpurun_out/splitk.txt]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402
from conv_sweep import L as LAYERS, FC, N as NIMG, bench  # noqa: E402

OUT = []


def say(*a):
    s = ' '.join(str(x) for x in a)
    print(s, flush=True)
    OUT.append(s)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    dev = 'cuda:0'
    layers = [(n, H, W, Cin, Cout, R, s, c) for (n, H, W, Cin, Cout, R, s, c) in LAYERS]
    layers += [(n, 1, 1, K, Cout, 1, 1, 1) for (n, M, K, Cout) in FC]
    tot_auto = tot_best = 0.0
    for name, H, W, Cin, Cout, R, stride, cnt in layers:
        Nn = 1024 if name.startswith('fc') else NIMG
        pad = R // 2
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
        M = Nn * Ho * Wo
        halo = R == 3 and stride == 1 and Cin % 16 == 0 and M >= 2000
        wgs = ((M + 63) // 64) * ((Cout + 63) // 64) if not halo else None
        if M > 40000 or name == 'fc1':
            continue
        x = torch.randn(Nn, H, W, Cin, device=dev)
        w = torch.randn(Cout, R, R, Cin, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        r = {}
        BF.conv_bfx_tuning()
        r['auto'] = bench(lambda: BF.conv2d_nhwc(x, w, b, stride=stride, pad=pad, relu=True), iters=20)
        used = BF.conv_bfx_last_launch()
        for s in (1, 2, 3, 4, 5, 6, 8):
            if s > 1 and R * R * Cin // 16 < 4 * s:
                continue
            if halo:
                BF.conv_bfx_tuning(halo_splits=s)
            else:
                BF.conv_bfx_tuning(0, s)
            r[s] = bench(lambda: BF.conv2d_nhwc(x, w, b, stride=stride, pad=pad, relu=True), iters=20)
        BF.conv_bfx_tuning()
        best = min((k for k in r if k != 'auto'), key=lambda k: r[k])
        tot_auto += r['auto'] * cnt
        tot_best += min(r[best], r['auto']) * cnt
        say('%-12s M %6d K %5d Cout %5d %s | auto (%s splits) %.4f | best %s %.4f | %s  x%d'
            % (name, M, R * R * Cin, Cout, 'halo' if halo else 'wgs %d' % wgs,
               used['halo_splits'] if halo else used['splits'], r['auto'], best, r[best],
               ' '.join('%s:%.4f' % (k, v) for k, v in r.items() if k != 'auto'), cnt))
    say('total: auto %.3f ms, best per layer %.3f ms' % (tot_auto, tot_best))
    if a.out:
        os.makedirs(os.path.dirname(a.out) or '.', exist_ok=True)
        with open(a.out, 'w') as f:
            f.write('\n'.join(OUT) + '\n')


if __name__ == '__main__':
    main()
