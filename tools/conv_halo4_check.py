#!/usr/bin/env python
"""Halo kernel variant 4 (filter slices by LDS-DMA) against variant 2 (register-staged slices) on the
GPU box: bit-identity and per-layer timing of cfg[1]'s 3x3 stride-1 layers.

    python tools/conv_halo4_check.py [--out gpurun_out/halo4.txt]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402
from conv_sweep import L as LAYERS, N as NIMG, bench  # noqa: E402

OUT = []


def say(*a):
    s = ' '.join(str(x) for x in a)
    print(s, flush=True)
    OUT.append(s)


def identity():
    torch.manual_seed(0)
    dev = 'cuda:0'
    ok = True
    os.environ['BGS_CONV_HALO'] = '1'
    for (N, H, W, Cin, Cout, relu, hs) in [(1, 8, 16, 16, 128, False, -1), (2, 13, 21, 64, 256, True, -1),
                                           (1, 25, 42, 256, 200, True, 1), (2, 50, 84, 32, 64, False, 1),
                                           (1, 3, 5, 48, 15, True, 3), (1, 19, 37, 128, 64, True, 4),
                                           (2, 40, 56, 64, 128, True, 2)]:
        x = torch.randn(N, H, W, Cin, device=dev) * torch.exp(torch.randn(N, H, W, Cin, device=dev))
        w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
        b = torch.randn(Cout, device=dev)
        BF.conv_bfx_tuning(halo_splits=hs, halo_variant=2)
        y2 = BF.conv2d_nhwc(x, w, b, stride=1, pad=1, relu=relu)
        u2 = BF.conv_bfx_last_launch()
        BF.conv_bfx_tuning(halo_splits=hs, halo_variant=4)
        y4 = BF.conv2d_nhwc(x, w, b, stride=1, pad=1, relu=relu)
        u4 = BF.conv_bfx_last_launch()
        BF.conv_bfx_tuning()
        good = torch.equal(y2, y4) and u2['halo_variant'] == 2 and u4['halo_variant'] == 4
        ok &= bool(good)
        say('halo N%d %dx%d %d->%d nb %d splits %d | v4 == v2 %s  %s'
            % (N, H, W, Cin, Cout, u4['halo_nb'], u4['halo_splits'], torch.equal(y2, y4), 'ok' if good else 'BAD'))
    os.environ.pop('BGS_CONV_HALO', None)
    say('IDENTICAL' if ok else 'MISMATCH')
    return ok


def sweep():
    dev = 'cuda:0'
    os.environ['BGS_CONV_HALO'] = '1'
    tot = dict(v2=0.0, v4=0.0)
    say('%-12s %8s %6s %5s | v2 (register-staged) | v4 DMA filter' % ('layer', 'M', 'K', 'Cout'))
    for name, H, W, Cin, Cout, R, stride, cnt in LAYERS:
        if not (R == 3 and stride == 1 and Cin % 16 == 0 and NIMG * H * W >= 2000):
            continue
        x = torch.randn(NIMG, H, W, Cin, device=dev)
        w = torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        M = NIMG * H * W
        gf = 2.0 * M * 9 * Cin * Cout / 1e9
        BF.conv_bfx_tuning(halo_variant=2)
        t2 = bench(lambda: BF.conv2d_nhwc(x, w, b, pad=1, relu=True), iters=20)
        BF.conv_bfx_tuning(halo_variant=4)
        t4 = bench(lambda: BF.conv2d_nhwc(x, w, b, pad=1, relu=True), iters=20)
        BF.conv_bfx_tuning()
        tot['v2'] += t2 * cnt
        tot['v4'] += t4 * cnt
        say('%-12s %8d %6d %5d | %6.3f (%5.1f) | %6.3f (%5.1f)   x%d'
            % (name, M, 9 * Cin, Cout, t2, gf / t2, t4, gf / t4, cnt))
    os.environ.pop('BGS_CONV_HALO', None)
    say('3x3 stride-1 layers per forward: v2 %.3f ms | v4 %.3f' % (tot['v2'], tot['v4']))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--no-sweep', action='store_true')
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    ok = identity()
    if not a.no_sweep:
        sweep()
    if a.out:
        os.makedirs(os.path.dirname(a.out) or '.', exist_ok=True)
        with open(a.out, 'w') as f:
            f.write('\n'.join(OUT) + '\n')
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
