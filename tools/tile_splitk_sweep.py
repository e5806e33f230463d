#!/usr/bin/env python
"""Tile x split-K sweep of the small-grid conv / linear shapes of cfg[1]:
BGS_CONV_TILE in {11 (64x64), 21 (128x64), 22 (128x128)} x BGS_CONV_SPLITK.  TFLOP/s per cell.
    python tools/tile_splitk_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402
from conv_sweep import FC, L, N, bench  # noqa: E402

CONFIGS = [('auto', None, None)] + [('t%s/s%s' % (t, k), t, k) for t in ('11', '21', '22')
                                    for k in ('1', '2', '4', '8')]


def main():
    dev = 'cuda:0'
    print('%-12s %7s %6s %6s | %s' % ('layer', 'M', 'K', 'Cout', ' '.join('%7s' % c[0] for c in CONFIGS)))
    rows = [(n, H, W, Cin, Cout, R, st, cnt) for n, H, W, Cin, Cout, R, st, cnt in L]
    rows += [(n, 1, M, K, Co, 1, 1, 1) for n, M, K, Co in FC]
    best_tot, auto_tot = 0.0, 0.0
    for name, H, W, Cin, Cout, R, stride, cnt in rows:
        pad = R // 2
        nimg = 1 if name.startswith('fc') else N
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
        M = nimg * Ho * Wo
        if ((M + 63) // 64) * ((Cout + 63) // 64) >= 1500 or Cout < 64:
            continue
        x = torch.randn(nimg, H, W, Cin, device=dev)
        w = torch.randn(Cout, R, R, Cin, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        flops = 2.0 * M * Cout * R * R * Cin
        cells, times = [], []
        for label, t, k in CONFIGS:
            for key, val in (('BGS_CONV_TILE', t), ('BGS_CONV_SPLITK', k)):
                if val is None:
                    os.environ.pop(key, None)
                else:
                    os.environ[key] = val
            ms = bench(lambda: BF.conv2d_nhwc(x, w, b, stride=stride, pad=pad, relu=True))
            times.append(ms)
            cells.append('%7.0f' % (flops / ms / 1e9))
        auto_tot += times[0] * cnt
        best_tot += min(times) * cnt
        bi = times.index(min(times))
        print('%-12s %7d %6d %6d | %s   best %s x%d' % (name, M, R * R * Cin, Cout, ' '.join(cells),
                                                       CONFIGS[bi][0], cnt))
    for key in ('BGS_CONV_TILE', 'BGS_CONV_SPLITK'):
        os.environ.pop(key, None)
    print('total ms: auto %.3f, best-per-layer %.3f' % (auto_tot, best_tot))


if __name__ == '__main__':
    main()
