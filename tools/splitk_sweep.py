#!/usr/bin/env python
"""Times the small-grid conv / linear shapes of cfg[1] with different split-K factors
(BGS_CONV_SPLITK = 1 disables, unset = the library's own choice).  python tools/splitk_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402
from conv_sweep import FC, L, N, bench  # noqa: E402


def main():
    dev = 'cuda:0'
    opts = ['1', 'auto', '2', '3', '4', '6', '8'] if len(sys.argv) < 2 else sys.argv[1].split(',')
    tot = {o: 0.0 for o in opts}
    print('%-12s %7s %6s %6s %5s | %s' % ('layer', 'M', 'K', 'Cout', 'WGs', '  '.join('%7s' % o for o in opts)))
    rows = [(n, H, W, Cin, Cout, R, st, cnt) for n, H, W, Cin, Cout, R, st, cnt in L]
    rows += [(n, 1, M, K, Co, 1, 1, 1) for n, M, K, Co in FC]
    for name, H, W, Cin, Cout, R, stride, cnt in rows:
        pad = R // 2
        nimg = 1 if name.startswith('fc') else N
        x = torch.randn(nimg, H, W, Cin, device=dev)
        w = torch.randn(Cout, R, R, Cin, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
        M = nimg * Ho * Wo
        wgs = ((M + 63) // 64) * ((Cout + 63) // 64)
        if wgs >= 1500:
            continue
        flops = 2.0 * M * Cout * R * R * Cin
        cells = []
        ref = None
        for o in opts:
            if o == 'auto':
                os.environ.pop('BGS_CONV_SPLITK', None)
            else:
                os.environ['BGS_CONV_SPLITK'] = o
            ms = bench(lambda: BF.conv2d_nhwc(x, w, b, stride=stride, pad=pad, relu=True))
            if o == '1':
                ref = BF.conv2d_nhwc(x, w, b, stride=stride, pad=pad, relu=True)
            else:
                y = BF.conv2d_nhwc(x, w, b, stride=stride, pad=pad, relu=True)
                err = float((y - ref).abs().max() / ref.abs().max().clamp(min=1e-6))
                assert err < 1e-4, (name, o, err)
            tot[o] += ms * cnt
            cells.append('%.3f(%3.0f)' % (ms, flops / ms / 1e9))
        print('%-12s %7d %6d %6d %5d | %s  x%d' % (name, M, R * R * Cin, Cout, wgs, '  '.join(cells), cnt))
    os.environ.pop('BGS_CONV_SPLITK', None)
    print('totals (ms per forward of these layers): ' + '  '.join('%s=%.3f' % (o, tot[o]) for o in opts))


if __name__ == '__main__':
    main()
