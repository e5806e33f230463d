#!/usr/bin/env python
"""Times the weight-gradient kernel on every trainable conv / linear shape of cfg[1] (selectp=0),
optionally with forced split counts:  python tools/wgrad_sweep.py [splits,splits,...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402
from conv_sweep import FC, L, N, bench  # noqa: E402


def main():
    dev = 'cuda:0'
    opts = ['auto'] + (sys.argv[1].split(',') if len(sys.argv) > 1 else [])
    tot = {o: 0.0 for o in opts}
    gf = 0.0
    rows = [r for r in L if not (r[0].startswith('stem') or r[0].startswith('l1.'))]
    rows += [(n, 1, M, K, Co, 1, 1, 1) for n, M, K, Co in FC]
    print('%-12s %7s %6s %6s | %s' % ('layer', 'M', 'K', 'Cout', '  '.join('%12s' % o for o in opts)))
    for name, H, W, Cin, Cout, R, stride, cnt in rows:
        pad = R // 2
        nimg = 1 if name.startswith('fc') else N
        if Cout % 4:
            Cout += 4 - Cout % 4
        x = torch.randn(nimg, H, W, Cin, device=dev)
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
        dy = torch.randn(nimg, Ho, Wo, Cout, device=dev)
        M = nimg * Ho * Wo
        flops = 2.0 * M * Cout * R * R * Cin
        cells = []
        for o in opts:
            if o == 'auto':
                os.environ.pop('BGS_WGRAD_SPLITS', None)
            else:
                os.environ['BGS_WGRAD_SPLITS'] = o
            ms = bench(lambda: BF.conv2d_wgrad_nhwc(x, dy, R, stride=stride, pad=pad, bias=True))
            tot[o] += ms * cnt
            cells.append('%.3f(%5.1f)' % (ms, flops / ms / 1e9))
        gf += flops * cnt / 1e9
        print('%-12s %7d %6d %6d | %s  x%d' % (name, M, R * R * Cin, Cout, '  '.join(cells), cnt))
    os.environ.pop('BGS_WGRAD_SPLITS', None)
    print('total GFLOP %.1f; ' % gf + '  '.join('%s=%.3f ms (%.1f TF)' % (o, tot[o], gf / tot[o]) for o in opts))


if __name__ == '__main__':
    main()
