#!/usr/bin/env python
"""Times every conv / linear shape of one cfg[1] forward (2 x 800x1344) with each tile
configuration of the implicit-GEMM kernel.  Run on the GPU box:  python tools/conv_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402

N = 2
# name, H, W, Cin, Cout, R, stride, count per forward
L = [('stem7x7', 800, 1344, 4, 64, 7, 2, 1)]
H1, W1 = 200, 336
L += [('l1.c1(64)', H1, W1, 64, 64, 1, 1, 1), ('l1.c1(256)', H1, W1, 256, 64, 1, 1, 2),
      ('l1.c2', H1, W1, 64, 64, 3, 1, 3), ('l1.c3', H1, W1, 64, 256, 1, 1, 4)]
for i, (pl, hw_in, hw, nb) in enumerate([(128, (200, 336), (100, 168), 4), (256, (100, 168), (50, 84), 6),
                                         (512, (50, 84), (25, 42), 3)], start=2):
    cin = pl * 2
    L += [('l%d.b0.c1' % i, hw_in[0], hw_in[1], cin, pl, 1, 1, 1),
          ('l%d.b0.c2s2' % i, hw_in[0], hw_in[1], pl, pl, 3, 2, 1),
          ('l%d.ds' % i, hw_in[0], hw_in[1], cin, pl * 4, 1, 2, 1),
          ('l%d.c1' % i, hw[0], hw[1], pl * 4, pl, 1, 1, nb - 1),
          ('l%d.c2' % i, hw[0], hw[1], pl, pl, 3, 1, nb - 1),
          ('l%d.c3' % i, hw[0], hw[1], pl, pl * 4, 1, 1, nb)]
for lvl, (h, w, c) in enumerate([(200, 336, 256), (100, 168, 512), (50, 84, 1024), (25, 42, 2048)]):
    L += [('fpn.lat%d' % lvl, h, w, c, 256, 1, 1, 1), ('fpn.out%d' % lvl, h, w, 256, 256, 3, 1, 1)]
for lvl, (h, w) in enumerate([(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]):
    L += [('rpn.conv%d' % lvl, h, w, 256, 256, 3, 1, 1), ('rpn.head%d' % lvl, h, w, 256, 15, 1, 1, 1)]
FC = [('fc1', 1024, 12544, 1024), ('fc2', 1024, 1024, 1024), ('fc_cls', 1024, 1024, 1236),
      ('fc_reg', 1024, 1024, 4924)]


def bench(fn, iters=10):
    # (the first launches after an idle period run up to 8 % slower — clock ramp; with 2 warm-up calls
    #  the FIRST configuration timed for a layer was measurably penalised: warm up properly)
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = 'cuda:0'
    pf = os.environ.get('SWEEP_PF', '2')
    os.environ['BGS_CONV_PF'] = pf
    print('register prefetch depth PF =', pf)
    tiles = os.environ.get('SWEEP_TILES', '0,22,21,11').split(',')
    tot = {t: 0.0 for t in tiles}
    tot['best'] = 0.0
    gf_total = 0.0
    print('%-14s %9s %7s %6s | %s' % ('layer', 'M', 'K', 'Cout', '  '.join('ms(TF)@' + t for t in tiles)))
    for name, H, W, Cin, Cout, R, stride, cnt in L:
        pad = R // 2
        x = torch.randn(N, H, W, Cin, device=dev)
        w = torch.randn(Cout, R, R, Cin, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        Ho = (H + 2 * pad - R) // stride + 1
        Wo = (W + 2 * pad - R) // stride + 1
        M, K = N * Ho * Wo, R * R * Cin
        gf = 2.0 * M * K * Cout / 1e9
        gf_total += gf * cnt
        row = []
        best = 1e9
        for t in tiles:
            os.environ['BGS_CONV_TILE'] = t
            ms = bench(lambda: BF.conv2d_nhwc(x, w, b, stride=stride, pad=pad, relu=True))
            row.append('%7.3f(%5.1f)' % (ms, gf / ms))
            tot[t] += ms * cnt
            if t != '0':
                best = min(best, ms)
        tot['best'] += best * cnt
        print('%-14s %9d %7d %6d | %s  x%d' % (name, M, K, Cout, '  '.join(row), cnt))
    for name, M, K, Cout in FC:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(Cout, K, device=dev) * 0.02
        b = torch.randn(Cout, device=dev)
        gf = 2.0 * M * K * Cout / 1e9
        gf_total += gf
        row = []
        best = 1e9
        for t in tiles:
            os.environ['BGS_CONV_TILE'] = t
            ms = bench(lambda: BF.linear(x, w, b, relu=True))
            row.append('%7.3f(%5.1f)' % (ms, gf / ms))
            tot[t] += ms
            if t != '0':
                best = min(best, ms)
        tot['best'] += best
        print('%-14s %9d %7d %6d | %s' % (name, M, K, Cout, '  '.join(row)))
    print('total GFLOP per forward (2 img): %.1f' % gf_total)
    for t in tiles + ['best']:
        print('tile %-4s total %.3f ms -> %.1f TFLOP/s' % (t, tot[t], gf_total / tot[t]))


if __name__ == '__main__':
    main()
