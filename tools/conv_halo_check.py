#!/usr/bin/env python
"""Experimental halo-resident 3x3 kernel (BGS_CONV_HALO=1) against the general implicit-GEMM
kernel: max error on several shapes (odd sizes, Cout not a multiple of 128), then timing on the
158-GFLOP P2 layer and a few others.   python tools/conv_halo_check.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402


def run(x, w, b, relu, halo):
    if halo:
        os.environ['BGS_CONV_HALO'] = '1'
    else:
        os.environ.pop('BGS_CONV_HALO', None)
    return BF.conv2d_nhwc(x, w, b, stride=1, pad=1, relu=relu)


def bench(fn, iters=10):
    for _ in range(2):
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
    torch.manual_seed(0)
    dev = 'cuda:0'
    ok = True
    for (N, H, W, Cin, Cout, relu) in [(1, 8, 16, 16, 128, False), (2, 13, 21, 64, 256, True),
                                       (1, 25, 42, 256, 200, True), (2, 50, 84, 32, 64, False),
                                       (1, 3, 5, 48, 15, True)]:
        x = torch.randn(N, H, W, Cin, device=dev)
        w = torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        ref = run(x, w, b, relu, False)
        got = run(x, w, b, relu, True)
        err = float((got - ref).abs().max() / ref.abs().max().clamp(min=1e-6))
        print('shape N%d %dx%d %d->%d relu=%d: rel err %.2e' % (N, H, W, Cin, Cout, relu, err))
        ok &= err < 1e-5
    print('CORRECT' if ok else 'MISMATCH')
    if not ok and '--time-anyway' not in sys.argv:
        return
    for name, (N, H, W, Cin, Cout) in [('fpn.out0', (2, 200, 336, 256, 256)), ('fpn.out1', (2, 100, 168, 256, 256)),
                                       ('l2.c2', (2, 100, 168, 128, 128)), ('l3.c2', (2, 50, 84, 256, 256)),
                                       ('l1.c2', (2, 200, 336, 64, 64))]:
        x = torch.randn(N, H, W, Cin, device=dev)
        w = torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        gf = 2.0 * N * H * W * 9 * Cin * Cout / 1e9
        t0 = bench(lambda: run(x, w, b, True, False))
        t1 = bench(lambda: run(x, w, b, True, True))
        print('%-9s general %.3f ms (%5.1f TF)   halo %.3f ms (%5.1f TF)' % (name, t0, gf / t0, t1, gf / t1))
    os.environ.pop('BGS_CONV_HALO', None)


if __name__ == '__main__':
    main()
