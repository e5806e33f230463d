#!/usr/bin/env python
"""Component ablation of conv_wgrad_bfx_kernel (timing only — results are wrong by construction).
Needs the `ablate` build:  python -m balancedgroupsoftmax_amd.csrc.build --variant ablate
                           BGS_LIB_VARIANT=ablate python tools/wgrad_ablate.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402

MODES = [(0, 'full kernel'), (1, 'no MFMA'), (2, 'no global loads after stage 0'), (4, 'no operand split'),
         (8, 'no LDS stores (and no split)'), (16, 'no fragment ds_reads'), (17, 'no MFMA, no ds_reads'),
         (12, 'no split, no LDS stores'), (14, 'no loads, no split, no stores (MFMA + ds_reads + barriers)'),
         (29, 'loads + barriers only'), (31, 'barriers only')]


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dev = 'cuda:0'
    BF.set_conv_math('bf16x6')
    for (N, H, W, Cin, Cout, R, tag) in [(2, 200, 336, 256, 256, 3, 'fpn.out0 3x3 256->256 M=134400'),
                                          (2, 50, 84, 256, 256, 3, 'l3.c2 3x3 256->256 M=8400'),
                                          (2, 100, 168, 512, 128, 1, 'l2.c1 1x1 512->128 M=33600'),
                                          (2, 50, 84, 256, 1024, 1, 'l3.c3 1x1 256->1024 M=8400')]:
        x = torch.randn(N, H, W, Cin, device=dev)
        dy = torch.randn(N, H, W, Cout, device=dev)
        gf = 2.0 * N * H * W * R * R * Cin * Cout / 1e9
        print(tag, flush=True)
        for mode, name in MODES:
            os.environ['BGS_WGRAD_ABLATE'] = str(mode)
            ms = timeit(lambda: BF.conv2d_wgrad_nhwc(x, dy, R, pad=R // 2, bias=True))
            print('  abl %2d  %-62s %7.3f ms  (%6.1f TF-equivalent)' % (mode, name, ms, gf / ms), flush=True)
        os.environ['BGS_WGRAD_ABLATE'] = '0'


if __name__ == '__main__':
    main()
