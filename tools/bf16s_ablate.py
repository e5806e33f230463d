#!/usr/bin/env python
"""Component ablation of conv_bf16s_kernel<P1X1, bf16 out, 3 stages> (timing only — results are wrong by
construction).  Needs the `ablate` build:  python -m balancedgroupsoftmax_amd.csrc.build --variant ablate
                                           BGS_LIB_VARIANT=ablate python tools/bf16s_ablate.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402

MODES = [(0, 'full kernel'), (1, 'no MFMA'), (2, 'no DMA issue after the prologue'), (4, 'no fragment ds_reads'),
         (8, 'no barrier'), (3, 'no MFMA, no DMA'), (5, 'no MFMA, no ds_reads'), (6, 'no DMA, no ds_reads (MFMA + barriers)'),
         (7, 'barriers only'), (12, 'no ds_reads, no barrier'), (14, 'MFMA only'), (15, 'empty loop')]


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def main():
    dev = 'cuda:0'
    for (N, H, W, Cin, Cout, tag) in [(2, 50, 84, 1024, 1024, 'l3 1x1 1024->1024 M=8400 (528 workgroups, 32 stages)'),
                                       (2, 50, 84, 512, 1024, 'l3.0.c1 512->1024 M=8400 (16 stages)'),
                                       (2, 200, 336, 256, 256, 'l1 1x1 256->256 M=134400 (2100 workgroups, 8 stages)')]:
        x = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
        w = torch.randn(Cout, 1, 1, Cin, device=dev) * 0.03
        b = torch.randn(Cout, device=dev)
        out = torch.empty(N, H, W, Cout, device=dev, dtype=torch.bfloat16)
        print(tag, flush=True)
        for mode, name in MODES:
            os.environ['BGS_BF16S_ABLATE'] = str(mode)
            us = timeit(lambda: BF.conv2d_nhwc(x, w, b, relu=True, out=out, out_dtype=torch.bfloat16))
            print('  abl %2d  %-42s %7.1f us' % (mode, name, us), flush=True)
        os.environ['BGS_BF16S_ABLATE'] = '0'


if __name__ == '__main__':
    main()
