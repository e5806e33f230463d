#!/usr/bin/env python
"""Component ablation of the wide-tile 1x1 kernel (timing only — results are wrong by construction).

    python -m balancedgroupsoftmax_amd.csrc.build --variant ablate
    BGS_LIB_VARIANT=ablate python tools/wide_ablate.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import capi, functional as BF  # noqa: E402
from wide_ab import bench  # noqa: E402

MODES = [(0, 'full kernel'), (1, 'no MFMA'), (2, 'no DMA after the prologue'), (16, 'no A split'),
         (4, 'no epilogue global loads / stores'), (8, 'no epilogue'), (9, 'no MFMA, no epilogue'),
         (10, 'no DMA, no epilogue (MFMA + reads + barriers)'), (11, 'no MFMA, no DMA, no epilogue (loop skeleton)'),
         (3, 'no MFMA, no DMA'), (5, 'no MFMA, no epilogue traffic')]


def main():
    dev = 'cuda:0'
    lib = capi.load()
    os.environ['BGS_CONV_HALO'] = '0'
    for (N, H, W, C, Co, tag) in [(2, 200, 336, 256, 256, 'fpn.lat0 256->256 M=134400 (2100 tiles)'),
                                  (2, 100, 168, 512, 256, 'fpn.lat1 512->256 M=33600 (526 tiles)'),
                                  (2, 50, 84, 256, 1024, 'l3.c3 256->1024 M=8400 (528 tiles)')]:
        x = torch.randn(N, H, W, C, device=dev)
        w = torch.randn(Co, 1, 1, C, device=dev) * 0.05
        b = torch.randn(Co, device=dev)
        gf = 2.0 * N * H * W * C * Co / 1e9
        for nst in (2, 3):
            print('%s, %d stages' % (tag, nst), flush=True)
            for mode, name in MODES:
                lib.bgs_conv_bfx_wide_tuning(2 | (mode << 8), nst, 1)
                us = bench(lambda: BF.conv2d_nhwc(x, w, b, relu=True), iters=20)
                print('  abl %2d  %-52s %7.1f us  (%6.1f TF-equivalent)' % (mode, name, us, gf / us * 1e3), flush=True)
    lib.bgs_conv_bfx_wide_tuning(1, 0, -1)


if __name__ == '__main__':
    main()
