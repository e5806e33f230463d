#!/usr/bin/env python
"""Component ablation of the two conv loops (timing only — results are wrong by construction):
which part of a K step the time goes to.  Needs the `ablate` build of the library:

    python -m balancedgroupsoftmax_amd.csrc.build --variant ablate
    BGS_LIB_VARIANT=ablate python tools/ablate.py [--out gpurun_out/ablate.txt]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402
from conv_sweep import bench  # noqa: E402

OUT = []


def say(*a):
    s = ' '.join(str(x) for x in a)
    print(s, flush=True)
    OUT.append(s)


HALO = [(0, 'full kernel'), (1, 'no MFMA'), (2, 'no filter loads/stores'), (4, 'no fragment ds_reads'),
        (8, 'no barriers'), (16, 'no patch reload'), (3, 'no MFMA, no filter traffic'),
        (5, 'no MFMA, no ds_reads'), (6, 'no filter traffic, no ds_reads (MFMA + barriers + patch)'),
        (14, 'MFMA + patch only'), (18, 'no filter traffic, no patch'), (26, '+ no barriers'),
        (30, 'MFMA only')]
DMA = [(0, 'full kernel'), (1, 'no MFMA'), (2, 'no DMA issue'), (4, 'no fragment ds_reads'),
       (8, 'no A split'), (3, 'no MFMA, no DMA'), (9, 'no MFMA, no split'), (12, 'no ds_reads, no split'),
       (6, 'no DMA, no ds_reads')]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    dev = 'cuda:0'
    lib = BF.capi.load()
    for (N, H, W, C, Co, tag) in [(2, 200, 336, 256, 256, 'P2 3x3 256->256 (2100 workgroups)'),
                                  (2, 50, 84, 256, 256, 'P4 3x3 256->256 (split 4)')]:
        x = torch.randn(N, H, W, C, device=dev)
        w = torch.randn(Co, 3, 3, C, device=dev) * 0.02
        b = torch.randn(Co, device=dev)
        gf = 2.0 * N * H * W * 9 * C * Co / 1e9
        say('halo kernel, %s' % tag)
        for mode, name in HALO:
            lib.bgs_conv3x3_halo_bfx_tuning(-1, 2 | (mode << 8))
            ms = bench(lambda: BF.conv2d_nhwc(x, w, b, pad=1, relu=True), iters=20)
            say('  abl %2d  %-55s %7.3f ms  (%6.1f TF-equivalent)' % (mode, name, ms, gf / ms))
        lib.bgs_conv3x3_halo_bfx_tuning(-1, 2)
    os.environ['BGS_CONV_HALO'] = '0'
    for (N, H, W, C, Co, tag) in [(2, 200, 336, 256, 256, 'fpn.lat0 1x1 256->256 M=134400'),
                                  (2, 100, 168, 512, 128, 'l2.c1 1x1 512->128 M=33600'),
                                  (2, 50, 84, 1024, 256, 'l3.c1 1x1 1024->256 M=8400 (split-K)'),
                                  (2, 25, 42, 2048, 512, 'l4.c1 1x1 2048->512 M=2100 (split-K)')]:
        x = torch.randn(N, H, W, C, device=dev)
        w = torch.randn(Co, 1, 1, C, device=dev) * 0.02
        b = torch.randn(Co, device=dev)
        gf = 2.0 * N * H * W * C * Co / 1e9
        say('64x64 ring kernel, %s' % tag)
        for mode, name in DMA:
            lib.bgs_conv3x3_halo_bfx_tuning(-1, 2 | (mode << 8))
            ms = bench(lambda: BF.conv2d_nhwc(x, w, b, relu=True), iters=20)
            say('  abl %2d  %-55s %7.3f ms  (%6.1f TF-equivalent)' % (mode, name, ms, gf / ms))
        lib.bgs_conv3x3_halo_bfx_tuning(-1, 2)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or '.', exist_ok=True)
        with open(a.out, 'w') as f:
            f.write('\n'.join(OUT) + '\n')


if __name__ == '__main__':
    main()
