#!/usr/bin/env python
"""Per-step time of the halo 3x3 kernel on an UNDER-FILLED grid (layer3 conv2: 2 x 50 x 84, 256 -> 256; 140 workgroups of
128 pixels x 128 channels before K is sliced): wall time of the launch (+ the split-K epilogue) for 1 / 2 / 4 / 8 slices
over the channel chunks and both pixel tiles, and the time per k step of one workgroup (144 steps / slices).  A step is 24
MFMAs per wave = 768 matrix-pipe cycles: a per-step time well above ~0.4 us x (waves per SIMD) is latency, not pipe.
python tools/halo_small_map_probe.py [N H W C]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402


def bench(fn, iters=20, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    N, H, W, C = [int(v) for v in sys.argv[1:5]] if len(sys.argv) > 4 else (2, 50, 84, 256)
    dev = 'cuda:0'
    BF.set_conv_math('bf16x6')
    x = torch.randn(N, H, W, C, device=dev)
    w = torch.randn(C, 3, 3, C, device=dev) * 0.02
    b = torch.randn(C, device=dev)
    f = lambda: BF.conv2d_nhwc(x, w, b, pad=1, relu=True)   # noqa: E731
    print('layer %d x %d x %d, %d -> %d, 3x3' % (N, H, W, C, C))
    for geom in (0, 1):
        for splits in (-1, 1, 2, 4, 8):
            BF.conv_bfx_tuning(halo_splits=splits, halo_geom=geom)
            t = min(bench(f), bench(f))
            last = BF.conv_bfx_last_launch()
            s = max(1, last['halo_splits'])
            steps = 9 * (C // 16) / s
            print('geom %d (%s) splits %2d%s: %6.1f us   -> %5.2f us per k step of a workgroup (%d steps)' % (
                geom, '8x16' if geom == 0 else '10x12', s, ' (auto)' if splits < 0 else '', t, t / steps, steps))
    BF.conv_bfx_tuning()


if __name__ == '__main__':
    main()
