#!/usr/bin/env python
"""A/B (round 6) of the 3x3 planes kernel (csrc/conv3x3_planes.hip: 8 x 8 pixels x 128 / 256 channels per workgroup, the
whole reduction in the workgroup, one barrier per 32-channel chunk) against the default halo dispatch (K sliced on small
grids + the reduction launch) on every 3x3 / stride-1 layer of one cfg[1] step: equality with the unsliced halo kernel
first, then HIP-event times per layer, interleaved.  python tools/planes3_ab.py  (BGS_BFX_PLANES3_NB=1/2 forces the
channels per workgroup)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import capi, functional as BF  # noqa: E402
from conv_sweep import L as LAYERS, N as NIMG  # noqa: E402
from planes_ab import bench  # noqa: E402


def main():
    dev = 'cuda:0'
    lib = capi.load()
    BF.set_conv_math('bf16x6')
    tot = {0: 0.0, 2: 0.0, 'auto': 0.0}
    print('%-12s %8s %6s %6s | %10s %10s %5s | %s' % ('layer', 'M', 'Cin', 'Cout', 'default us', 'planes us', 'ch/wg', 'x count'))
    for name, H, W, Cin, Cout, R, stride, cnt in LAYERS:
        if R != 3 or stride != 1 or Cin % 32 or Cout % 128:
            continue
        x = torch.randn(NIMG, H, W, Cin, device=dev)
        w = torch.randn(Cout, 3, 3, Cin, device=dev) * 0.02
        b = torch.randn(Cout, device=dev)
        f = lambda: BF.conv2d_nhwc(x, w, b, pad=1, relu=True)   # noqa: E731
        lib.bgs_conv3x3_planes_enable(0)
        BF.conv_bfx_tuning(halo_splits=1)
        y0 = f()
        BF.conv_bfx_tuning()
        lib.bgs_conv3x3_planes_enable(2)
        y2 = f()
        nb = lib.bgs_conv3x3_planes_last_launch()
        if not nb:                 # (the caller did not route this layer to the halo entry point at all: tiny maps)
            print('%-12s %8d %6d %6d | not routed to the 3x3 halo entry point' % (name, NIMG * H * W, Cin, Cout))
            continue
        assert torch.equal(y0, y2), (name, float((y0 - y2).abs().max()))
        lib.bgs_conv3x3_planes_enable(1)
        f()
        taken = lib.bgs_conv3x3_planes_last_launch() != 0
        t = {0: 1e9, 2: 1e9}
        for rep in range(2):
            for mode in ((0, 2) if rep == 0 else (2, 0)):
                lib.bgs_conv3x3_planes_enable(mode)
                t[mode] = min(t[mode], bench(f))
        for mode in t:
            tot[mode] += t[mode] * cnt
        tot['auto'] += (t[2] if taken else t[0]) * cnt
        print('%-12s %8d %6d %6d | %10.1f %10.1f %5d | x%d %s%s' % (name, NIMG * H * W, Cin, Cout, t[0], t[2], 128 * nb, cnt,
                                                                     '<-- planes wins' if t[2] < 0.97 * t[0] else '',
                                                                     '  [automatic mode takes it]' if taken else ''), flush=True)
    lib.bgs_conv3x3_planes_enable(-1)
    print('sum over the 3x3 / stride-1 layers of a step (x count): default %.1f us, planes everywhere %.1f us, automatic rule %.1f us'
          % (tot[0], tot[2], tot['auto']))


if __name__ == '__main__':
    main()
