#!/usr/bin/env python
"""A/B (round 6) of the planes-in-LDS 1x1 kernel (csrc/conv1x1_planes.hip: 64 pixels x 256 channels per workgroup, A split
once per K chunk into LDS planes) against the default dispatch (wide 128 x 128 / 64 x 64 ring) on every eligible 1x1 layer
of one cfg[1] step (2 x 800 x 1344): bit-equality first, then HIP-event times per layer, interleaved.
python tools/planes_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import capi, functional as BF  # noqa: E402
from conv_sweep import L as LAYERS, N as NIMG  # noqa: E402


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
    return e0.elapsed_time(e1) / iters * 1e3          # us


def main():
    dev = 'cuda:0'
    lib = capi.load()
    BF.set_conv_math('bf16x6')
    tot = {0: 0.0, 2: 0.0}
    print('%-12s %8s %6s %6s %4s | %10s %10s | %s' % ('layer', 'M', 'K', 'Cout', 'res', 'default us', 'planes us', 'x count'))
    for name, H, W, Cin, Cout, R, stride, cnt in LAYERS:
        if R != 1 or stride not in (1, 2) or Cin % 64 or Cout % 128:
            continue
        # residual as in the model: conv3 of a bottleneck adds the block input (mode 1), FPN laterals below the top add the
        # upsampled coarser level (mode 2), the rest none
        rm = 1 if '.c3' in name else (2 if name.startswith('fpn.lat') and not name.endswith('3') else 0)
        x = torch.randn(NIMG, H, W, Cin, device=dev)
        w = torch.randn(Cout, 1, 1, Cin, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        res = None
        if rm == 1:
            res = torch.randn(NIMG, H, W, Cout, device=dev)
        elif rm == 2:
            res = torch.randn(NIMG, H // 2, W // 2, Cout, device=dev)
        relu = rm == 1 or '.c1' in name
        f = lambda: BF.conv2d_nhwc(x, w, b, stride=stride, relu=relu, residual=res, residual_mode=rm)   # noqa: E731
        lib.bgs_conv1x1_planes_enable(0)
        y0 = f()
        assert not lib.bgs_conv1x1_planes_last_launch()
        lib.bgs_conv1x1_planes_enable(2)
        y2 = f()
        assert lib.bgs_conv1x1_planes_last_launch(), name
        same = torch.equal(y0, y2)
        if not same:       # the default plan slices K on this layer (another summation order): compare the unsliced default
            lib.bgs_conv1x1_planes_enable(0)
            BF.conv_bfx_tuning(0, 1)
            y1 = f()
            BF.conv_bfx_tuning(0, -1)
            assert torch.equal(y1, y2), (name, float((y1 - y2).abs().max()))
        t = {0: 1e9, 2: 1e9}
        for rep in range(2):
            for mode in ((0, 2) if rep == 0 else (2, 0)):
                lib.bgs_conv1x1_planes_enable(mode)
                t[mode] = min(t[mode], bench(f))
        for mode in t:
            tot[mode] += t[mode] * cnt
        print('%-12s %8d %6d %6d %4d | %10.1f %10.1f | x%d %s' % (name, NIMG * ((H - 1) // stride + 1) * ((W - 1) // stride + 1), Cin, Cout, rm, t[0], t[2], cnt,
                                                                   ('<-- planes wins' if t[2] < 0.97 * t[0] else '') +
                                                                   ('' if same else '  (default slices K)')), flush=True)
    lib.bgs_conv1x1_planes_enable(-1)
    print('sum over the eligible layers of a step (x count): default %.1f us, planes %.1f us' % (tot[0], tot[2]))


if __name__ == '__main__':
    main()
