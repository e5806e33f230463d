#!/usr/bin/env python
"""Timing-only ablations of the planes-in-LDS 1x1 kernel (csrc/conv1x1_planes.hip, template ABL; results are WRONG in every
arm but 0): where does the launch spend its time beyond the MFMAs?  One child process per arm (the arm is latched from
BGS_BFX_PLANES_ABLATE at the first launch): python tools/planes_ablate.py [N H W Cin Cout [res_mode]]   (default fpn.lat0)
arms: 0 full | 1 filter fragments loaded once | 2 activation tile loaded once | 3 = 1 + 2 | 4 no output stores |
7 = no operand loads, no stores (MFMAs + LDS + skeleton) | 8 no MFMAs | 12 no MFMAs, no stores | 15 skeleton only"""
import os
import subprocess
import sys

ARMS = [0, 1, 2, 3, 4, 7, 8, 12, 15, 0]


def child(shape):
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from balancedgroupsoftmax_amd import capi, functional as BF
    N, H, W, Cin, Cout, rm = shape
    lib = capi.load()
    BF.set_conv_math('bf16x6')
    lib.bgs_conv1x1_planes_enable(2)
    dev = 'cuda:0'
    x = torch.randn(N, H, W, Cin, device=dev)
    w = torch.randn(Cout, 1, 1, Cin, device=dev) * 0.05
    b = torch.randn(Cout, device=dev)
    res = None
    if rm == 1:
        res = torch.randn(N, H, W, Cout, device=dev)
    elif rm == 2:
        res = torch.randn(N, H // 2, W // 2, Cout, device=dev)
    f = lambda: BF.conv2d_nhwc(x, w, b, relu=rm == 1, residual=res, residual_mode=rm)   # noqa: E731
    for _ in range(10):
        f()
    assert lib.bgs_conv1x1_planes_last_launch()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    print('%.1f' % sorted(ts)[len(ts) // 2])


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        child([int(v) for v in sys.argv[2:8]])
        return
    shape = [int(v) for v in sys.argv[1:]] or [2, 200, 336, 256, 256]
    if len(shape) == 5:
        shape.append(0)
    print('shape N H W Cin Cout res_mode =', shape)
    for a in ARMS:
        env = dict(os.environ, BGS_BFX_PLANES_ABLATE=str(a))
        out = subprocess.run([sys.executable, os.path.abspath(__file__), '--child'] + [str(v) for v in shape], env=env,
                             capture_output=True, text=True, timeout=300)
        t = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else 'failed: ' + out.stderr[-300:]
        print('ABL %2d: %s us' % (a, t))


if __name__ == '__main__':
    main()
