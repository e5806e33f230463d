"""A/B: residual tile of the 64 x 64 ring kernel's epilogue requested ahead of the K loop (default) vs read in the
epilogue (bgs_conv_bfx_tuning tile bit 12).  Bit-equality, then HIP-event times, alternating arms."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balancedgroupsoftmax_amd import capi, functional as BF

dev = 'cuda:0'
os.environ['BGS_CONV_HALO'] = '0'


def bench(fn, iters=30, warm=10):
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
    layers = [  # name, N, H, W, Cin, Cout, res_mode, count per step
        ('l1.c3', 2, 200, 336, 64, 256, 1, 3), ('l2.c3', 2, 100, 168, 128, 512, 1, 4), ('l3.c3', 2, 50, 84, 256, 1024, 1, 6),
        ('l4.c3', 2, 25, 42, 512, 2048, 1, 3), ('fpn.lat0', 2, 200, 336, 256, 256, 2, 1), ('fpn.lat1', 2, 100, 168, 512, 256, 2, 1),
        ('fpn.lat2', 2, 50, 84, 1024, 256, 2, 1), ('odd', 3, 37, 29, 64, 132, 1, 0)]
    capi.load().bgs_conv_bfx_wide_tuning(0, 0, -1)       # the ring kernel on every layer
    tot = [0.0, 0.0]
    for name, N, H, W, Cin, Cout, rm, cnt in layers:
        torch.manual_seed(1)
        x = torch.randn(N, H, W, Cin, device=dev)
        w = torch.randn(Cout, 1, 1, Cin, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        res = torch.randn(N, H, W, Cout, device=dev) if rm == 1 else torch.randn(N, H // 2, W // 2, Cout, device=dev)
        fn = lambda: BF.conv2d_nhwc(x, w, b, stride=1, pad=0, relu=True, residual=res, residual_mode=rm)  # noqa: E731
        BF.conv_bfx_tuning(0x1000, -1)
        y0 = fn().clone()
        BF.conv_bfx_tuning(0, -1)
        y1 = fn().clone()
        eq = torch.equal(y0, y1)
        t = [[], []]
        for rep in range(3):
            for arm, tile in ((0, 0x1000), (1, 0)):
                BF.conv_bfx_tuning(tile, -1)
                t[arm].append(bench(fn))
        BF.conv_bfx_tuning(0, -1)
        a, c = min(t[0]), min(t[1])
        tot[0] += a * cnt
        tot[1] += c * cnt
        print('%-9s x%d | epilogue read %6.1f us | ahead of the K loop %6.1f us | equal %s | launch %s' %
              (name, cnt, a, c, eq, BF.conv_bfx_last_launch()), flush=True)
    print('per step: %.1f -> %.1f us' % tuple(tot))


if __name__ == '__main__':
    main()
