"""A/B of the cfg[4] bf16-mode trunk layers: fp32 tensors (operands rounded in the kernel: 8-wave ring /
grouped LDS kernel) vs bf16 tensors in HBM (csrc/conv_bf16s.hip).  X101-64x4d shapes at 2 x 800x1344.

    python tools/bf16s_ab.py            # prints a table (us per launch, TFLOP/s)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402

DEV = 'cuda:0'


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
    torch.manual_seed(0)
    BF.set_conv_math('bf16')
    N = 2
    convs = [  # name, H, W, Cin, Cout, residual
        ('l1.0.c1 64->256', 200, 336, 64, 256, False),
        ('l1.c1 256->256', 200, 336, 256, 256, False),
        ('l1.c3 256->256 +res', 200, 336, 256, 256, True),
        ('l2.c1 512->512', 100, 168, 512, 512, False),
        ('l2.c3 512->512 +res', 100, 168, 512, 512, True),
        ('l3.c1 1024->1024', 50, 84, 1024, 1024, False),
        ('l3.c3 1024->1024 +res', 50, 84, 1024, 1024, True),
        ('l4.c1 2048->2048', 25, 42, 2048, 2048, False),
        ('l4.c3 2048->2048 +res', 25, 42, 2048, 2048, True),
        ('fpn.lat0 256->256 f32out', 200, 336, 256, 256, False),
    ]
    from balancedgroupsoftmax_amd import capi
    lib = capi.load()
    print('| layer | fp32 tensors us | bf16 tensors us | speed-up | bf16s TFLOP/s | bf16 tensors, register-staged us |')
    print('|---|---|---|---|---|---|')
    tot = [0.0, 0.0]
    for name, H, W, Cin, Cout, res in convs:
        x = torch.randn(N, H, W, Cin, device=DEV)
        w = torch.randn(Cout, 1, 1, Cin, device=DEV) * 0.05
        b = torch.randn(Cout, device=DEV)
        r = torch.randn(N, H, W, Cout, device=DEV) if res else None
        xb = x.to(torch.bfloat16)
        rb = r.to(torch.bfloat16) if res else None
        o32 = torch.empty(N, H, W, Cout, device=DEV)
        f32out = 'f32out' in name
        ob = torch.empty(N, H, W, Cout, device=DEV, dtype=torch.float32 if f32out else torch.bfloat16)
        t0 = timeit(lambda: BF.conv2d_nhwc(x, w, b, relu=True, residual=r, out=o32))
        t1 = timeit(lambda: BF.conv2d_nhwc(xb, w, b, relu=True, residual=rb, out=ob,
                                           out_dtype=ob.dtype))
        lib.bgs_conv_bf16s_tuning(1)
        t2 = timeit(lambda: BF.conv2d_nhwc(xb, w, b, relu=True, residual=rb, out=ob,
                                           out_dtype=ob.dtype))
        lib.bgs_conv_bf16s_tuning(0)
        fl = 2.0 * N * H * W * Cin * Cout
        tot[0] += t0
        tot[1] += t1
        print('| %s | %.1f | %.1f | %.2f | %.0f | %.1f |' % (name, t0, t1, t0 / t1, fl / t1 * 1e-6, t2))
    grouped = [('l1 C256 cg4', 200, 336, 256), ('l2 C512 cg8', 100, 168, 512),
               ('l3 C1024 cg16', 50, 84, 1024), ('l4 C2048 cg32', 25, 42, 2048)]
    for name, H, W, C in grouped:
        cg = C // 64
        x = torch.randn(N, H, W, C, device=DEV)
        w = torch.randn(C, 3, 3, cg, device=DEV) * 0.1
        b = torch.randn(C, device=DEV)
        xb = x.to(torch.bfloat16)
        t0 = timeit(lambda: BF.grouped_conv3x3_nhwc(x, w, b, 64, relu=True))
        t1 = timeit(lambda: BF.grouped_conv3x3_nhwc(xb, w, b, 64, relu=True))
        fl = 2.0 * N * H * W * C * 9 * cg
        tot[0] += t0
        tot[1] += t1
        print('| grouped 3x3 %s | %.1f | %.1f | %.2f | %.0f |' % (name, t0, t1, t0 / t1, fl / t1 * 1e-6))
    print('| sum (one launch each) | %.1f | %.1f | %.2f | |' % (tot[0], tot[1], tot[0] / tot[1]))


if __name__ == '__main__':
    main()
