"""Would a bf16 pyramid pay in the bf16 mode of cfg[4]?  The FPN output / RPN 3x3 convs (256 -> 256) per level with fp32 tensors
(halo kernel, operands rounded in the kernel) against bf16 tensors in and out (csrc/conv_bf16s.hip, implicit GEMM), and the
lateral 1x1 convs with fp32 / bf16 output.   python tools/bf16_pyramid_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import functional as BF
from conv_sweep import bench
dev = 'cuda:0'
BF.set_conv_math('bf16')
tot = {'f32': 0.0, 'bf16': 0.0}
for name, H, W in (('P2', 200, 336), ('P3', 100, 168), ('P4', 50, 84), ('P5', 25, 42), ('P6', 13, 21)):
    x = torch.randn(2, H, W, 256, device=dev); w = torch.randn(256, 3, 3, 256, device=dev) * 0.02; b = torch.randn(256, device=dev)
    xb = x.bfloat16()
    t32 = min(bench(lambda: BF.conv2d_nhwc(x, w, b, pad=1, relu=True), iters=20) for _ in range(3))
    t16 = min(bench(lambda: BF.conv2d_nhwc(xb, w, b, pad=1, relu=True), iters=20) for _ in range(3))
    t16f = min(bench(lambda: BF.conv2d_nhwc(xb, w, b, pad=1, relu=True, out_dtype=torch.float32), iters=20) for _ in range(3))
    print('%s 3x3 256->256: fp32 tensors %.4f ms | bf16 in/out %.4f ms | bf16 in, fp32 out %.4f ms' % (name, t32, t16, t16f), flush=True)
    tot['f32'] += 2 * t32 if name != 'P6' else t32
    tot['bf16'] += 2 * t16 if name != 'P6' else t16
print('FPN out + RPN conv over the levels: fp32 tensors %.3f ms | bf16 tensors %.3f ms' % (tot['f32'], tot['bf16']))
