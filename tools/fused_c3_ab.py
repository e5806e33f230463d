#!/usr/bin/env python
"""A/B (round 6): a frozen ResNet-50 layer1 bottleneck at the BASELINE size (2 x 200 x 336) with conv2 -> conv3 fused into
one launch (bgs_conv3x3_c3_fused_nhwc_f32_bfx) against the two launches; HIP events, interleaved.
python tools/fused_c3_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402

dev = 'cuda:0'
BF.set_conv_math('bf16x6')
N, H, W = 2, 200, 336
t1 = torch.relu(torch.randn(N, H, W, 64, device=dev))
x = torch.relu(torch.randn(N, H, W, 256, device=dev))
w1 = torch.randn(64, 1, 1, 256, device=dev) * 0.05
w2 = torch.randn(64, 3, 3, 64, device=dev) * 0.05
w3 = torch.randn(256, 1, 1, 64, device=dev) * 0.1
b1, b2, b3 = torch.randn(64, device=dev), torch.randn(64, device=dev), torch.randn(256, device=dev)


def timed(f, iters=30):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def two():
    t2 = BF.conv2d_nhwc(t1, w2, b2, pad=1, relu=True)
    return BF.conv2d_nhwc(t2, w3, b3, relu=True, residual=x)


def fused():
    return BF.conv3x3_c3_fused_nhwc(t1, w2, b2, w3, b3, residual=x, relu3=True)


def block(fuse):
    def f():
        o = BF.conv2d_nhwc(x, w1, b1, relu=True)
        return fused_tail(o) if fuse else two_tail(o)
    return f


def two_tail(o):
    t2 = BF.conv2d_nhwc(o, w2, b2, pad=1, relu=True)
    return BF.conv2d_nhwc(t2, w3, b3, relu=True, residual=x)


def fused_tail(o):
    return BF.conv3x3_c3_fused_nhwc(o, w2, b2, w3, b3, residual=x, relu3=True)


assert torch.equal(two(), fused())
for rep in range(3):
    print('conv2 -> conv3 + residual + ReLU (layer1, 2 x 200 x 336): two launches %.1f us | fused %.1f us || whole block '
          '(conv1 + tail): three launches %.1f us | two launches %.1f us'
          % (timed(two), timed(fused), timed(block(False)), timed(block(True))), flush=True)
c2 = timed(lambda: BF.conv2d_nhwc(t1, w2, b2, pad=1, relu=True))
t2 = BF.conv2d_nhwc(t1, w2, b2, pad=1, relu=True)
c3 = timed(lambda: BF.conv2d_nhwc(t2, w3, b3, relu=True, residual=x))
print('alone: conv2 %.1f us, conv3 %.1f us' % (c2, c3))
