#!/usr/bin/env python
"""HIP-event timing of the roofline layer (FPN P2 3x3 conv): python tools/conv_p2_time.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402
dev = 'cuda:0'
x = torch.randn(2, 200, 336, 256, device=dev)
w = torch.randn(256, 3, 3, 256, device=dev) * 0.02
b = torch.randn(256, device=dev)
out = torch.empty(2, 200, 336, 256, device=dev)
for _ in range(5):
    BF.conv2d_nhwc(x, w, b, pad=1, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    BF.conv2d_nhwc(x, w, b, pad=1, out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print('%s: %.3f ms -> %.1f TFLOP/s' % (os.environ.get('TAG', 'conv P2'), ms, 2.0 * 2 * 200 * 336 * 256 * 256 * 9 / ms / 1e9))
