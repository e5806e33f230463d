#!/usr/bin/env python
"""Launches the roofline layer of bench.py (FPN P2 output conv: 2x200x336, 3x3, 256->256) a few
times — the target of tools/pmc_conv.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)."""
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
for _ in range(6):
    BF.conv2d_nhwc(x, w, b, pad=1, out=out)
torch.cuda.synchronize()
print('ok')
