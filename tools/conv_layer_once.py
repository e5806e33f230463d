#!/usr/bin/env python
"""Launch ONE conv layer a few times (target of tools/pmc_layer.sh):
    python tools/conv_layer_once.py N H W Cin Cout R stride [tile] [splitk] [halo 0|1]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402

a = [int(v, 0) for v in sys.argv[1:]]
N, H, W, Cin, Cout, R, stride = a[:7]
tile = a[7] if len(a) > 7 else 0
splitk = a[8] if len(a) > 8 else -1
os.environ['BGS_CONV_HALO'] = str(a[9]) if len(a) > 9 else '0'
dev = 'cuda:0'
x = torch.randn(N, H, W, Cin, device=dev)
w = torch.randn(Cout, R, R, Cin, device=dev) * 0.02
b = torch.randn(Cout, device=dev)
BF.conv_bfx_tuning(tile, splitk)
for _ in range(6):
    y = BF.conv2d_nhwc(x, w, b, stride=stride, pad=R // 2, relu=True)
torch.cuda.synchronize()
print('ok', BF.conv_bfx_last_launch())
