#!/usr/bin/env python
"""Launch ONE bf16-storage 1x1 conv layer a few times (target of tools/pmc_bf16s.sh):
    python tools/bf16s_layer_once.py N H W Cin Cout [residual 0|1]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402

a = [int(v, 0) for v in sys.argv[1:]]
N, H, W, Cin, Cout = a[:5]
res = bool(a[5]) if len(a) > 5 else False
dev = 'cuda:0'
x = torch.randn(N, H, W, Cin, device=dev).to(torch.bfloat16)
w = torch.randn(Cout, 1, 1, Cin, device=dev) * 0.03
b = torch.randn(Cout, device=dev)
r = torch.randn(N, H, W, Cout, device=dev).to(torch.bfloat16) if res else None
for _ in range(6):
    y = BF.conv2d_nhwc(x, w, b, relu=True, residual=r)
torch.cuda.synchronize()
print('ok', float(y.float().abs().mean()))
