"""Launches the fused stem kernel a few times at the cfg[1] size for a rocprofv3 pass: python tools/stem_once.py [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balancedgroupsoftmax_amd import functional as BF
BF.set_conv_math('bf16x6')
img = torch.randn(2, 3, 800, 1344, device='cuda:0')
w = torch.randn(64, 7, 7, 4, device='cuda:0') * 0.1; b = torch.randn(64, device='cuda:0')
ws = BF.stem_fused_split_weights(w)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    BF.stem_fused(img, ws, b)
torch.cuda.synchronize()
print('done')
