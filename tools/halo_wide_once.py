"""Launches the halo 3x3 kernel on the two-round map (2 x 256 x 256, 256 -> 256) a few times for a rocprofv3 pass:
python tools/halo_wide_once.py <wide 0|1|2> <ablation 0|1|2|4|6|7> [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balancedgroupsoftmax_amd import capi, functional as BF
lib = capi.load()
wide, abl = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 8
BF.set_conv_math('bf16x6')
N, H, W, Cin, Cout = 2, 256, 256, 256, 256
x = torch.randn(N, H, W, Cin, device='cuda:0'); w = torch.randn(Cout, 3, 3, Cin, device='cuda:0') * 0.05; b = torch.randn(Cout, device='cuda:0')
lib.bgs_conv3x3_halo_bfx_tuning(-1, (abl << 8) | ((wide + 1) << 24))
for _ in range(iters):
    BF.conv2d_nhwc(x, w, b, pad=1, relu=True)
torch.cuda.synchronize()
print('done', BF.conv_bfx_last_launch())
