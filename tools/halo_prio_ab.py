"""Halo 3x3 kernel: wave-priority experiments (bgs_conv3x3_halo_bfx_tuning bits 20..23) on the P2 / P3 / layer3 sizes.
flags 8 none | 1 (default) s_setprio 1 around the MFMA cluster | 2 static priority = hardware wave slot | 3 both."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import functional as BF
from conv_sweep import bench
dev = 'cuda:0'
for name, H, W, C in (('P2 256', 200, 336, 256), ('P3 256', 100, 168, 256), ('l3.c2', 50, 84, 256), ('l2.c2 128', 100, 168, 128)):
    x = torch.randn(2, H, W, C, device=dev); w = torch.randn(C, 3, 3, C, device=dev) * 0.05; b = torch.randn(C, device=dev)
    f = lambda: BF.conv2d_nhwc(x, w, b, pad=1, relu=True)
    BF.conv_bfx_tuning()
    y0 = f().clone()
    row = []
    for rnd in range(2):
        for fl in (8, 1, 2, 3):
            BF.conv_bfx_tuning(halo_flags=fl)
            y = f()
            assert torch.equal(y, y0)
            row.append('f%d %.4f' % (fl, bench(f, iters=20)))
    BF.conv_bfx_tuning()
    print(name, ' | '.join(row), flush=True)
