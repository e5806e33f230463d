"""Timing-only: the halo kernel's P2-level step with 1 of 3 / none of the filter-plane DMAs (results are wrong).
    python -m balancedgroupsoftmax_amd.csrc.build --variant ablate; BGS_LIB_VARIANT=ablate python tools/halo_dma_ablate.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import functional as BF
from conv_sweep import bench
dev = 'cuda:0'
for name, H, W, C in (('P2 256', 200, 336, 256), ('P3 256', 100, 168, 256)):
    x = torch.randn(2, H, W, C, device=dev); w = torch.randn(C, 3, 3, C, device=dev) * 0.05; b = torch.randn(C, device=dev)
    f = lambda: BF.conv2d_nhwc(x, w, b, pad=1, relu=True)
    row = []
    for rnd in range(2):
        for fl, nm in ((0, 'all three planes'), (4, 'one plane'), (8, 'no filter DMA')):
            BF.conv_bfx_tuning(halo_flags=fl)
            row.append('%s %.4f ms' % (nm, bench(f, iters=20)))
    BF.conv_bfx_tuning()
    print(name, ' | '.join(row), flush=True)
