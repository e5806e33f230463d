"""K = 64 layers of layer1 on the 64 x 64 ring: 3-stage vs 4-stage ring (bgs_conv_bfx_tuning bits 10 / 11)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balancedgroupsoftmax_amd import capi, functional as BF
from respf_ab import bench

dev = 'cuda:0'
layers = [('l1.c3', 64, 256, True), ('l1.ds', 64, 256, False), ('l1.b0.c1', 64, 64, False), ('l1.c1', 256, 64, False),
          ('l2.c3', 128, 512, True)]
capi.load().bgs_conv_bfx_wide_tuning(0, 0, -1)
for name, Cin, Cout, has_res in layers:
    H, W = (200, 336) if name.startswith('l1') else (100, 168)
    x = torch.randn(2, H, W, Cin, device=dev)
    w = torch.randn(Cout, 1, 1, Cin, device=dev) * 0.05
    b = torch.randn(Cout, device=dev)
    res = torch.randn(2, H, W, Cout, device=dev) if has_res else None
    fn = lambda: BF.conv2d_nhwc(x, w, b, relu=True, residual=res, residual_mode=1 if has_res else 0)  # noqa: E731
    out = []
    for rep in range(2):
        for tile in (0x400, 0x800):
            BF.conv_bfx_tuning(tile, -1)
            out.append('%s %.1f' % ('n3' if tile == 0x400 else 'n4', bench(fn)))
    BF.conv_bfx_tuning(0, -1)
    print(name, ' | '.join(out), BF.conv_bfx_last_launch()['ring_stages'], flush=True)
