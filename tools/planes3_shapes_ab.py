"""A/B (round 6) of the automatic rule of the 3x3 planes kernel on shapes OUTSIDE the cfg[1] trunk: the 14 x 14 RoI maps of the
Mask R-CNN head (N = RoIs), single images, larger batches.  python tools/planes3_shapes_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import capi, functional as BF
from planes_ab import bench
lib = capi.load(); BF.set_conv_math('bf16x6'); dev = 'cuda:0'
for (N, H, W, C) in ((256, 14, 14, 256), (64, 14, 14, 256), (512, 14, 14, 256), (2, 25, 42, 256), (8, 50, 84, 256), (1, 200, 336, 256), (1, 100, 168, 256)):
    x = torch.randn(N, H, W, C, device=dev); w = torch.randn(C, 3, 3, C, device=dev) * 0.02; b = torch.randn(C, device=dev)
    f = lambda: BF.conv2d_nhwc(x, w, b, pad=1, relu=True)
    t = {}
    for rep in range(2):
        for mode in (0, 1):
            lib.bgs_conv3x3_planes_enable(mode); f()
            took = lib.bgs_conv3x3_planes_last_launch()
            t[mode] = min(t.get(mode, 1e9), bench(f))
            if mode: tk = took
    print('%4d x %3d x %3d x %d: default %7.1f us | automatic %7.1f us (planes kernel took it: %d)' % (N, H, W, C, t[0], t[1], tk))
lib.bgs_conv3x3_planes_enable(-1)
