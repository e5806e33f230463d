"""A/B (round 6): the stride-2 form of the 3x3 planes kernel against the 64 x 64 operand ring on the three layerN.0.conv2 shapes
of cfg[1].  python tools/planes3_s2_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import capi, functional as BF
from planes_ab import bench
lib = capi.load(); BF.set_conv_math('bf16x6'); dev = 'cuda:0'
for name, (N, H, W, C) in (('l2.b0.c2s2', (2, 200, 336, 128)), ('l3.b0.c2s2', (2, 100, 168, 256)), ('l4.b0.c2s2', (2, 50, 84, 512))):
    x = torch.randn(N, H, W, C, device=dev); w = torch.randn(C, 3, 3, C, device=dev) * 0.02; b = torch.randn(C, device=dev)
    f = lambda: BF.conv2d_nhwc(x, w, b, stride=2, pad=1, relu=True)
    t = {}
    for rep in range(2):
        for mode in (0, 2):
            lib.bgs_conv3x3_planes_enable(mode); f()
            t[mode] = min(t.get(mode, 1e9), bench(f))
    print('%-12s default (operand ring) %7.1f us | planes stride 2 %7.1f us' % (name, t[0], t[2]))
lib.bgs_conv3x3_planes_enable(-1)
