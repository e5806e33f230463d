#!/usr/bin/env python
"""Round 6: ResNet-50 layer1 inside the real detector (stem output of the bench inputs) with the fused conv2 -> conv3 launch on / off,
side-stream forks on / off, whole layer and one block alone; HIP-event loop of bench.timed_loop.  python tools/fused_c3_layer1_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from balancedgroupsoftmax_amd import functional as BF
dev = torch.device('cuda', 0)
step = bench.DetectorStep(dev, 0, 1, 2, 1)
bb = step.model.backbone
with torch.no_grad():
    x0 = bb._forward_stem(step.img)
layer1 = getattr(bb, bb.res_layers[0])
for forks in ('1', '0'):
    os.environ['BGS_LEVEL_FORK'] = forks
    os.environ['BGS_SHORTCUT_FORK'] = forks
    for rep in range(2):
        for env in ('1', '0'):
            os.environ['BGS_FUSED_C3'] = env
            def f():
                with torch.no_grad():
                    x = x0
                    for blk in layer1:
                        x = blk.run(x, blk.folded())
                    return x
            print('forks=%s BGS_FUSED_C3=%s layer1: %.1f us' % (forks, env, bench.timed_loop(f, 30, 10, 1) * 1e6 / 30), flush=True)
            def g():
                with torch.no_grad():
                    blk = layer1[1]
                    return blk.run(xx, blk.folded())
            xx = f()
            print('      block 1 alone: %.1f us' % (bench.timed_loop(g, 30, 10, 1) * 1e6 / 30), flush=True)
