"""The fused stem kernel (conv 7x7 / s2 + ReLU + 3x3 / s2 max-pool from the NCHW image, csrc/stem_fused.hip) against the
three-launch chain it replaces, at the cfg[1] size 2 x 3 x 800 x 1344."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import functional as BF
from conv_sweep import bench
dev = 'cuda:0'
BF.set_conv_math('bf16x6')
img = torch.randn(2, 3, 800, 1344, device=dev)
w = torch.randn(64, 7, 7, 4, device=dev) * 0.1; w[..., 3] = 0; b = torch.randn(64, device=dev)
ws = BF.stem_fused_split_weights(w)
chain = lambda: BF.maxpool3x3s2_nhwc(BF.conv2d_nhwc(BF.nchw_to_nhwc4(img), w, b, stride=2, pad=3, relu=True))
fused = lambda: BF.stem_fused(img, ws, b)
a, c = fused(), chain()
print('max |fused - chain| / scale = %.2e' % (float((a - c).abs().max()) / float(c.abs().max())))
for rnd in range(2):
    tc, tf = bench(chain, iters=30), bench(fused, iters=30)
    gf = 2.0 * 2 * 400 * 672 * 147 * 64 / 1e9
    print('chain (3 launches) %.1f us | fused %.1f us (%.0f TF of useful conv flops)' % (tc * 1e3, tf * 1e3, gf / tf), flush=True)
