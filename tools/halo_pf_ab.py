"""A/B of the halo 3x3 kernel variants 4 (default) and 5 (A fragments of tap t+1 prefetched across
the step barrier) on the 3x3 stride-1 layers of a cfg[1] forward; interleaved, bit-identity asserted."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import functional as BF
from conv_sweep import L as LAYERS, N as NIMG, bench
dev = 'cuda:0'
tot = {4: 0.0, 5: 0.0}
for name, H, W, Cin, Cout, R, stride, cnt in LAYERS:
    if R != 3 or stride != 1 or NIMG * H * W < 2000:
        continue
    x = torch.randn(NIMG, H, W, Cin, device=dev); w = torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05; b = torch.randn(Cout, device=dev)
    r, ys = {}, {}
    for rnd in range(3):
        for v in (4, 5):
            BF.conv_bfx_tuning(0, -1, -1, v)
            f = lambda: BF.conv2d_nhwc(x, w, b, pad=1, relu=True)
            ys[v] = f()
            r[v] = min(r.get(v, 1e9), bench(f, iters=20))
    assert torch.equal(ys[4], ys[5]), name
    gf = 2.0 * NIMG * H * W * 9 * Cin * Cout / 1e9
    print('%-10s M %6d Cin %3d Cout %3d | v4 %.4f ms (%.0f TF) | v5 %.4f ms (%.0f TF)  x%d' % (
        name, NIMG * H * W, Cin, Cout, r[4], gf / r[4], r[5], gf / r[5], cnt), flush=True)
    for k in tot:
        tot[k] += r[k] * cnt
BF.conv_bfx_tuning()
print('total 3x3 per forward: variant 4 %.3f ms  variant 5 %.3f ms' % (tot[4], tot[5]))
