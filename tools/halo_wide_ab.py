"""A/B of the halo 3x3 kernel's wide pixel tile (variant 7: 16 x 16 pixels x 128 channels per workgroup, two
workgroups per CU, whole rounds + a variant-4 launch for the left-over rows) against variant 4 (8 x 16 pixels, three
workgroups per CU) on the 3x3 stride-1 layers of a cfg[1] forward with M >= 30000; interleaved, bit-identity asserted.
python tools/halo_wide_ab.py [math=bf16x6|bf16] [iters=20]     arms: 0 = off, 1 = automatic schedule, 2 = all on variant 7"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import functional as BF
from conv_sweep import L as LAYERS, N as NIMG, bench
dev = 'cuda:0'
math = sys.argv[1] if len(sys.argv) > 1 else 'bf16x6'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
BF.set_conv_math(math)
arms = (0, 1, 2)
tot = {a: 0.0 for a in arms}
for name, H, W, Cin, Cout, R, stride, cnt in LAYERS:
    if R != 3 or stride != 1 or NIMG * H * W < 30000 or Cout % 128:
        continue
    x = torch.randn(NIMG, H, W, Cin, device=dev); w = torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05; b = torch.randn(Cout, device=dev)
    r, ys, used = {}, {}, {}
    for rnd in range(3):
        for a in arms:
            BF.conv_bfx_tuning(halo_wide=a)
            f = lambda: BF.conv2d_nhwc(x, w, b, pad=1, relu=True)
            ys[a] = f()
            used[a] = BF.conv_bfx_last_launch()
            r[a] = min(r.get(a, 1e9), bench(f, iters=iters))
    assert torch.equal(ys[0], ys[1]) and torch.equal(ys[0], ys[2]), name
    gf = 2.0 * NIMG * H * W * 9 * Cin * Cout / 1e9
    print('%-10s M %6d Cin %3d Cout %3d | v4 %.4f ms (%.0f TF) | schedule %.4f ms (%.0f TF; %d wide + %d tail units) | all wide %.4f ms (%.0f TF; %d units)  x%d' % (
        name, NIMG * H * W, Cin, Cout, r[0], gf / r[0], r[1], gf / r[1], used[1]['halo_wide_units'], used[1]['halo_tail_units'],
        r[2], gf / r[2], used[2]['halo_wide_units'], cnt), flush=True)
    for k in tot:
        tot[k] += r[k] * cnt
BF.conv_bfx_tuning()
print('total (these layers) per forward: v4 %.3f ms | schedule %.3f ms | all wide %.3f ms' % (tot[0], tot[1], tot[2]))
