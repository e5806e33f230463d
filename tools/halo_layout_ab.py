"""A/B of the halo 3x3 kernel's patch layout: variant 6 (48-byte padded patch rows: 2-way bank conflicts in every
lane group of the A-fragment reads) against the default (32-byte pixels, k halves swapped on odd row + column) on the
3x3 stride-1 layers of a cfg[1] forward; interleaved, bit-identity asserted.  Variant 6 exists for the NB = 2 /
8 x 16 instantiation only (the other layers run the same kernel in both arms).
python tools/halo_layout_ab.py A B compares variants A and B (4 default, 6 padded rows, 7 one filter buffer)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import functional as BF
from conv_sweep import L as LAYERS, N as NIMG, bench
dev = 'cuda:0'
import sys as _s
A, B = (int(_s.argv[1]), int(_s.argv[2])) if len(_s.argv) > 2 else (6, 4)
tot = {A: 0.0, B: 0.0}
for name, H, W, Cin, Cout, R, stride, cnt in LAYERS:
    if R != 3 or stride != 1 or NIMG * H * W < 2000:
        continue
    x = torch.randn(NIMG, H, W, Cin, device=dev); w = torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05; b = torch.randn(Cout, device=dev)
    r, ys = {}, {}
    for rnd in range(3):
        for v in (A, B):
            BF.conv_bfx_tuning(0, -1, -1, v)
            f = lambda: BF.conv2d_nhwc(x, w, b, pad=1, relu=True)
            ys[v] = f()
            r[v] = min(r.get(v, 1e9), bench(f, iters=20))
    assert torch.equal(ys[A], ys[B]), name
    gf = 2.0 * NIMG * H * W * 9 * Cin * Cout / 1e9
    print('%-10s M %6d Cin %3d Cout %3d | variant %d %.4f ms (%.0f TF) | variant %d %.4f ms (%.0f TF)  x%d' % (
        name, NIMG * H * W, Cin, Cout, A, r[A], gf / r[A], B, r[B], gf / r[B], cnt), flush=True)
    for k in tot:
        tot[k] += r[k] * cnt
BF.conv_bfx_tuning()
print('total 3x3 per forward: variant %d %.3f ms  variant %d %.3f ms' % (A, tot[A], B, tot[B]))
