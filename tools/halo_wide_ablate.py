"""Timing-only ablation of the wide-pixel-tile halo kernel (variant 7) on a map that is exactly two rounds of 512 units
(2 x 256 x 256, 256 -> 256) and on the P2 map: where do the cycles beyond the 48 MFMAs per step go?
arms: v4 | v7 | v7 without the second half's fragment reads (1) | without patch restaging (2) | without filter DMA (4) | 6 | 7
(ablated results are wrong by construction; the unablated arms are asserted bit-identical)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import capi, functional as BF
from conv_sweep import bench
lib = capi.load()
dev = 'cuda:0'
BF.set_conv_math('bf16x6')
for (N, H, W, Cin, Cout) in ((2, 256, 256, 256, 256), (2, 200, 336, 256, 256)):
    x = torch.randn(N, H, W, Cin, device=dev); w = torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05; b = torch.randn(Cout, device=dev)
    f = lambda: BF.conv2d_nhwc(x, w, b, pad=1, relu=True)
    gf = 2.0 * N * H * W * 9 * Cin * Cout / 1e9
    res = {}
    for rnd in range(2):
        for name, wide, abl in (('v4', 0, 0), ('v7', 1, 0), ('v7-prio', 1, -2), ('v7-noprio', 1, -3), ('v7-fb32', 1, -1), ('v7-frag2', 1, 1), ('v7-patch', 1, 2), ('v7-dma', 1, 4), ('v7-patch-dma', 1, 6), ('v7-all', 1, 7)):
            if abl == -1 and os.environ.get('BGS_HALO_FB32') != '1':
                continue      # the FB32 arm needs BGS_HALO_FB32=1 at process start (round 6: the mode decides the weight-buffer layout)
            fl = {-2: 2, -3: 8}.get(abl, 0)          # flags: 2 = static priority by wave slot, 8 = no priority games at all
            lib.bgs_conv3x3_halo_bfx_tuning(-1, (max(abl, 0) << 8) | ((wide + 1) << 24) | (fl << 20))
            y = f()
            if name == 'v4':
                y4 = y
            if name in ('v7', 'v7-fb32', 'v7-prio', 'v7-noprio'):
                assert torch.equal(y, y4), name
            res[name] = min(res.get(name, 1e9), bench(f, iters=20))
    u = BF.conv_bfx_last_launch()
    print('%dx%dx%d %d->%d (%d wide + %d tail units): ' % (N, H, W, Cin, Cout, u['halo_wide_units'], u['halo_tail_units']) +
          ' | '.join('%s %.4f ms (%.0f TF)' % (k, v, gf / v) for k, v in res.items()), flush=True)
BF.conv_bfx_tuning()
