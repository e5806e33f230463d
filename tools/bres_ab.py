"""A/B of the filter-resident 1x1 kernel (csrc/conv1x1_bres.hip) against the 64 x 64 operand ring on
the eligible layers of a cfg[1] forward (interleaved in one process, bit-identity asserted)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import capi, functional as BF
from conv_sweep import L as LAYERS, N as NIMG, bench
lib = capi.load()
dev = 'cuda:0'
tot = {'ring': 0.0, 'bres': 0.0}
for name, H, W, Cin, Cout, R, stride, cnt in LAYERS:
    if R != 1 or Cin not in (64, 128, 256) or Cout % 256:
        continue
    x = torch.randn(NIMG, H, W, Cin, device=dev); w = torch.randn(Cout, 1, 1, Cin, device=dev) * 0.05; b = torch.randn(Cout, device=dev)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = torch.randn(NIMG, Ho, Wo, Cout, device=dev) if 'c3' in name else None
    r = {}
    ys = {}
    for rnd in range(2):
        for key, on in (('ring', 0), ('bres', 2)):
            lib.bgs_conv1x1_bres_enable(on)
            f = lambda: BF.conv2d_nhwc(x, w, b, stride=stride, relu=True, residual=res)
            ys[key] = f()
            assert lib.bgs_conv1x1_bres_last_launch() == (1 if on else 0), (name, key)
            t = bench(f, iters=20)
            r[key] = min(r.get(key, 1e9), t)
    assert torch.equal(ys['ring'], ys['bres']), name
    M = NIMG * Ho * Wo
    gf = 2.0 * M * Cin * Cout / 1e9
    byts = (M * Cin + M * Cout * (2 if res is not None else 1)) * 4 / 1e6
    print('%-10s M %6d K %4d Cout %4d | ring %.4f ms | bres %.4f ms (%.0f TF, %.2f TB/s algorithmic)  x%d' % (
        name, M, Cin, Cout, r['ring'], r['bres'], gf / r['bres'], byts / r['bres'] / 1e3, cnt), flush=True)
    for k in tot:
        tot[k] += r[k] * cnt
lib.bgs_conv1x1_bres_enable(1)
print('total per forward: ring %.3f ms  bres %.3f ms' % (tot['ring'], tot['bres']))
