#!/usr/bin/env python
"""DMA-ring kernel with 128x128 / 128x64 / 64x128 workgroup tiles (conv_igemm_bfx_ring_kernel) on the
GPU box: bit-identity against the 64x64 ring kernel's results is NOT expected (different K-split /
summation order across tiles is the same, but split-K plans differ), so correctness is checked against
an fp64 reference; then per-layer timing of cfg[1]'s implicit-GEMM layers.

    python tools/conv_ring_check.py [--out gpurun_out/ring.txt]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402
from conv_sweep import L as LAYERS, FC, N as NIMG, bench  # noqa: E402
from conv_bfx_check import ref64, absdot  # noqa: E402

OUT = []
RING = 0x1000


def say(*a):
    s = ' '.join(str(x) for x in a)
    print(s, flush=True)
    OUT.append(s)


def errors():
    torch.manual_seed(0)
    dev = 'cuda:0'
    ok = True
    os.environ['BGS_CONV_HALO'] = '0'
    cases = [
        # name, N, H, W, Cin, Cout, R, stride, pad, relu, res_mode, splitk
        ('1x1', 2, 20, 24, 64, 96, 1, 1, 0, True, 0, 1),
        ('1x1 res odd M', 1, 17, 23, 128, 200, 1, 1, 0, True, 1, 1),
        ('1x1 up2 res', 1, 16, 24, 64, 256, 1, 1, 0, False, 2, 1),
        ('1x1 s2', 2, 31, 45, 256, 128, 1, 2, 0, False, 0, 1),
        ('3x3 s2', 2, 31, 45, 32, 64, 3, 2, 1, True, 0, 1),
        ('3x3 s1 pad', 1, 13, 21, 48, 80, 3, 1, 1, True, 0, 1),
        ('stem 7x7', 1, 64, 96, 4, 64, 7, 2, 3, True, 0, 1),
        ('1x1 splitk4', 2, 13, 21, 512, 128, 1, 1, 0, True, 1, 4),
        ('3x3 splitk5', 1, 13, 21, 256, 256, 3, 1, 1, True, 0, 5),
        ('rpn head 15', 2, 25, 42, 256, 15, 1, 1, 0, False, 0, 1),
        ('fc 1236', 130, 1, 1, 1024, 1236, 1, 1, 0, False, 0, 3),
    ]
    for name, N, H, W, Cin, Cout, R, stride, pad, relu, rm, sk in cases:
        x = torch.randn(N, H, W, Cin, device=dev)
        w = torch.randn(Cout, R, R, Cin, device=dev) / (R * R * Cin) ** 0.5
        b = torch.randn(Cout, device=dev)
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
        res = None
        if rm == 1:
            res = torch.randn(N, Ho, Wo, Cout, device=dev)
        elif rm == 2:
            res = torch.randn(N, Ho // 2, Wo // 2, Cout, device=dev)
        ref = ref64(x.cpu(), w.cpu(), b.cpu(), stride, pad, relu, None if res is None else res.cpu(), rm)
        den = absdot(x.cpu(), w.cpu(), stride, pad).max().item()
        BF.conv_bfx_tuning(11, sk)
        y0 = BF.conv2d_nhwc(x, w, b, stride=stride, pad=pad, relu=relu, residual=res, residual_mode=rm)
        e0 = (y0.cpu().double() - ref).abs().max().item() / den
        line = '%-16s 64x64 %.2e |' % (name, e0)
        for tile in (22, 21, 12):
            BF.conv_bfx_tuning(tile | RING, sk)
            y = BF.conv2d_nhwc(x, w, b, stride=stride, pad=pad, relu=relu, residual=res, residual_mode=rm)
            used = BF.conv_bfx_last_launch()
            e = (y.cpu().double() - ref).abs().max().item() / den
            same = torch.equal(y, y0)
            good = e < max(2 * e0, 2e-7) and (used['tile'] & RING) and (used['tile'] & 0xff) == tile
            ok &= bool(good)
            line += ' t%d %.2e%s%s' % (tile, e, ' =' if same else '', '' if good else ' BAD(%#x)' % used['tile'])
        BF.conv_bfx_tuning()
        say(line)
    os.environ.pop('BGS_CONV_HALO', None)
    say('CORRECT' if ok else 'MISMATCH')
    return ok


def sweep():
    dev = 'cuda:0'
    os.environ['BGS_CONV_HALO'] = '0'
    tot = {'auto': 0.0, 'best': 0.0}
    say('%-12s %8s %6s %5s | auto (64x64 ring / 128x128 reg) | ring 128x128 | 128x64 | 64x128 | best' % ('layer', 'M', 'K', 'Cout'))
    layers = [(n, H, W, Cin, Cout, R, s, c) for (n, H, W, Cin, Cout, R, s, c) in LAYERS]
    layers += [(n, 1, 1, K, Cout, 1, 1, 1) for (n, M, K, Cout) in FC]
    for name, H, W, Cin, Cout, R, stride, cnt in layers:
        if R == 3 and stride == 1 and NIMG * H * W >= 2000:
            continue                      # the halo-kernel layers
        Nn = 1024 if name.startswith('fc') else NIMG
        pad = R // 2
        x = torch.randn(Nn, H, W, Cin, device=dev)
        w = torch.randn(Cout, R, R, Cin, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
        M = Nn * Ho * Wo
        gf = 2.0 * M * R * R * Cin * Cout / 1e9
        BF.conv_bfx_tuning()
        r = {'auto': bench(lambda: BF.conv2d_nhwc(x, w, b, stride=stride, pad=pad, relu=True))}
        for tile in (22, 21, 12):
            if tile == 22 and M * Cout < 128 * 128 * 16:
                continue
            for sk in ((-1,) if M >= 30000 else (-1, 2, 4, 8)):
                if sk > 1 and R * R * Cin // 16 < 4 * sk:
                    continue
                BF.conv_bfx_tuning(tile | RING, sk)
                r['t%d/%d' % (tile, sk)] = bench(lambda: BF.conv2d_nhwc(x, w, b, stride=stride, pad=pad, relu=True))
        BF.conv_bfx_tuning()
        best = min(r, key=r.get)
        tot['auto'] += r['auto'] * cnt
        tot['best'] += r[best] * cnt
        say('%-12s %8d %6d %5d | %6.3f (%5.1f) | %-8s %6.3f (%5.1f)  [%s]  x%d'
            % (name, M, R * R * Cin, Cout, r['auto'], gf / r['auto'], best, r[best], gf / r[best],
               ' '.join('%s:%.3f' % kv for kv in sorted(r.items())), cnt))
    os.environ.pop('BGS_CONV_HALO', None)
    say('igemm layers per forward: auto %.3f ms | best per layer %.3f ms' % (tot['auto'], tot['best']))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--no-sweep', action='store_true')
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    ok = errors()
    if not a.no_sweep:
        sweep()
    if a.out:
        os.makedirs(os.path.dirname(a.out) or '.', exist_ok=True)
        with open(a.out, 'w') as f:
            f.write('\n'.join(OUT) + '\n')
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
