#!/usr/bin/env python
"""Split-form activations (bf16 hi / mid / lo planes) through the bf16x6 conv kernels, on the GPU box:

  1. bit-identity: conv on ``split_act(x)`` == conv on ``x``; the epilogue's split-form output ==
     ``split_act(y)``; the three planes sum to ``y`` exactly;
  2. per-layer timing of cfg[1]'s implicit-GEMM layers: fp32 input (A re-split inside the K loop) vs
     split-form input (3- and 4-stage rings), with and without the split-form output.

    python tools/conv_split_check.py [--out gpurun_out/split_sweep.txt]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402
from conv_sweep import L as LAYERS, FC, N as NIMG, bench  # noqa: E402

OUT = []


def say(*a):
    s = ' '.join(str(x) for x in a)
    print(s, flush=True)
    OUT.append(s)


def identity():
    torch.manual_seed(0)
    dev = 'cuda:0'
    ok = True
    os.environ['BGS_CONV_HALO'] = '0'
    cases = [
        # name, N, H, W, Cin, Cout, R, stride, pad, relu, res_mode, splitk, nst3
        ('1x1', 2, 20, 24, 64, 96, 1, 1, 0, True, 0, -1, 0),
        ('1x1 nst3', 2, 20, 24, 64, 96, 1, 1, 0, True, 0, -1, 1),
        ('1x1 res odd M', 1, 17, 23, 128, 200, 1, 1, 0, True, 1, -1, 0),
        ('1x1 up2 res', 1, 16, 24, 64, 256, 1, 1, 0, False, 2, 1, 0),
        ('1x1 s2', 2, 31, 45, 256, 128, 1, 2, 0, False, 0, 1, 0),
        ('3x3 s2', 2, 31, 45, 32, 64, 3, 2, 1, True, 0, 1, 1),
        ('3x3 s1 pad', 1, 13, 21, 48, 80, 3, 1, 1, True, 0, 1, 0),
        ('1x1 splitk4', 2, 13, 21, 512, 128, 1, 1, 0, True, 1, 4, 0),
        ('3x3 splitk5', 1, 13, 21, 256, 256, 3, 1, 1, True, 0, 5, 1),
        ('rpn head 15', 2, 25, 42, 256, 15, 1, 1, 0, False, 0, -1, 0),
        ('fc 1024', 130, 1, 1, 1024, 1236, 1, 1, 0, False, 0, -1, 0),
        ('K=16 tail', 1, 9, 11, 16, 64, 1, 1, 0, False, 0, 1, 0),
    ]
    for name, N, H, W, Cin, Cout, R, stride, pad, relu, rm, sk, nst3 in cases:
        x = torch.randn(N, H, W, Cin, device=dev) * torch.exp(torch.randn(N, H, W, Cin, device=dev))
        w = torch.randn(Cout, R, R, Cin, device=dev) / (R * R * Cin) ** 0.5
        b = torch.randn(Cout, device=dev)
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
        res = None
        if rm == 1:
            res = torch.randn(N, Ho, Wo, Cout, device=dev)
        elif rm == 2:
            res = torch.randn(N, Ho // 2, Wo // 2, Cout, device=dev)
        BF.conv_bfx_tuning(11, sk)
        y0 = BF.conv2d_nhwc(x, w, b, stride=stride, pad=pad, relu=relu, residual=res, residual_mode=rm)
        xp = BF.split_act(x)
        BF.conv_bfx_tuning(11 | (0x400 if nst3 else 0), sk)
        y1, yp1 = BF.conv2d_nhwc_split(None, xp, w, b, stride=stride, pad=pad, relu=relu, residual=res,
                                       residual_mode=rm, want_f32=True, want_planes=True)
        used = BF.conv_bfx_last_launch()
        _, yp2 = BF.conv2d_nhwc_split(x, None, w, b, stride=stride, pad=pad, relu=relu, residual=res,
                                      residual_mode=rm, want_f32=False, want_planes=True)
        BF.conv_bfx_tuning()
        ref_p = BF.split_act(y0) if Cout % 4 == 0 else None
        same_y = torch.equal(y0, y1)
        same_p = ref_p is None or (torch.equal(ref_p.view(torch.int16), yp1.view(torch.int16)) and
                                   torch.equal(ref_p.view(torch.int16), yp2.view(torch.int16)))
        s3 = (yp1[0].float() + yp1[1].float()) + yp1[2].float()
        exact = torch.equal(s3, y0)
        sum_x = torch.equal((xp[0].float() + xp[1].float()) + xp[2].float(), x)
        good = same_y and same_p and exact and sum_x and (used['tile'] & 0x800)
        ok &= bool(good)
        say('%-16s tile %#x splits %d | y identical %s  planes identical %s  planes sum == y %s  x planes exact %s  %s'
            % (name, used['tile'], used['splits'], same_y, same_p, exact, sum_x, 'ok' if good else 'BAD'))
    os.environ.pop('BGS_CONV_HALO', None)
    say('IDENTICAL' if ok else 'MISMATCH')
    return ok


def sweep():
    dev = 'cuda:0'
    os.environ['BGS_CONV_HALO'] = '0'
    tot = dict(f32in=0.0, pl4=0.0, pl3=0.0, best=0.0, best_po=0.0)
    say('%-12s %8s %6s %5s | fp32-in auto | planes-in NST4 / NST3 | + planes out (no f32) NST4 / NST3 | split_act of the input'
        % ('layer', 'M', 'K', 'Cout'))
    layers = [(n, H, W, Cin, Cout, R, s, c) for (n, H, W, Cin, Cout, R, s, c) in LAYERS]
    layers += [(n, 1, 1, K, Cout, 1, 1, 1) for (n, M, K, Cout) in FC]
    for name, H, W, Cin, Cout, R, stride, cnt in layers:
        if Cin % 16 or (R == 3 and stride == 1 and 2 * H * W >= 2000):
            continue                      # stem (Cin = 4) and the halo-kernel layers
        Nn = 1024 if name.startswith('fc') else NIMG
        pad = R // 2
        x = torch.randn(Nn, H, W, Cin, device=dev)
        w = torch.randn(Cout, R, R, Cin, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
        M = Nn * Ho * Wo
        gf = 2.0 * M * R * R * Cin * Cout / 1e9
        xp = BF.split_act(x)
        BF.conv_bfx_tuning()
        t0 = bench(lambda: BF.conv2d_nhwc(x, w, b, stride=stride, pad=pad, relu=True))
        r = {}
        for nst, key in ((4, 'pl4'), (3, 'pl3')):
            BF.conv_bfx_tuning(11 | (0x400 if nst == 3 else 0), -1)
            r[key] = bench(lambda: BF.conv2d_nhwc_split(None, xp, w, b, stride=stride, pad=pad, relu=True))
            r[key + 'po'] = bench(lambda: BF.conv2d_nhwc_split(None, xp, w, b, stride=stride, pad=pad, relu=True,
                                                               want_f32=False, want_planes=Cout % 4 == 0 or True))
        BF.conv_bfx_tuning()
        ts = bench(lambda: BF.split_act(x))
        tot['f32in'] += t0 * cnt
        tot['pl4'] += r['pl4'] * cnt
        tot['pl3'] += r['pl3'] * cnt
        tot['best'] += min(t0, r['pl4'], r['pl3']) * cnt
        tot['best_po'] += min(r['pl4po'], r['pl3po']) * cnt
        say('%-12s %8d %6d %5d | %6.3f (%5.1f) | %6.3f (%5.1f) / %6.3f (%5.1f) | %6.3f / %6.3f | %6.3f   x%d'
            % (name, M, R * R * Cin, Cout, t0, gf / t0, r['pl4'], gf / r['pl4'], r['pl3'], gf / r['pl3'],
               r['pl4po'], r['pl3po'], ts, cnt))
    os.environ.pop('BGS_CONV_HALO', None)
    say('igemm layers per forward: fp32-in %.3f ms | planes-in NST4 %.3f  NST3 %.3f | best of three %.3f | '
        'planes in + planes-only out (best NST) %.3f' % (tot['f32in'], tot['pl4'], tot['pl3'], tot['best'], tot['best_po']))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--no-sweep', action='store_true')
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    ok = identity()
    if not a.no_sweep:
        sweep()
    if a.out:
        os.makedirs(os.path.dirname(a.out) or '.', exist_ok=True)
        with open(a.out, 'w') as f:
            f.write('\n'.join(OUT) + '\n')
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
