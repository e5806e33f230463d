#!/usr/bin/env python
"""bf16x6 convolution kernels (csrc/conv_bfx.hip) on the GPU box:

  1. error of the fp32 MFMA kernel and of the bf16x6 kernels against an fp64 torch-CPU reference on
     a set of shapes that covers every epilogue / tile / stride / dgrad mode;
  2. per-layer timing of cfg[1]'s forward shapes (2 x 800x1344): fp32 MFMA kernel vs bf16x6
     (tile x bk sweep, halo variant for the 3x3 stride-1 layers).

    python tools/conv_bfx_check.py [--quick] [--out gpurun_out/bfx_sweep.txt]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402
from conv_sweep import L as LAYERS, FC, N as NIMG, bench  # noqa: E402

OUT = []


def say(*a):
    s = ' '.join(str(x) for x in a)
    print(s, flush=True)
    OUT.append(s)


def ref64(x, w, b, stride, pad, relu, res, res_mode):
    y = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2),
                 None if b is None else b.double(), stride=stride, padding=pad)
    if res is not None:
        r = res.double().permute(0, 3, 1, 2)
        if res_mode == 2:
            r = F.interpolate(r, scale_factor=2, mode='nearest')
        y = y + r
    if relu:
        y = y.clamp(min=0)
    return y.permute(0, 2, 3, 1).contiguous()


def absdot(x, w, stride, pad):
    return F.conv2d(x.double().abs().permute(0, 3, 1, 2), w.double().abs().permute(0, 3, 1, 2),
                    stride=stride, padding=pad).permute(0, 2, 3, 1)


def run(mode, fn):
    prev = BF.set_conv_math(mode)
    try:
        return fn()
    finally:
        BF.set_conv_math(prev)


def errors():
    torch.manual_seed(0)
    dev = 'cuda:0'
    ok = True
    cases = [
        # N, H, W, Cin, Cout, R, stride, pad, relu, res_mode, tuning(tile,bk,splitk)
        ('1x1 64x64 tile', 2, 20, 24, 64, 96, 1, 1, 0, True, 0, (11, -1)),
        ('1x1 64x64 bk32', 2, 20, 24, 64, 96, 1, 1, 0, True, 0, (11, -1)),
        ('1x1 res', 1, 17, 23, 128, 200, 1, 1, 0, True, 1, (0, -1)),
        ('1x1 up2 res', 1, 16, 24, 64, 256, 1, 1, 0, False, 2, (22, 1)),
        ('3x3 s2 128x64', 2, 31, 45, 32, 64, 3, 2, 1, True, 0, (21, 1)),
        ('3x3 s2 64x128', 2, 31, 45, 32, 160, 3, 2, 1, True, 0, (12, 1)),
        ('3x3 128x128 bk32', 1, 33, 47, 64, 130, 3, 1, 1, False, 0, (22, 1)),
        ('stem 7x7', 1, 64, 96, 4, 64, 7, 2, 3, True, 0, (21, 1)),
        ('stem 7x7 bk32', 1, 64, 96, 4, 64, 7, 2, 3, True, 0, (21, 1)),
        ('1x1 splitk4', 2, 13, 21, 512, 128, 1, 1, 0, True, 1, (11, 4)),
        ('1x1 splitk3 bk32', 2, 13, 21, 512, 128, 1, 1, 0, True, 1, (11, 3)),
        ('3x3 splitk', 1, 13, 21, 256, 256, 3, 1, 1, True, 0, (11, 5)),
        ('rpn head 15', 2, 25, 42, 256, 15, 1, 1, 0, False, 0, (0, -1)),
        ('fc 1024x12544', 64, 1, 1, 12544, 1024, 1, 1, 0, True, 0, (0, -1)),
        ('fc_cls 1236', 130, 1, 1, 1024, 1236, 1, 1, 0, False, 0, (0, -1)),
    ]
    os.environ['BGS_CONV_HALO'] = '0'
    for name, N, H, W, Cin, Cout, R, stride, pad, relu, rm, tune in cases:
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
        y32 = run('f32', lambda: BF.conv2d_nhwc(x, w, b, stride=stride, pad=pad, relu=relu,
                                                residual=res, residual_mode=rm))
        BF.conv_bfx_tuning(*tune)
        y6 = run('bf16x6', lambda: BF.conv2d_nhwc(x, w, b, stride=stride, pad=pad, relu=relu,
                                                  residual=res, residual_mode=rm))
        used = BF.conv_bfx_last_launch()
        BF.conv_bfx_tuning()
        e32 = (y32.cpu().double() - ref).abs().max().item() / den
        e6 = (y6.cpu().double() - ref).abs().max().item() / den
        good = e6 < max(2 * e32, 2e-7)
        ok &= good
        say('%-20s tile %#x splits %d | err/sum|ab|: f32 %.2e  bf16x6 %.2e  %s'
            % (name, used['tile'], used['splits'], e32, e6, 'ok' if good else 'BAD'))
    # halo kernel
    os.environ['BGS_CONV_HALO'] = '1'
    for (N, H, W, Cin, Cout, relu, hs) in [(1, 8, 16, 16, 128, False, -1), (2, 13, 21, 64, 256, True, -1),
                                           (1, 25, 42, 256, 200, True, 1), (2, 50, 84, 32, 64, False, 1),
                                           (1, 3, 5, 48, 15, True, 3), (1, 19, 37, 128, 64, True, 4)]:
        x = torch.randn(N, H, W, Cin, device=dev)
        w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
        b = torch.randn(Cout, device=dev)
        ref = ref64(x.cpu(), w.cpu(), b.cpu(), 1, 1, relu, None, 0)
        den = absdot(x.cpu(), w.cpu(), 1, 1).max().item()
        y32 = run('f32', lambda: BF.conv2d_nhwc(x, w, b, stride=1, pad=1, relu=relu))
        BF.conv_bfx_tuning(halo_splits=hs)
        y6 = run('bf16x6', lambda: BF.conv2d_nhwc(x, w, b, stride=1, pad=1, relu=relu))
        used = BF.conv_bfx_last_launch()
        BF.conv_bfx_tuning()
        e32 = (y32.cpu().double() - ref).abs().max().item() / den
        e6 = (y6.cpu().double() - ref).abs().max().item() / den
        good = e6 < max(2 * e32, 2e-7)
        ok &= good
        say('halo N%d %dx%d %d->%d nb %d splits %d | f32 %.2e  bf16x6 %.2e  %s'
            % (N, H, W, Cin, Cout, used['halo_nb'], used['halo_splits'], e32, e6, 'ok' if good else 'BAD'))
    os.environ.pop('BGS_CONV_HALO', None)
    # dgrad (stride 1 with residual + mask, stride 2)
    for (N, H, W, Cin, Cout, R, stride, pad) in [(1, 20, 28, 64, 128, 3, 1, 1), (2, 21, 29, 32, 64, 3, 2, 1),
                                                 (1, 16, 24, 256, 64, 1, 1, 0), (1, 16, 24, 128, 256, 1, 2, 0)]:
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
        dy = torch.randn(N, Ho, Wo, Cout, device=dev)
        w = torch.randn(Cout, R, R, Cin, device=dev) / (R * R * Cout) ** 0.5
        mask = torch.randn(N, H, W, Cin, device=dev)
        res = torch.randn(N, H, W, Cin, device=dev)
        xr = torch.zeros(N, Cin, H, W, dtype=torch.float64, requires_grad=True)
        yr = F.conv2d(xr, w.cpu().double().permute(0, 3, 1, 2), stride=stride, padding=pad)
        (gx,) = torch.autograd.grad(yr, xr, dy.cpu().double().permute(0, 3, 1, 2))
        ref = (gx.permute(0, 2, 3, 1) + res.cpu().double()) * (mask.cpu() > 0)
        d32 = run('f32', lambda: BF.conv2d_dgrad_nhwc(dy, w, (H, W), stride=stride, pad=pad,
                                                      residual=res, mask=mask))
        d6 = run('bf16x6', lambda: BF.conv2d_dgrad_nhwc(dy, w, (H, W), stride=stride, pad=pad,
                                                        residual=res, mask=mask))
        sc = ref.abs().max().item()
        e32 = (d32.cpu().double() - ref).abs().max().item() / sc
        e6 = (d6.cpu().double() - ref).abs().max().item() / sc
        good = e6 < max(2 * e32, 5e-7)
        ok &= good
        say('dgrad %dx%d %d<-%d k%d s%d | rel: f32 %.2e  bf16x6 %.2e  %s'
            % (H, W, Cin, Cout, R, stride, e32, e6, 'ok' if good else 'BAD'))
    say('CORRECT' if ok else 'MISMATCH')
    return ok


def sweep(quick):
    dev = 'cuda:0'
    tot32 = totb = 0.0
    say('%-12s %8s %6s %5s | %-12s | %s' % ('layer', 'M', 'K', 'Cout', 'f32 ms(TF)', 'bf16x6: best cfg ms (TF)  [all]'))
    layers = [(n, H, W, Cin, Cout, R, s, c) for (n, H, W, Cin, Cout, R, s, c) in LAYERS]
    layers += [(n, 1, 1, K, Cout, 1, 1, 1) for (n, M, K, Cout) in FC]
    for name, H, W, Cin, Cout, R, stride, cnt in layers:
        Nn = 1024 if name.startswith('fc') else NIMG
        pad = R // 2
        x = torch.randn(Nn, H, W, Cin, device=dev)
        w = torch.randn(Cout, R, R, Cin, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
        M = Nn * Ho * Wo
        gf = 2.0 * M * R * R * Cin * Cout / 1e9
        os.environ.pop('BGS_CONV_HALO', None)
        t32 = run('f32', lambda: bench(lambda: BF.conv2d_nhwc(x, w, b, stride=stride, pad=pad, relu=True)))
        res = {}
        os.environ['BGS_CONV_HALO'] = '0'
        cfgs = [(0, -1)] if quick else [(0, -1), (11, -1), (11 | 0x100, -1), (12, -1), (22, -1)]
        for tile, sk in cfgs:
            if tile == 22 and M * Cout < 128 * 128 * 64:
                continue
            if sk > 1 and R * R * Cin // 16 < 2 * sk:
                continue
            BF.conv_bfx_tuning(tile, sk)
            res['t%x/%d' % (tile, sk)] = run('bf16x6', lambda: bench(
                lambda: BF.conv2d_nhwc(x, w, b, stride=stride, pad=pad, relu=True)))
        BF.conv_bfx_tuning()
        if R == 3 and stride == 1 and Cin % 16 == 0:
            os.environ['BGS_CONV_HALO'] = '1'
            for hv, hs in ([(2, -1)] if quick else [(1, -1), (2, -1), (2, 1), (2, 2), (2, 4), (2, 8)]):
                BF.conv_bfx_tuning(halo_splits=hs, halo_variant=hv)
                res['halo%d/%d' % (hv, hs)] = run('bf16x6', lambda: bench(
                    lambda: BF.conv2d_nhwc(x, w, b, stride=1, pad=1, relu=True)))
            BF.conv_bfx_tuning()
        os.environ.pop('BGS_CONV_HALO', None)
        best = min(res, key=res.get)
        tot32 += t32 * cnt
        totb += res[best] * cnt
        say('%-12s %8d %6d %5d | %6.3f (%5.1f) | %-9s %6.3f (%5.1f)  [%s]'
            % (name, M, R * R * Cin, Cout, t32, gf / t32, best, res[best], gf / res[best],
               ' '.join('%s:%.3f' % kv for kv in sorted(res.items()))))
    say('forward conv+fc total: f32 %.3f ms   bf16x6 (best per layer) %.3f ms' % (tot32, totb))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--quick', action='store_true')
    ap.add_argument('--no-sweep', action='store_true')
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    ok = errors()
    if not a.no_sweep:
        sweep(a.quick)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or '.', exist_ok=True)
        with open(a.out, 'w') as f:
            f.write('\n'.join(OUT) + '\n')
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
