#!/usr/bin/env python
"""A/B of the wide-tile 1x1 kernel (csrc/conv_bfx_wide.hip: 128 x 128, four M-stacked waves) against the
64 x 64 operand ring on every 1x1 / linear layer of one cfg[1] step (2 x 800x1344), in ONE process:
bit-equality of the unsliced arms first, then HIP-event times per layer and arm.

    python tools/wide_ab.py [--quick]        (GPU box)
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import capi, functional as BF  # noqa: E402
from conv_sweep import L as LAYERS, FC, N as NIMG  # noqa: E402


def bench(fn, iters=20, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3          # us


def wide(mode, nst=0, splitk=-1, flags=0):
    capi.load().bgs_conv_bfx_wide_tuning(int(mode) | (int(flags) << 16), int(nst), int(splitk))


def last():
    v = capi.load().bgs_conv_bfx_wide_last_launch()
    return dict(ran=v & 1, nst=(v >> 4) & 15, splits=(v >> 8) & 0xfff, nbw=2 if v & 0x100000 else 4)


def correctness(dev):
    """wide (forced, unsliced) == ring, bit for bit: odd M, stride 2, both residual modes, ReLU, Cout % 128 != 0."""
    cases = [
        # N, H, W, Cin, Cout, stride, relu, res_mode
        (2, 50, 84, 256, 1024, 1, True, 1),
        (2, 50, 84, 1024, 256, 1, True, 0),
        (1, 37, 29, 64, 256, 1, False, 0),            # odd M: row clamp
        (2, 100, 168, 256, 512, 2, False, 0),         # stride-2 projection shortcut
        (2, 50, 84, 512, 256, 1, False, 2),           # FPN lateral: nearest-2x upsampled residual
        (1024, 1, 1, 1024, 1236, 1, False, 0),        # fc_cls: Cout % 128 != 0: column clamp
        (3, 17, 23, 128, 132, 1, True, 1),            # both edges
    ]
    ok = True
    for (N, H, W, Cin, Cout, stride, relu, rm) in cases:
        torch.manual_seed(N * 1000 + Cin)
        x = torch.randn(N, H, W, Cin, device=dev)
        w = torch.randn(Cout, 1, 1, Cin, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        res = None
        if rm == 1:
            res = torch.randn(N, Ho, Wo, Cout, device=dev)
        elif rm == 2:
            res = torch.randn(N, Ho // 2, Wo // 2, Cout, device=dev)
        kw = dict(stride=stride, pad=0, relu=relu, residual=res, residual_mode=rm)
        wide(0)
        BF.conv_bfx_tuning(0, 1)                      # the ring unsliced too (fc_cls: its plan slices K four ways)
        y0 = BF.conv2d_nhwc(x, w, b, **kw)
        BF.conv_bfx_tuning(0, -1)
        assert not last()['ran']
        outs = {}
        for nst in (2, 3):
            wide(2, nst, 1)
            y1 = BF.conv2d_nhwc(x, w, b, **kw)
            u = last()
            assert u['ran'] and u['nst'] == nst and u['splits'] == 1, u
            outs[nst] = bool(torch.equal(y0, y1))
        # sliced K: not the same summation order; compare with fp64
        wide(2, 0, 2)
        ws_needed = BF.capi.load().bgs_conv_bfx_workspace_bytes(N * Ho * Wo, Cout, Cin)
        y2 = BF.conv2d_nhwc(x, w, b, **kw)
        u2 = last()
        ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(),
                                         stride=stride).permute(0, 2, 3, 1)
        if rm == 1:
            ref = ref + res.double()
        elif rm == 2:
            ref = ref + res.double().repeat_interleave(2, 1).repeat_interleave(2, 2)
        if relu:
            ref = ref.clamp_min(0)
        e0 = float((y0.double() - ref).abs().max())
        e2 = float((y2.double() - ref).abs().max())
        good = outs[2] and outs[3] and e2 < 4 * max(e0, 1e-6)
        ok = ok and good
        print('case N%d %dx%d %d->%d s%d relu%d res%d | nst2 == ring: %s  nst3 == ring: %s | max err vs fp64: ring %.2e '
              'sliced-wide(x%d, ws %d) %.2e  %s' % (N, H, W, Cin, Cout, stride, relu, rm, outs[2], outs[3], e0,
                                                   u2['splits'], ws_needed, e2, 'ok' if good else 'FAIL'), flush=True)
    wide(1)
    print('CORRECT' if ok else 'WRONG', flush=True)
    return ok


def stress(dev, reps=60):
    """Every (ring depth, K slicing) arm `reps` times per shape under full occupancy (a second stream keeps the
    chip busy with a copy): every repetition must reproduce the first result bit for bit (unsliced: the ring's)."""
    shapes = [(2, 50, 84, 256, 1024, 1, True, 1), (2, 100, 168, 512, 256, 1, False, 2), (2, 200, 336, 256, 256, 1, True, 0),
              (1024, 1, 1, 1024, 1236, 1, False, 0), (2, 25, 42, 2048, 512, 1, True, 0)]
    bad = 0
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    for (N, H, W, Cin, Cout, stride, relu, rm) in shapes:
        x = torch.randn(N, H, W, Cin, device=dev)
        w = torch.randn(Cout, 1, 1, Cin, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        res = None
        if rm == 1:
            res = torch.randn(N, Ho, Wo, Cout, device=dev)
        elif rm == 2:
            res = torch.randn(N, Ho // 2, Wo // 2, Cout, device=dev)
        kw = dict(stride=stride, pad=0, relu=relu, residual=res, residual_mode=rm)
        wide(0)
        BF.conv_bfx_tuning(0, 1)
        y_ring = BF.conv2d_nhwc(x, w, b, **kw).clone()
        BF.conv_bfx_tuning(0, -1)
        for nst in (2, 3):
            for sk in (1, 2, 4):
                wide(2, nst, sk)
                first = None
                n_bad = 0
                for r in range(reps):
                    if r % 3 == 0:
                        with torch.cuda.stream(side):
                            junk.add_(1)
                    y = BF.conv2d_nhwc(x, w, b, **kw)
                    if sk == 1:
                        n_bad += int(not torch.equal(y, y_ring))
                    else:
                        if first is None:
                            first = y.clone()
                            n_bad += int(float((y - y_ring).abs().max()) > 1e-4 * float(y_ring.abs().max()))
                        n_bad += int(not torch.equal(y, first))
                u = last()
                bad += n_bad
                print('stress N%d %dx%d %d->%d res%d | nst %d slices %d (ran x%d) : %d / %d mismatches' %
                      (N, H, W, Cin, Cout, rm, nst, sk, u['splits'], n_bad, reps), flush=True)
    torch.cuda.synchronize()
    wide(1)
    print('STRESS ' + ('CLEAN' if bad == 0 else 'FAILED (%d)' % bad), flush=True)
    return bad == 0


def flags_sweep(dev):
    """WideArgs::flags (bit 0 priority per residency slot, bit 1 staggered start, bit 2 DMA issue between the MFMA
    halves) x ring depth on a few layers; every arm must equal the ring bit for bit."""
    shapes = [('fpn.lat0', 2, 200, 336, 256, 256, 1), ('l2.ds', 2, 200, 336, 256, 512, 2), ('fpn.lat1', 2, 100, 168, 512, 256, 1),
              ('l3.c3', 2, 50, 84, 256, 1024, 1), ('l3.ds', 2, 100, 168, 512, 1024, 2), ('l2.b0.c1', 2, 200, 336, 256, 128, 1)]
    fl = [0, 1, 2, 4, 5, 6, 3, 7]
    print('%-10s | ring | %s' % ('layer', '  '.join('n%d/f%d' % (n, f) for n in (2, 3) for f in fl)))
    for name, N, H, W, Cin, Cout, stride in shapes:
        x = torch.randn(N, H, W, Cin, device=dev)
        w = torch.randn(Cout, 1, 1, Cin, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        fn = lambda: BF.conv2d_nhwc(x, w, b, stride=stride, pad=0, relu=True)  # noqa: E731
        wide(0)
        y0 = fn().clone()
        t0 = bench(fn)
        row = []
        for nst in (2, 3):
            for f in fl:
                wide(2, nst, 1, f)
                y = fn()
                assert last()['ran']
                eq = torch.equal(y, y0)
                row.append('%6.1f%s' % (bench(fn), '' if eq else '!'))
        print('%-10s | %5.1f | %s' % (name, t0, '  '.join(row)), flush=True)
    wide(1)


def main():
    dev = 'cuda:0'
    quick = '--quick' in sys.argv
    os.environ['BGS_CONV_HALO'] = '0'
    if '--stress' in sys.argv:
        sys.exit(0 if stress(dev) else 1)
    if '--flags' in sys.argv:
        return flags_sweep(dev)
    if not correctness(dev):
        sys.exit(1)
    layers = [(n, H, W, Cin, Cout, R, s, c) for (n, H, W, Cin, Cout, R, s, c) in LAYERS if R == 1]
    layers += [(n, 1, 1, K, Cout, 1, 1, 1) for (n, M, K, Cout) in FC]
    arms = [('ring', (0, 0, -1)), ('w-auto', (2, 0, -1)), ('w-n2', (2, 2, 1)), ('w-n3', (2, 3, 1))]
    if not quick:
        arms += [('w-k2', (2, 0, 2)), ('w-k4', (2, 0, 4))]
    if '--narrow' in sys.argv:      # the 128 x 64 tile (bits 4..7 of nst = 2) against the ring and the 128 x 128 tile
        arms = [('ring', (0, 0, -1)), ('w4-auto', (2, 0, -1)), ('w2-n2', (2, 2 | 0x20, 1)), ('w2-n3', (2, 3 | 0x20, 1)),
                ('w2-auto', (2, 0x20, -1)), ('ring', (0, 0, -1)), ('w2-n2', (2, 2 | 0x20, 1))]
    tot = {a: 0.0 for a, _ in arms}
    tot['best'] = 0.0
    if '--narrow' in sys.argv:      # bit-identity of the 128 x 64 tile first
        for (N_, H_, W_, Ci, Co, st_, relu, rm) in [(2, 50, 84, 256, 1024, 1, True, 1), (1, 37, 29, 64, 64, 1, False, 0),
                                                     (2, 100, 168, 256, 512, 2, False, 0), (3, 17, 23, 128, 132, 1, True, 1),
                                                     (2, 50, 84, 512, 256, 1, False, 2)]:
            x = torch.randn(N_, H_, W_, Ci, device=dev); w = torch.randn(Co, 1, 1, Ci, device=dev) * 0.05
            b = torch.randn(Co, device=dev)
            Ho, Wo = (H_ - 1) // st_ + 1, (W_ - 1) // st_ + 1
            res = torch.randn(N_, Ho, Wo, Co, device=dev) if rm == 1 else (torch.randn(N_, Ho // 2, Wo // 2, Co, device=dev) if rm == 2 else None)
            kw = dict(stride=st_, pad=0, relu=relu, residual=res, residual_mode=rm)
            wide(0); BF.conv_bfx_tuning(0, 1); y0 = BF.conv2d_nhwc(x, w, b, **kw).clone(); BF.conv_bfx_tuning(0, -1)
            for nst in (2, 3):
                wide(2, nst | 0x20, 1)
                y1 = BF.conv2d_nhwc(x, w, b, **kw)
                u = last()
                print('128x64 nst %d: N%d %dx%d %d->%d s%d res%d ran=%s nbw=%d equal=%s' % (
                    nst, N_, H_, W_, Ci, Co, st_, rm, u['ran'], u['nbw'], bool(torch.equal(y0, y1))), flush=True)
    print('%-12s %7s %6s %5s %6s | %s' % ('layer', 'M', 'K', 'Cout', 'tiles', '  '.join('%9s' % a for a, _ in arms)))
    for name, H, W, Cin, Cout, R, stride, cnt in layers:
        Nn = 1024 if name.startswith('fc') else NIMG
        if Cout < (64 if '--narrow' in sys.argv else 128) or Cin % 16:
            continue
        x = torch.randn(Nn, H, W, Cin, device=dev)
        w = torch.randn(Cout, 1, 1, Cin, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        M = Nn * Ho * Wo
        res = torch.randn(Nn, Ho, Wo, Cout, device=dev) if '.c3' in name else None
        row = []
        best = 1e9
        for a, (mode, nst, sk) in arms:
            wide(mode, nst, sk)
            fn = lambda: BF.conv2d_nhwc(x, w, b, stride=stride, pad=0, relu=True, residual=res)  # noqa: E731
            fn()
            u = last()
            us = bench(fn)
            tag = '%7.1f%s' % (us, ('/%d%s' % (u['nst'], 'x%d' % u['splits'] if u['splits'] > 1 else '')) if u['ran'] else '  ')
            row.append('%9s' % tag)
            tot[a] += us * cnt
            best = min(best, us)
        tot['best'] += best * cnt
        tiles = ((M + 127) // 128) * ((Cout + 127) // 128)
        gf = 2.0 * M * Cin * Cout / 1e9
        print('%-12s %7d %6d %5d %6d | %s  x%d  (%.1f GF, best %.0f TF)' % (name, M, Cin, Cout, tiles, '  '.join(row),
                                                                           cnt, gf, gf / best * 1e3), flush=True)
    wide(1)
    print('per step (us): ' + '  '.join('%s %.0f' % (k, v) for k, v in tot.items()))


if __name__ == '__main__':
    main()
