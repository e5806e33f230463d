#!/usr/bin/env python
"""Sustained fp32 MFMA rate of the chip (register operands, no memory): python tools/mfma_peak.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balancedgroupsoftmax_amd import capi  # noqa: E402

lib = capi.load()
out = torch.zeros(4, device='cuda')
st = capi.current_stream(out.device)
for blocks, sign in ((256, 1), (512, 1), (1024, 1), (2048, 1)):
    iters = 20000 * sign
    for _ in range(2):
        lib.bgs_selftest_mfma_peak(blocks, iters, capi.ptr(out), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        lib.bgs_selftest_mfma_peak(blocks, iters, capi.ptr(out), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    flops = blocks * 4.0 * 4 * abs(iters) * 4096
    print('blocks=%5d (%.1f waves/SIMD) %s operands: %.3f ms -> %.1f TFLOP/s' % (
        blocks, blocks * 4 / 1024.0, 'random' if sign < 0 else 'constant', ms, flops / ms / 1e9))

# ---- bf16 MFMA (v_mfma_f32_32x32x16_bf16), the instruction of the bf16x6 / bf16 conv kernels: zero vs random operands
print('v_mfma_f32_32x32x16_bf16, 24 MFMAs per loop trip in the product order of the bf16x6 kernels:')
for blocks in (256, 512, 768, 1024):
    iters = 2000
    for rnd in (0, 1):
        for _ in range(2):
            lib.bgs_selftest_mfma_peak_bf16(blocks, iters, rnd, capi.ptr(out), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.bgs_selftest_mfma_peak_bf16(blocks, iters, rnd, capi.ptr(out), st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        flops = blocks * 4.0 * 24 * iters * 32768
        print('blocks=%5d (%.2f waves/SIMD) %-6s operands: %.3f ms -> %7.1f TFLOP/s bf16 = %6.1f TFLOP/s of bf16x6 work '
              '(%.2f of 2500 / 416.7)' % (blocks, blocks * 4 / 1024.0, 'random' if rnd else 'zero', ms, flops / ms / 1e9,
                                          flops / ms / 1e9 / 6, flops / ms / 1e9 / 2500))
