"""GPU parity tests of the bf16 STORAGE kernels (csrc/conv_bf16s.hip; BASELINE cfg[4] bf16 mode =
the reference's `wrap_fp16_model` + `Fp16OptimizerHook`, mmdet/core/fp16/decorators.py:8-80,
hooks.py:11-127: half tensors in memory, half x half products, fp32 accumulate).

Oracle: torch-CPU fp64 convolution of the SAME bf16-valued inputs with the bf16-rounded filter
(every product of two bf16 values is exact in fp32 / fp64, so the only freedom the kernel has is
the fp32 summation order and the final rounding to bf16: the result must sit within one bf16 ulp
of the rounded fp64 value, and mostly ON it)."""
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from balancedgroupsoftmax_amd import functional as BF

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def check_bf16(got, ref64, what, frac_exact=0.97):
    """``got`` bf16 (or fp32) tensor vs an fp64 reference: within 1 bf16 ulp (2^-8 relative, plus the
    fp32 accumulation noise near zero), and at least ``frac_exact`` of the elements equal to the
    correctly rounded value."""
    g = got.detach().cpu().to(torch.float64)
    scale = float(ref64.abs().max())
    if got.dtype == torch.bfloat16:
        tol = ref64.abs() * 2.0 ** -7 + 2e-6 * scale
        exact = (g == ref64.to(torch.bfloat16).to(torch.float64)).double().mean().item()
        assert exact >= frac_exact, (what, exact)
    else:
        tol = torch.full_like(ref64, 2e-6 * scale)
    bad = (g - ref64).abs() > tol
    assert not bool(bad.any()), (what, int(bad.sum()), float((g - ref64).abs().max()), scale)


CONV_CASES = [
    # name, N, H, W, Cin, Cout, R, stride, pad, bias, relu, res ('', 'bf16', 'f32', 'up_f32'), out dtype
    ('l1_conv1_k64', 2, 25, 42, 64, 256, 1, 1, 0, True, True, '', 'bf16'),
    ('conv3_res_bf16', 2, 25, 42, 256, 256, 1, 1, 0, True, True, 'bf16', 'bf16'),
    ('l3_k1024', 1, 50, 84, 1024, 1024, 1, 1, 0, True, True, 'bf16', 'bf16'),
    ('downsample_s2', 2, 28, 30, 256, 512, 1, 2, 0, True, False, '', 'bf16'),
    ('3x3_r50_conv2', 2, 20, 30, 64, 64, 3, 1, 1, True, True, '', 'bf16'),
    ('3x3_s2', 1, 33, 47, 128, 128, 3, 2, 1, True, True, '', 'bf16'),
    ('ragged_cout_k', 1, 9, 11, 24, 200, 3, 1, 1, True, False, 'f32', 'bf16'),
    ('fpn_lateral_f32_out', 2, 16, 24, 512, 256, 1, 1, 0, True, False, 'up_f32', 'f32'),
    ('fpn_top_f32_out', 2, 8, 12, 2048, 256, 1, 1, 0, True, False, '', 'f32'),
    ('no_bias', 1, 17, 19, 64, 72, 1, 1, 0, False, True, '', 'bf16'),
]


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d_bf16_storage_vs_fp64_of_the_same_bf16_operands(case):
    name, N, H, W, Cin, Cout, R, stride, pad, use_bias, relu, res, out = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    x = bf16_round(torch.randn(N, H, W, Cin, generator=g))
    w = torch.randn(Cout, R, R, Cin, generator=g) * (2.0 / (R * R * Cin)) ** 0.5
    b = torch.randn(Cout, generator=g) if use_bias else None
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    r, rmode = None, 0
    if res in ('bf16', 'f32'):
        r, rmode = torch.randn(N, Ho, Wo, Cout, generator=g), 1
        if res == 'bf16':
            r = bf16_round(r)
    elif res == 'up_f32':
        r, rmode = torch.randn(N, Ho // 2, Wo // 2, Cout, generator=g), 2
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), bf16_round(w).double().permute(0, 3, 1, 2),
                   None if b is None else b.double(), stride=stride, padding=pad)
    if rmode == 1:
        ref = ref + r.double().permute(0, 3, 1, 2)
    elif rmode == 2:
        ref = ref + F.interpolate(r.double().permute(0, 3, 1, 2), scale_factor=2, mode='nearest')
    if relu:
        ref = torch.relu(ref)
    ref = ref.permute(0, 2, 3, 1).contiguous()
    rd = None
    if r is not None:
        rd = r.to(DEV).to(torch.bfloat16 if res == 'bf16' else torch.float32).contiguous()
    BF.launch_census(reset=True)
    y = BF.conv2d_nhwc(x.to(DEV).to(torch.bfloat16), w.to(DEV), None if b is None else b.to(DEV),
                       stride=stride, pad=pad, relu=relu, residual=rd, residual_mode=rmode,
                       out_dtype=torch.bfloat16 if out == 'bf16' else torch.float32)
    assert BF.launch_census()['bf16s'] == 1
    assert y.dtype == (torch.bfloat16 if out == 'bf16' else torch.float32)
    check_bf16(y, ref, name)


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d_bf16_storage_register_staged_variant_is_bit_identical(case):
    """``bgs_conv_bf16s_tuning(1)``: the operands reach LDS through registers instead of by LDS-DMA —
    the same tile, the same summation order, the same bits."""
    from balancedgroupsoftmax_amd import capi
    name, N, H, W, Cin, Cout, R, stride, pad, use_bias, relu, res, out = case
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) + 1)
    x = torch.randn(N, H, W, Cin, generator=g).to(DEV).to(torch.bfloat16)
    w = (torch.randn(Cout, R, R, Cin, generator=g) * (2.0 / (R * R * Cin)) ** 0.5).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV) if use_bias else None
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    r, rmode = None, 0
    if res in ('bf16', 'f32'):
        r, rmode = torch.randn(N, Ho, Wo, Cout, generator=g).to(DEV), 1
        if res == 'bf16':
            r = r.to(torch.bfloat16)
    elif res == 'up_f32':
        r, rmode = torch.randn(N, Ho // 2, Wo // 2, Cout, generator=g).to(DEV), 2
    od = torch.bfloat16 if out == 'bf16' else torch.float32
    ys = []
    lib = capi.load()
    try:
        for variant in (0, 1):
            lib.bgs_conv_bf16s_tuning(variant)
            ys.append(BF.conv2d_nhwc(x, w, b, stride=stride, pad=pad, relu=relu, residual=r,
                                     residual_mode=rmode, out_dtype=od))
    finally:
        lib.bgs_conv_bf16s_tuning(0)
    assert torch.equal(ys[0], ys[1])


def test_conv2d_bf16_storage_equals_the_bf16_operand_mode_on_fp32_tensors():
    """The storage kernel computes what the operand-rounding mode computes: the same bf16 x bf16
    products with fp32 accumulation — on inputs that are already bf16 values the fp32 results agree
    to accumulation order."""
    g = torch.Generator().manual_seed(7)
    x = bf16_round(torch.randn(2, 40, 56, 512, generator=g)).to(DEV)
    w = (torch.randn(512, 1, 1, 512, generator=g) * 0.06).to(DEV)
    b = torch.randn(512, generator=g).to(DEV)
    prev = BF.set_conv_math('bf16')
    try:
        y32 = BF.conv2d_nhwc(x, w, b, relu=True)
    finally:
        BF.set_conv_math(prev)
    y16 = BF.conv2d_nhwc(x.to(torch.bfloat16), w, b, relu=True, out_dtype=torch.float32)
    assert float((y32 - y16).abs().max()) <= 2e-6 * float(y32.abs().max())


GROUPED_CASES = [
    # name, N, H, W, C, groups, stride
    ('x101_l1_cg4', 2, 24, 40, 256, 64, 1),
    ('x101_l2_cg8', 1, 25, 42, 512, 64, 1),
    ('x101_l3_cg16', 1, 20, 33, 1024, 64, 1),
    ('x101_l4_cg32', 1, 13, 21, 2048, 64, 1),
    ('x101_l2_0_cg8_s2', 1, 50, 84, 512, 64, 2),
    ('x101_l3_0_cg16_s2', 1, 25, 42, 1024, 64, 2),
    ('x101_l4_0_cg32_s2', 1, 13, 21, 2048, 64, 2),
    ('x50_32x4d_cg4', 1, 16, 16, 128, 32, 1),
    ('c_not_64_direct', 1, 9, 10, 48, 3, 1),
]


@pytest.mark.parametrize('case', GROUPED_CASES, ids=[c[0] for c in GROUPED_CASES])
def test_grouped_conv3x3_bf16_storage_vs_fp64(case):
    name, N, H, W, C, groups, stride = case
    cg = C // groups
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    x = bf16_round(torch.randn(N, H, W, C, generator=g))
    w = torch.randn(C, 3, 3, cg, generator=g) * (2.0 / (9 * cg)) ** 0.5
    b = torch.randn(C, generator=g)
    ref = torch.relu(F.conv2d(x.double().permute(0, 3, 1, 2), bf16_round(w).double().permute(0, 3, 1, 2),
                              b.double(), stride=stride, padding=1, groups=groups))
    ref = ref.permute(0, 2, 3, 1).contiguous()
    BF.launch_census(reset=True)
    y = BF.grouped_conv3x3_nhwc(x.to(DEV).to(torch.bfloat16), w.to(DEV), b.to(DEV), groups,
                                stride=stride, relu=True)
    assert BF.launch_census()['grouped_bf16s'] == 1
    assert y.dtype == torch.bfloat16 and tuple(y.shape) == tuple(ref.shape)
    check_bf16(y, ref, name)


def test_maxpool_into_the_bf16_trunk_is_the_rounded_fp32_pool():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 37, 50, 64, generator=g)
    ref = F.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1).to(torch.bfloat16)
    y = BF.maxpool3x3s2_nhwc(x.to(DEV), out_dtype=torch.bfloat16)
    assert y.dtype == torch.bfloat16 and torch.equal(y.cpu(), ref)


def test_resnext_trunk_and_fpn_with_bf16_storage_track_the_fp32_storage_mode():
    """Whole frozen trunk + neck: ResNeXt-50 32x4d + FPN under the 'bf16' arithmetic with bf16 tensors
    in HBM against the same arithmetic on fp32 tensors.  The two differ by one extra rounding per
    layer output (the reference's half tensors do the same): pyramid levels agree to a few bf16 ulps
    of their range; trainable layers and grad mode fall back to fp32 storage."""
    import balancedgroupsoftmax_amd as bgs
    from balancedgroupsoftmax_amd.registry import BACKBONES, NECKS, build_from_cfg
    from oracle import det_oracle
    torch.manual_seed(0)
    backbone = build_from_cfg(dict(type='ResNeXt', depth=50, groups=32, base_width=4, num_stages=4,
                                   out_indices=(0, 1, 2, 3), frozen_stages=1, style='pytorch'), BACKBONES)
    neck = build_from_cfg(dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256,
                               num_outs=5), NECKS)
    with torch.no_grad():
        det_oracle.fill_detector(backbone.state_dict(), 5)
        det_oracle.fill_detector(neck.state_dict(), 6)
    backbone.to(DEV).eval()
    neck.to(DEV).eval()
    for p in list(backbone.parameters()) + list(neck.parameters()):
        p.requires_grad_(False)
    img = torch.randn(1, 3, 224, 288, generator=torch.Generator().manual_seed(1)).to(DEV)
    prev = BF.set_conv_math('bf16')
    outs = {}
    try:
        for storage in (False, True):
            prev_s = BF.set_bf16_storage(storage)
            BF.launch_census(reset=True)
            with torch.no_grad():
                c = backbone(img)
                p = neck(c)
            census = BF.launch_census()
            BF.set_bf16_storage(prev_s)
            assert all(t.dtype == (torch.bfloat16 if storage else torch.float32) for t in c)
            assert all(t.dtype == torch.float32 for t in p)
            if storage:
                assert census['grouped_bf16s'] == 16 and census['bf16s'] == 16 * 2 + 4 + 4, census
            else:
                assert census['grouped_bf16s'] == 0 and census['bf16s'] == 0, census
            outs[storage] = [t.float().cpu() for t in p]
        # a trainable block keeps fp32 tensors
        for q in backbone.layer4.parameters():
            q.requires_grad_(True)
        c = backbone(img)
        assert all(t.dtype == torch.float32 for t in c)
    finally:
        BF.set_conv_math(prev)
    for a, b in zip(outs[False], outs[True]):
        rng = float(a.abs().max())
        assert float((a - b).abs().max()) < 0.06 * rng, (float((a - b).abs().max()), rng)
        assert float((a - b).abs().mean()) < 0.006 * rng, (float((a - b).abs().mean()), rng)
