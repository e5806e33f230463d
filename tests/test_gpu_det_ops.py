"""GPU parity tests of the detector ops behind the C ABI: fp32-MFMA implicit-GEMM conv /
linear, max pooling, multi-level RoIAlign, batched NMS — against the CPU oracles
(oracle/det_oracle.py; torch-CPU conv; the compiled reference nms_cpu.cpp)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from balancedgroupsoftmax_amd import functional as BF
from oracle import build_ref, det_oracle

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def krsc(w_oihw):
    return np.ascontiguousarray(np.transpose(w_oihw, (0, 2, 3, 1)))


CONV_CASES = [
    # name, N, H, W, Cin, Cout, R, stride, pad, bias, relu, res_mode
    ('1x1_small', 2, 25, 42, 64, 128, 1, 1, 0, True, True, 0),
    ('3x3_res_relu', 2, 20, 30, 32, 96, 3, 1, 1, True, True, 1),
    ('3x3_s2', 1, 33, 47, 16, 40, 3, 2, 1, False, False, 0),
    ('7x7_stem', 1, 64, 96, 4, 64, 7, 2, 3, True, True, 0),
    ('1x1_s2_downsample', 2, 28, 28, 64, 256, 1, 2, 0, True, False, 0),
    ('fpn_lateral_upsample_add', 2, 16, 24, 48, 64, 1, 1, 0, True, False, 2),
    ('tile_128x128', 2, 64, 128, 32, 512, 3, 1, 1, True, True, 0),
    ('tile_128x64', 2, 128, 256, 16, 64, 1, 1, 0, True, False, 0),
    ('ragged_cout_k', 1, 9, 11, 12, 70, 3, 1, 1, True, False, 0),
]


MATHS = ['bf16x6', 'f32']


@pytest.fixture
def conv_math(request):
    """Runs the test body under one of the two conv arithmetic modes and restores every
    process-wide tuning hook afterwards."""
    prev = BF.set_conv_math(request.param)
    yield request.param
    BF.set_conv_math(prev)
    BF.conv_tuning()
    BF.conv_bfx_tuning()


@pytest.mark.parametrize('conv_math', MATHS, indirect=True)
@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d_vs_torch_cpu(case, conv_math):
    name, N, H, W, Cin, Cout, R, stride, pad, use_bias, relu, res_mode = case
    import zlib
    rs = np.random.RandomState(zlib.crc32(name.encode()))
    x = rs.standard_normal((N, H, W, Cin)).astype(np.float32)
    w = (rs.standard_normal((Cout, Cin, R, R)) / np.sqrt(Cin * R * R)).astype(np.float32)
    b = rs.standard_normal(Cout).astype(np.float32) if use_bias else None
    Ho = (H + 2 * pad - R) // stride + 1
    Wo = (W + 2 * pad - R) // stride + 1
    res = res_full = None
    if res_mode == 1:
        res = res_full = rs.standard_normal((N, Ho, Wo, Cout)).astype(np.float32)
    elif res_mode == 2:
        res = rs.standard_normal((N, Ho // 2, Wo // 2, Cout)).astype(np.float32)
        res_full = np.repeat(np.repeat(res, 2, axis=1), 2, axis=2)   # F.interpolate(nearest, 2x)
    y = BF.conv2d_nhwc(dev(x), dev(krsc(w)), None if b is None else dev(b), stride=stride, pad=pad,
                       relu=relu, residual=None if res is None else dev(res),
                       residual_mode=res_mode).cpu().numpy()
    exp = det_oracle.conv2d_nhwc(x, w, b, stride, pad, relu, res_full, dtype='float64')
    assert y.shape == exp.shape
    err = np.abs(y - exp).max()
    assert err < 2e-5 * max(1.0, np.abs(exp).max()), err
    exp32 = det_oracle.conv2d_nhwc(x, w, b, stride, pad, relu, res_full)
    assert np.abs(y - exp32).max() < 1e-4 * max(1.0, np.abs(exp32).max())


def _conv_ref64(x, w, b, stride, pad, relu):
    y = F.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2),
                 torch.from_numpy(w).double().permute(0, 3, 1, 2),
                 torch.from_numpy(b).double(), stride=stride, padding=pad)
    return (y.relu() if relu else y).permute(0, 2, 3, 1).numpy()


# Every template instantiation the two launchers can choose (csrc/conv_igemm.hip launch_conv:
# tile {22, 21, 11} x BK {16, 32} x UP {1, 2}; csrc/conv_bfx.hip: tile {22, 21, 12, 11} x UP {1, 2}),
# forced through the tuning hooks and CONFIRMED through the last-launch queries — the size-based
# defaults only reach the larger tiles at M >= 100,000.
F32_INST = [(t, bk, up) for t in (22, 21, 11) for bk in (16, 32) for up in (1, 2)]
BFX_INST = [(t, up) for t in (22, 21, 12, 11, 11 | 0x800, 11 | 0x100) for up in (1, 2)]
# 64x64 tile: default = the LDS-DMA ring kernel with 3 stages (five workgroups / CU), 0x800 = its 4-stage
# instantiation, 0x100 = register-staged


def _inst_problem(up, seed):
    """up = 1: a forward 3x3 conv with bias + ReLU; up = 2: the data gradient of a stride-2 3x3
    conv (the zero-upsampled read path).  Sizes ragged against every tile (M = 2*37*45 = 3330 /
    dgrad 2*37*45 input pixels, Cout 200 / Cin 72)."""
    rs = np.random.RandomState(seed)
    if up == 1:
        N, H, W, Cin, Cout = 2, 37, 45, 40, 200
        x = rs.standard_normal((N, H, W, Cin)).astype(np.float32)
        w = (rs.standard_normal((Cout, 3, 3, Cin)) / np.sqrt(9 * Cin)).astype(np.float32)
        b = rs.standard_normal(Cout).astype(np.float32)
        exp = _conv_ref64(x, w, b, 1, 1, True)
        return (lambda: BF.conv2d_nhwc(dev(x), dev(w), dev(b), pad=1, relu=True)), exp
    N, H, W, Cin, Cout = 2, 37, 45, 72, 136
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    dy = rs.standard_normal((N, Ho, Wo, Cout)).astype(np.float32)
    w = (rs.standard_normal((Cout, 3, 3, Cin)) / np.sqrt(9 * Cout)).astype(np.float32)
    xr = torch.zeros(N, Cin, H, W, dtype=torch.float64, requires_grad=True)
    yr = F.conv2d(xr, torch.from_numpy(w).double().permute(0, 3, 1, 2), stride=2, padding=1)
    (gx,) = torch.autograd.grad(yr, xr, torch.from_numpy(dy).double().permute(0, 3, 1, 2))
    return (lambda: BF.conv2d_dgrad_nhwc(dev(dy), dev(w), (H, W), stride=2, pad=1)), \
        gx.permute(0, 2, 3, 1).numpy()


@pytest.mark.parametrize('inst', F32_INST, ids=['t%d_bk%d_up%d' % i for i in F32_INST])
def test_conv_f32_every_instantiation(inst):
    tile, bk, up = inst
    prev = BF.set_conv_math('f32')
    try:
        run, exp = _inst_problem(up, tile * 100 + bk + up)
        BF.conv_tuning(tile=tile, bk=bk, splitk=1)
        got = run().cpu().numpy()
        used = BF.conv_last_launch()
        assert (used['tile'], used['bk'], used['up'], used['splits']) == (tile, bk, up, 1), used
        assert np.abs(got - exp).max() < 2e-5 * np.abs(exp).max()
        BF.conv_tuning(tile=tile, bk=bk, splitk=3)           # the same tile with split-K
        got = run().cpu().numpy()
        assert BF.conv_last_launch()['splits'] == 3
        assert np.abs(got - exp).max() < 2e-5 * np.abs(exp).max()
    finally:
        BF.conv_tuning()
        BF.set_conv_math(prev)


@pytest.mark.parametrize('inst', BFX_INST, ids=['t%d_up%d' % i for i in BFX_INST])
def test_conv_bfx_every_instantiation(inst):
    tile, up = inst
    prev = BF.set_conv_math('bf16x6')
    os.environ['BGS_CONV_HALO'] = '0'
    try:
        run, exp = _inst_problem(up, tile * 100 + up)
        if up == 1:      # + the 1x1 / no-padding specialisation of the DMA kernels (stride 2, ragged)
            rs = np.random.RandomState(tile)
            x1 = rs.standard_normal((2, 27, 35, 48)).astype(np.float32)
            w1 = (rs.standard_normal((72, 1, 1, 48)) / 7).astype(np.float32)
            b1 = rs.standard_normal(72).astype(np.float32)
            exp1 = _conv_ref64(x1, w1, b1, 2, 0, False)
            BF.conv_bfx_tuning(tile=tile, splitk=1)
            got1 = BF.conv2d_nhwc(dev(x1), dev(w1), dev(b1), stride=2).cpu().numpy()
            assert np.abs(got1 - exp1).max() < 2e-5 * np.abs(exp1).max()
        for splitk in (1, 3):
            BF.conv_bfx_tuning(tile=tile, splitk=splitk)
            got = run().cpu().numpy()
            used = BF.conv_bfx_last_launch()
            # tile 11 runs the LDS-DMA ring kernel (reported with bit 9), 11 | 0x100 the register-
            # staged one
            expect = {11: 11 | 0x200, 11 | 0x800: 11 | 0x200, 11 | 0x100: 11}.get(tile, tile)
            assert (used['tile'], used['splits']) == (expect, splitk), used
            if tile & 0xff == 11 and not tile & 0x100:
                assert used['ring_stages'] == (4 if tile & 0x800 else 3), used
            assert np.abs(got - exp).max() < 2e-5 * np.abs(exp).max()
    finally:
        os.environ.pop('BGS_CONV_HALO', None)
        BF.conv_bfx_tuning()
        BF.set_conv_math(prev)


def _bf16_rne(x):
    """fp32 array -> (bf16 bits, that bf16 as fp32): round to nearest even, subnormals kept (v_cvt_pk_bf16_f32)."""
    b = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((b + 0x7fff + ((b >> 16) & 1)) >> 16).astype(np.uint16)
    return r, (r.astype(np.uint32) << 16).view(np.float32)


def test_bfx_split_planes_match_the_numpy_restatement():
    """The exact three-way split every bf16x6 kernel applies to its operands (csrc/bfx_split.h: hi = bf16(x), mid =
    bf16(x - hi), lo = bf16(x - hi - mid); the residuals formed by v_dot2c_f32_bf16 on the packed pair), observed through
    ``bgs_conv_bfx_split_weights``: all three planes equal the numpy restatement bit for bit — both elements of a
    packed pair (the inline-constant encoding of the multiplier got the low element wrong), large and tiny magnitudes,
    rounding ties, fp32 subnormals (kept, not flushed) — and hi + mid + lo == x exactly (|x| > 1e-30)."""
    g = np.random.default_rng(0)
    n = 1 << 14
    cases = [
        (g.standard_normal(n) * np.exp2(g.integers(-30, 31, n))),
        np.maximum(g.standard_normal(n), 0),
        np.concatenate([np.exp2(g.integers(-20, 20, n // 2)) * (1 + np.exp2(-8.0)), np.exp2(g.integers(-40, 40, n // 2))]),
        g.uniform(1, 3, n) * np.exp2(g.integers(100, 127, n)),
        g.uniform(1, 2, n) * np.exp2(g.integers(-125, -100, n)),
        g.uniform(0, 1, n) * 1.1e-38,
        g.standard_normal(37),                                          # K % 16 != 0: zero-padded tail
    ]
    for x in cases:
        x = (x * np.where(g.integers(0, 2, x.size) > 0, 1, -1)).astype(np.float32)
        K = x.size
        KC = 2 * ((K + 31) // 32)                          # planes [3][KC][rows][16], K padded to 32
        buf = BF.bfx_split_weights(torch.from_numpy(x).to('cuda:0').view(1, K), cache=False)
        raw = buf.cpu().numpy()[:3 * KC * 32].view(np.uint16).reshape(3, KC * 16)
        h, hf = _bf16_rne(x)
        r = x - hf
        m, mf = _bf16_rne(r)
        lo, lf = _bf16_rne(r - mf)
        for got, want in zip(raw, (h, m, lo)):
            assert np.array_equal(got[:K], want) and not got[K:].any()
        big = (np.abs(x) > 1e-30) | (x == 0)              # below that the low plane runs out of bf16 exponent range
        assert np.array_equal((hf.astype(np.float64) + mf.astype(np.float64) + lf.astype(np.float64))[big],
                              x.astype(np.float64)[big])


def test_bfx_error_not_above_f32_mfma():
    """The bf16x6 kernels are not a reduced-precision mode: against an fp64 reference their error
    (normalised by sum |a||b|, the fp32 rounding scale of the reduction) is at the level of the
    fp32 MFMA kernel's — for a deep reduction (K = 4608), wide-dynamic-range operands (so that the
    mid / lo planes matter) and the halo kernel."""
    rs = np.random.RandomState(77)
    N, H, W, Cin, Cout = 1, 24, 40, 512, 128
    x = (rs.standard_normal((N, H, W, Cin)) * np.exp(rs.uniform(-4, 4, (N, H, W, Cin)))).astype(np.float32)
    w = (rs.standard_normal((Cout, 3, 3, Cin)) * np.exp(rs.uniform(-4, 4, (Cout, 3, 3, Cin))) / 70).astype(np.float32)
    b = np.zeros(Cout, np.float32)
    exp = _conv_ref64(x, w, b, 1, 1, False)
    den = _conv_ref64(np.abs(x), np.abs(w), b, 1, 1, False).max()
    errs = {}
    for math, halo in (('f32', '0'), ('bf16x6', '0'), ('bf16x6', '1')):
        prev = BF.set_conv_math(math)
        os.environ['BGS_CONV_HALO'] = halo
        try:
            y = BF.conv2d_nhwc(dev(x), dev(w), dev(b), pad=1).cpu().numpy().astype(np.float64)
        finally:
            os.environ.pop('BGS_CONV_HALO', None)
            BF.set_conv_math(prev)
        errs[(math, halo)] = np.abs(y - exp).max() / den
    assert errs[('f32', '0')] < 5e-7, errs
    for k in (('bf16x6', '0'), ('bf16x6', '1')):
        assert errs[k] < max(1.5 * errs[('f32', '0')], 1.5e-7), errs


def test_linear_matches_fc_shapes():
    """RoI-head FC shapes (convfc_bbox_head.py): 12544 -> 1024 (+ReLU), 1024 -> 1236."""
    rs = np.random.RandomState(5)
    x = rs.standard_normal((130, 12544)).astype(np.float32)
    w1 = (rs.standard_normal((1024, 12544)) / 112).astype(np.float32)
    b1 = rs.standard_normal(1024).astype(np.float32)
    h = BF.linear(dev(x), dev(w1), dev(b1), relu=True)
    exp = np.maximum(x.astype(np.float64) @ w1.astype(np.float64).T + b1, 0)
    assert np.abs(h.cpu().numpy() - exp).max() < 5e-5 * np.abs(exp).max()
    w2 = (rs.standard_normal((1236, 1024)) / 32).astype(np.float32)
    z = BF.linear(h, dev(w2), None)
    exp2 = exp @ w2.astype(np.float64).T
    assert np.abs(z.cpu().numpy() - exp2).max() < 5e-5 * np.abs(exp2).max()


def test_linear_autograd_vs_torch():
    """dX, dW, db of the MFMA linear (what trains fc_cls under selectp=1) vs torch autograd."""
    rs = np.random.RandomState(8)
    x = rs.standard_normal((200, 1024)).astype(np.float32)
    w = (rs.standard_normal((1236, 1024)) / 32).astype(np.float32)
    b = rs.standard_normal(1236).astype(np.float32)
    gy = rs.standard_normal((200, 1236)).astype(np.float32)
    xt, wt, bt = (torch.from_numpy(a).double().requires_grad_(True) for a in (x, w, b))
    (torch.nn.functional.linear(xt, wt, bt) * torch.from_numpy(gy).double()).sum().backward()
    xd, wd, bd = (dev(a).requires_grad_(True) for a in (x, w, b))
    y = BF.linear_autograd(xd, wd, bd)
    (y * dev(gy)).sum().backward()
    for got, exp in ((xd.grad, xt.grad), (wd.grad, wt.grad), (bd.grad, bt.grad)):
        assert float((got.cpu().double() - exp).abs().max()) < 5e-5 * float(exp.abs().max())
    # frozen input / bias-free variants
    xd2 = dev(x)
    wd2 = dev(w).requires_grad_(True)
    BF.linear_autograd(xd2, wd2, None, relu=True).sum().backward()
    exp = (torch.relu(torch.from_numpy(x).double() @ torch.from_numpy(w).double().t()) > 0).double()
    expw = exp.t() @ torch.from_numpy(x).double()
    assert float((wd2.grad.cpu().double() - expw).abs().max()) < 5e-5 * float(expw.abs().max())


@pytest.mark.parametrize('shape', [(1, 96, 160), (2, 67, 93), (1, 33, 31), (2, 224, 320)],
                         ids=lambda c: 'x'.join(str(v) for v in c))
def test_fused_stem_conv_relu_maxpool_vs_torch_fp64_and_the_three_launch_chain(shape):
    """``bgs_stem_conv7x7s2_relu_maxpool_nchw_f32`` (round 5: mmdet/models/backbones/resnet.py:522-533 conv1 -> BN
    (folded) -> ReLU -> MaxPool2d(3, 2, 1) in one launch from the NCHW image) == torch-CPU fp64 within the bf16x6
    family's bound and == the chain nchw_to_nhwc4 -> conv2d (K = 7 * 7 * 4) -> maxpool3x3s2 it replaces (same
    products, another summation order); odd sizes: tiles hanging over the pooled map, conv rows / columns that exist
    only as pool padding."""
    N, H, W = shape
    g = torch.Generator().manual_seed(H * 3 + W)
    img = torch.randn(N, 3, H, W, generator=g) * torch.exp(0.5 * torch.randn(N, 3, H, W, generator=g))
    w = torch.randn(64, 3, 7, 7, generator=g) * (2.0 / 147) ** 0.5
    b = torch.randn(64, generator=g) * 0.5
    exp = F.max_pool2d(F.relu(F.conv2d(img.double(), w.double(), b.double(), stride=2, padding=3)), 3, 2, 1)
    exp = exp.permute(0, 2, 3, 1)
    wk = torch.nn.functional.pad(w.permute(0, 2, 3, 1), (0, 1)).contiguous()        # [64, 7, 7, 4] as the fold hands it over
    prev = BF.set_conv_math('bf16x6')
    try:
        BF.launch_census(reset=True)
        ws = BF.stem_fused_split_weights(dev(wk))
        got = BF.stem_fused(dev(img), ws, dev(b))
        assert BF.launch_census()['stem_fused'] == 1
        chain = BF.maxpool3x3s2_nhwc(BF.conv2d_nhwc(BF.nchw_to_nhwc4(dev(img)), dev(wk), dev(b), stride=2, pad=3,
                                                    relu=True))
    finally:
        BF.set_conv_math(prev)
    scale = float(exp.abs().max())
    assert tuple(got.shape) == tuple(exp.shape) == tuple(chain.shape)
    assert float((got.cpu().double() - exp).abs().max()) <= 2e-6 * scale
    assert float((got - chain).abs().max()) <= 1e-5 * scale
    assert float(got.min()) >= 0.0


def test_resnet_stem_takes_the_fused_kernel_when_frozen_and_the_chain_otherwise(monkeypatch):
    """Dispatch: a frozen stem under bf16x6 launches the fused kernel once (no NHWC image copy, no max-pool launch);
    ``BGS_STEM_FUSED=0``, the fp32-MFMA arithmetic and a trainable stem keep the chain; same feature maps."""
    import balancedgroupsoftmax_amd as bgs
    from balancedgroupsoftmax_amd.config import to_config_dict
    torch.manual_seed(3)
    net = bgs.build_backbone(to_config_dict(dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3),
                                                 frozen_stages=1, style='pytorch'))).to(DEV).eval()
    img = torch.randn(1, 3, 128, 160, device=DEV)
    prev = BF.set_conv_math('bf16x6')
    try:
        with torch.no_grad():
            BF.launch_census(reset=True)
            a = net(img)
            assert BF.launch_census()['stem_fused'] == 1
            monkeypatch.setenv('BGS_STEM_FUSED', '0')
            BF.launch_census(reset=True)
            b = net(img)
            assert BF.launch_census()['stem_fused'] == 0
            monkeypatch.delenv('BGS_STEM_FUSED')
            BF.set_conv_math('f32')
            BF.launch_census(reset=True)
            net(img)
            assert BF.launch_census()['stem_fused'] == 0
        for u, v in zip(a, b):
            assert float((u - v).abs().max()) <= 2e-5 * float(v.abs().max())
    finally:
        BF.set_conv_math(prev)


def test_maxpool3x3s2():
    rs = np.random.RandomState(6)
    for (N, H, W, C) in [(2, 17, 23, 8), (1, 64, 96, 64)]:
        x = rs.standard_normal((N, H, W, C)).astype(np.float32)
        y = BF.maxpool3x3s2_nhwc(dev(x)).cpu().numpy()
        exp = torch.nn.functional.max_pool2d(torch.from_numpy(x).permute(0, 3, 1, 2), 3, 2, 1)
        np.testing.assert_array_equal(y, exp.permute(0, 2, 3, 1).numpy())


@pytest.mark.parametrize('C', [16, 256])
def test_roi_align_multilevel_vs_oracle(C):
    rs = np.random.RandomState(7 + C)
    strides = [4, 8, 16, 32]
    feats = [rs.standard_normal((2, 200 // (s // 4), 336 // (s // 4), C)).astype(np.float32)
             for s in strides]
    K = 48
    wh = np.exp(rs.uniform(np.log(8), np.log(700), (K, 2)))
    xy = np.stack([rs.uniform(-20, 1300, K), rs.uniform(-20, 780, K)], 1)
    rois = np.concatenate([rs.randint(0, 2, (K, 1)), xy, xy + wh], 1).astype(np.float32)
    rois[0] = [0, 5000, 5000, 5100, 5100]          # completely outside -> zeros
    rois[1] = [1, 10, 10, 5, 5]                     # malformed (x2 < x1): width clamps to >= 0
    rois[2:6] = [[0, 30, 40, 80, 100], [1, 30, 40, 180, 200], [0, 30, 40, 330, 360],
                 [1, 30, 40, 700, 650]]             # one RoI for each of the four levels
    out, lv = BF.roi_align_nhwc([dev(f) for f in feats], dev(rois), strides, return_levels=True)
    exp, exp_lv = det_oracle.roi_align_multilevel(feats, rois, strides)
    np.testing.assert_array_equal(lv.cpu().numpy(), exp_lv)
    assert set(exp_lv.tolist()) == {0, 1, 2, 3}
    err = np.abs(out.cpu().numpy() - exp).max()
    # sample coordinates are fp32 chains whose fma contraction is compiler-defined (also in the
    # reference's nvcc build): 1-ulp coordinate differences move a bilinear tap by ~1e-5
    assert err < 1e-4 * max(1.0, np.abs(exp).max()), err
    assert not out[0].any()


@pytest.mark.parametrize('pool', [1, 2])
def test_roi_align_tap_grid_kernel_equals_the_sample_at_a_time_kernel_bit_for_bit(pool):
    """The forward kernel that loads every distinct feature pixel of a bin once (``roi_align_fwd_grid_kernel``, default)
    and the one that walks the 2 x 2 sample points one after the other (``BGS_ROI_DEDUP=0``): same weights, same products,
    same summation order — identical bits, on RoIs of every level incl. ones that stick out of the image, are degenerate
    or lie completely outside, with and without accumulation into ``out``, C a multiple of 256 or not."""
    rs = np.random.RandomState(31 + pool)
    strides = [4, 8, 16, 32]
    for C in ((256, 64) if pool == 1 else (256,)):
        feats = [dev(rs.standard_normal((2, 200 // (s // 4), 336 // (s // 4), C)).astype(np.float32)) for s in strides]
        K = 600
        wh = np.exp(rs.uniform(np.log(4), np.log(900), (K, 2)))
        xy = np.stack([rs.uniform(-60, 1340, K), rs.uniform(-60, 800, K)], 1)
        rois = np.concatenate([rs.randint(0, 2, (K, 1)), xy, xy + wh], 1).astype(np.float32)
        rois[0] = [0, 5000, 5000, 5100, 5100]
        rois[1] = [1, 10, 10, 5, 5]
        rois[2] = [0, 1300, 780, 1400, 900]
        rois[3] = [1, 100.25, 50.5, 100.25, 50.5]            # a single point: every sample in one pixel cell
        r = dev(rois)
        base = dev(rs.standard_normal((K, 7, 7, C)).astype(np.float32))
        got = {}
        try:
            for mode in ('0', '1'):
                os.environ['BGS_ROI_DEDUP'] = mode
                acc = base.clone()
                got[mode] = (BF.roi_align_nhwc(feats, r, strides, out_size=7, pool=pool),
                             BF.roi_align_nhwc(feats, r, strides, out_size=7, pool=pool, out=acc))
        finally:
            os.environ.pop('BGS_ROI_DEDUP', None)
        assert torch.equal(got['0'][0], got['1'][0]) and torch.equal(got['0'][1], got['1'][1])
        assert got['1'][0].abs().sum() > 0 and not got['1'][0][0].any()


@pytest.mark.parametrize('mode', [0, 1])
def test_nms_batched_vs_oracle_and_reference(mode):
    counts = [2000, 1337, 64, 65, 1, 0, 500]
    nmax = 2000
    P = len(counts)
    boxes = np.zeros((P, nmax, 5), np.float32)
    sorted_dets = []
    for p, n in enumerate(counts):
        d = det_oracle.make_boxes(n, seed=100 + p) if n else np.zeros((0, 5), np.float32)
        d = d[np.argsort(-d[:, 4], kind='stable')]
        boxes[p, :n] = d
        sorted_dets.append(d)
    thr = 0.7
    keep, kc = BF.nms_batched(dev(boxes), torch.tensor(counts, dtype=torch.int32, device=DEV), thr,
                              iou_mode=mode)
    keep, kc = keep.cpu().numpy(), kc.cpu().numpy()
    ref = build_ref.load_nms_cpu()
    for p, n in enumerate(counts):
        exp = det_oracle.nms(sorted_dets[p], thr, mode='cpu' if mode else 'cuda')
        assert kc[p] == len(exp), (p, kc[p], len(exp))
        np.testing.assert_array_equal(keep[p, :kc[p]], exp)
        if mode == 1 and ref is not None and n:
            r = ref.nms(torch.from_numpy(sorted_dets[p]), thr).numpy()
            np.testing.assert_array_equal(keep[p, :kc[p]], r)


def test_nms_properties_and_max_keep():
    """Idempotence: NMS of the kept set keeps everything; max_keep truncates the prefix."""
    d = det_oracle.make_boxes(1500, seed=9)
    d = d[np.argsort(-d[:, 4], kind='stable')]
    cnt = torch.tensor([1500], dtype=torch.int32, device=DEV)
    keep, kc = BF.nms_batched(dev(d[None]), cnt, 0.5)
    k = keep[0, :int(kc[0])].cpu().numpy()
    kept = np.zeros((1, 1500, 5), np.float32)
    kept[0, :len(k)] = d[k]
    keep2, kc2 = BF.nms_batched(dev(kept), torch.tensor([len(k)], dtype=torch.int32, device=DEV), 0.5)
    assert int(kc2[0]) == len(k)
    np.testing.assert_array_equal(keep2[0, :len(k)].cpu().numpy(), np.arange(len(k)))
    keep3, kc3 = BF.nms_batched(dev(d[None]), cnt, 0.5, max_keep=10)
    assert int(kc3[0]) == 10
    np.testing.assert_array_equal(keep3[0, :10].cpu().numpy(), k[:10])


@pytest.mark.parametrize('case', [
    # (N, L, nmax, num, kept counts per (image, level), quantised scores -> ties across levels)
    (2, 5, 2000, 2000, None, False),                     # the cfg[1] shape: ~10,000 kept of which 2000 are taken
    (2, 5, 2000, 2000, None, True),                      # scores on a coarse grid: many exact ties across levels
    (1, 3, 300, 1000, [[7, 0, 12]], False),              # fewer kept boxes than `num`: trailing invalid slots
    (3, 4, 500, 64, None, True),
], ids=['cfg1', 'cfg1_ties', 'short', 'small_num'])
def test_proposal_tail_merge_equals_the_topk_tail(case):
    """``bgs_nms_merge_select`` (the per-image top ``max_num`` over the levels as an L-way merge of the levels'
    score-sorted kept lists: ONE launch) == ``bgs_nms_gather`` + ``bgs_topk_sorted_f32`` + ``bgs_gather_boxes``
    (rpn_head.py:99-103 as eleven launches) — the same boxes in the same order, bit for bit, ties across levels
    included (lower level first: the composite order of the radix select), ``valid`` identical."""
    N, L, nmax, num, counts, ties = case
    rs = np.random.RandomState(nmax + num + int(ties))
    boxes = np.zeros((N * L, nmax, 5), np.float32)
    for r in range(N * L):
        d = det_oracle.make_boxes(nmax, seed=100 + r)
        if ties:
            d[:, 4] = np.round(d[:, 4] * 40) / 40
        boxes[r] = d[np.argsort(-d[:, 4], kind='stable')]
    cnt = torch.full((N * L,), nmax, dtype=torch.int32, device=DEV)
    keep, kc = BF.nms_batched(dev(boxes), cnt, 0.7)
    if counts is not None:
        kc = torch.tensor(np.array(counts, np.int32).reshape(-1), device=DEV)
    num = min(num, L * nmax)                              # as RPNHead._nms_and_select does
    props, valid = BF.nms_merge_select(dev(boxes), keep, kc, N, num)
    kept, kept_scores = BF.nms_gather(dev(boxes), keep, kc)
    flat, flat_s = kept.view(N, L * nmax, 5), kept_scores.view(N, L * nmax)
    top_s, top_i = BF.topk_sorted([flat_s], [num], num)
    props0, valid0 = BF.gather_boxes(flat, top_i.view(N, num), top_s.view(N, num))
    assert torch.equal(valid, valid0)
    nv = valid.sum(1).cpu().numpy()
    total = kc.view(N, L).sum(1).clamp(max=num).cpu().numpy()
    np.testing.assert_array_equal(nv, total)
    for n in range(N):
        a, b = props[n, :nv[n]].cpu().numpy(), props0[n, :nv[n]].cpu().numpy()
        assert (np.diff(a[:, 4]) <= 0).all()
        np.testing.assert_array_equal(a[:, 4], b[:, 4])            # the same scores in the same order, always
        # Which of several boxes EQUAL to the num-th score make the cut is unspecified in the radix select (arrival
        # order of its collect pass, as in torch.topk); the merge takes them in concatenated order.  Everything above
        # that last tie group must agree box for box; inside it the merge's choice must be the FIRST ones.
        cut = int((a[:, 4] > a[-1, 4]).sum()) if nv[n] == num else nv[n]
        np.testing.assert_array_equal(a[:cut], b[:cut])
        if cut < nv[n]:
            fs = flat_s[n].cpu().numpy()
            first = np.nonzero(fs == a[-1, 4])[0][:nv[n] - cut]
            np.testing.assert_array_equal(a[cut:], flat[n].cpu().numpy()[first])
        assert float(props[n, nv[n]:].abs().max() if nv[n] < num else 0.0) == 0.0


def test_proposal_tail_merge_with_nan_and_equal_scores_fills_every_slot():
    """Ranks come from a TOTAL order (the uint32 key image of the score, as in bgs_topk_sorted_f32), so NaN scores
    (sorted above +inf upstream) and runs of equal scores across levels cannot make two entries claim one output
    slot or leave a slot below the kept total unwritten."""
    N, L, nmax, num = 2, 4, 64, 200
    boxes = np.zeros((N * L, nmax, 5), np.float32)
    for r in range(N * L):
        d = det_oracle.make_boxes(nmax, seed=300 + r)
        d[:, 4] = np.round(d[:, 4] * 8) / 8                # eight distinct values: long tie runs across levels
        d = d[np.argsort(-d[:, 4], kind='stable')]
        if r % 2 == 0:
            d[:3, 4] = np.nan                              # NaNs lead a descending list (key order)
        boxes[r] = d
        boxes[r, :, 0] = r * 1000 + np.arange(nmax)        # every box identifiable
    keep = torch.arange(nmax, dtype=torch.int32, device=DEV).repeat(N * L, 1).contiguous()
    kc = torch.tensor([50, 64, 0, 33, 64, 1, 17, 64], dtype=torch.int32, device=DEV)
    props = torch.full((N, num, 5), -7.0, device=DEV)
    valid = torch.full((N, num), 9, dtype=torch.uint8, device=DEV)
    props, valid = BF.nms_merge_select(dev(boxes), keep, kc, N, num, out=(props, valid))
    for n in range(N):
        total = min(int(kc.view(N, L)[n].sum()), num)
        v = valid[n].cpu().numpy()
        assert (v[:total] == 1).all() and (v[total:] == 0).all()
        ids = props[n, :total, 0].cpu().numpy()
        assert len(set(ids.tolist())) == total and (ids >= 0).all()     # a permutation of kept boxes, none twice
        sc = props[n, :total, 4].cpu().numpy()
        lead = int(np.isnan(sc).sum())
        assert np.isnan(sc[:lead]).all() and (np.diff(sc[lead:]) <= 0).all()
        assert float(props[n, total:].abs().max()) == 0.0 if total < num else True


# ---------------------------------------------------------------- multiclass NMS (test-time path)
def _mc_cases():
    import json
    import os
    from tests.golden import make_golden_det
    z = np.load(os.path.join(os.path.dirname(make_golden_det.__file__), 'multiclass_nms_golden.npz'))
    return z, json.loads(bytes(z['__cases__']).decode()), make_golden_det.case_inputs


@pytest.mark.parametrize('name', ['c31_cut', 'c11_agnostic_all', 'c1231_lvis', 'c1231_thr',
                                  'c21_empty', 'c5_nocap'])
def test_multiclass_nms_vs_executed_reference_golden(name):
    """One batched launch over all classes == the reference's per-class Python loop (executed on
    CPU with its own nms_cpu.cpp -> IoU >= thr semantics -> iou_mode=1).  Bit-exact boxes,
    scores, labels and order."""
    from balancedgroupsoftmax_amd.post_processing import multiclass_nms
    z, cases, case_inputs = _mc_cases()
    case = [c for c in cases if c['name'] == name][0]
    boxes, scores = case_inputs(case)
    db, dl = multiclass_nms(dev(boxes), dev(scores), case['score_thr'],
                            dict(type='nms', iou_thr=case['iou_thr']), case['max_num'], iou_mode=1)
    assert dl.dtype == torch.int64 and db.shape[1:] == (5,)
    np.testing.assert_array_equal(dl.cpu().numpy(), z[name + '/det_labels'])
    np.testing.assert_array_equal(db.cpu().numpy(), z[name + '/det_bboxes'])


@pytest.mark.parametrize('name', ['c31_cut', 'c11_agnostic_all', 'c5_nocap'])
def test_multiclass_nms_gpu_semantics_vs_oracle(name):
    """Default iou_mode=0 (IoU > thr, nms_kernel.cu:60) against the numpy restatement."""
    from balancedgroupsoftmax_amd.post_processing import bbox2result, multiclass_nms
    _, cases, case_inputs = _mc_cases()
    case = [c for c in cases if c['name'] == name][0]
    boxes, scores = case_inputs(case)
    eb, el = det_oracle.multiclass_nms(boxes, scores, case['score_thr'], case['iou_thr'],
                                       case['max_num'], mode='cuda')
    db, dl = multiclass_nms(dev(boxes), dev(scores), case['score_thr'],
                            dict(type='nms', iou_thr=case['iou_thr']), case['max_num'])
    np.testing.assert_array_equal(dl.cpu().numpy(), el)
    np.testing.assert_array_equal(db.cpu().numpy(), eb)
    res = bbox2result(db, dl, case['C'])
    assert len(res) == case['C'] - 1 and sum(r.shape[0] for r in res) == eb.shape[0]
    for c, r in enumerate(res):
        np.testing.assert_array_equal(r, eb[el == c])


def test_multiclass_nms_score_factors_and_padding_rows():
    from balancedgroupsoftmax_amd.post_processing import multiclass_nms
    boxes, scores = det_oracle.make_multiclass_case(120, 9, 7, agnostic=True, clusters=4)
    scores[100:] = -1.0                                  # padding rows of a fixed-shape list
    db, dl = multiclass_nms(dev(boxes), dev(scores), 0.0, dict(type='nms', iou_thr=0.5), 1000)
    eb, el = det_oracle.multiclass_nms(boxes[:100], scores[:100], 0.0, 0.5, 1000, mode='cuda')
    np.testing.assert_array_equal(db.cpu().numpy(), eb)
    np.testing.assert_array_equal(dl.cpu().numpy(), el)


# ---------------------------------------------------------------- conv backward (selectp = 0)
BWD_CASES = [
    # name, N, H, W, Cin, Cout, k, stride, pad
    ('3x3s1', 2, 20, 28, 128, 128, 3, 1, 1),
    ('3x3s2_odd', 2, 21, 27, 128, 64, 3, 2, 1),
    ('3x3s2_even', 1, 24, 32, 64, 128, 3, 2, 1),
    ('1x1s1', 2, 25, 42, 256, 64, 1, 1, 0),
    ('1x1s2', 2, 26, 40, 256, 128, 1, 2, 0),
    ('1x1_thin16', 2, 50, 84, 256, 16, 1, 1, 0),
    ('linear', 1, 1, 300, 512, 132, 1, 1, 0),
    ('3x3s1_wide', 1, 13, 21, 256, 256, 3, 1, 1),
]


def _torch_conv_grads(x, w_oihw, dy, stride, pad):
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).double().requires_grad_(True)
    wt = torch.from_numpy(w_oihw).double().requires_grad_(True)
    y = torch.nn.functional.conv2d(xt, wt, None, stride=stride, padding=pad)
    y.backward(torch.from_numpy(dy).permute(0, 3, 1, 2).double())
    return (xt.grad.permute(0, 2, 3, 1).contiguous().numpy(),
            wt.grad.permute(0, 2, 3, 1).contiguous().numpy())


@pytest.mark.parametrize('case', BWD_CASES, ids=[c[0] for c in BWD_CASES])
def test_conv2d_dgrad_wgrad_vs_torch_autograd(case):
    """dx / dw / db of the implicit-GEMM conv against fp64 torch-CPU autograd of F.conv2d."""
    name, N, H, W, Cin, Cout, k, stride, pad = case
    rs = np.random.RandomState(hash(name) % 1000)
    x = rs.randn(N, H, W, Cin).astype(np.float32)
    w = (rs.randn(Cout, Cin, k, k) / np.sqrt(Cin * k * k)).astype(np.float32)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    dy = rs.randn(N, Ho, Wo, Cout).astype(np.float32)
    edx, edw = _torch_conv_grads(x, w, dy, stride, pad)
    wk = dev(krsc(w))
    dx = BF.conv2d_dgrad_nhwc(dev(dy), wk, (H, W), stride=stride, pad=pad)
    assert np.abs(dx.cpu().numpy() - edx).max() <= 2e-5 * max(1.0, np.abs(edx).max())
    dw, db = BF.conv2d_wgrad_nhwc(dev(x), dev(dy), k, stride=stride, pad=pad, bias=True)
    assert dw.shape == (Cout, k, k, Cin)
    assert np.abs(dw.cpu().numpy() - edw).max() <= 2e-5 * np.abs(edw).max()
    edb = dy.astype(np.float64).sum((0, 1, 2))
    assert np.abs(db.cpu().numpy() - edb).max() <= 2e-5 * max(1.0, np.abs(edb).max())
    # accumulate: second call doubles
    BF.conv2d_wgrad_nhwc(dev(x), dev(dy), k, stride=stride, pad=pad, bias=True, dw=dw, db=db,
                         accumulate=True)
    assert np.abs(dw.cpu().numpy() - 2 * edw).max() <= 4e-5 * np.abs(edw).max()
    assert np.abs(db.cpu().numpy() - 2 * edb).max() <= 4e-5 * max(1.0, np.abs(edb).max())


def test_conv2d_dgrad_epilogues_and_wgrad_reproducibility():
    """residual (same-shape / 2x2 sum-pooled) + ReLU-backward mask in the dgrad epilogue; the
    split-reduction wgrad is bitwise reproducible run to run."""
    rs = np.random.RandomState(5)
    N, H, W, Cin, Cout = 2, 12, 18, 64, 128
    dy = rs.randn(N, H, W, Cout).astype(np.float32)
    w = (rs.randn(Cout, Cin, 3, 3) * 0.05).astype(np.float32)
    x = rs.randn(N, H, W, Cin).astype(np.float32)
    res1 = rs.randn(N, H, W, Cin).astype(np.float32)
    res3 = rs.randn(N, 2 * H, 2 * W, Cin).astype(np.float32)
    edx, _ = _torch_conv_grads(x, w, dy, 1, 1)
    wk = dev(krsc(w))
    got = BF.conv2d_dgrad_nhwc(dev(dy), wk, (H, W), 1, 1, residual=dev(res1), mask=dev(x)).cpu().numpy()
    exp = np.where(x > 0, edx + res1, 0)
    assert np.abs(got - exp).max() < 5e-5
    got = BF.conv2d_dgrad_nhwc(dev(dy), wk, (H, W), 1, 1, residual=dev(res3), residual_mode=3).cpu().numpy()
    pooled = res3.reshape(N, H, 2, W, 2, Cin).sum((2, 4))
    assert np.abs(got - (edx + pooled)).max() < 5e-5
    big_x = dev(rs.randn(2, 60, 84, 64).astype(np.float32))
    big_dy = dev(rs.randn(2, 60, 84, 32).astype(np.float32))
    a = BF.conv2d_wgrad_nhwc(big_x, big_dy, 3, 1, 1)
    b = BF.conv2d_wgrad_nhwc(big_x, big_dy, 3, 1, 1)
    assert torch.equal(a, b)


# ---------------------------------------------------------------- RoIAlign backward (selectp = 0)
def _ml_case(seed, C=16):
    rs = np.random.RandomState(seed)
    strides = [4, 8, 16, 32]
    feats = [rs.randn(2, 64 // (s // 4), 96 // (s // 4), C).astype(np.float32) for s in strides]
    K = 60
    ctr = rs.uniform(0, 1, (K, 2)) * np.array([384, 256])
    size = np.exp(rs.uniform(np.log(8), np.log(500), (K, 2)))
    rois = np.concatenate([rs.randint(0, 2, (K, 1)), ctr - size / 2, ctr + size / 2], 1).astype(np.float32)
    return strides, feats, rois


def test_roi_align_backward_vs_oracle_and_adjoint():
    strides, feats, rois = _ml_case(11)
    rs = np.random.RandomState(12)
    g = rs.randn(rois.shape[0], 7, 7, feats[0].shape[3]).astype(np.float32)
    dfeats = [torch.zeros(f.shape, device=DEV) for f in feats]
    BF.roi_align_nhwc_bwd(dev(g), dev(rois), dfeats, strides)
    lv = det_oracle.map_roi_levels(rois, 4)
    for i, s in enumerate(strides):
        idx = np.nonzero(lv == i)[0]
        exp = det_oracle.roi_align_backward(g[idx], rois[idx], 1.0 / s, feats[i].shape, 2)
        got = dfeats[i].cpu().numpy()
        assert np.abs(got - exp).max() <= 1e-4 * max(1.0, np.abs(exp).max()), i
    # adjoint identity with the HIP forward
    out = BF.roi_align_nhwc([dev(f) for f in feats], dev(rois), strides).cpu().numpy().astype(np.float64)
    lhs = (out * g).sum()
    rhs = sum((f.astype(np.float64) * d.cpu().numpy()).sum() for f, d in zip(feats, dfeats))
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))


def test_roi_align_autograd_accumulates_into_feature_grads():
    strides, feats, rois = _ml_case(13)
    ft = [dev(f).requires_grad_(True) for f in feats]
    out = BF.roi_align_nhwc_autograd(ft, dev(rois), strides)
    (out * 2.0).sum().backward()
    ones = np.ones((rois.shape[0], 7, 7, feats[0].shape[3]), np.float32) * 2.0
    lv = det_oracle.map_roi_levels(rois, 4)
    for i, s in enumerate(strides):
        idx = np.nonzero(lv == i)[0]
        exp = det_oracle.roi_align_backward(ones[idx], rois[idx], 1.0 / s, feats[i].shape, 2)
        assert np.abs(ft[i].grad.cpu().numpy() - exp).max() <= 1e-4 * max(1.0, np.abs(exp).max())


@pytest.mark.parametrize('shape', [(1, 8, 16, 16, 128, False, -1), (2, 13, 21, 64, 256, True, -1),
                                   (1, 25, 42, 256, 200, True, 1), (2, 50, 84, 32, 64, False, 1),
                                   (1, 3, 5, 48, 15, True, 3), (1, 19, 37, 128, 64, True, 4),
                                   (2, 40, 56, 64, 128, True, 2)])
def test_halo_bfx_variants_bit_identical(monkeypatch, shape):
    """The bf16x6 halo kernel's default variant (4: filter slices by LDS-DMA, both Cout tile
    widths, dummy DMA pieces, channel-chunk split-K) == variant 2 (register-staged slices) bit for
    bit: same operand values, same MFMA order — only the way the slices reach LDS differs."""
    N, H, W, Cin, Cout, relu, hs = shape
    rs = np.random.RandomState(H * 11 + Cout)
    x = (rs.standard_normal((N, H, W, Cin)) * np.exp(rs.standard_normal((N, H, W, Cin)))).astype(np.float32)
    w = (rs.standard_normal((Cout, 3, 3, Cin)) / (9 * Cin) ** 0.5).astype(np.float32)
    b = rs.standard_normal(Cout).astype(np.float32)
    prev = BF.set_conv_math('bf16x6')
    monkeypatch.setenv('BGS_CONV_HALO', '1')
    try:
        BF.conv_bfx_tuning(halo_splits=hs, halo_variant=2)
        y2 = BF.conv2d_nhwc(dev(x), dev(w), dev(b), pad=1, relu=relu)
        u2 = BF.conv_bfx_last_launch()
        BF.conv_bfx_tuning(halo_splits=hs)
        y4 = BF.conv2d_nhwc(dev(x), dev(w), dev(b), pad=1, relu=relu)
        u4 = BF.conv_bfx_last_launch()
        assert (u2['halo_variant'], u4['halo_variant']) == (2, 4)
        assert u2['halo_nb'] == u4['halo_nb'] and u2['halo_splits'] == u4['halo_splits']
        assert torch.equal(y2, y4)
    finally:
        BF.conv_bfx_tuning()
        BF.set_conv_math(prev)


@pytest.mark.parametrize('case', [
    # (N, H, W, Cin, Cout, relu, mode, math): mode 2 = every eligible layer on the wide units, 1 = the automatic schedule
    (1, 16, 16, 16, 128, False, 2, 'bf16x6'),           # exactly one unit
    (2, 23, 37, 64, 256, True, 2, 'bf16x6'),            # tiles hanging over both image edges, two Cout tiles, two images
    (1, 50, 84, 256, 128, True, 2, 'bf16x6'),           # 16 channel chunks
    (1, 40, 56, 32, 256, True, 2, 'bf16'),              # the bf16 mode's one-plane instantiation
    (2, 200, 336, 16, 256, True, 1, 'bf16x6'),          # cfg[1]'s P2 map: 1092 units -> 1008 wide + the last rows on variant 4
    (2, 200, 336, 16, 256, False, 1, 'bf16'),
    (3, 136, 200, 32, 128, True, 1, 'bf16x6'),          # 351 px tiles x 1: below one round -> stays on variant 4
], ids=lambda c: 'x'.join(str(v) for v in c))
def test_halo_wide_pixel_tile_is_bit_identical_to_variant_4(case, monkeypatch):
    """``conv3x3_halo_bfx7_kernel`` (round 5: 16 x 16 pixels x 128 channels per workgroup, half the filter bytes
    per MFMA of the 8 x 16 tile; mmdet/models/necks/fpn.py:131-134 / anchor_heads/rpn_head.py:30-35 at the P2 level)
    accumulates every output in variant 4's order: BIT-IDENTICAL, alone (mode 2) and under the two-launch schedule
    (whole rounds of 512 wide units + the left-over image rows on variant 4), with the ReLU-backward mask of the
    data-gradient call, and == fp64 within the kernel family's bound."""
    N, H, W, Cin, Cout, relu, mode, math = case
    g = torch.Generator().manual_seed(H * 13 + W + Cin)
    x = torch.randn(N, H, W, Cin, generator=g) * torch.exp(torch.randn(N, H, W, Cin, generator=g))
    w = torch.randn(Cout, 3, 3, Cin, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, generator=g)
    monkeypatch.setenv('BGS_CONV_HALO', '1')
    prev = BF.set_conv_math(math)
    try:
        xd, wd, bd = dev(x), dev(w), dev(b)
        # (halo_splits=1: the wide units do not take part in the channel-chunk split of the small grids)
        BF.conv_bfx_tuning(halo_splits=1, halo_wide=0)
        y4 = BF.conv2d_nhwc(xd, wd, bd, pad=1, relu=relu)
        u4 = BF.conv_bfx_last_launch()
        assert u4['halo_variant'] == 4 and u4['halo_wide_units'] == 0
        BF.launch_census(reset=True)
        BF.conv_bfx_tuning(halo_splits=1, halo_wide=mode)
        y7 = BF.conv2d_nhwc(xd, wd, bd, pad=1, relu=relu)
        u7 = BF.conv_bfx_last_launch()
        census = BF.launch_census()
        ty, tx, tn = (H + 15) // 16, (W + 15) // 16, Cout // 128
        units = N * ty * tx * tn
        if mode == 2:
            assert u7['halo_variant'] == 7 and u7['halo_wide_units'] == units and u7['halo_tail_units'] == 0, u7
            assert census['halo_wide'] == 1 and census['halo_bfx4'] == 0
        elif units >= 512 and math == 'bf16x6':             # (the bf16 mode keeps variant 4 in the automatic mode: measured)
            rows = units // 512 * 512 // (tx * tn)
            assert u7['halo_variant'] == 7 and u7['halo_wide_units'] == rows * tx * tn, u7
            n_a, r_a = divmod(rows, ty)
            ty4 = (H + 7) // 8
            assert u7['halo_tail_units'] == (N * ty4 * tx - (n_a * ty4 + 2 * r_a) * tx) * tn, u7
            assert census['halo_wide'] == 1 and census['halo_bfx4'] == (1 if u7['halo_tail_units'] else 0)
        else:
            assert u7['halo_variant'] == 4 and u7['halo_wide_units'] == 0
        assert torch.equal(y7, y4)
        if math == 'bf16x6':
            exp = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), b.double(), padding=1)
            exp = (exp.relu() if relu else exp).permute(0, 2, 3, 1)
            assert float((y7.cpu().double() - exp).abs().max()) <= 2e-6 * float(exp.abs().max())
        if Cin % 128 == 0:
            # the data-gradient call of the same kernel pair: mask epilogue (conv2d_dgrad_nhwc routes 3x3 / stride 1
            # through the halo entry with the flipped, transposed filter; its output channels are the forward's Cin)
            dy = dev(torch.randn(N, H, W, Cout, generator=g))
            mask = dev(torch.randn(N, H, W, Cin, generator=g))
            BF.conv_bfx_tuning(halo_splits=1, halo_wide=0)
            a = BF.conv2d_dgrad_nhwc(dy, wd, (H, W), 1, 1, mask=mask)
            assert BF.conv_bfx_last_launch()['halo_variant'] == 4
            BF.conv_bfx_tuning(halo_splits=1, halo_wide=2)
            bb = BF.conv2d_dgrad_nhwc(dy, wd, (H, W), 1, 1, mask=mask)
            assert BF.conv_bfx_last_launch()['halo_variant'] == 7
            assert torch.equal(a, bb)
    finally:
        BF.conv_bfx_tuning()
        BF.set_conv_math(prev)


@pytest.mark.parametrize('conv_math', MATHS, indirect=True)
@pytest.mark.parametrize('shape', [(1, 8, 16, 16, 128, False), (2, 13, 21, 64, 256, True),
                                   (1, 25, 42, 256, 200, True), (2, 50, 84, 32, 64, False),
                                   (1, 3, 5, 48, 15, True), (1, 200, 513, 16, 128, True)])
def test_conv3x3_halo_kernel_vs_torch_cpu(monkeypatch, shape, conv_math):
    """csrc/conv_halo.hip / the halo kernel of csrc/conv_bfx.hip (input patch + halo staged in LDS
    once per channel chunk, shared by the nine taps) == torch-CPU F.conv2d and == the general
    implicit-GEMM kernel: tiles that hang over the image, Cout that is not a multiple of the
    128-wide tile, one pixel tile only, and a shape large enough for the DEFAULT dispatch to
    choose it."""
    N, H, W, Cin, Cout, relu = shape
    rs = np.random.RandomState(H * 7 + Cin)
    x = rs.randn(N, H, W, Cin).astype(np.float32)
    w = (rs.randn(Cout, 3, 3, Cin) * 0.05).astype(np.float32)
    b = rs.randn(Cout).astype(np.float32)
    exp = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(w).permute(0, 3, 1, 2),
                   torch.from_numpy(b), padding=1)
    if relu:
        exp = exp.relu()
    exp = exp.permute(0, 2, 3, 1)
    tol = 2e-5 * float(exp.abs().max())
    monkeypatch.setenv('BGS_CONV_HALO', '1')
    got = BF.conv2d_nhwc(dev(x), dev(w), dev(b), pad=1, relu=relu)
    assert float((got.cpu() - exp).abs().max()) <= tol
    if conv_math == 'bf16x6':
        for hs in (1, 2):                                      # channel-chunk split-K of the halo kernel
            BF.conv_bfx_tuning(halo_splits=hs)
            g2 = BF.conv2d_nhwc(dev(x), dev(w), dev(b), pad=1, relu=relu)
            assert BF.conv_bfx_last_launch()['halo_splits'] == min(hs, Cin // 16)
            assert float((g2.cpu() - exp).abs().max()) <= tol
        BF.conv_bfx_tuning()
    monkeypatch.setenv('BGS_CONV_HALO', '0')
    gen = BF.conv2d_nhwc(dev(x), dev(w), dev(b), pad=1, relu=relu)
    assert float((gen.cpu() - exp).abs().max()) <= tol
    monkeypatch.delenv('BGS_CONV_HALO')
    dflt = BF.conv2d_nhwc(dev(x), dev(w), dev(b), pad=1, relu=relu)
    if conv_math == 'f32':
        use_halo = BF._use_halo_kernel(N * H * W, Cout)
        assert use_halo == (N * H * W >= 100000 and Cout % 128 == 0)
    else:
        use_halo = BF._use_halo_bfx(N * H * W, Cout)
        assert use_halo == (N * H * W >= 2000)
    assert torch.equal(dflt, got if use_halo else gen)


@pytest.mark.parametrize('conv_math', MATHS, indirect=True)
def test_conv_split_k_matches_single_pass(conv_math):
    """Split-K (small-M layers) == the single-pass kernel for every epilogue: bias+ReLU, same-shape
    and upsampled residuals (forward), sum-pooled residual + ReLU mask (data gradient)."""
    rs = np.random.RandomState(9)
    N, H, W, Cin, Cout = 2, 10, 14, 512, 128                  # 5 x 2 workgroups, 16..144 K tiles
    x = dev(rs.randn(N, H, W, Cin).astype(np.float32))
    w = dev((rs.randn(Cout, 3, 3, Cin) * 0.02).astype(np.float32))
    b = dev(rs.randn(Cout).astype(np.float32))
    res1 = dev(rs.randn(N, H, W, Cout).astype(np.float32))
    res2 = dev(rs.randn(N, H // 2, W // 2, Cout).astype(np.float32))
    dy = dev(rs.randn(N, H, W, Cout).astype(np.float32))
    res3 = dev(rs.randn(N, 2 * H, 2 * W, Cin).astype(np.float32))
    mask = dev(rs.randn(N, H, W, Cin).astype(np.float32))

    def run():
        return [BF.conv2d_nhwc(x, w, b, pad=1, relu=True),
                BF.conv2d_nhwc(x, w, b, pad=1, residual=res1),
                BF.conv2d_nhwc(x, w, None, pad=1, residual=res2, residual_mode=2),
                BF.conv2d_dgrad_nhwc(dy, w, (H, W), 1, 1, residual=res3, residual_mode=3, mask=mask),
                BF.conv2d_dgrad_nhwc(dy, w, (H, W), 1, 1, residual=mask, mask=mask)]
    def force(k):
        if conv_math == 'f32':
            BF.conv_tuning(splitk=k)
        else:
            BF.conv_bfx_tuning(splitk=k if k else -1)
    force(1)
    ref = run()
    for f in (2, 5, 8):
        force(f)
        for a, e in zip(run(), ref):
            assert float((a - e).abs().max()) <= 2e-5 * float(e.abs().max())
    force(0)
    for a, e in zip(run(), ref):                              # the library's own choice
        assert float((a - e).abs().max()) <= 2e-5 * float(e.abs().max())


@pytest.mark.parametrize('out_size', [7, 14])
def test_roi_align_hip_vs_compiled_reference_kernels(out_size):
    """HIP RoIAlign forward + backward against the reference's own ROIAlignForward/Backward
    templates (roi_align_kernel.cu) compiled as host code into oracle/_ref/roi_align_ref.so."""
    if build_ref.load_roi_align() is None:
        pytest.skip('oracle/_ref/roi_align_ref.so not available')
    rs = np.random.RandomState(out_size)
    H, W, C = 23, 31, 8
    feat = rs.randn(2, H, W, C).astype(np.float32)
    K = 50
    ctr = rs.uniform(-0.1, 1.1, (K, 2)) * np.array([W * 8, H * 8])
    size = np.exp(rs.uniform(np.log(2), np.log(400), (K, 2)))
    rois = np.concatenate([rs.randint(0, 2, (K, 1)), ctr - size / 2, ctr + size / 2], 1).astype(np.float32)
    rois[0] = [0, 5.0, 5.0, 5.0, 5.0]
    rois[1] = [1, -40.0, -30.0, 20.0, 10.0]
    rois[2] = [0, W * 8 - 10.0, H * 8 - 12.0, W * 8 + 60.0, H * 8 + 50.0]
    exp = build_ref.roi_align_reference(feat.transpose(0, 3, 1, 2), rois, 0.125, out_size)
    # single level with stride 8: finest_scale huge -> every RoI maps to level 0
    got = BF.roi_align_nhwc([dev(feat)], dev(rois), [8], out_size=out_size, finest_scale=1e9)
    np.testing.assert_allclose(got.cpu().numpy().transpose(0, 3, 1, 2), exp, rtol=1e-4, atol=1e-5)
    g = rs.randn(K, out_size, out_size, C).astype(np.float32)
    eb = build_ref.roi_align_reference_backward(g.transpose(0, 3, 1, 2), rois, 0.125, (2, C, H, W))
    d = [torch.zeros(2, H, W, C, device=DEV)]
    BF.roi_align_nhwc_bwd(dev(g), dev(rois), d, [8], finest_scale=1e9)
    np.testing.assert_allclose(d[0].cpu().numpy().transpose(0, 3, 1, 2), eb, rtol=1e-3, atol=1e-4)


# ---------------------------------------------------------------- batched sorted top-k (RPN pre-selection)
def test_topk_sorted_vs_torch_all_rpn_levels():
    """bgs_topk_sorted_f32 == torch.topk(sorted) per (image, level): values exactly, indices
    exactly where the values are distinct (ties among EQUAL values are unordered in both)."""
    g = torch.Generator().manual_seed(0)
    N, nmax = 2, 2000
    lens = [201600, 50400, 12600, 3150, 819]
    rows = [(torch.randn(N, n, generator=g) * 0.7 - 1.0).to(DEV) for n in lens]
    ks = [min(n, nmax) for n in lens]
    vals, idx = BF.topk_sorted(rows, ks, nmax)
    assert tuple(vals.shape) == (N, 5, nmax) and idx.dtype == torch.int64
    for l, (r, k) in enumerate(zip(rows, ks)):
        ev, ei = r.topk(k, dim=1)
        assert torch.equal(vals[:, l, :k], ev), l
        assert torch.equal(r.gather(1, idx[:, l, :k]), ev), l          # indices point at the values
        distinct = torch.ones_like(ev, dtype=torch.bool)
        distinct[:, 1:] &= ev[:, 1:] != ev[:, :-1]
        distinct[:, :-1] &= ev[:, :-1] != ev[:, 1:]
        assert torch.equal(idx[:, l, :k][distinct], ei[distinct]), l
        assert not vals[:, l, k:].any() and not idx[:, l, k:].any()    # zero fill beyond k
        for i in range(N):
            assert idx[i, l, :k].unique().numel() == k


def test_topk_sorted_reads_fused_head_layout_in_place():
    """``inner=A``: the row is the first A of C channels per pixel (the objectness logits inside
    the fused RPN output ``[N, H, W, 5A]``) — same result as top-k of the gathered copy."""
    g = torch.Generator().manual_seed(4)
    fused = [torch.randn(2, h, w, 15, generator=g).to(DEV) for h, w in [(50, 84), (13, 21), (4, 5)]]
    ks = [2000, 2000, 2000]
    vals, idx = BF.topk_sorted(fused, [min(k, f[0].numel() // 5) for k, f in zip(ks, fused)], 2000,
                               inner=3)
    for l, f in enumerate(fused):
        r = f[..., :3].reshape(2, -1)
        k = min(2000, r.shape[1])
        ev, ei = r.topk(k, dim=1)
        assert torch.equal(vals[:, l, :k], ev) and torch.equal(idx[:, l, :k], ei)
        assert not vals[:, l, k:].any()


@pytest.mark.parametrize('case', ['ties', 'negzero', 'final', 'tiny'])
def test_topk_sorted_edge_cases(case):
    g = torch.Generator().manual_seed(1)
    if case == 'ties':        # heavily quantised values: thousands equal to the threshold
        r = (torch.randint(0, 7, (3, 30000), generator=g).float() - 3.0).to(DEV)
        k, kmax = 2000, 2048
    elif case == 'negzero':   # sign handling of the key transform: -0.0 < +0.0 in key order only
        r = torch.cat([torch.randn(2, 5000, generator=g), torch.zeros(2, 10), -torch.zeros(2, 10),
                       torch.full((2, 3), float('-inf')), torch.full((2, 2), float('inf'))], 1).to(DEV)
        k, kmax = 4096, 4096
    elif case == 'final':     # the post-NMS selection: -1 marks invalid slots
        r = torch.rand(2, 10000, generator=g)
        r[:, ::3] = -1.0
        r = r.to(DEV)
        k, kmax = 2000, 2000
    else:                     # fewer elements than k; k == 1; a single element
        r = torch.randn(4, 5, generator=g).to(DEV)
        k, kmax = 16, 16
    vals, idx = BF.topk_sorted([r.contiguous()], [k], kmax)
    kk = min(k, r.shape[1])
    ev, _ = r.topk(kk, dim=1)
    assert torch.equal(vals[:, 0, :kk], ev)
    assert torch.equal(r.gather(1, idx[:, 0, :kk]), ev)
    for i in range(r.shape[0]):
        assert idx[i, 0, :kk].unique().numel() == kk
    assert not vals[:, 0, kk:].any()
    if case == 'tiny':
        v1, i1 = BF.topk_sorted([r.contiguous()], [1], 4)
        assert torch.equal(v1[:, 0, 0], r.max(dim=1).values) and torch.equal(i1[:, 0, 0], r.argmax(dim=1))


@pytest.mark.parametrize('case', [
    # N, H, W, Cin, Cout, R, stride, pad, relu, res, halo
    (2, 20, 24, 64, 96, 1, 1, 0, True, False, False),
    (1, 17, 23, 128, 200, 1, 1, 0, True, True, False),
    (2, 31, 45, 32, 64, 3, 2, 1, True, False, False),
    (1, 64, 96, 4, 64, 7, 2, 3, True, False, False),
    (2, 13, 21, 512, 128, 1, 1, 0, False, True, False),
    (2, 13, 21, 64, 256, 3, 1, 1, True, False, True),
    (1, 19, 37, 128, 64, 3, 1, 1, False, False, True),
    (2, 50, 84, 32, 64, 3, 1, 1, True, False, True),
], ids=lambda c: 'x'.join(str(v) for v in c[:8]))
def test_conv_bf16_mode_is_bf16_rounded_operands_with_fp32_accumulate(monkeypatch, case):
    """conv_math = 'bf16' (cfg[4], planes = 1 of the bf16x6 kernels: the 64x64 DMA ring, the halo
    kernel, split-K): the result is the convolution of the bf16-ROUNDED operands accumulated in fp32 —
    checked against an fp64 convolution of the rounded operands (so the only difference left is fp32
    summation order), and the ring kernel == the register-staged kernel bit for bit."""
    N, H, W, Cin, Cout, R, stride, pad, relu, with_res, halo = case
    rs = np.random.RandomState(H * 13 + Cout)
    x = rs.standard_normal((N, H, W, Cin)).astype(np.float32)
    w = (rs.standard_normal((Cout, R, R, Cin)) / (R * R * Cin) ** 0.5).astype(np.float32)
    b = rs.standard_normal(Cout).astype(np.float32)
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    res = rs.standard_normal((N, Ho, Wo, Cout)).astype(np.float32) if with_res else None
    xr = torch.from_numpy(x).bfloat16().double()
    wr = torch.from_numpy(w).bfloat16().double()
    exp = F.conv2d(xr.permute(0, 3, 1, 2), wr.permute(0, 3, 1, 2), torch.from_numpy(b).double(),
                   stride=stride, padding=pad).permute(0, 2, 3, 1)
    if res is not None:
        exp = exp + torch.from_numpy(res).double()
    if relu:
        exp = exp.clamp(min=0)
    scale = float(F.conv2d(xr.abs().permute(0, 3, 1, 2), wr.abs().permute(0, 3, 1, 2), stride=stride,
                           padding=pad).max())
    prev = BF.set_conv_math('bf16')
    monkeypatch.setenv('BGS_CONV_HALO', '1' if halo else '0')
    try:
        for sk in (1, 2):
            if halo:
                BF.conv_bfx_tuning(halo_splits=sk)
            else:
                BF.conv_bfx_tuning(11, sk)
            got = BF.conv2d_nhwc(dev(x), dev(w), dev(b), stride=stride, pad=pad, relu=relu,
                                 residual=None if res is None else dev(res))
            used = BF.conv_bfx_last_launch()
            assert float((got.cpu().double() - exp).abs().max()) <= 2e-6 * scale, used
            if halo:
                assert used['halo_variant'] == 4 and used['halo_splits'] == sk
                BF.conv_bfx_tuning(halo_splits=sk, halo_variant=2)
            else:
                assert used['tile'] == 11 | 0x200 and used['splits'] == sk, used     # the DMA ring ran
                BF.conv_bfx_tuning(11 | 0x100, sk)
            old = BF.conv2d_nhwc(dev(x), dev(w), dev(b), stride=stride, pad=pad, relu=relu,
                                 residual=None if res is None else dev(res))
            assert torch.equal(old, got)
    finally:
        BF.conv_bfx_tuning()
        BF.set_conv_math(prev)


@pytest.mark.parametrize('case', [
    # N, H, W, Cin, Cout, R, stride, pad, relu, res_mode, splitk
    (2, 40, 56, 256, 320, 1, 1, 0, True, 1, -1),
    (2, 67, 95, 256, 256, 1, 2, 0, False, 0, -1),
    (2, 91, 123, 128, 272, 3, 2, 1, True, 0, -1),
    (2, 40, 56, 512, 256, 1, 1, 0, True, 2, 3),
    (1, 46, 50, 256, 1236, 1, 1, 0, False, 0, -1),
], ids=lambda c: 'x'.join(str(v) for v in c))
def test_conv_bf16_mode_large_layers_run_the_8_wave_128x128_ring(monkeypatch, case):
    """bf16 mode, M >= 2048 / Cout >= 256 / K >= 256: the 8-wave 128 x 128 ring kernel (ragged M and Cout
    tiles, stride 2, 3x3 with padding, residual modes, split-K, its two-half LDS epilogue) == the 64 x 64
    ring bit for bit (same products, same K order)."""
    N, H, W, Cin, Cout, R, stride, pad, relu, rm, sk = case
    rs = np.random.RandomState(H + Cout)
    x = rs.standard_normal((N, H, W, Cin)).astype(np.float32)
    w = (rs.standard_normal((Cout, R, R, Cin)) / (R * R * Cin) ** 0.5).astype(np.float32)
    b = rs.standard_normal(Cout).astype(np.float32)
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    res = None
    if rm == 1:
        res = rs.standard_normal((N, Ho, Wo, Cout)).astype(np.float32)
    elif rm == 2:
        res = rs.standard_normal((N, Ho // 2, Wo // 2, Cout)).astype(np.float32)
    prev = BF.set_conv_math('bf16')
    monkeypatch.setenv('BGS_CONV_HALO', '0')
    try:
        BF.conv_bfx_tuning(0, sk)
        got = BF.conv2d_nhwc(dev(x), dev(w), dev(b), stride=stride, pad=pad, relu=relu,
                             residual=None if res is None else dev(res), residual_mode=rm)
        used = BF.conv_bfx_last_launch()
        assert used['tile'] == 22 | 0x200, used                     # 128 x 128, DMA ring
        if sk > 0:
            assert used['splits'] == sk
        BF.conv_bfx_tuning(11, sk)
        ref = BF.conv2d_nhwc(dev(x), dev(w), dev(b), stride=stride, pad=pad, relu=relu,
                             residual=None if res is None else dev(res), residual_mode=rm)
        assert BF.conv_bfx_last_launch()['tile'] == 11 | 0x200
        assert torch.equal(got, ref)
    finally:
        BF.conv_bfx_tuning()
        BF.set_conv_math(prev)


@pytest.fixture
def planes_kernel_off():
    """The dispatch-asserting tests of the kernels BEHIND conv1x1_planes_bfx_kernel / conv3x3_planes_bfx_kernel in the
    launch order of the bf16x6 entry points: the planes kernels (round 6; first choice for the layers they take) are
    switched off for the test."""
    from balancedgroupsoftmax_amd import capi
    lib = capi.load()
    lib.bgs_conv1x1_planes_enable(0)
    lib.bgs_conv3x3_planes_enable(0)
    yield
    lib.bgs_conv1x1_planes_enable(-1)
    lib.bgs_conv3x3_planes_enable(-1)


@pytest.mark.usefixtures('planes_kernel_off')
@pytest.mark.parametrize('shape', [
    # (N, H, W, Cin, Cout, stride, residual mode)
    (1, 67, 75, 256, 256, 1, 0),      # M = 5025: last tile has 1 row
    (2, 50, 84, 256, 256, 1, 2),      # fpn lateral with the nearest-2x-upsampled top-down add
    (1, 80, 96, 64, 256, 1, 1),       # layer1 conv3: K = 64, residual + ReLU
    (2, 50, 84, 128, 512, 1, 1),      # layer2 conv3: K = 128, two 256-channel slabs
    (1, 100, 168, 256, 512, 2, 0),    # layer2 projection shortcut: stride 2
    (2, 50, 84, 256, 1024, 1, 1),     # layer3 conv3: four slabs
])
def test_conv1x1_filter_resident_kernel_is_bit_identical_to_the_operand_ring(shape):
    """``conv1x1_bres_kernel`` (csrc/conv1x1_bres.hip: split filter resident in registers, activation
    tiles DMA'd once) against the 64 x 64 operand ring it replaces on the short-reduction 1x1 layers:
    BIT-IDENTICAL outputs (same products, same accumulation order), and both against torch-CPU
    ``F.conv2d`` incl. bias / residual / ReLU; dispatch asserted through the last-launch query."""
    import torch.nn.functional as F
    from balancedgroupsoftmax_amd import capi
    lib = capi.load()
    N, H, W, Cin, Cout, stride, rmode = shape
    g = torch.Generator().manual_seed(Cin * 7 + Cout + stride)
    x = torch.randn(N, H, W, Cin, generator=g)
    w = torch.randn(Cout, 1, 1, Cin, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = None
    if rmode == 1:
        res = torch.randn(N, Ho, Wo, Cout, generator=g)
    elif rmode == 2:
        res = torch.randn(N, Ho // 2, Wo // 2, Cout, generator=g)
    prev = BF.set_conv_math('bf16x6')
    try:
        outs = []
        for on in (2, 0):
            lib.bgs_conv1x1_bres_enable(on)
            y = BF.conv2d_nhwc(dev(x), dev(w), dev(b), stride=stride, relu=True,
                               residual=None if res is None else dev(res), residual_mode=rmode)
            assert lib.bgs_conv1x1_bres_last_launch() == (1 if on else 0)
            outs.append(y.cpu())
        assert torch.equal(outs[0], outs[1])
    finally:
        lib.bgs_conv1x1_bres_enable(0)
        BF.set_conv_math(prev)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), stride=stride)
    if rmode == 1:
        ref = ref + res.permute(0, 3, 1, 2).double()
    elif rmode == 2:
        ref = ref + F.interpolate(res.permute(0, 3, 1, 2).double(), scale_factor=2, mode='nearest')
    ref = torch.relu(ref).permute(0, 2, 3, 1)
    err = float((outs[0].double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, err


@pytest.mark.usefixtures('planes_kernel_off')
def test_conv1x1_filter_resident_kernel_as_data_gradient_with_mask():
    """The same kernel under ``bgs_conv2d_dgrad_nhwc_f32_bfx_ws`` (1x1, stride 1: the data gradient of
    a 1x1 conv is a 1x1 conv with the transposed filter) with the ReLU-backward mask and a residual
    gradient in the epilogue: bit-identical to the ring kernel."""
    from balancedgroupsoftmax_amd import capi
    lib = capi.load()
    g = torch.Generator().manual_seed(5)
    N, H, W, Cin, Cout = 2, 50, 84, 1024, 256          # dgrad: K = Cout = 256 -> Cin = 1024 channels
    dy = torch.randn(N, H, W, Cout, generator=g)
    w = torch.randn(Cout, 1, 1, Cin, generator=g) * 0.05
    res = torch.randn(N, H, W, Cin, generator=g)
    mask = torch.randn(N, H, W, Cin, generator=g)
    prev = BF.set_conv_math('bf16x6')
    try:
        outs = []
        for on in (2, 0):
            lib.bgs_conv1x1_bres_enable(on)
            dx = BF.conv2d_dgrad_nhwc(dev(dy), dev(w), (H, W), residual=dev(res), mask=dev(mask))
            assert lib.bgs_conv1x1_bres_last_launch() == (1 if on else 0)
            outs.append(dx.cpu())
        assert torch.equal(outs[0], outs[1])
    finally:
        lib.bgs_conv1x1_bres_enable(0)
        BF.set_conv_math(prev)


@pytest.mark.usefixtures('planes_kernel_off')
@pytest.mark.parametrize('shape', [
    # (N, H, W, Cin, Cout, pixel tile the fewest-tiles mode takes: 0 = 8 x 16, 1 = 10 x 12, 2 = 5 x 21)
    (2, 50, 84, 256, 256, 1),      # stride-16 maps (layer3 conv2, P4): 35 tiles per image instead of 42
    (2, 25, 42, 512, 512, 2),      # stride-32 maps (layer4 conv2, P5): 10 instead of 12
    (1, 37, 51, 64, 64, 0),        # ragged edges on both axes, NB = 1 (20 tiles either way: stays 8 x 16)
    (2, 200, 336, 32, 128, 0),     # stride-4 map: stays 8 x 16
    (2, 100, 168, 32, 128, 0),     # stride-8 map: 140 vs 143 tiles is inside the 5 % margin: stays 8 x 16
], ids=lambda c: 'x'.join(str(v) for v in c))
def test_halo_kernel_pixel_tile_geometries_are_bit_identical(shape, monkeypatch):
    """``conv3x3_halo_bfx4_kernel<.., GTH, GTW>``: the 10 x 12 and 5 x 21 pixel tiles (the small maps of
    mmdet/models/backbones/resnet.py:220-266 / necks/fpn.py:131-134, where the 8 x 16 tile wastes up to 28 % of its
    MFMA rows on pixels outside the map) against the 8 x 16 tile: every output is the same sum in the same order —
    BIT-IDENTICAL, unsliced and with the channel chunks sliced two ways; the fewest-tiles choice per map size
    and the default (8 x 16: the tighter tiles measured flat in the cfg[1] step) are asserted through the
    last-launch query; bias + ReLU epilogue; fp64 bound."""
    N, H, W, Cin, Cout, auto = shape
    g = torch.Generator().manual_seed(H * 7 + W)
    x = torch.randn(N, H, W, Cin, generator=g)
    w = torch.randn(Cout, 3, 3, Cin, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, generator=g)
    monkeypatch.setenv('BGS_CONV_HALO', '1')
    prev = BF.set_conv_math('bf16x6')
    try:
        for splits in (1, 2):
            outs = {}
            for geom in (0, 1, 2):
                BF.conv_bfx_tuning(halo_splits=splits, halo_geom=geom, halo_wide=0)    # (the 128-pixel kernel's tiles)
                outs[geom] = BF.conv2d_nhwc(dev(x), dev(w), dev(b), pad=1, relu=True).cpu()
                used = BF.conv_bfx_last_launch()
                assert used['halo_variant'] == 4 and used['halo_geom'] == geom and used['halo_splits'] == splits, used
            assert torch.equal(outs[1], outs[0]) and torch.equal(outs[2], outs[0])
        BF.conv_bfx_tuning(halo_geom=3)                     # "fewest tiles per image"
        BF.conv2d_nhwc(dev(x), dev(w), dev(b), pad=1, relu=True)
        assert BF.conv_bfx_last_launch()['halo_geom'] == auto
        BF.conv_bfx_tuning()                                # the default stays 8 x 16 (measured flat in the step)
        BF.conv2d_nhwc(dev(x), dev(w), dev(b), pad=1, relu=True)
        assert BF.conv_bfx_last_launch()['halo_geom'] == 0
    finally:
        BF.conv_bfx_tuning()
        BF.set_conv_math(prev)
    ref = torch.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), padding=1))
    err = float((outs[0].double() - ref.permute(0, 2, 3, 1)).abs().max() / ref.abs().max())
    assert err < 2e-6, err


def _wide_last():
    from balancedgroupsoftmax_amd import capi
    v = capi.load().bgs_conv_bfx_wide_last_launch()
    return dict(ran=v & 1, nst=(v >> 4) & 15, splits=(v >> 8) & 0xfff, nbw=2 if v & 0x100000 else 4)


@pytest.mark.usefixtures('planes_kernel_off')
@pytest.mark.parametrize('nbw', [4, 2])
@pytest.mark.parametrize('nst', [2, 3])
@pytest.mark.parametrize('shape', [
    # (N, H, W, Cin, Cout, stride, relu, residual mode)
    (2, 50, 84, 256, 1024, 1, True, 1),       # layer3 conv3: residual + ReLU, 528 tiles
    (1, 37, 29, 64, 256, 1, False, 0),        # M = 1073: the last row tile is clamped (49 live rows), K = 64
    (2, 100, 168, 256, 512, 2, False, 0),     # layer2 projection shortcut: stride 2
    (2, 50, 84, 512, 256, 1, False, 2),       # FPN lateral: nearest-2x-upsampled top-down add
    (256, 1, 1, 1024, 1236, 1, False, 0),     # fc_cls: Cout % 128 = 84: clamped filter rows
    (3, 17, 23, 128, 132, 1, True, 1),        # both edges ragged
], ids=lambda c: 'x'.join(str(int(v)) for v in c))
def test_bfx_wide_tile_kernel_is_bit_identical_to_the_operand_ring(shape, nst, nbw):
    """``conv1x1_bfx_wide_kernel`` (csrc/conv_bfx_wide.hip: 128 x 128 or 128 x 64 tile, four M-stacked waves, wave-private A,
    rows / columns past the edge clamped instead of zero-filled) in both ring depths against the 64 x 64 operand
    ring on the 1x1 layers of mmdet/models/backbones/resnet.py:220-266, necks/fpn.py:101-141 and the FC heads:
    BIT-IDENTICAL outputs when K is not sliced (same products, same order), both within fp32 rounding of fp64
    torch; sliced K within the same bound; the dispatch asserted through the last-launch query."""
    from balancedgroupsoftmax_amd import capi
    lib = capi.load()
    N, H, W, Cin, Cout, stride, relu, rm = shape
    g = torch.Generator().manual_seed(Cin * 3 + Cout + stride)
    x = torch.randn(N, H, W, Cin, generator=g)
    w = torch.randn(Cout, 1, 1, Cin, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = None
    if rm == 1:
        res = torch.randn(N, Ho, Wo, Cout, generator=g)
    elif rm == 2:
        res = torch.randn(N, Ho // 2, Wo // 2, Cout, generator=g)
    kw = dict(stride=stride, relu=relu, residual=None if res is None else dev(res), residual_mode=rm)
    prev = BF.set_conv_math('bf16x6')
    try:
        lib.bgs_conv_bfx_wide_tuning(0, 0, -1)
        BF.conv_bfx_tuning(0, 1)                       # the ring unsliced (its plan slices K for the small grids)
        ring = BF.conv2d_nhwc(dev(x), dev(w), dev(b), **kw).cpu()
        BF.conv_bfx_tuning(0, -1)
        assert not _wide_last()['ran']
        lib.bgs_conv_bfx_wide_tuning(2, nst | (nbw << 4), 1)   # bits 4..7: tile width (2: 128 x 64)
        wide = BF.conv2d_nhwc(dev(x), dev(w), dev(b), **kw).cpu()
        assert _wide_last() == dict(ran=1, nst=nst, splits=1, nbw=nbw)
        assert torch.equal(wide, ring)
        lib.bgs_conv_bfx_wide_tuning(2, nst | (nbw << 4), 2)   # K sliced two ways through the slab epilogue
        sliced = BF.conv2d_nhwc(dev(x), dev(w), dev(b), **kw).cpu()
        assert _wide_last()['ran'] and _wide_last()['splits'] == 2
    finally:
        lib.bgs_conv_bfx_wide_tuning(1, 0, -1)
        BF.conv_bfx_tuning()
        BF.set_conv_math(prev)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), stride=stride)
    if rm == 1:
        ref = ref + res.permute(0, 3, 1, 2).double()
    elif rm == 2:
        ref = ref + F.interpolate(res.permute(0, 3, 1, 2).double(), scale_factor=2, mode='nearest')
    if relu:
        ref = torch.relu(ref)
    ref = ref.permute(0, 2, 3, 1)
    for got in (wide, sliced):
        err = float((got.double() - ref).abs().max() / ref.abs().max())
        assert err < 2e-6, err


@pytest.mark.usefixtures('planes_kernel_off')
def test_bfx_wide_tile_kernel_as_data_gradient_with_mask_and_in_the_automatic_mode():
    """(i) the wide kernel under ``bgs_conv2d_dgrad_nhwc_f32_bfx_ws`` (the data gradient of a 1x1 conv is a 1x1 conv
    with the transposed filter) with the ReLU-backward mask and a residual gradient in the epilogue == the ring,
    bit for bit; (ii) the automatic mode takes the large-grid K >= 256 layers (fpn.lat0-sized here) and leaves the
    528-tile K = 256 layer and the K = 64 layers to the ring."""
    from balancedgroupsoftmax_amd import capi
    lib = capi.load()
    g = torch.Generator().manual_seed(11)
    N, H, W, Cin, Cout = 2, 50, 84, 1024, 256          # dgrad: K = Cout = 256 -> Cin = 1024 channels
    dy = torch.randn(N, H, W, Cout, generator=g)
    w = torch.randn(Cout, 1, 1, Cin, generator=g) * 0.05
    res = torch.randn(N, H, W, Cin, generator=g)
    mask = torch.randn(N, H, W, Cin, generator=g)
    prev = BF.set_conv_math('bf16x6')
    try:
        outs = []
        for mode in (2, 0):
            lib.bgs_conv_bfx_wide_tuning(mode, 0, 1)
            lib.bgs_conv1x1_bres_enable(0)
            BF.conv_bfx_tuning(0, -1 if mode else 1)
            dx = BF.conv2d_dgrad_nhwc(dev(dy), dev(w), (H, W), residual=dev(res), mask=dev(mask))
            BF.conv_bfx_tuning(0, -1)
            assert _wide_last()['ran'] == (1 if mode else 0)
            outs.append(dx.cpu())
        assert torch.equal(outs[0], outs[1])
        lib.bgs_conv_bfx_wide_tuning(1, 0, -1)
        lib.bgs_conv1x1_bres_enable(0)
        took = {}
        for name, (n, h, wd, ci, co) in dict(lat0=(2, 200, 336, 256, 256), l3c3=(2, 50, 84, 256, 1024),
                                             l1c3=(2, 100, 168, 64, 256), lat1=(2, 100, 168, 512, 256)).items():
            BF.conv2d_nhwc(torch.randn(n, h, wd, ci, device=DEV), torch.randn(co, 1, 1, ci, device=DEV),
                           torch.zeros(co, device=DEV))
            u = _wide_last()
            took[name] = (u['ran'], u['nbw'] if u['ran'] else 0)
        # 128 x 128 on the large grids with K >= 256; 128 x 64 (three stages) where >= 1000 of them remain with
        # K >= 256; the 64 x 64 ring on K = 64
        assert took == dict(lat0=(1, 4), l3c3=(1, 2), l1c3=(0, 0), lat1=(1, 4)), took
    finally:
        lib.bgs_conv_bfx_wide_tuning(1, 0, -1)
        lib.bgs_conv1x1_bres_enable(1)
        BF.conv_bfx_tuning()
        BF.set_conv_math(prev)


@pytest.mark.parametrize('case', [
    # name, N, H, W, Cin, Cout, k, stride, pad
    ('3x3s1', 2, 40, 56, 128, 128, 3, 1, 1),
    ('3x3s2', 1, 41, 55, 64, 256, 3, 2, 1),
    ('1x1', 2, 50, 84, 256, 512, 1, 1, 0),
    ('fc_cls', 1, 1, 1024, 1024, 1236, 1, 1, 0),
    ('ragged', 1, 19, 23, 100, 132, 3, 1, 1),      # Cout, K and M not multiples of the tile
], ids=lambda c: c[0])
def test_wgrad_bf16x6_kernel_vs_fp64_and_not_worse_than_the_fp32_mfma_kernel(case):
    """``conv_wgrad_bfx_kernel`` (weight gradient on the bf16 matrix cores, both operands split
    exactly into three bf16 planes after an in-register transpose): dw / db against fp64 torch
    autograd; its error is bounded by 1.5x the fp32-MFMA wgrad kernel's own on the same inputs
    (it is an fp32-faithful mode, not a reduced-precision one); bitwise reproducible; the bf16 mode
    (planes = 1) equals the fp64 product of bf16-ROUNDED operands; accumulate adds."""
    from balancedgroupsoftmax_amd import capi
    lib = capi.load()
    name, N, H, W, Cin, Cout, k, stride, pad = case
    rs = np.random.RandomState(len(name) * 31 + Cout)
    x = (rs.randn(N, H, W, Cin) * np.exp(rs.uniform(-3, 3, size=(N, H, W, 1)))).astype(np.float32)
    w = (rs.randn(Cout, Cin, k, k) / np.sqrt(Cin * k * k)).astype(np.float32)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    dy = (rs.randn(N, Ho, Wo, Cout) * np.exp(rs.uniform(-3, 3, size=(N, Ho, Wo, 1)))).astype(np.float32)
    _, edw = _torch_conv_grads(x, w, dy, stride, pad)
    edb = dy.astype(np.float64).sum((0, 1, 2))
    scale = np.abs(edw).max()
    prev = BF.set_conv_math('bf16x6')
    try:
        # 2 = the bf16x6 kernel on every eligible layer (the default routes reductions of <= 1536 rows,
        # the FC heads, to the fp32-MFMA kernel, which is faster there)
        lib.bgs_conv2d_wgrad_bfx_enable(2)
        BF.launch_census(reset=True)
        dw, db = BF.conv2d_wgrad_nhwc(dev(x), dev(dy), k, stride=stride, pad=pad, bias=True)
        assert BF.launch_census()['wgrad_bfx'] == 1
        dw2 = BF.conv2d_wgrad_nhwc(dev(x), dev(dy), k, stride=stride, pad=pad)
        assert torch.equal(dw, dw2)                                  # fixed-order reduction
        err_bfx = np.abs(dw.cpu().numpy() - edw).max() / scale
        assert np.abs(db.cpu().numpy() - edb).max() <= 2e-5 * max(1.0, np.abs(edb).max())
        lib.bgs_conv2d_wgrad_bfx_enable(0)
        dwf = BF.conv2d_wgrad_nhwc(dev(x), dev(dy), k, stride=stride, pad=pad)
        lib.bgs_conv2d_wgrad_bfx_enable(2)
        err_f32 = np.abs(dwf.cpu().numpy() - edw).max() / scale
        print('%s: wgrad error vs fp64: bf16x6 %.2e, fp32 MFMA %.2e' % (name, err_bfx, err_f32))
        assert err_bfx <= max(1.5 * err_f32, 2e-7) and err_bfx < 2e-5
        BF.conv2d_wgrad_nhwc(dev(x), dev(dy), k, stride=stride, pad=pad, bias=True, dw=dw, db=db,
                             accumulate=True)
        assert np.abs(dw.cpu().numpy() - 2 * edw).max() <= 4e-5 * scale
        BF.set_conv_math('bf16')
        dwb = BF.conv2d_wgrad_nhwc(dev(x), dev(dy), k, stride=stride, pad=pad)
        rx = torch.from_numpy(x).bfloat16().float().numpy()
        rdy = torch.from_numpy(dy).bfloat16().float().numpy()
        _, edwb = _torch_conv_grads(rx, w, rdy, stride, pad)
        assert np.abs(dwb.cpu().numpy() - edwb).max() <= 2e-5 * np.abs(edwb).max()
    finally:
        lib.bgs_conv2d_wgrad_bfx_enable(1)
        BF.set_conv_math(prev)


@pytest.mark.parametrize('case', [
    # Cout, Cin, k, with BN, conv bias, cin padded to
    (64, 3, 7, True, False, 4),        # stem
    (128, 256, 1, True, False, None),
    (256, 256, 3, True, False, None),
    (512, 512, 3, True, False, None),  # largest filter row: 4608 floats
    (256, 256, 3, False, True, None),  # FPN / RPN conv: no BN, a bias
    (12, 256, 1, False, True, None),
    (1024, 16, 3, True, False, None),  # ResNeXt grouped conv2 (Cin / groups)
])
def test_fused_bn_fold_forward_and_backward_vs_torch_autograd(case):
    """``bgs_fold_conv_bn_fwd`` / ``_bwd`` (csrc/bn_fold.hip: eval-mode BN folded into the filter, one
    launch each way) against torch autograd of the tensor-op formula it replaces
    (w * gamma / sqrt(var + eps) permuted to KRSC, beta - mean * scale): folded filter / bias exactly
    equal up to 1 ulp, gradients of w, gamma, beta and the conv bias to fp32 rounding."""
    Cout, Cin, k, with_bn, with_cb, cinp = case
    g = torch.Generator().manual_seed(Cout + Cin + k)
    w = torch.randn(Cout, Cin, k, k, generator=g)
    cb = torch.randn(Cout, generator=g) if with_cb else None
    gamma, beta = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    mean, var = torch.randn(Cout, generator=g), torch.rand(Cout, generator=g) + 0.1
    eps = 1e-5
    leaves = [t.clone().double().requires_grad_(True) for t in (w, gamma, beta)] + \
        ([cb.clone().double().requires_grad_(True)] if with_cb else [])
    wd, gd, bd = leaves[:3]
    if with_bn:
        scale = gd / torch.sqrt(var.double() + eps)
        wf_ref = (wd * scale.view(-1, 1, 1, 1)).permute(0, 2, 3, 1)
        bf_ref = bd - mean.double() * scale + (leaves[3] * scale if with_cb else 0)
    else:
        wf_ref = wd.permute(0, 2, 3, 1)
        bf_ref = leaves[3] if with_cb else torch.zeros(Cout, dtype=torch.float64)
    if cinp:
        wf_ref = torch.nn.functional.pad(wf_ref, (0, cinp - Cin))
    gw = torch.randn(wf_ref.shape, generator=g).double()
    gb = torch.randn(Cout, generator=g).double()
    ((wf_ref * gw).sum() + (bf_ref * gb).sum()).backward()
    params = [dev(t).requires_grad_(True) for t in (w, gamma, beta)]
    cbd = dev(cb).requires_grad_(True) if with_cb else None
    if with_bn:
        wf, bf = BF.fold_conv_bn(params[0], cbd, params[1], params[2], dev(mean), dev(var), eps,
                                 cin_padded=cinp)
    else:
        wf, bf = BF.fold_conv_bn(params[0], cbd, cin_padded=cinp)
    assert tuple(wf.shape) == tuple(wf_ref.shape) and wf.is_contiguous()
    np.testing.assert_allclose(wf.detach().cpu().numpy(), wf_ref.detach().numpy(), rtol=3e-7, atol=1e-30)
    np.testing.assert_allclose(bf.detach().cpu().numpy(), bf_ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    ((wf * dev(gw.float())).sum() + (bf * dev(gb.float())).sum()).backward()
    np.testing.assert_allclose(params[0].grad.cpu().numpy(), wd.grad.numpy(), rtol=1e-5, atol=1e-6)
    if with_bn:
        tol = 1e-4 * max(1.0, float(gd.grad.abs().max()))
        assert float((params[1].grad.cpu().double() - gd.grad).abs().max()) < tol
        np.testing.assert_allclose(params[2].grad.cpu().numpy(), bd.grad.numpy(), rtol=1e-5, atol=1e-6)
    else:
        assert params[1].grad is None and params[2].grad is None
    if with_cb:
        np.testing.assert_allclose(cbd.grad.cpu().numpy(), leaves[3].grad.numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('shape', [(2, 40, 56, 128, 128), (1, 50, 84, 256, 256), (2, 25, 42, 512, 512)])
def test_conv3x3_data_gradient_runs_the_halo_kernel_and_equals_the_operand_ring(shape, monkeypatch):
    """3x3 / stride 1 / pad 1 data gradients (bottleneck conv2, FPN outputs, RPN convs) go through the
    halo-resident forward kernel on the flipped, transposed filter, with the ReLU-backward mask in
    its epilogue: == the general operand-ring path (same products; summation order differs between
    the two kernels) and == fp64 torch autograd."""
    N, H, W, Cin, Cout = shape
    rs = np.random.RandomState(Cin + H)
    x = rs.randn(N, H, W, Cin).astype(np.float32)
    w = (rs.randn(Cout, Cin, 3, 3) / np.sqrt(Cin * 9)).astype(np.float32)
    dy = rs.randn(N, H, W, Cout).astype(np.float32)
    edx, _ = _torch_conv_grads(x, w, dy, 1, 1)
    wk = dev(krsc(w))
    prev = BF.set_conv_math('bf16x6')
    try:
        BF.launch_census(reset=True)
        a = BF.conv2d_dgrad_nhwc(dev(dy), wk, (H, W), 1, 1, mask=dev(x))
        assert BF.launch_census()['halo_bfx4'] == 1
        monkeypatch.setenv('BGS_DGRAD_HALO', '0')
        BF.launch_census(reset=True)
        b = BF.conv2d_dgrad_nhwc(dev(dy), wk, (H, W), 1, 1, mask=dev(x))
        assert BF.launch_census()['halo_bfx4'] == 0
    finally:
        BF.set_conv_math(prev)
    exp = np.where(x > 0, edx, 0)
    scale = np.abs(exp).max()
    assert np.abs(a.cpu().numpy() - exp).max() <= 2e-5 * scale
    assert np.abs(a.cpu().numpy() - b.cpu().numpy()).max() <= 2e-6 * scale


@pytest.mark.parametrize('shape', [(256, 3, 3, 128), (512, 1, 1, 256), (15, 1, 1, 256), (64, 3, 3, 20), (100, 7, 7, 4)])
def test_fused_dgrad_filter_split_equals_flip_permute_split(shape):
    """``bgs_conv_bfx_split_weights_dgrad``: the split planes of the data-gradient filter in one launch
    are byte-identical to flip + permute + contiguous + ``bgs_conv_bfx_split_weights``."""
    Cout, R, S, Cin = shape
    g = torch.Generator().manual_seed(Cout * 7 + Cin)
    w = (torch.randn(Cout, R, S, Cin, generator=g) * torch.exp(torch.rand(Cout, 1, 1, 1, generator=g) * 8 - 4)).to(DEV)
    a = BF.bfx_split_weights_dgrad(w, cache=False)
    b = BF.bfx_split_weights(BF.dgrad_filter(w).view(Cin, R * S * Cout), cache=False)
    assert a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize('clip', [35.0, 1e9, None])
def test_fused_clip_sgd_step_vs_torch_clip_grad_norm_and_sgd(clip):
    """``bgs_sgd_clip_step`` (csrc/optim.hip) against ``torch.nn.utils.clip_grad_norm_`` +
    ``torch.optim.SGD.step`` (the reference's DistOptimizerHook tail, dist_utils.py:55-58) over three
    steps: 70 tensors (two launches per phase), sizes around the 16K chunk and the 16-byte vector width,
    clipping active (max_norm 35 against a norm of ~1e3), inactive (1e9) and absent; a parameter without
    a gradient is skipped; the Fp16 hook's unscale (grad_scale = 1 / 512) rides in the same pass."""
    from balancedgroupsoftmax_amd import train
    g = torch.Generator().manual_seed(11)
    sizes = [(3,), (17,), (16384,), (16385,), (70000,), (257, 129), (64, 3, 7, 7), (1236, 1024)] + [(50 + i,) for i in range(62)]
    init = [torch.randn(*sz, generator=g) for sz in sizes]
    grads = [[torch.randn(*sz, generator=g) * 3.0 for sz in sizes] for _ in range(3)]
    grad_clip = None if clip is None else dict(max_norm=clip, norm_type=2)

    def run(fused, scale):
        ps = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
        extra = torch.nn.Parameter(torch.ones(5, device=DEV))          # never gets a gradient
        opt = torch.optim.SGD(ps + [extra], lr=0.02, momentum=0.9, weight_decay=1e-4)
        norms = []
        for step in range(3):
            for p, gr in zip(ps, grads[step]):
                p.grad = (gr * scale).to(DEV)
            if fused:
                f = train.FusedClipSGD(opt, ps + [extra], grad_clip) if step == 0 else f
                f.step(1.0 / scale)
                norms.append(float(f.total_norm))
            else:
                if scale != 1.0:
                    torch._foreach_div_([p.grad for p in ps], scale)
                if grad_clip is not None:
                    norms.append(float(torch.nn.utils.clip_grad_norm_(ps, clip, 2)))
                opt.step()
        return ps, [opt.state[p]['momentum_buffer'] for p in ps], norms, extra

    for scale in (1.0, 512.0):
        pa, ma, na, ea = run(True, scale)
        pb, mb, nb, eb = run(False, scale)
        assert torch.equal(ea, eb)
        for a, b in zip(pa, pb):
            assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), a.shape
        for a, b in zip(ma, mb):
            assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), a.shape
        for a, b in zip(pa, pb):       # the clipped gradient is written back, as the hook clips in place
            assert float((a.grad - b.grad).abs().max()) <= 2e-6 * float(b.grad.abs().max())
        if grad_clip is not None:
            assert np.allclose(na, nb, rtol=1e-5), (na, nb)


def test_fused_clip_sgd_step_invalidates_the_version_keyed_weight_caches():
    """The fused step writes the parameters through raw pointers; every cache keyed on ``Parameter._version`` — the
    split bf16 planes of a frozen-looking weight (``functional.bfx_split_weights``), the folded conv+BN weights of an
    eval-mode module (``backbone._FoldCache``) — must see the update as it would after ``torch.optim.SGD.step()``:
    eval, fused step, eval gives the NEW weights' output (a train / eval / train / eval sequence in one
    process silently reused the first eval's planes)."""
    from balancedgroupsoftmax_amd import train
    from balancedgroupsoftmax_amd.backbone import cached_fold
    g = torch.Generator().manual_seed(3)
    conv = torch.nn.Conv2d(32, 64, 1, bias=True).to(DEV)
    bn = torch.nn.BatchNorm2d(64).to(DEV).eval()
    x = torch.randn(2, 16, 16, 32, generator=g).to(DEV)
    prev = BF.set_conv_math('bf16x6')
    try:
        def eval_out():
            with torch.no_grad():
                w, b = cached_fold(conv, bn)
                return BF.conv2d_nhwc(x, w, b)                    # frozen_weight=True: planes cached by version
        y0 = eval_out()
        v0 = conv.weight._version
        ps = list(conv.parameters())
        opt = torch.optim.SGD(ps, lr=0.5, momentum=0.0)
        for p_ in ps:
            p_.grad = torch.ones_like(p_)
        f = train.FusedClipSGD(opt, ps, None)
        f.step()
        assert conv.weight._version > v0 and conv.bias._version > 0
        y1 = eval_out()
        with torch.no_grad():
            ref = F.conv2d(x.permute(0, 3, 1, 2).double(), conv.weight.double(), conv.bias.double())
            ref = F.batch_norm(ref, bn.running_mean.double(), bn.running_var.double(), bn.weight.double(),
                               bn.bias.double(), False, 0.0, bn.eps).permute(0, 2, 3, 1)
        assert float((y1.double() - ref).abs().max()) < 1e-4 * float(ref.abs().max())
        assert float((y1 - y0).abs().max()) > 1.0          # lr 0.5 x ones moved every weight by 0.5
    finally:
        BF.set_conv_math(prev)


def test_dist_optimizer_step_uses_the_fused_kernels_on_the_gpu_and_torch_under_the_switch(monkeypatch):
    from balancedgroupsoftmax_amd import train
    ps = [torch.nn.Parameter(torch.randn(300, 20, device=DEV)), torch.nn.Parameter(torch.randn(7, device=DEV))]
    opt = train.build_optimizer(ps, dict(type='SGD', lr=0.01, momentum=0.9, weight_decay=1e-4))
    assert train.DistOptimizerStep(ps, opt, dict(max_norm=35, norm_type=2)).fused is not None
    assert train.DistOptimizerStep(ps, opt, dict(max_norm=35, norm_type=1)).fused is None
    monkeypatch.setenv('BGS_FUSED_SGD', '0')
    assert train.DistOptimizerStep(ps, opt, dict(max_norm=35, norm_type=2)).fused is None


def test_image_batch_to_padded_nhwc_in_one_launch():
    img = torch.randn(2, 3, 37, 53, generator=torch.Generator().manual_seed(2))
    ref = F.pad(img.permute(0, 2, 3, 1), (0, 1)).contiguous()
    assert torch.equal(BF.nchw_to_nhwc4(img.to(DEV)).cpu(), ref)


@pytest.mark.parametrize('case', [
    # name, N, H, W (input of the forward conv), Cin, Cout, k, pad, residual + mask
    ('l2.0.conv2', 2, 40, 56, 128, 128, 3, 1, False),
    ('l3.0.conv2 + fork', 1, 24, 36, 64, 256, 3, 1, True),
    ('l3.0.downsample', 2, 20, 28, 256, 512, 1, 0, True),
    ('ragged tiles', 1, 18, 22, 48, 80, 3, 1, False),
])
@pytest.mark.parametrize('math', ['bf16x6', 'bf16'])
def test_stride2_data_gradient_parity_rows_equal_the_zero_upsampled_form(case, math):
    """Stride-2 data gradients with the GEMM rows grouped by output-pixel parity (conv_bfx.hip UP == 3: only
    the taps that meet non-zeros of the zero-upsampled dy are multiplied; three of the four classes of a 1x1
    filter are pure epilogues) against the zero-upsampled form (the skipped products are exact zeros; equal up to
    the summation order of a split-K run) and against torch autograd."""
    from balancedgroupsoftmax_amd import capi
    lib = capi.load()
    name, N, H, W, Cin, Cout, k, pad, fork = case
    rs = np.random.RandomState(len(name) + Cout)
    x = rs.randn(N, H, W, Cin).astype(np.float32)
    w = (rs.randn(Cout, Cin, k, k) / np.sqrt(Cin * k * k)).astype(np.float32)
    Ho, Wo = (H + 2 * pad - k) // 2 + 1, (W + 2 * pad - k) // 2 + 1
    dy = rs.randn(N, Ho, Wo, Cout).astype(np.float32)
    res = rs.randn(N, H, W, Cin).astype(np.float32) if fork else None
    mask = rs.randn(N, H, W, Cin).astype(np.float32) if fork else None
    edx, _ = _torch_conv_grads(x, w, dy, 2, pad)
    if fork:
        edx = (edx + res) * (mask > 0)
    wk = dev(krsc(w))
    prev = BF.set_conv_math(math)
    outs = []
    try:
        for on in (1, 0):
            lib.bgs_conv_dgrad_parity_enable(on)
            dx = BF.conv2d_dgrad_nhwc(dev(dy), wk, (H, W), stride=2, pad=pad,
                                      residual=None if res is None else dev(res),
                                      mask=None if mask is None else dev(mask))
            tile = BF.conv_bfx_last_launch()['tile']
            assert bool(tile & 0x2000) == bool(on), hex(tile)
            outs.append(dx.cpu())
    finally:
        lib.bgs_conv_dgrad_parity_enable(1)
        BF.set_conv_math(prev)
    # (the zero-upsampled arm may take split-K: same products, another summation order)
    scale = float(outs[1].abs().max())
    assert float((outs[0] - outs[1]).abs().max()) <= (2e-6 if math == 'bf16x6' else 2e-6) * scale, \
        float((outs[0] - outs[1]).abs().max()) / scale
    tol = 2e-5 if math == 'bf16x6' else 3e-2
    assert np.abs(outs[0].numpy() - edx).max() <= tol * np.abs(edx).max()


@pytest.mark.parametrize('case', [
    # N, H, W, residual, relu3
    (1, 8, 16, True, True),             # exactly one tile
    (1, 13, 21, True, True),            # ragged in both directions (tile rows / columns past the image)
    (2, 40, 56, False, True),           # no residual
    (2, 40, 56, True, False),           # no final ReLU
    (2, 200, 336, True, True),          # ResNet-50 layer1 at the BASELINE size (2100 tiles)
], ids=lambda c: 'x'.join(str(int(v)) for v in c))
def test_fused_conv2_conv3_bottleneck_tail_is_bit_identical_to_the_two_launches(case, monkeypatch):
    """``conv3x3_c3_fused_bfx_kernel`` (round 6): conv2 (3x3, 64 -> 64, folded BN, ReLU) -> conv3 (1x1, 64 -> 256, folded
    BN) + residual + ReLU of a frozen bottleneck (mmdet/models/backbones/resnet.py:239-266) in ONE launch, the 64-channel
    intermediate kept in LDS as split bf16 planes.  Every output is accumulated in the unfused kernels' order (the halo
    kernel's taps / chunks, the ring kernel's k steps and plane products, bias then residual then clamp): BIT-IDENTICAL
    to ``conv2d_nhwc(3x3)`` + ``conv2d_nhwc(1x1, residual=)``, and within the family's bound of fp64."""
    N, H, W, with_res, relu3 = case
    g = torch.Generator().manual_seed(H * 7 + W)
    x = torch.relu(torch.randn(N, H, W, 64, generator=g) * torch.exp(torch.randn(N, H, W, 64, generator=g)))
    w2 = torch.randn(64, 3, 3, 64, generator=g) * (2.0 / (9 * 64)) ** 0.5
    b2 = torch.randn(64, generator=g) * 0.5
    w3 = torch.randn(256, 1, 1, 64, generator=g) * (2.0 / 64) ** 0.5
    b3 = torch.randn(256, generator=g) * 0.5
    res = torch.randn(N, H, W, 256, generator=g) if with_res else None
    monkeypatch.setenv('BGS_CONV_HALO', '1')
    prev = BF.set_conv_math('bf16x6')
    try:
        xd, w2d, b2d, w3d, b3d = dev(x), dev(w2), dev(b2), dev(w3), dev(b3)
        rd = dev(res) if with_res else None
        BF.conv_bfx_tuning(halo_splits=1, halo_wide=0)
        t2 = BF.conv2d_nhwc(xd, w2d, b2d, pad=1, relu=True)
        ref = BF.conv2d_nhwc(t2, w3d, b3d, relu=relu3, residual=rd)
        out = BF.conv3x3_c3_fused_nhwc(xd, w2d, b2d, w3d, b3d, residual=rd, relu3=relu3)
        torch.cuda.synchronize()
        assert torch.equal(out, ref), float((out - ref).abs().max())
        # fp64 reference of the chain (the fp32 intermediate rounded as the kernels round it)
        t2_64 = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w2.double().permute(0, 3, 1, 2), b2.double(),
                                           padding=1).relu().float().double()
        y64 = torch.nn.functional.conv2d(t2_64, w3.double().permute(0, 3, 1, 2), b3.double())
        if with_res:
            y64 = y64 + res.double().permute(0, 3, 1, 2)
        if relu3:
            y64 = y64.relu()
        err = (out.cpu().double() - y64.permute(0, 2, 3, 1)).abs().max().item()
        assert err < 2e-4 * max(1.0, y64.abs().max().item()), err
    finally:
        BF.conv_bfx_tuning()
        BF.set_conv_math(prev)


def test_frozen_layer1_bottleneck_takes_the_fused_tail_and_keeps_its_bits(monkeypatch):
    """``backbone.Bottleneck.run`` on a frozen 64 -> 64 -> 256 block over a map large enough for the halo kernel: the fused
    conv2 -> conv3 launch (default) == the three-launch block (``BGS_FUSED_C3`` off), bit for bit, with and without the
    projection shortcut."""
    from balancedgroupsoftmax_amd import backbone as B
    monkeypatch.setenv('BGS_CONV_HALO', '1')
    prev = BF.set_conv_math('bf16x6')
    try:
        for ds, inpl in ((True, 64), (False, 256)):
            torch.manual_seed(5 + inpl)
            blk = B.Bottleneck(inpl, 64, stride=1, downsample=ds).to(DEV).eval()
            with torch.no_grad():
                for m in blk.modules():
                    if isinstance(m, torch.nn.BatchNorm2d):
                        m.running_mean.normal_(0, 0.1)
                        m.running_var.uniform_(0.5, 1.5)
                        m.weight.normal_(1, 0.1)
                        m.bias.normal_(0, 0.1)
            for p_ in blk.parameters():
                p_.requires_grad = False
            x = torch.relu(torch.randn(2, 160, 288, inpl, device=DEV))       # 720 pixel tiles: the fused path's range
            assert BF.fused_c3_eligible(torch.empty(2, 160, 288, 64, device=DEV), blk.folded()['c2'][0],
                                        blk.folded()['c3'][0], 1, None)
            with torch.no_grad():
                f = blk.folded()
                on = BF.set_fused_c3(True)
                BF.launch_census(reset=True)
                y1 = blk.run(x, f)
                assert BF.launch_census()['fused_c3'] == 1
                BF.set_fused_c3(False)
                BF.launch_census(reset=True)
                y0 = blk.run(x, f)
                assert BF.launch_census()['fused_c3'] == 0
                BF.set_fused_c3(on)
            assert torch.equal(y1, y0)
    finally:
        BF.set_conv_math(prev)


@pytest.mark.parametrize('case', [
    # N, H, W, Cin, Cout, relu, res_mode
    (1, 13, 10, 64, 256, True, 1),            # M = 130: ragged last tile, one K chunk
    (2, 50, 84, 256, 1024, True, 1),          # conv3 of layer3: four channel tiles
    (2, 50, 84, 1024, 256, True, 0),          # conv1 of layer3: 16 K chunks
    (2, 20, 28, 512, 256, False, 2),          # FPN lateral: nearest-2x upsampled residual
    (1, 7, 9, 128, 512, False, 0),            # M = 63: less than one tile
    (2, 200, 336, 256, 256, False, 2),        # fpn.lat0 at the BASELINE size
    (2, 100, 168, 512, 128, True, 0),         # layer2 conv1: 128 output channels (the 128-channel workgroup)
    (2, 50, 84, 1024, 256, True, 0),          # layer3 conv1: 132 pixel tiles -> two 128-channel workgroups per tile
    (1, 33, 21, 64, 384, False, 1),           # three 128-channel slabs, ragged last pixel tile
    (2, 100, 168, 256, 512, False, -2),       # projection shortcut of layer2: 1x1 / stride 2 (res_mode -2 = stride 2, no residual)
    (1, 15, 11, 128, 256, True, -2),          # stride 2 on an odd map (Ho = 8, Wo = 6)
], ids=lambda c: 'x'.join(str(int(v)) for v in c))
def test_planes_in_lds_1x1_kernel_is_bit_identical_to_the_ring_and_wide_kernels(case):
    """``conv1x1_planes_bfx_kernel`` (round 6, VERDICT r5 item 1: the A-operand split out of the MFMA loop — the fp32 A tile
    of a 64-deep K chunk is split ONCE per 64-pixel x 256-channel workgroup into bf16 planes in LDS; resnet.py:220-266,
    fpn.py:118-127) accumulates the k steps and the six plane products of a step in the existing kernels' order and runs
    their epilogue: BIT-IDENTICAL to the default dispatch (64 x 64 ring / 128 x 128 wide kernel, unsliced), with both
    residual modes, ReLU, ragged M; and within the family's bound of fp64."""
    from balancedgroupsoftmax_amd import capi
    lib = capi.load()
    N, H, W, Cin, Cout, relu, rm = case
    stride = 1
    if rm == -2:
        stride, rm = 2, 0
    g = torch.Generator().manual_seed(H * 31 + Cin)
    x = torch.randn(N, H, W, Cin, generator=g) * torch.exp(torch.randn(N, H, W, Cin, generator=g))
    w = torch.randn(Cout, 1, 1, Cin, generator=g) * (2.0 / Cin) ** 0.5
    b = torch.randn(Cout, generator=g)
    res = None
    if rm == 1:
        res = torch.randn(N, H, W, Cout, generator=g)
    elif rm == 2:
        res = torch.randn(N, H // 2, W // 2, Cout, generator=g)
    prev = BF.set_conv_math('bf16x6')
    try:
        xd, wd, bd = dev(x), dev(w), dev(b)
        rd = dev(res) if res is not None else None
        kw = dict(stride=stride, relu=relu, residual=rd, residual_mode=rm)
        lib.bgs_conv1x1_planes_enable(0)
        BF.conv_bfx_tuning(0, 1)                       # (unsliced reference arm: split-K is another summation order)
        y0 = BF.conv2d_nhwc(xd, wd, bd, **kw)
        BF.conv_bfx_tuning(0, -1)
        assert not lib.bgs_conv1x1_planes_last_launch()
        lib.bgs_conv1x1_planes_enable(2)
        y1 = BF.conv2d_nhwc(xd, wd, bd, **kw)
        assert lib.bgs_conv1x1_planes_last_launch() in (1, 2)          # = channels per workgroup / 128
        torch.cuda.synchronize()
        assert torch.equal(y0, y1), float((y0 - y1).abs().max())
        y64 = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), b.double(),
                                         stride=stride)
        if rm == 1:
            y64 = y64 + res.double().permute(0, 3, 1, 2)
        elif rm == 2:
            y64 = y64 + torch.nn.functional.interpolate(res.double().permute(0, 3, 1, 2), scale_factor=2, mode='nearest')
        if relu:
            y64 = y64.relu()
        scale = (x.abs().double().reshape(-1, Cin) @ w.abs().double().reshape(Cout, Cin).t()).max().item()
        assert (y1.cpu().double() - y64.permute(0, 2, 3, 1)).abs().max().item() < 1e-6 * scale
    finally:
        lib.bgs_conv1x1_planes_enable(-1)
        BF.conv_bfx_tuning()
        BF.set_conv_math(prev)


@pytest.mark.parametrize('case', [
    # N, H, W, Cin, Cout, relu
    (2, 50, 84, 256, 256, True),          # layer3 conv2 / the stride-16 pyramid level at the BASELINE size
    (2, 25, 42, 512, 512, True),          # layer4 conv2: ragged tiles (25 = 3 x 8 + 1, 42 = 5 x 8 + 2)
    (2, 100, 168, 128, 128, True),        # layer2 conv2: the 128-channel workgroup
    (2, 31, 37, 256, 256, False),         # ragged in both directions, no ReLU
    (2, 37, 29, 64, 384, False),          # three 128-channel slabs, two 32-channel chunks
    (32, 9, 8, 32, 128, True),            # ONE chunk, tiles of one partial row
], ids=lambda c: 'x'.join(str(int(v)) for v in c))
def test_planes_3x3_kernel_is_bit_identical_to_the_unsliced_halo_kernel(case):
    """``conv3x3_planes_bfx_kernel`` (csrc/conv3x3_planes.hip, round 6: 8 x 8 output pixels x 128 / 256 channels per
    workgroup, the whole reduction inside the workgroup, patch planes in LDS per 32-channel chunk; the 3x3 / stride-1 convs of
    mmdet/models/backbones/resnet.py:239-252, necks/fpn.py:129-141, anchor_heads/rpn_head.py:30-35 on the small maps)
    accumulates the 16-channel chunks, the nine taps of a chunk and the six plane products of a k step in the halo kernel's
    order: BIT-IDENTICAL to ``conv3x3_halo_bfx4_kernel`` with ONE K slice (every pixel incl. the zero-padded border and the
    ragged last tiles), within fp32 summation order of the sliced default, and within the family's bound of fp64."""
    from balancedgroupsoftmax_amd import capi
    lib = capi.load()
    N, H, W, Cin, Cout, relu = case
    g = torch.Generator().manual_seed(H * 17 + Cin)
    x = torch.randn(N, H, W, Cin, generator=g) * torch.exp(torch.randn(N, H, W, Cin, generator=g))
    w = torch.randn(Cout, 3, 3, Cin, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, generator=g)
    prev = BF.set_conv_math('bf16x6')
    try:
        xd, wd, bd = dev(x), dev(w), dev(b)
        lib.bgs_conv3x3_planes_enable(0)
        BF.conv_bfx_tuning(halo_splits=1)              # the halo kernel, one K slice (a forced slice count also keeps the
        y0 = BF.conv2d_nhwc(xd, wd, bd, pad=1, relu=relu)        # planes kernel out of the way)
        assert BF.conv_bfx_last_launch()['halo_variant'] == 4 and not lib.bgs_conv3x3_planes_last_launch()
        BF.conv_bfx_tuning()
        yd = BF.conv2d_nhwc(xd, wd, bd, pad=1, relu=relu)        # the default plan (K sliced on these grids)
        lib.bgs_conv3x3_planes_enable(2)
        y1 = BF.conv2d_nhwc(xd, wd, bd, pad=1, relu=relu)
        assert lib.bgs_conv3x3_planes_last_launch() in (1, 2)          # = channels per workgroup / 128
        assert BF.conv_bfx_last_launch()['halo_variant'] == 9
        torch.cuda.synchronize()
        assert torch.equal(y0, y1), float((y0 - y1).abs().max())
        y64 = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), b.double(),
                                         padding=1)
        if relu:
            y64 = y64.clamp(min=0)
        y64 = y64.permute(0, 2, 3, 1)
        scale = float(y64.abs().max())
        tol = 2e-6 * max(1.0, (9 * Cin / 2304.0) ** 0.5)           # fp32 accumulation over K = 9 Cin terms (heavy-tailed x)
        e1, ed = float((y1.cpu().double() - y64).abs().max()) / scale, float((yd.cpu().double() - y64).abs().max()) / scale
        assert e1 < tol and ed < tol, (e1, ed, tol)
    finally:
        lib.bgs_conv3x3_planes_enable(-1)
        BF.conv_bfx_tuning()
        BF.set_conv_math(prev)


@pytest.mark.parametrize('case', [
    # R, N, H, W, Cin_fwd, Cout_fwd, residual
    (1, 2, 50, 84, 1024, 256, True),      # layer3 conv1 backward: 256 -> 1024 channels, the skip connection's gradient added
    (1, 2, 100, 168, 128, 512, False),    # layer2 conv3 backward: 512 -> 128 channels (the 128-channel workgroup)
    (3, 2, 50, 84, 256, 256, False),      # layer3 conv2 backward
    (3, 1, 100, 168, 128, 128, False),    # layer2 conv2 backward, one image
], ids=lambda c: 'x'.join(str(int(v)) for v in c))
def test_planes_kernels_as_data_gradients_with_the_relu_mask_are_bit_identical(case):
    """The data gradient of a stride-1 conv is the same conv of ``dy`` with the flipped / transposed filter, gated by the
    ReLU-backward mask of the conv's input (``bgs_conv2d_dgrad_nhwc_f32_bfx_ws``; the backward of
    mmdet/models/backbones/resnet.py:220-266 under ``selectp = 0``): with the mask (and the residual gradient) in their
    epilogues the planes kernels take these launches too — BIT-IDENTICAL to the kernels behind them (1x1: the default
    dispatch unsliced; 3x3: the halo kernel with one K slice), and both within fp32 rounding of fp64 torch."""
    from balancedgroupsoftmax_amd import capi
    lib = capi.load()
    R, N, H, W, Cin, Cout, with_res = case
    g = torch.Generator().manual_seed(R * 100 + Cin)
    dy = torch.randn(N, H, W, Cout, generator=g)
    w = torch.randn(Cout, R, R, Cin, generator=g) * (2.0 / (R * R * Cout)) ** 0.5
    mask = torch.randn(N, H, W, Cin, generator=g)
    res = torch.randn(N, H, W, Cin, generator=g) if with_res else None
    prev = BF.set_conv_math('bf16x6')
    try:
        kw = dict(pad=R // 2, residual=None if res is None else dev(res), mask=dev(mask))
        lib.bgs_conv1x1_planes_enable(0)
        lib.bgs_conv3x3_planes_enable(0)
        BF.conv_bfx_tuning(0, 1, halo_splits=1)          # reference arm: no K slices anywhere
        d0 = BF.conv2d_dgrad_nhwc(dev(dy), dev(w), (H, W), **kw)
        BF.conv_bfx_tuning()
        lib.bgs_conv1x1_planes_enable(2)
        lib.bgs_conv3x3_planes_enable(2)
        d1 = BF.conv2d_dgrad_nhwc(dev(dy), dev(w), (H, W), **kw)
        took = lib.bgs_conv1x1_planes_last_launch() if R == 1 else lib.bgs_conv3x3_planes_last_launch()
        assert took in (1, 2), took
        torch.cuda.synchronize()
        assert torch.equal(d0, d1), float((d0 - d1).abs().max())
    finally:
        lib.bgs_conv1x1_planes_enable(-1)
        lib.bgs_conv3x3_planes_enable(-1)
        BF.conv_bfx_tuning()
        BF.set_conv_math(prev)
    ref = torch.nn.functional.conv_transpose2d(dy.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), padding=R // 2)
    ref = ref.permute(0, 2, 3, 1)
    if res is not None:
        ref = ref + res.double()
    ref = torch.where(mask > 0, ref, torch.zeros_like(ref))
    err = float((d1.cpu().double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, err


@pytest.mark.parametrize('case', [
    # N, H, W, Cin, Cout, relu
    (2, 200, 336, 128, 128, True),        # layer2.0 conv2 at the BASELINE size
    (2, 100, 168, 256, 256, True),        # layer3.0 conv2
    (2, 50, 84, 512, 512, True),          # layer4.0 conv2: ragged output tiles (25 x 42)
    (1, 37, 29, 64, 384, False),          # odd input sizes (Ho = 19, Wo = 15), three 128-channel slabs
    (3, 16, 18, 32, 128, True),           # two chunks only, even sizes (the last input row / column is never read)
], ids=lambda c: 'x'.join(str(int(v)) for v in c))
def test_planes_3x3_stride2_kernel_vs_the_operand_ring_and_fp64(case):
    """``conv3x3s2_planes_bfx_kernel`` (csrc/conv3x3_planes.hip: the 17 x 17 input patch of 8 x 8 output pixels as four
    parity sub-grids in LDS; the stride-2 conv2 of mmdet/models/backbones/resnet.py:239-252) against the 64 x 64 operand
    ring it replaces: the same products in another fp32 summation order (chunk-major / tap-major), so equality within
    fp32 rounding, and both within the family's bound of fp64 torch (every pixel incl. the zero-padded border)."""
    from balancedgroupsoftmax_amd import capi
    lib = capi.load()
    N, H, W, Cin, Cout, relu = case
    g = torch.Generator().manual_seed(H * 13 + Cin)
    x = torch.randn(N, H, W, Cin, generator=g) * torch.exp(torch.randn(N, H, W, Cin, generator=g))
    w = torch.randn(Cout, 3, 3, Cin, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, generator=g)
    prev = BF.set_conv_math('bf16x6')
    try:
        xd, wd, bd = dev(x), dev(w), dev(b)
        lib.bgs_conv3x3_planes_enable(0)
        y0 = BF.conv2d_nhwc(xd, wd, bd, stride=2, pad=1, relu=relu)
        assert not lib.bgs_conv3x3_planes_last_launch()
        lib.bgs_conv3x3_planes_enable(2)
        y1 = BF.conv2d_nhwc(xd, wd, bd, stride=2, pad=1, relu=relu)
        assert lib.bgs_conv3x3_planes_last_launch() == 0x11          # stride-2 form, 128 channels per workgroup
        torch.cuda.synchronize()
    finally:
        lib.bgs_conv3x3_planes_enable(-1)
        BF.set_conv_math(prev)
    y64 = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), b.double(),
                                     stride=2, padding=1)
    if relu:
        y64 = y64.clamp(min=0)
    y64 = y64.permute(0, 2, 3, 1)
    assert tuple(y1.shape) == tuple(y64.shape)
    scale = float(y64.abs().max())
    tol = 2e-6 * max(1.0, (9 * Cin / 2304.0) ** 0.5)
    e0, e1 = float((y0.cpu().double() - y64).abs().max()) / scale, float((y1.cpu().double() - y64).abs().max()) / scale
    assert e0 < tol and e1 < tol, (e0, e1, tol)
    assert float((y0 - y1).abs().max()) / scale < tol
