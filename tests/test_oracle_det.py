"""CPU: pins of the detector-op oracles (oracle/det_oracle.py)."""
import numpy as np
import pytest
import torch

from oracle import build_ref, det_oracle


@pytest.fixture(scope='module')
def ref_nms():
    m = build_ref.load_nms_cpu()
    if m is None:
        pytest.skip('oracle/_ref/nms_cpu_ref.so not built and reference tree absent')
    return m


@pytest.mark.parametrize('n,seed', [(1, 0), (2, 1), (50, 2), (300, 3), (1000, 4)])
@pytest.mark.parametrize('thr', [0.3, 0.5, 0.7])
def test_nms_restatement_vs_compiled_reference(ref_nms, n, seed, thr):
    """oracle nms(mode='cpu') == the reference's own nms_cpu.cpp compiled from source."""
    dets = det_oracle.make_boxes(n, seed)
    exp = ref_nms.nms(torch.from_numpy(dets), float(thr)).numpy()
    got = det_oracle.nms(dets, thr, mode='cpu')
    np.testing.assert_array_equal(got, exp)


def test_nms_tie_semantics_differ_between_cpu_and_cuda(ref_nms):
    """F8: IoU exactly == thr is suppressed by nms_cpu (>=) but kept by nms_cuda (>)."""
    dets = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 4, 0.8]], dtype=np.float32)  # IoU = 0.5
    assert ref_nms.nms(torch.from_numpy(dets), 0.5).tolist() == [0]
    assert det_oracle.nms(dets, 0.5, 'cpu').tolist() == [0]
    assert det_oracle.nms(dets, 0.5, 'cuda').tolist() == [0, 1]


def test_nms_empty():
    assert det_oracle.nms(np.zeros((0, 5), np.float32), 0.5).shape == (0,)


def test_roi_align_analytic_properties():
    """No executed reference exists for RoIAlign; the restatement must satisfy what the CUDA
    kernel's formulas imply: constant map -> constant; f(y,x) = a*y + b*x + c -> value at the
    mean sample position (interior RoIs); fully-outside RoI -> 0."""
    H, W, C = 40, 50, 3
    const = np.full((1, H, W, C), 2.5, np.float32)
    rois = np.array([[0, 10, 8, 60, 70], [0, 0, 0, 30, 20]], np.float32)
    out = det_oracle.roi_align_forward(const, rois, 0.25)
    assert np.allclose(out, 2.5, atol=1e-6)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    ramp = np.stack([0.5 * yy + 0.25 * xx + 1.0, yy * 1.0, xx * 1.0], -1)[None].astype(np.float32)
    roi = np.array([[0, 40, 32, 120, 100]], np.float32)
    out = det_oracle.roi_align_forward(ramp, roi, 0.25)
    sw, sh = 40 * 0.25, 32 * 0.25
    bw, bh = ((120 + 1) * 0.25 - sw) / 7, ((100 + 1) * 0.25 - sh) / 7
    for ph in range(7):
        for pw in range(7):
            y = sh + (ph + 0.5) * bh
            x = sw + (pw + 0.5) * bw
            assert np.allclose(out[0, ph, pw], [0.5 * y + 0.25 * x + 1.0, y, x], atol=1e-4)
    far = np.array([[0, 4000, 4000, 4100, 4100]], np.float32)
    assert not det_oracle.roi_align_forward(ramp, far, 0.25).any()


def test_map_roi_levels_matches_torch_formula():
    rs = np.random.RandomState(0)
    wh = np.exp(rs.uniform(np.log(4), np.log(900), (500, 2)))
    xy = rs.uniform(0, 400, (500, 2))
    rois = np.concatenate([np.zeros((500, 1)), xy, xy + wh], 1).astype(np.float32)
    t = torch.from_numpy(rois)
    scale = torch.sqrt((t[:, 3] - t[:, 1] + 1) * (t[:, 4] - t[:, 2] + 1))
    exp = torch.floor(torch.log2(scale / 56 + 1e-6)).clamp(min=0, max=3).long().numpy()
    np.testing.assert_array_equal(det_oracle.map_roi_levels(rois, 4), exp)
    assert set(exp.tolist()) == {0, 1, 2, 3}


# ---------------------------------------------------------------- multiclass NMS (test-time path)
def _mc_golden():
    import json
    import os
    from tests.golden import make_golden_det
    z = np.load(os.path.join(os.path.dirname(make_golden_det.__file__), 'multiclass_nms_golden.npz'))
    cases = json.loads(bytes(z['__cases__']).decode())
    return z, cases, make_golden_det.case_inputs


@pytest.mark.parametrize('name', ['c31_cut', 'c11_agnostic_all', 'c1231_thr', 'c21_empty',
                                  'c5_nocap', 'c1231_lvis'])
def test_multiclass_nms_restatement_vs_executed_reference(name):
    """oracle.multiclass_nms(mode='cpu') == the reference's multiclass_nms run on CPU with its own
    compiled nms_cpu.cpp (golden vectors; tests/golden/make_golden_det.py)."""
    z, cases, case_inputs = _mc_golden()
    case = [c for c in cases if c['name'] == name][0]
    boxes, scores = case_inputs(case)
    bb, ll = det_oracle.multiclass_nms(boxes, scores, case['score_thr'], case['iou_thr'],
                                       case['max_num'], mode='cpu')
    np.testing.assert_array_equal(ll, z[name + '/det_labels'])
    np.testing.assert_array_equal(bb, z[name + '/det_bboxes'])


def test_roi_align_backward_restatement_is_the_adjoint_of_the_forward():
    """<RoIAlign(x), g> == <x, RoIAlign^T(g)> for the two numpy restatements (the reference's
    gradcheck recipe, ops/roi_align/gradcheck.py:11-30, states the same property numerically)."""
    rs = np.random.RandomState(3)
    feat = rs.randn(2, 9, 11, 3).astype(np.float32)
    rois = np.array([[0, 1.5, 2.0, 30.0, 25.5], [1, -6.0, -3.0, 12.0, 40.0],
                     [1, 20.0, 10.0, 47.0, 38.0], [0, 5.0, 5.0, 5.0, 5.0]], np.float32)
    g = rs.randn(4, 3, 3, 3).astype(np.float32)
    out = det_oracle.roi_align_forward(feat, rois, 0.25, 3, 3, 2)
    dfeat = det_oracle.roi_align_backward(g, rois, 0.25, feat.shape, 2)
    lhs = float((out.astype(np.float64) * g).sum())
    rhs = float((feat.astype(np.float64) * dfeat).sum())
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))


# ---------------------------------------------------------------- RoIAlign pinned to the reference
@pytest.fixture(scope='module')
def ref_roi():
    if build_ref.load_roi_align() is None:
        pytest.skip('oracle/_ref/roi_align_ref.so not built and reference tree absent')
    return build_ref


def _roi_cases(seed, K=40, H=23, W=31):
    rs = np.random.RandomState(seed)
    ctr = rs.uniform(-0.1, 1.1, (K, 2)) * np.array([W * 8, H * 8])
    size = np.exp(rs.uniform(np.log(2), np.log(400), (K, 2)))
    rois = np.concatenate([rs.randint(0, 2, (K, 1)), ctr - size / 2, ctr + size / 2], 1).astype(np.float32)
    rois[0] = [0, 5.0, 5.0, 5.0, 5.0]                 # degenerate 1-pixel box
    rois[1] = [1, -40.0, -30.0, 20.0, 10.0]           # hangs over the top-left corner
    rois[2] = [0, W * 8 - 10.0, H * 8 - 12.0, W * 8 + 60.0, H * 8 + 50.0]   # over the bottom-right
    rois[3] = [1, 30.0, 30.0, 10.0, 10.0]             # malformed (x2 < x1): width clamps to 0
    return rois


@pytest.mark.parametrize('out_size', [7, 14])
def test_roi_align_forward_restatement_vs_compiled_reference(ref_roi, out_size):
    """oracle.roi_align_forward == the reference's own ROIAlignForward<float> (roi_align_kernel.cu
    :63-124, compiled as host code by oracle/build_ref.py)."""
    rs = np.random.RandomState(out_size)
    feat = rs.randn(2, 23, 31, 6).astype(np.float32)
    rois = _roi_cases(out_size)
    exp = ref_roi.roi_align_reference(feat.transpose(0, 3, 1, 2), rois, 0.125, out_size)
    got = det_oracle.roi_align_forward(feat, rois, 0.125, out_size, out_size, 2)
    np.testing.assert_allclose(got.transpose(0, 3, 1, 2), exp, rtol=1e-6, atol=1e-6)
    assert np.abs(exp).max() > 0.1


def test_roi_align_backward_restatement_vs_compiled_reference(ref_roi):
    rs = np.random.RandomState(2)
    rois = _roi_cases(5, K=25)
    g = rs.randn(25, 7, 7, 4).astype(np.float32)
    exp = ref_roi.roi_align_reference_backward(g.transpose(0, 3, 1, 2), rois, 0.125, (2, 4, 23, 31))
    got = det_oracle.roi_align_backward(g, rois, 0.125, (2, 23, 31, 4), 2)
    np.testing.assert_allclose(got.transpose(0, 3, 1, 2), exp, rtol=1e-4, atol=1e-5)
