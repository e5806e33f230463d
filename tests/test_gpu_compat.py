"""GPU: the operator-level drop-in boundary as the REFERENCE calls it.

* ``balancedgroupsoftmax_amd.compat.roi_align_cuda`` / ``.nms_cuda`` carry the exact pybind11
  signatures of the reference's extension modules.  ``/root/reference`` does not travel to the GPU
  box, so the reference's Python callers are restated here line for line (``RoIAlignFunction``,
  mmdet/ops/roi_align/roi_align.py:9-53; ``nms``, mmdet/ops/nms/nms_wrapper.py:8-49) with the compat
  modules bound where they import their extensions, and the results are compared with the
  reference's OWN kernels built for the host (``oracle/_ref``: ``ROIAlignForward/Backward`` of
  roi_align_kernel.cu, ``nms_cpu.cpp``) and the restated ``>`` rule of nms_kernel.cu.
* The reference-side ctypes binding printed in INTEGRATION.md is EXECUTED (the code block is
  extracted from the document) against the fixtures of the executed reference class.
"""
import os
import re

import numpy as np
import pytest
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from balancedgroupsoftmax_amd import capi
from balancedgroupsoftmax_amd.compat import nms_cuda, roi_align_cuda
from oracle import build_ref, det_oracle, gs_oracle
from tests.golden_util import case_setup, golden

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class RoIAlignFunction(Function):
    """mmdet/ops/roi_align/roi_align.py:9-53 with ``roi_align_cuda`` = the compat module."""

    @staticmethod
    def forward(ctx, features, rois, out_size, spatial_scale, sample_num=0):
        out_h, out_w = (out_size, out_size) if isinstance(out_size, int) else out_size
        ctx.spatial_scale = spatial_scale
        ctx.sample_num = sample_num
        ctx.save_for_backward(rois)
        ctx.feature_size = features.size()
        batch_size, num_channels, data_height, data_width = features.size()
        num_rois = rois.size(0)
        output = features.new_zeros(num_rois, num_channels, out_h, out_w)
        if features.is_cuda:
            roi_align_cuda.forward(features, rois, out_h, out_w, spatial_scale, sample_num, output)
        else:
            raise NotImplementedError
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        feature_size = ctx.feature_size
        rois = ctx.saved_tensors[0]
        assert feature_size is not None and grad_output.is_cuda
        batch_size, num_channels, data_height, data_width = feature_size
        out_w, out_h = grad_output.size(3), grad_output.size(2)
        grad_input = None
        if ctx.needs_input_grad[0]:
            grad_input = rois.new_zeros(batch_size, num_channels, data_height, data_width)
            roi_align_cuda.backward(grad_output.contiguous(), rois, out_h, out_w, ctx.spatial_scale,
                                    ctx.sample_num, grad_input)
        return grad_input, None, None, None, None


def _rois(rs, K, N, W, H):
    xy = rs.uniform(-20, [W - 10, H - 10], size=(K, 2))
    wh = np.exp(rs.uniform(np.log(4), np.log(300), size=(K, 2)))
    b = rs.randint(0, N, size=(K, 1)).astype(np.float32)
    return np.concatenate([b, xy, xy + wh], 1).astype(np.float32)


@pytest.mark.parametrize('out_size,scale', [(7, 0.25), (14, 0.125), ((7, 5), 1.0 / 16)])
def test_roi_align_cuda_signature_vs_compiled_reference_kernels(out_size, scale):
    rs = np.random.RandomState(11)
    N, C, H, W = 2, 24, 40, 56
    feat = rs.standard_normal((N, C, H, W)).astype(np.float32)
    rois = _rois(rs, 37, N, W / scale, H / scale)
    f = torch.from_numpy(feat).to(DEV).requires_grad_(True)
    r = torch.from_numpy(rois).to(DEV)
    out = RoIAlignFunction.apply(f, r, out_size, scale, 2)
    oh, ow = (out_size, out_size) if isinstance(out_size, int) else out_size
    assert tuple(out.shape) == (37, C, oh, ow)
    if oh == ow:
        exp = build_ref.roi_align_reference(feat, rois, scale, oh, 2)
        assert np.abs(out.detach().cpu().numpy() - exp).max() < 1e-5 * max(1.0, np.abs(exp).max())
    else:   # the host build of the reference kernels takes a square out_size: use the restatement
        exp = det_oracle.roi_align_forward(np.ascontiguousarray(feat.transpose(0, 2, 3, 1)), rois,
                                           scale, oh, ow, 2).transpose(0, 3, 1, 2)
        assert np.abs(out.detach().cpu().numpy() - exp).max() < 1e-5 * max(1.0, np.abs(exp).max())
    g = rs.standard_normal(tuple(out.shape)).astype(np.float32)
    out.backward(torch.from_numpy(g).to(DEV))
    if oh == ow:
        expg = build_ref.roi_align_reference_backward(g, rois, scale, (N, C, H, W), 2)
        assert np.abs(f.grad.cpu().numpy() - expg).max() < 1e-4 * max(1.0, np.abs(expg).max())
    # backward ACCUMULATES into the caller's buffer (the reference kernel's atomicAdd)
    acc = torch.ones((N, C, H, W), device=DEV)
    assert roi_align_cuda.backward(torch.from_numpy(g).to(DEV), r, oh, ow, scale, 2, acc) == 1
    # (two runs of an fp32-atomics scatter: the summation order differs from launch to launch, and `acc`
    #  starts at 1.0 — the comparison carries the rounding of up to a few hundred unordered additions)
    assert torch.allclose(acc - 1.0, f.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('sample_num', [0, 1, 2, 3])
def test_roi_align_cuda_every_sample_num_vs_compiled_reference_kernels(sample_num):
    """The whole ``sample_num`` range of the reference interface (roi_align_kernel.cu:95-99): n x n grids and the
    ADAPTIVE grid (0: ceil(roi_size / pooled_size) samples per axis, different for every RoI), forward and backward,
    against the reference's own ``ROIAlignForward`` / ``ROIAlignBackward`` compiled for the host (oracle/_ref)."""
    rs = np.random.RandomState(31 + sample_num)
    N, C, H, W, scale, K = 2, 16, 40, 56, 0.25, 41
    feat = rs.standard_normal((N, C, H, W)).astype(np.float32)
    rois = _rois(rs, K, N, W / scale, H / scale)
    f = torch.from_numpy(feat).to(DEV).requires_grad_(True)
    r = torch.from_numpy(rois).to(DEV)
    out = RoIAlignFunction.apply(f, r, 7, scale, sample_num)
    exp = build_ref.roi_align_reference(feat, rois, scale, 7, sample_num)
    assert np.abs(out.detach().cpu().numpy() - exp).max() < 1e-5 * max(1.0, np.abs(exp).max())
    g = rs.standard_normal(tuple(out.shape)).astype(np.float32)
    out.backward(torch.from_numpy(g).to(DEV))
    expg = build_ref.roi_align_reference_backward(g, rois, scale, (N, C, H, W), sample_num)
    assert np.abs(f.grad.cpu().numpy() - expg).max() < 1e-4 * max(1.0, np.abs(expg).max())


@pytest.mark.parametrize('sample_num', [0, 2])
def test_roi_align_cuda_fp16_tensors(sample_num):
    """The half instantiation of the reference's dispatch (roi_align_kernel.cu:136,281): fp16 features / rois /
    output / gradients through the same signature; compared with the compiled reference kernels run in fp32 on the
    fp16-rounded inputs, to fp16 resolution (this path accumulates in fp32 and rounds once)."""
    rs = np.random.RandomState(7 + sample_num)
    N, C, H, W, scale, K = 2, 32, 30, 44, 0.125, 29
    feat = rs.standard_normal((N, C, H, W)).astype(np.float16)
    rois = np.round(_rois(rs, K, N, W / scale, H / scale)).astype(np.float16)      # integers: exact in fp16
    f = torch.from_numpy(feat).to(DEV).requires_grad_(True)
    r = torch.from_numpy(rois).to(DEV)
    out = RoIAlignFunction.apply(f, r, 7, scale, sample_num)
    assert out.dtype == torch.float16
    exp = build_ref.roi_align_reference(feat.astype(np.float32), rois.astype(np.float32), scale, 7, sample_num)
    assert np.abs(out.detach().float().cpu().numpy() - exp).max() < 2e-3 * max(1.0, np.abs(exp).max())
    g = rs.standard_normal(tuple(out.shape)).astype(np.float16)
    out.backward(torch.from_numpy(g).to(DEV))
    assert f.grad.dtype == torch.float16
    expg = build_ref.roi_align_reference_backward(g.astype(np.float32), rois.astype(np.float32), scale,
                                                  (N, C, H, W), sample_num)
    assert np.abs(f.grad.float().cpu().numpy() - expg).max() < 4e-3 * max(1.0, np.abs(expg).max())


def test_roi_align_cuda_passes_the_references_own_gradcheck_recipe():
    """mmdet/ops/roi_align/gradcheck.py:11-30, restated (the file does not travel): 2 images of 16 x 15 x 15
    features, 20 RoIs in the lower-right half, ``RoIAlign(3, 1 / 8)`` (sample_num = 0: the adaptive grid) and
    ``RoIAlign(3, 1 / 8, 2)``, ``gradcheck(..., atol=1e-3, eps=1e-3)`` — through the compat module."""
    from torch.autograd import gradcheck
    rs = np.random.RandomState(0)
    feat_size, spatial_scale, num_imgs, num_rois = 15, 1.0 / 8, 2, 20
    img_size = feat_size / spatial_scale
    batch_ind = rs.randint(num_imgs, size=(num_rois, 1))
    rois = rs.rand(num_rois, 4) * img_size * 0.5
    rois[:, 2:] += img_size * 0.5
    rois = np.hstack((batch_ind, rois))
    g = torch.Generator().manual_seed(0)
    feat = torch.randn(num_imgs, 16, feat_size, feat_size, generator=g).to(DEV).requires_grad_(True)
    rois = torch.from_numpy(rois).float().to(DEV)

    class RoIAlign(torch.nn.Module):                 # mmdet/ops/roi_align/roi_align.py:58-86 (use_torchvision=False)
        def __init__(self, out_size, spatial_scale, sample_num=0):
            super().__init__()
            self.out_size, self.spatial_scale, self.sample_num = out_size, float(spatial_scale), int(sample_num)

        def forward(self, features, rois):
            return RoIAlignFunction.apply(features, rois, self.out_size, self.spatial_scale, self.sample_num)

    assert gradcheck(RoIAlign(3, spatial_scale), (feat, rois), atol=1e-3, eps=1e-3)
    assert gradcheck(RoIAlign(3, spatial_scale, 2), (feat, rois), atol=1e-3, eps=1e-3)


def test_roi_align_cuda_error_convention():
    f = torch.zeros((1, 4, 8, 8), device=DEV)
    out = torch.zeros((3, 4, 7, 7), device=DEV)
    bad = torch.zeros((3, 4), device=DEV)
    assert roi_align_cuda.forward(f, bad, 7, 7, 1.0, 2, out) == 0          # "wrong roi size", rc 0
    assert roi_align_cuda.backward(out, bad, 7, 7, 1.0, 2, f) == 0
    with pytest.raises(RuntimeError, match='CUDAtensor'):
        roi_align_cuda.forward(f.cpu(), torch.zeros((3, 5)), 7, 7, 1.0, 2, out.cpu())
    with pytest.raises(RuntimeError, match='contiguous'):
        roi_align_cuda.forward(f.permute(0, 1, 3, 2)[..., ::2], torch.zeros((3, 5), device=DEV), 7, 7,
                               1.0, 2, out)
    with pytest.raises(NotImplementedError):          # float64: the one dtype of the reference's dispatch without a kernel
        roi_align_cuda.forward(f.double(), torch.zeros((3, 5), device=DEV).double(), 7, 7, 1.0, 2, out.double())
    with pytest.raises(RuntimeError, match='like the other operands'):
        roi_align_cuda.forward(f.half(), torch.zeros((3, 5), device=DEV), 7, 7, 1.0, 2, out.half())
    empty = torch.zeros((0, 5), device=DEV)
    assert roi_align_cuda.forward(f, empty, 7, 7, 1.0, 2, torch.zeros((0, 4, 7, 7), device=DEV)) == 1


def nms_wrapper(dets, iou_thr):
    """mmdet/ops/nms/nms_wrapper.py:26-49 for a CUDA tensor input."""
    dets_th = dets
    if dets_th.shape[0] == 0:
        inds = dets_th.new_zeros(0, dtype=torch.long)
    else:
        assert dets_th.is_cuda
        inds = nms_cuda.nms(dets_th, iou_thr)
    return dets[inds, :], inds


@pytest.mark.parametrize('n,thr', [(1, 0.5), (63, 0.5), (64, 0.7), (700, 0.7), (2000, 0.5), (2000, 0.3)])
def test_nms_cuda_signature_unsorted_dets_original_order_indices(n, thr):
    dets = det_oracle.make_boxes(n, seed=n + int(thr * 10))
    rs = np.random.RandomState(n)
    dets = dets[rs.permutation(n)]                                  # ANY order in
    kept, inds = nms_wrapper(torch.from_numpy(dets).to(DEV), thr)
    assert inds.dtype == torch.long and inds.is_cuda
    exp = det_oracle.nms(dets, thr, mode='cuda')                    # the `>` rule of nms_kernel.cu:60
    np.testing.assert_array_equal(inds.cpu().numpy(), exp)          # ascending original indices
    np.testing.assert_array_equal(kept.cpu().numpy(), dets[exp])
    # the compiled nms_cpu.cpp (`>=`) agrees wherever no IoU equals the threshold exactly
    ref = build_ref.load_nms_cpu()
    if ref is not None:
        cpu_keep = ref.nms(torch.from_numpy(dets), thr).numpy()
        if len(cpu_keep) == len(exp):
            np.testing.assert_array_equal(cpu_keep, exp)


def test_nms_cuda_empty_and_cpu_inputs():
    out = nms_cuda.nms(torch.zeros((0, 5), device=DEV), 0.5)
    assert out.dtype == torch.long and out.numel() == 0 and not out.is_cuda     # nms_cuda.cpp:12-13
    with pytest.raises(RuntimeError, match='CUDAtensor'):
        nms_cuda.nms(torch.zeros((4, 5)), 0.5)


def _integration_snippet():
    """The ```python block of INTEGRATION.md that defines the reference-side ctypes binding."""
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', text, flags=re.S)
    code = [b for b in blocks if 'class GroupSoftmaxLoss' in b and 'def gs_prepare' in b]
    assert len(code) == 1
    return code[0].replace("ctypes.CDLL('libbgs.so')", 'ctypes.CDLL(%r)' % capi.lib_path())


@pytest.mark.parametrize('name', ['n512_cfg1', 'n1024_cfg2', 'n256_ratio2', 'n96_9bins'])
def test_integration_md_reference_side_binding_is_executed(name):
    """``gs_prepare`` + ``GroupSoftmaxLoss`` exactly as printed (the
    drop-in body of ``GSBBoxHeadWith0.loss``, gs_bbox_head_with0.py:147-171) against the losses /
    gradients the EXECUTED reference class produced for the same inputs and sampled weights."""
    ns = {}
    exec(compile(_integration_snippet(), 'INTEGRATION.md', 'exec'), ns)
    case, l2b, ps, fg_splits, cls_w, batch = case_setup(name)
    g = golden()
    w, avg = g.get(name, 'weights'), g.get(name, 'avg')
    labels = torch.from_numpy(batch['labels']).to(DEV)
    l2b_t = torch.from_numpy(l2b).to(DEV)
    # label remap through the printed gs_prepare (bit-exact gather; its own sampling is device RNG)
    bl, w_dev, avg_dev = ns['gs_prepare'](labels, l2b_t, float(case.get('ratio', 8.0)), 1234)
    np.testing.assert_array_equal(bl.cpu().numpy(), gs_oracle.remap_labels(batch['labels'], l2b))
    assert tuple(w_dev.shape) == tuple(w.shape) and float(avg_dev.min()) >= 1.0
    # loss + gradient with the reference's own sampled weights
    z = torch.from_numpy(batch['logits']).to(DEV).requires_grad_(True)
    per_bin = ns['GroupSoftmaxLoss'].apply(z, bl, np.asarray(ps), torch.from_numpy(w).to(DEV),
                                           torch.from_numpy(avg).to(DEV))
    np.testing.assert_allclose(per_bin.detach().cpu().numpy(), g.get(name, 'losses'), rtol=1e-4,
                               atol=1e-5)
    per_bin.sum().backward()
    rows = g.get(name, 'grad_rows')
    np.testing.assert_allclose(z.grad.cpu().numpy()[rows], g.get(name, 'grad_sub'), rtol=1e-4,
                               atol=1e-7)
