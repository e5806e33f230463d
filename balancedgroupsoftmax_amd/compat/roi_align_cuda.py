"""``roi_align_cuda`` of the reference (mmdet/ops/roi_align/src/roi_align_cuda.cpp:27-85) over
``bgs_roi_align_nhwc_fwd`` / ``bgs_roi_align_nhwc_bwd`` and their fp16 forms (csrc/roi_align.hip).

``forward(features, rois, pooled_height, pooled_width, spatial_scale, sample_num, output) -> int``
``backward(top_grad, rois, pooled_height, pooled_width, spatial_scale, sample_num, bottom_grad) -> int``

* ``features`` / ``bottom_grad`` ``[N, C, H, W]``, ``output`` / ``top_grad`` ``[K, C, ph, pw]``,
  ``rois [K, 5]`` = (batch index, x1, y1, x2, y2): CUDA and contiguous (``CHECK_INPUT``,
  roi_align_cuda.cpp:20-25: anything else raises), all of ONE floating dtype — float32 or float16
  (the reference's kernels are instantiated for double, float and half, roi_align_kernel.cu:136,281;
  float64 raises ``NotImplementedError`` here: no BAGS config or test of the reference uses it);
* the caller owns ``output`` (``features.new_zeros(...)``, roi_align.py:23) and ``bottom_grad``
  (``rois.new_zeros(...)``, roi_align.py:45-46); ``backward`` ACCUMULATES into ``bottom_grad``
  exactly like the reference kernel's ``atomicAdd`` (roi_align_kernel.cu:243-262);
* returns 1; ``rois.size(1) != 5`` prints "wrong roi size" and returns 0 (:39-42, :69-72);
* ``sample_num``: any value >= 0 as in the reference (roi_align_kernel.cu:95-99) — n > 0 samples an n x n grid
  per bin, 0 = adaptive ``ceil(roi_size / pooled_size)`` per axis (``RoIAlign(3, 1 / 8)`` of the reference's own
  mmdet/ops/roi_align/gradcheck.py:29);
* one feature level, so the level rule of the native entry point is moot; legacy box semantics
  (``roi_end = (x2 + 1) * scale``) as in roi_align_kernel.cu:82-90;
* fp16: arithmetic in fp32, results rounded once (the reference's half kernel rounds after every
  operation: this path is the more accurate one); gradients are accumulated in fp32 and added into the
  fp16 ``bottom_grad`` once.
"""
import ctypes

import torch

from .. import capi

_DTYPES = (torch.float32, torch.float16)


def _check_input(t, name, dtype):
    if not t.is_cuda:
        raise RuntimeError('%s must be a CUDAtensor ' % name)          # CHECK_CUDA
    if not t.is_contiguous():
        raise RuntimeError('%s must be contiguous ' % name)            # CHECK_CONTIGUOUS
    if t.dtype == torch.float64:
        raise NotImplementedError('%s: float64 has no kernel in libbgs (float32 / float16 do)' % name)
    if t.dtype not in _DTYPES or t.dtype != dtype:
        raise RuntimeError('%s must be %s like the other operands (float32 or float16)' % (name, dtype))


def _level_args(t_nhwc, spatial_scale):
    return ((ctypes.c_int * 1)(int(t_nhwc.shape[1])), (ctypes.c_int * 1)(int(t_nhwc.shape[2])),
            (ctypes.c_float * 1)(float(spatial_scale)))


def forward(features, rois, pooled_height, pooled_width, spatial_scale, sample_num, output):
    for t, n in ((features, 'features'), (rois, 'rois'), (output, 'output')):
        _check_input(t, n, features.dtype)
    if rois.size(1) != 5:
        print('wrong roi size')
        return 0
    sample_num = int(sample_num)
    if sample_num < 0:
        raise ValueError('sample_num must be >= 0')
    lib = capi.load()
    N, C, H, W = features.shape
    K = rois.size(0)
    ph, pw = int(pooled_height), int(pooled_width)
    assert tuple(output.shape) == (K, C, ph, pw), (tuple(output.shape), (K, C, ph, pw))
    if K == 0:
        return 1
    half = features.dtype == torch.float16
    x = features.permute(0, 2, 3, 1).contiguous()                      # NHWC for the kernel
    out = torch.empty((K, ph, pw, C), dtype=features.dtype, device=features.device)
    hs, ws, sc = _level_args(x, spatial_scale)
    ptrs = (ctypes.c_void_p * 1)(x.data_ptr())
    r32 = rois.float() if half else rois
    st = capi.current_stream(features.device)
    if half:
        rc = lib.bgs_roi_align_nhwc_fwd_f16(ptrs, hs, ws, sc, 1, N, 56.0, capi.ptr(r32), K, C, ph, pw, sample_num,
                                            capi.ptr(out), None, st)
        capi.check('bgs_roi_align_nhwc_fwd_f16', rc)
    else:
        rc = lib.bgs_roi_align_nhwc_fwd(ptrs, hs, ws, sc, 1, N, 56.0, capi.ptr(r32), K, C, ph, pw, sample_num,
                                        capi.ptr(out), None, st)
        capi.check('bgs_roi_align_nhwc_fwd', rc)
    output.copy_(out.permute(0, 3, 1, 2))
    return 1


def backward(top_grad, rois, pooled_height, pooled_width, spatial_scale, sample_num, bottom_grad):
    for t, n in ((top_grad, 'top_grad'), (rois, 'rois'), (bottom_grad, 'bottom_grad')):
        _check_input(t, n, top_grad.dtype)
    if rois.size(1) != 5:
        print('wrong roi size')
        return 0
    sample_num = int(sample_num)
    if sample_num < 0:
        raise ValueError('sample_num must be >= 0')
    lib = capi.load()
    N, C, H, W = bottom_grad.shape
    K = rois.size(0)
    ph, pw = int(pooled_height), int(pooled_width)
    assert tuple(top_grad.shape) == (K, C, ph, pw), (tuple(top_grad.shape), (K, C, ph, pw))
    if K == 0:
        return 1
    half = top_grad.dtype == torch.float16
    dout = top_grad.permute(0, 2, 3, 1).contiguous()
    dfeat = torch.zeros((N, H, W, C), dtype=torch.float32, device=bottom_grad.device)
    hs, ws, sc = _level_args(dfeat, spatial_scale)
    ptrs = (ctypes.c_void_p * 1)(dfeat.data_ptr())
    r32 = rois.float() if half else rois
    st = capi.current_stream(bottom_grad.device)
    if half:
        rc = lib.bgs_roi_align_nhwc_bwd_f16(ptrs, hs, ws, sc, 1, N, 56.0, capi.ptr(r32), K, C, ph, pw, sample_num,
                                            capi.ptr(dout), st)
        capi.check('bgs_roi_align_nhwc_bwd_f16', rc)
    else:
        rc = lib.bgs_roi_align_nhwc_bwd(ptrs, hs, ws, sc, 1, N, 56.0, capi.ptr(r32), K, C, ph, pw, sample_num,
                                        capi.ptr(dout), st)
        capi.check('bgs_roi_align_nhwc_bwd', rc)
    bottom_grad.add_(dfeat.permute(0, 3, 1, 2).to(bottom_grad.dtype))  # the kernel's atomicAdd
    return 1
