"""``roi_align_cuda`` of the reference (mmdet/ops/roi_align/src/roi_align_cuda.cpp:27-85) over
``bgs_roi_align_nhwc_fwd`` / ``bgs_roi_align_nhwc_bwd`` (csrc/roi_align.hip).

``forward(features, rois, pooled_height, pooled_width, spatial_scale, sample_num, output) -> int``
``backward(top_grad, rois, pooled_height, pooled_width, spatial_scale, sample_num, bottom_grad) -> int``

* ``features`` / ``bottom_grad`` ``[N, C, H, W]``, ``output`` / ``top_grad`` ``[K, C, ph, pw]``,
  ``rois [K, 5]`` = (batch index, x1, y1, x2, y2): CUDA, contiguous, float32 (``CHECK_INPUT``,
  roi_align_cuda.cpp:20-25: anything else raises);
* the caller owns ``output`` (``features.new_zeros(...)``, roi_align.py:23) and ``bottom_grad``
  (``rois.new_zeros(...)``, roi_align.py:45-46); ``backward`` ACCUMULATES into ``bottom_grad``
  exactly like the reference kernel's ``atomicAdd`` (roi_align_kernel.cu:243-262);
* returns 1; ``rois.size(1) != 5`` prints "wrong roi size" and returns 0 (:39-42, :69-72);
* one feature level, so the level rule of the native entry point is moot; legacy box semantics
  (``roi_end = (x2 + 1) * scale``) as in roi_align_kernel.cu:82-90;
* only ``sample_num == 2`` has a kernel (every shipped config: ``sample_num=2``): other values
  raise ``NotImplementedError`` instead of silently sampling differently.
"""
import ctypes

import torch

from .. import capi


def _check_input(t, name):
    if not t.is_cuda:
        raise RuntimeError('%s must be a CUDAtensor ' % name)          # CHECK_CUDA
    if not t.is_contiguous():
        raise RuntimeError('%s must be contiguous ' % name)            # CHECK_CONTIGUOUS
    if t.dtype != torch.float32:
        raise RuntimeError('%s must be float32 (the only dtype of the BAGS configs)' % name)


def _level_args(t_nhwc, spatial_scale):
    return ((ctypes.c_int * 1)(int(t_nhwc.shape[1])), (ctypes.c_int * 1)(int(t_nhwc.shape[2])),
            (ctypes.c_float * 1)(float(spatial_scale)))


def forward(features, rois, pooled_height, pooled_width, spatial_scale, sample_num, output):
    for t, n in ((features, 'features'), (rois, 'rois'), (output, 'output')):
        _check_input(t, n)
    if rois.size(1) != 5:
        print('wrong roi size')
        return 0
    if int(sample_num) != 2:
        raise NotImplementedError('roi_align_cuda.forward: sample_num=%r (libbgs has the '
                                  'sample_num=2 kernel of the shipped configs)' % (sample_num,))
    lib = capi.load()
    N, C, H, W = features.shape
    K = rois.size(0)
    ph, pw = int(pooled_height), int(pooled_width)
    assert tuple(output.shape) == (K, C, ph, pw), (tuple(output.shape), (K, C, ph, pw))
    if K == 0:
        return 1
    x = features.permute(0, 2, 3, 1).contiguous()                      # NHWC for the kernel
    out = torch.empty((K, ph, pw, C), dtype=torch.float32, device=features.device)
    hs, ws, sc = _level_args(x, spatial_scale)
    ptrs = (ctypes.c_void_p * 1)(x.data_ptr())
    rc = lib.bgs_roi_align_nhwc_fwd(ptrs, hs, ws, sc, 1, N, 56.0, capi.ptr(rois), K, C, ph, pw, 2,
                                    capi.ptr(out), None, capi.current_stream(features.device))
    capi.check('bgs_roi_align_nhwc_fwd', rc)
    output.copy_(out.permute(0, 3, 1, 2))
    return 1


def backward(top_grad, rois, pooled_height, pooled_width, spatial_scale, sample_num, bottom_grad):
    for t, n in ((top_grad, 'top_grad'), (rois, 'rois'), (bottom_grad, 'bottom_grad')):
        _check_input(t, n)
    if rois.size(1) != 5:
        print('wrong roi size')
        return 0
    if int(sample_num) != 2:
        raise NotImplementedError('roi_align_cuda.backward: sample_num=%r' % (sample_num,))
    lib = capi.load()
    N, C, H, W = bottom_grad.shape
    K = rois.size(0)
    ph, pw = int(pooled_height), int(pooled_width)
    assert tuple(top_grad.shape) == (K, C, ph, pw), (tuple(top_grad.shape), (K, C, ph, pw))
    if K == 0:
        return 1
    dout = top_grad.permute(0, 2, 3, 1).contiguous()
    dfeat = torch.zeros((N, H, W, C), dtype=torch.float32, device=bottom_grad.device)
    hs, ws, sc = _level_args(dfeat, spatial_scale)
    ptrs = (ctypes.c_void_p * 1)(dfeat.data_ptr())
    rc = lib.bgs_roi_align_nhwc_bwd(ptrs, hs, ws, sc, 1, N, 56.0, capi.ptr(rois), K, C, ph, pw, 2,
                                    capi.ptr(dout), capi.current_stream(bottom_grad.device))
    capi.check('bgs_roi_align_nhwc_bwd', rc)
    bottom_grad.add_(dfeat.permute(0, 3, 1, 2))                        # the kernel's atomicAdd
    return 1
