"""Signature-compatible stand-ins for the two pybind11 extension modules the reference imports
on this path, backed by ``libbgs.so``:

* :mod:`.roi_align_cuda` — ``mmdet/ops/roi_align/src/roi_align_cuda.cpp:27-85``
  (``forward`` / ``backward``; imported by ``mmdet/ops/roi_align/roi_align.py:6``);
* :mod:`.nms_cuda`       — ``mmdet/ops/nms/src/nms_cuda.cpp:8-17`` (``nms``; imported by
  ``mmdet/ops/nms/nms_wrapper.py:4``).

A maintainer of the reference drops them in without touching any caller::

    import sys
    from balancedgroupsoftmax_amd.compat import roi_align_cuda, nms_cuda
    sys.modules['mmdet.ops.roi_align.roi_align_cuda'] = roi_align_cuda
    sys.modules['mmdet.ops.nms.nms_cuda'] = nms_cuda

Same argument order, layouts (NCHW features / outputs, unsorted ``dets``), ownership (the caller
allocates ``output`` / ``bottom_grad``), return values (``1`` / ``0`` + "wrong roi size",
original-order keep indices) and input checks (CUDA + contiguous) as the extensions.  The native
ABI underneath is NHWC / pre-sorted (``include/bgs.h``); the transposes and the score sort are done
here, on the device.
"""
from . import nms_cuda, roi_align_cuda  # noqa: F401
