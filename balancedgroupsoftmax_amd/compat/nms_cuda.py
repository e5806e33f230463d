"""``nms_cuda`` of the reference (mmdet/ops/nms/src/nms_cuda.cpp:8-17 -> nms_kernel.cu:70-131) over
``bgs_nms_batched`` (csrc/nms.hip).

``nms(dets, threshold) -> LongTensor``: ``dets [n, 5]`` = (x1, y1, x2, y2, score) in ANY order on
the GPU; returns the kept rows' ORIGINAL indices in ascending order (nms_kernel.cu:127-130), on
the same device — greedy suppression in descending-score order, IoU with the legacy ``+1`` extents,
suppressed when ``IoU > threshold`` (nms_kernel.cu:19-31, 60).  ``dets.numel() == 0`` returns an
empty CPU LongTensor like nms_cuda.cpp:12-13.  A non-CUDA ``dets`` raises (``CHECK_CUDA``).

The result has a data-dependent length, so — like the reference extension, which copies its
suppression mask to the host — this call synchronises; the training path of this package never
calls it (fixed-shape ``functional.nms_batched``).
"""
import torch

from .. import functional as BF


def nms(dets, threshold):
    if not dets.is_cuda:
        raise RuntimeError('dets must be a CUDAtensor ')               # CHECK_CUDA
    if dets.numel() == 0:
        return torch.empty((0,), dtype=torch.long)                     # CPU, nms_cuda.cpp:12-13
    assert dets.dim() == 2 and dets.size(1) == 5, tuple(dets.shape)
    d = dets.detach().to(torch.float32)
    order = d[:, 4].sort(0, descending=True)[1]                        # nms_kernel.cu:74-75
    boxes = d.index_select(0, order).contiguous().unsqueeze(0)         # [1, n, 5]
    counts = torch.full((1,), d.size(0), dtype=torch.int32, device=d.device)
    keep, keep_count = BF.nms_batched(boxes, counts, float(threshold), iou_mode=0)
    k = int(keep_count[0])                                             # the one host sync
    kept_sorted_pos = keep[0, :k].long()
    return order.index_select(0, kept_sorted_pos).sort(0)[0]           # nms_kernel.cu:127-130
