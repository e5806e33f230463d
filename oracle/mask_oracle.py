"""TEST INFRASTRUCTURE ONLY (CPU oracle) — the mask branch of the BAGS Mask R-CNN (cfg 4).

* ``resize_linear_u8``   cv2.resize(src, (w, h), interpolation=INTER_LINEAR) for uint8 images, i.e.
  what ``mmcv.imresize`` (mmcv 0.2.x, default 'bilinear') calls from
  mmdet/core/mask/mask_target.py:31.  OpenCV is NOT installed in this image and is not part of the
  reference tree: this follows OpenCV's published fixed-point algorithm
  (modules/imgproc/src/resize.cpp: INTER_RESIZE_COEF_BITS = 11, HResizeLinear + VResizeLinear
  with FixedPtCast<int, uchar, 22>) — **parity unpinned** against an executed cv2.
* ``mask_target_single`` mmdet/core/mask/mask_target.py:16-38 around it (pinned by reading only).
* ``mask_cross_entropy`` mmdet/models/losses/cross_entropy_loss.py:54-61 — pinned against the
  EXECUTED reference ``FCNMaskHead`` (tests/golden/make_golden_mask.py).
* ``fill_mask_head``     seeded parameter values shared by the golden generator and the tests
  (the fixture stores outputs only).
"""
import numpy as np

F32 = np.float32


def _coef(v):
    """saturate_cast<short>(v * 2048): cvRound = round half to even."""
    return int(np.clip(np.rint(np.float32(v) * np.float32(2048.0)), -32768, 32767))


def resize_linear_u8(src, dsize):
    """src ``[h, w]`` uint8 -> ``[dh, dw]`` uint8; ``dsize = (dw, dh)`` as cv2 takes it."""
    src = np.asarray(src, dtype=np.uint8)
    h, w = src.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    if (dw, dh) == (w, h):
        return src.copy()
    sx_scale, sy_scale = float(w) / dw, float(h) / dh
    xs, ax = [], []
    for dx in range(dw):
        fx = np.float32((dx + 0.5) * sx_scale - 0.5)
        sx = int(np.floor(fx))
        fx = np.float32(fx - np.float32(sx))
        if sx < 0:
            fx, sx = np.float32(0), 0
        if sx >= w - 1:
            fx, sx = np.float32(0), w - 1
        xs.append(sx)
        ax.append((_coef(np.float32(1.0) - fx), _coef(fx)))
    out = np.zeros((dh, dw), dtype=np.uint8)
    s = src.astype(np.int64)
    for dy in range(dh):
        fy = np.float32((dy + 0.5) * sy_scale - 0.5)
        sy = int(np.floor(fy))
        fy = np.float32(fy - np.float32(sy))
        b0, b1 = _coef(np.float32(1.0) - fy), _coef(fy)
        r0, r1 = min(max(sy, 0), h - 1), min(max(sy + 1, 0), h - 1)
        for dx in range(dw):
            sx = xs[dx]
            sx1 = min(sx + 1, w - 1)
            a0, a1 = ax[dx]
            S0 = int(s[r0, sx]) * a0 + int(s[r0, sx1]) * a1
            S1 = int(s[r1, sx]) * a0 + int(s[r1, sx1]) * a1
            v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
            out[dy, dx] = np.uint8(min(max(v, 0), 255))
    return out


def mask_target_single(pos_proposals, pos_assigned_gt_inds, gt_masks, mask_size):
    """``[P,4]`` boxes, ``[P]`` gt indices, ``[G,H,W]`` uint8 -> ``[P,S,S]`` float32."""
    pos_proposals = np.asarray(pos_proposals, dtype=F32)
    out = []
    for i in range(pos_proposals.shape[0]):
        gt = gt_masks[int(pos_assigned_gt_inds[i])]
        x1, y1, x2, y2 = pos_proposals[i, :4].astype(np.int32)
        w = max(x2 - x1 + 1, 1)
        h = max(y2 - y1 + 1, 1)
        crop = gt[y1:y1 + h, x1:x1 + w]
        out.append(resize_linear_u8(crop, (mask_size, mask_size)))
    if not out:
        return np.zeros((0, mask_size, mask_size), dtype=F32)
    return np.stack(out).astype(F32)


def make_gt_masks(num, H, W, boxes, seed):
    """Synthetic GT bitmaps: an ellipse or a rectangle inside each box (SURVEY §8d cfg 4)."""
    rs = np.random.RandomState(seed)
    masks = np.zeros((num, H, W), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    for g in range(num):
        x1, y1, x2, y2 = boxes[g]
        if rs.rand() < 0.5:
            cx, cy = (x1 + x2) / 2.0, (y1 + y2) / 2.0
            rx, ry = max((x2 - x1) / 2.0, 1.0), max((y2 - y1) / 2.0, 1.0)
            masks[g] = (((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1.0).astype(np.uint8)
        else:
            masks[g, int(max(y1, 0)):int(y2) + 1, int(max(x1, 0)):int(x2) + 1] = 1
    return masks


def mask_cross_entropy(pred_slice, target):
    """mean BCE-with-logits over ``[P,S,S]`` (cross_entropy_loss.py:54-61), float64."""
    z = np.asarray(pred_slice, dtype=np.float64)
    t = np.asarray(target, dtype=np.float64)
    return float((np.maximum(z, 0) - z * t + np.log1p(np.exp(-np.abs(z)))).mean())


def fill_mask_head(state_dict, seed):
    """In-place seeded values for every tensor of an FCNMaskHead state_dict (name order):
    He-like scale for weights, small biases.  Works on torch tensors of either implementation."""
    import torch
    rs = np.random.RandomState(seed)
    for name in sorted(state_dict.keys()):
        t = state_dict[name]
        if name.endswith('weight'):
            fan_in = int(np.prod(t.shape[1:])) if t.dim() > 1 else int(t.shape[0])
            v = rs.standard_normal(tuple(t.shape)).astype(F32) * F32(np.sqrt(2.0 / max(fan_in, 1)))
        else:
            v = (rs.standard_normal(tuple(t.shape)) * 0.05).astype(F32)
        t.copy_(torch.from_numpy(v))
    return state_dict
