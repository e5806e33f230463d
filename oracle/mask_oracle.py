"""TEST INFRASTRUCTURE ONLY (CPU oracle) — the mask branch of the BAGS Mask R-CNN (cfg 4).

* ``resize_linear_u8``   cv2.resize(src, (w, h), interpolation=INTER_LINEAR) for uint8 images, i.e.
  what ``mmcv.imresize`` (mmcv 0.2.x, default 'bilinear') calls from
  mmdet/core/mask/mask_target.py:31.  OpenCV is NOT installed in this image and is not part of the
  reference tree: this follows OpenCV's published fixed-point algorithm
  (modules/imgproc/src/resize.cpp: INTER_RESIZE_COEF_BITS = 11, HResizeLinear + VResizeLinear
  with FixedPtCast<int, uchar, 22>) — **parity unpinned** against an executed cv2.
* ``mask_target_single`` mmdet/core/mask/mask_target.py:16-38 around it (pinned by reading only).
* ``resize_linear_f32`` / ``seg_masks_dense``  the float32 resize and the paste of ``FCNMaskHead.get_seg_masks``
  (fcn_mask_head.py:125-181) without the RLE step; the paste logic is pinned against the EXECUTED reference
  method with this resize injected as ``mmcv.imresize`` (tests/test_mask_cpu.py), the resize itself is unpinned.
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


def resize_linear_f32(src, dsize):
    """cv2.resize(src float32 [h, w], (dw, dh), interpolation=INTER_LINEAR) — what ``mmcv.imresize`` does to the
    28 x 28 mask probabilities in ``FCNMaskHead.get_seg_masks`` (fcn_mask_head.py:170).  Follows OpenCV's float path
    (modules/imgproc/src/resize.cpp: scale = 1 / (dsize / ssize) in double; fx = (float)((dx + 0.5) * scale - 0.5);
    sx < 0 -> (0, fx = 0); sx >= w - 1 -> (w - 1, fx = 0); row indices clipped with fy kept; HResizeLinear
    ``S[sx] * (1 - fx) + S[sx + 1] * fx`` — ``S[sx]`` alone where sx + 1 leaves the row — then VResizeLinear
    ``row0 * (1 - fy) + row1 * fy``, float32 throughout; an unchanged size returns the source).  **Parity
    unpinned** against an executed cv2 (not installed; its SIMD paths may fuse the multiply-adds, and an exact 2x
    reduction takes the INTER_AREA fast path there: both move a value by an ulp at most, which the > 0.5
    threshold behind it only sees on exact ties)."""
    src = np.asarray(src, dtype=F32)
    h, w = src.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    if (dw, dh) == (w, h):
        return src.copy()
    sx_scale, sy_scale = 1.0 / (float(dw) / w), 1.0 / (float(dh) / h)
    out = np.zeros((dh, dw), dtype=F32)
    cols = []
    for dx in range(dw):
        fx = F32((dx + 0.5) * sx_scale - 0.5)
        sx = int(np.floor(fx))
        fx = F32(fx - F32(sx))
        if sx < 0:
            fx, sx = F32(0), 0
        if sx >= w - 1:
            fx, sx = F32(0), w - 1
        cols.append((sx, sx + 1 if sx + 1 < w else -1, F32(1) - fx, fx))
    for dy in range(dh):
        fy = F32((dy + 0.5) * sy_scale - 0.5)
        sy = int(np.floor(fy))
        fy = F32(fy - F32(sy))
        r0, r1 = min(max(sy, 0), h - 1), min(max(sy + 1, 0), h - 1)
        b0, b1 = F32(1) - fy, fy
        for dx, (c0, c1, a0, a1) in enumerate(cols):
            if c1 >= 0:
                h0 = F32(F32(src[r0, c0] * a0) + F32(src[r0, c1] * a1))
                h1 = F32(F32(src[r1, c0] * a0) + F32(src[r1, c1] * a1))
            else:
                h0, h1 = src[r0, c0], src[r1, c0]
            out[dy, dx] = F32(F32(h0 * b0) + F32(h1 * b1))
    return out


def seg_masks_dense(mask_probs, bboxes, scale_factor, thr, img_h, img_w, return_margin=False):
    """fcn_mask_head.py:156-176 without the RLE step: ``mask_probs [n, S, S]`` float32 (the detection's own class
    channel after the sigmoid), ``bboxes [n, >=4]`` -> ``uint8 [n, img_h, img_w]``.  The part of a box that leaves
    the image is clipped (numpy's slice assignment raises there).  ``return_margin``: also ``[n, img_h, img_w]``
    float32 |value - thr| inside the boxes (inf outside): the pixels a differently rounded resize could flip."""
    mask_probs = np.asarray(mask_probs, dtype=F32)
    bboxes = np.asarray(bboxes, dtype=F32)[:, :4]
    n = bboxes.shape[0]
    out = np.zeros((n, img_h, img_w), dtype=np.uint8)
    margin = np.full((n, img_h, img_w), np.inf, dtype=F32) if return_margin else None
    for i in range(n):
        bbox = (bboxes[i, :] / F32(scale_factor)).astype(np.int32)
        w = max(bbox[2] - bbox[0] + 1, 1)
        h = max(bbox[3] - bbox[1] + 1, 1)
        bm = resize_linear_f32(mask_probs[i], (w, h))
        y0, x0 = int(bbox[1]), int(bbox[0])
        ys, xs = max(y0, 0), max(x0, 0)
        ye, xe = min(y0 + h, img_h), min(x0 + w, img_w)
        if ye <= ys or xe <= xs:
            continue
        sub = bm[ys - y0:ye - y0, xs - x0:xe - x0]
        out[i, ys:ye, xs:xe] = (sub > F32(thr)).astype(np.uint8)
        if return_margin:
            margin[i, ys:ye, xs:xe] = np.abs(sub - F32(thr))
    return (out, margin) if return_margin else out


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
