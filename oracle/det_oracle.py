"""TEST INFRASTRUCTURE ONLY (CPU oracle) — numpy restatements of the reference's native ops
on the detector path.  Never imported by the product package.

* ``roi_align_forward``  mmdet/ops/roi_align/src/roi_align_kernel.cu:16-124 (the reference has
  NO CPU RoIAlign: roi_align.py:27-28 raises).  PINNED: oracle/build_ref.py compiles the
  reference's own ``ROIAlignForward`` / ``ROIAlignBackward`` templates from that file as host
  code (oracle/_ref/roi_align_ref.so) and tests/test_oracle_det.py checks this restatement
  against them (forward 1e-6, backward 1e-4).
* ``roi_align_backward`` roi_align_kernel.cu:149-266 (pinned as above; additionally checked as the
  exact adjoint of the forward restatement).
* ``map_roi_levels``     mmdet/models/roi_extractors/single_level.py:54-73
* ``nms``                mmdet/ops/nms/src/nms_cpu.cpp:5-59 (``>=``) and
                         mmdet/ops/nms/src/nms_kernel.cu:13-131 (``>``); the ``>=`` flavour is
                         pinned against the COMPILED reference (oracle/_ref/nms_cpu_ref.so).
* ``conv2d_nhwc``        the reference delegates to nn.Conv2d (torch); the oracle is torch-CPU
                         F.conv2d in fp32 (fp64 for the "truth").
"""
import numpy as np

F32 = np.float32


def map_roi_levels(rois, num_levels, finest_scale=56):
    rois = np.asarray(rois, dtype=F32)
    scale = np.sqrt((rois[:, 3] - rois[:, 1] + F32(1)) * (rois[:, 4] - rois[:, 2] + F32(1)))
    lv = np.floor(np.log2(scale / F32(finest_scale) + F32(1e-6)))
    return np.clip(lv, 0, num_levels - 1).astype(np.int64)


def _bilinear(feat, H, W, y, x):
    """feat [H, W, C] -> [C]; roi_align_kernel.cu:16-61 in fp32."""
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return np.zeros(feat.shape[2], dtype=F32)
    y = F32(max(y, 0.0))
    x = F32(max(x, 0.0))
    y_low, x_low = int(y), int(x)
    if y_low >= H - 1:
        y_high = y_low = H - 1
        y = F32(y_low)
    else:
        y_high = y_low + 1
    if x_low >= W - 1:
        x_high = x_low = W - 1
        x = F32(x_low)
    else:
        x_high = x_low + 1
    ly, lx = F32(y - F32(y_low)), F32(x - F32(x_low))
    hy, hx = F32(1.0) - ly, F32(1.0) - lx
    w1, w2, w3, w4 = hy * hx, hy * lx, ly * hx, ly * lx
    return (w1 * feat[y_low, x_low] + w2 * feat[y_low, x_high] + w3 * feat[y_high, x_low] +
            w4 * feat[y_high, x_high]).astype(F32)


def roi_align_forward(feat_nhwc, rois, spatial_scale, out_h=7, out_w=7, sample_num=2):
    """Single level.  feat ``[N,H,W,C]`` fp32, rois ``[K,5]`` -> ``[K,out_h,out_w,C]``
    (the reference's ``[K,C,ph,pw]`` is the (0,3,1,2) transpose of this)."""
    feat = np.asarray(feat_nhwc, dtype=F32)
    rois = np.asarray(rois, dtype=F32)
    N, H, W, C = feat.shape
    K = rois.shape[0]
    ss = F32(spatial_scale)
    out = np.zeros((K, out_h, out_w, C), dtype=F32)
    for k in range(K):
        b = int(rois[k, 0])
        start_w, start_h = rois[k, 1] * ss, rois[k, 2] * ss
        end_w, end_h = (rois[k, 3] + F32(1)) * ss, (rois[k, 4] + F32(1)) * ss
        rw = F32(max(end_w - start_w, 0.0))
        rh = F32(max(end_h - start_h, 0.0))
        bh, bw = F32(rh / F32(out_h)), F32(rw / F32(out_w))
        for ph in range(out_h):
            for pw in range(out_w):
                acc = np.zeros(C, dtype=F32)
                for iy in range(sample_num):
                    y = F32(start_h + F32(ph) * bh + F32(iy + 0.5) * bh / F32(sample_num))
                    for ix in range(sample_num):
                        x = F32(start_w + F32(pw) * bw + F32(ix + 0.5) * bw / F32(sample_num))
                        acc += _bilinear(feat[b], H, W, y, x)
                out[k, ph, pw] = acc / F32(sample_num * sample_num)
    return out


def roi_align_backward(dout, rois, spatial_scale, feat_shape, sample_num=2):
    """Restatement of ROIAlignBackward (roi_align_kernel.cu:149-266), single level, fp64
    accumulation (the reference accumulates with fp32 atomics in arbitrary order).
    ``dout [K,out_h,out_w,C]`` -> ``dfeat [N,H,W,C]``."""
    dout = np.asarray(dout, dtype=np.float64)
    rois = np.asarray(rois, dtype=F32)
    N, H, W, C = feat_shape
    K, out_h, out_w, _ = dout.shape
    ss = F32(spatial_scale)
    dfeat = np.zeros(feat_shape, dtype=np.float64)
    count = float(sample_num * sample_num)
    for k in range(K):
        b = int(rois[k, 0])
        start_w, start_h = rois[k, 1] * ss, rois[k, 2] * ss
        end_w, end_h = (rois[k, 3] + F32(1)) * ss, (rois[k, 4] + F32(1)) * ss
        rw = F32(max(end_w - start_w, 0.0))
        rh = F32(max(end_h - start_h, 0.0))
        bh, bw = F32(rh / F32(out_h)), F32(rw / F32(out_w))
        for ph in range(out_h):
            for pw in range(out_w):
                g = dout[k, ph, pw]
                for iy in range(sample_num):
                    y = F32(start_h + F32(ph) * bh + F32(iy + 0.5) * bh / F32(sample_num))
                    for ix in range(sample_num):
                        x = F32(start_w + F32(pw) * bw + F32(ix + 0.5) * bw / F32(sample_num))
                        if y < -1.0 or y > H or x < -1.0 or x > W:
                            continue
                        yy, xx = F32(max(y, 0.0)), F32(max(x, 0.0))
                        y_low, x_low = int(yy), int(xx)
                        if y_low >= H - 1:
                            y_high = y_low = H - 1
                            yy = F32(y_low)
                        else:
                            y_high = y_low + 1
                        if x_low >= W - 1:
                            x_high = x_low = W - 1
                            xx = F32(x_low)
                        else:
                            x_high = x_low + 1
                        ly, lx = F32(yy - F32(y_low)), F32(xx - F32(x_low))
                        hy, hx = F32(1.0) - ly, F32(1.0) - lx
                        dfeat[b, y_low, x_low] += g * float(hy * hx) / count
                        dfeat[b, y_low, x_high] += g * float(hy * lx) / count
                        dfeat[b, y_high, x_low] += g * float(ly * hx) / count
                        dfeat[b, y_high, x_high] += g * float(ly * lx) / count
    return dfeat


def roi_align_multilevel(feats_nhwc, rois, strides, out_size=7, sample_num=2, finest_scale=56):
    """SingleRoIExtractor.forward (single_level.py:89-107)."""
    lv = map_roi_levels(rois, len(feats_nhwc), finest_scale)
    K = rois.shape[0]
    C = feats_nhwc[0].shape[3]
    out = np.zeros((K, out_size, out_size, C), dtype=F32)
    for i, (f, s) in enumerate(zip(feats_nhwc, strides)):
        idx = np.nonzero(lv == i)[0]
        if idx.size:
            out[idx] = roi_align_forward(f, rois[idx], 1.0 / s, out_size, out_size, sample_num)
    return out, lv


def nms(dets, thr, mode='cuda'):
    """Greedy NMS on ``[n,5]`` (x1,y1,x2,y2,score), legacy +1 areas.  ``mode='cuda'``:
    suppress on IoU > thr (nms_kernel.cu:60); ``'cpu'``: IoU >= thr (nms_cpu.cpp:55).
    Returns kept ORIGINAL indices in ascending order (nms_kernel.cu:127-130, nms_cpu.cpp:58)."""
    dets = np.asarray(dets, dtype=F32)
    n = dets.shape[0]
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    x1, y1, x2, y2, sc = (dets[:, i] for i in range(5))
    areas = (x2 - x1 + F32(1)) * (y2 - y1 + F32(1))
    order = np.argsort(-sc, kind='stable')
    suppressed = np.zeros(n, dtype=bool)
    thr = F32(thr)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(F32(0), xx2 - xx1 + F32(1))
        h = np.maximum(F32(0), yy2 - yy1 + F32(1))
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        hit = (ovr > thr) if mode == 'cuda' else (ovr >= thr)
        suppressed[rest[hit]] = True
    return np.nonzero(~suppressed)[0].astype(np.int64)


def conv2d_nhwc(x_nhwc, w_oihw, bias=None, stride=1, pad=0, relu=False, residual=None,
                dtype='float32'):
    """torch-CPU reference of the fused conv (what nn.Conv2d + folded BN + ReLU compute)."""
    import torch
    import torch.nn.functional as Fn
    dt = getattr(torch, dtype)
    x = torch.from_numpy(np.ascontiguousarray(x_nhwc)).to(dt).permute(0, 3, 1, 2)
    w = torch.from_numpy(np.ascontiguousarray(w_oihw)).to(dt)
    b = None if bias is None else torch.from_numpy(np.ascontiguousarray(bias)).to(dt)
    y = Fn.conv2d(x, w, b, stride=stride, padding=pad)
    if residual is not None:
        y = y + torch.from_numpy(np.ascontiguousarray(residual)).to(dt).permute(0, 3, 1, 2)
    if relu:
        y = torch.relu(y)
    return y.permute(0, 2, 3, 1).contiguous().numpy()


def make_boxes(n, seed, img_w=1333, img_h=800, cluster=True):
    """Score-unsorted synthetic detections with heavy overlaps (clustered around few centres)."""
    rs = np.random.RandomState(seed)
    if cluster:
        nc = max(1, n // 12)
        cx = rs.uniform(0, img_w, nc)[rs.randint(0, nc, n)] + rs.normal(0, 12, n)
        cy = rs.uniform(0, img_h, nc)[rs.randint(0, nc, n)] + rs.normal(0, 12, n)
    else:
        cx, cy = rs.uniform(0, img_w, n), rs.uniform(0, img_h, n)
    w = np.exp(rs.uniform(np.log(16), np.log(400), n))
    h = np.exp(rs.uniform(np.log(16), np.log(400), n))
    x1 = np.clip(cx - w / 2, 0, img_w - 1)
    y1 = np.clip(cy - h / 2, 0, img_h - 1)
    x2 = np.clip(cx + w / 2, 0, img_w - 1)
    y2 = np.clip(cy + h / 2, 0, img_h - 1)
    sc = rs.uniform(0, 1, n)
    return np.stack([x1, y1, x2, y2, sc], 1).astype(F32)


def make_multiclass_case(n, C, seed, agnostic=False, clusters=12, img=(800, 1333), peak=0.6):
    """Seeded test-time inputs: ``n`` RoIs clustered around ``clusters`` objects, class-specific
    (or agnostic) decoded boxes ``[n, 4C]`` and softmax-like scores ``[n, C]`` (column 0 = bg)."""
    rs = np.random.RandomState(seed)
    H, W = img
    ctr = rs.uniform(0.15, 0.85, size=(clusters, 2)) * np.array([W, H])
    size = rs.uniform(40, 300, size=(clusters, 2))
    which = rs.randint(0, clusters, size=n)
    c = ctr[which] + rs.normal(0, 6, size=(n, 2))
    s = size[which] * np.exp(rs.normal(0, 0.08, size=(n, 2)))
    base = np.concatenate([c - s / 2, c + s / 2], axis=1)                   # [n,4]
    if agnostic:
        boxes = base.astype(F32)
    else:
        boxes = (base[:, None, :] + rs.normal(0, 2.0, size=(n, C, 4))).reshape(n, 4 * C).astype(F32)
    logits = rs.normal(0, 1, size=(n, C))
    fav = rs.randint(1, C, size=clusters)[which]
    logits[np.arange(n), fav] += rs.uniform(2, 8, size=n) * peak / 0.6
    logits[:, 0] += 2.0
    e = np.exp(logits - logits.max(1, keepdims=True))
    scores = (e / e.sum(1, keepdims=True)).astype(F32)
    return boxes, scores


def multiclass_nms(multi_bboxes, multi_scores, score_thr, iou_thr, max_num=-1, mode='cuda'):
    """Restatement of mmdet/core/post_processing/bbox_nms.py:31-66 on numpy: classes 1..C-1 in
    order; per class the rows with score > thr, NMS, survivors in ORIGINAL row order (what both
    nms_cpu.cpp:58 and nms_kernel.cu:127-130 return); concatenated class-major; if more than
    ``max_num`` survive, the top ``max_num`` by score (descending)."""
    multi_bboxes = np.asarray(multi_bboxes, F32)
    multi_scores = np.asarray(multi_scores, F32)
    n, C = multi_scores.shape
    out_b, out_l = [], []
    for i in range(1, C):
        sel = multi_scores[:, i] > F32(score_thr)
        if not sel.any():
            continue
        b = multi_bboxes[sel] if multi_bboxes.shape[1] == 4 else multi_bboxes[sel, 4 * i:4 * i + 4]
        d = np.concatenate([b, multi_scores[sel, i][:, None]], axis=1)
        keep = nms(d, iou_thr, mode)
        out_b.append(d[keep])
        out_l.append(np.full(keep.shape[0], i - 1, dtype=np.int64))
    if not out_b:
        return np.zeros((0, 5), F32), np.zeros((0,), np.int64)
    bb, ll = np.concatenate(out_b), np.concatenate(out_l)
    if max_num >= 0 and bb.shape[0] > max_num:
        order = np.argsort(-bb[:, 4], kind='stable')[:max_num]
        bb, ll = bb[order], ll[order]
    return bb, ll


def fill_detector(state_dict, seed):
    """In-place seeded values for EVERY tensor of a detector state_dict (sorted name order), for
    the end-to-end goldens (tests/golden/make_golden_e2e.py): He-scaled conv / fc weights, small
    biases, BatchNorm statistics that keep an eval-mode trunk well conditioned (variance in
    [0.5, 1.5], the residual branch's last BN scaled down so that 16 bottlenecks do not blow the
    activations up).  Works on the reference's and on this package's modules alike (same names)."""
    import torch
    rs = np.random.RandomState(seed)
    for name in sorted(state_dict.keys()):
        t = state_dict[name]
        shape = tuple(t.shape)
        if name.endswith('num_batches_tracked'):
            continue
        if name.endswith('running_var'):
            v = rs.uniform(0.5, 1.5, size=shape)
        elif name.endswith('running_mean'):
            v = rs.standard_normal(shape) * 0.1
        elif t.dim() == 1 and name.endswith('weight'):            # BatchNorm gamma
            lo, hi = (0.2, 0.4) if '.bn3.' in name else (0.8, 1.2)
            v = rs.uniform(lo, hi, size=shape)
        elif name.endswith('weight'):
            fan_in = int(np.prod(shape[1:]))
            v = rs.standard_normal(shape) * np.sqrt(2.0 / max(fan_in, 1))
            if name.startswith('neck.lateral_convs'):
                v = v * 0.04       # the eval-mode trunk leaves activations of std ~25: back to O(1)
            elif 'rpn_reg' in name:
                v = v * 0.2        # proposal deltas of std ~0.3 (target_stds = 1)
        else:
            v = rs.standard_normal(shape) * 0.05
        t.copy_(torch.from_numpy(np.asarray(v, dtype=F32)))
    return state_dict
