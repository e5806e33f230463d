"""TEST INFRASTRUCTURE ONLY (CPU oracle) — not part of the product path.

numpy restatement of the reference's Balanced-Group-Softmax head arithmetic.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product package never does (it fails loudly when the HIP
library is missing instead of falling back to CPU).

Pinned against the *executed reference class* (``GSBBoxHeadWith0`` imported through
``oracle/ref_import.py``) by ``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``
and ``tests/test_oracle_golden.py``.  The reference itself holds no golden vectors
for this path, so those executed-reference fixtures are the pin.

Every function cites the reference lines it restates (paths relative to the
reference root).
"""
import numpy as np


# ----------------------------------------------------------------------------
# label remap + "others" sampling
# ----------------------------------------------------------------------------
def remap_labels(labels, label2binlabel):
    """``new_bin_label = label2binlabel[i][labels]`` for every bin i.

    mmdet/models/bbox_heads/gs_bbox_head_with0.py:96-99.  int64 ``[B, N]``.
    """
    labels = np.asarray(labels, dtype=np.int64)
    l2b = np.asarray(label2binlabel, dtype=np.int64)
    return l2b[:, labels]


def sample_others(bin_label, others_sample_ratio, cls_weight=None):
    """mmdet/models/bbox_heads/gs_bbox_head_with0.py:63-89 (and the ``Reweight``
    variant, gs_bbox_head_with0_reweight.py:57-87, when ``cls_weight`` is given).

    Uses numpy's *global* RNG exactly like the reference (``np.random.choice`` on
    the ascending index list of non-fg rows, ``replace=False``), so seeding
    ``np.random.seed`` reproduces the reference's draw bit-for-bit.
    """
    bin_label = np.asarray(bin_label, dtype=np.int64)
    fg = (bin_label > 0).astype(np.int64)
    fg_num = int(fg.sum())
    if fg_num == 0:
        return np.zeros_like(bin_label, dtype=np.float64)
    bg_idx = np.nonzero(1 - fg)[0]
    bg_num = bg_idx.shape[0]
    bg_sample_num = int(fg_num * others_sample_ratio)
    if bg_sample_num >= bg_num:
        weight = np.ones_like(bin_label)
    else:
        sample_idx = np.random.choice(bg_idx, (bg_sample_num,), replace=False)
        weight = fg.copy()
        weight[sample_idx] = 1
    weight = weight.astype(np.float64)
    if cls_weight is not None:
        weight = weight * np.asarray(cls_weight, dtype=np.float64)[bin_label]
    return weight


def remap_and_sample(labels, label2binlabel, others_sample_ratio, cls_weights=None):
    """``_remap_labels`` (gs_bbox_head_with0.py:91-112): bin labels, weights, avg factors.

    Returns ``bin_labels`` int64 [B,N], ``weights`` float32 [B,N], ``avg`` float32 [B]
    (``avg_factor = max(sum(weight).float(), 1.)``).
    """
    bl = remap_labels(labels, label2binlabel)
    B, N = bl.shape
    w = np.zeros((B, N), dtype=np.float64)
    for i in range(B):
        if i < 1:
            w[i] = 1.0
        else:
            cw = None if cls_weights is None else cls_weights[i - 1]
            w[i] = sample_others(bl[i], others_sample_ratio, cw)
    # torch.sum(weight).float(): the sum is taken in the weight dtype
    # (int64, or float64 for the reweight variant) and then cast to fp32.
    avg = np.maximum(w.sum(axis=1).astype(np.float32), np.float32(1.0))
    return bl, w.astype(np.float32), avg.astype(np.float32)


# ----------------------------------------------------------------------------
# per-bin cross entropy (loss + gradient)
# ----------------------------------------------------------------------------
def group_softmax_loss(logits, bin_labels, weights, avg, pred_slice, dtype=np.float64,
                       grad_scale=None):
    """Per-bin weighted CE and its gradient.

    loss_i = sum_r w_i[r] * (logsumexp(z[r, s_i:s_i+n_i]) - z[r, s_i + b_i[r]]) / A_i
      — gs_bbox_head_with0.py:160-171 -> losses/cross_entropy_loss.py:9-19
      (``F.cross_entropy(reduction='none')``) -> losses/utils.py:26-52
      (``loss * weight`` then ``loss.sum() / avg_factor``).
    dz[r, s_i + j] = g_i * (w_i[r] / A_i) * (softmax_j - [j == b_i[r]])  (autograd of the above).

    ``dtype=np.float64`` gives the "true" value; ``np.float32`` mimics the fp32 path.
    Returns ``losses [B]``, ``dlogits [N, W]``.
    """
    z = np.asarray(logits).astype(dtype)
    N, W = z.shape
    B = len(pred_slice)
    g = np.ones(B, dtype=dtype) if grad_scale is None else np.asarray(grad_scale, dtype=dtype)
    losses = np.zeros(B, dtype=dtype)
    dz = np.zeros((N, W), dtype=dtype)
    rows = np.arange(N)
    for i in range(B):
        s, n = int(pred_slice[i][0]), int(pred_slice[i][1])
        zi = z[:, s:s + n]
        m = zi.max(axis=1, keepdims=True)
        e = np.exp(zi - m)
        se = e.sum(axis=1, keepdims=True)
        lse = (np.log(se) + m)[:, 0]
        bl = np.asarray(bin_labels[i], dtype=np.int64)
        wi = np.asarray(weights[i]).astype(dtype)
        ai = dtype(avg[i])
        per_row = lse - zi[rows, bl]
        losses[i] = (per_row * wi).sum() / ai
        p = e / se
        p[rows, bl] -= 1.0
        dz[:, s:s + n] = p * (g[i] * wi / ai)[:, None]
    return losses, dz


# ----------------------------------------------------------------------------
# box regression loss
# ----------------------------------------------------------------------------
def smooth_l1_bbox_loss(bbox_pred, labels, bbox_targets, bbox_weights, num_classes,
                        beta=1.0, reg_class_agnostic=False, dtype=np.float64):
    """``loss_bbox`` part of ``loss`` (gs_bbox_head_with0.py:173-185) with
    ``smooth_l1_loss`` (losses/smooth_l1_loss.py:9-15), ``avg_factor = N``.

    Returns ``loss`` (scalar) and dense ``dbbox_pred`` (same shape as ``bbox_pred``).
    """
    bp = np.asarray(bbox_pred).astype(dtype)
    labels = np.asarray(labels, dtype=np.int64)
    t = np.asarray(bbox_targets).astype(dtype)
    bw = np.asarray(bbox_weights).astype(dtype)
    N = bp.shape[0]
    pos = np.nonzero(labels > 0)[0]
    grad = np.zeros_like(bp)
    if reg_class_agnostic:
        view = bp.reshape(N, 1, 4)
        cls = np.zeros_like(labels)
    else:
        view = bp.reshape(N, num_classes, 4)
        cls = labels
    gview = grad.reshape(view.shape)
    pred = view[pos, cls[pos]]
    d = pred - t[pos]
    ad = np.abs(d)
    loss_el = np.where(ad < beta, 0.5 * ad * ad / beta, ad - 0.5 * beta)
    loss = (loss_el * bw[pos]).sum() / dtype(N)
    gel = np.where(ad < beta, d / beta, np.sign(d)) * bw[pos] / dtype(N)
    gview[pos, cls[pos]] = gel
    return loss, grad


# ----------------------------------------------------------------------------
# inference score merge
# ----------------------------------------------------------------------------
def merge_score(logits, pred_slice, fg_splits, num_classes, dtype=np.float64):
    """``_merge_score`` (gs_bbox_head_with0.py:239-273): softmax inside each bin;
    ``merge[:, 0] = p_0[:, 0]``; for class c = fg_splits[i][k-1]:
    ``merge[:, c] = p_0[:, 1] * p_{i+1}[:, k]``.  Rows do not sum to 1.
    """
    z = np.asarray(logits).astype(dtype)
    N = z.shape[0]
    scores = []
    for s, n in pred_slice:
        zi = z[:, int(s):int(s) + int(n)]
        m = zi.max(axis=1, keepdims=True)
        e = np.exp(zi - m)
        scores.append(e / e.sum(axis=1, keepdims=True))
    fg_merge = np.zeros((N, num_classes), dtype=dtype)
    for i, split in enumerate(fg_splits):
        fg_merge[:, np.asarray(split, dtype=np.int64)] = scores[i + 1][:, 1:]
    fg_merge = scores[0][:, 1:2] * fg_merge
    merge = np.zeros((N, num_classes), dtype=dtype)
    merge[:, 0] = scores[0][:, 0]
    merge[:, 1:] = fg_merge[:, 1:]
    return merge


# ----------------------------------------------------------------------------
# deterministic synthetic inputs shared by golden generation, tests and bench
# ----------------------------------------------------------------------------
def make_roi_batch(n, width, num_classes, seed, fg_fraction=0.25, with_bbox=False,
                   logit_scale=1.0):
    """cfg-1-style RoI batch: ``randn(n, width)`` logits, the first
    ``fg_fraction`` of the rows carry uniform foreground labels (positives first, as
    ``bbox_target_single`` orders them, mmdet/core/bbox/bbox_target.py:35-61).

    Uses the legacy ``RandomState`` stream so the same seed yields the same bytes on
    every machine / numpy version.
    """
    rs = np.random.RandomState(seed)
    logits = (rs.standard_normal((n, width)) * logit_scale).astype(np.float32)
    labels = np.zeros(n, dtype=np.int64)
    nfg = int(round(n * fg_fraction))
    if nfg > 0:
        labels[:nfg] = rs.randint(1, num_classes, size=nfg)
    out = dict(logits=logits, labels=labels)
    if with_bbox:
        out['bbox_pred'] = rs.standard_normal((n, 4 * num_classes)).astype(np.float32)
        out['bbox_targets'] = rs.standard_normal((n, 4)).astype(np.float32)
        bw = np.zeros((n, 4), dtype=np.float32)
        bw[labels > 0] = 1.0
        out['bbox_weights'] = bw
    return out
