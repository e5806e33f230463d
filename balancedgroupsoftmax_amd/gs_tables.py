"""Group tables for Balanced Group Softmax: construction rule + on-disk formats.

The reference builds three files offline from the LVIS train annotation
(``tools/lvis_analyse.py:11-58`` ``get_cate_gs`` and ``:60-98`` ``get_split``):

* ``label2binlabel.pt``   int64 ``[B, C]``  class id -> label inside bin ``b`` (0 = "others")
* ``pred_slice_with0.pt`` int64 ``[B, 2]``  (start, length) of bin ``b`` in the ``C + B``-wide logits
* ``valsplit.pkl``        dict of int arrays: class ids of every foreground bin, in bin-label order

Neither the files nor the annotation json ship with the reference
(``README.md:83-87``), so bin sizes are *runtime data*.  This module restates the
rule for arbitrary instance counts / thresholds and reads/writes the same formats,
so real ``./data/lvis/*.pt`` files drop in unchanged.
"""
import os
import pickle

import numpy as np
import torch

# keys hard-coded by the reference loader (gs_bbox_head_with0.py:45-49)
FG_SPLIT_KEYS_5 = ['(0, 10)', '[10, 100)', '[100, 1000)', '[1000, ~)']


def split_keys(thresholds):
    """Key names in the style of ``tools/lvis_analyse.py:87-91`` for any threshold list."""
    thresholds = list(thresholds)
    if thresholds == [10, 100, 1000]:
        return list(FG_SPLIT_KEYS_5)
    keys = ['(0, %d)' % thresholds[0]]
    for lo, hi in zip(thresholds[:-1], thresholds[1:]):
        keys.append('[%d, %d)' % (lo, hi))
    keys.append('[%d, ~)' % thresholds[-1])
    return keys


def build_group_tables(instance_counts, thresholds=(10, 100, 1000)):
    """Restates ``get_cate_gs`` + ``get_split`` (tools/lvis_analyse.py:11-98).

    Args:
        instance_counts: int array ``[C]``; entry 0 (background) is ignored.
            ``instance_counts[c]`` = train instance count of category id ``c``.
        thresholds: ascending bin upper bounds; ``len(thresholds) + 2`` bins result
            (bin 0 = {bg, fg}; bin i>=1 = classes with count in
            ``[thresholds[i-2], thresholds[i-1])``).

    Returns:
        label2binlabel int64 ``[B, C]``, pred_slice int64 ``[B, 2]``,
        fg_split ``dict(key -> int64 array)`` incl. 'normal', 'background', 'all'.
    """
    counts = np.asarray(instance_counts)
    C = counts.shape[0]
    thresholds = list(thresholds)
    B = len(thresholds) + 2
    binlabel_count = [1] * B
    l2b = np.zeros((B, C), dtype=np.int64)
    l2b[0, 1:] = 1
    binlabel_count[0] += 1
    members = [[] for _ in range(B - 1)]
    for cid in range(1, C):  # ascending category id == LVIS cats dict order
        n = int(counts[cid])
        b = 1 + int(np.searchsorted(thresholds, n, side='right'))
        l2b[b, cid] = binlabel_count[b]
        binlabel_count[b] += 1
        members[b - 1].append(cid)
    pred_slice = np.zeros((B, 2), dtype=np.int64)
    start = 0
    for i, n in enumerate(binlabel_count):
        pred_slice[i] = (start, n)
        start += n
    splits = {}
    for key, m in zip(split_keys(thresholds), members):
        splits[key] = np.array(m, dtype=np.int64)
    splits['normal'] = np.arange(1, C)
    splits['background'] = np.zeros((1,), dtype=np.int64)
    splits['all'] = np.arange(C)
    return l2b, pred_slice, splits


def synthetic_instance_counts(num_classes=1231, seed=0):
    """Long-tailed synthetic counts: ``floor(10 ** U(0, 4.3))`` (cfg 1).

    With ``num_classes=1231, seed=0`` the 5-bin widths are [2, 285, 312, 266, 371].
    """
    rs = np.random.RandomState(seed)
    ins = np.floor(10.0 ** rs.uniform(0.0, 4.3, size=num_classes - 1)).astype(np.int64)
    return np.concatenate([np.ones(1, dtype=np.int64), ins])


def synthetic_group_tables(num_classes=1231, seed=0, thresholds=(10, 100, 1000)):
    return build_group_tables(synthetic_instance_counts(num_classes, seed), thresholds)


def bin_class_weights(instance_counts, label2binlabel):
    """Restates ``get_bin_weight`` (tools/lvis_analyse.py:449-484): per-bin class
    weights for ``GSBBoxHeadWith0Reweight`` (inverse frequency, mean-normalised,
    clipped to [0.1, 5], with weight 1 prepended for the bin's "others" slot)."""
    counts = np.asarray(instance_counts, dtype=np.float64).copy()
    counts[0] = 1
    weight = 1.0 / counts
    out = []
    for i in range(1, label2binlabel.shape[0]):
        idx = np.where(label2binlabel[i] > 0)
        binw = weight[idx]
        binw = binw / binw.mean()
        binw = np.where(binw > 5, 5, binw)
        binw = np.where(binw < 0.1, 0.1, binw)
        out.append(np.hstack((np.ones(1,), binw)))
    return out


def save_group_tables(out_dir, label2binlabel, pred_slice, fg_split, bin_cls_weight=None):
    """Writes the reference's three (optionally four) files into ``out_dir``."""
    os.makedirs(out_dir, exist_ok=True)
    paths = dict(label2binlabel=os.path.join(out_dir, 'label2binlabel.pt'),
                 pred_slice=os.path.join(out_dir, 'pred_slice_with0.pt'),
                 fg_split=os.path.join(out_dir, 'valsplit.pkl'))
    torch.save(torch.from_numpy(np.ascontiguousarray(label2binlabel)), paths['label2binlabel'])
    torch.save(torch.from_numpy(np.ascontiguousarray(pred_slice)), paths['pred_slice'])
    with open(paths['fg_split'], 'wb') as f:
        pickle.dump(fg_split, f)
    if bin_cls_weight is not None:
        paths['bin_cls_weight'] = os.path.join(out_dir, 'bins_cls_weight.pkl')
        with open(paths['bin_cls_weight'], 'wb') as f:
            pickle.dump(bin_cls_weight, f)
    return paths


def load_group_tables(label2binlabel, pred_slice, fg_split):
    """Loads the reference formats (gs_bbox_head_with0.py:37-49). Returns CPU tensors
    plus the list of foreground splits ordered by bin (bin 1 first)."""
    l2b = torch.load(label2binlabel, map_location='cpu').long().contiguous()
    ps = torch.load(pred_slice, map_location='cpu').long().contiguous()
    with open(fg_split, 'rb') as f:
        split = pickle.load(f)
    B = l2b.shape[0]
    skip = ('normal', 'background', 'all')
    keys = [k for k in split.keys() if k not in skip]
    if B == 5 and all(k in split for k in FG_SPLIT_KEYS_5):
        keys = list(FG_SPLIT_KEYS_5)
    if len(keys) != B - 1:
        raise ValueError('fg_split has %d foreground groups, label2binlabel has %d bins'
                         % (len(keys), B))
    fg_splits = [torch.as_tensor(np.asarray(split[k]), dtype=torch.long) for k in keys]
    return l2b, ps, fg_splits


def class_to_column(label2binlabel, pred_slice):
    """For every class id c>=1: the logit column that holds its in-bin score,
    i.e. ``start_b + label2binlabel[b, c]`` for the unique fg bin b with a non-zero
    entry.  Column for c=0 is the bg column ``start_0 + 0``.  int32 ``[C]``.

    This is the inverse form of the ``fg_splits`` scatter used by ``_merge_score``
    (gs_bbox_head_with0.py:258-259): ``fg_split[b-1][k-1] = c  <=>  L[b, c] = k``.
    """
    l2b = torch.as_tensor(label2binlabel).long()
    ps = torch.as_tensor(pred_slice).long()
    B, C = l2b.shape
    col = torch.zeros(C, dtype=torch.int32)
    col[0] = int(ps[0, 0])
    fg = l2b[1:]  # [B-1, C]
    owner = (fg > 0).long()
    if not bool((owner[:, 1:].sum(0) == 1).all()):
        raise ValueError('every foreground class must belong to exactly one foreground bin')
    b = owner.argmax(0) + 1  # [C]
    cols = ps[b, 0] + l2b[b, torch.arange(C)]
    col[1:] = cols[1:].int()
    return col
