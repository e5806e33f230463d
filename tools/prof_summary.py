#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid): calls, avg/median/min ns.

    python tools/prof_summary.py gpurun_out/<tag>/prof/bench_kernel_trace.csv > profiles/<name>.md
"""
import collections
import csv
import sys


CATS = [('conv: fused bottleneck tail (3x3 -> 1x1 + residual)', ('conv3x3_c3_fused', 'bottleneck_tail_planes')), ('conv: 3x3 (planes / halo kernels)', ('conv3x3_halo', 'conv3x3_planes', 'conv3x3s2_planes')), ('conv: implicit GEMM', ('conv_igemm', 'conv_bf16s', 'conv1x1_bres', 'conv1x1_bfx_wide', 'conv1x1_planes')),
        ('conv: split-K epilogue', ('conv_splitk',)), ('conv: weight split / wgrad / grouped / pool',
                                                       ('bfx_split', 'conv_wgrad', 'grouped_conv', 'maxpool', 'fold_', 'wgrad_reduce', 'nchw_to', 'stem_')),
        ('torch glue (elementwise / copy / cat / reduce / index)', ('at::native', 'rocclr', 'at::cuda', 'cub::', 'rocprim', 'hipcub')),
        ('optimizer (fused clip + SGD)', ('sgd_',)),
        ('GroupSoftmax + box loss', ('gs_', 'bbox_sl1', 'reduce_partials')),
        ('targets / sampling / RPN loss', ('iou_', 'rpn_loss', 'sample_', 'rcnn_targets', 'random_keys', 'decode_proposals')),
        ('top-k / NMS', ('topk_', 'nms_', 'gather_boxes')), ('RoIAlign', ('roi_align',)), ('mask / resize', ('mask_', 'resize_'))]


def categories(rows, steps):
    """Per-step GPU time by kernel family (the trace holds `steps` eager steps incl. warm-up)."""
    tot = collections.OrderedDict((c, [0, 0]) for c, _ in CATS)
    tot['other'] = [0, 0]
    for r in rows:
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        name = r['Kernel_Name']
        for c, keys in CATS:
            if any(k in name for k in keys):
                break
        else:
            c = 'other'
        tot[c][0] += d
        tot[c][1] += 1
    alln = sum(v[0] for v in tot.values())
    global LAST_FAMILIES
    LAST_FAMILIES = collections.OrderedDict(
        (c, dict(ms=round(ns / steps / 1e6, 4), launches=round(n / steps, 1))) for c, (ns, n) in tot.items())
    LAST_FAMILIES['all kernels'] = dict(ms=round(alln / steps / 1e6, 4),
                                        launches=round(sum(v[1] for v in tot.values()) / steps, 1))
    print('| kernel family | ms / step | launches / step | % |')
    print('|---|---|---|---|')
    for c, (ns, n) in tot.items():
        print('| %s | %.3f | %.1f | %.1f |' % (c, ns / steps / 1e6, n / steps, 100.0 * ns / alln))
    print('| **all kernels** | %.3f | %.1f | 100 |' % (alln / steps / 1e6, sum(v[1] for v in tot.values()) / steps))
    print()


def steady_steps(rows, markers=('stem_conv7x7s2_relu_maxpool', 'maxpool3x3s2')):
    """The kernels of the LAST complete steps of the trace (between occurrences of a kernel that
    runs once per step — the fused stem kernel, or the stem's max-pool of the three-launch chain): warm-up work (BN
    folding, filter splits, allocator fills) is excluded."""
    rows = sorted(rows, key=lambda r: int(r['Start_Timestamp']))
    marks = []
    for marker in markers:
        marks = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
        if len(marks) >= 3:
            break
    if len(marks) < 3:
        return rows, 0
    n = min(4, len(marks) - 1)
    return rows[marks[-1 - n]:marks[-1]], n


def main(path, steps=None):
    rows = list(csv.DictReader(open(path)))
    if steps:
        srows, n = steady_steps(rows)
        if n:
            print('steady state: the last %d complete steps of the trace (per step)\n' % n)
            categories(srows, n)
            rows = srows
        else:
            categories(rows, steps)
    agg = collections.defaultdict(list)
    for r in rows:
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        name = r['Kernel_Name']
        name = name.replace('(anonymous namespace)::', '').replace('void ', '')
        name = name.split('(')[0][:70]
        key = (name, int(r['Grid_Size_X']), int(r['Workgroup_Size_X']), r['VGPR_Count'],
               r['SGPR_Count'], r['LDS_Block_Size'])
        agg[key].append(d)
    total = sum(sum(v) for v in agg.values())
    print('| kernel | grid (threads) | wg | vgpr | sgpr | lds B | calls | avg ns | median ns | min ns | % time |')
    print('|---|---|---|---|---|---|---|---|---|---|---|')
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        v = sorted(v)
        print('| %s | %d | %d | %s | %s | %s | %d | %d | %d | %d | %.1f |' % (
            k[0], k[1], k[2], k[3], k[4], k[5], len(v), sum(v) / len(v), v[len(v) // 2], v[0],
            100.0 * sum(v) / total))


LAST_FAMILIES = None


if __name__ == '__main__':
    # prof_summary.py trace.csv [steps] [families.json source-label]: the third argument also writes
    # the per-family table as JSON (bench.py reads profiles/step_families.json into roofline_step)
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None)
    if len(sys.argv) > 3 and LAST_FAMILIES is not None:
        import json
        with open(sys.argv[3], 'w') as f:
            json.dump(dict(families_ms=LAST_FAMILIES,
                           source=sys.argv[4] if len(sys.argv) > 4 else sys.argv[1]), f, indent=1)
