#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid): calls, avg/median/min ns.

    python tools/prof_summary.py gpurun_out/<tag>/prof/bench_kernel_trace.csv > profiles/<name>.md
"""
import collections
import csv
import sys


def main(path):
    rows = list(csv.DictReader(open(path)))
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


if __name__ == '__main__':
    main(sys.argv[1])
