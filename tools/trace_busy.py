#!/usr/bin/env python
"""GPU occupancy of a rocprofv3 --kernel-trace CSV over the last complete steps (between `sgd_apply_kernel` launches):
span per step, time with at least one kernel running (union of the launch intervals), the sum of the launch durations
(average concurrency = sum / union), the idle gaps, and which kernels run ALONE for how long.
    python tools/trace_busy.py trace.csv [steps=8]"""
import collections
import csv
import sys


def short(n):
    return n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]


def main(path, nsteps=8):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))
    ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Stream_Id', r.get('Queue_Id', '?'))) for r in rows]
    marks = [i for i, e in enumerate(ev) if 'sgd_apply_kernel' in e[2]]
    if len(marks) < nsteps + 1:
        nsteps = len(marks) - 1
    lo, hi = ev[marks[-1 - nsteps]][1], ev[marks[-1]][1]
    sel = [e for e in ev if e[0] >= lo and e[1] <= hi]
    span = hi - lo
    # sweep
    pts = []
    for s, e, n, q in sel:
        pts.append((s, 1, n)); pts.append((e, -1, n))
    pts.sort(key=lambda p: (p[0], p[1]))
    live = collections.Counter()
    nlive, last, union = 0, lo, 0
    conc_time = collections.Counter()
    alone = collections.Counter()
    gaps = []
    lastname = '-'
    for t, d, n in pts:
        dt = t - last
        if dt > 0:
            conc_time[min(nlive, 6)] += dt
            if nlive == 0:
                gaps.append((dt, lastname, n))
            else:
                union += dt
            if nlive == 1:
                alone[next(k for k, v in live.items() if v > 0)] += dt
        last = t
        if d < 0:
            lastname = n
        nlive += d
        live[n] += d
    tot = sum(e - s for s, e, _, _ in sel)
    print('steps %d | span %.3f ms / step | busy (>= 1 kernel) %.3f ms / step (%.1f %%) | sum of launch durations %.3f ms / step | '
          'average concurrency while busy %.2f | launches / step %.1f | streams %d'
          % (nsteps, span / nsteps / 1e6, union / nsteps / 1e6, 100.0 * union / span, tot / nsteps / 1e6, tot / max(union, 1),
             len(sel) / nsteps, len({q for _, _, _, q in sel})))
    print('time with k kernels in flight (ms / step): ' + '  '.join('%s%d: %.3f' % ('>=' if k == 6 else '', k, v / nsteps / 1e6)
                                                                    for k, v in sorted(conc_time.items())))
    big = sorted(gaps, reverse=True)
    print('idle gaps: %d / step, %.3f ms / step; the largest (us): %s' % (len(gaps) / nsteps, sum(g[0] for g in gaps) / nsteps / 1e6,
                                                                        ' '.join('%.1f' % (g[0] / 1e3) for g in big[:10])))
    where = collections.Counter()
    for g, a, b in gaps:
        where[(a, b)] += g
    print('idle time by (kernel that ended -> kernel that started), ms / step:')
    for (a, b), v in where.most_common(8):
        print('  %.3f  %s -> %s' % (v / nsteps / 1e6, a[:40], b[:40]))
    print('kernels running ALONE (ms / step):')
    for n, v in alone.most_common(14):
        print('  %-62s %.3f' % (n, v / nsteps / 1e6))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8)
