#!/usr/bin/env python
"""Round 6 A/B of the bandwidth-bound GroupSoftmax kernels: row-per-WAVE (private LDS row, no workgroup barrier)
against row-per-workgroup, for the loss (bgs_gs_loss_fwd_bwd, loss_out = NULL: the streaming kernel alone) and the
score merge (bgs_gs_merge_score), next to a plain device copy of the same bytes on the same box.
HIP events around back-to-back launches, batches repeated until two agree within 1 %.  python tools/gs_stream_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balancedgroupsoftmax_amd import capi, functional as BF, gs_tables  # noqa: E402


def timed(run, iters, settle=8):
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    prev = None
    for _ in range(settle):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        if prev is not None and abs(us - prev) <= 0.01 * prev:
            break
        prev = us
    return us


def main():
    dev = 'cuda:0'
    lib = capi.load()
    C = 1231
    l2b, ps, _ = gs_tables.build_group_tables(gs_tables.synthetic_instance_counts(C, seed=0))
    ps_keep, ps_ptr = capi.host_i64(ps)
    c2c = gs_tables.class_to_column(l2b, ps).to(dev)
    sizes = [int(x) for x in os.environ.get('SIZES', '4096,8192,16384,65536,262144').split(',')]
    for N in sizes:
        z = torch.randn(N, 1236, device=dev)
        labels = torch.randint(0, C, (N,), device=dev)
        labels[N // 4:] = 0
        bl, w, avg = BF.gs_prepare(labels, torch.from_numpy(l2b).to(dev), 8.0, seed=3)
        dz = torch.empty_like(z)
        sc = torch.empty(N, C, device=dev)
        ws = torch.empty(lib.bgs_gs_loss_workspace_bytes(N, 5), dtype=torch.uint8, device=dev)
        st = capi.current_stream(z.device)
        iters = 50 if N <= 65536 else 15
        us_copy = timed(lambda: dz.copy_(z), iters)
        row = ['copy %7.1f us %5.2f TB/s' % (us_copy, 2 * 4944.0 * N / us_copy / 1e6)]
        lib.bgs_gs_loss_wavepriv_min_rows(0)
        for mode in (3, 4, 6, 7, 3, 6):
            lib.bgs_gs_loss_tuning(mode)

            def run():
                rc = lib.bgs_gs_loss_fwd_bwd(capi.ptr(z), capi.ptr(bl), ps_ptr, capi.ptr(w), capi.ptr(avg), N, 5, 1236,
                                             None, capi.ptr(dz), capi.ptr(ws), st)
                assert rc == 0
            us = timed(run, iters)
            row.append('loss m%d %7.1f us %5.2f TB/s' % (mode, us, 9916.0 * N / us / 1e6))
        lib.bgs_gs_loss_tuning(5)
        lib.bgs_gs_loss_wavepriv_min_rows(-1)
        for mode in (0, 2, 3, 0, 2):
            lib.bgs_gs_merge_tuning(mode, 0)

            def run():
                rc = lib.bgs_gs_merge_score(capi.ptr(z), ps_ptr, capi.ptr(c2c), N, C, 5, 1236, capi.ptr(sc), st)
                assert rc == 0
            us = timed(run, iters)
            row.append('merge m%d %7.1f us %5.2f TB/s' % (mode, us, 9868.0 * N / us / 1e6))
        lib.bgs_gs_merge_tuning(1, -1)
        print('N %7d | %s' % (N, ' | '.join(row)), flush=True)
        del z, dz, sc


if __name__ == '__main__':
    main()
