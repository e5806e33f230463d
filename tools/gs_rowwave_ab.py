#!/usr/bin/env python
"""A/B of the row-per-workgroup GroupSoftmax loss kernel's next-row prefetch at the bandwidth-bound sizes
(HIP events around the kernel alone: bgs_gs_loss_fwd_bwd with loss_out = NULL).  python tools/gs_rowwave_ab.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balancedgroupsoftmax_amd import capi, functional as BF, gs_tables  # noqa: E402


def main():
    dev = 'cuda:0'
    lib = capi.load()
    C = 1231
    l2b, ps, _ = gs_tables.build_group_tables(gs_tables.synthetic_instance_counts(C, seed=0))
    ps_keep, ps_ptr = capi.host_i64(ps)
    for N in (8192, 16384, 65536, 262144):
        z = torch.randn(N, 1236, device=dev)
        labels = torch.randint(0, C, (N,), device=dev)
        labels[N // 4:] = 0
        bl, w, avg = BF.gs_prepare(labels, torch.from_numpy(l2b).to(dev), 8.0, seed=3)
        dz = torch.empty_like(z)
        ws = torch.empty(lib.bgs_gs_loss_workspace_bytes(N, 5), dtype=torch.uint8, device=dev)
        st = capi.current_stream(z.device)
        row = []
        for pf in (0, 1, 2, 3, 4, 1, 4):
            lib.bgs_gs_loss_tuning(pf)

            def run():
                rc = lib.bgs_gs_loss_fwd_bwd(capi.ptr(z), capi.ptr(bl), ps_ptr, capi.ptr(w), capi.ptr(avg), N, 5, 1236,
                                             None, capi.ptr(dz), capi.ptr(ws), st)
                assert rc == 0
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            row.append('pf%d %8.1f us %5.2f TB/s' % (pf, us, 9916.0 * N / us / 1e6))
        print('N %7d | %s' % (N, ' | '.join(row)), flush=True)
    lib.bgs_gs_loss_tuning(3)


if __name__ == '__main__':
    main()
