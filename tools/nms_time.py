#!/usr/bin/env python
"""Times bgs_nms_batched on the training shape (10 problems x 2000 boxes) and the test-time
shape (1230 classes x 1000 boxes).  python tools/nms_time.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402


def boxes(P, n, seed):
    g = torch.Generator().manual_seed(seed)
    ctr = torch.rand(P, n, 2, generator=g) * torch.tensor([1333., 800.])
    wh = torch.exp(torch.rand(P, n, 2, generator=g) * 3.0 + 2.5)
    sc = torch.sort(torch.rand(P, n, generator=g), dim=1, descending=True).values
    return torch.cat([ctr - wh / 2, ctr + wh / 2, sc[..., None]], -1).cuda().contiguous()


for P, n, thr in ((10, 2000, 0.7), (1230, 1000, 0.5)):
    b = boxes(P, n, P)
    cnt = torch.full((P,), n, dtype=torch.int32, device='cuda')
    for _ in range(3):
        keep, kc = BF.nms_batched(b, cnt, thr)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        keep, kc = BF.nms_batched(b, cnt, thr)
    e1.record()
    torch.cuda.synchronize()
    print('P=%d n=%d: %.3f ms per call, kept/problem %.1f' % (P, n, e0.elapsed_time(e1) / 20, float(kc.float().mean())))
