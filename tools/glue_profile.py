#!/usr/bin/env python
"""Which lines of the package issue the small torch ops of one training step?  One eager step of
the bench's DetectorStep under a TorchDispatchMode that records every non-view aten op together
with the innermost Python frame inside balancedgroupsoftmax_amd/ (or bench.py).

    python tools/glue_profile.py [--selectp 1] > gpurun_out/glue.txt
"""
import argparse
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

VIEWS = ('view', 'reshape', '_unsafe_view', 'slice', 'select', 'expand', 'permute', 'transpose', 't.',
         'unsqueeze', 'squeeze', 'detach', 'alias', 'as_strided', 'split', 'unbind', 'narrow',
         'empty', 'size', 'stride', 'is_', '_local_scalar', 'lift_fresh', 'unfold', 'chunk',
         'empty_like', 'new_empty', 'empty_strided', 'sym_', 'stack_trace', 'set_', 'resize_')


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.agg = collections.defaultdict(collections.Counter)

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, '__name__') else str(func)
        full = str(func)
        base = full.replace('aten.', '').split('.')[0]
        if not any(base == v.rstrip('.') or base.startswith(v) for v in VIEWS):
            where = '?'
            for fr in reversed(traceback.extract_stack(limit=40)):
                fn = fr.filename
                if 'balancedgroupsoftmax_amd/' in fn or fn.endswith('bench.py'):
                    where = '%s:%d %s' % (fn.split('balancedgroupsoftmax_amd/')[-1].split('/')[-1]
                                           if 'balancedgroupsoftmax_amd/' in fn else 'bench.py', fr.lineno, fr.name)
                    break
            self.agg[where][base] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--selectp', type=int, default=1)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    step = bench.DetectorStep(dev, 0, 1, 2, a.selectp)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    log = Log()
    with log:
        step()
    torch.cuda.synchronize()
    tot = sum(sum(c.values()) for c in log.agg.values())
    print('non-view aten ops issued by one step: %d' % tot)
    for w, c in sorted(log.agg.items(), key=lambda kv: -sum(kv[1].values())):
        print('%4d  %-46s %s' % (sum(c.values()), w, ', '.join('%s x%d' % kv for kv in c.most_common())[:200]))


if __name__ == '__main__':
    main()
