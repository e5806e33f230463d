"""A/B of the fused GroupSoftmax head kernel's counting scheme (bgs_gs_head_variant: 0 = flag words, 1 = bit
planes, 2 / 3 = bit planes with 2 / 4 rows per workgroup, 4 / 5 = those with direct gradient stores): the main kernel under back-to-back HIP events and the whole head step under hipGraph replay, several
interleaved rounds.  GPU only.  `python tools/gs_head_ab.py [rounds] [variants, comma separated]`."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

import bench  # noqa: E402
from balancedgroupsoftmax_amd import capi  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    VARIANTS = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0, 1, 2, 3, 4, 5]
    dev = torch.device('cuda', 0)
    lib = capi.load()
    out = {}
    for n in (1024, 512, 2048):
        inp = bench.make_inputs(n, 0, dev)
        for rnd in range(rounds):
            for variant in VARIANTS:
                lib.bgs_gs_head_variant(variant)
                r = bench.kernel_roofline(inp, n, iters=300, kernel='fused')
                rec = out.setdefault('n%d_v%d' % (n, variant), dict(kernel_us=[], step_us=[], step_any_us=[]))
                rec['kernel_us'].append(r['us_per_launch'])
                if n == 1024:
                    g = bench.gs_head_metric(inp, n, steps=300, warmup=20)
                    rec['step_us'].append(g['us_per_step'])
                    rec['step_any_us'].append(g['us_per_step_any_upstream'])
    lib.bgs_gs_head_variant(-1)
    for k in sorted(out):
        print(k, json.dumps(out[k]))


if __name__ == '__main__':
    main()
