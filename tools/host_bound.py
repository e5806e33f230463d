"""Is the cfg[1] step bound by the host's launch path?  Per policy (sequential eager with forks, pipelined depth 5): wall
time until the LAST launch of n steps is enqueued (no synchronisation inside), wall time until the GPU has finished, and
the CPU time the launching thread spent (time.thread_time).  enqueue ~ total and cpu ~ enqueue  => the host is the
bottleneck; enqueue << total => the GPU is.
python tools/host_bound.py [steps=40] [--cascade | --htc | --mask] [--bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
argv = sys.argv[1:]
sys.argv = sys.argv[:1]
import torch
import bench
from balancedgroupsoftmax_amd import functional as BF

n = int(argv[0]) if argv else 40
flags = argv[1:]                                   # e.g. --cascade --bf16
dev = torch.device('cuda', 0)
if '--bf16' in flags:
    BF.set_conv_math('bf16')
x101 = '--cascade' in flags or '--htc' in flags
step = bench.DetectorStep(dev, 0, 1, 2, 3 if x101 else 1, mask='--mask' in flags, cascade='--cascade' in flags,
                          htc='--htc' in flags, conv_math='bf16' if '--bf16' in flags else 'bf16x6')


def measure(fn, name):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    out = []
    for rep in range(3):
        t0, c0 = time.perf_counter(), time.thread_time()
        for _ in range(n):
            fn()
        t1, c1 = time.perf_counter(), time.thread_time()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out.append(((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, (c1 - c0) / n * 1e3))
    e, t, c = sorted(out, key=lambda v: v[1])[1]
    print('%-18s enqueue %.3f ms / step | until the GPU is done %.3f ms / step | launching thread CPU %.3f ms / step' % (name, e, t, c), flush=True)


measure(step, 'eager (forks)')
os.environ['BGS_LEVEL_FORK'] = '0'
measure(step, 'eager (no forks)')
os.environ.pop('BGS_LEVEL_FORK')
for d in (3, 5):
    fn = step.pipelined(depth=d)
    measure(fn, 'pipelined depth %d' % d)
    fn.drain()
    torch.cuda.synchronize()
