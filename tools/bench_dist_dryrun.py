#!/usr/bin/env python
"""CPU dry run of bench.py's N > 1 orchestration (bench_dist.run) over gloo: eight ranks, a stand-in step with the
interface of bench.DetectorStep (a tiny CPU model, the product's own ``train.allreduce_grads`` as its gradient exchange,
a fake trunk pipeline), every diagnostic forced to fail or hang in turn through the BGS_BENCH_FAIL / BGS_BENCH_HANG test
hooks.  What it shows: every scenario ends with ONE emitted line on rank 0 that carries the headline measurement, the
failed diagnostic as ``{"error": ...}``, the others measured; a hung rank costs the diagnostics' time budget, not the line.

    python tools/bench_dist_dryrun.py [--world 8] > profiles/r10_bench_dist_dryrun.txt

tests/test_bench_dist_cpu.py runs the same scenarios and asserts on them.
"""
import argparse
import json
import os
import sys
import tempfile
import time
import types

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SCENARIOS = [
    # name, environment, what must hold
    ('clean', {}, 'every diagnostic measured'),
    ('fail_calibration_on_rank5', {'BGS_BENCH_FAIL': 'calibration@5'}, 'calibration skipped on every rank, headline measured'),
    ('fail_exchange_check_on_rank3', {'BGS_BENCH_FAIL': 'grad_exchange_check@3'}, 'check reports the error, allreduce_us measured'),
    ('fail_allreduce_us_everywhere', {'BGS_BENCH_FAIL': 'allreduce_us'}, 'allreduce_us is an error, the check ran'),
    ('fail_n1_reference', {'BGS_BENCH_FAIL': 'n1_reference@0'}, 'n1_same_invocation is an error'),
    ('fail_all_diagnostics', {'BGS_BENCH_FAIL': 'grad_exchange_check@1,allreduce_us@2,n1_reference@0'}, 'three errors, one line'),
    ('hang_exchange_check_on_rank2', {'BGS_BENCH_HANG': 'grad_exchange_check@2', 'BGS_BENCH_DIAG_SECONDS': '4'},
     'the watchdog prints the line with the headline after 4 s'),
    ('hang_before_timed_region', {'BGS_BENCH_HANG': 'calibration@6', 'BGS_BENCH_WALL_SECONDS': '4'},
     'no headline: an error line, every rank leaves'),
]


class StandInStep(object):
    """bench.DetectorStep's interface on a 3-layer CPU model: compute() = forward + backward, apply() = the product's
    flat gradient all-reduce (train.allreduce_grads) + SGD; `pipelined(depth)` = the same step behind a fake queue."""

    def __init__(self, rank, world):
        from balancedgroupsoftmax_amd import train
        self.train = train
        self.world = world
        torch.manual_seed(0)
        self.model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8))
        self.params = list(self.model.parameters())
        opt = torch.optim.SGD(self.params, lr=0.01)
        self.step_fn = types.SimpleNamespace(overlap=None, optimizer=opt)
        self.x = torch.randn(4, 16, generator=torch.Generator().manual_seed(1000 + rank))
        self.last = {}

    def compute(self, feats=None):
        self.step_fn.optimizer.zero_grad(set_to_none=True)
        loss = self.model(self.x).pow(2).mean()
        loss.backward()
        self.last = {'loss': loss.detach()}

    def apply(self):
        self.train.allreduce_grads(self.params, self.world)
        self.step_fn.optimizer.step()

    def __call__(self):
        self.compute()
        self.apply()

    def can_pipeline(self):
        return True

    def pipelined(self, depth=None):
        def step():
            self.compute()
            self.apply()
        step.drain = lambda: None
        step.depth = depth or 4
        return step


def _worker(rank, world, port, out_path):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    import datetime
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    import bench_dist
    step = StandInStep(rank, world)
    args = types.SimpleNamespace(steps=5, warmup=2, imgs=2)

    def emit(f):
        f = {k: v for k, v in f.items() if k != 'step_fn'}
        with open(out_path, 'a') as fh:
            fh.write(json.dumps(f) + '\n')

    bench_dist.run(step, args, rank, world, lambda: StandInStep(0, 1), emit)
    dist.destroy_process_group()


def run_scenario(env, world=8, port=None):
    """-> (list of emitted lines (dicts), wall seconds, exit codes ok)."""
    port = port or (36000 + (os.getpid() * 7 + int(time.time())) % 2000)
    fd, out_path = tempfile.mkstemp(prefix='bench_dist_dry_', suffix='.jsonl')
    os.close(fd)
    saved = {k: os.environ.get(k) for k in ('BGS_BENCH_FAIL', 'BGS_BENCH_HANG', 'BGS_BENCH_DIAG_SECONDS',
                                            'BGS_BENCH_WALL_SECONDS', 'BGS_BENCH_DIST_CALIB')}
    for k in saved:
        os.environ.pop(k, None)
    os.environ.update(env)
    t0 = time.time()
    ok = True
    try:
        mp.spawn(_worker, args=(world, port, out_path), nprocs=world, join=True)
    except Exception as e:      # a rank that left with a non-zero code
        ok = False
        sys.stderr.write('scenario raised: %r\n' % (e,))
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    wall = time.time() - t0
    with open(out_path) as fh:
        lines = [json.loads(ln) for ln in fh if ln.strip()]
    os.unlink(out_path)
    return lines, wall, ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--world', type=int, default=8)
    a = ap.parse_args()
    print('bench_dist.run dry run: gloo, world size %d, stand-in step (tools/bench_dist_dryrun.py)' % a.world)
    for i, (name, env, expect) in enumerate(SCENARIOS):
        lines, wall, ok = run_scenario(env, a.world, port=36100 + 17 * i + os.getpid() % 1000)
        print('\n== %s  env=%s\n   expected: %s\n   wall %.1f s, all ranks left with code 0: %s, lines emitted on rank 0: %d'
              % (name, env, expect, wall, ok, len(lines)))
        for ln in lines:
            print('   ' + json.dumps(ln)[:1500])


if __name__ == '__main__':
    main()
