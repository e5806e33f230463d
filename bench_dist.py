"""N > 1 side of bench.py: one process per GPU over ``torch.distributed`` (RCCL; gloo in the CPU dry run).

The reference's whole parallel story is tools/dist_train.sh:8-9 + mmdet/core/utils/dist_utils.py:9-58 + the loop of
mmdet/apis/train.py:143-205: rank-local iterations and ONE flat gradient all-reduce per step.  This module holds what
``bench.py --gpus N`` does around that step — launch-policy calibration, the timed region, and the diagnostics that ride
in the JSON line — written so that the first contact with a multi-GPU node produces a line instead of a hang:

* order: build -> (untimed) calibration -> the K timed steps -> diagnostics.  The headline is measured BEFORE any
  diagnostic runs; a diagnostic can only cost its own field.
* every diagnostic is ``local phase -> agree() -> collective phase``: whatever can raise on one rank alone (building a
  second model, a forward / backward, a test hook) runs before ONE all-reduce of an ok flag; the collectives of the
  diagnostic are issued only when every rank said ok, so no rank waits alone inside an all-reduce.  A failure becomes
  ``{"error": ...}`` in that diagnostic's field.
* a watchdog bounds the diagnostics (``BGS_BENCH_DIAG_SECONDS``, default 90 s) and everything after the
  model build (``BGS_BENCH_WALL_SECONDS``, default 200 s): on expiry rank 0 prints the line with what it has (or an error line if the
  headline itself is missing) and every rank leaves with ``os._exit`` — a hung collective cannot hold the job.
* default arms for N > 1: eager launches and the per-rank trunk pipeline (depth 4).  The other arms (forks off, depth 5)
  are ``BGS_BENCH_DIST_CALIB=full``; the hipGraph-with-RCCL child job is ``BGS_BENCH_DIST_GRAPH_CHILD=1`` (bench.py).
* test hooks (tests/test_bench_dist_cpu.py drives all of them over gloo, world size 8, on a stand-in step):
  ``BGS_BENCH_FAIL=<name>[@rank]`` raises in the local phase of diagnostic ``<name>`` (on one rank or all),
  ``BGS_BENCH_HANG=<name>[@rank]`` sleeps there instead (the watchdog's case).

Nothing here touches the oracle or the reference; the step object is the caller's (``bench.DetectorStep`` on the GPU,
a tiny CPU model in the dry run) and only needs ``__call__``, ``compute``, ``params``, ``train.allreduce_grads``,
``step_fn.overlap``, ``step_fn.optimizer``, ``can_pipeline`` and ``pipelined``.
"""
import json
import os
import sys
import threading
import time

import torch

DIAG_NAMES = ('calibration', 'grad_exchange_check', 'allreduce_us', 'n1_reference')


def _backend():
    import torch.distributed as dist
    return dist.get_backend()


def _coll_device():
    return 'cuda' if _backend() == 'nccl' else 'cpu'


def sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        if dist.get_backend() == 'nccl':
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def timed_loop(fn, steps, warmup, world):
    """bench.py's contract: W untimed calls, then exactly K calls bracketed by barrier + synchronize; returns the MAX over
    ranks (``timed_loop.last_local_dt`` keeps this rank's own time)."""
    for _ in range(warmup):
        fn()
    sync()
    barrier(world)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    barrier(world)
    dt = time.perf_counter() - t0
    timed_loop.last_local_dt = dt
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=_coll_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


timed_loop.last_local_dt = 0.0


def gather_scalar(value, world):
    """every rank's float, as a list (one all-gather)."""
    import torch.distributed as dist
    mine = torch.tensor([float(value)], dtype=torch.float64, device=_coll_device())
    allr = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    return [float(t.item()) for t in allr]


def agree(ok, world):
    """ONE all-reduce (MIN) of an ok flag: True only when every rank said ok.  Called between the local phase of a
    diagnostic and its collectives."""
    if world == 1:
        return bool(ok)
    import torch.distributed as dist
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=_coll_device())
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def _hook(kind, name, rank):
    spec = os.environ.get(kind, '')
    if not spec:
        return False
    for item in spec.split(','):
        n, _, r = item.partition('@')
        if n == name and (r == '' or int(r) == rank):
            return True
    return False


def test_hooks(name, rank):
    """``BGS_BENCH_FAIL`` / ``BGS_BENCH_HANG`` (module docstring): called at the top of every local phase."""
    if _hook('BGS_BENCH_HANG', name, rank):
        time.sleep(3600)
    if _hook('BGS_BENCH_FAIL', name, rank):
        raise RuntimeError('forced failure of %s on rank %d (BGS_BENCH_FAIL test hook)' % (name, rank))


class Watchdog(object):
    """Bounds a phase in wall time.  ``on_expire()`` runs on the watchdog thread (rank 0: print the line) and the process
    then leaves through ``os._exit(code)`` — the only exit a thread blocked inside a collective cannot hold up."""

    def __init__(self, seconds, on_expire, code=0):
        self.seconds, self.on_expire, self.code = float(seconds), on_expire, code
        self._done = threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def _run(self):
        if self._done.wait(self.seconds):
            return
        try:
            self.on_expire()
        finally:
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(self.code)

    def cancel(self):
        self._done.set()


def _err(e):
    return {'error': ('%s: %s' % (type(e).__name__, e))[:300]}


# ---------------------------------------------------------------------------------------------------------------------
# calibration (before the timed region)
# ---------------------------------------------------------------------------------------------------------------------
def calibrate(step, world, rank):
    """`--launch auto`, N > 1: the launch policies are calibrated UNTIMED on all ranks (8 steps per arm, the gradient
    exchange included; the max over ranks decides, so every rank takes the same arm) and the K timed steps run under
    the fastest.  Default arms: eager launches with the side-stream forks, and the per-rank trunk pipeline of
    train.TrunkPipeline (depth 4) when the trunk is frozen; ``BGS_BENCH_DIST_CALIB=full`` adds forks-off and depth 5,
    ``=0`` skips the calibration (eager).  Whether an arm exists is decided LOCALLY and agreed on before its first
    collective; an arm that raises inside its steps cannot be recovered from (its collectives are half issued) and ends
    the run through the caller.  -> (record, function to time, pipeline depth or 0)."""
    mode = os.environ.get('BGS_BENCH_DIST_CALIB', '1')
    if mode == '0':
        return dict(chosen='forks_on', note='calibration skipped (BGS_BENCH_DIST_CALIB=0)'), step, 0
    ok, can_pipe = True, False
    try:
        test_hooks('calibration', rank)
        can_pipe = bool(step.can_pipeline()) and not os.environ.get('BGS_BENCH_NO_PIPELINE')
    except Exception as e:
        ok, err = False, _err(e)
    if not agree(ok, world):
        rec = dict(chosen='forks_on', note='calibration skipped: a rank failed before its first arm')
        if not ok:
            rec.update(err)
        return rec, step, 0
    can_pipe = agree(can_pipe, world)
    names = [('forks_on', '1', 0)]
    if mode == 'full':
        names.append(('forks_off', '0', 0))
    if can_pipe:
        names.append(('pipelined', '1', 4))
        if mode == 'full':
            names.append(('pipelined_depth5', '1', 5))
    arms, per_rank = {}, {}
    for name, val, pipe in names:
        os.environ['BGS_LEVEL_FORK'] = val
        fn = step.pipelined(depth=pipe) if pipe else step
        dt = timed_loop(fn, 8, 4 if name == 'forks_on' else 2, world)
        if pipe:
            fn.drain()
            sync()
        arms[name] = round(dt * 1e3 / 8, 3)
        per_rank[name] = [round(t, 3) for t in gather_scalar(timed_loop.last_local_dt * 1e3 / 8, world)]
    chosen, depth = 'forks_on', 0
    if 'forks_off' in arms and arms['forks_off'] < 0.99 * arms['forks_on']:
        chosen = 'forks_off'
    pipes = [n for n in ('pipelined', 'pipelined_depth5') if n in arms]
    if pipes:
        best = min(pipes, key=lambda n: arms[n])
        if arms[best] < 0.99 * arms[chosen]:
            chosen, depth = 'pipelined', (4 if best == 'pipelined' else 5)
    os.environ['BGS_LEVEL_FORK'] = '0' if chosen == 'forks_off' else '1'
    rec = dict(chosen=chosen, ms_by_rank=per_rank,
               note='untimed calibration on all ranks (8 eager steps per arm incl. the gradient exchange, max over '
                    'ranks); the K timed steps ran under `chosen`; arms: %s (BGS_BENCH_DIST_CALIB=full for forks-off '
                    'and depth 5; the whole-step-graph arm is opt-in: BGS_BENCH_DIST_GRAPH_CHILD=1)'
                    % ', '.join(n for n, _, _ in names))
    for name in arms:
        rec['eager_%s_ms' % name] = arms[name]
    if depth:
        rec['pipeline_depth'] = depth
    return rec, (step.pipelined(depth=depth) if depth else step), depth


# ---------------------------------------------------------------------------------------------------------------------
# diagnostics (after the timed region, under the watchdog)
# ---------------------------------------------------------------------------------------------------------------------
def exchange_check(step, world, rank):
    """the N-rank exchanged gradient == the mean of the N single-rank gradients on the same
    per-rank inputs.  Local phase: one forward + backward, the local gradients kept.  Collective phase: the product's
    exchange (train.allreduce_grads: flat SUM all-reduce / world, mmdet/core/utils/dist_utils.py:9-41), an all-gather
    of the local gradients, the comparison."""
    import torch.distributed as dist
    ok, res, local, params = True, None, None, None
    try:
        test_hooks('grad_exchange_check', rank)
        if step.step_fn.overlap is not None:
            res = dict(checked=False, why='bucketed exchange overlapped with backward (selectp=0): the local gradients '
                                          'are replaced bucket by bucket')
        else:
            step.compute()
            params = [p for p in step.params if p.grad is not None]
            local = torch.cat([p.grad.reshape(-1) for p in params]).float().clone()
            sync()
    except Exception as e:
        ok, res = False, dict(checked=False, **_err(e))
    if not agree(ok, world):
        return res if not ok else dict(checked=False, error='another rank failed in the local phase')
    if not agree(res is None, world):             # (the overlap case is the same on every rank; agreed on anyway)
        return res or dict(checked=False, why='another rank runs the overlapped exchange')
    n_el = [int(v) for v in gather_scalar(local.numel(), world)]
    if len(set(n_el)) != 1:                       # ranks disagree on which parameters have gradients: no flat layout
        step.step_fn.optimizer.zero_grad(set_to_none=False)
        return dict(checked=False, error='ranks hold different gradient sets: %s elements' % n_el)
    step.train.allreduce_grads(step.params, world)
    got = torch.cat([p.grad.reshape(-1) for p in params]).float()
    if dist.get_backend() != 'nccl':
        local, got = local.cpu(), got.cpu()
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    mean = torch.stack(gathered).sum(0) / world
    diff = float((got - mean).abs().max())
    scale = float(mean.abs().max())
    differ = float((gathered[0] - gathered[-1]).abs().max()) if world > 1 else 0.0
    step.step_fn.optimizer.zero_grad(set_to_none=False)
    return dict(checked=True, ok=bool(diff <= 1e-5 * scale + 1e-12), max_abs_diff=diff, max_abs_grad=scale,
                ranks_see_different_data=bool(differ > 0), elements=int(local.numel()),
                what='allreduce_grads(trainable grads) vs mean of the all-gathered per-rank gradients')


def allreduce_us(step, world, rank, iters=10):
    """wall time of the gradient exchange alone (flat fp32 all-reduce + /world + unflatten), max over ranks."""
    ok, err = True, None
    try:
        test_hooks('allreduce_us', rank)
        if step.step_fn.overlap is not None:
            ok, err = False, dict(skipped='overlapped bucketed exchange: no separate exchange phase')
        for p in step.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
    except Exception as e:
        ok, err = False, _err(e)
    if not agree(ok, world):
        return err or dict(error='another rank failed in the local phase')
    for _ in range(3):
        step.train.allreduce_grads(step.params, world)
    sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        step.train.allreduce_grads(step.params, world)
    sync()
    us = (time.perf_counter() - t0) * 1e6 / iters
    step.step_fn.optimizer.zero_grad(set_to_none=False)
    return round(max(gather_scalar(us, world)), 2)


def n1_reference(make_one_rank_step, steps, warmup, rank, world, ms_per_step_n, depth, imgs):
    """The same step on ONE rank of the same node in the same invocation (rank 0 alone, the other ranks waiting at the
    closing barrier) so that the line carries the weak-scaling efficiency against a number taken on this very box — the
    driver computes its own from separate runs; this one removes box-to-box spread.  No collective inside: a failure on
    rank 0 costs the field only."""
    out = None
    if rank == 0:
        try:
            test_hooks('n1_reference', rank)
            one = make_one_rank_step()
            fn1 = one.pipelined(depth=depth if depth > 1 else 4) if (depth and one.can_pipeline()) else one
            dt = timed_loop(fn1, steps, max(warmup, 4), 1)
            if fn1 is not one:
                fn1.drain()
                sync()
            ms1 = dt * 1e3 / steps
            out = dict(n1_same_invocation=dict(
                ms_per_step=round(ms1, 3), img_per_s=round(imgs * steps / dt, 3),
                launch='eager launches%s, rank 0 alone, other ranks idle'
                       % ((' (%d-stage pipeline)' % fn1.depth) if fn1 is not one else '')),
                weak_scaling_eff=round(ms1 / ms_per_step_n, 4))
            del one
        except Exception as e:
            out = dict(n1_same_invocation=_err(e))
    else:
        try:
            test_hooks('n1_reference', rank)
        except Exception:
            pass
    barrier(world)
    return out


def run(step, args, rank, world, make_one_rank_step, emit, launch_auto=True):
    """Everything bench.py does for N > 1 between the model build and the JSON line.  ``emit(fields)`` is called exactly
    once, on rank 0, with: ms_per_step, dt, pipelined depth, per-rank times, launch calibration, the diagnostics —
    from the main thread when all went well, from the watchdog thread with what exists when a phase ran out of time."""
    state = dict(fields=None, emitted=False)
    lock = threading.Lock()

    def emit_once(extra=None):
        with lock:
            if state['emitted']:
                return
            state['emitted'] = True
            if rank == 0:
                f = dict(state['fields'] or {})
                if extra:
                    f.update(extra)                       # (`watchdog` in the fields: the caller prints at once, no GPU work)
                emit(f if state['fields'] is not None else dict(error=(extra or {}).get('watchdog', 'no measurement')))

    wall = float(os.environ.get('BGS_BENCH_WALL_SECONDS', '200'))
    wd_all = Watchdog(wall, lambda: emit_once(dict(watchdog='wall budget of %.0f s exceeded in the %s phase'
                                                           % (wall, state.get('phase', '?')))), code=0)
    state['phase'] = 'calibration'
    calib, fn, depth = (None, step, 0)
    if launch_auto:
        calib, fn, depth = calibrate(step, world, rank)
    state['phase'] = 'timed'
    dt = timed_loop(fn, args.steps, args.warmup, world)
    if depth:
        fn.drain()
        sync()
    ms_per_step = dt * 1e3 / args.steps
    rank_ms = [round(t, 3) for t in gather_scalar(timed_loop.last_local_dt * 1e3 / args.steps, world)]
    fields = dict(dt=dt, ms_per_step=ms_per_step, pipeline_depth=depth, rank_ms=rank_ms, launch_calibration=calib,
                  step_fn=fn, diagnostics={})
    state['fields'] = fields
    # ---- diagnostics: bounded, each one guarded; the headline above is already safe
    state['phase'] = 'diagnostics'
    budget = float(os.environ.get('BGS_BENCH_DIAG_SECONDS', '90'))
    wd = Watchdog(budget, lambda: emit_once(dict(watchdog='diagnostics exceeded %.0f s (%s unfinished)'
                                                         % (budget, state.get('diag', '?')))), code=0)
    diag = fields['diagnostics']
    if not os.environ.get('BGS_BENCH_NO_DIAG'):
        state['diag'] = 'grad_exchange_check'
        diag['grad_exchange_check'] = exchange_check(step, world, rank)
        state['diag'] = 'allreduce_us'
        diag['allreduce_us'] = allreduce_us(step, world, rank)
        if not os.environ.get('BGS_BENCH_NO_N1_REFERENCE'):
            state['diag'] = 'n1_reference'
            n1 = n1_reference(make_one_rank_step, args.steps, args.warmup, rank, world, ms_per_step, depth, args.imgs)
            if n1:
                diag.update(n1)
    wd.cancel()
    wd_all.cancel()
    emit_once()
    barrier(world)
    return fields


def dumps_line(out):
    return json.dumps(out)
