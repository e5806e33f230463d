#!/usr/bin/env python
"""Benchmark of the MI355X-native Balanced Group Softmax hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload gs_head]

Contract: W untimed warm-up steps, then exactly K timed steps bracketed by barrier +
``torch.cuda.synchronize()``; the MAX over ranks is taken and rank 0 prints ONE JSON line.

Workload ``gs_head`` (this round): the RoI-head loss of BASELINE.json config[1]
(Faster R-CNN R50-FPN + BAGS, 2 img/GPU x 512 RoI = 1024 RoIs x 1236 logits, 25 % fg):
one step = ``_remap_labels`` (device sampling) + fused GroupSoftmax loss forward+backward
(+ the reduce / scale kernels) + the box-regression loss, inputs resident in HBM.
The JSON line carries ``roofline`` for the dominant kernel (the fused streaming kernel,
HIP-event timed on its own stream) and ``cpu_baseline`` = the torch-CPU port of the
reference's loss()+backward() timed on this host (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from balancedgroupsoftmax_amd import capi  # noqa: E402
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402
from balancedgroupsoftmax_amd import gs_tables  # noqa: E402
import bench_dist  # noqa: E402
# (round 6: the workloads, the per-kernel rooflines and the CPU baseline live in their own modules; every name stays
#  importable from here — the golden generators and the tests use `bench.detector_cfg`, `bench.DetectorStep`, ...)
from bench_workloads import (NUM_CLASSES, HBM_PEAK_GBS, make_inputs, GsHeadStep, detector_cfg, _cfg_variant,  # noqa: E402,F401
                             DetectorStep, CONV_MATH_NOTE, try_graph)
from bench_rooflines import (conv_roofline, _event_time_us, _pmc_traffic, kernel_roofline, capture_head_inputs,  # noqa: E402,F401
                             roi_footprint_bytes, hbm_kernel_rooflines, gs_head_metric, STEP_GFLOP, step_layer_floor,
                             roofline_step)
from bench_cpu_baseline import _cpu_time_threads, cpu_baseline, cpu_baseline_detector  # noqa: E402,F401
from bench_dist import barrier, timed_loop  # noqa: E402,F401  (the timed-region contract lives there: one copy for N = 1 and N > 1)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--workload', default='detector', choices=['detector', 'gs_head'])
    ap.add_argument('--imgs', type=int, default=2, help='images per GPU per step (cfg: imgs_per_gpu=2)')
    ap.add_argument('--rois', type=int, default=1024, help='RoIs per GPU per step (2 img x 512)')
    ap.add_argument('--cascade', action='store_true',
                    help='cfg[4]: Cascade R-CNN X101-64x4d-FPN + BAGS, 3 stages (fp32)')
    ap.add_argument('--htc', action='store_true',
                    help='Hybrid Task Cascade X101-64x4d-FPN + BAGS (gs_htc_x101_64x4d_fpn_20e_16gpu_'
                         'lvis): cascade + 3 HTCMaskHeads + semantic branch (fp32)')
    ap.add_argument('--selectp', type=int, default=1, choices=[0, 1, 3],
                    help='1 (as shipped): train bbox_head.fc_cls only; 0: train everything '
                         '(tools/train.py:49-57)')
    ap.add_argument('--mask', action='store_true',
                    help='cfg[3]: add the Mask R-CNN branch (gs_mask_rcnn_r50_fpn_1x_lvis)')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the secondary measurements (selectp=0, Mask R-CNN) of the N=1 run')
    ap.add_argument('--child', action='store_true',
                    help='(internal) measurement sub-process of the single-GPU run')
    ap.add_argument('--no-roofline', action='store_true',
                    help='skip the per-kernel roofline timings (used by the extras sub-runs)')
    ap.add_argument('--no-graph', action='store_true', help='time eager launches, not hipGraph replay')
    ap.add_argument('--launch', choices=('auto', 'graph', 'eager', 'pipelined'), default='auto',
                    help='one GPU: auto = calibrate hipGraph replay, eager launches and the trunk pipeline untimed and '
                         'time the fastest (default) | graph | eager | pipelined (train.TrunkPipeline without '
                         'calibration, depth BGS_BENCH_PIPELINE_DEPTH or 5: for traces)')
    ap.add_argument('--dist-graph', action='store_true',
                    help='N > 1 over RCCL: capture the WHOLE step, gradient all-reduce included, into '
                         'one hipGraph per rank (default for N > 1: eager launches)')
    ap.add_argument('--conv-math', default=None, choices=['bf16x6', 'f32', 'bf16'],
                    help="conv / linear arithmetic: 'bf16x6' (default; bf16 MFMA on exactly split fp32 "
                         "operands, fp32-faithful), 'f32' (v_mfma_f32_32x32x2_f32) or 'bf16' (operands "
                         "rounded to bf16, fp32 accumulate: the reduced-precision mode of cfg[4])")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    a = ap.parse_args()
    if a.conv_math is not None:
        BF.set_conv_math(a.conv_math)
    a.conv_math = BF.conv_math()
    if a.steps is None:
        a.steps = 30 if a.workload == 'detector' else 200
    if a.warmup is None:
        a.warmup = 5 if a.workload == 'detector' else 20
    return a


def init_dist(args):
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # (test hooks: BGS_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and BGS_DIST_BACKEND=gloo
    #  swaps the collective backend, so the multi-process code path can be exercised on a
    #  single-GPU box; the real runs use one GPU per rank over RCCL)
    if os.environ.get('BGS_BENCH_ONE_DEVICE'):
        local = 0
        # two PROCESSES time-sharing one GPU, each with side streams: the device scheduler thrashes between the
        # processes' queues at every event edge (185 ms per step instead of 15: profiles/r8q_dist2_onegpu*.json).
        # One rank per GPU — the real configuration — has no second process on the device; the hook turns the
        # side-stream forks off for itself.
        os.environ.setdefault('BGS_LEVEL_FORK', '0')
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('BGS_DIST_BACKEND', 'nccl')                   # nccl = RCCL on ROCm
        import datetime
        # a collective that one rank never joins raises after this long instead of holding the job for torch's default
        # 10 minutes (the watchdogs of bench_dist.run print the line earlier still)
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=int(os.environ.get('BGS_BENCH_COLL_TIMEOUT', '150'))))
    elif os.environ.get('BGS_BENCH_SELF_GROUP'):
        # test hook: a 1-rank RCCL group whose all-reduce really runs, so that the N > 1 launch
        # policies (eager exchange, --dist-graph) can be exercised on a single-GPU box
        import socket
        import torch.distributed as dist
        from balancedgroupsoftmax_amd import train as BT
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        dist.init_process_group(backend='nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0,
                                world_size=1)
        BT.exchange_at_world_size_one(True)
    return rank, local, world


def extras(dev, args):
    """Secondary single-GPU measurements of the same step in the other §8 configurations (not the
    headline value): selectp=0 (train everything but the frozen stem+layer1), the Mask R-CNN
    config (cfg[3]) and the X101-64x4d cascade (cfg[4], fp32).  Each runs in its OWN process
    (this script with --no-extras), so a failure there can never take the headline line down."""
    import subprocess
    res = {}
    runs = (('f32_mfma_math_selectp1', ['--conv-math', 'f32']),
            ('selectp0', ['--selectp', '0']),
            ('mask_rcnn_selectp1', ['--mask']),
            ('mask_rcnn_selectp0', ['--mask', '--selectp', '0']),
            ('cascade_x101_64x4d_selectp3_fp32', ['--cascade', '--selectp', '3']),
            ('cascade_x101_64x4d_selectp3_bf16', ['--cascade', '--selectp', '3', '--conv-math', 'bf16']),
            ('cascade_x101_64x4d_selectp3_bf16_fp32storage',
             ['--cascade', '--selectp', '3', '--conv-math', 'bf16'], {'BGS_BF16_STORAGE': '0'}),
            ('htc_x101_64x4d_selectp3_fp32', ['--htc', '--selectp', '3']),
            ('htc_x101_64x4d_selectp3_bf16', ['--htc', '--selectp', '3', '--conv-math', 'bf16']))
    for run in runs:
        key, flags = run[0], run[1]
        env = dict(os.environ, **run[2]) if len(run) > 2 else None
        cmd = [sys.executable, os.path.abspath(__file__), '--workload', 'detector', '--steps', '10',
               '--warmup', '3', '--imgs', str(args.imgs), '--no-extras', '--no-cpu-baseline',
               '--no-roofline'] + (['--conv-math', args.conv_math] if '--conv-math' not in flags else []) \
            + flags + (['--no-graph'] if args.no_graph else []) + ['--launch', args.launch]
        try:
            out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420, env=env)
            line = [ln for ln in out.stdout.decode().splitlines() if ln.startswith('{')]
            if out.returncode != 0 or not line:
                res[key] = {'error': 'rc=%d %s' % (out.returncode, out.stderr.decode()[-160:])}
                continue
            d = json.loads(line[-1])
            res[key] = {'img_per_s': d['value'], 'ms_per_step': d['ms_per_step'],
                        'conv_math': d['config'].get('conv_math_mode'),
                        'trainable_params': d['config']['trainable_params'],
                        'loss': d['last_losses']['loss'], 'launch': d['config']['launch']}
        except Exception as e:  # pragma: no cover
            res[key] = {'error': repr(e)[:200]}
    # test time (informational): simple_test on one 800 x 1344 image, 1000 proposals, score_thr = 0, 1230 classes
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'infer_time.py'), '20'],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        line = [ln for ln in out.stdout.decode().splitlines() if ln.startswith('{')]
        res['inference_simple_test'] = json.loads(line[-1]) if out.returncode == 0 and line else \
            {'error': 'rc=%d %s' % (out.returncode, out.stderr.decode()[-160:])}
    except Exception as e:  # pragma: no cover
        res['inference_simple_test'] = {'error': repr(e)[:200]}
    return res


def run_graph_child(args):
    """Single-GPU headline measurement (hipGraph replay of the whole step) in a child process: a
    GPU fault inside a graph replay cannot be caught in-process, so the parent keeps the ability
    to fall back to eager launches and still print its line.  The timed region (barrier + sync
    around exactly K steps) lives entirely in the child."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--workload', 'detector', '--child',
           '--gpus', '1', '--steps', str(args.steps), '--warmup', str(args.warmup),
           '--imgs', str(args.imgs), '--selectp', str(args.selectp), '--no-extras',
           '--no-cpu-baseline', '--no-roofline', '--conv-math', args.conv_math]
    cmd += (['--mask'] if args.mask else []) + (['--cascade'] if args.cascade else [])
    cmd += ['--htc'] if args.htc else []
    cmd += ['--dist-graph'] if args.dist_graph else []
    cmd += ['--launch', args.launch]
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        line = [ln for ln in out.stdout.decode().splitlines() if ln.startswith('{')]
        if out.returncode == 0 and line:
            return json.loads(line[-1])
        sys.stderr.write('graph-replay child failed (rc=%d): %s\n'
                         % (out.returncode, out.stderr.decode()[-300:]))
    except Exception as e:  # pragma: no cover
        sys.stderr.write('graph-replay child failed: %r\n' % (e,))
    return None


def finish_line(out, args, dev, world):
    """Secondary measurements + per-kernel rooflines + CPU baseline, then the ONE JSON line."""
    # (per-kernel rooflines first, the other configurations after them; every measurement here is standalone)
    run_extras = world == 1 and not args.no_extras and args.selectp == 1 and not args.mask \
        and not args.cascade and not args.htc
    if not args.no_roofline:
        out['roofline'] = conv_roofline(dev, args.conv_math)
        if args.conv_math != 'f32':
            out['roofline_f32_mfma_kernel'] = conv_roofline(dev, 'f32')
        if args.conv_math == 'bf16x6':
            # the halo kernels the planes kernel of `roofline` replaced on this layer (round 6), same layer, same box: the
            # one-launch form and the two-launch wide schedule (what the pipelined steps of rounds 4 - 5 ran)
            out['roofline_halo_kernel'] = conv_roofline(dev, args.conv_math, planes3=False)
            out['roofline_wide_schedule'] = conv_roofline(dev, args.conv_math, wide=1, planes3=False)
        rs = roofline_step(out, args)
        if rs:
            out['roofline_step'] = rs
        gs_inp = make_inputs(1024, seed=1000, dev=dev)
        out['roofline_gs_loss'] = kernel_roofline(gs_inp, 1024, kernel='fused')
        out['roofline_gs_loss_rowwave'] = kernel_roofline(gs_inp, 1024, kernel='rowwave')
        big = make_inputs(65536, seed=7, dev=dev)
        out['roofline_gs_loss_n65536'] = kernel_roofline(big, 65536, iters=30, kernel='rowwave')
        del big
        if world == 1:
            out['gs_head'] = gs_head_metric(gs_inp, 1024)
        del gs_inp
        torch.cuda.empty_cache()
        if world == 1 and not args.mask and not args.cascade and not args.htc:
            try:
                out.update(hbm_kernel_rooflines(dev, args.conv_math))
            except Exception as e:  # pragma: no cover  (never lose the line to a secondary measurement)
                out['roofline_hbm_kernels_error'] = repr(e)[:300]
            torch.cuda.empty_cache()
    if run_extras:
        out['also_measured'] = extras(dev, args)
    if world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline(1024, args.cpu_seconds)
        cb['note'] = ('GroupSoftmax loss()+backward() only: the reference cannot run the whole '
                      'detector on CPU as shipped (its RoIAlign has no CPU path, roi_align.py:27-28); '
                      'see cpu_baseline_detector for the whole iteration with its ops built for the host')
        out['cpu_baseline'] = cb
        out['cpu_baseline_1thread'] = {'value': cb['threads_tried'].get('1'), 'unit': 'us/RoI',
                                       'cores': 1, 'kind': cb['kind']}
        cbd = cpu_baseline_detector()
        if cbd:
            out['cpu_baseline_detector'] = cbd
    print(json.dumps(out), flush=True)


def run_dist_graph_children(args, rank, local, world):
    """N > 1: the whole-step hipGraph policy (RCCL all-reduce captured with the rest of the step)
    measured in CHILD processes — one per rank, forming their own process group on another port —
    exactly as the 1-GPU path isolates its graph replay: a fault or a hang inside a replay cannot
    take the parent's eager measurement (and its JSON line) down.  Runs BEFORE the parent builds its
    model, so the GPU is the child's alone.  Returns a dict on rank 0 (None elsewhere)."""
    import socket
    import subprocess
    import torch.distributed as dist
    port = torch.zeros(1, dtype=torch.int64, device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
    if rank == 0:
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port[0] = sk.getsockname()[1]
    dist.broadcast(port, 0)
    env = dict(os.environ, MASTER_PORT=str(int(port.item())), RANK=str(rank), WORLD_SIZE=str(world),
               LOCAL_RANK=str(local))
    env.setdefault('MASTER_ADDR', '127.0.0.1')
    for k in list(env):
        if k.startswith('TORCHELASTIC_'):
            env.pop(k)
    cmd = [sys.executable, os.path.abspath(__file__), '--workload', 'detector', '--child', '--dist-graph',
           '--gpus', str(world), '--steps', str(min(args.steps, 10)), '--warmup', '3',
           '--imgs', str(args.imgs), '--selectp', str(args.selectp), '--no-extras', '--no-cpu-baseline',
           '--no-roofline', '--conv-math', args.conv_math]
    res = dict(ok=False)
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        try:
            so, se = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            p.kill()                      # exactly the child this rank started
            so, se = p.communicate()
            res['error'] = 'timeout'
        line = [ln for ln in so.decode().splitlines() if ln.startswith('{')]
        if p.returncode == 0 and line:
            d = json.loads(line[-1])
            res = dict(ok='hipGraph' in d['config']['launch'], ms_per_step=d['ms_per_step'],
                       img_per_s=d['value'], launch=d['config']['launch'], steps=d['steps'])
        elif 'error' not in res:
            res['error'] = 'rc=%s %s' % (p.returncode, se.decode()[-200:])
    except Exception as e:  # pragma: no cover
        res['error'] = repr(e)[:200]
    barrier(world)
    return res if rank == 0 else None


def detector_line(args, step, world, imgs_per_s, ms_per_step, graph, pipelined, depth, self_group=False):
    """The fields of the detector JSON line that describe WHAT was timed (shared by the N = 1 and N > 1 paths)."""
    cfg_name = 'gs_faster_rcnn_r50_fpn_1x_lvis_with0_bg8 (cfg[1])'
    if args.htc:
        cfg_name = 'gs_htc_x101_64x4d_fpn_20e_16gpu_lvis (cfg[4], HTC)'
    elif args.cascade:
        cfg_name = 'gs_cascade_rcnn_x101_64x4d_fpn_1x_lvis (cfg[4])'
    elif args.mask:
        cfg_name = 'gs_mask_rcnn_r50_fpn_1x_lvis (cfg[3])'
    lv = {k: round(float(v), 5) for k, v in step.last.items()}
    return {
        'metric': 'img/s fwd+bwd R50-FPN+BAGS 1333x800, 512 RoI (BASELINE metric: img/s/GPU '
                  'fwd+bwd R50-FPN+BAGS 1333x800, 512 RoI; GroupSoftmax us/RoI)',
        'value': round(imgs_per_s, 3), 'unit': 'img/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': {'f32': 'f32', 'bf16': 'bf16 (conv/linear operands rounded to bf16, fp32 accumulate; %s; '
                                               'fp32 master weights and losses)'
                                               % ('bf16 storage of the frozen trunk activations'
                                                  if os.environ.get('BGS_BF16_STORAGE', '1') != '0'
                                                  else 'fp32 storage'),
                  'bf16x6': 'f32 (conv/linear products on the bf16 MFMA from exact 3-way bf16 '
                            'splits, fp32 accumulate: fp32-faithful)'}[args.conv_math],
        'data': 'synthetic',
        'config': {'conv_math_mode': args.conv_math, 'conv_math': CONV_MATH_NOTE[args.conv_math],
                   'workload': cfg_name + ' training '
                               'iteration %s, grad all-reduce, clip 35, SGD): '
                               '%d img/GPU, 3x800x1344 (1333x800 padded /32), 20 GT/img, '
                               '512 RoI/img, 1231 classes, 5 bins; random-init weights'
                               % (('as shipped (selectp=1: full forward incl. RPN '
                                   'losses/proposals/NMS/assign/sample/RoIAlign/FC heads/'
                                   'GroupSoftmax loss, backward through fc_cls')
                                  if args.selectp == 1 else
                                  ('with selectp=0 (train everything but the frozen stem + '
                                   'layer1: full forward and full backward through heads, '
                                   'RoIAlign, RPN, FPN, ResNet layer2-4'), args.imgs),
                   'selectp': args.selectp, 'mask_branch': bool(args.mask),
                   'cascade_x101': bool(args.cascade), 'htc_x101': bool(args.htc),
                   'trainable_params': int(sum(p.numel() for p in step.params)),
                   'imgs_per_gpu': args.imgs, 'rois_per_img': 512,
                   'launch': ('hipGraph replay of the whole step (forward+losses+backward+'
                              + ('RCCL all-reduce+' if (world > 1 or self_group) else '') +
                              'clip+SGD)') if graph else
                   ('eager launches, %d-stage software pipeline (train.TrunkPipeline): the frozen trunk of the '
                    'batches ahead, in %d piece(s) on their own streams (depth 3: backbone(i+2) | FPN(i+1)), beside '
                    'batch i\'s RPN / proposal chain / RoI heads / losses / backward / optimizer step; one pass of '
                    'every piece + one head pass per timed step, bit-identical training (tests/test_gpu_e2e.py)'
                    % (depth, depth - 1) if pipelined else 'eager launches'),
                   'parallelism': 'dp%d (one process per GPU; flat fp32 all-reduce of the '
                                  '%d trainable grads over RCCL)'
                                  % (world, sum(p.numel() for p in step.params)),
                   'ranks': world,
                   'collective_backend': (__import__('torch.distributed').distributed.get_backend()
                                          if world > 1 else None)},
        'img_per_s_per_gpu': round(imgs_per_s / world, 3),
        'last_losses': lv,
    }


def main_detector_dist(args, rank, local, world, dev, step, dist_graph):
    """N > 1 (one process per GPU, launched by torch.distributed.run): calibration, the K timed steps and the
    diagnostics run through bench_dist.run — headline first, every diagnostic guarded and bounded (its docstring)."""
    import torch.distributed as dist

    def make_one_rank_step():
        return DetectorStep(dev, 0, 1, args.imgs, args.selectp, args.mask, args.cascade, args.htc,
                            conv_math=args.conv_math)

    def emit(f):
        if 'ms_per_step' not in f:              # the watchdog fired before the timed region finished
            print(json.dumps({'metric': 'img/s fwd+bwd R50-FPN+BAGS 1333x800, 512 RoI', 'value': None, 'unit': 'img/s',
                              'n_gpus': world, 'error': f.get('error', 'no measurement')}), flush=True)
            return
        depth = f['pipeline_depth']
        out = detector_line(args, step, world, args.imgs * world * args.steps / f['dt'], f['ms_per_step'], None,
                            bool(depth), depth)
        out['rccl_ranks'] = world if dist.get_backend() == 'nccl' else 0
        out['launch_policy'] = 'eager launches (the RCCL all-reduce between backward and the optimizer)'
        out['ms_per_step_by_rank'] = dict(min=min(f['rank_ms']), max=max(f['rank_ms']), all=f['rank_ms'])
        calib = f.get('launch_calibration')
        if calib is not None:
            out['launch_calibration'] = calib
            if depth and calib.get('eager_forks_on_ms') is not None:
                out['ms_per_step_eager'] = calib['eager_forks_on_ms']
        for k, v in (f.get('diagnostics') or {}).items():
            out[k] = v
        if dist_graph is not None:
            out['dist_graph_policy'] = dist_graph
        if f.get('watchdog'):
            # printed from the watchdog thread while the main thread may be stuck in a collective: no GPU work here
            out['watchdog'] = f['watchdog']
            print(json.dumps(out), flush=True)
            return
        finish_line(out, args, dev, world)

    bench_dist.run(step, args, rank, world, make_one_rank_step, emit,
                   launch_auto=args.launch == 'auto' and not os.environ.get('BGS_BENCH_NO_DIST_CALIB'))


def main_detector(args, rank, local, world, dev):
    fallback_note = None
    if world == 1 and not args.no_graph and not args.child:
        out = run_graph_child(args)
        if out is not None:
            finish_line(out, args, dev, world)
            return
        fallback_note = 'hipGraph replay failed in the measurement child; eager launches instead'
        args.no_graph = True
    if args.child and os.environ.get('BGS_BENCH_CHILD_FAIL'):     # test hook for the fallback path
        os._exit(134)
    dist_graph = None
    if world > 1 and not args.child and not args.no_graph and not args.dist_graph \
            and os.environ.get('BGS_BENCH_DIST_GRAPH_CHILD'):
        # opt-in (round 6): the whole-step hipGraph with the RCCL all-reduce captured, in child processes (up to 180 s)
        dist_graph = run_dist_graph_children(args, rank, local, world)
    step = DetectorStep(dev, rank, world, args.imgs, args.selectp, args.mask, args.cascade,
                        args.htc, conv_math=args.conv_math)
    if world > 1 and not args.child and not args.dist_graph:
        return main_detector_dist(args, rank, local, world, dev, step, dist_graph)
    # Launch policy.  The iteration is free of host synchronisation, so on ONE GPU the whole
    # step (forward, losses, backward, clip, SGD: ~560 launches) is captured into a single
    # hipGraph and replayed.  The graph must own the ENTIRE step: on ROCm 7.2 a large graph whose
    # replays are interleaved with eager launches (an eager optimizer step, or the host-side
    # philox bookkeeping of torch.randint inside a captured region) faults after a few dozen
    # replays (tools/two_graphs_repro.py reproduces it: "variants plain" vs "variants whole";
    # DESIGN.md §5).  With N > 1 the gradient all-reduce (RCCL) sits between backward and the
    # optimizer, so multi-GPU runs launch eagerly (main_detector_dist above) — the step is GPU-bound and eager launches
    # cost nothing measurable (10.99 vs 10.96 ms); what is left here for N > 1 is `--dist-graph` (the whole step incl.
    # the collective in one hipGraph per rank) and its measurement children.
    graph = None
    self_group = bool(os.environ.get('BGS_BENCH_SELF_GROUP'))
    rccl = world > 1 and __import__('torch.distributed').distributed.get_backend() == 'nccl'
    # Launch policy on one GPU, `--launch auto` (default): the step forks independent launches onto a side stream
    # (functional.forked); replayed from a hipGraph those forks cost event edges, launched eagerly they run as real
    # concurrent streams but pay the host's launch path — which of the two is faster depends on the box and its
    # host (6.30 vs 6.20 ms on one, 6.40 vs 6.42 on another).  Both are calibrated UNTIMED (8 steps each; the eager
    # one before the capture, so that no replay ever follows an eager step that followed a replay: DESIGN 5) and
    # the official K steps are then timed under the faster policy.
    calib = None
    can_graph = not args.no_graph and ((world == 1 and not self_group) or
                                       (args.dist_graph and (rccl or self_group)))
    auto = can_graph and world == 1 and not self_group and args.launch == 'auto'
    pipe_fn = None
    if auto:
        calib = dict(eager_ms=round(timed_loop(step, 8, 6, 1) * 1e3 / 8, 3))      # (6 warm-up steps: lazy folds / splits / caches)
        if step.can_pipeline() and not os.environ.get('BGS_BENCH_NO_PIPELINE'):
            # third policy (round 5): eager launches with the NEXT batch's frozen trunk on its own stream beside this
            # batch's heads / losses / backward / optimizer step (train.TrunkPipeline: bit-identical training)
            depths = [int(os.environ['BGS_BENCH_PIPELINE_DEPTH'])] if os.environ.get('BGS_BENCH_PIPELINE_DEPTH') \
                else [3, 4, 5]
            best_depth = None
            for dpt in depths:
                pipe_fn = step.pipelined(depth=dpt)
                ms = round(timed_loop(pipe_fn, 8, 3, 1) * 1e3 / 8, 3)
                pipe_fn.drain()
                torch.cuda.synchronize()
                calib['eager_pipelined_depth%d_ms' % dpt] = ms
                if best_depth is None or ms < calib['eager_pipelined_ms']:
                    best_depth, calib['eager_pipelined_ms'] = dpt, ms
            calib['pipeline_depth'] = best_depth
    if can_graph and args.launch not in ('eager', 'pipelined'):
        # --dist-graph: the RCCL all-reduce is captured with the rest of the step (the communicator
        # is set up by the eager warm-up iterations inside try_graph)
        graph = try_graph(step)
    fn = graph.replay if graph is not None else step
    if auto and graph is not None:
        calib['graph_ms'] = round(timed_loop(graph.replay, 8, 3, 1) * 1e3 / 8, 3)
        if calib['eager_ms'] < 0.99 * calib['graph_ms']:
            fn = step
        calib['chosen'] = 'eager' if fn is step else 'graph'
    pipelined = False
    if auto and pipe_fn is not None:
        best = min(calib['eager_ms'], calib.get('graph_ms', 1e9))
        if calib['eager_pipelined_ms'] < 0.99 * best:
            fn = step.pipelined(depth=calib['pipeline_depth'])      # (a fresh pipeline: its first features are launched here, untimed)
            pipelined = True
            calib['chosen'] = 'eager_pipelined'
    if args.launch == 'pipelined' and world == 1 and step.can_pipeline():
        fn, pipelined = step.pipelined(depth=int(os.environ.get('BGS_BENCH_PIPELINE_DEPTH', 5))), True
    dt = timed_loop(fn, args.steps, args.warmup, world)
    if pipelined:
        fn.drain()                             # (the features launched by the last timed call: consumed by nobody)
        torch.cuda.synchronize()
    ms_per_step = dt * 1e3 / args.steps
    imgs_per_s = args.imgs * world * args.steps / dt
    rank_ms = None
    if world > 1:       # every rank's own wall time of the timed region (the line reports the max)
        rank_ms = [round(t, 3) for t in bench_dist.gather_scalar(timed_loop.last_local_dt * 1e3 / args.steps, world)]
    ms_eager = None
    ms_graph = None
    if graph is not None and (fn is step or pipelined):      # eager was the timed policy: the graph figure from the calibration
        ms_graph = calib['graph_ms']
        graph = None                           # (the line's `launch` describes what was timed)
    if pipelined:
        ms_eager = (calib or {}).get('eager_ms')
    elif graph is not None:     # every rank: the same step launched eagerly, for the graph-vs-eager figure
        ms_eager = round(timed_loop(step, 5, 2, world) * 1e3 / 5, 3)
    if rank == 0:
        out = detector_line(args, step, world, imgs_per_s, ms_per_step, graph, pipelined,
                            fn.depth if pipelined else 0, self_group)
        if ms_eager is not None:
            out['ms_per_step_eager'] = ms_eager
        if ms_graph is not None:
            out['ms_per_step_graph'] = ms_graph
        if calib is not None:
            out['launch_calibration'] = dict(calib, note='untimed calibration of both launch policies (8 steps each) '
                                                    'before the timed region; the K timed steps ran under `chosen`')
        if world > 1:
            import torch.distributed as dist
            out['rccl_ranks'] = world if dist.get_backend() == 'nccl' else 0
            out['launch_policy'] = 'hipGraph replay incl. the RCCL all-reduce' if graph else \
                'eager launches (the RCCL all-reduce between backward and the optimizer)'
            out['ms_per_step_by_rank'] = dict(min=min(rank_ms), max=max(rank_ms), all=rank_ms)
        if self_group:
            out['config']['collective_backend'] = 'nccl (1-rank group: test hook BGS_BENCH_SELF_GROUP)'
        if fallback_note:
            out['config']['launch_note'] = fallback_note
        finish_line(out, args, dev, world)
    barrier(world)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run (one
    process per GPU, the reference's tools/dist_train.sh:8-9 layout) and relay its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node',
           str(args.gpus), '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the product has no CPU path)')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(spawn_ranks(args))
    rank, local, world = init_dist(args)
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d: the line would report the wrong '
                         'n_gpus' % (args.gpus, world))
    if world > 1:
        import torch.distributed as dist
        assert dist.get_world_size() == args.gpus
    dev = torch.device('cuda', local)
    if args.workload == 'detector':
        return main_detector(args, rank, local, world, dev)
    n = args.rois
    inp = make_inputs(n, seed=1000 + rank, dev=dev)
    step = GsHeadStep(inp)
    graph = None if args.no_graph else try_graph(step)
    fn = graph.replay if graph is not None else step
    dt = timed_loop(fn, args.steps, args.warmup, world)
    ms_per_step = dt * 1e3 / args.steps
    us_per_roi = dt * 1e6 / (args.steps * n * world)

    if rank == 0:
        # eager (python launch overhead included) for reference
        dt_eager = None
        if graph is not None and world == 1:
            dt_eager = timed_loop(step, min(args.steps, 100), 5, 1) * 1e3 / min(args.steps, 100)
        out = {
            'metric': 'GroupSoftmax us/RoI (loss fwd+bwd; BASELINE metric: img/s/GPU fwd+bwd '
                      'R50-FPN+BAGS 1333x800, 512 RoI; GroupSoftmax us/RoI)',
            'value': round(us_per_roi, 6), 'unit': 'us/RoI', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 5),
            'higher_is_better': False, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'gs_head: RoI-head loss of gs_faster_rcnn_r50_fpn_1x_lvis_'
                                   'with0_bg8 (cfg[1]): %d RoIs/GPU (2 img x 512) x 1236 logits, '
                                   '1231 classes, 5 bins, 25%% fg; _remap_labels + fused '
                                   'GroupSoftmax loss fwd+bwd + loss_bbox' % n,
                       'rois_per_gpu': n, 'launch': 'hipGraph replay' if graph else 'eager',
                       'parallelism': 'dp%d (independent RoI batches, no data-path collective)'
                                      % world},
            'rois_per_s': round(n * world * args.steps / dt, 1),
        }
        if dt_eager is not None:
            out['ms_per_step_eager'] = round(dt_eager, 5)
        out['roofline'] = kernel_roofline(inp, n, kernel='fused' if n <= BF.GS_FUSED_MAX_ROWS else 'rowwave')
        out['roofline_rowwave'] = kernel_roofline(inp, n, kernel='rowwave')
        big = make_inputs(65536, seed=7, dev=dev)
        out['roofline_n65536'] = kernel_roofline(big, 65536, iters=30, kernel='rowwave')
        del big
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(n, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    barrier(world)


if __name__ == '__main__':
    main()
