"""bench.py's `cpu_baseline` legs: the reference's own CPU path (oracle/_ref, kind "reference") or the numpy / torch port
(kind "port") timed on the host cores of the GPU box, rank 0, N = 1, on a bounded sample.  The ONLY part of bench.py that
may touch oracle/ — as the thing timed beside the product, never as the product.  Split out of bench.py in round 6."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from balancedgroupsoftmax_amd import capi  # noqa: E402,F401
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402,F401
from balancedgroupsoftmax_amd import gs_tables  # noqa: E402,F401
from bench_workloads import NUM_CLASSES  # noqa: E402,F401


def _cpu_time_threads(fn, n, seconds, cores):
    """median time of fn() per thread count; best of 1 / 8 / 32 / min(cores, 64)."""
    best, tried = None, {}
    counts = [nt for nt in sorted(set([1, 8, 32, min(cores, 64)])) if nt <= cores]
    for nt in counts:
        torch.set_num_threads(nt)
        for _ in range(3):
            fn(0)
        times = []
        t_end = time.perf_counter() + seconds / max(len(counts), 1)
        while time.perf_counter() < t_end and len(times) < 500:
            t0 = time.perf_counter()
            fn(len(times))
            times.append(time.perf_counter() - t0)
        med = float(np.median(times))
        tried[str(nt)] = round(med * 1e6 / n, 4)
        if best is None or med < best[0]:
            best = (med, nt, len(times))
    return best, tried


def cpu_baseline(n, seconds):
    """The reference's CPU path on THIS host's cores: the reference's own
    ``GSBBoxHeadWith0.loss()`` + ``backward()`` (gs_bbox_head_with0.py:147-186), imported from the
    head closure that oracle/build_ref.py stages under oracle/_ref/ (``kind: "reference"``); when
    that is absent, the torch-CPU port of it (oracle/gs_torch_port.py, ``kind: "port"``)."""
    import tempfile
    from oracle import build_ref, gs_oracle, gs_torch_port
    counts = gs_tables.synthetic_instance_counts(NUM_CLASSES, seed=0)
    l2b, ps, _ = gs_tables.build_group_tables(counts)
    batch = gs_oracle.make_roi_batch(n, int(ps[:, 1].sum()), NUM_CLASSES, seed=0)
    z, lab = torch.from_numpy(batch['logits']), torch.from_numpy(batch['labels'])
    l2b_t, ps_t = torch.from_numpy(l2b), torch.from_numpy(ps)
    cores = os.cpu_count() or 1

    def port(i):
        np.random.seed(i)
        gs_torch_port.gs_loss_fwd_bwd(z, lab, l2b_t, ps_t, 8.0)

    out = None
    root = build_ref.reference_python_root()
    if root is not None:
        try:
            from oracle import ref_import
            ref_import.install_stubs(root=root)
            tmp = tempfile.mkdtemp(prefix='bgs_ref_tables_')
            gs_tables.save_group_tables(tmp, *gs_tables.synthetic_group_tables())
            head = ref_import.build_reference_head(tmp)
            zr = z.clone().requires_grad_(True)

            def ref(i):
                np.random.seed(i)
                zr.grad = None
                losses = head.loss(zr, None, lab, None, None, None)
                sum(losses.values()).backward()

            (med, nt, cnt), tried = _cpu_time_threads(ref, n, seconds * 0.7, cores)
            out = dict(value=round(med * 1e6 / n, 4), unit='us/RoI', cores=nt, kind='reference',
                       host_cores=cores, threads_tried=tried,
                       sample='%d x (loss()+backward()) of the reference class GSBBoxHeadWith0 itself '
                              '(mmdet/models/bbox_heads/gs_bbox_head_with0.py, imported from %s under the '
                              'dependency stubs of oracle/ref_import.py) on N=%d RoIs x 1236 logits (cls '
                              'branch, numpy sampling incl.), median; best of 1/8/32/64 threads = %d; '
                              'torch %s' % (cnt, 'the reference tree' if root == build_ref.REF else
                                            'oracle/_ref/reference_py (staged by oracle/build_ref.py)',
                                            n, nt, torch.__version__))
            seconds *= 0.3
        except Exception as e:  # pragma: no cover
            sys.stderr.write('reference-class cpu_baseline failed (%r); timing the port\n' % (e,))
            out = None
    (med, nt, cnt), tried = _cpu_time_threads(port, n, seconds, cores)
    pd = dict(value=round(med * 1e6 / n, 4), unit='us/RoI', cores=nt, kind='port',
              host_cores=cores, threads_tried=tried,
              sample='%d x (loss+backward) of the torch-CPU port of GSBBoxHeadWith0.loss '
                     '(oracle/gs_torch_port.py) on N=%d RoIs x 1236 logits, median; best thread count = %d'
                     % (cnt, n, nt))
    if out is None:
        return pd
    out['port'] = pd
    return out


def cpu_baseline_detector(live=True):
    """The EXECUTED reference detector's whole training iteration on CPU (cfg[1]: the same shapes, GT count and
    sampler sizes as this bench, ``selectp=1``).  LIVE on this host's cores when the reference closure staged by
    oracle/build_ref.py travels with the tree (``oracle/_ref/reference_py`` + the host-built ops of oracle/_ref):
    tools/ref_cpu_detector_time.py in a child process (the import stubs patch ``torch.Tensor.cuda``: kept out of
    this process), one warm-up + one timed iteration per thread count, best of 16 / 64 threads — a bounded sample
    (~25 s).  The record measured in the authoring container (8 cores) is quoted next to it, or alone when the
    live leg cannot run."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, 'profiles', 'r2v_reference_detector_cpu_time.jsonl')
    out = None
    try:
        recs = [json.loads(ln) for ln in open(path) if ln.strip().startswith('{')]
        out = dict(kind='reference', where='authoring container (not this host)', records=recs,
                   source='profiles/r2v_reference_detector_cpu_time.jsonl')
    except Exception:  # pragma: no cover
        out = None
    if not live:
        return out
    try:
        from oracle import build_ref
        root = build_ref.reference_python_root()
        if root is None:
            return out
        cores = os.cpu_count() or 1
        best = None
        tried = []
        for nt in sorted({min(16, cores), min(64, cores)}):
            env = dict(os.environ, BGS_REFERENCE_ROOT=root, CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES='',
                       OMP_NUM_THREADS=str(nt))
            r = subprocess.run([sys.executable, os.path.join(here, 'tools', 'ref_cpu_detector_time.py'), '--iters', '1',
                                '--selectp', '1', '--threads', str(nt)], env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.DEVNULL, timeout=240)
            lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith('{')]
            if r.returncode != 0 or not lines:
                continue
            rec = json.loads(lines[-1])
            tried.append((nt, rec['s_per_iter']))
            if best is None or rec['s_per_iter'] < best['s_per_iter']:
                best = rec
        if best is None:
            return out
        live_rec = dict(kind='reference', where='this host', value=best['img_per_s'], unit='img/s',
                        s_per_iter=best['s_per_iter'], cores=best['threads'], host_cores=cores,
                        threads_tried=tried,
                        sample='1 training iteration (2 x 3x800x1344, 20 GT / image, shipped samplers, selectp=1: '
                               'forward + losses + backward) of the executed reference detector after one warm-up '
                               'iteration, per thread count; %s; ops = the reference\'s nms_cpu.cpp / RoIAlign kernels '
                               'built for the host (oracle/_ref); torch %s'
                               % ('the reference tree' if root == build_ref.REF else
                                  'oracle/_ref/reference_py (staged by oracle/build_ref.py)', best['torch']))
        if out is not None:
            live_rec['authoring_container_record'] = out
        return live_rec
    except Exception as e:  # pragma: no cover
        sys.stderr.write('live cpu_baseline_detector failed (%r); quoting the committed record\n' % (e,))
        return out
