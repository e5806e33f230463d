"""Multi-process GPU tests of the data-parallel path (mmdet/core/utils/dist_utils.py:9-58, mmdet/apis/train.py:143-205).

Two modes per test:
  * ``rccl``  — one process per GPU over RCCL (backend "nccl"), the real configuration: runs when the box has at least two
                GPUs and SKIPS cleanly otherwise (the round-end boxes so far had one);
  * ``gloo1`` — the same workers, both ranks on cuda:0 over gloo (device tensors through gloo): exercises every line of the
                test on a 1-GPU box, so the rccl arm does not meet an unexercised worker on its first multi-GPU node.
"""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from balancedgroupsoftmax_amd import train

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = ['rccl', 'gloo1']


def _need(mode, world=2):
    if mode == 'rccl' and torch.cuda.device_count() < world:
        pytest.skip('needs %d GPUs (this box has %d)' % (world, torch.cuda.device_count()))


def _init(rank, world, port, mode):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import datetime
    dev = torch.device('cuda', rank if mode == 'rccl' else 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl' if mode == 'rccl' else 'gloo', rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=120))
    return dev


def _net(dev):
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(12, 32), torch.nn.ReLU(), torch.nn.Linear(32, 32), torch.nn.ReLU(),
                              torch.nn.Linear(32, 4)).to(dev)
    return net


def _flat_worker(rank, world, port, mode, out_dir):
    dev = _init(rank, world, port, mode)
    try:
        net = _net(dev)
        params = list(net.parameters())
        x = torch.randn(16, 12, generator=torch.Generator().manual_seed(100 + rank)).to(dev)
        net(x).pow(2).mean().backward()
        local = [p.grad.clone() for p in params]
        train.allreduce_grads(params, world)
        torch.cuda.synchronize()
        torch.save(dict(local=[g.cpu() for g in local], got=[p.grad.cpu() for p in params]),
                   os.path.join(out_dir, 'flat%d.pt' % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('mode', MODES)
def test_allreduce_grads_on_device_is_the_mean_of_the_rank_gradients(mode, tmp_path):
    """the exchanged gradient == the mean of the single-rank gradients on the same per-rank inputs, the same
    bits on both ranks (flat SUM all-reduce / world, dist_utils.py:22-28) — device tensors, RCCL when two GPUs exist."""
    _need(mode)
    world = 2
    mp.spawn(_flat_worker, args=(world, 39000 + os.getpid() % 1500, mode, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), 'flat%d.pt' % k)) for k in range(world)]
    for i in range(len(r[0]['got'])):
        assert torch.equal(r[0]['got'][i], r[1]['got'][i])
        mean = (r[0]['local'][i] + r[1]['local'][i]) / world
        assert torch.allclose(r[0]['got'][i], mean, rtol=1e-6, atol=1e-8)
        assert not torch.equal(r[0]['local'][i], r[1]['local'][i])          # the ranks really saw different data


def _overlap_worker(rank, world, port, mode, out_dir, overlap):
    dev = _init(rank, world, port, mode)
    try:
        net = _net(dev)
        unused = torch.nn.Linear(3, 3).to(dev)                               # trainable, never in the graph
        params = list(net.parameters()) + list(unused.parameters())
        opt = train.build_optimizer(params, dict(type='SGD', lr=0.1, momentum=0.9, weight_decay=1e-4))
        step = train.DistOptimizerStep(params, opt, dict(max_norm=35, norm_type=2), world_size=world,
                                       overlap=overlap, bucket_bytes=256)
        if overlap:
            assert step.overlap is not None and len(step.overlap.buckets) >= 3
        g = torch.Generator().manual_seed(200 + rank)
        for it in range(3):
            x = torch.randn(16, 12, generator=g).to(dev)
            step(net(x).pow(2).mean())
        torch.cuda.synchronize()
        assert all(p.grad is None or not p.grad.abs().sum() > 0 for p in unused.parameters())
        torch.save([p.detach().cpu() for p in net.parameters()],
                   os.path.join(out_dir, '%s%d.pt' % ('ov' if overlap else 'fl', rank)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('mode', MODES)
def test_overlapped_bucketed_exchange_on_device_equals_the_flat_one(mode, tmp_path):
    """``OverlappedGradExchange`` (buckets launched from the backward hooks, asynchronous all-reduces on the
    collective's own stream) ends three optimizer steps with the weights of the flat exchange, identical on both ranks."""
    _need(mode)
    world = 2
    base = 40600 + os.getpid() % 1500
    mp.spawn(_overlap_worker, args=(world, base, mode, str(tmp_path), True), nprocs=world, join=True)
    mp.spawn(_overlap_worker, args=(world, base + 1, mode, str(tmp_path), False), nprocs=world, join=True)
    ov = [torch.load(os.path.join(str(tmp_path), 'ov%d.pt' % k)) for k in range(world)]
    fl = [torch.load(os.path.join(str(tmp_path), 'fl%d.pt' % k)) for k in range(world)]
    for i in range(len(ov[0])):
        assert torch.equal(ov[0][i], ov[1][i]) and torch.equal(fl[0][i], fl[1][i])
        assert torch.allclose(ov[0][i], fl[0][i], rtol=1e-6, atol=1e-7)


def _detector_worker(rank, world, port, mode, out_dir, steps, depth):
    dev = _init(rank, world, port, mode) if world > 1 else torch.device('cuda', 0)
    try:
        if world == 1:
            torch.cuda.set_device(dev)
        sys.path.insert(0, ROOT)
        os.environ.setdefault('BGS_LEVEL_FORK', '0')        # (two processes may share one device here; every arm is bit-identical)
        import bench
        # the SAME data on every rank (DetectorStep seeds its inputs by `rank`: rank 0's everywhere): the mean over ranks
        # of identical gradients is that gradient exactly, so the N-rank weights must equal the single-rank ones bit for bit
        step = bench.DetectorStep(dev, 0, world, 1, selectp=1)
        fn = step.pipelined(depth=depth) if depth else step
        for _ in range(steps):
            fn()
        if depth:
            fn.drain()
        torch.cuda.synchronize()
        torch.save(dict(w=[p.detach().cpu() for p in step.params],
                        last={k: float(v) for k, v in step.last.items()}),
                   os.path.join(out_dir, 'det_w%d_r%d.pt' % (world, rank)))
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.parametrize('mode', MODES)
def test_pipelined_training_steps_per_rank_are_bit_identical_to_single_rank(mode, tmp_path):
    """Three pipelined (train.TrunkPipeline, depth 3) cfg[1] training steps on each of two ranks that hold the same batch
    == the same three steps in a single process: losses of the last step and the trained ``fc_cls`` bit for bit on both
    ranks — the gradient exchange, the clip and the SGD step of the N-rank path change nothing but the averaging."""
    _need(mode)
    out = str(tmp_path)
    base = 42200 + os.getpid() % 1500
    mp.spawn(_detector_worker, args=(2, base, mode, out, 3, 3), nprocs=2, join=True)
    mp.spawn(_detector_worker, args=(1, base + 1, mode, out, 3, 3), nprocs=1, join=True)
    single = torch.load(os.path.join(out, 'det_w1_r0.pt'))
    for r in range(2):
        got = torch.load(os.path.join(out, 'det_w2_r%d.pt' % r))
        assert got['last'] == single['last'], (got['last'], single['last'])
        for a, b in zip(got['w'], single['w']):
            assert torch.equal(a, b)
