"""CPU, world_size = 2 over gloo: the gradient exchange step of the data-parallel path
(mmdet/core/utils/dist_utils.py:9-41 semantics: flat SUM all-reduce, then / world_size) and the
optimizer-step hook order."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from balancedgroupsoftmax_amd import train


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.manual_seed(0)                              # identical parameters on both ranks
        fc = torch.nn.Linear(8, 5)
        other = torch.nn.Linear(3, 2)
        for p in other.parameters():
            p.requires_grad = False                       # frozen params are not exchanged
        params = list(fc.parameters())
        opt = train.build_optimizer(params, dict(type='SGD', lr=0.1, momentum=0.9,
                                                 weight_decay=0.0001))
        step = train.DistOptimizerStep(params, opt, dict(max_norm=35, norm_type=2), world_size=world)
        g = torch.Generator().manual_seed(100 + rank)     # rank-local data
        x = torch.randn(16, 8, generator=g)
        loss = fc(x).pow(2).mean()
        step(loss)
        torch.save(dict(grad=[p.grad.clone() for p in params], w=[p.detach().clone() for p in params]),
                   os.path.join(out_dir, 'rank%d.pt' % rank))
        # asynchronous form returns a finisher
        for p in params:
            p.grad = torch.full_like(p, float(rank + 1))
        fin = train.allreduce_grads(params, world, async_op=True)
        fin()
        assert all(torch.allclose(p.grad, torch.full_like(p, 1.5)) for p in params)
    finally:
        dist.destroy_process_group()


def test_allreduce_grads_mean_of_rank_grads(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(str(tmp_path), 'rank0.pt'))
    r1 = torch.load(os.path.join(str(tmp_path), 'rank1.pt'))
    # every rank ends with the same (averaged) gradient and the same updated weights
    for a, b in zip(r0['grad'], r1['grad']):
        assert torch.allclose(a, b, atol=1e-7)
    for a, b in zip(r0['w'], r1['w']):
        assert torch.allclose(a, b, atol=1e-7)
    # and that gradient is the mean of the two single-rank gradients
    torch.manual_seed(0)
    fc = torch.nn.Linear(8, 5)
    singles = []
    for rank in range(world):
        g = torch.Generator().manual_seed(100 + rank)
        x = torch.randn(16, 8, generator=g)
        fc.zero_grad()
        fc(x).pow(2).mean().backward()
        singles.append([p.grad.clone() for p in fc.parameters()])
    for i, a in enumerate(r0['grad']):
        assert torch.allclose(a, (singles[0][i] + singles[1][i]) / 2, atol=1e-6)


def test_single_process_is_noop():
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    assert train.allreduce_grads([p], 1) is None and torch.equal(p.grad, torch.full((3,), 2.0))


def _overlap_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(12, 32), torch.nn.ReLU(), torch.nn.Linear(32, 32),
                                  torch.nn.ReLU(), torch.nn.Linear(32, 4))
        unused = torch.nn.Linear(3, 3)                      # never receives a gradient
        params = list(net.parameters()) + list(unused.parameters())
        opt = train.build_optimizer(params, dict(type='SGD', lr=0.1, momentum=0.9, weight_decay=1e-4))
        # tiny buckets: 5 of them, launched from the backward hooks while backward still runs
        step = train.DistOptimizerStep(params, opt, dict(max_norm=35, norm_type=2), world_size=world,
                                       overlap=True, bucket_bytes=256)
        assert step.overlap is not None and len(step.overlap.buckets) >= 3
        g = torch.Generator().manual_seed(200 + rank)
        for it in range(2):                                  # two iterations: state resets
            x = torch.randn(16, 12, generator=g)
            step(net(x).pow(2).mean())
        torch.save(dict(w=[p.detach().clone() for p in params]),
                   os.path.join(out_dir, 'ov_rank%d.pt' % rank))
    finally:
        dist.destroy_process_group()


def _flat_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(12, 32), torch.nn.ReLU(), torch.nn.Linear(32, 32),
                                  torch.nn.ReLU(), torch.nn.Linear(32, 4))
        params = list(net.parameters())
        opt = train.build_optimizer(params, dict(type='SGD', lr=0.1, momentum=0.9, weight_decay=1e-4))
        step = train.DistOptimizerStep(params, opt, dict(max_norm=35, norm_type=2), world_size=world,
                                       overlap=False)
        g = torch.Generator().manual_seed(200 + rank)
        for it in range(2):
            x = torch.randn(16, 12, generator=g)
            step(net(x).pow(2).mean())
        torch.save(dict(w=[p.detach().clone() for p in params]),
                   os.path.join(out_dir, 'flat_rank%d.pt' % rank))
    finally:
        dist.destroy_process_group()


def test_overlapped_bucketed_exchange_equals_flat_allreduce(tmp_path):
    """The backward-overlapped bucketed exchange ends in the same weights as the reference-style
    flat all-reduce after backward, on every rank, incl. parameters that get no gradient."""
    world = 2
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_overlap_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    mp.spawn(_flat_worker, args=(world, port + 1, str(tmp_path)), nprocs=world, join=True)
    ov = [torch.load(os.path.join(str(tmp_path), 'ov_rank%d.pt' % r)) for r in range(world)]
    fl = [torch.load(os.path.join(str(tmp_path), 'flat_rank%d.pt' % r)) for r in range(world)]
    n = len(fl[0]['w'])
    for a, b in zip(ov[0]['w'], ov[1]['w']):
        assert torch.allclose(a, b, atol=1e-7)
    for a, b in zip(ov[0]['w'][:n], fl[0]['w']):
        assert torch.allclose(a, b, atol=1e-6)


def _one_rank_worker(rank, world, port):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        fc = torch.nn.Linear(4, 3)
        params = list(fc.parameters())
        for p in params:
            p.grad = torch.full_like(p, 2.0)
        calls = []
        real = dist.all_reduce

        def spy(t, *a, **k):
            calls.append(t.numel())
            return real(t, *a, **k)

        dist.all_reduce = spy
        try:
            assert train.allreduce_grads(params, 1) is None and calls == []       # default: no collective
            prev = train.exchange_at_world_size_one(True)
            assert prev is False
            train.allreduce_grads(params, 1)
            assert calls == [sum(p.numel() for p in params)]                      # ONE flat all-reduce
            assert all(torch.equal(p.grad, torch.full_like(p, 2.0)) for p in params)   # sum / 1
        finally:
            dist.all_reduce = real
            train.exchange_at_world_size_one(False)
    finally:
        dist.destroy_process_group()


def test_exchange_at_world_size_one_hook_runs_the_collective():
    """The test hook behind ``BGS_BENCH_SELF_GROUP`` (bench.py --dist-graph on a single-GPU box): with it
    the gradient exchange issues its flat all-reduce even in a 1-rank group, without it it does not."""
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_one_rank_worker, args=(1, port), nprocs=1, join=True)


def _ws8_net():
    torch.manual_seed(0)
    # 12*20+20 + 20*20+20 + 20*7+7 + 7*3+3 floats: with 600-byte buckets the LAST bucket (the first layer's
    # tensors, produced last by backward) is smaller than the others
    net = torch.nn.Sequential(torch.nn.Linear(12, 20), torch.nn.Tanh(), torch.nn.Linear(20, 20), torch.nn.Tanh(),
                              torch.nn.Linear(20, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3))
    unused = torch.nn.Linear(5, 5)                          # trainable but never in the graph: no gradient
    return net, unused


def _ws8_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        net, unused = _ws8_net()
        params = list(net.parameters()) + list(unused.parameters())
        ex = train.OverlappedGradExchange(params, world, bucket_bytes=600)
        sizes = [sum(p.numel() * 4 for p in b) for b in ex.buckets]
        assert len(ex.buckets) >= 3 and sizes[-1] < max(sizes)
        assert any(id(p) in ex._bucket_of and len(ex.buckets[ex._bucket_of[id(p)]]) > 2 for p in unused.parameters())
        g = torch.Generator().manual_seed(300 + rank)
        for it in range(2):                                  # the per-iteration state resets
            for p in params:
                p.grad = None
            x = torch.randn(6, 12, generator=g)
            net(x).pow(2).mean().backward()                  # hooks launch the bucket all-reduces
            ex.finish()
        assert all(p.grad is None for p in unused.parameters())
        if rank in (0, world - 1):
            torch.save([p.grad.clone() for p in net.parameters()], os.path.join(out_dir, 'ws8_rank%d.pt' % rank))
        ex.remove()
    finally:
        dist.destroy_process_group()


def test_overlapped_exchange_world_size_8_uneven_last_bucket_and_a_parameter_without_gradient(tmp_path):
    """Eight ranks over gloo (the node size the data-parallel path is specified for, dist_utils.py:9-58): the
    bucketed backward-overlapped exchange leaves on every rank the MEAN over the eight rank-local gradients,
    with a short last bucket and with trainable parameters that receive no gradient (they are skipped, their
    bucket still completes)."""
    world = 8
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_ws8_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    first = torch.load(os.path.join(str(tmp_path), 'ws8_rank0.pt'))
    last = torch.load(os.path.join(str(tmp_path), 'ws8_rank%d.pt' % (world - 1)))
    net, _ = _ws8_net()
    mean = [torch.zeros_like(p) for p in net.parameters()]
    for rank in range(world):
        g = torch.Generator().manual_seed(300 + rank)
        for it in range(2):
            x = torch.randn(6, 12, generator=g)
        net.zero_grad()
        net(x).pow(2).mean().backward()
        for m, p in zip(mean, net.parameters()):
            m += p.grad / world
    for a, b, m in zip(first, last, mean):
        assert torch.equal(a, b)                            # the same bits on every rank
        assert torch.allclose(a, m, atol=1e-6, rtol=1e-5)


def _uneven_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        net = torch.nn.Linear(6, 4)
        extra = torch.nn.Linear(4, 4)                       # in the graph of rank 0 only
        unused = torch.nn.Linear(3, 3)                      # in no rank's graph
        params = list(net.parameters()) + list(extra.parameters()) + list(unused.parameters())
        ex = train.OverlappedGradExchange(params, world, bucket_bytes=1 << 20)   # ONE bucket: never completed by hooks
        assert len(ex.buckets) == 1
        x = torch.randn(5, 6, generator=torch.Generator().manual_seed(40 + rank))
        for it in range(2):
            for p in params:
                p.grad = None
            y = net(x)
            if rank == 0:
                y = extra(y)
            y.pow(2).mean().backward()
            ex.finish()
        assert all(p.grad is None for p in unused.parameters())           # nobody contributed: dropped, as in the reference
        assert all(p.grad is not None for p in extra.parameters())        # rank 0 contributed: EVERY rank keeps the mean
        torch.save([p.grad.clone() for p in list(net.parameters()) + list(extra.parameters())],
                   os.path.join(out_dir, 'uneven_rank%d.pt' % rank))
        ex.remove()
    finally:
        dist.destroy_process_group()


def test_overlapped_exchange_keeps_ranks_consistent_when_only_one_rank_has_a_gradient(tmp_path):
    """a parameter with a gradient on rank 0 and none on rank 1.  The zero placeholder of rank 1 takes
    part in the SUM; afterwards BOTH ranks must hold the same averaged gradient (rank 1 used to reset its copy to
    ``None`` and the replicas drifted apart), while a parameter without a gradient on every rank is ``None`` again
    on every rank (dist_utils.py:33-38 leaves it out; SGD skips it)."""
    world = 2
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_uneven_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a = torch.load(os.path.join(str(tmp_path), 'uneven_rank0.pt'))
    b = torch.load(os.path.join(str(tmp_path), 'uneven_rank1.pt'))
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    # expected: mean over ranks, rank 1 contributing zero to `extra`
    torch.manual_seed(0)
    net = torch.nn.Linear(6, 4)
    extra = torch.nn.Linear(4, 4)
    mean = [torch.zeros_like(p) for p in list(net.parameters()) + list(extra.parameters())]
    for rank in range(world):
        net.zero_grad()
        extra.zero_grad()
        x = torch.randn(5, 6, generator=torch.Generator().manual_seed(40 + rank))
        y = net(x)
        if rank == 0:
            y = extra(y)
        y.pow(2).mean().backward()
        for m, p in zip(mean, list(net.parameters()) + list(extra.parameters())):
            if p.grad is not None:
                m += p.grad / world
    for x, m in zip(a, mean):
        assert torch.allclose(x, m, atol=1e-6, rtol=1e-5)
