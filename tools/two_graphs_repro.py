import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
mode = sys.argv[1]
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)

def run(selectp, graph, steps, tag):
    st = bench.DetectorStep(dev, 0, 1, 2, selectp=selectp)
    g = bench.try_graph(st.compute) if graph else None
    for i in range(steps):
        if g is not None:
            g.replay()
            st.apply()
        else:
            st()
    torch.cuda.synchronize()
    print(tag, 'selectp', selectp, 'graph', g is not None, 'loss', float(st.last['loss']), flush=True)
    return st, g

if mode == 'sp0_long':
    run(0, True, 40, 'A')
elif mode == 'sp1_then_sp0':
    a = run(1, True, 10, 'A')
    b = run(0, True, 10, 'B')
elif mode == 'sp1_del_then_sp0':
    a = run(1, True, 10, 'A')
    del a
    torch.cuda.empty_cache()
    b = run(0, True, 10, 'B')
elif mode == 'sp1eager_then_sp0':
    a = run(1, False, 10, 'A')
    b = run(0, True, 10, 'B')
elif mode == 'sp0_then_sp0':
    a = run(0, True, 10, 'A')
    b = run(0, True, 10, 'B')
elif mode == 'sp0_eager':
    run(0, False, 13, 'A')
elif mode == "sp0_eager2":
    run(0, False, 2, 'A')
elif mode == 'eager_default_then_graph':
    st = bench.DetectorStep(dev, 0, 1, 2, selectp=0)
    for i in range(3):
        st.compute()
    torch.cuda.synchronize()
    print('eager done', float(st.last['loss']), flush=True)
    g = bench.try_graph(st.compute)
    for i in range(12):
        g.replay()
    torch.cuda.synchronize()
    print('graph replays done', float(st.last['loss']), flush=True)
elif mode == 'graph_only_norms':
    st = bench.DetectorStep(dev, 0, 1, 2, selectp=0)
    g = bench.try_graph(st.compute)
    for i in range(12):
        g.replay()
    torch.cuda.synchronize()
    print('graph replays done', float(st.last['loss']), flush=True)
elif mode == 'sp0_eager_long':
    st = bench.DetectorStep(dev, 0, 1, 2, selectp=0)
    for i in range(int(sys.argv[2])):
        st.compute()          # no optimizer step: weights fixed, only the sampling changes
        if i % 10 == 0:
            torch.cuda.synchronize()
            print(i, float(st.last['loss']), flush=True)
    torch.cuda.synchronize()
    print('done', flush=True)
elif mode == 'graph_replays':
    st = bench.DetectorStep(dev, 0, 1, 2, selectp=int(sys.argv[4]) if len(sys.argv) > 4 else 0)
    g = bench.try_graph(st.compute)
    n, do_apply = int(sys.argv[2]), sys.argv[3] == 'apply'
    for i in range(n):
        g.replay()
        if do_apply:
            st.apply()
        if i % 10 == 0:
            torch.cuda.synchronize()
            print(i, float(st.last['loss']), flush=True)
    torch.cuda.synchronize()
    print('done', flush=True)
elif mode == 'bisect':
    # graph_replays noapply with parts of the randomness frozen:  bisect <n> <what>
    what = sys.argv[3]
    from balancedgroupsoftmax_amd import assign as A
    from balancedgroupsoftmax_amd import functional as BF
    if 'keys' in what:
        _cache = {}
        def fixed_keys(n, device, generator=None):
            if n not in _cache:
                g = torch.Generator(device=device); g.manual_seed(n)
                _cache[n] = torch.randint(0, A._KEY_MAX, (n,), device=device, dtype=torch.int64, generator=g)
            return _cache[n]
        A._random_keys = fixed_keys
    st = bench.DetectorStep(dev, 0, 1, 2, selectp=1)
    if 'nodraw' in what:
        head = st.model.bbox_head
        orig = BF.gs_prepare
        def gp(labels, l2b, ratio, seed=None, cls_weight=None, seed_offset=None):
            return orig(labels, l2b, ratio, seed=1234, cls_weight=cls_weight, seed_offset=None)
        import balancedgroupsoftmax_amd.bbox_heads as BH
        BH.BF.gs_prepare = gp
    g = bench.try_graph(st.compute)
    for i in range(int(sys.argv[2])):
        g.replay()
        if i % 20 == 0:
            torch.cuda.synchronize()
            print(i, float(st.last['loss']), flush=True)
    torch.cuda.synchronize()
    print('done', what, flush=True)
elif mode == 'pool_check':
    from balancedgroupsoftmax_amd import functional as BF
    rec = {}
    orig_w = BF.conv2d_wgrad_nhwc
    def wg(*a, **k):
        out = orig_w(*a, **k)
        t = out[0] if isinstance(out, tuple) else out
        rec.setdefault('bwd', []).append((t.data_ptr(), torch.cuda.is_current_stream_capturing(), torch.cuda.current_stream().cuda_stream))
        return out
    BF.conv2d_wgrad_nhwc = wg
    orig_c = BF.conv2d_nhwc
    def cf(*a, **k):
        out = orig_c(*a, **k)
        rec.setdefault('fwd', []).append((out.data_ptr(), torch.cuda.is_current_stream_capturing(), torch.cuda.current_stream().cuda_stream))
        return out
    BF.conv2d_nhwc = cf
    st = bench.DetectorStep(dev, 0, 1, 2, selectp=1)
    g = bench.try_graph(st.compute)
    snap = torch.cuda.memory_snapshot()
    def pool_of(ptr):
        for seg in snap:
            if seg['address'] <= ptr < seg['address'] + seg['total_size']:
                return seg.get('segment_pool_id')
        return None
    for k in ('fwd', 'bwd'):
        caps = [r for r in rec[k] if r[1]]
        print(k, 'calls during capture:', len(caps), 'streams', set(r[2] for r in caps))
        print('   pools:', set(pool_of(r[0]) for r in caps))
    print('pools of persistent .grad:', pool_of(st.params[0].grad.data_ptr()))
elif mode == 'train_trace':
    use_graph = sys.argv[2] == 'graph'
    st = bench.DetectorStep(dev, 0, 1, 2, selectp=1)
    g = bench.try_graph(st.compute) if use_graph else None
    for i in range(int(sys.argv[3])):
        if g is not None:
            g.replay()
        else:
            st.compute()
        st.apply()
        if i >= 25 or i % 5 == 0:
            torch.cuda.synchronize()
            w = st.model.bbox_head.fc_cls.weight
            print(i, 'loss %.4f' % float(st.last['loss']), 'w absmax %.3e' % float(w.abs().max()),
                  'grad absmax %.3e' % float(w.grad.abs().max()), flush=True)
    print('done', flush=True)
elif mode == 'variants':
    v = sys.argv[2]
    st = bench.DetectorStep(dev, 0, 1, 2, selectp=1)
    if v == 'whole':
        g = bench.try_graph(st.__call__)
    else:
        g = bench.try_graph(st.compute)
    for i in range(int(sys.argv[3])):
        g.replay()
        if v == 'sync':
            torch.cuda.synchronize()
            st.apply()
        elif v == 'syncboth':
            torch.cuda.synchronize()
            st.apply()
            torch.cuda.synchronize()
        elif v == 'plain':
            st.apply()
        if i % 20 == 0:
            torch.cuda.synchronize()
            print(i, 'loss %.4f' % float(st.last['loss']), flush=True)
    torch.cuda.synchronize()
    print('done', v, flush=True)
elif mode == 'sidestream':
    st = bench.DetectorStep(dev, 0, 1, 2, selectp=1)
    g = bench.try_graph(st.compute)
    s2 = torch.cuda.Stream()
    cur = torch.cuda.current_stream()
    for i in range(int(sys.argv[2])):
        g.replay()
        s2.wait_stream(cur)
        with torch.cuda.stream(s2):
            st.apply()
        cur.wait_stream(s2)
        if i % 20 == 0:
            torch.cuda.synchronize()
            print(i, 'loss %.4f' % float(st.last['loss']), flush=True)
    torch.cuda.synchronize()
    print('done sidestream', flush=True)
elif mode == 'graphstream':
    # replay the graph on a dedicated stream, eager work stays on the default stream
    st = bench.DetectorStep(dev, 0, 1, 2, selectp=1)
    g = bench.try_graph(st.compute)
    s2 = torch.cuda.Stream()
    cur = torch.cuda.current_stream()
    for i in range(int(sys.argv[2])):
        s2.wait_stream(cur)
        with torch.cuda.stream(s2):
            g.replay()
        cur.wait_stream(s2)
        st.apply()
        if i % 20 == 0:
            torch.cuda.synchronize()
            print(i, 'loss %.4f' % float(st.last['loss']), flush=True)
    torch.cuda.synchronize()
    print('done graphstream', flush=True)
