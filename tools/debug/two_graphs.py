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
