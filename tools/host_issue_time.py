"""How long does the host need to ISSUE one eager cfg[1] step (no synchronisation inside), against the GPU time of the
step?  Eager launching stays GPU-bound only while the first is clearly below the second."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda:0')
step = bench.DetectorStep(dev, 0, 1, 2, 1)
for _ in range(6):
    step()
torch.cuda.synchronize()
for rnd in range(3):
    n = 4
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('host issue %.2f ms / step | wall %.2f ms / step' % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(4):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(28)
