import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
st = bench.DetectorStep(dev, 0, 1, 2, selectp=1)
for _ in range(5):
    st()
torch.cuda.synchronize()
ts = []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print('host enqueue ms (median) %.2f, step wall ms (median) %.2f' % (sorted(t[0] for t in ts)[10] * 1e3, sorted(t[1] for t in ts)[10] * 1e3))
