import sys, os, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
st = bench.DetectorStep(dev, 0, 1, 2, selectp=0)
st.compute()
torch.cuda.synchronize()
gc.collect()
live = [o for o in gc.get_objects() if isinstance(o, torch.Tensor) and o.grad_fn is not None]
print('live non-leaf tensors with grad_fn after compute():', len(live))
for t in live[:20]:
    refs = [type(r).__name__ for r in gc.get_referrers(t)][:6]
    print(tuple(t.shape), type(t.grad_fn).__name__, refs)
    for r in gc.get_referrers(t):
        if isinstance(r, dict):
            ks = [k for k, v in r.items() if v is t]
            print('   dict keys:', ks[:5], [k for k in list(r.keys())[:8]])
        elif isinstance(r, (list, tuple)):
            print('   seq len', len(r), [type(x).__name__ for x in gc.get_referrers(r)][:4])
