import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
st = bench.DetectorStep(dev, 0, 1, 2, selectp=0)
names = [n for n, p in st.model.named_parameters() if p.requires_grad]
params = [p for n, p in st.model.named_parameters() if p.requires_grad]
def norms():
    torch.cuda.synchronize()
    return torch.stack([p.grad.norm() for p in params]).cpu()
eg = []
for i in range(3):
    st.compute()
    eg.append(norms())
g = bench.try_graph(st.compute)
gg = []
for i in range(3):
    g.replay()
    gg.append(norms())
e = torch.stack(eg).mean(0)
r = torch.stack(gg).mean(0)
ratio = (r / e.clamp(min=1e-12))
idx = torch.argsort((ratio.log().abs()), descending=True)
print('eager grad-norm spread (step to step):', float((eg[0] / eg[1].clamp(min=1e-12)).log().abs().max()))
for i in idx[:25].tolist():
    print('%-45s eager %.4e graph %.4e ratio %.3f' % (names[i], e[i], r[i], ratio[i]))
print('nan in graph grads:', int(torch.isnan(torch.stack(gg)).sum()), 'median ratio', float(ratio.median()))
print('loss eager/graph', float(st.last['loss']))
