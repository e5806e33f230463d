"""Soak of train.TrunkPipeline at the bench's full size (cfg[1], 2 x 3 x 800 x 1344, shipped sampler sizes): n optimizer steps
over two alternating batches WITHOUT any host synchronisation inside the loop, sequential loop vs pipelined (depths 4, 5, 3):
the final fc_cls parameters and the last step's losses must be BIT-IDENTICAL (tests/test_gpu_e2e.py holds the same statement on
six steps with a host read-back per step; this one leaves the allocator and the streams to themselves).
python tools/pipe_soak.py [steps=200]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
argv = sys.argv[1:]
sys.argv = sys.argv[:1]
import torch
import bench
from balancedgroupsoftmax_amd import functional as BF

n = int(argv[0]) if argv else 200
dev = torch.device('cuda', 0)
step = bench.DetectorStep(dev, 0, 1, 2, 1, conv_math='bf16x6')
img_a = step.img
img_b = (torch.flip(img_a, dims=[3]) * 0.9 + 0.05).contiguous()
init = [p.detach().clone() for p in step.params]
mom = None


def reset():
    with torch.no_grad():
        for p, v in zip(step.params, init):
            p.copy_(v)
            p.grad = None
    for c in BF._KEY_COUNTERS.values():
        c.zero_()
    step.model.bbox_head._draw.zero_()
    for st in step.step_fn.optimizer.state.values():                     # momentum buffers of the previous run
        for k, v in st.items():
            if torch.is_tensor(v):
                v.zero_()


def run(depth):
    reset()
    torch.cuda.synchronize()
    imgs = [img_a if (i // 3) % 2 == 0 else img_b for i in range(n + 8)]
    if depth:
        pipe = step.train.TrunkPipeline(step.model, depth=depth)
        for k in range(pipe.depth - 1):
            pipe.push(imgs[k])
    for i in range(n):
        step.img = imgs[i]
        feats = None
        if depth:
            feats = pipe.take()
            pipe.push(imgs[i + pipe.depth - 1])
        step.compute(feats)
        step.apply()
    if depth:
        pipe.drain()
    torch.cuda.synchronize()
    step.img = img_a
    return [p.detach().clone() for p in step.params], {k: float(v) for k, v in step.last.items()}


w0, l0 = run(0)
print('sequential: %d steps, last loss %.6f, |fc_cls| %.6f' % (n, l0['loss'], float(w0[0].abs().sum())), flush=True)
ok = True
for d in (4, 5, 3, 5):
    w, l = run(d)
    same = all(torch.equal(a, b) for a, b in zip(w0, w)) and l == l0
    ok &= same
    print('pipelined depth %d: identical parameters and last losses: %s (last loss %.6f)' % (d, same, l['loss']), flush=True)
w1, l1 = run(0)
print('sequential again: identical: %s' % (all(torch.equal(a, b) for a, b in zip(w0, w1)) and l1 == l0))
print('SOAK', 'OK' if ok else 'FAILED')
sys.exit(0 if ok else 1)
