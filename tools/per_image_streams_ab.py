"""Feasibility: the frozen trunk (backbone + FPN + RPN head) of the cfg[1] step on the 2-image batch in one stream vs
one image per stream on two real streams (eager launches; inner side-stream forks off in both arms)."""
import os, sys, time
os.environ['BGS_LEVEL_FORK'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda:0')
step = bench.DetectorStep(dev, 0, 1, 2, 1)
m = step.model
img = step.img
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def trunk(x):
    with torch.no_grad():
        f = m.extract_feat(x)
        return m.rpn_head(f)

def batch():
    trunk(img)

def per_image():
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(main)
    done = []
    for s, i in ((s1, 0), (s2, 1)):
        s.wait_event(ev)
        with torch.cuda.stream(s):
            trunk(img[i:i + 1])
            e = torch.cuda.Event(); e.record(s); done.append(e)
    for e in done:
        main.wait_event(e)

def wall(fn, reps=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

for rnd in range(3):
    print('batch of 2, one stream: %.3f ms | one image per stream: %.3f ms' % (wall(batch), wall(per_image)), flush=True)
os.environ['BGS_LEVEL_FORK'] = '1'
print('batch of 2 with the side-stream forks: %.3f ms' % wall(batch))
