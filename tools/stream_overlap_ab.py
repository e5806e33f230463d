"""Do independent small-grid convs overlap when issued on forked streams — eagerly and inside a hipGraph?
The five pyramid levels of an FPN output conv (3x3, 256 -> 256, 2 images) are independent; P4..P6 launch 168 / 48 / 16
workgroups on 256 CUs.  Arms: one stream; P2 on the main stream and P3..P6 on a side stream; one stream per level.
Each arm eagerly and as a captured graph (wall clock of 50 back-to-back repetitions)."""
import sys, time
import torch
sys.path.insert(0, ".")
from balancedgroupsoftmax_amd import functional as BF

dev = torch.device("cuda:0")
torch.manual_seed(0)
sizes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
xs = [torch.randn(2, h, w, 256, device=dev) for h, w in sizes]
wt = torch.randn(256, 3, 3, 256, device=dev) * 0.02
b = torch.zeros(256, device=dev)
outs = [torch.empty(2, h, w, 256, device=dev) for h, w in sizes]
side = [torch.cuda.Stream(device=dev) for _ in range(4)]


def conv(i):
    BF.conv2d_nhwc(xs[i], wt, b, stride=1, pad=1, out=outs[i])


def serial():
    for i in range(5):
        conv(i)


def fork(groups):
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(main)
    joins = []
    for s, idx in zip(side, groups[1:]):
        s.wait_event(ev)
        with torch.cuda.stream(s):
            for i in idx:
                conv(i)
            e = torch.cuda.Event()
            e.record(s)
            joins.append(e)
    for i in groups[0]:
        conv(i)
    for e in joins:
        main.wait_event(e)


arms = {
    "one stream": serial,
    "P2 | P3..P6": lambda: fork([[0], [1, 2, 3, 4]]),
    "P2 | P3 | P4..P6": lambda: fork([[0], [1], [2, 3, 4]]),
    "one stream per level": lambda: fork([[0], [1], [2], [3], [4]]),
    "P3..P6 only, one stream": lambda: [conv(i) for i in (1, 2, 3, 4)],
}

# two latency-bound chains (20 dependent one-workgroup kernels each) + one big conv: do chains overlap each other /
# the conv inside a replayed graph?
ta = torch.zeros(256, device=dev)
tb = torch.zeros(256, device=dev)


def chain(t, n=20):
    for _ in range(n):
        t.add_(1.0)


def chains_serial():
    chain(ta); chain(tb)


def chains_forked():
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(main)
    side[0].wait_event(ev)
    with torch.cuda.stream(side[0]):
        chain(tb)
        e = torch.cuda.Event(); e.record(side[0])
    chain(ta)
    main.wait_event(e)


def chain_and_conv_serial():
    conv(1); chain(ta, 40)


def chain_and_conv_forked():
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(main)
    side[0].wait_event(ev)
    with torch.cuda.stream(side[0]):
        chain(ta, 40)
        e = torch.cuda.Event(); e.record(side[0])
    conv(1)
    main.wait_event(e)


arms.update({"2 chains x 20, one stream": chains_serial, "2 chains x 20, forked": chains_forked,
             "conv P3 + chain x 40, one stream": chain_and_conv_serial,
             "conv P3 + chain x 40, forked": chain_and_conv_forked})


def wall(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


ref = None
for name, fn in arms.items():
    with torch.cuda.stream(torch.cuda.Stream(device=dev)):
        e = wall(fn)
        chk = [o.clone() for o in outs]
        BF.reset_workspaces()
        g = torch.cuda.CUDAGraph()
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            fn()
        gt = wall(g.replay)
    if ref is None:
        ref = chk
    same = all(torch.equal(a, c) for a, c in zip(chk, ref)) if (name.startswith("P2") or name.startswith("one")) else "-"
    print("%-28s eager %7.1f us | graph %7.1f us | equal %s" % (name, e, gt, same), flush=True)
