"""Interleaved A/B of the whole cfg[1] training step (eager launches on the real side streams) between arms given as
    name:KEY=val,KEY=val;name2:...
KEY = an environment variable that the library reads at every call (BGS_LEVEL_FORK, BGS_ROI_XCD, ...), or
HALO_WIDE = 0/1/2 (bgs_conv3x3_halo_bfx_tuning: the wide pixel tile of the halo kernel).
python tools/step_ab.py "v4:HALO_WIDE=0;wide:HALO_WIDE=1" [rounds=5] [steps=20] [selectp=1] [extra bench flags...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
argv = sys.argv[1:]
sys.argv = sys.argv[:1]
import torch
import bench
from balancedgroupsoftmax_amd import functional as BF

spec = argv[0] if argv else 'v4:HALO_WIDE=0;wide:HALO_WIDE=1'
rounds = int(argv[1]) if len(argv) > 1 else 5
steps = int(argv[2]) if len(argv) > 2 else 20
selectp = int(argv[3]) if len(argv) > 3 else 1
flags = argv[4:]
arms = []
for part in spec.split(';'):
    name, _, kv = part.partition(':')
    arms.append((name, [tuple(x.split('=')) for x in kv.split(',') if x]))
keys = sorted({k for _, kvs in arms for k, _ in kvs} - {'MAIN_PRIO', 'PIPE'})


def apply(kvs):
    d = dict(kvs)
    if 'SPLITK' in keys or 'HALO_WIDE' in keys:
        # SPLITK=1: no K slicing anywhere (conv ring / wide kernels and the halo kernel's channel-chunk split)
        sk = int(d.get('SPLITK', -1))
        BF.conv_bfx_tuning(0, sk, halo_splits=(1 if sk == 1 else -1), halo_wide=int(d.get('HALO_WIDE', -1)))
    for k in keys:
        if k in ('HALO_WIDE', 'SPLITK'):
            continue
        if k == 'HALO_WIDE':
            BF.conv_bfx_tuning(halo_wide=int(d.get(k, 1)))
        elif k in d:
            os.environ[k] = d[k]
        else:
            os.environ.pop(k, None)


dev = torch.device('cuda', 0)
step = bench.DetectorStep(dev, 0, 1, 2, selectp, mask='--mask' in flags, cascade='--cascade' in flags,
                          htc='--htc' in flags, conv_math='bf16' if '--bf16' in flags else 'bf16x6')
res = {n: [] for n, _ in arms}
hi = torch.cuda.Stream(device=dev, priority=-1)          # MAIN_PRIO=1: the whole step on a high-priority stream (side lanes normal)


def run(kvs, k, w):
    d = dict(kvs)

    def body():
        if d.get('PIPE') == '2':            # prototype: three stages — backbone(i+2) | neck(i+1) | heads(i)
            m = step.model
            st = {}

            def launch_backbone():
                with torch.no_grad():
                    with BF.forked(dev, lane=3) as fk:
                        c = m.backbone(step.img)
                fk.hold(step.img)
                return fk, c

            def launch_neck(fkc):
                fk1, c = fkc
                with torch.no_grad():
                    with BF.forked(dev, lane=4) as fk:
                        fk1.join()                      # (inside the block: lane 4 waits for the backbone's stream)
                        x = m.neck(c)
                fk.hold(c)
                return fk, x

            st['b'] = launch_backbone()
            st['n'] = launch_neck(st['b'])
            st['b'] = launch_backbone()

            def fn():
                fk, x = st['n']
                fk.join()
                st['n'] = launch_neck(st['b'])
                st['b'] = launch_backbone()
                BF._PIPELINE_ACTIVE[0] = 1
                step.compute(x)
                step.apply()
                BF._PIPELINE_ACTIVE[0] = 0

            t = bench.timed_loop(fn, k, w, 1)
            st['n'][0].join(); st['b'][0].join()
            torch.cuda.synchronize()
            return t
        if d.get('PIPE') in ('1', '3', '4', '5', '6'):            # train.TrunkPipeline: the frozen trunk of the batches ahead beside this batch's heads
            fn = step.pipelined(depth={'1': 2, '3': 3, '4': 4, '5': 5, '6': 6}[d['PIPE']])
            t = bench.timed_loop(fn, k, w, 1)
            fn.drain()
            torch.cuda.synchronize()
            return t
        return bench.timed_loop(step, k, w, 1)

    if d.get('MAIN_PRIO') == '1':
        hi.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(hi):
            t = body()
        torch.cuda.current_stream().wait_stream(hi)
        return t
    return body()


for n, kvs in arms:
    apply(kvs)
    run(kvs, 4, 4)
for r in range(rounds):
    for n, kvs in (arms if r % 2 == 0 else arms[::-1]):
        apply(kvs)
        res[n].append(run(kvs, steps, 3) * 1e3 / steps)
BF.conv_bfx_tuning()
for n, _ in arms:
    v = res[n]
    print('%-16s ms/step: min %.3f  median %.3f  all %s' % (n, min(v), sorted(v)[len(v) // 2], ' '.join('%.3f' % x for x in v)), flush=True)
