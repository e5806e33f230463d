"""RoIAlign forward on the operands of a real cfg[1] iteration (bench.capture_head_inputs): the tap-grid kernel (every
distinct pixel of a bin loaded once; default) against the sample-at-a-time kernel (BGS_ROI_DEDUP=0), interleaved, hipEvent.
Also counts how many of the 16 taps per bin are distinct on these RoIs.   python tools/roi_dedup_ab.py [iters=200]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
argv = sys.argv[1:]
sys.argv = sys.argv[:1]
import numpy as np
import torch
import bench
from balancedgroupsoftmax_amd import functional as BF

iters = int(argv[0]) if argv else 200
dev = torch.device('cuda', 0)
cap = bench.capture_head_inputs(dev)
feats, rois = cap['feats'], cap['rois']


def run():
    return BF.roi_align_nhwc(feats, rois, cap['strides'], cap['out_size'], cap['sample_num'], cap['finest_scale'])


def timed():
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


res = {'0': [], '1': []}
outs = {}
for rnd in range(4):
    for m in (('0', '1') if rnd % 2 == 0 else ('1', '0')):
        os.environ['BGS_ROI_DEDUP'] = m
        outs[m] = run()
        res[m].append(timed())
os.environ.pop('BGS_ROI_DEDUP')
assert torch.equal(outs['0'], outs['1'])
print('K = %d RoIs x 49 bins x %d channels; identical bits: True' % (rois.shape[0], feats[0].shape[-1]))
for m, name in (('0', 'sample-at-a-time'), ('1', 'tap grid')):
    v = sorted(res[m])
    print('%-18s us / launch: min %.1f  median %.1f  all %s' % (name, v[0], v[len(v) // 2], ' '.join('%.1f' % x for x in res[m])))
# backward (selectp=0): one atomic per tap against one per distinct pixel
dout = torch.randn_like(outs['1'])
dfe = {m: [torch.zeros_like(x) for x in feats] for m in ('0', '1')}


def timed_bwd(m):
    os.environ['BGS_ROI_DEDUP'] = m
    fn = lambda: BF.roi_align_nhwc_bwd(dout, rois, dfe[m], cap['strides'], cap['sample_num'], cap['finest_scale'])
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 50 * 1e3


resb = {'0': [], '1': []}
for rnd in range(4):
    for m in (('0', '1') if rnd % 2 == 0 else ('1', '0')):
        resb[m].append(timed_bwd(m))
os.environ.pop('BGS_ROI_DEDUP')
for m in ('0', '1'):
    for x in dfe[m]:
        x.zero_()
    os.environ['BGS_ROI_DEDUP'] = m
    BF.roi_align_nhwc_bwd(dout, rois, dfe[m], cap['strides'], cap['sample_num'], cap['finest_scale'])
os.environ.pop('BGS_ROI_DEDUP')
err = max(float((a - b).abs().max()) for a, b in zip(dfe['0'], dfe['1']))
ref = max(float(a.abs().max()) for a in dfe['0'])
for m, name in (('0', 'bwd, atomic per tap'), ('1', 'bwd, per distinct px')):
    v = sorted(resb[m])
    print('%-22s us / launch: min %.1f  median %.1f  all %s' % (name, v[0], v[len(v) // 2], ' '.join('%.1f' % x for x in resb[m])))
print('backward: max |difference| between the two %.3g (max |gradient| %.3g)' % (err, ref))
# distinct taps per bin on these RoIs (host restatement of the row / column equality)
r = rois.cpu().numpy().astype(np.float64)
strides = np.asarray(cap['strides'], dtype=np.float64)
scale = np.sqrt((r[:, 3] - r[:, 1] + 1) * (r[:, 4] - r[:, 2] + 1))
lvl = np.clip(np.floor(np.log2(scale / cap['finest_scale'] + 1e-6)), 0, len(strides) - 1).astype(int)
cnt = []
for k in range(r.shape[0]):
    ss = 1.0 / strides[lvl[k]]
    H, W = feats[lvl[k]].shape[1:3]
    x0, y0, x1, y1 = r[k, 1] * ss, r[k, 2] * ss, (r[k, 3] + 1) * ss, (r[k, 4] + 1) * ss
    bw, bh = max(x1 - x0, 0) / 7, max(y1 - y0, 0) / 7
    for axis, (s0, b, n) in enumerate(((y0, bh, H), (x0, bw, W))):
        pass
    def axis_sets(s0, b, n):
        out = []
        for p in range(7):
            idx = set()
            for i in range(2):
                v = s0 + p * b + (i + .5) * b / 2
                if not (-1 <= v <= n):
                    idx.add(0); continue
                v = max(v, 0); lo = int(v)
                if lo >= n - 1: lo = hi = n - 1
                else: hi = lo + 1
                idx.add(lo); idx.add(hi)
            out.append(len(idx))
        return out
    ry, cx = axis_sets(y0, bh, H), axis_sets(x0, bw, W)
    cnt += [a * b for a in ry for b in cx]
cnt = np.asarray(cnt)
print('distinct pixels per bin (of 16 taps): mean %.2f  | histogram %s' % (cnt.mean(), dict(zip(*np.unique(cnt, return_counts=True)))))
