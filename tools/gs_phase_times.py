"""Phase times of the fused GroupSoftmax head kernel from in-kernel s_memtime marks
(bgs_gs_head_debug_timestamps): per workgroup, averaged over the grid.  The marks live in the ONE-row-per-
workgroup kernels (bgs_gs_head_variant 0 = flag words, 1 = bit planes; `python tools/gs_phase_times.py [variant]`,
default 1); the multi-row kernel the library picks by default for N <= 2048 carries none."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from balancedgroupsoftmax_amd import capi, functional as BF
from bench import make_inputs, NUM_CLASSES
lib = capi.load()
lib.bgs_gs_head_variant(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
dev = torch.device('cuda:0')
n = 1024
inp = make_inputs(n, 1000, dev)
buf = torch.zeros(2048 * 8, dtype=torch.int64, device=dev)
z = inp['logits'].clone().requires_grad_(True)
def run():
    t, tot, _ = BF.gs_head_step(z, inp['labels'], inp['l2b'], inp['ps_np'], 8.0, 1, bbox_pred=inp['bbox_pred'],
                                bbox_targets=inp['bbox_targets'], bbox_weights=inp['bbox_weights'],
                                num_reg_classes=NUM_CLASSES)
for _ in range(5):
    run()
torch.cuda.synchronize()
lib.bgs_gs_head_debug_timestamps(capi.ptr(buf))
run()
torch.cuda.synchronize()
lib.bgs_gs_head_debug_timestamps(None)
t = buf.cpu().numpy().reshape(2048, 8)[:n].astype(np.float64)
t0 = t[:, 0].min()
names = ['start', 'loads landed + LDS written', 'barrier 1', 'flags + ballots', 'barrier 2', 'bins done (wave 0)', 'barrier 3', 'end']
print('marks relative to the first workgroup start (ticks of s_memtime), mean / min / max over %d workgroups' % n)
for i, nm in enumerate(names):
    d = t[:, i] - t0
    print('%-28s %9.0f %9.0f %9.0f   (+%.0f since previous mark)' % (nm, d.mean(), d.min(), d.max(), (t[:, i] - t[:, max(i - 1, 0)]).mean()))
print('workgroup lifetime mean %.0f ticks; kernel span %.0f ticks' % ((t[:, 7] - t[:, 0]).mean(), t[:, 7].max() - t0))
