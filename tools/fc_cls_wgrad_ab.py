"""fc_cls weight gradient (the one trainable layer of the shipped selectp=1 step: dW [1236 -> padded 1236, 1024] =
dy^T [1024 RoIs x 1236] x [1024 RoIs x 1024], on the critical path behind the GroupSoftmax head): fp32-MFMA kernel
against the bf16x6 kernel (bgs_conv2d_wgrad_bfx_enable(2): also reductions of <= 1536 rows) over the split counts.
python tools/fc_cls_wgrad_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import capi, functional as BF
from conv_sweep import bench
lib = capi.load()
dev = 'cuda:0'
BF.set_conv_math('bf16x6')
M, K, Nout = 1024, 1024, 1236
x = torch.randn(M, 1, 1, K, device=dev); dy = torch.randn(M, 1, 1, Nout, device=dev)
ref = (dy.view(M, Nout).double().t() @ x.view(M, K).double())
for mode, name in ((0, 'fp32 MFMA'), (2, 'bf16x6')):
    lib.bgs_conv2d_wgrad_bfx_enable(mode)
    for splits in (0, 1, 2, 3, 4, 6, 8, 12, 16):
        if splits:
            os.environ['BGS_WGRAD_SPLITS'] = str(splits)
        else:
            os.environ.pop('BGS_WGRAD_SPLITS', None)
        f = lambda: BF.conv2d_wgrad_nhwc(x, dy, 1, bias=True)
        dw, db = f()
        err = float((dw.view(Nout, K).double() - ref).abs().max() / ref.abs().max())
        t = min(bench(f, iters=30) for _ in range(2))
        print('%-10s splits %-4s %.2f us  (%.0f TF)  rel err %.2e' % (name, splits or 'auto', t * 1e3, 2.0 * M * K * Nout / t / 1e9, err), flush=True)
os.environ.pop('BGS_WGRAD_SPLITS', None)
lib.bgs_conv2d_wgrad_bfx_enable(1)
