import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import functional as BF
from conv_sweep import L as LAYERS, FC, N as NIMG, bench
os.environ['BGS_CONV_HALO'] = '0'
dev='cuda:0'
layers = [(n, H, W, Cin, Cout, R, s, c) for (n, H, W, Cin, Cout, R, s, c) in LAYERS]
layers += [(n, 1, 1, K, Cout, 1, 1, 1) for (n, M, K, Cout) in FC]
tot={'n4':0,'n3':0,'auto':0}
for name, H, W, Cin, Cout, R, stride, cnt in layers:
    if R == 3 and stride == 1 and NIMG*H*W >= 2000: continue
    Nn = 1024 if name.startswith('fc') else NIMG
    pad = R // 2
    x = torch.randn(Nn, H, W, Cin, device=dev); w = torch.randn(Cout, R, R, Cin, device=dev) * 0.05; b = torch.randn(Cout, device=dev)
    r={}
    for key,t in (('n4',0x800),('n3',0x400),('auto',0)):
        BF.conv_bfx_tuning(t, -1)
        y = BF.conv2d_nhwc(x, w, b, stride=stride, pad=pad, relu=True)
        if key=='n4': y4=y
        else: assert torch.equal(y, y4), name
        r[key]=bench(lambda: BF.conv2d_nhwc(x, w, b, stride=stride, pad=pad, relu=True), iters=20)
        u=BF.conv_bfx_last_launch()
        tot[key]+=r[key]*cnt
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    wgs=((Nn*Ho*Wo+63)//64)*((Cout+63)//64)*u['splits']
    print('%-12s wgs %5d (x%d splits) | 4-stage %.4f | 3-stage %.4f | auto(%d) %.4f  x%d' % (name, wgs, u['splits'], r['n4'], r['n3'], u['ring_stages'], r['auto'], cnt), flush=True)
BF.conv_bfx_tuning()
print('total: 4-stage %.3f  3-stage %.3f  auto %.3f' % (tot['n4'], tot['n3'], tot['auto']))
