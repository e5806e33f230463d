import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from balancedgroupsoftmax_amd import functional as BF
torch.manual_seed(0)
for (N, H, W) in ((2, 128, 192), (2, 800, 1344)):
    img = torch.randn(N, 3, H, W)
    w = torch.randn(64, 3, 7, 7) * (2.0 / 147) ** 0.5
    b = torch.randn(64) * 0.3
    exp = F.max_pool2d(F.relu(F.conv2d(img.double(), w.double(), b.double(), stride=2, padding=3)), 3, 2, 1).permute(0, 2, 3, 1)
    cpu32 = F.max_pool2d(F.relu(F.conv2d(img, w, b, stride=2, padding=3)), 3, 2, 1).permute(0, 2, 3, 1).double()
    wk = torch.nn.functional.pad(w.permute(0, 2, 3, 1), (0, 1)).contiguous().cuda()
    x = img.cuda(); bb = b.cuda()
    a = BF.stem_fused(x, BF.stem_fused_split_weights(wk), bb).cpu().double()
    c = BF.maxpool3x3s2_nhwc(BF.conv2d_nhwc(BF.nchw_to_nhwc4(x), wk, bb, stride=2, pad=3, relu=True)).cpu().double()
    BF.set_conv_math('f32')
    f = BF.maxpool3x3s2_nhwc(BF.conv2d_nhwc(BF.nchw_to_nhwc4(x), wk, bb, stride=2, pad=3, relu=True)).cpu().double()
    BF.set_conv_math('bf16x6')
    sc = float(exp.abs().max())
    for name, t in (('fused', a), ('chain bf16x6', c), ('chain f32 MFMA', f), ('torch CPU fp32', cpu32)):
        d = (t - exp).abs()
        print('%dx%dx%d %-16s max %.2e rms %.2e (of scale %.2f)' % (N, H, W, name, float(d.max()) / sc, float(d.pow(2).mean().sqrt()) / sc, sc))
