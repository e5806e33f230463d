"""The operand split's residual by v_dot2c_f32_bf16 (csrc/bfx_split.h) against the subtract form.
   python tools/split_ab.py planes   -> the planes bgs_conv_bfx_split_weights writes for classes of special inputs,
                                        compared with the numpy restatement (tests/ carries the same check)
   python tools/split_ab.py layers   -> every conv shape of one cfg[1] forward, default dispatch, ms per launch
Run each under the default library and under BGS_LIB_PATH=.../libbgs_splitsub.so (tools/split_ab.sh)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from balancedgroupsoftmax_amd import functional as BF, capi
from conv_sweep import L as LAYERS, N as NIMG, FC, bench

dev = 'cuda:0'


def bf16_rne(x):
    """fp32 array -> (uint16 bf16 bits, the bf16 value as fp32), round to nearest even (v_cvt_pk_bf16_f32)."""
    b = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((b + 0x7fff + ((b >> 16) & 1)) >> 16).astype(np.uint32)
    nan = np.isnan(x)
    r = np.where(nan, (b >> 16) | 0x40, r).astype(np.uint16)
    return r, (r.astype(np.uint32) << 16).view(np.float32)


def split3_np(x):
    h, hf = bf16_rne(x)
    with np.errstate(invalid='ignore', over='ignore'):
        r = (x - hf).astype(np.float32)
        m, mf = bf16_rne(r)
        r2 = (r - mf).astype(np.float32)
    l, _ = bf16_rne(r2)
    return h, m, l


def planes_gpu(x):
    K = x.size
    buf = BF.bfx_split_weights(torch.from_numpy(x).to(dev).view(1, K), cache=False)
    torch.cuda.synchronize()
    KC = 2 * ((K + 31) // 32)
    raw = buf.cpu().numpy()[:3 * KC * 16 * 2].view(np.uint16).reshape(3, KC * 16)
    return raw[0, :K], raw[1, :K], raw[2, :K]


def planes():
    g = np.random.default_rng(0)
    n = 1 << 16
    cls = {
        'normal, exponents -30..30': (g.standard_normal(n) * np.exp2(g.integers(-30, 31, n))).astype(np.float32),
        'activations (relu of normal)': np.maximum(g.standard_normal(n), 0).astype(np.float32),
        'ties and powers of two': np.concatenate([np.exp2(g.integers(-20, 20, n // 2)) * (1 + np.exp2(-8.0)),
                                                  np.exp2(g.integers(-40, 40, n // 2))]).astype(np.float32),
        'large (1e30..3e38)': (g.uniform(1, 3, n) * np.exp2(g.integers(100, 127, n))).astype(np.float32),
        'tiny normal (1e-37..1e-30)': (g.uniform(1, 2, n) * np.exp2(g.integers(-125, -100, n))).astype(np.float32),
        'fp32 subnormal': (g.uniform(0, 1, n) * 1.1e-38).astype(np.float32),
    }
    for name, x in cls.items():
        x = x * np.where(g.integers(0, 2, x.size) > 0, 1, -1).astype(np.float32)
        h, m, l = planes_gpu(x)
        H, M, Lo = split3_np(x)
        print('%-32s hi mismatches %6d  mid %6d  lo %6d  (of %d; mid flushed to zero where the restatement is not: %d, lo: %d)' % (
            name, int((h != H).sum()), int((m != M).sum()), int((l != Lo).sum()), x.size,
            int(((m & 0x7fff) == 0)[m != M].sum()), int(((l & 0x7fff) == 0)[l != Lo].sum())), flush=True)
    print('library:', capi.lib_path())


def layers():
    BF.set_conv_math('bf16x6')
    tot = 0.0
    for name, H, W, Cin, Cout, R, stride, cnt in LAYERS:
        if name == 'stem7x7':
            continue
        pad = R // 2
        x = torch.randn(NIMG, H, W, Cin, device=dev); w = torch.randn(Cout, R, R, Cin, device=dev) * 0.05; b = torch.randn(Cout, device=dev)
        ms = min(bench(lambda: BF.conv2d_nhwc(x, w, b, stride=stride, pad=pad, relu=True), iters=20) for _ in range(3))
        tot += ms * cnt
        print('%-12s %.4f ms x%d' % (name, ms, cnt), flush=True)
    for name, M, K, Cout in FC:
        x = torch.randn(M, 1, 1, K, device=dev); w = torch.randn(Cout, 1, 1, K, device=dev) * 0.02; b = torch.randn(Cout, device=dev)
        ms = min(bench(lambda: BF.conv2d_nhwc(x, w, b), iters=20) for _ in range(3))
        tot += ms
        print('%-12s %.4f ms' % (name, ms), flush=True)
    print('TOTAL per forward %.4f ms   library: %s' % (tot, capi.lib_path()))


if __name__ == '__main__':
    {'planes': planes, 'layers': layers}[sys.argv[1]]()
