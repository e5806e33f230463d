"""Timing-only (round 6): what the halo kernel's P2-level launch pays for BESIDES the loop that tools/mfma_lds_overlap.hip
times in isolation (411 ns per workgroup step against 648 in the launch): arms of BGS_HALO_ABL (bit 0 patch loaded once,
bit 1 patch split / stored once, bit 2 no epilogue) x the filter-DMA flags of tools/halo_dma_ablate.py.  One child process
per BGS_HALO_ABL value (read once).  Results are WRONG in every arm but the first.
    python -m balancedgroupsoftmax_amd.csrc.build --variant ablate; BGS_LIB_VARIANT=ablate python tools/halo_ablate2.py"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def child():
    import torch
    from balancedgroupsoftmax_amd import functional as BF
    from conv_sweep import bench
    dev = 'cuda:0'
    H, W, C = 200, 336, 256
    x = torch.randn(2, H, W, C, device=dev); w = torch.randn(C, 3, 3, C, device=dev) * 0.05; b = torch.randn(C, device=dev)
    f = lambda: BF.conv2d_nhwc(x, w, b, pad=1, relu=True)
    row = []
    for fl, nm in ((0, 'filter DMA'), (8, 'no filter DMA')):
        BF.conv_bfx_tuning(halo_flags=fl)
        row.append('%s %.4f ms' % (nm, min(bench(f, iters=20), bench(f, iters=20))))
    BF.conv_bfx_tuning()
    print(' | '.join(row))


if len(sys.argv) > 1 and sys.argv[1] == '--child':
    child()
else:
    names = {0: 'full kernel', 1: 'patch loaded once', 2: 'patch stored once', 3: 'no patch path', 4: 'no epilogue',
             7: 'no patch path, no epilogue'}
    for abl in (0, 1, 2, 3, 4, 7, 0):
        env = dict(os.environ, BGS_HALO_ABL=str(abl), BGS_LIB_VARIANT='ablate')
        out = subprocess.run([sys.executable, os.path.abspath(__file__), '--child'], env=env, capture_output=True, text=True, timeout=300)
        print('%-28s %s' % (names[abl], out.stdout.strip().splitlines()[-1] if out.stdout.strip() else 'failed ' + out.stderr[-300:]), flush=True)
