#!/usr/bin/env python
"""Times ONE training iteration of the EXECUTED REFERENCE detector on the host CPU, at the
BASELINE configuration (cfg[1]: gs_faster_rcnn_r50_fpn_1x_lvis_with0_bg8, 2 x 3x800x1344,
20 GT/img, 512 RoI/img, selectp=1 as shipped and selectp=0).  Authoring container only (needs
/root/reference); the reference's ops are its own sources built for the host (oracle/build_ref.py:
nms_cpu.cpp, the RoIAlign kernels of roi_align_kernel.cu as host code) — BASELINE.md §3 item 6.

    python tools/ref_cpu_detector_time.py [--iters 2] [--threads N]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=2)
    ap.add_argument('--threads', type=int, default=os.cpu_count())
    ap.add_argument('--selectp', type=int, default=1)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    from balancedgroupsoftmax_amd.config import to_config_dict
    from bench import detector_cfg
    from tests.golden import make_golden_train as T
    from tests.golden import make_golden_e2e as E
    T._bind_reference_ops(forbid_draws=False)      # the reference's own ops; numpy samplers as shipped
    from mmdet.models import build_detector
    tmp = tempfile.mkdtemp(prefix='bgs_refcpu_')
    model_cfg, train_cfg = detector_cfg(tmp)
    model = build_detector(to_config_dict(model_cfg), train_cfg=to_config_dict(train_cfg),
                           test_cfg=to_config_dict(E.TEST_CFG))
    model.train()
    if a.selectp == 1:
        for n, p in model.named_parameters():
            p.requires_grad = n.startswith('bbox_head.fc_cls')
    g = torch.Generator().manual_seed(1000)
    H, W, imgs = 800, 1344, 2
    img = torch.randn(imgs, 3, H, W, generator=g)
    metas = [dict(img_shape=(800, 1333, 3), pad_shape=(H, W, 3), ori_shape=(800, 1333, 3),
                  scale_factor=1.0, flip=False) for _ in range(imgs)]
    gtb, gtl = [], []
    for _ in range(imgs):
        wh = torch.exp(torch.rand(20, 2, generator=g) * (np.log(400) - np.log(16)) + np.log(16))
        xy = torch.rand(20, 2, generator=g) * (torch.tensor([1333., 800.]) - wh).clamp(min=1)
        gtb.append(torch.cat([xy, xy + wh], 1))
        gtl.append(torch.randint(1, 1231, (20,), generator=g))
    times = []
    for it in range(a.iters + 1):
        np.random.seed(it)
        t0 = time.perf_counter()
        losses = model.forward_train(img, metas, gtb, gtl)
        total = sum(sum(v) if isinstance(v, list) else v for k, v in losses.items() if 'loss' in k)
        model.zero_grad()
        total.backward()
        dt = time.perf_counter() - t0
        if it:
            times.append(dt)
        print('iter %d: %.2f s  loss %.4f' % (it, dt, float(total)), flush=True)
    print(json.dumps({'what': 'executed reference GroupSoftmax detector, one training iteration on CPU',
                      'selectp': a.selectp, 'threads': a.threads, 'cpu_count': os.cpu_count(),
                      's_per_iter': round(float(np.median(times)), 2),
                      'img_per_s': round(imgs / float(np.median(times)), 4),
                      'torch': torch.__version__}))


if __name__ == '__main__':
    main()
