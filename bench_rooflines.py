"""bench.py's per-kernel rooflines: HIP-event timings of the dominant conv kernel, the GroupSoftmax
kernels, RoIAlign, `_merge_score`, IoU / assignment, and the per-layer floor of the whole step.  Split out of bench.py in
round 6; the fields they produce are unchanged."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from balancedgroupsoftmax_amd import capi  # noqa: E402,F401
from balancedgroupsoftmax_amd import functional as BF  # noqa: E402,F401
from balancedgroupsoftmax_amd import gs_tables  # noqa: E402,F401
from bench_workloads import NUM_CLASSES, HBM_PEAK_GBS, DetectorStep, make_inputs, GsHeadStep, try_graph  # noqa: E402,F401
from bench_dist import timed_loop  # noqa: E402


def conv_roofline(dev, math, iters=20, wide=0, planes3=True):
    """Dominant kernel of the detector step: the 3x3 convolution of the largest pyramid level — since round 6 the 8 x 8-pixel
    planes kernel (csrc/conv3x3_planes.hip) under the default dispatch; ``planes3=False`` times the halo kernels it
    replaced (``wide``: their two-launch schedule) on the same layer for comparison.  Timed on the largest single
    layer (FPN output conv on P2: 2x200x336 pixels, 3x3, 256->256 = 158.5 algorithmic GFLOP) with
    HIP events on the launch stream.  `achieved` = ALGORITHMIC flops / time.  Peaks
    (MI355X_MICROARCH.md): fp32 matrix 157.3 TFLOP/s; bf16 matrix 2500 TFLOP/s dense — the bf16x6
    kernel spends SIX bf16 MFMA passes per algorithmic fp32 multiply-add, so its matrix-pipe
    ceiling in algorithmic flops is 2500 / 6 = 416.7 TFLOP/s (frac = matrix-pipe busy fraction)."""
    prev = BF.set_conv_math(math)
    prev_wide = BF.set_halo_wide(1) if wide else None
    lib = capi.load()
    if not planes3:
        lib.bgs_conv3x3_planes_enable(0)
    used = {}
    took_planes3 = 0
    try:
        x = torch.randn(2, 200, 336, 256, device=dev)
        w = torch.randn(256, 3, 3, 256, device=dev) * 0.02
        b = torch.randn(256, device=dev)
        out = torch.empty(2, 200, 336, 256, device=dev)
        # warm-up: the first ~10 launches after an idle period run 8-10 % slower (clock ramp: 0.81 vs
        # 0.74 ms on the P2 layer); inside the step the kernel runs warm (rocprofv3 average 0.743 ms,
        # profiles/r3k_detector_prof_summary.md), and that is the state a roofline should describe
        for _ in range(25):
            BF.conv2d_nhwc(x, w, b, pad=1, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            BF.conv2d_nhwc(x, w, b, pad=1, out=out)
        e1.record()
        torch.cuda.synchronize()
        used = BF.conv_bfx_last_launch() if math != 'f32' else {}
        took_planes3 = lib.bgs_conv3x3_planes_last_launch() if math == 'bf16x6' else 0
    finally:
        BF.set_conv_math(prev)
        if wide:
            BF.set_halo_wide(prev_wide)
        if not planes3:
            lib.bgs_conv3x3_planes_enable(-1)
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * 2 * 200 * 336 * 256 * 256 * 9
    tf = flops / (ms * 1e-3) / 1e12
    if math == 'bf16x6':
        kname = 'conv3x3_halo_bfx4_kernel<2>'
        kdesc = kname + (' (halo-resident A operand split to 3 bf16 planes in LDS, filter slices by '
                         'LDS-DMA, v_mfma_f32_32x32x16_bf16 x 6)')
        peak, passes = 2500.0 / 6.0, 6
        if took_planes3:
            kname = 'conv3x3_planes_bfx_kernel<%d>' % took_planes3
            kdesc = kname + (' (8 x 8 output pixels x %d channels per workgroup, the whole reduction in the workgroup; the '
                             '10 x 10 patch of a 32-channel chunk split once to 3 bf16 planes in LDS, one barrier per 18 k '
                             'steps; filter fragments by buffer loads from L2; v_mfma_f32_32x32x16_bf16 x 6)'
                             % (128 * took_planes3))
        elif wide and used.get('halo_wide_units'):
            # the two-launch schedule the trunk pipeline switches on (bgs_conv3x3_halo_bfx_wide(1)): whole rounds of
            # 16 x 16-pixel units (two workgroups per CU) + the left-over rows on the 8 x 16-pixel kernel
            kname = 'conv3x3_halo_bfx7_kernel<3>+conv3x3_halo_bfx4_kernel<2>'
            kdesc = ('conv3x3_halo_bfx7_kernel<3> on %d units of 16 x 16 pixels x 128 channels (two workgroups per CU, '
                     'whole rounds of 512) + conv3x3_halo_bfx4_kernel<2> on the %d left-over 8 x 16-pixel units: two '
                     'launches per layer, bit-identical to the one-launch form (tests/test_gpu_det_ops.py); '
                     'ms_per_launch is the PAIR' % (used['halo_wide_units'], used['halo_tail_units']))
    elif math == 'bf16':
        kname = 'conv3x3_halo_bfx3_kernel<2,1>'
        kdesc = kname + ' (operands rounded to bf16, v_mfma_f32_32x32x16_bf16 x 1)'
        peak, passes = 2500.0, 1
    else:
        halo = BF._use_halo_kernel(2 * 200 * 336, 256)
        kname = 'conv3x3_halo_f32_kernel' if halo else 'conv_igemm_f32_kernel<2,2,16,1>'
        kdesc = kname + ' (v_mfma_f32_32x32x2_f32)'
        peak, passes = 157.3, 1
    traffic = src = None
    ent = {}
    try:      # HBM-side bytes per launch from the committed PMC passes (separate rocprofv3 --pmc runs)
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles',
                               'pmc_traffic.json')) as f:
            ent = json.load(f)[kname]['fpn_p2_out_2x200x336_3x3_256_256']
        traffic, src = ent['traffic_bytes_per_launch'], ent['source']
        if ent.get('tree'):
            src = '%s; tree %s' % (src, ent['tree'])
    except Exception:
        pass
    r = dict(bound='mfma', achieved=round(tf, 2), peak=round(peak, 1), unit='TFLOP/s',
             frac=round(tf / peak, 4), traffic=traffic,
             traffic_source=('committed PMC measurement of this kernel on this layer, not collected '
                             'in this run: %s' % src) if src else None,
             algorithmic_bytes=277610496, kernel=kdesc, ms_per_launch=round(ms, 4),
             flops_per_launch=flops,
             layer='FPN output conv P2: N=2, 200x336, 3x3, 256->256 (M=134400, K=2304)',
             timing='hipEvent over %d back-to-back launches after 25 warm-up launches' % iters)
    if ent.get('matrix_pipe_busy'):
        # committed PMC pass of the same kernel on the same layer (not collected in this run): the
        # fraction of cycles the matrix pipe was busy, and the clock the chip sustained under it —
        # `frac` is priced against the 2.4 GHz data-sheet peak
        r.update(matrix_pipe_busy_pmc=ent['matrix_pipe_busy'],
                 effective_clock_ghz_pmc=ent['effective_clock_ghz'],
                 pmc_note='frac x 2.4 / %.1f ~ matrix_pipe_busy_pmc (%s)'
                          % (ent['effective_clock_ghz'], ent.get('counters', '')))
    if passes > 1:
        r.update(mfma_dtype='bf16', mfma_passes_per_flop=passes,
                 matrix_pipe_tflops=round(tf * passes, 1), peak_bf16_dense=2500.0,
                 peak_note='algorithmic-flop ceiling of the bf16x6 kernel = 2500 (bf16 dense MFMA) / 6 '
                           'passes; against the fp32-MFMA peak (157.3) the same launch is %.2fx'
                           % (tf / 157.3))
    return r


def _event_time_us(launch, iters, warm=20, settle=0):
    """HIP-event time per launch over `iters` back-to-back launches.  ``settle`` > 0: the batch is repeated (at most
    `settle` times) until two consecutive batches agree within 1 % and the last one is returned — a streaming
    kernel's first ~20 ms after an idle or compute-bound phase run 10 - 15 % slower (rowwave kernel at N = 65,536
    on fresh inputs: 141, 126, 123, 121, 121 us for five consecutive batches of 50 launches; the memory-side clocks
    ramp), and the roofline is a steady-state figure."""
    for _ in range(warm):
        launch()
    torch.cuda.synchronize()
    prev = None
    for _ in range(max(1, settle)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            launch()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        if prev is not None and abs(us - prev) <= 0.01 * prev:
            break
        prev = us
    return us


def _pmc_traffic(kernel, n):
    """HBM bytes per launch from the PMC counters: collected in separate rocprofv3 passes
    (tools/pmc_traffic.sh), corrected as the microarch guide prescribes, and committed under
    profiles/ — bench.py itself cannot run the profiler."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
            ent = json.load(f)[kernel][str(n)]
        src = ent.get('source')
        if src and ent.get('tree'):          # the tree the PMC pass ran on (commit that added its summary to profiles/)
            src = '%s; tree %s' % (src, ent['tree'])
        return ent['traffic_bytes_per_launch'], src
    except Exception:
        return None, None


def kernel_roofline(inp, n, iters=300, kernel='fused'):
    """HIP-event timing of ONE GroupSoftmax kernel alone, back to back on the current stream.
    ``kernel='fused'``: the kernel the detector step and the gs_head step actually launch for
    N <= 4096 (``gs_head_multi_kernel`` for N <= 2048 — 4 or 2 rows per workgroup behind one shared
    prologue —, ``gs_head_fused_kernel`` beyond; ``bgs_gs_head_variant_used``: main launch of
    bgs_gs_head_step with loss_out = NULL — label remap + "others" draw + loss + gradient + box branch).  ``kernel='rowwave'``: the plain loss
    kernel (main launch of bgs_gs_loss_fwd_bwd; the path for N > 4096 / reweighted heads).
    Algorithmic bytes per RoI: W*4 read + W*4 written + 8 (label) + B*4."""
    lib = capi.load()
    W, B = inp['W'], inp['ps_np'].shape[0]
    dev = inp['logits'].device
    ps_keep, ps_ptr = capi.host_i64(inp['ps_np'])
    dl = torch.empty_like(inp['logits'])
    ws = torch.empty(lib.bgs_gs_loss_workspace_bytes(n, B), dtype=torch.uint8, device=dev)
    st = capi.current_stream(dev)
    if kernel == 'fused':
        avg = torch.empty(B, dtype=torch.float32, device=dev)
        cbits = BF.gs_class_bin_mask(inp['l2b'])
        variant = lib.bgs_gs_head_variant_used(n)
        kname = {0: 'gs_head_fused_kernel<4,true,true,0>', 1: 'gs_head_fused_kernel<4,true,true,1>',
                 2: 'gs_head_multi_kernel<4,true,true,2>', 3: 'gs_head_multi_kernel<4,true,true,4>',
                 4: 'gs_head_multi_kernel<4,true,true,2,direct>',
                 5: 'gs_head_multi_kernel<4,true,true,4,direct>'}.get(variant, 'gs_head kernel variant %d' % variant)

        def launch():
            rc = lib.bgs_gs_head_step(capi.ptr(inp['logits']), capi.ptr(inp['labels']), capi.ptr(inp['l2b']),
                                      capi.ptr(cbits), None, ps_ptr, None, n, NUM_CLASSES, B, W, 8.0,
                                      12345, None, capi.ptr(inp['bbox_pred']),
                                      capi.ptr(inp['bbox_targets']), capi.ptr(inp['bbox_weights']),
                                      NUM_CLASSES, 1.0, 1.0, None, None, capi.ptr(dl), None,
                                      capi.ptr(avg), None, None, capi.ptr(ws), st)
            capi.check('bgs_gs_head_step', rc)
    else:
        bl, w, avg = BF.gs_prepare(inp['labels'], inp['l2b'], 8.0, seed=1)
        kname = 'gs_loss_rowwave_kernel<4,true>'

        def launch():
            rc = lib.bgs_gs_loss_fwd_bwd(capi.ptr(inp['logits']), capi.ptr(bl), ps_ptr, capi.ptr(w),
                                         capi.ptr(avg), n, B, W, None, capi.ptr(dl), capi.ptr(ws), st)
            capi.check('bgs_gs_loss_fwd_bwd', rc)

    us = _event_time_us(launch, iters, settle=8 if n >= 16384 else 0)
    bytes_per_roi = W * 4 + W * 4 + 8 + B * 4
    achieved = bytes_per_roi * n / (us * 1e-6) / 1e9
    traffic, src = _pmc_traffic(kname, n)
    return dict(bound='hbm', achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                traffic_source=('committed PMC measurement, not collected in this run: %s' % src) if src else None,
                kernel=kname, us_per_launch=round(us, 3),
                launched_by=('the detector step and the gs_head step (N <= 4096)' if kernel == 'fused'
                             else 'heads with N > 4096 rows or per-class reweighting (after gs_prepare)'),
                algorithmic_bytes_per_roi=bytes_per_roi, rois_per_launch=n,
                timing='hipEvent over %d back-to-back launches (includes the ~1.5 us '
                       'inter-kernel boundary)%s' % (iters, '; batches repeated until two agree within 1 % '
                                                     '(steady state, see _event_time_us)' if n >= 16384 else ''))


def capture_head_inputs(dev, conv_math='bf16x6'):
    """One eager cfg[1] iteration with spies on the RoI extractor and the RPN assigner: the REAL operands of the
    HBM-bound helper kernels (sampled RoIs + the NHWC pyramid; anchors, inside flags, gts) for their standalone
    rooflines and for tools/kernel_once.py (the PMC passes)."""
    step = DetectorStep(dev, 0, 1, 2, 1, conv_math=conv_math)
    cap = {}
    ext = step.model.bbox_roi_extractor
    orig = ext.forward

    def spy(feats, rois, *a, **k):
        cap['feats'] = [f.detach() for f in feats[:ext.num_inputs]]
        cap['rois'] = rois.detach().clone()
        cap['strides'] = list(ext.featmap_strides)
        cap['out_size'], cap['sample_num'], cap['finest_scale'] = ext.out_size, ext.sample_num, ext.finest_scale
        return orig(feats, rois, *a, **k)

    ext.forward = spy
    orig_assign = BF.iou_assign

    def spy_assign(boxes, gt_cat, offs, pos, neg, minpos=0.0, valid=None, shared_boxes=False, **k):
        if shared_boxes and 'anchors' not in cap:
            cap['anchors'], cap['gt_cat'], cap['gt_offs'] = boxes, gt_cat.clone(), list(offs)
            cap['assign_thr'] = (pos, neg, minpos)
            cap['inside'] = valid
        return orig_assign(boxes, gt_cat, offs, pos, neg, minpos, valid=valid, shared_boxes=shared_boxes, **k)

    BF.iou_assign = spy_assign
    try:
        os.environ['BGS_RPN_LOSS_FORK'] = '0'
        step()
        torch.cuda.synchronize()
    finally:
        BF.iou_assign = orig_assign
        ext.forward = orig
        os.environ.pop('BGS_RPN_LOSS_FORK', None)
    del step
    return cap


def roi_footprint_bytes(rois, shapes, strides, C, out_size=7, sample_num=2, finest_scale=56.0):
    """the unique input footprint of every RoI, exactly, from the RoIs themselves — the set of
    feature-map pixels its out x out x sample_num^2 bilinear sample points touch (level map single_level.py:69-72,
    sample geometry and the clamping / out-of-bounds rules of roi_align_kernel.cu:16-61,86-118), x C x 4 bytes.
    Returns (sum over RoIs of per-RoI footprints, bytes of the UNION over all RoIs, per-level RoI counts)."""
    r = rois.detach().cpu().numpy().astype(np.float32)
    L = len(strides)
    f32 = np.float32
    scale = np.sqrt((r[:, 3] - r[:, 1] + f32(1)) * (r[:, 4] - r[:, 2] + f32(1)))
    lvl = np.clip(np.floor(np.log2(scale / f32(finest_scale) + f32(1e-6))), 0, L - 1).astype(np.int64)
    per_roi = 0
    union = [dict() for _ in range(L)]
    g = (np.arange(out_size * sample_num, dtype=np.float32) + f32(0.5)) / f32(sample_num)   # sample offsets in bins
    for k in range(r.shape[0]):
        l = int(lvl[k])
        n = int(r[k, 0])
        H, W = shapes[l]
        ss = f32(1.0 / strides[l])
        x1, y1 = r[k, 1] * ss, r[k, 2] * ss
        rw = max((r[k, 3] + f32(1)) * ss - x1, f32(0))
        rh = max((r[k, 4] + f32(1)) * ss - y1, f32(0))
        ys = y1 + g * (rh / f32(out_size))
        xs = x1 + g * (rw / f32(out_size))

        def axis(v, S):
            ok = (v >= -1.0) & (v <= S)
            v = np.maximum(v, 0)
            lo = np.minimum(v.astype(np.int64), S - 1)
            hi = np.minimum(lo + 1, S - 1)
            return ok, lo, hi

        oky, ylo, yhi = axis(ys, H)
        okx, xlo, xhi = axis(xs, W)
        rows = np.unique(np.concatenate([ylo[oky], yhi[oky]]))
        cols = np.unique(np.concatenate([xlo[okx], xhi[okx]]))
        per_roi += rows.size * cols.size
        u = union[l].setdefault(n, np.zeros((H, W), dtype=bool))
        if rows.size and cols.size:
            u[np.ix_(rows, cols)] = True
    union_px = sum(int(m.sum()) for d in union for m in d.values())
    counts = [int((lvl == l).sum()) for l in range(L)]
    return per_roi * C * 4, union_px * C * 4, counts


def hbm_kernel_rooflines(dev, conv_math='bf16x6'):
    """SURVEY.md 8(d)'s other HBM-bound kernels on the operands of a real cfg[1] iteration: RoIAlign forward,
    `_merge_score` (R = 1000 as at test time, and R = 65,536 where the roofline applies), IoU / assignment of the
    RPN's 268,569 anchors x 2 images.  hipEvent time of back-to-back launches; `traffic` from the committed PMC
    passes (tools/pmc_hbm_kernels.sh -> profiles/pmc_traffic.json)."""
    res = {}
    cap = capture_head_inputs(dev, conv_math)
    feats, rois = cap['feats'], cap['rois']
    K, C = int(rois.shape[0]), int(feats[0].shape[3])
    shapes = [(int(f.shape[1]), int(f.shape[2])) for f in feats]
    out_bytes = K * cap['out_size'] ** 2 * C * 4
    pyramid = sum(int(f.numel()) * 4 for f in feats)
    fp_sum, fp_union, counts = roi_footprint_bytes(rois, shapes, cap['strides'], C, cap['out_size'],
                                                   cap['sample_num'], cap['finest_scale'])
    us = _event_time_us(lambda: BF.roi_align_nhwc(feats, rois, cap['strides'], cap['out_size'], cap['sample_num'],
                                                  cap['finest_scale']), 50, settle=4)
    alg = out_bytes + min(pyramid, fp_sum)
    kname = 'roi_align_fwd_grid_kernel<1,false>'      # (round 5: every distinct pixel of a bin loaded once)
    tr, src = _pmc_traffic(kname, K)
    res['roofline_roi_align'] = dict(
        bound='hbm', achieved=round(alg / (us * 1e-6) / 1e9, 1), peak=HBM_PEAK_GBS, unit='GB/s',
        frac=round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), traffic=tr,
        traffic_source=('committed PMC measurement, not collected in this run: %s' % src) if src else None,
        kernel=kname, us_per_launch=round(us, 2), rois_per_launch=K, rois_per_level=counts,
        algorithmic_bytes=alg, output_bytes=out_bytes, pyramid_bytes=pyramid,
        sum_of_per_roi_footprints=fp_sum, union_of_footprints=fp_union,
        bytes_issued_by_the_taps=K * cap['out_size'] ** 2 * cap['sample_num'] ** 2 * 4 * C * 4,
        frac_with_union_footprint=round((out_bytes + fp_union) / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
        note='SURVEY 8(d): bytes = output + min(pyramid, sum of per-RoI unique footprints), footprints computed '
             'exactly from the sampled RoIs of a real iteration (roi_footprint_bytes); `union_of_footprints` is '
             'what a perfect cache would fetch once')
    del feats, cap['feats']
    # _merge_score: read W*4 + write C*4 per RoI
    tdir = __import__('tempfile').mkdtemp(prefix='bgs_tables_')
    counts_t = gs_tables.synthetic_instance_counts(NUM_CLASSES, seed=0)
    l2b, ps, _ = gs_tables.build_group_tables(counts_t)
    c2c = gs_tables.class_to_column(l2b, ps).to(dev)
    W = int(ps[:, 1].sum())
    for R, iters in ((1000, 200), (65536, 30)):
        z = torch.randn(R, W, device=dev)
        us = _event_time_us(lambda: BF.gs_merge_score(z, ps, c2c, NUM_CLASSES), iters, settle=6 if R > 4096 else 0)
        alg = R * (W * 4 + NUM_CLASSES * 4)
        # (round 6: from 4096 rows the row-per-WAVE kernel — wave-private LDS row, 16-byte score stores — takes over)
        kname = 'gs_merge_wavepriv_kernel' if R >= 4096 else 'gs_merge_rowwave_kernel'
        tr, src = _pmc_traffic(kname, R)
        res['roofline_merge_score' + ('' if R == 1000 else '_n%d' % R)] = dict(
            bound='hbm', achieved=round(alg / (us * 1e-6) / 1e9, 1), peak=HBM_PEAK_GBS, unit='GB/s',
            frac=round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), traffic=tr,
            traffic_source=('committed PMC measurement, not collected in this run: %s' % src) if src else None,
            kernel=kname, us_per_launch=round(us, 2), rois_per_launch=R,
            algorithmic_bytes_per_roi=W * 4 + NUM_CLASSES * 4,
            note='includes the [R, 1231] output allocation of the wrapper (no launch)')
        del z
    # IoU + MaxIoUAssigner of the RPN: per image A anchors x (16 B box + 1 B inside flag) read, 4 B written
    if 'anchors' in cap:
        A = int(cap['anchors'].shape[0])
        N = len(cap['gt_offs']) - 1
        pos, neg, minpos = cap['assign_thr']
        us = _event_time_us(lambda: BF.iou_assign(cap['anchors'], cap['gt_cat'], cap['gt_offs'], pos, neg, minpos,
                                                  valid=cap['inside'], shared_boxes=True), 100)
        alg = N * A * (16 + 1 + 4) + int(cap['gt_cat'].numel()) * 4
        tr, src = _pmc_traffic('iou_gtmax_kernel+iou_assign_kernel', A)
        res['roofline_iou_assign'] = dict(
            bound='hbm', achieved=round(alg / (us * 1e-6) / 1e9, 1), peak=HBM_PEAK_GBS, unit='GB/s',
            frac=round(alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), traffic=tr,
            traffic_source=('committed PMC measurement, not collected in this run: %s' % src) if src else None,
            kernel='fill_i32_kernel + iou_gtmax_kernel + iou_assign_kernel (bgs_iou_assign: 3 launches)',
            us_per_call=round(us, 2), anchors=A, images=N, gts=int(cap['gt_cat'].shape[0]),
            algorithmic_bytes=alg,
            note='HBM-bound by class (SURVEY 8d) but 11 MB per call: three dependent launches of ~5 us each are '
                 'launch / latency bound, the figure to read is us_per_call')
    return res


def gs_head_metric(inp, n, steps=300, warmup=20):
    """The BASELINE metric's second half, 'GroupSoftmax us/RoI', as the whole head-loss step
    (GSBBoxHeadWith0.loss() + backward(): label remap, 'others' sampling, per-bin loss forward and
    backward, box loss, the sums) on a 1024-RoI batch resident in HBM — what `--workload gs_head`
    reports as its `value`, here as a field of the default line."""
    step = GsHeadStep(inp)
    graph = try_graph(step)
    fn = graph.replay if graph is not None else step
    dt = timed_loop(fn, steps, warmup, 1)
    # the same step under an arbitrary upstream gradient (the scaling launch of the autograd edge runs)
    step_g = GsHeadStep(inp, unit_root=False)
    graph_g = try_graph(step_g)
    dt_g = timed_loop(graph_g.replay if graph_g is not None else step_g, steps, warmup, 1)
    return dict(value=round(dt * 1e6 / (steps * n), 6), unit='us/RoI', us_per_step=round(dt * 1e6 / steps, 2),
                us_per_step_any_upstream=round(dt_g * 1e6 / steps, 2),
                rois_per_step=n, steps=steps,
                launch='hipGraph replay' if graph is not None else 'eager launches',
                what='bgs_gs_head_step + total.backward(unit_gradient): main kernel (label remap + others '
                     'sampling + per-bin loss fwd + bwd + box branch) + reduce (6 terms, total, draw counter); '
                     'the gradient the forward wrote is the answer when the root gradient is the library\'s '
                     'constant 1 (no launch on the autograd edge); us_per_step_any_upstream = the same step '
                     'under any other upstream gradient (one scaling launch more); launch-latency bound')


STEP_GFLOP = {
    # algorithmic flops of one 2-image step (~212 GMAC = 424 GFLOP forward per
    # image; selectp=1 adds dW_cls only, selectp=0 ~3x minus the frozen stem + layer1)
    1: 2 * 426.0, 0: 2 * 1200.0,
}


def step_layer_floor(imgs, conv_math):
    """Per-layer floor of the selectp = 1 cfg[1] step: sum over the conv / linear layers of
    max(flops / matrix-pipe peak, algorithmic bytes / 8 TB/s) — a K = 64 layer of ResNet layer1 is priced by the
    bytes it must move (input + output + residual + filter), not by its MFMAs.  Layer shapes: resnet.py:220-266,
    522-533 (stem 7x7 / s2 + max-pool, stages 3-4-6-3), fpn.py:101-141, rpn_head.py:30-35, convfc_bbox_head.py:
    132-168, at 800 x 1344.  Returns (floor_ms, mfma_part_ms, hbm_bound_ms, n_layers, hbm_bound_layers)."""
    peak = {'bf16x6': 2500.0 / 6.0, 'f32': 157.3, 'bf16': 2500.0}[conv_math] * 1e12
    hbm = HBM_PEAK_GBS * 1e9
    N = imgs
    layers = []     # (name, M = output pixels, K, Cout, input bytes, extra bytes (residual), count)

    def conv(name, H, W, Cin, Cout, R, stride, count=1, residual=False):
        Ho, Wo = H // stride, W // stride
        M = N * Ho * Wo
        inb = N * H * W * Cin * 4
        layers.append((name, M, R * R * Cin, Cout, inb, M * Cout * 4 if residual else 0, count))

    conv('stem', 800, 1344, 3, 64, 7, 2)            # (+ max-pool: its 34 MB output is what leaves)
    H1, W1 = 200, 336
    conv('l1.c1(64)', H1, W1, 64, 64, 1, 1)
    conv('l1.c1(256)', H1, W1, 256, 64, 1, 1, 2)
    conv('l1.c2', H1, W1, 64, 64, 3, 1, 3)
    conv('l1.c3', H1, W1, 64, 256, 1, 1, 3, residual=True)
    conv('l1.ds', H1, W1, 64, 256, 1, 1)
    for pl, (hi, wi), nb in ((128, (200, 336), 4), (256, (100, 168), 6), (512, (50, 84), 3)):
        ho, wo = hi // 2, wi // 2
        conv('c1', hi, wi, pl * 2, pl, 1, 1)
        conv('c2s2', hi, wi, pl, pl, 3, 2)
        conv('ds', hi, wi, pl * 2, pl * 4, 1, 2)
        conv('c1', ho, wo, pl * 4, pl, 1, 1, nb - 1)
        conv('c2', ho, wo, pl, pl, 3, 1, nb - 1)
        conv('c3', ho, wo, pl, pl * 4, 1, 1, nb, residual=True)
    for (h, w, c) in ((200, 336, 256), (100, 168, 512), (50, 84, 1024), (25, 42, 2048)):
        conv('fpn.lat', h, w, c, 256, 1, 1, residual=(h != 25))     # (top-down add fused in the lateral's epilogue)
        conv('fpn.out', h, w, 256, 256, 3, 1)
    for (h, w) in ((200, 336), (100, 168), (50, 84), (25, 42), (13, 21)):
        conv('rpn.conv', h, w, 256, 256, 3, 1)
        conv('rpn.head', h, w, 256, 15, 1, 1)
    R = 512 * N
    for name, K, Cout in (('fc1', 12544, 1024), ('fc2', 1024, 1024), ('fc_cls', 1024, 1236), ('fc_reg', 1024, 4924)):
        layers.append((name, R, K, Cout, R * K * 4, 0, 1))
    layers.append(('fc_cls.dW', 1236, R, 1024, R * (1236 + 1024) * 4, 0, 1))
    floor = mfma_ms = hbm_ms = 0.0
    nl = nh = 0
    for name, M, K, Cout, inb, extra, count in layers:
        flops = 2.0 * M * K * Cout
        outb = M * Cout * 4 if name != 'stem' else M * Cout          # (the stem's map is pooled 4:1 before it leaves)
        byts = inb + outb + extra + K * Cout * 4
        t_m, t_b = flops / peak, byts / hbm
        floor += count * max(t_m, t_b)
        mfma_ms += count * t_m
        nl += count
        if t_b > t_m:
            nh += count
            hbm_ms += count * t_b
    return floor * 1e3, mfma_ms * 1e3, hbm_ms * 1e3, nl, nh


def roofline_step(out, args):
    """The WHOLE step against the matrix-pipe ceiling (the line's `roofline` describes the best
    layer of the dominant kernel only): algorithmic GFLOP per step / ms_per_step / ceiling, plus the
    per-family kernel time of the last committed rocprofv3 trace (profiles/step_families.json)."""
    if args.mask or args.cascade or args.htc or args.selectp not in STEP_GFLOP:
        return None
    gf = STEP_GFLOP[args.selectp] * args.imgs / 2.0
    peak = {'bf16x6': 2500.0 / 6.0, 'f32': 157.3, 'bf16': 2500.0}[args.conv_math]
    tf = gf / out['ms_per_step']          # GFLOP / ms = TFLOP/s
    r = dict(bound='mfma', achieved=round(tf, 1), peak=round(peak, 1), unit='TFLOP/s',
             frac=round(tf / peak, 4), gflop_per_step=gf, ms_per_step=out['ms_per_step'],
             note='algorithmic flops of the whole iteration (conv + FC; SURVEY.md 8d) per GPU / wall '
                  'time per step / the arithmetic mode\'s matrix-pipe ceiling; the step also holds '
                  'HBM- and latency-bound kernels (targets, NMS, RoIAlign, losses, optimizer)')
    if args.selectp == 1:
        fl, mm, hb, nl, nh = step_layer_floor(args.imgs, args.conv_math)
        r['per_layer_floor'] = dict(
            floor_ms=round(fl, 3), frac=round(fl / out['ms_per_step'], 4), mfma_only_ms=round(mm, 3),
            hbm_bound_layers=nh, layers=nl, hbm_bound_ms=round(hb, 3),
            note='sum over the %d conv / linear launches of max(flops / %.1f TFLOP/s, algorithmic bytes / 8 TB/s); '
                 '%d of them (ResNet layer1, the stem, the RPN heads, fc_cls dW) are priced by their bytes; frac = '
                 'floor / ms_per_step' % (nl, peak, nh))
    try:
        with open(os.path.join(ROOT, 'profiles', 'step_families.json')) as f:
            fam = json.load(f)
        r['families_ms'] = fam['families_ms']
        r['families_source'] = fam['source']
    except Exception:
        pass
    return r
