"""Tensor-level wrappers over the C ABI (``include/bgs.h``).

PyTorch is plumbing here: device memory, the current HIP stream and autograd graph
edges.  All arithmetic happens in the hand-written gfx950 kernels of ``libbgs.so``.
Every wrapper requires CUDA(ROCm) tensors and raises otherwise — there is no CPU
path in the product.
"""
import ctypes
import itertools
import os

import torch

from . import capi

_WS = {}


def _workspace(nbytes, device):
    """Per (device, stream) scratch buffer; kernels of one stream are serialised."""
    key = (device.index, capi.raw_stream(device))
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf


def reset_workspaces():
    """Drop every cached scratch buffer.  Call before capturing a NEW hipGraph in a process that
    already captured one: a buffer handed out during an earlier capture lives in that graph's
    private memory pool and must not be baked into another graph."""
    _WS.clear()


_SIDE = {}
_PIPELINE_ACTIVE = [0]      # non-zero (the id of the owning train.TrunkPipeline) while one has batches in flight


def level_fork_enabled():
    """Small pyramid levels on a side stream (FPN output convs, RPN head): ``BGS_LEVEL_FORK=0`` turns it off.  Off as
    well while a ``train.TrunkPipeline`` has batches in flight: its pieces already run on three or four streams, and
    forks inside every one of them oversubscribe the hardware queues (depth 3: 6.17 ms per step with the inner
    forks, 5.74 without; profiles/r9h)."""
    return os.environ.get('BGS_LEVEL_FORK', '1') != '0' and not _PIPELINE_ACTIVE[0]


def shortcut_fork_enabled():
    """Projection shortcuts of frozen residual blocks on the side stream (backbone.Bottleneck.run):
    ``BGS_SHORTCUT_FORK=0`` turns it off (``BGS_LEVEL_FORK=0`` turns every fork off)."""
    return level_fork_enabled() and os.environ.get('BGS_SHORTCUT_FORK', '1') != '0'


def rpn_loss_fork_enabled():
    """The RPN loss chain (~12 short dependent launches) on the side stream (detectors._rpn_forward_train).  Default
    ``BGS_RPN_LOSS_FORK=auto``: when launching eagerly (two real streams: 6.45 -> 6.30 ms per step), not while a
    hipGraph is being captured (inside a replayed graph the fork's event edges cost more than the overlap returns:
    6.44 -> 6.49 ms); ``1`` / ``0`` force it."""
    v = os.environ.get('BGS_RPN_LOSS_FORK', 'auto')
    if v == 'auto':
        return level_fork_enabled() and not torch.cuda.is_current_stream_capturing()
    return level_fork_enabled() and v != '0'


class forked(object):
    """``with forked(device) as f: <launches>`` issues the block on the device's side stream, ordered after
    everything enqueued on the current stream so far; ``f.join()`` makes the current stream wait for it.

    Why: the stride-16 / 32 / 64 pyramid levels launch 168 / 48 / 16 workgroups on 256 CUs; next to the P2-level
    launch of the same layer (2100 workgroups) they fill CUs that would idle in its tail instead of taking their
    own serial slot (tools/stream_overlap_ab.py: 1.10 -> 1.02 ms for the five FPN output convs, eagerly and in
    a replayed hipGraph).  Dependent chains of tiny kernels do NOT gain from a fork (an event edge costs more than
    a back-to-back launch: 69 -> 139 us for two chains of 20), so only conv launches are forked.
    Tensors allocated inside the block belong to the side stream's pool; they are consumed after ``join()`` and
    freed after it, and the next block starts with an event recorded after those consumers: no reuse race.
    Tensors allocated on the MAIN stream that the block only reads are a different matter: if the host drops its
    last reference before ``join()``, the caching allocator may hand the block to a later main-stream allocation
    that is not ordered against the side stream.  ``hold(*tensors)`` keeps such inputs alive until ``join()``
    (which orders the main stream after the block) — every long-lived fork must hold what it reads."""

    def __init__(self, device, lane=0):
        # lane 0: short forks that are joined before the next one opens; lanes 1, 2: long-lived branches (the RPN loss
        # chain, the mask branch) that stay open across other forks — each lane is its own stream, so a short fork
        # never queues behind a long one
        # (one set of lanes PER PARENT stream: the trunk stage of train.TrunkPipeline forks its small pyramid levels from
        #  its own stream while the head stage forks from the main one — sharing a lane would queue the two stages'
        #  unrelated forks behind each other)
        key = (device.index if device.index is not None else torch.cuda.current_device(), int(lane),
               torch.cuda.current_stream(device).cuda_stream)
        if key not in _SIDE:
            _SIDE[key] = torch.cuda.Stream(device=device)
        self.side = _SIDE[key]
        self.device = device
        self.done = None
        self._held = None

    def hold(self, *tensors):
        """Keep main-stream tensors (or nests of them) that the forked block reads alive until ``join()``."""
        if self._held is None:
            self._held = []
        self._held.extend(tensors)
        return self

    def __enter__(self):
        main = torch.cuda.current_stream(self.device)
        ev = torch.cuda.Event()
        ev.record(main)
        self.side.wait_event(ev)
        self._ctx = torch.cuda.stream(self.side)
        self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        self.done = torch.cuda.Event()
        self.done.record(self.side)
        self._ctx.__exit__(*exc)
        return False

    def join(self):
        torch.cuda.current_stream(self.device).wait_event(self.done)
        # from here on every main-stream launch is ordered after the block: its inputs may be recycled
        self._held = None


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                'balancedgroupsoftmax_amd ops run only on the GPU (hand-written HIP kernels); '
                'got a %s tensor. There is no CPU fallback.' % t.device)


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


# ----------------------------------------------------------------------------------------
# label remap + sampling  (GSBBoxHeadWith0._remap_labels, gs_bbox_head_with0.py:91-112)
# ----------------------------------------------------------------------------------------
_seed_counter = itertools.count(1)


def gs_prepare(labels, label2binlabel, others_sample_ratio, seed=None, cls_weight=None,
               seed_offset=None, row_weights=None):
    """Device-side ``_remap_labels``: returns ``bin_labels [B,N] i32`` (the
    ``label2binlabel[b][labels]`` gather), ``weights [B,N] f32``, ``avg [B] f32``.  No host sync.

    ``seed``: draw identifier of the counter-based RNG; ``None`` takes the next value of
    a process-wide counter mixed with ``torch.initial_seed()`` (reproducible runs).
    ``seed_offset``: optional device int64 ``[1]`` draw counter added to the seed on the
    device (bump it with a tensor op; needed under hipGraph replay where ``seed`` is frozen).
    ``row_weights``: the detector's ``label_weights [N]``; rows <= 0 are padding slots of a
    fixed-shape batch and are left out of every count, draw and loss.
    """
    _require_cuda(labels, label2binlabel, cls_weight, row_weights)
    lib = capi.load()
    labels = labels.contiguous()
    assert labels.dtype == torch.int64 and label2binlabel.dtype == torch.int64
    B, C = label2binlabel.shape
    N = labels.numel()
    dev = labels.device
    if seed is None:
        seed = (torch.initial_seed() * 0x9E3779B1 + next(_seed_counter)) & 0xFFFFFFFFFFFFFFFF
    weights = torch.empty((B, N), dtype=torch.float32, device=dev)
    avg = torch.empty((B,), dtype=torch.float32, device=dev)
    bl = torch.empty((B, N), dtype=torch.int32, device=dev)
    cw_stride = 0
    if cls_weight is not None:
        assert cls_weight.dtype == torch.float32 and cls_weight.dim() == 2
        assert cls_weight.shape[0] == B - 1 and cls_weight.is_contiguous()
        cw_stride = cls_weight.shape[1]
    if row_weights is not None:
        row_weights = _f32c(row_weights)
        assert row_weights.numel() == N
    rc = lib.bgs_gs_prepare(capi.ptr(labels), capi.ptr(label2binlabel), capi.ptr(cls_weight),
                            cw_stride, capi.ptr(row_weights), N, C, B, float(others_sample_ratio),
                            int(seed),
                            capi.ptr(seed_offset), capi.ptr(bl), capi.ptr(weights), capi.ptr(avg),
                            capi.current_stream(dev))
    capi.check('bgs_gs_prepare', rc)
    return bl, weights, avg


# ----------------------------------------------------------------------------------------
# fused group-softmax loss  (GSBBoxHeadWith0.loss classification part, :160-171)
# ----------------------------------------------------------------------------------------
def _gs_loss_launch(logits, bin_labels, pred_slice_host, weights, avg, want_grad):
    lib = capi.load()
    N, W = logits.shape
    B = bin_labels.shape[0]
    dev = logits.device
    loss = torch.empty((B,), dtype=torch.float32, device=dev)
    dlogits = torch.empty_like(logits) if want_grad else None
    ws = _workspace(lib.bgs_gs_loss_workspace_bytes(N, B), dev)
    ps_keep, ps_ptr = capi.host_i64(pred_slice_host)
    rc = lib.bgs_gs_loss_fwd_bwd(capi.ptr(logits), capi.ptr(bin_labels), ps_ptr,
                                 capi.ptr(weights), capi.ptr(avg), N, B, W, capi.ptr(loss),
                                 capi.ptr(dlogits), capi.ptr(ws), capi.current_stream(dev))
    capi.check('bgs_gs_loss_fwd_bwd', rc)
    return loss, dlogits


class _GroupSoftmaxLoss(torch.autograd.Function):
    """losses[B] = fused kernel; the gradient w.r.t. the logits is produced by the SAME
    launch and only rescaled by the upstream scalars in backward (early-out when they are 1)."""

    @staticmethod
    def forward(ctx, logits, bin_labels, pred_slice_host, weights, avg):
        want_grad = logits.requires_grad
        z = _f32c(logits)
        loss, dlogits = _gs_loss_launch(z, bin_labels, pred_slice_host, weights, avg, want_grad)
        ctx.dlogits = dlogits
        ctx.pred_slice_host = pred_slice_host
        ctx.in_dtype = logits.dtype
        ctx.prev_g = None
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_loss):
        dl = ctx.dlogits
        if dl is None:
            return None, None, None, None, None
        if ctx.prev_g is not None:
            # The kernel's dlogits buffer is scaled IN PLACE and handed to autograd (no second
            # [N, W] tensor per step); a second backward through a retained graph would rescale a
            # tensor the first one already gave away (and could not recover bins whose first
            # upstream grad was 0).  Refuse instead of returning wrong gradients.
            raise RuntimeError('group_softmax_loss: the fused dlogits buffer was consumed by the '
                               'first backward; a second backward through a retained graph is not '
                               'supported — call the loss again')
        lib = capi.load()
        g = grad_loss.detach().to(torch.float32).contiguous()
        N, W = dl.shape
        ps_keep, ps_ptr = capi.host_i64(ctx.pred_slice_host)
        rc = lib.bgs_gs_scale_grad(capi.ptr(dl), ps_ptr, capi.ptr(g), N, ps_keep.shape[0], W,
                                   capi.current_stream(dl.device))
        capi.check('bgs_gs_scale_grad', rc)
        ctx.prev_g = g
        out = dl if ctx.in_dtype == torch.float32 else dl.to(ctx.in_dtype)
        return out, None, None, None, None


class _GsHeadFusedLoss(torch.autograd.Function):
    """``_remap_labels`` + ``_sample_others`` + loss + gradient in one streaming launch
    (``bgs_gs_head_loss_fused``); backward as :class:`_GroupSoftmaxLoss`."""

    @staticmethod
    def forward(ctx, logits, labels, l2b, pred_slice_host, ratio, seed, seed_offset, row_weights,
                debug):
        lib = capi.load()
        z = _f32c(logits)
        N, W = z.shape
        B, C = l2b.shape
        dev = z.device
        want_grad = logits.requires_grad
        loss = torch.empty((B,), dtype=torch.float32, device=dev)
        avg = torch.empty((B,), dtype=torch.float32, device=dev)
        dlogits = torch.empty_like(z) if want_grad else None
        bl = torch.empty((B, N), dtype=torch.int32, device=dev) if debug else None
        w = torch.empty((B, N), dtype=torch.float32, device=dev) if debug else None
        ws = _workspace(lib.bgs_gs_loss_workspace_bytes(N, B), dev)
        ps_keep, ps_ptr = capi.host_i64(pred_slice_host)
        rw = None if row_weights is None else _f32c(row_weights)
        rc = lib.bgs_gs_head_loss_fused(capi.ptr(z), capi.ptr(labels), capi.ptr(l2b), capi.ptr(rw),
                                        ps_ptr, N, C, B, W, float(ratio), int(seed),
                                        capi.ptr(seed_offset), capi.ptr(loss), capi.ptr(dlogits),
                                        capi.ptr(avg), capi.ptr(bl), capi.ptr(w), capi.ptr(ws),
                                        capi.current_stream(dev))
        capi.check('bgs_gs_head_loss_fused', rc)
        ctx.dlogits = dlogits
        ctx.pred_slice_host = pred_slice_host
        ctx.in_dtype = logits.dtype
        ctx.prev_g = None
        ctx.mark_non_differentiable(avg)
        if debug:
            ctx.mark_non_differentiable(bl, w)
            return loss, avg, bl, w
        return loss, avg

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_loss, *_unused):
        out = _GroupSoftmaxLoss.backward(ctx, grad_loss)[0]
        return (out,) + (None,) * 8


_CLASS_BITS = {}


def gs_class_bin_mask(label2binlabel):
    """``[C]`` uint16 device table, bit b = class is foreground in bin b
    (``bgs_gs_class_bin_mask``); cached per table tensor (the tables are constants of a head)."""
    _require_cuda(label2binlabel)
    key = (label2binlabel.data_ptr(), label2binlabel._version, tuple(label2binlabel.shape))
    hit = _CLASS_BITS.get(key)
    if hit is not None:
        return hit[1]
    lib = capi.load()
    B, C = label2binlabel.shape
    out = torch.empty(((C + 7) // 8 * 8,), dtype=torch.int16, device=label2binlabel.device)   # padded: read in 16-byte pieces
    rc = lib.bgs_gs_class_bin_mask(capi.ptr(label2binlabel), C, B, capi.ptr(out),
                                   capi.current_stream(label2binlabel.device))
    capi.check('bgs_gs_class_bin_mask', rc)
    if len(_CLASS_BITS) > 64:
        _CLASS_BITS.clear()
    _CLASS_BITS[key] = (label2binlabel, out)
    return out


_UNIT_GRAD = {}
_UNIT_LEN = 64


def unit_gradient(device, n=1):
    """The library's constant all-ones float32 tensor of a device (``[n]``, ``n <= 64``: a prefix view of ONE
    cached buffer): a root gradient for ``total.backward(unit_gradient(dev))`` that is not refilled every step
    (``backward()`` without an argument launches a fill) and that the loss edges of this package RECOGNISE —
    :func:`gs_head_step`'s backward, the head's term split and ``train.parse_losses`` hand it on by identity, so
    that "every upstream factor is 1" is known on the host and the gradient the forward produced is the answer
    without a launch.  Recognition = same storage AND an untouched version counter: a tensor somebody wrote to
    in place (the buffer is an ordinary CUDA tensor) is NOT taken for ones — the general path runs."""
    device = torch.device(device)
    if device.type == 'cuda' and device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    ent = _UNIT_GRAD.get(device)
    if ent is None or ent[0]._version != ent[1]:
        t = torch.ones(_UNIT_LEN, dtype=torch.float32, device=device)
        ent = _UNIT_GRAD[device] = (t, t._version)
    assert 1 <= n <= _UNIT_LEN
    return ent[0][:n]


def _is_unit_gradient(g):
    """``g`` is (a view of the head of) the cached all-ones buffer, and nobody has written to that buffer."""
    ent = _UNIT_GRAD.get(g.device)
    if ent is None or g.dtype != torch.float32:
        return False
    t, ver = ent
    return (t._version == ver and g.data_ptr() == t.data_ptr() and 1 <= g.numel() <= _UNIT_LEN
            and g.is_contiguous())


class _UnbindTermsFn(torch.autograd.Function):
    """``terms.unbind(0)`` for the fused head's loss vector with a backward that keeps the identity of unit
    gradients: when every term's upstream gradient IS the library's unit gradient (``loss.backward(
    unit_gradient(dev))`` through ``train.parse_losses``), the vector handed to the head's backward is the cached
    ones buffer itself — no stack launch, and the head launches nothing either.  Otherwise: one stack."""

    @staticmethod
    def forward(ctx, terms):
        ctx.n = terms.shape[0]
        ctx.set_materialize_grads(False)
        return tuple(terms.detach().unbind(0))

    @staticmethod
    def backward(ctx, *gs):
        if all(g is None for g in gs):
            return None
        if all(g is not None and g.numel() == 1 and _is_unit_gradient(g) for g in gs) and ctx.n <= _UNIT_LEN:
            return unit_gradient(gs[0].device, ctx.n)
        ref = next(g for g in gs if g is not None)
        return torch.stack([ref.new_zeros(()) if g is None else g.reshape(()) for g in gs])


def unbind_terms(terms):
    """The fused head's ``terms [B + 1]`` as B + 1 scalars (see :class:`_UnbindTermsFn`)."""
    return _UnbindTermsFn.apply(terms)


class _GsHeadStepFn(torch.autograd.Function):
    """``bgs_gs_head_step``: the whole ``GSBBoxHeadWith0.loss()`` as main kernel + reduce.  Returns
    ``terms [B + 1]`` = {per-bin losses (x loss weights), loss_bbox}, ``total [1]`` = their sum and
    ``avg [B]``; backward is ONE scaling launch over the gradients the forward already produced
    (early-out on the device when the upstream factors are 1), and NO launch when the upstream
    gradient of ``total`` is :func:`unit_gradient` itself."""

    @staticmethod
    def forward(ctx, logits, bbox_pred, labels, l2b, class_bits, pred_slice_host,
                bin_loss_weight_host, ratio, seed, counter, row_weights, bbox_targets, bbox_weights,
                num_reg_classes, beta, box_loss_weight, debug):
        import numpy as np
        lib = capi.load()
        z = _f32c(logits)
        N, W = z.shape
        B, C = l2b.shape
        dev = z.device
        box = bbox_pred is not None
        p = _f32c(bbox_pred) if box else None
        R = int(num_reg_classes) if box else 1
        if box:
            assert tuple(p.shape) == (N, 4 * R), (p.shape, N, R)
        terms = torch.empty((B + 1,), dtype=torch.float32, device=dev)
        total = torch.empty((1,), dtype=torch.float32, device=dev)
        avg = torch.empty((B,), dtype=torch.float32, device=dev)
        dlogits = torch.empty_like(z) if logits.requires_grad else None
        dbbox = torch.empty_like(p) if (box and bbox_pred.requires_grad) else None
        bl = torch.empty((B, N), dtype=torch.int32, device=dev) if debug else None
        w = torch.empty((B, N), dtype=torch.float32, device=dev) if debug else None
        ws = _workspace(lib.bgs_gs_loss_workspace_bytes(N, B), dev)
        ps_keep, ps_ptr = capi.host_i64(pred_slice_host)
        lw_keep = None if bin_loss_weight_host is None else \
            np.ascontiguousarray(bin_loss_weight_host, dtype=np.float32)
        lw_ptr = None if lw_keep is None else lw_keep.ctypes.data_as(ctypes.c_void_p)
        rw = None if row_weights is None else _f32c(row_weights)
        rc = lib.bgs_gs_head_step(
            capi.ptr(z), capi.ptr(labels), capi.ptr(l2b), capi.ptr(class_bits), capi.ptr(rw), ps_ptr,
            lw_ptr, N, C, B, W, float(ratio), int(seed), capi.ptr(counter), capi.ptr(p),
            capi.ptr(_f32c(bbox_targets)) if box else None,
            capi.ptr(_f32c(bbox_weights)) if box else None, R, float(beta), float(box_loss_weight),
            capi.ptr(terms), capi.ptr(total), capi.ptr(dlogits), capi.ptr(dbbox), capi.ptr(avg),
            capi.ptr(bl), capi.ptr(w), capi.ptr(ws), capi.current_stream(dev))
        capi.check('bgs_gs_head_step', rc)
        ctx.grads = (dlogits, dbbox)
        ctx.meta = (pred_slice_host, N, B, W, R, logits.dtype, None if not box else bbox_pred.dtype)
        ctx.consumed = False
        ctx.set_materialize_grads(False)     # an unused output (terms or total) arrives as None, not zeros
        if debug:
            ctx.mark_non_differentiable(avg, bl, w)
            return terms, total, avg, bl, w
        ctx.mark_non_differentiable(avg)
        return terms, total, avg

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_terms, g_total, *_unused):
        dlogits, dbbox = ctx.grads
        ps_host, N, B, W, R, zdt, pdt = ctx.meta
        if (dlogits is None and dbbox is None) or (g_terms is None and g_total is None):
            return (None,) * 17
        if ctx.consumed:     # the buffers are scaled in place and handed to autograd (see _GroupSoftmaxLoss)
            raise RuntimeError('gs_head_step: the fused gradient buffers were consumed by the first '
                               'backward; call the loss again')
        if (g_terms is None and _is_unit_gradient(g_total)) or \
                (g_total is None and g_terms is not None and g_terms.numel() == B + 1 and _is_unit_gradient(g_terms)):
            pass          # the root gradient is the library's unit gradient, handed down by identity (directly to
            #               `total`, or to every term through parse_losses): every factor is 1 — no launch
        else:
            lib = capi.load()
            gt = None if g_terms is None else g_terms.detach().to(torch.float32).contiguous()
            gT = None if g_total is None else g_total.detach().to(torch.float32).contiguous()
            ps_keep, ps_ptr = capi.host_i64(ps_host)
            dev = (dlogits if dlogits is not None else dbbox).device
            rc = lib.bgs_gs_head_step_scale_grad(capi.ptr(dlogits), capi.ptr(dbbox), ps_ptr, capi.ptr(gt),
                                                 capi.ptr(gT), N, B, W, R, capi.current_stream(dev))
            capi.check('bgs_gs_head_step_scale_grad', rc)
        ctx.consumed = True
        ctx.grads = (None, None)      # sole owner now: AccumulateGrad can take the buffer instead of cloning it
        gz = dlogits if (dlogits is None or zdt == torch.float32) else dlogits.to(zdt)
        gp = dbbox if (dbbox is None or pdt == torch.float32) else dbbox.to(pdt)
        del dlogits, dbbox
        return (gz, gp) + (None,) * 15


def gs_head_step(cls_score, labels, label2binlabel, pred_slice, others_sample_ratio, seed,
                 draw_counter=None, row_weights=None, bin_loss_weight=None, bbox_pred=None,
                 bbox_targets=None, bbox_weights=None, num_reg_classes=1, beta=1.0,
                 box_loss_weight=1.0, debug=False, class_bits='auto'):
    """The whole BAGS head loss as two launches (``bgs_gs_head_step``): returns ``(terms, total,
    avg)`` with ``terms [B + 1]`` = per-bin classification losses (times ``bin_loss_weight``, a HOST
    sequence) followed by ``loss_bbox`` (0 without ``bbox_pred``), ``total [1]`` = their sum (both
    differentiable) and ``avg [B]``.  ``draw_counter``: device int64 ``[1]``, read as this call's
    draw index and advanced BY THE KERNEL (fresh "others" samples under hipGraph replay without a
    tensor op).  ``class_bits``: ``'auto'`` builds / reuses :func:`gs_class_bin_mask`, ``None`` lets
    every workgroup derive the table from ``label2binlabel``.  Limits of :func:`gs_head_loss_fused`."""
    _require_cuda(cls_score, labels, label2binlabel, row_weights, bbox_pred, bbox_targets,
                  bbox_weights, draw_counter)
    assert labels.dtype == torch.int64 and label2binlabel.dtype == torch.int64
    assert cls_score.dim() == 2 and 0 < cls_score.shape[0] <= GS_FUSED_MAX_ROWS
    if draw_counter is not None:
        assert draw_counter.dtype == torch.int64 and draw_counter.numel() == 1
    if bbox_pred is not None:
        assert bbox_targets is not None and bbox_weights is not None
    label2binlabel = label2binlabel.contiguous()
    if isinstance(class_bits, str):
        class_bits = gs_class_bin_mask(label2binlabel)
    return _GsHeadStepFn.apply(cls_score, bbox_pred, labels.contiguous(), label2binlabel, class_bits,
                               _host_pred_slice(pred_slice), bin_loss_weight,
                               float(others_sample_ratio), int(seed), draw_counter, row_weights,
                               bbox_targets, bbox_weights, int(num_reg_classes), float(beta),
                               float(box_loss_weight), bool(debug))


GS_FUSED_MAX_ROWS = 4096


def gs_head_loss_fused(cls_score, labels, label2binlabel, pred_slice, others_sample_ratio,
                       seed, seed_offset=None, row_weights=None, debug=False):
    """The BAGS head's classification loss straight from ``labels``: ``(losses [B], avg [B])``
    (``debug=True`` adds the bin labels and sample weights ``[B, N]`` the kernel used).
    Requires ``N <= 4096`` rows, ``B <= 15`` and no per-class reweighting; same values, bit for
    bit, as :func:`gs_prepare` + :func:`group_softmax_loss` with the same seed."""
    _require_cuda(cls_score, labels, label2binlabel, row_weights)
    assert labels.dtype == torch.int64 and label2binlabel.dtype == torch.int64
    assert cls_score.dim() == 2 and 0 < cls_score.shape[0] <= GS_FUSED_MAX_ROWS
    return _GsHeadFusedLoss.apply(cls_score, labels.contiguous(), label2binlabel.contiguous(),
                                  _host_pred_slice(pred_slice), float(others_sample_ratio), int(seed),
                                  seed_offset, row_weights, bool(debug))


def group_softmax_loss(cls_score, bin_labels, pred_slice, weights=None, avg=None):
    """Per-bin weighted CE of the ``[N, W]`` logits -> ``[B]`` losses (differentiable
    w.r.t. ``cls_score``).

    ``bin_labels [B,N] i32`` / ``weights [B,N]`` / ``avg [B]`` as produced by ``gs_prepare``;
    ``pred_slice``: HOST ``[B,2]`` (start, length) table (numpy / list / CPU tensor).
    """
    _require_cuda(cls_score, bin_labels, weights, avg)
    assert cls_score.dim() == 2 and bin_labels.dtype == torch.int32
    assert bin_labels.is_contiguous() and bin_labels.shape[1] == cls_score.shape[0]
    ps = _host_pred_slice(pred_slice)
    assert ps.shape == (bin_labels.shape[0], 2)
    if weights is not None:
        weights = _f32c(weights)
        assert weights.shape == tuple(bin_labels.shape)
    if avg is not None:
        avg = _f32c(avg)
    return _GroupSoftmaxLoss.apply(cls_score, bin_labels, ps, weights, avg)


def _host_pred_slice(pred_slice):
    import numpy as np
    if isinstance(pred_slice, torch.Tensor):
        if pred_slice.is_cuda:
            raise RuntimeError('pred_slice must live on the host (static metadata passed by '
                               'value to the kernels); keep a CPU copy instead of syncing')
        pred_slice = pred_slice.numpy()
    return np.ascontiguousarray(pred_slice, dtype=np.int64)


# ----------------------------------------------------------------------------------------
# inference score merge  (GSBBoxHeadWith0._merge_score, :239-273)
# ----------------------------------------------------------------------------------------
def gs_merge_score(cls_score, pred_slice, cls2col, num_classes):
    _require_cuda(cls_score, cls2col)
    lib = capi.load()
    z = _f32c(cls_score)
    N, W = z.shape
    ps_keep, ps_ptr = capi.host_i64(_host_pred_slice(pred_slice))
    B = ps_keep.shape[0]
    assert cls2col.dtype == torch.int32 and cls2col.numel() == num_classes
    out = torch.empty((N, num_classes), dtype=torch.float32, device=z.device)
    rc = lib.bgs_gs_merge_score(capi.ptr(z), ps_ptr, capi.ptr(cls2col), N,
                                num_classes, B, W, capi.ptr(out), capi.current_stream(z.device))
    capi.check('bgs_gs_merge_score', rc)
    return out


# ----------------------------------------------------------------------------------------
# box regression loss  (loss_bbox branch, :173-185)
# ----------------------------------------------------------------------------------------
class _BBoxSmoothL1(torch.autograd.Function):

    @staticmethod
    def forward(ctx, bbox_pred, labels, bbox_targets, bbox_weights, num_reg_classes, beta,
                avg_factor, loss_weight):
        lib = capi.load()
        p = _f32c(bbox_pred)
        N = p.shape[0]
        R = int(num_reg_classes)
        assert p.shape[1] == 4 * R
        dev = p.device
        want_grad = bbox_pred.requires_grad
        loss = torch.empty((1,), dtype=torch.float32, device=dev)
        dpred = torch.empty_like(p) if want_grad else None
        ws = _workspace(lib.bgs_bbox_loss_workspace_bytes(N), dev)
        rc = lib.bgs_bbox_smooth_l1_fwd_bwd(
            capi.ptr(p), capi.ptr(labels), capi.ptr(bbox_targets), capi.ptr(bbox_weights), N, R,
            float(beta), float(avg_factor), float(loss_weight), capi.ptr(loss), capi.ptr(dpred),
            capi.ptr(ws), capi.current_stream(dev))
        capi.check('bgs_bbox_smooth_l1_fwd_bwd', rc)
        ctx.dpred = dpred
        ctx.in_dtype = bbox_pred.dtype
        return loss[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_loss):
        if ctx.dpred is None:
            return (None,) * 8
        g = ctx.dpred * grad_loss  # upstream scalar; dense like the autograd result
        if ctx.in_dtype != torch.float32:
            g = g.to(ctx.in_dtype)
        return (g,) + (None,) * 7


def bbox_smooth_l1_loss(bbox_pred, labels, bbox_targets, bbox_weights, num_reg_classes,
                        beta=1.0, avg_factor=None, loss_weight=1.0):
    """``loss_weight * sum(smooth_l1(pred[pos, labels[pos]] - targets[pos]) * w[pos]) / avg``."""
    _require_cuda(bbox_pred, labels, bbox_targets, bbox_weights)
    if avg_factor is None:
        avg_factor = bbox_targets.shape[0]
    return _BBoxSmoothL1.apply(bbox_pred, labels.contiguous(), _f32c(bbox_targets),
                               _f32c(bbox_weights), num_reg_classes, beta, avg_factor,
                               loss_weight)


# ----------------------------------------------------------------------------------------
# conv / linear on the matrix cores (NHWC fp32 activations, [Cout,R,S,Cin] fp32 weights)
#   math 'bf16x6' (default): bf16 MFMA on exactly split operands, fp32-faithful (csrc/conv_bfx.hip)
#   math 'f32':              v_mfma_f32_32x32x2_f32 (csrc/conv_igemm.hip, csrc/conv_halo.hip)
#   math 'bf16':             operands rounded to bf16 for the MFMA, fp32 accumulate / storage: the
#                            arithmetic of the reference's fp16 autocast (mmdet/core/fp16/) — cfg[4]
# ----------------------------------------------------------------------------------------
_CONV_MATHS = ('bf16x6', 'f32', 'bf16')


def _checked_conv_math(mode, origin):
    if mode not in _CONV_MATHS:
        raise ValueError('%s: unknown conv math %r (expected one of %s)' % (origin, mode, _CONV_MATHS))
    return mode


# a typo in the environment ('fp32', 'BF16X6') must not select an arithmetic silently: validated at import
_CONV_MATH = [_checked_conv_math(os.environ.get('BGS_CONV_MATH', 'bf16x6'), 'BGS_CONV_MATH')]


def set_conv_math(mode):
    """'bf16x6' | 'f32' | 'bf16'; returns the previous mode.  The first two give fp32-accurate
    results (the split kernel's error against fp64 is not above the fp32 MFMA kernel's; 'f32'
    keeps the bit-exact fp32 fma chain); 'bf16' is the reduced-precision mode of cfg[4]."""
    _checked_conv_math(mode, 'set_conv_math')
    prev = _CONV_MATH[0]
    _CONV_MATH[0] = mode
    return prev


def conv_math():
    return _CONV_MATH[0]


class conv_math_scope(object):
    """``with conv_math_scope('bf16'): ...`` — the mode is restored on exit (also on error)."""

    def __init__(self, mode):
        self.mode = _checked_conv_math(mode, 'conv_math_scope')

    def __enter__(self):
        self.prev = set_conv_math(self.mode)
        return self

    def __exit__(self, *exc):
        set_conv_math(self.prev)
        return False


# Split planes of FROZEN filters only, one entry per storage address: {data_ptr: (version, shape,
# source tensor, planes)}.  The source is kept alive so that its address cannot be handed to another
# tensor while the entry exists; an entry whose version went stale is overwritten in place, so the
# cache never holds more than one split per live frozen tensor.  Trained filters (and per-step
# temporaries such as the folded copies of a trained trunk) are NEVER cached: their split launch is
# part of every step — inside a captured hipGraph it is re-run by every replay, so an eager forward
# after replayed optimizer steps (which do not bump ``_version``) cannot read stale planes.
# bf16 STORAGE of the frozen trunk's activations in the 'bf16' arithmetic mode (csrc/conv_bf16s.hip;
# the memory side of the reference's wrap_fp16_model, mmdet/core/fp16/decorators.py:8-80).
# BGS_BF16_STORAGE=0 keeps fp32 tensors in HBM (operands rounded inside the kernels), the A/B arm.
_BF16_STORAGE = [os.environ.get('BGS_BF16_STORAGE', '1') != '0']


def set_bf16_storage(on):
    """-> previous value.  Effective only while the arithmetic mode is ``'bf16'``."""
    prev = _BF16_STORAGE[0]
    _BF16_STORAGE[0] = bool(on)
    return prev


def bf16_storage_active():
    return _CONV_MATH[0] == 'bf16' and _BF16_STORAGE[0]


# Residual-fork fusion on the trainable trunk (backbone.Bottleneck.run): the block output's ReLU backward and
# the fork's gradient sum ride in the dgrad epilogue of the next block's first conv.  BGS_FORK_FUSION=0: the
# separate threshold / add passes (A/B).
_FORK_FUSION = [os.environ.get('BGS_FORK_FUSION', '1') != '0']


def fork_fusion_enabled():
    return _FORK_FUSION[0]


def set_fork_fusion(on):
    prev = _FORK_FUSION[0]
    _FORK_FUSION[0] = bool(on)
    return prev


_SPLIT_CACHE = {}
_SPLIT_CACHE_MAX = 1024


class CacheFence(object):
    """Orders the FIRST cross-stream use of a cached tensor behind the launches that wrote it (split planes
    and folded weights first built inside a pipeline piece are written on that piece's side stream and then read from
    other streams with nothing in between).  ``CacheFence()`` right after the producing launches records an event on
    the current stream; ``wait()`` before a use makes the current stream wait for it when it is another stream and
    the event has not completed — and forgets the event once it has, so a warm cache costs one attribute test.  Under
    stream capture it does nothing (a captured step is ordered by its own fork / join edges)."""
    __slots__ = ('event', 'stream')

    def __init__(self, device=None):
        self.event = None
        self.stream = None
        if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            st = torch.cuda.current_stream(device)
            self.event = torch.cuda.Event()
            self.event.record(st)
            self.stream = st.cuda_stream

    def wait(self, device=None):
        ev = self.event
        if ev is None:
            return
        if torch.cuda.is_current_stream_capturing():
            return
        if ev.query():
            self.event = None
            return
        st = torch.cuda.current_stream(device)
        if st.cuda_stream != self.stream:
            st.wait_event(ev)


def clear_split_cache():
    _SPLIT_CACHE.clear()


def bfx_split_weights(w2d, cache=True):
    """``w2d [rows, K]`` fp32 -> the three bf16 planes ``bgs_conv2d_nhwc_f32_bfx_ws`` streams.
    ``cache=True`` declares the tensor FROZEN (the caller decides: a detached view of a trained
    parameter has ``requires_grad == False`` too)."""
    _require_cuda(w2d)
    lib = capi.load()
    assert w2d.dtype == torch.float32 and w2d.dim() == 2 and w2d.is_contiguous()
    rows, K = w2d.shape
    cacheable = cache and not w2d.requires_grad
    key = w2d.data_ptr()
    if cacheable:
        hit = _SPLIT_CACHE.get(key)
        if hit is not None and hit[0] == w2d._version and hit[1] == (rows, K):
            hit[4].wait(w2d.device)
            return hit[3]
    out = torch.empty(lib.bgs_conv_bfx_weight_bytes(rows, K), dtype=torch.uint8, device=w2d.device)
    rc = lib.bgs_conv_bfx_split_weights(capi.ptr(w2d), capi.ptr(out), rows, K,
                                        capi.current_stream(w2d.device))
    capi.check('bgs_conv_bfx_split_weights', rc)
    if cacheable:
        if key not in _SPLIT_CACHE and len(_SPLIT_CACHE) >= _SPLIT_CACHE_MAX:
            _SPLIT_CACHE.pop(next(iter(_SPLIT_CACHE)))      # oldest entry (insertion order)
        _SPLIT_CACHE[key] = (w2d._version, (rows, K), w2d, out, CacheFence(w2d.device))
    return out


def presplit(*weights):
    """Fill the split cache for frozen conv filters ``[Cout,R,S,Cin]`` on the CURRENT stream — before a
    :class:`forked` block whose launches (on two streams) share them."""
    for w in weights:
        if w is not None and w.is_cuda and w.dim() == 4 and not w.requires_grad and \
                (_CONV_MATH[0] != 'f32' or bf16_storage_active()):
            bfx_split_weights(w.view(w.shape[0], -1), cache=True)


def bfx_split_weights_dgrad(w_krsc, cache=True):
    """The split planes of the DATA-GRADIENT filter of ``w_krsc [Cout,R,S,Cin]`` (= ``bfx_split_weights``
    of ``dgrad_filter(w_krsc).view(Cin, R*S*Cout)``) in one launch (``bgs_conv_bfx_split_weights_dgrad``).
    ``cache`` as in :func:`bfx_split_weights`."""
    _require_cuda(w_krsc)
    lib = capi.load()
    assert w_krsc.dtype == torch.float32 and w_krsc.dim() == 4 and w_krsc.is_contiguous()
    Cout, R, S, Cin = w_krsc.shape
    cacheable = cache and not w_krsc.requires_grad
    key = (w_krsc.data_ptr(), 'dgrad')
    if cacheable:
        hit = _SPLIT_CACHE.get(key)
        if hit is not None and hit[0] == w_krsc._version and hit[1] == (Cout, R, S, Cin):
            hit[4].wait(w_krsc.device)
            return hit[3]
    out = torch.empty(lib.bgs_conv_bfx_weight_bytes(Cin, R * S * Cout), dtype=torch.uint8,
                      device=w_krsc.device)
    rc = lib.bgs_conv_bfx_split_weights_dgrad(capi.ptr(w_krsc), capi.ptr(out), Cout, R, S, Cin,
                                              capi.current_stream(w_krsc.device))
    capi.check('bgs_conv_bfx_split_weights_dgrad', rc)
    if cacheable:
        if key not in _SPLIT_CACHE and len(_SPLIT_CACHE) >= _SPLIT_CACHE_MAX:
            _SPLIT_CACHE.pop(next(iter(_SPLIT_CACHE)))
        _SPLIT_CACHE[key] = (w_krsc._version, (Cout, R, S, Cin), w_krsc, out, CacheFence(w_krsc.device))
    return out


def conv_bfx_tuning(tile=0, splitk=-1, halo_splits=-1, halo_variant=0, halo_geom=-1, halo_flags=0, halo_wide=-1):
    """Process-wide tuning / test hook of the bf16x6 kernels (see include/bgs_tuning.h).  ``halo_geom``: the
    pixel tile of the halo kernel, -1 default | 0: 8 x 16 | 1: 10 x 12 | 2: 5 x 21 | 3: fewest tiles per image.
    ``halo_wide``: the 16 x 16-pixel units of variant 7, -1 leave as is (the all-default call restores the
    environment's default) | 0 off | 1 automatic | 2 every eligible layer."""
    lib = capi.load()
    lib.bgs_conv_bfx_tuning(int(tile), int(splitk))
    lib.bgs_conv3x3_halo_bfx_tuning(int(halo_splits), int(halo_variant) | ((int(halo_geom) + 1) << 16) |
                                    (int(halo_flags) << 20) | ((int(halo_wide) + 1) << 24))


def set_halo_wide(mode):
    """The wide-pixel-tile mode of the halo 3x3 kernel alone (``bgs_conv3x3_halo_bfx_wide``): -1 environment default |
    0 off | 1 automatic schedule | 2 every eligible layer; returns the previous mode."""
    return capi.load().bgs_conv3x3_halo_bfx_wide(int(mode))


def conv_bfx_last_launch():
    """-> dict(tile, splits, halo_nb, halo_splits, ...) of the last bf16x6 launches.  ``halo_variant`` 7: the wide
    pixel tile ran (``halo_wide_units`` 256-pixel units) followed by ``halo_tail_units`` 128-pixel units on variant 4."""
    import ctypes
    lib = capi.load()
    a, c, d, e, f, g = (ctypes.c_int() for _ in range(6))
    lib.bgs_conv_bfx_last_launch(ctypes.byref(a), ctypes.byref(c))
    lib.bgs_conv3x3_halo_bfx_last_launch(ctypes.byref(d), ctypes.byref(e))
    lib.bgs_conv3x3_halo_bfx_last_wide(ctypes.byref(f), ctypes.byref(g))
    return dict(tile=a.value & ~0x400, ring_stages=3 if a.value & 0x400 else 4, splits=c.value,
                halo_nb=d.value & 0xff, halo_variant=(d.value >> 8) & 0xff, halo_geom=d.value >> 16,
                halo_splits=e.value, halo_wide_units=f.value, halo_tail_units=g.value)


CENSUS = dict(bf16_ring8=0, grouped_lds=1, halo_bfx4=2, dma_ring64=3, gs_head_fused=4, conv1x1_bres=5,
              wgrad_bfx=6, roi_bwd_gather=7, bf16s=8, grouped_bf16s=9, bfx_wide=10, gs_scale_grad=11, halo_wide=12, stem_fused=13, fused_c3=14,
              planes_3x3=15, planes_1x1=16)


def launch_census(reset=False):
    """-> {family: launches since the last reset} (``bgs_launch_census``; include/bgs.h BGS_CENSUS_*)."""
    lib = capi.load()
    out = {k: lib.bgs_launch_census(v, 0) for k, v in CENSUS.items()}
    if reset:
        lib.bgs_launch_census(0, 1)
    return out


def conv_tuning(tile=0, bk=0, splitk=0, noswizzle=0):
    """Process-wide tuning / test hook of the fp32 MFMA conv kernel (see include/bgs.h)."""
    capi.load().bgs_conv_tuning(int(tile), int(bk), int(splitk), int(noswizzle))


def conv_last_launch():
    """-> dict(tile, bk, up, splits) of the last fp32 MFMA conv launch."""
    import ctypes
    a, b, c, d = (ctypes.c_int() for _ in range(4))
    capi.load().bgs_conv_last_launch(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d))
    return dict(tile=a.value, bk=b.value, up=c.value, splits=d.value)


def _use_halo_bfx(M, Cout):
    """bf16x6 path: the halo kernel wins on every 3x3 stride-1 layer of cfg[1] down to the 25x42
    maps (profiles/r2g_bfx_sweep.txt: P2 0.74 vs 1.01 ms, P3 0.22 vs 0.28, M = 8400: 0.078 vs
    0.089, M = 2100: 0.035 vs 0.039); the 13x21 level (M = 546) stays on the general kernel."""
    env = os.environ.get('BGS_CONV_HALO')
    if env == '1':
        return True
    if env == '0':
        return False
    return M >= 2000


def _use_halo_kernel(M, Cout):
    """``BGS_CONV_HALO`` = 1: every eligible 3x3 layer, 0: never; unset: where it was measured
    faster than the general kernel (tools/conv_halo_check.py, profiles/r3m_conv_halo.txt): the
    large-M, Cout % 128 == 0 layers (FPN output conv and RPN conv on P2: 114 vs 98 TFLOP/s); its
    fixed 128 x 128 tile without split-K loses on the smaller ones."""
    env = os.environ.get('BGS_CONV_HALO')
    if env == '1':
        return True
    if env == '0':
        return False
    return M >= 100000 and Cout % 128 == 0


_FUSED_C3 = [os.environ.get('BGS_FUSED_C3', '1') != '0']


def set_fused_c3(on):
    """-> previous value.  ``BGS_FUSED_C3=0`` / ``set_fused_c3(False)``: a frozen bottleneck runs conv2 and conv3 as two
    launches (the A/B arm; bit-identical)."""
    prev = _FUSED_C3[0]
    _FUSED_C3[0] = bool(on)
    return prev


def fused_c3_eligible(x, w2, w3, stride, residual):
    """conv2 -> conv3 of a frozen bottleneck in one launch (``bgs_conv3x3_c3_fused_nhwc_f32_bfx``, round 6): fp32 NHWC
    activations under the fp32-faithful arithmetic, 3x3 / stride 1, 64 -> 64 -> 256 channels (ResNet-50 layer1), the map
    large enough for the halo kernel, nothing that needs a gradient."""
    env = os.environ.get('BGS_FUSED_C3')              # (read at every call: tools/step_ab.py switches it between arms)
    on = _FUSED_C3[0] if env is None else env != '0'
    return (on and _CONV_MATH[0] == 'bf16x6' and x.is_cuda and x.dtype == torch.float32 and stride == 1
            and tuple(w2.shape[:3]) == (64, 3, 3) and w2.shape[3] == 64 and tuple(w3.shape) == (256, 1, 1, 64)
            and x.shape[3] == 64 and _use_halo_bfx(x.shape[0] * x.shape[1] * x.shape[2], 64)
            # (>= 700 pixel tiles: below that the unfused conv2 splits its channel chunks over gridDim.z — another
            #  summation order — and a short grid gains nothing from the fusion; csrc/conv_bfx.hip halo_bfx_plan)
            and x.shape[0] * ((x.shape[1] + 7) // 8) * ((x.shape[2] + 15) // 16) >= 700
            and not (torch.is_grad_enabled() and (x.requires_grad or w2.requires_grad or w3.requires_grad or
                                                  (residual is not None and residual.requires_grad))))


def conv3x3_c3_fused_nhwc(x, w2_krsc, bias2, w3_krsc, bias3, residual=None, relu3=True, out=None):
    """``y = act(conv1x1(relu(conv3x3(x, w2) + bias2), w3) + bias3 + residual)`` in ONE launch (the second half of a frozen
    bottleneck, resnet.py:239-266); bit-identical to the two ``conv2d_nhwc`` calls it replaces."""
    _require_cuda(x, w2_krsc, bias2, w3_krsc, bias3, residual)
    lib = capi.load()
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    N, H, W, Cmid = x.shape
    Cout3 = w3_krsc.shape[0]
    assert tuple(w2_krsc.shape) == (Cmid, 3, 3, Cmid) and tuple(w3_krsc.shape) == (Cout3, 1, 1, Cmid)
    if residual is not None:
        assert tuple(residual.shape) == (N, H, W, Cout3) and residual.is_contiguous() and residual.dtype == torch.float32
    if out is None:
        out = torch.empty((N, H, W, Cout3), dtype=torch.float32, device=x.device)
    w2s = bfx_split_weights(w2_krsc.view(Cmid, 9 * Cmid), cache=True)
    w3s = bfx_split_weights(w3_krsc.view(Cout3, Cmid), cache=True)
    rc = lib.bgs_conv3x3_c3_fused_nhwc_f32_bfx(capi.ptr(x), capi.ptr(w2s), capi.ptr(bias2), capi.ptr(w3s),
                                               capi.ptr(bias3), capi.ptr(residual), capi.ptr(out), N, H, W, Cmid,
                                               Cout3, int(bool(relu3)), capi.current_stream(x.device))
    capi.check('bgs_conv3x3_c3_fused_nhwc_f32_bfx', rc)
    return out


def conv2d_nhwc(x, w_krsc, bias=None, stride=1, pad=0, relu=False, residual=None,
                residual_mode=0, out=None, frozen_weight=True, out_dtype=None):
    """``y = act(conv(x, w) + bias + residual)``; x ``[N,H,W,Cin]``, w ``[Cout,R,S,Cin]``.
    Forward only (the shipped BAGS configs freeze every conv: selectp=1, tools/train.py:49-57).
    ``frozen_weight=False``: ``w_krsc`` is (a detached view of) a tensor that changes from step to
    step — its bf16 planes are split on every call and never cached.
    ``x`` in bf16 selects the bf16 storage kernels (cfg[4] bf16 mode; ``out_dtype``: bf16 (default)
    or fp32 for consumers outside the trunk)."""
    _require_cuda(x, w_krsc, bias, residual)
    lib = capi.load()
    assert x.dtype in (torch.float32, torch.bfloat16) and w_krsc.dtype == torch.float32
    assert x.is_contiguous() and w_krsc.is_contiguous() and x.dim() == 4 and w_krsc.dim() == 4
    N, H, W, Cin = x.shape
    Cout, R, S, Cin2 = w_krsc.shape
    assert Cin == Cin2, (x.shape, w_krsc.shape)
    Ho = (H + 2 * pad - R) // stride + 1
    Wo = (W + 2 * pad - S) // stride + 1
    if residual is not None and residual_mode == 0:
        residual_mode = 1
    if residual is not None:
        exp = (N, Ho, Wo, Cout) if residual_mode == 1 else (N, Ho // 2, Wo // 2, Cout)
        assert tuple(residual.shape) == exp and residual.is_contiguous(), (residual.shape, exp)
    if x.dtype == torch.bfloat16:
        # bf16 storage mode (csrc/conv_bf16s.hip): bf16 activations, the bf16(w) plane, fp32 accumulate
        if out_dtype is None:
            out_dtype = torch.bfloat16
        assert out_dtype in (torch.float32, torch.bfloat16)
        assert residual is None or residual.dtype in (torch.float32, torch.bfloat16)
        if out is None:
            out = torch.empty((N, Ho, Wo, Cout), dtype=out_dtype, device=x.device)
        assert out.dtype == out_dtype
        wsplit = bfx_split_weights(w_krsc.view(Cout, R * S * Cin), cache=bool(frozen_weight))
        rc = lib.bgs_conv2d_nhwc_bf16s(capi.ptr(x), capi.ptr(wsplit), capi.ptr(bias), capi.ptr(residual),
                                       residual_mode if residual is not None else 0,
                                       int(residual is not None and residual.dtype == torch.bfloat16),
                                       capi.ptr(out), int(out_dtype == torch.bfloat16), N, H, W, Cin,
                                       Cout, R, S, stride, pad, int(bool(relu)),
                                       capi.current_stream(x.device))
        capi.check('bgs_conv2d_nhwc_bf16s', rc)
        return out
    assert out_dtype in (None, torch.float32), 'fp32 activations produce fp32 (bf16 out needs bf16 in)'
    assert residual is None or residual.dtype == torch.float32
    if out is None:
        out = torch.empty((N, Ho, Wo, Cout), dtype=torch.float32, device=x.device)
    halo_ok = R == 3 and S == 3 and stride == 1 and pad == 1 and residual is None and Cin % 16 == 0
    if _CONV_MATH[0] != 'f32':
        planes = 3 if _CONV_MATH[0] == 'bf16x6' else 1
        wsplit = bfx_split_weights(w_krsc.view(Cout, R * S * Cin), cache=bool(frozen_weight))
        st = capi.current_stream(x.device)
        if halo_ok and _use_halo_bfx(N * Ho * Wo, Cout):
            wsb = lib.bgs_conv3x3_halo_bfx_workspace_bytes(N, H, W, Cin, Cout)
            ws = _workspace(wsb, x.device) if wsb else None
            rc = lib.bgs_conv3x3_halo_nhwc_f32_bfx(capi.ptr(x), capi.ptr(wsplit), capi.ptr(bias),
                                                   capi.ptr(out), N, H, W, Cin, Cout,
                                                   int(bool(relu)), planes, capi.ptr(ws), wsb, st)
            capi.check('bgs_conv3x3_halo_nhwc_f32_bfx', rc)
            return out
        wsb = lib.bgs_conv_bfx_workspace_bytes(N * Ho * Wo, Cout, R * S * Cin)
        ws = _workspace(wsb, x.device) if wsb else None
        rc = lib.bgs_conv2d_nhwc_f32_bfx_ws(capi.ptr(x), capi.ptr(wsplit), capi.ptr(bias),
                                            capi.ptr(residual), capi.ptr(out), N, H, W, Cin, Cout,
                                            R, S, stride, pad, int(bool(relu)), residual_mode,
                                            planes, capi.ptr(ws), wsb, st)
        capi.check('bgs_conv2d_nhwc_f32_bfx_ws', rc)
        return out
    if halo_ok and _use_halo_kernel(N * Ho * Wo, Cout):
        # halo-resident 3x3 kernel (csrc/conv_halo.hip): the input patch is staged in LDS once per
        # channel chunk and shared by the nine taps
        rc = lib.bgs_conv3x3_halo_nhwc_f32(capi.ptr(x), capi.ptr(w_krsc), capi.ptr(bias),
                                           capi.ptr(out), N, H, W, Cin, Cout, int(bool(relu)),
                                           capi.current_stream(x.device))
        capi.check('bgs_conv3x3_halo_nhwc_f32', rc)
        return out
    wsb = lib.bgs_conv2d_workspace_bytes(N * Ho * Wo, Cout)      # split-K scratch (small-M layers)
    ws = _workspace(wsb, x.device) if wsb else None
    rc = lib.bgs_conv2d_nhwc_f32_ws(capi.ptr(x), capi.ptr(w_krsc), capi.ptr(bias),
                                    capi.ptr(residual), capi.ptr(out), N, H, W, Cin, Cout, R, S,
                                    stride, pad, int(bool(relu)), residual_mode, capi.ptr(ws),
                                    wsb, capi.current_stream(x.device))
    capi.check('bgs_conv2d_nhwc_f32_ws', rc)
    return out


def dgrad_filter(w_krsc):
    """``[Cout,R,S,Cin]`` -> ``[Cin,R,S,Cout]`` with both spatial axes flipped: the filter layout
    ``bgs_conv2d_dgrad_nhwc_f32`` streams (K-major for the transposed problem)."""
    return w_krsc.detach().permute(3, 1, 2, 0).flip(1, 2).contiguous()


def conv2d_dgrad_nhwc(dy, w_krsc, in_hw, stride=1, pad=0, residual=None, residual_mode=0,
                      mask=None, wt=None, frozen_weight=False):
    """Data gradient of :func:`conv2d_nhwc`: ``dy [N,Ho,Wo,Cout]`` -> ``dx [N,H,W,Cin]``.
    ``residual`` is added (mode 1 same shape, mode 3: ``[N,2H,2W,Cin]`` 2x2-sum-pooled), then
    ``mask > 0`` gates the result (ReLU backward of the conv's input)."""
    _require_cuda(dy, w_krsc, residual, mask)
    lib = capi.load()
    assert dy.dtype == torch.float32 and dy.is_contiguous() and dy.dim() == 4
    Cout, R, S, Cin = w_krsc.shape
    N, Ho, Wo, Cout2 = dy.shape
    H, W = in_hw
    assert Cout == Cout2 and Ho == (H + 2 * pad - R) // stride + 1 and \
        Wo == (W + 2 * pad - S) // stride + 1, (dy.shape, w_krsc.shape, in_hw)
    if Cout % 4 != 0:      # e.g. the fused RPN head (15 outputs): zero-pad the reduction axis
        padc = 4 - Cout % 4
        dy = torch.nn.functional.pad(dy, (0, padc))
        w_krsc = torch.nn.functional.pad(w_krsc.detach(), (0, 0, 0, 0, 0, 0, 0, padc))
        Cout += padc
        wt = None
    wt_is_temp = wt is None
    fused_split = wt is None and _CONV_MATH[0] != 'f32' and os.environ.get('BGS_DGRAD_FUSED_SPLIT', '1') != '0'
    if wt is None and not fused_split:
        wt = dgrad_filter(w_krsc)
    if residual is not None and residual_mode == 0:
        residual_mode = 1
    if residual is not None:
        exp = (N, H, W, Cin) if residual_mode == 1 else (N, 2 * H, 2 * W, Cin)
        assert tuple(residual.shape) == exp and residual.is_contiguous(), (residual.shape, exp)
    if mask is not None:
        assert tuple(mask.shape) == (N, H, W, Cin) and mask.is_contiguous()
    dx = torch.empty((N, H, W, Cin), dtype=torch.float32, device=dy.device)
    if _CONV_MATH[0] != 'f32':
        if fused_split:
            # flip + transpose + split in one launch; cached for frozen filters (a detached view of a
            # trained filter changes from step to step: w_krsc.requires_grad tells nothing about it,
            # so only tensors the caller passes as frozen parameters are cached — see _ConvFn.backward)
            wt_split = bfx_split_weights_dgrad(w_krsc.detach().contiguous(), cache=bool(frozen_weight))
        else:
            wt_split = bfx_split_weights(wt.view(Cin, R * S * Cout), cache=not wt_is_temp)
        if R == 3 and S == 3 and stride == 1 and pad == 1 and residual is None and Cout % 16 == 0 \
                and _use_halo_bfx(N * H * W, Cin) and os.environ.get('BGS_DGRAD_HALO', '1') != '0':
            # the data gradient of a 3x3 / stride 1 / pad 1 conv is the same conv of dy with the
            # flipped, transposed filter: the halo-resident kernel of the forward pass (ReLU-backward
            # mask in its epilogue) instead of the general operand ring
            wsb = lib.bgs_conv3x3_halo_bfx_workspace_bytes(N, H, W, Cout, Cin)
            ws = _workspace(wsb, dy.device) if wsb else None
            rc = lib.bgs_conv3x3_halo_nhwc_f32_bfx_ex(
                capi.ptr(dy), capi.ptr(wt_split), None, capi.ptr(mask), capi.ptr(dx), N, H, W, Cout,
                Cin, 0, 3 if _CONV_MATH[0] == 'bf16x6' else 1, capi.ptr(ws), wsb,
                capi.current_stream(dy.device))
            if rc == 0:
                return dx
            if rc != 2:          # BGS_ERR_UNSUPPORTED (an A/B variant without the mask): general path
                capi.check('bgs_conv3x3_halo_nhwc_f32_bfx_ex', rc)
        wsb = lib.bgs_conv_bfx_workspace_bytes(N * H * W, Cin, R * S * Cout)
        ws = _workspace(wsb, dy.device) if wsb else None
        rc = lib.bgs_conv2d_dgrad_nhwc_f32_bfx_ws(capi.ptr(dy), capi.ptr(wt_split),
                                                  capi.ptr(residual), capi.ptr(mask), capi.ptr(dx),
                                                  N, H, W, Cin, Cout, R, S, stride, pad,
                                                  residual_mode,
                                                  3 if _CONV_MATH[0] == 'bf16x6' else 1,
                                                  capi.ptr(ws), wsb,
                                                  capi.current_stream(dy.device))
        capi.check('bgs_conv2d_dgrad_nhwc_f32_bfx_ws', rc)
        return dx
    wsb = lib.bgs_conv2d_workspace_bytes(N * H * W, Cin)
    ws = _workspace(wsb, dy.device) if wsb else None
    rc = lib.bgs_conv2d_dgrad_nhwc_f32_ws(capi.ptr(dy), capi.ptr(wt), capi.ptr(residual),
                                          capi.ptr(mask), capi.ptr(dx), N, H, W, Cin, Cout, R, S,
                                          stride, pad, residual_mode, capi.ptr(ws), wsb,
                                          capi.current_stream(dy.device))
    capi.check('bgs_conv2d_dgrad_nhwc_f32_ws', rc)
    return dx


def conv2d_wgrad_nhwc(x, dy, ksize, stride=1, pad=0, bias=False, dw=None, db=None,
                      accumulate=False):
    """Weight (and bias) gradient of :func:`conv2d_nhwc`: ``x [N,H,W,Cin]``, ``dy [N,Ho,Wo,Cout]``
    -> ``dw [Cout,R,S,Cin]`` (``db [Cout]``).  ``accumulate`` adds into the given ``dw``/``db``."""
    _require_cuda(x, dy)
    lib = capi.load()
    assert x.dtype == torch.float32 and dy.dtype == torch.float32
    assert x.is_contiguous() and dy.is_contiguous() and x.dim() == 4 and dy.dim() == 4
    N, H, W, Cin = x.shape
    R = S = int(ksize)
    Cout = dy.shape[3]
    assert tuple(dy.shape[:3]) == (N, (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1)
    dev = x.device
    if Cout % 4 != 0:      # pad the output-channel axis, slice the result
        assert dw is None and db is None and not accumulate
        padc = 4 - Cout % 4
        res = conv2d_wgrad_nhwc(x, torch.nn.functional.pad(dy, (0, padc)), ksize, stride, pad, bias)
        if bias:
            return res[0][:Cout].contiguous(), res[1][:Cout].contiguous()
        return res[:Cout].contiguous()
    if dw is None:
        assert not accumulate
        dw = torch.empty((Cout, R, S, Cin), dtype=torch.float32, device=dev)
    if bias and db is None:
        assert not accumulate
        db = torch.empty((Cout,), dtype=torch.float32, device=dev)
    if _CONV_MATH[0] != 'f32':      # bf16 matrix cores: bf16x6 (fp32-faithful) or bf16 operands
        ws = _workspace(lib.bgs_conv2d_wgrad_bfx_workspace_bytes(N, H, W, Cin, Cout, R, S, stride, pad), dev)
        rc = lib.bgs_conv2d_wgrad_nhwc_f32_bfx(capi.ptr(x), capi.ptr(dy), capi.ptr(dw),
                                               capi.ptr(db) if bias else None, N, H, W, Cin, Cout, R,
                                               S, stride, pad, int(bool(accumulate)),
                                               3 if _CONV_MATH[0] == 'bf16x6' else 1, capi.ptr(ws),
                                               capi.current_stream(dev))
        capi.check('bgs_conv2d_wgrad_nhwc_f32_bfx', rc)
        return (dw, db) if bias else dw
    ws = _workspace(lib.bgs_conv2d_wgrad_workspace_bytes(N, H, W, Cin, Cout, R, S, stride, pad), dev)
    rc = lib.bgs_conv2d_wgrad_nhwc_f32(capi.ptr(x), capi.ptr(dy), capi.ptr(dw),
                                       capi.ptr(db) if bias else None, N, H, W, Cin, Cout, R, S,
                                       stride, pad, int(bool(accumulate)), capi.ptr(ws),
                                       capi.current_stream(dev))
    capi.check('bgs_conv2d_wgrad_nhwc_f32', rc)
    return (dw, db) if bias else dw


class _FoldConvBNFn(torch.autograd.Function):
    """``(w [Cout,Cin,R,S], conv_bias, gamma, beta) -> (wf [Cout,R,S,CinP], bf [Cout])``: eval-mode
    BatchNorm folded into the filter (``bgs_fold_conv_bn_fwd``) with the fold's backward as ONE launch
    (``bgs_fold_conv_bn_bwd``: dw, dconv_bias, dgamma, dbeta) — as tensor ops the fold is ~20
    elementwise / permute launches per conv and step on the `selectp = 0` path."""

    @staticmethod
    def forward(ctx, w, conv_bias, gamma, beta, mean, var, eps, cin_padded):
        lib = capi.load()
        wc = _f32c(w)
        Cout, Cin, R, S = wc.shape
        cinp = int(cin_padded) if cin_padded else Cin
        dev = wc.device
        cb = None if conv_bias is None else _f32c(conv_bias)
        bn = gamma is not None
        g, b_, m, v = (_f32c(t) for t in (gamma, beta, mean, var)) if bn else (None,) * 4
        wf = torch.empty((Cout, R, S, cinp), dtype=torch.float32, device=dev)
        bf = torch.empty((Cout,), dtype=torch.float32, device=dev)
        rc = lib.bgs_fold_conv_bn_fwd(capi.ptr(wc), capi.ptr(cb), capi.ptr(g), capi.ptr(b_),
                                      capi.ptr(m), capi.ptr(v), float(eps), Cout, Cin, R, S, cinp,
                                      capi.ptr(wf), capi.ptr(bf), capi.current_stream(dev))
        capi.check('bgs_fold_conv_bn_fwd', rc)
        ctx.save_for_backward(wc, cb, g, m, v)
        ctx.cfg = (float(eps), cinp, conv_bias is not None, bn)
        return wf, bf

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dwf, dbf):
        wc, cb, g, m, v = ctx.saved_tensors
        eps, cinp, has_cb, bn = ctx.cfg
        lib = capi.load()
        Cout, Cin, R, S = wc.shape
        dev = wc.device
        need = ctx.needs_input_grad
        dw = torch.empty_like(wc) if need[0] else None
        dcb = torch.empty((Cout,), dtype=torch.float32, device=dev) if (has_cb and need[1]) else None
        dg = torch.empty((Cout,), dtype=torch.float32, device=dev) if (bn and need[2]) else None
        db = torch.empty((Cout,), dtype=torch.float32, device=dev) if (bn and need[3]) else None
        rc = lib.bgs_fold_conv_bn_bwd(capi.ptr(dwf.contiguous()), capi.ptr(dbf.contiguous()),
                                      capi.ptr(wc), capi.ptr(cb), capi.ptr(g), capi.ptr(m), capi.ptr(v),
                                      eps, Cout, Cin, R, S, cinp, capi.ptr(dw), capi.ptr(dcb),
                                      capi.ptr(dg), capi.ptr(db), capi.current_stream(dev))
        capi.check('bgs_fold_conv_bn_bwd', rc)
        return dw, dcb, dg, db, None, None, None, None


def fold_conv_bn(weight, conv_bias=None, bn_weight=None, bn_bias=None, running_mean=None,
                 running_var=None, eps=1e-5, cin_padded=None):
    """-> ``(wf [Cout,R,S,CinP], bf [Cout])``, differentiable w.r.t. ``weight``, ``conv_bias``,
    ``bn_weight``, ``bn_bias`` (one launch forward, one backward)."""
    _require_cuda(weight, conv_bias, bn_weight, bn_bias, running_mean, running_var)
    return _FoldConvBNFn.apply(weight, conv_bias, bn_weight, bn_bias, running_mean, running_var,
                               float(eps), cin_padded)


class _ConvFn(torch.autograd.Function):
    """Differentiable fused conv: ``y = act(conv(x, w) + b + residual)`` with gradients for
    x, w, b and the residual (the `selectp = 0` path).

    backward: ``dz = dy * (y > 0)`` (ReLU), ``dx = dgrad(dz, w)``, ``dw, db = wgrad(x, dz)``,
    ``dresidual = dz`` (mode 1) or its 2x2 sum-pool (mode 2: nearest-2x upsampled residual).

    ReLU-backward placement: ``relu='consumers'`` declares that EVERY consumer of ``y`` is a
    ``_ConvFn`` called with ``mask_input=True`` — each of them gates its own ``dx`` with
    ``(y > 0)`` inside the dgrad kernel's epilogue (the mask is linear, so gating the addends
    equals gating their sum) and this node skips the separate masking pass over ``dy``."""

    @staticmethod
    def forward(ctx, x, w, bias, residual, stride, pad, relu, residual_mode, mask_input,
                passthrough=False):
        # w.requires_grad: a trained parameter or the per-step fold of one (its detached view has
        # requires_grad == False, so the decision is taken HERE): split every step, never cached
        y = conv2d_nhwc(x.detach(), w.detach(), None if bias is None else bias.detach(),
                        stride=stride, pad=pad, relu=bool(relu),
                        residual=None if residual is None else residual.detach(),
                        residual_mode=residual_mode, frozen_weight=not w.requires_grad)
        ctx.cfg = (stride, pad, relu, residual_mode if residual is not None else 0,
                   bias is not None, mask_input)
        ctx.math = _CONV_MATH[0]          # backward replays in the arithmetic of this forward
        ctx.save_for_backward(x, w, y if relu is True else None)
        if passthrough:
            # second output: an alias of x for the OTHER consumer of x at a fork (the identity path
            # of a residual block).  Its gradient arrives in this node's backward together with
            # dy, so the fork's `g_conv + g_identity` is the dgrad kernel's residual epilogue
            # instead of a separate pass over the map
            ctx.set_materialize_grads(False)
            return y, x.view_as(x)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy, dalias=None):
        x, w, y = ctx.saved_tensors
        stride, pad, relu, res_mode, has_bias, mask_input = ctx.cfg
        if dy is None:            # (passthrough nodes only: y itself had no consumer)
            return (dalias,) + (None,) * 9
        dz = dy.contiguous()
        if relu is True:          # one fused pass: dy * (y > 0)
            dz = torch.ops.aten.threshold_backward(dz, y, 0.0)
        need_x, need_w, need_b, need_r = ctx.needs_input_grad[:4]
        dx = dw = db = dres = None
        with conv_math_scope(ctx.math):
            if need_x:
                dx = conv2d_dgrad_nhwc(dz, w, (x.shape[1], x.shape[2]), stride=stride, pad=pad,
                                       residual=None if dalias is None else dalias.contiguous(),
                                       mask=x if mask_input else None,
                                       frozen_weight=not w.requires_grad)
            if need_w or (has_bias and need_b):
                out = conv2d_wgrad_nhwc(x, dz, w.shape[1], stride=stride, pad=pad, bias=has_bias)
                dw, db = out if has_bias else (out, None)
        if res_mode and need_r:
            if res_mode == 1:
                dres = dz
            else:
                n, h, wd, c = dz.shape
                dres = dz.view(n, h // 2, 2, wd // 2, 2, c).sum(dim=(2, 4))
        return dx, dw, db, dres, None, None, None, None, None, None


class _ReluGateFn(torch.autograd.Function):
    """Identity whose backward applies the ReLU gate of its (post-ReLU) input: turns a
    ``relu='consumers'`` conv output into an ordinary tensor that ANY consumer may use."""

    @staticmethod
    def forward(ctx, y):
        ctx.save_for_backward(y)
        return y.view_as(y)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return torch.ops.aten.threshold_backward(g.contiguous(), y, 0.0)


def relu_gate(y):
    """Public-boundary adapter for ``relu='consumers'`` outputs (see :class:`_ConvFn`): the
    returned tensor carries the ReLU backward itself.  No-op when nothing is being recorded."""
    if torch.is_grad_enabled() and y.requires_grad and getattr(y, '_bgs_consumers_mask', False):
        return _ReluGateFn.apply(y)
    return y


def conv2d_autograd(x, w_krsc, bias=None, stride=1, pad=0, relu=False, residual=None,
                    residual_mode=0, mask_input=False, out_dtype=None, passthrough=False):
    """:func:`conv2d_nhwc` that records an autograd node when any input requires grad.
    ``relu``: False / True / ``'consumers'`` (see :class:`_ConvFn`); ``mask_input``: ``x`` is the
    output of a ``relu='consumers'`` conv; ``passthrough``: returns ``(y, x_alias)`` — hand ``x_alias``
    to the other consumer of ``x`` (a residual identity path) and the fork's gradient sum rides in
    this conv's dgrad epilogue.  The contract is checked where it can be: outputs of
    ``relu='consumers'`` convs are tagged, and feeding a tagged tensor to a conv WITHOUT
    ``mask_input=True`` raises (its gradient would skip the ReLU gate)."""
    ts = [t for t in (x, w_krsc, bias, residual) if t is not None]
    if torch.is_grad_enabled() and any(t.requires_grad for t in ts):
        if x.dtype != torch.float32:
            raise RuntimeError('conv2d_autograd: bf16-stored activations are forward-only (frozen trunk); '
                               'convert to fp32 before a trainable layer')
        if getattr(x, '_bgs_consumers_mask', False) and not mask_input:
            raise RuntimeError("conv2d_autograd: x is the output of a relu='consumers' conv (its ReLU "
                               "backward is delegated to its consumers) but mask_input is False")
        if residual is not None and residual_mode == 0:
            residual_mode = 1
        out = _ConvFn.apply(x, w_krsc, bias, residual, stride, pad, relu, residual_mode,
                            bool(mask_input), bool(passthrough))
        y = out[0] if passthrough else out
        if relu == 'consumers':
            y._bgs_consumers_mask = True
        return out
    y = conv2d_nhwc(x, w_krsc, bias, stride=stride, pad=pad, relu=bool(relu),
                    residual=residual, residual_mode=residual_mode, out_dtype=out_dtype)
    return (y, x) if passthrough else y


def linear(x, weight, bias=None, relu=False, frozen_weight=None):
    """``act(x @ weight.T + bias)`` for ``x [M,K]``, ``weight [Cout,K]`` (nn.Linear layout).
    ``frozen_weight`` (default: ``not weight.requires_grad``, evaluated BEFORE detaching) decides
    whether the bf16 planes of the weight may be cached (see :func:`bfx_split_weights`)."""
    M, K = x.shape
    if frozen_weight is None:
        frozen_weight = not weight.requires_grad
    x = _f32c(x)
    weight = _f32c(weight)
    y = conv2d_nhwc(x.view(M, 1, 1, K), weight.view(weight.shape[0], 1, 1, K),
                    None if bias is None else _f32c(bias), relu=relu, frozen_weight=frozen_weight)
    return y.view(M, weight.shape[0])


class _LinearFn(torch.autograd.Function):
    """y = x @ W^T + b on the fp32 matrix cores, with gradients for x, W, b.

    dW = dY^T X and dX = dY W are the same implicit-GEMM kernel fed with transposed copies
    (both operands must be K-contiguous).  Under selectp=1 only ``fc_cls`` takes this path:
    dW_cls = dlogits^T x is 1.3 GMAC."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return linear(x.detach(), weight.detach(), None if bias is None else bias.detach(),
                      frozen_weight=not weight.requires_grad)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous().float()
        gx = gw = gb = None
        M, K = x.shape
        Nout = weight.shape[0]
        if ctx.needs_input_grad[0]:
            gx = linear(gy, weight.detach().t().contiguous(), frozen_weight=False)   # [M,N] x [K,N]^T
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or want_b:
            # dW[n,k] = sum_m gy[m,n] x[m,k]: the split-reduction wgrad kernel, no transposes
            if Nout % 4 == 0 and K % 4 == 0:
                out = conv2d_wgrad_nhwc(_f32c(x).view(M, 1, 1, K), gy.view(M, 1, 1, Nout), 1,
                                        bias=want_b)
                gw, gb = (out[0], out[1]) if want_b else (out, None)
                gw = gw.view(Nout, K)
            else:
                gw = linear(gy.t().contiguous(), x.detach().t().contiguous(), frozen_weight=False)
                gb = gy.sum(0) if want_b else None
        return gx, gw, gb


def linear_autograd(x, weight, bias=None, relu=False, mask_input=False):
    """``act(x @ weight.T + bias)``: fused forward-only kernel when nothing requires grad,
    otherwise the differentiable path (``relu`` / ``mask_input`` as in :func:`conv2d_autograd`)."""
    needs = torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or
                                         (bias is not None and bias.requires_grad))
    if not needs:
        return linear(x, weight, bias, relu=bool(relu))
    M, K = x.shape
    Nout = weight.shape[0]
    y = _ConvFn.apply(x.float().contiguous().view(M, 1, 1, K),
                      weight.float().contiguous().view(Nout, 1, 1, K),
                      None if bias is None else bias.float(), None, 1, 0, relu, 0,
                      bool(mask_input))
    return y.view(M, Nout)


def _grouped_conv3x3_launch(x, w, bias, groups, stride, relu):
    lib = capi.load()
    N, H, W, C = x.shape
    if x.dtype == torch.bfloat16:      # bf16 storage mode: bf16 in, bf16 out (csrc/conv_bf16s.hip)
        out = torch.empty((N, (H - 1) // stride + 1, (W - 1) // stride + 1, C), dtype=torch.bfloat16,
                          device=x.device)
        rc = lib.bgs_grouped_conv3x3_nhwc_bf16s(capi.ptr(x), capi.ptr(w), capi.ptr(bias), capi.ptr(out),
                                                N, H, W, C, int(groups), int(stride), int(bool(relu)),
                                                capi.current_stream(x.device))
        capi.check('bgs_grouped_conv3x3_nhwc_bf16s', rc)
        return out
    out = torch.empty((N, (H - 1) // stride + 1, (W - 1) // stride + 1, C), dtype=torch.float32,
                      device=x.device)
    if _CONV_MATH[0] == 'bf16' and stride == 1 and C % 64 == 0:
        # cfg[4] bf16 mode: operands rounded to bf16 in the kernel, fp32 accumulate (the other
        # layouts keep the fp32 kernel: more precise than the mode asks for)
        rc = lib.bgs_grouped_conv3x3_nhwc_bf16ops(capi.ptr(x), capi.ptr(w), capi.ptr(bias),
                                                  capi.ptr(out), N, H, W, C, int(groups), 1,
                                                  int(bool(relu)), capi.current_stream(x.device))
        capi.check('bgs_grouped_conv3x3_nhwc_bf16ops', rc)
        return out
    rc = lib.bgs_grouped_conv3x3_nhwc_f32(capi.ptr(x), capi.ptr(w), capi.ptr(bias), capi.ptr(out),
                                          N, H, W, C, int(groups), int(stride), int(bool(relu)),
                                          capi.current_stream(x.device))
    capi.check('bgs_grouped_conv3x3_nhwc_f32', rc)
    return out


def grouped_dgrad_filter(w, groups):
    """``w [C,3,3,cg]`` -> ``wt[g*cg+cl][2-r][2-s][co_local] = w[g*cg+co_local][r][s][cl]``: the
    filter of the data gradient (transposed inside each group, both spatial axes flipped)."""
    C, _, _, cg = w.shape
    return w.detach().view(groups, cg, 3, 3, cg).flip(2, 3).permute(0, 4, 2, 3, 1) \
        .reshape(C, 3, 3, cg).contiguous()


class _GroupedConvFn(torch.autograd.Function):
    """Differentiable grouped 3x3 conv (+ bias + ReLU): ResNeXt conv2 under ``selectp = 0``
    (resnext.py:47-57).  dx = the MFMA forward kernel on dy with the per-group transposed filter,
    dw / db = ``bgs_grouped_conv3x3_wgrad_nhwc_f32``."""

    @staticmethod
    def forward(ctx, x, w, bias, groups, stride, relu):
        y = _grouped_conv3x3_launch(x.detach(), w.detach(), None if bias is None else bias.detach(),
                                    groups, stride, relu)
        ctx.cfg = (groups, stride, bool(relu), bias is not None)
        ctx.save_for_backward(x, w, y if relu else None)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        groups, stride, relu, has_bias = ctx.cfg
        lib = capi.load()
        dz = dy.contiguous()
        if relu:
            dz = torch.ops.aten.threshold_backward(dz, y, 0.0)
        N, H, W, C = x.shape
        st = capi.current_stream(x.device)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = grouped_dgrad_filter(w, groups)
            dx = torch.empty_like(x)
            rc = lib.bgs_grouped_conv3x3_dgrad_nhwc_f32(capi.ptr(dz), capi.ptr(wt), capi.ptr(dx), N,
                                                        H, W, C, groups, stride, st)
            capi.check('bgs_grouped_conv3x3_dgrad_nhwc_f32', rc)
        want_b = has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or want_b:
            dw = torch.empty_like(w)
            db = torch.empty((C,), dtype=torch.float32, device=x.device) if want_b else None
            ws = _workspace(lib.bgs_grouped_conv3x3_wgrad_workspace_bytes(N, H, W, C, groups, stride),
                            x.device)
            rc = lib.bgs_grouped_conv3x3_wgrad_nhwc_f32(capi.ptr(x), capi.ptr(dz), capi.ptr(dw),
                                                        capi.ptr(db), N, H, W, C, groups, stride, 0,
                                                        capi.ptr(ws), st)
            capi.check('bgs_grouped_conv3x3_wgrad_nhwc_f32', rc)
        return dx, dw, db, None, None, None


def grouped_conv3x3_nhwc(x, w, bias, groups, stride=1, relu=False):
    """Grouped 3x3 / pad 1 conv (ResNeXt conv2): x ``[N,H,W,C]``, w ``[C,3,3,C/groups]``; records
    an autograd node when an input requires grad."""
    _require_cuda(x, w, bias)
    assert x.dtype in (torch.float32, torch.bfloat16) and x.is_contiguous() and x.dim() == 4
    assert w.is_contiguous() and w.dtype == torch.float32
    N, H, W, C = x.shape
    assert tuple(w.shape) == (C, 3, 3, C // groups), (w.shape, C, groups)
    if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or
                                    (bias is not None and bias.requires_grad)):
        if x.dtype != torch.float32:
            raise RuntimeError('grouped_conv3x3_nhwc: bf16-stored activations are forward-only')
        return _GroupedConvFn.apply(x, w, bias, int(groups), int(stride), bool(relu))
    return _grouped_conv3x3_launch(x, w, bias, groups, stride, relu)


def nchw_to_nhwc4(img):
    """``[N, C <= 4, H, W]`` image batch -> ``[N, H, W, 4]`` fp32, channels zero-padded (the stem conv's
    16-byte pixels) in one launch (``bgs_nchw_to_nhwc4_f32``)."""
    _require_cuda(img)
    lib = capi.load()
    x = img.detach().to(torch.float32).contiguous()
    N, C, H, W = x.shape
    out = torch.empty((N, H, W, 4), dtype=torch.float32, device=x.device)
    rc = lib.bgs_nchw_to_nhwc4_f32(capi.ptr(x), capi.ptr(out), N, C, H, W, capi.current_stream(x.device))
    capi.check('bgs_nchw_to_nhwc4_f32', rc)
    return out


def stem_fused_enabled():
    """The one-launch stem (conv 7x7 / s2 + ReLU + max-pool from the NCHW image, csrc/stem_fused.hip):
    ``BGS_STEM_FUSED=0`` keeps the three-launch chain (A/B; read at every call)."""
    return os.environ.get('BGS_STEM_FUSED', '1') != '0' and _CONV_MATH[0] == 'bf16x6'


def stem_fused_split_weights(w_krsc):
    """Folded stem filter ``[64, 7, 7, >=3]`` -> the split planes ``bgs_stem_conv7x7s2_relu_maxpool_nchw_f32`` reads."""
    _require_cuda(w_krsc)
    lib = capi.load()
    w = _f32c(w_krsc)
    assert w.dim() == 4 and tuple(w.shape[:3]) == (64, 7, 7) and w.shape[3] >= 3, tuple(w.shape)
    out = torch.empty(lib.bgs_stem_fused_weight_bytes(), dtype=torch.uint8, device=w.device)
    rc = lib.bgs_stem_fused_split_weights(capi.ptr(w), int(w.shape[3]), capi.ptr(out), capi.current_stream(w.device))
    capi.check('bgs_stem_fused_split_weights', rc)
    return out


def stem_fused(img, wsplit, bias):
    """``relu(conv7x7s2(img) + bias)`` max-pooled 3x3 / s2 / p1 in ONE launch: ``img [N, 3, H, W]`` (NCHW, as the
    reference hands it over) -> ``[N, PH, PW, 64]`` NHWC fp32.  Forward only (the frozen stem of the BAGS configs)."""
    _require_cuda(img, wsplit, bias)
    lib = capi.load()
    x = img.detach().to(torch.float32).contiguous()
    N, C, H, W = x.shape
    assert C == 3
    CH, CW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    PH, PW = (CH - 1) // 2 + 1, (CW - 1) // 2 + 1
    out = torch.empty((N, PH, PW, 64), dtype=torch.float32, device=x.device)
    rc = lib.bgs_stem_conv7x7s2_relu_maxpool_nchw_f32(capi.ptr(x), capi.ptr(wsplit), capi.ptr(_f32c(bias)),
                                                      capi.ptr(out), N, H, W, capi.current_stream(x.device))
    capi.check('bgs_stem_conv7x7s2_relu_maxpool_nchw_f32', rc)
    return out


def _maxpool_launch(x, out_dtype=torch.float32):
    lib = capi.load()
    N, H, W, C = x.shape
    if out_dtype == torch.bfloat16:    # entry of the bf16-stored trunk
        out = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), dtype=torch.bfloat16, device=x.device)
        rc = lib.bgs_maxpool3x3s2_nhwc_f32_to_bf16(capi.ptr(x), capi.ptr(out), N, H, W, C,
                                                   capi.current_stream(x.device))
        capi.check('bgs_maxpool3x3s2_nhwc_f32_to_bf16', rc)
        return out
    out = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), dtype=torch.float32,
                      device=x.device)
    rc = lib.bgs_maxpool3x3s2_nhwc_f32(capi.ptr(x), capi.ptr(out), N, H, W, C,
                                       capi.current_stream(x.device))
    capi.check('bgs_maxpool3x3s2_nhwc_f32', rc)
    return out


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return _maxpool_launch(x.detach())

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        lib = capi.load()
        N, H, W, C = x.shape
        dx = torch.empty_like(x)
        rc = lib.bgs_maxpool3x3s2_bwd_nhwc_f32(capi.ptr(x), capi.ptr(dy.contiguous()), capi.ptr(dx),
                                               N, H, W, C, capi.current_stream(x.device))
        capi.check('bgs_maxpool3x3s2_bwd_nhwc_f32', rc)
        return dx


def maxpool3x3s2_nhwc(x, out_dtype=torch.float32):
    """3x3 / stride 2 / pad 1 max pooling (ResNet stem); differentiable when ``x`` requires grad
    (``frozen_stages < 1``).  ``out_dtype=torch.bfloat16``: the pooled map is stored in bf16."""
    _require_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    if torch.is_grad_enabled() and x.requires_grad:
        assert out_dtype == torch.float32
        return _MaxPoolFn.apply(x)
    return _maxpool_launch(x, out_dtype)


# ----------------------------------------------------------------------------------------
# RoIAlign (multi-level, NHWC) and batched NMS
# ----------------------------------------------------------------------------------------
def roi_align_nhwc(feats, rois, featmap_strides, out_size=7, sample_num=2, finest_scale=56,
                   return_levels=False, pool=1, out=None):
    """feats: list of ``[N,H_l,W_l,C]`` maps; rois ``[K,5]`` -> ``[K, out, out, C]``.
    ``pool``: RoIAlign on the ``(out*pool)^2`` grid averaged down to ``out^2`` in the kernel
    (== ``F.adaptive_avg_pool2d`` of the finer result);  ``out``: accumulate INTO this tensor."""
    import ctypes
    _require_cuda(rois, *feats)
    lib = capi.load()
    L = len(feats)
    N, _, _, C = feats[0].shape
    for f in feats:
        assert f.dtype == torch.float32 and f.is_contiguous() and f.shape[0] == N and f.shape[3] == C
    rois = _f32c(rois)
    K = rois.shape[0]
    ph, pw = (out_size, out_size) if isinstance(out_size, int) else tuple(out_size)
    accumulate = out is not None
    if accumulate:
        assert tuple(out.shape) == (K, ph, pw, C) and out.is_contiguous() and \
            out.dtype == torch.float32, (out.shape, (K, ph, pw, C))
    else:
        out = torch.empty((K, ph, pw, C), dtype=torch.float32, device=rois.device)
    lv = torch.empty((K,), dtype=torch.int32, device=rois.device) if return_levels else None
    ptrs = (ctypes.c_void_p * L)(*[f.data_ptr() for f in feats])
    hs = (ctypes.c_int * L)(*[int(f.shape[1]) for f in feats])
    ws = (ctypes.c_int * L)(*[int(f.shape[2]) for f in feats])
    sc = (ctypes.c_float * L)(*[1.0 / s for s in featmap_strides])
    rc = lib.bgs_roi_align_nhwc_fwd_ex(ptrs, hs, ws, sc, L, N, float(finest_scale), capi.ptr(rois),
                                       K, C, ph, pw, sample_num, int(pool), int(accumulate),
                                       capi.ptr(out), capi.ptr(lv),
                                       capi.current_stream(rois.device))
    capi.check('bgs_roi_align_nhwc_fwd_ex', rc)
    return (out, lv) if return_levels else out


def roi_align_nhwc_bwd(dout, rois, dfeats, featmap_strides, sample_num=2, finest_scale=56, pool=1):
    """Scatter ``dout [K,ph,pw,C]`` into the per-level gradient maps ``dfeats`` (accumulated
    into, in place, with fp32 atomics)."""
    import ctypes
    _require_cuda(dout, rois, *dfeats)
    lib = capi.load()
    L = len(dfeats)
    N, _, _, C = dfeats[0].shape
    for f in dfeats:
        assert f.dtype == torch.float32 and f.is_contiguous() and f.shape[0] == N and f.shape[3] == C
    rois = _f32c(rois)
    dout = _f32c(dout)
    K, ph, pw, C2 = dout.shape
    assert C2 == C and rois.shape[0] == K
    ptrs = (ctypes.c_void_p * L)(*[f.data_ptr() for f in dfeats])
    hs = (ctypes.c_int * L)(*[int(f.shape[1]) for f in dfeats])
    ws = (ctypes.c_int * L)(*[int(f.shape[2]) for f in dfeats])
    sc = (ctypes.c_float * L)(*[1.0 / s for s in featmap_strides])
    rc = lib.bgs_roi_align_nhwc_bwd_ex(ptrs, hs, ws, sc, L, N, float(finest_scale), capi.ptr(rois),
                                       K, C, ph, pw, sample_num, int(pool), capi.ptr(dout),
                                       capi.current_stream(rois.device))
    capi.check('bgs_roi_align_nhwc_bwd_ex', rc)
    return dfeats


class _RoIAlignFn(torch.autograd.Function):
    """Differentiable multi-level RoIAlign (w.r.t. the feature maps; RoIs carry no gradient, as
    in the reference: roi_align.py:52).  With ``base`` the pooled result is added into it in
    place (HTC semantic fusion: ``bbox_feats += pooled_semantic``); its gradient is ``dout``."""

    @staticmethod
    def forward(ctx, rois, strides, out_size, sample_num, finest_scale, pool, base, *feats):
        ctx.save_for_backward(rois)
        ctx.cfg = (tuple(strides), sample_num, finest_scale, pool, [tuple(f.shape) for f in feats],
                   base is not None)
        if base is not None:
            ctx.mark_dirty(base)
        return roi_align_nhwc([f.detach() for f in feats], rois, strides, out_size, sample_num,
                              finest_scale, pool=pool, out=base)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        rois, = ctx.saved_tensors
        strides, sample_num, finest_scale, pool, shapes, has_base = ctx.cfg
        dout = dout.contiguous()
        dfeats = [None] * len(shapes)
        if any(ctx.needs_input_grad[7:]):
            dfeats = [torch.zeros(s, dtype=torch.float32, device=dout.device) for s in shapes]
            roi_align_nhwc_bwd(dout, rois, dfeats, strides, sample_num, finest_scale, pool=pool)
        dbase = dout if has_base and ctx.needs_input_grad[6] else None
        return (None, None, None, None, None, None, dbase) + tuple(dfeats)


def roi_align_nhwc_autograd(feats, rois, featmap_strides, out_size=7, sample_num=2,
                            finest_scale=56, pool=1, add_to=None):
    """``add_to``: RoI features ``[K, out, out, C]`` that receive ``+= pooled`` in place."""
    ts = list(feats) + ([add_to] if add_to is not None else [])
    if torch.is_grad_enabled() and any(t.requires_grad for t in ts):
        return _RoIAlignFn.apply(rois, tuple(featmap_strides), out_size, sample_num, finest_scale,
                                 pool, add_to, *feats)
    return roi_align_nhwc(feats, rois, featmap_strides, out_size, sample_num, finest_scale,
                          pool=pool, out=add_to)


def resize_bilinear_nhwc(x, size):
    """``F.interpolate(x, size, mode='bilinear', align_corners=True)`` on an NHWC map."""
    _require_cuda(x)
    lib = capi.load()
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    N, H, W, C = x.shape
    Ho, Wo = int(size[0]), int(size[1])
    y = torch.empty((N, Ho, Wo, C), dtype=torch.float32, device=x.device)
    rc = lib.bgs_resize_bilinear_nhwc_f32(capi.ptr(x), capi.ptr(y), N, H, W, C, Ho, Wo, 1,
                                          capi.current_stream(x.device))
    capi.check('bgs_resize_bilinear_nhwc_f32', rc)
    return y


class _ResizeBilinearFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, size):
        ctx.in_shape = tuple(x.shape)
        return resize_bilinear_nhwc(x.detach(), size)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        lib = capi.load()
        N, H, W, C = ctx.in_shape
        dy = dy.contiguous()
        dx = torch.zeros(ctx.in_shape, dtype=torch.float32, device=dy.device)
        rc = lib.bgs_resize_bilinear_nhwc_bwd_f32(capi.ptr(dy), capi.ptr(dx), N, H, W, C,
                                                  dy.shape[1], dy.shape[2], 1,
                                                  capi.current_stream(dy.device))
        capi.check('bgs_resize_bilinear_nhwc_bwd_f32', rc)
        return dx, None


def resize_bilinear_nhwc_autograd(x, size):
    if torch.is_grad_enabled() and x.requires_grad:
        return _ResizeBilinearFn.apply(x, tuple(size))
    return resize_bilinear_nhwc(x, size)


def topk_sorted(rows, ks, kmax, inner=None):
    """Sorted top-k of several row sets in one launch set (``bgs_topk_sorted_f32``).

    ``rows``: list of L float32 tensors, one per set, ``[N, ...]`` with contiguous per-image
    blocks; ``ks``: list of L ints (``k_l <= kmax <= 4096``).  ``inner=None``: each per-image
    block is a flat row.  ``inner=A``: the block is ``[n_pix, C]`` and the row consists of the
    FIRST ``A`` channels of every pixel (``n_pix * A`` elements, index ``pix * A + a``) — the
    objectness logits inside the fused RPN head output, read in place.
    Returns ``(values, indices)`` ``[N, L, kmax]``: per (image, set) the ``min(k_l, len)`` largest
    entries in descending order and their row positions, zeros beyond.  ``N * L <= 64``."""
    import ctypes
    _require_cuda(*rows)
    lib = capi.load()
    N = rows[0].shape[0]
    L = len(rows)
    dev = rows[0].device
    ptrs, lens, kk, inn, pit = [], [], [], [], []
    for i in range(N):
        for r, k in zip(rows, ks):
            assert r.shape[0] == N and r.dtype == torch.float32 and r.is_contiguous()
            per = r[0].numel()
            ptrs.append(r.data_ptr() + i * per * 4)
            if inner is None:
                lens.append(per)
                inn.append(0)
                pit.append(0)
            else:
                C = int(r.shape[-1])
                lens.append(per // C * int(inner))
                inn.append(int(inner))
                pit.append(C)
            kk.append(int(k))
    P = N * L
    vals = torch.empty((N, L, kmax), dtype=torch.float32, device=dev)
    idx = torch.empty((N, L, kmax), dtype=torch.int64, device=dev)
    ws = _workspace(lib.bgs_topk_workspace_bytes(P, int(kmax)), dev)
    rc = lib.bgs_topk_sorted_f32((ctypes.c_void_p * P)(*ptrs), _c_int_array(lens), _c_int_array(kk),
                                 _c_int_array(inn) if inner is not None else None,
                                 _c_int_array(pit) if inner is not None else None,
                                 P, int(kmax), capi.ptr(vals), capi.ptr(idx), capi.ptr(ws),
                                 capi.current_stream(dev))
    capi.check('bgs_topk_sorted_f32', rc)
    return vals, idx


def nms_batched(boxes, counts, iou_thr, iou_mode=0, max_keep=0):
    """boxes ``[P,nmax,5]`` sorted by descending score per problem, counts ``[P]`` i32 ->
    ``keep [P,nmax]`` i32 (ascending indices), ``keep_count [P]`` i32.  No host sync."""
    _require_cuda(boxes, counts)
    lib = capi.load()
    boxes = _f32c(boxes)
    assert boxes.dim() == 3 and boxes.shape[2] == 5 and counts.dtype == torch.int32
    P, nmax, _ = boxes.shape
    dev = boxes.device
    keep = torch.empty((P, nmax), dtype=torch.int32, device=dev)
    keep_count = torch.empty((P,), dtype=torch.int32, device=dev)
    ws = _workspace(lib.bgs_nms_workspace_bytes(P, nmax), dev)
    rc = lib.bgs_nms_batched(capi.ptr(boxes), capi.ptr(counts), P, nmax, float(iou_thr),
                             int(iou_mode), int(max_keep), capi.ptr(keep), capi.ptr(keep_count),
                             capi.ptr(ws), capi.current_stream(dev))
    capi.check('bgs_nms_batched', rc)
    return keep, keep_count


def nms_gather(boxes, keep, keep_count):
    """``boxes [R,nmax,5]``, ``keep [R,nmax]`` / ``keep_count [R]`` of :func:`nms_batched` ->
    ``(kept [R,nmax,5], scores [R,nmax])``: the kept boxes in fixed-shape rows, score -1 in the slots
    past ``keep_count`` (``bgs_nms_gather``: one launch)."""
    _require_cuda(boxes, keep, keep_count)
    lib = capi.load()
    assert boxes.dtype == torch.float32 and boxes.is_contiguous() and boxes.dim() == 3 and boxes.shape[2] == 5
    assert keep.dtype == torch.int32 and keep.is_contiguous() and keep_count.dtype == torch.int32
    R, nmax, _ = boxes.shape
    out = torch.empty_like(boxes)
    sc = torch.empty((R, nmax), dtype=torch.float32, device=boxes.device)
    rc = lib.bgs_nms_gather(capi.ptr(boxes), capi.ptr(keep), capi.ptr(keep_count.contiguous()), R, nmax,
                            capi.ptr(out), capi.ptr(sc), capi.current_stream(boxes.device))
    capi.check('bgs_nms_gather', rc)
    return out, sc


def nms_merge_select(boxes, keep, keep_count, num_images, num, out=None):
    """The proposal tail in one launch (``bgs_nms_merge_select``): ``boxes [N*L,nmax,5]`` (rows sorted by
    descending score), ``keep`` / ``keep_count`` of :func:`nms_batched` -> ``(props [N,num,5], valid [N,num] bool)``:
    the ``num`` best kept boxes of each image over its L levels, in descending score order.  Every slot is
    written (ranks are a permutation under the kernel's total order; the slots past the kept total are zeroed).
    ``out``: optional pre-allocated ``(props f32, valid u8)`` pair (tests poison it to prove full coverage)."""
    _require_cuda(boxes, keep, keep_count)
    lib = capi.load()
    assert boxes.dtype == torch.float32 and boxes.is_contiguous() and boxes.dim() == 3 and boxes.shape[2] == 5
    assert keep.dtype == torch.int32 and keep.is_contiguous() and keep_count.dtype == torch.int32
    R, nmax, _ = boxes.shape
    N = int(num_images)
    assert R % N == 0
    if out is None:
        props = torch.empty((N, int(num), 5), dtype=torch.float32, device=boxes.device)
        valid = torch.empty((N, int(num)), dtype=torch.uint8, device=boxes.device)
    else:
        props, valid = out
        assert props.shape == (N, int(num), 5) and props.dtype == torch.float32 and props.is_contiguous()
        assert valid.shape == (N, int(num)) and valid.dtype == torch.uint8 and valid.is_contiguous()
    rc = lib.bgs_nms_merge_select(capi.ptr(boxes), capi.ptr(keep), capi.ptr(keep_count.contiguous()), N, R // N,
                                  nmax, int(num), capi.ptr(props), capi.ptr(valid),
                                  capi.current_stream(boxes.device))
    capi.check('bgs_nms_merge_select', rc)
    return props, valid.view(torch.bool)


def gather_boxes(flat, idx, scores):
    """``props [N,num,5] = flat[n, idx[n,j]]``, ``valid [N,num] (bool) = scores >= 0``
    (``bgs_gather_boxes``: one launch)."""
    _require_cuda(flat, idx, scores)
    lib = capi.load()
    assert flat.dtype == torch.float32 and flat.is_contiguous() and flat.dim() == 3 and flat.shape[2] == 5
    assert idx.dtype == torch.int64 and idx.is_contiguous() and scores.dtype == torch.float32 \
        and scores.is_contiguous() and idx.shape == scores.shape
    N, T, _ = flat.shape
    num = idx.shape[1]
    props = torch.empty((N, num, 5), dtype=torch.float32, device=flat.device)
    valid = torch.empty((N, num), dtype=torch.uint8, device=flat.device)
    rc = lib.bgs_gather_boxes(capi.ptr(flat), capi.ptr(idx), capi.ptr(scores), N, T, num, capi.ptr(props),
                              capi.ptr(valid), capi.current_stream(flat.device))
    capi.check('bgs_gather_boxes', rc)
    return props, valid.view(torch.bool)


# ----------------------------------------------------------------------------------------
# fused target assignment / RPN loss / proposal decode / RoI targets (csrc/det_targets.hip)
# ----------------------------------------------------------------------------------------
def _c_int_array(vals):
    import ctypes
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def _c_float_array(vals):
    import ctypes
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])


def _c_ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def iou_assign(boxes, gt_cat, gt_offsets, pos_iou_thr, neg_iou_thr, min_pos_iou=0.0, valid=None,
               shared_boxes=False, return_max_overlaps=False):
    """``MaxIoUAssigner`` for N images at once -> ``assigned [N,A]`` int32.

    boxes: ``[A, >=4]`` with ``shared_boxes=True`` (the same anchors for every image) or
    ``[N, A, >=4]``; gt_cat ``[sum G, 4]`` with python ``gt_offsets`` (len N+1); valid ``[N,A]`` u8."""
    _require_cuda(boxes, gt_cat, valid)
    lib = capi.load()
    boxes = boxes if boxes.dtype == torch.float32 and boxes.is_contiguous() else _f32c(boxes)
    gt_cat = _f32c(gt_cat)
    N = len(gt_offsets) - 1
    for i in range(N):
        # max_iou_assigner.py:76-77: an image without GT boxes is an error in the reference; the
        # fused kernel would silently mark every box of that image "ignore" (-1)
        if gt_offsets[i + 1] - gt_offsets[i] <= 0:
            raise ValueError('No gt or proposals (image %d of the batch has no GT box)' % i)
    if shared_boxes:
        A, bs = boxes.shape[0], boxes.shape[1]
        img_stride = 0
    else:
        assert boxes.shape[0] == N
        A, bs = boxes.shape[1], boxes.shape[2]
        img_stride = A * bs
    if isinstance(neg_iou_thr, (tuple, list)):
        lo, hi = neg_iou_thr
    else:
        lo, hi = 0.0, neg_iou_thr
    dev = boxes.device
    if valid is not None:
        assert valid.dtype == torch.uint8 and tuple(valid.shape) == (N, A) and valid.is_contiguous()
    assigned = torch.empty((N, A), dtype=torch.int32, device=dev)
    mo = torch.empty((N, A), dtype=torch.float32, device=dev) if return_max_overlaps else None
    ws = _workspace(lib.bgs_iou_assign_workspace_bytes(N, A, int(gt_offsets[-1])), dev)
    rc = lib.bgs_iou_assign(capi.ptr(boxes), img_stride, bs, capi.ptr(valid), capi.ptr(gt_cat),
                            _c_int_array(gt_offsets), N, A, float(pos_iou_thr), float(lo),
                            float(hi), float(min_pos_iou), capi.ptr(assigned), capi.ptr(mo),
                            capi.ptr(ws), capi.current_stream(dev))
    capi.check('bgs_iou_assign', rc)
    return (assigned, mo) if return_max_overlaps else assigned


def _rpn_loss_launch(level_outs, num_anchors, anchors, assigned, pos_mask, neg_mask, gt_cat,
                     gt_offsets, means, stds, beta, pos_weight, loss_weight_cls, loss_weight_bbox):
    lib = capi.load()
    L = len(level_outs)
    N = level_outs[0].shape[0]
    for o in level_outs:
        assert o.dtype == torch.float32 and o.is_contiguous() and o.shape[-1] == 5 * num_anchors
    hw = [int(o.shape[1] * o.shape[2]) for o in level_outs]
    A = sum(hw) * num_anchors
    assert tuple(assigned.shape) == (N, A) and assigned.dtype == torch.int32
    assert pos_mask.dtype == torch.uint8 and neg_mask.dtype == torch.uint8
    assert anchors.shape == (A, 4) and anchors.is_contiguous()
    dev = anchors.device
    lc = torch.empty((L,), dtype=torch.float32, device=dev)
    lb = torch.empty((L,), dtype=torch.float32, device=dev)
    nt = torch.empty((1,), dtype=torch.float32, device=dev)
    ws = _workspace(lib.bgs_rpn_loss_workspace_bytes(N, A, L), dev)
    rc = lib.bgs_rpn_loss(_c_ptr_array(level_outs), _c_int_array(hw), L, num_anchors,
                          capi.ptr(anchors), capi.ptr(assigned), capi.ptr(pos_mask),
                          capi.ptr(neg_mask), capi.ptr(gt_cat),
                          _c_int_array(gt_offsets), N, _c_float_array(means), _c_float_array(stds),
                          float(beta), float(pos_weight), float(loss_weight_cls),
                          float(loss_weight_bbox), capi.ptr(lc), capi.ptr(lb), capi.ptr(nt),
                          capi.ptr(ws), capi.current_stream(dev))
    capi.check('bgs_rpn_loss', rc)
    return lc, lb, nt


class _RpnLossFn(torch.autograd.Function):
    """Differentiable (w.r.t. the fused head outputs) form of the fused RPN loss."""

    @staticmethod
    def forward(ctx, cfg, anchors, assigned, pos_mask, neg_mask, gt_cat, *level_outs):
        outs = [o.detach() for o in level_outs]
        lc, lb, nt = _rpn_loss_launch(outs, cfg[0], anchors, assigned, pos_mask, neg_mask, gt_cat,
                                      *cfg[1:])
        ctx.cfg = cfg
        ctx.save_for_backward(anchors, assigned, pos_mask, neg_mask, gt_cat, nt, *outs)
        ctx.mark_non_differentiable(nt)
        return lc, lb, nt

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_lc, g_lb, _g_nt):
        anchors, assigned, pos_mask, neg_mask, gt_cat, nt = ctx.saved_tensors[:6]
        outs = list(ctx.saved_tensors[6:])
        (num_anchors, gt_offsets, means, stds, beta, pos_weight, w_cls, w_bbox) = ctx.cfg
        lib = capi.load()
        L = len(outs)
        N = outs[0].shape[0]
        hw = [int(o.shape[1] * o.shape[2]) for o in outs]
        douts = [torch.zeros_like(o) for o in outs]
        rc = lib.bgs_rpn_loss_grad(_c_ptr_array(outs), _c_ptr_array(douts), _c_int_array(hw), L,
                                   num_anchors, capi.ptr(anchors), capi.ptr(assigned),
                                   capi.ptr(pos_mask), capi.ptr(neg_mask), capi.ptr(gt_cat),
                                   _c_int_array(gt_offsets), N, _c_float_array(means),
                                   _c_float_array(stds), float(beta), float(pos_weight),
                                   float(w_cls), float(w_bbox), capi.ptr(nt),
                                   capi.ptr(_f32c(g_lc)), capi.ptr(_f32c(g_lb)),
                                   capi.current_stream(anchors.device))
        capi.check('bgs_rpn_loss_grad', rc)
        return (None,) * 6 + tuple(douts)


def rpn_loss(level_outs, num_anchors, anchors, assigned, pos_mask, neg_mask, gt_cat, gt_offsets,
             means, stds, beta, pos_weight=-1, loss_weight_cls=1.0, loss_weight_bbox=1.0):
    """Per-level RPN losses ``(loss_cls [L], loss_bbox [L], num_total_samples [1])`` from the
    fused head outputs ``[N,H,W,A+4A]`` of every level and the sampled-anchor masks.
    Differentiable w.r.t. ``level_outs`` when they require grad."""
    _require_cuda(anchors, assigned, pos_mask, neg_mask, gt_cat, *level_outs)
    pos_mask, neg_mask, gt_cat = pos_mask.contiguous(), neg_mask.contiguous(), _f32c(gt_cat)
    args = (num_anchors, tuple(int(o) for o in gt_offsets), tuple(means), tuple(stds), beta,
            pos_weight, loss_weight_cls, loss_weight_bbox)
    if torch.is_grad_enabled() and any(o.requires_grad for o in level_outs):
        return _RpnLossFn.apply(args, anchors, assigned, pos_mask, neg_mask, gt_cat, *level_outs)
    return _rpn_loss_launch(list(level_outs), num_anchors, anchors, assigned, pos_mask, neg_mask,
                            gt_cat, *args[1:])


_KEY_COUNTERS = {}


def _draw_counter(device, name):
    """Per (device, sampler) draw counter (device int64 ``[1]``), advanced with a tensor op per
    call.  One counter per sampler: the RPN sampler may run on a second stream next to the RoI
    sampler (detectors._rpn_forward_train), and a shared counter would be a cross-stream race."""
    key = (device.index, name)
    ctr = _KEY_COUNTERS.get(key)
    if ctr is None:
        ctr = torch.zeros(1, dtype=torch.int64, device=device)
        _KEY_COUNTERS[key] = ctr
    ctr.add_(1)
    return ctr


def random_keys(n, device):
    """``[n]`` int64 sampling keys in ``[0, 2^62)`` from the counter-based device RNG
    (``bgs_random_keys``).  A per-device draw counter is advanced with a tensor op per call, so
    the launch is hipGraph-replayable and still draws fresh keys every replay."""
    lib = capi.load()
    device = torch.device(device)
    ctr = _draw_counter(device, 'keys')
    out = torch.empty((n,), dtype=torch.int64, device=device)
    seed = (torch.initial_seed() * 0x9E3779B1 + 0x5851F42D) & 0xFFFFFFFFFFFFFFFF
    rc = lib.bgs_random_keys(seed, capi.ptr(ctr), int(n), capi.ptr(out),
                             capi.current_stream(device))
    capi.check('bgs_random_keys', rc)
    return out


def sample_pos_neg(assigned, num, pos_fraction, neg_pos_ub=-1):
    """RPN RandomSampler for a whole batch in one launch: ``assigned [N, A]`` int32 ->
    ``(pos_mask, neg_mask) [N, A]`` uint8 with the exact counts of base_sampler.py:56-73."""
    _require_cuda(assigned)
    lib = capi.load()
    assert assigned.dim() == 2 and assigned.dtype == torch.int32 and assigned.is_contiguous()
    N, A = assigned.shape
    dev = assigned.device
    ctr = _draw_counter(dev, 'pos_neg')
    pos = torch.empty((N, A), dtype=torch.uint8, device=dev)
    neg = torch.empty((N, A), dtype=torch.uint8, device=dev)
    seed = (torch.initial_seed() * 0x9E3779B1 + 0x2545F491) & 0xFFFFFFFFFFFFFFFF
    ws = _workspace(lib.bgs_sample_pos_neg_workspace_bytes(N), dev)
    rc = lib.bgs_sample_pos_neg(capi.ptr(assigned), N, A, int(num), float(pos_fraction),
                                float(neg_pos_ub), seed, capi.ptr(ctr), capi.ptr(pos), capi.ptr(neg),
                                capi.ptr(ws), capi.current_stream(dev))
    capi.check('bgs_sample_pos_neg', rc)
    return pos, neg


def sample_rois(assigned_list, num, pos_fraction, gt_counts=None):
    """RoI-head RandomSampler for a batch: per image ``assigned [A_n]`` int32 (candidates = GT
    boxes then proposals) -> ``inds [N, num]`` int64 (positives first), ``is_pos``, ``valid``
    ``[N, num]`` uint8.  One launch; ``A_n <= 4096``.  ``gt_counts`` (list of ints: the GT boxes in
    front of each image's candidates): also returns ``gt_ind [N, num]`` int32 (assigned gt index of
    every sampled row, -1 = none) and ``is_gt [N, num]`` uint8 from the same launch."""
    _require_cuda(*assigned_list)
    lib = capi.load()
    N = len(assigned_list)
    dev = assigned_list[0].device
    for a in assigned_list:
        assert a.dtype == torch.int32 and a.is_contiguous() and a.dim() == 1
    ctr = _draw_counter(dev, 'rois')
    inds = torch.empty((N, num), dtype=torch.int64, device=dev)
    is_pos = torch.empty((N, num), dtype=torch.uint8, device=dev)
    valid = torch.empty((N, num), dtype=torch.uint8, device=dev)
    seed = (torch.initial_seed() * 0x9E3779B1 + 0x68E31DA4) & 0xFFFFFFFFFFFFFFFF
    if gt_counts is not None:
        gt_ind = torch.empty((N, num), dtype=torch.int32, device=dev)
        is_gt = torch.empty((N, num), dtype=torch.uint8, device=dev)
        rc = lib.bgs_sample_rois_ex(_c_ptr_array(assigned_list),
                                    _c_int_array([int(a.numel()) for a in assigned_list]),
                                    _c_int_array([int(g) for g in gt_counts]), N, int(num),
                                    float(pos_fraction), seed, capi.ptr(ctr), capi.ptr(inds),
                                    capi.ptr(is_pos), capi.ptr(valid), capi.ptr(gt_ind), capi.ptr(is_gt),
                                    capi.current_stream(dev))
        capi.check('bgs_sample_rois_ex', rc)
        return inds, is_pos, valid, gt_ind, is_gt
    rc = lib.bgs_sample_rois(_c_ptr_array(assigned_list),
                             _c_int_array([int(a.numel()) for a in assigned_list]), N, int(num),
                             float(pos_fraction), seed, capi.ptr(ctr), capi.ptr(inds),
                             capi.ptr(is_pos), capi.ptr(valid), capi.current_stream(dev))
    capi.check('bgs_sample_rois', rc)
    return inds, is_pos, valid


def decode_proposals(level_outs, level_counts, num_anchors, anchors, top_idx, top_logit, img_hw,
                     means, stds, wh_ratio_clip=16 / 1000):
    """``[N,L,nmax,5]`` decoded + clamped proposals (score = sigmoid of the top logit)."""
    _require_cuda(anchors, top_idx, top_logit, *level_outs)
    lib = capi.load()
    L = len(level_outs)
    N, L2, nmax = top_idx.shape
    assert L2 == L and top_idx.dtype == torch.int64 and top_idx.is_contiguous()
    assert top_logit.dtype == torch.float32 and top_logit.is_contiguous()
    hw = [int(o.shape[1] * o.shape[2]) for o in level_outs]
    out = torch.empty((N, L, nmax, 5), dtype=torch.float32, device=anchors.device)
    flat_hw = []
    for h, w in img_hw:
        flat_hw += [int(h), int(w)]
    rc = lib.bgs_decode_proposals(_c_ptr_array(level_outs), _c_int_array(hw),
                                  _c_int_array(level_counts), L, num_anchors, capi.ptr(anchors),
                                  capi.ptr(top_idx), capi.ptr(top_logit), N, _c_int_array(flat_hw),
                                  _c_float_array(means), _c_float_array(stds),
                                  float(wh_ratio_clip), nmax, capi.ptr(out),
                                  capi.current_stream(anchors.device))
    capi.check('bgs_decode_proposals', rc)
    return out


def refine_boxes(rois, labels, bbox_pred, img_shapes, means, stds, wh_ratio_clip=16 / 1000):
    """Cascade stage hand-over (``regress_by_class`` + ``delta2bbox`` for all images) in one launch:
    ``rois [K,5]`` (image index, box), ``labels [K]`` int64 (ignored when ``bbox_pred`` has 4 columns:
    class-agnostic), ``bbox_pred [K, 4 or 4*classes]``, ``img_shapes`` = per-image ``(h, w)`` clip bounds
    -> ``[K,4]`` boxes."""
    _require_cuda(rois, labels, bbox_pred)
    lib = capi.load()
    K, cols = bbox_pred.shape
    assert rois.dtype == torch.float32 and bbox_pred.dtype == torch.float32
    assert rois.is_contiguous() and bbox_pred.is_contiguous() and tuple(rois.shape) == (K, 5)
    lab = None
    if cols > 4:
        assert labels.dtype == torch.int64 and labels.is_contiguous() and labels.numel() == K
        lab = labels
    out = torch.empty((K, 4), dtype=torch.float32, device=rois.device)
    hw = [int(v) for s_ in img_shapes for v in s_[:2]]
    rc = lib.bgs_refine_boxes(capi.ptr(rois), capi.ptr(lab), capi.ptr(bbox_pred), K, cols,
                              _c_int_array(hw), len(img_shapes), _c_float_array(means),
                              _c_float_array(stds), float(wh_ratio_clip), capi.ptr(out),
                              capi.current_stream(rois.device))
    capi.check('bgs_refine_boxes', rc)
    return out


def rcnn_targets(boxes_list, assigned_list, inds_list, valid_list, gt_labels_list, gt_cat,
                 gt_offsets, num, means, stds, pos_weight=-1):
    """Sampled RoIs -> ``rois [N*num,5]``, ``labels`` i64, ``label_weights``, ``bbox_targets``,
    ``bbox_weights`` (positives first per image, as bbox_target_single lays them out)."""
    _require_cuda(gt_cat, *boxes_list)
    lib = capi.load()
    N = len(boxes_list)
    dev = gt_cat.device
    for b, a, i, g in zip(boxes_list, assigned_list, inds_list, gt_labels_list):
        assert b.dtype == torch.float32 and b.is_contiguous() and b.dim() == 2
        assert a.dtype == torch.int32 and a.is_contiguous()
        assert i.dtype == torch.int64 and i.is_contiguous() and i.numel() == num
        assert g.dtype == torch.int64 and g.is_contiguous()
    valid_u8 = [None if v is None else v.to(torch.uint8).contiguous() for v in valid_list]
    rois = torch.empty((N * num, 5), dtype=torch.float32, device=dev)
    labels = torch.empty((N * num,), dtype=torch.int64, device=dev)
    lw = torch.empty((N * num,), dtype=torch.float32, device=dev)
    bt = torch.empty((N * num, 4), dtype=torch.float32, device=dev)
    bw = torch.empty((N * num, 4), dtype=torch.float32, device=dev)
    rc = lib.bgs_rcnn_targets(_c_ptr_array(boxes_list), _c_int_array([b.shape[1] for b in boxes_list]),
                              _c_ptr_array(assigned_list), _c_ptr_array(inds_list),
                              _c_ptr_array(valid_u8) if any(v is not None for v in valid_u8) else None,
                              _c_ptr_array(gt_labels_list), capi.ptr(_f32c(gt_cat)),
                              _c_int_array(gt_offsets), N, num, _c_float_array(means),
                              _c_float_array(stds), float(pos_weight), capi.ptr(rois),
                              capi.ptr(labels), capi.ptr(lw), capi.ptr(bt), capi.ptr(bw),
                              capi.current_stream(dev))
    capi.check('bgs_rcnn_targets', rc)
    return rois, labels, lw, bt, bw


def selftest_wave_reduce(values):
    """Runs the device self-test of the wave reduction primitive (64 floats in)."""
    _require_cuda(values)
    lib = capi.load()
    v = _f32c(values)
    assert v.numel() == 64
    out = torch.empty(4, dtype=torch.float32, device=v.device)
    capi.check('bgs_selftest_wave_reduce',
               lib.bgs_selftest_wave_reduce(capi.ptr(v), capi.ptr(out),
                                            capi.current_stream(v.device)))
    return out


# ----------------------------------------------------------------------------------------
# mask branch (csrc/mask_head.hip): targets, single-channel logits, fused BCE
# ----------------------------------------------------------------------------------------
def mask_target(gt_masks, rois, gt_inds, valid, mask_size):
    """``gt_masks``: per image a uint8 device tensor ``[G_n, H, W]``; ``rois [P, >=5]`` float
    (batch_ind, x1, y1, x2, y2); ``gt_inds [P]`` int32; ``valid [P]`` bool/uint8 or None ->
    ``[P, S, S]`` float targets (mask_target.py:16-38 incl. the cv2 INTER_LINEAR uint8 resize)."""
    import ctypes
    _require_cuda(rois, gt_inds, valid, *gt_masks)
    lib = capi.load()
    N = len(gt_masks)
    Hm, Wm = int(gt_masks[0].shape[1]), int(gt_masks[0].shape[2])
    for m in gt_masks:
        assert m.dtype == torch.uint8 and m.is_contiguous() and tuple(m.shape[1:]) == (Hm, Wm)
    rois = _f32c(rois)
    P = rois.shape[0]
    gt_inds = gt_inds.to(torch.int32).contiguous()
    v8 = None if valid is None else valid.to(torch.uint8).contiguous()
    out = torch.empty((P, mask_size, mask_size), dtype=torch.float32, device=rois.device)
    ptrs = (ctypes.c_void_p * N)(*[m.data_ptr() if m.numel() else None for m in gt_masks])
    ng = (ctypes.c_int * N)(*[int(m.shape[0]) for m in gt_masks])
    rc = lib.bgs_mask_target(ptrs, ng, N, Hm, Wm, capi.ptr(rois), int(rois.shape[1]),
                             capi.ptr(gt_inds), capi.ptr(v8), P, int(mask_size), capi.ptr(out),
                             capi.current_stream(rois.device))
    capi.check('bgs_mask_target', rc)
    return out


def mask_paste(probs, boxes, scale_factor, thr, img_h, img_w):
    """``FCNMaskHead.get_seg_masks`` without the RLE step (fcn_mask_head.py:156-176) in one launch
    (``bgs_mask_paste_u8``): ``probs [K, S, S]`` float = sigmoid of each detection's own class channel,
    ``boxes [K, >=4]`` -> dense ``uint8 [K, img_h, img_w]``: the box (divided by ``scale_factor``, truncated to int as
    the reference does) receives ``cv2.resize(prob, (w, h), INTER_LINEAR) > thr``, everything else is 0."""
    _require_cuda(probs, boxes)
    lib = capi.load()
    probs, boxes = _f32c(probs), _f32c(boxes)
    K, S, S2 = probs.shape
    assert S == S2 and boxes.dim() == 2 and boxes.shape[0] == K and boxes.shape[1] >= 4
    out = torch.empty((K, int(img_h), int(img_w)), dtype=torch.uint8, device=probs.device)
    if K == 0:
        return out
    rc = lib.bgs_mask_paste_u8(capi.ptr(probs), capi.ptr(boxes), int(boxes.shape[1]), K, S, float(scale_factor),
                               float(thr), int(img_h), int(img_w), capi.ptr(out),
                               capi.current_stream(probs.device))
    capi.check('bgs_mask_paste_u8', rc)
    return out


def mask_gt_logits(feat, weight, bias, labels):
    """``feat [P, pixels, C]``, ``weight [K, C]``, ``labels [P]`` -> ``[P, pixels]`` logits of each
    RoI's own class channel (no gradient: test-time / inspection path)."""
    _require_cuda(feat, weight, bias, labels)
    lib = capi.load()
    feat, weight = _f32c(feat), _f32c(weight)
    P, pix, C = feat.shape
    out = torch.empty((P, pix), dtype=torch.float32, device=feat.device)
    rc = lib.bgs_mask_gt_logits(capi.ptr(feat), capi.ptr(weight),
                                capi.ptr(None if bias is None else _f32c(bias)),
                                capi.ptr(labels.to(torch.int64).contiguous()), P, pix, C,
                                weight.shape[0], capi.ptr(out), capi.current_stream(feat.device))
    capi.check('bgs_mask_gt_logits', rc)
    return out


class _MaskBceFn(torch.autograd.Function):
    """loss[1] = mean over (valid RoIs x pixels) of BCE(<feat, W[label]> + b[label], target);
    gradients for feat, W, b come from the same launch (scaled by the upstream scalar after)."""

    @staticmethod
    def forward(ctx, feat, weight, bias, labels, target, valid):
        lib = capi.load()
        f, w = _f32c(feat), _f32c(weight)
        b = None if bias is None else _f32c(bias)
        P, pix, C = f.shape
        dev = f.device
        lab = labels.to(torch.int64).contiguous()
        tgt = _f32c(target)
        v8 = None if valid is None else valid.to(torch.uint8).contiguous()
        nvalid = torch.full((), float(P), device=dev) if v8 is None else v8.sum().float()
        norm = (1.0 / (nvalid.clamp(min=1.0) * pix)).reshape(1).contiguous()
        need = [feat.requires_grad, weight.requires_grad, bias is not None and bias.requires_grad]
        dfeat = torch.empty_like(f) if need[0] else None
        dw = torch.zeros_like(w) if (need[1] or need[2]) else None
        db = torch.zeros(w.shape[0], dtype=torch.float32, device=dev) if dw is not None else None
        part = torch.empty((lib.bgs_mask_bce_partials(P),), dtype=torch.float32, device=dev)
        rc = lib.bgs_mask_bce(capi.ptr(f), capi.ptr(w), capi.ptr(b), capi.ptr(lab), capi.ptr(tgt),
                              capi.ptr(v8), capi.ptr(norm), P, pix, C, w.shape[0], capi.ptr(part),
                              capi.ptr(dfeat), capi.ptr(dw), capi.ptr(db),
                              capi.current_stream(dev))
        capi.check('bgs_mask_bce', rc)
        ctx.grads = (dfeat, dw, db if bias is not None else None)
        return (part.sum() * norm[0]).reshape(1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        dfeat, dw, db = ctx.grads
        s = g.reshape(())
        return (None if dfeat is None else dfeat * s, None if dw is None else dw * s,
                None if db is None else db * s, None, None, None)


def mask_bce(feat, weight, bias, labels, target, valid=None):
    """Fused single-channel ``conv_logits`` + ``mask_cross_entropy`` -> loss ``[1]``."""
    _require_cuda(feat, weight, bias, labels, target, valid)
    assert feat.dim() == 3 and weight.dim() == 2 and feat.shape[2] == weight.shape[1]
    return _MaskBceFn.apply(feat, weight, bias, labels, target, valid)
