"""Training-step glue of the BAGS hot path (what the reference spreads over tools/train.py,
mmdet/apis/train.py and mmdet/core/utils/dist_utils.py — mmcv's Runner itself is not in the
reference tree).

* ``parse_losses``            mmdet/apis/train.py:17-34 — every key containing 'loss' is summed
  (lists are summed over their entries); no ``.item()`` per key here.
* ``select_training_param``   tools/train.py:49-91 — ``selectp``: 0 all, 1 ``bbox_head.fc_cls``
  only (every shipped BAGS config), 2 the whole bbox head, 3 the cascade stages' ``fc_cls``.
* ``allreduce_grads``         mmdet/core/utils/dist_utils.py:9-41 — ONE flat fp32 buffer, SUM
  all-reduce (RCCL over xGMI), divide by world size.  Launched asynchronously right after
  backward on the communication stream; with ``selectp=1`` the payload is 5.07 MB.
* ``DistOptimizerStep``       dist_utils.py:51-58 — zero_grad, backward, all-reduce, clip
  (max_norm=35, L2), SGD step.
* ``wrap_fp16_model`` / ``Fp16OptimizerStep``  mmdet/core/fp16/hooks.py:11-127 — the reduced-
  precision mode of BASELINE cfg[4] ("bf16"): conv / linear operands rounded to bf16 for the MFMA,
  fp32 accumulate; fp32 master weights, loss scaling.
"""
from collections import OrderedDict

import torch
import torch.distributed as dist


def _scalar(v):
    return v if v.dim() == 0 else v.mean()


def _total(vals):
    """Sum of 0-dim tensors as ONE stack + ONE reduction (``sum(...)`` is a chain of adds, each a
    launch forward and a node backward; the detectors produce 16-27 loss scalars per iteration)."""
    vals = [v.reshape(()) for v in vals]
    if len(vals) == 1:
        return vals[0]
    return torch.stack(vals).sum()


_SEGMENT_MATS = {}


def _segment_matrix(lengths, device):
    """[n_keys + 1, n_scalars] boolean matrix: row k selects the scalars of key k, the last row the scalars of
    every key containing 'loss' — cached per (structure, device)."""
    key = (lengths, str(device))
    m = _SEGMENT_MATS.get(key)
    if m is None:
        n = sum(l for l, _ in lengths)
        m = torch.zeros(len(lengths) + 1, n, dtype=torch.bool)
        off = 0
        for k, (l, is_loss) in enumerate(lengths):
            m[k, off:off + l] = True
            if is_loss:
                m[len(lengths), off:off + l] = True
            off += l
        m = m.to(device)
        if len(_SEGMENT_MATS) > 64:
            _SEGMENT_MATS.clear()
        _SEGMENT_MATS[key] = m
    return m


def parse_losses(losses):
    """mmdet/apis/train.py:24-47 (``parse_losses``): per entry the mean (of each element of a
    list, summed), the total over every key containing ``'loss'``.  Same values.  Every per-key sum
    AND the total come from ONE stack of all loss scalars, ONE product with a cached 0/1 segment matrix
    and ONE row-sum (three launches forward, whatever the number of keys) instead of a stack + sum
    per list-valued key and one more for the total."""
    names, flat, lengths = [], [], []
    for name, value in losses.items():
        if isinstance(value, torch.Tensor):
            vals = [_scalar(value)]
        elif isinstance(value, (list, tuple)):
            vals = [_scalar(v) for v in value]
        else:
            raise TypeError('{} is not a tensor or list of tensors'.format(name))
        names.append(name)
        flat.extend(v.reshape(()) for v in vals)
        lengths.append((len(vals), 'loss' in name))
    log_vars = OrderedDict()
    same = len({(v.device, v.dtype) for v in flat}) == 1 and flat and flat[0].is_floating_point()
    if not same or len(flat) == 1:
        off = 0
        for name, (l, _) in zip(names, lengths):
            log_vars[name] = _total(flat[off:off + l])
            off += l
        loss = _total([v for k, v in log_vars.items() if 'loss' in k])
        log_vars['loss'] = loss
        return loss, log_vars
    sums = _LossSumsFn.apply(tuple(lengths), *flat)
    for k, name in enumerate(names):
        log_vars[name] = sums[k]
    loss = sums[len(names)]
    log_vars['loss'] = loss
    return loss, log_vars


class _LossSumsFn(torch.autograd.Function):
    """Every per-key sum and the total of ``parse_losses`` from ONE stack, ONE masked select and ONE row sum
    (three launches forward whatever the number of keys).  `where`, not a product with a 0/1 matrix: 0 * NaN is
    NaN, and one diverged term (or a non-finite metric such as ``acc``, which the total excludes) must poison only
    its own key and the total it belongs to — as the reference's per-key sums do.

    Backward hands each scalar the gradient of the sums it belongs to WITHOUT arithmetic when only the total is
    differentiated (the training step): every 'loss' scalar receives the total's upstream gradient tensor
    itself.  With ``loss.backward(functional.unit_gradient(dev).reshape(()))`` — what ``DistOptimizerStep`` does —
    that tensor is the library's unit gradient, which the fused GroupSoftmax head recognises by identity: no launch
    between the root and the head's gradient buffers."""

    @staticmethod
    def forward(ctx, lengths, *flat):
        ctx.lengths = lengths
        ctx.set_materialize_grads(False)
        stacked = torch.stack([v.detach() for v in flat])
        seg = _segment_matrix(lengths, flat[0].device)
        sums = torch.where(seg, stacked.unsqueeze(0), 0.0).sum(1)
        return tuple(sums.unbind(0))

    @staticmethod
    def backward(ctx, *gs):
        g_total = gs[-1]
        out = [None]
        for k, (l, is_loss) in enumerate(ctx.lengths):
            g = gs[k]
            if is_loss and g_total is not None:
                g = g_total if g is None else g + g_total
            out.extend([g] * l)
        return tuple(out)


def select_training_param(model, selectp):
    """Returns the list of parameters left trainable."""
    if selectp == 0:
        return [p for p in model.parameters() if p.requires_grad]
    for p in model.parameters():
        p.requires_grad = False
    if selectp == 1:
        chosen = [model.bbox_head.fc_cls]
    elif selectp == 2:
        chosen = [model.bbox_head]
    elif selectp == 3:
        chosen = [h.fc_cls for h in model.bbox_head]
    else:
        raise ValueError('selectp must be 0, 1, 2 or 3')
    out = []
    for m in chosen:
        for p in m.parameters():
            p.requires_grad = True
            out.append(p)
    return out


def build_optimizer(params, cfg):
    """``optimizer = dict(type='SGD', lr, momentum, weight_decay[, paramwise_options])``
    (mmdet/apis/train.py:63-140).

    ``params``: a model (``nn.Module``, possibly wrapped with ``.module`` — the reference's own argument), an
    iterable of ``(name, parameter)`` pairs, or a plain list of parameters (plain branch only: the paramwise
    rules are keyed on parameter NAMES).

    Plain branch (:95-100): the global settings for every parameter that requires a gradient.
    ``paramwise_options`` (:101-140; none of ``configs/bags/*`` sets it): one param group per parameter —
    ``(bn|gn)(\\d+)?.(weight|bias)`` names get ``weight_decay * norm_decay_mult``; other ``.bias`` names get
    ``lr * bias_lr_mult`` and ``weight_decay * bias_decay_mult``; frozen parameters keep a group of their own
    with the global settings (the reference keeps them "to align with model.parameters()" for its fp16 master
    copies).  With more than one group ``DistOptimizerStep`` takes the ``clip_grad_norm_`` +
    ``torch.optim.SGD.step`` route instead of the fused single-group kernels (``_fused_sgd_eligible``)."""
    import re
    cfg = dict(cfg)
    paramwise = cfg.pop('paramwise_options', None)
    typ = cfg.pop('type')
    named = None
    if isinstance(params, torch.nn.Module):
        model = params.module if hasattr(params, 'module') else params
        named = list(model.named_parameters())
        plain = [p for _, p in named if p.requires_grad]
    else:
        params = list(params)
        if params and isinstance(params[0], (tuple, list)) and len(params[0]) == 2 \
                and isinstance(params[0][0], str):
            named = [(n, p) for n, p in params]
            plain = [p for _, p in named if p.requires_grad]
        else:
            plain = params
    if paramwise is None:
        return getattr(torch.optim, typ)(plain, **cfg)
    if not isinstance(paramwise, dict):
        raise AssertionError('paramwise_options must be a dict')
    if named is None:
        raise TypeError('paramwise_options are keyed on parameter names: pass the model or (name, parameter) pairs')
    base_lr = cfg['lr']
    base_wd = cfg.get('weight_decay', None)
    if 'bias_decay_mult' in paramwise or 'norm_decay_mult' in paramwise:
        assert base_wd is not None, 'weight_decay must be given when a decay multiplier is'
    bias_lr_mult = paramwise.get('bias_lr_mult', 1.)
    bias_decay_mult = paramwise.get('bias_decay_mult', 1.)
    norm_decay_mult = paramwise.get('norm_decay_mult', 1.)
    groups = []
    for name, param in named:
        group = {'params': [param]}
        if param.requires_grad:
            if re.search(r'(bn|gn)(\d+)?.(weight|bias)', name):
                if base_wd is not None:
                    group['weight_decay'] = base_wd * norm_decay_mult
            elif name.endswith('.bias'):
                group['lr'] = base_lr * bias_lr_mult
                if base_wd is not None:
                    group['weight_decay'] = base_wd * bias_decay_mult
        groups.append(group)
    return getattr(torch.optim, typ)(groups, **cfg)


_EXCHANGE_AT_ONE = [False]


def exchange_at_world_size_one(flag):
    """Test hook: run the gradient all-reduce even in a 1-rank process group, so that the collective's
    place in the step (and in a captured hipGraph of it) can be exercised on a single GPU."""
    prev = _EXCHANGE_AT_ONE[0]
    _EXCHANGE_AT_ONE[0] = bool(flag)
    return prev


def allreduce_grads(params, world_size, async_op=False):
    """Flatten -> one SUM all-reduce -> /world_size -> unflatten (dist_utils.py:9-41)."""
    grads = [p.grad for p in params if p.requires_grad and p.grad is not None]
    if (world_size == 1 and not _EXCHANGE_AT_ONE[0]) or not grads:
        return None
    flat = torch.cat([g.reshape(-1) for g in grads])
    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op)

    def finish():
        if work is not None and async_op:
            work.wait()
        flat.div_(world_size)
        off = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n

    if async_op:
        return finish
    finish()
    return None


class OverlappedGradExchange(object):
    """Bucketed gradient all-reduce overlapped with backward (what the north star asks of the
    `selectp = 0` data-parallel path; the reference's own hook exchanges one flat buffer AFTER
    backward, dist_utils.py:9-41 — same result: mean over ranks of every gradient).

    Parameters are grouped, in reverse registration order (= the order backward produces their
    gradients), into buckets of ``bucket_bytes``; a post-accumulate-grad hook on every parameter
    counts its bucket down and, when the bucket is complete, packs it into a flat buffer and
    launches ONE asynchronous SUM all-reduce (RCCL over xGMI: ring collectives are per-link
    bound, so the default 32 MB keeps each ring step bandwidth- rather than latency-dominated
    while still leaving >= 6 buckets of the 191 MB to hide behind the conv backward).
    ``finish()`` waits, divides by the world size and scatters the means back into ``p.grad``.
    """

    def __init__(self, params, world_size, bucket_bytes=32 << 20, process_group=None):
        self.params = [p for p in params if p.requires_grad]
        self.world_size = world_size
        self.group = process_group
        self.buckets = []
        cur, size = [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for p in b:
                self._bucket_of[id(p)] = bi
        self._pending = [len(b) for b in self.buckets]
        self._inflight = []
        self._handles = []
        if world_size > 1:
            for p in self.params:
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _launch(self, bi, had_grad=None):
        """Packs bucket ``bi`` and starts its all-reduce.  Behind the gradients ride ``len(bucket)`` flags — 1 where this
        rank HAD a gradient for the parameter, 0 where ``finish()`` put a zero placeholder — so that after the SUM every
        rank knows, per parameter, whether ANY rank contributed (the layout is the same on every rank whether the bucket
        was completed by the hooks or flushed by ``finish()``)."""
        bucket = self.buckets[bi]
        dev = bucket[0].grad.device
        if had_grad is None:
            flags = torch.ones(len(bucket), dtype=bucket[0].grad.dtype, device=dev)
        else:
            flags = torch.tensor(had_grad, dtype=bucket[0].grad.dtype, device=dev)
        flat = torch.cat([p.grad.reshape(-1) for p in bucket] + [flags])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._inflight.append((bi, flat, work))

    def _on_grad(self, p):
        bi = self._bucket_of[id(p)]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def finish(self):
        """Call after ``backward()``: flushes the buckets the hooks did not complete (a parameter without a gradient
        this iteration), waits for the collectives and writes the averaged gradients back.

        Every incomplete bucket is flushed on every rank — also one in which THIS rank has no gradient at all — with a
        zero placeholder for each missing gradient, so the ranks issue the same collectives with the same layout.  A
        placeholder is dropped again (``p.grad = None``: the reference's ``_allreduce_coalesced`` leaves such a
        parameter out, dist_utils.py:33-38, and SGD then skips it — no weight decay) only where NO rank contributed;
        where another rank did, every rank keeps the same averaged gradient, so the replicas cannot drift apart.
        (Ranks must still complete the same buckets through the hooks in the same order — the usual data-parallel
        contract; the reference has it too.)"""
        if self.world_size == 1:
            return
        filled = {}
        for bi, left in enumerate(self._pending):
            if left > 0:
                had = []
                for p in self.buckets[bi]:
                    had.append(0.0 if p.grad is None else 1.0)
                    if p.grad is None:
                        p.grad = torch.zeros_like(p)          # keeps the bucket's flat layout equal on every rank
                        filled[id(p)] = p
                self._launch(bi, had)
        for bi, flat, work in self._inflight:
            work.wait()
            bucket = self.buckets[bi]
            nb = len(bucket)
            contributed = None
            if any(id(p) in filled for p in bucket):
                contributed = flat[flat.numel() - nb:].tolist()   # host read only on this (rare) path
            flat.div_(self.world_size)
            off = 0
            for k, p in enumerate(bucket):
                n = p.numel()
                if contributed is not None and id(p) in filled and contributed[k] == 0.0:
                    p.grad = None                             # no rank had a gradient: as in the reference
                else:
                    p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
        self._inflight = []
        self._pending = [len(b) for b in self.buckets]

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []


def _fused_sgd_eligible(optimizer, params, grad_clip):
    """The fused clip + SGD kernels (csrc/optim.hip) cover what the BAGS configs use: ONE param group of
    torch.optim.SGD without dampening / nesterov / maximize, L2 clipping, fp32 CUDA parameters."""
    import os
    if os.environ.get('BGS_FUSED_SGD', '1') == '0' or type(optimizer) is not torch.optim.SGD:
        return False
    if len(optimizer.param_groups) != 1:
        return False
    g = optimizer.param_groups[0]
    if g.get('dampening', 0) != 0 or g.get('nesterov', False) or g.get('maximize', False):
        return False
    if grad_clip is not None and float(grad_clip.get('norm_type', 2)) != 2.0:
        return False
    return all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in params)


class FusedClipSGD(object):
    """``clip_grad_norm_`` + ``torch.optim.SGD.step`` for all trainable tensors as two launch phases
    (``bgs_sgd_clip_step``; the reference's ``DistOptimizerHook.after_train_iter`` tail, dist_utils.py:55-58).
    The momentum buffers ARE the wrapped optimizer's ``state[p]['momentum_buffer']`` (created as zeros
    — ``momentum * 0 + d`` is torch's first-step clone), so ``optimizer.state_dict()`` stays what a
    torch-driven run would save.  Parameters whose ``.grad`` is None are skipped, as torch does."""

    def __init__(self, optimizer, params, grad_clip=None):
        self.optimizer = optimizer
        self.params = list(params)
        self.max_norm = float(grad_clip['max_norm']) if grad_clip is not None else 0.0
        self.total_norm = None
        self._ws = None

    def step(self, grad_scale=1.0):
        import ctypes
        from . import capi
        lib = capi.load()
        group = self.optimizer.param_groups[0]
        ps = [p for p in self.params if p.grad is not None]
        if not ps:
            return
        dev = ps[0].device
        bufs = []
        for p in ps:
            st = self.optimizer.state[p]
            if st.get('momentum_buffer') is None:
                st['momentum_buffer'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            bufs.append(st['momentum_buffer'])
            if not p.grad.is_contiguous():
                p.grad = p.grad.contiguous()
        n = len(ps)
        arr = ctypes.c_void_p * n
        numel = (ctypes.c_longlong * n)(*[p.numel() for p in ps])
        wsb = lib.bgs_sgd_clip_workspace_bytes(numel, n)
        if self._ws is None or self._ws.numel() < wsb or self._ws.device != dev:
            self._ws = torch.empty(max(wsb, 1 << 16), dtype=torch.uint8, device=dev)
            self.total_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        rc = lib.bgs_sgd_clip_step(arr(*[p.data_ptr() for p in ps]), arr(*[p.grad.data_ptr() for p in ps]),
                                   arr(*[b.data_ptr() for b in bufs]), numel, n, self.max_norm,
                                   float(grad_scale), float(group['lr']), float(group.get('momentum', 0.0)),
                                   float(group.get('weight_decay', 0.0)), capi.ptr(self._ws),
                                   self._ws.numel(), capi.ptr(self.total_norm), capi.current_stream(dev))
        capi.check('bgs_sgd_clip_step', rc)
        _bump_versions(ps)


def _bump_versions(tensors):
    """The fused kernels write the parameters through raw pointers, which autograd's version counters do not
    see; every cache in this package that is keyed on ``Parameter._version`` (split bf16 planes, folded BN,
    permuted fc1 weights, the mask / semantic / deconv head folds) must notice the update exactly as it would
    after ``torch.optim.SGD.step()``'s in-place ops.  ``increment_version`` is bookkeeping only — no launch,
    legal under hipGraph capture (both forms below)."""
    try:
        torch.autograd.graph.increment_version(tensors)
    except (AttributeError, TypeError):                      # older torch: the private per-tensor form
        for t in tensors:
            torch._C._increment_version(t)


class DistOptimizerStep(object):
    """One optimizer step with the reference's hook order (dist_utils.py:51-58)."""

    def __init__(self, params, optimizer, grad_clip=None, world_size=1, overlap=None,
                 bucket_bytes=32 << 20):
        self.params = list(params)
        self.optimizer = optimizer
        self.grad_clip = grad_clip
        self.world_size = world_size
        # overlap the exchange with backward when there is something to hide it behind: more
        # than one bucket worth of gradients (selectp = 0: 191 MB; selectp = 1: 5 MB -> flat)
        nbytes = sum(p.numel() * p.element_size() for p in self.params)
        if overlap is None:
            overlap = world_size > 1 and nbytes > bucket_bytes
        self.overlap = OverlappedGradExchange(self.params, world_size, bucket_bytes) \
            if (overlap and world_size > 1) else None
        # clip + SGD as two launch phases over all tensors (csrc/optim.hip) where the configuration
        # allows; the torch foreach path otherwise (and under BGS_FUSED_SGD=0)
        self.fused = FusedClipSGD(optimizer, self.params, grad_clip) \
            if _fused_sgd_eligible(optimizer, self.params, grad_clip) else None

    def _clip_and_step(self, loss_scale=1.0):
        """``loss_scale``: the factor the loss was multiplied by before backward (1: none).  The torch path
        divides the gradients by it, as the reference does (hooks.py:72-76); the fused kernel multiplies by its
        reciprocal — the same bits for the power-of-two scales the reference's configs use, one rounding apart
        otherwise."""
        if self.fused is not None:
            self.fused.step(1.0 / loss_scale if loss_scale != 1.0 else 1.0)
            return
        if loss_scale != 1.0:
            grads = [p.grad for p in self.params if p.grad is not None]
            if grads:
                torch._foreach_div_(grads, float(loss_scale))
        if self.grad_clip is not None:
            torch.nn.utils.clip_grad_norm_(self.params, self.grad_clip['max_norm'],
                                           self.grad_clip.get('norm_type', 2))
        self.optimizer.step()

    def exchange_and_update(self):
        """all-reduce -> clip -> step, for gradients already produced by backward."""
        if self.overlap is not None:
            self.overlap.finish()          # buckets were launched from the backward hooks
        else:
            allreduce_grads(self.params, self.world_size)
        self._clip_and_step()

    def __call__(self, loss):
        self.optimizer.zero_grad(set_to_none=False)
        backward_unit(loss)
        self.exchange_and_update()


class TrunkPipeline(object):
    """Software pipeline of a training loop whose trunk is FROZEN (the shipped ``selectp = 1`` / ``3``,
    tools/train.py:49-57: only ``fc_cls`` trains).  The trunk of the batches AHEAD — in ``depth`` - 1 pieces, each on
    its own HIP stream — runs while the current batch's RPN losses, proposal / NMS / target chain (~0.4 ms of
    single-workgroup launches during which the chip idles), RoI heads, GroupSoftmax loss, backward, gradient exchange
    and optimizer step run on the main stream:

        depth 2:  extract_feat(i + 1)                                          | heads(i)
        depth 3:  backbone(i + 2)            | neck(i + 1)                     | heads(i)
        depth 4:  stem .. layer2 (i + 3)     | layer3 .. layer4 (i + 2) | neck(i + 1) | heads(i)
        depth 5 / 6: the residual stages in three / four groups

    (the small-grid launches of the deep stages — 168 / 96 workgroups on 256 CUs — and the latency-bound head chain
    fill each other's idle CUs: 6.22 ms sequential -> 6.06 / 5.78 / .. ms per step, profiles/r9h).  The features do not
    depend on anything the optimizer updates, so every step computes exactly what the sequential loop computes — same
    losses, same weights, bit for bit (tests/test_gpu_e2e.py) — and every loop iteration still holds one pass of every
    piece of the trunk and one head pass.

        pipe = TrunkPipeline(model, depth=3)
        for img in first `pipe.depth - 1` batches: pipe.push(img)        # prologue
        for i, batch in enumerate(loader):          # a loader that hands out images depth - 1 batches ahead
            feats = pipe.take()
            pipe.push(images_of_batch[i + pipe.depth - 1])
            losses = model(batch.img, batch.meta, return_loss=True, ..., feats=feats)
            ...backward, DistOptimizerStep...

    Raises if the trunk trains (``selectp = 0``): its features then depend on the previous step's update.

    Test time (``inference=True``): the same pieces ahead of ``simple_test(img, meta, feats=pipe.take())`` — the next
    image's trunk beside this image's RPN / NMS / RoI head / 1230-class NMS and the D2H copy of its result."""

    def __init__(self, model, depth=2, lane=3, inference=False):
        # ``inference=True``: a test loop (``model(img, meta, return_loss=False, feats=pipe.take())``) — nothing updates
        # the parameters between batches, so the frozen-trunk condition does not apply
        if not inference and not model.trunk_is_frozen():
            raise ValueError('TrunkPipeline needs a frozen backbone / neck (selectp = 1 or 3); with a trainable trunk '
                             'the next batch\'s features depend on this step\'s update')
        bb = model.backbone
        nst = len(getattr(bb, 'res_layers', ()))
        split = model.with_neck and hasattr(bb, 'forward_partial') and nst >= 2
        if depth >= 3 and not split:
            depth = 2
        depth = max(2, min(int(depth), nst + 2 if split else 2))
        if depth == 2:
            self.stages = [model.extract_feat]
        elif depth == 3:
            self.stages = [bb, model.neck]
        else:
            # depth - 2 contiguous groups of residual stages (4: stem .. layer2 | layer3 .. layer4; 6: one stage each)
            k = depth - 2
            cuts = [(nst * g + k - 1) // k for g in range(k + 1)]          # 0 = cuts[0] < ... < cuts[k] = nst
            self.stages = []
            for g in range(k):
                first, last = cuts[g], cuts[g + 1] - 1

                def piece(st, first=first, last=last, final=(g == k - 1)):
                    if first == 0:
                        r = bb.forward_partial(st, -1, last)
                    else:
                        r = bb.forward_partial(st[0], first, last, st[1])
                    return r[1] if final else r
                self.stages.append(piece)
            self.stages.append(model.neck)
        self.model = model
        self.depth = depth
        self.lane = lane
        self._slots = [None] * len(self.stages)                 # slot j: (fork, output) of stage j for some batch
        self._device = next(model.parameters()).device          # (also before the first image: push(None) on an empty pipeline)

    @staticmethod
    def _record(obj, stream):
        """``record_stream`` on every tensor of a nest: the caching allocator then hands a block back to its own
        stream's pool only after the work ``stream`` has queued at the time of the free.  (A piece's input lives in
        the PREVIOUS piece's pool and is read on this piece's stream; holding a reference until some later join is
        not enough here, because the stream that reuses the block never waits for this one.)"""
        if torch.is_tensor(obj):
            if obj.is_cuda:
                obj.record_stream(stream)
        elif isinstance(obj, (tuple, list)):
            for o in obj:
                TrunkPipeline._record(o, stream)

    def _launch(self, j, inp, producer):
        from . import functional as BF
        with torch.no_grad():
            with BF.forked(self._device, lane=self.lane + j) as fk:
                if producer is not None:
                    producer.join()                             # (inside the block: THIS stage's stream waits for it)
                self._record(inp, fk.side)
                out = self.stages[j](inp)
        return fk, out

    def push(self, img):
        """Feed the next batch's images: every piece of the trunk advances by one batch (launched deepest first, each
        on its own stream, ordered after everything enqueued on the current stream so far).  ``img = None`` (the
        loader has run dry): the batches in flight advance, nothing new enters."""
        from . import functional as BF
        if img is not None:
            self._device = img.device
        elif all(s is None for s in self._slots):
            return                               # nothing in flight and nothing new: the process-wide switches stay as they are
        if not BF._PIPELINE_ACTIVE[0]:
            # (armed only when a slot is or becomes occupied — a trailing push(None) after the last take() used
            #  to re-arm the switches with nothing in flight and leave them set until drain() / __del__)
            BF._PIPELINE_ACTIVE[0] = id(self)    # (functional.level_fork_enabled: no forks inside the pieces or beside the heads; the value names the owner)
            # ... and without forks beside them the P2 halo convs take the whole-rounds schedule of variant 7
            # (bit-identical), unless the caller has chosen a mode of his own
            self._wide_prev = BF.set_halo_wide(1)
            if self._wide_prev >= 0:
                BF.set_halo_wide(self._wide_prev)
        n = len(self.stages)
        assert self._slots[n - 1] is None, 'take() the finished features first'
        for j in range(n - 1, 0, -1):
            if self._slots[j - 1] is not None:
                fk, out = self._slots[j - 1]
                self._slots[j] = self._launch(j, out, fk)
                self._slots[j - 1] = None
        if img is not None:
            self._slots[0] = self._launch(0, img, None)
        elif all(s is None for s in self._slots):
            self._deactivate()

    def prefetch(self, img):              # depth 2: push == prefetch
        self.push(img)

    def take(self):
        """The features of the oldest batch in flight, with the current stream ordered after their producers."""
        from . import functional as BF
        n = len(self.stages)
        assert self._slots[n - 1] is not None, 'push() %d batches first' % n
        fk, feats = self._slots[n - 1]
        fk.join()
        self._record(feats, torch.cuda.current_stream(self._device))     # (allocated on the last piece's stream, read here)
        self._slots[n - 1] = None
        if all(s is None for s in self._slots):
            self._deactivate()
        return feats

    def __del__(self):
        try:                     # (a pipeline dropped with batches in flight must not leave the process-wide switches set)
            self._deactivate()
        except Exception:
            pass

    def _deactivate(self):
        from . import functional as BF
        if BF._PIPELINE_ACTIVE[0] == id(self):       # (only the pipeline that set the switches clears them: a dropped
            BF._PIPELINE_ACTIVE[0] = 0               #  pipeline's __del__ must not touch a live one's)
            if getattr(self, '_wide_prev', 0) < 0:
                BF.set_halo_wide(-1)

    def drain(self):
        """Join and drop whatever is in flight (end of the loop)."""
        from . import functional as BF
        for j, s in enumerate(self._slots):
            if s is not None:
                s[0].join()
                self._slots[j] = None
        self._deactivate()


def backward_unit(loss):
    """``loss.backward()`` with the library's cached unit gradient as the root gradient (same values; saves
    autograd's ones_like fill, and the loss edges of this package pass it down by identity — see
    ``functional.unit_gradient``)."""
    if loss.is_cuda and loss.dtype == torch.float32 and loss.numel() == 1:
        from . import functional as BF
        loss.backward(BF.unit_gradient(loss.device).reshape(loss.shape))
    else:
        loss.backward()


def wrap_fp16_model(model, math='bf16'):
    """The reference converts the model with ``model.half()`` and keeps an fp32 copy of the weights
    in the optimizer (hooks.py:40-44,86-94); normalisation layers and every ``@force_fp32`` loss
    (the GroupSoftmax / box / mask losses, gs_bbox_head_with0.py:147) stay in fp32.

    Here the SAME arithmetic comes from the kernels instead of from tensor dtypes: under
    ``conv_math = 'bf16'`` every conv / linear rounds both operands to bf16 on their way into the
    matrix cores (``planes = 1`` of csrc/conv_bfx.hip) and accumulates in fp32.  Parameters and
    activations stay fp32 in HBM, so the parameters ARE the fp32 master weights (no copy, no
    copy-back), eval-mode BatchNorm is folded in fp32, and the losses see fp32 logits — what
    ``patch_norm_fp32`` / ``force_fp32`` arrange in the reference.

    The mode is scoped to THIS model, as ``model.half()`` is: forward pre / post hooks switch the
    process-wide conv arithmetic for the duration of ``model(...)`` only, and every conv autograd
    node replays its backward in the mode its forward ran in — another model of the process (an
    fp32 teacher, an evaluation copy) keeps its own arithmetic.  Returns the conv math that stays
    in force OUTSIDE the model (``unwrap_fp16_model`` removes the hooks).

    **Scope, read this:** the hooks fire on ``model(...)`` (``nn.Module.__call__``) ONLY.  Calling
    ``model.forward_train`` / ``simple_test`` / ``extract_feat`` or a sub-module directly does NOT enter
    the bf16 mode (it runs in whatever ``functional.conv_math()`` is in force, normally bf16x6) —
    wrap such calls in ``with fp16_scope(model): ...``.  The return value is the OUTSIDE mode, not a
    "previous mode" to restore: nothing needs restoring (round-2 callers that passed it back to
    ``set_conv_math`` get the same end state)."""
    from . import functional as BF
    assert math in ('bf16',), math
    unwrap_fp16_model(model)
    model._conv_math = math
    stack = []

    def _enter(_m, _args, _kwargs=None):
        stack.append(BF.set_conv_math(model._conv_math))

    def _exit(_m, _args, _out):
        if stack:
            BF.set_conv_math(stack.pop())

    model._conv_math_hooks = (model.register_forward_pre_hook(_enter),
                              model.register_forward_hook(_exit, always_call=True))
    return BF.conv_math()


class fp16_scope(object):
    """``with fp16_scope(model): feats = model.extract_feat(img)`` — the arithmetic mode of a model wrapped by
    :func:`wrap_fp16_model` for code that calls its methods or sub-modules directly instead of ``model(...)``.
    A model that is not wrapped leaves the mode alone.  Restored on exit, also on error; re-entrant."""

    def __init__(self, model):
        self.math = getattr(model, '_conv_math', None)
        self._prev = []

    def __enter__(self):
        from . import functional as BF
        self._prev.append(BF.set_conv_math(self.math) if self.math else None)
        return self

    def __exit__(self, *exc):
        from . import functional as BF
        prev = self._prev.pop()
        if prev is not None:
            BF.set_conv_math(prev)
        return False


def unwrap_fp16_model(model):
    for h in getattr(model, '_conv_math_hooks', ()):
        h.remove()
    model._conv_math_hooks = ()
    model._conv_math = None


class Fp16OptimizerStep(DistOptimizerStep):
    """``Fp16OptimizerHook.after_train_iter`` (hooks.py:58-83): scale the loss, backward, all-reduce
    the gradients of the fp32 weights, scale them back, clip, step.  (bf16 has the exponent range
    of fp32, so the scale is not needed for range; it is kept for the hook's contract and is exact
    for powers of two.)"""

    def __init__(self, params, optimizer, grad_clip=None, world_size=1, loss_scale=512.0, **kw):
        super().__init__(params, optimizer, grad_clip=grad_clip, world_size=world_size, **kw)
        self.loss_scale = float(loss_scale)

    def exchange_and_update(self):
        if self.overlap is not None:
            self.overlap.finish()
        else:
            allreduce_grads(self.params, self.world_size)
        self._clip_and_step(self.loss_scale)

    def __call__(self, loss):
        self.optimizer.zero_grad(set_to_none=False)
        (loss * self.loss_scale).backward()
        self.exchange_and_update()
