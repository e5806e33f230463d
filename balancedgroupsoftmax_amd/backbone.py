"""ResNet trunk + FPN neck behind the reference's ``BACKBONES`` / ``NECKS`` registry keys.

* ``ResNet``  mmdet/models/backbones/resnet.py:332-542 (Bottleneck :86-266, style='pytorch':
  stride on the 3x3 conv; ``frozen_stages``; ``norm_eval=True`` => BatchNorm always uses its
  running statistics, :535-542)
* ``FPN``     mmdet/models/necks/fpn.py:9-141

Parameter *names and shapes* are the reference's (``conv1.weight``, ``layer2.0.bn1.running_var``,
``layer1.0.downsample.0.weight``, ``lateral_convs.1.conv.bias`` ...), so reference checkpoints
load unchanged — the ``nn.Conv2d`` / ``nn.BatchNorm2d`` objects are parameter containers only.
The arithmetic runs in the fp32-MFMA implicit-GEMM kernel (csrc/conv_igemm.hip):

* activations are NHWC end to end (channels = GEMM K axis contiguous);
* eval-mode BatchNorm is folded into the conv weights/bias (legal because ``norm_eval=True``),
  weights are re-laid-out once to ``[Cout, R, S, Cin]``; the fold is cached and refreshed
  when a parameter's version counter changes (load_state_dict, optimizer step);
* bias, residual add and ReLU are fused into the conv epilogue; the FPN top-down
  ``lateral + nearest_2x(upper)`` is the epilogue's upsampled-residual mode.

Training modes (tools/train.py:49-57): the shipped BAGS configs train only
``bbox_head.fc_cls`` (``selectp=1``) — then every conv here is a plain forward launch on cached
folded weights.  With trainable trunk parameters (``selectp=0``) the fold is recomputed per step
as differentiable tensor code (so autograd unfolds ``dW', db'`` into the conv weight and the BN
affine parameters) and every conv records ``functional._ConvFn`` (dgrad / wgrad kernels,
csrc/conv_igemm.hip + csrc/conv_wgrad.hip).  The stem + max-pool have no backward here: they are
frozen in every BAGS config (``frozen_stages=1``); with ``frozen_stages=-1`` the stem trains too
(weight / BN gradients of the 7x7 conv, max-pool backward).
"""
import contextlib

import torch
import torch.nn as nn

from . import functional as BF
from .registry import BACKBONES, NECKS


def _trainable(*modules):
    """True when gradients are being recorded and some parameter of ``modules`` wants one."""
    return torch.is_grad_enabled() and any(
        p.requires_grad for m in modules if m is not None for p in m.parameters())


def _fold_conv_bn(conv, bn, pad_cin_to=None):
    """-> (w [Cout,R,S,Cin] contiguous, bias [Cout]) with the eval-mode BN folded in.  On the GPU the
    fold is ONE launch (``functional.fold_conv_bn``: csrc/bn_fold.hip) and, when the conv / BN
    parameters are being trained, ONE more in backward (dW, dgamma, dbeta, dbias from dW', db');
    CPU tensors (module construction / state-dict tests: the product never computes there) take the
    tensor-op form of the same formula."""
    w = conv.weight
    if w.is_cuda:
        ctx = contextlib.nullcontext() if _trainable(conv, bn) else torch.no_grad()
        with ctx:
            if bn is not None:
                return BF.fold_conv_bn(w, conv.bias, bn.weight, bn.bias, bn.running_mean,
                                       bn.running_var, bn.eps, cin_padded=pad_cin_to)
            return BF.fold_conv_bn(w, conv.bias, cin_padded=pad_cin_to)
    ctx = contextlib.nullcontext() if _trainable(conv, bn) else torch.no_grad()
    with ctx:
        w = conv.weight.float()
        cout = w.shape[0]
        if bn is not None:
            scale = bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps)
            shift = bn.bias.float() - bn.running_mean.float() * scale
            w = w * scale.view(-1, 1, 1, 1)
            b = shift if conv.bias is None else shift + conv.bias.float() * scale
        else:
            b = conv.bias.float() if conv.bias is not None else w.new_zeros(cout)
        w = w.permute(0, 2, 3, 1)
        if pad_cin_to is not None and w.shape[3] < pad_cin_to:
            w = torch.nn.functional.pad(w, (0, pad_cin_to - w.shape[3]))
        return w.contiguous(), b.contiguous()


def _tensor_slots(mods):
    """(registration dict, name) of every parameter / buffer below the modules ``mods`` (direct walk of the registration
    dicts), and those dicts themselves (parameters, buffers and sub-modules of every module seen, empty ones included)."""
    slots, dicts, moddicts = [], [], []
    stack = list(mods)[::-1]
    while stack:
        m = stack.pop()
        for d in (m._parameters, m._buffers):
            dicts.append(d)
            for k in d:
                slots.append((d, k))
        dicts.append(m._modules)
        moddicts.append(m._modules)
        stack.extend(m._modules.values())
    return slots, dicts, moddicts


class _FoldCache(object):
    """Folded weights keyed by the identity, version counter and storage address of the tensors they were built from.

    ``get(mods, build)``: ``mods`` is a module or a tuple of modules.  The host cost of the check is on the launch path of
    every frozen layer (a quarter of the launching thread's time once, a tenth before round 5), so it reads the CURRENT
    tensor objects through the registration dicts listed once (a re-registered parameter or buffer is a new object under
    the same name: seen), compares (id, _version, data_ptr) and re-lists the dicts only when a name disappears, the number
    of registered names (parameters, buffers, sub-modules) changes, a sub-module is REPLACED under its name (the identities
    of the registered sub-modules are part of the cheap check: ADVICE r5 — the slots would otherwise keep pointing at the
    old module's tensors) or another set of modules is passed."""

    def __init__(self):
        self.key = None
        self.data = None
        self.mods = None
        self.slots = None
        self.count = -1
        self.fence = None

    def _list(self, mods):
        self.mods = tuple(id(m) for m in mods)
        self.slots, self.dicts, self.moddicts = _tensor_slots(mods)
        self.count = sum(len(d) for d in self.dicts)
        self.subs = self._sub_ids()
        self.key = None

    def _sub_ids(self):
        return [id(v) for d in self.moddicts for v in d.values()]

    def get(self, module, build):
        mods = module if isinstance(module, tuple) else (module,)
        if self.slots is None or self.mods != tuple(id(m) for m in mods) or \
                self.count != sum(len(d) for d in self.dicts) or self.subs != self._sub_ids():
            self._list(mods)
        try:
            tensors = [d[k] for d, k in self.slots]
        except KeyError:                                   # a name was removed: list again
            self._list(mods)
            tensors = [d[k] for d, k in self.slots]
        grad = torch.is_grad_enabled()
        key = []
        for t in tensors:
            if t is None:
                continue
            if grad and t.requires_grad:
                return build()       # trained parameters: the fold must be on this step's tape
            key.append((id(t), t._version, t.data_ptr()))
        if key != self.key:
            self.data = build()
            self.key = key
            # built (torch ops) on the current stream — possibly a pipeline piece's side stream — and read from whatever
            # stream the module runs on later: the first cross-stream use waits for the build (functional.CacheFence)
            self.fence = BF.CacheFence() if any(t is not None and t.is_cuda for t in tensors) else None
        elif self.fence is not None:
            self.fence.wait()
            if self.fence.event is None:
                self.fence = None
        return self.data


def cached_fold(conv, bn=None, pad_cin_to=None):
    """:func:`_fold_conv_bn` through a per-module :class:`_FoldCache` (kept on ``conv``): a frozen layer is
    folded — and, downstream, split into its bf16 planes — ONCE per parameter version instead of on every
    forward call (the mask / semantic heads call their convs 24+ times per HTC iteration); a trained layer is
    folded on this step's tape, as before."""
    cache = conv.__dict__.get('_bgs_fold_cache')
    if cache is None:
        cache = conv.__dict__['_bgs_fold_cache'] = _FoldCache()
    mods = (conv, bn) if bn is not None else (conv,)
    return cache.get(mods, lambda: _fold_conv_bn(conv, bn, pad_cin_to=pad_cin_to))


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=False, groups=1, base_width=4):
        super().__init__()
        # ResNeXt (resnext.py:19-22): width = floor(planes * base_width / 64) * groups
        width = planes if groups == 1 else (planes * base_width // 64) * groups
        self.groups, self.width = groups, width
        self.conv1 = nn.Conv2d(inplanes, width, 1, stride=1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1, groups=groups, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.stride = stride
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential(
                nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * 4))
        self._cache = _FoldCache()

    def folded(self):
        return self._cache.get(self, self._build_fold)

    def _build_fold(self):
        f = dict(c1=_fold_conv_bn(self.conv1, self.bn1), c2=_fold_conv_bn(self.conv2, self.bn2),
                 c3=_fold_conv_bn(self.conv3, self.bn3))
        if self.downsample is not None:
            f['ds'] = _fold_conv_bn(self.downsample[0], self.downsample[1])
        return f

    def run_bf16_storage(self, x, f):
        """The frozen block with bf16 activations in HBM (cfg[4] bf16 mode, csrc/conv_bf16s.hip):
        bf16 in, bf16 out, fp32 accumulate / bias / residual add / ReLU inside the kernels."""
        identity, fk = x, None
        if 'ds' in f:
            if x.is_cuda and BF.shortcut_fork_enabled():      # projection shortcut beside conv1 / conv2 (see run)
                with BF.forked(x.device) as fk:
                    identity = BF.conv2d_nhwc(x, f['ds'][0], f['ds'][1], stride=self.stride)
            else:
                identity = BF.conv2d_nhwc(x, f['ds'][0], f['ds'][1], stride=self.stride)
        out = BF.conv2d_nhwc(x, f['c1'][0], f['c1'][1], relu=True)
        if self.groups > 1:
            out = BF.grouped_conv3x3_nhwc(out, f['c2'][0], f['c2'][1], self.groups, stride=self.stride,
                                          relu=True)
        else:
            out = BF.conv2d_nhwc(out, f['c2'][0], f['c2'][1], stride=self.stride, pad=1, relu=True)
        if fk is not None:
            fk.join()
        return BF.conv2d_nhwc(out, f['c3'][0], f['c3'][1], relu=True, residual=identity)

    def run(self, x, f):
        if x.dtype == torch.bfloat16:
            return self.run_bf16_storage(x, f)
        # Residual fork (x feeds conv1 / the projection shortcut AND the identity path): with a trainable
        # predecessor the first conv hands an alias of x to the other consumer (`passthrough`), so both
        # gradients meet in ITS backward and their sum — and the ReLU gate of x, when x is the
        # `relu='consumers'` output of the previous block — ride in that conv's dgrad epilogue instead of
        # an add pass and a threshold pass over the map per block (functional._ConvFn).
        tagged = bool(getattr(x, '_bgs_consumers_mask', False))
        fuse = BF.fork_fusion_enabled() and torch.is_grad_enabled() and x.requires_grad
        if tagged and not fuse:
            raise RuntimeError("Bottleneck.run: x is a relu='consumers' block output but fork fusion is off")
        out_relu = 'consumers' if BF.fork_fusion_enabled() else True
        xin = x
        identity = x
        fk = None
        if 'ds' in f and not fuse and x.is_cuda and BF.shortcut_fork_enabled() and \
                not (torch.is_grad_enabled() and any(t.requires_grad for t in f['ds'] if t is not None)):
            # frozen block: the projection shortcut (a 1x1 / stride-s conv of the block input) is independent of
            # conv1 -> conv2 and runs beside them on the side stream; conv3's epilogue consumes it after the join
            # (the stride-16 / 32 stages launch 264 - 528 workgroups per conv: two launches fill the chip)
            with BF.forked(x.device) as fk:
                identity = BF.conv2d_nhwc(x, f['ds'][0], f['ds'][1], stride=self.stride)
        elif 'ds' in f:
            if fuse:
                identity, xin = BF.conv2d_autograd(x, f['ds'][0], f['ds'][1], stride=self.stride,
                                                   mask_input=tagged, passthrough=True)
            else:
                identity = BF.conv2d_autograd(x, f['ds'][0], f['ds'][1], stride=self.stride)
        first_pt = fuse and 'ds' not in f          # conv1 is the fork's first conv
        first_mask = tagged and 'ds' not in f      # (behind a projection shortcut the alias is ungated:
        #                                             the shortcut's epilogue gates the sum)
        # conv1 -> conv2 -> conv3 is a chain of single consumers: the ReLU backward of o1 / o2
        # rides in the epilogue of the next conv's dgrad (relu='consumers' + mask_input)
        if self.groups > 1:      # ResNeXt: grouped 3x3 (csrc/grouped_conv.hip)
            out = BF.conv2d_autograd(xin, f['c1'][0], f['c1'][1], relu=True, mask_input=first_mask,
                                     passthrough=first_pt)
            if first_pt:
                out, identity = out
            out = BF.grouped_conv3x3_nhwc(out, f['c2'][0], f['c2'][1], self.groups,
                                          stride=self.stride, relu=True)
            if fk is not None:
                fk.join()
            return BF.conv2d_autograd(out, f['c3'][0], f['c3'][1], relu=out_relu, residual=identity)
        out = BF.conv2d_autograd(xin, f['c1'][0], f['c1'][1], relu='consumers', mask_input=first_mask,
                                 passthrough=first_pt)
        if first_pt:
            out, identity = out
        if not fuse and BF.fused_c3_eligible(out, f['c2'][0], f['c3'][0], self.stride, identity):
            # frozen 64 -> 64 -> 256 block on a large map (ResNet-50 layer1): conv2 -> conv3 + residual + ReLU in one
            # launch, the 64-channel intermediate stays in LDS (csrc/conv_bfx.hip, round 6; bit-identical)
            if fk is not None:
                fk.join()
            return BF.conv3x3_c3_fused_nhwc(out, f['c2'][0], f['c2'][1], f['c3'][0], f['c3'][1], residual=identity,
                                            relu3=True)
        out = BF.conv2d_autograd(out, f['c2'][0], f['c2'][1], stride=self.stride, pad=1,
                                 relu='consumers', mask_input=True)
        if fk is not None:
            fk.join()
        return BF.conv2d_autograd(out, f['c3'][0], f['c3'][1], relu=out_relu, residual=identity,
                                  mask_input=True)


@BACKBONES.register_module
class ResNet(nn.Module):
    arch_settings = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}

    def __init__(self, depth, num_stages=4, strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1),
                 out_indices=(0, 1, 2, 3), style='pytorch', frozen_stages=-1, conv_cfg=None,
                 norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, dcn=None,
                 stage_with_dcn=(False, False, False, False), gcb=None,
                 stage_with_gcb=(False, False, False, False), gen_attention=None,
                 stage_with_gen_attention=((), (), (), ()), with_cp=False,
                 zero_init_residual=True, groups=1, base_width=4):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError('invalid depth {} for resnet (bottleneck depths only)'.format(depth))
        if dcn:
            raise NotImplementedError('dcn (deformable convolution, gs_htc_dconv_* config) has no '
                                      'kernel in this build')
        if style != 'pytorch' or tuple(dilations) != (1, 1, 1, 1) or gcb or gen_attention:
            raise NotImplementedError('only the plain pytorch-style trunk of the BAGS configs')
        if norm_cfg.get('type', 'BN') != 'BN' or not norm_eval:
            raise NotImplementedError('BatchNorm in eval mode only (norm_eval=True is what every '
                                      'BAGS config uses); it is folded into the convs')
        self.depth, self.num_stages = depth, num_stages
        self.out_indices, self.frozen_stages, self.norm_eval = out_indices, frozen_stages, norm_eval
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        blocks = self.arch_settings[depth][:num_stages]
        inplanes = 64
        self.res_layers = []
        for i, nb in enumerate(blocks):
            planes = 64 * 2 ** i
            layers = []
            for j in range(nb):
                stride = strides[i] if j == 0 else 1
                layers.append(Bottleneck(inplanes, planes, stride,
                                         downsample=(j == 0 and (stride != 1 or
                                                                 inplanes != planes * 4)),
                                         groups=groups, base_width=base_width))
                inplanes = planes * 4
            name = 'layer{}'.format(i + 1)
            self.add_module(name, nn.Sequential(*layers))
            self.res_layers.append(name)
        self.feat_dim = inplanes
        self._cache = _FoldCache()
        self._stem_split = _FoldCache()      # split planes of the fused stem kernel
        self._freeze_stages()

    def _freeze_stages(self):
        """resnet.py:483-494."""
        if self.frozen_stages >= 0:
            for m in (self.conv1, self.bn1):
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            for p in getattr(self, 'layer{}'.format(i)).parameters():
                p.requires_grad = False

    def init_weights(self, pretrained=None):
        """resnet.py:496-520 (kaiming for convs, BN = 1/0, zero-init of the last BN)."""
        if isinstance(pretrained, str):
            # resnet.py:496-499: load_checkpoint(self, pretrained, strict=False)
            if '://' in pretrained:      # torchvision:// / open-mmlab:// / http(s):// model-zoo URLs
                import warnings
                warnings.warn('pretrained=%r is a model-zoo URL and there is no network here: the '
                              'backbone keeps its RANDOM initialisation (frozen stages included). '
                              'Pass a local checkpoint path, or load a state_dict explicitly '
                              '(checkpoint.load_checkpoint / load_from).' % (pretrained,),
                              RuntimeWarning, stacklevel=2)
            else:
                from .checkpoint import load_checkpoint
                load_checkpoint(self, pretrained, strict=False)     # raises if the file is missing
                return
        elif pretrained is not None:
            raise TypeError('pretrained must be a str or None')
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        for m in self.modules():
            if isinstance(m, Bottleneck):
                nn.init.constant_(m.bn3.weight, 0)

    def _build_stem(self):
        return _fold_conv_bn(self.conv1, self.bn1, pad_cin_to=4)

    @staticmethod
    def to_nhwc4(img):
        """NCHW image batch -> NHWC with the channel axis zero-padded to 4 (16-byte pixels)."""
        if img.is_cuda and not (torch.is_grad_enabled() and img.requires_grad):
            return BF.nchw_to_nhwc4(img)          # one launch instead of a fill + a strided copy
        x = img.float().permute(0, 2, 3, 1)
        return torch.nn.functional.pad(x, (0, 4 - x.shape[3])).contiguous()

    def forward(self, img):
        """img ``[N,3,H,W]`` (as the reference) -> tuple of NHWC feature maps."""
        return self.forward_partial(img, -1, len(self.res_layers) - 1)[1]

    def forward_partial(self, x, first, last, outs=()):
        """Stages ``first .. last`` of the trunk (``first = -1``: ``x`` is the image batch and the stem runs first;
        otherwise ``x`` is the activation in front of ``res_layers[first]``) -> ``(activation, outs)`` with the output
        maps of the stages in ``out_indices`` appended to ``outs``.  ``forward`` = all of it; ``train.TrunkPipeline``
        runs the pieces of a frozen trunk on different streams for different batches."""
        outs = list(outs)
        if first < 0:
            x = self._forward_stem(x)
            first = 0
        for i in range(first, last + 1):
            for blk in getattr(self, self.res_layers[i]):
                x = blk.run(x, blk.folded())
            if i in self.out_indices:
                # a trainable stage hands on a `relu='consumers'` tensor (its ReLU backward rides in the
                # next block's dgrad epilogue); what LEAVES the trunk is an ordinary tensor any consumer
                # may use: relu_gate applies the gate to the gradient coming back (a no-op otherwise)
                outs.append(BF.relu_gate(x))
        return x, tuple(outs)

    def _forward_stem(self, img):
        # frozen_stages >= 1 (every BAGS config): a plain forward launch on the cached fold;
        # frozen_stages < 1: the fold is on the tape (resnet.py:483-494), the conv records its
        # weight / bias gradient (the image needs none) and the max-pool its routing
        stem = self._cache.get((self.conv1, self.bn1), self._build_stem)
        frozen_stem = not (torch.is_grad_enabled() and (img.requires_grad or stem[0].requires_grad or
                                                        stem[1].requires_grad))
        if img.is_cuda and frozen_stem and img.shape[1] == 3 and tuple(stem[0].shape[:3]) == (64, 7, 7) \
                and BF.stem_fused_enabled():
            # round 5: conv1 + ReLU + max-pool in ONE launch straight from the NCHW image (csrc/stem_fused.hip): the
            # [N, H/2, W/2, 64] conv map (137 MB at cfg[1]) and the NHWC copy of the image never reach HBM
            ws = self._stem_split.get((self.conv1, self.bn1),
                                      lambda: BF.stem_fused_split_weights(stem[0].detach()))
            x = BF.stem_fused(img, ws, stem[1].detach())
        else:
            x = self.to_nhwc4(img)
            x = BF.conv2d_autograd(x, stem[0], stem[1], stride=2, pad=3, relu=True)
            # cfg[4] bf16 mode with a frozen trunk: the activations of layer1..4 live in HBM as bf16
            # (the storage side of wrap_fp16_model, mmdet/core/fp16/decorators.py:8-80)
            storage = (BF.bf16_storage_active() and not x.requires_grad and
                       not _trainable(*[getattr(self, n) for n in self.res_layers]))
            x = BF.maxpool3x3s2_nhwc(x, out_dtype=torch.bfloat16 if storage else torch.float32)
        return x

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self


@BACKBONES.register_module
class ResNeXt(ResNet):
    """mmdet/models/backbones/resnext.py:95-222: the ResNet layout with grouped 3x3 convs
    (``groups=64, base_width=4`` = X101-64x4d in configs/bags/gs_cascade_rcnn_x101_64x4d_*.py).
    Parameter names / shapes are the reference's."""

    def __init__(self, groups=1, base_width=4, **kwargs):
        super().__init__(groups=groups, base_width=base_width, **kwargs)
        self.groups, self.base_width = groups, base_width


class ConvModule(nn.Module):
    """Container with the reference's ``.conv`` attribute name (mmdet/models/utils/conv_module.py)."""

    def __init__(self, cin, cout, k, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=padding)
        self.padding = padding


@NECKS.register_module
class FPN(nn.Module):

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1,
                 add_extra_convs=False, extra_convs_on_inputs=True,
                 relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None,
                 norm_cfg=None, activation=None):
        super().__init__()
        assert isinstance(in_channels, list)
        if add_extra_convs or norm_cfg is not None or activation is not None or start_level != 0 \
                or end_level != -1:
            raise NotImplementedError('FPN variant outside the BAGS configs')
        self.in_channels, self.out_channels, self.num_outs = in_channels, out_channels, num_outs
        self.num_ins = len(in_channels)
        self.lateral_convs = nn.ModuleList(ConvModule(c, out_channels, 1) for c in in_channels)
        self.fpn_convs = nn.ModuleList(ConvModule(out_channels, out_channels, 3, padding=1)
                                       for _ in in_channels)
        self._cache = _FoldCache()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                nn.init.constant_(m.bias, 0)

    def _build_fold(self):
        return dict(lat=[_fold_conv_bn(m.conv, None) for m in self.lateral_convs],
                    out=[_fold_conv_bn(m.conv, None) for m in self.fpn_convs])

    def forward(self, inputs):
        """inputs: NHWC C2..C5 -> NHWC P2..P6 (fpn.py:101-141)."""
        assert len(inputs) == self.num_ins
        f = self._cache.get(self, self._build_fold)
        n = self.num_ins
        if inputs[0].dtype != torch.float32 and torch.is_grad_enabled() and _trainable(self):
            inputs = [t.float() for t in inputs]      # a trainable neck records fp32 operands
        lat = [None] * n
        # bf16-stored trunk maps (cfg[4] bf16 mode) are read as they are; the pyramid itself is fp32
        # (trunk maps that are `relu='consumers'` block outputs: the lateral's dgrad epilogue gates them)
        gate = [bool(getattr(t, '_bgs_consumers_mask', False)) for t in inputs]
        lat[n - 1] = BF.conv2d_autograd(inputs[n - 1], *f['lat'][n - 1], out_dtype=torch.float32,
                                        mask_input=gate[n - 1])
        if n > 1 and lat[n - 1].is_cuda and BF.level_fork_enabled():
            # The output convs of the small levels (168 / 48 workgroups on 256 CUs) go to the side stream as soon as
            # their lateral exists (functional.forked: each block waits for what the main stream has issued so
            # far), beside the remaining laterals and the P2-level output conv on the main stream.
            outs, fk = [None] * n, None
            for i in range(n - 1, 0, -1):
                with BF.forked(lat[i].device) as fk:
                    outs[i] = BF.conv2d_autograd(lat[i], *f['out'][i], pad=1)
                    if i == n - 1:
                        extra = []
                        for _ in range(self.num_outs - n):   # F.max_pool2d(x, 1, stride=2) == subsampling
                            extra.append((extra[-1] if extra else outs[i])[:, ::2, ::2, :].contiguous())
                lat[i - 1] = BF.conv2d_autograd(inputs[i - 1], *f['lat'][i - 1], residual=lat[i],
                                                residual_mode=2, out_dtype=torch.float32, mask_input=gate[i - 1])
            outs[0] = BF.conv2d_autograd(lat[0], *f['out'][0], pad=1)
            fk.join()                                        # (the side stream is in order: its last block is its last work)
            return tuple(outs + extra)
        for i in range(n - 2, -1, -1):   # lateral_i + nearest_2x(lateral_{i+1}), fused
            lat[i] = BF.conv2d_autograd(inputs[i], *f['lat'][i], residual=lat[i + 1],
                                        residual_mode=2, out_dtype=torch.float32, mask_input=gate[i])
        outs = [BF.conv2d_autograd(lat[i], *f['out'][i], pad=1) for i in range(n)]
        for _ in range(self.num_outs - n):   # F.max_pool2d(x, 1, stride=2) == subsampling
            outs.append(outs[-1][:, ::2, ::2, :].contiguous())
        return tuple(outs)
