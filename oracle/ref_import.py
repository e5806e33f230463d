"""TEST INFRASTRUCTURE ONLY — never imported by the product package.

Imports the *actual* reference implementation (``/root/reference``, an mmdetection
v1.0rc0 fork) on CPU by stubbing the dependencies that are absent from this image
(mmcv, pycocotools, lvis, cv2, the compiled CUDA extensions) and neutralising the
hard-coded ``.cuda()`` calls (reference: mmdet/models/bbox_heads/gs_bbox_head_with0.py:37-49,85).

``/root/reference`` exists only in the authoring container, never on the GPU box:
this module is used solely by ``tests/golden/make_golden.py`` (fixture generation)
and by CPU tests that are skipped when the reference tree is absent.
"""
import os
import sys
import types
from unittest.mock import MagicMock

REFERENCE_ROOT = os.environ.get('BGS_REFERENCE_ROOT', '/root/reference')


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'mmdet'))


class _Base(object):
    def __init__(self, *a, **k):
        pass


def _noop(*a, **k):
    return None


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _obj_from_dict(info, parent=None, default_args=None):
    """mmcv.runner.obj_from_dict semantics (type -> getattr(parent, type)(**kw))."""
    args = dict(info)
    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        obj_type = getattr(parent, obj_type) if parent is not None else sys.modules[obj_type]
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    return obj_type(**args)


_INSTALLED = False


def install_stubs(root=None):
    """Install the import stubs and put the reference tree on sys.path (idempotent).
    ``root``: an alternative location of the reference's python modules — bench.py's
    ``cpu_baseline`` leg passes the head closure staged by oracle/build_ref.py
    (``oracle/_ref/reference_py``) on the GPU box, where ``/root/reference`` does not exist."""
    global _INSTALLED, REFERENCE_ROOT
    if _INSTALLED:
        return
    if root is not None:
        REFERENCE_ROOT = root
    if not reference_available():
        raise RuntimeError('reference tree not found at %s' % REFERENCE_ROOT)
    import torch

    def _slice_list(in_list, lens):
        out, i = [], 0
        for n in lens:
            out.append(in_list[i:i + n])
            i += n
        return out

    mm = _stub('mmcv', is_str=lambda x: isinstance(x, str), Config=MagicMock(),
               imresize=MagicMock(), imdenormalize=MagicMock(), bbox_flip=MagicMock(),
               slice_list=_slice_list, concat_list=lambda ls: sum(ls, []),
               imrescale=MagicMock(), imflip=MagicMock(), impad=MagicMock(),
               imnormalize=MagicMock(), __path__=[])
    inits = dict(constant_init=_noop, kaiming_init=_noop, normal_init=_noop,
                 xavier_init=_noop, uniform_init=_noop, caffe2_xavier_init=_noop)
    _stub('mmcv.cnn', VGG=torch.nn.Module, __path__=[], **inits)
    _stub('mmcv.cnn.weight_init', **inits)
    _stub('mmcv.runner', load_checkpoint=_noop, OptimizerHook=_Base, Hook=_Base,
          Runner=_Base, DistSamplerSeedHook=_Base, get_dist_info=lambda: (0, 1),
          obj_from_dict=_obj_from_dict, __path__=[])
    _stub('mmcv.runner.utils', get_dist_info=lambda: (0, 1))
    _stub('mmcv.parallel', collate=_noop, scatter=_noop, MMDataParallel=_Base,
          MMDistributedDataParallel=_Base, DataContainer=_Base)
    _stub('mmcv.parallel.data_container', DataContainer=_Base)
    mm.runner = sys.modules['mmcv.runner']
    mm.cnn = sys.modules['mmcv.cnn']
    mm.parallel = sys.modules['mmcv.parallel']
    for n in ['pycocotools', 'pycocotools.mask', 'pycocotools.coco',
              'pycocotools.cocoeval', 'lvis', 'lvis.lvis', 'terminaltables',
              'imagecorruptions', 'albumentations', 'cv2',
              'mmdet.ops.nms.nms_cpu', 'mmdet.ops.nms.nms_cuda',
              'mmdet.ops.nms.soft_nms_cpu', 'mmdet.ops.roi_align.roi_align_cuda',
              'mmdet.ops.roi_pool.roi_pool_cuda', 'mmdet.ops.dcn.deform_conv_cuda',
              'mmdet.ops.dcn.deform_pool_cuda',
              'mmdet.ops.sigmoid_focal_loss.sigmoid_focal_loss_cuda',
              'mmdet.ops.masked_conv.masked_conv2d_cuda']:
        sys.modules[n] = MagicMock()
    _stub('mmdet.version', __version__='1.0.rc0+oracle', short_version='1.0.rc0')
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # the reference hard-codes .cuda(); on this CPU-only oracle it must be identity
    torch.Tensor.cuda = lambda self, *a, **k: self
    _INSTALLED = True


class AttrDict(dict):
    """gs_config is read by attribute in the reference (gs_bbox_head_with0.py:29,34)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def build_reference_head(table_dir, cls_name='GSBBoxHeadWith0', others_sample_ratio=8.0,
                         num_classes=1231, reg_class_agnostic=False, bin_cls_weight=None,
                         num_bins=5):
    """Construct the reference's own head class from table files in ``table_dir``."""
    install_stubs()
    import importlib
    modname = {'GSBBoxHeadWith0': 'gs_bbox_head_with0',
               'GSBBoxHeadWith0Reweight': 'gs_bbox_head_with0_reweight'}[cls_name]
    mod = importlib.import_module('mmdet.models.bbox_heads.' + modname)
    cls = getattr(mod, cls_name)
    ce = dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)
    gs = AttrDict(label2binlabel=os.path.join(table_dir, 'label2binlabel.pt'),
                  pred_slice=os.path.join(table_dir, 'pred_slice_with0.pt'),
                  fg_split=os.path.join(table_dir, 'valsplit.pkl'),
                  others_sample_ratio=others_sample_ratio, loss_bg=dict(ce),
                  num_bins=num_bins, loss_bin=dict(ce))
    if bin_cls_weight is not None:
        gs['bin_cls_weight'] = bin_cls_weight
    head = cls(num_fcs=2, in_channels=256, fc_out_channels=1024, gs_config=gs,
               roi_feat_size=7, num_classes=num_classes,
               target_means=[0., 0., 0., 0.], target_stds=[0.1, 0.1, 0.2, 0.2],
               reg_class_agnostic=reg_class_agnostic, loss_cls=dict(ce),
               loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0))
    return head
