"""Component registries with the reference's names and lookup behaviour.

The drop-in boundary of the hot path is mmdet's registry (reference:
mmdet/utils/registry.py:6-76 and mmdet/models/registry.py:1-9): a class is registered
under its ``__name__`` by ``@HEADS.register_module`` and instantiated from a config dict
``dict(type='Name', **kwargs)`` by ``build_from_cfg``.  Behaviour kept: duplicate name ->
``KeyError``; unknown ``type`` -> ``KeyError``; non-class -> ``TypeError``;
``default_args`` only fill keys the config does not set.
"""
import inspect


class Registry(object):

    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    name = property(lambda self: self._name)
    module_dict = property(lambda self: self._module_dict)

    def __repr__(self):
        return '{}(name={}, items={})'.format(type(self).__name__, self._name,
                                              sorted(self._module_dict))

    def __contains__(self, key):
        return key in self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def _add(self, key, cls):
        if not inspect.isclass(cls):
            raise TypeError('module must be a class, but got {}'.format(type(cls)))
        if key in self._module_dict:
            raise KeyError('{} is already registered in {}'.format(key, self._name))
        self._module_dict[key] = cls

    def register_module(self, cls):
        """Decorator form used throughout the reference: ``@HEADS.register_module``."""
        self._add(getattr(cls, '__name__', None), cls)
        return cls

    def register_alias(self, key, cls):
        """Second key for an existing class.  ``GSBBoxHead`` needs it: the reference has no
        source for that name, only config strings
        (configs/ablations/gs_faster_rcnn_r50_fpn_1x_lvis.py:35)."""
        self._add(key, cls)


def build_from_cfg(cfg, registry, default_args=None):
    if not (isinstance(cfg, dict) and 'type' in cfg):
        raise AssertionError('cfg must be a dict with a "type" key')
    if not (default_args is None or isinstance(default_args, dict)):
        raise AssertionError('default_args must be a dict or None')
    kwargs = {k: v for k, v in cfg.items() if k != 'type'}
    wanted = cfg['type']
    if isinstance(wanted, str):
        cls = registry.get(wanted)
        if cls is None:
            raise KeyError('{} is not in the {} registry'.format(wanted, registry.name))
    elif inspect.isclass(wanted):
        cls = wanted
    else:
        raise TypeError('type must be a str or valid type, but got {}'.format(type(wanted)))
    for k, v in (default_args or {}).items():
        kwargs.setdefault(k, v)
    return cls(**kwargs)


BACKBONES = Registry('backbone')
NECKS = Registry('neck')
ROI_EXTRACTORS = Registry('roi_extractor')
SHARED_HEADS = Registry('shared_head')
HEADS = Registry('head')
LOSSES = Registry('loss')
DETECTORS = Registry('detector')
