"""Minimal ``mmcv.Config.fromfile`` equivalent (mmcv is not in the reference tree;
call site: tools/train.py:96).  A config is a Python file whose top-level names become
keys; dicts are attribute-accessible (``gs_config.num_bins``,
mmdet/models/bbox_heads/gs_bbox_head_with0.py:29,34)."""
import os
import runpy


class ConfigDict(dict):
    """dict with attribute access; missing attributes raise AttributeError."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError("'ConfigDict' object has no attribute '%s'" % name)

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        del self[name]


def to_config_dict(obj):
    if isinstance(obj, dict):
        return ConfigDict((k, to_config_dict(v)) for k, v in obj.items())
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_config_dict(v) for v in obj)
    return obj


class Config(object):

    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, '_cfg_dict', to_config_dict(cfg_dict or {}))
        object.__setattr__(self, '_filename', filename)

    @staticmethod
    def fromfile(filename):
        filename = os.path.abspath(os.path.expanduser(filename))
        if not os.path.isfile(filename):
            raise FileNotFoundError('file "%s" does not exist' % filename)
        if not filename.endswith('.py'):
            raise IOError('Only py type is supported')
        ns = runpy.run_path(filename)
        cfg = {k: v for k, v in ns.items()
               if not k.startswith('__') and not callable(v) and not hasattr(v, '__loader__')}
        return Config(cfg, filename=filename)

    @property
    def filename(self):
        return self._filename

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = to_config_dict(value)

    def __contains__(self, name):
        return name in self._cfg_dict

    def get(self, name, default=None):
        return self._cfg_dict.get(name, default)

    def __repr__(self):
        return 'Config (path: %s): %r' % (self._filename, dict(self._cfg_dict))
