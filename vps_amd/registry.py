"""Name -> class registries and python-file configs: the plugin surface of the reference
(mmdet/utils/registry.py:6-76, mmdet/models/registry.py:1-11, mmdet/models/builder.py:9-45 and the subset of
mmcv.Config that tools/test_vpq.py:100-131 uses), re-implemented so `configs/cityscapes/fusetrack.py`-style
model dicts build the HIP-backed components by their reference names.
"""
import importlib.util
import os


class ConfigDict(dict):
    """dict with attribute access; `hasattr(cfg, key)` is True iff the key exists (the reference relies on this
    for `hasattr(self.test_cfg, 'flownet2')`, panoptic_fusetrack.py:59-64,90-91,513,560)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    @staticmethod
    def wrap(obj):
        if isinstance(obj, dict):
            return ConfigDict({k: ConfigDict.wrap(v) for k, v in obj.items()})
        if isinstance(obj, (list, tuple)):
            return type(obj)(ConfigDict.wrap(v) for v in obj)
        return obj


class Config(ConfigDict):
    @staticmethod
    def fromfile(path):
        path = os.path.abspath(path)
        spec = importlib.util.spec_from_file_location('_vps_cfg_' + str(abs(hash(path))), path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        items = {k: v for k, v in vars(mod).items() if not k.startswith('__') and not callable(v)
                 and not isinstance(v, type(os))}
        cfg = Config(ConfigDict.wrap(items))
        dict.__setattr__(cfg, 'filename', path)
        return cfg


class Registry:
    def __init__(self, name):
        self._name = name
        self._classes = {}

    name = property(lambda s: s._name)
    module_dict = property(lambda s: s._classes)

    def get(self, key):
        return self._classes.get(key)

    def register_module(self, cls):
        if not isinstance(cls, type):
            raise TypeError('register_module expects a class, got %r' % (cls,))
        if cls.__name__ in self._classes:
            raise KeyError('%s is already registered in %s' % (cls.__name__, self._name))
        self._classes[cls.__name__] = cls
        return cls

    def __repr__(self):
        return 'Registry(%s: %s)' % (self._name, sorted(self._classes))


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict) or 'type' not in cfg:
        raise TypeError('cfg must be a dict with a "type" key, got %r' % (cfg,))
    args = dict(cfg)
    kind = args.pop('type')
    if isinstance(kind, str):
        cls = registry.get(kind)
        if cls is None:
            raise KeyError('%s is not in the %s registry' % (kind, registry.name))
    elif isinstance(kind, type):
        cls = kind
    else:
        raise TypeError('type must be a str or a class, got %r' % (kind,))
    for k, v in (default_args or {}).items():
        args.setdefault(k, v)
    return cls(**args)


BACKBONES = Registry('backbone')
NECKS = Registry('neck')
EXTRA_NECKS = Registry('extra_neck')
PANOPTIC = Registry('panoptic')
ROI_EXTRACTORS = Registry('roi_extractor')
SHARED_HEADS = Registry('shared_head')
HEADS = Registry('head')
LOSSES = Registry('loss')
DETECTORS = Registry('detector')


def _build(cfg, registry, default_args=None):
    if isinstance(cfg, (list, tuple)):
        import torch.nn as nn
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_backbone(cfg): return _build(cfg, BACKBONES)
def build_neck(cfg): return _build(cfg, NECKS)
def build_extra_neck(cfg): return _build(cfg, EXTRA_NECKS)
def build_panoptic(cfg): return _build(cfg, PANOPTIC)
def build_roi_extractor(cfg): return _build(cfg, ROI_EXTRACTORS)
def build_shared_head(cfg): return _build(cfg, SHARED_HEADS)
def build_head(cfg): return _build(cfg, HEADS)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return _build(cfg, DETECTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))
