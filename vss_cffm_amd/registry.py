"""Registry / build-from-config machinery the CFFM heads plug into.

The reference registers its heads in ``mmseg.models.builder.HEADS`` (an ``mmcv.utils.Registry``,
mmseg/models/builder.py:6-10) with ``@HEADS.register_module()`` (cffm_head.py:40,160,303) and builds them
with ``build_head(cfg)`` -> ``build_from_cfg`` (builder.py:13-48): pop ``type``, look the class up, pass the
rest as kwargs.  mmcv is not available on either box, so this module re-provides exactly that contract
(same names: ``Registry``, ``build_from_cfg``, ``BACKBONES/NECKS/HEADS/LOSSES/SEGMENTORS``, ``build_head`` ...).
"""
import inspect


class Registry:
    def __init__(self, name):
        self._name = name
        self._classes = {}

    name = property(lambda self: self._name)
    module_dict = property(lambda self: self._classes)

    def __len__(self):
        return len(self._classes)

    def __contains__(self, key):
        return key in self._classes

    def __repr__(self):
        return '%s(name=%s, items=%s)' % (type(self).__name__, self._name, sorted(self._classes))

    def get(self, key):
        return self._classes.get(key)

    def _add(self, cls, name=None, force=False):
        if not inspect.isclass(cls):
            raise TypeError('only classes can be registered, got %r' % (cls,))
        for key in ([name] if isinstance(name, str) else (name or [cls.__name__])):
            if key in self._classes and not force:
                raise KeyError('%s is already registered in %s' % (key, self._name))
            self._classes[key] = cls
        return cls

    def register_module(self, name=None, force=False, module=None):
        """``@REG.register_module()`` decorator or ``REG.register_module(module=cls)`` call."""
        if module is not None:
            return self._add(module, name, force)
        return lambda cls: self._add(cls, name, force)


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError('cfg must be a dict, got %s' % type(cfg))
    if 'type' not in cfg and not (default_args and 'type' in default_args):
        raise KeyError('`cfg` or `default_args` must contain the key "type", got %r' % (cfg,))
    args = dict(cfg)
    for k, v in (default_args or {}).items():
        args.setdefault(k, v)
    kind = args.pop('type')
    if isinstance(kind, str):
        cls = registry.get(kind)
        if cls is None:
            raise KeyError('%s is not in the %s registry' % (kind, registry.name))
    elif inspect.isclass(kind):
        cls = kind
    else:
        raise TypeError('type must be a str or a class, got %s' % type(kind))
    return cls(**args)


BACKBONES = Registry('backbone')
NECKS = Registry('neck')
HEADS = Registry('head')
LOSSES = Registry('loss')
SEGMENTORS = Registry('segmentor')


def build(cfg, registry, default_args=None):
    if isinstance(cfg, (list, tuple)):
        import torch.nn as nn
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_backbone(cfg):
    return build(cfg, BACKBONES)


def build_neck(cfg):
    return build(cfg, NECKS)


def build_head(cfg):
    return build(cfg, HEADS)


def build_loss(cfg):
    return build(cfg, LOSSES)


def build_segmentor(cfg, train_cfg=None, test_cfg=None):
    return build(cfg, SEGMENTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))
