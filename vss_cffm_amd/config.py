"""Loader for the reference's python-file configs (``local_configs/cffm/**.py``), unchanged.

Semantics re-provided from mmcv 1.3 ``Config`` as the CFFM configs use them: the file is
executed; its public module-level names are the keys; ``_base_`` (str or list) names files relative to the
including file which are loaded first (a key defined by two bases is an error); the child is merged over the
bases -- dict over dict recursively, anything else replaces; ``_delete_=True`` inside a child dict drops the
base dict instead of merging; attribute access; ``merge_from_dict`` with dotted keys for ``--options``.
"""
import os
import types

BASE_KEY, DELETE_KEY = '_base_', '_delete_'


class ConfigDict(dict):
    """dict with attribute access; missing attributes raise AttributeError."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError("'ConfigDict' object has no attribute '%s'" % name)

    def __setattr__(self, name, value):
        self[name] = value


def _wrap(obj):
    if isinstance(obj, dict):
        return ConfigDict((k, _wrap(v)) for k, v in obj.items())
    if isinstance(obj, list):
        return [_wrap(v) for v in obj]
    if isinstance(obj, tuple):
        return tuple(_wrap(v) for v in obj)
    return obj


def _merge(child, base):
    """child over base (both plain dicts); returns a new dict."""
    out = dict(base)
    for k, v in child.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get(DELETE_KEY, False):
            out[k] = _merge(v, out[k])
        elif isinstance(v, dict):
            out[k] = {kk: vv for kk, vv in v.items() if kk != DELETE_KEY}
        else:
            out[k] = v
    return out


def _load_file(path):
    path = os.path.abspath(os.path.expanduser(path))
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    if not path.endswith('.py'):
        raise IOError('only .py configs are supported: %s' % path)
    scope = {'__file__': path}
    with open(path) as f:
        exec(compile(f.read(), path, 'exec'), scope)
    own = {k: v for k, v in scope.items()
           if not k.startswith('__') and not isinstance(v, (types.ModuleType, types.FunctionType, type))}
    bases = own.pop(BASE_KEY, [])
    merged = {}
    for b in ([bases] if isinstance(bases, str) else list(bases)):
        sub = _load_file(os.path.join(os.path.dirname(path), b))
        dup = set(sub) & set(merged)
        if dup:
            raise KeyError('duplicate key(s) %s in the bases of %s' % (sorted(dup), path))
        merged.update(sub)
    return _merge(own, merged)


class Config:
    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, '_cfg_dict', _wrap(cfg_dict or {}))
        object.__setattr__(self, '_filename', filename)

    @staticmethod
    def fromfile(filename):
        return Config(_load_file(filename), filename=filename)

    filename = property(lambda self: self._filename)

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __contains__(self, name):
        return name in self._cfg_dict

    def __setattr__(self, name, value):
        self._cfg_dict[name] = _wrap(value)

    __setitem__ = __setattr__

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def keys(self):
        return self._cfg_dict.keys()

    def to_dict(self):
        def plain(o):
            if isinstance(o, dict):
                return {k: plain(v) for k, v in o.items()}
            if isinstance(o, (list, tuple)):
                return type(o)(plain(v) for v in o)
            return o
        return plain(self._cfg_dict)

    def merge_from_dict(self, options):
        """``{'model.decode_head.num_classes': 19}`` style overrides (tools/train.py --options)."""
        nested = {}
        for dotted, v in options.items():
            d = nested
            parts = dotted.split('.')
            for p in parts[:-1]:
                d = d.setdefault(p, {})
            d[parts[-1]] = v
        object.__setattr__(self, '_cfg_dict', _wrap(_merge(nested, self.to_dict())))
