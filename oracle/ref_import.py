"""Import the reference's own modules from /root/reference as the parity pin.  TEST INFRASTRUCTURE.

Only usable in the build container (the GPU box has no /root/reference): used by
``tests/golden/make_golden.py`` to emit golden vectors and by
``tests/test_oracle_vs_reference.py`` (skipped when the tree is absent).  The reference needs
``timm`` (absent here) for three trivial symbols; they are stubbed exactly as SURVEY.md
Appendix B describes.  Nothing is copied: the reference files are executed where they lie.
"""
import contextlib
import importlib.util
import io
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get('CFFM_REFERENCE_ROOT', '/root/reference')


def available():
    return os.path.isfile(os.path.join(REF_ROOT, 'mmseg/models/decode_heads/cffm_module/cffm_transformer.py'))


def _stub_timm():
    if 'timm.models.layers' in sys.modules and getattr(sys.modules['timm.models.layers'], '_cffm_stub', False):
        return

    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    timm = types.ModuleType('timm')
    models = types.ModuleType('timm.models')
    layers = types.ModuleType('timm.models.layers')
    layers.DropPath, layers.to_2tuple, layers.trunc_normal_ = DropPath, to_2tuple, torch.nn.init.trunc_normal_
    layers._cffm_stub = True
    timm.models, models.layers = models, layers
    for name, mod in (('timm', timm), ('timm.models', models), ('timm.models.layers', layers)):
        sys.modules.setdefault(name, mod)


def _load(rel, name):
    _stub_timm()
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_cache = {}


def cffm_transformer():
    if 'ct' not in _cache:
        _cache['ct'] = _load('mmseg/models/decode_heads/cffm_module/cffm_transformer.py', '_ref_cffm_transformer')
    return _cache['ct']


def swin_2d():
    if 'sw' not in _cache:
        _cache['sw'] = _load('mmseg/models/decode_heads/pvt/swin_transformer_2d.py', '_ref_swin_transformer_2d')
    return _cache['sw']


def build_basic_layer(depth, dim=256):
    """BasicLayer3d3 with the kwargs of cffm_head.py:74-95 (ctor prints are swallowed)."""
    with contextlib.redirect_stdout(io.StringIO()):
        m = cffm_transformer().BasicLayer3d3(
            dim=dim, depth=depth, num_heads=8, window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None,
            drop=0., attn_drop=0., drop_path=0., norm_layer=nn.LayerNorm, pool_method='fc', downsample=None,
            focal_level=2, focal_window=5, expand_size=3, use_conv_embed=False, use_shift=False,
            use_pre_norm=False, use_checkpoint=False, focal_l_clips=[1, 2, 3], focal_kernel_clips=[7, 5, 3])
    return m.eval()


def build_cluster_layer(depth=1, dim=256):
    """BasicLayer_cluster with the kwargs of cffm_head.py:369-382."""
    with contextlib.redirect_stdout(io.StringIO()):
        m = swin_2d().BasicLayer_cluster(
            dim=dim, depth=depth, num_heads=8, window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None,
            drop=0., attn_drop=0., drop_path=0., norm_layer=nn.LayerNorm, downsample=None, use_checkpoint=False)
    return m.eval()
