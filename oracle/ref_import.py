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


# ------------------------------------------------------------------------------------------------ level 2
def import_mmseg_models():
    """Import the reference's ``mmseg.models`` package (whole heads through its registry) with permissive
    stand-ins for the third-party packages that are absent here (level 2): a real mini
    ``Registry`` / ``build_from_cfg`` / ``ConvModule`` (BatchNorm2d standing in for SyncBN exactly as the
    reference's own CPU tests do, tests/test_models/test_forward.py:186) and placeholder classes for the rest.
    Used only to generate / check golden vectors of the head; never shipped."""
    if 'mm' in _cache:
        return _cache['mm']
    _stub_timm()

    class _Stub(types.ModuleType):
        __path__ = []

        def __getattr__(self, name):
            if name.startswith('__'):
                raise AttributeError(name)
            cls = type(name, (nn.Module,), {})
            setattr(self, name, cls)
            return cls

    def passthrough_decorator(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    class Registry:
        def __init__(self, name):
            self.name, self.module_dict = name, {}

        def get(self, key):
            return self.module_dict.get(key)

        def register_module(self, name=None, force=False, module=None):
            def deco(cls):
                self.module_dict[name or cls.__name__] = cls
                return cls
            return deco(module) if module is not None else deco

    def build_from_cfg(cfg, registry, default_args=None):
        args = dict(cfg)
        for k, v in (default_args or {}).items():
            args.setdefault(k, v)
        return registry.get(args.pop('type'))(**args)

    class ConvModule(nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size, norm_cfg=None, act_cfg=dict(type='ReLU'), **kw):
            super().__init__()
            self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, bias=norm_cfg is None)
            self.bn = nn.BatchNorm2d(out_channels) if norm_cfg is not None else None
            self.activate = nn.ReLU(inplace=True)

        def forward(self, x):
            x = self.conv(x)
            if self.bn is not None:
                x = self.bn(x)
            return self.activate(x)

    names = ['mmcv', 'mmcv.cnn', 'mmcv.cnn.bricks', 'mmcv.cnn.utils', 'mmcv.cnn.utils.flops_counter', 'mmcv.runner',
             'mmcv.utils', 'mmcv.utils.parrots_wrapper', 'mmcv.ops', 'mmcv.parallel', 'mmcv.image', 'mmcv.fileio',
             'IPython', 'cv2', 'fast_pytorch_kmeans', 'terminaltables', 'torchvision', 'timm.models.registry',
             'timm.models.vision_transformer']
    mods = {}
    for n in names:
        mods[n] = _Stub(n)
        sys.modules[n] = mods[n]
    for n, m in mods.items():
        if '.' in n and n.rsplit('.', 1)[0] in mods:
            setattr(mods[n.rsplit('.', 1)[0]], n.rsplit('.', 1)[1], m)
    mods['mmcv'].__version__ = '1.3.0'
    mods['mmcv.utils'].Registry = Registry
    mods['mmcv.utils'].build_from_cfg = build_from_cfg
    mods['mmcv.utils'].print_log = print
    mods['mmcv.cnn'].ConvModule = ConvModule
    mods['mmcv.cnn'].normal_init = lambda m, mean=0, std=1, bias=0: (nn.init.normal_(m.weight, mean, std),
                                                                      nn.init.constant_(m.bias, bias))
    for reg in ('UPSAMPLE_LAYERS', 'CONV_LAYERS', 'NORM_LAYERS', 'ACTIVATION_LAYERS', 'PLUGIN_LAYERS'):
        setattr(mods['mmcv.cnn'], reg, Registry(reg))
        setattr(mods['mmcv.cnn.bricks'], reg, getattr(mods['mmcv.cnn'], reg))
    mods['mmcv.runner'].auto_fp16 = passthrough_decorator
    mods['mmcv.runner'].force_fp32 = passthrough_decorator
    mods['timm.models.registry'].register_model = passthrough_decorator
    mods['timm.models.vision_transformer']._cfg = lambda **k: dict(k)
    mods['IPython'].embed = lambda *a, **k: None
    sys.modules['timm.models'].registry = mods['timm.models.registry']
    sys.modules['timm.models'].vision_transformer = mods['timm.models.vision_transformer']
    sys.dont_write_bytecode = True
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    with contextlib.redirect_stdout(io.StringIO()):
        import mmseg.models as M
    _cache['mm'] = M
    return M


def head_cfg(kind='CFFMHead_clips_resize1_8', in_channels=(32, 64, 160, 256), depths=1, num_classes=124, dropout_ratio=0.1):
    """Head kwargs as the B0/B1 configs give them (local_configs/cffm/B*/cffm.b*.480x480.vspw2.160k.py merged
    over _base_/models/segformer.py)."""
    return dict(type=kind, in_channels=list(in_channels), in_index=[0, 1, 2, 3], feature_strides=[4, 8, 16, 32],
                channels=128, dropout_ratio=dropout_ratio, num_classes=num_classes,
                norm_cfg=dict(type='SyncBN', requires_grad=True), align_corners=False,
                decoder_params=dict(embed_dim=256, depths=depths),
                loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0), num_clips=4)


def build_reference_head(**kw):
    M = import_mmseg_models()
    with contextlib.redirect_stdout(io.StringIO()):
        return M.build_head(head_cfg(**kw))
