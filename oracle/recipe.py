"""Deterministic synthetic parameters / inputs shared by the golden-vector generator, the
parity tests, smoke() and bench.py's cpu_baseline leg.  TEST INFRASTRUCTURE.

Values come from numpy's MT19937 ``RandomState`` (bit-stable across numpy versions and
machines), so a fixture only has to store *expected outputs*: inputs and the state_dict are
regenerated from (name, shape, seed).  The scales follow SURVEY.md section 8(d): the
reference's default init leaves the in-window bias table at zero and the softmax nearly
uniform, which is too easy, so ``qkv.weight ~ N(0,0.08)`` and every bias table ``~ N(0,0.5)``.
"""
import zlib

import numpy as np
import torch


def _rs(name, seed):
    return np.random.RandomState((zlib.crc32(name.encode()) + 7919 * seed) % (2 ** 31))


def _scale_for(name):
    if name.endswith('norm1.weight') or name.endswith('norm2.weight') or name.endswith('bn.weight'):
        return 'gain', 0.2
    if 'relative_position_bias_table' in name:
        return 'normal', 0.5
    if 'pool_layers' in name and name.endswith('weight'):
        return 'pool', 0.05
    if name.endswith('qkv.weight') or name.endswith('qkv_cluster.weight'):
        return 'normal', 0.08
    if name.endswith('running_var'):
        return 'var', 0.3
    if name.endswith('running_mean'):
        return 'normal', 0.2
    if name.endswith('.weight'):
        return 'normal', 0.05
    if name.endswith('.bias'):
        return 'normal', 0.1
    return 'normal', 0.05


def synth_tensor(name, shape, seed=0, dtype=torch.float32):
    kind, s = _scale_for(name)
    r = _rs(name, seed).standard_normal(tuple(shape)).astype(np.float64)
    if kind == 'gain':
        r = 1.0 + s * r
    elif kind == 'pool':
        r = 1.0 / max(1, shape[-1]) + s * r / max(1, shape[-1]) * 4.0
    elif kind == 'var':
        r = 1.0 + s * np.abs(r)
    else:
        r = s * r
    return torch.from_numpy(r).to(dtype)


def synth_state(module_or_shapes, seed=0, dtype=torch.float32):
    """Fill every floating-point entry of a state_dict (or {name: shape}) from the recipe.
    Integer buffers (index tables, num_batches_tracked) are left untouched / skipped."""
    if hasattr(module_or_shapes, 'state_dict'):
        items = {k: v for k, v in module_or_shapes.state_dict().items()}
    else:
        items = dict(module_or_shapes)
    out = {}
    for k, v in items.items():
        if torch.is_tensor(v):
            if not v.dtype.is_floating_point:
                continue
            out[k] = synth_tensor(k, v.shape, seed, dtype)
        else:
            out[k] = synth_tensor(k, v, seed, dtype)
    return out


def synth_input(name, shape, seed=1, scale=1.5, dtype=torch.float32):
    r = _rs('input:' + name, seed).standard_normal(tuple(shape)) * scale
    return torch.from_numpy(r).to(dtype)


# {name: shape} of one CffmTransformerBlock3d3, C = 256
def block_param_shapes(c=256, nh=8, mlp_ratio=4):
    hid = int(c * mlp_ratio)
    return {
        'pool_layers.0.weight': (1, 49), 'pool_layers.0.bias': (1,),
        'pool_layers_clips.0.weight': (1, 49), 'pool_layers_clips.0.bias': (1,),
        'pool_layers_clips.1.weight': (1, 9), 'pool_layers_clips.1.bias': (1,),
        'pool_layers_clips.2.weight': (1, 4), 'pool_layers_clips.2.bias': (1,),
        'norm1.weight': (c,), 'norm1.bias': (c,),
        'attn.relative_position_bias_table': (169, nh),
        'attn.relative_position_bias_table_to_neighbors': (1, nh, 49, 132),
        'attn.relative_position_bias_table_to_windows.0': (nh, 121),
        'attn.relative_position_bias_table_to_windows_clips.0': (nh, 169),
        'attn.relative_position_bias_table_to_windows_clips.1': (nh, 121),
        'attn.relative_position_bias_table_to_windows_clips.2': (nh, 81),
        'attn.qkv.weight': (3 * c, c), 'attn.qkv.bias': (3 * c,),
        'attn.proj.weight': (c, c), 'attn.proj.bias': (c,),
        'norm2.weight': (c,), 'norm2.bias': (c,),
        'mlp.fc1.weight': (hid, c), 'mlp.fc1.bias': (hid,),
        'mlp.fc2.weight': (c, hid), 'mlp.fc2.bias': (c,),
    }


def layer_state(depth, seed=0, dtype=torch.float32, c=256):
    shapes = {}
    for i in range(depth):
        for k, s in block_param_shapes(c).items():
            shapes['blocks.%d.%s' % (i, k)] = s
    return synth_state(shapes, seed, dtype)


def gtc_block_param_shapes(c=256, nh=8, mlp_ratio=4):
    hid = int(c * mlp_ratio)
    return {
        'norm1.weight': (c,), 'norm1.bias': (c,),
        'attn.relative_position_bias_table': (169, nh),
        'attn.qkv.weight': (3 * c, c), 'attn.qkv.bias': (3 * c,),
        'attn.proj.weight': (c, c), 'attn.proj.bias': (c,),
        'attn.qkv_cluster.weight': (2 * c, c), 'attn.qkv_cluster.bias': (2 * c,),
        'attn.proj_cluster.weight': (c, c), 'attn.proj_cluster.bias': (c,),
        'norm2.weight': (c,), 'norm2.bias': (c,),
        'mlp.fc1.weight': (hid, c), 'mlp.fc1.bias': (hid,),
        'mlp.fc2.weight': (c, hid), 'mlp.fc2.bias': (c,),
    }


def gtc_layer_state(depth=1, seed=0, dtype=torch.float32, c=256):
    shapes = {}
    for i in range(depth):
        for k, s in gtc_block_param_shapes(c).items():
            shapes['blocks.%d.%s' % (i, k)] = s
    return synth_state(shapes, seed, dtype)
