"""ctypes binding of libcffm_hip.so (C ABI: include/cffm_hip.h).

The product path loads ONLY ``libcffm_hip.so`` (built by ``__graft_entry__.build()`` with hipcc for
gfx950) and raises if it is missing -- there is no CPU fallback.  ``bind(path)`` is the generic
signature binder; the test-suite uses it to load the emulator build of the same sources
(tests/emu.py) and ``_override`` is the single seam through which it does so.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libcffm_hip.so')
ABI_VERSION = 10

vp, ci, cl, cd, cf = C.c_void_p, C.c_int, C.c_long, C.c_double, C.c_float


class Geom(C.Structure):
    _fields_ = [(n, ci) for n in ('B', 'H0', 'W0', 'Hp', 'Wp', 'gy', 'gx', 'nW', 'HW', 'RC')]


class BlockPtrs(C.Structure):
    """cffm_block_params / cffm_block_grads (same field order)."""
    _fields_ = [('norm1_w', vp), ('norm1_b', vp), ('pool_w', vp * 4), ('pool_b', vp * 4), ('rpb_own', vp),
                ('rpb_ring', vp), ('rpb_pool', vp * 4), ('qkv_w', vp), ('qkv_b', vp), ('proj_w', vp), ('proj_b', vp),
                ('norm2_w', vp), ('norm2_b', vp), ('fc1_w', vp), ('fc1_b', vp), ('fc2_w', vp), ('fc2_b', vp)]


class BlockWs(C.Structure):
    _fields_ = [(n, cl) for n in ('mean1', 'rstd1', 'M', 'zall', 'qkv', 'bias', 'biasT', 'lse', 'ao', 'x1', 'mean2',
                                  'rstd2', 'z2', 'hraw', 'act', 'x2', 'w_split', 'w_frag', 'ao_t', 'zall_t', 'total')]


class GtcPtrs(C.Structure):
    """cffm_gtc_params / cffm_gtc_grads (same field order)."""
    _fields_ = [(n, vp) for n in ('norm1_w', 'norm1_b', 'qkv_w', 'qkv_b', 'kv_w', 'kv_b', 'proj_w', 'proj_b', 'norm2_w', 'norm2_b',
                                  'fc1_w', 'fc1_b', 'fc2_w', 'fc2_b')]


GP, BP = C.POINTER(Geom), C.POINTER(BlockPtrs)
GTP = C.POINTER(GtcPtrs)
P4 = vp * 4

SIGNATURES = {
    'cffm_abi_version': (ci, []),
    'cffm_last_error': (C.c_char_p, []),
    'cffm_geom_init': (ci, [GP, ci, ci, ci]),
    'cffm_block_ws_layout': (ci, [GP, C.POINTER(BlockWs)]),
    'cffm_layer_saved_floats': (cl, [GP, ci]),
    'cffm_layer_scratch_floats': (cl, [GP]),
    'cffm_profile_enable': (ci, [C.c_longlong]),
    'cffm_profile_sample_every': (ci, [ci]),
    'cffm_side_streams': (ci, [ci]),
    'cffm_branch_begin': (vp, [vp, ci]),
    'cffm_branch_join': (ci, [vp]),
    'cffm_branch_mark': (ci, [vp]),
    'cffm_defer_begin': (vp, [vp]),
    'cffm_defer_join': (ci, [vp]),
    'cffm_add_inplace': (ci, [vp, vp, cl, vp]),
    'cffm_branch_take': (vp, [vp, ci]),
    'cffm_fuse_compose_fwd': (ci, [vp, vp, vp, vp, ci, ci, vp, vp, vp]),
    'cffm_fuse_compose_bwd': (ci, [vp, vp, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp]),
    'cffm_profile_stage_count': (ci, []),
    'cffm_profile_null_pair': (ci, [vp]),
    'cffm_profile_stage_name': (C.c_char_p, [ci]),
    'cffm_profile_collect': (ci, [vp, vp]),
    'cffm_profile_collect_graph': (ci, [vp, vp, ci]),
    'cffm_transpose': (ci, [vp, vp, ci, ci, ci, cl, cl, vp]),
    'cffm_pool_matrix': (ci, [P4, vp, vp]),
    'cffm_pool_matrix_bwd': (ci, [vp, P4, vp]),
    'cffm_grad_slices_padded': (None, [ci]),
    'cffm_ln_pool_fwd': (ci, [GP, vp, cl, vp, cl, vp, vp, vp, P4, vp, vp, vp, vp]),
    'cffm_ln_pool_bwd': (ci, [GP, vp, cl, vp, cl, vp, vp, vp, vp, vp, vp, vp, vp, cl, ci, vp, cl, vp, vp, vp, P4, vp]),
    'cffm_bias_assemble': (ci, [vp, vp, P4, vp, vp, vp]),
    'cffm_bias_scatter': (ci, [vp, vp, vp, P4, vp]),
    'cffm_linear_qkv_fwd': (ci, [vp, vp, vp, vp, cl, vp]),
    'cffm_attn_fwd': (ci, [GP, vp, vp, vp, vp, vp, vp, vp]),
    'cffm_attn_bwd': (ci, [GP, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    'cffm_linear_fwd': (ci, [vp, vp, vp, cl, ci, ci, vp]),
    'cffm_linear_bias_fwd': (ci, [vp, vp, vp, vp, cl, ci, ci, vp]),
    'cffm_linear_bwd_input': (ci, [vp, vp, vp, cl, ci, ci, vp]),
    'cffm_linear_bwd_weight': (ci, [vp, vp, vp, cl, ci, ci, vp]),
    'cffm_linear_bwd_weight_group': (ci, [vp, ci, vp]),
    'cffm_split4': (ci, [vp, vp, cl, vp]),
    'cffm_linear_bwd_weight_split': (ci, [vp, vp, vp, cl, ci, ci, vp]),
    'cffm_linear_bwd_weight_split_group': (ci, [vp, ci, vp]),
    'cffm_tfrag_floats': (cl, [cl, ci]),
    'cffm_dw_stream': (ci, [ci]),
    'cffm_tfrag_pack': (ci, [vp, vp, cl, ci, vp]),
    'cffm_linear_bwd_weight_tfrag': (ci, [vp, vp, vp, cl, ci, ci, vp]),
    'cffm_linear_bwd_weight_tfrag_group': (ci, [vp, ci, vp]),
    'cffm_linear_gelu_fwd': (ci, [vp, vp, vp, vp, vp, cl, ci, ci, vp]),
    'cffm_linear_residual_fwd': (ci, [vp, vp, vp, vp, vp, cl, ci, ci, vp]),
    'cffm_colsum': (ci, [vp, cl, ci, vp, vp]),
    'cffm_panel_pack_weight': (ci, [vp, ci, ci, ci, vp, vp]),
    'cffm_mlp_records': (cl, [cl]),
    'cffm_mlp_fwd': (ci, [vp, vp, cl, ci] + [vp] * 15 + [cl, vp]),
    'cffm_mlp_bwd': (ci, [vp] * 18 + [cl, vp]),
    'cffm_mlp_fwd_tfrag': (ci, [vp, vp, cl, ci] + [vp] * 18 + [cl, vp]),
    'cffm_mlp_bwd_tfrag': (ci, [vp] * 21 + [cl, vp]),
    'cffm_residual_ln': (ci, [vp, cl, ci, vp, vp, vp, vp, vp, vp, vp, vp, cl, vp]),
    'cffm_ln_bwd_residual': (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, cl, ci, vp, vp, vp]),
    'cffm_layernorm_fwd': (ci, [vp, vp, vp, vp, vp, vp, cl, vp]),
    'cffm_bias_gelu': (ci, [vp, vp, vp, cl, ci, vp]),
    'cffm_gelu_bwd': (ci, [vp, vp, vp, cl, ci, vp, vp]),
    'cffm_residual_out': (ci, [vp, vp, vp, vp, cl, vp]),
    'cffm_block_forward': (ci, [GP, BP, vp, cl, vp, cl, vp, vp, vp, vp, vp]),
    'cffm_block_backward': (ci, [GP, BP, BP, vp, cl, vp, cl, vp, vp, vp, vp, vp, vp, vp, cl, ci, vp, cl, vp, vp]),
    'cffm_layer_forward': (ci, [GP, ci, BP, vp, vp, vp, vp, vp, vp, vp]),
    'cffm_layer_backward': (ci, [GP, ci, BP, BP, vp, cl, vp, vp, vp, vp, vp, vp, vp, vp]),
    'cffm_layer_backward_range': (ci, [GP, ci, BP, BP, vp, cl, vp, vp, vp, vp, vp, vp, vp, ci, ci, vp]),
    'cffm_layer_forward_full': (ci, [GP, ci, BP, vp, vp, vp, vp, vp, vp, vp]),
    'cffm_layer_backward_full': (ci, [GP, ci, BP, BP, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, vp]),
    'cffm_gtc_attn_fwd': (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, vp]),
    'cffm_gtc_attn_bwd': (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp]),
    'cffm_gtc_ws_floats': (cl, [ci, ci, ci]),
    'cffm_gtc_block_forward': (ci, [GTP, vp, vp, vp, vp, ci, ci, ci, vp]),
    'cffm_gtc_block_backward': (ci, [GTP, GTP, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp]),
    'cffm_segfuse_fwd': (ci, [vp, vp, vp * 3, ci * 3, ci * 3, ci, ci, ci, ci, vp]),
    'cffm_segfuse_bwd': (ci, [vp, vp * 3, ci * 3, ci * 3, ci, ci, ci, ci, vp]),
    'cffm_seg_counts': (ci, [vp, vp, cl, ci, ci, ci, vp, vp]),
    'cffm_vc_counts': (ci, [vp, vp, ci, cl, ci, vp, vp]),
    'cffm_layer_forward_rows': (ci, [GP, ci, BP, vp, vp, vp, vp, vp, vp, vp]),
    'cffm_layer_backward_rows': (ci, [GP, ci, BP, BP, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    'cffm_colstats_records': (cl, [cl]),
    'cffm_colstats': (ci, [vp, cl, vp, vp]),
    'cffm_bn_relu_pool_records': (cl, [ci, ci, ci]),
    'cffm_bn_relu_pool_fwd': (ci, [vp, vp, vp, vp, vp, vp, ci, ci, ci, vp]),
    'cffm_bn_relu_pool_bwd1': (ci, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp]),
    'cffm_bn_bwd2': (ci, [vp, vp, vp, vp, vp, vp, vp, cl, vp]),
    'cffm_bn_finalize_fwd': (ci, [vp, cl, cd, vp, vp, vp, vp, cf, cf, vp, vp]),
    'cffm_bn_finalize_bwd': (ci, [vp, cl, cd, vp, vp, ci, vp, vp]),
    'cffm_rows_resize_fwd': (ci, [vp, cl, vp, cl, ci, ci, ci, ci, ci, ci, vp]),
    'cffm_rows_resize_bwd': (ci, [vp, cl, vp, cl, ci, ci, ci, ci, ci, ci, vp]),
    'cffm_clip_format': (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, C.c_float * 3, C.c_float * 3, ci, C.c_float, ci, ci, vp]),
    'cffm_clip_format_photo': (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, C.c_float * 3, C.c_float * 3, ci, C.c_float, ci, ci, vp, vp, vp]),
    'cffm_clip_format_hsv': (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, C.c_float * 3, C.c_float * 3, ci, C.c_float, ci, ci, vp, vp, vp, vp, vp, vp]),
    'cffm_clip_resize': (ci, [vp, vp, ci, ci, ci, vp, vp, ci, ci, vp]),
    'cffm_upce_blocks': (cl, [ci, ci, ci]),
    'cffm_upce_fwd': (ci, [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp]),
    'cffm_upce_bwd': (ci, [vp, vp, vp, vp, C.c_float, vp, ci, ci, ci, ci, ci, ci, ci, vp]),
    'cffm_upce_maps_finalize': (ci, [vp, ci, cl, vp, vp, vp, vp]),
    'cffm_upce_maps_fwd': (ci, [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, cl, cl, ci, ci, vp]),
    'cffm_upce_maps_bwd': (ci, [vp, vp, vp, vp, vp, vp, C.c_float, vp, ci, ci, ci, ci, ci, ci, ci, ci, cl, cl, ci, ci, vp]),
    'cffm_adamw_step': (ci, [vp, ci, cd, cd, cd, cd, cd, ci, vp]),
    'cffm_adamw_step_dev': (ci, [vp, ci, vp, cd, cd, cd, cd, cd, vp, vp]),
    'cffm_adamw_step_rows': (ci, [vp, ci, vp, vp, vp, vp, ci, vp, vp, vp]),
}


class CffmError(RuntimeError):
    pass


def bind(path):
    """Load a library exporting the C ABI of include/cffm_hip.h and attach the signatures."""
    lib = C.CDLL(path)
    missing = [n for n in SIGNATURES if not hasattr(lib, n)]
    if missing:
        raise CffmError('%s does not export %s' % (path, ', '.join(missing)))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.cffm_abi_version() != ABI_VERSION:
        raise CffmError('%s: ABI version %d, expected %d' % (path, lib.cffm_abi_version(), ABI_VERSION))
    return lib


_lib = None
_override = None   # set only by tests/emu.py (emulator build of the same kernel sources)


def get():
    """The product library.  Raises CffmError when it has not been built: no fallback."""
    global _lib
    if _override is not None:
        return _override
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise CffmError('libcffm_hip.so is missing (%s): build it with `python -c "import __graft_entry__ as g; '
                            'g.build()"`; the CFFM hot path has no CPU fallback' % LIB_PATH)
        _lib = bind(LIB_PATH)
    return _lib


def check(rc, lib):
    if rc != 0:
        raise CffmError('libcffm_hip: %s (code %d)' % (lib.cffm_last_error().decode(), rc))
