"""The C-ABI library loads and exports every symbol include/cffm_hip.h declares (no compute, no GPU)."""
import ctypes
import os
import re

import pytest

from vss_cffm_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'cffm_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(cffm_[a-z0-9_]+)\s*\(', text)))


def test_header_and_binding_agree():
    syms = declared_symbols()
    assert len(syms) >= 30
    assert set(syms) == set(_lib.SIGNATURES), set(syms) ^ set(_lib.SIGNATURES)


@pytest.mark.skipif(not os.path.isfile(_lib.LIB_PATH), reason='libcffm_hip.so not built (run __graft_entry__.build())')
def test_product_library_exports_every_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), s
    lib.cffm_abi_version.restype = ctypes.c_int
    assert lib.cffm_abi_version() == _lib.ABI_VERSION


def test_geom_and_layout_through_abi():
    from tests import emu
    lib = emu.lib()
    g = _lib.Geom()
    assert lib.cffm_geom_init(ctypes.byref(g), 2, 60, 60) == 0
    assert (g.Hp, g.Wp, g.gy, g.gx, g.nW, g.HW, g.RC) == (63, 63, 9, 9, 81, 3600, 5184)
    assert lib.cffm_geom_init(ctypes.byref(g), 0, 60, 60) != 0
    assert b'bad sizes' in lib.cffm_last_error()


def test_no_cpu_fallback():
    """The product path must fail loudly off-GPU (no oracle / CPU fallback behind it)."""
    import torch
    from vss_cffm_amd import ops
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(_lib.CffmError):
        ops.cffm_layer(torch.zeros(1, 4, 256, 8, 8), 1, [torch.zeros(1)] * ops.NPB)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'vss_cffm_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
                assert 'libcffm_emu' not in src or f == '_lib.py', f
