"""TEST INFRASTRUCTURE: run the HIP kernel *sources* on the CPU through the fiber emulator
(vss_cffm_amd/csrc/hipemu.h) so kernel logic is checked against the oracle where no GPU exists.
This is not a product path: vss_cffm_amd never loads libcffm_emu.so by itself."""
import contextlib
import os
import subprocess

from vss_cffm_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_SO = os.path.join(ROOT, 'tests', 'libcffm_emu.so')
_emu = None


def build(force=False):
    srcdir = os.path.join(ROOT, 'vss_cffm_amd', 'csrc')
    newest = max(os.path.getmtime(os.path.join(srcdir, f)) for f in os.listdir(srcdir))
    if force or not os.path.isfile(EMU_SO) or os.path.getmtime(EMU_SO) < newest:
        subprocess.check_call(['bash', os.path.join(ROOT, 'build_native.sh'), '--emu'])
    return EMU_SO


def lib():
    global _emu
    if _emu is None:
        _emu = _lib.bind(build())
    return _emu


@contextlib.contextmanager
def active():
    """Route vss_cffm_amd's operators to the emulator build (CPU tensors) inside the block."""
    prev = _lib._override
    _lib._override = lib()
    try:
        yield _lib._override
    finally:
        _lib._override = prev
