"""CPU checks of the C-ABI boundary: the library builds/loads, exports every symbol that
include/cna_hip.h declares, the ctypes table covers the header, and the product refuses to
run without a GPU instead of falling back."""
import os
import re

import pytest

from cna_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, 'include', 'cna_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(cna_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_ffi.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _ffi.load()
    names = header_functions()
    assert len(names) >= 35
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(_ffi.SIGNATURES) == names
    assert lib.cna_abi_version() == 1
    assert lib.cna_kernel_name(2) == b'nam_step'


def test_header_cites_the_reference_for_every_compute_entry_point():
    text = open(os.path.join(ROOT, 'include', 'cna_hip.h')).read()
    assert text.count('_nam.py:') >= 10 and text.count('_association.py:') >= 6 and text.count('_stats.py:') >= 2


def test_no_cpu_fallback_without_gpu():
    import ctypes as C
    lib = _ffi.load()
    n = C.c_int(0)
    lib.cna_device_count(C.byref(n))
    if n.value > 0:
        pytest.skip('a GPU is visible')
    from cna_amd.engine import Engine
    with pytest.raises(_ffi.CnaHipError):
        Engine()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'cna_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert 'oracle' not in src.replace('no oracle', ''), os.path.join(dp, f)
