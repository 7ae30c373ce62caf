"""CPU tier: the C-ABI library builds, loads and exports every symbol include/gpbayes.h declares;
the product path fails loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def _declared(header="gpbayes.h"):
    txt = open(os.path.join(REPO, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gpb_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    from gpbayestools_hic_amd.build import build_native
    build_native()
    from gpbayestools_hic_amd import _native
    return _native.load()


def test_exports_every_declared_symbol(lib):
    names = _declared()
    assert len(names) >= 25
    for n in names + _declared("gpbayes_debug.h"):
        assert hasattr(lib, n), n


def test_binding_covers_headers():
    """the boundary header and the debug header are bound one to one, and the boundary holds no test hooks"""
    from gpbayestools_hic_amd import _native
    assert sorted(_native.BOUNDARY) == _declared()
    assert sorted(_native.DEBUG) == _declared("gpbayes_debug.h")
    assert not [n for n in _declared() if re.match(r"gpb_(test|debug|probe|profile)_", n)]


def test_version(lib):
    assert lib.gpb_version() == 100


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    assert lib.gpb_ctx_create(0, None, ctypes.byref(h)) == -4     # GPB_E_NODEV
    from gpbayestools_hic_amd import GPEngine
    from gpbayestools_hic_amd._native import GPBError
    with pytest.raises(GPBError):
        GPEngine(0)
