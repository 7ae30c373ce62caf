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
    build_native(debug_variants=True)
    from gpbayestools_hic_amd import _native
    return _native.load(False)


def _exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], stdout=subprocess.PIPE, check=True).stdout.decode()
    return sorted(ln.split()[2] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] == "T")


def test_product_library_exports_the_boundary_header_and_nothing_else(lib):
    """libgpbayes.so: every function include/gpbayes.h declares, and no other (no test hook, no tuning hook, nothing of the C++
    inside: -fvisibility=hidden).  libgpbayes_debug.so: the same plus include/gpbayes_debug.h."""
    from gpbayestools_hic_amd import _native
    names = _declared()
    assert len(names) >= 40
    assert _exported(_native.LIB_PATHS[False]) == names
    assert _exported(_native.LIB_PATHS[True]) == sorted(names + _declared("gpbayes_debug.h"))
    hooks = [n for n in names if re.match(r"gpb_(test|debug|probe)_", n)]
    assert hooks == ["gpb_debug_has_variants"]                      # (which build is this?)
    for n in names:
        assert callable(getattr(lib, n)), n


def test_binding_covers_headers(lib):
    """the boundary header and the debug header are bound one to one; in the product library the debug hooks are stubs that say
    where they live"""
    from gpbayestools_hic_amd import _native
    assert sorted(_native.BOUNDARY) == _declared()
    assert sorted(_native.DEBUG) == _declared("gpbayes_debug.h")
    with pytest.raises(_native.GPBError, match="debug build"):
        lib.gpb_test_philox(None, 0, None, None)
    dbg = _native.load(True)
    assert dbg is not lib and dbg.is_debug and not lib.is_debug
    assert dbg.gpb_debug_has_variants() == 1 and lib.gpb_debug_has_variants() == 0
    with _native.debug_library():
        assert _native.load() is dbg
    assert _native.load() is (dbg if os.environ.get("GPB_DEBUG_LIB") == "1" else lib)


def test_version(lib):
    assert lib.gpb_version() == 110


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
