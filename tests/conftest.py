import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def relerr(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    den = np.maximum(np.abs(b), np.finfo(float).tiny)
    return float(np.max(np.abs(a - b) / den)) if a.size else 0.0


def maxrel(a, b):
    """max |a-b| / max|b|  (matrix-level relative error)"""
    a = np.asarray(a, float); b = np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def _debug_library_or_skip():
    from gpbayestools_hic_amd import _native
    if not os.path.exists(_native.LIB_PATHS[True]):
        pytest.skip("libgpbayes_debug.so is not built (python -m gpbayestools_hic_amd.build --debug-variants)")
    return _native


@pytest.fixture
def debug_lib():
    """engines created inside the test bind libgpbayes_debug.so — the product sources compiled with -DGPB_DEBUG_VARIANTS: the test
    hooks of include/gpbayes_debug.h (loopback ranks, device draws, tile trace, one rank's share of a sharded step) and the
    measured-and-rejected kernel variants.  The product library exports none of them; both libraries live side by side."""
    nat = _debug_library_or_skip()
    with nat.debug_library():
        yield nat


def debug_engine():
    """a GPEngine bound to the debug library (see debug_lib), whatever the process default is"""
    _debug_library_or_skip()
    from gpbayestools_hic_amd import GPEngine
    return GPEngine(0, debug=True)
