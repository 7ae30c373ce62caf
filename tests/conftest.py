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
