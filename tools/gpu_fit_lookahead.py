#!/usr/bin/env python3
"""Lookahead variants of the blocked Cholesky (far part of the trailing update on a side stream): off / on (low-priority side stream).  One engine per variant (the side stream is created once)."""
import os
os.environ.setdefault("GPB_DEBUG_LIB", "1")      # the sweeps switch to kernel variants of the debug build
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpbayestools_hic_amd import GPEngine, synth  # noqa: E402
from gpu_fit_timing import timed  # noqa: E402

P = 10
for N, d in ((2048, 20), (4096, 20)):
    row = {"N": N}
    for outer in (256, 512):
        for look in (0, 1):
            eng = GPEngine(0)
            eng.set_data(synth.lhs(N, d), np.random.default_rng(1).standard_normal((P, N)), "RBF", 0.1)
            eng.set_theta(synth.fixed_theta(d, P))
            eng.tune("chol_outer", outer); eng.tune("chol_lookahead", look)
            row[f"outer{outer}_look{look}"] = round(timed(eng, 4), 3)
            eng.close()
    print(json.dumps(row), flush=True)
