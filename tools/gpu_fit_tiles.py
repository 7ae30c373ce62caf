#!/usr/bin/env python3
"""Fit at fixed theta: tile choice of the panel-end SYRK and of the triangular-inverse levels (tune keys)."""
import os
os.environ.setdefault("GPB_DEBUG_LIB", "1")      # the sweeps switch to kernel variants of the debug build
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpbayestools_hic_amd import GPEngine, synth  # noqa: E402
from gpu_fit_timing import timed  # noqa: E402

P = 10
for N, d in ((2048, 20), (4096, 20)):
    eng = GPEngine(0)
    eng.set_data(synth.lhs(N, d), np.random.default_rng(1).standard_normal((P, N)), "RBF", 0.1)
    eng.set_theta(synth.fixed_theta(d, P))
    row = {"N": N}
    for syrk in (0, 64, 128):
        for trtri in (0, 64, 128):
            eng.tune("syrk_tile", syrk); eng.tune("trtri_tile", trtri)
            row[f"syrk{syrk}_trtri{trtri}"] = round(timed(eng, 4), 3)
    print(json.dumps(row), flush=True)
    eng.close()
