#!/usr/bin/env python3
"""`rocprofv3 --kernel-trace --stats -- python3 tools/gpu_fit_profile.py N [reps]`: the fit at fixed theta only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpbayestools_hic_amd import GPEngine, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
d, P = 20, 10
eng = GPEngine(0)
eng.set_data(synth.lhs(N, d), np.random.default_rng(1).standard_normal((P, N)), "RBF", 0.1)
eng.set_theta(synth.fixed_theta(d, P))
for key in sys.argv[3:]:
    k, v = key.split("=")
    eng.tune(k, int(v))
for _ in range(reps):
    eng.factor()
eng.sync()
