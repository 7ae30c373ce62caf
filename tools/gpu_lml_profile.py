#!/usr/bin/env python3
"""`rocprofv3 --kernel-trace --stats -- python3 tools/gpu_lml_profile.py N [reps]`: LML + gradient evaluations of 10 GPs only."""
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
theta = synth.fixed_theta(d, P)
for _ in range(reps):
    eng.lml(theta, eval_gradient=True)
eng.sync()
