#!/usr/bin/env python3
"""`rocprofv3 --kernel-trace --stats -- python3 tools/gpu_lml_profile.py N [reps] [P] [key=value ...]`: LML + gradient evaluations of P GPs only
(default 10; 63 = the batch train_emulators evaluates per lock-step round).  Prints the wall time per evaluation."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpbayestools_hic_amd import GPEngine, synth  # noqa: E402

pos = [a for a in sys.argv[1:] if "=" not in a]
N = int(pos[0]) if len(pos) > 0 else 2048
reps = int(pos[1]) if len(pos) > 1 else 10
d, P = 20, (int(pos[2]) if len(pos) > 2 else 10)
eng = GPEngine(0)
eng.set_data(synth.lhs(N, d), np.random.default_rng(1).standard_normal((P, N)), "RBF", 0.1)
theta = synth.fixed_theta(d, P)
tunes = dict(a.split("=") for a in sys.argv[1:] if "=" in a)          # e.g. kinv_tile=64
for k, v in tunes.items():
    eng.tune(k, int(v))
import time  # noqa: E402
eng.lml(theta, eval_gradient=True)
eng.sync()
t0 = time.perf_counter()
for _ in range(reps):
    eng.lml(theta, eval_gradient=True)
eng.sync()
print({"N": N, "P": P, "tune": tunes, "ms_per_lml_grad": round((time.perf_counter() - t0) / reps * 1e3, 3)})
