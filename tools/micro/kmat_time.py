"""K(X,X) assembly alone (gpb_profile_fit_piece 'kmat'), sustained: reps launches back to back between two events, at
BASELINE cfg 4 (N = 2048, RBF) and cfg 5 (N = 4096, Matern-5/2) shapes, 10 GPs, d = 20.  usage: kmat_time.py [reps=3000]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gpbayestools_hic_amd import GPEngine, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for N, kind in ((2048, "RBF"), (4096, "Matern25"), (1024, "RBF"), (4096, "RBF"), (2048, "Matern15")):
    d, P = 20, 10
    eng = GPEngine(0)
    eng.set_data(synth.lhs(N, d), np.random.default_rng(1).standard_normal((P, N)), kind, 0.1)
    eng.set_theta(synth.fixed_theta(d, P))
    for _ in range(20):
        eng.fit_piece("kmat")
    torch.cuda.synchronize()
    n = max(50, reps * (2048 * 2048) // (N * N))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = torch.cuda.current_stream()
    e0.record(st)
    for _ in range(n):
        eng.fit_piece("kmat")
    e1.record(st); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print("N %d %-9s %4d GPs: %7.1f us per K build (%d launches)   %.3f of 4 N^2 P bytes at 8 TB/s" %
          (N, kind, P, us, n, 4.0 * N * N * P / 8e12 / (us * 1e-6)), flush=True)
    eng.factor(); eng.close()
