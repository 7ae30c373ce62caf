#!/usr/bin/env python3
"""Where does a context's set-up time go?  gpb_gp_set / gpb_gp_set_multi / set_theta / factor / close, repeated."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpbayestools_hic_amd import GPEngine, synth
import torch
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
def t(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3, r
N, d = 1000, 20
X = synth.lhs(N, d); rng = np.random.default_rng(0)
for P in (7, 63):
    Z = rng.standard_normal((P, N)); th = synth.fixed_theta(d, P)
    for rep in range(3):
        a, eng = t(lambda: GPEngine(0))
        b, _ = t(lambda: eng.set_data(X, Z, "RBF", 0.1))
        c, _ = t(lambda: eng.set_theta(th))
        e, _ = t(lambda: eng.factor())
        f, _ = t(lambda: eng.lml(th))
        g, _ = t(lambda: eng.close())
        print(f"P={P:3d} single-design: ctor {a:.2f} set_data {b:.2f} set_theta {c:.2f} factor {e:.2f} lml {f:.2f} close {g:.2f} ms")
    for rep in range(3):
        a, eng = t(lambda: GPEngine(0))
        b, _ = t(lambda: eng.set_data_multi([X] * P, list(Z), "RBF", 0.1))
        f, _ = t(lambda: eng.lml(th))
        g, _ = t(lambda: eng.close())
        print(f"P={P:3d} multi:         ctor {a:.2f} set_data_multi {b:.2f} lml {f:.2f} close {g:.2f} ms")
