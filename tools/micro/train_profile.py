#!/usr/bin/env python3
"""Where does the time of train_emulators go?  Nine emulators (N = 1000, 20 parameters, 63 GPs): wall time, the time inside the device
log-marginal-likelihood calls (GPEngine.lml / lml_subset, timed by wrapping them), and a cProfile of one call.  The measurement that
found the two host stalls of profiles/r04_train_batching.txt."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, "/root/repo")
from gpbayestools_hic_amd import Emulator, synth
from gpbayestools_hic_amd import emulator as E
from gpbayestools_hic_amd.engine import GPEngine
import torch
torch.zeros(1, device="cuda")
def make(wd, N=1000, d=20, M=60):
    pf = os.path.join(wd, "par.txt"); synth.write_parameter_file(pf, np.zeros(d), np.ones(d)); emus = []
    for i in range(9):
        X = synth.lhs(N, d, seed=synth.SEED + 100 + i); tp = os.path.join(wd, "t%d.pkl" % i)
        synth.write_training_pickle(tp, X, synth.observables(X, M, seed=synth.SEED + 200 + i), 0.01)
        emus.append(Emulator(training_set_path=tp, parameter_file=pf, npc=6 + i % 3))
    return emus
stat = {"calls": 0, "t": 0.0, "gps": 0}
for name in ("lml", "lml_subset"):
    orig = getattr(GPEngine, name)
    def wrap(self, *a, _o=orig, **k):
        t0 = time.perf_counter(); r = _o(self, *a, **k); stat["t"] += time.perf_counter() - t0; stat["calls"] += 1
        return r
    setattr(GPEngine, name, wrap)
GPEngine.lml_active = GPEngine.lml_subset
E.train_emulators(make(tempfile.mkdtemp(), N=128)[:2])
for k in stat: stat[k] = 0
emus = make(tempfile.mkdtemp())
t0 = time.perf_counter(); E.train_emulators(emus); dt = time.perf_counter() - t0
print("train_emulators", round(dt, 3), "s; device evaluation calls", stat["calls"], "time in them", round(stat["t"], 3), "s")
# coarse split of one train_emulators call
import cProfile, pstats
emus = make(tempfile.mkdtemp())
pr = cProfile.Profile(); pr.enable(); E.train_emulators(emus); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
