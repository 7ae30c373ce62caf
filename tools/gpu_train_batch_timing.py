#!/usr/bin/env python3
"""Nine emulators of a chain (N = 1000 each, 20 parameters, 60 observables, 6-8 GPs: the reference's analysis size,
examples/RunBayesianAnalysis.ipynb:35-48) trained with their full hyper-parameter searches: one after the other
(Emulator.trainEmulator per emulator, as examples/EmulatorTraining.ipynb:124-138 does) against train_emulators (all 63
searches in one lock-step batch).  Prints wall times and whether theta* is identical.

    python tools/gpu_train_batch_timing.py [nrestarts] [N]"""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpbayestools_hic_amd import Emulator, synth  # noqa: E402
from gpbayestools_hic_amd.emulator import train_emulators  # noqa: E402


def make(wd, nrestarts, N, d=20, M=60):
    emus = []
    pf = os.path.join(wd, "par.txt")
    synth.write_parameter_file(pf, np.zeros(d), np.ones(d))
    for i in range(9):
        X = synth.lhs(N, d, seed=synth.SEED + 100 + i)
        Y = synth.observables(X, M, seed=synth.SEED + 200 + i)
        tp = os.path.join(wd, "t%d.pkl" % i)
        synth.write_training_pickle(tp, X, Y, 0.01)
        emus.append(Emulator(training_set_path=tp, parameter_file=pf, npc=6 + i % 3, nrestarts=nrestarts))
    return emus


def main():
    import torch
    nrestarts = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    torch.zeros(1, device="cuda"); torch.cuda.synchronize()
    warm = make(tempfile.mkdtemp(), 0, 128)[:1]          # page in scipy, the kernels and the thread pool before timing
    warm[0].trainEmulatorAutoMask()
    out = {"emulators": 9, "N": N, "d": 20, "gps": sum(6 + i % 3 for i in range(9)), "nrestarts": nrestarts}
    seq = make(tempfile.mkdtemp(), nrestarts, N)
    np.random.seed(5)
    t0 = time.perf_counter()
    for e in seq:
        e.trainEmulatorAutoMask()
    torch.cuda.synchronize()
    out["one_after_the_other_s"] = round(time.perf_counter() - t0, 3)
    tog = make(tempfile.mkdtemp(), nrestarts, N)
    np.random.seed(5)
    t0 = time.perf_counter()
    train_emulators(tog)
    torch.cuda.synchronize()
    out["train_emulators_s"] = round(time.perf_counter() - t0, 3)
    out["speedup"] = round(out["one_after_the_other_s"] / out["train_emulators_s"], 2)
    out["theta_identical"] = bool(all(np.array_equal(a.thetas_, b.thetas_) for a, b in zip(seq, tog)))
    out["lml_identical"] = bool(all(np.array_equal(np.asarray(a.lml_), np.asarray(b.lml_)) for a, b in zip(seq, tog)))
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
