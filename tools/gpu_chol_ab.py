#!/usr/bin/env python3
"""Cholesky piece alone (product library), N = 1024 / 2048 / 4096, 10 GPs: microseconds per gpb_profile_fit_piece("potrf") with
the K build's time taken off, and the triangular inverse.     python tools/gpu_chol_ab.py [N ...] [P=63] [key=value ...]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpbayestools_hic_amd import GPEngine, synth  # noqa: E402


def timed(fn, reps=10):
    import torch
    fn(); fn(); torch.cuda.synchronize()
    t_heat = __import__("time").perf_counter()
    while __import__("time").perf_counter() - t_heat < 0.05:
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3          # us


def main():
    P = 10
    sizes = [int(a) for a in sys.argv[1:] if "=" not in a] or [1024, 2048, 4096]
    opts = [a.split("=") for a in sys.argv[1:] if "=" in a]
    for k, v in opts:
        if k == "P":                                   # number of GPs per launch (the batched regime of train_emulators: 63)
            P = int(v)
    opts = [kv for kv in opts if kv[0] != "P"]
    for N in sizes:
        d = 15 if N == 1024 else 20
        kernel = "Matern25" if N == 4096 else "RBF"
        eng = GPEngine(0)
        eng.set_data(synth.lhs(N, d), np.random.default_rng(1).standard_normal((P, N)), kernel, 0.1)
        eng.set_theta(synth.fixed_theta(d, P))
        for k, v in opts:
            eng.tune(k, int(v))
        eng.factor()
        kmat = min(timed(lambda: eng.fit_piece("kmat")) for _ in range(3))

        def chol():
            eng.fit_piece("kmat"); eng.fit_piece("potrf")
        row = {"N": N, "P": P, "opts": dict(opts), "kmat_us": round(kmat, 1),
               "potrf_us": [round(timed(chol) - kmat, 1) for _ in range(3)]}
        eng.factor()
        row["trtri_us"] = [round(timed(lambda: eng.fit_piece("trtri")), 1) for _ in range(2)]
        row["alpha_us"] = round(timed(lambda: eng.fit_piece("alpha")), 1)
        row["factor_ms"] = round(timed(lambda: eng.factor(), 5) / 1e3, 3)
        row["chol_frac_of_peak"] = round(P * N ** 3 / 3 / (min(row["potrf_us"]) * 1e-6) / 78.6e12, 4)
        print(json.dumps(row), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
