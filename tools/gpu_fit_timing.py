#!/usr/bin/env python3
"""Fit at fixed hyper-parameters (K + blocked Cholesky + L^-1 + alpha, all P GPs): 64- vs 128-wide tiles for the
K=64 trailing updates inside an outer panel; checks that both give the same bits."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpbayestools_hic_amd import GPEngine, synth  # noqa: E402


def main():
    import torch
    P = 10
    for N, d in ((1024, 15), (2048, 20), (4096, 20)):
        eng = GPEngine(0)
        eng.set_data(synth.lhs(N, d), np.random.default_rng(1).standard_normal((P, N)), "RBF", 0.1)
        eng.set_theta(synth.fixed_theta(d, P))
        row = {"N": N, "P": P}
        Ls = {}
        for tile in (128, 64):
            eng.tune("chol_inner_tile", tile)
            for _ in range(2):
                eng.factor()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                eng.factor()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            row[f"inner{tile}_ms"] = round(ms, 3)
            row[f"inner{tile}_chol_equiv_tflops"] = round(P * N ** 3 / 3 / (ms * 1e-3) / 1e12, 2)
            if N <= 2048:
                Ls[tile] = eng.get("L"), eng.get("Linv")
        if Ls:
            row["bit_identical"] = bool(np.array_equal(Ls[64][0], Ls[128][0]) and np.array_equal(Ls[64][1], Ls[128][1]))
        print(json.dumps(row), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
