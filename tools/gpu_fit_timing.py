#!/usr/bin/env python3
"""Fit at fixed hyper-parameters (K build + blocked Cholesky + L^-1 + alpha, all P GPs): round 1's three-launch
schedule against the two-launch schedule with the next diagonal block fused into the update (gpb_chol.hip), and the
outer panel width.  Checks both factors against each other (not bit-identical: the diagonal block is factored with
different sub-block products) and the factorisation residual."""
import json
import os
os.environ.setdefault("GPB_DEBUG_LIB", "1")      # the sweeps switch to kernel variants of the debug build
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpbayestools_hic_amd import GPEngine, synth  # noqa: E402


def timed(eng, reps=5):
    import torch
    for _ in range(2):
        eng.factor()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.factor()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    P = 10
    sizes = ((1024, 15), (2048, 20), (4096, 20)) if len(sys.argv) < 2 else [(int(a), 20) for a in sys.argv[1:]]
    for N, d in sizes:
        eng = GPEngine(0)
        eng.set_data(synth.lhs(N, d), np.random.default_rng(1).standard_normal((P, N)), "RBF", 0.1)
        eng.set_theta(synth.fixed_theta(d, P))
        row = {"N": N, "P": P}
        Ls = {}
        for algo, outer, look in ((0, 512, 0), (1, 512, 0), (1, 512, 1), (1, 256, 0), (1, 256, 1), (1, 1024, 0), (1, 1024, 1),
                                  (1, 128, 0), (1, 128, 1), (1, 2048, 0), (1, 512, 1), (1, 256, 1)):
            if outer > N:
                continue
            eng.tune("chol_algo", algo)
            eng.tune("chol_outer", outer)
            eng.tune("chol_lookahead", look)
            ms = timed(eng)
            tag = f"algo{algo}_outer{outer}_look{look}"
            while tag + "_ms" in row:
                tag += "_again"
            row[tag + "_ms"] = round(ms, 3)
            row[tag + "_chol_equiv_tflops"] = round(P * N ** 3 / 3 / (ms * 1e-3) / 1e12, 2)
            if N <= 2048 and outer == 512 and look == (1 if algo else 0):
                Ls[algo] = eng.get("L"), eng.get("Linv")
        if Ls:
            L0, X0 = Ls[0]
            L1, X1 = Ls[1]
            row["L_maxrel_new_vs_r01"] = float(np.max(np.abs(L1 - L0)) / np.max(np.abs(L0)))
            row["Linv_maxrel_new_vs_r01"] = float(np.max(np.abs(X1 - X0)) / np.max(np.abs(X0)))
            v = np.random.default_rng(2).standard_normal((N, 3))
            row["Linv_L_minus_I"] = float(np.max(np.abs(X1[0] @ (L1[0] @ v) - v)))
        print(json.dumps(row), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
