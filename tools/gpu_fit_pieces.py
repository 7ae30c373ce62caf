#!/usr/bin/env python3
"""Time the pieces of gpb_gp_factor apart (gpb_profile_fit_piece: K build, Cholesky, triangular inverse, alpha) at the design
sizes of BASELINE configs 2, 4 and 5, 10 GPs, with A/B over the K-build kernel (tune kmat_mfma) and the Cholesky schedule
(tune chol_algo).    python tools/gpu_fit_pieces.py [N ...]"""
import json
import os
os.environ.setdefault("GPB_DEBUG_LIB", "1")      # the sweeps switch to kernel variants of the debug build
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpbayestools_hic_amd import GPEngine, synth  # noqa: E402


def timed(fn, reps=10):
    import torch
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3          # us


def main():
    P = 10
    sizes = [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096]
    for N in sizes:
        d = 15 if N == 1024 else 20
        kernel = "Matern25" if N == 4096 else "RBF"
        eng = GPEngine(0)
        eng.set_data(synth.lhs(N, d), np.random.default_rng(1).standard_normal((P, N)), kernel, 0.1)
        eng.set_theta(synth.fixed_theta(d, P))
        eng.factor()
        row = {"N": N, "P": P, "d": d, "kernel": kernel}
        for km in (0, 1, 0, 1):
            eng.tune("kmat_mfma", km)
            row.setdefault(f"kmat_mfma{km}_us", []).append(round(timed(lambda: eng.fit_piece("kmat")), 1))
        Kref = None
        for algo in (0, 1, 0, 1):
            eng.tune("chol_algo", algo)

            def chol():
                eng.fit_piece("kmat"); eng.fit_piece("potrf")
            t = timed(chol) - row["kmat_mfma1_us"][-1]
            row.setdefault(f"potrf_algo{algo}_us", []).append(round(t, 1))
        eng.tune("chol_algo", 1)
        for tile in (0, 64, 128):                        # tile of the end-of-panel trailing updates (0 = rule)
            eng.tune("syrk_tile", tile)

            def chol():
                eng.fit_piece("kmat"); eng.fit_piece("potrf")
            row[f"potrf_syrk_tile{tile}_us"] = round(timed(chol) - row["kmat_mfma1_us"][-1], 1)
        eng.tune("syrk_tile", 0)
        eng.factor()
        for tile in (0, 64, 128):
            eng.tune("trtri_tile", tile)
            row[f"trtri_tile{tile}_us"] = round(timed(lambda: eng.fit_piece("trtri")), 1)
        eng.tune("trtri_tile", 0)
        row["trtri_us"] = round(timed(lambda: eng.fit_piece("trtri")), 1)
        row["alpha_us"] = round(timed(lambda: eng.fit_piece("alpha")), 1)
        row["factor_ms"] = round(timed(lambda: eng.factor(), 5) / 1e3, 3)
        bytes_k = 4.0 * eng.N * eng.N * P                 # lower triangle, 8 bytes per entry
        fpair = (3 * d + 3) if kernel == "RBF" else (3 * d + 10)
        flops_k = 0.5 * N * N * fpair * P
        best = min(row["kmat_mfma1_us"])
        row["kmat_bound_us"] = round(max(bytes_k / 8e12, flops_k / 78.6e12) * 1e6, 1)
        row["kmat_frac_of_bound"] = round(row["kmat_bound_us"] / best, 3)
        print(json.dumps(row), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
