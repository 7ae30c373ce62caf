#!/usr/bin/env python3
"""Calibration point for BASELINE metric (iii): what the vendor's batched Cholesky (torch.linalg.cholesky -> hipSOLVER / rocSOLVER
potrf_batched) and triangular inverse (torch.linalg.solve_triangular against I) reach on this box at the shapes bench.py's
extras.fit_fixed_theta_* time — P SPD matrices of N x N, fp64 — next to this library's own pieces (gpb_profile_fit_piece).
Measurement only: nothing in the product path calls the vendor solver.   usage: gpu_potrf_reference.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gpbayestools_hic_amd import GPEngine, synth  # noqa: E402


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    for N, P, d in ((1024, 10, 15), (2048, 10, 20), (4096, 10, 20), (1024, 63, 20)):
        eng = GPEngine(0)
        eng.set_data(synth.lhs(N, d), np.random.default_rng(1).standard_normal((P, N)), "RBF", 0.1)
        eng.set_theta(synth.fixed_theta(d, P))
        eng.factor()
        # the very matrices the library factors: K(X,X) + (noise + alpha) I, rebuilt on the host side of torch from L
        L = torch.as_tensor(eng.get("L"), device="cuda")
        A = L @ L.transpose(1, 2)
        A = 0.5 * (A + A.transpose(1, 2))
        eye = torch.eye(N, dtype=torch.float64, device="cuda").expand(P, N, N).contiguous()
        t_v = timed(lambda: torch.linalg.cholesky(A), 5)
        Lv = torch.linalg.cholesky(A)
        t_i = timed(lambda: torch.linalg.solve_triangular(Lv, eye, upper=False), 3)
        t_own = {pc: timed(lambda pc=pc: eng.fit_piece(pc), 5) for pc in ("potrf", "trtri")}
        eng.factor()
        fl = P * N ** 3 / 3.0
        err = float((Lv - L).abs().max() / L.abs().max())
        print(json.dumps({"N": N, "matrices": P,
                          "vendor_cholesky_ms": round(t_v * 1e3, 3), "vendor_cholesky_tflops": round(fl / t_v / 1e12, 2),
                          "vendor_cholesky_frac_of_peak": round(fl / t_v / 1e12 / 78.6, 3),
                          "this_library_cholesky_ms": round(t_own["potrf"] * 1e3, 3),
                          "this_library_cholesky_frac_of_peak": round(fl / t_own["potrf"] / 1e12 / 78.6, 3),
                          "vendor_triangular_inverse_ms": round(t_i * 1e3, 3),
                          "this_library_triangular_inverse_ms": round(t_own["trtri"] * 1e3, 3),
                          "max_rel_diff_of_the_factors": err}), flush=True)
        eng.close()
        del A, L, Lv, eye
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
