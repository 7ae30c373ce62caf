#!/usr/bin/env python3
"""First-contact GPU probe: fp64 issue rates and per-kernel timings of the hot path at a config."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPB_DEBUG_LIB", "1")      # measurement hooks and kernel variants: the debug library (libgpbayes_debug.so)
from gpbayestools_hic_amd import GPEngine, synth  # noqa: E402


def main():
    import torch
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    c = synth.CONFIGS[cfg]
    N, d, M, P, W = c["N"], c["d"], c["M"], c["P"], c["W"]
    eng = GPEngine(0)
    out = {"mfma_f64_tflops": eng.probe_fp64(0), "valu_f64_tflops": eng.probe_fp64(1),
           "both_tflops": eng.probe_fp64(2), "mfma_f64_cycles_per_instr": eng.probe_fp64(3),
           "clock_ghz_dense_mfma": eng.probe_fp64(4)}
    print(json.dumps(out), flush=True)
    X = synth.lhs(N, d)
    Z = np.random.default_rng(1).standard_normal((P, N))
    eng.set_data(X, Z, c["kernel"], 0.1)
    eng.set_theta(synth.fixed_theta(d, P))
    t0 = time.time(); eng.factor(); eng.sync(); t1 = time.time()
    t2 = time.time(); eng.factor(); eng.sync(); t3 = time.time()
    print(json.dumps({"factor_first_s": t1 - t0, "factor_s": t3 - t2,
                      "chol_gflops_equiv": P * N ** 3 / 3 / (t3 - t2) / 1e9}), flush=True)
    Xs = torch.as_tensor(synth.walkers(W, d), device="cuda")
    for _ in range(3):
        m, v = eng.predict(Xs)
    torch.cuda.synchronize()
    t0 = time.time()
    reps = 10
    for _ in range(reps):
        m, v = eng.predict(Xs)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / reps
    flops = P * W * (N * (3 * d + 3) + N * N + 4 * N)
    print(json.dumps({"predict_ms": dt * 1e3, "predict_tflops_alg": flops / dt / 1e12,
                      "walker_evals_per_s": W / dt}), flush=True)


if __name__ == "__main__":
    main()
