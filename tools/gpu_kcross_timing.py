#!/usr/bin/env python3
"""k_kcross alone (K*^T + mean partials; `predict(..., return_var=False)` = k_kcross + k_finalize) against the batch size, on
full (not compacted) batches of BASELINE config 4's emulator: python tools/gpu_kcross_timing.py [cfg]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpbayestools_hic_amd import GPEngine, synth  # noqa: E402


def main():
    import torch
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    c = synth.CONFIGS[cfg]
    N, d, P = c["N"], c["d"], c["P"]
    eng = GPEngine(0)
    eng.set_data(synth.lhs(N, d), np.random.default_rng(1).standard_normal((P, N)), c["kernel"], 0.1)
    eng.set_theta(synth.fixed_theta(d, P)); eng.factor()
    for W in (128, 256, 512, 1024, 2048, 4096):
        Xs = torch.as_tensor(synth.walkers(W, d), device="cuda")
        for _ in range(3):
            eng.predict(Xs, return_var=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            eng.predict(Xs, return_var=False)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(json.dumps({"W": W, "kcross_plus_finalize_us": round(us, 1), "pairs_per_us": round(P * N * W / us)}), flush=True)


if __name__ == "__main__":
    main()
