#!/usr/bin/env python3
"""k_predict / k_kcross time vs walker-batch size (what a rank sees at 1/2/4/8-way sharding)."""
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
    for W in (128, 256, 512, 1024, 2048):
        Xs = torch.as_tensor(synth.walkers(W, d), device="cuda")
        row = {"W": W}
        for tile, xcd in ((64, 0), (64, 1), (64, 2), (128, 0), (128, 1), (128, 2)):
            eng.force_tile(tile)
            eng.tune("xcd", xcd)
            for _ in range(2):
                eng.predict(Xs)
            eng.profile(True)
            for _ in range(5):
                eng.predict(Xs)
            n, ms, units = eng.profile_read()
            eng.profile(False)
            row[f"t{tile}x{xcd}"] = [round(ms / n, 4), round(units / n * N * N / (ms / n * 1e-3) / 1e12, 1)]
        eng.tune("xcd", -1)
        # whole predict call (kcross + predict + finalize) with the automatic choice
        eng.force_tile(0)
        eng.tune("waves", 4); eng.tune("wgs64", 6)
        for _ in range(2):
            eng.predict(Xs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            eng.predict(Xs)
        e1.record(); torch.cuda.synchronize()
        row["predict_call_ms"] = round(e0.elapsed_time(e1) / 5, 4)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
