#!/usr/bin/env python3
"""k_predict time vs walker-batch size and tile size (what a rank sees at 1/2/4/8-way sharding)."""
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
        row = {"W": W}
        for tile, xcd in ((64, 0), (64, 1), (128, 0), (128, 1)):
            eng.force_tile(tile, -11 - xcd)
            for _ in range(2):
                eng.predict(Xs)
            eng.profile(True)
            for _ in range(5):
                eng.predict(Xs)
            n, ms, units = eng.profile_read()
            eng.profile(False)
            row[f"t{tile}x{xcd}_ms"] = round(ms / n, 4)
            row[f"t{tile}x{xcd}_tf"] = round(units / n * N * N / (ms / n * 1e-3) / 1e12, 1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
