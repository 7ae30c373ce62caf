#!/usr/bin/env python3
"""k_predict time vs walker-batch size (what a rank sees at 1/2/4/8-way sharding), A/B over the tile
shape (128x128, 64x64, 64x32 = "32"), the XCD affinity, the persistent workgroups per CU and the static
tile orders of fully resident grids."""
import json
import os
os.environ.setdefault("GPB_DEBUG_LIB", "1")      # the sweeps switch to kernel variants of the debug build
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpbayestools_hic_amd import GPEngine, synth  # noqa: E402


def timed(eng, Xs, reps=8):
    for _ in range(2):
        eng.predict(Xs)
    eng.profile(True)
    for _ in range(reps):
        eng.predict(Xs)
    n, ms, units = eng.profile_read()
    eng.profile(False)
    return ms / n, units / n


def main():
    import torch
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    sizes = [int(a) for a in sys.argv[2:]] or [128, 256, 512, 1024, 2048]
    c = synth.CONFIGS[cfg]
    N, d, P = c["N"], c["d"], c["P"]
    eng = GPEngine(0)
    eng.set_data(synth.lhs(N, d), np.random.default_rng(1).standard_normal((P, N)), c["kernel"], 0.1)
    eng.set_theta(synth.fixed_theta(d, P)); eng.factor()
    for W in sizes:
        Xs = torch.as_tensor(synth.walkers(W, d), device="cuda")
        row = {"W": W}
        ref = None
        for tile in (64, 128, 32):
            eng.force_tile(tile)
            for xcd in (0, 1):
                eng.tune("xcd", xcd)
                for res in (0, 1, 2, 3):
                    eng.tune("resident", res)
                    ms, units = timed(eng, Xs)
                    row[f"t{tile}x{xcd}r{res}"] = [round(ms, 4), round(units * N * N / (ms * 1e-3) / 1e12, 1)]
                    m, v = eng.predict(Xs)
                    if ref is None:
                        ref = v.clone()
                    elif not torch.equal(ref, v):
                        row[f"t{tile}x{xcd}r{res}_MISMATCH"] = True
        eng.tune("xcd", -1); eng.force_tile(0); eng.tune("resident", 0)
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
