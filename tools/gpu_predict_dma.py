#!/usr/bin/env python3
"""The 64-row predict tiles with LDS-DMA operand staging (tune predict_dma 1) against the register-staged tiles: same bits?
time per launch?  cfg 4 emulator, full batches: python tools/gpu_predict_dma.py"""
import json
import os
os.environ.setdefault("GPB_DEBUG_LIB", "1")      # the LDS-DMA tiles are a variant of the debug build
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpbayestools_hic_amd import GPEngine, synth  # noqa: E402


def main():
    import torch
    c = synth.CONFIGS[4]
    N, d, P = c["N"], c["d"], c["P"]
    eng = GPEngine(0)
    eng.set_data(synth.lhs(N, d), np.random.default_rng(1).standard_normal((P, N)), c["kernel"], 0.1)
    eng.set_theta(synth.fixed_theta(d, P)); eng.factor()
    for W, ft in ((256, 32), (512, 32), (512, 64), (1024, 64), (1024, 65), (2048, 65)):
        Xs = torch.as_tensor(synth.walkers(W, d), device="cuda")
        eng.force_tile(ft)
        row = {"W": W, "tile": {32: "64x32", 64: "64x64", 65: "64x128"}[ft]}
        out = {}
        for dma in (0, 1, 0, 1):
            eng.tune("predict_dma", dma)
            for _ in range(3):
                m, v = eng.predict(Xs)
            eng.profile(True)
            for _ in range(20):
                m, v = eng.predict(Xs)
            eng.sync()
            n_l, ms_l, _u = eng.profile_read()
            eng.profile(False)
            row.setdefault("dma_us" if dma else "reg_us", []).append(round(ms_l / n_l * 1e3, 1))
            out[dma] = (m.clone(), v.clone())
        row["same_bits"] = bool(torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]))
        row["max_abs_var_diff"] = float((out[0][1] - out[1][1]).abs().max())
        print(json.dumps(row), flush=True)
    eng.force_tile(0)


if __name__ == "__main__":
    main()
