#!/usr/bin/env python3
"""Wall time of a full Emulator.trainEmulatorAutoMask() (hyper-parameter search included) on synthetic data."""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpbayestools_hic_amd import Emulator, synth  # noqa: E402


def main():
    import torch                                    # page the runtime in before anything is timed
    torch.zeros(1, device="cuda"); torch.cuda.synchronize()
    for (N, d, M, npc) in ((1024, 15, 16, 4), (2048, 20, 64, 10)):
        wd = tempfile.mkdtemp()
        X = synth.lhs(N, d); Y = synth.observables(X, M)
        tp, pf = os.path.join(wd, "t.pkl"), os.path.join(wd, "p.txt")
        synth.write_training_pickle(tp, X, Y, 0.01)
        synth.write_parameter_file(pf, np.zeros(d), np.ones(d))
        emu = Emulator(training_set_path=tp, parameter_file=pf, npc=npc)
        t0 = time.time()
        emu.trainEmulatorAutoMask()
        dt = time.time() - t0
        eng = emu._engine_ready()
        t1 = time.time()
        for _ in range(5):
            eng.lml(emu.thetas_, eval_gradient=True)
        ev = (time.time() - t1) / 5
        print(json.dumps({"N": N, "d": d, "M": M, "npc": npc, "train_s": round(dt, 3),
                          "lml_grad_eval_ms_all_gps": round(ev * 1e3, 2), "lml": [round(float(v), 3) for v in emu.lml_],
                          "scores": [round(float(s), 4) for s in emu.gp_scores_]}), flush=True)


if __name__ == "__main__":
    main()
