#!/usr/bin/env python3
"""Where and when every k_predict tile of one launch ran (debug trace): per-CU load, start/finish spread, the
critical tiles.  usage: gpu_tile_trace.py [W=256] [tile=0] [resident=2]"""
import json
import os
os.environ.setdefault("GPB_DEBUG_LIB", "1")      # the sweeps switch to kernel variants of the debug build
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpbayestools_hic_amd import GPEngine, synth  # noqa: E402


def main():
    import torch
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    res = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    c = synth.CONFIGS[4]
    N, d, P = c["N"], c["d"], c["P"]
    eng = GPEngine(0)
    eng.set_data(synth.lhs(N, d), np.random.default_rng(1).standard_normal((P, N)), c["kernel"], 0.1)
    eng.set_theta(synth.fixed_theta(d, P)); eng.factor()
    eng.force_tile(tile); eng.tune("resident", res)
    Xs = torch.as_tensor(synth.walkers(W, d), device="cuda")
    for _ in range(3):
        eng.predict(Xs)
    eng.tile_trace(1 << 15)
    eng.predict(Xs)
    r = eng.tile_trace_read().astype(np.int64)
    eng.tile_trace(0)
    hw, xcc, gp, ib, wt, t0, t1 = (r[:, i] for i in range(7))
    cu = xcc * 64 + ((hw >> 13) & 7) * 16 + ((hw >> 8) & 15)          # dense enough: (XCC, SE, CU)
    tmin = t0.min()
    t0 = (t0 - tmin) / 100.0
    t1 = (t1 - tmin) / 100.0                                           # microseconds
    out = {"W": W, "tiles": int(len(r)), "span_us": round(float(t1.max()), 1),
           "start_spread_us": round(float(t0.max()), 1), "distinct_cus": int(len(set(cu.tolist())))}
    work = (ib + 1).astype(float)                                      # K-loop length in 64-row units (64-row tiles)
    per_cu_work, per_cu_end, per_cu_n = {}, {}, {}
    for c_, w_, e_ in zip(cu.tolist(), work.tolist(), t1.tolist()):
        per_cu_work[c_] = per_cu_work.get(c_, 0.0) + w_
        per_cu_end[c_] = max(per_cu_end.get(c_, 0.0), e_)
        per_cu_n[c_] = per_cu_n.get(c_, 0) + 1
    ws = np.array(list(per_cu_work.values())); es = np.array(list(per_cu_end.values())); ns = np.array(list(per_cu_n.values()))
    out["tiles_per_cu_min_max"] = [int(ns.min()), int(ns.max())]
    out["work_per_cu_min_mean_max"] = [round(float(ws.min()), 1), round(float(ws.mean()), 1), round(float(ws.max()), 1)]
    out["cu_finish_us_min_mean_max"] = [round(float(es.min()), 1), round(float(es.mean()), 1), round(float(es.max()), 1)]
    # correlation between a CU's summed work and its finish time; duration of the heaviest tiles
    out["corr_work_finish"] = round(float(np.corrcoef(ws, es)[0, 1]), 3)
    heavy = ib == ib.max()
    out["heaviest_tile_us_min_mean_max"] = [round(float((t1 - t0)[heavy].min()), 1), round(float((t1 - t0)[heavy].mean()), 1),
                                            round(float((t1 - t0)[heavy].max()), 1)]
    # time per K-step (16 columns) of a tile as a function of how long it ran
    ksteps = (ib + 1) * 4
    out["us_per_kstep_heaviest"] = round(float(((t1 - t0)[heavy] / ksteps[heavy]).mean()), 3)
    light = ib <= 3
    out["us_per_kstep_lightest"] = round(float(((t1 - t0)[light] / ksteps[light]).mean()), 3)
    print(json.dumps(out), flush=True)
    # the ten CUs that finish last: their tiles
    order = sorted(per_cu_end, key=per_cu_end.get)[-5:]
    for c_ in order:
        m = cu == c_
        print("CU", c_, "finish", round(per_cu_end[c_], 1), "tiles (ib, start, end):",
              sorted([(int(a), round(float(b), 1), round(float(e), 1)) for a, b, e in zip(ib[m], t0[m], t1[m])], reverse=True))


if __name__ == "__main__":
    main()
