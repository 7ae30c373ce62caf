#!/usr/bin/env python3
"""Full-batch k_predict launch under two tile -> XCD maps: where the time goes (per-tile trace of the persistent kernel).
usage: gpu_tile_trace_full.py [W=2048] [xcd modes ...= -1 3]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPB_DEBUG_LIB", "1")      # measurement hooks and kernel variants: the debug library (libgpbayes_debug.so)
from gpbayestools_hic_amd import GPEngine, synth  # noqa: E402


def main():
    import torch
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    modes = sys.argv[2:] or ["0", "3"]
    c = synth.CONFIGS[4]
    N, d, P = c["N"], c["d"], c["P"]
    eng = GPEngine(0)
    eng.set_data(synth.lhs(N, d), np.random.default_rng(1).standard_normal((P, N)), c["kernel"], 0.1)
    eng.set_theta(synth.fixed_theta(d, P)); eng.factor()
    Xs = torch.as_tensor(synth.walkers(W, d), device="cuda")
    for rep in range(0 if os.environ.get("GPB_TRACE_DUMP") else 3):        # untraced: HIP events around the k_predict launches alone
        for m in modes:
            eng.tune("xcd", int(m))
            for _ in range(3):
                eng.predict(Xs)
            eng.profile(True)
            for _ in range(20):
                eng.predict(Xs)
            eng.sync()
            n_l, ms_l, _u = eng.profile_read()
            eng.profile(False)
            print(json.dumps({"xcd": m, "untraced_k_predict_us_per_launch": round(ms_l / n_l * 1e3, 1), "launches": n_l}), flush=True)
    for rep in range(2):
        for m in modes:
            eng.tune("xcd", int(m))
            for _ in range(3):
                eng.predict(Xs)
            eng.tile_trace(1 << 15)
            eng.predict(Xs)
            r = eng.tile_trace_read().astype(np.int64)
            eng.tile_trace(0)
            if os.environ.get("GPB_TRACE_DUMP"):
                np.save(os.path.join(os.environ["GPB_TRACE_DUMP"], "tiles_%s_%d.npy" % (m, rep)), r)
            hw, xcc, gp, ib, wt, t0, t1, blk = (r[:, i] for i in range(8))
            dt = ((t1 - t0) & 0xffffffff) / 100.0
            tmin = t0.min()
            s = ((t0 - tmin) & 0xffffffff) / 100.0
            e = s + dt
            span = float(e.max())
            nslot = len(set(blk.tolist()))
            last = {}
            first = {}
            for b, ss, ee in zip(blk.tolist(), s.tolist(), e.tolist()):
                last[b] = max(last.get(b, 0.0), ee)
                first[b] = min(first.get(b, 1e30), ss)
            tail = np.array([span - v for v in last.values()])
            head = np.array(list(first.values()))
            ksteps = (ib + 1) * 8.0                    # 128-row tiles, 16-deep K-steps
            out = {"xcd": m, "tiles": int(len(r)), "slots": nslot, "span_us": round(span, 1),
                   "busy_fraction": round(float(dt.sum() / (nslot * span)), 4),
                   "tail_idle_us_mean_max": [round(float(tail.mean()), 1), round(float(tail.max()), 1)],
                   "head_idle_us_mean_max": [round(float(head.mean()), 1), round(float(head.max()), 1)],
                   "us_per_kstep_by_rowblock": {int(i): round(float((dt[ib == i] / ksteps[ib == i]).mean()), 4)
                                                for i in sorted(set(ib.tolist()))},
                   "us_per_kstep_all": round(float(dt.sum() / ksteps.sum()), 4)}
            # how many tiles ran on an XCD other than their queue's (stolen)
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
