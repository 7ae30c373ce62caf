#!/usr/bin/env python3
"""A nine-emulator chain at the size of the reference's real analyses (RunBayesianAnalysis.ipynb:35-48: nine
emulators, ~540 observables) on one MI355X: log-posterior of 2048-row batches and stretch-move steps, the whole chain
through one C call (gpb_chain_logpost / gpb_chain_emcee_run: rows outside the box skipped, no Python between the
emulators) against the per-emulator calls sequenced from Python over all rows."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPB_DEBUG_LIB", "1")      # measurement hooks and kernel variants: the debug library (libgpbayes_debug.so)
from gpbayestools_hic_amd import StretchSampler, synth  # noqa: E402
from gpbayestools_hic_amd.workload import build_multi_chain  # noqa: E402


def main():
    import torch
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    d, nw = 20, 4096
    for a in sys.argv[2:]:
        if a.startswith("--walkers="):
            nw = int(a[10:])
    specs = [(N, 60, 6 + i % 3, ("RBF", "Matern25", "RBF")[i % 3]) for i in range(9)]
    chain, emus, info = build_multi_chain(specs, d)
    for a in sys.argv[2:]:
        if a.startswith("--tune="):                  # e.g. --tune=chain_batch:0 (set on every emulator's engine)
            k, v = a[7:].split(":")
            for e in emus:
                if k == "force_tile":                    # 0 = rule, 32 / 64 / 65 / 128 (65 = 64 rows x 128 walkers)
                    e._engine_ready().force_tile(int(v))
                else:
                    e._engine_ready().tune(k, int(v))
            print(json.dumps({"tune": {k: int(v)}}), flush=True)
    X = torch.as_tensor(synth.walkers(nw // 2, d), device="cuda")
    row = {"emulators": 9, "N": N, "d": d, "observables": chain.nobs, "GPs": sum(s[2] for s in specs), "rows": nw // 2}
    outs = {}
    for tag, flag in (("one_call", True), ("per_emulator_calls", False)):
        chain.use_chain_call = flag
        for _ in range(3):
            out = chain.log_prob_device(X)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            out = chain.log_prob_device(X)
        torch.cuda.synchronize()
        row["logpost_ms_" + tag] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
        outs[tag] = out.cpu().numpy()
    row["same_bits"] = bool(np.array_equal(outs["one_call"], outs["per_emulator_calls"]))
    row["rows_inside_box"] = int(np.isfinite(outs["one_call"]).sum())
    X0 = synth.walkers(nw, d)
    if "--ball" in sys.argv:                         # burnt-in start: every proposal row inside the box
        X0 = synth.walkers_ball(nw, info["xstar"], 1e-10)
        row["start"] = "ball"
    for tag, flag in (("c_loop", True), ("host_loop_per_emulator_calls", False)):
        chain.use_chain_call = flag
        s = StretchSampler(chain, nw, seed=1)
        assert (s._resident_engine() is not None) == flag
        s.run(X0, 5, status=10 ** 9, store=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = s.run(None, 20, status=10 ** 9, store=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        row["step_ms_" + tag] = round(dt * 1e3, 3)
        row["walker_evals_per_s_" + tag] = round(nw / dt)
        outs[tag] = last
    row["same_ensemble"] = bool(np.array_equal(outs["c_loop"], outs["host_loop_per_emulator_calls"]))
    print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
