#!/usr/bin/env python3
"""The nine-emulator chain of bench.py's extras.nine_emulator_chain (nine emulators, N = 1000 each, 63 GPs, 540 observables, 20
parameters, 4096 walkers; src/mcmc.py:153-166, examples/RunBayesianAnalysis.ipynb:35-48), burnt-in ensemble, STEPS stretch-move steps
through gpb_chain_emcee_run — the workload to put under `rocprofv3 --kernel-trace --stats` for the per-kernel split of a half-step:

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_multi -o run -- python3 tools/gpu_multi_chain_profile.py 30
    python3 tools/kernel_trace_summary.py gpurun_out/prof_multi/*/run_kernel_trace.csv > profiles/r05_multi_chain_kernel_stats.csv

Prints one JSON line: ms per step, the predict launches' HIP-event time per step, and the share of the step outside them."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpbayestools_hic_amd import StretchSampler, synth  # noqa: E402
from gpbayestools_hic_amd.workload import build_multi_chain  # noqa: E402


def main():
    import torch
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    nw, d = 4096, 20
    for a in sys.argv[2:]:
        if a.startswith("--walkers="):
            nw = int(a[10:])
    specs = [(1000, 60, 6 + i % 3, ("RBF", "Matern25", "RBF")[i % 3]) for i in range(9)]
    mapped = "--mapped" in sys.argv[2:]          # every emulator with parameterTrafoPCA: nine parameter maps per half-step
    chain, emus, info = build_multi_chain(specs, d, mapped=mapped)
    for a in sys.argv[2:]:
        if a.startswith("--tune="):
            k, v = a[7:].split(":")
            for e in emus:
                e._engine_ready().tune(k, int(v))
    gps = sum(s[2] for s in specs)
    X0 = synth.walkers_ball(nw, info["xstar"], 1e-10)
    heat = StretchSampler(chain, nw, seed=6)
    heat.run(X0, 20, status=10 ** 9, store=False)
    del heat
    s = StretchSampler(chain, nw, seed=5)
    assert s._resident_engine() is not None
    s.run(X0, 3, status=10 ** 9, store=False)
    e0 = emus[0]._engine_ready()
    e0.profile(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.run(None, steps, status=10 ** 9, store=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    n_l, ms_l, u_l = e0.profile_read()
    e0.profile(False)
    rows = u_l / gps
    print(json.dumps({"walkers": nw, "steps": steps, "parameterTrafoPCA": mapped, "ms_per_step": dt * 1e3, "predict_launches_per_step": n_l / steps,
                      "predict_ms_per_step": ms_l / steps, "outside_predict_ms_per_step": dt * 1e3 - ms_l / steps,
                      "outside_predict_share": 1.0 - ms_l / steps / (dt * 1e3),
                      "rows_inside_box_fraction": rows / (nw * steps),
                      "k_predict_frac_of_peak_by_N2": gps * rows * 1000.0 ** 2 / (ms_l * 1e-3) / 1e12 / 78.6}), flush=True)


if __name__ == "__main__":
    main()
