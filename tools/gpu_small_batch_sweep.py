#!/usr/bin/env python3
"""One rank's share of an 8-way sharded cfg-4 step (256 walkers per batch): launch-geometry knobs of the small kernels
around k_predict (k_kcross walkers per lane / chunks per workgroup, finalize fusion)."""
import os
os.environ.setdefault("GPB_DEBUG_LIB", "1")      # the sweeps switch to kernel variants of the debug build
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpu_shard_sim import timed  # noqa: E402


def main():
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd.sampler import StretchSampler
    from gpbayestools_hic_amd.workload import build_chain
    chain, emu, info = build_chain(4)
    eng = emu._engine_ready()
    nw = 2 * info["W"]
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    eng.tune("sim_ranks", world)
    s = StretchSampler(chain, nw, seed=1)
    s.run(synth.walkers(nw, info["d"]), 3, store=False, status=10 ** 9)
    base = {"kcross_wpl": 2, "kcross_chunks": 0, "fuse_finalize": 1}
    for knob, values in (("kcross_wpl", (2, 1)), ("kcross_chunks", (0, 1, 2, 4, 8)), ("fuse_finalize", (1, 0))):
        for v in values:
            for k, b in base.items():
                eng.tune(k, b)
            eng.tune(knob, v)
            print(json.dumps({"world": world, knob: v, "ms_per_step": round(timed(s, 60), 4)}), flush=True)
    for k, b in base.items():
        eng.tune(k, b)
    eng.tune("kcross_wpl", 1)
    for v in (1, 2, 4):
        eng.tune("kcross_chunks", v)
        print(json.dumps({"world": world, "kcross_wpl": 1, "kcross_chunks": v, "ms_per_step": round(timed(s, 60), 4)}), flush=True)


if __name__ == "__main__":
    main()
