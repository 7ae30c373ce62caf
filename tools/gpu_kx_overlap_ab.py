#!/usr/bin/env python3
"""A/B of option key 48 in ONE process (the review's item 6: K*^T of the second GP group under the first group's predict launch,
two streams and two events per batch): ms per stretch-move step, burnt-in ensemble, for key 48 = 0 (one K*^T launch, one predict
launch: the default) and several splits, alternating, on
  (a) BASELINE cfg 4 (N = 2048, 10 GPs, 4096 walkers), whole step and one rank's share of eight (sim_ranks hook: debug library);
  (b) the nine-emulator chain of bench.py's extras (63 GPs, N = 1000, 4096 walkers).
Same bits for every split (tests/test_gpu_multi_emulator.py).  usage: gpu_kx_overlap_ab.py [rounds=3]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPB_DEBUG_LIB", "1")
from gpbayestools_hic_amd import StretchSampler, synth  # noqa: E402
from gpbayestools_hic_amd.workload import build_chain, build_multi_chain  # noqa: E402

PCTS = (0, 50, 30, 70, 20)


def ms_per_step(chain, nw, X0, steps=40):
    import torch
    heat = StretchSampler(chain, nw, seed=99)
    heat.run(X0, 60, store=False, status=10 ** 9)
    del heat
    s = StretchSampler(chain, nw, seed=1)
    assert s._resident_engine() is not None
    s.run(X0, 5, store=False, status=10 ** 9)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.run(None, steps, store=False, status=10 ** 9)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    chain4, emu4, info4 = build_chain(4)
    e4 = emu4._engine_ready()
    nw = 2 * info4["W"]
    X04 = synth.walkers_ball(nw, info4["xstar"], 1e-13)
    specs = [(1000, 60, 6 + i % 3, ("RBF", "Matern25", "RBF")[i % 3]) for i in range(9)]
    chain9, emus9, info9 = build_multi_chain(specs, 20)
    e9 = emus9[0]._engine_ready()
    X09 = synth.walkers_ball(4096, info9["xstar"], 1e-13)
    res = {"cfg4_whole_step": {}, "cfg4_share_of_8": {}, "nine_emulator_chain": {}}
    for r in range(rounds):
        for pct in PCTS:
            e4.tune("kx_overlap", pct)
            res["cfg4_whole_step"].setdefault(pct, []).append(round(ms_per_step(chain4, nw, X04), 4))
            e4.tune("sim_ranks", 8)
            res["cfg4_share_of_8"].setdefault(pct, []).append(round(ms_per_step(chain4, nw, X04), 4))
            e4.tune("sim_ranks", 0)
            e9.tune("kx_overlap", pct)
            res["nine_emulator_chain"].setdefault(pct, []).append(round(ms_per_step(chain9, 4096, X09), 4))
        print(json.dumps({"round": r, "ms_per_step_by_percent_of_GPs_in_the_first_group": res}), flush=True)
    e4.tune("kx_overlap", 0); e9.tune("kx_overlap", 0)
    best = {k: {p: min(v) for p, v in d.items()} for k, d in res.items()}
    print(json.dumps({"best_of_rounds": best,
                      "gain_vs_0_percent": {k: {p: round(100.0 * (1.0 - t / d[0]), 2) for p, t in d.items() if p} for k, d in best.items()}}))


if __name__ == "__main__":
    main()
