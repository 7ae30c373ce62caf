#!/usr/bin/env python3
"""One step of the C-driven sampling loop as plain launches and as replays of its HIP graph (gpb_debug_graph_probe),
BASELINE config 4, unsharded and as one rank's share of an 8-way split: what a graph would save."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPB_DEBUG_LIB", "1")      # measurement hooks and kernel variants: the debug library (libgpbayes_debug.so)


def main():
    import torch
    from gpbayestools_hic_amd import StretchSampler, synth
    from gpbayestools_hic_amd import _native as nat
    from gpbayestools_hic_amd.workload import build_chain
    chain, emu, info = build_chain(4)
    eng = emu._engine_ready()
    nw = 2 * info["W"]
    s = StretchSampler(chain, nw, seed=1)
    s.run(synth.walkers(nw, info["d"]), 5, status=10 ** 9, store=False)
    _, ctxs, E = s._resident_engine()
    lo, hi = chain._box(s.device)
    for ranks in (0, 8):
        eng.tune("sim_ranks", ranks)
        a, b = C.c_double(0.0), C.c_double(0.0)
        eng._ck(eng.lib.gpb_debug_graph_probe(ctxs, E, nat.ptr(s.pos), nat.ptr(s.lp), nw, 12345, 2.0, nat.ptr(lo), nat.ptr(hi),
                                              float("-inf"), chain.inside_const, 50, C.byref(a), C.byref(b)))
        print(json.dumps({"ranks_simulated": max(ranks, 1), "ms_per_step_plain_launches": round(a.value, 4),
                          "ms_per_step_graph_replay": round(b.value, 4)}), flush=True)
    eng.tune("sim_ranks", 0)


if __name__ == "__main__":
    main()
