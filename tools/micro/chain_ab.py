#!/usr/bin/env python3
"""A/B in ONE process of two builds of the product library on whole step loops: tools/micro/_bin/libgpbayes_before.so (built from
the previous commit's sources) against the in-tree libgpbayes.so, alternating, same seeds — the nine-emulator chain (plain and
with parameterTrafoPCA on every emulator) and BASELINE config 3 / 4.  Prints one JSON line per workload: ms per step of each
round, the gain, and whether both builds leave the same ensemble bit for bit.
usage: chain_ab.py [nine] [nine-mapped] [cfg3] [cfg4]   (default: all four)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gpbayestools_hic_amd import _native as nat, synth, StretchSampler
from gpbayestools_hic_amd.workload import build_chain, build_multi_chain

BEFORE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", "libgpbayes_before.so")
AFTER = nat.LIB_PATHS[False]
SPECS = [(1000, 60, 6 + i % 3, ("RBF", "Matern25", "RBF")[i % 3]) for i in range(9)]


def bind(path):
    nat.LIB_PATHS[False] = path
    nat._libs[False] = None                        # (the loader caches one product library per process: bind this path afresh)


def build(name):
    if name.startswith("nine"):
        chain, emus, info = build_multi_chain(SPECS, 20, mapped=name == "nine-mapped")
        return chain, emus[0], info, 4096, 1e-10
    cfg = int(name[3:])
    chain, emu, info = build_chain(cfg)
    return chain, emu, info, 2 * info["W"], 1e-7


def ms_per_step(chain, nw, X0, steps):
    s = StretchSampler(chain, nw, seed=5)
    assert s._resident_engine() is not None
    s.run(X0, 3, status=10 ** 9, store=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.run(None, steps, status=10 ** 9, store=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, s


def main():
    names = [a for a in sys.argv[1:] if not a.startswith("-")] or ["nine", "nine-mapped", "cfg3", "cfg4"]
    for name in names:
        bind(BEFORE); ca, ea, info, nw, ball = build(name)
        bind(AFTER); cb, eb, _, _, _ = build(name)
        assert ea._engine_ready().lib is not eb._engine_ready().lib, "the two builds must be two libraries"
        X0 = synth.walkers_ball(nw, info["xstar"], ball, lo=info["lo"], hi=info["hi"])
        steps = 30 if name.startswith("nine") else (200 if name == "cfg3" else 40)
        for c in (ca, cb):                          # clocks up
            ms_per_step(c, nw, X0, steps)
        row = {"workload": name, "walkers": nw, "steps": steps, "before_ms": [], "after_ms": []}
        for _ in range(4):
            a, sa = ms_per_step(ca, nw, X0, steps)
            b, sb = ms_per_step(cb, nw, X0, steps)
            row["before_ms"].append(round(a, 4)); row["after_ms"].append(round(b, 4))
        row["same_bits"] = bool(torch.equal(sa.pos, sb.pos) and torch.equal(sa.lp, sb.lp))
        row["gain_percent"] = round(100.0 * (1.0 - min(row["after_ms"]) / min(row["before_ms"])), 2)
        print(json.dumps(row), flush=True)
        del sa, sb, ca, cb, ea, eb
        import gc; gc.collect()


if __name__ == "__main__":
    main()
