#!/usr/bin/env python3
"""A/B in ONE process of two builds of the product library on the cross kernel: tools/micro/_bin/libgpbayes_before.so (built from the
previous commit) against the in-tree libgpbayes.so, alternating: K*^T + mean partials (predict(..., return_var=False)) on full
batches of BASELINE config 4 / 3 / 5, and the headline step.  usage: kcross_ab.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gpbayestools_hic_amd import _native as nat, synth
from gpbayestools_hic_amd.engine import GPEngine

BEFORE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", "libgpbayes_before.so")
AFTER = nat.LIB_PATHS[False]


def engine(path, cfg):
    nat.LIB_PATHS[False] = path
    nat._libs[False] = None                        # (the loader caches one product library per process: bind this path afresh)
    c = synth.CONFIGS[cfg]
    e = GPEngine(0)
    e.set_data(synth.lhs(c["N"], c["d"]), np.random.default_rng(1).standard_normal((c["P"], c["N"])), c["kernel"], 0.1)
    e.set_theta(synth.fixed_theta(c["d"], c["P"])); e.factor()
    return e, c


def us(e, Xs, reps=30):
    for _ in range(5):
        e.predict(Xs, return_var=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        e.predict(Xs, return_var=False)
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps * 1e3, 1)


for cfg in (4, 3, 5):
    ea, c = engine(BEFORE, cfg)
    eb, _ = engine(AFTER, cfg)
    assert ea.lib is not eb.lib, "the two builds must be two libraries"
    for W in (256, 512, 2048, 8192):
        Xs = torch.as_tensor(synth.walkers(W, c["d"]), device="cuda")
        row = {"cfg": cfg, "W": W, "before_us": [], "after_us": []}
        for _ in range(3):
            row["before_us"].append(us(ea, Xs)); row["after_us"].append(us(eb, Xs))
        ma, va = ea.predict(Xs); mb, vb = eb.predict(Xs)
        row["same_bits"] = bool(torch.equal(ma, mb) and torch.equal(va, vb))
        row["gain_percent"] = round(100.0 * (1.0 - min(row["after_us"]) / min(row["before_us"])), 1)
        print(json.dumps(row), flush=True)
    ea.close(); eb.close()
