#!/usr/bin/env python3
"""A/B in ONE process of two builds of the product library on LML + gradient evaluations (gpb_gp_lml): tools/micro/_bin/libgpbayes_before.so
against the in-tree libgpbayes.so, alternating; prints ms per evaluation of each round and the largest relative difference of the
gradients.  usage: lml_ab.py [N:P:kernel ...]   (default 1000:63:RBF 2048:10:RBF 1024:10:RBF 4096:10:Matern25 1000:63:Matern15)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gpbayestools_hic_amd import _native as nat, synth
from gpbayestools_hic_amd.engine import GPEngine

BEFORE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", "libgpbayes_before.so")
AFTER = nat.LIB_PATHS[False]


def engine(path, N, P, kernel, d=20):
    nat.LIB_PATHS[False] = path
    nat._libs[False] = None
    e = GPEngine(0)
    e.set_data(synth.lhs(N, d), np.random.default_rng(1).standard_normal((P, N)), kernel, 0.1)
    return e


def ms(e, theta, reps):
    e.lml(theta, eval_gradient=True); e.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        e.lml(theta, eval_gradient=True)
    e.sync()
    return round((time.perf_counter() - t0) / reps * 1e3, 4)


for spec in (sys.argv[1:] or ["1000:63:RBF", "2048:10:RBF", "1024:10:RBF", "4096:10:Matern25", "1000:63:Matern15"]):
    N, P, kernel = spec.split(":"); N, P = int(N), int(P)
    ea, eb = engine(BEFORE, N, P, kernel), engine(AFTER, N, P, kernel)
    assert ea.lib is not eb.lib
    theta = synth.fixed_theta(20, P) + 0.1 * np.random.default_rng(2).standard_normal((P, 22))
    reps = 20 if N <= 2048 else 6
    row = {"N": N, "P": P, "kernel": kernel, "before_ms": [], "after_ms": []}
    for _ in range(3):
        row["before_ms"].append(ms(ea, theta, reps)); row["after_ms"].append(ms(eb, theta, reps))
    (va, ga), (vb, gb) = ea.lml(theta, eval_gradient=True), eb.lml(theta, eval_gradient=True)
    row["lml_same_bits"] = bool(np.array_equal(va, vb))
    row["grad_max_rel_diff"] = float(np.max(np.abs(ga - gb) / np.maximum(np.abs(ga), 1e-300)))
    row["gain_percent"] = round(100.0 * (1.0 - min(row["after_ms"]) / min(row["before_ms"])), 2)
    print(json.dumps(row), flush=True)
    ea.close(); eb.close()
