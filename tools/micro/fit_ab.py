"""A/B in ONE process of the fit at fixed theta (gpb_gp_factor: K build, Cholesky, L^-1, alpha; 10 GPs, d = 20) and of its Cholesky
piece alone: libgpbayes.so against a libgpbayes_debug.so built with other defines (GPB_DEBUG_EXTRA_DEFINES=... python -m
gpbayestools_hic_amd.build --debug-variants --force), alternating.  usage: fit_ab.py [rounds=4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gpbayestools_hic_amd import GPEngine, synth, _native as nat

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def fit_engine(N, kind):
    eng = GPEngine(0)
    eng.set_data(synth.lhs(N, 20), np.random.default_rng(1).standard_normal((10, N)), kind, 0.1)
    eng.set_theta(synth.fixed_theta(20, 10))
    eng.factor()
    return eng


def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for N, kind in ((1024, "RBF"), (1536, "RBF"), (2048, "RBF"), (3072, "RBF"), (4096, "Matern25")):
    ea = fit_engine(N, kind)
    with nat.debug_library():
        eb = fit_engine(N, kind)
    n = max(10, 200 * 1024 * 1024 // (N * N))
    out = {"plain": [], "other": []}
    for r in range(rounds):
        for tag, e in (("plain", ea), ("other", eb)):
            tf = timed(e.factor, n)
            def chol():
                e.fit_piece("kmat"); e.fit_piece("potrf")
            tk = timed(lambda: e.fit_piece("kmat"), n)
            tc = timed(chol, n) - tk
            e.factor()
            out[tag].append((tf, tc))
    for tag in ("plain", "other"):
        print("N %d %-8s %-5s fit ms: %s | Cholesky alone ms: %s" % (N, kind, tag, " ".join("%.3f" % x[0] for x in out[tag]),
                                                                   " ".join("%.3f" % x[1] for x in out[tag])), flush=True)
    ea.close(); eb.close()
