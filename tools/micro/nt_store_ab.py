"""A/B in ONE process: libgpbayes.so (plain stores) against libgpbayes_debug.so built with other store hints
(GPB_DEBUG_EXTRA_DEFINES="-DGPB_NT_KSTAR -DGPB_NT_K" python -m gpbayestools_hic_amd.build --debug-variants --force), alternating:
  * ms/step of the resident stretch-move loop, burnt-in ensemble, at BASELINE cfg 4 (4096 walkers) and cfg 3 (1024 walkers);
  * fit at fixed theta (gpb_gp_factor) and its K build alone at N = 1024 / 2048 / 4096, 10 GPs, d = 20.
usage: nt_store_ab.py [rounds=4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gpbayestools_hic_amd import GPEngine, StretchSampler, synth, _native as nat
from gpbayestools_hic_amd.workload import build_chain

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def make(cfg, nw):
    chain, emu, info = build_chain(cfg)
    X0 = synth.walkers_ball(nw, info["xstar"], 1e-9, lo=info["lo"], hi=info["hi"])
    s = StretchSampler(chain, nw, seed=12345)
    s.run(X0, 30, status=10 ** 9, store=False)
    return s


def steps(s, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s.run(None, n, status=10 ** 9, store=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


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


for cfg, nw, n in ((4, 4096, 40), (3, 1024, 200)):
    a = make(cfg, nw)
    with nat.debug_library():
        b = make(cfg, nw)
    assert a._engine().lib is not b._engine().lib
    res = {"plain": [], "hint": []}
    for r in range(rounds):
        res["plain"].append(steps(a, n)); res["hint"].append(steps(b, n))
    print("cfg %d, %d walkers, ms/step: plain %s | hint %s" % (cfg, nw, " ".join("%.4f" % x for x in res["plain"]),
                                                               " ".join("%.4f" % x for x in res["hint"])), flush=True)
    del a, b
for N, kind in ((1024, "RBF"), (2048, "RBF"), (4096, "Matern25")):
    ea = fit_engine(N, kind)
    with nat.debug_library():
        eb = fit_engine(N, kind)
    n = max(10, 200 * 1024 * 1024 // (N * N))
    out = {"plain": [], "hint": []}
    for r in range(rounds):
        for tag, e in (("plain", ea), ("hint", eb)):
            tf = timed(e.factor, n)
            tk = timed(lambda: e.fit_piece("kmat"), 4 * n); e.factor()
            out[tag].append((tf, tk * 1e3))
    for tag in ("plain", "hint"):
        print("N %d %s %-5s fit ms: %s | K build us: %s" % (N, kind, tag, " ".join("%.3f" % x[0] for x in out[tag]),
                                                            " ".join("%.1f" % x[1] for x in out[tag])), flush=True)
    ea.close(); eb.close()
