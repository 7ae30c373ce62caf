#!/usr/bin/env python3
"""Randomised parity soak (GPU box, imports the oracle: a test tool, not product): GPs of random shape — N 3..600 design points,
d 1..40 inputs, P 1..9 GPs, the three kernel families, W 1..500 query points, theta drawn inside the ranges the tests use and
sometimes at the edges of the reference's search box — predict (mean 1e-11 of the largest, variance 1e-10 relative) and
LML + gradient (1e-10 / 1e-9) against oracle/gp_oracle.py.  Prints every violation and a summary.
usage: gpu_parity_soak.py [cases=150] [seed=0]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpbayestools_hic_amd import GPEngine, synth  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402


def relerr(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), np.finfo(float).tiny))) if a.size else 0.0


def maxrel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    kinds = ["RBF", "Matern15", "Matern25"]
    bad, worst = [], {"mean": 0.0, "var": 0.0, "lml": 0.0, "grad": 0.0}
    t0 = time.time()
    eng = GPEngine(0)
    for c in range(cases):
        N = int(rng.choice([3, 7, 33, 63, 64, 65, 100, 127, 128, 129, 200, 257, 400, 600]))
        d = int(rng.choice([1, 2, 3, 5, 8, 9, 15, 16, 17, 20, 24, 31, 33, 40]))
        P = int(rng.integers(1, 10))
        W = int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129, 255, 300, 500]))
        kind = kinds[int(rng.integers(0, 3))]
        kid = O.KIND_NAMES[kind]
        X = synth.lhs(N, d, seed=1000 + c)
        Z = np.sin(X @ rng.standard_normal((d, P))).T + 0.05 * rng.standard_normal((P, N))
        edge = rng.random() < 0.25
        lo_l = 0.1 if kind == "RBF" else (1e-3 if edge else 0.05)        # (x the unit extent of the design)
        th = np.array([np.concatenate([[rng.uniform(-0.5, 0.7)],
                                       np.log(rng.uniform(0.5, 3.0, d) if not edge else
                                              np.where(rng.random(d) < 0.2, lo_l, rng.uniform(0.5, 3.0, d))),
                                       [np.log(rng.uniform(0.02, 0.2))]]) for _ in range(P)])
        Xs = rng.random((W, d))
        tag = dict(case=c, N=N, d=d, P=P, W=W, kind=kind, edge=bool(edge))
        try:
            eng.set_data(X, Z, kind, alpha=0.1); eng.set_theta(th); eng.factor()
            m, v = eng.predict(Xs)
            val, grad = eng.lml(th)
            eng.set_theta(th); eng.factor()
        except Exception as e:                       # a refusal or an error is a finding too
            bad.append(dict(tag, error="%s: %s" % (type(e).__name__, e)))
            print(json.dumps(bad[-1]), flush=True)
            continue
        for p in range(P):
            L, a = O.gp_factor(X, Z[p], th[p], kid, 0.1)
            mo, vo = O.gp_predict(Xs, X, th[p], L, a, kid)
            vl, gl = O.lml(th[p], X, Z[p], kid, 0.1, eval_gradient=True)
            e = {"mean": maxrel(m[:, p], mo), "var": relerr(v[:, p], vo), "lml": abs(val[p] - vl) / abs(vl),
                 "grad": maxrel(grad[p], gl)}
            lim = {"mean": 1e-11, "var": 1e-10, "lml": 1e-10, "grad": 1e-9}
            for k in e:
                worst[k] = max(worst[k], e[k])
                if not e[k] < lim[k]:
                    bad.append(dict(tag, gp=p, what=k, err=e[k], form=float(eng.get("form")[p])))
                    print(json.dumps(bad[-1]), flush=True)
        if c % 25 == 24:
            print(json.dumps({"done": c + 1, "seconds": round(time.time() - t0, 1), "violations": len(bad), "worst": worst}), flush=True)
    print(json.dumps({"cases": cases, "seed": seed, "violations": len(bad), "worst": worst, "seconds": round(time.time() - t0, 1)}))


if __name__ == "__main__":
    main()
