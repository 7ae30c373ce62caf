#!/usr/bin/env python3
"""Randomised soak of the whole log-posterior path (files in the reference's formats -> Emulator -> Chain -> gpb_logpost) against the
oracle's restatement (oracle/gp_oracle.py: OracleEmulator + log_prob): random N, d, M, npc, kernel family, fixed theta, walkers drawn
around the box so that some rows fall outside it; bar: |got - ref| <= 1e-10 max(|ref|, 1) on every finite row (a log-posterior that
happens to cross zero has no relative accuracy: at |ref| = 0.003 the oracle's own lean and faithful forms differ by 1.4e-10 relative,
profiles/r04_parity_soak.txt; the purely relative error is reported beside it), -inf exactly where the oracle has it.
A test tool (it imports the oracle), not product.  usage: gpu_posterior_soak.py [cases=40] [seed=0]"""
import json, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpbayestools_hic_amd import synth  # noqa: E402
from gpbayestools_hic_amd.workload import build_chain  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    bad, worst, worst_rel, t0 = [], 0.0, 0.0, time.time()
    for c in range(cases):
        N = int(rng.choice([20, 64, 100, 129, 250, 400]))
        d = int(rng.choice([2, 3, 5, 8, 15, 20]))
        M = int(rng.choice([3, 8, 17, 32, 64, 80]))
        P = int(rng.integers(2, min(M, 12) + 1))
        kernel = ["RBF", "Matern15", "Matern25"][int(rng.integers(0, 3))]
        W = int(rng.choice([1, 7, 64, 130, 300]))
        synth.CONFIGS[99] = dict(N=N, d=d, M=M, P=P, kernel=kernel, W=W)
        tag = dict(case=c, N=N, d=d, M=M, P=P, kernel=kernel, W=W)
        with tempfile.TemporaryDirectory() as wd:
            try:
                chain, emu, info = build_chain(99, workdir=wd)
                X = rng.uniform(-0.15, 1.15, size=(W, d)) if rng.random() < 0.7 else synth.walkers(W, d, seed=c)
                got = np.asarray(chain.log_posterior(X))
                got_ll = np.asarray(chain.log_likelihood(X, finite=True))
            except Exception as e:
                bad.append(dict(tag, error="%s: %s" % (type(e).__name__, e))); print(json.dumps(bad[-1]), flush=True); continue
            oe = O.OracleEmulator(info["X"], info["Y"], info["lo"], info["hi"], P, kind=O.KIND_NAMES[kernel]).fit(synth.fixed_theta(d, P))
            yexp = info["yexp"]; cexp = np.diag((0.05 * np.abs(yexp)) ** 2)
            pf = lambda x, e: oe.predict(x, True, e)
            ref = O.log_prob(X, info["lo"], info["hi"], pf, yexp, cexp)
            ref_ll = O.log_prob(X, info["lo"], info["hi"], pf, yexp, cexp, finite=True, posterior=False)
            for name, g, r in (("log_posterior", got, ref), ("log_likelihood(finite)", got_ll, ref_ll)):
                fin = np.isfinite(r) & (np.abs(r) < 1e299)
                if not np.array_equal(g[~fin], r[~fin]):
                    bad.append(dict(tag, what=name, err="rows outside the box differ")); print(json.dumps(bad[-1]), flush=True)
                if fin.any():
                    e = float(np.max(np.abs(g[fin] - r[fin]) / np.maximum(np.abs(r[fin]), 1.0)))
                    worst = max(worst, e)
                    worst_rel = max(worst_rel, float(np.max(np.abs(g[fin] - r[fin]) / np.abs(r[fin]))))
                    if not e < 1e-10:
                        bad.append(dict(tag, what=name, err=e)); print(json.dumps(bad[-1]), flush=True)
        if c % 10 == 9:
            print(json.dumps({"done": c + 1, "violations": len(bad), "worst": worst, "seconds": round(time.time() - t0, 1)}), flush=True)
    print(json.dumps({"cases": cases, "seed": seed, "violations": len(bad), "worst_err_over_max(|ref|,1)": worst, "worst_purely_relative": worst_rel, "seconds": round(time.time() - t0, 1)}))


if __name__ == "__main__":
    main()
