#!/usr/bin/env python3
"""Randomised soak of chains of SEVERAL emulators (the reference's Chain._predict over an emuList, src/mcmc.py:153-166; here one call of
the C ABI with batched cross / predict / likelihood launches) against the oracle's chain_predict + log_prob: 2..7 emulators with random
designs (N 30..260), observables (3..70 each), GPs (2..8), kernel families, over d = 2..12 parameters; rows around the box.  Bars:
|got - ref| <= 1e-10 max(|ref|, 1), -inf exactly where the oracle has it, and the same BITS with the batching switched off
(chain_batch 0) and sequenced per emulator from Python.  A test tool.  usage: gpu_multi_chain_soak.py [cases=40] [seed=0]"""
import json, os, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpbayestools_hic_amd import synth  # noqa: E402
from gpbayestools_hic_amd.workload import build_multi_chain  # noqa: E402
from test_gpu_multi_emulator import _oracle_chain  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad, worst, t0 = [], 0.0, time.time()
    for c in range(cases):
        E = int(rng.integers(2, 8)); D = int(rng.integers(2, 13))
        specs = []
        for _ in range(E):
            M = int(rng.integers(3, 71)); P = int(rng.integers(2, min(M, 8) + 1))
            specs.append((int(rng.choice([30, 64, 100, 128, 129, 200, 260])), M, P, ["RBF", "Matern15", "Matern25"][int(rng.integers(0, 3))]))
        W = int(rng.choice([1, 9, 64, 200, 700]))
        # one case in five: 20 parameters and parameterTrafoPCA on some of the emulators (their maps in one launch, their cross
        # kernels grouped by PADDED input count).  The oracle restates no parameter PCA (its parity is G7's): these cases check
        # the rows outside the box and the bit-identity of the three forms only
        flags = [False] * E
        if rng.random() < 0.2:
            D = 20
            flags = [bool(rng.random() < 0.7) for _ in range(E)]
        tag = dict(case=c, E=E, D=D, W=W, specs=specs, mapped=flags)
        with tempfile.TemporaryDirectory() as wd:
            try:
                chain, emus, info = build_multi_chain(specs, D, workdir=wd, mapped=flags)
                X = rng.uniform(-0.1, 1.1, size=(W, D)) if rng.random() < 0.6 else np.clip(info["xstar"] + 0.05 * rng.standard_normal((W, D)), 0.0, 1.0)
                got = np.asarray(chain.log_posterior(X))
                if any(flags):
                    ins = np.all((X > 0.0) & (X < 1.0), axis=1)
                    ref, inside, fin = None, ins, np.zeros(W, dtype=bool)
                    if not np.all(np.isfinite(got[ins])):
                        bad.append(dict(tag, err="non-finite log-posterior inside the box")); print(json.dumps(bad[-1]), flush=True)
                else:
                    ref = _oracle_chain(info)(X)
                    inside = fin = np.isfinite(ref)
                if not np.array_equal(np.isneginf(got), ~inside):
                    bad.append(dict(tag, err="rows outside the box differ")); print(json.dumps(bad[-1]), flush=True)
                if fin.any():
                    e = float(np.max(np.abs(got[fin] - ref[fin]) / np.maximum(np.abs(ref[fin]), 1.0)))
                    worst = max(worst, e)
                    if not e < 1e-10:
                        bad.append(dict(tag, err=e)); print(json.dumps(bad[-1]), flush=True)
                engs = [e_._engine_ready() for e_ in emus]
                engs[0].tune("chain_batch", 0)
                same = np.array_equal(np.asarray(chain.log_posterior(X)), got)
                engs[0].tune("chain_batch", 1)
                chain.use_chain_call = False
                same = same and np.array_equal(np.asarray(chain.log_posterior(X)), got)
                chain.use_chain_call = True
                if not same:
                    bad.append(dict(tag, err="bits differ between the batched call and the per-emulator forms")); print(json.dumps(bad[-1]), flush=True)
                for g in engs:
                    g.close()
            except Exception as e:
                bad.append(dict(tag, error="%s: %s" % (type(e).__name__, str(e)[:300]))); print(json.dumps(bad[-1]), flush=True)
        if c % 10 == 9:
            print(json.dumps({"done": c + 1, "violations": len(bad), "worst": worst, "seconds": round(time.time() - t0, 1)}), flush=True)
    print(json.dumps({"cases": cases, "violations": len(bad), "worst_err_over_max(|ref|,1)": worst, "seconds": round(time.time() - t0, 1)}))


if __name__ == "__main__":
    main()
