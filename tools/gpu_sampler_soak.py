#!/usr/bin/env python3
"""Randomised soak of the device stretch-move sampler against the oracle's restatement of emcee's algorithm (oracle/stretch_oracle.py),
bit for bit: random ensemble sizes (even, 4..600), dimensions 1..12, seeds over the whole uint64 range, both split modes, 6 steps each on
the toy density of tests/test_gpu_sampler_step.py (hard wall at |x| = 4: proposals that are never accepted).  A test tool.
usage: gpu_sampler_soak.py [cases=60] [seed=0]"""
import json, os, sys, time, types
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_sampler_step import _oracle_chain, _toy  # noqa: E402


def main():
    import torch
    from gpbayestools_hic_amd import StretchSampler
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad, t0 = [], time.time()
    for c in range(cases):
        nw = 2 * int(rng.integers(2, 301))
        d = int(rng.integers(1, 13))
        seed = int(rng.integers(0, 2 ** 63)) * 2 + int(rng.integers(0, 2))
        randomize = bool(rng.integers(0, 2))
        fake = types.SimpleNamespace(ndim=d, device=0, min=np.full(d, -9.0), max=np.full(d, 9.0), emuList=[])

        def logprob_dev(X_dev, out):
            out.copy_(torch.as_tensor(_toy(X_dev.cpu().numpy()), device=out.device))
            return out
        X0 = rng.normal(size=(nw, d)) * 1.8
        s = StretchSampler(fake, nw, seed=seed, logprob_device=logprob_dev, randomize_split=randomize)
        s.run(X0, 6)
        X, lp, nacc, hist = _oracle_chain(X0, 6, seed, randomize, None, _toy)
        ok = all(np.array_equal(s.chain[:, n], Xn) and np.array_equal(s.lnprobability[:, n], lpn) for n, (Xn, lpn) in enumerate(hist))
        ok = ok and np.array_equal(s.naccept.cpu().numpy(), nacc)
        if not ok:
            bad.append(dict(case=c, nw=nw, d=d, seed=seed, randomize=randomize)); print(json.dumps(bad[-1]), flush=True)
    print(json.dumps({"cases": cases, "violations": len(bad), "seconds": round(time.time() - t0, 1)}))


if __name__ == "__main__":
    main()
