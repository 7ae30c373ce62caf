#!/usr/bin/env python3
"""Randomised soak of the SHARDED resident step loop (gpb_chain_emcee_run with R > 1) on one GPU: R contexts joined by the debug
library's loopback communicator (everything of the multi-GPU loop but the wire) must end on the unsharded run's ensemble bit for bit —
random R 2..8, ensemble sizes with nw / 2 a multiple of R (the resident loop's condition; other sizes take the host-driven loop), random emulator shapes, walkers at the edge of the box (compacted batches,
ragged shares, ranks without a live row), both balance modes.  A test tool (debug library).  usage: gpu_shard_soak.py [cases=30] [seed=0]"""
import json, os, sys, tempfile, time
if __name__ == "__main__":                           # (imported by tests/test_gpu_soak.py under its debug_lib fixture instead)
    os.environ.setdefault("GPB_DEBUG_LIB", "1")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpbayestools_hic_amd import synth  # noqa: E402
from gpbayestools_hic_amd.workload import build_chain  # noqa: E402
from test_gpu_sampler import _loopback_ranks_reproduce_the_unsharded_run  # noqa: E402


def main():
    selftest = "--selftest" in sys.argv         # rank 1 gets a different experiment vector: the comparison must FAIL (the check is live)
    if selftest:
        sys.argv.remove("--selftest")
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad, t0 = [], time.time()
    for c in range(cases):
        R = int(rng.choice([2, 3, 4, 5, 8]))
        N = int(rng.choice([40, 100, 129, 200])); d = int(rng.choice([2, 4, 8, 13])); M = int(rng.choice([3, 6, 12, 20]))
        P = int(rng.integers(2, min(M, 6) + 1)); kernel = ["RBF", "Matern15", "Matern25"][int(rng.integers(0, 3))]
        nw = 2 * R * int(rng.integers(max(1, (d + R - 1) // R), max(2, 150 // R) + 1))      # the resident loop shards even splits: (nw / 2) % R == 0
        balance = int(rng.choice([0, 2]))
        nsteps = int(rng.integers(3, 7))
        synth.CONFIGS[99] = dict(N=N, d=d, M=M, P=P, kernel=kernel, W=nw)
        tag = dict(case=c, R=R, N=N, d=d, M=M, P=P, kernel=kernel, nw=nw, balance=balance, nsteps=nsteps)
        with tempfile.TemporaryDirectory() as wd:
            try:
                built = []
                for r in range(R):
                    os.mkdir(os.path.join(wd, "r%d" % r))
                    built.append(build_chain(99, workdir=os.path.join(wd, "r%d" % r)))
                mode = int(rng.integers(0, 3))
                if mode == 0:
                    X0 = synth.walkers(nw, d, seed=100 + c)                       # spread out: about half of the proposals leave the box
                elif mode == 1:
                    X0 = np.clip(built[0][2]["xstar"] + 0.02 * rng.standard_normal((nw, d)), 0.01, 0.99)      # burnt-in
                else:
                    X0 = synth.walkers(nw, d, seed=100 + c); X0[:: int(rng.integers(2, 6)), 0] = 0.9995     # many at the edge
                if selftest:
                    import torch
                    ch = built[1][0]; ch._prepare_blocks()
                    eng1 = built[1][1]._engine_ready()
                    yexp = built[1][2]["yexp"] * 1.01
                    eng1.set_likelihood(yexp, np.diag((0.05 * np.abs(yexp)) ** 2))
                _loopback_ranks_reproduce_the_unsharded_run(built, nw, nsteps, X0, balance)
                for b in built:
                    b[1]._engine_ready().close()
            except Exception as e:
                bad.append(dict(tag, error="%s: %s" % (type(e).__name__, str(e)[:300]))); print(json.dumps(bad[-1]), flush=True)
        if c % 10 == 9:
            print(json.dumps({"done": c + 1, "violations": len(bad), "seconds": round(time.time() - t0, 1)}), flush=True)
    print(json.dumps({"cases": cases, "violations": len(bad), "seconds": round(time.time() - t0, 1)}))


if __name__ == "__main__":
    main()
