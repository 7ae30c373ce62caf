#!/usr/bin/env python3
"""cProfile of Chain.log_posterior on a small batch (BASELINE config 1, 64 rows): where the host time of one call goes."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpbayestools_hic_amd import synth
from gpbayestools_hic_amd.workload import build_chain
chain, emu, info = build_chain(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
X = synth.walkers(64, info["d"], seed=3)
for _ in range(20):
    chain.log_posterior(X)
pr = cProfile.Profile()
pr.enable()
for _ in range(500):
    chain.log_posterior(X)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
