#!/usr/bin/env python3
"""Emulator.predict(numpy in, numpy out) at cfg 2's 10 000 points: page-locked result arrays (the default for large
results, _native.host_empty) against fresh pageable ones."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpbayestools_hic_amd import synth  # noqa: E402
from gpbayestools_hic_amd.workload import build_chain  # noqa: E402
import gpbayestools_hic_amd._native as nat  # noqa: E402

_, emu2, info2 = build_chain(2)
Xh = synth.walkers(10000, info2["d"])
for i in range(5):
    t0 = time.perf_counter(); m, c = emu2.predict(Xh, return_cov=True, extra_std=0.0); th = time.perf_counter() - t0
    print("page-locked result, call", i, "ms", round(th * 1e3, 2), c.shape, flush=True)
    if i == 2:
        del m, c
pinned = nat.host_empty
nat.host_empty = lambda shape, pinned_from=0: np.empty(shape)
for i in range(3):
    t0 = time.perf_counter(); m2, c2 = emu2.predict(Xh, return_cov=True, extra_std=0.0); th = time.perf_counter() - t0
    print("pageable result, call", i, "ms", round(th * 1e3, 2), flush=True)
nat.host_empty = pinned
print("same numbers", np.array_equal(c2, c), np.array_equal(m2, m))
