#!/usr/bin/env python3
"""scipy.linalg.svd of a 1000 x 60 (and 2048 x 64) matrix against the number of BLAS threads (threadpoolctl)."""
import time
import numpy as np
from scipy import linalg
from threadpoolctl import ThreadpoolController
ctl = ThreadpoolController()
print([(i["user_api"], i["internal_api"], i["num_threads"]) for i in ctl.info()])
rng = np.random.default_rng(0)
for shape in ((1000, 60), (2048, 64)):
    S = rng.standard_normal(shape)
    for lim in (None, 1, 2, 4, 8, 16):
        def run():
            t0 = time.perf_counter(); linalg.svd(S, full_matrices=False); return (time.perf_counter() - t0) * 1e3
        if lim is None:
            ts = [run() for _ in range(5)]
        else:
            with ctl.limit(limits=lim, user_api="blas"):
                ts = [run() for _ in range(5)]
        t0 = time.perf_counter()
        with ctl.limit(limits=4, user_api="blas"):
            pass
        ctx_ms = (time.perf_counter() - t0) * 1e3
        print(shape, "threads", lim, "svd ms", round(sorted(ts)[2], 2), "(entering the limit context:", round(ctx_ms, 3), "ms)")
