#!/usr/bin/env python3
"""Per-call host latency of the drop-in entry points on small batches (BASELINE config 1 and 4): what a caller that works row by row
or in small batches pays on top of the kernels."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpbayestools_hic_amd import synth, StretchSampler
from gpbayestools_hic_amd.workload import build_chain
import torch

def med(fn, n=200):
    for _ in range(10): fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return round(sorted(ts)[n // 2] * 1e6, 1)

for cfg in (1, 4):
    chain, emu, info = build_chain(cfg)
    d = info["d"]
    out = {"cfg": cfg}
    for W in (1, 64, 512):
        X = synth.walkers(W, d, seed=3)
        out[f"log_posterior_{W}_us"] = med(lambda: chain.log_posterior(X))
        out[f"predict_mean_{W}_us"] = med(lambda: emu.predict(X, return_cov=False))
        out[f"predict_cov_{W}_us"] = med(lambda: emu.predict(X, return_cov=True, extra_std=0.0))
    Xd = torch.as_tensor(synth.walkers(64, d, seed=3), device="cuda"); lp = torch.empty(64, dtype=torch.float64, device="cuda")
    def dev():
        chain.log_prob_device(Xd, lp); torch.cuda.synchronize()
    out["log_prob_device_64_us"] = med(dev)
    s = StretchSampler(chain, 64, seed=1)
    s.run(synth.walkers(64, d, seed=4), 2, status=10 ** 9, store=False)
    def step():
        s.run(None, 1, status=10 ** 9, store=False)
    out["sampler_run_1_step_us"] = med(step, 100)
    def step10():
        s.run(None, 10, status=10 ** 9, store=False)
    out["sampler_run_10_steps_us"] = med(step10, 50)
    print(json.dumps(out), flush=True)
