#!/usr/bin/env python3
"""What would folding the block likelihood into k_predict's tail cost?  (debug library, option key 46: every predict tile releases
its partial sums and takes a ticket of its walker tile; the last tile of a walker tile acquires, reads that tile's partials and
runs a chain as long as one walker's P x P factorisation.)  HIP-event time of the predict launch with the probe off / on, next to
the separate likelihood kernel's time (the difference of a log-likelihood call and a predict call), for a rank's share of cfg 4
(256 rows), cfg 3's half-ensemble (512 rows) and a full cfg-4 batch (2048 rows)."""
import json
import os
import sys
import time

os.environ.setdefault("GPB_DEBUG_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpbayestools_hic_amd import synth  # noqa: E402
from gpbayestools_hic_amd.workload import build_chain  # noqa: E402


def main():
    import torch
    for cfg, W in ((4, 256), (3, 512), (4, 2048)):
        chain, emu, info = build_chain(cfg)
        eng = emu._engine_ready()
        chain._prepare_blocks()
        X = torch.as_tensor(synth.walkers(W, info["d"], seed=5), device="cuda")
        out = torch.empty(W, dtype=torch.float64, device="cuda")
        row = {"cfg": cfg, "N": info["N"], "rows": W}
        ref = None
        for probe in (0, 1, 2, 0, 1, 2):
            eng.tune("fuse_probe", probe)
            for _ in range(20):
                eng.loglike(X, out=out, check=False)
            torch.cuda.synchronize()
            eng.profile(True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                eng.loglike(X, out=out, check=False)
            e1.record(); torch.cuda.synchronize()
            n, ms, _ = eng.profile_read()
            eng.profile(False)
            row.setdefault(f"probe{probe}_k_predict_us", []).append(round(ms / n * 1e3, 2))
            row.setdefault(f"probe{probe}_loglike_call_us", []).append(round(e0.elapsed_time(e1) / 50 * 1e3, 2))
            if ref is None:
                ref = out.clone()
            assert torch.equal(ref, out)                       # the probe changes no result
        eng.tune("fuse_probe", 0)
        print(json.dumps(row), flush=True)
        emu._engine.close()


if __name__ == "__main__":
    main()
