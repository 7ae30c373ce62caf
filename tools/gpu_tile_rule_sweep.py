#!/usr/bin/env python3
"""Does the tile-shape rule of the predict launch pick the fastest shape?  C-driven step loop (compacted batches) of cfg 3 / 4 / 5's
emulators at several batch sizes, burnt-in (every row live) and from the uniform start (~half the rows live), with the shape
forced to 64x32 / 64x64 / 64x128 / 128x128 against the rule's own choice (0).
    python tools/gpu_tile_rule_sweep.py [cfg ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(sampler, steps):
    import torch
    sampler.run(None, 4, store=False, status=10 ** 9)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sampler.run(None, steps, store=False, status=10 ** 9)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd.sampler import StretchSampler
    from gpbayestools_hic_amd.workload import build_chain
    cfgs = [int(a) for a in sys.argv[1:]] or [3, 4, 5]
    rows_of = {3: (128, 256, 512, 1024, 2048), 4: (128, 256, 512, 1024, 2048), 5: (256, 512, 1024, 2048, 4096)}
    names = {0: "rule", 32: "64x32", 64: "64x64", 65: "64x128", 128: "128x128"}
    for cfg in cfgs:
        chain, emu, info = build_chain(cfg)
        eng = emu._engine_ready()
        for rows in rows_of[cfg]:
            nw = 2 * rows
            for start in ("ball", "uniform"):
                X0 = synth.walkers_ball(nw, info["xstar"], 1e-13) if start == "ball" else synth.walkers(nw, info["d"])
                out = {"config": cfg, "N": info["N"], "rows_per_batch": rows, "start": start}
                steps = max(6, min(40, int(60 / (1e-6 * info["N"] ** 2 * rows * info["P"] / 6e4 + 0.05))))
                for tile in (0, 32, 64, 65, 128):
                    eng.force_tile(tile)
                    s = StretchSampler(chain, nw, seed=1)
                    s.run(X0, 3, store=False, status=10 ** 9)
                    out[names[tile]] = round(timed(s, steps), 4)
                eng.force_tile(0)
                best = min((v, k) for k, v in out.items() if k in names.values() and k != "rule")
                out["best"] = best[1]
                out["rule_over_best"] = round(out["rule"] / best[0], 3)
                print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
