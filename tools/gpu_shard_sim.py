#!/usr/bin/env python3
"""Single-GPU estimate of the per-rank step time under R-way walker sharding (no collective):
rank 0's shard is evaluated for real, the other rows are left untouched.  Timing only."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class FakeShard:
    def __init__(self, world):
        self.world = world

    def logprob(self, fn, X, out):
        W = X.shape[0]
        chunk = -(-W // self.world)
        fn(X[:chunk], out[:chunk])
        out[chunk:].fill_(-1e9)           # never accepted; stands in for the gathered remote rows
        return out


def main():
    import torch
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd.sampler import StretchSampler
    from gpbayestools_hic_amd.workload import build_chain
    chain, emu, info = build_chain(4)
    nw = 2 * info["W"]
    worlds = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]
    for world in worlds:
        s = StretchSampler(chain, nw, seed=1, sharding=FakeShard(world) if world > 1 else None)
        s.run(synth.walkers(nw, info["d"]), 3, store=False, status=10 ** 9)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.run(None, 20, store=False, status=10 ** 9)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        # host-only enqueue cost: same loop, timed without the final sync
        t0 = time.perf_counter()
        for _ in range(1):
            s.run(None, 20, store=False, status=10 ** 9)
        print(json.dumps({"world": world, "ms_per_step": round(dt * 1e3, 3),
                          "speedup_vs_1": None}), flush=True)


if __name__ == "__main__":
    main()
