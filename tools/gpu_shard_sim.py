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
        # host-only enqueue cost: same loop against a stubbed log-probability (no GPU work to wait for)
        s2 = StretchSampler(chain, nw, seed=1, logprob_device=lambda X, out: out)
        s2.run(synth.walkers(nw, info["d"]), 2, store=False, status=10 ** 9)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s2.run(None, 50, store=False, status=10 ** 9)
        host_stub = (time.perf_counter() - t0) / 50
        # enqueue time of the real loop (returns before the GPU has finished)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(1):
            s._step_loop_only = True
        import types
        t0 = time.perf_counter()
        n_enq = 30
        eng = s._engine(); lib, h = eng.lib, eng.h
        from gpbayestools_hic_amd import _native as nat
        for step in range(n_enq):
            for half in (0, 1):
                eng._ck(lib.gpb_stretch_propose(h, nat.ptr(s.pos), nw, s.ndim, half, s.seed, 10000 + step, s.a, nat.ptr(s.q), nat.ptr(s.factor), 1))
                s._eval(s.q, s.lpq)
                eng._ck(lib.gpb_stretch_accept(h, nat.ptr(s.pos), nat.ptr(s.lp), nw, s.ndim, half, s.seed, 10000 + step, nat.ptr(s.q), nat.ptr(s.factor), nat.ptr(s.lpq), nat.ptr(s.naccept), 1))
        host_enq = (time.perf_counter() - t0) / n_enq
        torch.cuda.synchronize()
        print(json.dumps({"world": world, "ms_per_step": round(dt * 1e3, 3), "host_enqueue_ms_per_step": round(host_enq * 1e3, 3),
                          "host_stub_ms_per_step": round(host_stub * 1e3, 3)}), flush=True)


if __name__ == "__main__":
    main()
