#!/usr/bin/env python3
"""Single-GPU measurement of ONE RANK's share of a walker-sharded step (BASELINE config 4, 4096 walkers):
gpb_emcee_run with the `sim_ranks` hook evaluates the first 1/R of every half-ensemble batch — exactly what rank 0 of R
does — and, with a one-rank RCCL communicator installed, still enqueues the two in-stream all-gathers per step
(ncclAllGather on the kernels' stream: launch and protocol cost without a wire).  Also: the same share through the
host-driven loop (Python enqueues every kernel), to show what the C loop removes.

    python tools/gpu_shard_sim.py [R ...] [--c-only] [--cfg=5] [--ball=1e-3]        # default 1 2 4 8, uniform start
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPB_DEBUG_LIB", "1")      # measurement hooks and kernel variants: the debug library (libgpbayes_debug.so)


class FakeShard:
    """host-driven loop, rank 0's rows only (round 1's simulation)"""

    def __init__(self, world):
        self.world = world

    def logprob(self, fn, X, out):
        chunk = -(-X.shape[0] // self.world)
        fn(X[:chunk], out[:chunk])
        out[chunk:].fill_(-1e9)
        return out


def timed(sampler, steps=40):
    import torch
    sampler.run(None, 5, store=False, status=10 ** 9)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sampler.run(None, steps, store=False, status=10 ** 9)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    import torch
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd.sampler import StretchSampler
    from gpbayestools_hic_amd.workload import build_chain
    cfg = 4
    for a in sys.argv[1:]:
        if a.startswith("--cfg="):
            cfg = int(a[6:])
    chain, emu, info = build_chain(cfg)
    for a in sys.argv[1:]:
        if a.startswith("--tune="):                  # e.g. --tune=tile_by_live:0
            k, v = a[7:].split(":")
            if k == "force_tile":                    # 0 = rule, 32 = 64x32, 64 = 64x64, 65 = 64x128, 128 = 128x128
                emu._engine_ready().force_tile(int(v))
            else:
                emu._engine_ready().tune(k, int(v))
            print(json.dumps({"tune": {k: int(v)}}), flush=True)
    eng = emu._engine_ready()
    nw = 2 * info["W"]
    for a in sys.argv[1:]:
        if a.startswith("--walkers="):
            nw = int(a[10:])
    X0 = synth.walkers(nw, info["d"])
    for a in sys.argv[1:]:
        if a.startswith("--ball="):                  # burnt-in start: walkers in a ball of this relative radius around theta*
            X0 = synth.walkers_ball(nw, info["xstar"], float(a[7:]))
            print(json.dumps({"start": "ball", "radius": float(a[7:])}), flush=True)
    c_only = "--c-only" in sys.argv
    worlds = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [1, 2, 4, 8]
    comm = False
    try:
        eng.dist_init(0, 1, eng.dist_uid())          # one-rank communicator: the collective is enqueued for real
        comm = True
    except Exception as e:                           # librccl missing: time without the collective and say so
        print(json.dumps({"warning": "no RCCL communicator: %s" % e}), flush=True)
    for world in worlds:
        row = {"config": cfg, "walkers": nw, "ranks_simulated": world, "walkers_per_rank_per_batch": nw // 2 // world, "collective_enqueued": comm and world > 1}
        eng.tune("sim_ranks", world if world > 1 else 0)
        # pre-heat on a scratch ensemble (clocks up whatever the step length: 8 steps of a 0.4 ms step are not enough; the
        # measured ensemble itself must stay short of ~60 steps from a 1e-13 ball, or it has spread to the box)
        heat = StretchSampler(chain, nw, seed=99)
        heat.run(X0, 100, store=False, status=10 ** 9)
        del heat
        s = StretchSampler(chain, nw, seed=1)
        assert s._resident_engine()[0] is eng
        s.run(X0, 3, store=False, status=10 ** 9)
        row["c_loop_ms_per_step"] = round(timed(s), 4)
        eng.profile(True)                                # a few more steps with the live-row counter and event timing on
        s.run(None, 5, store=False, status=10 ** 9)
        n_l, ms_l, units = eng.profile_read()
        eng.profile(False)
        P = info["P"]
        row["rows_inside_box_fraction"] = round(units / (n_l * P * (nw // 2 // world)), 3)
        row["k_predict_us_per_launch"] = round(ms_l / n_l * 1e3, 1)
        eng.tune("sim_ranks", 0)
        if c_only:
            print(json.dumps(row), flush=True)
            continue
        s2 = StretchSampler(chain, nw, seed=1, sharding=FakeShard(world) if world > 1 else None)
        s2._resident_engine = lambda: None            # force the host-driven loop
        s2.run(X0, 3, store=False, status=10 ** 9)
        row["python_loop_ms_per_step_no_collective"] = round(timed(s2), 4)
        print(json.dumps(row), flush=True)
    base = None
    if comm:
        eng.dist_finalize()


if __name__ == "__main__":
    main()
