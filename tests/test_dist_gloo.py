"""
CPU tier, world_size 2 over gloo: the walker-sharding layer (dist.WalkerSharding) that the N>1 bench
and sampler use.  The per-row evaluator is the CPU oracle here (test-only); on the GPU box the same
layer wraps the HIP log-posterior and the collective is RCCL.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO, golden


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _oracle_logpost():
    from oracle import gp_oracle as O
    g = golden("g5_chain.npz")
    emus = []
    for tag in ("A", "B"):
        e = O.OracleEmulator(g["X"], g[f"Y_{tag}"], g["lo"], g["hi"], int(g[f"npc_{tag}"]))
        emus.append(e.fit(g[f"thetas_{tag}"]))
    pf = lambda X, es: O.chain_predict(emus, X, es)

    def fn(X_t, out_t):
        lp = O.log_prob(X_t.numpy(), g["lo"], g["hi"], pf, g["expdata"], g["expdata_cov"])
        out_t.copy_(torch.from_numpy(lp))
        return out_t
    return g, fn


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from gpbayestools_hic_amd.dist import WalkerSharding, init_from_env
    r, w, _ = init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    sh = WalkerSharding()
    g, fn = _oracle_logpost()
    res = {}
    for W in (64, 37, 1):                       # even, ragged, fewer rows than ranks
        X = torch.from_numpy(np.ascontiguousarray(g["Xw"][:W]))
        out = torch.empty(W, dtype=torch.float64)
        sh.logprob(fn, X, out)
        res[W] = out.numpy().copy()
        assert sh.rows(W)[2] * world >= W
    dist.barrier()
    q.put((rank, res))
    dist.destroy_process_group()


def test_walker_sharding_world2_matches_unsharded():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g, fn = _oracle_logpost()
    for W in (64, 37, 1):
        X = torch.from_numpy(np.ascontiguousarray(g["Xw"][:W]))
        ref = fn(X, torch.empty(W, dtype=torch.float64)).numpy()
        for rank, res in got:
            assert np.array_equal(res[W], ref, equal_nan=True), (rank, W)      # every rank holds the full, identical vector
        assert np.allclose(ref[g["inside"][:W]], g["log_posterior"][:W][g["inside"][:W]], rtol=1e-10)


def test_rows_partition():
    from gpbayestools_hic_amd.dist import WalkerSharding
    for world in (1, 2, 3, 8):
        for W in (0, 1, 5, 8, 2048, 2049):
            seen = []
            for r in range(world):
                sh = WalkerSharding.__new__(WalkerSharding); sh.rank, sh.world = r, world
                r0, r1, chunk = sh.rows(W)
                assert 0 <= r0 <= r1 <= W and r1 - r0 <= chunk
                seen += list(range(r0, r1))
            assert seen == list(range(W))
