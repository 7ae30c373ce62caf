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


# ---------------------------------------------------------------- fit-side sharding (dist.GPSharding)
class _OracleLMLEngine:
    """Stands in for GPEngine.lml on the CPU tier: batched LML and gradient from the oracle."""

    def __init__(self, X, Z):
        self.X, self.Z = X, Z

    def lml(self, theta, eval_gradient=True):
        from oracle import gp_oracle as O
        out = [O.lml(theta[p], self.X, self.Z[p], O.KIND_RBF, 0.1, eval_gradient=True) for p in range(len(self.Z))]
        return np.array([o[0] for o in out]), np.array([o[1] for o in out])

    def close(self):
        pass


def _fit_problem(P):
    from oracle import gp_oracle as O
    rng = np.random.default_rng(11)
    N, d = 40, 3
    X = rng.random((N, d))
    Z = np.array([np.sin(X @ rng.standard_normal(d)) + 0.05 * rng.standard_normal(N) for _ in range(P)])
    theta0, bounds = O.default_theta0_bounds(np.zeros(d), np.ones(d), O.KIND_RBF)
    return X, Z, np.asarray(theta0), np.asarray(bounds)


def _fit_worker(rank, world, port, q):
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from gpbayestools_hic_amd.dist import GPSharding, init_from_env
    from gpbayestools_hic_amd.emulator import search_hyperparameters
    init_from_env(backend="gloo")
    res = {}
    for P in (3, 1):                            # 3 GPs over 2 ranks; 1 GP: rank 1 has nothing to search
        X, Z, theta0, bounds = _fit_problem(P)
        sh = GPSharding()
        served = []
        th, val = search_hyperparameters(lambda idx: (served.append(list(idx)), _OracleLMLEngine(X, Z[idx]))[1],
                                         P, theta0, bounds, 0, sh, close=True)
        res[P] = (th, val, served)
    dist.barrier()
    q.put((rank, res))
    dist.destroy_process_group()


def test_fit_sharding_world2_matches_unsharded():
    from gpbayestools_hic_amd.emulator import search_hyperparameters
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fit_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for P in (3, 1):
        X, Z, theta0, bounds = _fit_problem(P)
        th, val = search_hyperparameters(lambda idx: _OracleLMLEngine(X, Z[idx]), P, theta0, bounds, 0)
        assert np.all(np.isfinite(val)) and not np.allclose(th, theta0)          # the search moved
        for rank in range(world):
            assert np.array_equal(got[rank][P][0], th) and np.array_equal(got[rank][P][1], val), (rank, P)
        assert got[0][P][2] == [list(range(0, P, 2))]
        assert got[1][P][2] == ([[1]] if P == 3 else [])
