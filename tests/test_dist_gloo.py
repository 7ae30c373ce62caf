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


# ---------------------------------------------------------------- more ranks, and the direct-collective hand-shake
class _FakeDirectEngine:
    """Stands in for GPEngine's gpb_dist_* binding on the CPU tier: the "direct" all-gather is carried by gloo, and
    one rank can be told to fail at a chosen stage (uid / init / allgather / wrong data)."""

    def __init__(self, rank, fail_rank=-1, fail_at=None):
        self.rank, self.fail_rank, self.fail_at = rank, fail_rank, fail_at
        self.calls = []

    def _maybe(self, stage):
        self.calls.append(stage)
        if self.rank == self.fail_rank and self.fail_at == stage:
            raise RuntimeError("injected failure at %s" % stage)

    def dist_uid(self):
        self._maybe("uid")
        return b"u" * 128

    def dist_available(self):
        self.calls.append("available")
        return not (self.rank == self.fail_rank and self.fail_at == "available")      # librccl missing on one rank

    def dist_init(self, rank, world, uid):
        assert uid == b"u" * 128
        self._maybe("init")

    def dist_allgather(self, send, recv):
        dist.all_gather_into_tensor(recv, send.clone())        # every rank takes part, then one may report a failure
        self._maybe("allgather")
        if self.rank == self.fail_rank and self.fail_at == "wrong":
            recv[0] += 1.0
        return recv

    def dist_finalize(self):
        self.calls.append("finalize")


def _many_worker(rank, world, port, q):
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from gpbayestools_hic_amd.dist import WalkerSharding, init_from_env
    init_from_env(backend="gloo")
    g, fn = _oracle_logpost()
    res = {}
    # (scenario, failing rank, stage): every rank must come out with the SAME decision and the same numbers
    for name, fail_rank, stage in (("ok", -1, None), ("uid", 0, "uid"), ("available", world - 2, "available"),
                                   ("init", world - 1, "init"), ("allgather", 1, "allgather"),
                                   ("wrong", world // 2, "wrong")):
        sh = WalkerSharding()
        eng = _FakeDirectEngine(rank, fail_rank, stage)
        why = sh.try_direct(eng)
        outs = {}
        for W in (64, 37, 3):                   # even split (in-place path), ragged, fewer rows than ranks
            X = torch.from_numpy(np.ascontiguousarray(g["Xw"][:W]))
            out = torch.empty(W, dtype=torch.float64)
            sh.logprob(fn, X, out)
            outs[W] = out.numpy().copy()
        res[name] = (why, sh.direct is not None, outs, list(eng.calls))
    dist.barrier()
    q.put((rank, res))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_walker_sharding_more_ranks_and_direct_handshake(world):
    """world 4 and 8 over gloo with the CPU oracle as the evaluator: the sharded vector equals the unsharded one on
    every rank, and WalkerSharding.try_direct keeps the C-ABI collective only when it worked on ALL ranks — a rank
    that fails at any stage (no uid, no librccl, communicator refused, collective raises, collective returns wrong data) makes
    every rank fall back to torch.distributed, after the same number of collectives on each."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_many_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    got = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    g, fn = _oracle_logpost()
    refs = {}
    for W in (64, 37, 3):
        X = torch.from_numpy(np.ascontiguousarray(g["Xw"][:W]))
        refs[W] = fn(X, torch.empty(W, dtype=torch.float64)).numpy()
    for name in ("ok", "uid", "available", "init", "allgather", "wrong"):
        kept = {got[r][name][1] for r in range(world)}
        assert kept == ({True} if name == "ok" else {False}), (name, kept)       # unanimous
        for r in range(world):
            why, direct, outs, calls = got[r][name]
            assert (why is None) == (name == "ok"), (name, r, why)
            for W in (64, 37, 3):
                # every rank holds the SAME vector, bit for bit; against the unsharded evaluation the CPU oracle's BLAS
                # may round a 16-row shard differently from a 64-row batch (the HIP path does not: tested on the GPU)
                assert np.array_equal(outs[W], got[0][name][2][W], equal_nan=True), (name, r, W)
                fin = np.isfinite(refs[W])
                assert np.array_equal(np.isfinite(outs[W]), fin)
                assert np.allclose(outs[W][fin], refs[W][fin], rtol=1e-12, atol=0.0), (name, r, W)
            # a communicator that was built is torn down again wherever the hand-shake is called off
            built = name in ("allgather", "wrong") or (name == "init" and r != world - 1)
            assert ("finalize" in calls) == built, (name, r, calls)
            # ncclCommInitRank is itself collective: when a rank has no id or no librccl the ranks vote BEFORE it and
            # NO rank enters it (one alone inside it would block for good)
            if name in ("uid", "available"):
                assert "init" not in calls, (name, r, calls)
            else:
                assert "init" in calls, (name, r, calls)
    # in the good case the direct path really carried the even-split batches
    assert all("allgather" in got[r]["ok"][3] for r in range(world))


# ---------------------------------------------------------------- replicated state: replicate + agree_state
class _FakeEmu:
    """the part of Emulator that WalkerSharding.replicate touches: host state through __getstate__ / __setstate__, a device
    index of its own, an engine that is dropped when the state is replaced"""

    def __init__(self, device, z):
        self.device, self.z, self._engine, self.closed = device, np.asarray(z, float), self, 0

    def close(self):
        self.closed += 1

    def __getstate__(self):
        return {"device": self.device, "z": self.z, "_engine": None, "closed": 0}

    def __setstate__(self, st):
        self.__dict__.update(st)

    def state_digest(self):
        import hashlib
        return hashlib.sha256(self.z.tobytes()).digest()


def _replica_worker(rank, world, port, q):
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import types
    from gpbayestools_hic_amd.dist import WalkerSharding, init_from_env
    from gpbayestools_hic_amd.mcmc import Chain
    init_from_env(backend="gloo")
    sh = WalkerSharding()
    # every rank "trained" its own replica: rank r's targets carry a rounding difference of its own (the SVD under another
    # BLAS threading), and rank 2's experiment block differs as well
    chain = types.SimpleNamespace(emuList=[_FakeEmu(rank, [1.0, 2.0 + 1e-16 * 4 * rank]), _FakeEmu(rank, [3.0])],
                                  expdata=np.array([[0.5 + (rank == 2)]]), expdata_cov=np.eye(1), min=np.zeros(2),
                                  max=np.ones(2), prior_volume_=1.0, _like_sig="stale")
    chain.state_digest = lambda: Chain.state_digest(chain)
    res = {"same": sh.agree_state(b"x" * 32)}
    try:
        sh.agree_state(chain.state_digest())
        res["differ"] = "no error"
    except RuntimeError as e:                   # on EVERY rank, also on those whose digest equals rank 0's
        res["differ"] = str(e)
    # the rows of a sharded batch: one all-reduce of a position-weighted checksum, every rank raises when they differ
    Xsame = torch.arange(60, dtype=torch.float64).reshape(12, 5) * 0.37
    sh.rows_agree_end(sh.rows_agree_begin(Xsame))
    res["rows"] = []
    for Xr in (Xsame + (1e-12 if rank == 1 else 0.0), Xsame.flip(0) if rank == 2 else Xsame):     # a changed bit; swapped rows
        try:
            sh.rows_agree_end(sh.rows_agree_begin(Xr.contiguous()))
            res["rows"].append("no error")
        except RuntimeError as e:
            res["rows"].append(str(e))
    # ... and with the state digest riding on the same all-reduce (Chain._log_prob: every sharded call, on every rank alike)
    res["rows_digest"] = []
    for dg in (b"d" * 32, bytes([rank == 1]) * 32, chain.state_digest()):
        try:
            sh.rows_agree_end(sh.rows_agree_begin(Xsame, dg))
            res["rows_digest"].append("no error")
        except RuntimeError as e:
            res["rows_digest"].append(str(e))
    sh.replicate(chain)
    res["after"] = sh.agree_state(chain.state_digest())
    sh.rows_agree_end(sh.rows_agree_begin(Xsame, chain.state_digest()))
    res["z"] = chain.emuList[0].z.copy()
    res["dev"] = [e.device for e in chain.emuList]
    res["exp"] = float(chain.expdata[0, 0])
    res["closed"] = "closed" in chain.emuList[0].__dict__ and rank != 0
    res["like_sig"] = chain._like_sig
    dist.barrier()
    q.put((rank, res))
    dist.destroy_process_group()


def test_replicas_are_rank_zeros_and_a_difference_raises_on_every_rank():
    world, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_replica_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        assert got[r]["same"] is True
        assert "replicas of the GP state differ" in got[r]["differ"], (r, got[r]["differ"])
        assert got[r]["after"] is True
        assert len(got[r]["rows"]) == 2 and all("different rows" in m for m in got[r]["rows"]), got[r]["rows"]
        rd = got[r]["rows_digest"]
        assert rd[0] == "no error" and all("replicas of the GP state differ" in m for m in rd[1:]), rd
        assert np.array_equal(got[r]["z"], got[0]["z"]) and got[r]["exp"] == 0.5          # rank 0's state everywhere ...
        assert got[r]["dev"] == [r, r]                                                    # ... on each rank's own device
        assert got[r]["like_sig"] == ("stale" if r == 0 else None)                        # likelihood blocks are re-installed
