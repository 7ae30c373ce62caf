"""GPU tier: the replicas of a walker-sharded run.  Two ranks share cuda:0 and rendezvous over gloo (one GPU per box).  Rank 1
"trains" on observables that differ from rank 0's in the last bits (what numpy's scaler / PCA SVD does under another BLAS
threading): StretchSampler.run refuses to start — on BOTH ranks — until WalkerSharding.replicate has handed rank 0's fitted
state to rank 1, and then both hold the single-process ensemble bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import REPO

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _build(workdir, perturb):
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd.workload import build_chain
    chain, emu, info = build_chain(1, workdir=workdir)
    if perturb:                                        # a replica whose targets differ in the last bits
        emu.model_data = emu.model_data * (1.0 + 4e-16)
        emu.trainEmulator([True] * emu.nev, kernel_type=info["kernel_type"], thetas=synth.fixed_theta(info["d"], info["P"]))
    return chain, emu, info


def _run(chain, info, sharding):
    from gpbayestools_hic_amd import StretchSampler, synth
    s = StretchSampler(chain, 64, seed=21, sharding=sharding)
    s.run(synth.walkers(64, info["d"], seed=4), 5)
    return s.chain, s.lnprobability


def _worker(rank, world, port, workdir, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from gpbayestools_hic_amd.dist import WalkerSharding, init_from_env
    init_from_env(backend="gloo")
    d = os.path.join(workdir, f"r{rank}"); os.makedirs(d, exist_ok=True)
    chain, emu, info = _build(d, perturb=(rank == 1))
    sh = WalkerSharding()
    digest0 = chain.state_digest()
    try:
        _run(chain, info, sh)
        refused = "ran"
    except RuntimeError as e:
        refused = str(e)
    sh.replicate(chain)
    out = _run(chain, info, sh)
    # the pocoMC call shape, sharded: log_likelihood(X, finite=True) batches in row shares over the ranks (Chain.shard_over)
    import numpy as np
    from gpbayestools_hic_amd import synth
    Xb = synth.walkers(301, info["d"], seed=8)
    Xb[::7, 1] = 1.25                                               # rows outside the box
    chain.shard_over(sh)
    ll = (chain.log_likelihood(Xb, finite=True), chain.log_posterior(Xb[:64]), chain.log_likelihood(Xb[:1], finite=True))
    try:
        chain.log_likelihood(Xb + (1e-9 if rank == 1 else 0.0), finite=True)
        differ = "ran"
    except RuntimeError as e:
        differ = str(e)
    chain.shard_over(None)
    dist.barrier()
    q.put((rank, refused, digest0, chain.state_digest(), out, ll, differ))
    dist.destroy_process_group()


def test_ranks_refuse_differing_replicas_and_agree_after_replicate(tmp_path):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs: p.start()
    got = {r[0]: r[1:] for r in (q.get(timeout=600) for _ in range(world))}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    one = tmp_path / "single"; one.mkdir()
    chain, emu, info = _build(str(one), perturb=False)
    ref = _run(chain, info, None)
    assert got[0][1] != got[1][1]                                   # the trainings really differed ...
    for r in range(world):
        assert "replicas of the GP state differ" in got[r][0], got[r][0]     # ... and BOTH ranks refused to sample
        assert got[r][2] == chain.state_digest()                   # after replicate: rank 0's state = the single process's
        assert np.array_equal(got[r][3][0], ref[0]) and np.array_equal(got[r][3][1], ref[1]), r
    from gpbayestools_hic_amd import synth
    Xb = synth.walkers(301, info["d"], seed=8)
    Xb[::7, 1] = 1.25
    want = (chain.log_likelihood(Xb, finite=True), chain.log_posterior(Xb[:64]), chain.log_likelihood(Xb[:1], finite=True))
    assert np.all(want[0][::7] == -1e300) and np.all(np.isfinite(want[0]))
    for r in range(world):
        for a, b in zip(got[r][4], want):
            assert np.array_equal(a, b), r                          # sharded batches = the single process's, bit for bit
        assert "different rows" in got[r][5], got[r][5]             # and both ranks refuse a batch that differs between them


def _flow_worker(rank, world, port, workdir, q):
    import faulthandler
    import traceback
    faulthandler.dump_traceback_later(200, exit=True)               # a rank that waits for a peer that died says where
    try:
        _flow(rank, world, port, workdir, q)
    except BaseException:                                           # noqa: BLE001  (the parent fails the test with this text)
        q.put((rank, "ERROR", traceback.format_exc(), None, None))
        raise


def _flow(rank, world, port, workdir, q):
    """the flow INTEGRATION.md documents for the pocoMC call shape: train, WalkerSharding.replicate(chain), chain.shard_over(sh),
    then log_likelihood batches — with NO sampler run in between, so that after replicate rank 0 still holds its engines while
    rank 1's were rebuilt: what one rank knows locally must not decide which collectives it enters"""
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import numpy as np
    import torch.distributed as dist
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd.dist import GPSharding, WalkerSharding, init_from_env
    init_from_env(backend="gloo")
    d = os.path.join(workdir, f"r{rank}"); os.makedirs(d, exist_ok=True)
    chain, emu, info = _build(d, perturb=(rank == 1))
    emu.fit_sharding = mine = GPSharding()
    sh = WalkerSharding()
    Xb = synth.walkers(129, info["d"], seed=8)
    chain.shard_over(sh)
    try:                                                            # differing replicas: every rank refuses the batch
        chain.log_likelihood(Xb, finite=True)
        refused = "ran"
    except RuntimeError as e:
        refused = str(e)
    sh.replicate(chain)
    rebuilt = emu._engine is not None                               # replicate rebuilds the device state before it returns
    kept = emu.fit_sharding is mine                                 # ... and leaves each rank its own fit-side sharding
    ll = [chain.log_likelihood(Xb, finite=True), chain.log_likelihood(Xb[:50], finite=True), chain.log_posterior(Xb)]
    emu._engine.close(); emu._engine = None                         # one rank's engines dropped (a pickle round trip): still
    if rank == 0:                                                   # the same collectives on both ranks
        emu._engine_ready()
    ll.append(chain.log_likelihood(Xb, finite=True))
    dist.barrier()
    q.put((rank, refused, rebuilt, kept, ll))
    dist.destroy_process_group()


def test_replicate_then_shard_over_then_batches_without_a_sampler_run(tmp_path):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_flow_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs: p.start()
    got = {}
    try:
        for _ in range(world):
            r = q.get(timeout=300)
            assert r[1] != "ERROR", "rank %d:\n%s" % (r[0], r[2])
            got[r[0]] = r[1:]
    finally:
        for p in procs:
            p.join(timeout=5 if len(got) < world else 120)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    one = tmp_path / "single"; one.mkdir()
    chain, emu, info = _build(str(one), perturb=False)
    from gpbayestools_hic_amd import synth
    Xb = synth.walkers(129, info["d"], seed=8)
    want = [chain.log_likelihood(Xb, finite=True), chain.log_likelihood(Xb[:50], finite=True), chain.log_posterior(Xb),
            chain.log_likelihood(Xb, finite=True)]
    for r in range(world):
        refused, rebuilt, kept, ll = got[r]
        assert "replicas of the GP state differ" in refused, refused
        assert rebuilt and kept
        for a, b in zip(ll, want):
            assert np.array_equal(a, b), r


def _mcmc_calls(chain):
    chain.run_mcmc(nsteps=12, nburnsteps=8, nwalkers=32, nthin=3, seed=7, status=100)       # burn-in, re-seeding, production
    first = chain.chain.copy()
    chain.run_mcmc(nsteps=6, nburnsteps=8, nwalkers=32, nthin=3, seed=8, status=100)        # resumes from the chain file
    return first, chain.chain.copy(), np.asarray(chain.acceptance_fraction).copy()


def _mcmc_worker(rank, world, port, workdir, q):
    import faulthandler
    import traceback
    faulthandler.dump_traceback_later(200, exit=True)
    try:
        sys.path.insert(0, REPO)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
        import torch.distributed as dist
        from gpbayestools_hic_amd.dist import WalkerSharding, init_from_env
        init_from_env(backend="gloo")
        d = os.path.join(workdir, f"r{rank}"); os.makedirs(d, exist_ok=True)
        chain, emu, info = _build(d, perturb=False)
        sh = WalkerSharding()
        sh.replicate(chain)
        chain.shard_over(sh)
        np.random.seed(5 if rank == 0 else 99)        # the start positions are rank 0's draws, whatever the others would draw
        out = _mcmc_calls(chain)
        wrote = os.path.exists(chain.mcmc_path)
        dist.barrier()
        # rank 0's chain file unreadable: EVERY rank raises (none is left waiting for rank 0 in a broadcast)
        if rank == 0:
            with open(chain.mcmc_path, "wb") as f:
                f.write(b"not a pickle")
        dist.barrier()
        try:
            chain.run_mcmc(nsteps=3, nburnsteps=4, nwalkers=32, seed=9, status=100)
            refused = None
        except RuntimeError as e:
            refused = str(e)
        dist.barrier()
        q.put((rank, "ok", out, (wrote, refused)))
        dist.destroy_process_group()
    except BaseException:                             # noqa: BLE001
        q.put((rank, "ERROR", traceback.format_exc(), None))
        raise


def test_run_mcmc_walker_sharded_over_two_ranks_is_the_single_gpu_chain(tmp_path):
    """Chain.run_mcmc after shard_over (src/mcmc.py:345-426 on several GPUs): rank 0's start, seed and chain file are everybody's,
    every half-step is evaluated in shares; both ranks end with the single-process chain — burn-in with re-seeding, production,
    thinning, resume — bit for bit, and only rank 0 writes the pickle."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mcmc_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs: p.start()
    got = {}
    try:
        for _ in range(world):
            r = q.get(timeout=300)
            assert r[1] != "ERROR", "rank %d:\n%s" % (r[0], r[2])
            got[r[0]] = r[2:]
    finally:
        for p in procs:
            p.join(timeout=5 if len(got) < world else 120)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    one = tmp_path / "single"; one.mkdir()
    chain, emu, info = _build(str(one), perturb=False)
    np.random.seed(5)
    ref = _mcmc_calls(chain)
    assert ref[0].shape == (32, 4, info["d"]) and ref[1].shape == (32, 6, info["d"])
    for r in range(world):
        out, (wrote, refused) = got[r]
        for a, b in zip(out, ref):
            assert np.array_equal(a, b), r
        assert wrote == (r == 0)
        assert refused is not None and "could not read" in refused, (r, refused)
