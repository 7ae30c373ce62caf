"""GPU tier: fit-side sharding (dist.GPSharding).  Two ranks share cuda:0 and rendezvous over gloo
(one GPU per box; the exchange is a host-side object all-gather, so the backend does not matter): each rank
searches its GPs on a sub-engine, and both must end with exactly the hyper-parameters, likelihoods and
predictions of the unsharded training."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import REPO

pytestmark = pytest.mark.gpu

N, D, M, NPC = 96, 4, 6, 5


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _train(workdir, sharded):
    from gpbayestools_hic_amd import Emulator, synth
    X = synth.lhs(N, D, seed=5)
    Y = synth.observables(X, M, seed=6)
    tp, pf = os.path.join(workdir, "train.pkl"), os.path.join(workdir, "par.txt")
    synth.write_training_pickle(tp, X, Y, 0.01)
    synth.write_parameter_file(pf, np.zeros(D), np.ones(D))
    emu = Emulator(training_set_path=tp, parameter_file=pf, npc=NPC)
    if sharded:
        from gpbayestools_hic_amd.dist import GPSharding
        emu.fit_sharding = GPSharding()
    emu.trainEmulatorAutoMask()
    mean, cov = emu.predict(synth.walkers(17, D, seed=9), return_cov=True)
    return emu.thetas_.copy(), np.asarray(emu.lml_).copy(), mean, cov


def _worker(rank, world, port, workdir, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch.distributed as dist
    from gpbayestools_hic_amd.dist import init_from_env
    init_from_env(backend="gloo")
    d = os.path.join(workdir, f"r{rank}"); os.makedirs(d, exist_ok=True)
    out = _train(d, sharded=True)
    dist.barrier()
    q.put((rank, out))
    dist.destroy_process_group()


def test_two_ranks_search_disjoint_gps_and_agree_with_one_rank(tmp_path):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs: p.start()
    got = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    one = tmp_path / "single"; one.mkdir()
    th, lml, mean, cov = _train(str(one), sharded=False)
    assert th.shape == (NPC, D + 2) and np.all(np.isfinite(lml))
    for rank in range(world):
        rth, rlml, rmean, rcov = got[rank]
        # a GP's search is independent of the batch it is evaluated in (fixed-order reductions per GP)
        assert np.array_equal(rth, th), rank
        assert np.array_equal(rlml, lml), rank
        assert np.array_equal(rmean, mean) and np.array_equal(rcov, cov), rank


def _train_two_with_restarts(workdir, sharded):
    """two emulators trained TOGETHER (train_emulators) with restarts: the first one's GPs dealt to the ranks (its searches,
    restarts included, in this rank's lock-step batch), the second unsharded on every rank.  np.random.seed fixes the restart
    points (sklearn draws them from numpy's global RandomState, sk:_gpr.py:259,318-325)."""
    from gpbayestools_hic_amd import Emulator, synth
    from gpbayestools_hic_amd.emulator import train_emulators
    emus = []
    for i, (npc, nre) in enumerate(((NPC, 2), (3, 1))):
        X = synth.lhs(N, D, seed=5 + i)
        tp, pf = os.path.join(workdir, "train%d.pkl" % i), os.path.join(workdir, "par.txt")
        synth.write_training_pickle(tp, X, synth.observables(X, M, seed=6 + i), 0.01)
        synth.write_parameter_file(pf, np.zeros(D), np.ones(D))
        emus.append(Emulator(training_set_path=tp, parameter_file=pf, npc=npc, nrestarts=nre))
    if sharded:
        from gpbayestools_hic_amd.dist import GPSharding
        emus[0].fit_sharding = GPSharding()
    np.random.seed(1234)
    train_emulators(emus)
    Xq = synth.walkers(9, D, seed=9)
    return [(e.thetas_.copy(), np.asarray(e.lml_).copy(), e.predict(Xq, return_cov=False)) for e in emus]


def _restart_worker(rank, world, port, workdir, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from gpbayestools_hic_amd.dist import init_from_env
    init_from_env(backend="gloo")
    d = os.path.join(workdir, f"r{rank}"); os.makedirs(d, exist_ok=True)
    out = _train_two_with_restarts(d, sharded=True)
    dist.barrier()
    q.put((rank, out))
    dist.destroy_process_group()


def test_train_emulators_with_a_sharded_fit_and_restarts(tmp_path):
    """round 4 raised NotImplementedError here.  The sharded emulator's searches equal the unsharded training's — and the
    sequential one's (Emulator.trainEmulator emulator after emulator, the same global RandomState) — bit for bit."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_restart_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs: p.start()
    got = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    one = tmp_path / "single"; one.mkdir()
    ref = _train_two_with_restarts(str(one), sharded=False)
    for rank in range(world):
        for (th, lml, mean), (rth, rlml, rmean) in zip(ref, got[rank]):
            assert np.array_equal(rth, th) and np.array_equal(rlml, lml) and np.array_equal(rmean, mean), rank
    # ... and what sequential trainings give from the same seed
    from gpbayestools_hic_amd import Emulator, synth
    seq = []
    np.random.seed(1234)
    for i, (npc, nre) in enumerate(((NPC, 2), (3, 1))):
        e = Emulator(training_set_path=str(one / ("train%d.pkl" % i)), parameter_file=str(one / "par.txt"), npc=npc, nrestarts=nre)
        e.trainEmulatorAutoMask()
        seq.append(e)
    for e, (th, lml, _) in zip(seq, ref):
        assert np.array_equal(e.thetas_, th) and np.array_equal(np.asarray(e.lml_), lml)


def test_c_abi_rccl_binding_single_rank():
    """gpb_dist_* (lazy dlopen of librccl, communicator per context, in-stream ncclAllGather) with a one-rank
    communicator: the only size a one-GPU box can form; more ranks run the same calls."""
    import torch
    from gpbayestools_hic_amd import GPEngine
    eng = GPEngine(0)
    uid = eng.dist_uid()
    assert len(uid) == 128 and any(uid)
    eng.dist_init(0, 1, uid)
    out = torch.zeros(1000, dtype=torch.float64, device="cuda")
    src = torch.arange(1000, dtype=torch.float64, device="cuda") * 0.5
    eng.dist_allgather(src, out)                       # separate buffers
    torch.cuda.synchronize()
    assert torch.equal(out, src)
    out.mul_(3.0)
    eng.dist_allgather(out[0:1000], out)               # in place (send = recv + rank * count)
    torch.cuda.synchronize()
    assert torch.equal(out, src * 3.0)
    with pytest.raises(ValueError):
        eng.dist_allgather(src[:10], out)              # recv must hold world * count
    eng.dist_finalize()
    eng.close()


def _direct_worker(port, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch
    import torch.distributed as dist
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd.dist import WalkerSharding
    from gpbayestools_hic_amd.sampler import StretchSampler
    from gpbayestools_hic_amd.workload import build_chain
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    chain, emu, info = build_chain(1)
    X0 = synth.walkers(64, info["d"])
    plain = StretchSampler(chain, 64, seed=7)
    plain.run(X0, 20, status=10 ** 9, store=False)
    sh = WalkerSharding()
    why = sh.try_direct(emu._engine_ready())
    sharded = StretchSampler(chain, 64, seed=7, sharding=sh)
    sharded.run(X0, 20, status=10 ** 9, store=False)
    # ragged batch (staging buffers) through the same communicator
    X = torch.as_tensor(synth.walkers(37, info["d"]), device="cuda")
    a = torch.empty(37, dtype=torch.float64, device="cuda"); b = torch.empty_like(a)
    fn = lambda Xr, o: chain.log_prob_device(Xr, out=o)
    fn(X, a); sh.logprob(fn, X, b)
    torch.cuda.synchronize()
    us_direct = sh.time_allgather(256, reps=50, warm=5)             # bench.py's wire probe, on both exchange paths
    keep, sh.direct = sh.direct, None
    us_torch = sh.time_allgather(256, reps=50, warm=5)
    sh.direct = keep
    q.put((why, sh.direct is not None, bool(torch.equal(plain.pos, sharded.pos)), bool(torch.equal(plain.lp, sharded.lp)),
           bool(torch.equal(a, b)), us_direct, us_torch))
    dist.destroy_process_group()


def test_sampler_over_the_direct_rccl_allgather_one_rank():
    """bench.py's N>1 exchange (WalkerSharding.try_direct -> gpb_dist_allgather in place on the kernel stream),
    formed with the one rank a one-GPU box has: self-check passes, the chain equals the unsharded one."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_direct_worker, args=(_free_port(), q))
    p.start()
    why, direct_on, same_pos, same_lp, ragged_ok, us_direct, us_torch = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert why is None and direct_on
    assert same_pos and same_lp and ragged_ok
    assert 0.0 < us_direct < 500.0 and 0.0 < us_torch < 500.0       # (one rank: launch and protocol cost without a wire)
