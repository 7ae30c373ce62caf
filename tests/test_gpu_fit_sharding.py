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
