"""
GPU tier: the device-resident stretch-move sampler.  emcee is not installed in the build
environment, so equivalence is statistical (moments of a known Gaussian target, acceptance
fraction), plus schema/resume semantics of Chain.run_mcmc (src/mcmc.py:345-426).
"""
import pickle
import types

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def test_stretch_move_samples_gaussian_target():
    import torch
    from gpbayestools_hic_amd import StretchSampler
    d, nw = 5, 256
    rng = np.random.default_rng(0)
    mu = rng.normal(size=d)
    B = rng.normal(size=(d, d))
    cov = B @ B.T / d + 0.3 * np.eye(d)
    prec = torch.as_tensor(np.linalg.inv(cov), device="cuda")
    mu_t = torch.as_tensor(mu, device="cuda")

    def logprob(X, out):
        r = X - mu_t
        out.copy_(-0.5 * torch.einsum("wi,ij,wj->w", r, prec, r))
        return out

    fake = types.SimpleNamespace(ndim=d, device=0, min=np.full(d, -50.0), max=np.full(d, 50.0), emuList=[])
    s = StretchSampler(fake, nw, seed=7, logprob_device=logprob)
    X0 = mu + 0.1 * rng.normal(size=(nw, d))
    s.run(X0, 300, store=False)
    s.reset()
    s.run(None, 1500)
    flat = s.flatchain
    assert flat.shape == (1500 * nw, d)
    assert s.chain.shape == (nw, 1500, d) and s.lnprobability.shape == (nw, 1500)
    sd = np.sqrt(np.diag(cov))
    assert np.all(np.abs(flat.mean(0) - mu) < 0.05 * sd)
    assert np.max(np.abs(np.cov(flat.T) - cov)) < 0.08 * np.max(np.abs(cov))
    af = s.acceptance_fraction
    assert 0.35 < af.mean() < 0.75          # emcee's stretch move (a=2) sits around 0.5-0.6 at d=5
    # log-probabilities stored with the chain belong to the stored positions
    r = flat - mu
    assert np.allclose(s.flatlnprobability, -0.5 * np.einsum("wi,ij,wj->w", r, np.linalg.inv(cov), r), rtol=1e-9, atol=1e-9)


def test_same_seed_reproduces_chain_and_shards_do_not_change_it():
    """Replicated counter-based RNG: identical seeds give identical chains, and evaluating the
    log-probability in shards (fake ranks on one GPU) leaves every number unchanged."""
    import torch
    from gpbayestools_hic_amd import StretchSampler
    d, nw = 4, 64
    fake = types.SimpleNamespace(ndim=d, device=0, min=np.full(d, -9.0), max=np.full(d, 9.0), emuList=[])

    def logprob(X, out):
        out.copy_(-0.5 * (X * X).sum(1) + torch.sin(X).sum(1))
        return out

    class FakeShards:                        # same interface as dist.WalkerSharding, serial over 3 "ranks"
        def logprob(self, fn, X, out):
            W = X.shape[0]; chunk = -(-W // 3)
            for r in range(3):
                a, b = min(r * chunk, W), min((r + 1) * chunk, W)
                if b > a: fn(X[a:b], out[a:b])
            return out

    X0 = np.random.default_rng(1).normal(size=(nw, d))
    runs = []
    for sh in (None, None, FakeShards()):
        s = StretchSampler(fake, nw, seed=123, logprob_device=logprob, sharding=sh)
        s.run(X0, 40)
        runs.append((s.chain, s.lnprobability))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][0], runs[2][0])
    assert np.array_equal(runs[0][1], runs[2][1])
    s2 = StretchSampler(fake, nw, seed=124, logprob_device=logprob)
    s2.run(X0, 40)
    assert not np.array_equal(s2.chain, runs[0][0])


def test_logging_ensemble_sampler_by_the_references_signature(tmp_path):
    """LoggingEnsembleSampler(nwalkers, ndim, log_prob_fn, pool=...).run_mcmc(X0, nsteps, status=...) as the
    reference drives it (src/mcmc.py:372-412): bound Chain.log_posterior stays on the device, any other callable is
    evaluated on the host — both walk the same chain as StretchSampler with that seed."""
    from gpbayestools_hic_amd import StretchSampler, synth
    from gpbayestools_hic_amd.mcmc import LoggingEnsembleSampler
    from gpbayestools_hic_amd.workload import build_chain
    chain, emu, info = build_chain(1, workdir=str(tmp_path))
    nw, d = 32, info["d"]
    X0 = synth.walkers(nw, d, seed=9)
    ref = StretchSampler(chain, nw, seed=77)
    ref.run(X0, 12)
    dev = LoggingEnsembleSampler(nw, d, chain.log_posterior, pool=chain, seed=77)
    last = dev.run_mcmc(X0, 12, status=5)
    assert np.array_equal(dev.chain, ref.chain) and np.array_equal(dev.lnprobability, ref.lnprobability)
    assert last.shape == (nw, d) and dev.chain.shape == (nw, 12, d) and dev.flatchain.shape == (nw * 12, d)
    host = LoggingEnsembleSampler(nw, d, lambda X: chain.log_posterior(X), seed=77)
    host.run_mcmc(X0, 12)
    assert np.array_equal(host.chain, ref.chain)          # same numbers through the host callback
    host.reset()
    assert host.iterations == 0 and np.all(host.acceptance_fraction == 0)


def test_chain_run_mcmc_schema_and_resume(tmp_path):
    from test_gpu_dropin import _chain
    g = golden("g5_chain.npz")
    ch = _chain(tmp_path, g)
    nw = 32
    ch.run_mcmc(nsteps=20, nburnsteps=10, nwalkers=nw, nthin=5, seed=3)
    with open(ch.mcmc_path, "rb") as f:
        data = pickle.load(f)
    assert set(data) == {"chain"} and data["chain"].shape == (nw, 4, ch.ndim)
    assert np.all((data["chain"] > ch.min) & (data["chain"] < ch.max))
    ch.run_mcmc(nsteps=10, nburnsteps=10, nwalkers=nw, nthin=5, seed=4)          # resumes: no burn-in, appends
    with open(ch.mcmc_path, "rb") as f:
        data2 = pickle.load(f)
    assert data2["chain"].shape == (nw, 6, ch.ndim)
    assert np.array_equal(data2["chain"][:, :4], data["chain"])
    ch.compute_log_likelihood_for_chain(str(tmp_path / "ll.pkl"))
    with open(tmp_path / "ll.pkl", "rb") as f:
        ll = pickle.load(f)["log_likelihood"]
    assert ll.shape == (nw, 6) and np.all(np.isfinite(ll))
    assert np.allclose(ll.reshape(-1), ch.log_likelihood(data2["chain"].reshape(-1, ch.ndim)), rtol=1e-12)


def test_log_posterior_is_batch_independent(tmp_path):
    from test_gpu_dropin import _chain
    g = golden("g5_chain.npz")
    ch = _chain(tmp_path, g)
    X = g["Xw"]
    full = ch.log_posterior(X)
    for sl in (slice(0, 1), slice(3, 40), slice(40, 64)):
        assert np.array_equal(ch.log_posterior(X[sl]), full[sl], equal_nan=True)


def test_split_permutation_is_a_bijection_and_changes_per_step():
    import torch
    from conftest import debug_engine
    from gpbayestools_hic_amd import _native as nat
    eng = debug_engine()                                           # gpb_test_split_perm: a hook of the debug library
    for n in (2, 6, 64, 100, 4096, 5000):
        outs = []
        for step in (0, 1, 2):
            out = torch.empty(n, dtype=torch.int64, device="cuda")
            eng._ck(eng.lib.gpb_test_split_perm(eng.h, n, 99, step, nat.ptr(out)))
            p = out.cpu().numpy()
            assert np.array_equal(np.sort(p), np.arange(n)), (n, step)
            outs.append(p)
        if n >= 64:
            assert not np.array_equal(outs[0], outs[1]) and not np.array_equal(outs[1], outs[2])
            # the induced halves look random: about half of the even positions land on even walkers
            frac = np.mean(outs[0][0::2] % 2 == 0)
            assert 0.3 < frac < 0.7
    eng.close()


def test_resident_c_loop_equals_the_host_driven_loop(tmp_path):
    """gpb_emcee_run (the C ABI enqueues every kernel of every step) against the loop that drives
    gpb_stretch_propose / log_prob_device / gpb_stretch_accept from Python: same chain, log-probabilities and
    acceptance counts bit for bit, also when continued in pieces and when the status interval cuts the run"""
    from gpbayestools_hic_amd import StretchSampler, synth
    from gpbayestools_hic_amd.workload import build_chain
    chain, emu, info = build_chain(1, workdir=str(tmp_path))
    nw, d = 48, info["d"]
    X0 = synth.walkers(nw, d, seed=21)
    X0[5, 0] = 1.7                                            # a walker that starts outside the prior box
    c = StretchSampler(chain, nw, seed=5)
    assert c._resident_engine()[0] is emu._engine_ready()
    c.run(X0, 7, status=3)
    c.run(None, 6, status=100)
    h = StretchSampler(chain, nw, seed=5)
    h._resident_engine = lambda: None                         # force the host-driven loop
    h.run(X0, 13, status=4)
    assert np.array_equal(c.chain, h.chain) and np.array_equal(c.lnprobability, h.lnprobability)
    assert np.array_equal(c.naccept.cpu().numpy(), h.naccept.cpu().numpy()) and c.iterations == h.iterations == 13
    assert np.isneginf(c.lnprobability[5, 0]) or np.isfinite(c.lnprobability[5, 0])
    # store=False keeps no chain but walks the same ensemble
    n = StretchSampler(chain, nw, seed=5)
    last = n.run(X0, 13, store=False)
    assert np.array_equal(last, c.chain[:, -1])


def test_resident_loop_with_a_one_rank_communicator(tmp_path):
    """the sharded form of gpb_emcee_run (rows of this rank + in-stream ncclAllGather) with the only communicator a
    one-GPU box can form: world 1, same numbers as the unsharded loop"""
    import ctypes
    import torch
    from gpbayestools_hic_amd import StretchSampler, synth
    from gpbayestools_hic_amd.workload import build_chain
    chain, emu, info = build_chain(1, workdir=str(tmp_path))
    eng = emu._engine_ready()
    nw = 32
    X0 = synth.walkers(nw, info["d"], seed=22)
    ref = StretchSampler(chain, nw, seed=9)
    ref.run(X0, 8)
    try:
        eng.dist_init(0, 1, eng.dist_uid())
    except Exception as e:
        pytest.skip("no RCCL communicator on this box: %s" % e)
    sh = types.SimpleNamespace(world=1, rank=0, direct=eng,
                               logprob=lambda fn, X, out: fn(X, out))
    s = StretchSampler(chain, nw, seed=9, sharding=sh)
    assert s._resident_engine()[0] is eng
    s.run(X0, 8)
    assert np.array_equal(s.chain, ref.chain) and np.array_equal(s.lnprobability, ref.lnprobability)
    # a sharded run lets the ranks AGREE that gpb_chain_emcee_prepare (state checks, workspaces) succeeded everywhere before
    # any of them enqueues the in-stream all-gathers: a rank whose peers report a failure raises without touching the
    # ensemble, and an own failure is what gets reported
    votes = []
    lib = eng.lib
    assert lib.gpb_chain_emcee_prepare((ctypes.c_void_p * 1)(eng.h), 1, nw) == 0
    assert lib.gpb_chain_emcee_prepare((ctypes.c_void_p * 1)(eng.h), 1, nw + 1) < 0          # odd ensemble
    assert lib.gpb_chain_emcee_prepare(None, 1, nw) < 0
    sh2 = types.SimpleNamespace(world=2, rank=0, direct=eng, logprob=lambda fn, X, out: fn(X, out),
                                _all_ok=lambda ok: (votes.append(ok), False)[1])              # "another rank failed"
    eng._dist_world = 1        # (the engine's communicator has one rank; the stub claims two: only the agreement is under test)
    s2 = StretchSampler(chain, nw, seed=9, sharding=sh2)
    s2.pos.copy_(s.pos); s2.lp.copy_(s.lp)
    before = s2.pos.clone()
    if s2._resident_engine() is not None:
        with pytest.raises(RuntimeError, match="another rank"):
            s2.run(None, 2)
        assert votes == [True] and torch.equal(s2.pos, before)
    eng.dist_finalize()


def test_one_ranks_share_measurement_hook(tmp_path, debug_lib):
    """the measurement hook of the debug library (option keys 26 / 32): one rank's share of a 4-way split on a single GPU (the
    rows it does not evaluate are rejected); the product library refuses the key"""
    from gpbayestools_hic_amd import GPEngine, StretchSampler, synth
    from gpbayestools_hic_amd._native import GPBError
    from gpbayestools_hic_amd.workload import build_chain
    chain, emu, info = build_chain(1, workdir=str(tmp_path))
    eng = emu._engine_ready()
    assert eng.has_variants
    nw = 32
    X0 = synth.walkers(nw, info["d"], seed=22)
    eng.tune("sim_ranks", 4)
    m = StretchSampler(chain, nw, seed=9)
    m.run(X0, 4)
    eng.tune("sim_ranks", 0)
    assert m.chain.shape == (nw, 4, info["d"]) and np.all(np.isfinite(m.chain))
    product = GPEngine(0, debug=False)
    with pytest.raises(GPBError, match="debug build"):
        product.tune("sim_ranks", 4)
    product.close()


def test_nan_log_probability_raises_like_emcee(tmp_path):
    """emcee aborts with "Probability function returned NaN"; on the device a NaN proposal is rejected and counted, and
    the sampler raises at its next status check (both loops)"""
    import torch
    from gpbayestools_hic_amd import StretchSampler, synth
    from gpbayestools_hic_amd import _native as nat
    from gpbayestools_hic_amd.workload import build_chain
    chain, emu, info = build_chain(1, workdir=str(tmp_path))
    nw = 32
    X0 = synth.walkers(nw, info["d"], seed=23)
    for force_host in (False, True):
        good_cov = chain.expdata_cov
        s = StretchSampler(chain, nw, seed=3)
        if force_host:
            s._resident_engine = lambda: None
        s.run(X0, 4)
        chain.expdata_cov = -0.5 * np.eye(chain.nobs)          # an indefinite covariance: every block is NaN
        before = (s.pos.clone(), s.lp.clone(), s.naccept.clone(), s.iterations, s._step_counter, s.chain.copy())
        with pytest.raises(ValueError, match="NaN"):
            s.run(None, 3)
        # emcee leaves its state untouched when it raises: the sampler is back in front of the offending block
        assert torch.equal(s.pos, before[0]) and torch.equal(s.lp, before[1]) and torch.equal(s.naccept, before[2])
        assert (s.iterations, s._step_counter) == before[3:5] and np.array_equal(s.chain, before[5])
        with pytest.raises(ValueError, match="initial log_prob was NaN"):
            StretchSampler(chain, nw, seed=3).run(X0, 1)
        chain.expdata_cov = good_cov
        s.run(None, 2)                                         # ... and carries on from there once the cause is gone
        s2 = StretchSampler(chain, nw, seed=3)
        s2.run(X0, 6)
        assert np.array_equal(s2.chain, s.chain)               # as if the failed call had never happened
        # a count left behind by another user of the context (gpb_stretch_accept on NaN log-probabilities, never read)
        # is not this run's: run() clears the counter when it starts
        eng = emu._engine_ready()
        nanlp = torch.full((nw // 2,), float("nan"), dtype=torch.float64, device="cuda")
        zeros = torch.zeros(nw // 2, dtype=torch.float64, device="cuda")
        eng._ck(eng.lib.gpb_stretch_accept(eng.h, nat.ptr(s2.pos.clone()), nat.ptr(s2.lp.clone()), nw, info["d"], 0, 1, 0,
                                           nat.ptr(s2.q), nat.ptr(zeros), nat.ptr(nanlp), None, 1))
        s3 = StretchSampler(chain, nw, seed=3)
        if force_host:
            s3._resident_engine = lambda: None
        s3.run(X0, 2)


def test_nan_in_a_later_status_block_keeps_the_blocks_before_it():
    """a NaN in the third status block of a call: the sampler goes back in front of THAT block; the two blocks before it stay —
    stored samples, step count and acceptance counts describe the same four steps, and the run can be continued"""
    from gpbayestools_hic_amd import StretchSampler
    d, nw = 3, 16
    fake = types.SimpleNamespace(ndim=d, device=0, min=np.full(d, -9.0), max=np.full(d, 9.0), emuList=[])
    calls = {"n": 0, "poison": True}

    def logprob(X, out):
        calls["n"] += 1
        out.copy_(-0.5 * (X * X).sum(1))
        if calls["poison"] and calls["n"] == 1 + 2 * 4 + 2:      # initial call + two half-steps per step: inside step 5
            out[0] = float("nan")
        return out

    X0 = np.random.default_rng(2).normal(size=(nw, d))
    s = StretchSampler(fake, nw, seed=5, logprob_device=logprob)
    with pytest.raises(ValueError, match="4 step"):
        s.run(X0, 8, status=2)
    assert s.iterations == 4 and s.chain.shape == (nw, 4, d) and s.lnprobability.shape == (nw, 4)
    assert np.array_equal(s.chain[:, -1], s.pos.cpu().numpy()) and np.array_equal(s.lnprobability[:, -1], s.lp.cpu().numpy())
    ref = StretchSampler(fake, nw, seed=5, logprob_device=lambda X, out: out.copy_(-0.5 * (X * X).sum(1)))
    ref.run(X0, 8, status=2)
    assert np.array_equal(s.chain, ref.chain[:, :4])
    calls["poison"] = False
    s.run(None, 4, status=2)                                      # carries on: the whole chain equals the clean run's
    assert np.array_equal(s.chain, ref.chain) and np.array_equal(s.naccept.cpu().numpy(), ref.naccept.cpu().numpy())


def test_box_test_in_the_proposal_kernel_equals_the_marking_kernel(tmp_path):
    """the C-driven loop's proposal kernel takes the prior-box test itself and gathers the rows inside the box (slots
    from a counter, in whatever order the walkers arrive), or leaves 0/1 flags for the gather kernel to rank (several
    256-row workgroups, the last one partial, many proposals outside the box) — same ensemble as with the compaction's
    own kernels and as the host-driven loop; likewise with the accept of a half-step and the proposal of the next in one
    launch (every walker group re-derives the pending accept decisions of the two walkers it reads) or in two"""
    from gpbayestools_hic_amd import StretchSampler, synth
    from gpbayestools_hic_amd.workload import build_chain
    chain, emu, info = build_chain(1, workdir=str(tmp_path))
    eng = emu._engine_ready()
    nw, d = 1364, info["d"]                                   # 682 rows per batch: 2 full workgroups + 170 rows
    X0 = synth.walkers(nw, d, seed=8)
    runs = {}
    for tag, premark, host in (("premark", 2, False), ("two_launches", 2, False), ("flags_only", 1, False),
                               ("mark_kernel", 0, False), ("host", 2, True)):
        eng.tune("premark", premark)
        eng.tune("fuse_accept_propose", 0 if tag == "two_launches" else 1)
        s = StretchSampler(chain, nw, seed=31, randomize_split=(tag != "flags_only"))
        if host:
            s._resident_engine = lambda: None
        s.run(X0, 3, status=100)
        s.run(None, 2, status=1)                              # continued, one step per C call
        runs[tag] = (s.chain, s.lnprobability, s.naccept.cpu().numpy())
    eng.tune("premark", 2)
    for tag in ("two_launches", "mark_kernel", "host"):
        for a, b in zip(runs["premark"], runs[tag]):
            assert np.array_equal(a, b), tag
    # the unshuffled split (emcee's randomize_split=False: identity permutation) through the fused launch and the host loop
    eng.tune("premark", 2)
    s = StretchSampler(chain, nw, seed=31, randomize_split=False)
    s._resident_engine = lambda: None
    s.run(X0, 3, status=100)
    s.run(None, 2, status=1)
    assert np.array_equal(s.chain, runs["flags_only"][0]) and np.array_equal(s.lnprobability, runs["flags_only"][1])
    s2 = StretchSampler(chain, nw, seed=31, randomize_split=False)
    s2.run(X0, 5, status=100)
    assert np.array_equal(s2.chain, s.chain) and np.array_equal(s2.naccept.cpu().numpy(), s.naccept.cpu().numpy())
    lnp = runs["premark"][1]
    assert 0 < runs["premark"][2].sum() < 5 * nw and np.isfinite(lnp).all()


@pytest.mark.parametrize("fuse,balance", [(1, 1), (0, 1), (1, 0), (0, 0)])
def test_every_ranks_share_of_the_c_loop_equals_the_host_loop_on_those_rows(tmp_path, fuse, balance, debug_lib):
    """the sharded C loop evaluates one rank's share of every batch — balanced: the r-th of R equal slices of the ordered
    list of ALL rows inside the box (k_balance_gather; the accept kernels find a row's value through its rank in that
    list); contiguous: the rows inside the box of proposals [r chunk, (r + 1) chunk).  Played for every rank r of 3 on one
    GPU (the other ranks' values stay -inf: rejected) it walks the same ensemble as the host-driven loop whose
    log-probability evaluates exactly those rows"""
    import torch
    from gpbayestools_hic_amd import StretchSampler, synth
    from gpbayestools_hic_amd.workload import build_chain
    chain, emu, info = build_chain(1, workdir=str(tmp_path))
    eng = emu._engine_ready()
    R, nw = 3, 1092                                           # 546 rows per batch, 182 per rank
    chunk = nw // 2 // R
    X0 = synth.walkers(nw, info["d"], seed=4)
    lo, hi = chain._box(torch.device("cuda", 0))
    eng.tune("fuse_accept_propose", fuse)
    eng.tune("balance_shards", 2 if balance else 0)
    slices = []
    for r in range(R):
        eng.tune("sim_ranks", R)
        eng.tune("sim_rank", r)
        c = StretchSampler(chain, nw, seed=17)
        assert c._resident_engine() is not None
        c.run(X0, 4, status=100)
        eng.tune("sim_ranks", 0)
        eng.tune("sim_rank", 0)

        def rows_of_rank(X_dev, out, r=r):
            out.fill_(float("-inf"))
            if X_dev.shape[0] == nw:                          # the starting positions: every rank evaluates them all
                return chain.log_prob_device(X_dev, out=out)
            if balance:
                inside = ((X_dev > lo) & (X_dev < hi)).all(dim=1)
                g = torch.cumsum(inside.to(torch.int64), 0) - 1
                per = max(-(-int(inside.sum().item()) // R), 1)
                mine = inside & (g // per == r)
                slices.append(int(mine.sum().item()))
                if mine.any():
                    out[mine] = chain.log_prob_device(X_dev[mine].contiguous())
            else:
                sl = slice(r * chunk, (r + 1) * chunk)
                out[sl] = chain.log_prob_device(X_dev[sl].contiguous())
            return out

        h = StretchSampler(chain, nw, seed=17, logprob_device=rows_of_rank)
        assert h._resident_engine() is None
        h.run(X0, 4, status=100)
        assert np.array_equal(c.chain, h.chain), r
        assert np.array_equal(c.lnprobability, h.lnprobability) and np.array_equal(c.naccept.cpu().numpy(), h.naccept.cpu().numpy())
        moved = np.any(c.chain[:, -1] != X0, axis=1)
        assert 0 < moved.sum() < nw
    if balance:
        assert max(slices) - min(s for s in slices if s) <= 60 and max(slices) <= chunk    # equal slices, within the collective's size
    eng.tune("fuse_accept_propose", 1)
    eng.tune("balance_shards", 0)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("balance", [0, 2])
@pytest.mark.parametrize("R", [2, 4, 8])
def test_sharded_c_loop_with_R_ranks_in_one_process(tmp_path, R, balance, debug_lib):
    """The R > 1 form of gpb_chain_emcee_run as a whole: R contexts (own streams, own host threads) joined by the loopback
    communicator of gpb_debug_loopback_group — per-rank row shares, in-stream all-gathers, accept steps fed by the other
    ranks' log-probabilities.  RCCL refuses two ranks on one device, so this is what a one-GPU box can run of it: everything
    but the wire.  Every rank must hold the ensemble of the unsharded run, bit for bit (replicated draws + gathered values).
    balance = 2: every rank takes an equal slice of the ordered list of ALL rows inside the box (k_balance_gather, tune key
    36; off by default) instead of the rows inside the box of its own contiguous share — the check a multi-rank run owed
    before that mode may be switched on."""
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd.workload import build_chain
    built = []
    for r in range(R):
        (tmp_path / ("r%d" % r)).mkdir()
        built.append(build_chain(1, workdir=str(tmp_path / ("r%d" % r))))
    d = built[0][2]["d"]
    nw, nsteps = 64, 7
    X0 = synth.walkers(nw, d, seed=31)
    X0[::5, 0] = 0.999                                               # a few walkers at the edge: proposals leave the box
    _loopback_ranks_reproduce_the_unsharded_run(built, nw, nsteps, X0, balance)


def _loopback_ranks_reproduce_the_unsharded_run(built, nw, nsteps, X0, balance=0):
    import ctypes
    import threading
    from gpbayestools_hic_amd import StretchSampler
    R = len(built)
    ref = StretchSampler(built[0][0], nw, seed=13)
    ref.run(X0, nsteps)
    engs = [b[1]._engine_ready() for b in built]
    lib = engs[0].lib
    for b in built:
        b[0]._prepare_blocks()
    assert lib.gpb_debug_loopback_group((ctypes.c_void_p * R)(*[e.h for e in engs]), R) == 0
    for e in engs:
        e.tune("balance_shards", balance)
    try:
        samplers, errors = [None] * R, []

        def work(r):
            try:
                sh = types.SimpleNamespace(world=R, rank=r, direct=engs[r], logprob=lambda fn, X, out: fn(X, out),
                                           _all_ok=lambda ok: ok)
                s = StretchSampler(built[r][0], nw, seed=13, sharding=sh)
                assert s._resident_engine() is not None and s._resident_engine()[0] is engs[r]
                s.run(X0, nsteps)
                samplers[r] = s
            except Exception as e:                                    # a failing rank would leave the others waiting
                errors.append((r, repr(e)))

        threads = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(R)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=120)
        assert not errors and all(not t.is_alive() for t in threads), errors
        for r in range(R):
            assert np.array_equal(samplers[r].chain, ref.chain), r
            assert np.array_equal(samplers[r].lnprobability, ref.lnprobability), r
            assert np.array_equal(samplers[r].naccept.cpu().numpy(), ref.naccept.cpu().numpy()), r
    finally:
        assert lib.gpb_debug_loopback_release(engs[0].h) == 0
    return ref


@pytest.mark.timeout(300)
def test_sharded_c_loop_when_the_gps_differ_in_their_distance_form(tmp_path, debug_lib):
    """The distance form of a GP's kernel matrices is chosen from theta alone (Gram form, or sklearn's difference form when a
    length scale is far below the design's extent: gpbayes.h GPB_GET_FORM): with one GP of the emulator on either side of the
    rule — two cross-kernel launches per batch — two loopback ranks still end on the unsharded ensemble bit for bit, and the
    ensemble is not the one the all-Gram emulator gives."""
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd.workload import build_chain
    R = 2
    built = []
    for r in range(R):
        (tmp_path / ("r%d" % r)).mkdir()
        chain, emu, info = build_chain(1, workdir=str(tmp_path / ("r%d" % r)))
        th = synth.fixed_theta(info["d"], info["P"])
        th[1, 1 + 3] = np.log(0.004)                                 # GP 1: parameter 3 at 0.004 of its extent, S = 6e4
        th[2, 1:1 + info["d"]] = np.log(0.11)                        # GP 2: every length scale short, S = 8 / 0.11^2 = 661: Gram
        emu.trainEmulator([True] * emu.nev, kernel_type=info["kernel_type"], thetas=th)
        built.append((chain, emu, info))
    eng = built[0][1]._engine_ready()
    assert eng.get("form").tolist() == [0.0, 1.0, 0.0, 0.0]
    d = built[0][2]["d"]
    nw = 64
    X0 = synth.walkers(nw, d, seed=31)
    X0[::5, 0] = 0.999
    ref = _loopback_ranks_reproduce_the_unsharded_run(built, nw, 7, X0)
    lp = built[0][0].log_posterior(X0)                               # and batch cuts change no bit either
    for sl in (slice(0, 1), slice(3, 40), slice(40, 64)):
        assert np.array_equal(built[0][0].log_posterior(X0[sl]), lp[sl])
    assert np.all(np.isfinite(ref.lnprobability))


@pytest.mark.timeout(300)
def test_sharded_c_loop_over_a_chain_of_emulators(tmp_path, debug_lib):
    """... and the same for a chain of three emulators (mixed kernels, batched cross / predict / likelihood launches) on two
    and three loopback ranks: the first emulator's context carries the communicator, every rank ends on the unsharded ensemble."""
    import ctypes
    import threading
    from gpbayestools_hic_amd import StretchSampler, synth
    from gpbayestools_hic_amd.workload import build_multi_chain
    specs = [(96, 12, 3, "RBF"), (128, 20, 4, "Matern25"), (80, 9, 2, "RBF")]
    D = 6
    for R in (2, 3):
        built = []
        for r in range(R):
            wd = tmp_path / ("R%d_r%d" % (R, r)); wd.mkdir()
            built.append(build_multi_chain(specs, D, workdir=str(wd)))
        nw, nsteps = 48, 5                                            # 24 rows a batch: 12 / 8 per rank
        X0 = np.clip(built[0][2]["xstar"] + 0.05 * np.random.default_rng(2).standard_normal((nw, D)), 0.02, 0.98)
        ref = StretchSampler(built[0][0], nw, seed=21)
        assert ref._resident_engine()[2] == 3
        ref.run(X0, nsteps)
        firsts = [b[1][0]._engine_ready() for b in built]            # the communicator lives on the chain's first context
        for b in built:
            b[0]._prepare_blocks()
        lib = firsts[0].lib
        assert lib.gpb_debug_loopback_group((ctypes.c_void_p * R)(*[e.h for e in firsts]), R) == 0
        try:
            samplers, errors = [None] * R, []

            def work(r):
                try:
                    sh = types.SimpleNamespace(world=R, rank=r, direct=firsts[r], logprob=lambda fn, X, out: fn(X, out),
                                               _all_ok=lambda ok: ok)
                    s = StretchSampler(built[r][0], nw, seed=21, sharding=sh)
                    assert s._resident_engine() is not None
                    s.run(X0, nsteps)
                    samplers[r] = s
                except Exception as e:
                    errors.append((r, repr(e)))

            threads = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(R)]
            for t in threads:
                t.start()
            for t in threads:
                t.join(timeout=120)
            assert not errors and all(not t.is_alive() for t in threads), errors
            for r in range(R):
                assert np.array_equal(samplers[r].chain, ref.chain), (R, r)
                assert np.array_equal(samplers[r].lnprobability, ref.lnprobability), (R, r)
        finally:
            assert lib.gpb_debug_loopback_release(firsts[0].h) == 0


def test_loopback_group_survives_a_member_being_destroyed(debug_lib):
    """test hook hygiene: a context that is destroyed without a release of its loopback group leaves the group; the group goes
    with its last member or with an explicit release"""
    import ctypes
    from gpbayestools_hic_amd import GPEngine
    a, b, c = GPEngine(0), GPEngine(0), GPEngine(0)
    lib = a.lib
    assert lib.gpb_debug_loopback_group((ctypes.c_void_p * 3)(a.h, b.h, c.h), 3) == 0
    assert lib.gpb_debug_loopback_group((ctypes.c_void_p * 2)(a.h, b.h), 2) < 0        # already members of a group
    b.close()
    assert lib.gpb_debug_loopback_release(a.h) == 0
    assert lib.gpb_debug_loopback_release(a.h) < 0                                     # released: no group
    assert lib.gpb_debug_loopback_group((ctypes.c_void_p * 2)(a.h, c.h), 2) == 0
    a.close(); c.close()                                                               # the group goes with its last member
