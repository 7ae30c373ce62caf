"""
GPU tier: BASELINE configs 2, 3 and 5 as WORKLOADS (workload.build_chain, the same objects bench.py times),
each against the oracle on a row sample taken from inside the full-size batch:
  cfg 2  1024 design pts x 15 params, RBF: Emulator.predict(return_cov=True) on 10 000 test points
  cfg 3  same GP, 1024-walker stretch move: 512-row log-posterior batches (M=32, P=10) and three sampler steps
         against emcee's algorithm evaluated with the ORACLE's log-posterior
  cfg 5  4096-pt Matern-5/2 design, 8192-row log_likelihood(finite=True) batch (P=10, M=64), driven through
         run_pocoMC with a recording stand-in for the absent `pocomc` package (src/mcmc.py:752-819)
(cfg 1 and cfg 4 are covered by test_gpu_dropin.py / test_gpu_fullsize.py.)
"""
import pickle
import sys
import types

import numpy as np
import pytest

from conftest import maxrel, relerr

pytestmark = pytest.mark.gpu


_ORACLES = {}


def _oracle(info, kind="RBF"):
    """the CPU oracle's emulator on the workload's data at the fixed timing hyper-parameters (cached per shape)"""
    from gpbayestools_hic_amd import synth
    from oracle import gp_oracle as O
    key = (info["N"], info["d"], info["M"], info["P"], kind)
    if key not in _ORACLES:
        oe = O.OracleEmulator(info["X"], info["Y"], info["lo"], info["hi"], info["P"], O.KIND_NAMES[kind])
        _ORACLES[key] = oe.fit(synth.fixed_theta(info["d"], info["P"]))
    return _ORACLES[key]


def _oracle_logprob(info, oe, X, **kw):
    from oracle import gp_oracle as O
    yexp = info["yexp"]
    cexp = np.diag((0.05 * np.abs(yexp)) ** 2)
    return O.log_prob(X, info["lo"], info["hi"], lambda x, e: oe.predict(x, True, e), yexp, cexp, **kw)


# ------------------------------------------------------------------------------------------- cfg 2
def test_cfg2_predict_10000_points_with_covariance(tmp_path):
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd.workload import build_chain
    _, emu, info = build_chain(2, workdir=str(tmp_path))
    assert (info["N"], info["d"], info["M"], info["P"]) == (1024, 15, 32, 10)
    Xs = synth.walkers(10000, info["d"], seed=31)
    mean, cov = emu.predict(Xs, return_cov=True, extra_std=0.0)
    assert mean.shape == (10000, 32) and cov.shape == (10000, 32, 32)
    oe = _oracle(info)
    rows = np.random.default_rng(2).choice(10000, 64, replace=False)
    m_ref, c_ref = oe.predict(Xs[rows], True, 0.0)
    assert relerr(mean[rows], m_ref) < 1e-11
    for i, r in enumerate(rows):
        assert maxrel(cov[r], c_ref[i]) < 1e-10
    # per-GP mean / variance of the same points (BASELINE metric ii is quoted on these)
    gm, gv = emu._engine_ready().predict(Xs)
    gm_ref, gv_ref = oe.gp_predict(Xs[rows])
    assert maxrel(gm[rows], gm_ref) < 1e-11 and relerr(gv[rows], gv_ref) < 1e-10
    assert np.all(gv > 0)
    # the host-to-host call (numpy in, page-locked numpy out) gives the bits of the device-resident one (torch in, torch out)
    import torch
    md, cd = emu._engine_ready().emu_predict(torch.as_tensor(Xs, device="cuda"), True, 0.0)
    assert np.array_equal(md.cpu().numpy(), mean) and np.array_equal(cd.cpu().numpy(), cov)
    del md, cd
    es_all = np.linspace(0.0, 0.3, 10000)                    # ... and with a per-row extra_std
    m3, c3 = emu.predict(Xs, return_cov=True, extra_std=es_all)
    _, c3_ref = oe.predict(Xs[rows], True, es_all[rows])
    assert np.array_equal(m3, mean) and maxrel(c3[rows], c3_ref) < 1e-10
    del m3, c3
    # mean-only call and a non-zero extra_std
    assert np.array_equal(emu.predict(Xs[:300], return_cov=False), mean[:300])
    es = np.linspace(0.0, 0.3, 64)
    _, c2 = emu.predict(Xs[rows], return_cov=True, extra_std=es)
    _, c2_ref = oe.predict(Xs[rows], True, es)
    assert maxrel(c2, c2_ref) < 1e-10


# ------------------------------------------------------------------------------------------- cfg 3
@pytest.fixture(scope="module")
def cfg3(tmp_path_factory):
    from gpbayestools_hic_amd.workload import build_chain
    chain, emu, info = build_chain(3, workdir=str(tmp_path_factory.mktemp("cfg3")))
    return chain, emu, info, _oracle(info)


def test_cfg3_512_row_log_posterior_batch(cfg3):
    from gpbayestools_hic_amd import synth
    chain, emu, info, oe = cfg3
    assert (info["N"], info["d"], info["M"], info["P"], info["W"]) == (1024, 15, 32, 10, 512)
    X = synth.walkers(512, info["d"], seed=41)
    X[7, 2] = 1.25; X[100, 14] = -0.5; X[333, 0] = 1.0; X[511, 5] = 0.0      # outside / on the boundary
    lp = chain.log_posterior(X)
    out = np.zeros(512, bool); out[[7, 100, 333, 511]] = True
    assert np.all(np.isneginf(lp[out])) and np.all(np.isfinite(lp[~out]))
    rows = np.concatenate([[7, 100, 333, 511], np.random.default_rng(3).choice(512, 44, replace=False)])
    ref = _oracle_logprob(info, oe, X[rows])
    fin = np.isfinite(ref)
    assert np.array_equal(np.isneginf(lp[rows]), ~fin)
    assert relerr(lp[rows][fin], ref[fin]) < 1e-10
    ll = chain.log_likelihood(X, finite=True)
    assert np.all(ll[out] == -1e300) and np.array_equal(ll[~out], lp[~out])


def test_cfg3_the_whole_run_1024_walkers_2000_steps(cfg3, tmp_path):
    """BASELINE config 3 as it is written: emcee stretch move, 1024 walkers x 2000 steps (src/mcmc.py:345-426 through
    Chain.run_mcmc: two-stage burn-in with re-seeding at the best points, production, thinning, the chain pickle).  The run's
    own numbers are checked against the device log-posterior re-evaluated at the stored positions, and a sample of those
    against the oracle."""
    import pickle as pk
    from gpbayestools_hic_amd import StretchSampler
    chain, emu, info, oe = cfg3
    nw, nsteps = 1024, 2000
    chain.mcmc_path = tmp_path / "chain.pkl"
    chain.chain = False
    chain.run_mcmc(nsteps=nsteps, nburnsteps=60, nwalkers=nw, nthin=10, status=500, seed=11)
    with open(chain.mcmc_path, "rb") as f:
        stored = pk.load(f)["chain"]
    assert stored.shape == (nw, nsteps // 10, info["d"]) and np.all(np.isfinite(stored))
    assert np.all((stored > info["lo"]) & (stored < info["hi"]))              # every stored position inside the prior box
    af = chain.acceptance_fraction
    assert af.shape == (nw,) and 0.15 < af.mean() < 0.8
    # the same production run again (same seed sequence: run_mcmc hands `seed` to the sampler): reproducible
    s = StretchSampler(chain, nw, seed=5)
    X0 = stored[:, -1, :]
    s.run(X0, 50, status=25)
    last = s.chain[:, -1]
    lp_again = chain.log_posterior(last)
    assert np.array_equal(lp_again, s.lnprobability[:, -1])                    # stored log-probabilities belong to the positions
    rows = np.random.default_rng(2).choice(nw, 24, replace=False)
    ref = _oracle_logprob(info, oe, last[rows])
    assert relerr(lp_again[rows], ref) < 1e-10
    # the posterior the ensemble settled on is tighter than the prior in the constrained directions
    assert stored[:, -1, :].std(0).min() < 0.25


def test_cfg3_stretch_move_1024_walkers_against_emcees_algorithm(cfg3):
    """three steps of the 1024-walker ensemble (six 512-row batches): the oracle's emcee step, evaluated with the
    ORACLE's log-posterior and the device's draws, takes the same accept decisions and ends on the same positions"""
    from gpbayestools_hic_amd import StretchSampler, synth
    from test_gpu_sampler_step import _oracle_chain
    chain, emu, info, oe = cfg3
    nw, seed = 1024, 2025
    X0 = synth.walkers(nw, info["d"], seed=43)
    s = StretchSampler(chain, nw, seed=seed)
    s.run(X0, 3)
    X, lp, nacc, _ = _oracle_chain(X0, 3, seed, True, s._engine(), lambda q: _oracle_logprob(info, oe, q))
    assert np.array_equal(s.naccept.cpu().numpy(), nacc) and 0 < nacc.sum() < 3 * nw
    assert np.array_equal(s.chain[:, -1], X)
    assert relerr(s.lnprobability[:, -1], lp) < 1e-10


# ------------------------------------------------------------------------------------------- cfg 5 + pocoMC
class _RecordingSampler:
    """stand-in for pocomc.Sampler (pocomc==1.2.6 is absent): records its keyword arguments, evaluates the
    likelihood callback once on an n_prior-row batch drawn from the prior widened past the box, returns canned
    posterior()/evidence() of the shapes pocoMC documents"""
    last = None

    def __init__(self, **kw):
        self.kw = kw
        _RecordingSampler.last = self

    def run(self, n_total=None, n_evidence=None):
        self.run_kw = dict(n_total=n_total, n_evidence=n_evidence)
        prior = self.kw["prior"]
        rng = np.random.default_rng(self.kw["random_state"])
        n = self.kw["n_prior"]
        X = np.column_stack([dist.ppf(rng.uniform(-0.002, 1.002, n).clip(0, 1)) + 0.0 for dist in prior.dists])
        edge = rng.random(n) < 0.01                       # rows pushed out of the box
        X[edge, 0] = X[edge, 0] + 2.0
        self.X = X
        assert self.kw["vectorize"] is True
        self.logl = np.asarray(self.kw["likelihood"](X, **self.kw["likelihood_kwargs"]))

    def posterior(self):
        k = 100
        w = np.full(k, 1.0 / k)
        return self.X[:k].copy(), w, self.logl[:k].copy(), np.zeros(k)

    def evidence(self):
        return -123.5, 0.25


class _Prior:
    def __init__(self, dists):
        self.dists = dists
        self.dim = len(dists)


@pytest.fixture(scope="module")
def cfg5(tmp_path_factory):
    from gpbayestools_hic_amd.workload import build_chain
    chain, emu, info = build_chain(5, workdir=str(tmp_path_factory.mktemp("cfg5")))
    return chain, emu, info


def test_cfg5_run_pocomc_8192_row_likelihood_batches(cfg5, monkeypatch):
    chain, emu, info = cfg5
    assert (info["N"], info["d"], info["M"], info["P"], info["kernel"]) == (4096, 20, 64, 10, "Matern25")
    fake = types.ModuleType("pocomc")
    fake.Sampler, fake.Prior = _RecordingSampler, _Prior
    monkeypatch.setitem(sys.modules, "pocomc", fake)
    # the reference's notebooks pass pool=12 (examples/RunBayesianAnalysis.ipynb:85): pocoMC would FORK workers that
    # inherit this chain; here the batch is already one device call and pool=None reaches the sampler
    chain.run_pocoMC(n_effective=512, n_active=256, n_prior=8192, n_total=1000, n_evidence=0, pool=12)
    smp = _RecordingSampler.last
    # the call shape of the reference (src/mcmc.py:798-805)
    kw = smp.kw
    assert set(kw) == {"prior", "likelihood", "likelihood_kwargs", "n_effective", "n_active", "n_prior", "sample",
                       "n_max_steps", "random_state", "vectorize", "pool"}
    assert kw["likelihood_kwargs"] == {"finite": True} and kw["vectorize"] is True and kw["pool"] is None
    assert (kw["n_effective"], kw["n_active"], kw["n_prior"], kw["sample"], kw["n_max_steps"], kw["random_state"]) == \
        (512, 256, 8192, "tpcn", 200, 42)
    assert kw["likelihood"].__self__ is chain and kw["likelihood"].__func__.__name__ == "log_likelihood"
    assert smp.run_kw == {"n_total": 1000, "n_evidence": 0}
    # default prior: uniform(min, max - min) per parameter (src/mcmc.py:786-792)
    assert kw["prior"].dim == chain.ndim
    for i, dist in enumerate(kw["prior"].dists):
        assert dist.support() == (chain.min[i], chain.max[i])
    # the 8192-row batch: -1e300 outside the open box, oracle values on a sample inside
    X, logl = smp.X, smp.logl
    assert X.shape == (8192, 20) and logl.shape == (8192,)
    inside = np.all((X > chain.min) & (X < chain.max), axis=1)
    assert 50 < np.count_nonzero(~inside) < 4000
    assert np.all(logl[~inside] == -1e300) and np.all(np.isfinite(logl)) and np.all(logl[inside] > -1e299)
    oe = _oracle(info, "Matern25")
    rows = np.concatenate([np.flatnonzero(~inside)[:4], np.flatnonzero(inside)[::200][:36]])
    ref = _oracle_logprob(info, oe, X[rows], finite=True, posterior=False)
    assert np.array_equal(ref == -1e300, ~inside[rows])
    ins = inside[rows]
    assert relerr(logl[rows][ins], ref[ins]) < 1e-10
    assert np.array_equal(chain.log_posterior(X[rows][ins]), logl[rows][ins])       # same numbers inside the box
    # output pickle: exactly the reference's six keys (src/mcmc.py:816-819) ...
    with open(chain.mcmc_path, "rb") as f:
        data = pickle.load(f)
    assert set(data) == {"chain", "weights", "logl", "logp", "logz", "logz_err"}
    assert data["chain"].shape == (100, 20) and data["weights"].shape == data["logl"].shape == data["logp"].shape == (100,)
    assert (data["logz"], data["logz_err"]) == (-123.5, 0.25)
    # ... which the reference's consumer reads by key and sorts by log-likelihood
    # (examples/generate_posterior_clusters.py:10-21, 23-45: the indexing below is that script's)
    order = np.argsort(data["logl"])[::-1]
    for key in ("chain", "weights", "logl", "logp"):
        assert data[key][order].shape == data[key].shape
    assert np.all(np.diff(data["logl"][order]) <= 0)


def test_run_pocomc_custom_prior_dimension_check(cfg5, monkeypatch):
    chain, _, _ = cfg5
    fake = types.ModuleType("pocomc")
    fake.Sampler, fake.Prior = _RecordingSampler, _Prior
    monkeypatch.setitem(sys.modules, "pocomc", fake)
    with pytest.raises(ValueError, match="prior.dim does not match"):
        chain.run_pocoMC(prior=_Prior([None] * (chain.ndim - 1)))
    _RecordingSampler.last = None
    from scipy.stats import norm
    custom = _Prior([norm(0.5, 0.1)] * chain.ndim)
    chain.run_pocoMC(n_prior=256, prior=custom)
    assert _RecordingSampler.last.kw["prior"] is custom and _RecordingSampler.last.X.shape == (256, chain.ndim)


def test_cfg5_matern25_predict_sample(cfg5):
    """per-GP mean / variance of the Matern-5/2 emulator at N = 4096 against the oracle (P = 10)"""
    from gpbayestools_hic_amd import synth
    chain, emu, info = cfg5
    oe = _oracle(info, "Matern25")
    Xs = synth.walkers(40, info["d"], seed=51)
    gm, gv = emu._engine_ready().predict(Xs)
    gm_ref, gv_ref = oe.gp_predict(Xs)
    assert maxrel(gm, gm_ref) < 1e-11 and relerr(gv, gv_ref) < 1e-10
