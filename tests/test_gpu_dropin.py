"""
GPU tier: the drop-in `Emulator` / `Chain` classes against vectors captured from the reference's
own src/emulator.py and src/mcmc.py (every flag combination the reference has).
"""
import os
import pickle

import numpy as np
import pytest

from conftest import golden, relerr, maxrel

pytestmark = pytest.mark.gpu

VARIANTS = {
    "pca_rbf": ("RBF", {}),
    "pca_trunc": ("RBF", {}),
    "nopca_rbf": ("RBF", dict(perform_no_PCA=True)),
    "logexp_rbf": ("RBF", dict(logTrafo=True, exp_and_cov_diagonal=True)),
    "pca_matern": ("Matern", {}),
}


def _make_emulator(tmp_path, g, kw, tag="e"):
    from gpbayestools_hic_amd import Emulator, synth
    tp, pf = str(tmp_path / f"{tag}_train.pkl"), str(tmp_path / f"{tag}_par.txt")
    synth.write_training_pickle(tp, g["X"], g["Y"], g["Yerr"])
    synth.write_parameter_file(pf, g["lo"], g["hi"])
    return Emulator(training_set_path=tp, parameter_file=pf, npc=int(g["npc"]), **kw)


@pytest.mark.parametrize("name", list(VARIANTS))
def test_g3_g4_emulator_at_reference_theta(tmp_path, name):
    g = golden(f"g3_emulator_{name}.npz")
    ktype, kw = VARIANTS[name]
    emu = _make_emulator(tmp_path, g, kw)
    assert maxrel(emu.model_data, g["model_data"]) < 1e-15
    emu.trainEmulator([True] * emu.nev, kernel_type=ktype, thetas=g["thetas"])
    assert maxrel(emu.scaler.mean_, g["scaler_mean"]) < 1e-14
    assert maxrel(emu.scaler.scale_, g["scaler_scale"]) < 1e-14
    if "trans_matrix" in g.files:
        assert maxrel(emu._trans_matrix, g["trans_matrix"]) < 1e-11
        Zpc = np.random.default_rng(4).standard_normal((3, 2, emu.npc))          # (..., npc) -> (..., nobs)
        want = Zpc @ g["trans_matrix"][:emu.npc] + emu.scaler.mean_
        assert maxrel(emu._inverse_transform(Zpc), want) < 1e-11                 # src/emulator.py:366-375
        assert maxrel(emu._cov_trunc, g["cov_trunc"]) < 1e-11
        assert maxrel(emu._var_trans, g["var_trans"]) < 1e-11
    assert relerr(emu.lml_, g["lml"]) < 1e-10
    assert maxrel(np.array([gp.alpha_ for gp in emu.gps]), g["alpha_"]) < 1e-10
    assert maxrel(np.array([np.diag(gp.L_) for gp in emu.gps]), g["Ldiag"]) < 1e-11
    mean, cov = emu.predict(g["Xs"], return_cov=True, extra_std=g["extra_std"])
    assert relerr(mean, g["mean"]) < 1e-11
    assert maxrel(cov, g["cov"]) < 1e-10
    mean0, cov0 = emu.predict(g["Xs"], return_cov=True, extra_std=0)      # scalar works (crashes in the reference under numpy>=2)
    assert maxrel(cov0, g["cov0"]) < 1e-10
    assert relerr(emu.predict(g["Xs"], return_cov=False), g["mean_only"]) < 1e-11
    # joint covariance between query points and posterior draws (src/emulator.py:608-633)
    for i, gp in enumerate(emu.gps):
        mm, cfull = gp.predict(g["Xs"][:12], return_cov=True)
        assert maxrel(cfull, g["gp_cov12"][i]) < 1e-9
        assert maxrel(mm, g["gp_mean"][:12, i]) < 1e-11
    if "sample_y" in g.files:
        np.random.seed(4242)
        ys = emu.sample_y(g["Xs"][:12], n_samples=5, random_state=7)
        assert ys.shape == g["sample_y"].shape
        assert maxrel(ys, g["sample_y"]) < 1e-6        # SVD-based draws amplify 1e-10 covariance differences
    else:
        assert emu.sample_y(g["Xs"][:4]) is None
    # pickling round trip drops and rebuilds the device state
    import dill
    emu2 = dill.loads(dill.dumps(emu))
    m2, c2 = emu2.predict(g["Xs"], return_cov=True, extra_std=g["extra_std"])
    assert np.array_equal(m2, mean) and np.array_equal(c2, cov)


@pytest.mark.parametrize("name", ["pca_rbf", "pca_matern", "nopca_rbf"])
def test_g3_hyperparameter_search_on_device(tmp_path, name):
    """fit() drop-in: L-BFGS-B over the device LML reaches the reference optimum
    (theta* to optimiser tolerance, LML* to 1e-7 relative; SURVEY §7 tier ii)."""
    g = golden(f"g3_emulator_{name}.npz")
    ktype, kw = VARIANTS[name]
    emu = _make_emulator(tmp_path, g, kw)
    emu.trainEmulator([True] * emu.nev, kernel_type=ktype)
    assert relerr(emu.lml_, g["lml"]) < 1e-7
    assert np.allclose(emu.thetas_, g["thetas"], atol=5e-3)
    mean, cov = emu.predict(g["Xs"], return_cov=True, extra_std=g["extra_std"])
    assert relerr(mean, g["mean"]) < 1e-4


def _chain(tmp_path, g):
    from gpbayestools_hic_amd import Chain, Emulator, synth
    pf = str(tmp_path / "par.txt")
    synth.write_parameter_file(pf, g["lo"], g["hi"])
    emus = []
    for tag in ("A", "B"):
        tp = str(tmp_path / f"{tag}.pkl")
        synth.write_training_pickle(tp, g["X"], g[f"Y_{tag}"], 0.01)
        e = Emulator(training_set_path=tp, parameter_file=pf, npc=int(g[f"npc_{tag}"]))
        e.trainEmulator([True] * e.nev, thetas=g[f"thetas_{tag}"])
        emus.append(e)
    ep = str(tmp_path / "exp.pkl")
    synth.write_experiment_pickle(ep, g["yexp"], g["yerr"])
    ch = Chain(mcmc_path=str(tmp_path / "mcmc" / "chain.pkl"), expdata_path=ep, model_parafile=pf)
    ch.emuList = emus
    return ch


def test_g5_chain_log_probabilities(tmp_path):
    g = golden("g5_chain.npz")
    ch = _chain(tmp_path, g)
    Xw, ins = g["Xw"], g["inside"]
    assert np.array_equal(ch.expdata, g["expdata"]) and np.array_equal(ch.expdata_cov, g["expdata_cov"])
    assert np.array_equal(ch.log_prior(Xw), g["log_prior"])
    pm, pc = ch._predict(Xw[ins], extra_std=0.0)
    assert relerr(pm, g["predict_mean"]) < 1e-11
    assert maxrel(pc, g["predict_cov"]) < 1e-10
    post = ch.log_posterior(Xw)
    assert np.all(np.isneginf(post[~ins])) and np.array_equal(np.isneginf(post), ~ins)
    assert relerr(post[ins], g["log_posterior"][ins]) < 1e-10
    like = ch.log_likelihood(Xw)
    assert relerr(like[ins], g["log_likelihood"][ins]) < 1e-10
    likef = ch.log_likelihood(Xw, finite=True)
    assert np.all(likef[~ins] == -1e300)
    assert relerr(likef[ins], g["log_likelihood_finite"][ins]) < 1e-10
    assert np.array_equal(ch.log_posterior(g["Xout"]), g["log_posterior_out"])      # all-outside batch
    assert np.array_equal(ch.log_likelihood(g["Xout"], finite=True), g["log_likelihood_out_finite"])
    assert relerr(ch.log_posterior(Xw[0]), g["log_posterior_1d"]) < 1e-10          # 1-D input is promoted
    assert relerr(ch.log_likelihood_point_by_point(Xw)[ins], g["log_likelihood"][ins]) < 1e-10
    # generic (foreign-emulator) path gives the same numbers through _predict + device MVN
    class Foreign:
        def __init__(self, e): self.e, self.nobs = e, e.nobs
        def predict(self, X, return_cov=True, extra_std=0): return self.e.predict(X, return_cov, extra_std)
    ch2 = _chain(tmp_path, g)
    ch2.emuList = [Foreign(e) for e in ch2.emuList]
    assert relerr(ch2.log_posterior(Xw)[ins], g["log_posterior"][ins]) < 1e-10


def test_g5_chain_in_one_call_equals_the_sequenced_calls(tmp_path):
    """gpb_chain_logpost / gpb_chain_emcee_run over the two emulators of G5 (block-diagonal covariance,
    src/mcmc.py:153-166) against the per-emulator calls sequenced from Python: identical bits, for the
    log-probabilities and for a sampled chain"""
    import ctypes
    from gpbayestools_hic_amd import StretchSampler
    g = golden("g5_chain.npz")
    ch = _chain(tmp_path, g)
    X = np.concatenate([g["Xw"], g["Xout"], g["Xw"][::-1]])
    one = ch.log_posterior(X)
    engs = [e._engine_ready() for e in ch.emuList]
    assert engs[0].lib.gpb_chain_supported((ctypes.c_void_p * 2)(*[e.h for e in engs]), 2) == 1
    ch.use_chain_call = False
    assert np.array_equal(ch.log_posterior(X), one)
    assert np.array_equal(ch.log_likelihood(X, finite=True), np.where(np.isneginf(one), -1e300, one))
    nw = 40
    lo, hi = ch.min, ch.max
    X0 = lo + (hi - lo) * np.random.default_rng(8).uniform(0.2, 0.8, (nw, ch.ndim))
    host = StretchSampler(ch, nw, seed=3)
    assert host._resident_engine() is None                   # two emulators: only the chain call drives them from C
    host.run(X0, 8, status=3)
    ch.use_chain_call = True
    res = StretchSampler(ch, nw, seed=3)
    assert res._resident_engine()[2] == 2
    res.run(X0, 5, status=100)
    res.run(None, 3, status=2)
    assert np.array_equal(res.chain, host.chain) and np.array_equal(res.lnprobability, host.lnprobability)
    assert np.array_equal(res.naccept.cpu().numpy(), host.naccept.cpu().numpy())
    assert np.any(res.naccept.cpu().numpy() > 0)
    # a chain whose emulators sit on different streams or lack a likelihood block is refused, not mis-evaluated
    from gpbayestools_hic_amd import GPEngine
    bare = GPEngine(0)
    assert engs[0].lib.gpb_chain_supported((ctypes.c_void_p * 2)(engs[0].h, bare.h), 2) == 0
    bare.close()


def test_g6_mvn_loglike_function():
    from gpbayestools_hic_amd import mvn_loglike
    g = golden("g6_mvn.npz")
    for M in (4, 16, 64):
        assert abs(mvn_loglike(g[f"y_{M}"][0], g[f"cov_{M}"][0]) - g[f"ll_{M}"][0]) < 1e-11 * abs(g[f"ll_{M}"][0])
    with pytest.raises(np.linalg.LinAlgError):
        mvn_loglike(np.ones(4), -np.eye(4))


def test_emulator_flag_validation(tmp_path):
    g = golden("g3_emulator_pca_rbf.npz")
    with pytest.raises(ValueError):
        _make_emulator(tmp_path, g, dict(exp_and_cov_diagonal=True))      # needs logTrafo (src/emulator.py:59-60)


# ---------------------------------------------------------------- hold-out helpers + estimator surface (G8)
@pytest.mark.parametrize("name,kw", [("pca", {}), ("log", dict(logTrafo=True))])
def test_g8_holdout_helpers(tmp_path, name, kw):
    """testEmulatorErrors / testEmulatorErrorsWithTrainingPoints (src/emulator.py:636-726) against the reference's
    own trainEmulator + predict on the same split: at the reference's hyper-parameters to the predict bars, and
    after this build's own L-BFGS-B search to optimiser tolerance."""
    g = golden(f"g8_holdout_{name}.npz")
    ntest, nobs = int(g["ntest"]), g["Y"].shape[1]
    emu = _make_emulator(tmp_path, g, kw)
    for tag, fn, nrows in (("test", emu.testEmulatorErrors, ntest),
                           ("train", emu.testEmulatorErrorsWithTrainingPoints, emu.nev - ntest)):
        pred, perr, truth, terr = fn(nTestPoints=ntest, thetas=g[f"{tag}_thetas"])
        assert pred.shape == perr.shape == truth.shape == terr.shape == (nrows, nobs)
        assert maxrel(truth, g[f"{tag}_truth"]) < 1e-15 and maxrel(terr, g[f"{tag}_truth_err"]) < 1e-14
        assert relerr(pred, g[f"{tag}_pred"]) < 1e-10
        assert relerr(perr, g[f"{tag}_pred_err"]) < 1e-9       # sqrt of a variance that carries 1e-10
        assert emu._X_train.shape[0] == emu.nev - ntest         # retrained on the first nev - ntest events
    # the full path: own hyper-parameter search (what the reference's helper does)
    pred, perr, truth, terr = emu.testEmulatorErrors(nTestPoints=ntest)
    assert np.allclose(emu.thetas_, g["test_thetas"], atol=2e-2)
    assert relerr(pred, g["test_pred"]) < 1e-3 and relerr(perr, g["test_pred_err"]) < 1e-2
    assert np.array_equal(truth, g["test_truth"])


def test_g8_fitted_gp_estimator_surface(tmp_path):
    """what the reference reads off `self.gps[i]` (src/emulator.py:316-328): score(X, z), kernel_ (theta, bounds,
    printed form), log_marginal_likelihood_value_, y_train_"""
    g = golden("g8_holdout_pca.npz")
    ntest = int(g["ntest"])
    emu = _make_emulator(tmp_path, g, {})
    emu.testEmulatorErrorsWithTrainingPoints(nTestPoints=ntest, thetas=g["train_thetas"])
    ntrain = emu.nev - ntest
    assert maxrel(np.array([gp.y_train_ for gp in emu.gps]), g["y_train"]) < 1e-11
    scores = np.array([gp.score(emu.design_points[:ntrain], g["y_train"][i]) for i, gp in enumerate(emu.gps)])
    assert np.max(np.abs(scores - g["gp_score"])) < 1e-11
    assert np.max(np.abs(emu.gp_scores_ - g["gp_score"])) < 1e-11
    for i, gp in enumerate(emu.gps):
        k = gp.kernel_
        assert str(k) == str(g["kernel_repr"][i])
        assert np.array_equal(k.theta, g["train_thetas"][i]) and k.n_dims == g["train_thetas"].shape[1]
        assert maxrel(k.bounds, g["kernel_bounds"]) < 1e-15
        assert "{}".format(k) == str(g["kernel_repr"][i]) and "kernel=" in repr(gp)
        assert np.isfinite(gp.log_marginal_likelihood_value_)
