"""
CPU tier: the oracle (oracle/gp_oracle.py) against vectors captured from the
reference itself (tools/make_goldens.py -> tests/golden/*.npz).
Tolerances follow SURVEY.md §8c: kernels 1e-14 (matrix-relative), L/alpha 1e-11,
mean 1e-12, var/cov 1e-10, log-posterior 1e-10 relative.
"""
import numpy as np
import pytest

from conftest import golden, relerr, maxrel
from oracle import gp_oracle as O

KINDS = {"rbf": O.KIND_RBF, "m15": O.KIND_MATERN15, "m25": O.KIND_MATERN25}
VARIANTS = {
    "pca_rbf": (O.KIND_RBF, O.MODE_PCA),
    "pca_trunc": (O.KIND_RBF, O.MODE_PCA),
    "nopca_rbf": (O.KIND_RBF, O.MODE_NO_PCA),
    "logexp_rbf": (O.KIND_RBF, O.MODE_EXPDIAG),
    "pca_matern": (O.KIND_MATERN15, O.MODE_PCA),
}


@pytest.mark.parametrize("name", list(KINDS))
def test_g1_kernels(name):
    g = golden("g1_kernels.npz")
    X, Xs, th = g["X"], g["Xs"], g["theta"]
    kind = KINDS[name]
    assert maxrel(O.kernel_train(X, th, kind), g[f"{name}_K"]) < 1e-14
    assert maxrel(O.kernel_cross(Xs, X, th, kind), g[f"{name}_Kcross"]) < 1e-14
    assert maxrel(O.kernel_train_grad(X, th, kind), g[f"{name}_G"]) < 1e-13
    assert np.allclose(O.prior_var(th, X.shape[1]), g[f"{name}_diag"], rtol=1e-15)


@pytest.mark.parametrize("name", list(KINDS))
@pytest.mark.parametrize("i", [0, 1, 2])
def test_g2_gpr(name, i):
    g = golden("g2_gpr.npz")
    X, z, Xs, th = g["X"], g["z"], g["Xs"], g["thetas"][i]
    kind = KINDS[name]
    L, a = O.gp_factor(X, z, th, kind, float(g["alpha"]))
    assert maxrel(a, g[f"{name}_{i}_alpha_"]) < 1e-11
    if i == 0:
        assert maxrel(L, g[f"{name}_{i}_L"]) < 1e-11
    v, grad = O.lml(th, X, z, kind, float(g["alpha"]), eval_gradient=True)
    assert abs(v - g[f"{name}_{i}_lml"]) < 1e-10 * abs(g[f"{name}_{i}_lml"])
    assert maxrel(grad, g[f"{name}_{i}_grad"]) < 1e-10
    m, var = O.gp_predict(Xs, X, th, L, a, kind)
    assert maxrel(m, g[f"{name}_{i}_mean"]) < 1e-12
    assert relerr(var, g[f"{name}_{i}_var"]) < 1e-10
    m2, var2 = O.gp_predict_faithful(Xs, X, th, L, a, kind)
    assert relerr(var2, g[f"{name}_{i}_var"]) < 1e-10


def _oracle_emulator(g, name):
    kind, mode = VARIANTS[name]
    e = O.OracleEmulator(g["X"], g["model_data"], g["lo"], g["hi"], int(g["npc"]), kind, mode)
    return e


@pytest.mark.parametrize("name", list(VARIANTS))
def test_g3_emulator_preprocessing_and_predict(name):
    g = golden(f"g3_emulator_{name}.npz")
    e = _oracle_emulator(g, name).fit(g["thetas"])
    assert maxrel(e.mu, g["scaler_mean"]) < 1e-14
    assert maxrel(e.scale, g["scaler_scale"]) < 1e-14
    if "trans_matrix" in g.files:
        assert maxrel(e.trans_matrix, g["trans_matrix"]) < 1e-11
        assert maxrel(e.cov_trunc, g["cov_trunc"]) < 1e-11
    assert maxrel(np.array(e.a), g["alpha_"]) < 1e-10
    assert maxrel(np.array([np.diag(L) for L in e.L]), g["Ldiag"]) < 1e-11
    assert relerr(np.array(e.lml_), g["lml"]) < 1e-10
    m, v = e.gp_predict(g["Xs"])
    assert maxrel(m, g["gp_mean"]) < 1e-11
    assert relerr(v, g["gp_var"]) < 1e-10
    for p in range(e.npc):        # joint covariance of the first 12 query points
        _, cfull = O.gp_predict_cov(g["Xs"][:12], e.X, e.thetas[p], e.L[p], e.a[p], e.kind)
        assert maxrel(cfull, g["gp_cov12"][p]) < 1e-10
    mean, cov = e.predict(g["Xs"], True, g["extra_std"])
    assert relerr(mean, g["mean"]) < 1e-11
    assert maxrel(cov, g["cov"]) < 1e-10
    mean0, cov0 = e.predict(g["Xs"], True, np.zeros(len(g["Xs"])))
    assert maxrel(cov0, g["cov0"]) < 1e-10
    assert relerr(e.predict(g["Xs"], False), g["mean_only"]) < 1e-11


@pytest.mark.parametrize("name", ["pca_rbf", "pca_matern"])
def test_g3_hyperparameter_search(name):
    """tier (ii) of SURVEY §7: theta* to optimiser tolerance, LML* to 1e-8 relative."""
    g = golden(f"g3_emulator_{name}.npz")
    kind, mode = VARIANTS[name]
    e = _oracle_emulator(g, name)
    th0, bnds = O.default_theta0_bounds(g["lo"], g["hi"], kind)
    assert np.allclose(bnds, g["theta_bounds"], rtol=1e-14, atol=1e-14)
    th, val = O.gp_fit_theta(e.X, np.ascontiguousarray(e.Z[:, 0]), th0, bnds, kind)
    assert abs(val - g["lml"][0]) < 1e-8 * abs(g["lml"][0])
    assert np.allclose(th, g["thetas"][0], atol=2e-3)


def _chain_emus(g):
    emus = []
    for tag in ("A", "B"):
        e = O.OracleEmulator(g["X"], g[f"Y_{tag}"], g["lo"], g["hi"], int(g[f"npc_{tag}"]))
        emus.append(e.fit(g[f"thetas_{tag}"]))
    return emus


@pytest.mark.parametrize("batched", [True, False])
def test_g5_chain(batched):
    g = golden("g5_chain.npz")
    emus = _chain_emus(g)
    lo, hi, Xw = g["lo"], g["hi"], g["Xw"]
    pf = lambda X, es: O.chain_predict(emus, X, es)
    assert np.array_equal(O.inside_box(Xw, lo, hi), g["inside"])
    assert np.array_equal(O.log_prior(Xw, lo, hi), g["log_prior"])
    pm, pc = pf(Xw[g["inside"]], 0.0 * Xw[g["inside"], -1])
    assert relerr(pm, g["predict_mean"]) < 1e-11
    assert maxrel(pc, g["predict_cov"]) < 1e-10
    kw = dict(yexp=g["expdata"], cov_exp=g["expdata_cov"], batched=batched)
    post = O.log_prob(Xw, lo, hi, pf, **kw)
    ins = g["inside"]
    assert np.all(np.isneginf(post[~ins]))
    assert relerr(post[ins], g["log_posterior"][ins]) < 1e-10
    like = O.log_prob(Xw, lo, hi, pf, posterior=False, **kw)
    assert relerr(like[ins], g["log_likelihood"][ins]) < 1e-10
    likef = O.log_prob(Xw, lo, hi, pf, posterior=False, finite=True, **kw)
    assert np.all(likef[~ins] == -1e300)
    assert relerr(likef[ins], g["log_likelihood_finite"][ins]) < 1e-10
    out = O.log_prob(g["Xout"], lo, hi, pf, **kw)
    assert np.array_equal(out, g["log_posterior_out"])
    one = O.log_prob(Xw[0], lo, hi, pf, **kw)
    assert relerr(one, g["log_posterior_1d"]) < 1e-10


@pytest.mark.parametrize("M", [4, 16, 64])
def test_g6_mvn(M):
    g = golden("g6_mvn.npz")
    y, cov, ll = g[f"y_{M}"], g[f"cov_{M}"], g[f"ll_{M}"]
    got = np.array([O.mvn_loglike(a, c) for a, c in zip(y, cov)])
    assert relerr(got, ll) < 1e-13
    assert relerr(O.mvn_loglike_batched(y, cov), ll) < 1e-11
    bad = -np.eye(M)
    assert np.isnan(O.mvn_loglike(y[0], bad))


@pytest.mark.parametrize("name", ["pca", "log"])
def test_g8_holdout_split(name):
    """hold-out validation (src/emulator.py:636-726): the oracle trained on the first nev - ntest events at the
    reference's hyper-parameters reproduces the reference's predictions for the held-out and the training events"""
    g = golden(f"g8_holdout_{name}.npz")
    ntest, npc = int(g["ntest"]), int(g["npc"])
    log = name == "log"
    Y = np.log(np.abs(g["Y"]) + 1e-30) if log else g["Y"]           # src/emulator.py:403-407
    ntr = Y.shape[0] - ntest
    for tag, rows in (("test", slice(ntr, None)), ("train", slice(0, ntr))):
        oe = O.OracleEmulator(g["X"][:ntr], Y[:ntr], g["lo"], g["hi"], npc).fit(g[f"{tag}_thetas"])
        mean, cov = oe.predict(g["X"][rows], True, 0.0)
        err = np.sqrt(np.diagonal(cov, axis1=1, axis2=2))
        if log:
            mean, err = np.exp(mean), err * np.exp(mean)
        assert relerr(mean, g[f"{tag}_pred"]) < 1e-10
        assert relerr(err, g[f"{tag}_pred_err"]) < 1e-9
        truth = np.exp(Y[rows]) if log else Y[rows]
        assert maxrel(truth, g[f"{tag}_truth"]) < 1e-15
    # estimator surface: R^2 of the predictive mean on the training targets (sklearn RegressorMixin.score)
    oe = O.OracleEmulator(g["X"][:ntr], Y[:ntr], g["lo"], g["hi"], npc).fit(g["train_thetas"])
    assert maxrel(oe.Z.T, g["y_train"]) < 1e-11
    m, _ = oe.gp_predict(g["X"][:ntr])
    z = g["y_train"].T
    r2 = 1.0 - ((z - m) ** 2).sum(0) / ((z - z.mean(0)) ** 2).sum(0)
    assert np.max(np.abs(r2 - g["gp_score"])) < 1e-11


def test_g10_learning_curve():
    """`Emulator.print_learning_curve` (src/emulator.py:424-462): the oracle's restatement of sklearn's learning_curve over the
    reference's GPR fits against the reference's own table.  The scores hang on where 50 L-BFGS-B searches end (the same scipy
    routine on the same objective): the bar is the optimiser's tolerance carried into R^2."""
    g = golden("g10_learning_curve.npz")
    mean, scale, _ = O.standardize_fit(g["Y"])
    Z, _, _, _ = O.pca_whiten_fit((g["Y"] - mean) / scale)
    got = O.learning_curve(g["X"], Z[:, :int(g["npc"])], g["lo"], g["hi"])
    ref = g["status"]
    assert got.shape == ref.shape and np.array_equal(got[:, :, 0], ref[:, :, 0])
    assert np.max(np.abs(got[:, :, 1:] - ref[:, :, 1:])) < 1e-5
