#!/usr/bin/env python3
"""
Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

    python tools/make_goldens.py            # needs /root/reference, scikit-learn, dill

The reference (Hendrik1704/GPBayesTools-HIC) has no tests or golden vectors of
its own, so parity is pinned by vectors captured here from the
unmodified reference code (src/emulator.py, src/mcmc.py) and from the scikit-learn
kernels it calls.  Only plain arrays (inputs + expected outputs) are written;
no reference source, bytecode or pickled reference object enters the repo.

`src/mcmc.py` imports `emcee` and `pocomc` at module level (src/mcmc.py:12,19);
neither package is installed here.  The functions captured below
(mvn_loglike, Chain.log_prior/log_likelihood/log_posterior/_predict) never call
into them, so the two module names are bound to empty placeholder modules for the
import only.  Sampler dynamics (emcee/pocoMC internals) are therefore NOT pinned.
"""
import os
import sys
import tempfile
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, "tests", "golden")
REF = "/root/reference"

sys.dont_write_bytecode = True
_work = tempfile.mkdtemp(prefix="gpb_ref_work_")
os.environ["WORKDIR"] = _work          # src/__init__.py:15-18 mkdirs cache/ at import
os.environ.setdefault("LOGLEVEL", "warning")
sys.path.insert(0, REF)

from gpbayestools_hic_amd import synth  # noqa: E402


def _import_reference():
    emcee = types.ModuleType("emcee")
    emcee.EnsembleSampler = type("EnsembleSampler", (), {})
    sys.modules.setdefault("emcee", emcee)
    sys.modules.setdefault("pocomc", types.ModuleType("pocomc"))
    from src.emulator import Emulator
    from src import mcmc
    return Emulator, mcmc


def g1_kernels():
    from sklearn.gaussian_process import kernels as sk
    rng = np.random.default_rng(101)
    N, d, W = 40, 6, 16
    X = synth.lhs(N, d, seed=102)
    Xs = rng.random((W, d))
    ls = rng.uniform(0.5, 2.5, d)
    c, noise = 1.7, 0.03
    theta = np.concatenate([[np.log(c)], np.log(ls), [np.log(noise)]])
    out = dict(X=X, Xs=Xs, theta=theta)
    for name, stat in (("rbf", sk.RBF(length_scale=ls)),
                       ("m15", sk.Matern(length_scale=ls, nu=1.5)),
                       ("m25", sk.Matern(length_scale=ls, nu=2.5))):
        k = sk.ConstantKernel(c) * stat + sk.WhiteKernel(noise)
        assert np.allclose(k.theta, theta)
        K, G = k(X, eval_gradient=True)
        out[f"{name}_K"] = K
        out[f"{name}_G"] = G
        out[f"{name}_Kcross"] = k(Xs, X)
        out[f"{name}_diag"] = k.diag(Xs)
    np.savez_compressed(os.path.join(OUT, "g1_kernels.npz"), **out)


def g2_gpr():
    from sklearn.gaussian_process import GaussianProcessRegressor as GPR
    from sklearn.gaussian_process import kernels as sk
    rng = np.random.default_rng(201)
    N, d, W = 64, 8, 24
    X = synth.lhs(N, d, seed=202)
    z = np.sin(X @ rng.standard_normal(d)) + 0.05 * rng.standard_normal(N)
    Xs = rng.random((W, d))
    thetas = []
    for i in range(3):
        ls = rng.uniform(0.4, 3.0, d)
        thetas.append(np.concatenate([[rng.uniform(-0.5, 0.8)], np.log(ls), [np.log(rng.uniform(0.02, 0.2))]]))
    thetas = np.array(thetas)
    out = dict(X=X, z=z, Xs=Xs, thetas=thetas, alpha=0.1)
    for name, mk in (("rbf", lambda ls: sk.RBF(length_scale=ls)),
                     ("m15", lambda ls: sk.Matern(length_scale=ls, nu=1.5)),
                     ("m25", lambda ls: sk.Matern(length_scale=ls, nu=2.5))):
        for i, th in enumerate(thetas):
            k = sk.ConstantKernel(np.exp(th[0])) * mk(np.exp(th[1:1 + d])) + sk.WhiteKernel(np.exp(th[-1]))
            gp = GPR(kernel=k, alpha=0.1, optimizer=None, copy_X_train=False).fit(X, z)
            lml, grad = gp.log_marginal_likelihood(th, eval_gradient=True)
            mean, cov = gp.predict(Xs, return_cov=True)
            out[f"{name}_{i}_alpha_"] = gp.alpha_
            out[f"{name}_{i}_lml"] = lml
            out[f"{name}_{i}_grad"] = grad
            out[f"{name}_{i}_mean"] = mean
            out[f"{name}_{i}_var"] = cov.diagonal().copy()
            if i == 0:
                out[f"{name}_{i}_L"] = gp.L_
    np.savez_compressed(os.path.join(OUT, "g2_gpr.npz"), **out)


VARIANTS = {
    # name: (N, d, M, npc, kernel_type, ctor kwargs)
    "pca_rbf":    (128, 8, 4, 4, "RBF", {}),
    "pca_trunc":  (128, 8, 6, 3, "RBF", {}),
    "nopca_rbf":  (96, 5, 3, 3, "RBF", dict(perform_no_PCA=True)),
    "logexp_rbf": (96, 5, 4, 2, "RBF", dict(logTrafo=True, exp_and_cov_diagonal=True)),
    "pca_matern": (96, 6, 4, 3, "Matern", {}),
}


def _make_inputs(name, N, d, M, seed):
    rng = np.random.default_rng(seed)
    lo = rng.uniform(-1.0, 0.5, d)
    hi = lo + rng.uniform(0.5, 3.0, d)
    X = synth.lhs(N, d, seed=seed + 1, lo=lo, hi=hi)
    U = (X - lo) / (hi - lo)
    Y = synth.observables(U, M, seed=seed + 2)
    Yerr = np.full_like(Y, 0.01)
    return lo, hi, X, Y, Yerr


def g3_g4_emulators(Emulator):
    trained = {}
    for vi, (name, (N, d, M, npc, ktype, kw)) in enumerate(VARIANTS.items()):
        lo, hi, X, Y, Yerr = _make_inputs(name, N, d, M, 300 + 10 * vi)
        tp = os.path.join(_work, f"{name}_train.pkl")
        pf = os.path.join(_work, f"{name}_par.txt")
        synth.write_training_pickle(tp, X, Y, Yerr)
        synth.write_parameter_file(pf, lo, hi)
        emu = Emulator(training_set_path=tp, parameter_file=pf, npc=npc, **kw)
        emu.trainEmulator([True] * emu.nev, kernel_type=ktype)
        rng = np.random.default_rng(900 + vi)
        Xs = lo + (hi - lo) * rng.random((32, d))
        extra = rng.uniform(0.0, 0.05, 32)
        per_gp = [gp.predict(Xs, return_cov=True) for gp in emu.gps]
        gp_mean = np.stack([m for m, _ in per_gp], axis=1)
        gp_var = np.stack([c.diagonal() for _, c in per_gp], axis=1)
        mean, cov = emu.predict(Xs, return_cov=True, extra_std=extra)
        mean0, cov0 = emu.predict(Xs, return_cov=True, extra_std=np.zeros(32))
        mean_only = emu.predict(Xs, return_cov=False)
        gp_cov_full = np.stack([c for _, c in per_gp], axis=0)[:, :12, :12]      # joint covariance of 12 points
        per_gp12 = [gp.predict(Xs[:12], return_cov=True)[1] for gp in emu.gps]
        samples = None
        if not kw.get("perform_no_PCA"):
            np.random.seed(4242)                                                  # neglected PCs use the global RNG
            samples = emu.sample_y(Xs[:12], n_samples=5, random_state=7)         # src/emulator.py:608-633
        out = dict(
            lo=lo, hi=hi, X=X, Y=Y, Yerr=Yerr, npc=npc, kernel_type=ktype,
            model_data=emu.model_data,
            thetas=np.array([gp.kernel_.theta for gp in emu.gps]),
            theta_bounds=emu.gps[0].kernel_.bounds,
            lml=np.array([gp.log_marginal_likelihood_value_ for gp in emu.gps]),
            alpha_=np.array([gp.alpha_ for gp in emu.gps]),
            Ldiag=np.array([gp.L_.diagonal() for gp in emu.gps]),
            Lrow_last=np.array([gp.L_[-1] for gp in emu.gps]),
            scaler_mean=emu.scaler.mean_, scaler_scale=emu.scaler.scale_, scaler_var=emu.scaler.var_,
            Xs=Xs, extra_std=extra, gp_mean=gp_mean, gp_var=gp_var,
            mean=mean, cov=cov, mean0=mean0, cov0=cov0, mean_only=mean_only,
            gp_cov12=np.stack(per_gp12, axis=0),
        )
        if samples is not None:
            out["sample_y"] = samples
        if not kw.get("perform_no_PCA"):
            out.update(pca_components=emu.pca.components_,
                       pca_explained_variance=emu.pca.explained_variance_,
                       pca_mean=emu.pca.mean_,
                       trans_matrix=emu._trans_matrix, var_trans=emu._var_trans,
                       cov_trunc=emu._cov_trunc)
        np.savez_compressed(os.path.join(OUT, f"g3_emulator_{name}.npz"), **out)
        trained[name] = (emu, lo, hi)
    return trained


def g5_chain(mcmc, Emulator):
    """Two emulators that share one parameter space (E=2: block-diagonal covariance,
    src/mcmc.py:153-166)."""
    N, d = 128, 8
    rng = np.random.default_rng(500)
    lo = rng.uniform(-1.0, 0.5, d)
    hi = lo + rng.uniform(0.5, 3.0, d)
    X = synth.lhs(N, d, seed=501, lo=lo, hi=hi)
    U = (X - lo) / (hi - lo)
    specs = [("A", 4, 4, 502), ("B", 6, 3, 503)]
    emus, out = [], dict(lo=lo, hi=hi, X=X)
    pf = os.path.join(_work, "chain_par.txt")
    synth.write_parameter_file(pf, lo, hi)
    for tag, M, npc, seed in specs:
        Y = synth.observables(U, M, seed=seed)
        tp = os.path.join(_work, f"chain_{tag}.pkl")
        synth.write_training_pickle(tp, X, Y, np.full_like(Y, 0.01))
        emu = Emulator(training_set_path=tp, parameter_file=pf, npc=npc)
        emu.trainEmulatorAutoMask()
        emus.append(emu)
        out[f"Y_{tag}"] = Y
        out[f"npc_{tag}"] = npc
        out[f"thetas_{tag}"] = np.array([gp.kernel_.theta for gp in emu.gps])
    xstar = lo + (hi - lo) * synth.truth_point(d, seed=504)
    yexp = np.concatenate([e.predict(xstar[None, :], return_cov=False)[0] for e in emus])
    yerr = 0.05 * np.abs(yexp)
    ep = os.path.join(_work, "chain_exp.pkl")
    synth.write_experiment_pickle(ep, yexp, yerr)
    chain = mcmc.Chain(mcmc_path=os.path.join(_work, "mcmc", "chain.pkl"),
                       expdata_path=ep, model_parafile=pf)
    chain.emuList = emus
    W = 64
    Xw = lo + (hi - lo) * rng.random((W, d))
    Xw[3, 2] = hi[2] + 0.1            # outside
    Xw[10, 0] = lo[0] - 1e-3          # outside
    Xw[17, 5] = lo[5]                 # exactly on the lower bound -> outside (strict)
    Xw[23, 7] = hi[7]                 # exactly on the upper bound -> outside (strict)
    Xw[40] = lo + (hi - lo) * 1e-9    # barely inside
    Xout = hi + 0.5 + rng.random((5, d))
    inside = np.all((Xw > lo) & (Xw < hi), axis=1)
    pm, pc = chain._predict(Xw[inside], extra_std=0.0)
    out.update(
        xstar=xstar, yexp=yexp, yerr=yerr, Xw=Xw, Xout=Xout, inside=inside,
        expdata=chain.expdata, expdata_cov=chain.expdata_cov,
        log_prior=chain.log_prior(Xw),
        log_likelihood=chain.log_likelihood(Xw),
        log_likelihood_finite=chain.log_likelihood(Xw, finite=True),
        log_posterior=chain.log_posterior(Xw),
        log_posterior_out=chain.log_posterior(Xout),
        log_likelihood_out_finite=chain.log_likelihood(Xout, finite=True),
        log_posterior_1d=chain.log_posterior(Xw[0]),
        predict_mean=pm, predict_cov=pc,
    )
    np.savez_compressed(os.path.join(OUT, "g5_chain.npz"), **out)


def g7_param_pca(Emulator):
    """parameterTrafoPCA=True (src/emulator.py:79-241, 492-551): 20 model parameters, three groups
    replaced by principal components of zeta/s(T), eta/s(mu_B), y_loss(y_init)."""
    N, d, M, npc = 96, 20, 4, 3
    rng = np.random.default_rng(700)
    lo = np.full(d, 0.1); hi = np.full(d, 1.0)
    lo[[2, 3, 4]], hi[[2, 3, 4]] = 0.0, 2.0                   # yloss_2,4,6
    lo[[12, 13, 14]], hi[[12, 13, 14]] = 0.01, 0.3            # eta_0,2,4
    lo[15], hi[15] = 0.0, 0.2                                  # zeta_max
    lo[16], hi[16] = 0.13, 0.3                                 # T_zeta0
    lo[[17, 18]], hi[[17, 18]] = 0.01, 0.15                    # sigma_plus, sigma_minus
    X = synth.lhs(N, d, seed=701, lo=lo, hi=hi)
    U = (X - lo) / (hi - lo)
    Y = synth.observables(U, M, seed=702)
    tp, pf = os.path.join(_work, "ppca_train.pkl"), os.path.join(_work, "ppca_par.txt")
    synth.write_training_pickle(tp, X, Y, np.full_like(Y, 0.01))
    synth.write_parameter_file(pf, lo, hi)
    emu = Emulator(training_set_path=tp, parameter_file=pf, npc=npc, parameterTrafoPCA=True)
    emu.trainEmulatorAutoMask()
    Xs = lo + (hi - lo) * rng.random((16, d))
    mean, cov = emu.predict(Xs, return_cov=True, extra_std=np.zeros(16))
    np.savez_compressed(
        os.path.join(OUT, "g7_param_pca.npz"), lo=lo, hi=hi, X=X, Y=Y, npc=npc,
        new_design_points=emu.PCA_new_design_points, design_min=emu.design_min, design_max=emu.design_max,
        n_components=np.array([emu.paramTrafoPCA_bulk.n_components_, emu.paramTrafoPCA_shear.n_components_,
                               emu.paramTrafoPCA_yloss.n_components_]),
        thetas=np.array([gp.kernel_.theta for gp in emu.gps]),
        lml=np.array([gp.log_marginal_likelihood_value_ for gp in emu.gps]),
        Xs=Xs, mean=mean, cov=cov)


def _holdout_by_the_reference(emu, ntest, on_training):
    """The two hold-out helpers (src/emulator.py:636-679, 682-726) call `self.predict(X, return_cov=True)` with the
    scalar default `extra_std=0`, which raises under numpy >= 2 (`np.array(0, copy=False)`, src/emulator.py:578;
    SURVEY §8 a6) — so the helpers themselves cannot run in this container.  Their steps are carried out here with
    the reference's own `trainEmulator` and `predict` (array-valued extra_std = 0, the same numbers): mask the last
    ntest events (:651-655), retrain, predict the validation events (:657-659), sqrt of the covariance diagonal,
    undo the log transform (:661-673), reshape to [-1, nobs] (:675-678)."""
    mask = [True] * emu.nev
    for i in range(emu.nev - ntest, emu.nev):
        mask[i] = False
    emu.trainEmulator(mask)
    vmask = np.array(mask if on_training else [not m for m in mask])
    Xv = emu.design_points_org_[vmask, :]
    pred, cov = emu.predict(Xv, return_cov=True, extra_std=np.zeros(Xv.shape[0]))
    pred_err = np.sqrt(np.array([cov[i].diagonal() for i in range(cov.shape[0])]))
    if emu.logTrafo_ and not emu.exp_and_cov_diagonal_:
        pred, pred_err = np.exp(pred), pred_err * np.exp(pred)
    if emu.logTrafo_:
        truth = np.exp(emu.model_data[vmask, :])
        truth_err = emu.model_data_err[vmask, :] * np.exp(emu.model_data[vmask, :])
    else:
        truth, truth_err = emu.model_data[vmask, :], emu.model_data_err[vmask, :]
    r = lambda a: np.array(a).reshape(-1, emu.nobs)
    return r(pred), r(pred_err), r(truth), r(truth_err)


def g8_holdout(Emulator):
    """Hold-out validation (src/emulator.py:636-726) and the sklearn estimator surface the reference reads off its
    fitted GPs (`gp.score`, `gp.kernel_`: src/emulator.py:316-328).  The hyper-parameters the reference's L-BFGS-B
    run ends on are stored so that the build can be checked at those values (tight) and after its own search
    (optimiser tolerance)."""
    for vi, (name, kw) in enumerate((("pca", {}), ("log", dict(logTrafo=True)))):
        N, d, M, npc, ntest = 64, 5, 4, 3, 6
        lo, hi, X, Y, Yerr = _make_inputs(name, N, d, M, 800 + 10 * vi)
        Y = np.abs(Y) + 0.5                                   # positive observables (logTrafo takes logs)
        tp = os.path.join(_work, f"holdout_{name}_train.pkl")
        pf = os.path.join(_work, f"holdout_{name}_par.txt")
        synth.write_training_pickle(tp, X, Y, Yerr)
        synth.write_parameter_file(pf, lo, hi)
        out = dict(lo=lo, hi=hi, X=X, Y=Y, Yerr=Yerr, npc=npc, ntest=ntest)
        emu = Emulator(training_set_path=tp, parameter_file=pf, npc=npc, **kw)
        try:                                                  # the record: the helper as shipped does not run here
            emu.testEmulatorErrors(nTestPoints=ntest)
            out["helper_runs_under_this_numpy"] = True
        except ValueError:
            out["helper_runs_under_this_numpy"] = False
        for tag, on_training in (("test", False), ("train", True)):
            pred, pred_err, truth, truth_err = _holdout_by_the_reference(emu, ntest, on_training)
            out[f"{tag}_pred"], out[f"{tag}_pred_err"] = pred, pred_err
            out[f"{tag}_truth"], out[f"{tag}_truth_err"] = truth, truth_err
            out[f"{tag}_thetas"] = np.array([gp.kernel_.theta for gp in emu.gps])
        # estimator surface after the last retraining (N - ntest points)
        Xt = emu.design_points[:N - ntest]
        Z = emu.pca.transform(emu.scaler.transform(emu.model_data[:N - ntest]))[:, :npc]
        out["gp_score"] = np.array([gp.score(Xt, Z[:, i]) for i, gp in enumerate(emu.gps)])
        out["kernel_repr"] = np.array([str(gp.kernel_) for gp in emu.gps])
        out["kernel_bounds"] = emu.gps[0].kernel_.bounds
        out["y_train"] = np.array([gp.y_train_ for gp in emu.gps])
        np.savez_compressed(os.path.join(OUT, f"g8_holdout_{name}.npz"), **out)


def g9_loading(Emulator):
    """a1 and the two host helpers: `_load_training_data_pickle` on a data set whose keys are STRINGS in shuffled order, with
    events over the relative-error threshold (discarded), a NaN error and a negative value; with and without the log transform
    and with a custom threshold (src/emulator.py:378-415); `getAvgTrainingDataRelError` (:418-421) and `outputPCAvsParam`
    (:244-249) on the loaded data."""
    import pickle
    rng = np.random.default_rng(77)
    N, d, M = 40, 3, 5
    lo, hi = np.zeros(d), np.ones(d)
    X = synth.lhs(N, d, seed=78)
    Y = 2.0 + synth.observables(X, M, seed=79)
    Yerr = 0.02 * np.abs(Y) * rng.uniform(0.2, 1.0, Y.shape)
    Yerr[3, 1] = 0.5 * abs(Y[3, 1])            # 50 % error: discarded at every threshold used here
    Yerr[11, 4] = 0.12 * abs(Y[11, 4])         # discarded at 0.1, kept at 0.2
    Yerr[17, 0] = 0.099 * abs(Y[17, 0])        # just under 0.1
    Y[22, 2] = -Y[22, 2]                       # a negative value: |err / val| and log(|val|)
    Yerr[30, 3] = np.nan                       # NaN error: the comparison is False (kept), nan_to_num -> 0
    data = {}
    for i in rng.permutation(N):               # insertion order shuffled; the loader sorts by int(key)
        data[str(int(i))] = {"parameter": X[i], "obs": np.array([Y[i], Yerr[i]])}
    tp = os.path.join(_work, "g9_train.pkl")
    pf = os.path.join(_work, "g9_par.txt")
    with open(tp, "wb") as f:
        pickle.dump(data, f)
    synth.write_parameter_file(pf, lo, hi)
    out = dict(X=X, Y=Y, Yerr=Yerr, order=np.array([int(k) for k in data.keys()]))
    for tag, kw in (("plain", {}), ("thr02", dict(max_rel_uncertainty_data=0.2)),
                    ("log", dict(logTrafo=True)), ("logthr", dict(logTrafo=True, max_rel_uncertainty_data=0.2))):
        with np.errstate(all="ignore"):
            emu = Emulator(training_set_path=tp, parameter_file=pf, npc=3, **kw)
            # (copies: the reference's StandardScaler(copy=False) makes outputPCAvsParam standardise model_data IN PLACE,
            # src/emulator.py:76,244-249 — a side effect nothing in the reference relies on and the drop-in does not copy)
            out[f"{tag}_design_points"] = emu.design_points.copy()
            out[f"{tag}_model_data"] = emu.model_data.copy()
            out[f"{tag}_model_data_err"] = emu.model_data_err.copy()
            out[f"{tag}_nev"] = emu.nev
            out[f"{tag}_avg_rel_err"] = emu.getAvgTrainingDataRelError()
            dp, Zt = emu.outputPCAvsParam()
            out[f"{tag}_pca_design"] = dp
            out[f"{tag}_pca_Zt"] = Zt
    np.savez_compressed(os.path.join(OUT, "g9_loading.npz"), **out)


def g10_learning_curve(Emulator):
    """`Emulator.print_learning_curve` (src/emulator.py:424-462): sklearn's learning_curve (5 unshuffled folds, train sizes 0.2 .. 0.9
    of a fold's training set) over GPR(1. * RBF(ptp, ptp x (.01, 100)) + White(1e-4, (1e-6, 1)), alpha = 0) fits of every principal
    component — 25 hyper-parameter searches per GP.  Stored: the inputs and the returned [train size, mean train R^2, mean test R^2]
    tables, plus the scaler / PCA state the call leaves behind (it refits both on ALL events)."""
    N, d, M, npc = 60, 3, 5, 2
    lo, hi, X, Y, Yerr = _make_inputs("learning_curve", N, d, M, 1000)
    tp = os.path.join(_work, "lc_train.pkl")
    pf = os.path.join(_work, "lc_par.txt")
    synth.write_training_pickle(tp, X, Y, Yerr)
    synth.write_parameter_file(pf, lo, hi)
    emu = Emulator(training_set_path=tp, parameter_file=pf, npc=npc)
    status = emu.print_learning_curve()
    np.savez_compressed(os.path.join(OUT, "g10_learning_curve.npz"), lo=lo, hi=hi, X=X, Y=Y, Yerr=Yerr, npc=npc,
                        status=np.array(status), scaler_mean=emu.scaler.mean_, pca_components=emu.pca.components_[:npc])


def _dump_trained(emu, prefix, out):
    """the attributes of a TRAINED reference emulator that Emulator.from_reference reads, as plain arrays"""
    out[prefix + "flags"] = np.array([emu.logTrafo_, emu.parameterTrafoPCA_, emu.exp_and_cov_diagonal_, emu.perform_no_PCA_], dtype=np.int64)
    out[prefix + "npc"] = emu.npc
    for name in ("design_points", "model_data", "model_data_err", "design_min", "design_max"):
        out[prefix + name] = np.array(getattr(emu, name), dtype=np.float64)
    for name in ("mean_", "scale_", "var_"):
        out[prefix + "scaler_" + name] = getattr(emu.scaler, name)
    if not emu.perform_no_PCA_:
        for name in ("mean_", "components_", "explained_variance_", "explained_variance_ratio_"):
            out[prefix + "pca_" + name] = getattr(emu.pca, name)
        out[prefix + "pca_n_components_"] = emu.pca.n_components_
    g0 = emu.gps[0]
    out[prefix + "gp_family"] = np.array([type(g0.kernel_.k1.k1).__name__, type(g0.kernel_.k1.k2).__name__, type(g0.kernel_.k2).__name__])
    out[prefix + "gp_nu"] = float(getattr(g0.kernel_.k1.k2, "nu", 0.0))
    out[prefix + "gp_alpha"] = float(g0.alpha)
    out[prefix + "gp_X_train"] = np.array(g0.X_train_)
    out[prefix + "gp_y_train"] = np.array([gp.y_train_ for gp in emu.gps])
    out[prefix + "gp_theta"] = np.array([gp.kernel_.theta for gp in emu.gps])
    out[prefix + "gp_lml"] = np.array([gp.log_marginal_likelihood_value_ for gp in emu.gps])
    if emu.parameterTrafoPCA_:
        out[prefix + "PCA_new_design_points"] = emu.PCA_new_design_points
        for tag in ("bulk", "shear", "yloss"):
            sc, pc = getattr(emu, "paramTrafoScaler_" + tag), getattr(emu, "paramTrafoPCA_" + tag)
            for name in ("mean_", "scale_", "var_"):
                out[prefix + tag + "_scaler_" + name] = getattr(sc, name)
            for name in ("mean_", "components_", "explained_variance_", "explained_variance_ratio_"):
                out[prefix + tag + "_pca_" + name] = getattr(pc, name)
            out[prefix + tag + "_pca_n_components_"] = pc.n_components_


def g11_trained_objects(Emulator):
    """Trained emulator OBJECTS of the reference as its pickles hold them (src/mcmc.py:145-150 loads them): the attributes
    `Emulator.from_reference` reads — fitted scaler / PCA, flags, the sklearn GPs' X_train_ / y_train_ / kernel_.theta / alpha, the
    parameter maps — and the object's own predictions, for: PCA + RBF trained on a MASKED event set, log transform +
    exp_and_cov_diagonal + Matern-3/2, perform_no_PCA, and parameterTrafoPCA."""
    out = {}
    cases = (("mask", 64, 6, 5, 3, "RBF", {}), ("logexp", 56, 5, 4, 2, "Matern", dict(logTrafo=True, exp_and_cov_diagonal=True)),
             ("nopca", 48, 4, 3, 3, "RBF", dict(perform_no_PCA=True)))
    for ci, (name, N, d, M, npc, ktype, kw) in enumerate(cases):
        lo, hi, X, Y, Yerr = _make_inputs("trained_" + name, N, d, M, 1100 + 10 * ci)
        if kw.get("logTrafo"):
            Y = np.abs(Y) + 0.5
        tp, pf = os.path.join(_work, f"tr_{name}_train.pkl"), os.path.join(_work, f"tr_{name}_par.txt")
        synth.write_training_pickle(tp, X, Y, Yerr)
        synth.write_parameter_file(pf, lo, hi)
        emu = Emulator(training_set_path=tp, parameter_file=pf, npc=npc, **kw)
        mask = np.ones(emu.nev, dtype=bool)
        if name == "mask":
            mask[[3, 17, 18, 40, 63]] = False
        emu.trainEmulator(mask, kernel_type=ktype)
        rng = np.random.default_rng(1150 + ci)
        Xs = lo + (hi - lo) * rng.random((12, d))
        es = np.linspace(0.0, 0.2, 12)
        mean, cov = emu.predict(Xs, return_cov=True, extra_std=es)
        out[name + "_lo"], out[name + "_hi"], out[name + "_Xs"], out[name + "_es"] = lo, hi, Xs, es
        out[name + "_mean"], out[name + "_cov"] = mean, cov
        _dump_trained(emu, name + "_", out)
    # parameterTrafoPCA (the inputs of g7)
    g7 = np.load(os.path.join(OUT, "g7_param_pca.npz"))
    tp, pf = os.path.join(_work, "tr_ppca_train.pkl"), os.path.join(_work, "tr_ppca_par.txt")
    synth.write_training_pickle(tp, g7["X"], g7["Y"], np.full_like(g7["Y"], 0.01))
    synth.write_parameter_file(pf, g7["lo"], g7["hi"])
    emu = Emulator(training_set_path=tp, parameter_file=pf, npc=int(g7["npc"]), parameterTrafoPCA=True)
    emu.trainEmulatorAutoMask()
    mean, cov = emu.predict(g7["Xs"], return_cov=True, extra_std=np.zeros(len(g7["Xs"])))
    out["ppca_lo"], out["ppca_hi"], out["ppca_Xs"], out["ppca_es"] = g7["lo"], g7["hi"], g7["Xs"], np.zeros(len(g7["Xs"]))
    out["ppca_mean"], out["ppca_cov"] = mean, cov
    _dump_trained(emu, "ppca_", out)
    np.savez_compressed(os.path.join(OUT, "g11_trained_objects.npz"), **out)


def g6_mvn(mcmc):
    rng = np.random.default_rng(600)
    out = {}
    for M in (4, 16, 64):
        ys, covs, lls = [], [], []
        for _ in range(6):
            B = rng.standard_normal((M, M))
            cov = B @ B.T / M + np.diag(rng.uniform(0.01, 0.5, M))
            y = rng.standard_normal(M)
            ys.append(y); covs.append(cov)
            lls.append(mcmc.mvn_loglike(y, cov))
        out[f"y_{M}"] = np.array(ys); out[f"cov_{M}"] = np.array(covs); out[f"ll_{M}"] = np.array(lls)
    np.savez_compressed(os.path.join(OUT, "g6_mvn.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    Emulator, mcmc = _import_reference()
    g1_kernels()
    g2_gpr()
    g3_g4_emulators(Emulator)
    g5_chain(mcmc, Emulator)
    g6_mvn(mcmc)
    g7_param_pca(Emulator)
    g8_holdout(Emulator)
    g9_loading(Emulator)
    g10_learning_curve(Emulator)
    g11_trained_objects(Emulator)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
