"""
Drop-in replacement for the reference's `src/emulator.py` `Emulator` (B1 protocol):
same constructor, `trainEmulatorAutoMask()`, `trainEmulator(eventMask, kernel_type)`,
`predict(X, return_cov, extra_std)` and attribute names, but every GP operation — kernel
matrices, Cholesky, alpha, log-marginal likelihood and gradient, predictive mean/variance,
the PC->observable covariance — runs in the HIP engine (GPEngine / include/gpbayes.h).
Host Python keeps file parsing, the N x M scaler/PCA SVD and the L-BFGS-B driver
(scipy.optimize, as sklearn uses: sk:_gpr.py:654-670).

Objects pickle/dill cleanly: device handles are dropped in __getstate__ and the factorisation
is rebuilt on the device from (X, Z, theta) at first use after loading (src/mcmc.py:145-150).
"""
import itertools
import logging
import os
import pickle
import threading

import numpy as np
import scipy.optimize

from .engine import (GPEngine, MODE_EXPDIAG, MODE_NO_PCA, MODE_NO_PCA_EXPDIAG, MODE_PCA,
                     NotPositiveDefinite)
from .preprocess import (Standardizer, WhitenedPCA, observable_transform,
                         parse_model_parameter_file)

log = logging.getLogger(__name__)

_STATE_SERIAL = itertools.count(1)     # one number per fitted state of any emulator of this process (Chain.state_digest_cached)

_KERNELS = {"RBF": ("RBF", (1e-1, 1e2)), "Matern": ("Matern15", (1e-3, 1e5)),
            "Matern25": ("Matern25", (1e-3, 1e5))}


def _check_random_state(seed):
    """sklearn.utils.check_random_state without importing sklearn."""
    if seed is None or seed is np.random:
        return np.random.mtrand._rand
    if isinstance(seed, (int, np.integer)):
        return np.random.RandomState(seed)
    if isinstance(seed, np.random.RandomState):
        return seed
    raise ValueError("%r cannot be used to seed a numpy.random.RandomState instance" % seed)


class FittedKernel:
    """`gp.kernel_` of a fitted GP: the sklearn composite `c * RBF|Matern(length_scale) + WhiteKernel(noise_level)`
    the reference builds (src/emulator.py:286-306), as a read-only view with the attributes and the printed form
    the reference uses (`theta`, `bounds`, `'... kernel: {}'.format(gp.kernel_)`, src/emulator.py:320-328;
    sk:kernels.py:887-888, 987-988, 1316-1317, 1437-1440, 1584-1593, 1783-1793)."""

    def __init__(self, theta, bounds, kernel_type):
        self.theta = np.array(theta, dtype=np.float64)          # [log c, log l_1..l_d, log noise]
        self.bounds = np.array(bounds, dtype=np.float64)        # log-space, like sklearn's kernel_.bounds
        self.kernel_type = kernel_type
        self.nu = {"RBF": None, "Matern": 1.5, "Matern25": 2.5}[kernel_type]

    @property
    def n_dims(self):
        return self.theta.shape[0]

    @property
    def constant_value(self):
        return float(np.exp(self.theta[0]))

    @property
    def length_scale(self):
        return np.exp(self.theta[1:-1])

    @property
    def noise_level(self):
        return float(np.exp(self.theta[-1]))

    def __repr__(self):
        ls = self.length_scale
        inner = "[{}]".format(", ".join(map("{0:.3g}".format, ls))) if ls.shape[0] > 1 else "{0:.3g}".format(ls[0])
        stat = "RBF(length_scale={})".format(inner) if self.nu is None else \
            "Matern(length_scale={}, nu={:.3g})".format(inner, self.nu)
        return "{0:.3g}**2 * {1} + WhiteKernel(noise_level={2:.3g})".format(
            np.sqrt(self.constant_value), stat, self.noise_level)


class FittedGP:
    """Read-only view of one fitted GP with the sklearn estimator surface the reference reads off `self.gps[i]`
    (src/emulator.py:309-328,553,621): kernel_, log_marginal_likelihood_value_, alpha_, L_,
    X_train_, y_train_, predict(), score(), sample_y()."""

    def __init__(self, emu, index):
        self._emu, self._i = emu, index

    @property
    def kernel_(self):
        emu = self._emu
        return FittedKernel(emu.thetas_[self._i], emu._theta0_bounds(emu.kernel_type_)[1], emu.kernel_type_)

    @property
    def kernel_theta(self):
        return self._emu.thetas_[self._i]

    @property
    def log_marginal_likelihood_value_(self):
        return float(self._emu.lml_[self._i])

    @property
    def X_train_(self):
        return self._emu._X_train

    @property
    def y_train_(self):
        return self._emu._Z_train[self._i]

    @property
    def alpha_(self):
        return self._emu._engine_ready().get("alpha")[self._i]

    @property
    def L_(self):
        return self._emu._engine_ready().get("L")[self._i]

    def predict(self, X, return_cov=False, return_std=False):
        m, v = self._emu._engine_ready().predict(np.atleast_2d(X), return_var=True)
        m, v = m[:, self._i], v[:, self._i]
        if return_std:
            return m, np.sqrt(np.clip(v, 0.0, None))       # sk:_gpr.py:479-485
        if return_cov:                                     # joint covariance of the query points (small W)
            mm, cc = self._emu._engine_ready().predict_cov(np.atleast_2d(X))
            return mm[:, self._i], cc[self._i]
        return m

    def score(self, X, y):
        """Coefficient of determination R^2 of the predictive mean (sklearn RegressorMixin.score, which the
        reference logs per GP: src/emulator.py:316-318): 1 - sum (y - mean)^2 / sum (y - ybar)^2."""
        y = np.asarray(y, dtype=np.float64)
        m = self.predict(X)
        return float(1.0 - ((y - m) ** 2).sum() / ((y - y.mean()) ** 2).sum())

    def sample_y(self, X, n_samples=1, random_state=0):
        """Draw from the GP posterior at X (sk:_gpr.py:498-540): mean/covariance from the device, the
        draw itself is numpy's RandomState.multivariate_normal exactly as sklearn calls it."""
        rng = _check_random_state(random_state)
        mean, cov = self.predict(X, return_cov=True)
        return rng.multivariate_normal(mean, cov, n_samples).T

    def __repr__(self):
        return "GaussianProcessRegressor(alpha={:g}, kernel={!r})".format(self._emu.alpha, self.kernel_)


class Emulator:
    def __init__(self, training_set_path=".", parameter_file="ABCD.txt",
                 npc=10, nrestarts=0, logTrafo=False, parameterTrafoPCA=False,
                 max_rel_uncertainty_data=0.1, exp_and_cov_diagonal=False,
                 perform_no_PCA=False, device=0):
        self.logTrafo_ = logTrafo
        self.parameterTrafoPCA_ = parameterTrafoPCA
        self.max_rel_uncertainty_data_ = max_rel_uncertainty_data
        self._load_training_data_pickle(training_set_path)
        self.exp_and_cov_diagonal_ = exp_and_cov_diagonal
        if self.exp_and_cov_diagonal_ and not self.logTrafo_:
            raise ValueError("exp_and_cov_diagonal can only be set to True if logTrafo is True.")
        self.perform_no_PCA_ = perform_no_PCA
        self.pardict = parse_model_parameter_file(parameter_file)
        self.design_min = np.array([v[1] for v in self.pardict.values()])
        self.design_max = np.array([v[2] for v in self.pardict.values()])
        self.npc = npc
        self.nrestarts = nrestarts
        self.nev, self.nobs = self.model_data.shape
        self.scaler = Standardizer()
        self.pca = WhitenedPCA()
        self.device = device
        self.alpha = 0.1                      # GPR(alpha=0.1), src/emulator.py:310
        self._engine = None
        self._trained = False
        self.fit_sharding = None              # dist.GPSharding: deal the npc hyper-parameter searches to the ranks
        if self.parameterTrafoPCA_:
            from .param_pca import ParameterPCA
            self._ppca = ParameterPCA(self.design_points, self.design_min, self.design_max)
            self.PCA_new_design_points = self._ppca.new_design_points
            self.design_min, self.design_max = self._ppca.design_min, self._ppca.design_max
            # the reference's attribute names for the three parameter groups (src/emulator.py:79-99)
            from . import param_pca as _pp
            self.targetVariance = _pp.TARGET_VARIANCE
            bulk, shear, yloss = self._ppca.groups
            self.indices_zeta_s_parameters = list(_pp.IDX_BULK)
            self.indices_eta_s_parameters = list(_pp.IDX_SHEAR)
            self.indices_yloss_parameters = list(_pp.IDX_YLOSS)
            self.paramTrafoScaler_bulk, self.paramTrafoPCA_bulk = bulk.scaler, bulk.pca
            self.paramTrafoScaler_shear, self.paramTrafoPCA_shear = shear.scaler, shear.pca
            self.paramTrafoScaler_yloss, self.paramTrafoPCA_yloss = yloss.scaler, yloss.pca

    @classmethod
    def from_reference(cls, ref, device=0):
        """Take over a TRAINED emulator object of the reference (`src.emulator.Emulator` after `trainEmulator`: what the dill
        pickles of examples/EmulatorTraining.ipynb hold and `Chain.loadEmulator` reads, src/mcmc.py:145-150) without retraining:
        its fitted scaler and PCA, the GPs' training inputs, targets and hyper-parameters (`gp.X_train_`, `gp.y_train_`,
        `gp.kernel_.theta`, `gp.alpha`: sk:_gpr.py:260-364), its flags and — with parameterTrafoPCA — its three fitted parameter
        maps go into a drop-in `Emulator`; the factorisation is redone on the device at first use.  Nothing of `ref` is called:
        attributes are read (duck-typed), so the reference package is needed only to unpickle `ref`.  Raises ValueError for an
        object that is not a trained emulator of that kind (an `EmulatorBAND`, a kernel outside RBF / Matern-3/2 / 5/2, GPs
        over different inputs)."""
        gps = list(getattr(ref, "gps", None) or [])
        if not gps or not all(hasattr(g, "kernel_") and hasattr(g, "X_train_") and hasattr(g, "y_train_") for g in gps):
            raise ValueError("from_reference: not a trained emulator with scikit-learn GPs (`gps[i].kernel_`, `X_train_`, `y_train_`)")

        def family(g):
            try:
                inner, white = g.kernel_.k1.k2, g.kernel_.k2
                name = type(inner).__name__
                if type(white).__name__ != "WhiteKernel" or type(g.kernel_.k1.k1).__name__ != "ConstantKernel":
                    raise AttributeError
            except AttributeError:
                raise ValueError("from_reference: kernel is not `c * RBF|Matern(length_scale) + WhiteKernel`") from None
            if name == "RBF":
                return "RBF"
            if name == "Matern" and float(inner.nu) in (1.5, 2.5):
                return "Matern" if float(inner.nu) == 1.5 else "Matern25"
            raise ValueError("from_reference: kernel family %s is not RBF / Matern-3/2 / Matern-5/2" % name)

        kinds = {family(g) for g in gps}
        # sklearn's normalize_y (sk:_gpr.py:265-282) rescales the targets and the predictions: the reference never sets it
        # (src/emulator.py:309-311) and the device state has no slot for it — such a GP would be adopted with wrong means and variances
        for g in gps:
            ym, ys = np.asarray(getattr(g, "_y_train_mean", 0.0)), np.asarray(getattr(g, "_y_train_std", 1.0))
            if bool(getattr(g, "normalize_y", False)) or np.any(ym != 0.0) or np.any(ys != 1.0):
                raise ValueError("from_reference: a GP was fitted with normalize_y (_y_train_mean / _y_train_std are not 0 / 1)")
            if np.ndim(g.alpha) != 0:
                raise ValueError("from_reference: a GP has a per-point alpha (array), not the reference's scalar")
        want = int(np.shape(ref.model_data)[1]) if bool(getattr(ref, "perform_no_PCA_", False)) else int(ref.npc)
        if len(gps) != want:
            raise ValueError("from_reference: %d GPs for %d %s" % (len(gps), want, "observables (perform_no_PCA)" if bool(
                getattr(ref, "perform_no_PCA_", False)) else "principal components (npc)"))
        X = np.ascontiguousarray(gps[0].X_train_, dtype=np.float64)
        if len(kinds) != 1 or len({float(g.alpha) for g in gps}) != 1 or \
                any(np.shape(g.X_train_) != X.shape or not np.array_equal(g.X_train_, X) for g in gps[1:]):
            raise ValueError("from_reference: the GPs differ in kernel family, alpha or training inputs")
        thetas = np.array([np.asarray(g.kernel_.theta, dtype=np.float64) for g in gps])
        if thetas.shape != (len(gps), X.shape[1] + 2):
            raise ValueError("from_reference: kernel_.theta is not [log c, log l_1..d, log noise] (anisotropic length scales expected)")
        emu = cls.__new__(cls)
        emu.logTrafo_ = bool(getattr(ref, "logTrafo_", False))
        emu.parameterTrafoPCA_ = bool(getattr(ref, "parameterTrafoPCA_", False))
        emu.max_rel_uncertainty_data_ = getattr(ref, "max_rel_uncertainty_data_", 0.1)
        emu.exp_and_cov_diagonal_ = bool(getattr(ref, "exp_and_cov_diagonal_", False))
        emu.perform_no_PCA_ = bool(getattr(ref, "perform_no_PCA_", False))
        emu.pardict = getattr(ref, "pardict", None)
        emu.design_min, emu.design_max = np.array(ref.design_min, dtype=np.float64), np.array(ref.design_max, dtype=np.float64)
        emu.design_points = np.array(ref.design_points, dtype=np.float64)
        emu.design_points_org_ = np.array(getattr(ref, "design_points_org_", ref.design_points), dtype=np.float64)
        emu.model_data = np.array(ref.model_data, dtype=np.float64)
        emu.model_data_err = np.array(getattr(ref, "model_data_err", np.zeros_like(emu.model_data)), dtype=np.float64)
        emu.npc, emu.nrestarts = int(ref.npc), int(getattr(ref, "nrestarts", 0))
        emu.nev, emu.nobs = emu.model_data.shape
        emu.scaler, emu.pca = Standardizer(), WhitenedPCA()
        for name in ("mean_", "scale_", "var_"):
            setattr(emu.scaler, name, np.array(getattr(ref.scaler, name), dtype=np.float64))
        if not emu.perform_no_PCA_:
            for name in ("mean_", "components_", "explained_variance_", "explained_variance_ratio_"):
                setattr(emu.pca, name, np.array(getattr(ref.pca, name), dtype=np.float64))
            emu.pca.n_components_ = int(ref.pca.n_components_)
        emu.device, emu.alpha = device, float(gps[0].alpha)
        emu._engine, emu._like_key, emu.fit_sharding = None, None, None
        if emu.parameterTrafoPCA_:
            from . import param_pca as _pp
            emu._ppca = _pp.ParameterPCA.from_fitted(
                [(ref.paramTrafoScaler_bulk, ref.paramTrafoPCA_bulk), (ref.paramTrafoScaler_shear, ref.paramTrafoPCA_shear),
                 (ref.paramTrafoScaler_yloss, ref.paramTrafoPCA_yloss)], ref.PCA_new_design_points, emu.design_min, emu.design_max)
            emu.PCA_new_design_points = emu._ppca.new_design_points
            emu.targetVariance = getattr(ref, "targetVariance", _pp.TARGET_VARIANCE)
            bulk, shear, yloss = emu._ppca.groups
            emu.indices_zeta_s_parameters = list(_pp.IDX_BULK)
            emu.indices_eta_s_parameters = list(_pp.IDX_SHEAR)
            emu.indices_yloss_parameters = list(_pp.IDX_YLOSS)
            emu.paramTrafoScaler_bulk, emu.paramTrafoPCA_bulk = bulk.scaler, bulk.pca
            emu.paramTrafoScaler_shear, emu.paramTrafoPCA_shear = shear.scaler, shear.pca
            emu.paramTrafoScaler_yloss, emu.paramTrafoPCA_yloss = yloss.scaler, yloss.pca
        emu._X_train = X
        emu._Z_train = np.ascontiguousarray([np.asarray(g.y_train_, dtype=np.float64).reshape(-1) for g in gps])
        emu.kernel_type_, emu._ngp = kinds.pop(), len(gps)
        emu.thetas_ = thetas
        emu.lml_ = np.array([float(getattr(g, "log_marginal_likelihood_value_", np.nan)) for g in gps])
        emu._state_serial = next(_STATE_SERIAL)
        emu._build_transform()
        emu._trained = True
        emu.gps = [FittedGP(emu, i) for i in range(emu._ngp)]
        return emu

    # scalar forms of the parametrised functions behind parameterTrafoPCA (src/emulator.py:102-126)
    def parametrization_zeta_over_s_vs_T(self, zeta_max, T_zeta0, sigma_plus, sigma_minus, T, mu_B):
        from .param_pca import zeta_over_s
        par = np.array([[zeta_max, T_zeta0, sigma_plus, sigma_minus]], dtype=np.float64)
        return float(zeta_over_s(par, T=np.array([T], dtype=np.float64), mu_B=mu_B)[0, 0])

    def parametrization_eta_over_s_vs_mu_B(self, eta_0, eta_2, eta_4, mu_B):
        from .param_pca import eta_over_s
        return float(eta_over_s(np.array([[eta_0, eta_2, eta_4]], dtype=np.float64),
                                mu_B=np.array([mu_B], dtype=np.float64))[0, 0])

    def parametrization_y_loss_vs_y_init(self, yloss_2, yloss_4, yloss_6, y_init):
        from .param_pca import y_loss
        return float(y_loss(np.array([[yloss_2, yloss_4, yloss_6]], dtype=np.float64),
                            y_init=np.array([y_init], dtype=np.float64))[0, 0])

    # ------------------------------------------------------------------ data loading
    def _load_training_data_pickle(self, dataFile):
        """{event_id -> {"parameter": f64[ndim], "obs": f64[2,nobs]}}; points whose largest
        relative statistical error exceeds max_rel_uncertainty_data are dropped; optional log
        transform (src/emulator.py:378-415)."""
        with open(dataFile, "rb") as fp:
            data = pickle.load(fp)
        X, Y, E = [], [], []
        dropped = 0
        for key in sorted(data.keys(), key=lambda s: int(s)):
            val, err = data[key]["obs"][0], data[key]["obs"][1]
            if np.abs(err / (val + 1e-16)).max() > self.max_rel_uncertainty_data_:
                dropped += 1
                continue
            X.append(data[key]["parameter"])
            if self.logTrafo_:
                Y.append(np.log(np.abs(val) + 1e-30))
                E.append(np.abs(err / (val + 1e-30)))
            else:
                Y.append(val)
                E.append(err)
        self.design_points = np.array(X)
        self.design_points_org_ = np.copy(self.design_points)
        self.model_data = np.array(Y)
        self.model_data_err = np.nan_to_num(np.abs(np.array(E)))
        log.info("training set: %d points kept, %d discarded", len(Y), dropped)

    def getAvgTrainingDataRelError(self):
        return np.mean(np.nan_to_num(self.model_data_err / self.model_data), axis=0)

    def outputPCAvsParam(self):
        Z = self.pca.fit_transform(self.scaler.fit_transform(self.model_data))[:, :self.npc]
        return self.design_points, Z.T

    # ------------------------------------------------------------------ training
    def trainEmulatorAutoMask(self):
        self.trainEmulator([True] * self.nev)

    def _theta0_bounds(self, kernel_type):
        """Initial theta and log-bounds of `1.*RBF(ptp, ptp x (1e-1,1e2)) + White(.05,(1e-2,1e2))`
        (src/emulator.py:286-306); the constant kernel's bounds are sklearn's default (1e-5,1e5)."""
        ptp = self.design_max - self.design_min
        lo, hi = _KERNELS[kernel_type][1]
        d = ptp.shape[0]
        theta0 = np.concatenate([[0.0], np.log(ptp), [np.log(0.05)]])
        bounds = np.empty((d + 2, 2))
        bounds[0] = np.log([1e-5, 1e5])
        bounds[1:1 + d, 0] = np.log(ptp * lo)
        bounds[1:1 + d, 1] = np.log(ptp * hi)
        bounds[d + 1] = np.log([1e-2, 1e2])
        return theta0, bounds

    def trainEmulator(self, eventMask, kernel_type="RBF", thetas=None):
        """Fit one GP per principal component (src/emulator.py:257-363).  `thetas` (optional,
        [npc, d+2]) skips the hyper-parameter search and factorises at the given values."""
        self._prepare_training(eventMask, kernel_type)
        eng = self._new_engine()
        if thetas is not None:
            thetas = np.array(thetas, dtype=np.float64).reshape(self._ngp, -1)
            lml = eng.lml(thetas, eval_gradient=False)
        else:
            thetas, lml = self._optimise(eng, kernel_type)
        self._finish_training(eng, thetas, lml)

    def _prepare_training(self, eventMask, kernel_type):
        """the host part in front of the GP fits: scaler, PCA, targets (src/emulator.py:257-285)"""
        if kernel_type not in _KERNELS:
            raise ValueError("Unknown kernel type: {}".format(kernel_type))
        mask = np.asarray(eventMask, dtype=bool)
        S = self.scaler.fit_transform(self.model_data[mask])
        if self.perform_no_PCA_:
            Z = S
        else:
            Z = self.pca.fit_transform(S)[:, :self.npc]
            log.info("%d PCs explain %.5f of variance", self.npc,
                     self.pca.explained_variance_ratio_[:self.npc].sum())
        X = (self.PCA_new_design_points if self.parameterTrafoPCA_ else self.design_points)[mask]
        self._X_train = np.ascontiguousarray(X, dtype=np.float64)
        self._Z_train = np.ascontiguousarray(Z.T, dtype=np.float64)       # [P, N]
        self.kernel_type_ = kernel_type
        self._ngp = self._Z_train.shape[0]

    def _finish_training(self, eng, thetas, lml):
        """the final factorisation at theta*, the observable transform and the per-GP scores (src/emulator.py:316-363)"""
        self.thetas_, self.lml_ = thetas, lml
        self._state_serial = next(_STATE_SERIAL)
        eng.set_theta(self.thetas_)
        eng.factor()
        self._build_transform()
        self._push_transform(eng)
        self._trained = True
        self.gps = [FittedGP(self, i) for i in range(self._ngp)]
        # R^2 of each GP on its training targets (src/emulator.py:316-318)
        m = eng.predict(self._X_train, return_var=False)
        zt = self._Z_train.T
        self.gp_scores_ = 1.0 - ((zt - m) ** 2).sum(0) / ((zt - zt.mean(0)) ** 2).sum(0)
        log.info("GP scores: %s", self.gp_scores_)
        if not self.perform_no_PCA_:                  # the reference's per-GP summary line (src/emulator.py:320-328)
            for n, gp in enumerate(self.gps):
                log.info("GP %d: %.5f of variance, LML = %.5g, Score = %.2f, kernel: %s", n,
                         self.pca.explained_variance_ratio_[n], gp.log_marginal_likelihood_value_,
                         self.gp_scores_[n], gp.kernel_)

    def _search_engine(self, idx, copies):
        """a fit-only context holding the GPs `idx` of this emulator `copies` times (the starts of their searches)"""
        return _SearchEngine(self.device, [self._X_train] * len(idx), [self._Z_train[i] for i in idx],
                             _KERNELS[self.kernel_type_][0], self.alpha, copies)

    def _optimise(self, eng, kernel_type, draws=None):
        """argmax LML per GP with scipy L-BFGS-B (sk:_gpr.py:296-337,654-670).  The P searches — and, with
        nrestarts > 0, the 1 + nrestarts starts of each — are independent; they run in lock-step threads so that every
        objective call is ONE batched device evaluation of the log-marginal likelihoods and gradients of all searches
        still running."""
        theta0, bounds = self._theta0_bounds(kernel_type)
        sh = self.fit_sharding
        copies = 1 + int(self.nrestarts)
        Np = -(-self._X_train.shape[0] // 64) * 64
        if copies * self._ngp * 3 * 8 * Np * Np > _BATCH_BYTES_MAX:     # the batch of all starts would not fit comfortably: start by start
            copies = 1
        if sh is None or sh.world == 1:
            if copies == 1:
                return search_hyperparameters(lambda idx: eng, self._ngp, theta0, bounds, self.nrestarts, draws=draws)
            return search_hyperparameters(lambda idx: self._search_engine(idx, copies), self._ngp, theta0, bounds,
                                          self.nrestarts, close=True, draws=draws)

        def sub_engine(idx):            # this rank's GPs only: same design, a subset of the target rows
            if copies > 1:
                return self._search_engine(idx, copies)
            sub = GPEngine(self.device)
            sub.set_data(self._X_train, self._Z_train[idx], _KERNELS[kernel_type][0], self.alpha)
            return sub
        return search_hyperparameters(sub_engine, self._ngp, theta0, bounds, self.nrestarts, sh, close=True, draws=draws)

    def _optimise_with_points(self, kernel_type, theta0_bounds, draws):
        """the search of this emulator alone from given restart points (train_emulators' fallback for batches that do not fit)"""
        theta0, bounds = theta0_bounds
        eng = self._new_engine()
        best_theta = np.tile(theta0, (self._ngp, 1))
        best_val = np.full(self._ngp, np.inf)
        for start in [best_theta.copy()] + [draws[:, r, :] for r in range(draws.shape[1] if draws.ndim == 3 else 0)]:
            th, val = _batched_lbfgsb(eng, start, bounds)
            better = val < best_val
            best_theta[better], best_val[better] = th[better], val[better]
        return best_theta, -best_val

    def _build_transform(self):
        if self.perform_no_PCA_:
            self._A = self._cov_trunc = self._trans_matrix = None
            return
        self._trans_matrix, self._A, self._cov_trunc = observable_transform(
            self.pca.components_, self.pca.explained_variance_, self.scaler.scale_, self.scaler.var_,
            self.npc)
        self._var_trans = np.einsum("ki,kj->kij", self._A, self._A).reshape(self.npc, self.nobs ** 2)

    @property
    def _mode(self):
        if self.perform_no_PCA_:
            return MODE_NO_PCA_EXPDIAG if self.exp_and_cov_diagonal_ else MODE_NO_PCA
        return MODE_EXPDIAG if self.exp_and_cov_diagonal_ else MODE_PCA

    def _push_transform(self, eng):
        if self.perform_no_PCA_:
            eng.set_transform(self._mode, self.scaler.mean_, scale=self.scaler.scale_)
        else:
            eng.set_transform(self._mode, self.scaler.mean_, A=self._A, cov_trunc=self._cov_trunc)
        if self.parameterTrafoPCA_:                  # device pre-pass for resident log-posterior loops
            eng.set_param_map(self._ppca, self.design_points.shape[1])

    # ------------------------------------------------------------------ engine lifetime
    def _new_engine(self):
        if self._engine is not None:
            self._engine._check_pid()
            self._engine.close()
        eng = GPEngine(self.device)
        eng.set_data(self._X_train, self._Z_train, _KERNELS[self.kernel_type_][0], self.alpha)
        if getattr(self, "predict_arithmetic", "fp64") == "int8":
            eng.tune("predict_sliced", 1)
        self._engine = eng
        self._like_key = None
        return eng

    def set_predict_arithmetic(self, which="fp64"):
        """"fp64" (default): V = L^-1 K*^T of every batch on the fp64 matrix cores; "int8": on the int8 matrix cores
        (gpb_ctx_option 51, csrc/gpb_sliced.hip — six 8-bit digit planes per operand, exact int32 sums, fp64 combine), 1.8-2.1x faster,
        the predictive variance within ~2e-11 relative of the fp64 kernel's, for an emulator whose GPs all have 1 + c / sigma_n^2 <= 128
        (others keep the fp64 kernel).  The choice is kept through pickling; a walker's bits do not depend on batch size, compaction
        or rank count in either arithmetic, but they differ between the two in the last digits: set it once per analysis."""
        if which not in ("fp64", "int8"):
            raise ValueError("predict arithmetic must be 'fp64' or 'int8'")
        self.predict_arithmetic = which
        if self._engine is not None:
            self._engine._check_pid()
            self._engine.tune("predict_sliced", 1 if which == "int8" else 0)
        self._state_serial = next(_STATE_SERIAL)      # (replicas of a sharded run must agree on it: it is part of the digest)
        return self

    def _engine_ready(self):
        """(Re)create the device state after unpickling, or in a worker forked BEFORE the parent touched the GPU.  A
        worker forked afterwards inherits a live handle on a context that did not survive the fork (the reference's
        pocoMC `pool=int`, src/mcmc.py:775-776,798-804): RuntimeError, before any HIP call."""
        if self._engine is not None:
            self._engine._check_pid()
        if self._engine is None:
            if not self._trained:
                raise RuntimeError("Emulator is not trained")
            eng = self._new_engine()
            eng.set_theta(self.thetas_)
            eng.factor()
            self._push_transform(eng)
        return self._engine

    def state_digest(self):
        """sha256 over everything the device state is built from (design, targets, theta*, kernel, transform arrays):
        equal digests = bit-identical replicas (the device side is deterministic)."""
        import hashlib
        h = hashlib.sha256()
        h.update(repr((self.kernel_type_, self._ngp, self.nobs, float(self.alpha), int(self._mode),
                       bool(self.parameterTrafoPCA_))).encode())
        if getattr(self, "predict_arithmetic", "fp64") != "fp64":      # (absent / fp64: the digest of earlier rounds' objects)
            h.update(b"predict_arithmetic=" + self.predict_arithmetic.encode())
        arrs = [self._X_train, self._Z_train, self.thetas_, self.scaler.mean_]
        arrs += [self.scaler.scale_] if self.perform_no_PCA_ else [self._A, self._cov_trunc]
        if self.parameterTrafoPCA_:          # the parameter-space map in front of the GPs (src/emulator.py:492-551)
            for g in self._ppca.groups:
                arrs += [g.scaler.mean_, g.scaler.scale_, g.pca.mean_, g.pca.components_]
        for a in arrs:
            a = np.ascontiguousarray(a, dtype=np.float64)
            h.update(repr(a.shape).encode()); h.update(a.tobytes())
        return h.digest()

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_engine"] = None
        st["_like_key"] = None
        st["fit_sharding"] = None
        st.pop("_state_serial", None)
        st.pop("gps", None)
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self.__dict__.setdefault("fit_sharding", None)
        self._state_serial = next(_STATE_SERIAL)
        if self._trained:
            self.gps = [FittedGP(self, i) for i in range(self._ngp)]

    # ------------------------------------------------------------------ prediction
    def print_learning_curve(self):
        """Learning curves of the GPs over the principal components (src/emulator.py:424-462): sklearn's `learning_curve` — five
        unshuffled folds (sk:model_selection/_split.py KFold), training sets of 0.2 / 0.4 / 0.6 / 0.8 / 0.9 of a fold's training
        events (the first n of them), scored by R^2 on those events and on the held-out fold — over fits of
        `GPR(1. * RBF(ptp, ptp x (.01, 100)) + WhiteKernel(.01**2, (.001**2, 1)), alpha=0.)`.  Returns, per GP, the table
        [train size, mean train score, mean test score].  The reference runs the 25 hyper-parameter searches of every GP one
        after the other; here the five folds of one train size (same padded size) and all GPs are ONE lock-step batch on the
        device.  The reference refits `self.scaler` / `self.pca` on all events as a side effect; the drop-in works on copies and
        leaves a trained emulator as it is."""
        S = Standardizer().fit_transform(self.model_data)
        Z = WhitenedPCA().fit_transform(S)[:, :self.npc]
        X = np.ascontiguousarray(self.PCA_new_design_points if self.parameterTrafoPCA_ else self.design_points, dtype=np.float64)
        ptp = self.design_max - self.design_min
        d, P, n = ptp.shape[0], Z.shape[1], X.shape[0]
        theta0 = np.concatenate([[0.0], np.log(ptp), [np.log(.01 ** 2)]])
        bounds = np.empty((d + 2, 2))
        bounds[0] = np.log([1e-5, 1e5])                   # the constant kernel's default bounds
        bounds[1:1 + d] = np.log(np.outer(ptp, (.01, 100)))
        bounds[d + 1] = np.log([.001 ** 2, 1])
        nfold = 5
        if n < nfold:
            raise ValueError("Cannot have number of splits n_splits=%d greater than the number of samples: n_samples=%d." % (nfold, n))
        fold_sizes = np.full(nfold, n // nfold)
        fold_sizes[:n % nfold] += 1
        stops = np.cumsum(fold_sizes)
        folds = [(np.r_[0:b - m, b:n], np.arange(b - m, b)) for b, m in zip(stops, fold_sizes)]       # (train, test), in order
        n_max = len(folds[0][0])
        sizes = np.unique(np.clip((np.array([0.2, 0.4, 0.6, 0.8, 0.9]) * n_max).astype(int), 1, n_max))
        r2 = lambda y, m: 1.0 - ((y - m) ** 2).sum(0) / ((y - y.mean(0)) ** 2).sum(0)
        train = np.empty((len(sizes), nfold, P))
        test = np.empty_like(train)
        for si, m in enumerate(sizes):
            rows = [tr[:m] for tr, _ in folds]
            # the searches of all folds and GPs at this size: virtual GP f * P + i = fold f, GP i
            se = _SearchEngine(self.device, [X[r] for r in rows for _ in range(P)], [Z[r, i] for r in rows for i in range(P)],
                               "RBF", 0.0)
            thetas, _ = search_hyperparameters(lambda idx: se, nfold * P, theta0, bounds, 0, close=True)
            for f, (r, (_, te)) in enumerate(zip(rows, folds)):
                eng = GPEngine(self.device)
                try:
                    eng.set_data(X[r], Z[r].T, "RBF", 0.0)
                    eng.set_theta(thetas[f * P:(f + 1) * P])
                    info = np.asarray(eng.factor(raise_on_fail=False))
                    good = np.nonzero(info == 0)[0]
                    if len(good) < P:
                        # sklearn's learning_curve records error_score = nan for a fit that fails and keeps the others
                        # (src/emulator.py:449-455: alpha = 0 and a noise bound of 1e-6 leave K barely positive definite);
                        # the engine installs no factorisation when any GP fails, so the others are factored again alone
                        log.warning("print_learning_curve: K not positive definite for GP(s) %s at train size %d, fold %d: scores set to NaN",
                                    np.nonzero(info != 0)[0].tolist(), m, f)
                        train[si, f], test[si, f] = np.nan, np.nan
                        if len(good):
                            eng.set_data(X[r], Z[r][:, good].T, "RBF", 0.0)
                            eng.set_theta(thetas[f * P:(f + 1) * P][good])
                            eng.factor()
                    if len(good):
                        train[si, f, good] = r2(Z[r][:, good], eng.predict(X[r], return_var=False))
                        test[si, f, good] = r2(Z[te][:, good], eng.predict(X[te], return_var=False))
                finally:
                    eng.close()
        trainStatus = []
        for i in range(P):
            trainStatus.append(np.array([sizes, train[:, :, i].mean(1), test[:, :, i].mean(1)]).transpose())
            log.info("GP %d:", i)
            for m, a, b in zip(sizes, train[:, :, i], test[:, :, i]):
                log.info("%d samples were used to train the model", m)
                log.info("The average train accuracy is %.2f", a.mean())
                log.info("The average test accuracy is %.2f", b.mean())
        return trainStatus

    def _map_parameters(self, X):
        return self._ppca.transform(X) if self.parameterTrafoPCA_ else X

    def predict(self, X, return_cov=True, extra_std=0):
        """mean[W,nobs] (and cov[W,nobs,nobs]) at X[W,ndim] (src/emulator.py:465-605).
        extra_std: scalar or length-W array added in quadrature to every GP's predictive std."""
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        Xg = np.ascontiguousarray(self._map_parameters(X))
        eng = self._engine_ready()
        W = Xg.shape[0]
        es = None
        if return_cov:
            es = np.ascontiguousarray(np.broadcast_to(np.asarray(extra_std, dtype=np.float64).reshape(-1), (W,)))
        # very long inputs go through in slabs sized for the device workspaces (K*^T: P x N doubles per row,
        # staged covariances: nobs^2 per row); a row's numbers do not depend on how the batch is cut
        per_row = 8 * (self._ngp * self._X_train.shape[0] + (self.nobs ** 2 if return_cov else 0))
        slab = int(min(max((8 << 30) // per_row, 1024), 1 << 17)) // 128 * 128
        if W <= slab:
            return eng.emu_predict(Xg, return_cov=return_cov, extra_std=es)
        mean = np.empty((W, self.nobs))
        cov = np.empty((W, self.nobs, self.nobs)) if return_cov else None
        for i0 in range(0, W, slab):
            sl = slice(i0, min(i0 + slab, W))
            part = eng.emu_predict(np.ascontiguousarray(Xg[sl]), return_cov=return_cov,
                                   extra_std=None if es is None else np.ascontiguousarray(es[sl]))
            if return_cov:
                mean[sl], cov[sl] = part
            else:
                mean[sl] = part
        return (mean, cov) if return_cov else mean

    def sample_y(self, X, n_samples=1, random_state=None):
        """Sample model output at X -> [n_samples_X, n_samples, nobs] (src/emulator.py:608-633): one
        posterior draw per emulated PC (each GP is handed the same `random_state`, as the reference does),
        standard-normal draws from numpy's global generator for the neglected PCs, then the PC ->
        observable map."""
        if self.perform_no_PCA_:
            log.warning("Sampling from raw data is not implemented.")
            return None
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        Xg = np.ascontiguousarray(self._map_parameters(X))
        mean, cov = self._engine_ready().predict_cov(Xg)                  # [W,P], [P,W,W]
        draws = []
        for i in range(self._ngp):
            rng = _check_random_state(random_state)
            draws.append(rng.multivariate_normal(mean[:, i], cov[i], n_samples).T[:, :, np.newaxis])
        rest = np.random.standard_normal((X.shape[0], n_samples, self.pca.n_components_ - self.npc))
        Z = np.concatenate(draws + [rest], axis=2)
        return self._inverse_transform(Z)

    def _inverse_transform(self, Z):
        """principal components Z[..., k] -> observables [..., nobs], using the first k rows of the PC ->
        observable map (src/emulator.py:366-375)"""
        Z = np.asarray(Z, dtype=np.float64)
        return np.dot(Z, self._trans_matrix[:Z.shape[-1]]) + self.scaler.mean_

    # ------------------------------------------------------------------ hold-out validation helpers
    def _holdout(self, nTestPoints, on_training, thetas=None):
        mask = np.ones(self.nev, dtype=bool)
        mask[self.nev - nTestPoints:] = False
        self.trainEmulator(mask, thetas=thetas)
        vmask = mask if on_training else ~mask
        pred, cov = self.predict(self.design_points_org_[vmask], return_cov=True)
        perr = np.sqrt(np.diagonal(cov, axis1=1, axis2=2))
        if self.logTrafo_ and not self.exp_and_cov_diagonal_:
            pred, perr = np.exp(pred), perr * np.exp(pred)
        if self.logTrafo_:
            truth = np.exp(self.model_data[vmask])
            terr = self.model_data_err[vmask] * truth
        else:
            truth, terr = self.model_data[vmask], self.model_data_err[vmask]
        r = lambda a: np.array(a).reshape(-1, self.nobs)
        return r(pred), r(perr), r(truth), r(terr)

    def testEmulatorErrors(self, nTestPoints=1, thetas=None):
        """Train on the first nev-nTestPoints points, predict the rest (src/emulator.py:636-679): returns
        (predictions, their errors, true values, their errors), each [nTestPoints, nobs].  `thetas` (extension)
        skips the hyper-parameter search of the retraining, as in trainEmulator."""
        return self._holdout(nTestPoints, on_training=False, thetas=thetas)

    def testEmulatorErrorsWithTrainingPoints(self, nTestPoints=1, thetas=None):
        """Same split, but predict the training points themselves (src/emulator.py:682-726)."""
        return self._holdout(nTestPoints, on_training=True, thetas=thetas)


_BATCH_BYTES_MAX = 96 << 30          # three N x N matrices per virtual GP of a search batch: beyond this the emulators train one by one


def train_emulators(emulators, eventMasks=None, kernel_type="RBF"):
    """Train the emulators of a chain TOGETHER: what `for emu in emulators: emu.trainEmulator(mask, kernel_type)` does
    (the reference fits dataset after dataset and GP after GP: examples/EmulatorTraining.ipynb:124-138,
    src/emulator.py:309-315), with the hyper-parameter searches of all GPs of all emulators — and all their restarts — in
    ONE lock-step batch per group of emulators whose designs pad to the same number of points (same number of parameters):
    at N ~ 1000 a fit is bound by the latency of its Cholesky chain, so 63 matrices per launch cost little more than 7.
    Every GP's search is the one its emulator's own training runs (same start points: numpy's global RandomState is drawn
    emulator after emulator, GP after GP; same objective values bit for bit), so theta* is identical.  `eventMasks`: one
    mask per emulator (default: all events)."""
    emulators = list(emulators)
    if eventMasks is None:
        eventMasks = [[True] * e.nev for e in emulators]
    for emu, mask in zip(emulators, eventMasks):
        emu._prepare_training(mask, kernel_type)
    kern = _KERNELS[kernel_type][0]
    # restart points in the order sequential trainings would draw them
    t0b = [emu._theta0_bounds(kernel_type) for emu in emulators]
    rs = _check_random_state(None)
    draws = [np.array([[rs.uniform(b[:, 0], b[:, 1]) for _ in range(int(emu.nrestarts))] for _ in range(emu._ngp)])
             for emu, (_, b) in zip(emulators, t0b)]
    groups = {}
    for i, emu in enumerate(emulators):
        sharded = emu.fit_sharding is not None and emu.fit_sharding.world > 1
        key = ("alone", i) if sharded else (emu.device, emu._X_train.shape[1], -(-emu._X_train.shape[0] // 64), float(emu.alpha))
        groups.setdefault(key, []).append(i)
    results = [None] * len(emulators)
    for key, members in groups.items():
        if key[0] == "alone":
            # a sharded emulator: its GPs are dealt to the ranks of ITS group (collective: every rank of that group is here), the
            # 1 + nrestarts starts of this rank's GPs in one lock-step batch, from the points drawn above — Emulator._optimise
            i = members[0]
            results[i] = emulators[i]._optimise(None, kernel_type, draws=draws[i])
            continue
        smax = 1 + max(int(emulators[i].nrestarts) for i in members)
        nvirt = sum(emulators[i]._ngp * (1 + int(emulators[i].nrestarts)) for i in members)
        if nvirt * 3 * 8 * (64 * key[2]) ** 2 > _BATCH_BYTES_MAX:
            # the batch's three N x N matrices per virtual GP would not fit comfortably (several emulators, or one with many
            # restarts): these emulators train one by one, start by start
            for i in members:
                emu = emulators[i]
                results[i] = emu._optimise_with_points(kernel_type, t0b[i], draws[i])
            continue
        Xs, Zs, start, bnd, owner = [], [], [], [], []
        for s_ in range(smax):                       # virtual GPs start-major: the first block is every GP's theta0 start
            for i in members:
                emu = emulators[i]
                if s_ > int(emu.nrestarts):
                    continue
                for g in range(emu._ngp):
                    Xs.append(emu._X_train); Zs.append(emu._Z_train[g])
                    start.append(t0b[i][0] if s_ == 0 else draws[i][g, s_ - 1])
                    bnd.append(t0b[i][1]); owner.append((i, g, s_))
        eng = GPEngine(emulators[members[0]].device)
        try:
            eng.set_data_multi(Xs, Zs, kern, float(emulators[members[0]].alpha))
            th, val = _batched_lbfgsb(eng, np.array(start), np.array(bnd))
        finally:
            eng.close()
        for i in members:
            emu = emulators[i]
            results[i] = (np.tile(t0b[i][0], (emu._ngp, 1)), np.full(emu._ngp, np.inf))
        for v, (i, g, s_) in enumerate(owner):       # start-major = sklearn's order per GP: the first of equal optima wins
            if val[v] < results[i][1][g]:
                results[i][0][g], results[i][1][g] = th[v], val[v]
        for i in members:
            results[i] = (results[i][0], -results[i][1])
    for emu, (thetas, lml) in zip(emulators, results):
        emu._finish_training(emu._new_engine(), thetas, lml)
    return emulators


def search_hyperparameters(make_engine, P, theta0, bounds, nrestarts=0, sharding=None, close=False, rng=None, draws=None):
    """theta*[P, d+2] and LML*[P]: L-BFGS-B from `theta0` plus `nrestarts` log-uniform starts per GP
    (sk:_gpr.py:296-337); `rng`: None (numpy's global RandomState, as sklearn), an int seed or a RandomState.  `make_engine(idx)` returns an object whose `.lml(theta[len(idx), d+2],
    eval_gradient=True)` serves the GPs `idx`.  With `sharding` (dist.GPSharding, SURVEY §8e "fit-side") the
    P searches are dealt round-robin to the ranks and ONE all-gather puts every result on every rank; a GP's
    search does not depend on which other GPs share its batch, so the result is the unsharded one.

    All 1 + nrestarts starts of all GPs run as ONE lock-step batch when the engine offers `restart_batch(n)` (an engine
    holding n copies of every GP: `_RestartEngine`): the fit is bound by the latency of its Cholesky chain, so (1 + nrestarts) P
    matrices per launch cost little more than P (the reference, like sklearn, runs start after start and GP after GP,
    src/emulator.py:309-315); searches that have converged leave the batch."""
    idx = np.arange(P) if sharding is None else sharding.mine(P)
    # restart points as sklearn draws them: GPR(random_state=None) takes numpy's GLOBAL RandomState (np.random.seed
    # makes a fit reproducible) and draws, GP after GP, one log-uniform theta per restart (sk:_gpr.py:259,318-325).
    # Every rank draws all P x nrestarts points and keeps its own, so a sharded fit equals the unsharded one.
    # `draws` [P, nrestarts, d+2]: the points already drawn (train_emulators draws for all its emulators up front, in order).
    nrestarts = int(nrestarts)
    if draws is None:
        rs = _check_random_state(rng)
        draws = np.array([[rs.uniform(bounds[:, 0], bounds[:, 1]) for _ in range(nrestarts)] for _ in range(P)])
    elif nrestarts and np.shape(draws)[:2] != (P, nrestarts):
        raise ValueError("search_hyperparameters: draws must be [P, nrestarts, d + 2]")
    best_theta = np.tile(theta0, (idx.size, 1))
    best_val = np.full(idx.size, np.inf)
    if idx.size:
        eng = make_engine(idx)
        try:
            starts = [np.tile(theta0, (idx.size, 1))]
            starts += [draws[idx, r, :] for r in range(nrestarts)]
            batch = getattr(eng, "restart_batch", None)
            if batch is not None and nrestarts > 0:
                # virtual GP v = s * n + i: start s of GP i (the engine holds the GPs' data once per start)
                th, val = _batched_lbfgsb(batch(1 + nrestarts), np.concatenate(starts, axis=0), bounds)
                results = [(th[s * idx.size:(s + 1) * idx.size], val[s * idx.size:(s + 1) * idx.size])
                           for s in range(1 + nrestarts)]
            else:
                results = [_batched_lbfgsb(eng, start, bounds) for start in starts]
            for th, val in results:                    # in sklearn's order: the first of equal optima wins (np.argmin)
                better = val < best_val
                best_theta[better], best_val[better] = th[better], val[better]
        finally:
            if close:
                eng.close()
    if sharding is None:
        return best_theta, -best_val
    return sharding.gather(P, idx, best_theta, -best_val)


class _SearchEngine:
    """The GPs of a hyper-parameter search on one fit-only device context (gpb_gp_set_multi): `Xs[i]`, `Zs[i]` the design and
    targets of GP i — of one emulator or of several — each stored `copies` times (the starts of its search).  `.lml(theta[V])`
    evaluates all V = copies x n virtual GPs, `.lml_active(idx, theta)` the ones still searching."""

    def __init__(self, device, Xs, Zs, kernel, alpha, copies=1):
        self.n = len(Xs)
        self.copies = int(copies)
        self.eng = GPEngine(device)
        self.eng.set_data_multi(list(Xs) * self.copies, list(Zs) * self.copies, kernel, alpha)

    def restart_batch(self, copies):
        assert copies == self.copies
        return self

    def lml(self, theta, eval_gradient=True):
        return self.eng.lml(theta, eval_gradient)

    def lml_active(self, idx, theta):
        return self.eng.lml_subset(idx, theta, True)

    def close(self):
        self.eng.close()


def _batched_lbfgsb(eng, start, bounds):
    """P independent L-BFGS-B minimisations of -LML_p(theta_p) in LOCK-STEP: every round of objective calls is served by one
    batched device evaluation — of the searches still running when the engine can evaluate a subset (`lml_active`), else of all
    P (`lml`; a finished search's theta stays where it ended).  Each search is scipy's own (sk:_gpr.py:654-670 calls
    scipy.optimize.minimize(method="L-BFGS-B", jac=True, bounds=...)): driven either by ONE thread that steps scipy's
    reverse-communication routine for all searches in turn (_lockstep_setulb: no thread switches, no GIL hand-overs — 63 scipy
    threads cost more per round than the device evaluation they wait for), or, where this scipy does not offer that routine in the
    form known here, by one scipy.optimize.minimize thread per search (_lockstep_threads).  Same theta*, bit for bit."""
    if _setulb_usable():
        return _lockstep_setulb(eng, start, bounds)
    return _lockstep_threads(eng, start, bounds)


def _evaluate_round(eng, cur, ids):
    """-LML and its gradient of the searches `ids` at cur[ids] (one batched device evaluation); rows of other searches NaN"""
    P = cur.shape[0]
    subset = getattr(eng, "lml_active", None)
    if subset is not None and len(ids) < P:
        v = np.full(P, np.nan); g = np.full((P, cur.shape[1]), np.nan)
        v[ids], g[ids] = subset(np.asarray(ids), cur[ids])
    else:
        v, g = eng.lml(cur, eval_gradient=True)
    return -v, -g


_SETULB_SIGNATURE = "setulb(m,x,l,u,nbd,f,g,factr,pgtol,wa,iwa,task,lsave,isave,dsave,maxls,ln_task)"
_setulb_ok = None


class _SetulbSearch:
    """One L-BFGS-B search as scipy.optimize._lbfgsb_py._minimize_lbfgsb runs it (scipy 1.15: m = 10, factr = 2.22e-9 / eps,
    pgtol = 1e-5, maxfun = maxiter = 15000, maxls = 20, x0 clipped into the bounds, the objective memoised on the latest x as
    scipy's ScalarFunction does), cut open at the objective call: advance() runs the routine until it wants f and g at a new x
    (returns True; feed(f, g) hands them over) or has finished (returns False; .x, .f are the result)."""

    def __init__(self, setulb, x0, bounds):
        n = x0.shape[0]
        m = 10
        self.setulb, self.m, self.maxls = setulb, m, 20
        self.factr, self.pgtol, self.maxfun, self.maxiter = 2.2204460492503131e-09 / np.finfo(float).eps, 1e-5, 15000, 15000
        self.low, self.up = np.array(bounds[:, 0], dtype=np.float64), np.array(bounds[:, 1], dtype=np.float64)
        self.nbd = np.full(n, 2, dtype=np.int32)                     # every hyper-parameter has both bounds (finite)
        self.x = np.array(np.clip(np.asarray(x0, dtype=np.float64).ravel(), self.low, self.up), dtype=np.float64)
        self.f = np.array(0.0, dtype=np.int32)                       # (scipy's own start values, types included)
        self.g = np.zeros((n,), dtype=np.int32)
        self.wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64)
        self.iwa = np.zeros(3 * n, dtype=np.int32)
        self.task, self.ln_task = np.zeros(2, dtype=np.int32), np.zeros(2, dtype=np.int32)
        self.lsave, self.isave, self.dsave = np.zeros(4, dtype=np.int32), np.zeros(44, dtype=np.int32), np.zeros(29, dtype=np.float64)
        self.n_iterations, self.nfev = 0, 0
        self._memo_x = None                                          # ScalarFunction evaluates at x0 when it is built:
        self.want = self.x.copy()                                    # the first request, before the routine is entered

    def feed(self, f, g):
        self._memo_x, self._memo = self.want, (f, np.atleast_1d(g))
        self.nfev += 1

    def advance(self):
        while True:
            self.g = self.g.astype(np.float64)
            self.setulb(self.m, self.x, self.low, self.up, self.nbd, self.f, self.g, self.factr, self.pgtol, self.wa, self.iwa,
                        self.task, self.lsave, self.isave, self.dsave, self.maxls, self.ln_task)
            if self.task[0] == 3:                                    # f and g at the current x
                if not np.array_equal(self.x, self._memo_x):
                    self.want = self.x.copy()
                    return True
                self.f, self.g = self._memo
            elif self.task[0] == 1:                                  # new iteration
                self.n_iterations += 1
                if self.n_iterations >= self.maxiter:
                    self.task[0], self.task[1] = 5, 504
                elif self.nfev > self.maxfun:
                    self.task[0], self.task[1] = 5, 502
            else:
                return False

    def resume(self):
        """after feed(): the routine asked for f and g at `want` = the current x"""
        self.f, self.g = self._memo
        return self.advance()


def _setulb_usable():
    """scipy's L-BFGS-B routine in the reverse-communication form this file knows (scipy 1.15's C port), PROVEN on a bounded test
    problem to give scipy.optimize.minimize's result bit for bit (x, fun, number of evaluations, every x it asked about);
    anything else — another signature, a different result — and the searches run as scipy.optimize.minimize threads"""
    global _setulb_ok
    if _setulb_ok is None:
        _setulb_ok = False
        try:
            if os.environ.get("GPB_LBFGSB_THREADS") == "1":
                return False
            from scipy.optimize import _lbfgsb_py
            setulb = _lbfgsb_py._lbfgsb.setulb
            if (setulb.__doc__ or "").strip() != _SETULB_SIGNATURE:
                return False

            def fg(x):          # a bounded, badly scaled problem that ends ON bounds, after line searches with several trials
                a = np.arange(1.0, x.shape[0] + 1.0)
                r = x - 0.3 * a
                return float(np.sum(a * r ** 2) + np.sum(np.cosh(0.5 * x)) + 5.0 * np.sin(x[0] * x[1])), \
                    2.0 * a * r + 0.5 * np.sinh(0.5 * x) + 5.0 * np.cos(x[0] * x[1]) * np.concatenate(([x[1], x[0]], np.zeros(x.shape[0] - 2)))
            bnd = np.array([[-1.0, 0.7], [-2.0, 2.0], [0.95, 3.0], [-4.0, 1.1], [0.0, 9.0]])
            for x0 in (np.array([0.5, -1.5, 2.5, -3.0, 8.0]), np.array([-5.0, 0.0, 0.0, 0.0, 1.0])):
                asked = []
                ref = scipy.optimize.minimize(lambda x: (asked.append(x.copy()), fg(x))[1], x0, method="L-BFGS-B", jac=True, bounds=bnd)
                s, mine = _SetulbSearch(setulb, x0, bnd), []
                more = True
                while more:
                    mine.append(s.want.copy())
                    s.feed(*fg(s.want))
                    more = s.resume() if len(mine) > 1 else s.advance()
                if not (np.array_equal(s.x, ref.x) and float(s.f) == float(ref.fun) and s.nfev == ref.nfev
                        and s.n_iterations == ref.nit and len(mine) == len(asked)
                        and all(np.array_equal(a, b) for a, b in zip(mine, asked))):
                    return False
            _setulb_ok = True
        except Exception:        # noqa: BLE001  (a private scipy interface: any surprise means "use the threads")
            _setulb_ok = False
    return _setulb_ok


def _lockstep_setulb(eng, start, bounds):
    from scipy.optimize import _lbfgsb_py
    setulb = _lbfgsb_py._lbfgsb.setulb
    P = start.shape[0]
    cur = np.array(start, dtype=np.float64)
    searches = [_SetulbSearch(setulb, start[p], bounds[p] if bounds.ndim == 3 else bounds) for p in range(P)]
    ids = list(range(P))
    first = True
    while ids:
        for p in ids:
            cur[p] = searches[p].want
        f, g = _evaluate_round(eng, cur, ids)
        nxt = []
        for p in ids:
            s = searches[p]
            s.feed(f[p], g[p])
            if (s.advance() if first else s.resume()):
                nxt.append(p)
        ids, first = nxt, False
    return np.array([s.x for s in searches]), np.array([float(s.f) for s in searches])


def _lockstep_threads(eng, start, bounds):
    """one scipy.optimize.minimize thread per search, meeting at every objective call (see _batched_lbfgsb)"""
    P = start.shape[0]
    cur = start.copy()
    results = [None] * P
    cond = threading.Condition()
    state = {"waiting": 0, "active": P, "round": 0, "val": None, "grad": None, "err": None}
    alive = np.ones(P, dtype=bool)

    def evaluate_round():
        try:
            state["val"], state["grad"] = _evaluate_round(eng, cur, np.flatnonzero(alive))
        except Exception as e:  # propagate to every waiting thread
            state["err"] = e
        state["waiting"] = 0
        state["round"] += 1
        cond.notify_all()

    def objective(p, th):
        with cond:
            cur[p] = th
            state["waiting"] += 1
            my_round = state["round"]
            if state["waiting"] == state["active"]:
                evaluate_round()
            else:
                while state["round"] == my_round:
                    cond.wait()
            if state["err"] is not None:
                raise state["err"]
            return state["val"][p], state["grad"][p]

    def worker(p):
        try:
            res = scipy.optimize.minimize(lambda th: objective(p, th), start[p], method="L-BFGS-B",
                                          jac=True, bounds=bounds[p] if bounds.ndim == 3 else bounds)
            results[p] = (res.x, res.fun)
        except Exception as e:
            results[p] = e
        finally:
            with cond:
                state["active"] -= 1
                alive[p] = False
                if state["active"] > 0 and state["waiting"] == state["active"]:
                    evaluate_round()

    threads = [threading.Thread(target=worker, args=(p,), daemon=True) for p in range(P)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for r in results:
        if isinstance(r, Exception):
            raise r
    return np.array([r[0] for r in results]), np.array([r[1] for r in results])
