"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.

CPU restatement (numpy + scipy.linalg only; no scikit-learn import) of the one
hot path of Hendrik1704/GPBayesTools-HIC:

    Emulator.trainEmulator -> GPR.fit            (K build, Cholesky, alpha, LML + grad)
    Chain.log_posterior -> Chain._predict -> Emulator.predict -> GPR.predict -> mvn_loglike

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker / the timed CPU baseline.

Citation convention (same as SURVEY.md):
    src/x.py:L        = /root/reference/src/x.py line L
    sk:_gpr.py:L      = scikit-learn 1.7.2 sklearn/gaussian_process/_gpr.py line L
    sk:kernels.py:L   = scikit-learn 1.7.2 sklearn/gaussian_process/kernels.py line L
scikit-learn is a third-party dependency of the reference (requirements.txt:9,
`scikit-learn>=1.4.0`), not vendored; its published algorithm (Rasmussen &
Williams Alg. 2.1) is restated here.

Pinning: the reference has no tests or golden vectors of its own (SURVEY.md §4,
§8c).  This restatement is pinned against outputs of the reference itself run in
the build container: tools/make_goldens.py imports /root/reference/src and writes
tests/golden/*.npz; tests/test_oracle_golden.py checks every function here
against those vectors.

theta layout everywhere: [log c, log l_1 .. log l_d, log sigma_n^2]
(sk:kernels.py Sum(Product(ConstantKernel, RBF|Matern), WhiteKernel).theta).
"""
import math

import numpy as np
from scipy.linalg import cholesky, cho_solve, solve_triangular, lapack
from scipy.spatial.distance import cdist, pdist, squareform

KIND_RBF = 0        # sk:kernels.py:1525-1575
KIND_MATERN15 = 1   # sk:kernels.py:1721-1723 (nu=1.5), the reference's "Matern" (src/emulator.py:292-297)
KIND_MATERN25 = 2   # sk:kernels.py:1724-1726 (nu=2.5), BASELINE cfg 5
KIND_NAMES = {"RBF": KIND_RBF, "Matern": KIND_MATERN15, "Matern15": KIND_MATERN15,
              "Matern25": KIND_MATERN25}

# constant the reference adds inside the box because extra_std == 0*X[:, -1]
# (src/mcmc.py:281,296-297 and :205,220-221):  2*log(0 + 1e-16) - 0
EXTRA_STD_CONST = 2.0 * math.log(1e-16)


# --------------------------------------------------------------------------- kernels
def _unpack(theta, d):
    theta = np.asarray(theta, dtype=np.float64)
    assert theta.shape == (d + 2,)
    c = math.exp(theta[0])
    ls = np.exp(theta[1:1 + d])
    noise = math.exp(theta[d + 1])
    return c, ls, noise


def _shape_fn(kind, r2=None, r=None):
    """k(r) with unit amplitude.  r2 = squared scaled distance, r = scaled distance."""
    if kind == KIND_RBF:
        return np.exp(-0.5 * r2)                                   # sk:kernels.py:1557,1565
    if kind == KIND_MATERN15:
        t = r * math.sqrt(3)
        return (1.0 + t) * np.exp(-t)                              # sk:kernels.py:1721-1723
    if kind == KIND_MATERN25:
        t = r * math.sqrt(5)
        return (1.0 + t + t ** 2 / 3.0) * np.exp(-t)               # sk:kernels.py:1724-1726
    raise ValueError(kind)


def kernel_cross(Xs, X, theta, kind=KIND_RBF):
    """K(X*, X) = c * k(|(x*-x)/l|); the WhiteKernel term is zero when Y is given
    (sk:kernels.py:1413-1414).  Returns [W, N]."""
    d = X.shape[1]
    c, ls, _ = _unpack(theta, d)
    if kind == KIND_RBF:
        r2 = cdist(Xs / ls, X / ls, metric="sqeuclidean")          # sk:kernels.py:1564
        return c * _shape_fn(kind, r2=r2)
    r = cdist(Xs / ls, X / ls, metric="euclidean")                 # sk:kernels.py:1717
    return c * _shape_fn(kind, r=r)


def kernel_train(X, theta, kind=KIND_RBF, alpha=0.0):
    """K(X, X) = c*k + sigma_n^2 I (+ alpha I as GPR.fit adds, sk:_gpr.py:347).
    Diagonal of the stationary part is forced to exactly 1 before scaling
    (sk:kernels.py:1559-1560, 1737-1738)."""
    N, d = X.shape
    c, ls, noise = _unpack(theta, d)
    if kind == KIND_RBF:
        K = squareform(_shape_fn(kind, r2=pdist(X / ls, metric="sqeuclidean")))
    else:
        K = squareform(_shape_fn(kind, r=pdist(X / ls, metric="euclidean")))
    np.fill_diagonal(K, 1.0)
    K *= c
    K[np.diag_indices_from(K)] += noise                            # sk:kernels.py:1401-1412
    K[np.diag_indices_from(K)] += alpha
    return K


def kernel_train_grad(X, theta, kind=KIND_RBF):
    """dK/dtheta_k, shape [N, N, d+2] (sk:kernels.py Sum/Product eval_gradient:
    833-866, 931-966; ConstantKernel 1276-1289; RBF 1574-1580; Matern 1744-1766;
    WhiteKernel 1401-1410)."""
    N, d = X.shape
    c, ls, noise = _unpack(theta, d)
    D = (X[:, None, :] - X[None, :, :]) ** 2 / ls ** 2            # [N,N,d]
    r2 = D.sum(-1)
    G = np.empty((N, N, d + 2))
    if kind == KIND_RBF:
        Ks = np.exp(-0.5 * r2)
        np.fill_diagonal(Ks, 1.0)
        G[:, :, 1:1 + d] = c * D * Ks[..., None]
    elif kind == KIND_MATERN15:
        r = np.sqrt(r2)
        Ks = _shape_fn(kind, r=r)
        np.fill_diagonal(Ks, 1.0)
        G[:, :, 1:1 + d] = c * 3.0 * D * np.exp(-np.sqrt(3.0 * r2))[..., None]
    elif kind == KIND_MATERN25:
        r = np.sqrt(r2)
        Ks = _shape_fn(kind, r=r)
        np.fill_diagonal(Ks, 1.0)
        t = np.sqrt(5.0 * r2)[..., None]
        G[:, :, 1:1 + d] = c * (5.0 / 3.0) * D * (t + 1.0) * np.exp(-t)
    else:
        raise ValueError(kind)
    G[:, :, 0] = c * Ks
    G[:, :, d + 1] = noise * np.eye(N)
    return G


# --------------------------------------------------------------------------- GP fit / LML
def gp_factor(X, z, theta, kind=KIND_RBF, alpha=0.1):
    """Final block of GPR.fit (sk:_gpr.py:346-364): K=k(X,X); K_ii+=alpha;
    L=chol(K) lower; alpha_=K^-1 z.  Raises numpy.linalg.LinAlgError if not PD."""
    K = kernel_train(X, theta, kind, alpha)
    L = cholesky(K, lower=True, check_finite=False)
    a = cho_solve((L, True), z, check_finite=False)
    return L, a


def lml(theta, X, z, kind=KIND_RBF, alpha=0.1, eval_gradient=False):
    """GPR.log_marginal_likelihood (sk:_gpr.py:537-652).  Non-PD -> (-inf, 0)."""
    N, d = X.shape
    K = kernel_train(X, theta, kind, alpha)
    try:
        L = cholesky(K, lower=True, check_finite=False)
    except np.linalg.LinAlgError:
        return (-np.inf, np.zeros(d + 2)) if eval_gradient else -np.inf
    a = cho_solve((L, True), z, check_finite=False)
    val = -0.5 * float(z @ a) - np.log(np.diag(L)).sum() - N / 2.0 * math.log(2 * math.pi)
    if not eval_gradient:
        return val
    Kinv = cho_solve((L, True), np.eye(N), check_finite=False)
    inner = np.outer(a, a) - Kinv
    G = kernel_train_grad(X, theta, kind)
    grad = 0.5 * np.einsum("ij,jik->k", inner, G)
    return val, grad


def default_theta0_bounds(design_min, design_max, kind=KIND_RBF):
    """Initial theta and log-bounds of the kernel the reference builds
    (src/emulator.py:286-306): 1.*RBF(ptp, ptp x (1e-1,1e2)) + White(.05,(1e-2,1e2));
    Matern: ptp x (1e-3,1e5), nu=1.5.  ConstantKernel default bounds (1e-5,1e5)
    (sk:kernels.py:1215)."""
    ptp = np.asarray(design_max, float) - np.asarray(design_min, float)
    d = ptp.shape[0]
    lo, hi = (1e-1, 1e2) if kind == KIND_RBF else (1e-3, 1e5)
    theta0 = np.concatenate([[0.0], np.log(ptp), [math.log(0.05)]])
    bounds = np.empty((d + 2, 2))
    bounds[0] = np.log([1e-5, 1e5])
    bounds[1:1 + d, 0] = np.log(ptp * lo)
    bounds[1:1 + d, 1] = np.log(ptp * hi)
    bounds[d + 1] = np.log([1e-2, 1e2])
    return theta0, bounds


def gp_fit_theta(X, z, theta0, bounds, kind=KIND_RBF, alpha=0.1):
    """Hyper-parameter search of GPR.fit with n_restarts_optimizer=0
    (sk:_gpr.py:296-337, 654-670): scipy L-BFGS-B on -LML with analytic gradient."""
    import scipy.optimize

    def obj(th):
        v, g = lml(th, X, z, kind, alpha, eval_gradient=True)
        return -v, -g

    res = scipy.optimize.minimize(obj, theta0, method="L-BFGS-B", jac=True, bounds=bounds)
    return res.x, -res.fun


def learning_curve(X, Z, design_min, design_max):
    """`Emulator.print_learning_curve` (src/emulator.py:424-462) for the GP targets Z[N, npc] over the design X[N, d]: sklearn's
    `learning_curve` with its defaults — KFold(5) without shuffling (contiguous test blocks, the first N % 5 one longer), train
    sizes (0.2, 0.4, 0.6, 0.8, 0.9) x the first fold's training count truncated to int, the FIRST n training events of a fold —
    over GPR(1. * RBF(ptp, ptp x (.01, 100)) + WhiteKernel(.01**2, (.001**2, 1)), alpha=0.) fits (one L-BFGS-B search each),
    scored by R^2 (GPR.score) on the n training events and on the held-out fold.  Returns [npc, sizes, (n, mean train, mean test)]."""
    X, Z = np.asarray(X, float), np.asarray(Z, float)
    N, d = X.shape
    ptp = np.asarray(design_max, float) - np.asarray(design_min, float)
    theta0 = np.concatenate([[0.0], np.log(ptp), [math.log(.01 ** 2)]])
    bounds = np.vstack([np.log([[1e-5, 1e5]]), np.log(np.outer(ptp, (.01, 100))), np.log([[.001 ** 2, 1]])])
    fold = np.full(5, N // 5)
    fold[:N % 5] += 1
    stops = np.cumsum(fold)
    folds = [(np.r_[0:b - m, b:N], np.arange(b - m, b)) for b, m in zip(stops, fold)]
    n_max = len(folds[0][0])
    sizes = np.unique(np.clip((np.array([0.2, 0.4, 0.6, 0.8, 0.9]) * n_max).astype(int), 1, n_max))
    r2 = lambda y, m: 1.0 - ((y - m) ** 2).sum() / ((y - y.mean()) ** 2).sum()
    out = np.empty((Z.shape[1], len(sizes), 3))
    for i in range(Z.shape[1]):
        for si, n in enumerate(sizes):
            tr_s, te_s = [], []
            for tr, te in folds:
                r = tr[:n]
                th, _ = gp_fit_theta(X[r], Z[r, i], theta0, bounds, KIND_RBF, 0.0)
                L, a = gp_factor(X[r], Z[r, i], th, KIND_RBF, 0.0)
                tr_s.append(r2(Z[r, i], kernel_cross(X[r], X[r], th) @ a))
                te_s.append(r2(Z[te, i], kernel_cross(X[te], X[r], th) @ a))
            out[i, si] = n, np.mean(tr_s), np.mean(te_s)
    return out


# --------------------------------------------------------------------------- GP predict
def prior_var(theta, d):
    """diag k(x*,x*) for the composite kernel: c*1 + sigma_n^2 (White adds on the
    diagonal because Y is None there, sk:kernels.py:1401-1412; no `alpha`)."""
    c, _, noise = _unpack(theta, d)
    return c + noise


def gp_predict(Xs, X, theta, L, a, kind=KIND_RBF):
    """Lean form of GPR.predict(return_cov=True) + diagonal (sk:_gpr.py:441-469,
    src/emulator.py:573-575): mean = K* a ; var = prior - sum_j V_j^2,
    V = L^-1 K*^T.  No clipping of negative variances."""
    Ks = kernel_cross(Xs, X, theta, kind)
    mean = Ks @ a
    V = solve_triangular(L, Ks.T, lower=True, check_finite=False)
    var = prior_var(theta, X.shape[1]) - np.einsum("ij,ij->j", V, V)
    return mean, var


def gp_predict_cov(Xs, X, theta, L, a, kind=KIND_RBF):
    """GPR.predict(return_cov=True) (sk:_gpr.py:441-469): mean and the full W x W covariance."""
    Ks = kernel_cross(Xs, X, theta, kind)
    mean = Ks @ a
    V = solve_triangular(L, Ks.T, lower=True, check_finite=False)
    d = X.shape[1]
    c, _, noise = _unpack(theta, d)
    Kss = kernel_cross(Xs, Xs, theta, kind)
    np.fill_diagonal(Kss, c)
    Kss[np.diag_indices_from(Kss)] += noise
    return mean, Kss - V.T @ V


def gp_predict_faithful(Xs, X, theta, L, a, kind=KIND_RBF):
    """What the reference really executes per GP: the full W x W covariance
    (sk:_gpr.py:460) and then its diagonal (src/emulator.py:573-575)."""
    Ks = kernel_cross(Xs, X, theta, kind)
    mean = Ks @ a
    V = solve_triangular(L, Ks.T, lower=True, check_finite=False)
    Kss = kernel_cross(Xs, Xs, theta, kind)
    d = X.shape[1]
    c, _, noise = _unpack(theta, d)
    np.fill_diagonal(Kss, c)            # k(X*) with Y=None forces the unit diagonal
    Kss[np.diag_indices_from(Kss)] += noise
    cov = Kss - V.T @ V
    return mean, cov.diagonal().copy()


# --------------------------------------------------------------------------- emulator (scaler + PCA + transforms)
def standardize_fit(Y):
    """StandardScaler.fit (src/emulator.py:76,260): population variance; zero
    scale -> 1."""
    mean = Y.mean(axis=0)
    var = Y.var(axis=0)
    scale = np.sqrt(var)
    scale[scale == 0.0] = 1.0
    return mean, scale, var


def pca_whiten_fit(S):
    """PCA(whiten=True, svd_solver='full').fit_transform (src/emulator.py:77,270):
    centre, LAPACK SVD, v-based sign flip, Z = U*sqrt(n-1)."""
    from scipy import linalg
    n = S.shape[0]
    mean = S.mean(axis=0)
    U, s, Vt = linalg.svd(S - mean, full_matrices=False)
    idx = np.argmax(np.abs(Vt), axis=1)
    signs = np.sign(Vt[np.arange(Vt.shape[0]), idx])
    U = U * signs[None, :]
    Vt = Vt * signs[:, None]
    ev = s ** 2 / (n - 1)
    Z = U * math.sqrt(n - 1)
    return Z, Vt, ev, mean


def emulator_transforms(components, explained_variance, scale, var, npc):
    """_trans_matrix / _var_trans / _cov_trunc (src/emulator.py:335-363)."""
    nobs = components.shape[1]
    T = components * np.sqrt(explained_variance[:, None]) * scale
    A = T[:npc]
    var_trans = np.einsum("ki,kj->kij", A, A).reshape(npc, nobs ** 2)
    B = T[npc:]
    cov_trunc = B.T @ B
    cov_trunc.flat[::nobs + 1] += 1e-4 * var
    return T, var_trans, cov_trunc


MODE_PCA = 0        # default
MODE_NO_PCA = 1     # perform_no_PCA=True
MODE_EXPDIAG = 2    # logTrafo=True + exp_and_cov_diagonal=True (on top of PCA)
MODE_NO_PCA_EXPDIAG = 3


def emulator_predict(gp_mean, gp_var, extra_std, *, mode, A=None, mu=None,
                     cov_trunc=None, scale=None, return_cov=True):
    """Emulator.predict after the per-GP calls (src/emulator.py:555-605).
    gp_mean, gp_var: [W, P].  PCA modes: A=[P,M] (_trans_matrix[:npc]), mu=scaler.mean_.
    no-PCA modes: scale, mu = scaler.scale_, scaler.mean_; cov=diag(gp_var) is NOT
    rescaled (src/emulator.py:589-592)."""
    W, P = gp_mean.shape
    no_pca = mode in (MODE_NO_PCA, MODE_NO_PCA_EXPDIAG)
    expdiag = mode in (MODE_EXPDIAG, MODE_NO_PCA_EXPDIAG)
    if not no_pca:
        mean = gp_mean @ A + mu                                     # :559-561, 373-374
    else:
        mean = gp_mean * scale + mu                                 # :563-565
    if expdiag:
        mean = np.exp(mean)                                         # :567-568
    if not return_cov:
        return mean
    gv = gp_var + np.asarray(extra_std, float).reshape(-1, 1) ** 2  # :578-579
    M = mean.shape[1]
    if not no_pca:
        vt = np.einsum("ki,kj->kij", A, A).reshape(P, M * M)
        cov = (gv @ vt).reshape(W, M, M) + cov_trunc                # :584-587
    else:
        cov = np.zeros((W, M, M))
        for i in range(W):
            cov[i] = np.diag(gv[i])                                 # :590-592
    if expdiag:
        for i in range(W):                                          # :594-601
            fstd = np.sqrt(np.diag(cov[i]))
            cov[i] = np.diag((fstd * mean[i]) ** 2)
    return mean, cov


# --------------------------------------------------------------------------- likelihood / posterior
def mvn_loglike(y, cov):
    """src/mcmc.py:23-65: dpotrf (upper, clean=False) -> dpotrs -> -y.a/2 - sum log diag.
    The reference's non-PD branch is dead code (:44-54); here info>0 is reported
    as NaN so that callers can see it."""
    L, info = lapack.dpotrf(cov, clean=False)
    if info != 0:
        return float("nan")
    a, info = lapack.dpotrs(L, y)
    return -0.5 * float(np.dot(y, a)) - np.log(L.diagonal()).sum()


def mvn_loglike_batched(dY, cov):
    """Lean batched form of map(mvn_loglike, dY, cov) (src/mcmc.py:293)."""
    Lc = np.linalg.cholesky(cov)
    v = np.linalg.solve(Lc, dY[..., None])[..., 0]   # general solve on a triangle: fine for an oracle
    return -0.5 * np.einsum("wi,wi->w", v, v) - np.log(np.diagonal(Lc, axis1=1, axis2=2)).sum(-1)


def inside_box(X, lo, hi):
    """strict inequalities (src/mcmc.py:182,194,275)."""
    return np.all((X > lo) & (X < hi), axis=1)


def log_prior(X, lo, hi):
    """Chain.log_prior (src/mcmc.py:169-185)."""
    X = np.array(X, ndmin=2, dtype=float)
    lp = np.log(np.ones(X.shape[0]) / np.prod(hi - lo))
    lp[~inside_box(X, lo, hi)] = -np.inf
    return lp


def log_prob(X, lo, hi, predict_fn, yexp, cov_exp, *, finite=False, posterior=True,
             batched=True):
    """Chain.log_posterior (src/mcmc.py:261-299; posterior=True, finite ignored) and
    Chain.log_likelihood (src/mcmc.py:188-222; posterior=False).
    predict_fn(X_inside, extra_std_arr) -> (mean[w,M], cov[w,M,M]) is Chain._predict
    (src/mcmc.py:153-166)."""
    X = np.array(X, ndmin=2, dtype=float)
    lp = np.zeros(X.shape[0])
    inside = inside_box(X, lo, hi)
    lp[~inside] = -1e300 if (finite and not posterior) else -np.inf
    if np.count_nonzero(inside) > 0:
        extra_std = 0.0 * X[inside, -1]
        mY, mC = predict_fn(X[inside], extra_std)
        dY = mY - yexp
        cov = mC + cov_exp
        if batched:
            lp[inside] += mvn_loglike_batched(dY, cov)
        else:
            lp[inside] += list(map(mvn_loglike, dY, cov))
        lp[inside] += EXTRA_STD_CONST
    return lp


# --------------------------------------------------------------------------- whole-emulator convenience (oracle-side "Emulator")
class OracleEmulator:
    """Plain-array restatement of a trained reference Emulator (src/emulator.py).
    Holds P GPs (X, theta_p, L_p, alpha_p) and the observable transform."""

    def __init__(self, X, Ydata, design_min, design_max, npc, kind=KIND_RBF,
                 mode=MODE_PCA, alpha=0.1):
        self.X = np.ascontiguousarray(X, float)
        self.kind, self.mode, self.alpha_reg = kind, mode, alpha
        self.design_min, self.design_max = np.asarray(design_min, float), np.asarray(design_max, float)
        self.mu, self.scale, self.var = standardize_fit(Ydata)
        S = (Ydata - self.mu) / self.scale
        self.nobs = Ydata.shape[1]
        if mode in (MODE_NO_PCA, MODE_NO_PCA_EXPDIAG):
            self.Z = S
            self.npc = self.nobs
            self.A = self.cov_trunc = None
        else:
            Zfull, comps, ev, _ = pca_whiten_fit(S)
            self.npc = npc
            self.Z = Zfull[:, :npc]
            self.components, self.explained_variance = comps, ev
            T, _, self.cov_trunc = emulator_transforms(comps, ev, self.scale, self.var, npc)
            self.trans_matrix = T
            self.A = T[:npc]
        self.thetas = None

    def fit(self, thetas=None):
        """thetas=None -> optimise like the reference; else fit at the given theta[P,d+2]."""
        P = self.npc
        th0, bnds = default_theta0_bounds(self.design_min, self.design_max, self.kind)
        self.thetas, self.L, self.a, self.lml_ = [], [], [], []
        for p in range(P):
            z = np.ascontiguousarray(self.Z[:, p])
            if thetas is None:
                th, _ = gp_fit_theta(self.X, z, th0, bnds, self.kind, self.alpha_reg)
            else:
                th = np.asarray(thetas[p], float)
            L, a = gp_factor(self.X, z, th, self.kind, self.alpha_reg)
            self.thetas.append(th); self.L.append(L); self.a.append(a)
            self.lml_.append(lml(th, self.X, z, self.kind, self.alpha_reg))
        self.thetas = np.array(self.thetas)
        return self

    def gp_predict(self, Xs, faithful=False):
        f = gp_predict_faithful if faithful else gp_predict
        out = [f(Xs, self.X, self.thetas[p], self.L[p], self.a[p], self.kind) for p in range(self.npc)]
        m = np.stack([o[0] for o in out], axis=1)
        v = np.stack([o[1] for o in out], axis=1)
        return m, v

    def predict(self, Xs, return_cov=True, extra_std=0.0, faithful=False):
        Xs = np.ascontiguousarray(Xs, float)
        m, v = self.gp_predict(Xs, faithful)
        es = np.broadcast_to(np.asarray(extra_std, float).reshape(-1), (Xs.shape[0],)) \
            if np.ndim(extra_std) else np.full(Xs.shape[0], float(extra_std))
        return emulator_predict(m, v, es, mode=self.mode, A=self.A, mu=self.mu,
                                cov_trunc=self.cov_trunc, scale=self.scale,
                                return_cov=return_cov)


def chain_predict(emus, X, extra_std_arr, faithful=False):
    """Chain._predict (src/mcmc.py:153-166): concatenated means, block-diagonal cov."""
    W = X.shape[0]
    M = sum(e.nobs for e in emus)
    mean = np.zeros((W, M)); cov = np.zeros((W, M, M))
    i0 = 0
    for e in emus:
        m, c = e.predict(X, True, extra_std_arr, faithful=faithful)
        n = m.shape[1]
        mean[:, i0:i0 + n] = m
        cov[:, i0:i0 + n, i0:i0 + n] = c
        i0 += n
    return mean, cov
