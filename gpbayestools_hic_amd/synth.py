"""
Seeded synthetic inputs for tests and bench.py.

Design: Latin hypercube on the unit cube, x_ij = (pi_j(i) + u_ij) / N.
Observables: Y = 2 + sin(X W1) + 0.5 cos(X W2) + 0.01 eps.
File formats are the reference's own (src/emulator.py:378-415, src/mcmc.py:302-324,
src/__init__.py:21-33) so that the same files feed the reference and this package.
"""
import pickle

import numpy as np

SEED = 20250829

# cfg -> (N, d, M, P, kernel, W per log-prob call)   BASELINE.md §4
CONFIGS = {
    1: dict(N=128, d=8, M=4, P=4, kernel="RBF", W=64),
    2: dict(N=1024, d=15, M=32, P=10, kernel="RBF", W=10000),
    3: dict(N=1024, d=15, M=32, P=10, kernel="RBF", W=512),
    4: dict(N=2048, d=20, M=64, P=10, kernel="RBF", W=2048),
    5: dict(N=4096, d=20, M=64, P=10, kernel="Matern25", W=8192),
}


def lhs(N, d, seed=SEED, lo=None, hi=None):
    rng = np.random.default_rng(seed)
    X = np.empty((N, d))
    for j in range(d):
        X[:, j] = (rng.permutation(N) + rng.random(N)) / N
    if lo is not None:
        X = lo + (np.asarray(hi) - np.asarray(lo)) * X
    return X


def observables(X, M, seed=SEED + 1, noise=0.01):
    rng = np.random.default_rng(seed)
    d = X.shape[1]
    W1 = rng.standard_normal((d, M))
    W2 = rng.standard_normal((d, M))
    eps = rng.standard_normal((X.shape[0], M))
    from .preprocess import single_thread_blas
    with single_thread_blas():             # the same bits whatever the process's BLAS threading (ranks of a sharded bench)
        return 2.0 + np.sin(X @ W1) + 0.5 * np.cos(X @ W2) + noise * eps


def truth_point(d, seed=SEED + 2):
    return np.random.default_rng(seed).random(d)


def walkers(W, d, seed=SEED + 3, lo=None, hi=None):
    X = np.random.default_rng(seed).random((W, d))
    if lo is not None:
        X = lo + (np.asarray(hi) - np.asarray(lo)) * X
    return X


def walkers_ball(W, centre, radius=1e-3, seed=SEED + 4, lo=None, hi=None):
    """A burnt-in ensemble: walkers in a small Gaussian ball around `centre` (what the reference's run_mcmc holds after
    re-seeding at its best points, src/mcmc.py:392-405), kept strictly inside the box [lo, hi] (default: the unit cube)."""
    centre = np.asarray(centre, dtype=np.float64)
    d = centre.shape[0]
    lo = np.zeros(d) if lo is None else np.asarray(lo, dtype=np.float64)
    hi = np.ones(d) if hi is None else np.asarray(hi, dtype=np.float64)
    span = hi - lo
    X = centre + radius * span * np.random.default_rng(seed).standard_normal((W, d))
    return np.clip(X, lo + 1e-6 * span, hi - 1e-6 * span)


def fixed_theta(d, P, c=1.0, ell=1.5, noise=0.05):
    """Timing-run hyper-parameters: c=1, l_j=1.5, sigma_n^2=0.05."""
    th = np.concatenate([[np.log(c)], np.full(d, np.log(ell)), [np.log(noise)]])
    return np.tile(th, (P, 1))


# ---- reference file formats -------------------------------------------------
def write_parameter_file(path, lo, hi, names=None):
    """`name: label, min, max  # comment` (src/__init__.py:21-33)."""
    d = len(lo)
    names = names or [f"p{j}" for j in range(d)]
    with open(path, "w") as f:
        f.write("# synthetic parameter file\n")
        for n, a, b in zip(names, lo, hi):
            f.write(f"{n}: {n}, {float(a)!r}, {float(b)!r}\n")


def write_training_pickle(path, X, Y, Yerr):
    """{event_id:str(int) -> {"parameter": f64[ndim], "obs": f64[2, nobs]}}
    (src/emulator.py:384-407)."""
    if np.ndim(Yerr) == 0:
        Yerr = np.full_like(Y, float(Yerr))
    d = {str(i): {"parameter": np.array(X[i]), "obs": np.stack([Y[i], Yerr[i]])}
         for i in range(X.shape[0])}
    with open(path, "wb") as f:
        pickle.dump(d, f)


def write_experiment_pickle(path, y, yerr):
    """One event; covariance = diag(err^2) (src/mcmc.py:307-322)."""
    with open(path, "wb") as f:
        pickle.dump({"0": {"parameter": np.zeros(1), "obs": np.stack([y, yerr])}}, f)
