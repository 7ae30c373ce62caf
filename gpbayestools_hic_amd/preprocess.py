"""
Host-side preprocessing of the training observables (one N x M SVD; SURVEY §8 a2 keeps
this on the host): standardisation, whitened PCA, and the PC -> observable transform arrays.
numpy/scipy only — scikit-learn is not needed at run time.
"""
import contextlib
import math

import numpy as np
from scipy import linalg

_blas_ctl = None


@contextlib.contextmanager
def single_thread_blas():
    """The host linear algebra of a training — one N x M SVD, two small products — on ONE BLAS thread.  Two reasons: (i) a threaded
    LAPACK rounds differently for every thread count, so two processes that train on the same data (the ranks of a sharded run:
    torch.distributed.run gives its ranks OMP_NUM_THREADS=1, a notebook has all cores) would disagree in the last bits of the GP
    targets, and with them in every log-probability — on one thread every process computes the same bits; (ii) it is faster at
    these sizes (1000 x 60: 1.2 ms on one thread against 3.4 on 64, tools/micro/svd_threads.py).  threadpoolctl when it is
    installed (it ships with scikit-learn, which the reference needs anyway); otherwise the caller's threading stays."""
    global _blas_ctl
    if _blas_ctl is None:
        try:
            from threadpoolctl import ThreadpoolController
            _blas_ctl = ThreadpoolController()
        except Exception:                  # not installed: nothing to pin
            _blas_ctl = False
    if _blas_ctl is False:
        yield
        return
    with _blas_ctl.limit(limits=1, user_api="blas"):
        yield


class Standardizer:
    """mean_/scale_/var_ as sklearn's StandardScaler exposes them (src/emulator.py:76,260)."""

    def fit_transform(self, Y):
        Y = np.asarray(Y, dtype=np.float64)
        self.mean_ = Y.mean(axis=0)
        self.var_ = Y.var(axis=0)
        self.scale_ = np.sqrt(self.var_)
        # (numerically) constant columns keep their values: scale 1, as sklearn does
        n, eps = Y.shape[0], np.finfo(np.float64).eps
        constant = self.var_ <= n * eps * self.var_ + (n * self.mean_ * eps) ** 2
        self.scale_[constant | (self.scale_ == 0.0)] = 1.0
        return (Y - self.mean_) / self.scale_

    def transform(self, Y):
        return (np.asarray(Y, dtype=np.float64) - self.mean_) / self.scale_

    def inverse_transform(self, S):
        return np.asarray(S) * self.scale_ + self.mean_


class WhitenedPCA:
    """Full-SVD PCA with whitening (src/emulator.py:77,270).  n_components: None (all),
    an int, or a float in (0,1) = smallest number of PCs reaching that explained-variance
    ratio (src/emulator.py:85)."""

    def __init__(self, n_components=None, whiten=True):
        self.n_components = n_components
        self.whiten = whiten

    def fit_transform(self, S):
        S = np.asarray(S, dtype=np.float64)
        n = S.shape[0]
        self.mean_ = S.mean(axis=0)
        with single_thread_blas():
            U, s, Vt = linalg.svd(S - self.mean_, full_matrices=False)
        # deterministic signs: largest-magnitude entry of every component is positive
        idx = np.argmax(np.abs(Vt), axis=1)
        sg = np.sign(Vt[np.arange(Vt.shape[0]), idx])
        sg[sg == 0] = 1.0
        U *= sg[None, :]
        Vt *= sg[:, None]
        ev = s ** 2 / (n - 1)
        ratio = ev / ev.sum()
        k = Vt.shape[0]
        if isinstance(self.n_components, float) and 0 < self.n_components < 1:
            k = int(np.searchsorted(np.cumsum(ratio), self.n_components, side="right") + 1)
        elif self.n_components is not None:
            k = int(self.n_components)
        self.n_components_ = k
        self.components_ = Vt[:k]
        self.explained_variance_ = ev[:k]
        self.explained_variance_ratio_ = ratio[:k]
        return U[:, :k] * (math.sqrt(n - 1) if self.whiten else s[:k])

    def fit(self, S):
        self.fit_transform(S)
        return self

    def transform(self, S):
        Z = (np.asarray(S, dtype=np.float64) - self.mean_) @ self.components_.T
        if self.whiten:
            Z = Z / np.sqrt(self.explained_variance_)
        return Z


def observable_transform(components, explained_variance, scale, var, npc):
    """_trans_matrix, A=_trans_matrix[:npc], _cov_trunc (src/emulator.py:335-363)."""
    nobs = components.shape[1]
    T = components * np.sqrt(explained_variance)[:, None] * scale
    A = np.ascontiguousarray(T[:npc])
    B = T[npc:]
    with single_thread_blas():
        cov_trunc = B.T @ B
    cov_trunc[np.diag_indices(nobs)] += 1e-4 * var
    return T, A, cov_trunc


def parse_model_parameter_file(parfile):
    """`name: label, min, max  # comment` -> {name: [label, min, max]} (src/__init__.py:21-33)."""
    pardict = {}
    with open(parfile, "r") as f:
        for line in f:
            body = line.split("#", 1)[0]
            if body.strip() == "":
                continue
            key, rest = body.split(":", 1)
            fields = [s.strip() for s in rest.split(",")]
            pardict[key] = [fields[0], float(fields[1]), float(fields[2])] + fields[3:]
    return pardict
