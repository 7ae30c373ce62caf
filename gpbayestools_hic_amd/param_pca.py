"""
Host-side parameter-space PCA of the reference's `parameterTrafoPCA=True` option
(src/emulator.py:79-241 at construction, :492-551 inside predict).

Three groups of model parameters are replaced by the leading principal components (99 % of the
variance) of the functions they parametrise, evaluated on 100-point grids:
    zeta/s(T)      columns [15,16,17,18] = (zeta_max, T_zeta0, sigma_plus, sigma_minus)   :102-108
    eta/s(mu_B)    columns [12,13,14]    = (eta_0, eta_2, eta_4)                           :111-117
    y_loss(y_init) columns [2,3,4]       = (yloss_2, yloss_4, yloss_6)                     :120-126
in that order; each step deletes its columns and appends the PCs, so the GP input becomes
[remaining original parameters, bulk PCs, shear PCs, yloss PCs].

The reference evaluates these with Python double loops per prediction row; here they are
vectorised numpy (the pre-transform is O(W*100), negligible next to the GP work and kept on the
host as SURVEY §8 a6 allows).  Branch conditions reproduce the reference's strict/non-strict
inequalities, including its values at the grid end points.
"""
import numpy as np

from .preprocess import Standardizer, WhitenedPCA

IDX_BULK = [15, 16, 17, 18]
IDX_SHEAR = [12, 13, 14]
IDX_YLOSS = [2, 3, 4]
T_GRID = np.linspace(0.0, 0.5, 100)
MUB_GRID = np.linspace(0.0, 0.6, 100)
YINIT_GRID = np.linspace(0.0, 6.2, 100)
TARGET_VARIANCE = 0.99


def zeta_over_s(par, T=T_GRID, mu_B=0.0):
    """par[:, (zeta_max, T_zeta0, sigma_plus, sigma_minus)] -> [n, len(T)]  (src/emulator.py:102-108)"""
    zmax, T0, sp, sm = (par[:, i:i + 1] for i in range(4))
    Tmu = T0 - 0.15 * mu_B ** 2.0
    sig = np.where(T[None, :] < T0, sm, sp)
    return zmax * np.exp(-(T[None, :] - Tmu) ** 2.0 / (2.0 * sig ** 2.0))


def eta_over_s(par, mu_B=MUB_GRID):
    """par[:, (eta_0, eta_2, eta_4)] -> [n, len(mu_B)]  (src/emulator.py:111-117)"""
    e0, e2, e4 = (par[:, i:i + 1] for i in range(3))
    m = mu_B[None, :]
    first = (0.0 < m) & (m <= 0.2)
    second = (0.2 < m) & (m < 0.4)
    return np.where(first, e0 + (e2 - e0) * (m / 0.2),
                    np.where(second, e2 + (e4 - e2) * ((m - 0.2) / 0.2), e4 + 0.0 * m))


def y_loss(par, y_init=YINIT_GRID):
    """par[:, (yloss_2, yloss_4, yloss_6)] -> [n, len(y_init)]  (src/emulator.py:120-126)"""
    y2, y4, y6 = (par[:, i:i + 1] for i in range(3))
    y = y_init[None, :]
    first = (0.0 < y) & (y <= 2.0)
    second = (2.0 < y) & (y < 4.0)
    return np.where(first, y2 * (y / 2.0),
                    np.where(second, y2 + (y4 - y2) * ((y - 2.0) / 2.0), y4 + (y6 - y4) * ((y - 4.0) / 2.0)))


class _Group:
    def __init__(self, idx, fn):
        self.idx, self.fn = idx, fn
        self.scaler = Standardizer()
        self.pca = WhitenedPCA(n_components=TARGET_VARIANCE, whiten=False)

    def fit(self, params):
        self.pcs = self.pca.fit_transform(self.scaler.fit_transform(self.fn(params)))
        return self.pcs

    def transform(self, params):
        return self.pca.transform(self.scaler.transform(self.fn(params)))


class ParameterPCA:
    def __init__(self, design_points, design_min, design_max):
        X = np.asarray(design_points, dtype=np.float64)
        self.groups = [_Group(IDX_BULK, zeta_over_s), _Group(IDX_SHEAR, eta_over_s), _Group(IDX_YLOSS, y_loss)]
        new = X
        dmin, dmax = np.asarray(design_min, float), np.asarray(design_max, float)
        for g in self.groups:
            pcs = g.fit(X[:, g.idx])                     # always from the ORIGINAL columns
            new = np.concatenate((np.delete(new, g.idx, axis=1), pcs), axis=1)
            dmin = np.concatenate((np.delete(dmin, g.idx), pcs.min(axis=0)))
            dmax = np.concatenate((np.delete(dmax, g.idx), pcs.max(axis=0)))
        self.new_design_points = new
        self.design_min, self.design_max = dmin, dmax
        self.n_components = [g.pca.n_components_ for g in self.groups]

    @classmethod
    def from_fitted(cls, fitted, new_design_points, design_min, design_max):
        """The map as ANOTHER object fitted it: `fitted` = [(scaler, pca)] of the bulk, shear and y-loss groups (anything with
        sklearn's attributes: scaler.mean_ / scale_ / var_, pca.mean_ / components_ / explained_variance_ /
        explained_variance_ratio_ / n_components_) — a trained emulator of the reference taken over without refitting
        (Emulator.from_reference)."""
        self = cls.__new__(cls)
        self.groups = [_Group(IDX_BULK, zeta_over_s), _Group(IDX_SHEAR, eta_over_s), _Group(IDX_YLOSS, y_loss)]
        for g, (sc, pc) in zip(self.groups, fitted):
            for name in ("mean_", "scale_", "var_"):
                setattr(g.scaler, name, np.array(getattr(sc, name), dtype=np.float64))
            for name in ("mean_", "components_", "explained_variance_", "explained_variance_ratio_"):
                setattr(g.pca, name, np.array(getattr(pc, name), dtype=np.float64))
            g.pca.n_components_ = int(pc.n_components_)
        self.new_design_points = np.array(new_design_points, dtype=np.float64)
        self.design_min, self.design_max = np.array(design_min, dtype=np.float64), np.array(design_max, dtype=np.float64)
        self.n_components = [g.pca.n_components_ for g in self.groups]
        return self

    def transform(self, X):
        """X[W, ndim_original] -> GP input [W, ndim_reduced]  (src/emulator.py:492-549)"""
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        new = X
        for g in self.groups:
            new = np.concatenate((np.delete(new, g.idx, axis=1), g.transform(X[:, g.idx])), axis=1)
        return new
