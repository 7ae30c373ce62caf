"""parameterTrafoPCA=True pre-transform (src/emulator.py:79-241, 492-551) against the reference."""
import numpy as np
import pytest

from conftest import golden, maxrel, relerr


def test_parameter_pca_matches_reference_cpu():
    from gpbayestools_hic_amd.param_pca import ParameterPCA
    g = golden("g7_param_pca.npz")
    pp = ParameterPCA(g["X"], g["lo"], g["hi"])
    assert list(pp.n_components) == list(g["n_components"])
    assert pp.new_design_points.shape == g["new_design_points"].shape
    assert maxrel(pp.new_design_points, g["new_design_points"]) < 1e-10
    assert maxrel(pp.design_min, g["design_min"]) < 1e-10 and maxrel(pp.design_max, g["design_max"]) < 1e-10
    # transform of the training points reproduces the fitted PCs; single rows work too
    assert maxrel(pp.transform(g["X"]), pp.new_design_points) < 1e-12
    assert pp.transform(g["X"][3]).shape == (1, pp.new_design_points.shape[1])


def test_function_families_match_reference_branches():
    """vectorised zeta/s, eta/s, y_loss against the reference's scalar branch logic, restated here"""
    from gpbayestools_hic_amd import param_pca as pp
    rng = np.random.default_rng(1)
    par4 = np.column_stack([rng.uniform(0, .2, 5), rng.uniform(.13, .3, 5), rng.uniform(.01, .15, 5), rng.uniform(.01, .15, 5)])
    ref = np.array([[p[0] * np.exp(-(T - p[1]) ** 2 / (2 * (p[3] if T < p[1] else p[2]) ** 2)) for T in pp.T_GRID] for p in par4])
    assert maxrel(pp.zeta_over_s(par4), ref) < 1e-15
    par3 = rng.uniform(0.01, 0.3, (5, 3))
    def eta(e0, e2, e4, m):
        if 0. < m <= 0.2: return e0 + (e2 - e0) * (m / 0.2)
        if 0.2 < m < 0.4: return e2 + (e4 - e2) * ((m - 0.2) / 0.2)
        return e4
    assert maxrel(pp.eta_over_s(par3), np.array([[eta(*p, m) for m in pp.MUB_GRID] for p in par3])) < 1e-15
    def yl(y2, y4, y6, y):
        if 0. < y <= 2.: return y2 * (y / 2.)
        if 2. < y < 4.: return y2 + (y4 - y2) * ((y - 2.) / 2.)
        return y4 + (y6 - y4) * ((y - 4.) / 2.)
    assert maxrel(pp.y_loss(par3), np.array([[yl(*p, y) for y in pp.YINIT_GRID] for p in par3])) < 1e-15


@pytest.mark.gpu
def test_emulator_with_parameter_pca_gpu(tmp_path):
    from gpbayestools_hic_amd import Emulator, synth
    g = golden("g7_param_pca.npz")
    tp, pf = str(tmp_path / "t.pkl"), str(tmp_path / "p.txt")
    synth.write_training_pickle(tp, g["X"], g["Y"], 0.01)
    synth.write_parameter_file(pf, g["lo"], g["hi"])
    emu = Emulator(training_set_path=tp, parameter_file=pf, npc=int(g["npc"]), parameterTrafoPCA=True)
    assert maxrel(emu.PCA_new_design_points, g["new_design_points"]) < 1e-10
    emu.trainEmulator([True] * emu.nev, thetas=g["thetas"])
    assert relerr(emu.lml_, g["lml"]) < 1e-8
    mean, cov = emu.predict(g["Xs"], return_cov=True, extra_std=0.0)
    assert relerr(mean, g["mean"]) < 1e-8
    assert maxrel(cov, g["cov"]) < 1e-7
