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


def test_emulator_exposes_reference_parameter_pca_surface(tmp_path):
    """attribute and method names notebooks use on a parameterTrafoPCA emulator (src/emulator.py:79-126);
    construction needs no device"""
    from gpbayestools_hic_amd import Emulator, synth
    from gpbayestools_hic_amd import param_pca as pp
    g = golden("g7_param_pca.npz")
    tp, pf = str(tmp_path / "t.pkl"), str(tmp_path / "p.txt")
    synth.write_training_pickle(tp, g["X"], g["Y"], 0.01)
    synth.write_parameter_file(pf, g["lo"], g["hi"])
    emu = Emulator(training_set_path=tp, parameter_file=pf, npc=int(g["npc"]), parameterTrafoPCA=True)
    assert emu.targetVariance == 0.99
    assert emu.indices_zeta_s_parameters == [15, 16, 17, 18] and emu.indices_eta_s_parameters == [12, 13, 14]
    assert emu.indices_yloss_parameters == [2, 3, 4]
    assert [emu.paramTrafoPCA_bulk.n_components_, emu.paramTrafoPCA_shear.n_components_,
            emu.paramTrafoPCA_yloss.n_components_] == list(g["n_components"])
    assert emu.paramTrafoScaler_bulk.mean_.shape == (100,)
    # scalar parametrisations agree with the vectorised grid functions, branch by branch
    z_lo = emu.parametrization_zeta_over_s_vs_T(0.1, 0.2, 0.05, 0.03, 0.15, 0.0)
    assert z_lo == pp.zeta_over_s(np.array([[0.1, 0.2, 0.05, 0.03]]), T=np.array([0.15]))[0, 0]
    z_hi = emu.parametrization_zeta_over_s_vs_T(0.1, 0.2, 0.05, 0.03, 0.25, 0.3)
    assert abs(z_hi - 0.1 * np.exp(-(0.25 - (0.2 - 0.15 * 0.3 ** 2)) ** 2 / (2 * 0.05 ** 2))) < 1e-16
    for m, want in ((0.1, 0.1 + (0.2 - 0.1) * 0.5), (0.3, 0.2 + (0.05 - 0.2) * ((0.3 - 0.2) / 0.2)), (0.5, 0.05), (0.0, 0.05)):
        assert abs(emu.parametrization_eta_over_s_vs_mu_B(0.1, 0.2, 0.05, m) - want) < 1e-16
    for y, want in ((1.0, 0.5), (3.0, 1.0 + (2.0 - 1.0) * 0.5), (5.0, 2.0 + (0.5 - 2.0) * 0.5)):
        assert abs(emu.parametrization_y_loss_vs_y_init(1.0, 2.0, 0.5, y) - want) < 1e-15


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
    assert relerr(emu.lml_, g["lml"]) < 1e-10         # measured 1e-15 (tests/diag_param_pca_errors.py)
    mean, cov = emu.predict(g["Xs"], return_cov=True, extra_std=0.0)
    assert relerr(mean, g["mean"]) < 1e-11            # measured 6e-15
    assert maxrel(cov, g["cov"]) < 1e-10              # measured 1.4e-15
    eng = emu._engine_ready()
    if eng.has_variants:                              # debug build (GPB_DEBUG_LIB=1): the difference form of the distance
        eng.tune("kcross_dot", 0)                     # holds the bars too
        mean_d, cov_d = emu.predict(g["Xs"], return_cov=True, extra_std=0.0)
        assert relerr(mean_d, g["mean"]) < 1e-11 and maxrel(cov_d, g["cov"]) < 1e-10
        eng.tune("kcross_dot", 1)


@pytest.mark.gpu
def test_chain_with_parameter_pca_uses_device_map(tmp_path):
    """Chain.log_posterior over a parameterTrafoPCA emulator: the device pre-pass (gpb_param_map) feeds the GP
    kernels, the prior box stays on the original parameters (src/emulator.py:492-551, src/mcmc.py:261-299)."""
    import torch
    from oracle import gp_oracle as O
    from gpbayestools_hic_amd import Chain, Emulator, synth
    g = golden("g7_param_pca.npz")
    tp, pf, ep = str(tmp_path / "t.pkl"), str(tmp_path / "p.txt"), str(tmp_path / "e.pkl")
    synth.write_training_pickle(tp, g["X"], g["Y"], 0.01)
    synth.write_parameter_file(pf, g["lo"], g["hi"])
    emu = Emulator(training_set_path=tp, parameter_file=pf, npc=int(g["npc"]), parameterTrafoPCA=True)
    emu.trainEmulator([True] * emu.nev, thetas=g["thetas"])
    eng = emu._engine_ready()
    Xd = torch.as_tensor(np.ascontiguousarray(g["Xs"]), device="cuda")
    assert maxrel(eng.param_map(Xd).cpu().numpy(), emu._ppca.transform(g["Xs"])) < 1e-12
    with pytest.raises(ValueError):
        eng.loglike(Xd)                                   # 20 raw columns are not a GP input of this engine
    yexp = g["mean"][0]
    err = 0.05 * np.abs(yexp)
    synth.write_experiment_pickle(ep, yexp, err)
    chain = Chain(mcmc_path=str(tmp_path / "mcmc" / "c.pkl"), expdata_path=ep, model_parafile=pf)
    chain.emuList = [emu]
    Xw = g["Xs"].copy()
    Xw[3, 0] = g["hi"][0] + 0.1                           # outside the prior box in an ORIGINAL parameter
    Xw[5, 16] = g["lo"][16]                               # on the boundary: outside (strict inequalities)
    inside = np.ones(len(Xw), bool); inside[[3, 5]] = False
    lp = chain.log_posterior(Xw)
    assert np.all(np.isneginf(lp[~inside])) and np.all(np.isfinite(lp[inside]))
    assert np.all(chain.log_likelihood(Xw, finite=True)[~inside] == -1e300)
    # against the reference's own predictions (golden) ...
    ref = np.array([O.mvn_loglike(m - yexp, c + np.diag(err ** 2)) for m, c in zip(g["mean"], g["cov"])]) + O.EXTRA_STD_CONST
    assert relerr(lp[inside], ref[inside]) < 1e-10
    # ... and against the host-mapped route of this build (isolates the device map)
    mY, mC = chain._predict(Xw[inside], 0.0)
    ref2 = np.array([O.mvn_loglike(m - yexp, c + np.diag(err ** 2)) for m, c in zip(mY, mC)]) + O.EXTRA_STD_CONST
    assert relerr(lp[inside], ref2) < 1e-10
    # the whole chain in one C call (gpb_chain_logpost: rows inside the box gathered, mapped, GP, likelihood) against
    # the same calls sequenced from Python over all rows: the same bits
    assert eng.lib.gpb_chain_supported((__import__("ctypes").c_void_p * 1)(eng.h), 1) == 1
    chain.use_chain_call = False
    assert np.array_equal(chain.log_posterior(Xw), lp)
    # ... and the C-driven sampling loop over the mapped emulator against the host-driven one
    from gpbayestools_hic_amd import StretchSampler
    nw = 44
    X0 = g["lo"] + (g["hi"] - g["lo"]) * np.random.default_rng(3).uniform(0.3, 0.7, (nw, len(g["lo"])))
    X0[7, 2] = g["hi"][2] + 1.0
    host = StretchSampler(chain, nw, seed=11)
    assert host._resident_engine() is None
    host.run(X0, 6, status=4)
    chain.use_chain_call = True
    res = StretchSampler(chain, nw, seed=11)
    assert res._resident_engine()[0] is eng
    res.run(X0, 6, status=5)
    assert np.array_equal(res.chain, host.chain) and np.array_equal(res.lnprobability, host.lnprobability)
    assert np.array_equal(res.naccept.cpu().numpy(), host.naccept.cpu().numpy())
    assert np.isfinite(res.lnprobability).any() and np.any(res.naccept.cpu().numpy() > 0)
