#!/usr/bin/env python3
"""Build-container check (needs /root/reference; no GPU): a REAL trained `src.emulator.Emulator`, dill-pickled as
examples/EmulatorTraining.ipynb does, goes through this package's `Chain.loadEmulator` and comes out as a drop-in `Emulator`
whose host-side state equals the reference object's — fitted transforms bit for bit, targets, hyper-parameters, parameter maps.
(The device side of the take-over is tests/test_gpu_from_reference.py, on the attribute arrays of tests/golden/g11_trained_objects.npz.)"""
import importlib.util
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [sys.argv[0]]
spec = importlib.util.spec_from_file_location("mg", os.path.join(REPO, "tools", "make_goldens.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)
RefEmulator, _ = mg._import_reference()

import dill  # noqa: E402
from gpbayestools_hic_amd import synth  # noqa: E402
from gpbayestools_hic_amd.emulator import Emulator  # noqa: E402
from gpbayestools_hic_amd.mcmc import Chain  # noqa: E402

g7 = np.load(os.path.join(REPO, "tests", "golden", "g7_param_pca.npz"))
for name, kw, ktype in (("pca", {}, "RBF"), ("logexp", dict(logTrafo=True, exp_and_cov_diagonal=True), "Matern"),
                        ("nopca", dict(perform_no_PCA=True), "RBF"), ("ppca", dict(parameterTrafoPCA=True), "RBF")):
    if name == "ppca":
        lo, hi, X, Y = g7["lo"], g7["hi"], g7["X"], g7["Y"]
        Yerr = np.full_like(Y, 0.01)
    else:
        lo, hi, X, Y, Yerr = mg._make_inputs("chk_" + name, 60, 5, 4, 1300)
        Y = np.abs(Y) + 0.5
    tp, pf, ep = (os.path.join(mg._work, f"chk_{name}_{x}") for x in ("t.pkl", "p.txt", "e.pkl"))
    synth.write_training_pickle(tp, X, Y, Yerr)
    synth.write_parameter_file(pf, lo, hi)
    ref = RefEmulator(training_set_path=tp, parameter_file=pf, npc=3, **kw)
    ref.trainEmulator([True] * ref.nev, kernel_type=ktype)
    path = os.path.join(mg._work, f"chk_{name}_emu.pkl")
    with open(path, "wb") as f:
        dill.dump(ref, f)
    yexp = ref.predict(np.atleast_2d(0.5 * (lo + hi)), return_cov=False)[0]
    synth.write_experiment_pickle(ep, yexp, 0.05 * np.abs(yexp) + 1e-3)
    chain = Chain(mcmc_path=os.path.join(mg._work, "mcmc", "c.pkl"), expdata_path=ep, model_parafile=pf)
    chain.loadEmulator([path])
    emu = chain.emuList[0]
    assert isinstance(emu, Emulator) and emu._trained, name
    assert np.array_equal(emu.thetas_, np.array([gp.kernel_.theta for gp in ref.gps]))
    assert np.array_equal(emu._Z_train, np.array([gp.y_train_ for gp in ref.gps])) and np.array_equal(emu._X_train, ref.gps[0].X_train_)
    if not kw.get("perform_no_PCA"):
        assert np.array_equal(emu._trans_matrix, ref._trans_matrix) and np.array_equal(emu._cov_trunc, ref._cov_trunc)
        assert np.array_equal(emu._var_trans, ref._var_trans)
    if kw.get("parameterTrafoPCA"):
        Xs = lo + (hi - lo) * np.random.default_rng(1).random((7, len(lo)))
        new = emu._map_parameters(Xs)
        assert new.shape == (7, ref.PCA_new_design_points.shape[1])
        # the reference's own map of its design points is PCA_new_design_points: ours must map the design to the same rows
        assert np.max(np.abs(emu._map_parameters(ref.design_points) - ref.PCA_new_design_points)) < 1e-12
    print(name, "ok:", type(emu).__name__, emu.kernel_type_, emu._X_train.shape, "thetas", emu.thetas_.shape)
print("all trained reference pickles were taken over")
