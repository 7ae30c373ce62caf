"""GPU tier: trained emulator OBJECTS of the reference taken over without retraining (Emulator.from_reference; Chain.loadEmulator on
the dill pickles examples/EmulatorTraining.ipynb writes, src/mcmc.py:145-150).  The reference cannot travel to the GPU box, so the
objects are rebuilt here from tests/golden/g11_trained_objects.npz (tools/make_goldens.py::g11_trained_objects: the attributes of
four trained reference emulators and their own predictions) as plain attribute holders of the same shape — from_reference only
reads attributes — and the adopted emulators must reproduce the reference's predictions."""
import types

import numpy as np
import pytest

from conftest import golden, maxrel, relerr

pytestmark = pytest.mark.gpu


class ConstantKernel: pass          # noqa: E701  (from_reference reads the kernel family off the class names, as of sklearn's kernels)
class WhiteKernel: pass             # noqa: E701
class RBF: pass                     # noqa: E701
class Matern:                       # noqa: E701
    nu = 1.5


def _ns(**kw):
    return types.SimpleNamespace(**kw)


def rebuild(g, name):
    """the trained reference object `name` of the fixture, as an attribute holder"""
    p = name + "_"
    fam = [str(x) for x in g[p + "gp_family"]]
    assert fam[0] == "ConstantKernel" and fam[2] == "WhiteKernel"
    inner = RBF() if fam[1] == "RBF" else Matern()
    if fam[1] == "Matern":
        inner.nu = float(g[p + "gp_nu"])
    flags = [bool(x) for x in g[p + "flags"]]
    gps = [_ns(kernel_=_ns(k1=_ns(k1=ConstantKernel(), k2=inner), k2=WhiteKernel(), theta=th), X_train_=g[p + "gp_X_train"],
               y_train_=y, alpha=float(g[p + "gp_alpha"]), log_marginal_likelihood_value_=float(l))
           for th, y, l in zip(g[p + "gp_theta"], g[p + "gp_y_train"], g[p + "gp_lml"])]
    ref = _ns(logTrafo_=flags[0], parameterTrafoPCA_=flags[1], exp_and_cov_diagonal_=flags[2], perform_no_PCA_=flags[3],
              npc=int(g[p + "npc"]), nrestarts=0, pardict=None, gps=gps,
              scaler=_ns(mean_=g[p + "scaler_mean_"], scale_=g[p + "scaler_scale_"], var_=g[p + "scaler_var_"]),
              **{k: g[p + k] for k in ("design_points", "model_data", "model_data_err", "design_min", "design_max")})
    if not flags[3]:
        ref.pca = _ns(n_components_=int(g[p + "pca_n_components_"]),
                      **{k: g[p + "pca_" + k] for k in ("mean_", "components_", "explained_variance_", "explained_variance_ratio_")})
    if flags[1]:
        ref.PCA_new_design_points = g[p + "PCA_new_design_points"]
        for tag in ("bulk", "shear", "yloss"):
            setattr(ref, "paramTrafoScaler_" + tag, _ns(**{k: g[p + tag + "_scaler_" + k] for k in ("mean_", "scale_", "var_")}))
            setattr(ref, "paramTrafoPCA_" + tag, _ns(n_components_=int(g[p + tag + "_pca_n_components_"]),
                                                      **{k: g[p + tag + "_pca_" + k] for k in
                                                         ("mean_", "components_", "explained_variance_", "explained_variance_ratio_")}))
    return ref


@pytest.mark.parametrize("name", ["mask", "logexp", "nopca", "ppca"])
def test_adopted_emulator_reproduces_the_references_predictions(name):
    from gpbayestools_hic_amd import Emulator
    g = golden("g11_trained_objects.npz")
    ref = rebuild(g, name)
    emu = Emulator.from_reference(ref)
    assert emu.kernel_type_ == ("Matern" if name == "logexp" else "RBF") and emu._trained
    assert np.array_equal(emu.thetas_, g[name + "_gp_theta"]) and emu._X_train.shape == g[name + "_gp_X_train"].shape
    mean, cov = emu.predict(g[name + "_Xs"], return_cov=True, extra_std=g[name + "_es"])
    assert relerr(mean, g[name + "_mean"]) < 1e-10
    assert maxrel(cov, g[name + "_cov"]) < 1e-9
    # the fitted-GP surface: LML at the adopted theta = the reference's log_marginal_likelihood_value_
    lml = emu._engine_ready().lml(emu.thetas_, eval_gradient=False)
    assert np.max(np.abs(lml - g[name + "_gp_lml"]) / np.abs(g[name + "_gp_lml"])) < 1e-10
    # ... and an adopted emulator is an ordinary drop-in one: it pickles (no device handle inside) and predicts the same bits again
    import dill
    again = dill.loads(dill.dumps(emu))
    m2, c2 = again.predict(g[name + "_Xs"], return_cov=True, extra_std=g[name + "_es"])
    assert np.array_equal(m2, mean) and np.array_equal(c2, cov)


def test_load_emulator_adopts_reference_pickles_and_keeps_other_objects_foreign(tmp_path):
    """Chain.loadEmulator: a pickled trained reference emulator runs through the device path (the chain's one-call log-posterior
    applies), adopt=False and objects that are no such emulator stay foreign; both give the same log-posterior."""
    import dill
    from gpbayestools_hic_amd import Chain, Emulator, synth
    g = golden("g11_trained_objects.npz")
    ref = rebuild(g, "mask")
    ref.predict = None                                   # (an attribute holder cannot predict: the foreign path is not used on it)
    path = str(tmp_path / "ref_emulator.pkl")
    with open(path, "wb") as f:
        dill.dump(ref, f)
    lo, hi = g["mask_lo"], g["mask_hi"]
    pf, ep = str(tmp_path / "p.txt"), str(tmp_path / "e.pkl")
    synth.write_parameter_file(pf, lo, hi)
    yexp = g["mask_mean"][0]
    synth.write_experiment_pickle(ep, yexp, 0.05 * np.abs(yexp) + 1e-3)
    chain = Chain(mcmc_path=str(tmp_path / "mcmc" / "c.pkl"), expdata_path=ep, model_parafile=pf)
    chain.loadEmulator([path])
    assert isinstance(chain.emuList[0], Emulator) and chain._native()
    X = lo + (hi - lo) * np.random.default_rng(3).uniform(-0.05, 1.05, (200, len(lo)))
    lp = chain.log_posterior(X)
    ins = np.all((X > lo) & (X < hi), axis=1)
    assert np.array_equal(np.isfinite(lp), ins) and ins.any() and not ins.all()
    # the same through an emulator adopted by hand
    chain2 = Chain(mcmc_path=str(tmp_path / "mcmc" / "c2.pkl"), expdata_path=ep, model_parafile=pf)
    chain2.emuList = [Emulator.from_reference(rebuild(g, "mask"))]
    assert np.array_equal(chain2.log_posterior(X), lp)
    chain3 = Chain(mcmc_path=str(tmp_path / "mcmc" / "c3.pkl"), expdata_path=ep, model_parafile=pf)
    chain3.loadEmulator([path], adopt=False)
    assert not isinstance(chain3.emuList[0], Emulator)


def test_from_reference_refuses_what_it_cannot_take_over():
    from gpbayestools_hic_amd import Emulator
    g = golden("g11_trained_objects.npz")
    with pytest.raises(ValueError):
        Emulator.from_reference(types.SimpleNamespace(gps=[]))                      # untrained
    ref = rebuild(g, "mask")
    ref.gps[1].kernel_.k1.k2 = type("RationalQuadratic", (), {})()
    with pytest.raises(ValueError):
        Emulator.from_reference(ref)                                                # a kernel family the device path lacks
    ref = rebuild(g, "mask")
    ref.gps[1].X_train_ = ref.gps[1].X_train_ + 1.0
    with pytest.raises(ValueError):
        Emulator.from_reference(ref)                                                # GPs over different inputs
