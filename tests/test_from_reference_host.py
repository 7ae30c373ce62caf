"""CPU tier: the host side of Emulator.from_reference (no device call until the first prediction): the attribute holders rebuilt
from tests/golden/g11_trained_objects.npz are read into drop-in emulators with the reference's hyper-parameters, targets, flags
and parameter maps; what cannot be taken over is refused."""
import types

import numpy as np
import pytest

from conftest import golden
from test_gpu_from_reference import rebuild


@pytest.mark.parametrize("name", ["mask", "logexp", "nopca", "ppca"])
def test_host_state_of_an_adopted_emulator(name):
    from gpbayestools_hic_amd.emulator import Emulator
    g = golden("g11_trained_objects.npz")
    ref = rebuild(g, name)
    emu = Emulator.from_reference(ref, device=0)
    assert emu._trained and emu._engine is None                          # nothing touched a device
    assert emu.kernel_type_ == ("Matern" if name == "logexp" else "RBF") and emu.alpha == 0.1
    assert np.array_equal(emu.thetas_, g[name + "_gp_theta"]) and np.array_equal(emu._Z_train, g[name + "_gp_y_train"])
    assert np.array_equal(emu._X_train, g[name + "_gp_X_train"]) and np.array_equal(emu.lml_, g[name + "_gp_lml"])
    flags = [bool(x) for x in g[name + "_flags"]]
    assert [emu.logTrafo_, emu.parameterTrafoPCA_, emu.exp_and_cov_diagonal_, emu.perform_no_PCA_] == flags
    assert (emu.nev, emu.nobs) == g[name + "_model_data"].shape
    if name == "mask":
        assert emu._X_train.shape[0] == emu.nev - 5                      # trained on a masked event set: the GPs' own inputs count
    if name == "nopca":
        assert emu._ngp == emu.nobs and emu._trans_matrix is None
    else:
        assert emu._trans_matrix.shape == (emu.nobs, emu.nobs) and emu._cov_trunc.shape == (emu.nobs, emu.nobs)
    if name == "ppca":
        # the adopted parameter maps send the design to the reference's own reduced design
        assert np.max(np.abs(emu._map_parameters(emu.design_points) - g["ppca_PCA_new_design_points"])) < 1e-12
        assert emu.PCA_new_design_points.shape[1] == emu._X_train.shape[1] == emu.thetas_.shape[1] - 2
    import dill
    again = dill.loads(dill.dumps(emu))                                   # an ordinary drop-in emulator: it pickles
    assert np.array_equal(again.thetas_, emu.thetas_) and again.state_digest() == emu.state_digest()
    assert [gp.kernel_.theta.shape for gp in again.gps] == [(emu.thetas_.shape[1],)] * emu._ngp


def test_refusals():
    from gpbayestools_hic_amd.emulator import Emulator
    g = golden("g11_trained_objects.npz")
    with pytest.raises(ValueError):
        Emulator.from_reference(types.SimpleNamespace(gps=None))
    ref = rebuild(g, "mask")
    ref.gps[0].alpha = 0.2
    with pytest.raises(ValueError):
        Emulator.from_reference(ref)                                      # GPs with different jitter
    ref = rebuild(g, "logexp")
    ref.gps[0].kernel_.k1.k2.nu = 0.5
    with pytest.raises(ValueError):
        Emulator.from_reference(ref)                                      # Matern-1/2: not on the device path


def test_refusals_for_what_the_device_state_cannot_hold():
    """A GP fitted with normalize_y, a per-point alpha, a GP count that does not match npc: ValueError, never a silent take-over."""
    from gpbayestools_hic_amd.emulator import Emulator
    g = golden("g11_trained_objects.npz")
    ref = rebuild(g, "mask")
    ref.gps[1]._y_train_mean, ref.gps[1]._y_train_std = np.float64(0.3), np.float64(2.0)
    with pytest.raises(ValueError, match="normalize_y"):
        Emulator.from_reference(ref)
    ref = rebuild(g, "mask")
    ref.gps[0].normalize_y = True
    with pytest.raises(ValueError, match="normalize_y"):
        Emulator.from_reference(ref)
    ref = rebuild(g, "mask")
    for gp in ref.gps:
        gp.alpha = np.full(gp.X_train_.shape[0], 0.1)
    with pytest.raises(ValueError, match="alpha"):
        Emulator.from_reference(ref)
    ref = rebuild(g, "mask")
    ref.gps = ref.gps[:-1]
    with pytest.raises(ValueError, match="GPs for"):
        Emulator.from_reference(ref)
    ref = rebuild(g, "nopca")
    ref.gps = ref.gps[:-1]
    with pytest.raises(ValueError, match="observables"):
        Emulator.from_reference(ref)


def test_load_emulator_keeps_an_incomplete_object_foreign(tmp_path):
    """Chain.loadEmulator: a pickle that looks like a reference emulator but lacks what from_reference reads (AttributeError,
    TypeError, KeyError — not only ValueError) stays a foreign emulator instead of aborting the load."""
    import dill
    from gpbayestools_hic_amd.mcmc import Chain

    chain = Chain.__new__(Chain)
    chain.emuList, chain.device = [], 0
    g = golden("g11_trained_objects.npz")
    cases = []
    ref = rebuild(g, "mask"); del ref.design_min; cases.append(ref)                   # AttributeError inside from_reference
    ref = rebuild(g, "mask"); ref.npc = None; cases.append(ref)                       # TypeError (int(None))
    ref = rebuild(g, "mask"); del ref.scaler.var_; cases.append(ref)                  # AttributeError on the scaler
    ref = rebuild(g, "mask"); ref.gps[0]._y_train_std = np.float64(3.0); cases.append(ref)   # ValueError (normalize_y)
    paths = []
    for i, ref in enumerate(cases):
        path = tmp_path / ("emu%d.pkl" % i)
        with open(path, "wb") as f:
            dill.dump(ref, f)
        paths.append(str(path))
    chain.loadEmulator(paths)
    assert len(chain.emuList) == len(cases)
    from gpbayestools_hic_amd.emulator import Emulator
    assert not any(isinstance(e, Emulator) for e in chain.emuList)
