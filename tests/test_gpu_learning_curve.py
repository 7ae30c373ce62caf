"""GPU tier: `Emulator.print_learning_curve` (src/emulator.py:424-462) against the reference's own output on the same data
(tests/golden/g10_learning_curve.npz, tools/make_goldens.py::g10_learning_curve: sklearn's learning_curve over 25 GPR fits per
principal component).  The drop-in runs the 25 x npc hyper-parameter searches as five lock-step batches on the device."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def test_learning_curve_tables_equal_the_references(tmp_path):
    from gpbayestools_hic_amd import Emulator, synth
    g = golden("g10_learning_curve.npz")
    tp, pf = str(tmp_path / "t.pkl"), str(tmp_path / "p.txt")
    synth.write_training_pickle(tp, g["X"], g["Y"], g["Yerr"])
    synth.write_parameter_file(pf, g["lo"], g["hi"])
    emu = Emulator(training_set_path=tp, parameter_file=pf, npc=int(g["npc"]))
    status = emu.print_learning_curve()
    ref = g["status"]                                   # [npc, 5 train sizes, (size, mean train R^2, mean test R^2)]
    assert len(status) == ref.shape[0]
    got = np.array(status)
    assert got.shape == ref.shape
    assert np.array_equal(got[:, :, 0], ref[:, :, 0])                   # 9, 19, 28, 38, 43 of a fold's 48 training events
    # the scores hang on where 25 L-BFGS-B searches per GP end: the same algorithm (scipy's) on the same objective — the bar is
    # the optimiser's own tolerance carried into R^2, not the 1e-10 of a fixed-theta prediction
    err = np.abs(got[:, :, 1:] - ref[:, :, 1:])
    assert err.max() < 1e-5, err
    # a trained emulator is left as it was (the reference refits self.scaler / self.pca inside the call; with all events in the
    # training, as here, to the same state)
    emu.trainEmulatorAutoMask()
    before = (emu.scaler.mean_.copy(), emu.pca.components_.copy(), emu.thetas_.copy())
    again = np.array(emu.print_learning_curve())
    assert np.array_equal(again, got)                                   # deterministic
    assert np.array_equal(emu.scaler.mean_, before[0]) and np.array_equal(emu.pca.components_, before[1])
    assert np.array_equal(emu.thetas_, before[2])
    assert np.allclose(emu.scaler.mean_, g["scaler_mean"], rtol=1e-13, atol=0)


def test_a_fit_that_is_not_positive_definite_scores_nan_and_keeps_the_others(tmp_path, monkeypatch):
    """sklearn's learning_curve (error_score = nan) records NaN for a failed fit and returns the other scores
    (src/emulator.py:449-455); here the failure is injected: the first scoring factorisation reports LAPACK's info = 3 for GP 1"""
    from gpbayestools_hic_amd import Emulator, GPEngine, synth
    import gpbayestools_hic_amd.emulator as E
    g = golden("g10_learning_curve.npz")
    tp, pf = str(tmp_path / "t.pkl"), str(tmp_path / "p.txt")
    synth.write_training_pickle(tp, g["X"], g["Y"], g["Yerr"])
    synth.write_parameter_file(pf, g["lo"], g["hi"])
    emu = Emulator(training_set_path=tp, parameter_file=pf, npc=int(g["npc"]))
    clean = np.array(emu.print_learning_curve())
    real, calls = GPEngine.factor, []

    def factor(self, raise_on_fail=True):
        info = np.asarray(real(self, raise_on_fail=raise_on_fail)).copy()
        if not raise_on_fail:                            # (the scoring engines ask for the info vector; the searches do not)
            calls.append(1)
            if len(calls) == 1 and len(info) > 1:
                info[1] = 3
        return info

    monkeypatch.setattr(E.GPEngine, "factor", factor)
    got = np.array(emu.print_learning_curve())
    assert calls
    # GP 1 at the first train size: one fold is NaN, so is the mean over the folds; every other entry is the clean run's
    assert np.isnan(got[1, 0, 1]) and np.isnan(got[1, 0, 2])
    mask = np.ones(got.shape, bool); mask[1, 0, 1:] = False
    assert np.array_equal(got[mask], clean[mask])
