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
