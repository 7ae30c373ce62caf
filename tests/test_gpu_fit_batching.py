"""GPU tier: the hyper-parameter searches of many GPs in one batch (gpb_gp_set_multi / gpb_gp_lml_subset).  The reference fits
GP after GP (src/emulator.py:309-315), start after start (sk:_gpr.py:318-337) and dataset after dataset
(examples/EmulatorTraining.ipynb:124-138); here the GPs of several emulators and all their restarts share every launch of
the log-marginal-likelihood evaluation, searches that have converged leave the batch — and every GP must end exactly where its
own one-emulator, start-by-start training ends: a GP's numbers do not depend on the batch it is evaluated in."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _emulator(workdir, tag, N, d, M, npc, seed, nrestarts=0):
    from gpbayestools_hic_amd import Emulator, synth
    X = synth.lhs(N, d, seed=seed)
    Y = synth.observables(X, M, seed=seed + 1)
    tp, pf = os.path.join(workdir, f"train{tag}.pkl"), os.path.join(workdir, f"par{tag}.txt")
    synth.write_training_pickle(tp, X, Y, 0.01)
    synth.write_parameter_file(pf, np.zeros(d), np.ones(d))
    return Emulator(training_set_path=tp, parameter_file=pf, npc=npc, nrestarts=nrestarts)


@pytest.mark.parametrize("kind", ["RBF", "Matern15", "Matern25"])
def test_gps_over_different_designs_in_one_context(kind):
    """gpb_gp_set_multi: five GPs over three designs of 100 / 128 / 70 points (all padded to 128): LML and gradient equal, bit for
    bit, what each design's own context gives — for the whole batch and for any subset in any order — and the oracle's to the
    G2 bars; the context refuses to predict."""
    from gpbayestools_hic_amd import GPEngine, synth
    from gpbayestools_hic_amd._native import GPBError
    from oracle import gp_oracle as O
    d = 5
    rng = np.random.default_rng(3)
    designs = [synth.lhs(n, d, seed=40 + i) for i, n in enumerate((100, 128, 70))]
    owner = [0, 0, 1, 2, 2]
    Zs = [np.sin(designs[o] @ rng.standard_normal(d)) + 0.1 * rng.standard_normal(designs[o].shape[0]) for o in owner]
    th = np.array([np.concatenate([[rng.uniform(-0.5, 0.5)], np.log(rng.uniform(0.3, 3.0, d)), [np.log(rng.uniform(0.02, 0.2))]])
                   for _ in owner])
    th[3, 1 + 2] = np.log(5e-4)                                   # one GP beyond the Gram form's range
    multi = GPEngine(0)
    multi.set_data_multi([designs[o] for o in owner], Zs, kind, 0.1)
    val, grad = multi.lml(th)
    assert multi.get("form").tolist() == [0.0, 0.0, 0.0, 1.0, 0.0]
    for g, o in enumerate(owner):
        one = GPEngine(0)
        one.set_data(designs[o], Zs[g][None, :], kind, 0.1)
        v1, g1 = one.lml(th[g][None, :])
        assert v1[0] == val[g] and np.array_equal(g1[0], grad[g]), g
        vo, go = O.lml(th[g], designs[o], Zs[g], O.KIND_NAMES[kind], 0.1, eval_gradient=True)
        assert abs(val[g] - vo) <= 1e-10 * abs(vo) and np.max(np.abs(grad[g] - go)) <= 1e-9 * max(np.max(np.abs(go)), 1.0)
        one.close()
    for idx in ([4], [3, 1], [2, 0, 4], [4, 3, 2, 1, 0]):
        vs, gs = multi.lml_subset(idx, th[idx])
        assert np.array_equal(vs, val[idx]) and np.array_equal(gs, grad[idx]), idx
    with pytest.raises(GPBError):
        multi.predict(np.zeros((3, d)))                            # no factorisation after a subset call ...
    multi.set_theta(th); multi.factor()
    L = multi.get("L")
    for g, o in enumerate(owner):
        n = designs[o].shape[0]
        Lo, _ = O.gp_factor(designs[o], Zs[g], th[g], O.KIND_NAMES[kind], 0.1)
        assert np.max(np.abs(L[g][:n, :n] - Lo)) <= 1e-11 * np.max(np.abs(Lo))
    with pytest.raises(GPBError, match="fit-only"):
        multi.predict(np.zeros((3, d)))                            # ... and never a prediction from GPs of different designs
    with pytest.raises(GPBError):
        multi.lml_subset([5], th[:1])
    with pytest.raises((GPBError, AssertionError)):
        bad = GPEngine(0)
        bad.set_data_multi([designs[0], synth.lhs(200, d)], [Zs[0], np.zeros(200)], kind, 0.1)    # 128 and 256 padded points
    multi.close()


def test_subset_evaluation_on_an_ordinary_context():
    from gpbayestools_hic_amd import GPEngine, synth
    N, d, P = 150, 4, 6
    rng = np.random.default_rng(8)
    X = synth.lhs(N, d, seed=2)
    Z = rng.standard_normal((P, N))
    th = synth.fixed_theta(d, P) + 0.3 * rng.standard_normal((P, d + 2))
    eng = GPEngine(0)
    eng.set_data(X, Z, "RBF", 0.1)
    val, grad = eng.lml(th)
    for idx in ([5], [1, 4], [0, 2, 3, 5]):
        vs, gs = eng.lml_subset(idx, th[idx])
        assert np.array_equal(vs, val[idx]) and np.array_equal(gs, grad[idx])
    eng.set_theta(th); eng.factor()                                # the context is whole again afterwards
    m, v = eng.predict(X[:9])
    assert np.all(np.isfinite(m)) and np.all(v > 0)
    eng.close()


def test_restarts_in_one_batch_equal_start_by_start(tmp_path):
    """nrestarts = 2: the three starts of every GP's search as ONE lock-step batch (a fit-only context with each GP three
    times) against the same search run start after start on the emulator's own context"""
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd import emulator as E
    emu = _emulator(str(tmp_path), "a", 90, 4, 6, 4, seed=5, nrestarts=2)
    np.random.seed(123)
    emu.trainEmulatorAutoMask()
    th_batched, lml_batched = emu.thetas_.copy(), np.asarray(emu.lml_).copy()
    theta0, bounds = emu._theta0_bounds("RBF")
    np.random.seed(123)
    eng = emu._engine_ready()
    th_seq, lml_seq = E.search_hyperparameters(lambda idx: eng, emu._ngp, theta0, bounds, 2)
    assert np.array_equal(th_batched, th_seq) and np.array_equal(lml_batched, lml_seq)
    eng.set_theta(emu.thetas_); eng.factor()                       # (the search left the context without a factorisation)
    moved = np.any(np.abs(th_batched - theta0) > 1e-3, axis=1)
    assert moved.all()
    Xq = synth.walkers(11, 4, seed=1)
    emu2 = _emulator(str(tmp_path), "b", 90, 4, 6, 4, seed=5, nrestarts=2)
    emu2.trainEmulator([True] * emu2.nev, thetas=th_seq)
    assert np.array_equal(emu.predict(Xq, return_cov=False), emu2.predict(Xq, return_cov=False))


@pytest.mark.parametrize("kernel_type", ["RBF", "Matern"])
def test_train_emulators_together_equals_one_after_the_other(tmp_path, kernel_type):
    """four emulators with their own designs (100 / 128 / 77 / 200 points: two groups by padded size), numbers of GPs and
    restarts: trained together and one after the other they end on identical hyper-parameters, likelihoods and predictions"""
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd.emulator import train_emulators
    spec = [(100, 5, 3, 0), (128, 7, 4, 1), (77, 5, 2, 0), (200, 6, 3, 1)]          # N, M, npc, nrestarts
    d = 4
    Xq = synth.walkers(13, d, seed=3)

    def make(sub):
        (tmp_path / sub).mkdir()
        return [_emulator(str(tmp_path / sub), str(i), N, d, M, npc, seed=60 + 3 * i, nrestarts=nr)
                for i, (N, M, npc, nr) in enumerate(spec)]
    seq = make("seq")
    np.random.seed(77)
    for emu in seq:
        emu.trainEmulator([True] * emu.nev, kernel_type=kernel_type)
    tog = make("tog")
    np.random.seed(77)
    assert train_emulators(tog, kernel_type=kernel_type) is not None
    for a, b in zip(seq, tog):
        assert np.array_equal(a.thetas_, b.thetas_) and np.array_equal(np.asarray(a.lml_), np.asarray(b.lml_))
        ma, ca = a.predict(Xq, return_cov=True)
        mb, cb = b.predict(Xq, return_cov=True)
        assert np.array_equal(ma, mb) and np.array_equal(ca, cb)
        assert np.array_equal(a.gp_scores_, b.gp_scores_)


def test_a_batch_that_would_not_fit_trains_the_emulators_one_by_one(tmp_path, monkeypatch):
    """train_emulators' fallback (the batch's matrices beyond the memory bound): each emulator's own search from the same restart
    points — identical results again"""
    from gpbayestools_hic_amd import emulator as E
    spec = [(100, 5, 3, 1), (128, 7, 2, 0)]
    def make(sub):
        (tmp_path / sub).mkdir()
        return [_emulator(str(tmp_path / sub), str(i), N, 4, M, npc, seed=80 + 3 * i, nrestarts=nr) for i, (N, M, npc, nr) in enumerate(spec)]
    a = make("a")
    np.random.seed(5)
    E.train_emulators(a)
    monkeypatch.setattr(E, "_BATCH_BYTES_MAX", 0)
    b = make("b")
    np.random.seed(5)
    E.train_emulators(b)
    for x, y in zip(a, b):
        assert np.array_equal(x.thetas_, y.thetas_) and np.array_equal(np.asarray(x.lml_), np.asarray(y.lml_))
