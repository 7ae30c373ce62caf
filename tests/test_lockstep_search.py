"""CPU tier: the lock-step driver of the hyper-parameter searches (emulator._batched_lbfgsb).  One thread stepping scipy's
reverse-communication L-BFGS-B routine for all searches must give what one scipy.optimize.minimize thread per search gives —
what sklearn itself runs (sk:_gpr.py:654-670) — bit for bit, whether the engine serves subsets of the searches or always all."""
import numpy as np
import pytest
import scipy.optimize

from gpbayestools_hic_amd import emulator as E


class _FakeEngine:
    """-LML_p(theta) = a smooth, badly scaled function with its own optimum per search; counts what it is asked"""

    def __init__(self, P, k, subset):
        rng = np.random.default_rng(11)
        self.c = rng.uniform(-1.5, 1.5, (P, k))
        self.w = rng.uniform(0.2, 30.0, (P, k))
        self.calls = []
        if subset:
            self.lml_active = self._subset

    def _one(self, p, th):
        r = th - self.c[p]
        v = -(np.sum(self.w[p] * r ** 2) + np.sum(np.cosh(0.3 * th)) + 2.0 * np.sin(th[0] * th[-1]))
        g = -(2.0 * self.w[p] * r + 0.3 * np.sinh(0.3 * th))
        g[0] -= 2.0 * np.cos(th[0] * th[-1]) * th[-1]
        g[-1] -= 2.0 * np.cos(th[0] * th[-1]) * th[0]
        return v, g

    def lml(self, theta, eval_gradient=True):
        self.calls.append(theta.shape[0])
        out = [self._one(p, theta[p]) for p in range(theta.shape[0])]
        return np.array([o[0] for o in out]), np.array([o[1] for o in out])

    def _subset(self, ids, theta):
        self.calls.append(len(ids))
        out = [self._one(int(p), theta[i]) for i, p in enumerate(ids)]
        return np.array([o[0] for o in out]), np.array([o[1] for o in out])


@pytest.mark.parametrize("subset", [True, False])
@pytest.mark.parametrize("per_search_bounds", [False, True])
def test_one_driver_thread_equals_one_scipy_thread_per_search(subset, per_search_bounds):
    assert E._setulb_usable()                      # this image's scipy (1.15): the single-thread driver is what runs
    P, k = 7, 6
    rng = np.random.default_rng(3)
    bounds = np.stack([np.full(k, -1.0), np.full(k, 1.2)], axis=1)
    if per_search_bounds:
        bounds = np.stack([bounds + 0.05 * p * np.array([-1.0, 1.0]) for p in range(P)])
    start = rng.uniform(-0.9, 1.1, (P, k))
    start[2] = 5.0                                 # a start outside the box: scipy clips it
    a, b = _FakeEngine(P, k, subset), _FakeEngine(P, k, subset)
    th_a, val_a = E._lockstep_setulb(a, start, bounds)
    th_b, val_b = E._lockstep_threads(b, start, bounds)
    assert np.array_equal(th_a, th_b) and np.array_equal(val_a, val_b)
    assert a.calls == b.calls                      # the same rounds, the same searches in each
    if subset:
        assert a.calls[0] == P and a.calls[-1] < P  # converged searches leave the batch
    # ... and each is the search scipy.optimize.minimize runs on its own
    for p in range(P):
        ref = scipy.optimize.minimize(lambda th: tuple(-x for x in a._one(p, th)), start[p], method="L-BFGS-B", jac=True,
                                      bounds=bounds[p] if per_search_bounds else bounds)
        assert np.array_equal(ref.x, th_a[p]) and ref.fun == val_a[p]
    assert np.any(th_a == bounds[..., 1].max()) or np.any(np.isclose(th_a, 1.2))     # some optimum sits on a bound


def test_threads_when_scipy_offers_another_routine(monkeypatch):
    monkeypatch.setattr(E, "_setulb_ok", None)
    monkeypatch.setattr(E, "_SETULB_SIGNATURE", "setulb(some, other, signature)")
    assert not E._setulb_usable()
    eng = _FakeEngine(3, 4, True)
    start = np.zeros((3, 4))
    bounds = np.stack([np.full(4, -2.0), np.full(4, 2.0)], axis=1)
    th, val = E._batched_lbfgsb(eng, start, bounds)
    monkeypatch.setattr(E, "_setulb_ok", None)
    monkeypatch.undo()
    assert E._setulb_usable()
    th2, val2 = E._batched_lbfgsb(_FakeEngine(3, 4, True), start, bounds)
    assert np.array_equal(th, th2) and np.array_equal(val, val2)
