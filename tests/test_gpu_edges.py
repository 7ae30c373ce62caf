"""GPU tier: degenerate and ragged shapes against the oracle (smallest sizes, odd sizes, empty batches)."""
import numpy as np
import pytest

from conftest import relerr, maxrel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,d,P,M,W,kind", [
    (1, 1, 1, 1, 1, "RBF"),            # one design point, one parameter, one observable
    (2, 1, 1, 1, 3, "Matern15"),
    (65, 3, 1, 2, 129, "RBF"),         # N, W just past a padding granule
    (127, 33, 2, 5, 64, "Matern25"),   # d > 32 (48-wide register path)
    (64, 2, 7, 9, 257, "RBF"),
])
def test_tiny_and_ragged_shapes(N, d, P, M, W, kind):
    from gpbayestools_hic_amd import GPEngine
    from gpbayestools_hic_amd.engine import MODE_PCA
    from oracle import gp_oracle as O
    rng = np.random.default_rng(N * 1000 + d)
    kid = O.KIND_NAMES[kind]
    X = rng.random((N, d))
    Z = rng.standard_normal((P, N))
    th = np.array([np.concatenate([[rng.uniform(-0.2, 0.3)], np.log(rng.uniform(0.5, 2.0, d)), [np.log(0.05)]])
                   for _ in range(P)])
    A = rng.standard_normal((P, M)); mu = rng.standard_normal(M)
    Bm = rng.standard_normal((M, M)); ctr = Bm @ Bm.T * 0.01 + 1e-3 * np.eye(M)
    yexp = rng.standard_normal(M); cexp = np.diag(rng.uniform(0.01, 0.1, M))
    eng = GPEngine(0)
    eng.set_data(X, Z, kind, 0.1); eng.set_theta(th); eng.factor()
    eng.set_transform(MODE_PCA, mu, A=A, cov_trunc=ctr)
    eng.set_likelihood(yexp, cexp)
    Xs = rng.random((W, d))
    m, v = eng.predict(Xs)
    mo = np.empty((W, P)); vo = np.empty((W, P))
    for p in range(P):
        L, a = O.gp_factor(X, Z[p], th[p], kid, 0.1)
        mo[:, p], vo[:, p] = O.gp_predict(Xs, X, th[p], L, a, kid)
    assert maxrel(m, mo) < 1e-11 and relerr(v, vo) < 1e-10
    mean, cov = eng.emu_predict(Xs, True, np.zeros(W))
    mref, cref = O.emulator_predict(mo, vo, np.zeros(W), mode=O.MODE_PCA, A=A, mu=mu, cov_trunc=ctr)
    assert maxrel(mean, mref) < 1e-11 and maxrel(cov, cref) < 1e-10
    ll = eng.loglike(Xs)
    ref = np.array([O.mvn_loglike(a_, c_) for a_, c_ in zip(mref - yexp, cref + cexp)])
    assert relerr(ll, ref) < 1e-10
    # empty batches are legal everywhere
    assert eng.predict(np.zeros((0, d)))[0].shape == (0, P)
    assert eng.emu_predict(np.zeros((0, d)))[1].shape == (0, M, M)
    assert eng.loglike(np.zeros((0, d))).shape == (0,)
    eng.close()


@pytest.mark.parametrize("kind", ["RBF", "Matern15", "Matern25"])
@pytest.mark.parametrize("c,ell,noise", [
    (1.0, 0.05, 1e-2),      # length scale far below the design spacing: K ~ diagonal
    (1.0, 20.0, 1e-2),      # nearly constant kernel: K close to rank one, variance by heavy cancellation
    (1e2, 1.0, 1e-6),       # sklearn's lower noise bound region, large amplitude
    (1e-3, 0.5, 1.0),       # noise dominated
    (1.0, 0.3, 1e-8),
])
def test_hyperparameters_at_the_edges_of_the_search_box(kind, c, ell, noise):
    """The optimiser of fit() visits the corners of the bounds (src/emulator.py:254-262); parity has to
    hold there as well, including where the predictive variance is a small difference of large terms."""
    from gpbayestools_hic_amd import GPEngine, synth
    from oracle import gp_oracle as O
    N, d, W = 300, 6, 200
    rng = np.random.default_rng(0)
    X = synth.lhs(N, d, seed=2); Xs = rng.random((W, d))
    z = np.sin(X @ rng.standard_normal(d))
    th = np.concatenate([[np.log(c)], np.log(np.full(d, ell)), [np.log(noise)]])
    eng = GPEngine(0)
    eng.set_data(X, z[None, :], kind, 0.1); eng.set_theta(th[None, :]); eng.factor()
    m, v = eng.predict(Xs)
    L, a = O.gp_factor(X, z, th, O.KIND_NAMES[kind], 0.1)
    mo, vo = O.gp_predict(Xs, X, th, L, a, O.KIND_NAMES[kind])
    assert np.max(np.abs(m[:, 0] - mo)) <= 1e-11 * np.max(np.abs(mo))
    assert np.max(np.abs(v[:, 0] - vo) / np.abs(vo)) < 1e-10      # element-wise, not norm-wise
    val, grad = eng.lml(th[None, :])
    vo_, go_ = O.lml(th, X, z, O.KIND_NAMES[kind], 0.1, eval_gradient=True)
    assert abs(val[0] - vo_) <= 1e-10 * abs(vo_)
    assert np.max(np.abs(grad[0] - go_)) <= 1e-9 * max(np.max(np.abs(go_)), 1.0)
    eng.close()


def test_argument_errors_are_reported_not_crashes():
    from gpbayestools_hic_amd import GPEngine
    from gpbayestools_hic_amd._native import GPBError
    eng = GPEngine(0)
    with pytest.raises(GPBError):
        eng.predict(np.zeros((3, 2)))                   # before set_data / factor
    eng.set_data(np.random.rand(10, 2), np.zeros((1, 10)))
    with pytest.raises(GPBError):
        eng.set_theta(np.full((1, 4), np.nan))
    with pytest.raises(GPBError):
        eng.factor()                                    # theta never set
    with pytest.raises(GPBError):
        eng.set_data(np.random.rand(10, 65), np.zeros((1, 10)))     # d > 64
    eng.close()


def test_two_walker_ensemble_runs():
    import types
    from gpbayestools_hic_amd import StretchSampler
    fake = types.SimpleNamespace(ndim=1, device=0, min=np.array([-5.0]), max=np.array([5.0]), emuList=[])
    s = StretchSampler(fake, 2, seed=1, logprob_device=lambda X, out: out.copy_(-0.5 * (X * X).sum(1)))
    s.run(np.array([[0.1], [-0.2]]), 50)
    assert s.chain.shape == (2, 50, 1) and np.all(np.isfinite(s.chain))


def test_long_host_batches_go_through_in_slabs(tmp_path):
    """a stored chain handed to log_likelihood_point_by_point (src/mcmc.py:225-258, 729-749) is longer than the
    slab the K*^T workspace is sized for: same numbers as row-by-row pieces, outside rows included"""
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd.workload import build_chain
    chain, emu, info = build_chain(1, workdir=str(tmp_path))
    W = (1 << 17) + 37                                     # one row more than a slab, and a ragged tail
    X = synth.walkers(W, info["d"], seed=11)
    X[::1001, 0] = 1.5
    lp = chain.log_likelihood_point_by_point(X)
    assert lp.shape == (W,) and np.all(np.isneginf(lp[::1001])) and np.all(np.isfinite(np.delete(lp, np.s_[::1001])))
    for sl in (slice(0, 300), slice((1 << 17) - 50, (1 << 17) + 37)):
        assert np.array_equal(chain.log_posterior(X[sl]), lp[sl])
    Xin = np.clip(X, 0.0, 1.0)
    mean = emu.predict(Xin, return_cov=False)              # Emulator.predict slabs the same way
    assert mean.shape == (W, emu.nobs)
    assert np.array_equal(mean[-60:], emu.predict(Xin[-60:], return_cov=False))
    m2, c2 = emu.predict(Xin[:40], return_cov=True, extra_std=0.0)
    assert np.array_equal(m2, mean[:40]) and c2.shape == (40, emu.nobs, emu.nobs)
