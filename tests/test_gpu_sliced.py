"""
GPU tier: the int8 form of V = L^-1 K*^T (option key 51, csrc/gpb_sliced.hip — six signed 8-bit digit planes per operand on
v_mfma_i32_32x32x32_i8, 21 exact digit products, fp64 combine) at the bars of the fp64 path it stands in for
(sk:_gpr.py:454-460, src/emulator.py:553,573-575): mean 1e-11, variance 1e-10 relative, log-posterior 1e-10 — against the
oracle, with shapes that exercise the row padding to 128, the front padding of the design, ragged batches and compaction.
The whole GPU suite also runs with the option forced on (GPB_PREDICT_SLICED=1: profiles/r06_sliced_suite.txt).
"""
import numpy as np
import pytest

from conftest import maxrel, relerr

pytestmark = pytest.mark.gpu


def _problem(N, d, P, kind, W, seed, sn2=0.05, c=1.0):
    from gpbayestools_hic_amd import synth
    rng = np.random.default_rng(seed)
    X = synth.lhs(N, d, seed=seed)
    Z = np.sin(X @ rng.standard_normal((d, P))).T + 0.05 * rng.standard_normal((P, N))
    th = synth.fixed_theta(d, P, ell=1.2, noise=sn2)
    th[:, 0] = np.log(c) + 0.1 * rng.standard_normal(P)
    Xs = rng.random((W, d))
    k = min(W, N, 16)
    Xs[:k] = X[:k]                                            # queries ON design points: the smallest variances
    return X, Z, th, Xs


@pytest.mark.parametrize("N,d,P,kind,W", [
    (1000, 15, 4, "RBF", 515),          # Np = 1024: 16 + 8 rows of padding in front and behind; ragged batch
    (320, 20, 3, "Matern15", 130),      # Np = 320: the last 128-row block is half empty
    (65, 3, 2, "Matern25", 1),          # Np = 128, one walker
    (40, 4, 2, "RBF", 70),              # Np = 64: half a row block, the K loop two steps long
    (900, 12, 3, "RBF", 300),           # Np = 960: 48 rows of padding in front, a whole 32-deep K-step of it skipped
    (2048, 20, 10, "RBF", 256),         # cfg 4's GPs on a rank's share of eight
    (640, 8, 9, "RBF", 1024),           # nine GPs: more than one super-block per row group
])
def test_int8_predict_against_the_oracle(N, d, P, kind, W):
    from gpbayestools_hic_amd import GPEngine
    from oracle import gp_oracle as O
    X, Z, th, Xs = _problem(N, d, P, kind, W, seed=N + W)
    eng = GPEngine(0)
    eng.set_data(X, Z, kind, alpha=0.1); eng.set_theta(th); eng.factor()
    eng.tune("predict_sliced", 0)                              # (whatever GPB_PREDICT_SLICED says: the fp64 kernel first)
    m64, v64 = eng.predict(Xs)
    eng.tune("predict_sliced", 1)
    m, v = eng.predict(Xs)
    assert not np.array_equal(v, v64)                          # the int8 kernel did run (the rule admits these GPs) ...
    assert relerr(v, v64) < 2e-11 and maxrel(m, m64) < 1e-13   # ... and stays close to the fp64 kernel
    kid = O.KIND_NAMES[kind]
    rows = np.unique(np.r_[0:min(W, 12), np.random.default_rng(1).choice(W, min(W, 20), replace=False)])
    for p in range(P):
        L, a = O.gp_factor(X, Z[p], th[p], kid, 0.1)
        mo, vo = O.gp_predict(Xs[rows], X, th[p], L, a, kid)
        assert maxrel(m[rows, p], mo) < 1e-11
        assert relerr(v[rows, p], vo) < 1e-10
    assert np.all(v > 0)
    # a walker's bits do not depend on the batch it arrives in: integer sums are exact, the fp64 epilogue's order is the row's
    for lo, hi in ((0, W // 2), (W // 2, W), (min(3, W - 1), W)):
        if hi > lo:
            mm, vv = eng.predict(Xs[lo:hi])
            assert np.array_equal(vv, v[lo:hi]) and np.array_equal(mm, m[lo:hi])
    # the accessor reads K*^T back from the digit planes (what the int8 kernel sees: within 2^(e_c - 48) of the fp64 values)
    Ks = eng.get("Kstar", hi - lo)
    for p in range(min(P, 2)):
        assert np.max(np.abs(Ks[p] - O.kernel_cross(Xs[lo:hi], X, th[p], kid))) < 1e-13 * max(1.0, float(np.exp(th[p, 0])))
    # the joint covariance reads the fp64 K*^T itself: that call runs the fp64 cross kernel whatever the option says
    if W >= 8:
        mc, cov = eng.predict_cov(Xs[:8])
        eng.tune("predict_sliced", 0)
        mc0, cov0 = eng.predict_cov(Xs[:8])
        eng.tune("predict_sliced", 1)
        assert np.array_equal(np.asarray(cov), np.asarray(cov0)) and np.array_equal(mc, mc0)
        assert relerr(np.diagonal(np.asarray(cov), axis1=1, axis2=2).T, v64[:8]) < 1e-9
    eng.close()


def test_the_rule_keeps_a_high_cancellation_context_on_the_fp64_kernel():
    """1 + c / sn2 > 128 for one GP of the context: every GP stays on the fp64 kernel (theta alone decides, never the batch);
    with the rule switched off (value 2, test hook) the int8 kernel runs there and is visibly less accurate"""
    from gpbayestools_hic_amd import GPEngine
    from oracle import gp_oracle as O
    X, Z, th, Xs = _problem(500, 5, 3, "RBF", 256, seed=9, sn2=0.05)
    th[1, 0], th[1, -1] = 3.0, np.log(1e-2)                     # c = 20, sn2 = 0.01: the worst corner of the search box
    th[:, 1:-1] = np.log(8.0)
    Xs[:64] = X[:64] + 1e-6
    eng = GPEngine(0)
    eng.set_data(X, Z, "RBF", alpha=0.1); eng.set_theta(th); eng.factor()
    eng.tune("predict_sliced", 0)
    m64, v64 = eng.predict(Xs)
    eng.tune("predict_sliced", 1)
    m1, v1 = eng.predict(Xs)
    assert np.array_equal(v1, v64) and np.array_equal(m1, m64)  # not admitted: the fp64 path, bit for bit
    L, a = O.gp_factor(X, Z[1], th[1], O.KIND_RBF, 0.1)
    _, vo = O.gp_predict(Xs[:64], X, th[1], L, a, O.KIND_RBF)
    assert relerr(v64[:64, 1], vo) < 1e-10
    eng.tune("predict_sliced", 2)
    _, v2 = eng.predict(Xs)
    e2 = relerr(v2[:64, 1], vo)
    assert 1e-12 < e2 < 1e-8                                    # 21 products at a cancellation factor of ~2000: what the rule is for
    th[1, 0], th[1, -1] = 0.0, np.log(0.05)                     # back inside the rule's range: admitted again
    eng.tune("predict_sliced", 1)
    eng.set_theta(th); eng.factor()
    _, v64b = (lambda e: (e.tune("predict_sliced", 0), e.predict(Xs))[1])(eng)
    eng.tune("predict_sliced", 1)
    _, v3 = eng.predict(Xs)
    assert not np.array_equal(v3, v64b) and relerr(v3, v64b) < 2e-11
    eng.close()


def test_chain_log_posterior_with_the_int8_kernel(tmp_path):
    """cfg 4's emulator at N = 320 through the drop-in classes: Chain.log_posterior / log_likelihood batches with rows outside
    the box (compaction on the device), against the oracle and against the sampler's own C-driven loop"""
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd.workload import build_chain
    from oracle import gp_oracle as O
    chain, emu, info = build_chain(4, workdir=str(tmp_path), N=320)
    eng = emu._engine_ready()
    d = info["d"]
    X = synth.walkers(300, d, seed=5)
    X[::7, 3] = 1.5; X[11, 0] = 0.0                            # outside / on the boundary
    eng.tune("predict_sliced", 0)
    lp64 = chain.log_posterior(X)
    eng.tune("predict_sliced", 1)
    lp = chain.log_posterior(X)
    assert np.array_equal(np.isneginf(lp), np.isneginf(lp64))
    fin = np.isfinite(lp)
    assert relerr(lp[fin], lp64[fin]) < 1e-10 and not np.array_equal(lp[fin], lp64[fin])
    oe = O.OracleEmulator(info["X"], info["Y"], info["lo"], info["hi"], info["P"], O.KIND_RBF).fit(synth.fixed_theta(d, info["P"]))
    yexp = info["yexp"]; cexp = np.diag((0.05 * np.abs(yexp)) ** 2)
    ref = O.log_prob(X, info["lo"], info["hi"], lambda x, e: oe.predict(x, True, e), yexp, cexp)
    assert np.array_equal(np.isneginf(ref), ~fin) and relerr(lp[fin], ref[fin]) < 1e-10
    ll = chain.log_likelihood(X, finite=True)
    assert np.all(ll[~fin] == -1e300) and np.array_equal(ll[fin], lp[fin])
    # compaction does not change a row's bits: the same rows alone, and one by one
    assert np.array_equal(chain.log_posterior(X[fin]), lp[fin])
    assert np.array_equal(np.asarray(chain.log_likelihood_point_by_point(X[:9])).reshape(-1), np.asarray(chain.log_likelihood(X[:9])).reshape(-1))
    # the stretch move's C-driven loop and the host-driven loop agree bit for bit with the int8 kernel inside
    from gpbayestools_hic_amd.sampler import StretchSampler
    X0 = synth.walkers_ball(256, info["xstar"], 1e-3, lo=info["lo"], hi=info["hi"])
    a = StretchSampler(chain, 256, seed=77)
    pa = a.run(X0, 6, status=10 ** 9, store=False)
    assert a._resident_engine() is not None
    b = StretchSampler(chain, 256, seed=77)
    b._resident_engine = lambda: None                         # force the host-driven loop
    pb = b.run(X0, 6, status=10 ** 9, store=False)
    assert np.array_equal(pa, pb)
    eng.tune("predict_sliced", 0)


def test_the_public_choice_of_arithmetic_survives_pickling_and_a_new_engine(tmp_path):
    """Emulator.set_predict_arithmetic / Chain.set_predict_arithmetic: the int8 kernel through the drop-in classes, kept by a
    pickled emulator (whose engine is rebuilt on load) and part of the state digest the replicas of a sharded run compare"""
    import dill
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd.workload import build_chain
    chain, emu, info = build_chain(4, workdir=str(tmp_path), N=320)
    emu._engine_ready().tune("predict_sliced", 0)              # (whatever GPB_PREDICT_SLICED says)
    X = synth.walkers(64, info["d"], seed=3)
    lp64, d64 = chain.log_posterior(X), emu.state_digest()
    assert chain.set_predict_arithmetic("int8") is chain and emu.predict_arithmetic == "int8"
    lp8, d8 = chain.log_posterior(X), emu.state_digest()
    assert d8 != d64 and not np.array_equal(lp8, lp64) and relerr(lp8, lp64) < 1e-10
    again = dill.loads(dill.dumps(emu))
    assert again.predict_arithmetic == "int8" and again.state_digest() == d8
    m8, c8 = emu.predict(X, return_cov=True)
    ma, ca = again.predict(X, return_cov=True)
    assert np.array_equal(np.asarray(ma), np.asarray(m8)) and np.array_equal(np.asarray(ca), np.asarray(c8))
    chain.set_predict_arithmetic("fp64")
    assert emu.state_digest() == d64 and np.array_equal(chain.log_posterior(X), lp64)
    with pytest.raises(ValueError):
        emu.set_predict_arithmetic("bf16")
