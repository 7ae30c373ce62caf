"""
GPU tier at BASELINE full sizes (config 4: N=2048, d=20, P=10 -> here P=3 to keep host memory small;
config 5 kernel: Matern-5/2 at N=4096): size-independent properties instead of an oracle run.
  * factorisation residual  K v == L (L^T v)  on random probes, L^-1 L == I on probes
  * GP identity at the training points:  mean(x_i) = z_i - (sigma_n^2 + alpha) * alpha_i   (K alpha = z)
  * 0 < var <= prior, var at training points equals  prior - [K_ii' - 2 s + s^2 (K^-1)_ii]
  * linearity of alpha and of the predictive mean in the training targets
  * bit-identical results for any batch split
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(eng, N, d, P, kernel, seed):
    from gpbayestools_hic_amd import synth
    rng = np.random.default_rng(seed)
    X = synth.lhs(N, d, seed=seed)
    Z = np.sin(X @ rng.standard_normal((d, P))).T + 0.05 * rng.standard_normal((P, N))
    th = synth.fixed_theta(d, P) + 0.1 * rng.standard_normal((P, d + 2))
    eng.set_data(X, Z, kernel, alpha=0.1)
    eng.set_theta(th)
    eng.factor()
    return X, Z, th


@pytest.mark.parametrize("N,d,kernel", [(2048, 20, "RBF"), (4096, 20, "Matern25")])
def test_fullsize_factor_and_predict_properties(N, d, kernel):
    from gpbayestools_hic_amd import GPEngine
    from oracle import gp_oracle as O
    P = 2
    eng = GPEngine(0)
    X, Z, th = _setup(eng, N, d, P, kernel, seed=N)
    rng = np.random.default_rng(1)
    L = eng.get("L"); Linv = eng.get("Linv"); alpha = eng.get("alpha")
    kid = O.KIND_NAMES[kernel]
    for p in range(P):
        K = O.kernel_train(X, th[p], kid, 0.1)                      # oracle kernel matrix only (cheap, elementwise)
        v = rng.standard_normal((N, 4))
        assert np.max(np.abs(L[p] @ (L[p].T @ v) - K @ v)) < 1e-11 * np.max(np.abs(K @ v))
        assert np.max(np.abs(Linv[p] @ (L[p] @ v) - v)) < 1e-10 * np.max(np.abs(v))
        assert np.max(np.abs(K @ alpha[p] - Z[p])) < 1e-10 * np.max(np.abs(Z[p]))
    idx = rng.choice(N, 300, replace=False)
    m, var = eng.predict(X[idx])
    for p in range(P):
        c, noise = np.exp(th[p, 0]), np.exp(th[p, -1])
        s = noise + 0.1
        assert np.max(np.abs(m[:, p] - (Z[p, idx] - s * alpha[p, idx]))) < 1e-10 * np.max(np.abs(Z[p]))
        kinv_ii = np.einsum("ki,ki->i", Linv[p][:, idx], Linv[p][:, idx])
        expect = (c + noise) - ((c + s) - 2 * s + s * s * kinv_ii)
        assert np.max(np.abs(var[:, p] - expect) / np.abs(expect)) < 1e-9
        assert np.all(var[:, p] > 0) and np.all(var[:, p] <= c + noise)
    Xs = rng.random((1000, d))
    m1, v1 = eng.predict(Xs)
    assert np.all(v1 > 0)
    for sl in (slice(0, 1), slice(17, 400), slice(400, 1000)):
        ms, vs = eng.predict(Xs[sl])
        assert np.array_equal(ms, m1[sl]) and np.array_equal(vs, v1[sl])
    eng.close()


def test_fullsize_linearity_in_targets():
    from gpbayestools_hic_amd import GPEngine, synth
    N, d = 2048, 20
    rng = np.random.default_rng(5)
    X = synth.lhs(N, d, seed=9)
    z1, z2 = rng.standard_normal(N), rng.standard_normal(N)
    eng = GPEngine(0)
    eng.set_data(X, np.stack([z1, z2, 2.0 * z1 - 0.5 * z2]), "RBF", 0.1)
    eng.set_theta(synth.fixed_theta(d, 3)); eng.factor()
    a = eng.get("alpha")
    assert np.max(np.abs(a[2] - (2.0 * a[0] - 0.5 * a[1]))) < 1e-11 * np.max(np.abs(a))
    Xs = rng.random((500, d))
    m, v = eng.predict(Xs)
    assert np.max(np.abs(m[:, 2] - (2.0 * m[:, 0] - 0.5 * m[:, 1]))) < 1e-11 * np.max(np.abs(m))
    assert np.array_equal(v[:, 0], v[:, 1]) and np.array_equal(v[:, 0], v[:, 2])   # variance ignores the targets
    eng.close()


def test_fullsize_log_posterior_cfg4_against_oracle_sample():
    """cfg 4 end to end (N=2048, d=20, M=64, P=10): 48 rows against the oracle (seconds on the host)."""
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd.workload import build_chain
    from oracle import gp_oracle as O
    chain, emu, info = build_chain(4)
    d, P = info["d"], info["P"]
    oe = O.OracleEmulator(info["X"], info["Y"], info["lo"], info["hi"], P).fit(synth.fixed_theta(d, P))
    Xw = synth.walkers(48, d, seed=77)
    Xw[5, 3] = 1.5; Xw[9, 0] = 0.0                                   # outside / on the boundary
    yexp = info["yexp"]; cexp = np.diag((0.05 * np.abs(yexp)) ** 2)
    ref = O.log_prob(Xw, info["lo"], info["hi"], lambda x, e: oe.predict(x, True, e), yexp, cexp)
    got = chain.log_posterior(Xw)
    ins = np.isfinite(ref)
    assert np.array_equal(np.isneginf(got), ~ins) and ins.sum() == 46
    assert np.max(np.abs(got[ins] - ref[ins]) / np.abs(ref[ins])) < 1e-10


def test_fullsize_log_posterior_cfg4_without_pca_p64_against_oracle_sample():
    """SURVEY 8(a)'s second shape of cfg 4: perform_no_PCA, one GP per observable (P = 64 GPs of N = 2048), the covariance left
    in standardized units as the reference leaves it (src/emulator.py:562-565,589-592), the dense 64 x 64 likelihood kernels
    instead of the low-rank form: 48 rows against the oracle (its 64 factorisations: about a minute on the host)."""
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd.workload import build_chain
    from oracle import gp_oracle as O
    chain, emu, info = build_chain(4, no_pca=True)
    d, P = info["d"], info["P"]
    assert P == info["M"] == 64 and emu.perform_no_PCA_ and len(emu.gps) == 64
    oe = O.OracleEmulator(info["X"], info["Y"], info["lo"], info["hi"], P, mode=O.MODE_NO_PCA).fit(synth.fixed_theta(d, P))
    Xw = synth.walkers(48, d, seed=78)
    Xw[5, 3] = 1.5; Xw[9, 0] = 0.0                                   # outside / on the boundary
    yexp = info["yexp"]; cexp = np.diag((0.05 * np.abs(yexp)) ** 2)
    ref = O.log_prob(Xw, info["lo"], info["hi"], lambda x, e: oe.predict(x, True, e), yexp, cexp)
    got = chain.log_posterior(Xw)
    ins = np.isfinite(ref)
    assert np.array_equal(np.isneginf(got), ~ins) and ins.sum() == 46
    assert np.max(np.abs(got[ins] - ref[ins]) / np.abs(ref[ins])) < 1e-10
    # the emulator's own outputs in this mode: mean in data units, covariance = diag(gp variance), NOT rescaled
    m, c = emu.predict(Xw[:6], return_cov=True, extra_std=0.0)
    mo, co = oe.predict(Xw[:6], True, 0.0)
    assert np.max(np.abs(m - mo) / np.abs(mo)) < 1e-11 and np.max(np.abs(c - co)) < 1e-10 * np.max(np.abs(co))
    # a 2048-row batch gives the same bits as its rows in small batches (the dense likelihood kernels, the 64-GP predict launch)
    Xb = synth.walkers(2048, d, seed=79)
    lp = chain.log_posterior(Xb)
    assert np.array_equal(chain.log_posterior(Xb[100:164]), lp[100:164]) and np.all(np.isfinite(lp))


@pytest.mark.parametrize("N,library", [(1024, "product"), (2048, "product"), (1024, "debug")])
def test_cholesky_schedules_agree(N, library):
    """the blocked Cholesky's schedules — two launches per step with the next diagonal block fused into the update
    (default), with and without the lookahead side stream — give the same factor bit
    for bit at equal outer panel width (the side stream only touches tiles nobody else touches at that time; the pivot
    arithmetic is round 1's), run to run; another panel width groups the trailing updates differently and agrees to
    rounding; and the factor is correct"""
    from gpbayestools_hic_amd import GPEngine
    from oracle import gp_oracle as O
    P, d = 3, 12
    if library == "debug":
        from conftest import debug_engine
        eng = debug_engine()
    else:
        eng = GPEngine(0)
    X, Z, th = _setup(eng, N, d, P, "RBF", seed=3 * N)
    auto_L, auto_X = eng.get("L"), eng.get("Linv")            # the default: panel width chosen by size
    eng.tune("chol_outer", 512)
    eng.factor()
    ref_L, ref_X = eng.get("L"), eng.get("Linv")
    assert np.max(np.abs(auto_L - ref_L)) < 1e-12 * np.max(np.abs(ref_L))
    assert np.max(np.abs(auto_X - ref_X)) < 1e-11 * np.max(np.abs(ref_X))
    for algo, outer, look in ((1, 512, 1), (1, 512, 0), (1, 512, 1), (1, 256, 1), (1, 256, 0), (1, 128, 1), (1, 0, 1)):
        eng.tune("chol_outer", outer); eng.tune("chol_lookahead", look)
        eng.factor()
        L, Xi = eng.get("L"), eng.get("Linv")
        if outer == 0:
            assert np.array_equal(L, auto_L) and np.array_equal(Xi, auto_X)       # run to run
        if outer == 512:
            assert np.array_equal(L, ref_L) and np.array_equal(Xi, ref_X), (algo, outer, look)
        else:
            assert np.max(np.abs(L - ref_L)) < 1e-12 * np.max(np.abs(ref_L)), (algo, outer, look)
            assert np.max(np.abs(Xi - ref_X)) < 1e-11 * np.max(np.abs(ref_X)), (algo, outer, look)
        if outer == 256 and look == 1:
            L256 = L
        if outer == 256 and look == 0:
            assert np.array_equal(L, L256)
    eng.tune("chol_outer", 0); eng.tune("chol_lookahead", 1)
    # column pairs (every second trailing update by two block columns at once: another grouping of the same sums) on and off
    pair = {}
    for mode in (2, 0, 2):
        eng.tune("chol_pair", mode)
        eng.factor()
        L, Xi = eng.get("L"), eng.get("Linv")
        assert np.max(np.abs(L - ref_L)) < 1e-12 * np.max(np.abs(ref_L)) and np.max(np.abs(Xi - ref_X)) < 1e-11 * np.max(np.abs(ref_X))
        if mode in pair:
            assert np.array_equal(L, pair[mode][0]) and np.array_equal(Xi, pair[mode][1])         # run to run
        pair[mode] = (L, Xi)
    eng.tune("chol_pair", 1)
    K = O.kernel_train(X, th[0], O.KIND_RBF, 0.1)
    Lo = np.linalg.cholesky(K)
    assert np.max(np.abs(ref_L[0] - Lo)) < 1e-11 * np.max(np.abs(Lo))
    v = np.random.default_rng(0).standard_normal((N, 3))
    assert np.max(np.abs(ref_X[0] @ (ref_L[0] @ v) - v)) < 1e-10
    # an indefinite matrix is reported with LAPACK's info (first non-positive pivot), whatever block it falls in
    eng.set_data(X, Z, "RBF", alpha=-1.5)
    eng.set_theta(th)
    info = eng.factor(raise_on_fail=False)
    Kbad = O.kernel_train(X, th[0], O.KIND_RBF, -1.5)
    from scipy.linalg import lapack
    _, ref_info = lapack.dpotrf(Kbad, lower=1)
    assert ref_info > 0 and info[0] == ref_info
    eng.close()


@pytest.mark.parametrize("N", [1601, 1700, 3072, 129, 190])
def test_cholesky_by_column_pairs_at_odd_and_even_block_counts(N):
    """the pair schedule (k_chol_update2: every second trailing update by two block columns at once) where the rule takes it
    (1024 <= Np <= 3072: 26, 27 and 48 block columns) and forced at three and three-with-a-ragged-end block columns: the factor
    against LAPACK's, the inverse against it, LAPACK's info on an indefinite matrix"""
    from gpbayestools_hic_amd import GPEngine
    from oracle import gp_oracle as O
    from scipy.linalg import lapack
    P, d = 2, 6
    eng = GPEngine(0)
    if N < 1024:
        eng.tune("chol_pair", 2)
    X, Z, th = _setup(eng, N, d, P, "Matern15", seed=N)
    L, Xi = eng.get("L"), eng.get("Linv")
    for p in range(P):
        K = O.kernel_train(X, th[p], O.KIND_MATERN15, 0.1)
        Lo = np.linalg.cholesky(K)
        assert np.max(np.abs(L[p] - Lo)) < 1e-11 * np.max(np.abs(Lo))
        v = np.random.default_rng(p).standard_normal((N, 2))
        assert np.max(np.abs(Xi[p] @ (Lo @ v) - v)) < 1e-9
    eng.set_data(X, Z, "Matern15", alpha=-1.2)
    eng.set_theta(th)
    info = eng.factor(raise_on_fail=False)
    _, ref_info = lapack.dpotrf(O.kernel_train(X, th[1], O.KIND_MATERN15, -1.2), lower=1)
    assert ref_info > 0 and info[1] == ref_info
    eng.close()
