"""
GPU tier (MI355X): the HIP path through the C ABI against (a) the golden vectors captured from
the reference and (b) the CPU oracle on seeded inputs.  Tolerances (fp64):
kernels/L/alpha 1e-11, mean 1e-11, var/cov 1e-10 relative, LML 1e-10, log-likelihood 1e-10.
"""
import numpy as np
import pytest

from conftest import golden, relerr, maxrel

pytestmark = pytest.mark.gpu

KINDS = {"rbf": ("RBF", 0), "m15": ("Matern15", 1), "m25": ("Matern25", 2)}


@pytest.fixture(scope="module")
def eng():
    from gpbayestools_hic_amd import GPEngine
    e = GPEngine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def deng():
    """an engine on the debug library: the self-test GEMM, the tile trace and the measured-and-rejected kernel variants"""
    from conftest import debug_engine
    e = debug_engine()
    yield e
    e.close()


# ---------------------------------------------------------------- MFMA tile engine
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("shape", [(128, 128, 16), (256, 384, 64), (192, 130, 48), (64, 64, 64), (130, 66, 32)])
def test_mfma_tile_gemm(deng, mode, shape):
    """Asymmetric operands catch a transposed C/D fragment map (v_mfma_f64_16x16x4_f64)."""
    eng = deng
    M, N, K = shape
    rng = np.random.default_rng(M * 7 + N * 3 + K + mode)
    A = rng.standard_normal((M, K)); B = rng.standard_normal((K, N))
    ref = A @ B
    if mode == 0:
        got = eng.test_gemm(A, B, 0)
    elif mode == 1:
        got = eng.test_gemm(A, np.ascontiguousarray(B.T), 1)
    else:
        got = eng.test_gemm(np.ascontiguousarray(A.T), B, 2)
    assert maxrel(got, ref) < 1e-13


def test_mfma_exact_integers(deng):
    eng = deng
    rng = np.random.default_rng(5)
    A = rng.integers(-8, 9, (128, 32)).astype(float); B = rng.integers(-8, 9, (32, 128)).astype(float)
    assert np.array_equal(eng.test_gemm(A, B, 0), A @ B)


# ---------------------------------------------------------------- G1/G2: kernels, factor, LML, predict
@pytest.mark.parametrize("name", list(KINDS))
def test_g1_kernel_matrix_via_factor(eng, name):
    g = golden("g1_kernels.npz")
    X, Xs, th = g["X"], g["Xs"], g["theta"]
    eng.set_data(X, np.zeros((1, X.shape[0])), KINDS[name][0], alpha=0.0)
    eng.set_theta(th[None, :])
    eng.factor()
    L = eng.get("L")[0]
    assert maxrel(L @ L.T, g[f"{name}_K"]) < 1e-13          # K = L L^T reproduces sklearn's K(X,X)
    Linv = eng.get("Linv")[0]
    assert np.max(np.abs(Linv @ L - np.eye(L.shape[0]))) < 1e-11
    assert np.all(np.triu(Linv, 1) == 0.0)


@pytest.mark.parametrize("name", list(KINDS))
@pytest.mark.parametrize("i", [0, 1, 2])
def test_g2_factor_lml_predict(eng, name, i):
    g = golden("g2_gpr.npz")
    X, z, Xs, th = g["X"], g["z"], g["Xs"], g["thetas"][i]
    eng.set_data(X, z[None, :], KINDS[name][0], alpha=float(g["alpha"]))
    val, grad = eng.lml(th[None, :])
    assert abs(val[0] - g[f"{name}_{i}_lml"]) < 1e-10 * abs(g[f"{name}_{i}_lml"])
    assert maxrel(grad[0], g[f"{name}_{i}_grad"]) < 1e-9
    eng.set_theta(th[None, :]); eng.factor()
    assert maxrel(eng.get("alpha")[0], g[f"{name}_{i}_alpha_"]) < 1e-10
    if i == 0:
        assert maxrel(eng.get("L")[0], g[f"{name}_{i}_L"]) < 1e-11
    m, v = eng.predict(Xs)
    assert maxrel(m[:, 0], g[f"{name}_{i}_mean"]) < 1e-11
    assert relerr(v[:, 0], g[f"{name}_{i}_var"]) < 1e-10


def test_not_positive_definite_is_reported(eng):
    from gpbayestools_hic_amd.engine import NotPositiveDefinite
    X = np.zeros((70, 3)); X[:, 0] = np.linspace(0, 1e-9, 70)    # (near-)duplicate points, no jitter
    eng.set_data(X, np.zeros((2, 70)), "RBF", alpha=-1.02)
    th = np.array([[0.0, 0, 0, 0, np.log(1e-2)]] * 2)
    eng.set_theta(th)
    with pytest.raises(NotPositiveDefinite):
        eng.factor()
    info = eng.factor(raise_on_fail=False)
    assert np.all(info > 0)
    from gpbayestools_hic_amd._native import GPBError
    with pytest.raises(GPBError):                                 # no factorisation installed: nothing runs on NaN factors
        eng.predict(np.zeros((2, 3)))
    val, grad = eng.lml(th)
    assert np.all(np.isneginf(val)) and np.all(grad == 0.0)       # sk:_gpr.py:588-589


# ---------------------------------------------------------------- G6: batched MVN
@pytest.mark.parametrize("M", [4, 16, 64])
def test_g6_mvn(eng, M):
    g = golden("g6_mvn.npz")
    got = eng.mvn_loglike(g[f"y_{M}"], g[f"cov_{M}"])
    assert relerr(got, g[f"ll_{M}"]) < 1e-11
    assert eng.last_not_pd == 0
    bad = g[f"cov_{M}"].copy(); bad[1] = -np.eye(M)
    got = eng.mvn_loglike(g[f"y_{M}"], bad)
    assert np.isnan(got[1]) and eng.last_not_pd == 1 and np.isfinite(got[0])


def test_mvn_large_block_global_slab(eng):
    """M > 128 leaves LDS for an HBM slab (real analyses reach sum M ~ 540)."""
    from oracle import gp_oracle as O
    rng = np.random.default_rng(9)
    M, W = 200, 5
    B = rng.standard_normal((W, M, M))
    cov = B @ B.transpose(0, 2, 1) / M + np.eye(M) * 0.1
    y = rng.standard_normal((W, M))
    ref = np.array([O.mvn_loglike(a, c) for a, c in zip(y, cov)])
    assert relerr(eng.mvn_loglike(y, cov), ref) < 1e-10


# ---------------------------------------------------------------- oracle parity at larger, ragged sizes
@pytest.mark.parametrize("N,d,P,W,kind", [(200, 5, 3, 77, "RBF"), (333, 11, 2, 300, "Matern15"),
                                          (1024, 15, 2, 513, "RBF"), (130, 20, 4, 1, "Matern25")])
def test_predict_matches_oracle(eng, N, d, P, W, kind):
    from oracle import gp_oracle as O
    from gpbayestools_hic_amd import synth
    kid = O.KIND_NAMES[kind]
    rng = np.random.default_rng(N + d)
    X = synth.lhs(N, d, seed=N)
    Z = np.sin(X @ rng.standard_normal((d, P))).T + 0.05 * rng.standard_normal((P, N))
    th = np.array([np.concatenate([[rng.uniform(-0.3, 0.5)], np.log(rng.uniform(0.5, 2.0, d)),
                                   [np.log(rng.uniform(0.02, 0.1))]]) for _ in range(P)])
    Xs = rng.random((W, d))
    eng.set_data(X, Z, kind, alpha=0.1); eng.set_theta(th); eng.factor()
    m, v = eng.predict(Xs)
    for p in range(P):
        L, a = O.gp_factor(X, Z[p], th[p], kid, 0.1)
        mo, vo = O.gp_predict(Xs, X, th[p], L, a, kid)
        assert maxrel(m[:, p], mo) < 1e-11
        assert relerr(v[:, p], vo) < 1e-10
    # empty batch and mean-only
    assert eng.predict(np.zeros((0, d)))[0].shape == (0, P)
    assert maxrel(eng.predict(Xs, return_var=False), m) == 0.0


def test_predict_is_batch_independent(eng):
    """A walker's numbers must not depend on how the batch is cut (bit-identical sharding)."""
    from gpbayestools_hic_amd import synth
    N, d, P = 300, 6, 2
    rng = np.random.default_rng(3)
    X = synth.lhs(N, d, seed=1)
    Z = rng.standard_normal((P, N))
    eng.set_data(X, Z, "RBF", 0.1); eng.set_theta(synth.fixed_theta(d, P)); eng.factor()
    Xs = rng.random((400, d))
    m, v = eng.predict(Xs)
    for sl in (slice(0, 1), slice(5, 133), slice(200, 400), slice(399, 400)):
        ms, vs = eng.predict(Xs[sl])
        assert np.array_equal(ms, m[sl]) and np.array_equal(vs, v[sl])


def test_lml_gradient_matches_oracle_and_finite_difference(eng):
    from oracle import gp_oracle as O
    from gpbayestools_hic_amd import synth
    N, d, P = 190, 7, 3
    rng = np.random.default_rng(17)
    X = synth.lhs(N, d, seed=4)
    Z = np.sin(X @ rng.standard_normal((d, P))).T
    th = np.array([np.concatenate([[0.2 * p], np.log(rng.uniform(0.6, 1.8, d)), [np.log(0.05)]]) for p in range(P)])
    for kind in ("RBF", "Matern15", "Matern25"):
        eng.set_data(X, Z, kind, 0.1)
        val, grad = eng.lml(th)
        for p in range(P):
            vo, go = O.lml(th[p], X, Z[p], O.KIND_NAMES[kind], 0.1, eval_gradient=True)
            assert abs(val[p] - vo) < 1e-10 * abs(vo)
            assert maxrel(grad[p], go) < 1e-9


def test_lml_gradient_with_more_than_thirty_parameters(eng):
    """d + 2 > 32 hyper-parameters: the gradient's final sum runs in two blocks of 32 (k_grad_final's grid.y); N = 700 puts
    66 tiles on its eight interleaved chunks"""
    from oracle import gp_oracle as O
    from gpbayestools_hic_amd import synth
    N, d, P = 700, 33, 2
    rng = np.random.default_rng(23)
    X = synth.lhs(N, d, seed=6)
    Z = np.sin(X @ rng.standard_normal((d, P)) * 0.5).T
    th = np.array([np.concatenate([[0.1 * p], np.log(rng.uniform(1.5, 3.0, d)), [np.log(0.05)]]) for p in range(P)])
    eng.set_data(X, Z, "RBF", 0.1)
    val, grad = eng.lml(th)
    for p in range(P):
        vo, go = O.lml(th[p], X, Z[p], O.KIND_NAMES["RBF"], 0.1, eval_gradient=True)
        assert abs(val[p] - vo) < 1e-10 * abs(vo)
        assert maxrel(grad[p], go) < 1e-9


# ---------------------------------------------------------------- fused log-likelihood: both MVN kernels vs the oracle
@pytest.mark.parametrize("M,P", [(4, 4), (13, 5), (32, 10), (41, 7), (64, 10), (64, 3), (100, 6),
                                 (20, 1), (24, 16), (24, 17), (3, 3)])      # low-rank form: P = 1, its largest P, one beyond
def test_loglike_fast_and_generic_paths(eng, M, P):
    from oracle import gp_oracle as O
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd.engine import MODE_PCA
    N, d, W = 256, 6, 203
    X = synth.lhs(N, d, seed=M)
    Y = synth.observables(X, M, seed=M + 1)
    oe = O.OracleEmulator(X, Y, np.zeros(d), np.ones(d), P).fit(synth.fixed_theta(d, P))
    eng.set_data(X, oe.Z.T, "RBF", 0.1); eng.set_theta(oe.thetas); eng.factor()
    eng.set_transform(MODE_PCA, oe.mu, A=oe.A, cov_trunc=oe.cov_trunc)
    yexp = oe.predict(synth.truth_point(d)[None, :], return_cov=False)[0]
    rng = np.random.default_rng(M * P)
    Bm = rng.standard_normal((M, M)) * 0.01
    cexp = np.diag((0.05 * np.abs(yexp)) ** 2) + Bm @ Bm.T       # full (non-diagonal) block is allowed
    eng.set_likelihood(yexp, cexp)
    Xw = synth.walkers(W, d, seed=5)
    mY, mC = oe.predict(Xw, True, np.zeros(W))
    ref = np.array([O.mvn_loglike(a, c) for a, c in zip(mY - yexp, mC + cexp)])
    # default for P <= 16: the low-rank form (a P x P Cholesky per walker: C = C0 + A^T D A with the same C0 for all)
    lowrank = eng.loglike(Xw).copy()
    assert eng.last_not_pd == 0
    assert relerr(lowrank, ref) < 1e-10
    eng.tune("fuse_finalize", 0); lr_unfused = eng.loglike(Xw).copy(); eng.tune("fuse_finalize", 1)
    assert np.array_equal(lr_unfused, lowrank)                   # the fused k_finalize sums keep k_finalize's order
    acc = eng.loglike(Xw, out=np.full(W, 2.5), accumulate=True)
    assert relerr(acc, lowrank + 2.5) < 1e-13
    eng.tune("lowrank", 0)                                       # the dense M x M kernels from here on
    fast = eng.loglike(Xw).copy()
    assert eng.last_not_pd == 0
    # two device paths to the same number (-1/2 q - 1/2 log det, both terms O(10 .. 100) and of either sign): the bar is
    # on the difference against the larger of |value| and 1 — a row whose terms cancel to ~0.1 is not 10x less accurate
    assert float(np.max(np.abs(lowrank - fast) / np.maximum(np.abs(fast), 1.0))) < 1e-12
    # the block kernels sum the predict partials themselves in k_finalize's order: same bits with and without the fusion
    eng.tune("fuse_finalize", 0); unfused = eng.loglike(Xw).copy(); eng.tune("fuse_finalize", 1)
    assert np.array_equal(unfused, fast)
    if 32 < M <= 64:
        # one wave per walker and one workgroup per walker apply the same operations to every element:
        # the batch-size switch between them never changes a bit
        eng.tune("mvn_wg_switch", 0); one_wave = eng.loglike(Xw).copy()
        eng.tune("mvn_wg_switch", 1 << 30); one_wg = eng.loglike(Xw).copy()
        eng.tune("mvn_wg_switch", 768)
        assert np.array_equal(one_wave, one_wg) and np.array_equal(one_wave, fast)
    eng.force_generic_mvn(True)
    gen = eng.loglike(Xw).copy()
    eng.force_generic_mvn(False)
    assert relerr(fast, ref) < 1e-10
    assert relerr(gen, ref) < 1e-10
    # accumulate=True adds onto an existing vector (multi-emulator blocks)
    acc = eng.loglike(Xw, out=np.full(W, 2.5), accumulate=True)
    assert relerr(acc, fast + 2.5) < 1e-13
    eng.tune("lowrank", 1)


def test_lowrank_loglike_reports_indefinite_covariances_like_the_dense_kernels(eng):
    """The low-rank form needs C0 = C_trunc + C_exp positive definite.  Nearly singular C0: both forms agree;
    indefinite C0: the set-up declines, the dense kernel runs and reports every row (NaN + count)."""
    from oracle import gp_oracle as O
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd.engine import MODE_PCA
    N, d, M, P, W = 128, 5, 12, 4, 150
    X = synth.lhs(N, d, seed=3); Y = synth.observables(X, M, seed=4)
    oe = O.OracleEmulator(X, Y, np.zeros(d), np.ones(d), P).fit(synth.fixed_theta(d, P))
    eng.set_data(X, oe.Z.T, "RBF", 0.1); eng.set_theta(oe.thetas); eng.factor()
    # a nearly singular C0 (1e-9 I + 1e-9 I): the hardest conditioning the low-rank set-up can be handed
    eng.set_transform(MODE_PCA, oe.mu, A=oe.A, cov_trunc=1e-9 * np.eye(M))
    yexp = oe.predict(synth.truth_point(d)[None, :], return_cov=False)[0]
    eng.set_likelihood(yexp, 1e-9 * np.eye(M))
    Xw = synth.walkers(W, d, seed=8)
    ok_lr = eng.loglike(Xw).copy(); n_lr = eng.last_not_pd
    eng.tune("lowrank", 0); ok_dn = eng.loglike(Xw).copy(); n_dn = eng.last_not_pd; eng.tune("lowrank", 1)
    assert n_lr == 0 and n_dn == 0 and relerr(ok_lr, ok_dn) < 1e-10
    eng.set_likelihood(yexp, -0.5 * np.eye(M))                   # C0 not positive definite: the low-rank set-up
    bad = eng.loglike(Xw)                                        # declines and the dense kernel reports the rows
    assert eng.last_not_pd == W and np.all(np.isnan(bad))


@pytest.mark.parametrize("c0", [1e-14, 1e-22])
def test_lowrank_logdet_does_not_overflow_for_tiny_c0(eng, c0):
    """npc = 16 (the low-rank form's largest) with C0 = c0 * I: the pivots of S = I + R D R^T are ~ var / c0, so all
    16 in one product overflow from ~1e19 each (c0 = 1e-22: the old kernel returned -inf silently); four per
    logarithm do not.  Reference: 60-digit Cholesky of the full M x M covariance (mpmath)."""
    mp = pytest.importorskip("mpmath")
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd.engine import MODE_PCA
    mp.mp.dps = 60
    N, d, M, P, W = 128, 4, 24, 16, 6
    rng = np.random.default_rng(12)
    X = synth.lhs(N, d, seed=5)
    Z = rng.standard_normal((P, N))
    A = rng.standard_normal((P, M)); mu = rng.standard_normal(M)
    eng.set_data(X, Z, "RBF", 0.1); eng.set_theta(synth.fixed_theta(d, P)); eng.factor()
    eng.set_transform(MODE_PCA, mu, A=A, cov_trunc=0.5 * c0 * np.eye(M))
    yexp = rng.standard_normal(M)
    eng.set_likelihood(yexp, 0.5 * c0 * np.eye(M))
    Xw = synth.walkers(W, d, seed=6)
    got = eng.loglike(Xw)
    assert eng.last_not_pd == 0 and np.all(np.isfinite(got))
    gm, gv = eng.predict(Xw)
    for w in range(W):
        C = mp.matrix(M, M)
        for i in range(M):
            for j in range(M):
                C[i, j] = mp.fsum(mp.mpf(gv[w, p]) * mp.mpf(A[p, i]) * mp.mpf(A[p, j]) for p in range(P))
            C[i, i] += mp.mpf(c0)
        y = mp.matrix([mp.fsum(mp.mpf(gm[w, p]) * mp.mpf(A[p, i]) for p in range(P)) + mp.mpf(mu[i]) - mp.mpf(yexp[i])
                       for i in range(M)])
        L = mp.cholesky(C)
        v = mp.lu_solve(L, y)
        ref = -mp.mpf(0.5) * mp.fsum(x * x for x in v) - mp.fsum(mp.log(L[i, i]) for i in range(M))
        assert abs(got[w] - float(ref)) <= 1e-10 * abs(float(ref)), (w, got[w], float(ref))


# ---------------------------------------------------------------- 64x64 tile variant (small walker batches / multi-GPU shards)
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_mfma_tile_gemm_64(deng, mode):
    eng = deng
    rng = np.random.default_rng(40 + mode)
    M, N, K = 192, 130, 48
    A = rng.standard_normal((M, K)); B = rng.standard_normal((K, N))
    args = [(A, B), (A, np.ascontiguousarray(B.T)), (np.ascontiguousarray(A.T), B)][mode]
    assert maxrel(eng.test_gemm(*args, mode, tile=64), A @ B) < 1e-13


@pytest.mark.parametrize("library", ["product", "debug"])
def test_predict_tile_sizes_are_bit_identical(request, library):
    """Both tile sizes reduce V^2 in the same tree, so the automatic switch (and hence the number of
    ranks a walker ensemble is sharded over) never changes a bit of the variance."""
    eng = request.getfixturevalue("eng" if library == "product" else "deng")
    from gpbayestools_hic_amd import synth
    rng = np.random.default_rng(8)
    for N in (448, 512):                                  # Np = 448 (odd number of 64-blocks) and 512
        d, P = 6, 3
        X = synth.lhs(N, d, seed=N)
        eng.set_data(X, rng.standard_normal((P, N)), "RBF", 0.1)
        eng.set_theta(synth.fixed_theta(d, P)); eng.factor()
        Xs = rng.random((300, d))
        eng.force_tile(128); m1, v1 = eng.predict(Xs)
        # the shapes the rule selects: 128x128 on ticket queues; 64x128, 64x64, 64x32 as static launches in three orders, two XCD
        # maps (the measured-and-rejected variants — 8-wave tiles, ticket queues for the 64-row tiles, folded row-block pairs,
        # LDS-DMA staging of the fp64 tiles, two more XCD maps — were deleted in round 6: profiles/HISTORY.md)
        for xcd in (0, 1):                                # tile -> XCD queue maps only reorder the work
            eng.tune("xcd", xcd)
            for tile in (64, 128, 32, 65):                # 32 = 64 rows x 32 walkers, 65 = 64 x 128
                eng.force_tile(tile)
                for order in (1, 2, 3):                   # static orders of the 64-row launches
                    eng.tune("resident", order)
                    m2, v2 = eng.predict(Xs)
                    assert np.array_equal(m1, m2) and np.array_equal(v1, v2), (tile, xcd, order)
        eng.tune("xcd", -1); eng.tune("resident", 2)
        from gpbayestools_hic_amd._native import GPBError
        for key, val in (("xcd", 2), ("resident", 0)):
            with pytest.raises(GPBError):
                eng.tune(key, val)                        # a deleted variant's value is refused, not silently ignored
        eng.force_tile(0)


def test_predict_tile_trace_covers_every_tile_once(deng):
    """debug hook: one record per (GP, row block, walker tile) of the static 64-row launch and of the 128 x 128 ticket-queue launch"""
    eng = deng
    from gpbayestools_hic_amd import synth
    rng = np.random.default_rng(21)
    N, d, P, W = 512, 5, 3, 256
    eng.set_data(synth.lhs(N, d, seed=3), rng.standard_normal((P, N)), "RBF", 0.1)
    eng.set_theta(synth.fixed_theta(d, P)); eng.factor()
    Xs = rng.random((W, d))
    for T in (64, 128):
        eng.force_tile(T)
        eng.tile_trace(4096)
        eng.predict(Xs)
        rec = eng.tile_trace_read()
        eng.tile_trace(0)
        tiles = {(int(r[2]), int(r[3]), int(r[4])) for r in rec}
        assert len(rec) == P * (N // T) * (W // T) == len(tiles)
        assert tiles == {(p, ib, wt) for p in range(P) for ib in range(N // T) for wt in range(W // T)}
        assert np.all((rec[:, 6].astype(np.int64) - rec[:, 5].astype(np.int64)) % (1 << 32) < 10_000_000)   # < 0.1 s
    eng.force_tile(0); eng.tune("resident", 2)


def test_product_and_debug_libraries_give_the_same_bits(eng, deng):
    """libgpbayes_debug.so is libgpbayes.so's sources plus hooks and variants: on their defaults the two give identical numbers
    (so what the hook-driven tests establish on the debug library holds for the product one)"""
    from gpbayestools_hic_amd import synth
    rng = np.random.default_rng(77)
    N, d, P, W = 200, 5, 3, 150
    X = synth.lhs(N, d, seed=9); Z = rng.standard_normal((P, N)); th = synth.fixed_theta(d, P) + 0.2 * rng.standard_normal((P, d + 2))
    Xs = rng.random((W, d))
    out = []
    for e in (eng, deng):
        e.set_data(X, Z, "Matern25", 0.1)
        val, grad = e.lml(th)
        e.set_theta(th); e.factor()
        out.append((val, grad, e.get("L"), e.get("alpha")) + tuple(e.predict(Xs)))
    import os
    if os.environ.get("GPB_DEBUG_LIB") != "1":             # (under GPB_DEBUG_LIB=1 the default engine is the debug library's too)
        assert eng.lib is not deng.lib and deng.has_variants and not eng.has_variants
    for a, b in zip(*out):
        assert np.array_equal(a, b)


# ---------------------------------------------------------------- the cache of freed device buffers
def test_pool_trim_hands_cached_buffers_back_and_contexts_still_work():
    """contexts come and go with every training and their large buffers are cached (csrc/gpb_pool.hip, GPB_POOL_MB); gpb_pool_trim
    gives the cache back to the driver, the next context allocates afresh and factors to the same bits"""
    import torch
    from gpbayestools_hic_amd import GPEngine, synth
    X = synth.lhs(700, 6)
    Z = np.random.default_rng(3).standard_normal((3, 700))
    th = synth.fixed_theta(6, 3)

    def alpha():
        e = GPEngine(0)
        e.set_data(X, Z, "RBF", 0.1); e.set_theta(th); e.factor()
        a = e.get("alpha")
        e.close()
        return a
    a0 = alpha()                                   # its K / L^-1 / T (3 x 704^2 doubles each, 11.9 MB) are cached now
    free_cached = torch.cuda.mem_get_info()[0]
    assert GPEngine(0).lib.gpb_pool_trim() == 0
    free_trimmed = torch.cuda.mem_get_info()[0]
    assert free_trimmed >= free_cached + 3 * 11 * (1 << 20)          # the three matrices went back to the driver
    assert np.array_equal(alpha(), a0) and np.array_equal(alpha(), a0)       # fresh allocation, then a cached one


def test_lml_gradient_bits_do_not_depend_on_the_kinv_tile_or_the_gp_count():
    """K^-1 = L^-T L^-1 of the LML gradient (k_kinv, sk:_gpr.py:627-629) on 64- or 128-wide tiles (option key 50; the rule picks by
    fill): every element's sum runs over k in the same order, so value and gradient are the same bits — and a GP's numbers do not
    depend on how many GPs share the launch (the 1-D grid deals tile t of all GPs before tile t + 1 of any)"""
    from gpbayestools_hic_amd import GPEngine, synth
    N, d, P = 300, 7, 5
    X = synth.lhs(N, d, seed=2)
    Z = np.random.default_rng(4).standard_normal((P, N))
    th = synth.fixed_theta(d, P) + 0.1 * np.random.default_rng(5).standard_normal((P, d + 2))
    outs = []
    for tile in (0, 64, 128):
        e = GPEngine(0)
        e.set_data(X, Z, "Matern25", 0.1)
        e.tune("kinv_tile", tile)
        outs.append(e.lml(th))
        e.close()
    for v, g in outs[1:]:
        assert np.array_equal(v, outs[0][0]) and np.array_equal(g, outs[0][1])
    e = GPEngine(0)
    e.set_data(X, Z[2:3], "Matern25", 0.1)                       # GP 2 alone
    v1, g1 = e.lml(th[2:3])
    e.close()
    assert v1[0] == outs[0][0][2] and np.array_equal(g1[0], outs[0][1][2])
