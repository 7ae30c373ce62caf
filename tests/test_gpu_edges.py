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


@pytest.mark.parametrize("kind,ell,where", [
    ("Matern15", 1e-3, "one"), ("Matern15", 1e-3, "all"), ("Matern15", 1e-2, "one"), ("Matern15", 1e-2, "all"),
    ("Matern25", 1e-3, "one"), ("Matern25", 1e-3, "all"), ("Matern25", 1e-2, "one"), ("Matern25", 1e-2, "all"),
    ("RBF", 0.1, "one"), ("RBF", 0.1, "all"),
])
def test_length_scales_at_the_lower_bound_of_the_references_search_box(kind, ell, where):
    """The reference lets L-BFGS-B take a length scale down to 1e-3 x the design's extent for "Matern" and 0.1 x for RBF
    (src/emulator.py:286-297).  There the Gram form of the squared distance, |a|^2 + |b|^2 - 2 a.b with |a|^2 ~ 2.5e5, loses
    ~1e-9 absolute in r^2 (3e-10 in K*, 2.5e-9 in the predictive variance), so such GPs are built in sklearn's difference
    form — chosen per GP from theta alone (gpbayes.h GPB_GET_FORM).  Queries sit within a length scale of design points
    along the short dimension(s), where K* is O(1) and r small: element-wise K*, mean, variance, K, LML and gradient."""
    from gpbayestools_hic_amd import GPEngine, synth
    from oracle import gp_oracle as O
    N, d, W = 1024, 20, 256
    kid = O.KIND_NAMES[kind]
    rng = np.random.default_rng(7)
    X = synth.lhs(N, d, seed=5)
    ptp = X.max(0) - X.min(0)
    ls = np.ones(d)
    short = np.arange(d) if where == "all" else np.array([4])
    ls[short] = ell * ptp[short]
    th = np.concatenate([[0.0], np.log(ls), [np.log(0.05)]])
    z = np.sin(X @ rng.standard_normal(d)) + 0.3 * rng.standard_normal(N)
    idx = rng.integers(0, N, W)
    Xs = X[idx] + rng.uniform(-0.02, 0.02, (W, d))
    Xs[:, short] = X[idx][:, short] + rng.uniform(-1.0, 1.0, (W, short.size)) * ls[short] * (0.25 if where == "all" else 1.0)
    Xs[:8] = X[idx[:8]]                                             # and a few queries ON design points (r = 0)
    S = float(np.sum((ptp / ls) ** 2))
    eng = GPEngine(0)
    eng.set_data(X, z[None, :], kind, 0.1); eng.set_theta(th[None, :])
    assert eng.get("form")[0] == (1.0 if S > 1024.0 else 0.0)
    assert S > 1024.0 or (kind == "RBF" and where == "one")         # every Matern case here is beyond the Gram form's range
    Ko = O.kernel_train(X, th, kid, 0.1)
    L, a = O.gp_factor(X, z, th, kid, 0.1)
    eng.fit_piece("kmat")                                           # K(X,X) alone (gpb_gp_factor overwrites it with L)
    Kg = eng.get("K")[0]
    assert np.max(np.abs(np.tril(Kg) - np.tril(Ko))) <= 1e-13
    eng.factor()
    m, v = eng.predict(Xs)
    Ks = eng.get("Kstar", W)[0]
    Kso = O.kernel_cross(Xs, X, th, kid)
    assert np.max(np.abs(Ks - Kso)) <= 1e-13                        # element-wise, c = 1
    assert Kso.max(1).min() > 0.05                                  # the queries do see their design points
    mo, vo = O.gp_predict(Xs, X, th, L, a, kid)
    assert np.max(np.abs(m[:, 0] - mo)) <= 1e-11 * np.max(np.abs(mo))
    assert np.max(np.abs(v[:, 0] - vo) / np.abs(vo)) < 1e-10
    val, grad = eng.lml(th[None, :])
    vo_, go_ = O.lml(th, X, z, kid, 0.1, eval_gradient=True)
    assert abs(val[0] - vo_) <= 1e-10 * abs(vo_)
    assert np.max(np.abs(grad[0] - go_)) <= 1e-9 * max(np.max(np.abs(go_)), 1.0)
    if kind == "Matern15" and ell == 1e-3 and where == "one":
        # what the rule protects against: the same GP forced into the Gram form misses the K* bar by orders of magnitude
        eng.tune("kcross_dot", 2)
        assert eng.get("form")[0] == 0.0
        eng.factor(); eng.predict(Xs)
        assert np.max(np.abs(eng.get("Kstar", W)[0] - Kso)) > 1e-11
        eng.tune("kcross_dot", 1)
        assert eng.get("form")[0] == 1.0
    eng.close()


def test_gps_of_one_emulator_in_different_distance_forms():
    """three GPs, the middle one beyond the Gram form's range: each is built in its own form (two cross-kernel and two K
    launches, each leaving the other's GPs alone), results as if each were alone, and no bit depends on the batch cut"""
    from gpbayestools_hic_amd import GPEngine, synth
    from oracle import gp_oracle as O
    N, d, P, W = 300, 6, 3, 333
    rng = np.random.default_rng(11)
    X = synth.lhs(N, d, seed=3)
    Z = rng.standard_normal((P, N))
    th = synth.fixed_theta(d, P, ell=0.8)
    th[1, 1 + 2] = np.log(2e-3)
    idx = rng.integers(0, N, W)
    Xs = np.clip(X[idx] + rng.uniform(-1e-3, 1e-3, (W, d)), 0.0, 1.0)
    for kind in ("RBF", "Matern15", "Matern25"):
        eng = GPEngine(0)
        eng.set_data(X, Z, kind, 0.1); eng.set_theta(th); eng.factor()
        assert eng.get("form").tolist() == [0.0, 1.0, 0.0]
        m, v = eng.predict(Xs)
        Ks = eng.get("Kstar", W)
        for p in range(P):
            L, a = O.gp_factor(X, Z[p], th[p], O.KIND_NAMES[kind], 0.1)
            mo, vo = O.gp_predict(Xs, X, th[p], L, a, O.KIND_NAMES[kind])
            assert np.max(np.abs(Ks[p] - O.kernel_cross(Xs, X, th[p], O.KIND_NAMES[kind]))) <= 1e-13
            assert np.max(np.abs(m[:, p] - mo)) <= 1e-11 * np.max(np.abs(mo))
            assert np.max(np.abs(v[:, p] - vo) / np.abs(vo)) < 1e-10
        for sl in (slice(0, 1), slice(5, 133), slice(200, 333)):
            ms, vs = eng.predict(Xs[sl])
            assert np.array_equal(ms, m[sl]) and np.array_equal(vs, v[sl])
        single = GPEngine(0)                                         # GP 1 alone: the same bits as in the mixed emulator
        single.set_data(X, Z[1:2], kind, 0.1); single.set_theta(th[1:2]); single.factor()
        m1, v1 = single.predict(Xs)
        assert np.array_equal(m1[:, 0], m[:, 1]) and np.array_equal(v1[:, 0], v[:, 1])
        single.close(); eng.close()


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


def test_torch_arguments_are_validated_before_their_pointers_reach_the_kernels():
    """a CPU tensor, a wrong dtype / shape / stride or a host extra_std next to device inputs must raise: the C ABI
    would read their addresses as device memory (checked on the host, never provoked on the card)"""
    import torch
    from gpbayestools_hic_amd import GPEngine
    from gpbayestools_hic_amd.engine import MODE_PCA
    rng = np.random.default_rng(0)
    N, d, P, M, W = 64, 3, 2, 4, 8
    eng = GPEngine(0)
    eng.set_data(rng.random((N, d)), rng.standard_normal((P, N)))
    eng.set_theta(np.tile(np.log([1.0, 1.0, 1.0, 1.0, 0.05]), (P, 1)))
    eng.factor()
    eng.set_transform(MODE_PCA, np.zeros(M), A=rng.standard_normal((P, M)), cov_trunc=0.1 * np.eye(M))
    eng.set_likelihood(np.zeros(M), 0.01 * np.eye(M))
    Xd = torch.as_tensor(rng.random((W, d)), device="cuda")
    lo, hi = torch.zeros(d, dtype=torch.float64, device="cuda"), torch.ones(d, dtype=torch.float64, device="cuda")
    out = torch.empty(W, dtype=torch.float64, device="cuda")
    ref = eng.logpost(Xd, out, False, lo, hi, -np.inf, 0.0).clone()
    bad_inputs = [Xd.cpu(), Xd.float(), Xd[:, :2], torch.empty((W, d, 1), dtype=torch.float64, device="cuda")]
    for bad in bad_inputs:
        for call in (lambda x: eng.predict(x), lambda x: eng.emu_predict(x), lambda x: eng.loglike(x),
                     lambda x: eng.logpost(x, out, False, lo, hi, -np.inf, 0.0)):
            with pytest.raises(ValueError):
                call(bad)
    with pytest.raises(ValueError):
        eng.logpost(Xd, out.cpu(), False, lo, hi, -np.inf, 0.0)
    with pytest.raises(ValueError):
        eng.logpost(Xd, out[:-1], False, lo, hi, -np.inf, 0.0)
    with pytest.raises(ValueError):
        eng.logpost(Xd, out, False, lo.cpu(), hi, -np.inf, 0.0)
    with pytest.raises(ValueError):
        eng.logpost(Xd, torch.empty(2 * W, dtype=torch.float64, device="cuda")[::2], False, lo, hi, -np.inf, 0.0)
    with pytest.raises(ValueError):
        eng.loglike(Xd, out=out.cpu())
    with pytest.raises(ValueError):
        eng.box_finish(Xd, lo, hi[:-1], -np.inf, 0.0, out)
    with pytest.raises(ValueError):
        eng.mvn_loglike(torch.zeros((W, M), dtype=torch.float64, device="cuda"), torch.zeros((W, M, M + 1), dtype=torch.float64, device="cuda"))
    with pytest.raises(ValueError):
        eng.mvn_loglike(torch.zeros((W, M), dtype=torch.float64, device="cuda"), torch.eye(M, dtype=torch.float64).repeat(W, 1, 1))
    with pytest.raises(ValueError):
        eng.emu_predict(Xd, extra_std=torch.zeros(W, dtype=torch.float64))            # host tensor next to device inputs
    # a transposed view is made contiguous, numbers / numpy extra_std are uploaded: same results as the plain calls
    Xt = Xd.t().contiguous().t()
    assert not Xt.is_contiguous() and torch.equal(eng.loglike(Xt), eng.loglike(Xd))
    m0, c0 = eng.emu_predict(Xd, extra_std=None)
    m1, c1 = eng.emu_predict(Xd, extra_std=0.0)
    m2, c2 = eng.emu_predict(Xd, extra_std=np.zeros(W))
    assert torch.equal(c0, c1) and torch.equal(c0, c2) and torch.equal(m0, m2)
    _, c3 = eng.emu_predict(Xd, extra_std=0.2)
    _, c4 = eng.emu_predict(Xd, extra_std=torch.full((W,), 0.2, dtype=torch.float64, device="cuda"))
    assert torch.equal(c3, c4) and not torch.equal(c3, c0)
    assert torch.equal(eng.logpost(Xd, out, False, lo, hi, -np.inf, 0.0), ref)
    eng.close()


def test_engine_follows_torchs_current_stream():
    """stream="torch": a call made inside `with torch.cuda.stream(s)` runs on s (where torch allocates and frees the
    caller's tensors), and back on the default stream afterwards — same numbers either way"""
    import torch
    from gpbayestools_hic_amd import GPEngine
    rng = np.random.default_rng(1)
    eng = GPEngine(0)
    eng.set_data(rng.random((64, 3)), rng.standard_normal((2, 64)))
    eng.set_theta(np.tile(np.log([1.0, 1.0, 1.0, 1.0, 0.05]), (2, 1)))
    eng.factor()
    Xd = torch.as_tensor(rng.random((100, 3)), device="cuda")
    m0, v0 = eng.predict(Xd)
    s0 = eng.lib.gpb_stream(eng.h)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        m1, v1 = eng.predict(Xd)
        assert eng.lib.gpb_stream(eng.h) == side.cuda_stream != s0
    side.synchronize()
    m2, v2 = eng.predict(Xd)
    assert eng.lib.gpb_stream(eng.h) == s0
    torch.cuda.synchronize()
    assert torch.equal(m0, m1) and torch.equal(v0, v1) and torch.equal(m0, m2)
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


@pytest.mark.parametrize("cfg,W", [(1, 64), (1, 257), (3, 512), (3, 1300)])
def test_compaction_to_rows_inside_the_box_changes_no_number(tmp_path, cfg, W):
    """gpb_logpost evaluates only the rows inside the prior box, as the reference does (src/mcmc.py:194-203, 275-283):
    the same numbers as evaluating every row (bit for bit, whatever the mix: none, some, most or all rows outside),
    through Chain.log_posterior, log_likelihood(finite=True) and the resident sampler."""
    from gpbayestools_hic_amd import StretchSampler, synth
    from gpbayestools_hic_amd.workload import build_chain
    chain, emu, info = build_chain(cfg, workdir=str(tmp_path))
    eng = emu._engine_ready()
    d = info["d"]
    rng = np.random.default_rng(W)
    base = synth.walkers(W, d, seed=W)
    for frac_out in (0.0, 0.3, 0.9, 1.0):
        X = base.copy()
        out = rng.random(W) < frac_out
        if frac_out == 1.0:
            out[:] = True
        X[out, rng.integers(0, d, out.sum())] = rng.choice([-0.25, 1.5, 0.0, 1.0], out.sum())   # outside or ON the boundary
        res = {}
        for compact in (1, 0):
            eng.tune("compact", compact)
            res[compact] = (chain.log_posterior(X), chain.log_likelihood(X, finite=True))
        eng.tune("compact", 1)
        assert np.array_equal(res[1][0], res[0][0]) and np.array_equal(res[1][1], res[0][1]), frac_out
        assert np.array_equal(np.isneginf(res[1][0]), out) and np.all(res[1][1][out] == -1e300)
    if cfg == 1:
        nw = 64
        X0 = synth.walkers(nw, d, seed=3)
        chains = {}
        for compact in (1, 0):
            eng.tune("compact", compact)
            s = StretchSampler(chain, nw, seed=11)
            s.run(X0, 6)
            chains[compact] = (s.chain, s.lnprobability)
        eng.tune("compact", 1)
        assert np.array_equal(chains[1][0], chains[0][0]) and np.array_equal(chains[1][1], chains[0][1])


def test_rows_with_nan_or_infinite_parameters_are_outside_the_box(tmp_path):
    """src/mcmc.py:275-283: `np.all((X > min) & (X < max), axis=1)` — a comparison with NaN is false, +-inf is outside:
    such rows get -inf (log_posterior) / -1e300 (finite=True) and never reach the emulators; the other rows of the batch
    keep their numbers bit for bit, on the host-buffer path and on the compacted device path."""
    from gpbayestools_hic_amd import synth
    from gpbayestools_hic_amd.workload import build_chain
    chain, emu, info = build_chain(1, workdir=str(tmp_path))
    d = info["d"]
    X = synth.walkers(96, d, seed=5)
    clean = chain.log_posterior(X)
    assert np.all(np.isfinite(clean))
    bad = X.copy()
    rows = {3: np.nan, 17: np.inf, 40: -np.inf, 95: np.nan}
    for r, v in rows.items():
        bad[r, r % d] = v
    bad[60, :] = np.nan                                           # every parameter NaN
    lp = chain.log_posterior(bad)
    ll = chain.log_likelihood(bad, finite=True)
    out = np.zeros(96, bool); out[list(rows) + [60]] = True
    assert np.array_equal(np.isneginf(lp), out) and np.all(ll[out] == -1e300)
    assert np.array_equal(lp[~out], clean[~out]) and np.all(np.isfinite(ll[~out]))
    eng = emu._engine_ready()
    for compact in (0, 1):                                        # evaluating every row / only the rows inside the box
        eng.tune("compact", compact)
        assert np.array_equal(chain.log_posterior(bad), lp)
    eng.tune("compact", 1)


def test_contexts_come_and_go_through_the_buffer_cache():
    """csrc/gpb_pool.hip keeps the device buffers a context releases for the next context of about that size (hipFree + hipMalloc of a
    large buffer is a driver round trip on this runtime): contexts of many shapes created, used, re-set and destroyed in any order —
    also while another context is alive on buffers of the same sizes — give the numbers a fresh context gives"""
    from gpbayestools_hic_amd import GPEngine, synth
    rng = np.random.default_rng(4)
    shapes = [(64, 3, 2), (200, 5, 3), (130, 4, 1), (200, 5, 3), (64, 3, 2), (333, 6, 2), (200, 5, 3)]
    ref = {}
    keep = []
    for rep in range(3):
        for n, (N, d, P) in enumerate(shapes):
            X = synth.lhs(N, d, seed=N + d); Z = np.random.default_rng(N).standard_normal((P, N))
            th = synth.fixed_theta(d, P)
            Xs = np.random.default_rng(d).random((37 + 64 * (n % 3), d))
            eng = GPEngine(0)
            eng.set_data(X, Z, "Matern15" if n % 2 else "RBF", 0.1); eng.set_theta(th); eng.factor()
            got = eng.predict(Xs) + (eng.lml(th)[0],)
            key = (n,)
            if key in ref:
                for a, b in zip(got, ref[key]):
                    assert np.array_equal(a, b), (rep, n)
            else:
                ref[key] = got
            if rng.random() < 0.4:
                keep.append(eng)                           # stays alive, on buffers that may have come from the cache
            else:
                eng.close()
            if keep and rng.random() < 0.5:
                old = keep.pop(int(rng.integers(len(keep))))
                old.set_data(synth.lhs(96, 2), np.zeros((1, 96)) + 1.0)       # re-set to another shape, then gone
                old.close()
    for e in keep:
        e.close()


@pytest.mark.parametrize("N", [100, 70, 127, 129, 1000])
def test_designs_that_are_not_a_multiple_of_64_points(N):
    """The stored matrices are padded to Np = a multiple of 64 rows with identity blocks — in FRONT of the design in whole 16-row
    units (the predict tiles' K-step: those leading zero rows of K*^T are skipped and the padded rows of V fall into the lightest
    row block), the remainder of fewer than 16 rows behind (csrc: pad_front, gp_set_impl).  Nothing of that may show at the
    boundary: K, L, L^-1, alpha and K* come back in the design's own numbering, LAPACK's info counts the design's pivots, and
    every number meets the oracle's — for front paddings of 16 / 48 / 0 / 48 / 16 rows and 12 / 10 / 1 / 15 / 8 behind."""
    from gpbayestools_hic_amd import GPEngine, synth
    from oracle import gp_oracle as O
    from scipy.linalg import lapack
    d, P, W = 5, 3, 37
    X = synth.lhs(N, d, seed=N)
    rng = np.random.default_rng(N)
    Z = rng.standard_normal((P, N))
    th = np.array([np.concatenate([[rng.uniform(-0.3, 0.3)], np.log(rng.uniform(0.5, 2.0, d)), [np.log(rng.uniform(0.02, 0.1))]])
                   for _ in range(P)])
    eng = GPEngine(0)
    eng.set_data(X, Z, "Matern25", 0.1)
    eng.set_theta(th)
    eng.fit_piece("kmat")                              # K(X,X) alone (the factorisation overwrites it with L)
    K = eng.get("K")
    eng.factor()
    Xs = synth.walkers(W, d, seed=N + 1)
    Xs[5] = X[N - 1]; Xs[6] = X[0]                     # queries ON the last and the first design point
    mean, var = eng.predict(Xs)
    L, Li, al, Ks = eng.get("L"), eng.get("Linv"), eng.get("alpha"), eng.get("Kstar", W)
    assert K.shape == (P, N, N) and Ks.shape == (P, W, N) and al.shape == (P, N)
    kind = O.KIND_NAMES["Matern25"]
    for p in range(P):
        Ko = O.kernel_train(X, th[p], kind, 0.1)
        Lo, ao = O.gp_factor(X, Z[p], th[p], kind, 0.1)
        assert np.max(np.abs(np.tril(K[p]) - np.tril(Ko))) < 1e-13 * np.max(np.abs(Ko))      # (the lower triangle is what is built)
        assert np.max(np.abs(L[p] - Lo)) < 1e-11 * np.max(np.abs(Lo)) and np.max(np.abs(al[p] - ao)) < 1e-10 * np.max(np.abs(ao))
        assert np.max(np.abs(Li[p] @ Lo - np.eye(N))) < 1e-10
        assert np.max(np.abs(Ks[p] - O.kernel_cross(Xs, X, th[p], kind))) < 1e-13
        mo, vo = O.gp_predict(Xs, X, th[p], Lo, ao, kind)
        assert np.max(np.abs(mean[:, p] - mo)) < 1e-11 * max(np.max(np.abs(mo)), 1.0)
        assert np.max(np.abs(var[:, p] - vo)) < 1e-10 * np.max(np.abs(vo))
    lml, grad = eng.lml(th)
    for p in range(P):
        vo, go = O.lml(th[p], X, Z[p], kind, 0.1, eval_gradient=True)
        assert abs(lml[p] - vo) < 1e-10 * abs(vo) and np.max(np.abs(grad[p] - go)) < 1e-9 * max(np.max(np.abs(go)), 1.0)
    # an indefinite matrix: LAPACK's info (first non-positive pivot, 1-based) in the DESIGN's numbering
    eng.set_data(X, Z, "Matern25", alpha=-1.3)
    eng.set_theta(th)
    info = eng.factor(raise_on_fail=False)
    for p in range(P):
        _, ref_info = lapack.dpotrf(O.kernel_train(X, th[p], kind, -1.3), lower=1)
        assert ref_info > 0 and info[p] == ref_info, (p, info[p], ref_info)
    eng.close()
