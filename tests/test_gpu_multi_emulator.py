"""
GPU tier: a chain of NINE emulators with different designs, kernels, numbers of GPs and observables over one parameter
space — the shape of the reference's real analyses (RunBayesianAnalysis.ipynb:35-48: nine emulators, ~540 observables;
Chain._predict concatenates them under a block-diagonal covariance, src/mcmc.py:153-166) — against the CPU oracle's
540 x 540 multivariate normal, and the C-driven sampling loop over all nine against emcee's algorithm evaluated with
the oracle's log-posterior.
"""
import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu

SPECS = [(96, 60, 4, "RBF"), (128, 60, 6, "Matern25"), (64, 60, 3, "RBF"), (160, 60, 5, "Matern15"), (80, 60, 4, "RBF"),
         (112, 60, 6, "RBF"), (72, 60, 3, "Matern25"), (144, 60, 5, "RBF"), (100, 60, 4, "RBF")]
D = 8


def _oracle_chain(info):
    from gpbayestools_hic_amd import synth
    from oracle import gp_oracle as O
    oes = []
    for i, ((N, M, P, kernel), (X, Y)) in enumerate(zip(info["specs"], info["data"])):
        oe = O.OracleEmulator(X, Y, info["lo"], info["hi"], P, O.KIND_NAMES[kernel])
        oes.append(oe.fit(synth.fixed_theta(info["d"], P, ell=1.2 + 0.1 * i, noise=0.03 + 0.01 * i)))
    cexp = np.diag((0.05 * np.abs(info["yexp"])) ** 2)
    return lambda X, **kw: O.log_prob(X, info["lo"], info["hi"], lambda x, e: O.chain_predict(oes, x, e), info["yexp"],
                                      cexp, **kw)


def test_nine_emulator_chain_against_the_oracle(tmp_path):
    import ctypes
    from gpbayestools_hic_amd import StretchSampler, synth
    from gpbayestools_hic_amd.workload import build_multi_chain
    chain, emus, info = build_multi_chain(SPECS, D, workdir=str(tmp_path))
    assert chain.nobs == 540 and len(chain.emuList) == 9
    logpost = _oracle_chain(info)
    X = synth.walkers(48, D, seed=5)
    X[3, 1] = 1.25
    X[17, 6] = 0.0                                            # on the boundary: outside
    X[40:] = info["xstar"] + 0.01 * np.random.default_rng(1).standard_normal((8, D))   # near the truth: large values
    ref = logpost(X)
    got = chain.log_posterior(X)
    ins = np.isfinite(ref)
    assert np.array_equal(np.isneginf(got), ~ins) and ins.sum() == 46
    assert relerr(got[ins], ref[ins]) < 1e-10
    assert relerr(chain.log_likelihood(X, finite=True)[ins], logpost(X, posterior=False, finite=True)[ins]) < 1e-10
    # the whole chain went through ONE call of the C ABI; sequenced per emulator from Python it gives the same bits
    engs = [e._engine_ready() for e in emus]
    assert engs[0].lib.gpb_chain_supported((ctypes.c_void_p * 9)(*[e.h for e in engs]), 9) == 1
    chain.use_chain_call = False
    assert np.array_equal(chain.log_posterior(X), got)
    chain.use_chain_call = True
    # ... and inside the call the emulators whose designs pad to the same size (here 96 / 128, then 80 / 112 / 72 design
    # points: Np = 128) share ONE predict launch over all their GPs (k_predict_static_multi / k_predict_multi): same bits as
    # one launch per emulator, for batches that select each of the tile shapes
    for W in (48, 700, 3000):
        Xb = X if W == 48 else synth.walkers(W, D, seed=W)
        one = chain.log_posterior(Xb)
        engs[0].tune("chain_batch", 0)
        assert np.array_equal(chain.log_posterior(Xb), one), W
        engs[0].tune("chain_batch", 1)
        for tile in (128, 65, 64, 32):
            for g in engs:
                g.force_tile(tile)
            assert np.array_equal(chain.log_posterior(Xb), one), (W, tile)
        for g in engs:
            g.force_tile(0)
    # three steps of the C-driven loop over the nine emulators against emcee's stretch move with the oracle's
    # log-posterior fed the same Philox draws (oracle/stretch_oracle.py)
    from test_gpu_sampler_step import _oracle_chain as emcee_by_the_oracle
    nw, seed = 32, 77
    X0 = np.clip(info["xstar"] + 0.05 * np.random.default_rng(4).standard_normal((nw, D)), 0.02, 0.98)
    s = StretchSampler(chain, nw, seed=seed)
    assert s._resident_engine()[2] == 9
    s.run(X0, 3, status=100)
    Xf, lp, nacc, hist = emcee_by_the_oracle(X0, 3, seed, True, s._engine(), logpost)
    assert np.array_equal(s.naccept.cpu().numpy(), nacc) and nacc.sum() > 0
    assert np.array_equal(s.chain[:, -1], Xf)                     # same decisions => same positions, bit for bit
    fin = np.isfinite(lp)
    assert relerr(s.lnprobability[:, -1][fin], lp[fin]) < 1e-10


def test_mixed_chain_mapped_and_plain_emulators_and_argument_checks(tmp_path):
    """a parameterTrafoPCA emulator (GPs over PCA-reduced parameters, src/emulator.py:492-551) and a plain one over the
    same 20 model parameters in ONE chain: gpb_chain_logpost maps the gathered rows for the first and feeds them as
    they are to the second; same bits as the per-emulator calls from Python.  Then the calls' argument checks."""
    import ctypes
    import torch
    from conftest import golden
    from gpbayestools_hic_amd import Chain, Emulator, GPEngine, StretchSampler, synth
    from gpbayestools_hic_amd import _native as nat
    g = golden("g7_param_pca.npz")
    d = len(g["lo"])
    tp, pf, ep = str(tmp_path / "t.pkl"), str(tmp_path / "p.txt"), str(tmp_path / "e.pkl")
    synth.write_training_pickle(tp, g["X"], g["Y"], 0.01)
    synth.write_parameter_file(pf, g["lo"], g["hi"])
    mapped = Emulator(training_set_path=tp, parameter_file=pf, npc=int(g["npc"]), parameterTrafoPCA=True)
    mapped.trainEmulator([True] * mapped.nev, thetas=g["thetas"])
    X2 = synth.lhs(90, d, seed=9, lo=g["lo"], hi=g["hi"])
    Y2 = synth.observables((X2 - g["lo"]) / (g["hi"] - g["lo"]), 24, seed=10)
    tp2 = str(tmp_path / "t2.pkl")
    synth.write_training_pickle(tp2, X2, Y2, 0.01)
    plain = Emulator(training_set_path=tp2, parameter_file=pf, npc=5)
    plain.trainEmulator([True] * plain.nev, thetas=synth.fixed_theta(d, 5, ell=2.0))
    x0 = 0.5 * (g["lo"] + g["hi"])
    yexp = np.concatenate([mapped.predict(x0[None], return_cov=False)[0], plain.predict(x0[None], return_cov=False)[0]])
    synth.write_experiment_pickle(ep, yexp, 0.05 * np.abs(yexp))
    chain = Chain(mcmc_path=str(tmp_path / "mcmc" / "c.pkl"), expdata_path=ep, model_parafile=pf)
    chain.emuList = [mapped, plain]
    X = g["lo"] + (g["hi"] - g["lo"]) * np.random.default_rng(5).uniform(-0.02, 1.02, (300, d))
    one = chain.log_posterior(X)
    ins = np.all((X > g["lo"]) & (X < g["hi"]), axis=1)
    assert 0 < ins.sum() < 300 and np.array_equal(np.isfinite(one), ins)
    chain.use_chain_call = False
    assert np.array_equal(chain.log_posterior(X), one)
    chain.use_chain_call = True
    # and in the other order (the plain emulator's context then owns the compaction)
    chain.emuList = [plain, mapped]
    chain.expdata = np.concatenate([chain.expdata[:, -24:], chain.expdata[:, :-24]], axis=1)
    n = chain.expdata_cov.shape[0]
    perm = np.r_[n - 24:n, 0:n - 24]
    chain.expdata_cov = chain.expdata_cov[np.ix_(perm, perm)]
    swapped = chain.log_posterior(X)
    assert np.array_equal(np.isfinite(swapped), ins)
    assert np.allclose(swapped[ins], one[ins], rtol=1e-12, atol=0)      # the blocks are summed in the other order
    s = StretchSampler(chain, 24, seed=2)
    assert s._resident_engine()[2] == 2
    s.run(x0 + 0.01 * (g["hi"] - g["lo"]) * np.random.default_rng(6).standard_normal((24, d)), 3, status=10)
    assert np.isfinite(s.lnprobability).all()

    # ---- argument checks of the chain calls
    e1, e2 = plain._engine_ready(), mapped._engine_ready()
    lib = e1.lib
    arr = (ctypes.c_void_p * 2)(e1.h, e2.h)
    Xd = torch.as_tensor(np.ascontiguousarray(X), device="cuda")
    out = torch.empty(len(X), dtype=torch.float64, device="cuda")
    lo, hi = chain._box(Xd.device)
    call = lambda a, E, W=len(X): lib.gpb_chain_logpost(a, E, nat.ptr(Xd), W, nat.ptr(out), nat.ptr(lo), nat.ptr(hi),
                                                        float("-inf"), 0.0)
    assert call(arr, 2) == 0 and call(arr, 2, 0) == 0                  # an empty batch is a no-op
    assert call(arr, 0) != 0 and call(None, 2) != 0 and call(arr, 65) != 0
    assert call((ctypes.c_void_p * 2)(e1.h, None), 2) != 0             # a null context in the list
    assert lib.gpb_chain_logpost(arr, 2, None, 5, nat.ptr(out), nat.ptr(lo), nat.ptr(hi), 0.0, 0.0) != 0
    assert call(arr, 2, -1) != 0
    other = GPEngine(0)                                                # fitted, but over 3 parameters and no likelihood
    other.set_data(synth.lhs(64, 3), np.random.default_rng(0).standard_normal((2, 64)), "RBF", 0.1)
    other.set_theta(synth.fixed_theta(3, 2))
    other.factor()
    bad = (ctypes.c_void_p * 2)(e1.h, other.h)
    assert lib.gpb_chain_supported(bad, 2) == 0 and call(bad, 2) != 0
    assert b"gpb_like_set" in lib.gpb_last_error(e1.h)
    assert lib.gpb_chain_emcee_run(arr, 2, nat.ptr(Xd), nat.ptr(out), 7, 1, 1, 0, 2.0, 1, nat.ptr(lo), nat.ptr(hi),
                                   float("-inf"), 0.0, None, None, None) != 0       # odd number of walkers
    other.close()


@pytest.mark.parametrize("d,nws", [(40, (2, 6, 34, 130)), (33, (66,)), (3, (4, 258, 40000))])
def test_c_loop_shapes_many_parameters_and_tiny_ensembles(tmp_path, d, nws):
    """the C-driven loop's walker-group kernels keep one parameter per lane up to 32 and two beyond (d <= 64): chains
    over 40, 33 and 3 parameters, ensembles from a single pair of walkers up to 40 000 (beyond 16 384 rows per batch the
    compaction keeps its own marking kernel), against the host-driven loop; and one rank's
    balanced slice of a 2-way split on the widest"""
    from gpbayestools_hic_amd import StretchSampler
    from gpbayestools_hic_amd.workload import build_multi_chain
    chain, emus, info = build_multi_chain([(72, 10, 3, "RBF"), (64, 6, 2, "Matern25")], d, workdir=str(tmp_path))
    eng = emus[0]._engine_ready()
    rng = np.random.default_rng(d)
    for nw in nws:
        X0 = np.clip(info["xstar"] + 0.2 * rng.standard_normal((nw, d)), -0.05, 1.05)      # a few walkers start outside
        c = StretchSampler(chain, nw, seed=nw)
        assert c._resident_engine()[2] == 2
        c.run(X0, 5, status=2)
        h = StretchSampler(chain, nw, seed=nw)
        h._resident_engine = lambda: None
        h.run(X0, 5, status=2)
        assert np.array_equal(c.chain, h.chain) and np.array_equal(c.lnprobability, h.lnprobability), nw
        assert np.array_equal(c.naccept.cpu().numpy(), h.naccept.cpu().numpy())
    nw = nws[-1]
    if eng.has_variants and (nw % 4 == 0 or (nw // 2) % 2 == 0):      # one rank's share: a hook of the debug library (GPB_DEBUG_LIB=1)
        eng.tune("sim_ranks", 2); eng.tune("sim_rank", 1); eng.tune("balance_shards", 2)
        s = StretchSampler(chain, nw, seed=1)
        s.run(X0, 3, status=10)
        eng.tune("sim_ranks", 0); eng.tune("sim_rank", 0); eng.tune("balance_shards", 0)
        assert np.isfinite(s.chain).all()


def test_chain_with_more_than_64_parameters(tmp_path):
    """The GPs' own input dimension is bounded by 64 (gpb_gp_set), the CHAIN's is not: a parameterTrafoPCA emulator
    (src/emulator.py:79-241, 492-551) maps d_in chain parameters to d_out GP inputs, and with one principal component
    per group 70 parameters become 63.  The C-driven loop's proposal kernels gather the rows inside the box lane by
    lane (two parameters per lane in registers, the rest recomputed) and the marking kernel stages its rows in column
    tiles: same ensemble as the host-driven loop, same log-posterior as the per-emulator calls."""
    from conftest import golden
    from gpbayestools_hic_amd import Chain, Emulator, StretchSampler, synth
    g = golden("g7_param_pca.npz")
    d = 70
    lo, hi = np.concatenate([g["lo"], np.zeros(d - 20)]), np.concatenate([g["hi"], np.ones(d - 20)])
    X = synth.lhs(80, d, seed=5, lo=lo, hi=hi)
    mid = 0.5 * (lo + hi)
    for group in ([15, 16, 17, 18], [12, 13, 14], [2, 3, 4]):          # one varying parameter per group: one PC each
        X[:, group[1:]] = mid[group[1:]]
    Y = synth.observables((X - lo) / (hi - lo), 6, seed=11)
    tp, pf, ep = str(tmp_path / "t.pkl"), str(tmp_path / "p.txt"), str(tmp_path / "e.pkl")
    synth.write_training_pickle(tp, X, Y, 0.01)
    synth.write_parameter_file(pf, lo, hi)
    emu = Emulator(training_set_path=tp, parameter_file=pf, npc=3, parameterTrafoPCA=True)
    assert emu.PCA_new_design_points.shape == (80, 63)
    emu.trainEmulator([True] * emu.nev, thetas=synth.fixed_theta(63, 3, ell=3.0))
    yexp = emu.predict(mid[None], return_cov=False)[0]
    synth.write_experiment_pickle(ep, yexp, 0.05 * np.abs(yexp))
    chain = Chain(mcmc_path=str(tmp_path / "mcmc" / "c.pkl"), expdata_path=ep, model_parafile=pf)
    chain.emuList = [emu]
    rng = np.random.default_rng(3)
    Xq = lo + (hi - lo) * rng.uniform(-0.002, 1.002, (600, d))          # 70 parameters: ~3/4 of the rows stay inside
    one = chain.log_posterior(Xq)
    ins = np.all((Xq > lo) & (Xq < hi), axis=1)
    assert 100 < ins.sum() < 590 and np.array_equal(np.isfinite(one), ins)
    chain.use_chain_call = False
    assert np.array_equal(chain.log_posterior(Xq), one)
    chain.use_chain_call = True
    # against the host arithmetic of the same emulator (Emulator.predict maps the parameters with numpy)
    from oracle import gp_oracle as O
    mean, cov = emu.predict(Xq[ins][:8], return_cov=True, extra_std=0.0)
    ref = np.array([O.mvn_loglike(m - chain.expdata[0], c + chain.expdata_cov) for m, c in zip(mean, cov)]) + O.EXTRA_STD_CONST
    assert relerr(one[ins][:8], ref) < 1e-10
    for nw in (6, 300):
        X0 = np.clip(mid + 0.05 * (hi - lo) * rng.standard_normal((nw, d)), lo - 0.01 * (hi - lo), hi + 0.01 * (hi - lo))
        for premark in (2, 1, 0):
            emu._engine_ready().tune("premark", premark)
            c = StretchSampler(chain, nw, seed=nw)
            assert c._resident_engine() is not None
            c.run(X0, 4, status=2)
            if premark == 2:
                h = StretchSampler(chain, nw, seed=nw)
                h._resident_engine = lambda: None
                h.run(X0, 4, status=2)
            assert np.array_equal(c.chain, h.chain) and np.array_equal(c.lnprobability, h.lnprobability), (nw, premark)
        emu._engine_ready().tune("premark", 2)
        assert c.acceptance_fraction.mean() > 0.0


def test_chain_block_likelihoods_split_over_workgroups_give_the_walks_bits(tmp_path):
    """option key 49: the block log-likelihoods of a chain's emulators as one workgroup per (walker tile, emulator) + an ordered sum
    (default) against one workgroup per walker tile walking the emulators: the same additions in the same order, bit for bit — on
    batches with rows outside the box, tiny batches, and through the resident step loop"""
    from gpbayestools_hic_amd import StretchSampler, synth
    from gpbayestools_hic_amd.workload import build_multi_chain
    chain, emus, info = build_multi_chain(SPECS, D, workdir=str(tmp_path))
    e0 = emus[0]._engine_ready()
    X = synth.walkers(1500, D, seed=21)
    X[::11, 3] = -0.25
    X0 = synth.walkers(192, D, seed=4)
    outs = {}
    for split in (1, 0, 1):
        e0.tune("lr_split", split)
        s = StretchSampler(chain, 192, seed=9)
        outs.setdefault(split, []).append((chain.log_posterior(X), chain.log_posterior(X[:3]), chain.log_likelihood(X[:65], finite=True),
                                           s.run(X0, 5, status=10 ** 9), s.lnprobability))
    for a, b in zip(outs[1][0], outs[0][0]):
        assert np.array_equal(a, b)
    for a, b in zip(outs[1][0], outs[1][1]):
        assert np.array_equal(a, b)
    assert np.isneginf(outs[1][0][0][::11]).all() and np.isfinite(outs[1][0][0][1::11]).all()
    ref = _oracle_chain(info)(X[:40])
    ins = np.isfinite(ref)
    assert relerr(outs[1][0][0][:40][ins], ref[ins]) < 1e-10


def test_parameter_maps_of_several_emulators_in_one_launch(tmp_path):
    """Three parameterTrafoPCA emulators (their own designs, so their own maps; src/emulator.py:492-551) and a plain one in one
    chain: the maps of the three run as ONE launch over the gathered rows (k_param_map_multi) — the same bits as one launch per
    emulator (tune chain_batch 0) and as the per-emulator calls from Python, through the log-posterior and the C step loop."""
    from conftest import golden
    from gpbayestools_hic_amd import Chain, Emulator, StretchSampler, synth
    g = golden("g7_param_pca.npz")
    lo, hi = g["lo"], g["hi"]
    d = len(lo)
    pf, ep = str(tmp_path / "p.txt"), str(tmp_path / "e.pkl")
    synth.write_parameter_file(pf, lo, hi)
    emus = []
    for i, (n, m, npc, on) in enumerate(((0, 4, 3, True), (80, 6, 2, True), (70, 5, 3, True), (90, 12, 4, False))):
        X = g["X"] if n == 0 else synth.lhs(n, d, seed=20 + i, lo=lo, hi=hi)
        Y = g["Y"] if n == 0 else synth.observables((X - lo) / (hi - lo), m, seed=30 + i)
        tp = str(tmp_path / ("t%d.pkl" % i))
        synth.write_training_pickle(tp, X, Y, 0.01)
        e = Emulator(training_set_path=tp, parameter_file=pf, npc=npc, parameterTrafoPCA=on)
        th = synth.fixed_theta(d, npc, ell=2.0)
        if on:          # length scales in units of each reduced input's extent: these GPs take the Gram form, and the three
            ext = np.ptp(e.PCA_new_design_points, axis=0)      # emulators (22, 23 and 22 reduced inputs) share ONE cross launch
            th = synth.fixed_theta(len(ext), npc, ell=2.0)
            th[:, 1:-1] += np.log(ext)[None, :]
        e.trainEmulator([True] * e.nev, thetas=g["thetas"] if n == 0 else th)
        emus.append(e)
    assert len({e.PCA_new_design_points.shape[1] for e in emus[:3]}) > 1          # different input counts, one padded count
    x0 = 0.5 * (lo + hi)
    yexp = np.concatenate([e.predict(x0[None], return_cov=False)[0] for e in emus])
    synth.write_experiment_pickle(ep, yexp, 0.05 * np.abs(yexp) + 1e-3)
    chain = Chain(mcmc_path=str(tmp_path / "mcmc" / "c.pkl"), expdata_path=ep, model_parafile=pf)
    chain.emuList = emus
    X = lo + (hi - lo) * np.random.default_rng(5).uniform(-0.01, 1.01, (700, d))
    ins = np.all((X > lo) & (X < hi), axis=1)
    one = chain.log_posterior(X)
    assert 0 < ins.sum() < 700 and np.array_equal(np.isfinite(one), ins)
    eng0 = emus[0]._engine_ready()
    eng0.tune("chain_batch", 0)
    assert np.array_equal(chain.log_posterior(X), one)                  # one map launch per emulator
    start = x0 + 0.01 * (hi - lo) * np.random.default_rng(6).standard_normal((40, d))
    s0 = StretchSampler(chain, 40, seed=3)
    s0.run(start, 6, status=10 ** 9)
    eng0.tune("chain_batch", 1)
    s1 = StretchSampler(chain, 40, seed=3)
    assert s1._resident_engine()[2] == 4
    s1.run(start, 6, status=10 ** 9)
    assert np.array_equal(s0.chain, s1.chain) and np.array_equal(s0.lnprobability, s1.lnprobability)
    chain.use_chain_call = False                                        # the per-emulator calls from Python
    assert np.array_equal(chain.log_posterior(X), one)
