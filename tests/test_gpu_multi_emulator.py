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
