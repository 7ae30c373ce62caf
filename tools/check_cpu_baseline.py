#!/usr/bin/env python3
"""
Is the oracle's *faithful* mode (bench.py's `cpu_baseline`, kind "port") the reference's work?

Runs in the BUILD container only (needs /root/reference): imports the reference's Emulator / Chain, builds the
BASELINE cfg-1 and cfg-3 shapes from the same synthetic files the bench uses, and times

    reference  Chain.log_posterior(X)                                   src/mcmc.py:261-299
    oracle     log_prob(..., faithful W x W covariance per GP, per-row LAPACK MVN)   oracle/gp_oracle.py

on identical inputs (median of 5 after one warm-up call).  Asserts that the two return the same numbers
(<= 1e-12 relative) and that the wall-time ratio oracle / reference lies within +-10 % at the cfg-3 shape
(BASELINE.md §3); cfg 1 runs for a few milliseconds, its ratio is printed, not asserted.

The hyper-parameter search is switched off on the reference side for this measurement (sklearn's
`GaussianProcessRegressor(optimizer=None)` bound into the reference's namespace at run time — its files are not
touched): both sides then factorise at the kernel's initial theta, and the time of a log-posterior call does not
depend on theta.

    python tools/check_cpu_baseline.py | tee profiles/r02_cpu_baseline_faithfulness.txt
"""
import functools
import os
import statistics
import sys
import tempfile
import time
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF = "/root/reference"
sys.dont_write_bytecode = True
_work = tempfile.mkdtemp(prefix="gpb_ref_work_")
os.environ["WORKDIR"] = _work
os.environ.setdefault("LOGLEVEL", "warning")
sys.path.insert(0, REF)

from gpbayestools_hic_amd import synth  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402


def _import_reference():
    emcee = types.ModuleType("emcee")
    emcee.EnsembleSampler = type("EnsembleSampler", (), {})
    sys.modules.setdefault("emcee", emcee)
    sys.modules.setdefault("pocomc", types.ModuleType("pocomc"))
    import src.emulator as ref_emu
    from src import mcmc as ref_mcmc
    from sklearn.gaussian_process import GaussianProcessRegressor
    ref_emu.GPR = functools.partial(GaussianProcessRegressor, optimizer=None)     # no L-BFGS-B: fit at theta0
    return ref_emu.Emulator, ref_mcmc


def median_time(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts), out


def run(cfg, Emulator, mcmc, assert_ratio):
    c = synth.CONFIGS[cfg]
    N, d, M, P, W = c["N"], c["d"], c["M"], c["P"], c["W"]
    lo, hi = np.zeros(d), np.ones(d)
    X = synth.lhs(N, d)
    Y = synth.observables(X, M)
    tp, pf, ep = (os.path.join(_work, f"cfg{cfg}_{n}") for n in ("train.pkl", "par.txt", "exp.pkl"))
    synth.write_training_pickle(tp, X, Y, 0.01)
    synth.write_parameter_file(pf, lo, hi)
    emu = Emulator(training_set_path=tp, parameter_file=pf, npc=P)
    emu.trainEmulatorAutoMask()
    thetas = np.array([gp.kernel_.theta for gp in emu.gps])
    xstar = synth.truth_point(d)
    yexp = emu.predict(xstar[None, :], return_cov=False)[0]
    synth.write_experiment_pickle(ep, yexp, 0.05 * np.abs(yexp))
    chain = mcmc.Chain(mcmc_path=os.path.join(_work, "mcmc", f"c{cfg}.pkl"), expdata_path=ep, model_parafile=pf)
    chain.emuList = [emu]
    oe = O.OracleEmulator(X, Y, lo, hi, P).fit(thetas)
    cexp = np.diag((0.05 * np.abs(yexp)) ** 2)
    Xw = synth.walkers(W, d, seed=synth.SEED + 7)
    t_ref, lp_ref = median_time(lambda: chain.log_posterior(Xw))
    t_orc, lp_orc = median_time(lambda: O.log_prob(Xw, lo, hi, lambda x, e: oe.predict(x, True, e, faithful=True),
                                                   yexp, cexp, batched=False))
    rel = float(np.max(np.abs(lp_orc - lp_ref) / np.abs(lp_ref)))
    ratio = t_orc / t_ref
    print(f"cfg {cfg} shape (N={N}, d={d}, M={M}, P={P}, W={W}): reference {t_ref:.4f} s, oracle faithful {t_orc:.4f} s, "
          f"ratio {ratio:.3f}, max rel diff {rel:.2e}, bit-equal {bool(np.array_equal(lp_orc, lp_ref))}")
    assert rel < 1e-12, rel
    if assert_ratio:
        assert 0.9 <= ratio <= 1.1, ratio
    return ratio


def main():
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count()
    import scipy
    import sklearn
    print(f"nproc {cores}, OMP_NUM_THREADS {os.environ.get('OMP_NUM_THREADS', 'unset')}, numpy {np.__version__}, "
          f"scipy {scipy.__version__}, scikit-learn {sklearn.__version__}; median of 5 calls after one warm-up")
    Emulator, mcmc = _import_reference()
    run(1, Emulator, mcmc, assert_ratio=False)
    run(3, Emulator, mcmc, assert_ratio=True)
    print("ok: the oracle's faithful mode returns the reference's numbers in the reference's time (+-10 %)")


if __name__ == "__main__":
    main()
