#!/usr/bin/env python3
"""
Is the oracle's *faithful* mode (bench.py's `cpu_baseline`, kind "port") the reference's work?

Runs in the BUILD container only (needs /root/reference): imports the reference's Emulator / Chain, builds the
BASELINE cfg-1 and cfg-3 shapes from the same synthetic files the bench uses, and times

    reference  Chain.log_posterior(X)                                   src/mcmc.py:261-299
    oracle     log_prob(..., faithful W x W covariance per GP, per-row LAPACK MVN)   oracle/gp_oracle.py

on identical inputs, the two sides called ALTERNATELY (reference, oracle, reference, ... : both see the same machine state — the
container's eight shared cores drift by +-15 % between two back-to-back blocks of calls, which is what a ratio of two separate
medians measured in rounds 2-4), REPS >= 7 pairs after one warm-up pair.  Asserts that the two return the same numbers, BIT FOR
BIT; reports the wall-time ratio oracle / reference as the median of the per-pair ratios with its spread (min, quartiles, max).
The ratio is evidence, not an assertion: a noisy box must not fail a parity tool.

The hyper-parameter search is switched off on the reference side for this measurement (sklearn's
`GaussianProcessRegressor(optimizer=None)` bound into the reference's namespace at run time — its files are not
touched): both sides then factorise at the kernel's initial theta, and the time of a log-posterior call does not
depend on theta.

    python tools/check_cpu_baseline.py | tee profiles/r05_cpu_baseline_faithfulness.txt
"""
import functools
import os
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


REPS = 9


def alternate(f_ref, f_orc, reps=REPS):
    """(times_ref, times_orc, last outputs): one warm-up pair, then `reps` pairs reference / oracle in turn"""
    f_ref(); f_orc()
    tr, to = [], []
    for _ in range(reps):
        t0 = time.perf_counter(); a = f_ref(); t1 = time.perf_counter(); b = f_orc(); t2 = time.perf_counter()
        tr.append(t1 - t0); to.append(t2 - t1)
    return np.array(tr), np.array(to), a, b


def run(cfg, Emulator, mcmc):
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
    tr, to, lp_ref, lp_orc = alternate(lambda: chain.log_posterior(Xw),
                                       lambda: O.log_prob(Xw, lo, hi, lambda x, e: oe.predict(x, True, e, faithful=True),
                                                          yexp, cexp, batched=False))
    equal = bool(np.array_equal(lp_orc, lp_ref))
    r = np.sort(to / tr)
    q = lambda f: float(np.quantile(r, f))
    print(f"cfg {cfg} shape (N={N}, d={d}, M={M}, P={P}, W={W}): bit-equal {equal}; {len(r)} alternating pairs: reference median "
          f"{np.median(tr):.4f} s (min {tr.min():.4f}, max {tr.max():.4f}), oracle faithful median {np.median(to):.4f} s (min "
          f"{to.min():.4f}, max {to.max():.4f}); per-pair ratio oracle / reference: median {q(0.5):.3f}, quartiles "
          f"{q(0.25):.3f} .. {q(0.75):.3f}, min {r[0]:.3f}, max {r[-1]:.3f}")
    assert equal, float(np.max(np.abs(lp_orc - lp_ref)))
    return q(0.5)


def main():
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count()
    import scipy
    import sklearn
    print(f"nproc {cores}, OMP_NUM_THREADS {os.environ.get('OMP_NUM_THREADS', 'unset')}, numpy {np.__version__}, "
          f"scipy {scipy.__version__}, scikit-learn {sklearn.__version__}; {REPS} alternating pairs after one warm-up pair")
    Emulator, mcmc = _import_reference()
    run(1, Emulator, mcmc)
    r3 = run(3, Emulator, mcmc)
    print(f"ok: the oracle's faithful mode returns the reference's numbers bit for bit; at the cfg-3 shape it takes {r3:.2f}x the "
          f"reference's wall time on this box (median of per-pair ratios; < 1 = the port is the faster of the two, i.e. bench.py's "
          f"cpu_baseline is, if anything, generous to the CPU)")


if __name__ == "__main__":
    main()
