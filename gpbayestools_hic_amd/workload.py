"""
Synthetic benchmark workloads (BASELINE.md §4 / SURVEY §8d) assembled through the drop-in classes:
files in the reference's formats -> Emulator (fixed timing hyper-parameters) -> Chain.
"""
import os
import tempfile

import numpy as np

from . import synth


def build_chain(cfg, workdir=None, device=0, N=None, W=None, no_pca=False):
    """Returns (chain, emulator, info) for BASELINE config `cfg` on `device`.  no_pca: the emulator with perform_no_PCA (one GP
    per observable, src/emulator.py:562-565,589-592: SURVEY 8a's "cfg 4, also P = 64"); info["P"] is then the observable count."""
    from .emulator import Emulator
    from .mcmc import Chain
    c = dict(synth.CONFIGS[cfg])
    if N is not None:
        c["N"] = N
    if no_pca:
        c["P"] = c["M"]
    Nn, d, M, P = c["N"], c["d"], c["M"], c["P"]
    workdir = workdir or tempfile.mkdtemp(prefix="gpb_bench_")
    lo, hi = np.zeros(d), np.ones(d)
    X = synth.lhs(Nn, d)
    Y = synth.observables(X, M)
    tp, pf, ep = (os.path.join(workdir, n) for n in ("train.pkl", "par.txt", "exp.pkl"))
    synth.write_training_pickle(tp, X, Y, 0.01)
    synth.write_parameter_file(pf, lo, hi)
    emu = Emulator(training_set_path=tp, parameter_file=pf, npc=P, device=device, perform_no_PCA=bool(no_pca))
    ktype = {"RBF": "RBF", "Matern15": "Matern", "Matern25": "Matern25"}[c["kernel"]]
    emu.trainEmulator([True] * emu.nev, kernel_type=ktype, thetas=synth.fixed_theta(d, P))
    xstar = synth.truth_point(d)
    yexp = emu.predict(xstar[None, :], return_cov=False)[0]
    synth.write_experiment_pickle(ep, yexp, 0.05 * np.abs(yexp))
    chain = Chain(mcmc_path=os.path.join(workdir, "mcmc", "chain.pkl"), expdata_path=ep, model_parafile=pf,
                  device=device)
    chain.emuList = [emu]
    info = dict(c, X=X, Y=Y, lo=lo, hi=hi, xstar=xstar, yexp=yexp, workdir=workdir, kernel_type=ktype, no_pca=bool(no_pca))
    return chain, emu, info


def build_multi_chain(specs, d, workdir=None, device=0, mapped=False):
    """A chain of several emulators over one parameter space, as real analyses run it (nine emulators, sum of
    observables ~540: RunBayesianAnalysis.ipynb:35-48; src/mcmc.py:139-166).  specs: [(N, M, P, kernel)], every
    emulator with its own design, observables and (fixed) hyper-parameters.  mapped (True, or one flag per emulator): with
    parameterTrafoPCA (the GPs over the PCA-reduced parameters, src/emulator.py:492-551; d = 20).  Returns (chain, emulators, info)."""
    from .emulator import Emulator
    from .mcmc import Chain
    workdir = workdir or tempfile.mkdtemp(prefix="gpb_multi_")
    lo, hi = np.zeros(d), np.ones(d)
    pf, ep = os.path.join(workdir, "par.txt"), os.path.join(workdir, "exp.pkl")
    synth.write_parameter_file(pf, lo, hi)
    xstar = synth.truth_point(d)
    emus, data, yexp = [], [], []
    flags = [bool(mapped)] * len(specs) if isinstance(mapped, (bool, int)) else [bool(m) for m in mapped]
    for i, (N, M, P, kernel) in enumerate(specs):
        mapped = flags[i]
        X = synth.lhs(N, d, seed=synth.SEED + 100 + i)
        Y = synth.observables(X, M, seed=synth.SEED + 200 + i)
        tp = os.path.join(workdir, "train%d.pkl" % i)
        synth.write_training_pickle(tp, X, Y, 0.01)
        emu = Emulator(training_set_path=tp, parameter_file=pf, npc=P, device=device, parameterTrafoPCA=mapped)
        ktype = {"RBF": "RBF", "Matern15": "Matern", "Matern25": "Matern25"}[kernel]
        th = synth.fixed_theta(d, P, ell=1.2 + 0.1 * i, noise=0.03 + 0.01 * i)
        if mapped:      # length scales in units of each reduced input's extent, as the reference's bounds are (src/emulator.py:292-297)
            ext = np.ptp(emu.PCA_new_design_points, axis=0)
            th = synth.fixed_theta(len(ext), P, ell=1.2 + 0.1 * i, noise=0.03 + 0.01 * i)
            th[:, 1:-1] += np.log(ext)[None, :]
        emu.trainEmulator([True] * emu.nev, kernel_type=ktype, thetas=th)
        yexp.append(emu.predict(xstar[None, :], return_cov=False)[0])
        emus.append(emu)
        data.append((X, Y))
    yexp = np.concatenate(yexp)
    synth.write_experiment_pickle(ep, yexp, 0.05 * np.abs(yexp))
    chain = Chain(mcmc_path=os.path.join(workdir, "mcmc", "chain.pkl"), expdata_path=ep, model_parafile=pf,
                  device=device)
    chain.emuList = emus
    return chain, emus, dict(d=d, lo=lo, hi=hi, specs=list(specs), data=data, yexp=yexp, xstar=xstar, workdir=workdir)


def flops_per_walker(N, d, P, M, kernel="RBF"):
    """Algorithmic flops of one walker's log-posterior."""
    fpair = 3 * d + 3 if kernel == "RBF" else 3 * d + 10
    return P * (N * fpair + N * N + 4 * N) + 2 * P * M + 2 * P * M * M + M * M + M ** 3 / 3 + 2 * M * M + M
