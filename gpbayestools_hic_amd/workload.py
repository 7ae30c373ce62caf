"""
Synthetic benchmark workloads (BASELINE.md §4 / SURVEY §8d) assembled through the drop-in classes:
files in the reference's formats -> Emulator (fixed timing hyper-parameters) -> Chain.
"""
import os
import tempfile

import numpy as np

from . import synth


def build_chain(cfg, workdir=None, device=0, N=None, W=None):
    """Returns (chain, emulator, info) for BASELINE config `cfg` on `device`."""
    from .emulator import Emulator
    from .mcmc import Chain
    c = dict(synth.CONFIGS[cfg])
    if N is not None:
        c["N"] = N
    Nn, d, M, P = c["N"], c["d"], c["M"], c["P"]
    workdir = workdir or tempfile.mkdtemp(prefix="gpb_bench_")
    lo, hi = np.zeros(d), np.ones(d)
    X = synth.lhs(Nn, d)
    Y = synth.observables(X, M)
    tp, pf, ep = (os.path.join(workdir, n) for n in ("train.pkl", "par.txt", "exp.pkl"))
    synth.write_training_pickle(tp, X, Y, 0.01)
    synth.write_parameter_file(pf, lo, hi)
    emu = Emulator(training_set_path=tp, parameter_file=pf, npc=P, device=device)
    ktype = {"RBF": "RBF", "Matern15": "Matern", "Matern25": "Matern25"}[c["kernel"]]
    emu.trainEmulator([True] * emu.nev, kernel_type=ktype, thetas=synth.fixed_theta(d, P))
    xstar = synth.truth_point(d)
    yexp = emu.predict(xstar[None, :], return_cov=False)[0]
    synth.write_experiment_pickle(ep, yexp, 0.05 * np.abs(yexp))
    chain = Chain(mcmc_path=os.path.join(workdir, "mcmc", "chain.pkl"), expdata_path=ep, model_parafile=pf,
                  device=device)
    chain.emuList = [emu]
    info = dict(c, X=X, Y=Y, lo=lo, hi=hi, xstar=xstar, yexp=yexp, workdir=workdir, kernel_type=ktype)
    return chain, emu, info


def flops_per_walker(N, d, P, M, kernel="RBF"):
    """Algorithmic flops of one walker's log-posterior."""
    fpair = 3 * d + 3 if kernel == "RBF" else 3 * d + 10
    return P * (N * fpair + N * N + 4 * N) + 2 * P * M + 2 * P * M * M + M * M + M ** 3 / 3 + 2 * M * M + M
