"""Fork contract of the B2 seam: pocoMC with `pool=int` calls the likelihood in FORKED workers that
inherited the whole Chain (/root/reference/src/mcmc.py:775-776, 798-804).  A GPU context does not survive a fork, so a
handle created in the parent must raise a Python error in the child before any HIP call — never a GPU fault.  The tests
change what os.getpid() reports instead of forking (never fork a GPU-initialised process on the GPU box)."""
import os

import numpy as np
import pytest

from gpbayestools_hic_amd import engine as E


class _Lib:
    """stands where libgpbayes.so would: any C-ABI call through it is a failure of the guard"""

    def __getattr__(self, name):
        raise AssertionError("C ABI reached in a forked child: %s" % name)


def _bare_engine():
    eng = object.__new__(E.GPEngine)             # no device in the CPU suite: a handle without a context behind it
    eng.lib, eng._h, eng._pid, eng.device = _Lib(), 0xdead, os.getpid(), 0
    eng._follow_torch, eng._stream, eng.N = False, None, 4
    return eng


def test_handle_is_refused_in_another_process(monkeypatch):
    eng = _bare_engine()
    assert eng.h == 0xdead                       # same process: handed out
    monkeypatch.setattr(os, "getpid", lambda: eng._pid + 1)
    for call in (lambda: eng.h, eng.sync, eng._need_data, eng._track_stream,
                 lambda: eng.set_data(np.zeros((4, 2)), np.zeros((1, 4))), lambda: eng.predict(np.zeros((1, 2)))):
        with pytest.raises(RuntimeError, match="does not survive a fork"):
            call()
    eng.close()                                  # the child's copy is dropped without touching the library
    assert eng._h is None


def test_emulator_refuses_inherited_engine(monkeypatch):
    from gpbayestools_hic_amd.emulator import Emulator
    emu = object.__new__(Emulator)
    emu._engine, emu._trained = _bare_engine(), True
    assert emu._engine_ready() is emu._engine
    monkeypatch.setattr(os, "getpid", lambda: emu._engine._pid + 1)
    with pytest.raises(RuntimeError, match="pool=None"):
        emu._engine_ready()
    with pytest.raises(RuntimeError, match="pool=None"):
        emu._new_engine()


def test_pickled_emulator_carries_no_handle():
    """what a 'spawn' worker (or dill, src/mcmc.py:145-150) receives: no engine, so it builds its own"""
    from gpbayestools_hic_amd.emulator import Emulator
    emu = object.__new__(Emulator)
    emu.__dict__.update(_engine=_bare_engine(), _like_key=1, fit_sharding=None, _trained=False)
    st = emu.__getstate__()
    assert st["_engine"] is None and st["_like_key"] is None


@pytest.mark.gpu
def test_real_engine_raises_after_simulated_fork(monkeypatch):
    from gpbayestools_hic_amd import GPEngine, synth
    from gpbayestools_hic_amd import mcmc
    eng = GPEngine(0)
    eng.set_data(synth.lhs(64, 3), np.zeros((1, 64)))
    real = os.getpid()
    util = mcmc._utility_engine(0)
    monkeypatch.setattr(os, "getpid", lambda: real + 1)
    with pytest.raises(RuntimeError, match="does not survive a fork"):
        eng.set_theta(np.zeros((1, 5)))
    with pytest.raises(RuntimeError, match="does not survive a fork"):
        mcmc._utility_engine(0)
    with pytest.raises(RuntimeError, match="does not survive a fork"):
        mcmc.mvn_loglike(np.zeros(2), np.eye(2))
    monkeypatch.setattr(os, "getpid", lambda: real)
    assert util.h is not None
    eng.set_theta(synth.fixed_theta(3, 1))       # back in the owning process: works
    eng.factor()
    eng.close()
