"""
GPU tier: the stretch-move step itself against the oracle's restatement of emcee's
RedBlueMove.propose + StretchMove.get_proposal (oracle/stretch_oracle.py), bit for bit:
  * the device's Philox4x32-10 reproduces Random123's known-answer vectors,
  * the draws k_propose / k_accept use (fetched through gpb_test_stretch_draws) equal the oracle's independent
    regeneration of them,
  * 10 steps of the device sampler == 10 iterations of emcee's algorithm fed with those draws (re-indexed into
    emcee's member order): positions, log-probabilities and acceptance counts identical to the last bit.
"""
import types

import numpy as np
import pytest

from oracle import stretch_oracle as S
from test_oracle_stretch import KAT

pytestmark = pytest.mark.gpu


def _engine():
    from gpbayestools_hic_amd import GPEngine
    return GPEngine(0)


def test_device_philox_known_answers_and_oracle():
    from gpbayestools_hic_amd import _native as nat
    eng = _engine()
    rng = np.random.default_rng(0)
    extra = rng.integers(0, 2 ** 32, size=(200, 6), dtype=np.uint64).astype(np.uint32)
    kat = np.array([[k[1][0], k[1][1], *k[0]] for k in KAT], dtype=np.uint32)
    inp = np.ascontiguousarray(np.concatenate([kat, extra]))
    out = np.zeros((len(inp), 4), dtype=np.uint32)
    eng._ck(eng.lib.gpb_test_philox(eng.h, len(inp), nat.ptr(inp), nat.ptr(out)))
    for i, (_, _, want) in enumerate(KAT):
        assert tuple(int(x) for x in out[i]) == want
    ref = np.stack(S.philox4x32_10((inp[:, 0], inp[:, 1]), tuple(inp[:, 2 + i] for i in range(4))), axis=1)
    assert np.array_equal(out, ref)
    eng.close()


def _device_draws(eng, nw, half, seed, step, randomize):
    import torch
    from gpbayestools_hic_amd import _native as nat
    nh = nw // 2
    u_z = torch.empty(nh, dtype=torch.float64, device="cuda")
    u_acc = torch.empty(nh, dtype=torch.float64, device="cuda")
    j = torch.empty(nh, dtype=torch.int64, device="cuda")
    perm = torch.empty(nw, dtype=torch.int64, device="cuda")
    eng._ck(eng.lib.gpb_test_stretch_draws(eng.h, nw, half, seed, step, 1 if randomize else 0, nat.ptr(u_z), nat.ptr(j),
                                           nat.ptr(u_acc), nat.ptr(perm)))
    torch.cuda.synchronize()
    return {"u_z": u_z.cpu().numpy(), "j": j.cpu().numpy(), "u_acc": u_acc.cpu().numpy(), "perm": perm.cpu().numpy()}


@pytest.mark.parametrize("nw", [2, 6, 64, 100, 4096])
def test_device_draws_equal_the_oracles_regeneration(nw):
    eng = _engine()
    for seed, step in ((12345, 0), (2 ** 63 + 17, 3), (7, 2 ** 32 - 1)):
        for half in (0, 1):
            for randomize in (True, False):
                got = _device_draws(eng, nw, half, seed, step, randomize)
                ref = S.device_draws(seed, step, half, nw, randomize)
                for k in ("u_z", "j", "u_acc", "perm"):
                    assert np.array_equal(got[k], ref[k]), (nw, seed, step, half, randomize, k)
    eng.close()


def _toy(q):
    """a banana-shaped log-density with a hard wall: -inf outside |x| < 4 (exercises the reject-always path)"""
    lp = -0.5 * (q[:, 0] ** 2 / 4.0 + np.sum((q[:, 1:] - 0.3 * q[:, :1] ** 2) ** 2, axis=1))
    return np.where(np.all(np.abs(q) < 4.0, axis=1), lp, -np.inf)


def _oracle_chain(X0, nsteps, seed, randomize, eng, logprob):
    """emcee's algorithm (oracle) driven by the device's draws fetched through the test hook"""
    nw = X0.shape[0]
    X, lp = X0.copy(), logprob(X0)
    nacc = np.zeros(nw, dtype=np.int64)
    hist = []
    for step in range(nsteps):
        dev = [_device_draws(eng, nw, h, seed, step, randomize) for h in (0, 1)]
        perm = dev[0]["perm"]
        inds = np.empty(nw, dtype=np.int64)
        inds[perm[0::2]], inds[perm[1::2]] = 0, 1
        draws = [S.emcee_order(perm, h, dev[h])[1:] for h in (0, 1)]
        X, lp, acc = S.stretch_step(X, lp, inds, draws, logprob)
        nacc += acc
        hist.append((X.copy(), lp.copy()))
    return X, lp, nacc, hist


@pytest.mark.parametrize("nw,d,randomize", [(32, 3, True), (32, 3, False), (10, 5, True), (256, 2, True)])
def test_ten_steps_match_emcees_algorithm_bit_for_bit(nw, d, randomize):
    import torch
    from gpbayestools_hic_amd import StretchSampler
    fake = types.SimpleNamespace(ndim=d, device=0, min=np.full(d, -9.0), max=np.full(d, 9.0), emuList=[])

    def logprob_dev(X_dev, out):                 # the toy density evaluated by the SAME numpy code as the oracle's
        out.copy_(torch.as_tensor(_toy(X_dev.cpu().numpy()), device=out.device))
        return out

    seed = 424242
    X0 = np.random.default_rng(nw + d).normal(size=(nw, d)) * 1.5
    s = StretchSampler(fake, nw, seed=seed, logprob_device=logprob_dev, randomize_split=randomize)
    s.run(X0, 10)
    eng = s._engine()
    X, lp, nacc, hist = _oracle_chain(X0, 10, seed, randomize, eng, _toy)
    chain, lnp = s.chain, s.lnprobability        # [nw, nsteps, d], [nw, nsteps]
    for n, (Xn, lpn) in enumerate(hist):
        assert np.array_equal(chain[:, n], Xn), ("positions differ at step", n)
        assert np.array_equal(lnp[:, n], lpn), ("log-probabilities differ at step", n)
    assert np.array_equal(s.naccept.cpu().numpy(), nacc)
    assert 0 < nacc.sum() < 10 * nw


def test_chain_log_posterior_steps_match_emcees_algorithm(tmp_path):
    """the same on the real device log-posterior (BASELINE config 1): emcee's algorithm with the ORACLE's
    log-posterior walks the same ensemble — accept decisions identical, log-probabilities within the 1e-10 bar"""
    from gpbayestools_hic_amd import StretchSampler, synth
    from gpbayestools_hic_amd.workload import build_chain
    from oracle import gp_oracle as O
    chain, emu, info = build_chain(1, workdir=str(tmp_path))
    d, P = info["d"], info["P"]
    oe = O.OracleEmulator(info["X"], info["Y"], info["lo"], info["hi"], P).fit(synth.fixed_theta(d, P))
    yexp = info["yexp"]; cexp = np.diag((0.05 * np.abs(yexp)) ** 2)
    logpost = lambda q: O.log_prob(q, info["lo"], info["hi"], lambda x, e: oe.predict(x, True, e), yexp, cexp)
    nw, seed = 32, 99
    X0 = synth.walkers(nw, d, seed=5)
    s = StretchSampler(chain, nw, seed=seed)
    s.run(X0, 10)
    X, lp, nacc, hist = _oracle_chain(X0, 10, seed, True, s._engine(), logpost)
    assert np.array_equal(s.naccept.cpu().numpy(), nacc)
    assert np.array_equal(s.chain[:, -1], X)                       # same decisions => same positions, bit for bit
    fin = np.isfinite(lp)
    assert np.max(np.abs(s.lnprobability[:, -1][fin] - lp[fin]) / np.abs(lp[fin])) < 1e-10
