"""CPU tier: the oracle's restatement of the sampler step (oracle/stretch_oracle.py).
Philox4x32-10 against Random123's published known-answer vectors (kat_vectors: the three philox4x32 10 lines),
the split permutation's bijectivity, and properties of emcee's stretch move that hold for any draws."""
import numpy as np

from oracle import stretch_oracle as S

# Random123 tests/kat_vectors, "philox4x32 10": counter words, key words, expected output
KAT = [((0x00000000,) * 4, (0x00000000, 0x00000000), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
       ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
       ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]


def test_philox_known_answers():
    for ctr, key, want in KAT:
        assert tuple(int(x) for x in S.philox4x32_10(key, ctr)) == want
    # vectorised call = element-wise scalar calls
    ks = np.array([k[1][0] for k in KAT]), np.array([k[1][1] for k in KAT])
    cs = tuple(np.array([k[0][i] for k in KAT]) for i in range(4))
    out = S.philox4x32_10(ks, cs)
    for i, (_, _, want) in enumerate(KAT):
        assert tuple(int(o[i]) for o in out) == want


def test_u01_range_and_resolution():
    assert S.u01(0, 0) == 0.0
    assert S.u01(0xFFFFFFFF, 0xFFFFFFFF) == 1.0 - 2.0 ** -53
    assert S.u01(0x80000000, 0) == 0.5


def test_split_perm_is_a_bijection_and_keyed_by_step():
    for n in (2, 6, 64, 100, 4096, 5000):
        p0, p1 = S.split_perm(99, 0, n), S.split_perm(99, 1, n)
        assert np.array_equal(np.sort(p0), np.arange(n)) and np.array_equal(np.sort(p1), np.arange(n))
        if n >= 64:
            assert not np.array_equal(p0, p1)
            assert not np.array_equal(p0, S.split_perm(100, 0, n))


def test_device_draws_shapes_and_ranges():
    d = S.device_draws(12345, 7, 1, 64)
    assert d["u_z"].shape == d["u_acc"].shape == d["j"].shape == (32,) and d["perm"].shape == (64,)
    assert np.all((d["u_z"] >= 0) & (d["u_z"] < 1)) and np.all((d["j"] >= 0) & (d["j"] < 32))
    other = S.device_draws(12345, 7, 0, 64)
    assert not np.array_equal(d["u_z"], other["u_z"]) and np.array_equal(d["perm"], other["perm"])
    assert np.array_equal(S.device_draws(1, 0, 0, 8, randomize_split=False)["perm"], np.arange(8))


def _random_draws(rng, nh):
    return [(rng.random(nh), rng.integers(0, nh, nh), rng.random(nh)) for _ in range(2)]


def test_stretch_step_properties():
    """What emcee's step guarantees whatever the draws: proposals lie on the line through the walker and its
    partner, rejected walkers keep position and log-probability bit for bit, accepted ones carry the proposal's,
    -inf proposals are never accepted, a NaN log-probability raises."""
    rng = np.random.default_rng(3)
    nw, d = 24, 3
    X = rng.normal(size=(nw, d))
    f = lambda q: -0.5 * np.sum(q * q, axis=1)
    lp = f(X)
    inds = np.arange(nw) % 2
    rng.shuffle(inds)
    draws = _random_draws(rng, nw // 2)
    X1, lp1, acc = S.stretch_step(X, lp, inds, draws, f)
    assert acc.dtype == bool and 0 < acc.sum() < nw
    assert np.array_equal(X1[~acc], X[~acc]) and np.array_equal(lp1[~acc], lp[~acc])
    assert np.array_equal(lp1[acc], f(X1[acc]))
    # first split: proposal geometry  q = c + zz (s - c)
    s, c = X[inds == 0], X[inds == 1]
    q, fac = S.stretch_get_proposal(s, [c], draws[0][0], draws[0][1])
    zz = ((2.0 - 1.0) * draws[0][0] + 1) ** 2 / 2.0
    assert np.all((zz >= 0.5) & (zz <= 2.0))
    assert np.allclose(q, c[draws[0][1]] + zz[:, None] * (s - c[draws[0][1]]), rtol=1e-14, atol=1e-14)
    assert np.array_equal(fac, (d - 1.0) * np.log(zz))
    # -inf outside a box: such proposals never replace a walker
    g = lambda q: np.where(np.all(np.abs(q) < 1.5, axis=1), f(q), -np.inf)
    X2, lp2, acc2 = S.stretch_step(X, g(X), inds, draws, g)
    assert np.all(np.isfinite(lp2[acc2]))
    try:
        S.stretch_step(X, lp, inds, draws, lambda q: np.full(len(q), np.nan))
    except ValueError as e:
        assert "NaN" in str(e)
    else:
        raise AssertionError("NaN log-probability must raise (emcee's contract)")


def test_emcee_order_reindexing_is_consistent():
    """device member k of half h = walker perm[2k+h]; after re-indexing, emcee's k-th member (ascending walker
    index) receives that walker's draws and its partner is the same walker"""
    nw = 40
    dev = S.device_draws(5, 3, 0, nw)
    perm = dev["perm"]
    mine, u_z, rint, u_acc = S.emcee_order(perm, 0, dev)
    inds = np.empty(nw, dtype=np.int64)
    inds[perm[0::2]] = 0
    inds[perm[1::2]] = 1
    members = np.flatnonzero(inds == 0)                  # emcee order
    comp = np.flatnonzero(inds == 1)
    for pos, w in enumerate(members):
        k = int(np.flatnonzero(mine == w)[0])           # device member holding walker w
        assert u_z[pos] == dev["u_z"][k] and u_acc[pos] == dev["u_acc"][k]
        assert comp[rint[pos]] == perm[2 * dev["j"][k] + 1]
