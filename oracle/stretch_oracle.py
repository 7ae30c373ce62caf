"""
ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.

CPU restatement of the sampler step the reference drives through
`LoggingEnsembleSampler` (src/mcmc.py:68-92, 372-412): emcee's `StretchMove` inside
`RedBlueMove`, which is what `emcee.EnsembleSampler.sample` executes per iteration when no
`moves=` is given (src/mcmc.py:372-374 passes none).

emcee is a third-party dependency of the reference (requirements.txt:5, `emcee>=3.1.4`), not
vendored and NOT installed in the build container, so its published algorithm is restated here:

    emcee 3.1.4  emcee/moves/red_blue.py  RedBlueMove.propose   (Goodman & Weare 2010, parallel
                 emcee/moves/stretch.py   StretchMove.get_proposal           stretch move, a = 2)
                 emcee/moves/move.py      Move.update

Pinning: the reference holds no tests or vectors for the sampler, emcee cannot be imported, and its
MT19937 stream order (shuffle, rand, randint, rand per walker) is therefore not reproducible here:
this file restates the step with the random draws as INPUTS.  `stretch_step` is emcee's arithmetic,
statement for statement; what feeds it in the parity tests are the device's own draws (fetched
through a test hook, and regenerated independently by `device_draws` below), re-indexed into
emcee's order.  That pins the move (complementary-set indexing, proposal arithmetic, accept rule,
state update); the stream of random numbers itself stays "parity unpinned" and is covered
statistically (tests/test_gpu_sampler.py).

The counter-based generator of the build (Philox4x32-10, Salmon et al. 2011 / Random123) and its
keyed split permutation are restated here as well, so that the device's draws are checked against
an independent implementation; `philox4x32_10` is pinned by Random123's published known-answer
vectors (tests/test_oracle_stretch.py).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)        # Random123 philox.h: PHILOX_M4x32_0/1
W0, W1 = 0x9E3779B9, 0xBB67AE85                              # PHILOX_W32_0/1 (key schedule)
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(key, ctr):
    """Philox4x32-10.  key: (k0, k1), ctr: (c0, c1, c2, c3); every entry a uint32 scalar or array
    (broadcast).  Returns four uint32 arrays.  Round (Random123 philox.h `_philox4x32round`):
        hi0:lo0 = M0 * c0 ; hi1:lo1 = M1 * c2 ; out = (hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0)
    key bumped by (W0, W1) between rounds."""
    c = [np.asarray(x, dtype=np.uint64) & MASK32 for x in ctr]
    c = list(np.broadcast_arrays(*c))
    k0 = np.asarray(key[0], dtype=np.uint64) & MASK32
    k1 = np.asarray(key[1], dtype=np.uint64) & MASK32
    for _ in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & MASK32, p1 & MASK32,
             ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & MASK32, p0 & MASK32]
        k0 = (k0 + np.uint64(W0)) & MASK32
        k1 = (k1 + np.uint64(W1)) & MASK32
    return tuple(x.astype(np.uint32) for x in c)


def u01(hi, lo):
    """53-bit uniform in [0, 1) from two 32-bit words: ((hi << 32 | lo) >> 11) * 2^-53."""
    b = ((np.asarray(hi, dtype=np.uint64) << np.uint64(32)) | np.asarray(lo, dtype=np.uint64)) >> np.uint64(11)
    return b.astype(np.float64) * (1.0 / 9007199254740992.0)


def _mix32(x):
    x = np.asarray(x, dtype=np.uint64) & MASK32
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & MASK32
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & MASK32
    x ^= x >> np.uint64(16)
    return x


def split_perm(seed, step, n):
    """The build's keyed permutation pi_step of [0, n) that shuffles the red/blue split (its stand-in for
    emcee's `model.random.shuffle(inds)`): 4-round Feistel network on 2*hb >= log2(n) bits, keys from
    philox(seed; 0xFFFFFFFF, step, 0, 7), cycle walking back into [0, n)."""
    seed = int(seed)
    b = 1
    while (1 << b) < n:
        b += 1
    hb = (b + 1) // 2
    k = philox4x32_10((seed & 0xFFFFFFFF, seed >> 32), (0xFFFFFFFF, step & 0xFFFFFFFF, 0, 7))
    k0, k1 = np.uint64(k[0]), np.uint64(k[1])
    mask = np.uint64((1 << hb) - 1)
    out = np.empty(n, dtype=np.int64)
    x = np.arange(n, dtype=np.uint64)
    todo = np.arange(n)
    while todo.size:
        L, R = x >> np.uint64(hb), x & mask
        for r in range(4):
            F = _mix32(R ^ ((k0 + np.uint64(r) * np.uint64(W0)) & MASK32)) ^ _mix32((k1 + np.uint64(r)) & MASK32)
            L, R = R, L ^ (F & mask)
        x = (L << np.uint64(hb)) | R
        done = x < np.uint64(n)
        out[todo[done]] = x[done].astype(np.int64)
        todo, x = todo[~done], x[~done]
    return out


def device_draws(seed, step, half, nwalkers, randomize_split=True):
    """Every random number the build's stretch move uses for (seed, step, half), regenerated on the host:
    walker k of the half gets (u_z, j) from philox(seed; k, step, half, 0) and u_acc from
    philox(seed; k, step, half, 1); j = (word2 * nhalf) >> 32 is the index INTO THE COMPLEMENTARY HALF;
    member k of half h is walker perm[2k + h]."""
    seed = int(seed)
    nh = nwalkers // 2
    key = (seed & 0xFFFFFFFF, seed >> 32)
    k = np.arange(nh, dtype=np.uint64)
    r = philox4x32_10(key, (k, step & 0xFFFFFFFF, half, 0))
    a = philox4x32_10(key, (k, step & 0xFFFFFFFF, half, 1))
    perm = split_perm(seed, step, nwalkers) if randomize_split else np.arange(nwalkers, dtype=np.int64)
    return {"u_z": u01(r[0], r[1]), "j": ((r[2].astype(np.uint64) * np.uint64(nh)) >> np.uint64(32)).astype(np.int64),
            "u_acc": u01(a[0], a[1]), "perm": perm}


# --------------------------------------------------------------------------- emcee's step
def stretch_get_proposal(s, c, u_z, rint, a=2.0):
    """emcee/moves/stretch.py StretchMove.get_proposal with `random.rand(Ns)` = u_z and
    `random.randint(Nc, size=(Ns,))` = rint supplied by the caller:
        zz = ((a - 1) * rand + 1) ** 2 / a ; factors = (ndim - 1) * log(zz)
        q = c[rint] - (c[rint] - s) * zz[:, None]"""
    c = np.concatenate(c, axis=0)
    ndim = s.shape[1]
    zz = ((a - 1.0) * u_z + 1) ** 2.0 / a
    factors = (ndim - 1.0) * np.log(zz)
    return c[rint] - (c[rint] - s) * zz[:, None], factors


def stretch_step(coords, log_prob, inds, draws, log_prob_fn, a=2.0):
    """One iteration of emcee/moves/red_blue.py RedBlueMove.propose (nsplits = 2) with StretchMove, the random
    draws supplied: inds[nwalkers] in {0, 1} is the (shuffled) split label of every walker; draws[split] =
    (u_z[Ns], rint[Ns], u_acc[Ns]) in the order of the split's members by ascending walker index (emcee's
    `state.coords[inds == split]`).  Returns (coords, log_prob, accepted) — new arrays."""
    coords = np.array(coords, dtype=np.float64)
    log_prob = np.array(log_prob, dtype=np.float64)
    nwalkers = coords.shape[0]
    accepted = np.zeros(nwalkers, dtype=bool)
    all_inds = np.arange(nwalkers)
    for split in range(2):
        S1 = inds == split
        sets = [coords[inds == j] for j in range(2)]
        s = sets[split]
        c = sets[:split] + sets[split + 1:]
        u_z, rint, u_acc = draws[split]
        q, factors = stretch_get_proposal(s, c, u_z, rint, a)
        new_log_probs = np.asarray(log_prob_fn(q), dtype=np.float64)
        if np.any(np.isnan(new_log_probs)):
            raise ValueError("Probability function returned NaN")          # emcee/ensemble.py compute_log_prob
        for i, (j, f, nlp) in enumerate(zip(all_inds[S1], factors, new_log_probs)):
            lnpdiff = f + nlp - log_prob[j]
            if lnpdiff > np.log(u_acc[i]):
                accepted[j] = True
        m1 = S1 & accepted                                                 # Move.update(old, new, accepted, S1)
        m2 = accepted[S1]
        coords[m1] = q[m2]
        log_prob[m1] = new_log_probs[m2]
    return coords, log_prob, accepted


def emcee_order(perm, half, dev):
    """Re-index the device's per-member draws for half `half` into emcee's order.  Device member k of half h is
    walker perm[2k + h] and its complementary pick j means walker perm[2j + 1 - h]; emcee orders a split's members
    and the complementary set by ascending walker index.  Returns (inds_label_of_members, u_z, rint, u_acc)."""
    nh = perm.shape[0] // 2
    mine = perm[2 * np.arange(nh) + half]
    comp = perm[2 * np.arange(nh) + 1 - half]
    order = np.argsort(mine)                          # emcee position -> device member
    comp_sorted = np.sort(comp)
    rank_in_comp = np.searchsorted(comp_sorted, comp[dev["j"]])      # device member -> index into emcee's c
    return mine, dev["u_z"][order], rank_in_comp[order], dev["u_acc"][order]
