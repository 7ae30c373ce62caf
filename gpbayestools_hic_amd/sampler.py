"""
Device-resident affine-invariant ensemble sampler: the emcee stretch move (a = 2) that the
reference drives through `LoggingEnsembleSampler` (src/mcmc.py:68-92, 372-412), with walker
positions, log-probabilities, proposals and the chain kept in HBM.

Per step and per half-ensemble ("red"/"blue"): gpb_stretch_propose -> log-probability of the
proposals (Chain.log_prob_device, optionally walker-sharded over ranks with one all-gather) ->
gpb_stretch_accept.  Nothing synchronises with the host inside the loop; Python only enqueues.
Random numbers come from a counter-based Philox generator keyed by (seed, step, half, walker),
so every rank of a sharded run generates identical proposals and accept decisions without
exchanging positions.

emcee itself is not installed in the build environment, so its MT19937 stream order cannot be
reproduced; equivalence with emcee is statistical (tests/test_gpu_sampler.py), while
log-probabilities at identical inputs are pinned by the golden vectors.
"""
import logging

import numpy as np

from . import _native as nat

log = logging.getLogger(__name__)


class StretchSampler:
    def __init__(self, chain, nwalkers, seed=None, a=2.0, logprob_device=None, sharding=None,
                 device=None, randomize_split=True):
        """chain: object with .min/.max/.ndim and log_prob_device(X_dev, out) (mcmc.Chain).
        logprob_device: optional override f(X_dev, out_dev) -> out_dev.
        sharding: optional dist.WalkerSharding (one process per GPU)."""
        import torch
        if nwalkers % 2:
            raise ValueError("nwalkers must be even")
        self.torch = torch
        self.chain_obj = chain
        self.ndim = int(chain.ndim)
        self.nwalkers = int(nwalkers)
        if self.nwalkers < 2 * self.ndim:
            log.warning("fewer walkers than 2*ndim")
        self.a = float(a)
        self.randomize_split = 1 if randomize_split else 0      # emcee's RedBlueMove default is True
        self.seed = int(np.random.SeedSequence(seed).generate_state(1, dtype=np.uint64)[0]) if seed is None \
            else int(seed)
        self.device = torch.device("cuda", chain.device if device is None else device)
        self.sharding = sharding
        self._logprob_override = logprob_device is not None
        self._logprob = logprob_device or chain.log_prob_device
        self._eng = None
        self._step_counter = 0
        nh = self.nwalkers // 2
        f64 = dict(dtype=torch.float64, device=self.device)
        self.pos = torch.empty((self.nwalkers, self.ndim), **f64)
        self.lp = torch.empty(self.nwalkers, **f64)
        self.q = torch.empty((nh, self.ndim), **f64)
        self.factor = torch.empty(nh, **f64)
        self.lpq = torch.empty(nh, **f64)
        self.naccept = torch.zeros(self.nwalkers, dtype=torch.int64, device=self.device)
        self.reset()

    # ------------------------------------------------------------------ helpers
    def _engine(self):
        if self._eng is None:
            emus = getattr(self.chain_obj, "emuList", None)
            if emus:
                self._eng = emus[0]._engine_ready()
            else:
                from .engine import GPEngine
                self._eng = GPEngine(self.device.index or 0)
        return self._eng

    def _eval(self, X_dev, out_dev):
        if self.sharding is not None:
            return self.sharding.logprob(self._logprob, X_dev, out_dev)
        return self._logprob(X_dev, out_dev)

    def reset(self):
        self._chain_dev = None
        self._lp_dev = None
        self.iterations = 0
        self.naccept.zero_()

    # ------------------------------------------------------------------ main loop
    def _resident_engine(self):
        """(engine, contexts, count) when gpb_chain_emcee_run can drive the whole loop from C (no Python per step), or
        None: the chain must be this package's Chain of native emulators (any number, with or without parameter maps),
        no log-probability override, and any sharding must go through the C ABI's own communicator on the FIRST
        emulator's engine with an even split (WalkerSharding.try_direct)."""
        ch = self.chain_obj
        emus = getattr(ch, "emuList", None)
        if self._logprob_override or not emus or not hasattr(ch, "_prepare_blocks"):
            return None
        if not all(hasattr(e, "_engine_ready") for e in emus):
            return None
        ch._prepare_blocks()
        eng = emus[0]._engine_ready()
        sh = self.sharding
        if sh is not None:
            if getattr(sh, "direct", None) is not eng or (self.nwalkers // 2) % sh.world:
                return None
        elif getattr(eng, "_dist_world", 0) > 1:
            return None                    # a communicator is installed but this sampler was not told to shard
        if ch.ndim != self.ndim:
            return None
        cc = ch._chain_contexts() if hasattr(ch, "_chain_contexts") else None
        if cc is None:
            # one emulator without a parameter map also runs uncompacted (non-PCA modes, tune("compact", 0))
            if len(emus) != 1 or getattr(emus[0], "parameterTrafoPCA_", False) or eng.d != self.ndim:
                return None
            eng._track_stream()
            cc = [eng], (nat.C.c_void_p * 1)(eng.h)
        return eng, cc[1], len(cc[0])

    def _nan_count(self, eng):
        """NaN log-probabilities the accept kernels have counted since the last call (synchronises; resets the counter)"""
        n = nat.c_i64(0)
        eng._ck(eng.lib.gpb_stretch_nan_count(eng.h, nat.C.byref(n), 1))
        return n.value

    def _snapshot(self):
        return (self.pos.clone(), self.lp.clone(), self.naccept.clone(), self.iterations, self._step_counter)

    def _restore(self, snap, keep_counter=False):
        """back to a snapshot; keep_counter: the ensemble only — the step counter (the key of the counter-based generators) runs
        on, so the next steps draw FRESH proposals from the restored ensemble instead of replaying the same ones"""
        self.pos.copy_(snap[0]); self.lp.copy_(snap[1]); self.naccept.copy_(snap[2])
        if not keep_counter:
            self.iterations, self._step_counter = snap[3], snap[4]

    def run(self, X0, nsteps, status=None, store=True):
        """Advance `nsteps` stretch-move steps from X0[nwalkers, ndim]; returns the final positions
        (numpy).  Mirrors LoggingEnsembleSampler.run_mcmc (src/mcmc.py:69-92)."""
        torch = self.torch
        eng = self._engine()
        lib, h = eng.lib, eng.h
        nw, d = self.nwalkers, self.ndim
        eng._track_stream()
        sh = self.sharding
        if sh is not None and getattr(sh, "world", 1) > 1 and hasattr(sh, "agree_state") and hasattr(self.chain_obj, "state_digest") \
                and not self._logprob_override:
            # sharded: rank r's rows are evaluated with rank r's GP state.  Every rank proves, before anything is accepted on
            # the strength of another rank's numbers, that all replicas are the same (raises on every rank otherwise) — at the
            # top of EVERY run(), unconditionally: whether a rank enters a collective must never hang on what only it knows
            # (which of its engines were rebuilt, whether it has sampled before)
            ch = self.chain_obj
            sh.agree_state(ch.state_digest_cached() if hasattr(ch, "state_digest_cached") else ch.state_digest())
        if X0 is not None:            # X0=None: continue from the resident state
            self.pos.copy_(torch.as_tensor(np.ascontiguousarray(X0, dtype=np.float64)))
            shd = self.sharding
            if shd is not None and getattr(shd, "world", 1) > 1 and hasattr(shd, "rows_agree_begin"):
                # replicated sampler: every rank must start from THE SAME ensemble (checksum all-reduce; raises on every rank)
                shd.rows_agree_end(shd.rows_agree_begin(self.pos))
            self._eval(self.pos, self.lp)
            if torch.isnan(self.lp).any().item():
                raise ValueError("The initial log_prob was NaN")          # emcee's message
        cd = ld = None
        if store:
            cd = torch.empty((nsteps, nw, d), dtype=torch.float64, device=self.device)
            ld = torch.empty((nsteps, nw), dtype=torch.float64, device=self.device)
        if status is None:
            status = max(nsteps // 10, 1)
        res = self._resident_engine()
        if res is not None and self.sharding is not None:
            # sharded C loop: whatever can fail on ONE rank (state checks, workspace allocation) is done now, and the
            # ranks agree on the outcome BEFORE any of them enqueues the in-stream all-gathers of gpb_chain_emcee_run —
            # a rank that failed inside that call would leave its peers waiting in a collective that never completes
            rc = res[0].lib.gpb_chain_emcee_prepare(res[1], res[2], nw)
            agree = getattr(self.sharding, "_all_ok", None) if getattr(self.sharding, "world", 1) > 1 else None
            if not (agree(rc == 0) if agree is not None else rc == 0):
                if rc != 0:
                    res[0]._ck(rc)
                raise RuntimeError("gpb_chain_emcee_prepare failed on another rank; no collective was enqueued")
        ceng = res[0] if res is not None else eng
        self._nan_count(ceng)       # the counter is per context: whatever an earlier, aborted user left behind is not ours
        n = 0
        while n < nsteps:
            m = min(status - n % status, nsteps - n)          # up to the next status line
            snap = self._snapshot()                           # device copies (nwalkers x ndim doubles): see the NaN check below
            if res is not None:
                # the C ABI enqueues all m steps itself (gpb_emcee_run): propose -> GP predict -> block likelihood +
                # prior box -> [in-stream all-gather] -> accept, no Python in between
                lo, hi = self.chain_obj._box(self.device)
                reng, ctxs, nemu = res
                reng._ck(lib.gpb_chain_emcee_run(ctxs, nemu, nat.ptr(self.pos), nat.ptr(self.lp), nw, m, self.seed,
                                                 self._step_counter, self.a, self.randomize_split, nat.ptr(lo),
                                                 nat.ptr(hi), float("-inf"), self.chain_obj.inside_const,
                                                 nat.ptr(cd[n:n + m]) if store else None,
                                                 nat.ptr(ld[n:n + m]) if store else None, nat.ptr(self.naccept)))
                self._step_counter += m
            else:
                for i in range(m):
                    step = self._step_counter
                    self._step_counter += 1
                    for half in (0, 1):
                        eng._ck(lib.gpb_stretch_propose(h, nat.ptr(self.pos), nw, d, half, self.seed, step, self.a,
                                                        nat.ptr(self.q), nat.ptr(self.factor), self.randomize_split))
                        self._eval(self.q, self.lpq)
                        eng._ck(lib.gpb_stretch_accept(h, nat.ptr(self.pos), nat.ptr(self.lp), nw, d, half, self.seed,
                                                       step, nat.ptr(self.q), nat.ptr(self.factor), nat.ptr(self.lpq),
                                                       nat.ptr(self.naccept), self.randomize_split))
                    if store:
                        cd[n + i].copy_(self.pos)
                        ld[n + i].copy_(self.lp)
            n += m
            self.iterations += m
            if n % status == 0 or n == nsteps:
                # emcee raises "Probability function returned NaN" at the offending step with its state untouched
                # (emcee/ensemble.py); the device loop rejects and counts such proposals and the count is read once per
                # status block (one synchronisation): on a NaN the sampler goes back to the state in front of the block
                bad = self._nan_count(ceng)
                if bad:
                    # back to the state in front of the offending status block; the blocks of this call before it stand: their
                    # samples are kept, so that chain / lnprobability, `iterations` and the acceptance counts all describe
                    # the same n - m steps (emcee, which checks every step, stops AT the offending step with everything
                    # before it stored: the difference is the block's granularity, `status`)
                    self._restore(snap)
                    if store and n - m > 0:
                        self._chain_dev = cd[:n - m].clone() if self._chain_dev is None else torch.cat([self._chain_dev, cd[:n - m]], 0)
                        self._lp_dev = ld[:n - m].clone() if self._lp_dev is None else torch.cat([self._lp_dev, ld[:n - m]], 0)
                    raise ValueError("Probability function returned NaN (%d proposal(s) within steps %d..%d of this run; "
                                     "the sampler is back at its state before step %d: the %d step(s) of this call before it "
                                     "are kept, stored and counted)" % (bad, n - m + 1, n, n - m + 1, n - m))
                af = self.acceptance_fraction
                log.info("step %d: acceptance fraction: mean %.4f, std %.4f, min %.4f, max %.4f",
                         n, af.mean(), af.std(), af.min(), af.max())
        if store:
            self._chain_dev = cd if self._chain_dev is None else torch.cat([self._chain_dev, cd], 0)
            self._lp_dev = ld if self._lp_dev is None else torch.cat([self._lp_dev, ld], 0)
        return self.pos.cpu().numpy()

    # ------------------------------------------------------------------ emcee-style accessors
    @property
    def acceptance_fraction(self):
        return self.naccept.cpu().numpy() / max(self.iterations, 1)

    @property
    def chain(self):
        """[nwalkers, nsteps, ndim] like emcee's `sampler.chain`."""
        return self._chain_dev.permute(1, 0, 2).contiguous().cpu().numpy()

    @property
    def lnprobability(self):
        """[nwalkers, nsteps]."""
        return self._lp_dev.permute(1, 0).contiguous().cpu().numpy()

    @property
    def flatchain(self):
        return self._chain_dev.reshape(-1, self.ndim).cpu().numpy()

    @property
    def flatlnprobability(self):
        return self._lp_dev.reshape(-1).cpu().numpy()


class _HostLogProb:
    """minimal chain-like object around a host callable f(X[w, ndim]) -> lp[w]"""

    def __init__(self, ndim, fn, device):
        self.ndim, self.fn, self.device, self.emuList = int(ndim), fn, int(device), []

    def log_prob_device(self, X_dev, out):
        import torch
        lp = np.asarray(self.fn(X_dev.cpu().numpy()), dtype=np.float64).reshape(-1)
        out.copy_(torch.as_tensor(lp, device=out.device))
        return out


class LoggingEnsembleSampler(StretchSampler):
    """The reference's sampler class by name and call signature (src/mcmc.py:68-92, 372-412):
    `LoggingEnsembleSampler(nwalkers, ndim, log_prob_fn, pool=...)`, `run_mcmc(X0, nsteps, status=)`, then
    `.chain`, `.flatchain`, `.lnprobability`, `.flatlnprobability`, `.acceptance_fraction`, `.reset()`.
    When `log_prob_fn` is `Chain.log_posterior` of this package the whole loop stays on the device;
    any other callable f(X[w, ndim]) -> lp[w] is called on the host once per half-ensemble (the stretch
    move itself still runs on the device).  `pool` is accepted and ignored: batches are already whole."""

    def __init__(self, nwalkers, ndim, log_prob_fn, pool=None, a=2.0, seed=None, device=0, **_ignored):
        from .mcmc import Chain
        owner = getattr(log_prob_fn, "__self__", None)
        if isinstance(owner, Chain) and owner._native() and getattr(log_prob_fn, "__func__", None) is Chain.log_posterior:
            if int(ndim) != owner.ndim:
                raise ValueError("ndim does not match the chain's number of parameters")
            super().__init__(owner, nwalkers, seed=seed, a=a)
        else:
            super().__init__(_HostLogProb(ndim, log_prob_fn, device), nwalkers, seed=seed, a=a)

    def run_mcmc(self, X0, nsteps, status=None, **_kwargs):
        log.info("running %d walkers for %d steps", self.nwalkers, nsteps)
        return self.run(X0, nsteps, status=status)
