"""
Drop-in replacement for the reference's `src/mcmc.py` log-posterior path and emcee driver
(B2 protocol): `mvn_loglike`, `Chain` with `log_prior / log_likelihood /
log_posterior / _predict / run_mcmc / run_pocoMC / compute_log_likelihood_for_chain`.

Every emulator prediction, covariance assembly, Cholesky and quadratic form runs in the HIP
engine.  When all emulators in `emuList` are this package's `Emulator`, a log-probability call
is E fused `gpb_loglike` block evaluations plus one box kernel; rows never leave HBM between
them.  Foreign emulators (anything with the reference's predict protocol) go through
`_predict` on the host and the generic batched device MVN (`gpb_mvn_loglike`).
"""
import logging
import math
import pickle
from pathlib import Path

import numpy as np

from .emulator import Emulator
from .engine import GPEngine
from .preprocess import parse_model_parameter_file
from .sampler import LoggingEnsembleSampler  # noqa: F401  (the reference defines it in this module)

log = logging.getLogger(__name__)

# the reference adds 2*log(extra_std + 1e-16) - extra_std/scale with extra_std == 0*X[:, -1]
# (src/mcmc.py:205,220-221 and :281,296-297): a constant inside the box
EXTRA_STD_CONST = 2.0 * math.log(1e-16)

_util_engine = None


def _utility_engine(device=0):
    global _util_engine
    if _util_engine is None:
        _util_engine = GPEngine(device)
    _util_engine._check_pid()            # created in the parent of a fork: RuntimeError before any HIP call
    return _util_engine


def mvn_loglike(y, cov):
    """Unnormalised multivariate-normal log-likelihood -1/2 y^T C^-1 y - 1/2 log det C
    (src/mcmc.py:23-65), evaluated by the device Cholesky kernel.  A covariance that is not
    positive definite raises numpy.linalg.LinAlgError (the reference's branch for this case is
    unreachable, src/mcmc.py:44-54)."""
    eng = _utility_engine()
    out = eng.mvn_loglike(np.asarray(y, float)[None, :], np.asarray(cov, float)[None, :, :])
    if eng.last_not_pd:
        raise np.linalg.LinAlgError("mvn_loglike: covariance is not positive definite")
    return float(out[0])


class Chain:
    def __init__(self, mcmc_path="./mcmc/chain.pkl", expdata_path="./exp_data.dat",
                 model_parafile="./model.dat", device=0):
        self.mcmc_path = Path(mcmc_path)
        self.mcmc_path.parent.mkdir(exist_ok=True)
        self.pardict = parse_model_parameter_file(model_parafile)
        self.ndim = len(self.pardict)
        self.label = [v[0] for v in self.pardict.values()]
        self.min = np.array([v[1] for v in self.pardict.values()])
        self.max = np.array([v[2] for v in self.pardict.values()])
        self.prior_volume_ = np.prod(self.max - self.min)
        self.expdata, self.expdata_cov = self._read_in_exp_data_pickle(expdata_path)
        self.nobs = self.expdata.shape[1]
        self.emuList = []
        self.chain = False
        self.device = device
        self._like_sig = None
        self.sharding = None                     # dist.WalkerSharding: see shard_over

    def __getstate__(self):                      # no device handles in pickles / forked workers
        st = dict(self.__dict__)
        st.pop("_box_dev", None)
        st.pop("_box_key", None)
        st["_like_sig"] = None
        st["sharding"] = None
        st.pop("_digest_cache", None)
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self.__dict__.setdefault("sharding", None)

    def shard_over(self, sharding):
        """Evaluate every log_posterior / log_likelihood batch in row shares over the ranks of `sharding`
        (dist.WalkerSharding; one process per GPU): rank r takes rows [r chunk, (r + 1) chunk) of the batch, one
        all-gather of the shares completes the vector on every rank.  This is the multi-GPU form of the call pocoMC makes —
        log_likelihood(X[n_active, ndim], finite=True) once per batch, src/mcmc.py:798-805 — with the sampler itself
        replicated: EVERY rank must make the same calls with the same rows, and all ranks must hold the same emulators
        (WalkerSharding.replicate).  Both are checked on EVERY call by the one all-reduce that rides in front of the batch (the
        rows' checksum and the four words of the state digest: a batch whose rows — or a replica whose model — differ between
        the ranks raises on all of them), so the sequence of collectives a rank issues never depends on what only that rank
        knows.  A row's value does not depend on the split.  `None` switches it off."""
        self.sharding = sharding
        return self

    # ------------------------------------------------------------------ inputs
    def _read_in_exp_data_pickle(self, filepath):
        """One experimental event: values and errors; covariance = diag(err^2)
        (src/mcmc.py:302-324)."""
        with open(filepath, "rb") as fp:
            data = pickle.load(fp)
        vals = np.array([data[k]["obs"][0] for k in data.keys()])
        errs = np.nan_to_num(np.abs(np.array([data[k]["obs"][1] for k in data.keys()])))
        cov = np.zeros((vals.shape[1], vals.shape[1]))
        np.fill_diagonal(cov, errs.flatten() ** 2)       # (nobs x nobs whatever the number of events, as the reference fills it)
        return vals, cov

    def state_digest_cached(self):
        """state_digest(), recomputed only when an emulator was re-trained / re-loaded or the experiment replaced (new arrays):
        the key is rank-local, and only ever decides which VALUE this rank contributes to a check — never whether it takes part"""
        # (an emulator's fitted state by its serial number — new with every training and every __setstate__, never reused;
        # the experiment block by identity AND a content probe: id() alone can come back after an array was dropped)
        # ... the WHOLE covariance (an in-place edit off the diagonal must not leave the digest stale): its bytes up to 128
        # observables, two weighted sums over every element beyond that (a 540 x 540 block is 2.3 MB per call otherwise)
        cov = np.atleast_2d(np.asarray(self.expdata_cov, dtype=np.float64))
        if cov.size <= 128 * 128:
            cov_probe = cov.tobytes()
        else:
            w = 1.0 + np.arange(cov.shape[1], dtype=np.float64) / cov.shape[1]
            u = 1.0 + np.arange(cov.shape[0], dtype=np.float64)[::-1] / cov.shape[0]
            cov_probe = (float(cov.sum()), float(u @ cov @ w), np.diagonal(cov).tobytes())
        key = (id(self.expdata), id(self.expdata_cov), np.asarray(self.expdata).tobytes(),
               cov_probe, self.min.tobytes(), self.max.tobytes(),
               tuple((id(e), getattr(e, "_state_serial", None)) for e in self.emuList))
        if any(k[1] is None for k in key[-1]):           # a foreign emulator: no serial to trust, hash every time
            return self.state_digest()
        c = getattr(self, "_digest_cache", None)
        if c is None or c[0] != key:
            c = self._digest_cache = (key, self.state_digest())
        return c[1]

    def state_digest(self):
        """sha256 over the emulators' state digests, the experiment block and the prior box: what the ranks of a walker-
        sharded run must hold identically (dist.WalkerSharding.agree_state)."""
        import hashlib
        h = hashlib.sha256()
        for e in self.emuList:
            h.update(e.state_digest() if hasattr(e, "state_digest") else repr(type(e)).encode())
        for a in (self.expdata, self.expdata_cov, self.min, self.max):
            a = np.ascontiguousarray(a, dtype=np.float64)
            h.update(repr(a.shape).encode()); h.update(a.tobytes())
        return h.digest()

    def loadEmulator(self, emulatorPathList, adopt=True):
        """src/mcmc.py:145-150.  A pickle that holds a trained emulator of the REFERENCE (its `src.emulator.Emulator` with
        scikit-learn GPs inside: what EmulatorTraining.ipynb dumps) is taken over onto the device without retraining
        (Emulator.from_reference) unless adopt=False; anything else with the predict protocol stays a foreign emulator
        (host predictions, device likelihood)."""
        import dill
        from .emulator import Emulator
        for path in emulatorPathList:
            with open(path, "rb") as f:
                emu = dill.load(f)
            if adopt and not isinstance(emu, Emulator) and getattr(emu, "gps", None) and hasattr(emu, "scaler"):
                try:
                    emu = Emulator.from_reference(emu, device=self.device)
                    log.info("%s: a trained emulator of the reference, taken over onto device %s", path, self.device)
                except (ValueError, AttributeError, TypeError, KeyError) as e:
                    # an object that is not quite a trained reference emulator (a missing attribute, an array alpha, a GP
                    # fitted with normalize_y): it keeps the predict protocol and stays a foreign emulator, as before
                    log.info("%s: kept as a foreign emulator (%s: %s)", path, type(e).__name__, e)
            self.emuList.append(emu)
        log.info("Number of Emulators: %d", len(self.emuList))

    def set_predict_arithmetic(self, which="fp64"):
        """the same choice for every drop-in emulator of the chain (Emulator.set_predict_arithmetic): "fp64" (default) or "int8" —
        foreign emulators are left alone; an emulator outside the int8 rule keeps the fp64 kernel by itself"""
        from .emulator import Emulator
        for e in self.emuList:
            if isinstance(e, Emulator):
                e.set_predict_arithmetic(which)
        self.__dict__.pop("_digest_cache", None)
        return self

    def random_pos(self, n=1):
        return np.random.uniform(self.min, self.max, (n, self.ndim))

    @staticmethod
    def map(f, args):
        """Lets the object stand in as an emcee `pool` so that the sampler hands the whole
        half-ensemble to the log-probability function in one call (src/mcmc.py:335-342)."""
        return f(args)

    # ------------------------------------------------------------------ model prediction
    def _predict(self, X, extra_std=0.0):
        """Concatenated means and block-diagonal covariance over the emulators
        (src/mcmc.py:153-166)."""
        n = X.shape[0]
        mean = np.zeros((n, self.nobs))
        cov = np.zeros((n, self.nobs, self.nobs))
        es = extra_std * X[:, -1]
        i0 = 0
        for emu in self.emuList:
            m, c = emu.predict(X, return_cov=True, extra_std=es)
            k = m.shape[1]
            mean[:, i0:i0 + k] = m
            cov[:, i0:i0 + k, i0:i0 + k] = c
            i0 += k
        return mean, cov

    # ------------------------------------------------------------------ log-probabilities
    def log_prior(self, X):
        """log(1/volume) inside the open box, -inf outside (src/mcmc.py:169-185)."""
        X = np.array(X, ndmin=2, dtype=np.float64)
        lp = np.log(np.ones(X.shape[0]) / self.prior_volume_)
        lp[~np.all((X > self.min) & (X < self.max), axis=1)] = -np.inf
        return lp

    def _native(self):
        return len(self.emuList) > 0 and all(isinstance(e, Emulator) for e in self.emuList)

    def _prepare_blocks(self):
        """Hand every emulator its slice of the experimental data.  The fused path needs the
        experimental covariance to be block-diagonal over the emulators (it is diagonal in the
        reference, src/mcmc.py:320-322)."""
        # (engines by their serial numbers: the id() of a closed engine can come back with the next one created)
        sig = (id(self.expdata), id(self.expdata_cov), tuple(id(e) for e in self.emuList),
               tuple(getattr(e._engine, "serial", None) for e in self.emuList))
        if sig == self._like_sig:
            return
        i0 = 0
        mask = np.zeros_like(self.expdata_cov, dtype=bool)
        for emu in self.emuList:
            k = emu.nobs
            eng = emu._engine_ready()
            eng.set_likelihood(self.expdata[0, i0:i0 + k], self.expdata_cov[i0:i0 + k, i0:i0 + k])
            mask[i0:i0 + k, i0:i0 + k] = True
            i0 += k
        if i0 != self.nobs:
            raise ValueError("emulators provide %d observables, experiment has %d" % (i0, self.nobs))
        if np.any(self.expdata_cov[~mask] != 0.0):
            raise ValueError("experimental covariance couples different emulators; "
                             "the fused likelihood needs it block-diagonal")
        self._like_sig = (sig[0], sig[1], sig[2], tuple(e._engine.serial for e in self.emuList))

    inside_const = EXTRA_STD_CONST               # what the reference adds to every row inside the box

    def _box(self, device):
        """the prior box resident in HBM, uploaded once per device"""
        import torch
        key = (str(device), self.min.tobytes(), self.max.tobytes())
        if getattr(self, "_box_key", None) != key:
            self._box_dev = (torch.as_tensor(self.min, dtype=torch.float64, device=device),
                             torch.as_tensor(self.max, dtype=torch.float64, device=device))
            self._box_key = key
        return self._box_dev

    use_chain_call = True                        # False: sequence the emulators from Python (A/B tests)

    def _chain_contexts(self):
        """(engines, ctypes array of their contexts) when the C ABI can evaluate the whole chain in one call
        (gpb_chain_logpost / gpb_chain_emcee_run: rows outside the prior box skipped, one emulator after the other on
        one stream), else None.  Call after _prepare_blocks()."""
        from . import _native as nat
        if not (self.use_chain_call and self._native()):
            return None
        engs = [e._engine_ready() for e in self.emuList]
        for g in engs:
            g._need_data()
            g._track_stream()
        arr = (nat.C.c_void_p * len(engs))(*[g.h for g in engs])
        if engs[0].lib.gpb_chain_supported(arr, len(engs)) != 1:
            return None
        return engs, arr

    def log_prob_device(self, X_dev, out=None, outside=-np.inf, lo_dev=None, hi_dev=None):
        """Device-resident log-posterior: X_dev torch.float64 cuda [W,ndim] -> lp [W] (no host
        sync).  Used by the resident sampler; `log_posterior`/`log_likelihood` wrap it."""
        import torch
        from . import _native as nat
        self._prepare_blocks()
        if out is None:
            out = torch.empty(X_dev.shape[0], dtype=torch.float64, device=X_dev.device)
        if lo_dev is None:
            lo_dev, hi_dev = self._box(X_dev.device)
        X_dev = X_dev.contiguous()
        last = len(self.emuList) - 1
        if X_dev.shape[1] != self.ndim:
            raise ValueError("log_prob_device: X has %d columns, the chain has %d parameters" % (X_dev.shape[1], self.ndim))
        cc = self._chain_contexts()
        if cc is not None:
            engs, arr = cc
            e0 = engs[0]
            X_dev = e0._dev(X_dev, (None, self.ndim), "X")
            e0._dev(lo_dev, (self.ndim,), "lo"), e0._dev(hi_dev, (self.ndim,), "hi")
            e0._dev(out, (X_dev.shape[0],), "out")
            e0._ck(e0.lib.gpb_chain_logpost(arr, len(engs), nat.ptr(X_dev), X_dev.shape[0], nat.ptr(out), nat.ptr(lo_dev),
                                            nat.ptr(hi_dev), float(outside), EXTRA_STD_CONST))
            return out
        for i, emu in enumerate(self.emuList):      # all engines enqueue on torch's current stream: ordered
            eng = emu._engine_ready()
            mapped = getattr(emu, "parameterTrafoPCA_", False)
            # parameterTrafoPCA emulators see the PCA-reduced parameters (src/emulator.py:492-551): device pre-pass
            Xg = eng.param_map(X_dev) if mapped else X_dev
            if i < last or mapped:
                eng.loglike(Xg, out=out, accumulate=(i > 0), check=False)
                if i == last:                        # the prior box is over the ORIGINAL parameters
                    eng.box_finish(X_dev, lo_dev, hi_dev, outside, EXTRA_STD_CONST, out)
            else:                                    # last block: likelihood + prior box + constant, one call
                eng.logpost(X_dev, out, i > 0, lo_dev, hi_dev, outside, EXTRA_STD_CONST)
        return out

    def _log_prob(self, X, outside):
        X = np.array(X, ndmin=2, dtype=np.float64)
        if self._native():
            import torch
            dev = torch.device("cuda", self.device)
            # long inputs (a stored chain, src/mcmc.py:729-749) go through in slabs: the K*^T workspace is
            # P x N x rows doubles per emulator; a row's result does not depend on how the batch is cut
            per_row = max(8 * e._ngp * e._X_train.shape[0] for e in self.emuList)
            slab = int(min(max((8 << 30) // per_row, 1024), 1 << 17)) // 128 * 128
            sh = self.sharding if (self.sharding is not None and getattr(self.sharding, "world", 1) > 1) else None
            # sharded: all ranks must hold the same model and be handed the same rows.  ONE all-reduce of the rows' checksum
            # and the state digest's words rides in front of every batch and is read after it — every rank issues it on every
            # call, whatever it knows locally (a rank whose engines were rebuilt by replicate(), a rank that has evaluated
            # before): the collective sequences of the ranks cannot drift apart
            digest = self.state_digest_cached() if sh is not None else None

            def evaluate(Xd):
                if sh is None:
                    return self.log_prob_device(Xd, outside=outside)
                pending = sh.rows_agree_begin(Xd, digest)
                out = torch.empty(Xd.shape[0], dtype=torch.float64, device=Xd.device)
                sh.logprob(lambda Xr, o: self.log_prob_device(Xr, out=o, outside=outside), Xd, out)
                sh.rows_agree_end(pending)
                return out
            if X.shape[0] <= slab:
                Xd = torch.as_tensor(np.ascontiguousarray(X), device=dev)
                return evaluate(Xd).cpu().numpy()
            out = np.empty(X.shape[0])
            for i0 in range(0, X.shape[0], slab):
                Xd = torch.as_tensor(np.ascontiguousarray(X[i0:i0 + slab]), device=dev)
                out[i0:i0 + slab] = evaluate(Xd).cpu().numpy()
            return out
        # generic path: foreign emulators predict on the host, the MVN runs on the device
        lp = np.zeros(X.shape[0])
        inside = np.all((X > self.min) & (X < self.max), axis=1)
        lp[~inside] = outside
        if np.count_nonzero(inside) > 0:
            Xi = X[inside]
            mY, mC = self._predict(Xi, 0.0 * Xi[:, -1])
            eng = _utility_engine(self.device)
            lp[inside] += eng.mvn_loglike(mY - self.expdata, mC + self.expdata_cov)
            lp[inside] += EXTRA_STD_CONST
        return lp

    def log_likelihood(self, X, extra_std_prior_scale=0.001, finite=False):
        """src/mcmc.py:188-222 (pocoMC calls it with finite=True)."""
        return self._log_prob(X, -1e300 if finite else -np.inf)

    def log_posterior(self, X, extra_std_prior_scale=.05):
        """src/mcmc.py:261-299 (the function emcee samples)."""
        return self._log_prob(X, -np.inf)

    def log_likelihood_point_by_point(self, X, extra_std_prior_scale=0.001):
        """Same values as the reference's per-row loop (src/mcmc.py:225-258), in one batch."""
        return self._log_prob(np.asarray(X), -np.inf)

    def compute_log_likelihood_for_chain(self, output_path="./mcmc/log_likelihood.pkl"):
        """src/mcmc.py:729-749."""
        if self.chain is False:
            with open(self.mcmc_path, "rb") as f:
                self.chain = pickle.load(f)["chain"]
        ll = self.log_likelihood_point_by_point(self.chain.reshape(-1, self.ndim))
        ll = ll.reshape(self.chain.shape[0], self.chain.shape[1])
        with open(output_path, "wb") as f:
            pickle.dump({"log_likelihood": ll}, f)

    # ------------------------------------------------------------------ samplers
    def run_mcmc(self, nsteps=500, nburnsteps=None, nwalkers=None, status=None, nthin=10,
                 skip_initial_state_check=False, seed=None):
        """Affine-invariant ensemble sampling with the reference's schedule (src/mcmc.py:345-426):
        resume from an existing chain pickle, else two-stage burn-in with re-seeding at the
        `nwalkers` best unique log-probabilities, then production; thinned chain appended to
        `{'chain': [nwalkers, nsteps/nthin, ndim]}`.  The stretch move runs on the device
        (sampler.StretchSampler) instead of emcee.  After `shard_over(WalkerSharding())` the walkers are sharded over the
        GPUs of the group (collective: every rank calls it with the same arguments)."""
        from .sampler import StretchSampler
        # Several GPUs (chain.shard_over(WalkerSharding()), one process per GPU, every rank makes this call): the sampler runs
        # replicated and walker-sharded — rank 0's start positions, seed and chain file are everybody's (broadcast: numpy's
        # global RandomState and a file system need not agree between processes), every half-step's rows are evaluated in shares
        # with one all-gather (in-stream RCCL when the group's backend is nccl), every rank ends with the same chain, rank 0
        # writes the pickle.  The samples are those of the single-GPU run with the same seed and start, bit for bit.
        sh = self.sharding if (self.sharding is not None and getattr(self.sharding, "world", 1) > 1) else None
        root = sh is None or sh.rank == 0
        share = (lambda obj: obj) if sh is None else sh.broadcast_object
        chain_data = {}
        failed = None
        if root:
            try:
                with open(self.mcmc_path, "rb") as f:
                    chain_data = pickle.load(f)
            except FileNotFoundError:
                pass
            except Exception as e:          # an unreadable chain file: the other ranks must not wait for rank 0 in a collective
                if sh is None:
                    raise
                failed = "%s: %s" % (type(e).__name__, e)
        failed = share(failed)
        if failed:
            raise RuntimeError("run_mcmc: rank 0 could not read %s (%s)" % (self.mcmc_path, failed))
        burn = share("chain" not in chain_data)
        if nburnsteps is None or nwalkers is None:
            log.error("must specify nburnsteps and nwalkers to start chain")
            return
        if sh is not None:
            if seed is None:
                seed = int(np.random.SeedSequence().generate_state(1, dtype=np.uint64)[0])
            seed = share(int(seed))
            if sh.direct is None and sh.backend() == "nccl" and self._native():
                sh.try_direct(self.emuList[0]._engine_ready())        # collective; torch.distributed's all-gather stays otherwise
        sampler = StretchSampler(self, nwalkers, seed=seed, sharding=sh)
        if burn:
            nburn0 = nburnsteps // 2
            sampler.run(share(self.random_pos(nwalkers) if root else None), nburn0, status=status)
            flat_lp = sampler.lnprobability.reshape(-1)
            flat_x = sampler.chain.reshape(-1, self.ndim)
            X0 = flat_x[np.unique(flat_lp, return_index=True)[1][-nwalkers:]]
            sampler.reset()
            X0 = sampler.run(X0, nburnsteps - nburn0, status=status)
            sampler.reset()
        else:
            X0 = share(chain_data["chain"][:, -1, :] if root else None)
        sampler.run(X0, nsteps, status=status)
        thinned = sampler.chain[:, ::nthin, :]
        if "chain" in chain_data:
            chain_data["chain"] = np.concatenate((chain_data["chain"], thinned), axis=1)
        else:
            chain_data["chain"] = thinned
        self.chain = chain_data["chain"] if root else thinned
        self.acceptance_fraction = sampler.acceptance_fraction
        if root:
            try:
                with open(self.mcmc_path, "wb") as f:
                    pickle.dump(chain_data, f)
            except Exception as e:
                if sh is None:
                    raise
                failed = "%s: %s" % (type(e).__name__, e)
        if sh is not None:
            self.chain = share(self.chain if root else None)          # (also: nobody returns before the file is written)
            failed = share(failed)
            if failed:
                raise RuntimeError("run_mcmc: rank 0 could not write %s (%s)" % (self.mcmc_path, failed))

    def run_pocoMC(self, n_effective=1000, n_active=250, n_prior=2000, sample="tpcn", n_max_steps=200,
                   random_state=42, n_total=5000, n_evidence=5000, pool=None, prior=None):
        """pocoMC driver with the reference's call shape and output schema (src/mcmc.py:752-819);
        the likelihood callback is this class's device-backed log_likelihood(finite=True)."""
        import pocomc
        from scipy.stats import uniform
        if pool is not None:
            # the reference hands `pool=12` to pocoMC, whose workers are FORKED copies of this Chain
            # (src/mcmc.py:775-776, 798-804; examples/RunBayesianAnalysis.ipynb:85).  Here the whole 8192-row batch is
            # one device call (vectorize=True); forked workers could not use the parent's GPU context anyway.
            log.info("run_pocoMC: pool=%r ignored — likelihood batches are already vectorised on the device", pool)
            pool = None
        if prior is None:
            prior = pocomc.Prior([uniform(self.min[i], self.max[i] - self.min[i]) for i in range(self.ndim)])
        elif self.ndim != prior.dim:
            raise ValueError("prior.dim does not match the model parameter space")
        sampler = pocomc.Sampler(prior=prior, likelihood=self.log_likelihood, likelihood_kwargs={"finite": True},
                                 n_effective=n_effective, n_active=n_active, n_prior=n_prior, sample=sample,
                                 n_max_steps=n_max_steps, random_state=random_state, vectorize=True, pool=pool)
        sampler.run(n_total=n_total, n_evidence=n_evidence)
        samples, weights, logl, logp = sampler.posterior()
        logz, logz_err = sampler.evidence()
        with open(self.mcmc_path, "wb") as f:
            pickle.dump({"chain": samples, "weights": weights, "logl": logl, "logp": logp, "logz": logz,
                         "logz_err": logz_err}, f)
