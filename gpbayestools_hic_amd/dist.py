"""
Walker sharding across the GPUs of one node: one process per GPU, every rank holds
the full GP state (replicated, a few hundred MB) and the full walker array; rank r evaluates
rows [r*chunk, (r+1)*chunk) of each log-probability batch and ONE all-gather of `chunk` float64
per rank puts the complete log-probability vector on every rank.  There is no other exchange:
proposals and accept draws are regenerated identically everywhere (sampler.py).

The collective is RCCL over xGMI: either torch.distributed's (backend "nccl"; "gloo" on CPU for tests) or,
after WalkerSharding.try_direct, the C ABI's own communicator (gpb_dist_allgather: ncclAllGather enqueued on
the very stream the kernels run on — measured 8 us less per collective than the hop through torch's
communicator stream).  No host synchronisation either way.  The payload is 2-16 KB: latency-bound.
"""
import os


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).
    Returns (rank, world, local_rank).  Single-process when WORLD_SIZE is absent or 1."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:     # GPB_DIST_BACKEND=gloo rehearses the N>1 path with several ranks on ONE GPU
            backend = os.environ.get("GPB_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if backend == "nccl":
            # one process per GPU: RCCL refuses two ranks on one device, and a rank folded onto another rank's GPU would time a
            # different job from the one asked for.  Refuse on every rank, before the rendezvous, with the numbers.
            nlocal = int(os.environ.get("LOCAL_WORLD_SIZE", world))
            if nlocal > ndev or local >= ndev:
                raise RuntimeError("init_from_env: %d ranks on this node but %d visible GPU(s) (LOCAL_RANK %d): the nccl backend "
                                   "needs one GPU per rank.  Start at most %d ranks, or rehearse the multi-rank control flow on "
                                   "one GPU with GPB_DIST_BACKEND=gloo" % (nlocal, ndev, local, ndev))
        elif ndev:
            local = local % ndev                # gloo rehearsal only: the ranks share the visible GPU(s)
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


class WalkerSharding:
    def __init__(self, rank=None, world=None, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self._buf = {}
        self.direct = None

    def enable_direct(self, engine):
        """Route the all-gather through the C ABI's own RCCL communicator (gpb_dist_*: ncclAllGather enqueued
        directly on the kernels' stream, no second stream and no event hops).  Collective: rank 0 creates the
        ncclUniqueId, torch.distributed carries it to the others.  ncclCommInitRank is itself collective — a rank
        that never reaches it leaves the others blocked inside it — so the ranks first VOTE, through
        torch.distributed, that every one of them has loaded librccl and holds the id; only a unanimous vote
        enters gpb_dist_init.  Raises (on every rank alike) when the vote fails.  bench.py goes through
        try_direct (below)."""
        uid, why = None, None
        if self.rank == 0:
            try:
                uid = engine.dist_uid()
            except Exception as e:      # still take part in the broadcast: every rank must see the same outcome
                uid, why = None, "rank 0 could not create a ncclUniqueId: %s" % e
        box = [uid]
        self.dist.broadcast_object_list(box, src=0, group=self.group)
        ready = box[0] is not None and len(bytes(box[0])) == 128
        if ready:
            try:
                ready = bool(engine.dist_available())
                if not ready:
                    why = "librccl could not be loaded on rank %d" % self.rank
            except Exception as e:
                ready, why = False, "%s: %s" % (type(e).__name__, e)
        if not self._all_ok(ready):     # nobody has entered ncclCommInitRank yet: all ranks leave together
            raise RuntimeError(why or "the direct RCCL path is not available on every rank (no communicator was created)")
        engine.dist_init(self.rank, self.world, box[0])
        self.direct = engine
        return self

    def _coll_device(self):
        """where torch.distributed's own collectives of this group want their tensors (gloo: host, nccl: this GPU)"""
        import torch
        if self.dist.get_backend(self.group) == "gloo" or not torch.cuda.is_available():
            return torch.device("cpu")
        return torch.device("cuda", torch.cuda.current_device())

    def backend(self):
        return self.dist.get_backend(self.group)

    def broadcast_object(self, obj, src=0):
        """rank `src`'s picklable object on every rank (collective)"""
        box = [obj if self.rank == src else None]
        self.dist.broadcast_object_list(box, src=src, group=self.group)
        return box[0]

    def _all_ok(self, ok):
        import torch
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self._coll_device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return int(t.item()) == 1

    # ---- replicated state: every rank must hold THE SAME emulators and experiment -------------------------------------
    # Walker sharding rests on it: rank r's rows are evaluated with rank r's GP state and accepted on every rank.  The
    # device side is deterministic (fixed-order reductions), the host side of a training is not across processes: the
    # scaler / PCA SVD of numpy rounds differently with a different number of BLAS threads (torch.distributed.run sets
    # OMP_NUM_THREADS=1 for its ranks only), and the GP targets with it.  So the ranks do not each trust their own fit:
    # `replicate` hands rank `src`'s fitted host state to everybody, `agree_state` proves the replicas equal before a run.
    def replicate(self, chain, src=0):
        """Rank `src` broadcasts the host state of the chain's emulators (what their pickles hold: scaler, PCA, design,
        targets, theta*, transforms) and the chain's experiment block and prior box; the other ranks install it and
        rebuild their device state from it before returning (set_data, set_theta, factor: milliseconds).  Each rank keeps its
        own device and its own fit-side sharding (Emulator.fit_sharding).  Collective."""
        box = [None]
        if self.rank == src:
            box[0] = ([e.__getstate__() for e in chain.emuList], chain.expdata, chain.expdata_cov, chain.min, chain.max)
        self.dist.broadcast_object_list(box, src=src, group=self.group)
        if self.rank != src:
            states, chain.expdata, chain.expdata_cov, chain.min, chain.max = box[0]
            if len(states) != len(chain.emuList):
                raise RuntimeError("replicate: rank %d holds %d emulators, rank %d %d" % (src, len(states), self.rank, len(chain.emuList)))
            for emu, st in zip(chain.emuList, states):
                eng = getattr(emu, "_engine", None)
                if eng is not None:
                    eng.close()
                st = dict(st, device=emu.device, fit_sharding=getattr(emu, "fit_sharding", None))
                emu.__setstate__(st)                       # (each rank keeps its own GPU and its own fit-side sharding)
            chain.prior_volume_ = float(__import__("numpy").prod(chain.max - chain.min))
            chain._like_sig = None
            chain.__dict__.pop("_digest_cache", None)
            for emu in chain.emuList:                      # the device state, now: set_data, set_theta, factor, transform
                if getattr(emu, "_trained", False) and hasattr(emu, "_engine_ready"):
                    emu._engine_ready()
        return chain

    def agree_state(self, digest):
        """All ranks hold the same 32-byte state digest (Chain.state_digest): MIN and MAX over the ranks of its four words
        coincide.  Raises on EVERY rank when they do not.  Collective."""
        import numpy as np
        import torch
        w = np.frombuffer(bytes(digest)[:32].ljust(32, b"\0"), dtype=np.int64).copy()
        lo = torch.as_tensor(w, device=self._coll_device())
        hi = lo.clone()
        self.dist.all_reduce(lo, op=self.dist.ReduceOp.MIN, group=self.group)
        self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX, group=self.group)
        if not torch.equal(lo, hi):
            raise RuntimeError("the ranks' replicas of the GP state differ (hyper-parameters, targets, transforms or the "
                               "experiment block): a sharded run would mix log-probabilities of different models.  Train on "
                               "one rank and call WalkerSharding.replicate(chain) before sampling.")
        return True

    def rows_agree_begin(self, X_dev, digest=None):
        """start the check that every rank was handed the same batch — and, with `digest` (Chain.state_digest: 32 bytes), holds
        the same model: a position-weighted 64-bit checksum of the rows' bits and the digest's four words, ONE max-all-reduce of
        (v, -v) over all of them enqueued behind it (asynchronous with nccl); rows_agree_end reads the outcome"""
        import numpy as np
        import torch
        v = X_dev.contiguous().view(torch.int64).reshape(-1)
        key = (v.numel(), v.device)
        if getattr(self, "_cs_key", None) != key:
            self._cs_w = torch.arange(v.numel(), dtype=torch.int64, device=v.device) * 2 + 1
            self._cs_key = key
        c = (((v * self._cs_w).sum() + v.numel()) >> 1).reshape(1)      # (>> 1: -c never overflows)
        if digest is not None:
            dk = (bytes(digest), self._coll_device())
            if getattr(self, "_dg_key", None) != dk:         # the digest's words on the collective's device: uploaded once
                w = np.frombuffer(dk[0][:32].ljust(32, b"\0"), dtype=np.int64) >> 1       # (>> 1: -w never overflows)
                self._dg_w, self._dg_key = torch.as_tensor(w.copy(), device=dk[1]), dk
            c = torch.cat([c.to(self._coll_device()), self._dg_w])
        t = torch.cat([c, -c]).to(self._coll_device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return t

    def rows_agree_end(self, t):
        """raises — on every rank — unless max(v) == min(v) over the ranks for every word (synchronises on the words)"""
        w = [int(x) for x in t.cpu()]
        n = len(w) // 2
        if w[0] != -w[n]:
            raise RuntimeError("the ranks of a sharded log-probability call were handed different rows: the sampler above it "
                               "must run replicated (same seed, same inputs on every rank)")
        if any(w[i] != -w[n + i] for i in range(1, n)):
            raise RuntimeError("the ranks' replicas of the GP state differ (hyper-parameters, targets, transforms or the "
                               "experiment block): a sharded call would mix log-probabilities of different models.  Train on "
                               "one rank and call WalkerSharding.replicate(chain) before evaluating.")

    def _drop_direct(self):
        if self.direct is not None:
            try:
                self.direct.dist_finalize()
            except Exception:
                pass
        self.direct = None

    def try_direct(self, engine):
        """enable_direct + a self-check against torch.distributed's all-gather on a test vector; every rank
        keeps the direct path only if it worked on ALL ranks.  Returns None when direct is on, else the reason
        (the exchange then stays on torch.distributed — both are RCCL, nothing leaves the GPUs).  The ranks agree
        after each stage, so that they always issue the same sequence of collectives."""
        import torch
        why = None
        try:
            self.enable_direct(engine)
        except Exception as e:          # missing librccl, ncclCommInitRank refused, ...
            why = "%s: %s" % (type(e).__name__, e)
        if not self._all_ok(why is None):
            self._drop_direct()
            return why or "direct path could not be set up on another rank"
        n, dev = 64, self._coll_device()
        mine = torch.arange(n, dtype=torch.float64, device=dev) + 1000.0 * self.rank
        ref = torch.empty(n * self.world, dtype=torch.float64, device=dev)
        self.dist.all_gather_into_tensor(ref, mine, group=self.group)
        try:
            got = torch.full_like(ref, -1.0)
            got[self.rank * n:(self.rank + 1) * n] = mine
            engine.dist_allgather(got[self.rank * n:(self.rank + 1) * n], got)
            if dev.type == "cuda":
                torch.cuda.synchronize()
            if not torch.equal(ref, got):
                why = "direct all-gather disagrees with torch.distributed"
        except Exception as e:
            why = "%s: %s" % (type(e).__name__, e)
        if not self._all_ok(why is None):
            self._drop_direct()
            return why or "direct all-gather failed its self-check on another rank"
        return None

    def time_allgather(self, count, reps=200, warm=20):
        """Microseconds per all-gather of `count` float64 per rank, back to back on the kernels' stream (HIP events; the MAX over
        the ranks) — the wire latency the sharded step pays twice: in place, through the path a sharded batch takes (the C ABI's
        in-stream ncclAllGather when `direct` is set, else torch.distributed's).  The collectives are enqueued BEHIND a few
        milliseconds of unrelated device work, so that they are all in the queue when the first one starts: the interval between the
        two events is device time, not the host's enqueue rate (as in the step loop, whose kernels the C ABI enqueues far ahead).
        Collective.  None under gloo (no CUDA all-gather: the rehearsal stages through the host and its time says nothing about a
        wire)."""
        import torch
        from . import _native as nat
        if self.backend() == "gloo":
            return None
        dev = torch.device("cuda", torch.cuda.current_device())
        buf = torch.zeros(count * self.world, dtype=torch.float64, device=dev)
        mine = buf[self.rank * count:(self.rank + 1) * count]
        if self.direct is not None:
            eng = self.direct
            eng._track_stream()
            eng._dev(mine, (count,), "send"); eng._dev(buf, (count * self.world,), "recv")     # validated once, then the bare call
            sp, rp, h, fn = nat.VP(mine.data_ptr()), nat.VP(buf.data_ptr()), eng.h, eng.lib.gpb_dist_allgather

            def one():
                eng._ck(fn(h, sp, rp, count))
        else:
            def one():
                self.dist.all_gather_into_tensor(buf, mine, group=self.group)
        for _ in range(warm):
            one()
        plug = torch.empty((4096, 4096), dtype=torch.float32, device=dev).normal_()
        torch.cuda.synchronize()
        self.dist.barrier(group=self.group)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(8):                             # ~ a few ms of device work in front: the host gets ahead of the device
            plug = torch.mm(plug, plug).clamp_(-1.0, 1.0)
        e0.record()
        for _ in range(reps):
            one()
        e1.record()
        torch.cuda.synchronize()
        t = torch.tensor([e0.elapsed_time(e1) / reps * 1e3], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def rows(self, W):
        """(r0, r1, chunk): this rank's row range of a W-row batch; chunk = ceil(W / world)."""
        chunk = -(-W // self.world)
        r0 = min(self.rank * chunk, W)
        return r0, min(r0 + chunk, W), chunk

    def logprob(self, fn, X, out):
        """out[:] = fn over all rows of X, computed in shards.  fn(X_rows, out_rows) -> out_rows
        must fill its output for exactly the rows it is given (torch tensors, any device)."""
        import torch
        W = X.shape[0]
        r0, r1, chunk = self.rows(W)
        if self.world > 1 and X.is_cuda and self.dist.get_backend(self.group) == "gloo":
            # rehearsal only (several ranks sharing one GPU, gloo has no CUDA all-gather): stage through PINNED host buffers with
            # stream-ordered copies and one stream synchronisation — the blocking pageable copies this branch used before
            # (local.cpu(), out.copy_(host tensor)) are where two of four ranks once stalled for good while six processes held
            # the GPU (profiles/r04_ranks_stall.txt)
            key = ("gloo", chunk, X.device, out.dtype)
            if key not in self._buf:
                self._buf[key] = (torch.zeros(chunk, dtype=out.dtype, device=X.device),
                                  torch.empty(chunk, dtype=out.dtype).pin_memory(),
                                  torch.empty(chunk * self.world, dtype=out.dtype).pin_memory())
            local, mine_h, gathered = self._buf[key]
            if r1 < r0 + chunk:
                local.zero_()
            if r1 > r0:
                fn(X[r0:r1], local[:r1 - r0])
            mine_h.copy_(local, non_blocking=True)
            torch.cuda.current_stream(X.device).synchronize()      # (also: the previous call's copy out of `gathered` is done)
            self.dist.all_gather_into_tensor(gathered, mine_h, group=self.group)
            out.copy_(gathered[:W], non_blocking=True)
            return out
        if (self.world > 1 or self.direct is not None) and W == chunk * self.world and out.is_contiguous():
            # even split: every rank writes its slice of `out` and the all-gather runs in place
            # (send buffer = receive buffer + rank*chunk) — no staging copies on the step's critical path
            mine = out[r0:r1]
            fn(X[r0:r1], mine)
            if self.direct is not None:
                self.direct.dist_allgather(mine, out)
            else:
                self.dist.all_gather_into_tensor(out, mine, group=self.group)
            return out
        key = (chunk, X.device, out.dtype)
        if key not in self._buf:
            self._buf[key] = (torch.zeros(chunk, dtype=out.dtype, device=X.device),
                              torch.empty(chunk * self.world, dtype=out.dtype, device=X.device))
        local, gathered = self._buf[key]
        if r1 > r0:
            fn(X[r0:r1], local[:r1 - r0])
        if self.world == 1:
            out.copy_(local[:W])
            return out
        if self.direct is not None:
            self.direct.dist_allgather(local, gathered)
        else:
            self.dist.all_gather_into_tensor(gathered, local, group=self.group)
        out.copy_(gathered[:W])
        return out


class GPSharding:
    """Fit-side sharding: the npc GPs of an emulator are independent until prediction, so their
    hyper-parameter searches — the wall-time of trainEmulator, src/emulator.py:309-315 — are dealt round-robin
    to the ranks (GP p -> rank p % world).  One object all-gather of (indices, theta, LML) afterwards; every rank
    then factorises all GPs at the gathered theta (3.7 ms at cfg 4, cheaper than broadcasting L and L^-1).
    Collective: every rank of the group must call Emulator.trainEmulator."""

    def __init__(self, rank=None, world=None, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world

    def mine(self, P):
        import numpy as np
        return np.arange(self.rank, P, self.world)

    def gather(self, P, idx, theta, val):
        """theta[P,k], val[P] on every rank from each rank's (idx, theta[len(idx),k], val[len(idx)])."""
        import numpy as np
        parts = [None] * self.world
        self.dist.all_gather_object(parts, (np.asarray(idx), np.asarray(theta), np.asarray(val)), group=self.group)
        k = next(t.shape[1] for _, t, _ in parts if t.ndim == 2 and t.shape[0])
        theta_all, val_all = np.empty((P, k)), np.empty(P)
        seen = np.zeros(P, dtype=bool)
        for i, t, v in parts:
            theta_all[i], val_all[i] = t.reshape(-1, k), v
            seen[i] = True
        if not seen.all():
            raise RuntimeError("GPSharding.gather: GPs %s were not searched by any rank" % np.flatnonzero(~seen))
        return theta_all, val_all
