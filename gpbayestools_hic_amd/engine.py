"""
GPEngine — thin Python handle on one gpb_ctx (include/gpbayes.h): the P independent GPs of one
emulator resident in HBM, plus its observable transform and likelihood block.
Host code only moves arguments; every number is produced by the HIP kernels.
"""
import ctypes as C
import itertools
import os

import numpy as np

from . import _native as nat

KERNEL_IDS = {"RBF": 0, "Matern": 1, "Matern15": 1, "Matern25": 2}
MODE_PCA, MODE_NO_PCA, MODE_EXPDIAG, MODE_NO_PCA_EXPDIAG = 0, 1, 2, 3


class NotPositiveDefinite(np.linalg.LinAlgError):
    pass


def _is_torch(x):
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda")


_SERIAL = itertools.count(1)


class GPEngine:
    def __init__(self, device=0, stream="torch", debug=None):
        """stream: "torch" = enqueue on torch's current stream of `device` (one ordered queue
        shared with torch copies and collectives), None = private stream, or a hipStream_t.
        debug: True binds libgpbayes_debug.so (test hooks, kernel variants); None = the process default (the product library
        unless GPB_DEBUG_LIB=1 or inside _native.debug_library())."""
        self.lib = nat.load(debug)
        if self.lib.gpb_device_count() <= 0:
            raise nat.GPBError("no HIP device visible: the gfx950 kernels cannot run (no CPU fallback)")
        h = nat.VP()
        rc = self.lib.gpb_ctx_create(int(device), None, C.byref(h))
        if rc != 0:
            raise nat.GPBError(f"gpb_ctx_create failed ({rc})")
        self._pid = os.getpid()          # device state does not survive a fork: see _check_pid
        self._h = h
        self.serial = next(_SERIAL)      # which context is this? (never reused within the process, unlike id() of a freed object)
        self.device = int(device)
        self._follow_torch = stream == "torch"
        self._stream = None
        if self._follow_torch:
            self._track_stream()
        elif stream is not None:
            self._ck(self.lib.gpb_ctx_set_stream(self.h, nat.VP(int(stream))))
        self.N = self.d = self.P = self.M = 0
        # GPB_PREDICT_SLICED=1 (2: the rule off): every engine of the process evaluates V = L^-1 K*^T on the int8 matrix pipe where
        # its accuracy rule admits the GPs (option key 51, csrc/gpb_sliced.hip) — the whole test suite runs under it this way
        if os.environ.get("GPB_PREDICT_SLICED", "0") not in ("", "0"):
            self.tune("predict_sliced", int(os.environ["GPB_PREDICT_SLICED"]))

    # ------------------------------------------------------------------ plumbing
    def _ck(self, rc):
        if rc < 0:
            raise nat.GPBError(f"{self.lib.gpb_last_error(self.h).decode()} (code {rc})")
        return rc

    def _check_pid(self):
        """A process forked AFTER this engine was created inherits the handle but not a usable HIP context (the
        runtime's state does not survive fork()): the reference's pocoMC `pool=int` workers are such processes
        (src/mcmc.py:775-776, 798-804).  Raise before any HIP call is issued on the dead context."""
        if self._pid != os.getpid():
            raise RuntimeError(
                "GPEngine was created in process %d, this is process %d: the GPU context does not survive a fork. "
                "Log-probability batches are already vectorised on the device — use pool=None, or start workers with "
                "the 'spawn' method before the parent touches the GPU" % (self._pid, os.getpid()))

    @property
    def h(self):
        """the gpb_ctx handle every C-ABI call takes: handing it out is where the fork guard sits"""
        self._check_pid()
        return self._h

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_pid", None) == os.getpid():       # never issue HIP calls on a context inherited by fork
                self.lib.gpb_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._check_pid()
        self._ck(self.lib.gpb_sync(self.h))

    def _need_data(self):
        self._check_pid()
        if self.N == 0:
            raise nat.GPBError("no GP data: call set_data / set_theta / factor first")

    def _track_stream(self):
        """stream="torch": the kernels must run on the stream torch allocates and frees the caller's tensors on.
        Called at the top of every method that takes torch tensors: if torch's current stream of this device has
        changed since the last call (`with torch.cuda.stream(s):`, a sampler on a side stream) the context is
        re-targeted onto it (gpb_ctx_set_stream drains the old stream first, so work already enqueued is ordered
        before anything that follows)."""
        self._check_pid()
        if not self._follow_torch:
            return
        import torch
        s = torch.cuda.current_stream(self.device).cuda_stream
        if s != self._stream:
            self._ck(self.lib.gpb_ctx_set_stream(self.h, nat.VP(s)))
            self._stream = s

    def _dev(self, x, shape, name, dtype="torch.float64", contiguous=True):
        """Validate a tensor whose raw pointer goes to the C ABI as DEVICE memory: a wrong device, dtype, shape
        or stride would be read by the kernels as an address (a GPU fault, not an exception).  `shape` entries
        of None are free.  Returns the tensor (made contiguous when `contiguous` is False = input may be a view)."""
        if not _is_torch(x):
            raise ValueError("%s: expected a torch tensor, got %s" % (name, type(x).__name__))
        if not x.is_cuda or x.device.index != self.device:
            raise ValueError("%s: tensor lives on %s, the engine on cuda:%d" % (name, x.device, self.device))
        if str(x.dtype) != dtype:
            raise ValueError("%s: expected %s, got %s" % (name, dtype, x.dtype))
        if x.dim() != len(shape) or any(s is not None and int(s) != int(n) for s, n in zip(shape, x.shape)):
            raise ValueError("%s: expected shape %s, got %s" % (name, tuple(shape), tuple(x.shape)))
        if not x.is_contiguous():
            if contiguous:
                raise ValueError("%s: must be contiguous" % name)
            x = x.contiguous()
        return x

    def _check_cols(self, X_dev, name="X"):
        """device inputs are used as they are: the kernels index them with the engine's d"""
        self._track_stream()
        return self._dev(X_dev, (None, self.d), name, contiguous=False)

    def _extra_std_dev(self, extra_std, W):
        """extra_std of the torch path as a device vector [W] (None = 0)"""
        if extra_std is None:
            return None
        import torch
        if _is_torch(extra_std):
            if extra_std.dim() == 0:
                extra_std = extra_std.reshape(1)
            if extra_std.numel() == 1 and W != 1:
                extra_std = extra_std.expand(W)
            return self._dev(extra_std, (W,), "extra_std", contiguous=False)
        es = np.array(np.broadcast_to(np.asarray(extra_std, dtype=np.float64).reshape(-1), (W,)))     # writable copy
        return torch.as_tensor(es, device=torch.device("cuda", self.device))

    # ------------------------------------------------------------------ GP state
    def set_data(self, X, Z, kernel="RBF", alpha=0.1):
        """X[N,d] design, Z[P,N] targets (one row per GP)."""
        self._check_pid()
        X, Z = nat.f64(X), nat.f64(Z)
        self.N, self.d = X.shape
        self.P = Z.shape[0]
        assert Z.shape[1] == self.N
        kid = KERNEL_IDS[kernel] if isinstance(kernel, str) else int(kernel)
        self.M, self.pmap_d_in, self.pmap_d_out = 0, -1, -1       # gpb_gp_set drops the transform, likelihood and map
        self._ck(self.lib.gpb_gp_set(self.h, self.N, self.d, self.P, nat.ptr(X), nat.ptr(Z), kid, float(alpha)))

    def set_data_multi(self, Xs, Zs, kernel="RBF", alpha=0.1):
        """P GPs, each over its own design: Xs[p] [N_p, d], Zs[p] [N_p] (gpb_gp_set_multi: the GPs of several emulators or
        the restarts of a search side by side; the designs must pad to the same multiple of 64 points).  Fit-only."""
        self._check_pid()
        import ctypes
        Xs = [nat.f64(x) for x in Xs]
        Zs = [nat.f64(z).reshape(-1) for z in Zs]
        self.P = len(Xs)
        assert self.P >= 1 and len(Zs) == self.P
        self.d = Xs[0].shape[1]
        for x, z in zip(Xs, Zs):
            assert x.ndim == 2 and x.shape[1] == self.d and z.shape[0] == x.shape[0]
        Ns = np.array([x.shape[0] for x in Xs], dtype=np.int64)
        self.N = int(Ns.max())
        kid = KERNEL_IDS[kernel] if isinstance(kernel, str) else int(kernel)
        self.M, self.pmap_d_in, self.pmap_d_out = 0, -1, -1
        xp = (ctypes.c_void_p * self.P)(*[x.ctypes.data for x in Xs])
        zp = (ctypes.c_void_p * self.P)(*[z.ctypes.data for z in Zs])
        self._ck(self.lib.gpb_gp_set_multi(self.h, self.P, self.d, nat.ptr(Ns), xp, zp, kid, float(alpha)))

    def lml_subset(self, idx, theta, eval_gradient=True):
        """LML (and gradient) of the stored GPs `idx` at theta[len(idx), d+2] in one launch sequence (gpb_gp_lml_subset);
        the context is left without a factorisation."""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        n = idx.shape[0]
        theta = nat.f64(theta).reshape(n, self.d + 2)
        val = np.empty(n)
        grad = np.empty((n, self.d + 2)) if eval_gradient else None
        info = np.zeros(n, dtype=np.int32)
        self._ck(self.lib.gpb_gp_lml_subset(self.h, n, nat.ptr(idx), nat.ptr(theta), nat.ptr(val), nat.ptr(grad), nat.ptr(info)))
        return (val, grad) if eval_gradient else val

    lml_active = lml_subset          # the name the lock-step search driver looks for (emulator._batched_lbfgsb)

    def set_theta(self, theta):
        theta = nat.f64(theta).reshape(self.P, self.d + 2)
        self._ck(self.lib.gpb_gp_set_theta(self.h, nat.ptr(theta)))
        self.theta = theta.copy()

    def factor(self, raise_on_fail=True):
        info = np.zeros(self.P, dtype=np.int32)
        rc = self._ck(self.lib.gpb_gp_factor(self.h, nat.ptr(info)))
        if rc > 0 and raise_on_fail:
            raise NotPositiveDefinite(
                f"kernel matrix of GP {int(np.flatnonzero(info)[0])} is not positive definite "
                f"(leading minor {rc}); try increasing alpha")
        return info

    def get(self, what, W=None):
        """K / L / Linv [P,N,N], alpha [P,N], "Kstar" [P,W,N] (K(X*, X) of the most recent batch of W rows), "form" [P]
        (0 = Gram form, 1 = difference form of the distances: chosen per GP from theta, gpbayes.h GPB_GET_FORM)."""
        sel = {"K": 0, "L": 1, "Linv": 2, "alpha": 3, "Kstar": 4, "form": 5}[what]
        if what == "Kstar":
            shape = (self.P, int(W), self.N)
        elif what == "form":
            shape = (self.P,)
        else:
            shape = (self.P, self.N) if what == "alpha" else (self.P, self.N, self.N)
        out = np.empty(shape)
        self._ck(self.lib.gpb_gp_get(self.h, sel, nat.ptr(out)))
        return out

    def lml(self, theta, eval_gradient=True):
        theta = nat.f64(theta).reshape(self.P, self.d + 2)
        val = np.empty(self.P)
        grad = np.empty((self.P, self.d + 2)) if eval_gradient else None
        info = np.zeros(self.P, dtype=np.int32)
        self._ck(self.lib.gpb_gp_lml(self.h, nat.ptr(theta), nat.ptr(val), nat.ptr(grad), nat.ptr(info)))
        self.theta = theta.copy()
        return (val, grad) if eval_gradient else val

    def predict(self, Xs, return_var=True):
        """per-GP mean[W,P] (and var[W,P]); numpy in -> numpy out, torch(cuda) in -> torch out."""
        self._need_data()
        if _is_torch(Xs):
            import torch
            Xs = self._check_cols(Xs, "Xs")
            W = Xs.shape[0]
            mean = torch.empty((W, self.P), dtype=torch.float64, device=Xs.device)
            var = torch.empty((W, self.P), dtype=torch.float64, device=Xs.device) if return_var else None
            self._ck(self.lib.gpb_gp_predict(self.h, nat.ptr(Xs), W, 1, nat.ptr(mean), nat.ptr(var)))
            return (mean, var) if return_var else mean
        Xs = nat.f64(Xs).reshape(-1, self.d)
        W = Xs.shape[0]
        mean = np.empty((W, self.P))
        var = np.empty((W, self.P)) if return_var else None
        self._ck(self.lib.gpb_gp_predict(self.h, nat.ptr(Xs), W, 0, nat.ptr(mean), nat.ptr(var)))
        return (mean, var) if return_var else mean

    def predict_cov(self, Xs):
        """per-GP mean[W,P] and full covariance cov[P,W,W] between the query points (numpy in/out)."""
        self._need_data()
        Xs = nat.f64(Xs).reshape(-1, self.d)
        W = Xs.shape[0]
        mean = np.empty((W, self.P))
        cov = nat.host_empty((self.P, W, W))
        self._ck(self.lib.gpb_gp_predict_cov(self.h, nat.ptr(Xs), W, 0, nat.ptr(mean), nat.ptr(cov)))
        return mean, cov

    # ------------------------------------------------------------------ emulator transform
    def set_transform(self, mode, mu, A=None, cov_trunc=None, scale=None):
        mu = nat.f64(mu)
        self.M = mu.shape[0]
        A = None if A is None else nat.f64(A)
        cov_trunc = None if cov_trunc is None else nat.f64(cov_trunc)
        scale = None if scale is None else nat.f64(scale)
        self._ck(self.lib.gpb_emu_set_transform(self.h, int(mode), self.M, nat.ptr(A), nat.ptr(mu),
                                                nat.ptr(cov_trunc), nat.ptr(scale)))

    def emu_predict(self, Xs, return_cov=True, extra_std=None):
        self._need_data()
        if _is_torch(Xs):
            import torch
            Xs = self._check_cols(Xs, "Xs")
            W = Xs.shape[0]
            es = self._extra_std_dev(extra_std, W)       # numbers / numpy are uploaded, tensors are validated
            mean = torch.empty((W, self.M), dtype=torch.float64, device=Xs.device)
            cov = torch.empty((W, self.M, self.M), dtype=torch.float64, device=Xs.device) if return_cov else None
            self._ck(self.lib.gpb_emu_predict(self.h, nat.ptr(Xs), W, 1, nat.ptr(es), nat.ptr(mean), nat.ptr(cov)))
            return (mean, cov) if return_cov else mean
        Xs = nat.f64(Xs).reshape(-1, self.d)
        W = Xs.shape[0]
        es = None if extra_std is None else nat.f64(np.broadcast_to(np.asarray(extra_std, float).reshape(-1), (W,)))
        mean = nat.host_empty((W, self.M))
        cov = nat.host_empty((W, self.M, self.M)) if return_cov else None
        self._ck(self.lib.gpb_emu_predict(self.h, nat.ptr(Xs), W, 0, nat.ptr(es), nat.ptr(mean), nat.ptr(cov)))
        return (mean, cov) if return_cov else mean

    # ------------------------------------------------------------------ likelihood block
    def set_likelihood(self, yexp, cov_exp):
        yexp, cov_exp = nat.f64(yexp).reshape(-1), nat.f64(cov_exp)
        assert yexp.shape[0] == self.M and cov_exp.shape == (self.M, self.M)
        self._ck(self.lib.gpb_like_set(self.h, nat.ptr(yexp), nat.ptr(cov_exp)))

    def loglike(self, Xs, out=None, accumulate=False, check=True):
        """Block log-likelihood for every row of Xs.  torch(cuda) in/out stays on the device
        and is asynchronous when check=False."""
        self._need_data()
        npd = C.c_int(0)
        if _is_torch(Xs):
            import torch
            Xs = self._check_cols(Xs, "Xs")
            W = Xs.shape[0]
            if out is None:
                out = torch.empty(W, dtype=torch.float64, device=Xs.device)
                accumulate = False
            else:
                self._dev(out, (W,), "out")
            self._ck(self.lib.gpb_loglike(self.h, nat.ptr(Xs), W, 1, nat.ptr(out),
                                          1 if accumulate else 0, C.byref(npd) if check else None))
        else:
            Xs = nat.f64(Xs).reshape(-1, self.d)
            W = Xs.shape[0]
            if out is None:
                out = np.empty(W)
                accumulate = False
            self._ck(self.lib.gpb_loglike(self.h, nat.ptr(Xs), W, 0, nat.ptr(out), 1 if accumulate else 0,
                                          C.byref(npd) if check else None))
        self.last_not_pd = npd.value
        return out

    def logpost(self, X_dev, out, accumulate, lo_dev, hi_dev, outside, const):
        """Fused device log-posterior of the last emulator block: log-likelihood (+= when accumulate),
        strict prior box, constant.  Asynchronous; torch cuda tensors only."""
        self._need_data()
        self._track_stream()
        X_dev = self._dev(X_dev, (None, self.d), "X")               # contiguous: `out` rows must line up with X rows
        self._dev(lo_dev, (self.d,), "lo"), self._dev(hi_dev, (self.d,), "hi")
        self._dev(out, (X_dev.shape[0],), "out")
        self._ck(self.lib.gpb_logpost(self.h, nat.ptr(X_dev), X_dev.shape[0], nat.ptr(out), 1 if accumulate else 0,
                                      nat.ptr(lo_dev), nat.ptr(hi_dev), float(outside), float(const)))
        return out

    def mvn_loglike(self, dY, cov):
        """Batched mvn_loglike on explicit dY[W,M], cov[W,M,M] (numpy or torch cuda)."""
        npd = C.c_int(0)
        if _is_torch(dY):
            import torch
            self._track_stream()
            dY = self._dev(dY, (None, None), "dY", contiguous=False)
            W, M = dY.shape
            cov = self._dev(cov, (W, M, M), "cov", contiguous=False)
            out = torch.empty(W, dtype=torch.float64, device=dY.device)
            self._ck(self.lib.gpb_mvn_loglike(self.h, nat.ptr(dY), nat.ptr(cov), W, M, 1,
                                              nat.ptr(out), C.byref(npd)))
        else:
            dY, cov = nat.f64(dY), nat.f64(cov)
            W, M = dY.shape
            out = np.empty(W)
            self._ck(self.lib.gpb_mvn_loglike(self.h, nat.ptr(dY), nat.ptr(cov), W, M, 0, nat.ptr(out),
                                              C.byref(npd)))
        self.last_not_pd = npd.value
        return out

    def box_finish(self, X_dev, lo_dev, hi_dev, outside, const, ll_dev):
        self._track_stream()
        X_dev = self._dev(X_dev, (None, None), "X")
        W, d = X_dev.shape
        self._dev(lo_dev, (d,), "lo"), self._dev(hi_dev, (d,), "hi"), self._dev(ll_dev, (W,), "ll")
        self._ck(self.lib.gpb_box_finish(self.h, nat.ptr(X_dev), W, d, nat.ptr(lo_dev), nat.ptr(hi_dev),
                                         float(outside), float(const), nat.ptr(ll_dev)))

    # ------------------------------------------------------------------ parameterTrafoPCA input map
    def set_param_map(self, ppca, d_in):
        """Upload the fitted parameter-space PCA (param_pca.ParameterPCA) for the device pre-pass."""
        from .param_pca import IDX_BULK, IDX_SHEAR, IDX_YLOSS, T_GRID, MUB_GRID, YINIT_GRID
        grids = (T_GRID, MUB_GRID, YINIT_GRID)
        idx = (IDX_BULK, IDX_SHEAR, IDX_YLOSS)
        G = len(ppca.groups)
        maxpc = int(max(g.pca.n_components_ for g in ppca.groups))
        # column bookkeeping exactly as the reference does it on the values (delete, then append)
        cols = np.arange(d_in, dtype=np.int64)
        desc = np.full((G, 6), -1, dtype=np.int32)
        tab = np.zeros((G, 4 + maxpc, 100))
        for gi, g in enumerate(ppca.groups):
            k = int(g.pca.n_components_)
            cols = np.concatenate((np.delete(cols, idx[gi]), -1 - (gi * maxpc + np.arange(k))))
            desc[gi, 0] = gi
            desc[gi, 1:1 + len(idx[gi])] = idx[gi]
            desc[gi, 5] = k
            tab[gi, 0], tab[gi, 1], tab[gi, 2], tab[gi, 3] = grids[gi], g.scaler.mean_, g.scaler.scale_, g.pca.mean_
            tab[gi, 4:4 + k] = g.pca.components_
        col_src = np.ascontiguousarray(cols, dtype=np.int32)
        self.pmap_d_in, self.pmap_d_out = int(d_in), int(col_src.shape[0])
        self._ck(self.lib.gpb_param_map_set(self.h, self.pmap_d_in, self.pmap_d_out, nat.ptr(col_src), G,
                                            nat.ptr(np.ascontiguousarray(desc)), nat.ptr(np.ascontiguousarray(tab)),
                                            maxpc))

    def param_map(self, X_dev, out=None):
        """X_dev[W, d_in] (torch cuda f64) -> GP input [W, d_out] on the device, asynchronous."""
        import torch
        self._track_stream()
        if getattr(self, "pmap_d_in", -1) < 0:
            raise nat.GPBError("param_map before set_param_map")
        X_dev = self._dev(X_dev, (None, self.pmap_d_in), "X", contiguous=False)
        W = X_dev.shape[0]
        if out is None:
            out = torch.empty((W, self.pmap_d_out), dtype=torch.float64, device=X_dev.device)
        else:
            self._dev(out, (W, self.pmap_d_out), "out")
        self._ck(self.lib.gpb_param_map(self.h, nat.ptr(X_dev), W, nat.ptr(out)))
        return out

    # ------------------------------------------------------------------ diagnostics
    def tile_trace(self, capacity):
        """arm (capacity > 0) or disarm (0) the per-tile placement/timing trace of the predict kernel"""
        self._ck(self.lib.gpb_debug_tile_trace(self.h, int(capacity)))
        self._trace_cap = int(capacity)

    def tile_trace_read(self):
        """records[n, 8] uint32: HW_ID, XCC_ID, gp, row block, walker tile, start, end (100 MHz ticks), blockIdx"""
        rec = np.zeros((self._trace_cap, 8), dtype=np.uint32)
        n = C.c_int64(0)
        self._ck(self.lib.gpb_debug_tile_trace_read(self.h, nat.ptr(rec), self._trace_cap, C.byref(n)))
        return rec[:n.value]

    # ------------------------------------------------------------------ RCCL without torch.distributed
    def dist_available(self):
        """True when librccl loads with the entry points gpb_dist_* use (no communicator is created)"""
        return self.lib.gpb_dist_available() == 1

    def dist_uid(self):
        """128-byte ncclUniqueId (rank 0 creates it and hands it to the other ranks out of band)."""
        uid = np.zeros(128, dtype=np.uint8)
        self._ck(self.lib.gpb_dist_uid(nat.ptr(uid)))
        return uid.tobytes()

    def dist_init(self, rank, nranks, uid):
        buf = np.frombuffer(bytes(uid), dtype=np.uint8).copy()
        if buf.size != 128:
            raise ValueError("uid must be the 128 bytes of dist_uid()")
        self._ck(self.lib.gpb_dist_init(self.h, int(rank), int(nranks), nat.ptr(buf)))
        self._dist_world = int(nranks)

    def dist_allgather(self, send, recv):
        """In-stream ncclAllGather of float64 device tensors: recv[r*n:(r+1)*n] = rank r's send[:n]
        (in place when send is recv[rank*n:(rank+1)*n])."""
        self._track_stream()
        n = send.numel()
        self._dev(send, (n,), "send")
        self._dev(recv, (n * getattr(self, "_dist_world", 0),), "recv")
        self._ck(self.lib.gpb_dist_allgather(self.h, nat.VP(send.data_ptr()), nat.VP(recv.data_ptr()), n))
        return recv

    def dist_finalize(self):
        self._ck(self.lib.gpb_dist_finalize(self.h))
        self._dist_world = 0

    def test_gemm(self, A, B, mode=0, tile=128):
        A, B = nat.f64(A), nat.f64(B)
        if mode == 0:
            M, K = A.shape; N = B.shape[1]
        elif mode == 1:
            M, K = A.shape; N = B.shape[0]
        else:
            K, M = A.shape; N = B.shape[1]
        Cm = np.empty((M, N))
        self._ck(self.lib.gpb_test_gemm(self.h, M, N, K, nat.ptr(A), nat.ptr(B), nat.ptr(Cm),
                                        mode | (4 if tile == 64 else 0)))
        return Cm

    def force_tile(self, tile=0, switch_tiles=0):
        """k_predict tile: 0 = by rule, 128, 64, 32 (64 rows x 32 walkers), 65 (64 x 128) — same bits whatever the shape;
        switch_tiles > 0: the rule's switch point to 128x128 tiles (tiles per 256 CUs)."""
        self.tune("force_tile", int(tile))
        if switch_tiles > 0:
            self.tune("tile_switch", int(switch_tiles))

    def tune(self, key, value):
        """option keys of gpb_ctx_option by name (launch geometry, tile rules, test hooks; 'predict_sliced': the int8 predict kernel)"""
        k = {"xcd": 0, "chol_outer": 4, "resident": 5, "narrow_switch": 7, "mvn_wg_switch": 8, "chol_inner_tile": 9,
             "tile_priority": 10, "fuse_finalize": 11, "trtri_tile": 12, "syrk_tile": 14, "tri_skip": 17, "kcross_dot": 18,
             "kcross_chunks": 19, "kcross_wpl": 20, "mid_switch": 22, "lowrank": 23, "chol_lookahead": 25, "sim_ranks": 26,
             "compact": 27, "tile_by_live": 28, "premark": 29, "fuse_accept_propose": 30, "sim_rank": 32, "tile_switch_c": 33,
             "mid_switch_c": 34, "narrow_switch_c": 35, "balance_shards": 36, "chain_batch": 40, "force_tile": 42, "generic_mvn": 43,
             "tile_switch": 44, "chol_pair": 47, "lr_split": 49, "kinv_tile": 50, "predict_sliced": 51}[key]
        self._ck(self.lib.gpb_ctx_option(self.h, k, int(value)))

    @property
    def has_variants(self):
        """True for the debug build (GPB_DEBUG_LIB=1): every measured-and-rejected kernel variant behind its tune key"""
        return self.lib.gpb_debug_has_variants() == 1

    def fit_piece(self, piece):
        """measurement hook: enqueue one piece of factor() alone ('kmat', 'potrf', 'trtri', 'alpha'); call factor() afterwards"""
        self._ck(self.lib.gpb_profile_fit_piece(self.h, {"kmat": 0, "potrf": 1, "trtri": 2, "alpha": 3}[piece]))

    def force_generic_mvn(self, on=True):
        """route the block log-likelihood through the generic Cholesky kernel (what M > 64 takes) whatever M"""
        self.tune("generic_mvn", 1 if on else 0)

    def profile(self, on=True):
        self._ck(self.lib.gpb_profile_enable(self.h, 1 if on else 0))

    def profile_read(self):
        """(launches, total_ms, gp_walker_pairs) of the timed k_predict launches since the last read."""
        n, ms, u = C.c_int64(0), C.c_double(0.0), C.c_double(0.0)
        self._ck(self.lib.gpb_profile_read(self.h, C.byref(n), C.byref(ms), C.byref(u)))
        return n.value, ms.value, u.value

    def probe_fp64(self, mode):
        out = C.c_double(0.0)
        self._ck(self.lib.gpb_probe_fp64(self.h, int(mode), C.byref(out)))
        return out.value
