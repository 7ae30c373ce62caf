"""
ctypes binding of include/gpbayes.h (the C-ABI drop-in boundary).

There is no CPU fallback: if the HIP library is missing or no gfx950 device is
visible, creating an engine raises.  torch is imported first so that this
process holds exactly one HIP runtime (libamdhip64.so.7) shared by torch tensors
and the kernels.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# libgpbayes.so: the product library (exports include/gpbayes.h, nothing else).  libgpbayes_debug.so: the same sources with
# -DGPB_DEBUG_VARIANTS — the measured-and-rejected kernel variants behind their option keys plus the test hooks of
# include/gpbayes_debug.h.  GPB_DEBUG_LIB=1 makes the debug build the default of a process (the sweeps in tools/);
# debug_library() does so for a block (tests that need a hook); both can be loaded side by side (-Bsymbolic, RTLD_LOCAL).
LIB_PATHS = {False: os.path.join(HERE, "libgpbayes.so"), True: os.path.join(HERE, "libgpbayes_debug.so")}
_default_debug = os.environ.get("GPB_DEBUG_LIB") == "1"
LIB_PATH = LIB_PATHS[_default_debug]

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
c_i64 = C.c_int64
c_u64 = C.c_uint64
VP = C.c_void_p

# name -> (restype, argtypes); BOUNDARY mirrors include/gpbayes.h one to one (the drop-in C ABI), DEBUG mirrors
# include/gpbayes_debug.h (test / tuning / measurement hooks)
BOUNDARY = {
    "gpb_version": (C.c_int, []),
    "gpb_device_count": (C.c_int, []),
    "gpb_ctx_create": (C.c_int, [C.c_int, VP, C.POINTER(VP)]),
    "gpb_ctx_destroy": (C.c_int, [VP]),
    "gpb_ctx_set_stream": (C.c_int, [VP, VP]),
    "gpb_sync": (C.c_int, [VP]),
    "gpb_pool_trim": (C.c_int, []),
    "gpb_last_error": (C.c_char_p, [VP]),
    "gpb_stream": (VP, [VP]),
    "gpb_gp_set": (C.c_int, [VP, c_i64, c_i64, c_i64, VP, VP, C.c_int, C.c_double]),
    "gpb_gp_set_multi": (C.c_int, [VP, c_i64, c_i64, VP, VP, VP, C.c_int, C.c_double]),
    "gpb_gp_lml_subset": (C.c_int, [VP, c_i64, VP, VP, VP, VP, VP]),
    "gpb_gp_set_theta": (C.c_int, [VP, VP]),
    "gpb_gp_factor": (C.c_int, [VP, VP]),
    "gpb_gp_get": (C.c_int, [VP, C.c_int, VP]),
    "gpb_gp_lml": (C.c_int, [VP, VP, VP, VP, VP]),
    "gpb_gp_predict": (C.c_int, [VP, VP, c_i64, C.c_int, VP, VP]),
    "gpb_gp_predict_cov": (C.c_int, [VP, VP, c_i64, C.c_int, VP, VP]),
    "gpb_emu_set_transform": (C.c_int, [VP, C.c_int, c_i64, VP, VP, VP, VP]),
    "gpb_emu_predict": (C.c_int, [VP, VP, c_i64, C.c_int, VP, VP, VP]),
    "gpb_like_set": (C.c_int, [VP, VP, VP]),
    "gpb_loglike": (C.c_int, [VP, VP, c_i64, C.c_int, VP, C.c_int, VP]),
    "gpb_logpost": (C.c_int, [VP, VP, c_i64, VP, C.c_int, VP, VP, C.c_double, C.c_double]),
    "gpb_mvn_loglike": (C.c_int, [VP, VP, VP, c_i64, c_i64, C.c_int, VP, VP]),
    "gpb_box_finish": (C.c_int, [VP, VP, c_i64, c_i64, VP, VP, C.c_double, C.c_double, VP]),
    "gpb_param_map_set": (C.c_int, [VP, c_i64, c_i64, VP, C.c_int32, VP, VP, C.c_int32]),
    "gpb_param_map": (C.c_int, [VP, VP, c_i64, VP]),
    "gpb_stretch_propose": (C.c_int, [VP, VP, c_i64, c_i64, C.c_int, c_u64, c_u64, C.c_double, VP, VP, C.c_int]),
    "gpb_stretch_accept": (C.c_int, [VP, VP, VP, c_i64, c_i64, C.c_int, c_u64, c_u64, VP, VP, VP, VP, C.c_int]),
    "gpb_stretch_nan_count": (C.c_int, [VP, VP, C.c_int]),
    "gpb_emcee_run": (C.c_int, [VP, VP, VP, c_i64, c_i64, c_u64, c_u64, C.c_double, C.c_int, VP, VP, C.c_double,
                                C.c_double, VP, VP, VP]),
    "gpb_chain_supported": (C.c_int, [VP, C.c_int]),
    "gpb_chain_logpost": (C.c_int, [VP, C.c_int, VP, c_i64, VP, VP, VP, C.c_double, C.c_double]),
    "gpb_chain_emcee_run": (C.c_int, [VP, C.c_int, VP, VP, c_i64, c_i64, c_u64, c_u64, C.c_double, C.c_int, VP, VP,
                                      C.c_double, C.c_double, VP, VP, VP]),
    "gpb_chain_emcee_prepare": (C.c_int, [VP, C.c_int, c_i64]),
    "gpb_dist_available": (C.c_int, []),
    "gpb_dist_uid": (C.c_int, [VP]),
    "gpb_dist_init": (C.c_int, [VP, C.c_int, C.c_int, VP]),
    "gpb_dist_allgather": (C.c_int, [VP, VP, VP, c_i64]),
    "gpb_dist_finalize": (C.c_int, [VP]),
    "gpb_ctx_option": (C.c_int, [VP, C.c_int, C.c_int]),
    "gpb_debug_has_variants": (C.c_int, []),
    "gpb_profile_enable": (C.c_int, [VP, C.c_int]),
    "gpb_profile_read": (C.c_int, [VP, VP, VP, VP]),
    "gpb_profile_fit_piece": (C.c_int, [VP, C.c_int]),
}
DEBUG = {
    "gpb_test_split_perm": (C.c_int, [VP, c_i64, c_u64, c_u64, VP]),
    "gpb_test_philox": (C.c_int, [VP, c_i64, VP, VP]),
    "gpb_test_stretch_draws": (C.c_int, [VP, c_i64, C.c_int, c_u64, c_u64, C.c_int, VP, VP, VP, VP]),
    "gpb_test_gemm": (C.c_int, [VP, c_i64, c_i64, c_i64, VP, VP, VP, C.c_int]),
    "gpb_debug_loopback_group": (C.c_int, [VP, C.c_int]),
    "gpb_debug_loopback_release": (C.c_int, [VP]),
    "gpb_debug_tile_trace": (C.c_int, [VP, c_i64]),
    "gpb_debug_tile_trace_read": (C.c_int, [VP, VP, c_i64, VP]),
    "gpb_probe_fp64": (C.c_int, [VP, C.c_int, VP]),
}
PROTOTYPES = dict(BOUNDARY, **DEBUG)

_libs = {False: None, True: None}


class GPBError(RuntimeError):
    pass


def _debug_only(name):
    def stub(*_a, **_k):
        raise GPBError(f"{name} is a hook of the debug build (libgpbayes_debug.so: `python -m gpbayestools_hic_amd.build "
                       "--debug-variants`, then GPB_DEBUG_LIB=1 or _native.debug_library())")
    return stub


class _Lib:
    """the loaded library: its entry points as attributes; in the product library the debug hooks are stubs that raise"""

    def __init__(self, cdll, debug):
        self._cdll, self.is_debug = cdll, debug
        for name, (res, args) in BOUNDARY.items():
            fn = getattr(cdll, name)
            fn.restype, fn.argtypes = res, args
            setattr(self, name, fn)
        for name, (res, args) in DEBUG.items():
            if debug:
                fn = getattr(cdll, name)
                fn.restype, fn.argtypes = res, args
                setattr(self, name, fn)
            else:
                setattr(self, name, _debug_only(name))


def load(debug=None):
    """Load libgpbayes.so — or, debug=True / GPB_DEBUG_LIB=1 / inside debug_library(), libgpbayes_debug.so (raises if it has
    not been built)."""
    debug = _default_debug if debug is None else bool(debug)
    if _libs[debug] is not None:
        return _libs[debug]
    path = LIB_PATHS[debug]
    if not os.path.exists(path):
        raise GPBError(
            f"{path} not found: build it with `python -m gpbayestools_hic_amd.build"
            + (" --debug-variants`" if debug else "`") + " (there is no CPU fallback)")
    try:
        import torch  # noqa: F401  (one shared HIP runtime per process)
    except Exception:  # pragma: no cover
        pass
    _libs[debug] = _Lib(C.CDLL(path), debug)
    return _libs[debug]


class debug_library:
    """with debug_library(): engines created inside the block bind libgpbayes_debug.so (test hooks, kernel variants); engines
    created before keep the library they were created with."""

    def __enter__(self):
        global _default_debug
        load(True)
        self._prev, _default_debug = _default_debug, True
        return self

    def __exit__(self, *exc):
        global _default_debug
        _default_debug = self._prev
        return False


def ptr(a):
    """void* of a C-contiguous float64/int32 numpy array, a torch tensor, an int, or None."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data_as(VP)
    if isinstance(a, int):
        return VP(a)
    if hasattr(a, "data_ptr"):
        return VP(a.data_ptr())
    raise TypeError(type(a))


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def host_empty(shape, pinned_from=1 << 20):
    """float64 result array on the host.  Large ones (>= `pinned_from` bytes) live in page-locked memory from torch's
    caching host allocator, so that the library's device-to-host copy is one DMA at the link's rate instead of a staged
    copy into fresh pageable pages (82 MB of covariances: 10.9 ms -> see DESIGN); the array is an ordinary numpy array
    that keeps its block alive, and the allocator takes the block back when the array is dropped."""
    n = int(np.prod(shape))
    if 8 * n >= pinned_from:
        try:
            import torch
            if torch.cuda.is_available():
                return torch.empty(tuple(int(x) for x in shape), dtype=torch.float64, pin_memory=True).numpy()
        except Exception:       # no torch / no pinned memory left: pageable memory is always correct
            pass
    return np.empty(shape)
