"""Build the HIP extension in-tree: gpbayestools_hic_amd/libgpbayes.so (gfx950 only)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgpbayes.so")
LIB_DEBUG = os.path.join(HERE, "libgpbayes_debug.so")
SOURCES = ["gpb_api.hip", "gpb_fit.hip", "gpb_chol.hip", "gpb_predict.hip", "gpb_sliced.hip", "gpb_like.hip", "gpb_cov.hip", "gpb_pmap.hip", "gpb_pool.hip"]
HEADERS = ["gpb_internal.h", "gemm_tile.h", "chol_block.h", "fast_math.h", os.path.join("..", "..", "include", "gpbayes.h"),
           os.path.join("..", "..", "include", "gpbayes_debug.h")]


def _stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_native(force=False, verbose=False, debug_variants=False):
    """libgpbayes.so: the product library (the kernels its own rules select).  debug_variants=True: libgpbayes_debug.so,
    the same sources with -DGPB_DEBUG_VARIANTS — every measured-and-rejected kernel variant behind its gpb_ctx_option key and
    the test hooks of include/gpbayes_debug.h, for the sweeps in tools/ and the tests that need them (GPB_DEBUG_LIB=1, or
    _native.debug_library() for one test)."""
    lib = LIB_DEBUG if debug_variants else LIB
    if not force and not _stale(lib):
        return lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    bdir = os.path.join(HERE, "build", "debug" if debug_variants else "product")
    os.makedirs(bdir, exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        # -fvisibility=hidden: the library exports the entry points of include/gpbayes.h (GPB_API) and, in the debug build,
        # include/gpbayes_debug.h — nothing of the C++ inside
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-c",
               os.path.join(CSRC, src), "-o", obj] + (["-DGPB_DEBUG_VARIANTS"] if debug_variants else [])
        if debug_variants:
            cmd += os.environ.get("GPB_DEBUG_EXTRA_DEFINES", "").split()      # A/B builds of the debug library only
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode()))
    # -Bsymbolic: the entry points call each other inside their own library, so that the product and the debug build can be
    # loaded into one process side by side (tests that need the debug hooks, _native.debug_library())
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", lib] + objs + ["-ldl"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stdout.decode()))
    return lib


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
    if "--debug-variants" in sys.argv:
        print(build_native(force="--force" in sys.argv, verbose=True, debug_variants=True))
