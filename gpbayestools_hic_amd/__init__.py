"""
gpbayestools_hic_amd — MI355X-native GP-emulator + MCMC log-posterior engine.

Drop-in for the hot path of Hendrik1704/GPBayesTools-HIC (src/emulator.py fit()/predict() and the
src/mcmc.py log-posterior loop): hand-written HIP kernels for gfx950 behind a C ABI
(include/gpbayes.h, ctypes), Python only for orchestration.  No CPU fallback.
"""
__version__ = "0.1.0"

__all__ = ["Emulator", "Chain", "mvn_loglike", "GPEngine", "StretchSampler", "LoggingEnsembleSampler",
           "WalkerSharding"]


def __getattr__(name):   # lazy: importing the package must not need torch / the built library
    if name == "Emulator":
        from .emulator import Emulator
        return Emulator
    if name in ("Chain", "mvn_loglike"):
        from . import mcmc
        return getattr(mcmc, name)
    if name == "GPEngine":
        from .engine import GPEngine
        return GPEngine
    if name in ("StretchSampler", "LoggingEnsembleSampler"):
        from . import sampler
        return getattr(sampler, name)
    if name == "WalkerSharding":
        from .dist import WalkerSharding
        return WalkerSharding
    raise AttributeError(name)
