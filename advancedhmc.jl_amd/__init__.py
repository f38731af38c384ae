"""advancedhmc.jl_amd — MI355X-native chain-batched HMC/NUTS trajectory engine.

Drop-in for the sampler-vec hot path of TuringLang/AdvancedHMC.jl: hand-written HIP kernels
(csrc/, gfx950) behind the C ABI of include/ahmc_hip.h, plus this host-side mirror of the
reference's operator interface.  The directory name is not a valid Python identifier; import it
through the root-level shim:  `import ahmc_amd`.
"""
from . import _capi as capi
from ._capi import AHMCError, ArgumentError, UnsupportedError, CLib, load_hip_library, hip_library_path
from .api import *  # noqa: F401,F403
from .api import (Engine, PhiloxRNG, sample, find_good_stepsize, stan_windows, EBFMI, renew, energy, neg_energy)
from .build import build_hip_library  # noqa: E402
from . import shard  # noqa: E402,F401
from . import diagnostics  # noqa: E402,F401
from .diagnostics import ess, bundle_samples  # noqa: E402,F401

__version__ = "0.1.0"
