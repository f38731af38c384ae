"""ctypes bindings for the C ABI declared in include/ahmc_hip.h.

`CLib(path)` binds every entry point of the header on one shared library.  The product
library is advancedhmc.jl_amd/csrc/libahmc_hip.so (HIP, gfx950) and is the ONLY library this
package ever opens by itself: `load_hip_library()` raises if it has not been built — there is
no CPU fallback.  (tests/ bind the same `CLib` class onto the CPU checker built under oracle/ to obtain the
checker; that path is never taken from inside the package.)
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

# --- enums (mirror include/ahmc_hip.h) --------------------------------------------------------
AHMC_ABI_VERSION = 6
OK, ERR_ARGUMENT, ERR_UNSUPPORTED, ERR_RUNTIME, ERR_STATE = 0, 1, 2, 3, 4
F32, F64 = 0, 1
METRIC_UNIT, METRIC_DIAG, METRIC_DENSE = 0, 1, 2
(TARGET_ISO_GAUSS, TARGET_DIAG_GAUSS, TARGET_FUNNEL, TARGET_HIER_GAUSS, TARGET_DENSE_GAUSS,
 TARGET_EXTERNAL, TARGET_PLUGIN, TARGET_KERNEL) = range(8)
KERNEL_HIP_FUNCTION, KERNEL_HIP_SYMBOL, KERNEL_HOST = 0, 1, 2
INTEGRATOR_LEAPFROG, INTEGRATOR_JITTERED, INTEGRATOR_TEMPERED = 0, 1, 2
TS_ENDPOINT, TS_MULTINOMIAL, TS_SLICE = 0, 1, 2
TC_CLASSIC, TC_GENERALISED, TC_STRICT = 0, 1, 2
ADAPT_NONE, ADAPT_STEPSIZE, ADAPT_MASSMATRIX, ADAPT_NAIVE, ADAPT_STAN = range(5)
VAR_WELFORD, VAR_NUTPIE, VAR_POOLED = 0, 1, 2
UNIQUE_ID_BYTES = 128

# stat fields: name -> (id, is_int)
STAT_FIELDS = {
    "n_steps": (0, True),
    "is_accept": (1, True),
    "acceptance_rate": (2, False),
    "log_density": (3, False),
    "hamiltonian_energy": (4, False),
    "hamiltonian_energy_error": (5, False),
    "max_hamiltonian_energy_error": (6, False),
    "tree_depth": (7, True),
    "numerical_error": (8, True),
    "step_size": (9, False),
    "nom_step_size": (10, False),
}


class KernelCfg(C.Structure):
    """ahmc_kernel_cfg"""
    _fields_ = [
        ("nuts", C.c_int32),
        ("sampler", C.c_int32),
        ("criterion", C.c_int32),
        ("max_depth", C.c_int32),
        ("delta_max", C.c_double),
        ("L", C.c_int64),
        ("lambda_", C.c_double),
        ("refresh_alpha", C.c_double),
    ]


class AdaptorState(C.Structure):
    """ahmc_adaptor_state"""
    _fields_ = [
        ("kind", C.c_int32),
        ("var_estimator", C.c_int32),
        ("init_buffer", C.c_int32),
        ("term_buffer", C.c_int32),
        ("window_size", C.c_int32),
        ("adapting", C.c_int32),
        ("has_da", C.c_int32),
        ("n_welford", C.c_int32),
        ("delta", C.c_double),
        ("stan_i", C.c_int64),
        ("n_adapts", C.c_int64),
        ("wv_n", C.c_int64),
        ("iteration", C.c_int64),
    ]


class AHMCError(RuntimeError):
    """A non-zero status from the C ABI.  `ArgumentError` mirrors Julia's exception of the
    same name (src/hamiltonian.jl:55-57, src/abstractmcmc.jl:64-70)."""

    def __init__(self, code, msg):
        super().__init__(f"[ahmc status {code}] {msg}")
        self.code = code


class ArgumentError(AHMCError, ValueError):
    pass


class UnsupportedError(AHMCError, NotImplementedError):
    pass


_vp, _i32, _i64, _u64, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double

# name -> (restype, argtypes); the complete symbol list of include/ahmc_hip.h
SIGNATURES = {
    "ahmc_create": (_i32, [_i32, _i32, _i64, _i64, _vp, C.POINTER(_vp)]),
    "ahmc_destroy": (_i32, [_vp]),
    "ahmc_last_error": (C.c_char_p, [_vp]),
    "ahmc_abi_version": (_i32, []),
    "ahmc_backend": (C.c_char_p, []),
    "ahmc_sync": (_i32, [_vp]),
    "ahmc_stream": (_vp, [_vp]),
    "ahmc_set_target": (_i32, [_vp, _i32, _vp, _i64]),
    "ahmc_set_target_plugin": (_i32, [_vp, C.c_char_p, _vp, _i64]),
    "ahmc_set_target_kernel": (_i32, [_vp, _i32, _vp, _i32, _i32, _vp]),
    "ahmc_set_metric": (_i32, [_vp, _i32, _vp, _i64]),
    "ahmc_get_metric": (_i32, [_vp, _vp, _i64]),
    "ahmc_set_stepsize": (_i32, [_vp, _vp, _i64]),
    "ahmc_get_stepsize": (_i32, [_vp, _vp]),
    "ahmc_set_integrator": (_i32, [_vp, _i32, _f64]),
    "ahmc_seed": (_i32, [_vp, _u64, _u64, _u64, _u64]),
    "ahmc_set_position": (_i32, [_vp, _vp, _vp]),
    "ahmc_set_phasepoint": (_i32, [_vp, _vp, _vp, _vp, _vp]),
    "ahmc_get_phasepoint": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "ahmc_refresh_momentum": (_i32, [_vp, _f64]),
    "ahmc_leapfrog": (_i32, [_vp, _i64]),
    "ahmc_lf_pre": (_i32, [_vp, _i32, _i64, _i64]),
    "ahmc_lf_post": (_i32, [_vp, _i32, _i64, _i64, _vp, _vp]),
    "ahmc_theta_ptr": (_vp, [_vp]),
    "ahmc_hmc_transition": (_i32, [_vp, _i64, _f64, _i32]),
    "ahmc_nuts_transition": (_i32, [_vp, _i32, _f64, _i32, _i32]),
    "ahmc_get_stat": (_i32, [_vp, _i32, _vp]),
    "ahmc_find_good_stepsize": (_i32, [_vp, _f64, _i32]),
    "ahmc_adaptor_init": (_i32, [_vp, _i32, _f64, _i32, _i32, _i32]),
    "ahmc_adapt": (_i32, [_vp, _i64, _i64, _vp, _vp]),
    "ahmc_adapt_point": (_i32, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "ahmc_set_var_estimator": (_i32, [_vp, _i32]),
    "ahmc_stan_windows": (_i32, [_i32, _i32, _i32, _i64, C.POINTER(_i64), C.POINTER(_i64),
                                 C.POINTER(_i64), _i32, C.POINTER(_i32)]),
    "ahmc_sample": (_i32, [_vp, C.POINTER(KernelCfg), _i64, _i64, _i32, _vp]),
    "ahmc_sample_from": (_i32, [_vp, C.POINTER(KernelCfg), _i64, _i64, _i64, _i32, _vp]),
    "ahmc_sample_reserve": (_i32, [_vp, C.POINTER(KernelCfg), _i64]),
    "ahmc_set_ref_compat": (_i32, [_vp, _i32]),
    "ahmc_get_accum": (_i32, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), _vp, _vp]),
    "ahmc_reset_accum": (_i32, [_vp]),
    "ahmc_get_accum_state": (_i32, [_vp, C.POINTER(_i64), _vp, _vp, _vp, _vp, _vp]),
    "ahmc_set_accum_state": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "ahmc_get_info": (_i32, [_vp, _i32, C.POINTER(_i64)]),
    "ahmc_ext_begin": (_i32, [_vp, C.POINTER(KernelCfg), _i32]),
    "ahmc_ext_find_good_stepsize_begin": (_i32, [_vp, _f64, _i32]),
    "ahmc_ext_pending": (_i32, [_vp, C.POINTER(_i64), _vp, _vp]),
    "ahmc_ext_advance": (_i32, [_vp, _vp, _vp]),
    "ahmc_ext_cancel": (_i32, [_vp]),
    "ahmc_get_adaptor_state": (_i32, [_vp, C.POINTER(AdaptorState), _vp, _vp]),
    "ahmc_set_adaptor_state": (_i32, [_vp, C.POINTER(AdaptorState), _vp, _vp]),
    "ahmc_comm_unique_id": (_i32, [_vp]),
    "ahmc_comm_init": (_i32, [_vp, _vp, _i32, _i32]),
    "ahmc_set_comm": (_i32, [_vp, _vp, _i32, _i32]),
    "ahmc_comm_info": (_i32, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "ahmc_gather_moments": (_i32, [_vp, _vp, _vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "ahmc_gather_state": (_i32, [_vp, _vp]),
    "ahmc_ebfmi": (_i32, [_vp, _vp]),
    "ahmc_ess": (_i32, [_vp, _vp, _i64, _vp]),
}


class CLib:
    """One loaded implementation of the ABI."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.path = os.path.abspath(path)
        self.dll = C.CDLL(self.path, mode=C.RTLD_GLOBAL if hasattr(C, "RTLD_GLOBAL") else 0)
        missing = []
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(self.dll, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.restype = res
            fn.argtypes = args
        if missing:
            raise ImportError(f"{self.path} does not export: {', '.join(missing)}")
        v = self.dll.ahmc_abi_version()
        if v != AHMC_ABI_VERSION:
            raise ImportError(f"{self.path}: ABI version {v}, expected {AHMC_ABI_VERSION}")
        self.backend = self.dll.ahmc_backend().decode()

    def check(self, code: int, ctx=None):
        if code == OK:
            return
        msg = self.dll.ahmc_last_error(ctx)
        msg = msg.decode("utf-8", "replace") if msg else ""
        if code == ERR_ARGUMENT:
            raise ArgumentError(code, msg)
        if code == ERR_UNSUPPORTED:
            raise UnsupportedError(code, msg)
        raise AHMCError(code, msg)


_HIP_LIB = None


def hip_library_path() -> str:
    # AHMC_HIP_LIB: another build of the SAME HIP engine (kernel-tuning experiments); its
    # backend string is still checked by load_hip_library()
    return os.environ.get("AHMC_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc",
                                                          "libahmc_hip.so")


def load_hip_library() -> CLib:
    """Open the HIP engine.  Fails loudly when it has not been built (no fallback)."""
    global _HIP_LIB
    if _HIP_LIB is None:
        path = hip_library_path()
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: build it with `python __graft_entry__.py` (hipcc, gfx950). "
                "This package has no CPU fallback.")
        # libtorch's bundled libamdhip64 must be the HIP runtime of the process if torch is used
        # at all (bench.py uses torch.distributed/RCCL): import torch first so that our library's
        # NEEDED libamdhip64.so.7 resolves to the copy already mapped.
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for single-process use
            pass
        _HIP_LIB = CLib(path)
        if not _HIP_LIB.backend.startswith("hip"):
            raise ImportError(f"{path} reports backend {_HIP_LIB.backend!r}, expected the HIP engine")
    return _HIP_LIB


def np_dtype(dtype_code: int):
    return np.float32 if dtype_code == F32 else np.float64


def dtype_code(dtype) -> int:
    dt = np.dtype(dtype)
    if dt == np.float32:
        return F32
    if dt == np.float64:
        return F64
    raise ArgumentError(ERR_ARGUMENT, f"element type must be float32 or float64, got {dt}")


class _OwningPtr(C.c_void_p):
    """void* that keeps the object it points into alive.  `as_ptr(np.ascontiguousarray(x))` hands the C call the
    address of a TEMPORARY: with a bare c_void_p the temporary is freed as soon as as_ptr returns, i.e. before the
    call runs (round-1 bug: the gradient of a user log-density was read from freed memory once D·N·8 B outgrew
    numpy's small-block cache).  The pointer object lives in the caller's argument tuple until the call returns,
    and so does what it refers to."""
    _keep = None


def as_ptr(a) -> C.c_void_p:
    """host numpy array, torch tensor (host or device) or raw int address -> void* (owning, see _OwningPtr)"""
    if a is None:
        return C.c_void_p(None)
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        p = _OwningPtr(a.ctypes.data)
    elif hasattr(a, "data_ptr"):
        p = _OwningPtr(a.data_ptr())
    else:
        raise TypeError(f"cannot take the address of {type(a)}")
    p._keep = a
    return p
