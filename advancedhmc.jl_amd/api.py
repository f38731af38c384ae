"""Host-side mirror of AdvancedHMC.jl's operator interface for the sampler-vec path.

Julia is absent from the build image, so the host side above the C ABI is written in Python
with the reference's names, argument meaning and error behaviour (so the parity tests read like
test/sampler-vec.jl, test/integrator.jl, test/trajectory.jl).  The Julia package extension that
binds the same ABI with `ccall` is julia/AdvancedHMCMI355XExt.jl.

Every class cites the reference definition it mirrors (paths under the AdvancedHMC.jl checkout).
Arrays are (D, N) column-major like a Julia Matrix: numpy arrays are converted with
`np.asfortranarray`, results come back as F-ordered (D, N) arrays (or (D,) for a single chain).
All compute happens in the C library the `Engine` is bound to — the HIP engine unless a test
injects another `CLib`.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from . import _capi as capi
from ._capi import ArgumentError


# ------------------------------------------------------------------------------------------------
# RNG handle (replaces `rng::AbstractRNG`)
# ------------------------------------------------------------------------------------------------
@dataclass
class PhiloxRNG:
    """Counter-based Philox4x32-10 stream family.  `chain_offset` is the global index of the
    first local chain (multi-GPU shards), `iteration` the transition counter."""
    seed: int = 0
    chain_offset: int = 0
    iteration: int = 0


def _rng_spec(rng, n_chains):
    """int | PhiloxRNG | sequence of PhiloxRNG (one per chain, src/utilities.jl:12-23)."""
    if rng is None:
        return PhiloxRNG(0), 1
    if isinstance(rng, (int, np.integer)):
        return PhiloxRNG(int(rng)), 1
    if isinstance(rng, PhiloxRNG):
        return rng, 1
    rngs = list(rng)
    if len(rngs) != n_chains:  # @argcheck length(rngs) == n_chains
        raise ArgumentError(capi.ERR_ARGUMENT, f"length(rngs) == n_chains must hold, got {len(rngs)} != {n_chains}")
    first = rngs[0]
    if all(r.seed == first.seed and r.chain_offset == first.chain_offset for r in rngs):
        return first, 0  # identically seeded RNG vector: every chain sees the same variates
    if all(r.seed == first.seed and r.chain_offset == first.chain_offset + k for k, r in enumerate(rngs)):
        return first, 1
    raise ArgumentError(capi.ERR_ARGUMENT, "a vector of RNGs must share one seed (same or consecutive streams)")


# ------------------------------------------------------------------------------------------------
# metrics (src/metric.jl:17-120)
# ------------------------------------------------------------------------------------------------
def _size_tuple(sz):
    if isinstance(sz, (int, np.integer)):
        return (int(sz),)
    return tuple(int(s) for s in sz)


class AbstractMetric:
    kind = None

    @property
    def D(self):
        return self.size[0]


class UnitEuclideanMetric(AbstractMetric):
    """src/metric.jl:17-35.  UnitEuclideanMetric([T,] sz)"""
    kind = capi.METRIC_UNIT

    def __init__(self, *args):
        if len(args) == 2:
            self.eltype, sz = np.dtype(args[0]), args[1]
        else:
            self.eltype, sz = np.dtype(np.float64), args[0]
        self.size = _size_tuple(sz)
        self.Minv = None

    def __repr__(self):
        return f"UnitEuclideanMetric({self.eltype}, {self.size})"


class DiagEuclideanMetric(AbstractMetric):
    """src/metric.jl:52-72.  DiagEuclideanMetric(M⁻¹) | DiagEuclideanMetric([T,] sz)"""
    kind = capi.METRIC_DIAG

    def __init__(self, *args):
        if len(args) == 2:
            T, sz = np.dtype(args[0]), _size_tuple(args[1])
            Minv = np.ones(sz, dtype=T, order="F")
        elif isinstance(args[0], np.ndarray):
            Minv = np.asfortranarray(args[0])
            if Minv.dtype not in (np.float32, np.float64):
                Minv = Minv.astype(np.float64)
        else:
            Minv = np.ones(_size_tuple(args[0]), dtype=np.float64, order="F")
        self.Minv = Minv
        self.eltype = Minv.dtype
        self.size = Minv.shape

    @property
    def sqrtMinv(self):
        return np.sqrt(self.Minv)

    def __repr__(self):
        return f"DiagEuclideanMetric({np.array2string(self.Minv.ravel(order='F')[:6], precision=3)} ...)"


class DenseEuclideanMetric(AbstractMetric):
    """src/metric.jl:89-120.  One (D, D) M⁻¹ shared by all chains (the reference has no batched
    dense metric, :103; sharing it is this engine's extension)."""
    kind = capi.METRIC_DENSE

    def __init__(self, *args):
        if len(args) == 2:
            T, D = np.dtype(args[0]), _size_tuple(args[1])[0]
            Minv = np.eye(D, dtype=T, order="F")
        elif isinstance(args[0], np.ndarray):
            Minv = np.asfortranarray(args[0])
        else:
            Minv = np.eye(_size_tuple(args[0])[0], dtype=np.float64, order="F")
        if Minv.ndim != 2 or Minv.shape[0] != Minv.shape[1]:
            raise ArgumentError(capi.ERR_ARGUMENT, "DenseEuclideanMetric needs a square M⁻¹")
        self.Minv = Minv
        self.eltype = Minv.dtype
        self.size = (Minv.shape[0],)


def renew(metric, Minv):
    """src/metric.jl:31,69,117"""
    if isinstance(metric, UnitEuclideanMetric):
        return metric
    return type(metric)(np.asarray(Minv))


# ------------------------------------------------------------------------------------------------
# targets: the (ℓπ, ∂ℓπ∂θ) pair of Hamiltonian (src/hamiltonian.jl:1-20) as built-in families
# ------------------------------------------------------------------------------------------------
@dataclass
class Target:
    kind: int
    D: int
    params: Optional[np.ndarray] = None


def IsoGaussian(D):
    """ℓπ of test/common.jl:76-77 (`Gaussian(zeros(D), ones(D))`)"""
    return Target(capi.TARGET_ISO_GAUSS, int(D))


def DiagGaussian(m, s):
    """test/common.jl:35-77 `Gaussian(m, s)` with the exact gradient (m-x)/s²"""
    m = np.asarray(m, dtype=np.float64).ravel()
    s = np.asarray(s, dtype=np.float64).ravel()
    if m.shape != s.shape:
        raise ArgumentError(capi.ERR_ARGUMENT, "DiagGaussian: m and s must have the same length")
    return Target(capi.TARGET_DIAG_GAUSS, m.size, np.concatenate([m, s]))


def Funnel(D):
    """Neal's funnel as defined in research/notebooks/geweke_test.ipynb cell 4"""
    return Target(capi.TARGET_FUNNEL, int(D))


def HierGaussian(D):
    """θ = (μ, log τ, x₁..x_{D-2}) hierarchical Gaussian"""
    return Target(capi.TARGET_HIER_GAUSS, int(D))


def DenseGaussian(P):
    """ℓπ = -½ θᵀPθ with P the (D, D) precision matrix"""
    P = np.asfortranarray(np.asarray(P, dtype=np.float64))
    return Target(capi.TARGET_DENSE_GAUSS, P.shape[0], P.ravel(order="F"))


@dataclass
class ExternalTarget:
    """User log-density evaluated by the caller: `fn(θ (D,N)) -> (ℓπ (N,), ∇ℓπ (D,N))`, the
    signature of `∂ℓπ∂θ` at test/common.jl:64-74 (what `Hamiltonian(metric, ℓπ, ∂ℓπ∂θ)` /
    LogDensityProblems hand the reference, src/AdvancedHMC.jl:163-186).  `Engine.step` runs through the
    split-step pair ahmc_lf_pre / ahmc_lf_post around this callback; whole transitions and
    find_good_stepsize run through the ask / tell calls ahmc_ext_* (`Engine._ext_drive`).  `fn` may
    also accept `fn(θ, chains)` — the indices of the columns that need evaluating — when
    `takes_chains` is set; the other columns of its result are ignored."""
    D: int
    fn: object
    kind: int = capi.TARGET_EXTERNAL
    params: Optional[np.ndarray] = None
    takes_chains: bool = False


@dataclass
class PluginTarget:
    """User log-density as a HIP DEVICE FUNCTION compiled INTO the engine's trajectory kernels (include/ahmc_user_target.h
    has the contract; ahmc_set_target_plugin): `source` is the path of a header defining
    `ahmc_user::logdensity<T, G, E>(params, D, theta, grad_neg, lane, d0)`.  The engine compiles it (hipcc, cached by content)
    for the context's element type and thread geometry the first time it is bound; from then on the density runs inside the
    same fused kernels as a built-in family — no host round trip.  HIP engine only (the CPU checker takes the same density
    as an ExternalTarget callback or a host KernelTarget)."""
    D: int
    source: str
    params: Optional[np.ndarray] = None
    kind: int = capi.TARGET_PLUGIN


@dataclass
class ObjectTarget:
    """User log-density as COMPILED device code — a relocatable object of `hipcc -fgpu-rdc -c` or raw amdgcn LLVM bitcode (`.bc`,
    what GPUCompiler.jl / AMDGPU.jl emit for a Julia function) that defines the C symbol of include/ahmc_user_target_object.h.
    `build_target_plugin_from_object` links it with the engine's trajectory kernels (device LTO: a bitcode density is inlined
    into the leaf loop) and the result is bound like a PluginTarget (ahmc_set_target_plugin): the fused kernels incl. the
    in-kernel warm-up, no host round trip, no per-leapfrog launch.  HIP engine only."""
    D: int
    object: str
    params: Optional[np.ndarray] = None
    kind: int = capi.TARGET_PLUGIN


@dataclass
class KernelTarget:
    """User log-density as a device KERNEL the engine launches itself (ahmc_set_target_kernel): `handle` is a hipFunction_t
    (handle_kind = capi.KERNEL_HIP_FUNCTION; what hipModuleGetFunction returns / AMDGPU.jl compiles a Julia kernel to), the
    host address of a __global__ symbol (KERNEL_HIP_SYMBOL), or — CPU checker only — a ctypes callback with the kernel's
    signature (KERNEL_HOST):  f(theta, lp, grad_neg, cols, n_cols, D, N, user)."""
    D: int
    handle: object
    handle_kind: int = capi.KERNEL_HIP_FUNCTION
    block_threads: int = 256
    chains_per_block: int = 1
    user: object = None
    kind: int = capi.TARGET_KERNEL
    params: Optional[np.ndarray] = None


@dataclass
class Hamiltonian:
    """src/hamiltonian.jl:1-20 (GaussianKinetic only, :18-20)"""
    metric: AbstractMetric
    target: object

    def __post_init__(self):
        if self.metric.size[0] != self.target.D:
            raise ArgumentError(capi.ERR_ARGUMENT, f"metric dimension {self.metric.size[0]} != target dimension {self.target.D}")


# ------------------------------------------------------------------------------------------------
# integrators (src/integrator.jl:71-209)
# ------------------------------------------------------------------------------------------------
class AbstractLeapfrog:
    kind = capi.INTEGRATOR_LEAPFROG
    param = 0.0

    def nom_step_size(self):
        return self.eps


class Leapfrog(AbstractLeapfrog):
    """src/integrator.jl:71-74; ϵ scalar or per-chain vector"""

    def __init__(self, eps):
        self.eps = np.asarray(eps, dtype=np.float64) if np.ndim(eps) else float(eps)


class JitteredLeapfrog(AbstractLeapfrog):
    """src/integrator.jl:112-156"""
    kind = capi.INTEGRATOR_JITTERED

    def __init__(self, eps0, jitter):
        self.eps = np.asarray(eps0, dtype=np.float64) if np.ndim(eps0) else float(eps0)
        self.param = float(jitter)


class TemperedLeapfrog(AbstractLeapfrog):
    """src/integrator.jl:174-209"""
    kind = capi.INTEGRATOR_TEMPERED

    def __init__(self, eps, alpha):
        self.eps = np.asarray(eps, dtype=np.float64) if np.ndim(eps) else float(eps)
        self.param = float(alpha)


# ------------------------------------------------------------------------------------------------
# trajectories / kernels (src/trajectory.jl:62-254, :414-449)
# ------------------------------------------------------------------------------------------------
class EndPointTS:
    code = capi.TS_ENDPOINT


class MultinomialTS:
    code = capi.TS_MULTINOMIAL


class SliceTS:
    code = capi.TS_SLICE


@dataclass
class FixedNSteps:
    L: int


@dataclass
class FixedIntegrationTime:
    lam: float


@dataclass
class ClassicNoUTurn:
    max_depth: int = 10
    delta_max: float = 1000.0
    code = capi.TC_CLASSIC


@dataclass
class GeneralisedNoUTurn:
    max_depth: int = 10
    delta_max: float = 1000.0
    code = capi.TC_GENERALISED


@dataclass
class StrictGeneralisedNoUTurn:
    max_depth: int = 10
    delta_max: float = 1000.0
    code = capi.TC_STRICT


_DYNAMIC = (ClassicNoUTurn, GeneralisedNoUTurn, StrictGeneralisedNoUTurn)


@dataclass
class Trajectory:
    """`Trajectory{TS}(integrator, termination_criterion)` (src/trajectory.jl:213-224)"""
    TS: type
    integrator: AbstractLeapfrog
    termination_criterion: object


class FullMomentumRefreshment:
    alpha = 0.0


@dataclass
class PartialMomentumRefreshment:
    alpha: float


class HMCKernel:
    """src/trajectory.jl:249-254"""

    def __init__(self, *args):
        if len(args) == 1:
            self.refreshment, self.tau = FullMomentumRefreshment(), args[0]
        else:
            self.refreshment, self.tau = args

    def cfg(self) -> capi.KernelCfg:
        tc = self.tau.termination_criterion
        k = capi.KernelCfg()
        k.sampler = self.tau.TS.code
        k.refresh_alpha = float(self.refreshment.alpha)
        if isinstance(tc, _DYNAMIC):
            k.nuts, k.criterion, k.max_depth, k.delta_max = 1, tc.code, tc.max_depth, tc.delta_max
        elif isinstance(tc, FixedNSteps):
            k.nuts, k.L, k.lambda_ = 0, tc.L, 0.0
        elif isinstance(tc, FixedIntegrationTime):
            k.nuts, k.L, k.lambda_ = 0, 0, tc.lam
        else:
            raise ArgumentError(capi.ERR_ARGUMENT, f"unknown termination criterion {tc!r}")
        return k


# ------------------------------------------------------------------------------------------------
# adaptors (src/adaptation/*.jl)
# ------------------------------------------------------------------------------------------------
class NoAdaptation:
    code = capi.ADAPT_NONE
    delta = 0.8


@dataclass
class StepSizeAdaptor:
    """StepSizeAdaptor(δ, integrator | ϵ) → NesterovDualAveraging (src/AdvancedHMC.jl:105-110)"""
    delta: float
    integrator: object = None
    code = capi.ADAPT_STEPSIZE


@dataclass
class MassMatrixAdaptor:
    """MassMatrixAdaptor(metric) → UnitMassMatrix / WelfordVar (src/AdvancedHMC.jl:112-118)"""
    metric: AbstractMetric
    code = capi.ADAPT_MASSMATRIX
    delta = 0.8
    estimator = 0  # AHMC_VAR_WELFORD


class PooledVar(MassMatrixAdaptor):
    """One (D,) M⁻¹ shared by all chains, estimated from all of them — and from all GPUs once the engine has a
    communicator (include/ahmc_hip.h AHMC_VAR_POOLED).  The reference has no counterpart: in
    matrix mode it resizes WelfordVar to one estimator per chain (src/adaptation/massmatrix.jl:103-121)."""
    estimator = 2  # AHMC_VAR_POOLED


class NutpieVar(MassMatrixAdaptor):
    """NutpieVar(size) (src/adaptation/massmatrix.jl:160-250): the nutpie-style diagonal estimator
    M⁻¹ = sqrt(var θ / var ∇ℓπ); a drop-in for the WelfordVar behind MassMatrixAdaptor(DiagEuclideanMetric),
    e.g. StanHMCAdaptor(NutpieVar(metric), StepSizeAdaptor(δ, lf)) (test/adaptation.jl:141-143)"""
    estimator = 1  # AHMC_VAR_NUTPIE


@dataclass
class NaiveHMCAdaptor:
    """src/adaptation/Adaptation.jl:41-64"""
    pc: MassMatrixAdaptor
    ssa: StepSizeAdaptor
    code = capi.ADAPT_NAIVE

    @property
    def delta(self):
        return self.ssa.delta


@dataclass
class StanHMCAdaptor:
    """src/adaptation/stan_adaptor.jl:52-103"""
    pc: MassMatrixAdaptor
    ssa: StepSizeAdaptor
    init_buffer: int = 75
    term_buffer: int = 50
    window_size: int = 25
    code = capi.ADAPT_STAN

    @property
    def delta(self):
        return self.ssa.delta


def stan_windows(n_adapts, init_buffer=75, term_buffer=50, window_size=25, lib: Optional[capi.CLib] = None):
    """initialize!(::StanHMCAdaptorState, ...) (src/adaptation/stan_adaptor.jl:13-50).
    Returns (window_start, window_end, window_splits)."""
    lib = lib or capi.load_hip_library()
    ws, we, ns = C.c_int64(), C.c_int64(), C.c_int32()
    splits = (C.c_int64 * 64)()
    lib.check(lib.dll.ahmc_stan_windows(init_buffer, term_buffer, window_size, n_adapts, C.byref(ws), C.byref(we),
                                        splits, 64, C.byref(ns)))
    return ws.value, we.value, [splits[i] for i in range(ns.value)]


# ------------------------------------------------------------------------------------------------
# phase point / transition carriers (src/hamiltonian.jl:88-107, src/trajectory.jl:18-23)
# ------------------------------------------------------------------------------------------------
@dataclass
class DualValue:
    value: np.ndarray
    gradient: np.ndarray


@dataclass
class PhasePoint:
    theta: np.ndarray
    r: np.ndarray
    lp: DualValue  # ℓπ: value = log density, gradient = -∇ℓπ (src/hamiltonian.jl:45-48)
    lk: DualValue  # ℓκ: value = -K(r); gradient (∂H∂r) is recomputed on demand


@dataclass
class Transition:
    z: PhasePoint
    stat: dict = field(default_factory=dict)


def neg_energy(z: PhasePoint):
    return z.lp.value + z.lk.value


def energy(z: PhasePoint):
    return -neg_energy(z)


# ------------------------------------------------------------------------------------------------
# Engine: one C context
# ------------------------------------------------------------------------------------------------
class Engine:
    """Binds a Hamiltonian and N chains to one `ahmc_ctx`.

    `lib=None` opens the HIP engine (and fails if it is not built).  `stream` is an optional
    hipStream_t (e.g. `torch.cuda.Stream().cuda_stream`) for HIP-event timing."""

    def __init__(self, h: Hamiltonian, n_chains: int, dtype=np.float64, rng=0, lib: Optional[capi.CLib] = None,
                 device: int = 0, stream: int = 0):
        self.lib = lib or capi.load_hip_library()
        self.h = h
        self.D, self.N = int(h.target.D), int(n_chains)
        self.dtype = np.dtype(dtype)
        self._ctx = C.c_void_p()
        self.lib.check(self.lib.dll.ahmc_create(device, capi.dtype_code(self.dtype), self.D, self.N,
                                                C.c_void_p(stream or None), C.byref(self._ctx)))
        self._external = isinstance(h.target, ExternalTarget)
        self.set_target(h.target)
        self.set_metric(h.metric)
        self.seed(rng)

    # -- plumbing --
    def _call(self, name, *args):
        self.lib.check(getattr(self.lib.dll, name)(self._ctx, *args), self._ctx)

    def close(self):
        if self._ctx:
            self.lib.dll.ahmc_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _mat(self, a, name):
        a = np.asarray(a)
        if a.ndim == 1:
            a = a.reshape(-1, 1)
        if a.shape != (self.D, self.N):
            raise ArgumentError(capi.ERR_ARGUMENT, f"{name} has size {a.shape}, expected {(self.D, self.N)}")
        return np.asfortranarray(a, dtype=self.dtype)

    def _out(self, vec=False):
        return np.empty((self.N,) if vec else (self.D, self.N), dtype=self.dtype, order="F")

    def _shape_out(self, a):
        return a[:, 0].copy() if (self.N == 1 and getattr(self, "_vector_mode", False)) else a

    # -- configuration --
    def set_target(self, target):
        p = None if target.params is None else np.ascontiguousarray(target.params, dtype=self.dtype)
        if isinstance(target, PluginTarget):
            from .build import build_target_plugin

            so = build_target_plugin(target.source, self.dtype, self.info("group_lanes"), self.info("elems_per_lane"))
            self._call("ahmc_set_target_plugin", so.encode(), capi.as_ptr(p), 0 if p is None else p.size)
            return
        if isinstance(target, ObjectTarget):
            from .build import build_target_plugin_from_object

            so = build_target_plugin_from_object(target.object, self.dtype, self.info("group_lanes"), self.info("elems_per_lane"),
                                                 n_params=-1 if p is None else p.size)
            self._call("ahmc_set_target_plugin", so.encode(), capi.as_ptr(p), 0 if p is None else p.size)
            return
        if isinstance(target, KernelTarget):
            h = target.handle
            self._kernel_keepalive = (h, target.user)  # (a ctypes callback must outlive the context's use of it)
            hp = C.cast(h, C.c_void_p) if not isinstance(h, (int, C.c_void_p)) else (C.c_void_p(h) if isinstance(h, int) else h)
            up = target.user if isinstance(target.user, C.c_void_p) else C.c_void_p(target.user)
            self._call("ahmc_set_target_kernel", int(target.handle_kind), hp, int(target.block_threads), int(target.chains_per_block), up)
            return
        self._call("ahmc_set_target", target.kind, capi.as_ptr(p), 0 if p is None else p.size)

    def set_metric(self, metric):
        if metric.kind == capi.METRIC_UNIT:
            self._call("ahmc_set_metric", metric.kind, None, 0)
        else:
            M = np.asfortranarray(metric.Minv, dtype=self.dtype)
            if metric.kind == capi.METRIC_DIAG and M.shape[0] != self.D:
                raise ArgumentError(capi.ERR_ARGUMENT, f"AxesMismatch: M⁻¹ has size {M.shape} but r has first axis {self.D}")
            self._call("ahmc_set_metric", metric.kind, capi.as_ptr(M), M.size)
        self.metric_kind = metric.kind

    def get_metric(self):
        if self.metric_kind == capi.METRIC_UNIT:
            return None
        # size is whatever was set last (D, D*N or D*D): probe via a D*N then D buffer
        for shape in ((self.D, self.N), (self.D,), (self.D, self.D)):
            out = np.empty(shape, dtype=self.dtype, order="F")
            code = self.lib.dll.ahmc_get_metric(self._ctx, capi.as_ptr(out), out.size)
            if code == capi.OK:
                return out
        self.lib.check(code, self._ctx)

    def set_integrator(self, lf: AbstractLeapfrog):
        eps = np.atleast_1d(np.asarray(lf.eps, dtype=self.dtype))
        if eps.size not in (1, self.N):
            raise ArgumentError(capi.ERR_ARGUMENT, f"step size vector has length {eps.size}, expected 1 or {self.N}")
        self._call("ahmc_set_stepsize", capi.as_ptr(eps), eps.size)
        self._call("ahmc_set_integrator", lf.kind, float(lf.param))

    def get_stepsize(self):
        out = self._out(vec=True)
        self._call("ahmc_get_stepsize", capi.as_ptr(out))
        return out

    def seed(self, rng, iteration=None):
        spec, stride = _rng_spec(rng, self.N)
        self._call("ahmc_seed", spec.seed, spec.chain_offset, stride, spec.iteration if iteration is None else iteration)

    # -- phase point --
    def set_position(self, theta, r=None):
        """phasepoint(h, θ, r) (src/hamiltonian.jl:115-119)"""
        self._vector_mode = np.ndim(theta) == 1
        th = self._mat(theta, "θ")
        if self._external:
            lp, grad = self.h.target.fn(th)
            rr = self._mat(r, "r") if r is not None else np.zeros_like(th)
            # the converted arrays are bound to names: the C call reads them (as_ptr also keeps them alive)
            lp_a = np.ascontiguousarray(lp, dtype=self.dtype).reshape(self.N)
            g_a = self._mat(-np.asarray(grad), "∇ℓπ")
            self._call("ahmc_set_phasepoint", capi.as_ptr(th), capi.as_ptr(rr), capi.as_ptr(lp_a), capi.as_ptr(g_a))
            return
        rr = None if r is None else self._mat(r, "r")
        self._call("ahmc_set_position", capi.as_ptr(th), capi.as_ptr(rr))

    def phasepoint(self) -> PhasePoint:
        th, r, g = self._out(), self._out(), self._out()
        lp, lk = self._out(True), self._out(True)
        self._call("ahmc_get_phasepoint", capi.as_ptr(th), capi.as_ptr(r), capi.as_ptr(lp), capi.as_ptr(g), capi.as_ptr(lk))
        if self.N == 1 and getattr(self, "_vector_mode", False):
            return PhasePoint(th[:, 0], r[:, 0], DualValue(lp[0], g[:, 0]), DualValue(lk[0], None))
        return PhasePoint(th, r, DualValue(lp, g), DualValue(lk, None))

    def theta(self):
        th = self._out()
        self._call("ahmc_get_phasepoint", capi.as_ptr(th), None, None, None, None)
        return self._shape_out(th)

    def set_ref_compat(self, on=True):
        """the reference's matrix-mode early exit (src/integrator.jl:252-258: every chain stops at the first step after which ANY chain is
        non-finite) for `step` and static EndPointTS transitions — off by default (each chain stops at its own first non-finite point)"""
        self._call("ahmc_set_ref_compat", 1 if on else 0)

    def refresh(self, refreshment=None):
        """refresh(rng, refreshment, h, z) (src/hamiltonian.jl:213-254)"""
        self._call("ahmc_refresh_momentum", float(getattr(refreshment, "alpha", 0.0)))

    def step(self, n_steps=1):
        """step(lf, h, z, n_steps) (src/integrator.jl:216-265)"""
        if self._external:
            n = abs(int(n_steps))
            fwd = 1 if n_steps > 0 else 0
            theta_view = self._out()
            for i in range(1, n + 1):
                self._call("ahmc_lf_pre", fwd, i, n)
                self._call("ahmc_get_phasepoint", capi.as_ptr(theta_view), None, None, None, None)
                lp, grad = self.h.target.fn(theta_view)
                lp_a = np.ascontiguousarray(lp, dtype=self.dtype).reshape(self.N)
                g_a = self._mat(-np.asarray(grad), "∇ℓπ")
                self._call("ahmc_lf_post", fwd, i, n, capi.as_ptr(lp_a), capi.as_ptr(g_a))
            return
        self._call("ahmc_leapfrog", int(n_steps))

    # -- transitions --
    def transition(self, kernel: HMCKernel):
        """transition(rng, h, κ, z) (src/sampler.jl:48-58)"""
        k = kernel.cfg()
        if self._external:
            self._call("ahmc_ext_begin", C.byref(k), 1)
            self._ext_drive()
            return
        if k.refresh_alpha != 0.0:
            # HMCKernel(PartialMomentumRefreshment(α), τ) (src/trajectory.jl:249-254, src/hamiltonian.jl:243-254): the refreshment is part of
            # the sample loop's kernel configuration at the boundary — ONE iteration of that loop is this transition (the iteration counter,
            # and with it every variate, continues as for the two calls below; the running accumulators count the draw, as for any kept one)
            self._call("ahmc_sample_from", C.byref(k), 1, 1, 0, 0, None)
            return
        if k.nuts:
            self._call("ahmc_nuts_transition", k.max_depth, k.delta_max, k.criterion, k.sampler)
        else:
            self._call("ahmc_hmc_transition", k.L, k.lambda_, k.sampler)

    def stats(self, fields: Optional[Sequence[str]] = None) -> dict:
        out = {}
        for name in fields or capi.STAT_FIELDS:
            fid, is_int = capi.STAT_FIELDS[name]
            buf = np.empty(self.N, dtype=np.int32 if is_int else self.dtype)
            self._call("ahmc_get_stat", fid, capi.as_ptr(buf))
            out[name] = buf
        for b in ("is_accept", "numerical_error"):
            if b in out:
                out[b] = out[b].astype(bool)
        return out

    def find_good_stepsize(self, initial_step_size=0.1, max_n_iters=100):
        """find_good_stepsize(rng, h, θ) per chain (src/trajectory.jl:768-837)"""
        if self._external:
            self._call("ahmc_ext_find_good_stepsize_begin", float(initial_step_size), int(max_n_iters))
            self._ext_drive()
        else:
            self._call("ahmc_find_good_stepsize", float(initial_step_size), int(max_n_iters))
        return self.get_stepsize()

    # -- external target: the ask / tell loop (include/ahmc_hip.h, ahmc_ext_*) --
    def _ext_drive(self):
        """Serve the engine's evaluation requests with the user's `fn` until the run started by
        ahmc_ext_begin / ahmc_ext_find_good_stepsize_begin is complete.  Returns the number of
        round trips (= evaluations of `fn`)."""
        n = C.c_int64()
        chains = np.empty(self.N, dtype=np.int32)
        theta = self._out()
        lp_buf = np.zeros(self.N, dtype=self.dtype)
        g_buf = np.zeros((self.D, self.N), dtype=self.dtype, order="F")
        trips = 0
        try:
            while True:
                self._call("ahmc_ext_pending", C.byref(n), capi.as_ptr(chains), capi.as_ptr(theta))
                if n.value == 0:
                    return trips
                idx = chains[:n.value]
                if self.h.target.takes_chains:
                    lp, grad = self.h.target.fn(theta, idx)
                else:
                    lp, grad = self.h.target.fn(theta)
                lp_buf[idx] = np.asarray(lp, dtype=self.dtype).reshape(self.N)[idx]
                g_buf[:, idx] = -np.asarray(grad, dtype=self.dtype).reshape(self.D, self.N)[:, idx]
                self._call("ahmc_ext_advance", capi.as_ptr(lp_buf), capi.as_ptr(g_buf))
                trips += 1
        except BaseException:
            self.lib.dll.ahmc_ext_cancel(self._ctx)
            raise

    # -- adaptation --
    def adaptor_init(self, adaptor):
        ib, tb, ws = (getattr(adaptor, "init_buffer", 75), getattr(adaptor, "term_buffer", 50),
                      getattr(adaptor, "window_size", 25))
        pc = adaptor if isinstance(adaptor, MassMatrixAdaptor) else getattr(adaptor, "pc", None)
        self._call("ahmc_set_var_estimator", int(getattr(pc, "estimator", 0)))
        self._call("ahmc_adaptor_init", adaptor.code, float(adaptor.delta), ib, tb, ws)

    def adapt(self, i, n_adapts, theta=None, alpha=None, grad=None):
        """adapt!(h, κ, adaptor, i, n_adapts, z_or_θ, α) (src/sampler.jl:72-90); θ/α default to the
        context's position and the last transition's acceptance_rate; `grad` = z.ℓπ.gradient makes the
        argument a phase point (needed by NutpieVar, src/adaptation/massmatrix.jl:238-243)"""
        th = None if theta is None else self._mat(theta, "θ")
        al = None if alpha is None else np.ascontiguousarray(np.broadcast_to(np.asarray(alpha, dtype=self.dtype), (self.N,)))
        if grad is None:
            self._call("ahmc_adapt", int(i), int(n_adapts), capi.as_ptr(th), capi.as_ptr(al))
        else:
            g = self._mat(grad, "∇")
            self._call("ahmc_adapt_point", int(i), int(n_adapts), capi.as_ptr(th), capi.as_ptr(g), capi.as_ptr(al))

    # -- bulk driver --
    def run(self, kernel: HMCKernel, n_samples, n_adapts=0, drop_warmup=False, samples_out=None, i_first=1):
        """The whole `sample` loop enqueued by one C call (no host synchronisation inside): iterations i_first …
        n_samples (i_first > 1 continues a checkpointed run, ahmc_sample_from).  With an ExternalTarget the loop runs
        here, one ask / tell transition + adapt! per iteration."""
        k = kernel.cfg()
        if self._external:
            if samples_out is not None:
                raise capi.UnsupportedError(capi.ERR_UNSUPPORTED, "samples_out with an ExternalTarget: use sample()")
            for i in range(int(i_first), int(n_samples) + 1):
                self.transition(kernel)
                self.adapt(i, int(n_adapts))
            return
        self._call("ahmc_sample_from", C.byref(k), int(i_first), int(n_samples), int(n_adapts), 1 if drop_warmup else 0,
                   capi.as_ptr(samples_out))

    def sync(self):
        self._call("ahmc_sync")

    # -- checkpoint / resume (HMCState, src/abstractmcmc.jl:11-27) --
    def get_state(self) -> dict:
        """Everything a resumed run needs: the phase point, h's metric, κ's nominal step sizes, the adaptor, the RNG
        counter.  `Engine.set_state` on a fresh engine of the same Hamiltonian continues the run bit for bit."""
        st = capi.AdaptorState()
        self._call("ahmc_get_adaptor_state", C.byref(st), None, None)
        da = np.empty((5, self.N), dtype=self.dtype) if st.has_da else None
        wv = np.empty((st.n_welford, self.N, self.D), dtype=self.dtype) if st.n_welford else None  # C-order view of (n, D, N) column-major
        self._call("ahmc_get_adaptor_state", C.byref(st), capi.as_ptr(da), capi.as_ptr(wv))
        z = self.phasepoint() if not getattr(self, "_vector_mode", False) else None
        if z is None:
            self._vector_mode = False
            z = self.phasepoint()
            self._vector_mode = True
        ntr = C.c_int64()
        acc = {"n_steps": np.empty(self.N, dtype=np.int64), "n_divergent": np.empty(self.N, dtype=np.int64),
               "sum_theta": np.empty((self.N, self.D), dtype=self.dtype), "sumsq_theta": np.empty((self.N, self.D), dtype=self.dtype),
               "energy_sums": np.empty((5, self.N), dtype=self.dtype)}
        self._call("ahmc_get_accum_state", C.byref(ntr), capi.as_ptr(acc["n_steps"]), capi.as_ptr(acc["n_divergent"]), capi.as_ptr(acc["sum_theta"]),
                   capi.as_ptr(acc["sumsq_theta"]), capi.as_ptr(acc["energy_sums"]))
        acc["n_transitions"] = ntr.value
        return {"adaptor": {k: getattr(st, k) for k, _ in capi.AdaptorState._fields_}, "da": da, "welford": wv,
                "theta": z.theta, "r": z.r, "lp": z.lp.value, "grad": z.lp.gradient,
                "metric": self.get_metric(), "metric_kind": self.metric_kind, "stepsize": self.get_stepsize(),
                "stepsize_scalar": bool(self.info("stepsize_scalar")), "accum": acc}

    def set_state(self, state: dict):
        if state["metric"] is not None:
            self._call("ahmc_set_metric", state["metric_kind"], capi.as_ptr(np.asfortranarray(state["metric"], dtype=self.dtype)), state["metric"].size)
        eps = np.ascontiguousarray(state["stepsize"], dtype=self.dtype)
        if state.get("stepsize_scalar"):   # ONE nominal ϵ stays one (κ's integrator is restored as it was: FixedIntegrationTime needs it, Q6)
            eps = eps[:1].copy()
        self._call("ahmc_set_stepsize", capi.as_ptr(eps), eps.size)
        th, r = self._mat(state["theta"], "θ"), self._mat(state["r"], "r")
        lp = np.ascontiguousarray(state["lp"], dtype=self.dtype).reshape(self.N)
        g = self._mat(state["grad"], "∇ℓπ")
        self._call("ahmc_set_phasepoint", capi.as_ptr(th), capi.as_ptr(r), capi.as_ptr(lp), capi.as_ptr(g))
        st = capi.AdaptorState(**state["adaptor"])
        self._call("ahmc_set_adaptor_state", C.byref(st), capi.as_ptr(state["da"]), capi.as_ptr(state["welford"]))
        acc = state.get("accum")
        if acc is not None:  # the running accumulators (Σθ, Σθ², Σ n_steps, the energy sums of EBFMI) are part of the checkpoint
            cont = lambda a, dt: np.ascontiguousarray(a, dtype=dt)  # noqa: E731
            self._call("ahmc_set_accum_state", int(acc["n_transitions"]), capi.as_ptr(cont(acc["n_steps"], np.int64)),
                       capi.as_ptr(cont(acc["n_divergent"], np.int64)), capi.as_ptr(cont(acc["sum_theta"], self.dtype)),
                       capi.as_ptr(cont(acc["sumsq_theta"], self.dtype)), capi.as_ptr(cont(acc["energy_sums"], self.dtype)))

    # -- multi-GPU: the final gather through the C ABI (RCCL inside) --
    def comm_unique_id(self) -> bytes:
        buf = (C.c_char * capi.UNIQUE_ID_BYTES)()
        self.lib.check(self.lib.dll.ahmc_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id: bytes, n_ranks: int, rank: int):
        buf = (C.c_char * capi.UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        self._call("ahmc_comm_init", buf, int(n_ranks), int(rank))

    def comm_info(self) -> dict:
        """what the communicator's own all-reduces reported when it was attached (ahmc_comm_info)"""
        v = [C.c_int64() for _ in range(4)]
        self._call("ahmc_comm_info", *[C.byref(x) for x in v])
        return dict(zip(("ranks_seen", "chains_total", "chains_min", "chains_max"), (x.value for x in v)))

    def reserve(self, kernel: "HMCKernel", n_samples: int):
        """announce a sampling run of n_samples transitions: its launch buffers are reserved now (ahmc_sample_reserve)"""
        if not self._external:
            k = kernel.cfg()
            self._call("ahmc_sample_reserve", C.byref(k), int(n_samples))

    def gather_moments(self) -> dict:
        mean, var = np.empty(self.D), np.empty(self.D)
        n, tot, ndiv = C.c_int64(), C.c_int64(), C.c_int64()
        self._call("ahmc_gather_moments", capi.as_ptr(mean), capi.as_ptr(var), C.byref(n), C.byref(tot), C.byref(ndiv))
        return {"mean": mean, "var": var, "n_draws": n.value, "total_n_steps": tot.value, "n_divergent": ndiv.value}

    def gather_state(self, theta_all_ptr):
        """ncclAllGather of θ into the caller's DEVICE buffer (D, N, n_ranks)"""
        self._call("ahmc_gather_state", theta_all_ptr if isinstance(theta_all_ptr, C.c_void_p) else capi.as_ptr(theta_all_ptr))

    # -- diagnostics computed where the data is --
    def ebfmi(self):
        """EBFMI (src/diagnosis.jl:1-3) per chain over the kept transitions of the last `run`"""
        out = self._out(vec=True)
        self._call("ahmc_ebfmi", capi.as_ptr(out))
        return out

    def ess(self, draws_ptr, n_draws):
        """ESS of every (dimension, chain) series of the (D, N, n_draws) device buffer `run(samples_out=…)` filled"""
        out = self._out()
        self._call("ahmc_ess", draws_ptr if isinstance(draws_ptr, C.c_void_p) else capi.as_ptr(draws_ptr), int(n_draws), capi.as_ptr(out))
        return out

    @property
    def stream(self):
        return self.lib.dll.ahmc_stream(self._ctx)

    def accum(self, moments=True):
        tot, ntr, ndiv = C.c_int64(), C.c_int64(), C.c_int64()
        s1 = self._out() if moments else None
        s2 = self._out() if moments else None
        self._call("ahmc_get_accum", C.byref(tot), C.byref(ntr), C.byref(ndiv), capi.as_ptr(s1), capi.as_ptr(s2))
        return {"total_n_steps": tot.value, "n_transitions": ntr.value, "n_divergent": ndiv.value,
                "sum_theta": s1, "sumsq_theta": s2}

    def reset_accum(self):
        self._call("ahmc_reset_accum")

    INFO = {"group_lanes": 0, "elems_per_lane": 1, "nuts_launches": 2, "nuts_batch": 3, "iteration": 4, "nuts_kernel_ns": 5,
            "nuts_warm_launches": 6, "nuts_warm_kernel_ns": 7, "dense_gemm_launches": 8, "dense_gemm_small_launches": 9, "dense_pipelines": 10,
            "dense_pool": 11, "nuts_draw_batch": 12, "dense_epoch_launches": 13, "stepsize_scalar": 14}

    def info(self, key):
        """engine introspection (ahmc_get_info): thread geometry, NUTS launch count / batch, iteration"""
        v = C.c_int64()
        self._call("ahmc_get_info", self.INFO[key], C.byref(v))
        return v.value


# ------------------------------------------------------------------------------------------------
# free functions with the reference's names
# ------------------------------------------------------------------------------------------------
def find_good_stepsize(rng, h: Hamiltonian, theta, initial_step_size=0.1, max_n_iters=100, dtype=np.float64,
                       lib=None):
    """src/trajectory.jl:768-852.  θ (D,) → scalar ϵ; θ (D,N) → per-chain ϵ (N,)"""
    theta = np.asarray(theta)
    N = 1 if theta.ndim == 1 else theta.shape[1]
    eng = Engine(h, N, dtype=dtype, rng=rng, lib=lib)
    try:
        eng.set_position(theta)
        eps = eng.find_good_stepsize(initial_step_size, max_n_iters)
    finally:
        eng.close()
    return float(eps[0]) if theta.ndim == 1 else eps


def sample(rng, h: Hamiltonian, kernel: HMCKernel, theta, n_samples: int, adaptor=None, n_adapts: Optional[int] = None,
           drop_warmup=False, verbose=False, progress=False, dtype=None, lib=None, device=0):
    """sample(rng, h, κ, θ, n_samples, adaptor, n_adapts; drop_warmup) (src/sampler.jl:159-248).

    Returns `(θs, stats)`: a list of (D, N) arrays (or (D,) for vector θ) and a list of stat
    dicts with the reference's field names plus `is_adapt` (:193).  One transition + adapt! per
    iteration is issued through the C ABI, exactly the loop of :182-228."""
    adaptor = adaptor if adaptor is not None else NoAdaptation()
    if n_adapts is None:
        n_adapts = min(n_samples // 10, 1000)
    if drop_warmup and isinstance(adaptor, NoAdaptation):
        raise AssertionError("Cannot drop warmup samples if there is no adaptation phase.")  # :172
    theta = np.asarray(theta)
    dtype = dtype or (theta.dtype if theta.dtype in (np.float32, np.float64) else np.float64)
    N = 1 if theta.ndim == 1 else theta.shape[1]
    eng = Engine(h, N, dtype=dtype, rng=rng, lib=lib, device=device)
    try:
        eng.set_integrator(kernel.tau.integrator)
        eng.set_position(theta)  # sample_init (:36-46)
        eng.adaptor_init(adaptor)
        k = kernel.cfg()
        thetas, stats = [], []
        one = capi.KernelCfg.from_buffer_copy(k)
        for i in range(1, n_samples + 1):
            if eng._external:
                eng.transition(kernel)  # the same transition with the user's ∂ℓπ∂θ served through ahmc_ext_*
            else:
                eng._call("ahmc_sample", C.byref(one), 1, 0, 0, None)  # transition(rng, h, κ, t.z) (:184)
            st = eng.stats()
            isadapted = i <= n_adapts and not isinstance(adaptor, NoAdaptation)
            eng.adapt(i, n_adapts)
            st["is_adapt"] = isadapted
            if not drop_warmup or i > n_adapts:
                thetas.append(eng.theta())
                stats.append(st)
        return thetas, stats
    finally:
        eng.close()


def EBFMI(energies):
    """src/diagnosis.jl:1-3; energies: sequence over iterations of scalars or (N,) arrays"""
    E = np.asarray(energies, dtype=np.float64)
    num = np.mean(np.diff(E, axis=0) ** 2, axis=0)
    return num / np.var(E, axis=0, ddof=1)
