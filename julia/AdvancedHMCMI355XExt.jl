# AdvancedHMCMI355XExt.jl — the reference-side binding of libahmc_hip.so (include/ahmc_hip.h).
#
# A package extension in the style of ext/AdvancedHMCCUDAExt.jl (which specialises `refresh` and
# `mh_accept_ratio` on `CuArray`, :6-34).  This one specialises the WHOLE transition on a device
# handle type, so that `sample(rng, h, κ, θ, n, adaptor, n_adapts)` (src/sampler.jl:159-248) and
# `AbstractMCMC.step` (src/abstractmcmc.jl:131-207) keep their signatures and run every
# transition + adapt! of the sampler-vec path as chain-batched HIP kernels.
#
# NOT EXECUTED in the build environment (no Julia there): it documents the `ccall` layer a
# maintainer adds; the Python mirror in advancedhmc.jl_amd/api.py drives the same ABI in the tests.
module AdvancedHMCMI355XExt

using AdvancedHMC
using AdvancedHMC: Hamiltonian, HMCKernel, Trajectory, PhasePoint, DualValue, Transition,
    AbstractMetric, UnitEuclideanMetric, DiagEuclideanMetric, Leapfrog, JitteredLeapfrog,
    TemperedLeapfrog, FixedNSteps, FixedIntegrationTime, EndPointTS, MultinomialTS, SliceTS,
    ClassicNoUTurn, GeneralisedNoUTurn, StrictGeneralisedNoUTurn, FullMomentumRefreshment,
    PartialMomentumRefreshment, step_size, nom_step_size
using AdvancedHMC.Adaptation: AbstractAdaptor, NoAdaptation, StepSizeAdaptor, MassMatrixAdaptor,
    NaiveHMCAdaptor, StanHMCAdaptor
using Random: AbstractRNG

const LIB = get(ENV, "AHMC_HIP_LIB", "libahmc_hip.so")

# --- enums of include/ahmc_hip.h -----------------------------------------------------------------
const F32, F64 = Cint(0), Cint(1)
const METRIC_UNIT, METRIC_DIAG, METRIC_DENSE = Cint(0), Cint(1), Cint(2)
const TARGET_ISO_GAUSS, TARGET_DIAG_GAUSS, TARGET_FUNNEL, TARGET_HIER_GAUSS = Cint.(0:3)
const TARGET_DENSE_GAUSS, TARGET_EXTERNAL, TARGET_PLUGIN, TARGET_KERNEL = Cint(4), Cint(5), Cint(6), Cint(7)
const KERNEL_HIP_FUNCTION, KERNEL_HIP_SYMBOL = Cint(0), Cint(1)
const VAR_WELFORD, VAR_NUTPIE, VAR_POOLED = Cint(0), Cint(1), Cint(2)
const TS_ENDPOINT, TS_MULTINOMIAL, TS_SLICE = Cint.(0:2)
const TC_CLASSIC, TC_GENERALISED, TC_STRICT = Cint.(0:2)
const ADAPT_NONE, ADAPT_STEPSIZE, ADAPT_MASSMATRIX, ADAPT_NAIVE, ADAPT_STAN = Cint.(0:4)

struct AHMCError <: Exception
    code::Int
    msg::String
end

function check(ctx, code)
    code == 0 && return nothing
    msg = unsafe_string(ccall((:ahmc_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx))
    code == 1 && throw(ArgumentError(msg))   # AHMC_ERR_ARGUMENT ↔ Julia ArgumentError
    throw(AHMCError(code, msg))
end

"Built-in log-density families the kernels evaluate in registers (AHMC_TARGET_*)."
abstract type DeviceTarget end
struct IsoGaussian <: DeviceTarget end
struct DiagGaussian{T} <: DeviceTarget
    m::Vector{T}
    s::Vector{T}
end
struct Funnel <: DeviceTarget end
struct HierGaussian <: DeviceTarget end
"The Hamiltonian's own `∂ℓπ∂θ` closure (any LogDensityProblems model, src/AdvancedHMC.jl:163-186): evaluated in Julia,
served to the engine through the ask / tell calls `ahmc_ext_*`."
struct UserDensity <: DeviceTarget end

"""
    MI355XChains{T}

Device-resident phase point of N chains: the handle type the extension dispatches on (the role
`CuArray` plays in ext/AdvancedHMCCUDAExt.jl).  Owns one `ahmc_ctx`.
"""
mutable struct MI355XChains{T<:AbstractFloat}
    ctx::Ptr{Cvoid}
    D::Int
    N::Int
    user_density::Bool   # target = UserDensity(): transitions go through ahmc_ext_* with h.∂ℓπ∂θ
    function MI355XChains{T}(D::Int, N::Int; device::Int=0) where {T}
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        code = ccall((:ahmc_create, LIB), Cint, (Cint, Cint, Int64, Int64, Ptr{Cvoid}, Ptr{Ptr{Cvoid}}),
                     device, T === Float32 ? F32 : F64, D, N, C_NULL, ref)
        code == 0 || throw(AHMCError(code, unsafe_string(ccall((:ahmc_last_error, LIB), Cstring, (Ptr{Cvoid},), C_NULL))))
        z = new{T}(ref[], D, N, false)
        finalizer(z -> ccall((:ahmc_destroy, LIB), Cint, (Ptr{Cvoid},), z.ctx), z)
        return z
    end
end

# --- configuration ----------------------------------------------------------------------------------
set_target!(z::MI355XChains, ::IsoGaussian) =
    check(z.ctx, ccall((:ahmc_set_target, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64), z.ctx, TARGET_ISO_GAUSS, C_NULL, 0))
set_target!(z::MI355XChains, ::Funnel) =
    check(z.ctx, ccall((:ahmc_set_target, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64), z.ctx, TARGET_FUNNEL, C_NULL, 0))
set_target!(z::MI355XChains, ::HierGaussian) =
    check(z.ctx, ccall((:ahmc_set_target, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64), z.ctx, TARGET_HIER_GAUSS, C_NULL, 0))
function set_target!(z::MI355XChains, ::UserDensity)
    check(z.ctx, ccall((:ahmc_set_target, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64), z.ctx, TARGET_EXTERNAL, C_NULL, 0))
    z.user_density = true
end
function set_target!(z::MI355XChains{T}, t::DiagGaussian) where {T}
    p = T[t.m; t.s]
    check(z.ctx, ccall((:ahmc_set_target, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{T}, Int64), z.ctx, TARGET_DIAG_GAUSS, p, length(p)))
end

set_metric!(z::MI355XChains, ::UnitEuclideanMetric) =
    check(z.ctx, ccall((:ahmc_set_metric, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Int64), z.ctx, METRIC_UNIT, C_NULL, 0))
function set_metric!(z::MI355XChains{T}, m::DiagEuclideanMetric) where {T}
    M = convert(Array{T}, m.M⁻¹)   # (D,) or (D, N): column-major, passed as is
    check(z.ctx, ccall((:ahmc_set_metric, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{T}, Int64), z.ctx, METRIC_DIAG, M, length(M)))
end

function set_metric!(z::MI355XChains{T}, m::DenseEuclideanMetric) where {T}
    M = convert(Matrix{T}, m.M⁻¹)   # (D, D) column-major, shared by all chains; runs on the MFMA engine
    check(z.ctx, ccall((:ahmc_set_metric, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{T}, Int64), z.ctx, METRIC_DENSE, M, length(M)))
end
# ℓπ = -½ θᵀPθ with the precision P applied as a GEMM (SURVEY §8d cfg4)
function set_dense_gaussian_target!(z::MI355XChains{T}, P::AbstractMatrix) where {T}
    Pm = convert(Matrix{T}, P)
    check(z.ctx, ccall((:ahmc_set_target, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{T}, Int64), z.ctx, TARGET_DENSE_GAUSS, Pm, length(Pm)))
end

function set_integrator!(z::MI355XChains{T}, lf) where {T}
    ϵ = T.(vcat(nom_step_size(lf)))
    check(z.ctx, ccall((:ahmc_set_stepsize, LIB), Cint, (Ptr{Cvoid}, Ptr{T}, Int64), z.ctx, ϵ, length(ϵ)))
    kind, param = lf isa JitteredLeapfrog ? (Cint(1), Float64(lf.jitter)) :
                  lf isa TemperedLeapfrog ? (Cint(2), Float64(lf.α)) : (Cint(0), 0.0)
    check(z.ctx, ccall((:ahmc_set_integrator, LIB), Cint, (Ptr{Cvoid}, Cint, Cdouble), z.ctx, kind, param))
end

seed!(z::MI355XChains, seed::Integer; chain_offset=0, iteration=0) =
    check(z.ctx, ccall((:ahmc_seed, LIB), Cint, (Ptr{Cvoid}, UInt64, UInt64, UInt64, UInt64), z.ctx, seed, chain_offset, 1, iteration))

# phasepoint(h, θ, r) (src/hamiltonian.jl:115-119): θ::Matrix{T} is passed zero-copy via pointer(θ)
function set_position!(z::MI355XChains{T}, θ::AbstractMatrix{T}) where {T}
    size(θ) == (z.D, z.N) || throw(ArgumentError("θ has size $(size(θ)), expected $((z.D, z.N))"))
    check(z.ctx, ccall((:ahmc_set_position, LIB), Cint, (Ptr{Cvoid}, Ptr{T}, Ptr{T}), z.ctx, θ, C_NULL))
end

function AdvancedHMC.PhasePoint(z::MI355XChains{T}) where {T}
    θ, r, g = (Matrix{T}(undef, z.D, z.N) for _ in 1:3)
    ℓπ, ℓκ = Vector{T}(undef, z.N), Vector{T}(undef, z.N)
    check(z.ctx, ccall((:ahmc_get_phasepoint, LIB), Cint, (Ptr{Cvoid}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}), z.ctx, θ, r, ℓπ, g, ℓκ))
    return PhasePoint(θ, r, DualValue(ℓπ, g), DualValue(ℓκ, similar(g)))
end

function getstat(z::MI355XChains{T}, field::Integer, ::Type{S}) where {T,S}
    out = Vector{S}(undef, z.N)
    check(z.ctx, ccall((:ahmc_get_stat, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{S}), z.ctx, field, out))
    return out
end

sampler_code(::Type{EndPointTS}) = TS_ENDPOINT
sampler_code(::Type{<:MultinomialTS}) = TS_MULTINOMIAL
sampler_code(::Type{<:SliceTS}) = TS_SLICE
criterion_code(::ClassicNoUTurn) = TC_CLASSIC
criterion_code(::GeneralisedNoUTurn) = TC_GENERALISED
criterion_code(::StrictGeneralisedNoUTurn) = TC_STRICT

# --- user log-densities: ask / tell (include/ahmc_hip.h, ahmc_ext_*) ------------------------------------
# PhasePoint(θ, r, ℓπ, ℓκ) with the caches computed by the Hamiltonian's own closure (src/hamiltonian.jl:115-119)
function set_position!(z::MI355XChains{T}, h::Hamiltonian, θ::AbstractMatrix{T}) where {T}
    size(θ) == (z.D, z.N) || throw(ArgumentError("θ has size $(size(θ)), expected $((z.D, z.N))"))
    ℓπ, ∇ℓπ = h.∂ℓπ∂θ(θ)                                  # ((N,), (D,N)) — test/common.jl:64-74
    r = zeros(T, z.D, z.N)
    check(z.ctx, ccall((:ahmc_set_phasepoint, LIB), Cint, (Ptr{Cvoid}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}),
                       z.ctx, Matrix{T}(θ), r, Vector{T}(ℓπ), Matrix{T}(-∇ℓπ)))
end

# Serve the engine's evaluation requests with h.∂ℓπ∂θ until the run started by ahmc_ext_begin /
# ahmc_ext_find_good_stepsize_begin is complete.  Only the pending columns are read by the engine, so a closure
# that evaluates all N columns (the sampler-vec convention) is served as is.
function ext_drive!(z::MI355XChains{T}, h::Hamiltonian) where {T}
    n = Ref{Int64}(0)
    θ = Matrix{T}(undef, z.D, z.N)
    try
        while true
            check(z.ctx, ccall((:ahmc_ext_pending, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int32}, Ptr{T}), z.ctx, n, C_NULL, θ))
            n[] == 0 && return nothing
            ℓπ, ∇ℓπ = h.∂ℓπ∂θ(θ)                          # ∂H∂θ(h, θ) — src/hamiltonian.jl:45-48
            check(z.ctx, ccall((:ahmc_ext_advance, LIB), Cint, (Ptr{Cvoid}, Ptr{T}, Ptr{T}), z.ctx, Vector{T}(ℓπ), Matrix{T}(-∇ℓπ)))
        end
    catch
        ccall((:ahmc_ext_cancel, LIB), Cint, (Ptr{Cvoid},), z.ctx)
        rethrow()
    end
end

kernel_cfg(κ::HMCKernel) = begin
    tc = κ.τ.termination_criterion
    nuts = tc isa AdvancedHMC.DynamicTerminationCriterion
    λ = tc isa FixedIntegrationTime ? Float64(tc.λ) : 0.0
    # struct ahmc_kernel_cfg { int32 nuts, sampler, criterion, max_depth; double delta_max; int64 L; double lambda, refresh_alpha; }
    (Cint(nuts), sampler_code(typeof(κ.τ).parameters[1]), nuts ? criterion_code(tc) : Cint(0),
     Cint(nuts ? tc.max_depth : 0), nuts ? Float64(tc.Δ_max) : 0.0, Int64(tc isa FixedNSteps ? tc.L : 0), λ,
     κ.refreshment isa PartialMomentumRefreshment ? Float64(κ.refreshment.α) : 0.0)
end

"""
    transition_user_density(h, κ, z)

`transition(rng, h, κ, z)` (src/sampler.jl:48-58) for a context whose target is `UserDensity()`: momentum refresh,
the whole static / NUTS trajectory, acceptance and statistics run in the engine, `h.∂ℓπ∂θ` runs here.
"""
function transition_user_density(h::Hamiltonian, κ::HMCKernel, z::MI355XChains{T}) where {T}
    set_integrator!(z, κ.τ.integrator)
    cfg = Ref(kernel_cfg(κ))
    check(z.ctx, ccall((:ahmc_ext_begin, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint), z.ctx, cfg, 1))
    ext_drive!(z, h)
end

"find_good_stepsize(rng, h, θ) per chain (src/trajectory.jl:768-837) with the Hamiltonian's own closure"
function find_good_stepsize_user_density(h::Hamiltonian, z::MI355XChains{T}; initial_step_size=0.1, max_n_iters::Int=100) where {T}
    check(z.ctx, ccall((:ahmc_ext_find_good_stepsize_begin, LIB), Cint, (Ptr{Cvoid}, Cdouble, Cint), z.ctx, Float64(initial_step_size), max_n_iters))
    ext_drive!(z, h)
    ϵ = Vector{T}(undef, z.N)
    check(z.ctx, ccall((:ahmc_get_stepsize, LIB), Cint, (Ptr{Cvoid}, Ptr{T}), z.ctx, ϵ))
    return ϵ
end

# --- the hook: transition(rng, h, κ::HMCKernel, z) (src/sampler.jl:48-58) ----------------------------
# static HMC
function AdvancedHMC.transition(
    ::Union{AbstractRNG,AbstractVector{<:AbstractRNG}}, h::Hamiltonian,
    κ::HMCKernel{<:FullMomentumRefreshment,<:Trajectory{TS,I,<:FixedNSteps}}, z::MI355XChains{T},
) where {TS,I,T}
    if z.user_density
        transition_user_density(h, κ, z)
    else
        set_integrator!(z, κ.τ.integrator)
        check(z.ctx, ccall((:ahmc_hmc_transition, LIB), Cint, (Ptr{Cvoid}, Int64, Cdouble, Cint),
                           z.ctx, κ.τ.termination_criterion.L, 0.0, sampler_code(TS)))
    end
    tstat = (
        n_steps=κ.τ.termination_criterion.L,
        is_accept=getstat(z, 1, Int32) .!= 0,
        acceptance_rate=getstat(z, 2, T),
        log_density=getstat(z, 3, T),
        hamiltonian_energy=getstat(z, 4, T),
        hamiltonian_energy_error=getstat(z, 5, T),
        numerical_error=any(getstat(z, 8, Int32) .!= 0),
        step_size=getstat(z, 9, T),
        nom_step_size=getstat(z, 10, T),
    )
    return Transition(z, tstat)
end

# NUTS, for all chains at once (the reference's dynamic transition is scalar-only, src/trajectory.jl:677-681)
function AdvancedHMC.transition(
    ::Union{AbstractRNG,AbstractVector{<:AbstractRNG}}, h::Hamiltonian,
    κ::HMCKernel{<:FullMomentumRefreshment,<:Trajectory{TS,I,TC}}, z::MI355XChains{T},
) where {TS,I,TC<:AdvancedHMC.DynamicTerminationCriterion,T}
    tc = κ.τ.termination_criterion
    if z.user_density
        transition_user_density(h, κ, z)
    else
        set_integrator!(z, κ.τ.integrator)
        check(z.ctx, ccall((:ahmc_nuts_transition, LIB), Cint, (Ptr{Cvoid}, Cint, Cdouble, Cint, Cint),
                           z.ctx, tc.max_depth, Float64(tc.Δ_max), criterion_code(tc), sampler_code(TS)))
    end
    tstat = (
        n_steps=getstat(z, 0, Int32),
        is_accept=trues(z.N),
        acceptance_rate=getstat(z, 2, T),
        log_density=getstat(z, 3, T),
        hamiltonian_energy=getstat(z, 4, T),
        hamiltonian_energy_error=getstat(z, 5, T),
        max_hamiltonian_energy_error=getstat(z, 6, T),
        tree_depth=getstat(z, 7, Int32),
        numerical_error=getstat(z, 8, Int32) .!= 0,
        step_size=getstat(z, 9, T),
        nom_step_size=getstat(z, 10, T),
    )
    return Transition(z, tstat)
end

# HMCKernel(PartialMomentumRefreshment(α), τ) (src/trajectory.jl:249-254, src/hamiltonian.jl:243-254), static or dynamic: at the boundary the
# refreshment is part of the sample loop's kernel configuration (`refresh_alpha` of ahmc_kernel_cfg), so ONE iteration of that loop is this
# transition — the iteration counter, and with it every variate, continues as for the two methods above (as `Engine.transition` of the
# Python mirror; HIP == CPU checker on it: tests/test_random_configurations.py, tests/test_random_call_sequences.py)
function AdvancedHMC.transition(
    ::Union{AbstractRNG,AbstractVector{<:AbstractRNG}}, h::Hamiltonian,
    κ::HMCKernel{<:PartialMomentumRefreshment,<:Trajectory{TS,I,TC}}, z::MI355XChains{T},
) where {TS,I,TC,T}
    if z.user_density
        transition_user_density(h, κ, z)
    else
        set_integrator!(z, κ.τ.integrator)
        cfg = Ref(kernel_cfg(κ))
        check(z.ctx, ccall((:ahmc_sample_from, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Cint, Ptr{T}),
                           z.ctx, cfg, 1, 1, 0, false, C_NULL))
    end
    dynamic = TC <: AdvancedHMC.DynamicTerminationCriterion
    tstat = (
        n_steps=getstat(z, 0, Int32),
        is_accept=dynamic ? trues(z.N) : getstat(z, 1, Int32) .!= 0,
        acceptance_rate=getstat(z, 2, T),
        log_density=getstat(z, 3, T),
        hamiltonian_energy=getstat(z, 4, T),
        hamiltonian_energy_error=getstat(z, 5, T),
        max_hamiltonian_energy_error=getstat(z, 6, T),
        tree_depth=getstat(z, 7, Int32),
        numerical_error=getstat(z, 8, Int32) .!= 0,
        step_size=getstat(z, 9, T),
        nom_step_size=getstat(z, 10, T),
    )
    return Transition(z, tstat)
end

# --- adapt!(h, κ, adaptor, i, n_adapts, z, α) (src/sampler.jl:72-90) ---------------------------------
adaptor_code(::NoAdaptation) = ADAPT_NONE
adaptor_code(::StepSizeAdaptor) = ADAPT_STEPSIZE
adaptor_code(::MassMatrixAdaptor) = ADAPT_MASSMATRIX
adaptor_code(::NaiveHMCAdaptor) = ADAPT_NAIVE
adaptor_code(::StanHMCAdaptor) = ADAPT_STAN

"""
    PooledVar

Marker for `adaptor_init!(z, adaptor, δ; pooled=true)`: ONE shared `(D,)` M⁻¹ estimated from all chains (and all GPUs
once `comm_init!` has been called) — `AHMC_VAR_POOLED`.  The reference has no counterpart: in matrix mode it resizes
WelfordVar to one estimator per chain (src/adaptation/massmatrix.jl:103-121).  Needs `DiagEuclideanMetric(M⁻¹::Vector)`.
"""
struct PooledVar end

function adaptor_init!(z::MI355XChains, a::AbstractAdaptor, δ::Real; pooled::Bool=false)
    pc = a isa MassMatrixAdaptor ? a : (hasproperty(a, :pc) ? a.pc : nothing)
    est = pooled ? VAR_POOLED : pc isa AdvancedHMC.Adaptation.NutpieVar ? VAR_NUTPIE : VAR_WELFORD   # src/adaptation/massmatrix.jl:160-250
    check(z.ctx, ccall((:ahmc_set_var_estimator, LIB), Cint, (Ptr{Cvoid}, Cint), z.ctx, est))
    ib, tb, ws = a isa StanHMCAdaptor ? (a.init_buffer, a.term_buffer, a.window_size) : (75, 50, 25)
    check(z.ctx, ccall((:ahmc_adaptor_init, LIB), Cint, (Ptr{Cvoid}, Cint, Cdouble, Cint, Cint, Cint),
                       z.ctx, adaptor_code(a), Float64(δ), ib, tb, ws))
end

function AdvancedHMC.Adaptation.adapt!(
    h::Hamiltonian, κ::HMCKernel, adaptor::AbstractAdaptor, i::Int, n_adapts::Int, z::MI355XChains, α,
)
    # θ = NULL and α = NULL: adapt on the context's own position and last acceptance rates
    check(z.ctx, ccall((:ahmc_adapt, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Ptr{Cvoid}, Ptr{Cvoid}), z.ctx, i, n_adapts, C_NULL, C_NULL))
    return h, κ, i <= n_adapts
end

"""
    sample_device(seed, h, κ, θ, n_samples, adaptor, n_adapts; drop_warmup=false)

The whole loop of src/sampler.jl:182-228 in ONE library call (`ahmc_sample`): no host
synchronisation between transitions.  Returns the kept draws as a (D, N, n_keep) array.
"""
function sample_device(seed::Integer, h::Hamiltonian, κ::HMCKernel, θ::Matrix{T}, n_samples::Int,
                       adaptor::AbstractAdaptor=NoAdaptation(), n_adapts::Int=min(div(n_samples, 10), 1_000);
                       target::DeviceTarget=IsoGaussian(), δ=0.8, drop_warmup::Bool=false) where {T}
    D, N = size(θ)
    z = MI355XChains{T}(D, N)
    set_target!(z, target); set_metric!(z, h.metric); set_integrator!(z, κ.τ.integrator)
    seed!(z, seed); set_position!(z, θ); adaptor_init!(z, adaptor, δ)
    cfg = Ref(kernel_cfg(κ))
    n_keep = n_samples - (drop_warmup ? n_adapts : 0)
    out = Array{T}(undef, D, N, n_keep)
    # ABI v5: announce the run, so that the buffers of its launches are reserved here and not inside the first launch
    check(z.ctx, ccall((:ahmc_sample_reserve, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), z.ctx, cfg, n_samples))
    check(z.ctx, ccall((:ahmc_sample, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Cint, Ptr{T}),
                       z.ctx, cfg, n_samples, n_adapts, drop_warmup, out))
    check(z.ctx, ccall((:ahmc_sync, LIB), Cint, (Ptr{Cvoid},), z.ctx))
    return out
end

"""
    ref_compat!(z, on=true)

ABI v6: the reference's matrix-mode early exit (`!isfinite(z) && break` over ALL columns, `src/integrator.jl:252-258`) for `step` and static
EndPointTS transitions — off by default (each chain stops at its own first non-finite point, the reference's scalar semantics).  A maintainer
who compares the device path with the CPU sampler-vec path bit for bit on a batch that contains a diverging chain switches it on.
"""
ref_compat!(z::MI355XChains, on::Bool=true) =
    check(z.ctx, ccall((:ahmc_set_ref_compat, LIB), Cint, (Ptr{Cvoid}, Cint), z.ctx, on ? 1 : 0))

# --- ABI v3 (+ v4: accum_state / restore_accum! below): checkpoint / resume, the final gather, device-side diagnostics ---------------------------------
# struct ahmc_adaptor_state (include/ahmc_hip.h): the value the reference carries in HMCState.adaptor
# (src/abstractmcmc.jl:11-27) — isbits, same field order and alignment as the C struct
struct AdaptorState
    kind::Cint; var_estimator::Cint; init_buffer::Cint; term_buffer::Cint; window_size::Cint
    adapting::Cint; has_da::Cint; n_welford::Cint
    delta::Cdouble
    stan_i::Int64; n_adapts::Int64; wv_n::Int64; iteration::Int64
end

"Everything a resumed run needs besides the Hamiltonian: phase point, metric, step sizes, adaptor, RNG counter."
struct Checkpoint{T}
    θ::Matrix{T}; r::Matrix{T}; ℓπ::Vector{T}; g::Matrix{T}
    M⁻¹::Union{Nothing,Array{T}}; metric_kind::Cint; ϵ::Vector{T}
    adaptor::AdaptorState; da::Matrix{T}; welford::Array{T,3}
end

function checkpoint(z::MI355XChains{T}, metric::AbstractMetric) where {T}
    st = Ref{AdaptorState}()
    check(z.ctx, ccall((:ahmc_get_adaptor_state, LIB), Cint, (Ptr{Cvoid}, Ptr{AdaptorState}, Ptr{Cvoid}, Ptr{Cvoid}), z.ctx, st, C_NULL, C_NULL))
    da = Matrix{T}(undef, z.N, st[].has_da != 0 ? 5 : 0)                     # (N, 5) column-major = C's (5, N)
    wv = Array{T,3}(undef, z.D, z.N, Int(st[].n_welford))                    # (D, N, n) = C's (n, D, N)
    check(z.ctx, ccall((:ahmc_get_adaptor_state, LIB), Cint, (Ptr{Cvoid}, Ptr{AdaptorState}, Ptr{T}, Ptr{T}), z.ctx, st,
                       isempty(da) ? C_NULL : da, isempty(wv) ? C_NULL : wv))
    pp = PhasePoint(z)
    ϵ = Vector{T}(undef, z.N)
    check(z.ctx, ccall((:ahmc_get_stepsize, LIB), Cint, (Ptr{Cvoid}, Ptr{T}), z.ctx, ϵ))
    one = Ref{Int64}(0)                                                      # AHMC_INFO_STEPSIZE_SCALAR = 14: ONE nominal ϵ stays one
    check(z.ctx, ccall((:ahmc_get_info, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Int64}), z.ctx, 14, one))
    one[] != 0 && (ϵ = ϵ[1:1])
    M = metric isa UnitEuclideanMetric ? nothing : similar(metric.M⁻¹, T)
    kind = metric isa UnitEuclideanMetric ? METRIC_UNIT : metric isa DiagEuclideanMetric ? METRIC_DIAG : METRIC_DENSE
    M === nothing || check(z.ctx, ccall((:ahmc_get_metric, LIB), Cint, (Ptr{Cvoid}, Ptr{T}, Int64), z.ctx, M, length(M)))
    return Checkpoint{T}(pp.θ, pp.r, pp.ℓπ.value, pp.ℓπ.gradient, M, kind, ϵ, st[], da, wv)
end

function restore!(z::MI355XChains{T}, c::Checkpoint{T}) where {T}
    c.M⁻¹ === nothing || check(z.ctx, ccall((:ahmc_set_metric, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{T}, Int64), z.ctx, c.metric_kind, c.M⁻¹, length(c.M⁻¹)))
    check(z.ctx, ccall((:ahmc_set_stepsize, LIB), Cint, (Ptr{Cvoid}, Ptr{T}, Int64), z.ctx, c.ϵ, length(c.ϵ)))
    check(z.ctx, ccall((:ahmc_set_phasepoint, LIB), Cint, (Ptr{Cvoid}, Ptr{T}, Ptr{T}, Ptr{T}, Ptr{T}), z.ctx, c.θ, c.r, c.ℓπ, c.g))
    st = Ref(c.adaptor)
    check(z.ctx, ccall((:ahmc_set_adaptor_state, LIB), Cint, (Ptr{Cvoid}, Ptr{AdaptorState}, Ptr{T}, Ptr{T}), z.ctx, st,
                       isempty(c.da) ? C_NULL : c.da, isempty(c.welford) ? C_NULL : c.welford))
end

"Continue the loop of src/sampler.jl:182-228 at iteration `i_first` (after `restore!`): `ahmc_sample_from`."
function resume_device!(z::MI355XChains{T}, κ::HMCKernel, i_first::Int, n_samples::Int, n_adapts::Int) where {T}
    cfg = Ref(kernel_cfg(κ))
    check(z.ctx, ccall((:ahmc_sample_from, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64, Int64, Cint, Ptr{T}),
                       z.ctx, cfg, i_first, n_samples, n_adapts, false, C_NULL))
end

# Multi-GPU (SURVEY §8e): one Julia process per GPU (Distributed / MPI.jl launches them), each with its own
# MI355XChains over its block of chains and seed!(…; chain_offset = first global chain).  Rank 0 makes the id, the host
# broadcasts the 128 bytes with whatever it already has (Distributed's remotecall, MPI.Bcast!), every rank joins.
comm_unique_id() = (id = Vector{UInt8}(undef, 128); code = ccall((:ahmc_comm_unique_id, LIB), Cint, (Ptr{UInt8},), id);
                    code == 0 || throw(AHMCError(code, "ahmc_comm_unique_id")); id)
comm_init!(z::MI355XChains, id::Vector{UInt8}, n_ranks::Integer, rank::Integer) =
    check(z.ctx, ccall((:ahmc_comm_init, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint), z.ctx, id, n_ranks, rank))
"an ncclComm_t the host already owns (its entry points are resolved from the RCCL copy loaded in this process)"
set_comm!(z::MI355XChains, comm::Ptr{Cvoid}, n_ranks::Integer, rank::Integer) =
    check(z.ctx, ccall((:ahmc_set_comm, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Cint, Cint), z.ctx, comm, n_ranks, rank))

"ABI v5: what the communicator itself measured when it was attached — ranks that answered, Σ / min / max of their chain counts."
function comm_info(z::MI355XChains)
    seen, tot, lo, hi = Ref{Int64}(0), Ref{Int64}(0), Ref{Int64}(0), Ref{Int64}(0)
    check(z.ctx, ccall((:ahmc_comm_info, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}), z.ctx, seen, tot, lo, hi))
    return (ranks_seen=seen[], chains_total=tot[], chains_min=lo[], chains_max=hi[])
end

"Pooled per-dimension mean / variance of the kept draws of ALL ranks (one ncclAllReduce of 2D+3 doubles)."
function gather_moments(z::MI355XChains)
    μ, σ² = Vector{Float64}(undef, z.D), Vector{Float64}(undef, z.D)
    n, steps, ndiv = Ref{Int64}(0), Ref{Int64}(0), Ref{Int64}(0)
    check(z.ctx, ccall((:ahmc_gather_moments, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}),
                       z.ctx, μ, σ², n, steps, ndiv))
    return (mean=μ, var=σ², n_draws=n[], n_steps=steps[], n_divergent=ndiv[])
end

# --- ABI v4: a user log-density ON THE DEVICE (no host round trip per leapfrog) ------------------------------
# (1) a device KERNEL the host already owns.  With AMDGPU.jl:  k = @roc launch=false my_kernel(θ, ℓπ, g, cols, n, D, N, user)
#     compiles a Julia kernel with the C-ABI signature of include/ahmc_hip.h; `k.fun.handle` (hipFunction_t) goes in here and the
#     engine launches it itself between its tree kernels — `transition`, `find_good_stepsize`, `sample_device` then run as for
#     a built-in family (DeviceTarget), not through ext_drive!.
"ahmc_set_target_kernel: `fun` = hipFunction_t; grid = ⌈n_cols / chains_per_block⌉ blocks of `block_threads` threads"
function set_target_kernel!(z::MI355XChains, fun::Ptr{Cvoid}; block_threads::Integer=256, chains_per_block::Integer=1, user::Ptr{Cvoid}=C_NULL)
    check(z.ctx, ccall((:ahmc_set_target_kernel, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}, Cint, Cint, Ptr{Cvoid}),
                       z.ctx, KERNEL_HIP_FUNCTION, fun, block_threads, chains_per_block, user))
end
# (2) a device FUNCTION in HIP C++ compiled into the engine's fused kernels (include/ahmc_user_target.h); `plugin_so` comes from
#     `python -c "from ahmc_amd.build import build_target_plugin; print(build_target_plugin(src, 'float64', G, E))"` (or the hipcc
#     line in that header) with (G, E) = the context's thread geometry.
function set_target_plugin!(z::MI355XChains{T}, plugin_so::AbstractString, params::Vector{T}=T[]) where {T}
    check(z.ctx, ccall((:ahmc_set_target_plugin, LIB), Cint, (Ptr{Cvoid}, Cstring, Ptr{T}, Int64), z.ctx, plugin_so,
                       isempty(params) ? C_NULL : params, length(params)))
end

# (3) COMPILED device code — what GPUCompiler.jl / AMDGPU.jl emit for a Julia function: amdgcn LLVM bitcode (or a relocatable device object)
#     defining the C symbol of include/ahmc_user_target_object.h (`ahmc_user_logdensity_f64` / `_f32`).  The build helper links it with the
#     engine's fused kernels under device LTO (the density is inlined into the leaf loop: built-in speed, INTEGRATION.md §3c) and the result
#     is bound like (2).  (G, E) of the context: AHMC_INFO_GROUP_LANES = 0, AHMC_INFO_ELEMS_PER_LANE = 1 of ahmc_get_info.
function set_target_object!(z::MI355XChains{T}, object_or_bitcode::AbstractString, params::Vector{T}=T[]; python::AbstractString="python",
                            repo::AbstractString=get(ENV, "AHMC_REPO", normpath(joinpath(@__DIR__, "..")))) where {T}
    isfile(object_or_bitcode) || throw(ArgumentError("set_target_object!: no such file: $(repr(object_or_bitcode))"))
    G, E = Ref{Int64}(0), Ref{Int64}(0)
    check(z.ctx, ccall((:ahmc_get_info, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Int64}), z.ctx, 0, G))
    check(z.ctx, ccall((:ahmc_get_info, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Int64}), z.ctx, 1, E))
    dt = T === Float32 ? "float32" : "float64"
    # the path and every number travel as ARGUMENTS (sys.argv), never inside the program text: a quote in a path is a quote in a path.
    # `repo` (the checkout that holds ahmc_amd.py) goes on sys.path, so the helper does not depend on what the `python` on PATH can import.
    code = "import sys; sys.path.insert(0, sys.argv[6]); from ahmc_amd.build import build_target_plugin_from_object as b; " *
           "print(b(sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])))"
    out, err = IOBuffer(), IOBuffer()
    cmd = `$python -c $code $(abspath(object_or_bitcode)) $dt $(G[]) $(E[]) $(length(params)) $repo`
    proc = run(pipeline(ignorestatus(cmd); stdout=out, stderr=err))
    success(proc) || error("set_target_object!: the build helper failed (exit $(proc.exitcode)):\n" * String(take!(err)))
    lines = split(strip(String(take!(out))), '\n')
    plugin_so = isempty(lines) ? "" : String(strip(lines[end]))
    isfile(plugin_so) || error("set_target_object!: the build helper returned $(repr(plugin_so)), which is not a file")
    set_target_plugin!(z, plugin_so, params)
end

"the running accumulators (Σθ, Σθ², Σ n_steps, divergences, the energy sums behind EBFMI) as part of a Checkpoint"
function accum_state(z::MI355XChains{T}) where {T}
    n = Ref{Int64}(0)
    steps, ndiv = Vector{Int64}(undef, z.N), Vector{Int64}(undef, z.N)
    Σθ, Σθ², E = Matrix{T}(undef, z.D, z.N), Matrix{T}(undef, z.D, z.N), Matrix{T}(undef, z.N, 5)
    check(z.ctx, ccall((:ahmc_get_accum_state, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Ptr{Int64}, Ptr{T}, Ptr{T}, Ptr{T}),
                       z.ctx, n, steps, ndiv, Σθ, Σθ², E))
    return (n_transitions=n[], n_steps=steps, n_divergent=ndiv, sum_theta=Σθ, sumsq_theta=Σθ², energy_sums=E)
end
restore_accum!(z::MI355XChains{T}, a) where {T} =
    check(z.ctx, ccall((:ahmc_set_accum_state, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{T}, Ptr{T}, Ptr{T}),
                       z.ctx, a.n_transitions, a.n_steps, a.n_divergent, a.sum_theta, a.sumsq_theta, a.energy_sums))

"EBFMI (src/diagnosis.jl:1-3) per chain over the kept transitions of the last `sample_device` / `resume_device!` call"
function AdvancedHMC.EBFMI(z::MI355XChains{T}) where {T}
    out = Vector{T}(undef, z.N)
    check(z.ctx, ccall((:ahmc_ebfmi, LIB), Cint, (Ptr{Cvoid}, Ptr{T}), z.ctx, out))
    return out
end

end # module
