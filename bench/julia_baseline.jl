# bench/julia_baseline.jl — the REAL AdvancedHMC.jl on the host CPU, for the configs of BASELINE.json that the
# reference can run (cfg1-3).  NOT executed in the build environment (no Julia there): bench.py's `cpu_baseline`
# is the C++ restatement under oracle/.  Anyone with Julia can run
#     julia -t auto --project=<AdvancedHMC checkout> bench/julia_baseline.jl [n_chains_nuts]
# and its numbers supersede the restatement's (BASELINE.md §2).
#
# Metric: chain-leapfrog-steps per second = Σ n_steps / sampling wall time (src/trajectory.jl:288,:728).
# The reference vectorises static HMC only (test/sampler-vec.jl); NUTS is one scalar chain per call, so the
# NUTS configs loop chains under Threads.@threads (src/trajectory.jl:626-635 is scalar-only).
using AdvancedHMC, Random, Statistics, LinearAlgebra

# the targets of SURVEY.md §8d ---------------------------------------------------------------
ℓπ_iso(θ::AbstractVector) = -sum(abs2, θ) / 2 - length(θ) * log(2π) / 2
ℓπ_iso(θ::AbstractMatrix) = vec(-sum(abs2, θ; dims=1) / 2 .- size(θ, 1) * log(2π) / 2)
∂ℓπ_iso(θ::AbstractVector) = (ℓπ_iso(θ), -θ)
∂ℓπ_iso(θ::AbstractMatrix) = (ℓπ_iso(θ), -θ)

function ℓπ_funnel(θ::AbstractVector)   # θ₁ ~ N(0, 3²), θ₂..D ~ N(0, e^{θ₁})
    y = θ[1]; n = length(θ) - 1
    return -(log(2π) + 2log(3) + y^2 / 9) / 2 - n * (log(2π) + y) / 2 - sum(abs2, @view θ[2:end]) * exp(-y) / 2
end
function ∂ℓπ_funnel(θ::AbstractVector)
    y = θ[1]; n = length(θ) - 1; ss = sum(abs2, @view θ[2:end]); ey = exp(-y)
    g = similar(θ)
    g[1] = -y / 9 - n / 2 + ss * ey / 2
    g[2:end] .= .-θ[2:end] .* ey
    return ℓπ_funnel(θ), g
end

# cfg1: D=10, 1 024 chains, Unit metric, static HMC L=16, vectorised (the reference's own sampler-vec path)
function cfg1(; D=10, N=1024, n_samples=2000)
    rng = MersenneTwister(0x5EED0001)
    h = Hamiltonian(UnitEuclideanMetric((D, N)), ℓπ_iso, ∂ℓπ_iso)
    κ = HMCKernel(Trajectory{EndPointTS}(Leapfrog(0.1), FixedNSteps(16)))
    θ0 = rand(rng, D, N)
    sample(rng, h, κ, θ0, 10; verbose=false, progress=false)   # compile
    t = @elapsed ((_, stats) = sample(rng, h, κ, θ0, n_samples; verbose=false, progress=false))
    steps = sum(s.n_steps for s in stats) * N
    println("cfg1  D=$D N=$N static HMC L=16 (vectorised, 1 thread): ", steps / t, " leapfrog-steps/s")
end

# cfg2 / cfg3: NUTS(0.8) + StanHMCAdaptor, scalar chains under threads
function nuts_chains(ℓπ, ∂ℓπ, D, n_chains; n_adapts=200, n_samples=100, seed=0x5EED0002)
    steps = zeros(Int, n_chains); secs = zeros(n_chains)
    Threads.@threads for c in 1:n_chains
        rng = MersenneTwister(seed + c)
        metric = DiagEuclideanMetric(D)
        h = Hamiltonian(metric, ℓπ, ∂ℓπ)
        θ0 = rand(rng, D)
        lf = Leapfrog(find_good_stepsize(rng, h, θ0))
        κ = HMCKernel(Trajectory{MultinomialTS}(lf, GeneralisedNoUTurn(; max_depth=10, Δ_max=1000.0)))
        adaptor = StanHMCAdaptor(MassMatrixAdaptor(metric), StepSizeAdaptor(0.8, lf))
        θs, _ = sample(rng, h, κ, θ0, n_adapts, adaptor, n_adapts; verbose=false, progress=false)
        secs[c] = @elapsed ((_, stats) = sample(rng, h, κ, θs[end], n_samples; verbose=false, progress=false))
        steps[c] = sum(s.n_steps for s in stats)
    end
    return sum(steps), maximum(secs), sum(secs)
end

function main()
    n = length(ARGS) >= 1 ? parse(Int, ARGS[1]) : 64 * Threads.nthreads()
    println("Julia ", VERSION, ", ", Threads.nthreads(), " threads, AdvancedHMC ", pkgversion(AdvancedHMC))
    cfg1()
    nuts_chains(ℓπ_iso, ∂ℓπ_iso, 128, Threads.nthreads(); n_adapts=20, n_samples=5)   # compile
    t = @elapsed ((steps, _, cpu) = nuts_chains(ℓπ_iso, ∂ℓπ_iso, 128, n))
    println("cfg2  D=128 iso Gaussian, NUTS+Stan, $n chains (sampling phase only): ", steps / (cpu / Threads.nthreads()),
            " leapfrog-steps/s on ", Threads.nthreads(), " threads (", steps / cpu, " per thread); whole call ", t, " s")
    t = @elapsed ((steps, _, cpu) = nuts_chains(ℓπ_funnel, ∂ℓπ_funnel, 32, n; seed=0x5EED0003))
    println("cfg3  D=32 funnel, NUTS+Stan, $n chains: ", steps / (cpu / Threads.nthreads()), " leapfrog-steps/s on ",
            Threads.nthreads(), " threads (", steps / cpu, " per thread)")
end

main()
