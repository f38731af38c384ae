# oracle/dump_golden.jl — escape hatch for pinning the oracle to the REAL reference (SURVEY.md §8c).
# NOT executed in the build environment (no Julia).  With Julia and an AdvancedHMC.jl checkout:
#     julia --project=<AdvancedHMC checkout> oracle/dump_golden.jl tests/golden/julia_golden.json
# It dumps RNG-free quantities on fixed inputs (the random streams of Julia and of this engine differ by
# construction: variate-level parity is unpinned); tests/test_oracle_golden.py::test_julia_golden replays the
# file against oracle/ when it exists.
using AdvancedHMC, LinearAlgebra, Random
import AdvancedHMC: phasepoint, step, neg_energy, ∂H∂r, isterminated, TurnStatistic, BinaryTree, combine, Termination
using AdvancedHMC.Adaptation: WelfordVar, NutpieVar, NesterovDualAveraging, adapt!, getM⁻¹, getϵ, reset!, initialize!, finalize!

function json(io, x)   # tiny JSON writer (no dependency)
    if x isa AbstractDict
        print(io, "{"); first = true
        for (k, v) in x
            first || print(io, ","); first = false
            print(io, "\"", k, "\":"); json(io, v)
        end
        print(io, "}")
    elseif x isa AbstractArray
        print(io, "["); for (i, v) in enumerate(x); i > 1 && print(io, ","); json(io, v); end; print(io, "]")
    elseif x isa AbstractFloat
        print(io, isfinite(x) ? repr(Float64(x)) : (isnan(x) ? "\"nan\"" : (x > 0 ? "\"inf\"" : "\"-inf\"")))
    elseif x isa Bool || x isa Integer
        print(io, x)
    else
        print(io, "\"", x, "\"")
    end
end

ℓπ(θ::AbstractMatrix) = vec(-sum(abs2, θ; dims=1) / 2 .- size(θ, 1) * log(2π) / 2)
∂ℓπ∂θ(θ::AbstractMatrix) = (ℓπ(θ), -θ)
ℓπ(θ::AbstractVector) = -sum(abs2, θ) / 2 - length(θ) * log(2π) / 2
∂ℓπ∂θ(θ::AbstractVector) = (ℓπ(θ), -θ)

function main(path)
    out = Dict{String,Any}()
    D, N = 5, 4
    θ = reshape(collect(range(-1.0, 1.0; length=D * N)), D, N)
    r = reshape(collect(range(0.5, -0.7; length=D * N)), D, N)
    Minv = reshape(collect(range(0.5, 1.5; length=D * N)), D, N)
    ϵ = collect(range(0.05, 0.2; length=N))
    for (name, metric) in (("unit", UnitEuclideanMetric((D, N))), ("diag", DiagEuclideanMetric(Minv)))
        h = Hamiltonian(metric, ℓπ, ∂ℓπ∂θ)
        z = phasepoint(h, θ, r)
        traj = Dict{String,Any}("theta" => vec(θ), "r" => vec(r), "eps" => ϵ, "lp0" => z.ℓπ.value, "lk0" => z.ℓκ.value)
        for n in (7, -4)
            z2 = step(Leapfrog(ϵ), h, z, n)
            traj["step$(n)"] = Dict("theta" => vec(z2.θ), "r" => vec(z2.r), "lp" => z2.ℓπ.value, "lk" => z2.ℓκ.value,
                                    "grad" => vec(z2.ℓπ.gradient))
        end
        name == "diag" && (traj["minv"] = vec(Minv))
        out["leapfrog_" * name] = traj
    end
    # tempered leapfrog (src/integrator.jl:198-209)
    h = Hamiltonian(UnitEuclideanMetric((D, N)), ℓπ, ∂ℓπ∂θ)
    zt = step(TemperedLeapfrog(0.1, 1.05), h, phasepoint(h, θ, r), 6)
    out["tempered"] = Dict("theta" => vec(zt.θ), "r" => vec(zt.r), "lp" => zt.ℓπ.value, "lk" => zt.ℓκ.value)
    # dual averaging on a fixed α sequence (src/adaptation/stepsize.jl:178-210)
    da = NesterovDualAveraging(0.8, 0.1)
    αs = [0.3, 0.95, 0.6, 1.0, 0.05, 0.8, 0.8, 0.7, 0.99, 0.4]
    epss = Float64[]
    for α in αs
        adapt!(da, zeros(2), α); push!(epss, getϵ(da))
    end
    finalize!(da)
    out["dual_averaging"] = Dict("alpha" => αs, "eps" => epss, "final" => getϵ(da))
    # WelfordVar / NutpieVar on a fixed sequence (src/adaptation/massmatrix.jl:141-157, :238-250)
    xs = [sin.(collect(1:3) .* k) .* [1.0, 2.0, 0.5] for k in 1:15]
    gs = [cos.(collect(1:3) .* k) ./ [1.0, 4.0, 0.25] for k in 1:15]
    wv = WelfordVar{Float64}((3,)); nv = NutpieVar{Float64}((3,))
    for (x, g) in zip(xs, gs)
        adapt!(wv, x, 1.0)
        adapt!(nv, AdvancedHMC.PhasePoint(x, x, AdvancedHMC.DualValue(0.0, g), AdvancedHMC.DualValue(0.0, x)), 1.0)
    end
    out["welford"] = Dict("x" => xs, "g" => gs, "var" => getM⁻¹(wv), "nutpie" => getM⁻¹(nv))
    open(path, "w") do io
        json(io, out)
    end
    println("wrote ", path)
end

main(length(ARGS) >= 1 ? ARGS[1] : "julia_golden.json")
