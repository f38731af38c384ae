#!/usr/bin/env python
"""Extract the known-answer values the reference's OWN test-suite pins for the hot path and write
them to tests/golden/reference_kats.json.

Runs only in the build container (needs /root/reference); the JSON is committed and is what the
tests read (the GPU box has no /root/reference).  Julia is not installed, so nothing here executes
the reference: every value is parsed out of the reference's test sources, and the script fails if
a pattern no longer matches (i.e. if the reference changes under us).
"""
import json
import os
import re
import sys

REF = os.environ.get("AHMC_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")


def src(rel):
    with open(os.path.join(REF, rel), encoding="utf-8") as f:
        return f.read()


def must(pattern, text, what):
    m = re.search(pattern, text, re.S)
    if not m:
        sys.exit(f"pattern for {what} not found — reference changed?")
    return m


def line_of(text, needle):
    return text[: text.index(needle)].count("\n") + 1


def main():
    kats = {"_source": "TuringLang/AdvancedHMC.jl v0.8.6 test suite (parsed, not executed)"}

    # --- Stan window schedule: test/adaptation.jl "Stan HMC adaptors" ---------------------------
    t = src("test/adaptation.jl")
    m = must(r"initialize!\(a, ([\d_]+)\)\s*@test a\.state\.window_start == (\d+)\s*@test a\.state\.window_end == (\d+)\s*"
             r"@test a\.state\.window_splits == \[([\d, ]+)\]", t, "window schedule")
    kats["stan_windows"] = {
        "cite": f"test/adaptation.jl:{line_of(t, 'a.state.window_start ==')}-{line_of(t, 'a.state.window_splits ==')}",
        "n_adapts": int(m.group(1).replace("_", "")), "init_buffer": 75, "term_buffer": 50, "window_size": 25,
        "window_start": int(m.group(2)), "window_end": int(m.group(3)),
        "window_splits": [int(x) for x in m.group(4).split(",")],
    }
    # defaults 75/50/25: src/adaptation/stan_adaptor.jl keyword defaults
    s = src("src/adaptation/stan_adaptor.jl")
    d = must(r"init_buffer::Int=(\d+),\s*term_buffer::Int=(\d+),\s*window_size::Int=(\d+)", s, "stan defaults")
    assert [int(d.group(i)) for i in (1, 2, 3)] == [75, 50, 25]

    # --- temper schedule: test/integrator.jl "temper" ---------------------------------------------
    t = src("test/integrator.jl")
    must(r"αsqrt = 2\.0\s*lf = TemperedLeapfrog\(ϵ, αsqrt\^2\)", t, "temper setup")
    calls = re.findall(r"r(\d) = AdvancedHMC\.temper\(lf, r, \(i=(\d), is_half=(true|false)\), (\d)\)", t)
    expect = dict(re.findall(r"@test r(\d) == (αsqrt|inv\(αsqrt\)) \* ones\(5\)", t))
    assert len(calls) == 6 and len(expect) == 6
    kats["temper"] = {
        "cite": f"test/integrator.jl:{line_of(t, 'αsqrt = 2.0')}-{line_of(t, '@test_throws BoundsError AdvancedHMC.temper')}",
        "alpha": 4.0,
        "cases": [{"i": int(i), "is_half": h == "true", "n_steps": int(n), "factor": 2.0 if expect[k] == "αsqrt" else 0.5}
                  for k, i, h, n in calls],
    }

    # --- BinaryTree combine: test/trajectory.jl "BinaryTree" -----------------------------------------
    t = src("test/trajectory.jl")
    m1 = must(r"t1 = AdvancedHMC\.BinaryTree\(z, z, AdvancedHMC\.TurnStatistic\(\), ([\d.]+), (\d+), (-?[\d.]+)\)", t, "t1")
    m2 = must(r"t2 = AdvancedHMC\.BinaryTree\(z, z, AdvancedHMC\.TurnStatistic\(\), ([\d.]+), (\d+), (-?[\d.]+)\)", t, "t2")
    m4 = must(r"t4 = AdvancedHMC\.BinaryTree\(z, z, AdvancedHMC\.TurnStatistic\(\), ([\d.]+), (\d+), (-?[\d.]+)\)", t, "t4")
    e3 = must(r"@test t3\.sum_α ≈ ([\d.]+) atol = ([\d.e-]+)\s*@test t3\.nα == (\d+)\s*@test t3\.ΔH_max == (-?[\d.]+)", t, "t3")
    e5 = must(r"@test t5\.ΔH_max == (-?[\d.]+)", t, "t5")
    f = lambda m: {"sum_alpha": float(m.group(1)), "n_alpha": int(m.group(2)), "dH_max": float(m.group(3))}  # noqa: E731
    kats["binary_tree_combine"] = {
        "cite": f"test/trajectory.jl:{line_of(t, 't1 = AdvancedHMC.BinaryTree')}-{line_of(t, '@test t5.ΔH_max')}",
        "t1": f(m1), "t2": f(m2), "t4": f(m4),
        "t3": {"sum_alpha": float(e3.group(1)), "atol": float(e3.group(2)), "n_alpha": int(e3.group(3)), "dH_max": float(e3.group(4))},
        "t5": {"dH_max": float(e5.group(1))},
    }

    # --- Termination algebra: the 4 + 16 assertions of "Termination" -------------------------------
    single = re.findall(r"@test AdvancedHMC\.isterminated\(t(\d)(\d)\) == (true|false)", t)
    prods = re.findall(r"@test AdvancedHMC\.isterminated\(t(\d)(\d) \* t(\d)(\d)\) == (true|false)", t)
    assert len(single) == 4 and len(prods) == 16
    kats["termination"] = {
        "cite": f"test/trajectory.jl:{line_of(t, 't00 = AdvancedHMC.Termination(false, false)')}-{line_of(t, 'isterminated(t11 * t11)')}",
        "single": [[int(a), int(b), r == "true"] for a, b, r in single],
        "product": [[int(a), int(b), int(c), int(d), r == "true"] for a, b, c, d, r in prods],
    }

    # --- tree sampler combine: "TreeSampler" ---------------------------------------------------------
    mw = must(r"w1 = (\d+)\s*s1 = AdvancedHMC\.MultinomialTS\(z1, log\(w1\)\)\s*w2 = (\d+)", t, "multinomial weights")
    mn = must(r"n1 = (\d+)\s*s1 = AdvancedHMC\.SliceTS\(z1, ℓu, n1\)\s*n2 = (\d+)", t, "slice counts")
    mr = must(r"@test mean\(s3_θ\) ≈ ones\(D\) \* w2 / \(w1 \+ w2\) rtol = ([\d.]+)", t, "pick ratio rtol")
    kats["sampler_combine"] = {
        "cite": f"test/trajectory.jl:{line_of(t, 'n_samples = 10_000')}-{line_of(t, 'w2 / (w1 + w2) rtol')}",
        "w1": int(mw.group(1)), "w2": int(mw.group(2)), "n1": int(mn.group(1)), "n2": int(mn.group(2)),
        "n_samples": 10000, "rtol": float(mr.group(1)),
    }

    # --- tolerances / fixtures of test/common.jl and the statistical tests ----------------------------
    c = src("test/common.jl")
    D = int(must(r"const D = (\d+)", c, "D").group(1))
    kats["common"] = {"cite": "test/common.jl:6-12", "D": D, "DETATOL": 1e-3 * D, "RNDATOL": 5e-2 * D * 2}
    sv = src("test/sampler-vec.jl")
    must(r"@test mean\(samples\) ≈ zeros\(D, n_chains\) atol = RNDATOL \* n_chains", sv, "sampler-vec tolerance")
    kats["sampler_vec"] = {"cite": "test/sampler-vec.jl:7-43", "n_chains": 5, "eps": 0.1, "n_steps": 10,
                           "n_samples": 20000, "n_adapts": 4000, "atol": 5e-2 * D * 2 * 5}
    it = src("test/integrator.jl")
    must(r"@test all\(x -> abs\(x - mean\(rs\)\) < 2e-3, rs\)", it, "oscillator bound")
    kats["harmonic_oscillator"] = {"cite": f"test/integrator.jl:{line_of(it, 'Analytical solution to Eq (2.11)')}-"
                                           f"{line_of(it, 'abs(x - mean(Hs)) < 2e-3')}",
                                   "eps": 0.01, "n_steps": 10000, "burn": 1000, "bound": 2e-3}

    # --- constants of the algorithms (parsed from src/, to catch drift) ---------------------------------
    tr = src("src/trajectory.jl")
    md = must(r"struct GeneralisedNoUTurn\{F<:AbstractFloat\} <: DynamicTerminationCriterion\s*max_depth::Int = (\d+)\s*Δ_max::F = ([\d.]+)", tr, "NUTS defaults")
    ss = src("src/adaptation/stepsize.jl")
    da = must(r"NesterovDualAveraging\(T\((\d+)//(\d+)\), T\((\d+)\), T\((\d+)//(\d+)\), δ, ϵ\)", ss, "DA constants")
    mm = src("src/adaptation/massmatrix.jl")
    wv = must(r"n, ϵ = T\(n\), T\(([\de.-]+)\)\s*return n / \(\(n \+ (\d+)\) \* \(n - 1\)\) \* M \.\+ ϵ \* \((\d+) / \(n \+ (\d+)\)\)", mm, "Welford regularisation")
    kats["constants"] = {
        "max_depth": int(md.group(1)), "delta_max": float(md.group(2)),
        "da_gamma": int(da.group(1)) / int(da.group(2)), "da_t0": float(da.group(3)), "da_kappa": int(da.group(4)) / int(da.group(5)),
        "welford_eps": float(wv.group(1)), "welford_shrink": int(wv.group(2)), "welford_n_min": 10,
        "cite": "src/trajectory.jl:434-437; src/adaptation/stepsize.jl:168-172; src/adaptation/massmatrix.jl:152-157",
    }

    with open(OUT, "w", encoding="utf-8") as fjson:
        json.dump(kats, fjson, indent=1, ensure_ascii=False)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
