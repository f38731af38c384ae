#!/usr/bin/env python
"""Where a leaf step's cycles go: the stage stamps of a measurement build (-DAHMC_LEAF_PROF=1, ahmc_nuts.hpp) on
the shapes of cfg2 / cfg3 / cfg5, for a LONE wave (one chain-chunk on an otherwise empty GPU: pure dependency latency) and at the
bench's occupancy (all chains: issue-slot sharing and memory contention included).

    python scripts/build_variant.py lp -DAHMC_LEAF_PROF=1
    AHMC_HIP_LIB=$PWD/advancedhmc.jl_amd/csrc/variants/libahmc_hip_lp.so python scripts/leaf_latency.py OUTDIR

Per case: a warm-up (StanHMCAdaptor; untimed, no records) so that step sizes and metric are the adapted ones, then ONE launch of draws
with the per-wave records on: cycles (shader clock, s_memtime) per stage divided by the leaf steps of the wave, averaged over waves
(weighted by leaf steps), and the same in µs at the shader clock the MFMA probe measured (profiles/r5_mfma_clock.jsonl: 2 387 MHz)."""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ahmc_amd as A  # noqa: E402

OUT = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/leaf_latency"
os.makedirs(OUT, exist_ok=True)
MHZ = 2387.0
STAGES = ["leapfrog_vector_work", "energy_allreduce", "weight_exp_divergence_stats", "merges", "park_and_loop", "doubling_top_level",
          "transition_prologue", "transition_epilogue"]
CASES = {"cfg2": dict(D=128, target=A.IsoGaussian, cpw=1, full=65536), "cfg3": dict(D=32, target=A.Funnel, cpw=4, full=65536),
         "cfg5": dict(D=2048, target=A.HierGaussian, cpw=1, full=8192)}


def run(lib, name, N, n_adapts, n_draws, tag):
    cs = CASES[name]
    D = cs["D"]
    metric = A.DiagEuclideanMetric(np.ones((D, N), order="F"))
    eng = A.Engine(A.Hamiltonian(metric, cs["target"](D)), N, rng=A.PhiloxRNG(0x5EED0000 + int(name[-1])), lib=lib)
    lf = A.Leapfrog(np.full(N, 0.1))
    eng.set_integrator(lf)
    eng.set_position(np.asfortranarray(np.random.default_rng(5).random((D, N))))
    eng.find_good_stepsize()
    eng.adaptor_init(A.StanHMCAdaptor(A.MassMatrixAdaptor(metric), A.StepSizeAdaptor(0.8, lf)))
    k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=10, delta_max=1000.0)))
    os.environ.pop("AHMC_WAVE_TIMELINE_OUT", None)
    eng.run(k, n_adapts, n_adapts)
    eng.sync()
    prefix = os.path.join(OUT, f"{name}_{tag}")
    for f in glob.glob(prefix + ".*.bin"):
        os.remove(f)
    os.environ["AHMC_WAVE_TIMELINE_OUT"] = prefix
    os.environ["AHMC_NUTS_DRAW_BATCH"] = str(n_draws)      # one launch
    eng.run(k, n_adapts + n_draws, n_adapts, i_first=n_adapts + 1)
    eng.sync()
    os.environ.pop("AHMC_WAVE_TIMELINE_OUT", None)
    os.environ.pop("AHMC_NUTS_DRAW_BATCH", None)
    acc = eng.accum()
    eng.close()
    recs = []
    for f in sorted(glob.glob(prefix + ".*.mode0.bin")):
        a = np.fromfile(f, dtype=np.uint64).reshape(-1, 24)
        recs.append(a[a[:, 1] > 0])
        os.remove(f)
    for f in glob.glob(prefix + ".*.bin"):
        os.remove(f)
    a = np.concatenate(recs).astype(np.float64)
    leaf = a[:, 16].sum()
    out = {"case": name, "occupancy": tag, "chains": N, "waves_recorded": int(len(a)), "draws": n_draws, "leaf_steps_of_the_waves": leaf,
           "chain_leapfrogs": float(acc["total_n_steps"]), "merges_per_leaf_step": a[:, 17].sum() / leaf,
           "leaf_steps_per_doubling": leaf / a[:, 18].sum(), "leaf_steps_per_transition_of_a_wave": leaf / a[:, 19].sum(),
           "reintegration_steps_per_leaf_step": a[:, 4].sum() / leaf,
           "lockstep": a[:, 3].sum() / (cs["cpw"] * a[:, 2].sum()) if cs["cpw"] > 1 else 1.0}
    cyc = {s: a[:, 8 + i].sum() / leaf for i, s in enumerate(STAGES)}
    out["cycles_per_leaf_step"] = cyc
    out["cycles_per_leaf_step_total"] = sum(cyc.values())
    out["cycles_in_the_leaf_loop"] = sum(cyc[s] for s in STAGES[:5])
    out["us_per_leaf_step_total_at_2387_mhz"] = out["cycles_per_leaf_step_total"] / MHZ
    out["us_per_leaf_step_in_the_leaf_loop"] = out["cycles_in_the_leaf_loop"] / MHZ
    # wall clock of the waves (100 MHz counter) per leaf step: what the shader-clock sum must agree with
    out["us_per_leaf_step_by_the_wall_clock_of_the_waves"] = float(((a[:, 1] - a[:, 0]) * 1e-2).sum() / leaf)
    return out


def main():
    lib = A.load_hip_library()
    res = []
    for name in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["cfg2", "cfg3", "cfg5"]):
        cs = CASES[name]
        res.append(run(lib, name, cs["cpw"], 300, 200, "lone_wave"))
        print(json.dumps(res[-1]), flush=True)
        res.append(run(lib, name, cs["full"], 300 if name != "cfg5" else 150, 32 if name != "cfg5" else 8, "full"))
        print(json.dumps(res[-1]), flush=True)
        with open(os.path.join(OUT, f"leaf_latency_{name}.json"), "w") as f:
            json.dump({"build": "-DAHMC_LEAF_PROF=1 (scripts/build_variant.py lp); stamps = s_memtime fenced by sched_barrier, per wave, summed over its leaf steps",
                       "stages": STAGES, "cases": [r for r in res if r["case"] == name]}, f, indent=1)


if __name__ == "__main__":
    main()
