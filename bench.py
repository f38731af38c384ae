#!/usr/bin/env python
"""bench.py — leapfrog-steps/sec of the chain-batched NUTS hot path on MI355X.

Workload (BASELINE.json configs[1]): D=128 isotropic Gaussian,
DiagEuclideanMetric (per-chain M⁻¹), NUTS(δ=0.8) = MultinomialTS + GeneralisedNoUTurn(max_depth
10, Δ_max 1000) with StanHMCAdaptor, 65 536 chains per GPU, Float64, synthetic θ0 ~ U(0,1).

A "step" is ONE NUTS transition of all chains.  Setup (untimed): find_good_stepsize + `--adapt`
Stan-adaptation transitions.  Then W warm-up steps and exactly K timed steps, bracketed by barrier
+ synchronize; MAX over ranks; rank 0 prints one JSON line.  `value` = Σ n_steps over all chains
and ranks in the timed region ÷ that time.

In the sampling phase the engine runs a batch of transitions per launch of the dominant kernel
k_nuts (chains are independent, so there is no per-transition barrier; `nuts_batch` transitions,
the K steps split evenly over ⌈K / nuts_batch⌉ launches).  The roofline object is per LAUNCH:
algorithmic bytes of the leapfrogs one launch executes ÷ the launch's duration, measured by HIP
events the engine records around each k_nuts launch on its own stream (AHMC_INFO_NUTS_KERNEL_NS) —
the same quantity rocprofv3's kernel trace reports for those launches (profiles/r1_kernel_stats.csv,
"timed launches" rows; the `--stats` average also covers the single-transition launches of the
adaptation phase).  `timed_region_ms_on_stream` is the whole K-step region incl. the helpers
(k_normals, the log-domain redo pass).

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL); chains shard with no
data-path collective (weak scaling, 65 536 chains per GPU, Philox chain offset = rank·N); the
only collectives are the timing reductions and the final gather of per-dimension moments.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md chip table)


def algorithmic_bytes_per_leapfrog(D, diag, itemsize):
    """B_lf = (6·D + D_M)·sizeof(T) + 4·sizeof(T)"""
    return (6 * D + (D if diag else 0)) * itemsize + 4 * itemsize


def measured_traffic(D, N):
    """HBM bytes per launch of the dominant kernel from the PMC counters (FETCH_SIZE x2 on gfx950 +
    WRITE_SIZE, separate rocprofv3 passes: profiles/r1_hbm_traffic.json).  bench.py cannot run
    rocprofv3 on itself, so the committed measurement is reported — only for the workload it was
    taken on; otherwise null."""
    try:
        with open(os.path.join(ROOT, "profiles", "r1_hbm_traffic.json")) as f:
            t = json.load(f)
        if (D, N) != (128, 65536):
            return None
        return t["hbm_bytes_per_launch"], t.get("transitions_per_launch")
    except Exception:
        return None


def measured_issue(D, N):
    """What actually bounds the fused kernel: VALU issue.  SQ_ACTIVE_INST_VALU / SIMD cycles and instructions per
    leapfrog from the committed PMC passes (profiles/r1_sq_counters.json), for the workload they were taken on."""
    try:
        with open(os.path.join(ROOT, "profiles", "r1_sq_counters.json")) as f:
            d = json.load(f)["derived"]
        if (D, N) != (128, 65536):
            return None
        return {"valu_busy_fraction_of_simd_cycles": d["valu_busy_fraction_of_simd_cycles"],
                "valu_instructions_per_leapfrog": d["valu_instructions_per_leapfrog"],
                "mean_waves_per_simd": d["mean_waves_per_simd"], "source": "profiles/r1_sq_counters.json"}
    except Exception:
        return None


def build_engine(A, lib, D, N, seed, chain_offset, stream=0, device=0, dtype=np.float64):
    metric = A.DiagEuclideanMetric(np.ones((D, N), order="F"))
    h = A.Hamiltonian(metric, A.IsoGaussian(D))
    eng = A.Engine(h, N, dtype=dtype, rng=A.PhiloxRNG(seed, chain_offset), lib=lib, device=device, stream=stream)
    lf = A.Leapfrog(np.full(N, 0.1))
    eng.set_integrator(lf)
    th0 = np.random.default_rng(seed + chain_offset).random((D, N))
    eng.set_position(np.asfortranarray(th0))
    eng.find_good_stepsize()
    eng.adaptor_init(A.StanHMCAdaptor(A.MassMatrixAdaptor(metric), A.StepSizeAdaptor(0.8, lf)))
    kernel = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=10, delta_max=1000.0)))
    return eng, kernel


def usable_cores():
    """CPU cores this process may actually use: min(visible CPUs, cgroup-v2 CPU quota).  The GPU
    boxes show 256 logical CPUs but run the container under `cpu.max 1600000 100000` (16 cores);
    oversubscribing the quota only adds throttling (measured: 32 threads 1.5e7, 256 threads 3e6)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


def cpu_baseline(A, D, n_adapt, steps, seed, chains, threads):
    """The CPU oracle (scalar restatement of the reference, OpenMP over chains) on a bounded
    sample of the same workload, timed on this host's usable cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import build_oracle  # test infrastructure: used here only as the timed CPU baseline

    lib = A.CLib(build_oracle.build())
    lib.dll.ahmco_set_num_threads.restype = int
    threads = lib.dll.ahmco_set_num_threads(int(threads))
    eng, kernel = build_engine(A, lib, D, chains, seed, 0)
    eng.run(kernel, n_adapt, n_adapt)
    eng.run(kernel, 1, 0)
    t0 = time.perf_counter()
    eng.run(kernel, steps, 0)
    dt = time.perf_counter() - t0
    acc = eng.accum(moments=False)
    eng.close()
    return acc["total_n_steps"] / dt, dt, threads


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--chains", type=int, default=65536, help="chains per GPU")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--adapt", type=int, default=200, help="untimed Stan adaptation transitions")
    ap.add_argument("--seed", type=int, default=0x5EED0002)
    ap.add_argument("--cpu-chains", type=int, default=0, help="0 = 128 per host core, capped at --chains")
    ap.add_argument("--cpu-steps", type=int, default=1500)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64", help="f64 = the reference default and the headline; f32 = what the "
                    "reference's CUDA smoke test uses (test/CUDA/cuda.jl:18), reported for information")
    ap.add_argument("--ess", type=int, default=0, help="after the timed region: K more transitions with the draws kept in HBM, "
                    "ESS/sec (min over dimensions, Geyer estimator, 256-chain subset) reported under config.ess")
    args = ap.parse_args()

    import torch
    import ahmc_amd as A

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or os.environ.get("AHMC_BENCH_FORCE_DIST"):  # (the env switch exercises the RCCL path on one GPU)
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    lib = A.load_hip_library()  # raises if the HIP engine is not built: no fallback
    D, N = args.dim, args.chains
    stream = torch.cuda.Stream(device=local_rank)
    np_dtype = np.float32 if args.dtype == "f32" else np.float64
    eng, kernel = build_engine(A, lib, D, N, args.seed, rank * N, stream=stream.cuda_stream, device=local_rank, dtype=np_dtype)

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # setup (untimed): adaptation, then W warm-up steps in the sampling phase
    eng.run(kernel, args.adapt, args.adapt)
    if args.warmup > 0:
        eng.run(kernel, args.warmup, 0)
    barrier()

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    launches0, kns0 = eng.info("nuts_launches"), eng.info("nuts_kernel_ns")
    t0 = time.perf_counter()
    ev0.record(stream)
    eng.run(kernel, args.steps, 0)  # K transitions, accumulators reset at the first one
    ev1.record(stream)
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1)

    acc = eng.accum(moments=True)
    n_leap = acc["total_n_steps"]
    n_launches = max(1, eng.info("nuts_launches") - launches0)
    nuts_ns = eng.info("nuts_kernel_ns") - kns0
    ess_info = None
    if args.ess > 0 and rank == 0:
        # ESS/sec (BASELINE.json's secondary metric): draws (K, N, D) written by k_nuts straight into HBM
        from ahmc_amd.diagnostics import ess as ess_fn

        draws = torch.empty((args.ess, N, D), dtype=torch.float64 if args.dtype == "f64" else torch.float32, device=f"cuda:{local_rank}")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng.run(kernel, args.ess, 0, samples_out=draws.data_ptr())
        eng.sync()
        dt_ess = time.perf_counter() - t1
        sub = draws[:, :256, :].cpu().numpy()                      # (K, 256 chains, D)
        e = ess_fn(sub, axis=0)                                    # (256, D) per chain and dimension
        per_chain = e.mean(axis=0)                                 # mean over the subset, per dimension
        ess_info = {"draws_per_chain": args.ess, "seconds": dt_ess, "min_over_dims_ess_per_chain": float(per_chain.min()),
                    "ess_per_sec": float(per_chain.min() * N * world / dt_ess),
                    "estimator": "Geyer initial monotone sequence on FFT autocorrelation, per chain, 256-chain subset"}
        del draws
    tt = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
    tn = torch.tensor([float(n_leap), float(acc["n_divergent"])], dtype=torch.float64, device=f"cuda:{local_rank}")
    # per-dimension pooled moments of this shard; the final gather over RCCL —
    # the same host code the 2-rank gloo test exercises (advancedhmc.jl_amd/shard.py)
    from ahmc_amd.shard import gather_moments, pooled_moments

    mom_np, n_draws = pooled_moments(acc["sum_theta"], acc["sumsq_theta"], acc["n_transitions"] * N)
    mom = torch.tensor(mom_np, dtype=torch.float64, device=f"cuda:{local_rank}")
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(tn, op=dist.ReduceOp.SUM)
    mean, var, _ = gather_moments(dist, mom, n_draws, f"cuda:{local_rank}")
    dt_max = float(tt.item())
    total_leap = float(tn[0].item())

    if rank == 0:
        B_lf = algorithmic_bytes_per_leapfrog(D, True, 8 if args.dtype == "f64" else 4)
        per_launch_s = nuts_ns / 1e9 / n_launches  # HIP events around k_nuts, engine stream
        achieved = (n_leap / n_launches) * B_lf / per_launch_s / 1e9  # this rank's dominant kernel
        traffic = measured_traffic(D, N)
        if traffic is not None:  # measured per launch of `transitions_per_launch`; rescale to this run's launches
            hbm, tpl = traffic
            traffic = hbm * (args.steps / n_launches) / tpl if tpl else hbm
        out = {
            "metric": "leapfrog-steps/sec (whole node) at n_chains x D",
            "value": total_leap / dt_max,
            "unit": "leapfrog-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"cfg2: D={D} iso Gaussian, DiagEuclideanMetric per-chain, NUTS(0.8) MultinomialTS+GeneralisedNoUTurn "
                            f"max_depth 10, StanHMCAdaptor ({args.adapt} untimed adaptation steps), {N} chains/GPU",
                "chains_per_gpu": N, "D": D, "parallelism": f"chain-shard x{world}",
                "mean_leapfrogs_per_transition": total_leap / (args.steps * N * world),
                "divergent": float(tn[1].item()),
                "max_abs_mean": float(np.abs(mean).max()), "max_abs_var_minus_1": float(np.abs(var - 1).max()),
                "ess": ess_info,
            },
            "roofline": {
                "bound": "hbm", "kernel": "k_nuts<%s,%d,%d,mode 0,iso>" % ("double" if args.dtype == "f64" else "float", eng.info("group_lanes"), eng.info("elems_per_lane")),
                "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_leapfrog": B_lf, "avg_launch_ms": per_launch_s * 1e3,
                "launches": n_launches, "transitions_per_launch": args.steps / n_launches,
                "timed_region_ms_on_stream": kernel_ms,
                "leapfrogs_per_launch": n_leap / n_launches,
                "note": "state-through-memory model of SURVEY 8d: a fused kernel keeps the trajectory in registers/LDS, so frac > 1 "
                        "is expected; the kernel is VALU-issue bound (see issue_bound)",
                "issue_bound": measured_issue(D, N),
            },
        }
        if not args.no_cpu_baseline and world == 1:  # (the contract: rank 0 at N = 1 only)
            try:
                cores = usable_cores()
                # bounded sample sized to the host: 256 chains per usable core (capped at the GPU's own
                # N), the same adaptation, then `cpu_steps` timed sampling transitions (~10-30 s in all)
                cpu_chains = args.cpu_chains or min(N, 256 * cores)
                cpu_steps = args.cpu_steps
                t_all = time.perf_counter()
                v, cdt, cores = cpu_baseline(A, D, args.adapt, cpu_steps, args.seed, cpu_chains, cores)
                out["cpu_baseline"] = {
                    "value": v, "unit": "leapfrog-steps/s", "cores": cores, "kind": "port",
                    "sample": f"{cpu_chains} chains x D={D}, same kernel/adaptor, {cpu_steps} timed transitions "
                              f"({cdt:.1f} s) after {args.adapt} adaptation steps ({time.perf_counter() - t_all:.1f} s in all); "
                              f"C++ restatement of the reference (oracle/), OpenMP over chains, {cores} threads = the container's "
                              f"CPU quota ({os.cpu_count()} logical CPUs visible), not Julia",
                }
            except Exception as ex:  # the baseline leg must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "error": repr(ex)}
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
