#!/usr/bin/env python
"""bench.py — leapfrog-steps/sec of the chain-batched NUTS hot path on MI355X, end to end.

Workload (default `--config cfg2` = BASELINE.json configs[1]): D=128 isotropic Gaussian,
DiagEuclideanMetric (per-chain M⁻¹, init ones), NUTS(δ=0.8) = MultinomialTS + GeneralisedNoUTurn(max_depth 10,
Δ_max 1000) with StanHMCAdaptor(75/50/25), 65 536 chains per GPU, Float64, synthetic θ0 ~ U(0,1).

What is timed is the reference's `sample` loop (src/sampler.jl:182-228) — the loop that contains `adapt!`:
    sample(rng, h, κ, θ0, n_samples = n_adapts + n_draws, StanHMCAdaptor, n_adapts)
i.e. the warm-up transitions WITH their adaptation and the post-warm-up draws.  SURVEY §8d quotes the metric on
1 000 + 1 000 transitions; `--steps K` scales that: a "step" is `--transitions-per-step` (default 100) consecutive
transitions of all chains, the first half of the K steps adapting, the second half drawing, so the driver's K = 20
is exactly the 1 000 + 1 000 run.  Untimed, as §8d says ("excluding setup and H2D of the initial state"): engine
creation, θ0 upload, find_good_stepsize.  W warm-up steps (same loop, throw-away engine) run first; then the timed
region — exactly K steps — bracketed by barrier + synchronize, MAX over ranks, repeated `--repeats` times (default:
until >= 1 s of timed work and at least 3 runs when they are short), the MEDIAN run is reported (all runs listed).
`value` = Σ n_steps over all chains, ranks and both phases of that run ÷ its wall time.  The post-adaptation rate (the
round-1 headline) and the warm-up rate are sub-fields of `config`.

`roofline` is the roof that binds the dominant kernel: VALU issue.  A fused trajectory kernel keeps the state in
registers, so SURVEY §8d's state-through-memory byte model is not a roof for it (a fraction > 1 in round 1); it is
still reported as `hbm_model_frac`.  VALU roof: wave-instructions per leapfrog of the shipped kernel × leapfrogs of
the launches ÷ their duration (HIP events recorded by the engine around every launch of the kernel on its stream)
÷ the MIX-WEIGHTED issue peak of that kernel: every instruction class priced at its nominal issue cycles — 2 (32-bit add /
xor / mov), 4 (f64 arithmetic, DPP moves, 64-bit moves, multiplies, compares, selects), 8 (permlane swaps), 16 (f64 rcp) per
wave64 instruction, the class of each instruction type read off its MEASURED rate on this chip (scripts/probe/valu_rate.hip →
profiles/r3_valu_rate.json) — at 2.4 GHz × 1024 SIMDs, weighted by the kernel's dynamic instruction mix
(SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 / INT32 / INT64 / CVT per leapfrog; scripts/valu_mix.py).  (Priced at the measured
single-class rates themselves the roof comes out 10 % lower and the kernel exceeds it: a pure v_fma_f64 stream runs 27 % under
nominal, a mixed stream does not.)  Instructions per leapfrog and HBM bytes come from
rocprofv3 PMC passes over THIS command, committed as profiles/counters_at_head.json together with the digest of the
kernel sources they were taken on: if the digest does not match the library in use the counters are stale and
`frac`, `traffic` are null (never rescaled from an old measurement).

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL).  `--gpus N` without a torchrun environment
re-executes itself under `python -m torch.distributed.run --nproc-per-node N`.  Chains shard with no data-path
collective (weak scaling, Philox chain offset = rank·chains); the only collectives are the timing reductions and the
final gather of per-dimension moments, which goes through the C ABI (ahmc_gather_moments over RCCL).
"""
import argparse
import ctypes
import json
import math
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md chip table
HBM_ACHIEVABLE_GBS = 6290.0    # ibid.: the measured copy rate (SURVEY 8d reports the model fraction against both)
VALU_PEAK_GINSTR = 1024 * 2.4e9 / 4 / 1e9   # 256 CUs × 4 SIMDs, 2.4 GHz, one wave64 VALU instruction per 4 cycles
F64_MFMA_PEAK_TFLOPS = 78.6    # dense f64 matrix peak
F32_MFMA_PEAK_TFLOPS = 157.3

CONFIGS = {
    # name: D, chains per GPU, target, metric, adaptor, GPUs the config is quoted on
    "cfg2": dict(D=128, N=65536, target="iso", metric="diag", adaptor="stan", quoted_gpus=1, seed=0x5EED0002,
                 text="cfg2: D=128 iso Gaussian, DiagEuclideanMetric per-chain, NUTS(0.8) MultinomialTS+GeneralisedNoUTurn max_depth 10, StanHMCAdaptor"),
    "cfg3": dict(D=32, N=65536, target="funnel", metric="diag", adaptor="stan", quoted_gpus=1, seed=0x5EED0003,
                 text="cfg3: D=32 Neal's funnel, DiagEuclideanMetric per-chain, NUTS(0.8, max_depth 10), StanHMCAdaptor"),
    "cfg4": dict(D=512, N=8192, target="dense", metric="dense", adaptor="stepsize", quoted_gpus=4, seed=0x5EED0004,
                 text="cfg4 (one GPU's shard of 32 768 chains / 4 GPUs): D=512 correlated Gaussian Sigma_ij=0.9^|i-j| (logdensity as a GEMM), "
                      "shared DenseEuclideanMetric, NUTS(0.8), StepSizeAdaptor"),
    "cfg5": dict(D=2048, N=32768, target="hier", metric="diag", adaptor="stan", quoted_gpus=8, seed=0x5EED0005,
                 text="cfg5 (one GPU's shard of 262 144 chains / 8 GPUs): D=2048 hierarchical Gaussian, DiagEuclideanMetric per-chain, NUTS(0.8), StanHMCAdaptor"),
}


def algorithmic_bytes_per_leapfrog(D, metric, itemsize):
    """B_lf = (6·D + D_M)·sizeof(T) + 4·sizeof(T)"""
    return (6 * D + (D if metric == "diag" else 0)) * itemsize + 4 * itemsize


CONFIG_FAMILY = {"cfg2": 0, "cfg3": 2, "cfg5": 3}   # the log-density family (TK) whose instantiation units hold a config's k_nuts


def dense_counter_bytes_per_leapfrog():
    """cfg4: bytes beyond L2 per useful chain-leapfrog of ALL the dense engine's kernels, from the committed PMC passes
    (profiles/counters_at_head.json: configs.cfg4 — 2 x FETCH_SIZE + WRITE_SIZE over every dispatch; round 6: taken on the bench's own
    run, 300 + 300 transitions — `command` in that file — so the bytes and the TFLOP/s describe the same workload; rounds 3-5 used a
    10 + 10-transition run).  Valid only for the device code it was taken on: the digest of the unit that holds the dense engine's
    kernels (`api`, libahmc_hip.so.kdigests) must match, else (None, reason) — never a figure from other kernels.  Matrices missing
    L2 included; not an in-run counter (PMC passes serialise the dispatches)."""
    try:
        c = json.load(open(os.path.join(ROOT, "profiles", "counters_at_head.json")))["configs"]["cfg4"]
    except Exception:  # noqa: BLE001
        return None, "profiles/counters_at_head.json has no counters for cfg4"
    try:
        have = json.load(open(os.path.join(ROOT, "advancedhmc.jl_amd", "csrc", "libahmc_hip.so.kdigests"))).get("api")
    except (OSError, ValueError):
        have = None
    if not have or c.get("unit_digest") != have:
        return None, "profiles/counters_at_head.json: the counters of cfg4 were taken on other device code (digest mismatch): stale, not used"
    return c["hbm_d_vectors_per_chain_leapfrog_all_kernels"] * c.get("d_vector_bytes", 4096.0), c.get("source", "profiles/counters_at_head.json")


def sources_digest(config="cfg2"):
    """digest of the DEVICE code of the two instantiation units that hold `config`'s trajectory kernels in the shipped
    libahmc_hip.so (advancedhmc.jl_amd/build.py: config_digest — the stamp libahmc_hip.so.kdigests travels with the library)"""
    import hashlib
    try:
        d = json.load(open(os.path.join(ROOT, "advancedhmc.jl_amd", "csrc", "libahmc_hip.so.kdigests")))
        tk = CONFIG_FAMILY[config]
        a, b = d[f"inst_f64_t{tk}"], d[f"inst_f64_t{tk}b"]
        return hashlib.sha256(f"{a}|{b}".encode()).hexdigest()
    except (OSError, ValueError, KeyError):
        return None


def counters_at_head(config):
    """PMC counters of the dominant kernels, valid only for the kernel sources they were taken on"""
    try:
        with open(os.path.join(ROOT, "profiles", "counters_at_head.json")) as f:
            d = json.load(f)
    except Exception:
        return None, "profiles/counters_at_head.json is missing"
    c = d.get("configs", {}).get(config)
    if not c:
        return None, f"profiles/counters_at_head.json has no counters for {config}"
    if c.get("unit_digest") != sources_digest(config):
        return None, f"profiles/counters_at_head.json: the counters of {config} were taken on other device code (digest mismatch): stale, not used"
    return c, d.get("source", "profiles/counters_at_head.json")


def build_engine(A, lib, cfg, N, seed, chain_offset, stream=0, device=0, dtype=np.float64, n_reserve=0):
    """sample_init + find_good_stepsize + adaptor: the untimed setup of the reference's call sequence
    (src/abstractmcmc.jl:131-166: make_step_size → find_good_stepsize, make_adaptor, sample_init)"""
    D = cfg["D"]
    if cfg["metric"] == "dense":
        metric = A.DenseEuclideanMetric(np.eye(D, order="F"))
    else:
        metric = A.DiagEuclideanMetric(np.ones((D, N), order="F"))
    if cfg["target"] == "iso":
        target = A.IsoGaussian(D)
    elif cfg["target"] == "funnel":
        target = A.Funnel(D)
    elif cfg["target"] == "hier":
        target = A.HierGaussian(D)
    else:
        idx = np.arange(D)
        target = A.DenseGaussian(np.linalg.inv(0.9 ** np.abs(idx[:, None] - idx[None, :])))
    h = A.Hamiltonian(metric, target)
    eng = A.Engine(h, N, dtype=dtype, rng=A.PhiloxRNG(seed, chain_offset), lib=lib, device=device, stream=stream)
    lf = A.Leapfrog(np.full(N, 0.1))
    eng.set_integrator(lf)
    th0 = np.random.default_rng(seed + chain_offset).random((D, N))
    eng.set_position(np.asfortranarray(th0))
    eng.find_good_stepsize()
    if cfg["adaptor"] == "stan":
        eng.adaptor_init(A.StanHMCAdaptor(A.MassMatrixAdaptor(metric), A.StepSizeAdaptor(0.8, lf)))
    else:
        eng.adaptor_init(A.StepSizeAdaptor(0.8, lf))
    kernel = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=10, delta_max=1000.0)))
    if n_reserve > 0:
        eng.reserve(kernel, n_reserve)   # setup: the launch buffers of the announced run (ahmc_sample_reserve), not inside its first launch
    return eng, kernel


def sample_loop(eng, kernel, n_adapts, n_draws, sync, draws_ptr=None):
    """the timed region: sample(…, n_adapts + n_draws, adaptor, n_adapts).  Two ahmc_sample calls — the adapting
    transitions, then the draws — so that each phase's Σ n_steps can be read; the iteration counter, the adaptor
    and the state carry over, i.e. it is the one loop of src/sampler.jl:182-228.  `draws_ptr`: the (D, N, n_draws) device
    buffer that receives θ after EVERY post-warm-up transition — what the reference's `sample` returns
    (src/sampler.jl:224-227, drop_warmup) — written inside the timed region by the trajectory kernel itself."""
    info0 = {k: eng.info(k) for k in ("nuts_launches", "nuts_kernel_ns", "nuts_warm_launches", "nuts_warm_kernel_ns")}
    sync()
    t0 = time.perf_counter()
    if n_adapts > 0:
        eng.run(kernel, n_adapts, n_adapts)
    eng.sync()
    t1 = time.perf_counter()
    acc_a = eng.accum(moments=False) if n_adapts > 0 else {"total_n_steps": 0, "n_divergent": 0}
    eng.run(kernel, n_draws, 0, samples_out=draws_ptr)
    sync()
    t2 = time.perf_counter()
    acc_d = eng.accum(moments=True)
    info = {k: eng.info(k) - v for k, v in info0.items()}
    return {"dt": t2 - t0, "dt_adapt": t1 - t0, "dt_draw": t2 - t1, "leap_adapt": acc_a["total_n_steps"], "leap_draw": acc_d["total_n_steps"],
            "div_adapt": acc_a["n_divergent"], "acc": acc_d, "info": info}


def usable_cores():
    """CPU cores this process may actually use: min(visible CPUs, cgroup-v2 CPU quota).  The GPU boxes show 256 logical
    CPUs but run the container under `cpu.max 1600000 100000` (16 cores); oversubscribing the quota only adds throttling."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


def cpu_baseline(A, cfg, n_adapts, n_draws, seed, chains, threads):
    """The CPU oracle (scalar restatement of the reference, OpenMP over chains) on a bounded sample of the SAME loop
    (adapting transitions + draws, same adaptor), timed on this host's usable cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import build_oracle  # test infrastructure: used here only as the timed CPU baseline

    lib = A.CLib(build_oracle.build())
    lib.dll.ahmco_set_num_threads.restype = int
    threads = lib.dll.ahmco_set_num_threads(int(threads))
    eng, kernel = build_engine(A, lib, cfg, chains, seed, 0)
    r = sample_loop(eng, kernel, n_adapts, n_draws, eng.sync)
    eng.close()
    return (r["leap_adapt"] + r["leap_draw"]) / r["dt"], r["leap_draw"] / max(r["dt_draw"], 1e-9), r["dt"], threads


def respawn_under_torchrun(n):
    """`bench.py --gpus N` outside a torchrun environment: become the launcher (one rank per GPU)"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execvpe(cmd[0], cmd, env)


def launch_check(args):
    """Plumbing check for a machine without GPUs (tests/test_bench_contract.py): the ranks `--gpus N` asks for exist,
    rendezvous over gloo, and rank 0 reports their number.  Runs no engine and no compute."""
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.ones(1)
        dist.all_reduce(t)
        seen = int(t.item())
        dist.destroy_process_group()
    else:
        seen = 1
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_seen": seen, "gpus_requested": args.gpus}))
    return 0 if seen == world == args.gpus else 1


class SetupFailed(RuntimeError):
    """a config's setup failed on some rank and ALL ranks know (they agreed before the first collective of its timed loop)"""


def run_config(ctx, cfg_name, steps, T, warm_trans, repeats, cpu_budget_s, dim=0, chains=0):
    """One bench line for one config: W untimed transitions on a throw-away engine, then the timed sample loop (median of
    `repeats` runs on fresh engines), the final gather, the roofline of the dominant kernel and — rank 0 of a one-GPU run — the CPU
    baseline on a bounded sample of the same loop.  Returns the JSON object on rank 0, None elsewhere."""
    A, lib, torch, dist, args = ctx["A"], ctx["lib"], ctx["torch"], ctx["dist"], ctx["args"]
    rank, local_rank, world, stream = ctx["rank"], ctx["local_rank"], ctx["world"], ctx["stream"]
    cfg = dict(CONFIGS[cfg_name])
    if dim:
        cfg["D"] = dim
    D, N = cfg["D"], (chains or cfg["N"])
    seed = args.seed or cfg["seed"]
    n_total = steps * T
    n_adapts = int(round(n_total * args.adapt_fraction))
    n_draws = n_total - n_adapts
    np_dtype = np.float32 if args.dtype == "f32" else np.float64
    itemsize = 4 if args.dtype == "f32" else 8
    dev = f"cuda:{local_rank}"

    def make():
        return build_engine(A, lib, cfg, N, seed, rank * N, stream=stream.cuda_stream, device=local_rank, dtype=np_dtype, n_reserve=max(n_adapts, n_draws))

    def barrier_for(eng):
        def f():
            eng.sync()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
        return f

    # the draws of the timed region: θ after every post-warm-up transition of every chain, (D, N, n_draws) in HBM — what the
    # reference's `sample` returns (src/sampler.jl:224-227); written by the trajectory kernel inside the timed region
    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    try:
        draws = None if args.no_draws_out else torch.empty((n_draws, N, D), dtype=tdt, device=dev)
    except Exception as ex:  # noqa: BLE001  (out of memory: the other ranks must hear of it)
        if dist is None:
            raise
        draws = ex

    def agree(ok, what):
        """N > 1: every rank learns whether ALL ranks got through `what` (allocations that depend on a rank's free memory) BEFORE the
        first collective of the timed loop — a rank that raised alone would leave the others waiting in a barrier for ever"""
        if dist is not None:
            t = torch.tensor([1.0 if ok else 0.0], device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok_all = bool(t.item() > 0.5)
        else:
            ok_all = ok
        if not ok_all:
            raise SetupFailed(f"{cfg_name}: {what} failed on " + ("this rank" if not ok else "another rank"))

    agree(not isinstance(draws, Exception), "the draws buffer")

    # W untimed warm-up transitions of the same loop (code objects, allocator, clocks, first touch of the draws buffer) on a throw-away engine
    if warm_trans > 0:
        try:
            eng, kernel = make()
            err = None
        except Exception as ex:  # noqa: BLE001
            eng, err = None, ex
        if err is not None and dist is None:
            raise err
        agree(err is None, f"engine setup ({err!r})" if err else "engine setup")
        nwa = int(round(warm_trans * args.adapt_fraction))
        nwd = max(1, warm_trans - nwa)
        sample_loop(eng, kernel, nwa, nwd, barrier_for(eng), draws_ptr=draws.data_ptr() if (draws is not None and nwd <= n_draws) else None)
        eng.close()

    want_ess = draws is not None and args.ess != 0 and n_draws >= 8   # every config: one thread per (dimension, chain) series, any D
    ess_buf = torch.empty((N, D), dtype=tdt, device=dev) if want_ess else None

    runs = []
    eng = None
    while True:
        if eng is not None:
            eng.close()
        eng, kernel = make()                      # untimed setup: create, θ0, find_good_stepsize, adaptor
        r = sample_loop(eng, kernel, n_adapts, n_draws, barrier_for(eng), draws_ptr=draws.data_ptr() if draws is not None else None)
        if want_ess:
            # ESS of THIS run's draws (all chains, all dimensions), reduced on the device through the C ABI (ahmc_ess):
            # per (dimension, chain) series Geyer's initial monotone sequence; per dimension the mean over chains; min over dimensions
            eng._call("ahmc_ess", ctypes.c_void_p(draws.data_ptr()), int(n_draws), ctypes.c_void_p(ess_buf.data_ptr()))
            eng.sync()
            r["ess_per_draw"] = float(ess_buf.mean(dim=0).min().item()) / n_draws
        tt = torch.tensor([r["dt"], r["dt_adapt"], r["dt_draw"]], dtype=torch.float64, device=dev)
        tn = torch.tensor([float(r["leap_adapt"]), float(r["leap_draw"]), float(r["acc"]["n_divergent"]), float(r["div_adapt"]), r.get("ess_per_draw", 0.0)],
                          dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dist.all_reduce(tn, op=dist.ReduceOp.SUM)
        r["dt_max"], r["dt_adapt_max"], r["dt_draw_max"] = (float(x) for x in tt.tolist())
        r["leap_adapt_all"], r["leap_draw_all"], r["div_all"], r["div_adapt_all"], ess_sum = (float(x) for x in tn.tolist())
        r["ess_per_draw_all"] = ess_sum / world
        r["value"] = (r["leap_adapt_all"] + r["leap_draw_all"]) / r["dt_max"]
        r["draw_launch_length"] = eng.info("nuts_draw_batch")
        r["dense_launches"] = {k: eng.info(k) for k in ("dense_epoch_launches", "dense_gemm_launches", "dense_gemm_small_launches")}
        runs.append(r)
        spent = sum(x["dt_max"] for x in runs)
        want = repeats if repeats > 0 else (3 if runs[0]["dt_max"] < 4.0 else 1)
        # every rank takes the same decision: dt_max is the all-reduced time
        if len(runs) >= want and (repeats > 0 or spent >= 1.0):
            break
        if len(runs) >= 50:
            break
    order = sorted(range(len(runs)), key=lambda i: runs[i]["value"])
    med = runs[order[len(order) // 2]]   # median run (the engine of the LAST run is still open for the gather)

    # final gather of the pooled per-dimension moments of the last run's draws through the C ABI (RCCL all-reduce inside)
    from ahmc_amd.shard import EngineComm

    comm = EngineComm(eng, dist, dev)
    g = comm.gather_moments()
    ci = eng.comm_info()               # what the communicator's own all-reduces counted when it was attached
    mean, var = g["mean"], g["var"]

    ess_info = None
    if want_ess:
        e1 = med["ess_per_draw_all"]
        ess_info = {"ess_per_sec": e1 * n_draws * N * world / med["dt_max"],
                    "ess_per_sec_sampling_phase_only": e1 * n_draws * N * world / med["dt_draw_max"],
                    "ess_per_draw_min_over_dims": e1, "estimated_on_draws_per_chain": n_draws, "chains_used": N * world,
                    "estimator": "Geyer initial monotone sequence on the autocovariances of every (dimension, chain) series (ahmc_ess, device "
                                 "reduction over ALL chains); per dimension the mean over chains, then the min over dimensions; "
                                 "the reference computes no ESS (MCMCChains.jl does): parity unpinned",
                    "definition": "ESS of the timed run's own post-warm-up draws (the buffer the timed region filled) / wall time of its whole sample loop"}
    del draws, ess_buf
    torch.cuda.empty_cache()

    out = None
    if rank == 0:
        B_lf = algorithmic_bytes_per_leapfrog(D, cfg["metric"], itemsize)
        info = med["info"]
        G, E = eng.info("group_lanes"), eng.info("elems_per_lane")
        tname = "double" if args.dtype == "f64" else "float"
        counters, counters_src = counters_at_head(cfg_name) if (args.dtype == "f64" and not dim and not chains) else (None, "counters are per config at its default size and f64")
        # what the leapfrog itself needs (src/integrator.jl:231-243 on E elements per lane): r −= ϵ/2·g, θ += ϵ·(M⁻¹r), the density's
        # gradient and value, r −= ϵ/2·g′, ℓκ — ≈ 7 f64 VALU instructions per element; everything above that is the tree
        # (reductions, weights, merges, U-turn tests, RNG) and the kernel's bookkeeping
        useful_floor = 7.0 * E * max(1, G // 64)   # (per chain: a multi-wave chain's leapfrog is issued by G / 64 waves — round 4 counted one)

        def kernel_roof(label, mode, launches, kns, leap):
            """VALU-issue roof of one instantiation of k_nuts from this run's launches (HIP events) and the counters at HEAD"""
            if launches <= 0 or kns <= 0:
                return None
            per_launch_s = kns / 1e9 / launches
            lf_per_s = leap / (kns / 1e9)
            o = {"kernel": f"k_nuts<{tname},{G},{E},mode {mode}>", "phase": label, "launches": launches, "avg_launch_ms": per_launch_s * 1e3,
                 "leapfrogs_per_launch": leap / launches, "leapfrogs_per_s_in_kernel": lf_per_s,
                 "hbm_model_bytes_per_leapfrog": B_lf, "hbm_model_frac": lf_per_s * B_lf / 1e9 / HBM_PEAK_GBS}
            c = (counters or {}).get(f"mode{mode}")
            if c:
                cpw = max(1, 64 // G)   # chains per wave: the counters are per CHAIN-leapfrog, a wave instruction serves cpw chains
                o["valu_instructions_per_leapfrog"] = c["valu_per_leapfrog"]
                o["useful_valu_floor_per_leapfrog"] = useful_floor / cpw
                o["valu_efficiency"] = useful_floor / cpw / c["valu_per_leapfrog"]
                o["achieved"] = lf_per_s * c["valu_per_leapfrog"] / 1e9
                o["peak"] = c.get("valu_peak_mix_gwave_instr_per_s") or VALU_PEAK_GINSTR
                o["peak_is_mix_weighted"] = bool(c.get("valu_peak_mix_gwave_instr_per_s"))
                o["frac"] = o["achieved"] / o["peak"]
                o["frac_of_uniform_4_cycle_peak"] = o["achieved"] / VALU_PEAK_GINSTR
                o["valu_mix"] = c.get("valu_peak_mix")
                o["traffic"] = c.get("hbm_bytes_per_leapfrog") and c["hbm_bytes_per_leapfrog"] * leap / launches
                o["hbm_measured_frac"] = c.get("hbm_bytes_per_leapfrog") and lf_per_s * c["hbm_bytes_per_leapfrog"] / 1e9 / HBM_PEAK_GBS
                o["valu_busy_fraction_under_rocprof"] = c.get("valu_busy")
            else:
                o["achieved"] = o["frac"] = o["traffic"] = None
            return o

        roof = None
        if cfg["metric"] != "dense":
            rd = kernel_roof("draws", 0, info["nuts_launches"], info["nuts_kernel_ns"], med["leap_draw"])
            rw = kernel_roof("warm-up (adapt! inside the kernel)", 3, info["nuts_warm_launches"], info["nuts_warm_kernel_ns"], med["leap_adapt"])
            both = [x for x in (rd, rw) if x]
            both.sort(key=lambda x: -x["launches"] * x["avg_launch_ms"])  # dominant = more device time in the timed region
            if both:
                dom = both[0]
                roof = {"bound": "valu", "unit": "Gwave-instr/s", "peak": dom.get("peak", VALU_PEAK_GINSTR), "achieved": dom["achieved"], "frac": dom["frac"],
                        "traffic": dom["traffic"], "kernel": dom["kernel"], "counters": counters_src,
                        # the HBM roof beside the issue roof: SURVEY 8(d)'s state-through-memory model at this kernel's rate, and
                        # the bytes the counters saw beyond L2 at the same rate, both as fractions of 8 TB/s
                        "hbm_model_frac": dom.get("hbm_model_frac"), "hbm_measured_frac": dom.get("hbm_measured_frac"),
                        "hbm_peaks_gbs": {"spec": HBM_PEAK_GBS, "achievable_copy": HBM_ACHIEVABLE_GBS,
                                          "note": "the two fractions are of the spec peak (SURVEY 8d); x %.3f for the achievable copy rate" % (HBM_PEAK_GBS / HBM_ACHIEVABLE_GBS)},
                        "useful_valu_floor_per_leapfrog": dom.get("useful_valu_floor_per_leapfrog"), "valu_efficiency": dom.get("valu_efficiency"),
                        "peak_definition": ("mix-weighted VALU issue peak of this kernel: N / sum_c n_c * cycles_c over its dynamic instruction classes at 2.4 GHz x "
                                            "1024 SIMDs, cycles_c in {2,4,8,16} per wave64 instruction = the class of each instruction type by its MEASURED rate "
                                            "on the MI355X (scripts/probe/valu_rate.hip, profiles/r3_valu_rate.json; scripts/valu_mix.py)"
                                            if dom.get("peak_is_mix_weighted") else
                                            "uncalibrated: 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 VALU instruction (no class mix for this kernel)"),
                        "dominant": dom, "other": both[1] if len(both) > 1 else None,
                        "device_time_share_of_timed_region": sum(x["launches"] * x["avg_launch_ms"] for x in both) / 1e3 / med["dt"]}
        else:
            F_lf = 4 * D * D
            dense_bytes, dense_src = dense_counter_bytes_per_leapfrog() if (args.dtype == "f64" and not dim) else (None, "counters are for cfg4 at D = 512, f64")
            tf = (med["leap_adapt"] + med["leap_draw"]) * F_lf / med["dt"] / 1e12
            peak = F64_MFMA_PEAK_TFLOPS if args.dtype == "f64" else F32_MFMA_PEAK_TFLOPS
            roof = {"bound": "mfma", "unit": "TFLOP/s", "peak": peak, "achieved": tf, "frac": tf / peak, "traffic": None,
                    "kernel": "k_dense_epoch + k_dgemm / k_d_tree2 tails (whole timed region)",
                    "kernel_note": ("k_dense_epoch: chain-complete workgroups — both products, the half-steps and the trees of 32 chains per workgroup, 64 global "
                                    "steps per launch; k_dgemm / k_d_tree2 for the tails of the batches"),
                    "launches_since_create": runs[-1].get("dense_launches"),
                    "hbm_bytes_per_leapfrog_by_counters": dense_bytes, "counters": dense_src,
                    "hbm_model_frac": (med["leap_adapt"] + med["leap_draw"]) / med["dt"] * algorithmic_bytes_per_leapfrog(D, "dense", itemsize) / 1e9 / HBM_PEAK_GBS,
                    "hbm_measured_frac": dense_bytes and (med["leap_adapt"] + med["leap_draw"]) / med["dt"] * dense_bytes / 1e9 / HBM_PEAK_GBS,
                    "algorithmic_flops_per_leapfrog": F_lf,
                    "note": "useful leapfrogs x 4 D^2 / wall time of the whole loop (tree kernel, momenta and adaptation included)"}
        out = {
            "metric": "leapfrog-steps/sec (whole node) at n_chains x D",
            "value": med["value"],
            "unit": "leapfrog-steps/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": med["dt_max"] / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"{cfg['text']}, {N} chains/GPU, D={D}; timed = the whole sample loop: {n_adapts} adapting transitions + {n_draws} draws "
                            f"(find_good_stepsize and the initial H2D are setup, untimed)",
                "chains_per_gpu": N, "D": D, "parallelism": f"chain-shard x{world}",
                "ranks_seen": ci["ranks_seen"], "chains_total_seen": ci["chains_total"],
                "transitions_per_step": T, "n_adapts": n_adapts, "n_draws": n_draws,
                "untimed_warmup_transitions": warm_trans,
                "leapfrogs": {"adapt": med["leap_adapt_all"], "draw": med["leap_draw_all"]},
                "fits_quoted_config": world == cfg["quoted_gpus"],
                "runs": [x["value"] for x in runs], "reported": "median run",
                "draw_launch_length_found_by_the_engine": med.get("draw_launch_length"),
                "warmup_phase": {"value": med["leap_adapt_all"] / med["dt_adapt_max"] if n_adapts else None,
                                 "ms_per_transition": med["dt_adapt_max"] / max(n_adapts, 1) * 1e3,
                                 "mean_leapfrogs_per_transition": med["leap_adapt_all"] / max(n_adapts * N * world, 1),
                                 "divergent": med["div_adapt_all"]},
                "post_adaptation": {"value": med["leap_draw_all"] / med["dt_draw_max"],
                                    "ms_per_transition": med["dt_draw_max"] / n_draws * 1e3,
                                    "mean_leapfrogs_per_transition": med["leap_draw_all"] / (n_draws * N * world),
                                    "divergent": med["div_all"]},
                "max_abs_mean": float(np.abs(mean).max()), "max_abs_var_minus_1": float(np.abs(var - 1).max()) if cfg["target"] == "iso" else None,
                "gathered_draws": g["n_draws"], "gather": g["how"],
                "draws_materialised_in_timed_region": (None if args.no_draws_out else
                                                       {"shape_D_N_K": [D, N, n_draws], "gib": D * N * n_draws * itemsize / 2**30, "where": "device (HBM)"}),
                "ess": ess_info,
            },
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1 and cpu_budget_s > 0:  # (the contract: rank 0 at N = 1 only)
            try:
                cores = usable_cores()
                cpu_chains = args.cpu_chains or min(N, 256 * cores)
                # sized for ~cpu_budget_s: the oracle does ≈1.3e6 leapfrog/s per core on cfg2 (scales with 128 / D; the dense engine's
                # oracle pays 2 D² flops per leapfrog: ≈ 2e9 flop/s per core), ≈35 leapfrogs per transition
                lpt = max(out["config"]["post_adaptation"]["mean_leapfrogs_per_transition"], out["config"]["warmup_phase"]["mean_leapfrogs_per_transition"] or 0)
                # (leapfrog/s per core, measured on this image's hosts: D <= 512 Diag ≈ 1.3e6·128/D; the multi-thousand-D hierarchical
                # target ≈ 2.8e4·2048/D — its warm-up trees are 200–300 leaves —; the dense target / metric pair ≈ 435·(512/D)²)
                rate = 435.0 * (512.0 / D) ** 2 if cfg["metric"] == "dense" else (2.8e4 * 2048.0 / D if D > 512 else 1.3e6 * 128.0 / D)
                if cfg["metric"] == "dense":
                    cpu_chains = args.cpu_chains or min(N, 16 * cores)
                if args.cpu_transitions:
                    cpu_T = args.cpu_transitions
                else:
                    cpu_T = int(max(20 if cfg["metric"] == "dense" or D > 512 else 40, min(n_total, cpu_budget_s * rate * cores / (cpu_chains * lpt))))
                    if cpu_T * cpu_chains * lpt / (rate * cores) > 2.0 * cpu_budget_s:   # still too long at the minimum transitions: fewer chains
                        cpu_chains = max(cores, int(cpu_budget_s * rate * cores / (cpu_T * lpt)))
                ca = int(round(cpu_T * args.adapt_fraction))
                v, v_draw, cdt, cores = cpu_baseline(A, cfg, ca, cpu_T - ca, seed, cpu_chains, cores)
                out["cpu_baseline"] = {
                    "value": v, "unit": "leapfrog-steps/s", "cores": cores, "kind": "port",
                    "sample": f"{cpu_chains} chains x D={D}, the same sample loop ({ca} adapting transitions + {cpu_T - ca} draws, same kernel / adaptor), "
                              f"{cdt:.1f} s; C++ restatement of the reference (oracle/), OpenMP over chains, {cores} threads = the container's "
                              f"CPU quota ({os.cpu_count()} logical CPUs visible), not Julia",
                    "post_adaptation_value": v_draw,
                }
                if ctx.get("single_thread_leg", True):
                    # single thread: what the reference's broadcast path uses; a smaller sample of the same loop
                    c1 = max(8, cpu_chains // (8 * cores))
                    t1n = max(20, cpu_T // 4)
                    v1, _, cdt1, _ = cpu_baseline(A, cfg, int(round(t1n * args.adapt_fraction)), t1n - int(round(t1n * args.adapt_fraction)), seed, c1, 1)
                    out["cpu_baseline"]["single_thread"] = {"value": v1, "cores": 1, "sample": f"{c1} chains, {t1n} transitions of the same loop, {cdt1:.1f} s"}
            except Exception as ex:  # the baseline leg must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "error": repr(ex)}
    comm.close()
    eng.close()
    return out


LINE_BUDGET = 6000   # characters of the ONE line the driver parses (its capture keeps ~8 KB of output tail)


def _sig(x, n=6):
    """numbers to n significant digits (the line is a record, not an archive: bench_detail.json keeps every bit)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if not math.isfinite(x):
            return None
        if x == int(x) and abs(x) < 1e15:
            return int(x)
        return float(f"{x:.{n}g}")
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    return x


def _roof_compact(r, with_launches):
    """the judge's fields of `roofline`: bound / unit / peak / achieved / frac / traffic / kernel (+ the launches the in-run HIP
    events timed); the instruction mix, the peak's definition and the counters' provenance stay in bench_detail.json"""
    if not r:
        return None
    o = {k: r.get(k) for k in ("bound", "unit", "peak", "achieved", "frac", "traffic")}
    o["kernel"] = (r.get("kernel") or "")[:64]
    for k in ("valu_efficiency", "device_time_share_of_timed_region", "hbm_bytes_per_leapfrog_by_counters", "hbm_model_frac", "hbm_measured_frac"):
        if r.get(k) is not None:
            o[k] = r[k]
    if with_launches:
        for which in ("dominant", "other"):
            x = r.get(which)
            if x:
                o[which] = {k: x.get(k) for k in ("kernel", "launches", "avg_launch_ms", "leapfrogs_per_launch", "frac", "traffic", "hbm_model_frac") if x.get(k) is not None}
    elif r.get("dominant"):
        x = r["dominant"]
        o["launches"], o["avg_launch_ms"] = x.get("launches"), x.get("avg_launch_ms")
    return o


def _cpu_compact(b, full):
    if not b:
        return None
    if b.get("value") is None:
        return {"value": None, "error": str(b.get("error"))[:120]}
    o = {k: b.get(k) for k in ("value", "unit", "cores", "kind")}
    if full:
        o["sample"] = (b.get("sample") or "")[:140]
        if b.get("single_thread"):
            o["single_thread_value"] = b["single_thread"].get("value")
    else:
        o.pop("unit", None)
    return o


def compact_line(full, detail_path=None):
    """The ONE line the driver parses, from the full record: the contract's headline keys, `roofline` and `cpu_baseline` with the
    fields the contract names, `config` with the workload and — default invocation — `config.secondary.{cfg3,cfg5,cfg4}` each as
    {value, ms_per_step, steps, workload, dtype, roofline, cpu_baseline}.  Everything else (instruction mixes, definitions, every run,
    ESS prose) is in `detail_path`.  Round 4's line had grown to 31.9 KB and the driver's 8 KB capture lost its head: this one is
    built to stay under LINE_BUDGET and, should a future field push it over, sheds optional fields instead of growing."""
    c = full["config"]
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                "dtype", "data")}
    cc = {"workload": c["workload"][:300]}
    for k in ("chains_per_gpu", "D", "parallelism", "ranks_seen", "transitions_per_step", "n_adapts", "n_draws", "leapfrogs", "runs", "reported",
              "fits_quoted_config", "draw_launch_length_found_by_the_engine", "max_abs_mean", "max_abs_var_minus_1"):
        if c.get(k) is not None:
            cc[k] = c[k]
    for ph in ("warmup_phase", "post_adaptation"):
        if c.get(ph):
            cc[ph] = {k: c[ph].get(k) for k in ("value", "mean_leapfrogs_per_transition", "divergent")}
    if c.get("ess"):
        cc["ess_per_sec"] = c["ess"].get("ess_per_sec")
        cc["ess_per_draw_min_over_dims"] = c["ess"].get("ess_per_draw_min_over_dims")
    if c.get("draws_materialised_in_timed_region"):
        cc["draws_gib_written_in_timed_region"] = c["draws_materialised_in_timed_region"].get("gib")
    sec = c.get("secondary")
    if sec:
        cs = {}
        for name, o in sec.items():
            if not o or "value" not in o:
                cs[name] = {"error": str((o or {}).get("error"))[:160]}
                continue
            oc = o["config"]
            cs[name] = {"value": o["value"], "ms_per_step": o["ms_per_step"], "steps": o["steps"], "dtype": o["dtype"],
                        "workload": oc["workload"].split(", timed =")[0].split("; timed =")[0][:200],
                        "n_adapts": oc["n_adapts"], "n_draws": oc["n_draws"], "chains_per_gpu": oc["chains_per_gpu"],
                        "ess_per_sec": (oc.get("ess") or {}).get("ess_per_sec"),
                        "roofline": _roof_compact(o.get("roofline"), False), "cpu_baseline": _cpu_compact(o.get("cpu_baseline"), False)}
        cc["secondary"] = cs
    if detail_path:
        cc["detail"] = detail_path
    out["config"] = cc
    out["roofline"] = _roof_compact(full.get("roofline"), True)
    if "cpu_baseline" in full:
        out["cpu_baseline"] = _cpu_compact(full["cpu_baseline"], True)
    out = _sig(out)
    # the budget is a hard one: shed optional fields, most dispensable first, rather than print a line the driver cannot hold
    for path in (("config", "runs"), ("roofline", "other"), ("config", "max_abs_var_minus_1"), ("config", "max_abs_mean"),
                 ("config", "leapfrogs"), ("roofline", "dominant"), ("cpu_baseline", "sample")):
        if len(json.dumps(out)) <= LINE_BUDGET:
            break
        out.get(path[0], {}).pop(path[1], None)
    if len(json.dumps(out)) > LINE_BUDGET:
        for o in out["config"].get("secondary", {}).values():
            o.pop("workload", None)
    if len(json.dumps(out)) > LINE_BUDGET:
        # last resort — a record must ALWAYS come out: the contract's headline keys, the roofline's and the CPU baseline's numbers,
        # the workload and where the rest is
        out = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                   "dtype", "data")}
        out["config"] = {"workload": c["workload"][:200], "detail": detail_path, "line_shortened": "over the size budget: headline only"}
        r, b = full.get("roofline") or {}, full.get("cpu_baseline") or {}
        out["roofline"] = {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
        out["cpu_baseline"] = {k: b.get(k) for k in ("value", "unit", "cores", "kind")}
        out = _sig(out)
    return out


# the default run reports the other single-GPU BASELINE configs beside the headline (config.secondary): (steps, transitions per
# step, untimed warm-up transitions, timed runs, seconds of CPU baseline) — sized so that the whole default invocation stays
# well under two minutes: cfg3 1 000 + 1 000, cfg5 100 + 100 and cfg4 300 + 300 on one GPU's shard (cfg4's batches end in a tail of
# few running chains: 2 / 6 steps measure 42.5 / 44.0 TFLOP/s, profiles/r5_experiments.md)
SECONDARY = {"cfg3": (20, 100, 100, 1, 5.0), "cfg5": (2, 100, 10, 1, 5.0), "cfg4": (6, 100, 10, 1, 5.0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="K timed steps; a step = --transitions-per-step transitions of all chains")
    ap.add_argument("--warmup", type=int, default=2, help="W untimed steps of the same loop on a throw-away engine")
    ap.add_argument("--transitions-per-step", type=int, default=100)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None, help="default: cfg2 (the headline) with the other one-GPU configs as config.secondary")
    ap.add_argument("--no-secondary", action="store_true", help="headline only")
    ap.add_argument("--chains", type=int, default=0, help="chains per GPU (0 = the config's)")
    ap.add_argument("--dim", type=int, default=0, help="D (0 = the config's)")
    ap.add_argument("--adapt-fraction", type=float, default=0.5, help="share of the K steps that adapt (SURVEY 8d: 1000 of 2000)")
    ap.add_argument("--repeats", type=int, default=0, help="timed runs (0 = until >= 1 s of timed work, at least 3 when a run is < 4 s)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-chains", type=int, default=0, help="0 = 256 per usable host core, capped at --chains")
    ap.add_argument("--cpu-transitions", type=int, default=0, help="transitions of the CPU sample (0 = sized for ~12 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64", help="f64 = the reference default and the headline; f32 = what the "
                    "reference's CUDA smoke test uses (test/CUDA/cuda.jl:18), reported for information")
    ap.add_argument("--ess", type=int, default=-1, help="ESS of the TIMED run's draws (every chain, every dimension; device reduction through ahmc_ess): "
                    "0 = off, anything else = on (every config; outside the timed region)")
    ap.add_argument("--no-draws-out", action="store_true", help="do not materialise the draws in the timed region (A/B of its cost; the reported line always does)")
    ap.add_argument("--detail", default="bench_detail.json", help="file (relative to the repository) that receives the FULL record; the printed line "
                    "is its compact form (<= 6 000 characters).  '' = none")
    ap.add_argument("--launch-check", action="store_true", help="spawn/rendezvous check only (gloo, no GPU, no compute)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        respawn_under_torchrun(args.gpus)  # does not return
    if args.launch_check:
        sys.exit(launch_check(args))

    import torch
    import ahmc_amd as A

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or let --gpus spawn them)")
    dist = None
    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("AHMC_BENCH_FORCE_DIST"):  # (the env switch exercises the RCCL path on one GPU)
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    lib = A.load_hip_library()  # raises if the HIP engine is not built: no fallback
    ctx = {"A": A, "lib": lib, "torch": torch, "dist": dist, "args": args, "rank": rank, "local_rank": local_rank, "world": world,
           "stream": torch.cuda.Stream(device=local_rank)}
    headline = args.config or "cfg2"
    T = args.transitions_per_step
    out = run_config(ctx, headline, args.steps, T, args.warmup * T, args.repeats, 12.0, dim=args.dim, chains=args.chains)
    # The other BASELINE configs a single GPU holds, each with its own value / roofline / cpu_baseline, beside the headline — in
    # the default invocation only (what the driver runs).  At N > 1: the configs BASELINE.json quotes on exactly N GPUs.
    if args.config is None and not args.no_secondary and not args.dim and args.dtype == "f64":
        sec = {}
        ctx["single_thread_leg"] = False
        for name, (st, t2, wt, rp, cpu_s) in SECONDARY.items():
            if world > 1 and CONFIGS[name]["quoted_gpus"] != world:
                continue
            try:
                # (--chains with the default invocation: the contract test's reduced run — every config at that many chains)
                o = run_config(ctx, name, st, t2, wt, rp, cpu_s, chains=min(args.chains, CONFIGS[name]["N"]) if args.chains else 0)
            except Exception as ex:  # a secondary config must never take the headline down with it
                o = {"error": repr(ex)} if rank == 0 else None
                if dist is not None and not isinstance(ex, SetupFailed):
                    # one rank failing INSIDE the loop leaves the others waiting in its collectives: no way to agree any more (setup
                    # failures — the likely ones: memory — are agreed on by all ranks before the first collective, see run_config).
                    # Rank 0 still prints the headline it has (computed by all ranks before any secondary); every other rank that
                    # fails exits non-zero, which makes the launcher (torch.distributed.run) end the whole job instead of leaving
                    # the others in a collective for ever.
                    print(f"bench.py rank {rank}: {name} failed after setup: {ex!r}", file=sys.stderr, flush=True)
                    if rank != 0:
                        os._exit(3)
                    out["config"]["secondary"] = dict(sec, **{name: o})
                    out["config"]["aborted"] = f"{name} failed after setup on rank 0"
                    _print_line(out, args)
                    os._exit(3)
            if rank == 0:
                sec[name] = o
        if rank == 0:
            out["config"]["secondary"] = sec
    if rank == 0:
        _print_line(out, args)
    if dist is not None:
        dist.destroy_process_group()


def _print_line(out, args):
    """rank 0: the full record to the detail file, its compact form as the LAST line of stdout"""
    try:  # RCCL writes its version banner to the C stdout when the first communicator is made: push it out first, so
        ctypes.CDLL(None).fflush(None)  # that the JSON line is the LAST line of stdout whatever the buffering
    except Exception:
        pass
    detail = None
    if args.detail:
        try:  # the full record (instruction mixes, every run, definitions, ESS prose): a side file, named in the line
            detail = args.detail if os.path.isabs(args.detail) else os.path.join(ROOT, args.detail)
            os.makedirs(os.path.dirname(detail), exist_ok=True)
            with open(detail, "w") as f:
                json.dump(out, f, indent=1)
        except OSError as ex:
            detail = None
            print(f"bench.py: cannot write {args.detail}: {ex}", file=sys.stderr)
    line = json.dumps(compact_line(out, os.path.relpath(detail, ROOT) if detail else None))  # (never over LINE_BUDGET: compact_line's last resort)
    print(line, flush=True)             # flushed NOW: with a process group alive the interpreter's exit path (RCCL / c10d
    sys.stdout.flush()                  # teardown) was seen to drop a block-buffered stdout — the line must not depend on it


if __name__ == "__main__":
    main()
