"""ABI v3: what surrounds the trajectory path — checkpoint / resume of the adaptor (HMCState,
src/abstractmcmc.jl:11-27), the variance estimator pooled over chains and ranks (§8f row 4), the final gather (§8e),
EBFMI (src/diagnosis.jl:1-3) and ESS on the device (§8f row 3).

CPU part: the oracle (same ABI) against numpy and against itself.  GPU part (`-m gpu`): the HIP engine against the
oracle / numpy through the same calls, and a one-rank RCCL communicator made by ahmc_comm_init."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import ahmc_amd as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make(lib, D, N, adaptor_kind, rng, shared_metric=False, seed=9, target=None, eps=0.2):
    shared_metric = shared_metric or "pooled" in adaptor_kind   # the pooled estimator adapts ONE (D,) M⁻¹
    minv = (0.5 + rng.random(D)) if shared_metric else np.asfortranarray(0.5 + rng.random((D, N)))
    metric = A.DiagEuclideanMetric(minv)
    h = A.Hamiltonian(metric, target or A.DiagGaussian(np.linspace(-1, 1, D), np.linspace(0.5, 2.0, D)))
    e = A.Engine(h, N, rng=seed, lib=lib)
    lf = A.Leapfrog(np.full(N, eps) * (0.7 + 0.6 * rng.random(N)))
    e.set_integrator(lf)
    e.set_position(rng.normal(size=(D, N)))
    ssa = A.StepSizeAdaptor(0.8, lf)
    ad = {"stan": A.StanHMCAdaptor(A.MassMatrixAdaptor(metric), ssa),
          "stan_nutpie": A.StanHMCAdaptor(A.NutpieVar(metric), ssa),
          "stan_pooled": A.StanHMCAdaptor(A.PooledVar(metric), ssa),
          "naive_pooled": A.NaiveHMCAdaptor(A.PooledVar(metric), ssa),
          "naive": A.NaiveHMCAdaptor(A.MassMatrixAdaptor(metric), ssa),
          "stepsize": ssa}[adaptor_kind]
    e.adaptor_init(ad)
    k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=6)))
    return e, k, h


def same_state(a, b):
    for key in ("theta", "r", "lp", "grad", "metric", "stepsize", "da", "welford"):
        if a[key] is None:
            assert b[key] is None, key
        else:
            np.testing.assert_array_equal(a[key], b[key], err_msg=key)
    assert a["adaptor"] == b["adaptor"]


def resume_case(lib, kind, rng_seed, bulk):
    D, N, n_adapts, n_total, cut = 5, 24, 160, 175, 93   # windows 76..110: the cut falls inside one, Welford state non-trivial
    runs = []
    for interrupted in (False, True):
        rng = np.random.default_rng(rng_seed)
        e, k, h = make(lib, D, N, kind, rng)
        if not interrupted:
            if bulk:
                e.run(k, n_total, n_adapts)
            else:
                for i in range(1, n_total + 1):
                    e.run(k, i, n_adapts, i_first=i)
        else:
            if bulk:
                e.run(k, cut, n_adapts)
            else:
                for i in range(1, cut + 1):
                    e.run(k, i, n_adapts, i_first=i)
            st = e.get_state()
            assert st["adaptor"]["iteration"] == cut and st["adaptor"]["adapting"] == 1
            e.close()
            e = A.Engine(h, N, rng=9, lib=lib)           # a fresh context: nothing but the checkpoint carries over
            e.set_integrator(A.Leapfrog(0.1))
            e.set_state(st)
            same_state(st, e.get_state())                 # the checkpoint round-trips
            e.run(k, n_total, n_adapts, i_first=cut + 1)
        runs.append((e.get_state(), e.stats()))
        e.close()
    same_state(runs[0][0], runs[1][0])
    for f in ("n_steps", "acceptance_rate", "hamiltonian_energy", "tree_depth"):
        np.testing.assert_array_equal(runs[0][1][f], runs[1][1][f], err_msg=f)


@pytest.mark.parametrize("kind", ["stan", "stan_nutpie", "naive", "stepsize", "stan_pooled"])
def test_adaptor_checkpoint_resumes_bit_for_bit(oracle, kind):
    """a run cut mid-warm-up (inside a Stan window) and resumed from ahmc_get/set_adaptor_state in a NEW context equals
    the uninterrupted run bit for bit: DAState, Welford (n, μ, M), window counter, RNG counter all round-trip"""
    resume_case(oracle, kind, 3, bulk=True)


def test_pooled_estimate_is_the_variance_of_all_pooled_draws(oracle, rng):
    """AHMC_VAR_POOLED: at a window end M⁻¹ (D,) = get_estimation (massmatrix.jl:152-157) of ALL draws of the window,
    every chain's, pooled — the per-chain Welford states merged by Chan's formula must equal the direct two-pass sums"""
    D, N, n_adapts = 4, 13, 150   # window 76..100, one split at 100 (stan_adaptor.jl:13-50)
    e, k, h = make(oracle, D, N, "stan_pooled", rng, shared_metric=True)
    draws = []
    for i in range(1, n_adapts + 1):
        th = rng.normal(size=(D, N)) * np.arange(1, D + 1)[:, None] + 3.0
        e.adapt(i, n_adapts, theta=th, alpha=np.full(N, 0.8))
        if 76 <= i <= 100:
            draws.append(th)
        m = e.get_metric()
        assert m.shape == (D,), "the metric stays one shared (D,) vector"
    x = np.concatenate(draws, axis=1)   # (D, 25·N)
    n = x.shape[1]
    M = ((x - x.mean(axis=1, keepdims=True)) ** 2).sum(axis=1)
    want = n / ((n + 5) * (n - 1)) * M + 1e-3 * (5 / (n + 5))
    np.testing.assert_allclose(e.get_metric(), want, rtol=1e-12)
    e.close()
    # a per-chain (D,N) metric is refused for the pooled estimator
    bad = A.Engine(A.Hamiltonian(A.DiagEuclideanMetric(np.ones((D, N), order="F")), A.IsoGaussian(D)), N, lib=oracle)
    bad.set_integrator(A.Leapfrog(0.1))
    with pytest.raises(A.ArgumentError):
        bad.adaptor_init(A.StanHMCAdaptor(A.PooledVar(A.DiagEuclideanMetric(np.ones((D, N)))), A.StepSizeAdaptor(0.8, A.Leapfrog(0.1))))
    bad.close()


def diag_checks(lib, rng, tol):
    D, N, K = 6, 40, 120
    e, k, h = make(lib, D, N, "stepsize", rng)
    e.run(k, 30, 30)
    draws = np.empty((K, N, D))                      # (D, N, K) column-major = what ahmc_sample writes
    e.run(k, K, 0, samples_out=draws)
    e.sync()
    eb = e.ebfmi()
    g = e.gather_moments()
    acc = e.accum()
    # the same run once more, one ahmc_sample call per iteration, to see every transition's energy
    rng2 = np.random.default_rng(20260925)
    e2, k2, _ = make(lib, D, N, "stepsize", rng2)
    e2.run(k2, 30, 30)
    E = []
    for _ in range(K):
        e2.run(k2, 1, 0)
        E.append(e2.stats()["hamiltonian_energy"].copy())
    np.testing.assert_allclose(eb, A.diagnostics.EBFMI(np.array(E)), rtol=tol)
    n = K * N
    mean = acc["sum_theta"].sum(axis=1) / n
    np.testing.assert_allclose(g["mean"], mean, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(g["var"], acc["sumsq_theta"].sum(axis=1) / n - mean ** 2, rtol=1e-9)
    assert g["n_draws"] == n and g["total_n_steps"] == acc["total_n_steps"] and g["n_divergent"] == acc["n_divergent"]
    np.testing.assert_allclose(mean, draws.mean(axis=(0, 1)), rtol=1e-9, atol=1e-12)
    e.close(); e2.close()
    return draws


def test_ebfmi_moments_and_ess_on_the_oracle(oracle, rng):
    draws = diag_checks(oracle, rng, 1e-10)
    K, N, D = draws.shape
    e = A.Engine(A.Hamiltonian(A.UnitEuclideanMetric(D), A.IsoGaussian(D)), N, lib=oracle)
    got = e.ess(draws, K)                            # (D, N)
    want = A.diagnostics.ess(draws, axis=0).T        # FFT autocovariances, the same truncation rule
    np.testing.assert_allclose(got, want, rtol=1e-8)
    # an AR(1) series with known integrated autocorrelation time: ESS/n ≈ (1 − φ)/(1 + φ)
    phi, K2 = 0.6, 4000
    x = np.zeros((K2, N, D))
    z = np.random.default_rng(1).normal(size=(K2, N, D))
    for t in range(1, K2):
        x[t] = phi * x[t - 1] + z[t]
    ratio = e.ess(x, K2).mean() / K2
    assert abs(ratio - (1 - phi) / (1 + phi)) < 0.03, ratio
    # a series that never moved (every transition rejected) has no autocorrelation to estimate: K — whatever the rounding of its own mean
    # does (Σx / K of K identical values can be an ulp off x; round 6: the estimators decided on γ₀ > 0 and answered ≈ 1, K or K² there)
    K3 = 23
    c = np.empty((K3, N, D))
    vals = np.random.default_rng(2).normal(size=(N, D)) * np.array([1e-3, 1.0, 0.1, 1e6, 1 / 3, 7.7])[np.arange(D) % 6]
    c[:] = vals
    c[:, 0, 0] = np.arange(K3)                        # one series that does move
    got, want = e.ess(c, K3), A.diagnostics.ess(c, axis=0).T
    still = np.ones((D, N), dtype=bool)
    still[0, 0] = False
    assert (got[still] == K3).all() and (want[still] == K3).all()
    np.testing.assert_allclose(got[0, 0], want[0, 0], rtol=1e-8)
    assert got[0, 0] < K3 / 4                          # a ramp is as autocorrelated as a series gets
    e.close()


GLOO_WORKER = r'''
import ctypes as C, os, sys
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
import ahmc_amd as A
from ahmc_amd.shard import chain_shard
sys.path.insert(0, os.path.join(%(root)r, "oracle")); import build_oracle
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
lib = A.CLib(build_oracle.build())
CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64, C.c_void_p)
def allgather(mine, out, count, user):            # the checker's cross-rank hook: an all-gather over gloo
    m = torch.from_numpy(np.ctypeslib.as_array(mine, shape=(count,)).copy())
    parts = [torch.empty_like(m) for _ in range(world)]
    dist.all_gather(parts, m)
    np.ctypeslib.as_array(out, shape=(count * world,))[:] = torch.cat(parts).numpy()
    return 0
cb = CB(allgather)
lib.dll.ahmco_set_allgather.argtypes = [C.c_void_p, CB, C.c_void_p, C.c_int32, C.c_int32]
D, N, n_adapts, n_total = 5, 22, 150, 160
off, cnt = chain_shard(N, rank, world)
full = np.random.default_rng(5)
minv, th0, epsv = 0.5 + full.random(D), full.normal(size=(D, N)), 0.2 * (0.7 + 0.6 * full.random(N))
metric = A.DiagEuclideanMetric(minv)
h = A.Hamiltonian(metric, A.DiagGaussian(np.linspace(-1, 1, D), np.linspace(0.5, 2.0, D)))
lf = A.Leapfrog(epsv[off:off + cnt])
e = A.Engine(h, cnt, rng=A.PhiloxRNG(17, chain_offset=off), lib=lib)
e.set_integrator(lf); e.set_position(th0[:, off:off + cnt])
lib.check(lib.dll.ahmco_set_allgather(e._ctx, cb, None, world, rank), e._ctx)
e.adaptor_init(A.StanHMCAdaptor(A.PooledVar(metric), A.StepSizeAdaptor(0.8, lf)))
k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=6)))
e.run(k, n_total, n_adapts)
g = e.gather_moments()
mine = torch.zeros((N, D), dtype=torch.float64); mine[off:off + cnt] = torch.from_numpy(np.ascontiguousarray(e.theta().T))
dist.all_reduce(mine)
if rank == 0:
    np.savez(%(out)r, theta=mine.numpy().T, metric=e.get_metric(), mean=g["mean"], var=g["var"], n=g["n_draws"], tot=g["total_n_steps"])
dist.destroy_process_group()
'''


def test_two_rank_pooled_adaptation_and_gather_match_single_process(tmp_path, oracle):
    """2 ranks over gloo: the pooled (D,) M⁻¹ needs every rank's chains at each window end (one all-gather per window)
    and ahmc_gather_moments pools the draws of all ranks — both must reproduce the single-process run of all chains"""
    out = str(tmp_path / "g.npz")
    script = tmp_path / "worker.py"
    script.write_text(GLOO_WORKER % {"root": ROOT, "out": out})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    got = np.load(out)
    D, N, n_adapts, n_total = 5, 22, 150, 160
    full = np.random.default_rng(5)
    minv, th0, epsv = 0.5 + full.random(D), full.normal(size=(D, N)), 0.2 * (0.7 + 0.6 * full.random(N))
    metric = A.DiagEuclideanMetric(minv)
    h = A.Hamiltonian(metric, A.DiagGaussian(np.linspace(-1, 1, D), np.linspace(0.5, 2.0, D)))
    lf = A.Leapfrog(epsv)
    e = A.Engine(h, N, rng=A.PhiloxRNG(17), lib=oracle)
    e.set_integrator(lf)
    e.set_position(th0)
    e.adaptor_init(A.StanHMCAdaptor(A.PooledVar(metric), A.StepSizeAdaptor(0.8, lf)))
    k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=6)))
    e.run(k, n_total, n_adapts)
    g = e.gather_moments()
    assert not np.allclose(e.get_metric(), minv), "the window end must have updated the pooled metric"
    # the pooled metric differs by rounding between 1 and 2 partitions (1e-16): compare the metric tightly and the
    # end state loosely (60 transitions of NUTS amplify a last-bit difference of M⁻¹ only mildly at this size)
    np.testing.assert_allclose(got["metric"], e.get_metric(), rtol=1e-12)
    assert got["n"] == g["n_draws"] == n_total * N
    close = np.isclose(got["theta"], e.theta(), rtol=1e-6, atol=1e-8).all(axis=0)
    assert close.mean() > 0.9, close.mean()
    np.testing.assert_allclose(got["mean"], g["mean"], atol=5e-2)
    e.close()


# ------------------------------------------------------------------------------------------------------------------
# GPU
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("kind,bulk", [("stan", True), ("stan_nutpie", True), ("stan", False), ("naive", False), ("stepsize", True), ("stan_pooled", True)])
def test_hip_adaptor_checkpoint_resumes_bit_for_bit(hip, kind, bulk):
    """the same on the HIP engine — bulk = the fused warm-up (adapt! inside k_nuts, batches of transitions), resumed
    with ahmc_sample_from; HIP against HIP, bit for bit"""
    resume_case(hip, kind, 4, bulk)


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["per_iteration", "fused"])
def test_hip_pooled_adaptation_matches_oracle(hip, oracle, path):
    """AHMC_VAR_POOLED on the HIP engine (k_pool_var / k_pool_finish; the fused warm-up ends its batches at the window
    splits) against the oracle: identical injected (θ, α) per iteration; then a real fused warm-up, compared on M⁻¹"""
    D, N, n_adapts = 24, 300, 150   # window 76..100, one split at 100
    if path == "per_iteration":
        engines = []
        for lib in (hip, oracle):
            e, k, h = make(lib, D, N, "stan_pooled", np.random.default_rng(8), shared_metric=True)
            engines.append(e)
        feed = np.random.default_rng(2)
        for i in range(1, n_adapts + 1):
            th = feed.normal(size=(D, N)) * np.linspace(0.5, 3, D)[:, None]
            al = feed.random(N)
            for e in engines:
                e.adapt(i, n_adapts, theta=th, alpha=al)
            if i in (75, 100, 101, 150):
                np.testing.assert_allclose(engines[0].get_metric(), engines[1].get_metric(), rtol=1e-11)
                np.testing.assert_allclose(engines[0].get_stepsize(), engines[1].get_stepsize(), rtol=1e-9)
        assert engines[0].get_metric().shape == (D,)
        for e in engines:
            e.close()
        return
    res = []
    for lib in (hip, oracle):
        e, k, h = make(lib, D, N, "stan_pooled", np.random.default_rng(8), shared_metric=True)
        m0 = e.get_metric().copy()
        e.run(k, 101, n_adapts)   # through the window end at 100 (one pooled update) and one transition beyond
        res.append((e.get_metric(), e.get_state()["adaptor"], m0))
        e.close()
    assert res[0][0].shape == (D,) and not np.allclose(res[0][0], res[0][2])
    assert res[0][1] == res[1][1]
    # 90 NUTS transitions with dual averaging in the loop: chains decorrelate from the oracle's at the 1e-6 level, the
    # pooled variance of 25 x 300 draws per dimension is a statistic of them — compare at the level that allows
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=0.15)
    true_var = np.linspace(0.5, 2.0, D) ** 2
    assert np.all(np.abs(res[0][0] / true_var - 1) < 0.25), res[0][0] / true_var


@pytest.mark.gpu
def test_hip_pooled_adaptation_through_a_one_rank_communicator(hip):
    """the rank-merge of the pooled estimator (ncclAllGather of the per-GPU partitions, k_pool_finish over R blocks) with a
    real RCCL communicator of ONE rank: the same chains as the run without a communicator, bit for bit"""
    D, N, n_adapts = 24, 300, 150
    out = []
    for with_comm in (False, True):
        e, k, h = make(hip, D, N, "stan_pooled", np.random.default_rng(8), shared_metric=True)
        if with_comm:
            e.comm_init(e.comm_unique_id(), 1, 0)
            assert e.comm_info()["ranks_seen"] == 1
        e.run(k, 110, n_adapts)
        out.append((e.get_metric().copy(), e.theta().copy(), e.get_stepsize().copy()))
        e.close()
    for a, b in zip(*out):
        np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
def test_hip_ebfmi_moments_ess_and_one_rank_rccl(hip, oracle, rng):
    """device-side reductions of the HIP engine (k_moments_reduce, the energy running sums of the transition kernels,
    k_ess) against numpy / the oracle's implementation, and the gather through a real RCCL communicator of one rank"""
    import torch

    D, N, K = 6, 40, 120
    e, k, h = make(hip, D, N, "stepsize", rng)
    e.run(k, 30, 30)
    draws_d = torch.empty((K, N, D), dtype=torch.float64, device="cuda")
    e.run(k, K, 0, samples_out=draws_d.data_ptr())
    e.sync()
    draws = draws_d.cpu().numpy()
    acc = e.accum()
    n = K * N
    mean = acc["sum_theta"].sum(axis=1) / n
    g0 = e.gather_moments()                           # no communicator: a world of one
    e.comm_init(e.comm_unique_id(), 1, 0)             # ncclCommInitRank(…, 1, id, 0): the all-reduce really runs
    assert e.comm_info() == {"ranks_seen": 1, "chains_total": N, "chains_min": N, "chains_max": N}   # comm_probe's two all-reduces ran
    g1 = e.gather_moments()
    for g in (g0, g1):
        np.testing.assert_allclose(g["mean"], mean, rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(g["var"], acc["sumsq_theta"].sum(axis=1) / n - mean ** 2, rtol=1e-9)
        assert g["n_draws"] == n and g["total_n_steps"] == acc["total_n_steps"]
    np.testing.assert_allclose(mean, draws.mean(axis=(0, 1)), rtol=1e-9, atol=1e-12)
    all_d = torch.empty((N, D), dtype=torch.float64, device="cuda")
    e.gather_state(all_d.data_ptr())                  # ncclAllGather over one rank
    np.testing.assert_array_equal(all_d.cpu().numpy().T, e.theta())
    # ESS: device kernel == the oracle's implementation of the same estimator == the FFT formulation
    got = e.ess(draws_d.data_ptr(), K)
    np.testing.assert_allclose(got, A.diagnostics.ess(draws, axis=0).T, rtol=1e-8)
    # EBFMI: the running sums of the fused kernel against the energies of the same chains replayed on the oracle
    eb = e.ebfmi()
    rng2 = np.random.default_rng(20260925)
    e2, k2, _ = make(oracle, D, N, "stepsize", rng2)
    e2.run(k2, 30, 30)
    E = []
    for _ in range(K):
        e2.run(k2, 1, 0)
        E.append(e2.stats()["hamiltonian_energy"].copy())
    want = A.diagnostics.EBFMI(np.array(E))
    ok = np.isclose(eb, want, rtol=1e-6)              # (a chain whose trajectory parted from the oracle's has other energies)
    assert ok.mean() >= 0.9, ok.mean()
    assert np.all(np.isfinite(eb)) and abs(np.median(eb) / np.median(want) - 1) < 0.05
    e.close(); e2.close()
