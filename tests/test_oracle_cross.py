"""The C++ oracle against a second, independent restatement of the reference (oracle/ahmc_ref.py: the Julia
source transcribed struct for struct into plain Python).  Both run the same Philox streams; every discrete
statistic and every float must agree bit for bit, transition after transition — which pins the oracle's control
logic (recursion order of build_tree, which draws are consumed when, sampler combination, the three U-turn
criteria, divergence, the statistics) to a second reading of src/trajectory.jl, not only to its known-answer tests.
"""
import os
import sys

import numpy as np
import pytest

import ahmc_amd as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ahmc_ref as R  # noqa: E402

TARGETS = {"iso": (R.iso_gaussian, A.IsoGaussian), "funnel": (R.funnel, A.Funnel)}
TS = {"multinomial": (R.MultinomialTS, A.MultinomialTS), "slice": (R.SliceTS, A.SliceTS)}
TC = {"classic": (R.CLASSIC, A.ClassicNoUTurn), "generalised": (R.GENERALISED, A.GeneralisedNoUTurn), "strict": (R.STRICT, A.StrictGeneralisedNoUTurn)}

FLOAT_STATS = ("acceptance_rate", "log_density", "hamiltonian_energy", "hamiltonian_energy_error", "max_hamiltonian_energy_error")
INT_STATS = ("n_steps", "tree_depth", "numerical_error", "is_accept")


@pytest.mark.parametrize("target", ["iso", "funnel"])
@pytest.mark.parametrize("metric", ["unit", "diag"])
@pytest.mark.parametrize("ts", ["multinomial", "slice"])
@pytest.mark.parametrize("tc", ["classic", "generalised", "strict"])
def test_nuts_transitions_bit_for_bit(oracle, rng, target, metric, ts, tc):
    D, N, n_trans, seed = 5, 12, 4, 0xA5A5
    fn, builtin = TARGETS[target]
    minv = None if metric == "unit" else (0.5 + rng.random((D, N)))
    eps = 0.35 * (0.5 + rng.random(N))
    th0 = rng.normal(size=(D, N))
    m = A.UnitEuclideanMetric(D) if minv is None else A.DiagEuclideanMetric(np.asfortranarray(minv))
    lf = A.Leapfrog(eps)
    eng = A.Engine(A.Hamiltonian(m, builtin(D)), N, rng=seed, lib=oracle)
    eng.set_integrator(lf)
    eng.set_position(th0)
    kernel = A.HMCKernel(A.Trajectory(TS[ts][1], lf, TC[tc][1](max_depth=6, delta_max=50.0)))
    ref = []
    for c in range(N):
        h = R.Hamiltonian(None if minv is None else [float(x) for x in minv[:, c]], fn, D)
        nt = R.NUTS(TS[ts][0], TC[tc][0], float(eps[c]), max_depth=6, delta_max=50.0)
        ref.append(R.sample_chain(seed, c, h, nt, [float(x) for x in th0[:, c]], n_trans))
    depths = set()
    for it in range(n_trans):
        eng.transition(kernel)
        st, z = eng.stats(), eng.phasepoint()
        for c in range(N):
            (th_ref, r_ref), st_ref = ref[c][0][it], ref[c][1][it]
            assert [float(x) for x in z.theta[:, c]] == th_ref, (it, c)
            assert [float(x) for x in z.r[:, c]] == r_ref, (it, c)
            for k in FLOAT_STATS:
                assert float(st[k][c]) == st_ref[k], (k, it, c)
            for k in INT_STATS:
                assert int(st[k][c]) == int(st_ref[k]), (k, it, c)
            depths.add(st_ref["tree_depth"])
    assert len(depths) > 1  # (trees of several sizes were built)
    eng.close()


@pytest.mark.parametrize("metric", ["unit", "diag"])
def test_static_endpoint_transitions_bit_for_bit(oracle, rng, metric):
    D, N, n_trans, seed, L = 4, 10, 5, 77, 6
    minv = None if metric == "unit" else (0.5 + rng.random((D, N)))
    eps = 0.4 * (0.5 + rng.random(N))
    th0 = rng.normal(size=(D, N))
    m = A.UnitEuclideanMetric(D) if minv is None else A.DiagEuclideanMetric(np.asfortranarray(minv))
    lf = A.Leapfrog(eps)
    eng = A.Engine(A.Hamiltonian(m, A.Funnel(D)), N, rng=seed, lib=oracle)
    eng.set_integrator(lf)
    eng.set_position(th0)
    kernel = A.HMCKernel(A.Trajectory(A.EndPointTS, lf, A.FixedNSteps(L)))
    ref = []
    for c in range(N):
        h = R.Hamiltonian(None if minv is None else [float(x) for x in minv[:, c]], R.funnel, D)
        ref.append(R.sample_chain(seed, c, h, ("hmc", float(eps[c]), L), [float(x) for x in th0[:, c]], n_trans))
    accepts = set()
    for it in range(n_trans):
        eng.transition(kernel)
        st, z = eng.stats(), eng.phasepoint()
        for c in range(N):
            (th_ref, r_ref), st_ref = ref[c][0][it], ref[c][1][it]
            assert [float(x) for x in z.theta[:, c]] == th_ref, (it, c)
            assert [float(x) for x in z.r[:, c]] == r_ref, (it, c)
            for k in ("acceptance_rate", "log_density", "hamiltonian_energy", "hamiltonian_energy_error"):
                assert float(st[k][c]) == st_ref[k], (k, it, c)
            assert bool(st["is_accept"][c]) == st_ref["is_accept"] and int(st["n_steps"][c]) == L
            accepts.add(st_ref["is_accept"])
    eng.close()


def test_philox_restatement_against_random123_vectors():
    """the Python Philox used above against the Random123 known-answer vectors (kat_vectors: philox4x32 10)"""
    assert R.philox4x32_10(0, 0, 0, 0, 0, 0) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)
    assert R.philox4x32_10(0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF) == (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)
    assert R.philox4x32_10(0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344, 0xA4093822, 0x299F31D0) == (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)


@pytest.mark.parametrize("ts", ["multinomial", "slice"])
def test_divergent_and_max_depth_trees_bit_for_bit(oracle, rng, ts):
    """large steps on the funnel (divergences: Δ_max exceeded, non-finite points) and tiny steps with max_depth 3
    (trees that end by depth, not by a U-turn)"""
    D, N, seed = 4, 16, 99
    th0 = rng.normal(size=(D, N)) * 2
    seen_div, seen_full = False, False
    for eps_v, max_depth, delta_max in ((2.5, 6, 10.0), (1e-3, 3, 1000.0)):
        lf = A.Leapfrog(np.full(N, eps_v))
        eng = A.Engine(A.Hamiltonian(A.UnitEuclideanMetric(D), A.Funnel(D)), N, rng=seed, lib=oracle)
        eng.set_integrator(lf)
        eng.set_position(th0)
        kernel = A.HMCKernel(A.Trajectory(TS[ts][1], lf, A.GeneralisedNoUTurn(max_depth=max_depth, delta_max=delta_max)))
        for it in range(3):
            eng.transition(kernel)
        st, z = eng.stats(), eng.phasepoint()
        for c in range(N):
            h = R.Hamiltonian(None, R.funnel, D)
            nt = R.NUTS(TS[ts][0], R.GENERALISED, eps_v, max_depth=max_depth, delta_max=delta_max)
            draws, stats = R.sample_chain(seed, c, h, nt, [float(x) for x in th0[:, c]], 3)
            assert [float(x) for x in z.theta[:, c]] == draws[-1][0], c
            for k in FLOAT_STATS:
                a, b = float(st[k][c]), stats[-1][k]
                assert a == b or (np.isnan(a) and np.isnan(b)), (k, c, a, b)
            for k in INT_STATS:
                assert int(st[k][c]) == int(stats[-1][k]), (k, c)
            seen_div |= any(s["numerical_error"] for s in stats)
            seen_full |= any(s["tree_depth"] == max_depth and s["n_steps"] == 2 ** max_depth - 1 for s in stats)
        eng.close()
    assert seen_div and seen_full


def test_static_multinomial_transitions_bit_for_bit(oracle, rng):
    D, N, n_trans, seed, L = 4, 10, 6, 31, 7
    minv = 0.5 + rng.random((D, N))
    eps = 0.35 * (0.5 + rng.random(N))
    th0 = rng.normal(size=(D, N))
    lf = A.Leapfrog(eps)
    eng = A.Engine(A.Hamiltonian(A.DiagEuclideanMetric(np.asfortranarray(minv)), A.Funnel(D)), N, rng=seed, lib=oracle)
    eng.set_integrator(lf)
    eng.set_position(th0)
    kernel = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.FixedNSteps(L)))
    zs = []
    for c in range(N):
        h = R.Hamiltonian([float(x) for x in minv[:, c]], R.funnel, D)
        zs.append(R.phasepoint(h, [float(x) for x in th0[:, c]], [0.0] * D))
    for it in range(n_trans):
        eng.transition(kernel)
        st, z = eng.stats(), eng.phasepoint()
        for c in range(N):
            h = R.Hamiltonian([float(x) for x in minv[:, c]], R.funnel, D)
            r = R.Rng(seed, c, it)
            zs[c], sr = R.hmc_multinomial_transition(r, h, float(eps[c]), L, R.refresh(r, h, zs[c]))
            assert [float(x) for x in z.theta[:, c]] == zs[c].theta, (it, c)
            assert [float(x) for x in z.r[:, c]] == zs[c].r, (it, c)
            for k in ("acceptance_rate", "log_density", "hamiltonian_energy", "hamiltonian_energy_error"):
                assert float(st[k][c]) == sr[k], (k, it, c)
    eng.close()


@pytest.mark.parametrize("kind", ["nuts", "hmc"])
def test_sample_with_stan_adaptor_bit_for_bit(oracle, rng, kind):
    """the whole `sample` loop with StanHMCAdaptor(WelfordVar, NesterovDualAveraging): window schedule, dual averaging
    (incl. its resets and finalize!), Welford pushes / updates / resets, renew of the metric and of the step size —
    step sizes, mass matrices and draws after 60 adaptation + 10 sampling transitions, bit for bit"""
    D, N, seed, n_samples, n_adapts = 3, 6, 5, 70, 60
    windows = (8, 6, 5)
    th0 = rng.normal(size=(D, N))
    metric = A.DiagEuclideanMetric(np.ones((D, N), order="F"))
    lf = A.Leapfrog(np.full(N, 0.3))
    if kind == "nuts":
        kernel = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=5)))
        kernel_of = lambda e: R.NUTS(R.MultinomialTS, R.GENERALISED, e, max_depth=5)  # noqa: E731
    else:
        kernel = A.HMCKernel(A.Trajectory(A.EndPointTS, lf, A.FixedNSteps(5)))
        kernel_of = lambda e: ("hmc", e, 5)  # noqa: E731
    adaptor = A.StanHMCAdaptor(A.MassMatrixAdaptor(metric), A.StepSizeAdaptor(0.8, lf), init_buffer=windows[0], term_buffer=windows[1],
                               window_size=windows[2])
    thetas, stats = A.sample(seed, A.Hamiltonian(metric, A.IsoGaussian(D)), kernel, th0, n_samples, adaptor, n_adapts, lib=oracle)
    for c in range(N):
        draws, st_ref, eps_ref, minv_ref = R.sample_chain_adapted(seed, c, R.iso_gaussian, [1.0] * D, 0.3, kernel_of, [float(x) for x in th0[:, c]],
                                                                  n_samples, n_adapts, windows=windows)
        for i in range(n_samples):
            assert [float(x) for x in thetas[i][:, c]] == draws[i][0], (i, c)
            assert float(stats[i]["step_size"][c]) == st_ref[i]["step_size"], (i, c)
            assert float(stats[i]["acceptance_rate"][c]) == st_ref[i]["acceptance_rate"], (i, c)
        assert float(stats[-1]["nom_step_size"][c]) == eps_ref and eps_ref != 0.3
        assert minv_ref != [1.0] * D


@pytest.mark.parametrize("metric", ["unit", "diag"])
@pytest.mark.parametrize("target", ["iso", "funnel"])
def test_find_good_stepsize_bit_for_bit(oracle, rng, metric, target):
    D, N, seed = 6, 20, 123
    fn, builtin = TARGETS[target]
    minv = None if metric == "unit" else (0.05 + 5 * rng.random((D, N)))
    th0 = rng.normal(size=(D, N)) * 2
    m = A.UnitEuclideanMetric(D) if minv is None else A.DiagEuclideanMetric(np.asfortranarray(minv))
    eng = A.Engine(A.Hamiltonian(m, builtin(D)), N, rng=seed, lib=oracle)
    eng.set_integrator(A.Leapfrog(0.1))
    eng.set_position(th0)
    eps = eng.find_good_stepsize()
    for c in range(N):
        h = R.Hamiltonian(None if minv is None else [float(x) for x in minv[:, c]], fn, D)
        assert float(eps[c]) == R.find_good_stepsize(seed, c, 0, h, [float(x) for x in th0[:, c]]), c
    assert len(set(float(e) for e in eps)) > 1
    eng.close()


def test_jitter_partial_refreshment_and_tempering_bit_for_bit(oracle, rng):
    """transition(rng, h, κ, z) (src/sampler.jl:48-58) with a JitteredLeapfrog + PartialMomentumRefreshment (NUTS), and a
    TemperedLeapfrog (static EndPointTS and NUTS)"""
    D, N, seed, n_trans = 4, 8, 17, 4
    th0 = rng.normal(size=(D, N))
    minv = 0.5 + rng.random((D, N))
    m = A.DiagEuclideanMetric(np.asfortranarray(minv))
    # (1) jitter + partial refreshment, NUTS
    lf = A.JitteredLeapfrog(0.3, 0.4)
    eng = A.Engine(A.Hamiltonian(m, A.Funnel(D)), N, rng=seed, lib=oracle)
    eng.set_integrator(lf)
    eng.set_position(th0)
    kernel = A.HMCKernel(A.PartialMomentumRefreshment(0.7), A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=5)))
    zs = []
    for c in range(N):
        h = R.Hamiltonian([float(x) for x in minv[:, c]], R.funnel, D)
        zs.append(R.phasepoint(h, [float(x) for x in th0[:, c]], [0.0] * D))
    for it in range(n_trans):
        eng.run(kernel, 1)
        st, z = eng.stats(), eng.phasepoint()
        for c in range(N):
            h = R.Hamiltonian([float(x) for x in minv[:, c]], R.funnel, D)
            r = R.Rng(seed, c, it)
            e = R.jitter(r, 0.3, 0.4)
            zs[c], sr = R.nuts_transition(r, h, R.NUTS(R.MultinomialTS, R.GENERALISED, e, max_depth=5), R.refresh(r, h, zs[c], 0.7))
            assert float(st["step_size"][c]) == e and e != 0.3
            assert [float(x) for x in z.theta[:, c]] == zs[c].theta and [float(x) for x in z.r[:, c]] == zs[c].r, (it, c)
            assert int(st["n_steps"][c]) == sr["n_steps"] and float(st["acceptance_rate"][c]) == sr["acceptance_rate"]
    eng.close()
    # (2) tempering: static EndPointTS, then NUTS
    eps = 0.25 * (0.5 + rng.random(N))
    lf = A.TemperedLeapfrog(eps, 1.1)
    eng = A.Engine(A.Hamiltonian(m, A.IsoGaussian(D)), N, rng=seed, lib=oracle)
    eng.set_integrator(lf)
    eng.set_position(th0)
    k_hmc = A.HMCKernel(A.Trajectory(A.EndPointTS, lf, A.FixedNSteps(5)))
    k_nuts = A.HMCKernel(A.Trajectory(A.SliceTS, lf, A.StrictGeneralisedNoUTurn(max_depth=5)))
    zs = []
    for c in range(N):
        h = R.Hamiltonian([float(x) for x in minv[:, c]], R.iso_gaussian, D)
        zs.append(R.phasepoint(h, [float(x) for x in th0[:, c]], [0.0] * D))
    for it in range(n_trans):
        nuts = it % 2 == 1
        eng.transition(k_nuts if nuts else k_hmc)
        st, z = eng.stats(), eng.phasepoint()
        for c in range(N):
            h = R.Hamiltonian([float(x) for x in minv[:, c]], R.iso_gaussian, D)
            r = R.Rng(seed, c, it)
            z0 = R.refresh(r, h, zs[c])
            if nuts:
                zs[c], sr = R.nuts_transition(r, h, R.NUTS(R.SliceTS, R.STRICT, float(eps[c]), max_depth=5, temper_alpha=1.1), z0)
            else:
                zs[c], sr = R.hmc_transition(r, h, float(eps[c]), 5, z0, temper_alpha=1.1)
            assert [float(x) for x in z.theta[:, c]] == zs[c].theta and [float(x) for x in z.r[:, c]] == zs[c].r, (it, c)
            assert float(st["hamiltonian_energy"][c]) == sr["hamiltonian_energy"]
    eng.close()


@pytest.mark.parametrize("ts,tc", [("multinomial", "generalised"), ("slice", "strict"), ("multinomial", "classic")])
def test_dense_metric_nuts_bit_for_bit(oracle, rng, ts, tc):
    """DenseEuclideanMetric (src/metric.jl:89-120,311-320; src/hamiltonian.jl:60-68,179-184): Cholesky factor,
    rand_momentum = U \\ z, ∂H∂r = M⁻¹r, neg_energy — the side of the oracle the MFMA engine is checked against"""
    D, N, n_trans, seed = 5, 8, 3, 2024
    B = rng.normal(size=(D, D))
    Minv = B @ B.T / D + np.eye(D)
    Minv = (Minv + Minv.T) / 2
    eps = 0.3 * (0.5 + rng.random(N))
    th0 = rng.normal(size=(D, N))
    lf = A.Leapfrog(eps)
    eng = A.Engine(A.Hamiltonian(A.DenseEuclideanMetric(Minv), A.Funnel(D)), N, rng=seed, lib=oracle)
    eng.set_integrator(lf)
    eng.set_position(th0)
    kernel = A.HMCKernel(A.Trajectory(TS[ts][1], lf, TC[tc][1](max_depth=5)))
    rows = [[float(x) for x in Minv[i]] for i in range(D)]
    ref = []
    for c in range(N):
        h = R.Hamiltonian(rows, R.funnel, D)
        ref.append(R.sample_chain(seed, c, h, R.NUTS(TS[ts][0], TC[tc][0], float(eps[c]), max_depth=5), [float(x) for x in th0[:, c]], n_trans))
    for it in range(n_trans):
        eng.transition(kernel)
        st, z = eng.stats(), eng.phasepoint()
        for c in range(N):
            (th_ref, r_ref), st_ref = ref[c][0][it], ref[c][1][it]
            assert [float(x) for x in z.theta[:, c]] == th_ref and [float(x) for x in z.r[:, c]] == r_ref, (it, c)
            for k in FLOAT_STATS:
                assert float(st[k][c]) == st_ref[k], (k, it, c)
            for k in INT_STATS:
                assert int(st[k][c]) == int(st_ref[k]), (k, it, c)
    eng.close()


def test_baseline_cfg1_plumbing_case(oracle):
    """BASELINE.json configs[0]: D = 10 isotropic Gaussian, UnitEuclideanMetric, static
    HMC(Leapfrog(0.1), FixedNSteps(16)), EndPointTS, 1 024 chains, θ0 ~ U(0,1), seed 0x5EED0001 — the reference's own
    CPU-runnable case.  The oracle's run: the first chains bit for bit against the second restatement, the whole batch
    against the moments the reference's test asserts (test/sampler-vec.jl:36-43: mean ≈ 0 within RNDATOL per chain)."""
    D, N, L, seed, n = 10, 1024, 16, 0x5EED0001, 300
    th0 = np.random.default_rng(seed).random((D, N))
    lf = A.Leapfrog(0.1)
    eng = A.Engine(A.Hamiltonian(A.UnitEuclideanMetric(D), A.IsoGaussian(D)), N, rng=seed, lib=oracle)
    eng.set_integrator(lf)
    eng.set_position(th0)
    kernel = A.HMCKernel(A.Trajectory(A.EndPointTS, lf, A.FixedNSteps(L)))
    eng.run(kernel, 5)
    z = eng.phasepoint()
    for c in range(6):
        h = R.Hamiltonian(None, R.iso_gaussian, D)
        draws, _ = R.sample_chain(seed, c, h, ("hmc", 0.1, L), [float(x) for x in th0[:, c]], 5)
        assert [float(x) for x in z.theta[:, c]] == draws[-1][0], c
    eng.run(kernel, n)  # accumulators are reset at the first kept transition of the call
    acc = eng.accum()
    assert acc["total_n_steps"] == n * N * L and acc["n_divergent"] == 0
    mean = acc["sum_theta"] / n
    var = acc["sumsq_theta"] / n - mean ** 2
    assert abs(mean.mean()) < 0.02 and abs(var.mean() + (mean ** 2).mean() - 1) < 0.05  # pooled over 3e5 draws per dimension
    assert eng.stats()["acceptance_rate"].mean() > 0.9  # ϵ = 0.1 on a unit Gaussian
    eng.close()


@pytest.mark.parametrize("case", ["stan_nutpie", "naive_welford", "naive_nutpie_hmcda"])
def test_other_adaptors_bit_for_bit(oracle, rng, case):
    """NutpieVar behind the Stan adaptor, NaiveHMCAdaptor (both estimators), and HMCDA's FixedIntegrationTime (the
    number of leapfrogs follows the adapted NOMINAL step size, src/trajectory.jl:241-243)"""
    D, N, seed, n_samples, n_adapts = 3, 5, 8, 45, 40
    windows = (6, 5, 4)
    th0 = rng.normal(size=(D, N))
    metric = A.DiagEuclideanMetric(np.ones((D, N), order="F"))
    nutpie = "nutpie" in case
    pc = A.NutpieVar(metric) if nutpie else A.MassMatrixAdaptor(metric)
    if case == "naive_nutpie_hmcda":
        lf = A.Leapfrog(0.25)  # FixedIntegrationTime needs ONE step size (src/trajectory.jl:241-243), so a single chain is run
        N = 1
        th0 = th0[:, :1]
        metric = A.DiagEuclideanMetric(np.ones((D, 1), order="F"))
        pc = A.NutpieVar(metric)
        kernel = A.HMCKernel(A.Trajectory(A.EndPointTS, lf, A.FixedIntegrationTime(1.5)))
        kernel_of = lambda e: ("hmcda", e, 1.5)  # noqa: E731
        eps0 = 0.25
    else:
        lf = A.Leapfrog(np.full(N, 0.3))
        kernel = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=5)))
        kernel_of = lambda e: R.NUTS(R.MultinomialTS, R.GENERALISED, e, max_depth=5)  # noqa: E731
        eps0 = 0.3
    ssa = A.StepSizeAdaptor(0.8, lf)
    naive = case.startswith("naive")
    adaptor = A.NaiveHMCAdaptor(pc, ssa) if naive else A.StanHMCAdaptor(pc, ssa, init_buffer=windows[0], term_buffer=windows[1], window_size=windows[2])
    thetas, stats = A.sample(seed, A.Hamiltonian(metric, A.Funnel(D)), kernel, th0, n_samples, adaptor, n_adapts, lib=oracle)
    for c in range(N):
        draws, st_ref, eps_ref, minv_ref = R.sample_chain_adapted(seed, c, R.funnel, [1.0] * D, eps0, kernel_of, [float(x) for x in th0[:, c]], n_samples,
                                                                  n_adapts, windows=windows, estimator=R.NutpieVar if nutpie else R.WelfordVar, naive=naive)
        for i in range(n_samples):
            assert [float(x) for x in np.atleast_2d(thetas[i].T).T[:, c]] == draws[i][0], (i, c)
            assert int(np.atleast_1d(stats[i]["n_steps"])[c]) == st_ref[i]["n_steps"], (i, c)
        assert float(np.atleast_1d(stats[-1]["nom_step_size"])[c]) == eps_ref and minv_ref != [1.0] * D


def test_dense_metric_with_welford_cov_adaptation_bit_for_bit(oracle, rng):
    """one chain with a DenseEuclideanMetric and StanHMCAdaptor → WelfordCov (massmatrix.jl:283-340), the reference's
    single-chain case: pushes, window-end estimate, the renewed metric's Cholesky factor, draws"""
    D, seed, n_samples, n_adapts = 3, 21, 50, 44
    windows = (6, 5, 4)
    th0 = rng.normal(size=(D, 1))
    metric = A.DenseEuclideanMetric(np.eye(D))
    lf = A.Leapfrog(0.3)
    kernel = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=5)))
    adaptor = A.StanHMCAdaptor(A.MassMatrixAdaptor(metric), A.StepSizeAdaptor(0.8, lf), init_buffer=windows[0], term_buffer=windows[1], window_size=windows[2])
    thetas, stats = A.sample(seed, A.Hamiltonian(metric, A.Funnel(D)), kernel, th0, n_samples, adaptor, n_adapts, lib=oracle)
    eye = [[1.0 if i == j else 0.0 for j in range(D)] for i in range(D)]
    draws, st_ref, eps_ref, minv_ref = R.sample_chain_adapted(seed, 0, R.funnel, eye, 0.3, lambda e: R.NUTS(R.MultinomialTS, R.GENERALISED, e, max_depth=5),
                                                              [float(x) for x in th0[:, 0]], n_samples, n_adapts, windows=windows, estimator=R.WelfordCov)
    for i in range(n_samples):
        assert [float(x) for x in np.asarray(thetas[i]).reshape(D)] == draws[i][0], i
    assert float(np.atleast_1d(stats[-1]["nom_step_size"])[0]) == eps_ref
    assert minv_ref != eye and minv_ref[0][1] != 0.0


def test_randomised_configurations_bit_for_bit(oracle):
    """120 random configurations (D 1..7, max_depth 1..7, step sizes from tiny to divergent, every sampler /
    criterion / metric / target, Δ_max from tight to loose): oracle == second restatement, bit for bit"""
    master = np.random.default_rng(987654321)
    n_div = n_deep = 0
    for case in range(120):
        D = int(master.integers(1, 8))
        N = int(master.integers(1, 5))
        max_depth = int(master.integers(1, 8))
        ts = ["multinomial", "slice"][int(master.integers(0, 2))]
        tc = ["classic", "generalised", "strict"][int(master.integers(0, 3))]
        target = ["iso", "funnel"][int(master.integers(0, 2))] if D > 1 else "iso"
        metric = ["unit", "diag", "dense"][int(master.integers(0, 3))]
        delta_max = float([5.0, 50.0, 1000.0][int(master.integers(0, 3))])
        eps = float(10 ** master.uniform(-2.5, 0.6)) * (0.5 + master.random(N))
        seed = int(master.integers(0, 2 ** 62))
        th0 = master.normal(size=(D, N)) * float(master.uniform(0.2, 3.0))
        fn, builtin = TARGETS[target]
        if metric == "unit":
            m, minvs = A.UnitEuclideanMetric(D), [None] * N
        elif metric == "diag":
            mv = 0.1 + 3 * master.random((D, N))
            m, minvs = A.DiagEuclideanMetric(np.asfortranarray(mv)), [[float(x) for x in mv[:, c]] for c in range(N)]
        else:
            B = master.normal(size=(D, D))
            M = B @ B.T / D + 0.5 * np.eye(D)
            M = (M + M.T) / 2
            m, minvs = A.DenseEuclideanMetric(M), [[[float(x) for x in M[i]] for i in range(D)]] * N
        lf = A.Leapfrog(eps)
        eng = A.Engine(A.Hamiltonian(m, builtin(D)), N, rng=seed, lib=oracle)
        eng.set_integrator(lf)
        eng.set_position(th0)
        kernel = A.HMCKernel(A.Trajectory(TS[ts][1], lf, TC[tc][1](max_depth=max_depth, delta_max=delta_max)))
        for _ in range(2):
            eng.transition(kernel)
        st, z = eng.stats(), eng.phasepoint()
        for c in range(N):
            h = R.Hamiltonian(minvs[c], fn, D)
            nt = R.NUTS(TS[ts][0], TC[tc][0], float(eps[c]), max_depth=max_depth, delta_max=delta_max)
            draws, stats = R.sample_chain(seed, c, h, nt, [float(x) for x in th0[:, c]], 2)
            tag = (case, c, D, max_depth, ts, tc, target, metric)
            assert [float(x) for x in z.theta[:, c]] == draws[-1][0], tag
            assert [float(x) for x in z.r[:, c]] == draws[-1][1], tag
            for k in FLOAT_STATS:
                a, b = float(st[k][c]), stats[-1][k]
                assert a == b or (np.isnan(a) and np.isnan(b)), (k,) + tag
            for k in INT_STATS:
                assert int(st[k][c]) == int(stats[-1][k]), (k,) + tag
            n_div += stats[-1]["numerical_error"]
            n_deep += stats[-1]["tree_depth"] == max_depth
        eng.close()
    assert n_div > 5 and n_deep > 5


def test_resumable_one_leapfrog_per_step_machine_equals_the_recursion():
    """experiments/r4/async_lockstep_model.py: the NUTS transition as a resumable state machine whose step() is exactly one
    leapfrog (the iterative formulation of the HIP kernel with every loop counter a member — what a lane group of an
    asynchronous kernel would carry) reproduces the recursive transcription bit for bit, divergent funnel trees included;
    and the schedule model runs (it is what DESIGN §7 item 3 quotes)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("async_lockstep_model", os.path.join(ROOT, "experiments", "r4", "async_lockstep_model.py"))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)
    assert M.check_machine(n_chains=4, n_transitions=10, D=8, eps=0.35, target="funnel") > 300
    assert M.check_machine(n_chains=2, n_transitions=6, D=6, eps=1.2, target="funnel") > 10      # short, mostly divergent trees
    assert M.check_machine(n_chains=3, n_transitions=8, D=16, eps=0.6, target="iso") > 100
    h = R.Hamiltonian([1.0] * 8, R.funnel, 8)

    def factory():
        return [M.GroupMachine(7, g, h, 0.3, [0.1 * (g + 1)] * 8, 5) for g in range(4)]
    r = M.simulate_wave(factory, 4, 0)
    assert r["transitions"] == 5 and r["chain_leapfrogs"] > 0 and r["today"] > 0 and r["async_"] > 0
