"""GPU side of the ask / tell protocol (ahmc_ext_*): the HIP engine driven with a user log-density must
reproduce the oracle running the same density built in (Float64: 1e-9 on energies / positions, discrete
statistics identical on every chain except at the oracle's own near-ties — the bars of test_gpu_parity.py), plus the step-synchronous engine's
variants (static MultinomialTS, partial refreshment, TemperedLeapfrog, Classic / Strict U-turn) and the split-step
pair ahmc_lf_pre / ahmc_lf_post.

History: in round 1 this file carried a module-wide xfail(strict=False) and every `engines()`-built test failed on
the MI355X behind it.  Cause: `_capi.as_ptr(<temporary array>)` handed ahmc_set_phasepoint the address of an array
that was freed before the call ran (advancedhmc.jl_amd/_capi.py, `_OwningPtr`) — the engine started every chain from
a garbage gradient once D*N*8 B outgrew numpy's small-block cache.  The marker is gone: a failure here is a failure.
"""
import ctypes as C

import numpy as np
import pytest

import ahmc_amd as A
import parity_util as PU
from test_external_target import TARGETS, iso_fn, make_metric

pytestmark = [pytest.mark.gpu,
              # every loop of the new host code is bounded (NUTS batches by their step bound, static HMC by L, the step-size
              # search by its iteration count); should one hang all the same, the run is cut short instead of stalling
              pytest.mark.timeout(240, method="thread")]

RT = 1e-9


def engines(hip, oracle, target, metric, N, lf, seed=7):
    fn, builtin = TARGETS[target]
    D = metric.D
    e_ext = A.Engine(A.Hamiltonian(metric, A.ExternalTarget(D, lambda th: fn(np.asarray(th, dtype=np.float64)))), N, rng=seed, lib=hip)
    e_ref = A.Engine(A.Hamiltonian(metric, builtin(D)), N, rng=seed, lib=oracle)
    for e in (e_ext, e_ref):
        e.set_integrator(lf)
    return e_ext, e_ref


def assert_close_state(e_ext, e_ref, what="external target"):
    """`e_ref` is the ORACLE engine: identical discrete decisions on every chain that the oracle did not take within 1e-9 of a tie
    (tests/parity_util.py; its margin record is read and reset here — one call per span of transitions)"""
    sa, sb = e_ext.stats(), e_ref.stats()
    same = (sa["n_steps"] == sb["n_steps"]) & (sa["is_accept"] == sb["is_accept"]) & (sa["tree_depth"] == sb["tree_depth"])
    same = PU.check_flips(same, PU.decision_margin(e_ref), np.float64, what)
    za, zb = e_ext.phasepoint(), e_ref.phasepoint()
    np.testing.assert_allclose(za.theta[:, same], zb.theta[:, same], rtol=RT, atol=RT)
    np.testing.assert_allclose(za.r[:, same], zb.r[:, same], rtol=RT, atol=RT)
    np.testing.assert_allclose(za.lp.gradient[:, same], zb.lp.gradient[:, same], rtol=RT, atol=RT)
    for k in ("acceptance_rate", "log_density", "hamiltonian_energy", "hamiltonian_energy_error", "max_hamiltonian_energy_error", "step_size"):
        np.testing.assert_allclose(sa[k][same], sb[k][same], rtol=1e-8, atol=1e-8, err_msg=k)
    return same


@pytest.mark.parametrize("TC", [A.ClassicNoUTurn, A.StrictGeneralisedNoUTurn])
@pytest.mark.parametrize("TS", [A.MultinomialTS, A.SliceTS])
@pytest.mark.parametrize("metric,target", [("dense", "dense"), ("dense", "funnel"), ("diag", "dense"), ("unit", "dense")])
def test_dense_engine_classic_and_strict_uturn(hip, oracle, rng, metric, target, TS, TC):
    """ClassicNoUTurn (src/trajectory.jl:551-557) and StrictGeneralisedNoUTurn (:579-617) in the step-synchronous tree
    kernel (k_d_tree_crit) against the oracle's recursion"""
    D, N = 16, 160
    B = rng.normal(size=(D, D))
    tgt = A.DenseGaussian(B @ B.T / D + np.eye(D)) if target == "dense" else A.Funnel(D)
    if metric == "dense":
        C2 = rng.normal(size=(D, D))
        m = A.DenseEuclideanMetric(C2 @ C2.T / D + np.eye(D))
    elif metric == "diag":
        m = A.DiagEuclideanMetric(np.asfortranarray(0.5 + rng.random((D, N))))
    else:
        m = A.UnitEuclideanMetric(D)
    lf = A.Leapfrog(np.full(N, 0.2) * (0.5 + rng.random(N)))
    h = A.Hamiltonian(m, tgt)
    e_g, e_o = A.Engine(h, N, rng=6, lib=hip), A.Engine(h, N, rng=6, lib=oracle)
    th0 = rng.normal(size=(D, N)) * 0.5
    for e in (e_g, e_o):
        e.set_integrator(lf)
        e.set_position(th0)
    kernel = A.HMCKernel(A.Trajectory(TS, lf, TC(max_depth=7)))
    for _ in range(4):
        e_g.transition(kernel)
        e_o.transition(kernel)
        same = assert_close_state(e_g, e_o)
        if not same.all():
            e_g.set_position(e_o.phasepoint().theta)
    assert e_g.info("dense_pool") == 1   # (round 6: both criteria on the point-pool kernel k_d_tree2<…, CRIT>, no longer the copying one)
    e_g.close(); e_o.close()


@pytest.mark.parametrize("TC", [A.GeneralisedNoUTurn, A.ClassicNoUTurn, A.StrictGeneralisedNoUTurn])
@pytest.mark.parametrize("target", ["iso", "funnel"])
@pytest.mark.parametrize("metric", ["unit", "diag_chain", "diag_shared", "dense"])
@pytest.mark.parametrize("TS", [A.MultinomialTS, A.SliceTS])
def test_hip_nuts_with_user_density(hip, oracle, rng, target, metric, TS, TC):
    D, N = 10, 200
    m = make_metric(metric, D, N, rng)
    lf = A.Leapfrog(np.full(N, 0.25) * (0.5 + rng.random(N)))
    e_ext, e_ref = engines(hip, oracle, target, m, N, lf)
    th0 = rng.normal(size=(D, N))
    e_ext.set_position(th0)
    e_ref.set_position(th0)
    kernel = A.HMCKernel(A.Trajectory(TS, lf, TC(max_depth=7)))
    for _ in range(3):
        e_ext.transition(kernel)
        e_ref.transition(kernel)
        same = assert_close_state(e_ext, e_ref)
        if not same.all():  # a chain that took another decision carries on from another state: re-align it
            e_ext.set_position(e_ref.phasepoint().theta)
    assert e_ext.info("iteration") == e_ref.info("iteration") == 3
    e_ext.close(); e_ref.close()


@pytest.mark.parametrize("TS", [A.EndPointTS, A.MultinomialTS])
@pytest.mark.parametrize("metric", ["unit", "diag_chain", "dense"])
def test_hip_static_hmc_with_user_density(hip, oracle, rng, metric, TS):
    D, N = 12, 130
    m = make_metric(metric, D, N, rng)
    lf = A.JitteredLeapfrog(0.15, 0.3)
    e_ext, e_ref = engines(hip, oracle, "funnel", m, N, lf)
    th0 = rng.normal(size=(D, N))
    e_ext.set_position(th0)
    e_ref.set_position(th0)
    kernel = A.HMCKernel(A.Trajectory(TS, lf, A.FixedNSteps(9)))
    for _ in range(4):  # (the coupled forward / backward split differs from transition to transition)
        e_ext.transition(kernel)
        e_ref.transition(kernel)
        same = assert_close_state(e_ext, e_ref)
        assert same.all()
    e_ext.close(); e_ref.close()


@pytest.mark.parametrize("metric,target", [("dense", "dense"), ("dense", "funnel"), ("diag", "dense"), ("unit", "dense")])
def test_dense_engine_static_multinomial(hip, oracle, rng, metric, target):
    """static HMC with MultinomialTS (src/trajectory.jl:369-390) on the step-synchronous engine — energies-only
    passes, randcat, re-integration (csrc/ahmc_dense_mn.hpp) — against the oracle's stored trajectories"""
    D, N = 24, 150
    B = rng.normal(size=(D, D))
    P = B @ B.T / D + np.eye(D)
    tgt = A.DenseGaussian(P) if target == "dense" else A.Funnel(D)
    if metric == "dense":
        C2 = rng.normal(size=(D, D))
        m = A.DenseEuclideanMetric(C2 @ C2.T / D + np.eye(D))
    elif metric == "diag":
        m = A.DiagEuclideanMetric(np.asfortranarray(0.5 + rng.random((D, N))))
    else:
        m = A.UnitEuclideanMetric(D)
    lf = A.Leapfrog(np.full(N, 0.12) * (0.5 + rng.random(N)))
    h = A.Hamiltonian(m, tgt)
    e_g = A.Engine(h, N, rng=5, lib=hip)
    e_o = A.Engine(h, N, rng=5, lib=oracle)
    th0 = rng.normal(size=(D, N)) * 0.5
    for e in (e_g, e_o):
        e.set_integrator(lf)
        e.set_position(th0)
    kernel = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.FixedNSteps(8)))
    for _ in range(4):
        e_g.transition(kernel)
        e_o.transition(kernel)
        sa, sb = e_g.stats(), e_o.stats()
        za, zb = e_g.phasepoint(), e_o.phasepoint()
        close = np.all(np.isclose(za.theta, zb.theta, rtol=1e-8, atol=1e-8), axis=0)  # (a different categorical pick shows as a different point)
        close = PU.check_flips(close, PU.decision_margin(e_o), np.float64, f"dense static multinomial {metric}/{target}")
        np.testing.assert_allclose(sa["acceptance_rate"], sb["acceptance_rate"], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(sa["hamiltonian_energy"][close], sb["hamiltonian_energy"][close], rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(za.r[:, close], zb.r[:, close], rtol=1e-8, atol=1e-8)
        assert (sa["n_steps"] == 8).all() and sa["is_accept"].all()
        if not close.all():
            e_g.set_position(zb.theta)
    e_g.close(); e_o.close()


@pytest.mark.parametrize("metric", ["unit", "diag_chain", "dense"])
def test_hip_find_good_stepsize_with_user_density(hip, oracle, rng, metric):
    D, N = 16, 128
    m = make_metric(metric, D, N, rng)
    e_ext, e_ref = engines(hip, oracle, "iso", m, N, A.Leapfrog(0.1))
    th0 = rng.normal(size=(D, N))
    e_ext.set_position(th0)
    e_ref.set_position(th0)
    eps_a, eps_b = e_ext.find_good_stepsize(), e_ref.find_good_stepsize()
    # powers of two times bisection midpoints: equal unless a test of the search sat on a tie
    PU.check_equal_or_near_tie(eps_a, eps_b, PU.decision_margin(e_ref), np.float64, f"find_good_stepsize, user density, {metric}")
    np.testing.assert_allclose(e_ext.phasepoint().theta, th0, rtol=0, atol=0)
    assert e_ext.info("iteration") == 0
    e_ext.close(); e_ref.close()


def test_hip_batch_of_transitions_and_device_arrays(hip, oracle, rng):
    """n_trans = 4 in one run; the evaluation is done with torch on the device, reading θ at ahmc_theta_ptr and
    handing device pointers to ahmc_ext_advance (no host staging)"""
    import torch

    D, N = 8, 300
    m = A.DiagEuclideanMetric(np.asfortranarray(0.5 + rng.random((D, N))))
    lf = A.Leapfrog(np.full(N, 0.3))
    e_ext, e_ref = engines(hip, oracle, "iso", m, N, lf)
    th0 = rng.normal(size=(D, N))
    e_ext.set_position(th0)
    e_ref.set_position(th0)
    kernel = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=6)))
    k = kernel.cfg()
    e_ext._call("ahmc_ext_begin", C.byref(k), 4)
    n = C.c_int64()
    theta_dev = torch.empty((N, D), dtype=torch.float64, device="cuda")  # row c = chain c: the (D,N) column-major layout
    trips = 0
    while True:
        e_ext._call("ahmc_ext_pending", C.byref(n), None, theta_dev.data_ptr())
        if n.value == 0:
            break
        lp = (-(1.8378770664093454835606594728112 + theta_dev * theta_dev) / 2).sum(dim=1)
        gneg = theta_dev.clone()  # -∇ℓπ = θ
        torch.cuda.synchronize()
        e_ext._call("ahmc_ext_advance", lp.data_ptr(), gneg.data_ptr())
        trips += 1
    for _ in range(4):
        e_ref.transition(kernel)
    assert e_ext.info("iteration") == 4 and trips > 4
    assert_close_state(e_ext, e_ref)  # (torch's reduction order differs from the oracle's loop)
    e_ext.close(); e_ref.close()


def test_hip_unsupported_combinations_and_state_errors(hip, rng):
    D, N = 4, 8
    lf = A.Leapfrog(0.1)
    e = A.Engine(A.Hamiltonian(A.UnitEuclideanMetric(D), A.ExternalTarget(D, iso_fn)), N, lib=hip)
    e.set_integrator(lf)
    e.set_position(rng.normal(size=(D, N)))
    kp = A.HMCKernel(A.PartialMomentumRefreshment(0.5), A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn())).cfg()
    with pytest.raises(A.UnsupportedError):  # the next momentum depends on the one a transition ends with: one per run
        e._call("ahmc_ext_begin", C.byref(kp), 2)
    k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn())).cfg()
    e._call("ahmc_ext_begin", C.byref(k), 1)
    with pytest.raises(A.AHMCError, match="run is in progress"):
        e.set_integrator(lf)
    e._call("ahmc_ext_cancel")
    e.set_integrator(lf)
    e.close()


@pytest.mark.parametrize("nuts", [False, True])
@pytest.mark.parametrize("metric,target", [("dense", "dense"), ("diag", "dense"), ("dense", "funnel")])
def test_dense_engine_partial_refreshment(hip, oracle, rng, metric, target, nuts):
    """PartialMomentumRefreshment(α) (src/hamiltonian.jl:243-254) on the step-synchronous engine: refresh, then a
    run of transitions through ahmc_sample (the momentum of each transition mixes in the one the previous ended with)"""
    D, N = 20, 140
    B = rng.normal(size=(D, D))
    tgt = A.DenseGaussian(B @ B.T / D + np.eye(D)) if target == "dense" else A.Funnel(D)
    if metric == "dense":
        C2 = rng.normal(size=(D, D))
        m = A.DenseEuclideanMetric(C2 @ C2.T / D + np.eye(D))
    else:
        m = A.DiagEuclideanMetric(np.asfortranarray(0.5 + rng.random((D, N))))
    lf = A.Leapfrog(np.full(N, 0.1) * (0.5 + rng.random(N)))
    h = A.Hamiltonian(m, tgt)
    e_g, e_o = A.Engine(h, N, rng=9, lib=hip), A.Engine(h, N, rng=9, lib=oracle)
    th0, r0 = rng.normal(size=(D, N)) * 0.5, rng.normal(size=(D, N))
    for e in (e_g, e_o):
        e.set_integrator(lf)
        e.set_position(th0, r0)
        e.refresh(A.PartialMomentumRefreshment(0.4))
    np.testing.assert_allclose(e_g.phasepoint().r, e_o.phasepoint().r, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(e_g.phasepoint().lk.value, e_o.phasepoint().lk.value, rtol=1e-9, atol=1e-9)
    tc = A.GeneralisedNoUTurn(max_depth=6) if nuts else A.FixedNSteps(6)
    kernel = A.HMCKernel(A.PartialMomentumRefreshment(0.4), A.Trajectory(A.MultinomialTS if nuts else A.EndPointTS, lf, tc))
    for _ in range(3):
        e_g.run(kernel, 1)
        e_o.run(kernel, 1)
        sa, sb = e_g.stats(), e_o.stats()
        same = (sa["n_steps"] == sb["n_steps"]) & (sa["is_accept"] == sb["is_accept"])
        same = PU.check_flips(same, PU.decision_margin(e_o), np.float64, f"partial refreshment {metric}/{target}")
        za, zb = e_g.phasepoint(), e_o.phasepoint()
        np.testing.assert_allclose(za.theta[:, same], zb.theta[:, same], rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(za.r[:, same], zb.r[:, same], rtol=1e-8, atol=1e-8)
        if not same.all():
            e_g.set_position(zb.theta, zb.r)
    e_g.close(); e_o.close()


def test_hip_user_density_partial_refreshment(hip, oracle, rng):
    D, N = 10, 100
    m = make_metric("diag_chain", D, N, rng)
    lf = A.Leapfrog(np.full(N, 0.2))
    e_ext, e_ref = engines(hip, oracle, "iso", m, N, lf)
    th0 = rng.normal(size=(D, N))
    e_ext.set_position(th0)
    e_ref.set_position(th0)
    for kernel in (A.HMCKernel(A.PartialMomentumRefreshment(0.6), A.Trajectory(A.EndPointTS, lf, A.FixedNSteps(5))),
                   A.HMCKernel(A.PartialMomentumRefreshment(0.6), A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=5)))):
        for _ in range(2):
            e_ext.transition(kernel)
            e_ref.run(kernel, 1)
            assert_close_state(e_ext, e_ref)
    e_ext.close(); e_ref.close()


@pytest.mark.parametrize("metric,target", [("dense", "dense"), ("diag", "dense"), ("dense", "funnel")])
def test_dense_engine_tempered_leapfrog(hip, oracle, rng, metric, target):
    """TemperedLeapfrog (src/integrator.jl:174-209) on the step-synchronous engine: step(lf, h, z, n) both ways,
    static HMC with EndPointTS and MultinomialTS; NUTS stays unsupported there"""
    D, N = 18, 120
    B = rng.normal(size=(D, D))
    tgt = A.DenseGaussian(B @ B.T / D + np.eye(D)) if target == "dense" else A.Funnel(D)
    if metric == "dense":
        C2 = rng.normal(size=(D, D))
        m = A.DenseEuclideanMetric(C2 @ C2.T / D + np.eye(D))
    else:
        m = A.DiagEuclideanMetric(np.asfortranarray(0.5 + rng.random((D, N))))
    lf = A.TemperedLeapfrog(np.full(N, 0.08) * (0.5 + rng.random(N)), 1.05)
    h = A.Hamiltonian(m, tgt)
    e_g, e_o = A.Engine(h, N, rng=4, lib=hip), A.Engine(h, N, rng=4, lib=oracle)
    th0, r0 = rng.normal(size=(D, N)) * 0.5, rng.normal(size=(D, N))
    for e in (e_g, e_o):
        e.set_integrator(lf)
        e.set_position(th0, r0)
    for n in (7, -4):
        for e in (e_g, e_o):
            e.step(n)
        za, zb = e_g.phasepoint(), e_o.phasepoint()
        np.testing.assert_allclose(za.theta, zb.theta, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(za.r, zb.r, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(za.lk.value, zb.lk.value, rtol=1e-9, atol=1e-8)
    for TS in (A.EndPointTS, A.MultinomialTS):
        kernel = A.HMCKernel(A.Trajectory(TS, lf, A.FixedNSteps(7)))
        for _ in range(3):
            e_g.transition(kernel)
            e_o.transition(kernel)
            sa, sb = e_g.stats(), e_o.stats()
            za, zb = e_g.phasepoint(), e_o.phasepoint()
            close = np.all(np.isclose(za.theta, zb.theta, rtol=1e-8, atol=1e-8), axis=0) & (sa["is_accept"] == sb["is_accept"])
            close = PU.check_flips(close, PU.decision_margin(e_o), np.float64, f"tempered static {TS.__name__}")
            np.testing.assert_allclose(sa["hamiltonian_energy"][close], sb["hamiltonian_energy"][close], rtol=1e-8, atol=1e-8)
            if not close.all():
                e_g.set_position(zb.theta, zb.r)
    # NUTS: every leaf is step(lf, h, z, 1), tempered up before its first half-step and down after its second
    for TC in (A.GeneralisedNoUTurn, A.StrictGeneralisedNoUTurn):
        kernel = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, TC(max_depth=6)))
        for _ in range(3):
            e_g.transition(kernel)
            e_o.transition(kernel)
            same = assert_close_state(e_g, e_o)
            if not same.all():
                e_g.set_position(e_o.phasepoint().theta, e_o.phasepoint().r)
    e_g.close(); e_o.close()


@pytest.mark.parametrize("TS,TC", [(A.EndPointTS, None), (A.MultinomialTS, None), (A.MultinomialTS, A.GeneralisedNoUTurn), (A.SliceTS, A.ClassicNoUTurn)])
def test_hip_user_density_tempered(hip, oracle, rng, TS, TC):
    D, N = 9, 80
    m = make_metric("diag_chain", D, N, rng)
    lf = A.TemperedLeapfrog(np.full(N, 0.15), 1.05)
    e_ext, e_ref = engines(hip, oracle, "funnel", m, N, lf)
    th0 = rng.normal(size=(D, N))
    e_ext.set_position(th0)
    e_ref.set_position(th0)
    kernel = A.HMCKernel(A.Trajectory(TS, lf, A.FixedNSteps(6) if TC is None else TC(max_depth=6)))
    for _ in range(3):
        e_ext.transition(kernel)
        e_ref.transition(kernel)
        assert_close_state(e_ext, e_ref)
    e_ext.close(); e_ref.close()


@pytest.mark.parametrize("TS", [A.EndPointTS, A.MultinomialTS])
def test_hip_batch_of_static_transitions(hip, oracle, rng, TS):
    D, N = 6, 90
    m = make_metric("diag_shared", D, N, rng)
    lf = A.Leapfrog(np.full(N, 0.3))
    e_ext, e_ref = engines(hip, oracle, "iso", m, N, lf)
    th0 = rng.normal(size=(D, N))
    e_ext.set_position(th0)
    e_ref.set_position(th0)
    kernel = A.HMCKernel(A.Trajectory(TS, lf, A.FixedNSteps(5)))
    k = kernel.cfg()
    e_ext._call("ahmc_ext_begin", C.byref(k), 3)
    e_ext._ext_drive()
    for _ in range(3):
        e_ref.transition(kernel)
    assert e_ext.info("iteration") == 3
    assert_close_state(e_ext, e_ref)
    e_ext.close(); e_ref.close()


def test_baseline_cfg1_on_the_gpu(hip, oracle):
    """BASELINE.json configs[0] (D = 10, Unit metric, static HMC with 16 leapfrogs, 1 024 chains) run by the fused
    k_hmc kernel against the oracle on the same streams (the kernel is the measured build's; only this test is new)"""
    D, N, L, seed = 10, 1024, 16, 0x5EED0001
    th0 = np.random.default_rng(seed).random((D, N))
    lf = A.Leapfrog(0.1)
    h = A.Hamiltonian(A.UnitEuclideanMetric(D), A.IsoGaussian(D))
    kernel = A.HMCKernel(A.Trajectory(A.EndPointTS, lf, A.FixedNSteps(L)))
    e_g, e_o = A.Engine(h, N, rng=seed, lib=hip), A.Engine(h, N, rng=seed, lib=oracle)
    for e in (e_g, e_o):
        e.set_integrator(lf)
        e.set_position(th0)
    for _ in range(5):
        e_g.transition(kernel)
        e_o.transition(kernel)
        sa, sb = e_g.stats(), e_o.stats()
        same = PU.check_flips(sa["is_accept"] == sb["is_accept"], PU.decision_margin(e_o), np.float64, "cfg1 on the GPU")
        np.testing.assert_allclose(e_g.phasepoint().theta[:, same], e_o.phasepoint().theta[:, same], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(sa["hamiltonian_energy"][same], sb["hamiltonian_energy"][same], rtol=1e-9, atol=1e-9)
        if not same.all():
            e_g.set_position(e_o.phasepoint().theta, e_o.phasepoint().r)
    e_g.close(); e_o.close()


def test_plain_c_program_on_the_hip_engine(hip, tmp_path):
    """tests/c_abi/ask_tell_demo.c (plain C99: user log-density through ask / tell, find_good_stepsize, step-size
    adaptation) linked against libahmc_hip.so — no Python, no torch in that process"""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sodir = os.path.dirname(A.hip_library_path())
    exe = str(tmp_path / "ask_tell_demo")
    subprocess.run(["gcc", "-std=c99", "-O1", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c_abi", "ask_tell_demo.c"), "-o", exe,
                    "-L", sodir, "-lahmc_hip", "-lm", f"-Wl,-rpath,{sodir}", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"],
                   check=True, capture_output=True)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=200)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.startswith("ok ")


def test_hip_randomised_configurations(hip, oracle):
    """60 random configurations on the step-synchronous engine — half with a user density through ahmc_ext_*, half with
    the dense metric / dense target — every kernel kind, sampler, criterion, integrator and refreshment, against the
    oracle running the same density built in"""
    master = np.random.default_rng(424242)
    mismatched = total = 0
    for case in range(60):
        ext = case % 2 == 0
        D = int(master.integers(2, 12))
        N = int(master.integers(3, 40))
        integ = int(master.integers(0, 3))
        eps = float(10 ** master.uniform(-1.3, -0.3)) * (0.5 + master.random(N))
        lf = (A.Leapfrog(eps), A.JitteredLeapfrog(float(eps[0]), 0.4), A.TemperedLeapfrog(eps, 1.0 + 0.1 * float(master.random())))[integ]
        alpha = [0.0, 0.0, float(master.uniform(0.1, 0.9))][int(master.integers(0, 3))]
        refresh = A.PartialMomentumRefreshment(alpha) if alpha else A.FullMomentumRefreshment()
        if master.random() < 0.6:
            TS_ = [A.MultinomialTS, A.SliceTS][int(master.integers(0, 2))]
            TC_ = [A.ClassicNoUTurn, A.GeneralisedNoUTurn, A.StrictGeneralisedNoUTurn][int(master.integers(0, 3))]
            traj = A.Trajectory(TS_, lf, TC_(max_depth=int(master.integers(1, 7))))
        else:
            TS_ = [A.EndPointTS, A.MultinomialTS][int(master.integers(0, 2))]
            traj = A.Trajectory(TS_, lf, A.FixedNSteps(int(master.integers(1, 9))))
        kernel = A.HMCKernel(refresh, traj)
        seed = int(master.integers(0, 2 ** 40))
        th0 = master.normal(size=(D, N))
        if ext:
            target = ["iso", "funnel"][int(master.integers(0, 2))]
            m = make_metric(["unit", "diag_chain", "diag_shared", "dense"][int(master.integers(0, 4))], D, N, master)
            e_g, e_o = engines(hip, oracle, target, m, N, lf, seed=seed)
        else:
            B = master.normal(size=(D, D))
            dense_target = master.random() < 0.6
            tgt = A.DenseGaussian(B @ B.T / D + np.eye(D)) if dense_target else A.Funnel(D)
            m = make_metric("dense" if (not dense_target or master.random() < 0.6) else ["unit", "diag_chain"][int(master.integers(0, 2))], D, N, master)
            h = A.Hamiltonian(m, tgt)
            e_g, e_o = A.Engine(h, N, rng=seed, lib=hip), A.Engine(h, N, rng=seed, lib=oracle)
            for e in (e_g, e_o):
                e.set_integrator(lf)
        e_g.set_position(th0)
        e_o.set_position(th0)
        tag = (case, ext, D, N, integ, alpha, type(traj.termination_criterion).__name__, traj.TS.__name__)
        for _ in range(2):
            if ext:
                e_g.transition(kernel)
            else:
                e_g.run(kernel, 1)
            e_o.run(kernel, 1)
            sa, sb = e_g.stats(), e_o.stats()
            za, zb = e_g.phasepoint(), e_o.phasepoint()
            same = (sa["n_steps"] == sb["n_steps"]) & (sa["is_accept"] == sb["is_accept"]) & np.all(np.isclose(za.theta, zb.theta, rtol=1e-7, atol=1e-7), axis=0)
            total += N
            mismatched += int((~same).sum())
            np.testing.assert_allclose(za.r[:, same], zb.r[:, same], rtol=1e-7, atol=1e-7, err_msg=str(tag))
            np.testing.assert_allclose(sa["hamiltonian_energy"][same], sb["hamiltonian_energy"][same], rtol=1e-7, atol=1e-7, err_msg=str(tag))
            np.testing.assert_allclose(sa["step_size"], sb["step_size"], rtol=1e-12, err_msg=str(tag))
            if not same.all():
                e_g.set_position(zb.theta, zb.r)
        e_g.close(); e_o.close()
    assert mismatched <= 0.01 * total, (mismatched, total)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("metric", ["unit", "diag_chain", "diag_shared"])
@pytest.mark.parametrize("where", ["host", "device"])
def test_hip_split_step_lf_pre_post(hip, oracle, rng, metric, dtype, where):
    """ahmc_lf_pre / ahmc_lf_post around a caller-side gradient (src/integrator.jl:231-243) on the HIP engine ==
    the oracle's fused step on the same density, built in — with host arrays (staged, the call owns them only
    until it returns) and with device arrays (torch), Leapfrog and TemperedLeapfrog, forwards and backwards"""
    import torch

    D, N = 24, 300
    m = make_metric(metric, D, N, rng)
    fn, builtin = TARGETS["funnel"]
    th0, r0 = rng.normal(size=(D, N)) * 0.5, rng.normal(size=(D, N))
    rt = 1e-10 if dtype == np.float64 else 2e-3
    for lf in (A.Leapfrog(np.full(N, 0.05) * (0.5 + rng.random(N))), A.TemperedLeapfrog(np.full(N, 0.04), 1.05)):
        ext = A.Engine(A.Hamiltonian(m, A.ExternalTarget(D, lambda th: fn(np.asarray(th, dtype=np.float64)))), N, dtype=dtype, lib=hip)
        ref = A.Engine(A.Hamiltonian(m, builtin(D)), N, dtype=dtype, lib=oracle)
        for e in (ext, ref):
            e.set_integrator(lf)
            e.set_position(th0, r0)
        for n_steps in (7, -4):
            ref.step(n_steps)
            if where == "host":
                ext.step(n_steps)
            else:
                n, fwd = abs(n_steps), 1 if n_steps > 0 else 0
                tdt = torch.float64 if dtype == np.float64 else torch.float32
                for i in range(1, n + 1):
                    ext._call("ahmc_lf_pre", fwd, i, n)
                    ext.sync()
                    th = ext.theta()
                    lp, g = fn(np.asarray(th, dtype=np.float64))
                    lp_d = torch.as_tensor(np.ascontiguousarray(lp), dtype=tdt).cuda()
                    g_d = torch.as_tensor(np.ascontiguousarray((-g).T), dtype=tdt).cuda().contiguous()  # row c = chain c
                    torch.cuda.synchronize()
                    ext._call("ahmc_lf_post", fwd, i, n, lp_d.data_ptr(), g_d.data_ptr())
                    ext.sync()
            za, zb = ext.phasepoint(), ref.phasepoint()
            np.testing.assert_allclose(za.theta, zb.theta, rtol=rt, atol=rt)
            np.testing.assert_allclose(za.r, zb.r, rtol=rt, atol=rt)
            np.testing.assert_allclose(za.lp.value, zb.lp.value, rtol=rt, atol=rt * 10)
            np.testing.assert_allclose(za.lk.value, zb.lk.value, rtol=rt, atol=rt * 10)
            np.testing.assert_allclose(za.lp.gradient, zb.lp.gradient, rtol=rt, atol=rt * 10)
        ext.close(); ref.close()
