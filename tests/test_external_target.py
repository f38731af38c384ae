"""Whole transitions with an EXTERNAL target: the ask / tell calls ahmc_ext_* (include/ahmc_hip.h).

The user's log-density — `h.∂ℓπ∂θ(θ)` of src/hamiltonian.jl:45-48, the LogDensityProblems surface of
src/AdvancedHMC.jl:163-186 — stays on the caller's side; the engine hands control back whenever a
leapfrog needs (ℓπ, ∇ℓπ).  CPU part (this file, not gpu-marked): the protocol and the Python driver
(`Engine._ext_drive`) on the oracle, where a run with the callback must reproduce the run with the
same density built in, bit for bit.  The HIP engine's side of the same protocol is compared with the
oracle in the gpu-marked tests at the bottom.
"""
import ctypes as C
import math

import numpy as np
import pytest

import ahmc_amd as A

LOG2PI = 1.8378770664093454835606594728112


def iso_fn(theta):
    """test/common.jl:40-44 with m = 0, s = 1, summed in index order like the oracle's loop"""
    terms = -(LOG2PI + theta * theta) / 2
    return np.cumsum(terms, axis=0)[-1], -theta


def funnel_fn(theta):
    """Neal's funnel as the oracle writes it (same libm calls, same association), one chain at a time"""
    D, N = theta.shape
    lp = np.empty(N)
    g = np.empty_like(theta)
    for c in range(N):
        th = theta[:, c]
        y = float(th[0])
        ss = 0.0
        for d in range(1, D):
            ss += float(th[d]) * float(th[d])
        if math.isfinite(y) and -y < 700:
            ey = math.exp(-y)
        else:
            ey = float(np.exp(np.float64(-y)))
        nm1 = float(D - 1)
        lp[c] = -(LOG2PI + 2 * math.log(3.0) + y * y / 9) / 2 - nm1 * (LOG2PI + y) / 2 - ss * ey / 2
        g[0, c] = -y / 9 - nm1 / 2 + ss * ey / 2
        g[1:, c] = -th[1:] * ey
    return lp, g


TARGETS = {"iso": (iso_fn, A.IsoGaussian), "funnel": (funnel_fn, A.Funnel)}


def make_metric(kind, D, N, rng):
    if kind == "unit":
        return A.UnitEuclideanMetric(D)
    if kind == "diag_chain":
        return A.DiagEuclideanMetric(np.asfortranarray(0.5 + rng.random((D, N))))
    if kind == "diag_shared":
        return A.DiagEuclideanMetric(0.5 + rng.random(D))
    B = rng.normal(size=(D, D))
    return A.DenseEuclideanMetric(B @ B.T / D + np.eye(D))


def pair(lib, target, metric, N, lf, seed=7, dtype=np.float64):
    """(engine with the callback, engine with the same density built in), same seeds"""
    fn, builtin = TARGETS[target]
    D = metric.D
    calls = {"n": 0}

    def counted(theta):
        calls["n"] += 1
        return fn(np.asarray(theta, dtype=np.float64))

    e_ext = A.Engine(A.Hamiltonian(metric, A.ExternalTarget(D, counted)), N, dtype=dtype, rng=seed, lib=lib)
    e_ref = A.Engine(A.Hamiltonian(metric, builtin(D)), N, dtype=dtype, rng=seed, lib=lib)
    for e in (e_ext, e_ref):
        e.set_integrator(lf)
    return e_ext, e_ref, calls


def assert_same_state(e_ext, e_ref, exact=True):
    za, zb = e_ext.phasepoint(), e_ref.phasepoint()
    sa, sb = e_ext.stats(), e_ref.stats()
    for k in ("n_steps", "is_accept", "tree_depth", "numerical_error"):
        np.testing.assert_array_equal(sa[k], sb[k], err_msg=k)
    cmp = np.testing.assert_array_equal if exact else (lambda a, b, err_msg="": np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-12, err_msg=err_msg))
    for k in ("acceptance_rate", "log_density", "hamiltonian_energy", "hamiltonian_energy_error", "max_hamiltonian_energy_error", "step_size"):
        cmp(sa[k], sb[k], err_msg=k)
    cmp(za.theta, zb.theta, err_msg="theta")
    cmp(za.r, zb.r, err_msg="r")
    cmp(za.lp.gradient, zb.lp.gradient, err_msg="grad")
    cmp(za.lp.value, zb.lp.value, err_msg="lp")
    cmp(za.lk.value, zb.lk.value, err_msg="lk")


@pytest.mark.parametrize("target", ["iso", "funnel"])
@pytest.mark.parametrize("metric", ["unit", "diag_chain", "dense"])
@pytest.mark.parametrize("TS,TC", [(A.MultinomialTS, A.GeneralisedNoUTurn), (A.SliceTS, A.GeneralisedNoUTurn),
                                   (A.MultinomialTS, A.ClassicNoUTurn), (A.SliceTS, A.StrictGeneralisedNoUTurn)])
def test_oracle_nuts_with_callback_equals_builtin(oracle, rng, target, metric, TS, TC):
    D, N = 6, 24
    m = make_metric(metric, D, N, rng)
    lf = A.Leapfrog(np.full(N, 0.3) * (0.5 + rng.random(N)))
    e_ext, e_ref, calls = pair(oracle, target, m, N, lf)
    th0 = rng.normal(size=(D, N))
    e_ext.set_position(th0)
    e_ref.set_position(th0)
    kernel = A.HMCKernel(A.Trajectory(TS, lf, TC(max_depth=6)))
    for _ in range(3):
        e_ext.transition(kernel)
        e_ref.transition(kernel)
        assert_same_state(e_ext, e_ref)
    assert e_ext.info("iteration") == e_ref.info("iteration") == 3
    assert calls["n"] > 3
    e_ext.close(); e_ref.close()


@pytest.mark.parametrize("TS", [A.EndPointTS, A.MultinomialTS])
@pytest.mark.parametrize("metric", ["unit", "diag_shared", "dense"])
def test_oracle_static_hmc_with_callback_equals_builtin(oracle, rng, TS, metric):
    D, N = 5, 16
    m = make_metric(metric, D, N, rng)
    lf = A.JitteredLeapfrog(0.2, 0.3)
    e_ext, e_ref, _ = pair(oracle, "funnel", m, N, lf)
    th0 = rng.normal(size=(D, N))
    e_ext.set_position(th0)
    e_ref.set_position(th0)
    kernel = A.HMCKernel(A.Trajectory(TS, lf, A.FixedNSteps(7)))
    for _ in range(3):
        e_ext.transition(kernel)
        e_ref.transition(kernel)
        assert_same_state(e_ext, e_ref)
    e_ext.close(); e_ref.close()


def test_oracle_find_good_stepsize_with_callback(oracle, rng):
    D, N = 8, 12
    m = make_metric("diag_chain", D, N, rng)
    e_ext, e_ref, calls = pair(oracle, "iso", m, N, A.Leapfrog(0.1))
    th0 = rng.normal(size=(D, N))
    e_ext.set_position(th0)
    e_ref.set_position(th0)
    eps_a = e_ext.find_good_stepsize()
    eps_b = e_ref.find_good_stepsize()
    np.testing.assert_array_equal(eps_a, eps_b)
    assert calls["n"] > 1 and len(set(eps_a)) > 1
    # the search leaves the phase point and the iteration counter alone
    np.testing.assert_array_equal(e_ext.phasepoint().theta, th0)
    assert e_ext.info("iteration") == 0
    e_ext.close(); e_ref.close()


def test_batch_of_transitions_runs_chains_asynchronously(oracle, rng):
    """ahmc_ext_begin(n_trans = 4): a chain that ends a transition starts its next one while others are
    still in an earlier one; the result is that of four single transitions"""
    D, N = 4, 10
    m = make_metric("unit", D, N, rng)
    lf = A.Leapfrog(np.full(N, 0.4))
    e_ext, e_ref, calls = pair(oracle, "iso", m, N, lf)
    th0 = rng.normal(size=(D, N))
    e_ext.set_position(th0)
    e_ref.set_position(th0)
    kernel = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=5)))
    k = kernel.cfg()
    e_ext._call("ahmc_ext_begin", C.byref(k), 4)
    trips = e_ext._ext_drive()
    for _ in range(4):
        e_ref.transition(kernel)
    assert_same_state(e_ext, e_ref)
    assert e_ext.info("iteration") == 4 and trips == calls["n"] - 1  # (set_position evaluated once)
    e_ext.close(); e_ref.close()


@pytest.mark.parametrize("TS", [A.EndPointTS, A.MultinomialTS])
def test_batch_of_static_transitions(oracle, rng, TS):
    D, N = 4, 10
    m = make_metric("diag_shared", D, N, rng)
    lf = A.Leapfrog(np.full(N, 0.3))
    e_ext, e_ref, _ = pair(oracle, "iso", m, N, lf)
    th0 = rng.normal(size=(D, N))
    e_ext.set_position(th0)
    e_ref.set_position(th0)
    kernel = A.HMCKernel(A.Trajectory(TS, lf, A.FixedNSteps(5)))
    k = kernel.cfg()
    e_ext._call("ahmc_ext_begin", C.byref(k), 3)
    e_ext._ext_drive()
    for _ in range(3):
        e_ref.transition(kernel)
    assert_same_state(e_ext, e_ref)
    assert e_ext.info("iteration") == 3
    e_ext.close(); e_ref.close()


def test_pending_list_and_partial_evaluation(oracle, rng):
    """`fn(θ, chains)`: only the listed columns are evaluated; garbage elsewhere must be ignored"""
    D, N = 3, 9
    seen = []

    def fn(theta, chains):
        seen.append(np.array(chains))
        lp = np.full(N, np.nan)
        g = np.full((D, N), np.nan)
        l, gg = iso_fn(theta[:, chains])
        lp[chains] = l
        g[:, chains] = gg
        return lp, g

    m = A.UnitEuclideanMetric(D)
    lf = A.Leapfrog(np.full(N, 0.5))
    tgt = A.ExternalTarget(D, fn, takes_chains=True)
    e_ext = A.Engine(A.Hamiltonian(m, tgt), N, rng=3, lib=oracle)
    e_ref = A.Engine(A.Hamiltonian(m, A.IsoGaussian(D)), N, rng=3, lib=oracle)
    th0 = rng.normal(size=(D, N))
    lp0, g0 = iso_fn(th0)
    for e in (e_ext, e_ref):
        e.set_integrator(lf)
    keep = [np.asfortranarray(th0), np.zeros((D, N), order="F"), np.ascontiguousarray(lp0), np.asfortranarray(-g0)]  # (alive across the call)
    e_ext._call("ahmc_set_phasepoint", *[A.capi.as_ptr(a) for a in keep])
    e_ref.set_position(th0)
    kernel = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=4)))
    e_ext.transition(kernel)
    e_ref.transition(kernel)
    assert_same_state(e_ext, e_ref)
    assert all(len(s) >= 1 for s in seen) and min(len(s) for s in seen) < N  # the tail of the run asks for fewer chains
    e_ext.close(); e_ref.close()


def test_sample_with_external_target_and_stan_adaptor(oracle, rng):
    """sample(rng, h, κ, θ, n, adaptor, n_adapts) with a user density == the same run with it built in"""
    D, N = 4, 8
    th0 = rng.normal(size=(D, N))
    out = []
    for tgt in (A.ExternalTarget(D, iso_fn), A.IsoGaussian(D)):
        metric = A.DiagEuclideanMetric(np.ones((D, N), order="F"))
        h = A.Hamiltonian(metric, tgt)
        lf = A.Leapfrog(np.full(N, 0.2))
        kernel = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=5)))
        adaptor = A.StanHMCAdaptor(A.MassMatrixAdaptor(metric), A.StepSizeAdaptor(0.8, lf), init_buffer=5, term_buffer=5, window_size=5)
        thetas, stats = A.sample(11, h, kernel, th0, 40, adaptor, 30, lib=oracle)
        out.append((thetas, stats))
    (ta, sa), (tb, sb) = out
    for a, b in zip(ta, tb):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(sa[-1]["step_size"], sb[-1]["step_size"])
    assert len({float(x) for x in sa[-1]["step_size"]}) > 1  # adapted per chain


def test_protocol_errors(oracle, rng):
    D, N = 3, 4
    lf = A.Leapfrog(0.1)
    kernel = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn()))
    k = kernel.cfg()
    # built-in target: STATE
    e = A.Engine(A.Hamiltonian(A.UnitEuclideanMetric(D), A.IsoGaussian(D)), N, lib=oracle)
    e.set_integrator(lf)
    e.set_position(rng.normal(size=(D, N)))
    with pytest.raises(A.AHMCError, match="not AHMC_TARGET_EXTERNAL"):
        e._call("ahmc_ext_begin", C.byref(k), 1)
    e.close()
    # external target
    e = A.Engine(A.Hamiltonian(A.UnitEuclideanMetric(D), A.ExternalTarget(D, iso_fn)), N, lib=oracle)
    e.set_integrator(lf)
    with pytest.raises(A.AHMCError, match="before set_phasepoint"):
        e._call("ahmc_ext_begin", C.byref(k), 1)
    e.set_position(rng.normal(size=(D, N)))
    with pytest.raises(A.AHMCError, match="no run in progress"):
        e._call("ahmc_ext_advance", None, None)
    with pytest.raises(A.ArgumentError):
        e._call("ahmc_ext_begin", C.byref(k), 0)
    n = C.c_int64(-1)
    e._call("ahmc_ext_pending", C.byref(n), None, None)
    assert n.value == 0
    e._call("ahmc_ext_begin", C.byref(k), 1)
    with pytest.raises(A.AHMCError, match="already in progress"):
        e._call("ahmc_ext_begin", C.byref(k), 1)
    e._call("ahmc_ext_pending", C.byref(n), None, None)
    assert n.value == N
    e._call("ahmc_ext_cancel")
    e._call("ahmc_ext_pending", C.byref(n), None, None)
    assert n.value == 0 and e.info("iteration") == 0

    # an exception inside the user's function cancels the run and propagates
    def boom(theta):
        raise ZeroDivisionError("user model failed")

    e.h.target.fn = boom
    with pytest.raises(ZeroDivisionError):
        e.transition(kernel)
    e._call("ahmc_ext_pending", C.byref(n), None, None)
    assert n.value == 0
    e.close()


def test_context_is_locked_during_a_run(oracle, rng):
    D, N = 3, 4
    lf = A.Leapfrog(0.1)
    e = A.Engine(A.Hamiltonian(A.UnitEuclideanMetric(D), A.ExternalTarget(D, iso_fn)), N, lib=oracle)
    e.set_integrator(lf)
    e.set_position(rng.normal(size=(D, N)))
    k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn())).cfg()
    e._call("ahmc_ext_begin", C.byref(k), 1)
    for call in (lambda: e.set_integrator(lf), lambda: e.seed(1), lambda: e.refresh(), lambda: e.adapt(1, 10)):
        with pytest.raises(A.AHMCError, match="run is in progress"):
            call()
    e.stats()  # reading is fine
    e._ext_drive()
    e.set_integrator(lf)
    e.close()


def test_oracle_partial_refreshment_with_callback(oracle, rng):
    """PartialMomentumRefreshment through the ask / tell run == through ahmc_sample with the density built in"""
    D, N = 5, 12
    m = make_metric("diag_chain", D, N, rng)
    lf = A.Leapfrog(np.full(N, 0.3))
    e_ext, e_ref, _ = pair(oracle, "iso", m, N, lf)
    th0 = rng.normal(size=(D, N))
    e_ext.set_position(th0)
    e_ref.set_position(th0)
    for kernel in (A.HMCKernel(A.PartialMomentumRefreshment(0.6), A.Trajectory(A.EndPointTS, lf, A.FixedNSteps(5))),
                   A.HMCKernel(A.PartialMomentumRefreshment(0.6), A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=5)))):
        for _ in range(3):
            e_ext.transition(kernel)
            e_ref.run(kernel, 1)
            assert_same_state(e_ext, e_ref)
    e_ext.close(); e_ref.close()


@pytest.mark.parametrize("TS,TC", [(A.EndPointTS, None), (A.MultinomialTS, None), (A.MultinomialTS, A.GeneralisedNoUTurn)])
def test_oracle_tempered_with_callback(oracle, rng, TS, TC):
    D, N = 5, 12
    m = make_metric("dense", D, N, rng)
    lf = A.TemperedLeapfrog(np.full(N, 0.2), 1.05)
    e_ext, e_ref, _ = pair(oracle, "funnel", m, N, lf)
    th0 = rng.normal(size=(D, N))
    e_ext.set_position(th0)
    e_ref.set_position(th0)
    kernel = A.HMCKernel(A.Trajectory(TS, lf, A.FixedNSteps(6) if TC is None else TC(max_depth=5)))
    for _ in range(3):
        e_ext.transition(kernel)
        e_ref.transition(kernel)
        assert_same_state(e_ext, e_ref)
    e_ext.close(); e_ref.close()


def test_randomised_configurations_callback_equals_builtin(oracle):
    """80 random configurations (kernel kind, sampler, criterion, metric, integrator, refreshment, sizes, batch
    length): the run through ahmc_ext_* with the density as a callback == the run with it built in, bit for bit"""
    master = np.random.default_rng(20260926)
    for case in range(80):
        D = int(master.integers(1, 7))
        N = int(master.integers(1, 9))
        target = ["iso", "funnel"][int(master.integers(0, 2))] if D > 1 else "iso"
        metric = ["unit", "diag_chain", "diag_shared", "dense"][int(master.integers(0, 4))]
        m = make_metric(metric, D, N, master)
        eps = float(10 ** master.uniform(-1.5, 0.3)) * (0.5 + master.random(N))
        integ = int(master.integers(0, 3))
        lf = (A.Leapfrog(eps), A.JitteredLeapfrog(float(eps[0]), 0.5), A.TemperedLeapfrog(eps, 1.0 + 0.2 * float(master.random())))[integ]
        alpha = [0.0, 0.0, float(master.uniform(0.1, 0.9))][int(master.integers(0, 3))]
        refresh = A.PartialMomentumRefreshment(alpha) if alpha else A.FullMomentumRefreshment()
        if master.random() < 0.6:
            TS_ = [A.MultinomialTS, A.SliceTS][int(master.integers(0, 2))]
            TC_ = [A.ClassicNoUTurn, A.GeneralisedNoUTurn, A.StrictGeneralisedNoUTurn][int(master.integers(0, 3))]
            traj = A.Trajectory(TS_, lf, TC_(max_depth=int(master.integers(1, 7)), delta_max=float([20.0, 1000.0][int(master.integers(0, 2))])))
        else:
            TS_ = [A.EndPointTS, A.MultinomialTS][int(master.integers(0, 2))]
            traj = A.Trajectory(TS_, lf, A.FixedNSteps(int(master.integers(1, 9))))
        kernel = A.HMCKernel(refresh, traj)
        n_trans = int(master.integers(1, 4))
        e_ext, e_ref, _ = pair(oracle, target, m, N, lf, seed=int(master.integers(0, 2 ** 40)))
        th0 = master.normal(size=(D, N)) * 1.5
        e_ext.set_position(th0)
        e_ref.set_position(th0)
        k = kernel.cfg()
        e_ext._call("ahmc_ext_begin", C.byref(k), n_trans)
        e_ext._ext_drive()
        e_ref.run(kernel, n_trans)
        za, zb = e_ext.phasepoint(), e_ref.phasepoint()
        sa, sb = e_ext.stats(), e_ref.stats()
        tag = (case, D, N, target, metric, integ, alpha, type(traj.termination_criterion).__name__, traj.TS.__name__, n_trans)
        np.testing.assert_array_equal(za.theta, zb.theta, err_msg=str(tag))
        np.testing.assert_array_equal(za.r, zb.r, err_msg=str(tag))
        for key in ("n_steps", "is_accept", "tree_depth", "numerical_error", "step_size", "acceptance_rate", "hamiltonian_energy"):
            np.testing.assert_array_equal(sa[key], sb[key], err_msg=str((key,) + tag))
        assert e_ext.info("iteration") == e_ref.info("iteration") == n_trans
        e_ext.close(); e_ref.close()
