"""Random SEQUENCES of calls on one context: the HIP engine against the oracle after every call.

tests/test_random_configurations.py draws a configuration and runs it from a fresh context.  A context also carries state between calls — the
dispatch order built from the step sizes, the prefetched normals of the launch that was expected to follow, launch lengths the scheduler
settled on, cached gradients, the adaptor, the "one scalar step size" flag, accumulators.  Here a seeded generator draws a sequence of a dozen calls
(transitions of changing kernels, bulk runs with and without an adaptor, new integrators / metrics / positions / seeds, leapfrog steps both ways,
find_good_stepsize, refresh, checkpoint round trips) and applies it to one context of either library; after every call the HIP engine's result is
held to the oracle's (the margin rule of tests/parity_util.py for what follows from decisions, 1e-8 for the rest on the agreeing chains), then the HIP
context is put on the oracle's complete state, so that the next call starts from identical inputs and anything stale in the HIP context shows."""
import os

import numpy as np
import pytest

import ahmc_amd as A
import parity_util as PU
from test_gpu_parity import compare_transition_stats

N_SEQ = int(os.environ.get("AHMC_RANDOM_SEQUENCES", "32"))
OPS = ["nuts", "nuts", "static", "run", "run", "run_adapt", "set_eps", "set_metric", "set_position", "refresh", "step", "find_eps", "seed",
       "adaptor", "checkpoint"]


def draw_sequence(i):
    rs = np.random.default_rng(30_000 + i)
    c = {"i": i, "D": int(rs.choice([1, 2, 5, 16, 33, 64, 100, 129, 300, 600])), "N": int(rs.choice([1, 2, 5, 64, 65, 130]))}
    c["target"] = str(rs.choice((["iso", "diag", "funnel", "hier"] if c["D"] >= 3 else ["iso", "diag"]) + (["dense"] if c["D"] <= 129 else [])))
    c["metric"] = str(rs.choice(["unit", "diag_chain", "diag_chain"] + (["dense"] if c["D"] <= 129 else [])))   # dense: the step-synchronous engine
    c["ops"] = [str(rs.choice(OPS)) for _ in range(12)]
    c["seed"] = int(rs.integers(1, 1 << 30))
    return c


def sync_from_oracle(g, o):
    """the HIP context takes the oracle's complete state (phase point, step sizes, metric, adaptor, RNG counter, accumulators)"""
    g.set_state(o.get_state())
    g.seed(A.PhiloxRNG(o._seed_now), iteration=o.info("iteration"))


def make_kernel(rs, lf, D, nuts):
    k = _make_kernel(rs, lf, D, nuts)
    alpha = float(rs.choice([0.0, 0.0, 0.3, 0.9]))
    return A.HMCKernel(A.PartialMomentumRefreshment(alpha), k.tau) if alpha else k


def _make_kernel(rs, lf, D, nuts):
    if nuts:
        TS = (A.MultinomialTS, A.SliceTS)[rs.integers(2)]
        TC = (A.GeneralisedNoUTurn, A.GeneralisedNoUTurn, A.ClassicNoUTurn, A.StrictGeneralisedNoUTurn)[rs.integers(4)]
        return A.HMCKernel(A.Trajectory(TS, lf, TC(max_depth=int(rs.integers(1, 6)), delta_max=float(rs.choice([1000.0, 10.0])))))
    TS = (A.EndPointTS, A.MultinomialTS)[rs.integers(2)]
    return A.HMCKernel(A.Trajectory(TS, lf, A.FixedNSteps(int(rs.integers(1, 9)))))


def make_lf(rs, D, N, base):
    eps = base * (0.5 + rs.random(N)) if rs.integers(2) else float(base * (0.5 + rs.random()))
    kind = rs.integers(4)
    if kind == 2:
        return A.JitteredLeapfrog(eps, float(rs.choice([0.1, 0.4])))
    if kind == 3:
        return A.TemperedLeapfrog(eps, float(rs.choice([1.02, 1.08])))
    return A.Leapfrog(eps)


def run_sequence(c, hip, oracle):
    rs = np.random.default_rng(c["seed"])
    D, N, dtype = c["D"], c["N"], np.float64
    from test_gpu_parity import make_target, _spd

    def new_metric():
        if c["metric"] == "dense":
            return A.DenseEuclideanMetric(_spd(D, rs))
        return A.DiagEuclideanMetric(np.asfortranarray(0.5 + rs.random((D, N))))

    metric = A.UnitEuclideanMetric((D, N)) if c["metric"] == "unit" else new_metric()
    h = A.Hamiltonian(metric, A.DenseGaussian(_spd(D, rs, 3.0)) if c["target"] == "dense" else make_target(c["target"], D, rs))
    base = (0.35 if c["target"] != "funnel" else 0.2) * D ** -0.25
    lf = make_lf(rs, D, N, base)
    seed = int(rs.integers(1, 1 << 16))
    g = A.Engine(h, N, dtype=dtype, rng=seed, lib=hip)
    o = A.Engine(h, N, dtype=dtype, rng=seed, lib=oracle)
    o._seed_now = seed
    log = []
    try:
        for e in (g, o):
            e.set_integrator(lf)
            e.set_position(0.5 * np.random.default_rng(c["seed"] + 1).normal(size=(D, N)))
            e.refresh()
        PU.reset_margin(o)
        adaptor_on = False
        for step, op in enumerate(c["ops"]):
            what = f"sequence {c['i']} (D={D} N={N} {c['target']}/{c['metric']}) call {step} {op}; before: {log}"
            if op in ("nuts", "static", "run", "run_adapt"):
                k = make_kernel(rs, lf, D, op != "static" and (op != "run" or rs.integers(2)))
                n = 1 if op in ("nuts", "static") else int(rs.integers(2, 6))
                i0 = o.info("iteration")
                adapt = op == "run_adapt" and adaptor_on
                for e in (g, o):
                    if op in ("nuts", "static"):
                        e.transition(k)
                    else:
                        # (iteration numbers continue: the Stan adaptor's own counter must match i_first, so adapting runs restart the adaptor below)
                        e.run(k, n, n if adapt else 0)
                sg, so = g.stats(), o.stats()
                clear = PU.decision_margin(o, reset=False) >= PU.MARGIN_BOUND[np.dtype(dtype)]
                same = compare_transition_stats(sg, so, dtype, o, what) & clear
                zg, zo = g.phasepoint(), o.phasepoint()
                tol = 1e-8 if not adapt else 1e-6
                np.testing.assert_allclose(zg.theta[:, same], zo.theta[:, same], rtol=tol, atol=tol, err_msg=what)
                np.testing.assert_allclose(zg.r[:, same], zo.r[:, same], rtol=tol, atol=tol, err_msg=what)
                if adapt:
                    np.testing.assert_allclose(g.get_stepsize()[same], o.get_stepsize()[same], rtol=1e-6, err_msg=what)
                assert g.info("iteration") == o.info("iteration") == i0 + n, what
            elif op == "set_eps":
                lf = make_lf(rs, D, N, base)
                for e in (g, o):
                    e.set_integrator(lf)
                np.testing.assert_array_equal(g.get_stepsize(), o.get_stepsize(), err_msg=what)
                adaptor_on = False
            elif op == "set_metric" and c["metric"] != "unit":
                m = new_metric()
                for e in (g, o):
                    e.set_metric(m)
                # (the phase point keeps the kinetic energy it was built with — an immutable PhasePoint under `renew`, src/metric.jl:69 — until the
                # next refresh; what the new metric does shows in the calls that follow)
                np.testing.assert_array_equal(g.get_metric(), o.get_metric(), err_msg=what)
            elif op == "set_position":
                th = (0.5 if rs.integers(3) else 3.0) * rs.normal(size=(D, N))
                r = rs.normal(size=(D, N)) if rs.integers(2) else None
                for e in (g, o):
                    e.set_position(th, r)
                zg, zo = g.phasepoint(), o.phasepoint()
                np.testing.assert_allclose(zg.lp.value, zo.lp.value, rtol=1e-10, atol=1e-10, err_msg=what)
                np.testing.assert_allclose(zg.lp.gradient, zo.lp.gradient, rtol=1e-10, atol=1e-10, err_msg=what)
                np.testing.assert_allclose(zg.r, zo.r, rtol=1e-10, atol=1e-10, err_msg=what)
            elif op == "refresh":
                for e in (g, o):
                    e.refresh()
                np.testing.assert_allclose(g.phasepoint().r, o.phasepoint().r, rtol=1e-10, atol=1e-10, err_msg=what)
            elif op == "step":
                n = int(rs.integers(1, 9)) * (1 if rs.integers(2) else -1)
                # conditioning of THIS integration, measured on the oracle: the same steps from a start point moved by a relative 1e-13.  A
                # chain whose end point moves by more than 1e-9 amplifies rounding by > 1e4 (a ballistic pass through log τ ≪ 0 of the
                # hierarchical target after an earlier step threw it out of the typical set does 1e16): not comparable at 1e-8, on any two engines
                z0 = o.phasepoint()
                o2 = A.Engine(h, N, dtype=dtype, rng=1, lib=oracle)
                try:
                    o2.set_integrator(lf)
                    if c["metric"] != "unit":
                        Mo = np.asarray(o.get_metric())
                        o2.set_metric(A.renew(h.metric, Mo.reshape((D, D), order="F") if c["metric"] == "dense" else Mo))
                    o2.lib.check(o2.lib.dll.ahmc_set_stepsize(o2._ctx, A.capi.as_ptr(o.get_stepsize()), N), o2._ctx)
                    o2.set_position(z0.theta * (1 + 1e-13), z0.r)
                    o2.step(n)
                    th2 = o2.theta()
                finally:
                    o2.close()
                for e in (g, o):
                    e.step(n)
                zg, zo = g.phasepoint(), o.phasepoint()
                with np.errstate(invalid="ignore"):
                    stable = (np.abs(th2 - zo.theta) <= 1e-9 * np.maximum(1.0, np.abs(zo.theta))).all(axis=0)
                fin = np.isfinite(zo.lp.value) & np.isfinite(zo.lk.value) & stable
                np.testing.assert_allclose(zg.theta[:, fin], zo.theta[:, fin], rtol=1e-8, atol=1e-8, err_msg=what)
                np.testing.assert_allclose(zg.r[:, fin], zo.r[:, fin], rtol=1e-8, atol=1e-8, err_msg=what)
            elif op == "find_eps":
                PU.reset_margin(o)
                eg, eo = g.find_good_stepsize(), o.find_good_stepsize()
                PU.check_equal_or_near_tie(eg, eo, PU.decision_margin(o), dtype, what)
                adaptor_on = False
            elif op == "seed":
                seed = int(rs.integers(1, 1 << 16))
                it = int(rs.integers(0, 1000))
                for e in (g, o):
                    e.seed(A.PhiloxRNG(seed), iteration=it)
                o._seed_now = seed
                adaptor_on = False    # (the adaptor's iteration count no longer matches)
            elif op == "adaptor":
                ad = A.StepSizeAdaptor(0.8, lf)
                for e in (g, o):
                    e.seed(A.PhiloxRNG(o._seed_now), iteration=0)      # the sample loop counts from 1: a fresh adaptor starts a fresh loop
                    e.adaptor_init(ad)
                adaptor_on = True
            elif op == "checkpoint":
                st = g.get_state()
                g.set_state(st)
                zg, zo = g.phasepoint(), o.phasepoint()
                np.testing.assert_allclose(zg.theta, zo.theta, rtol=1e-8, atol=1e-8, err_msg=what)
            log.append(op)
            if op in ("run_adapt",):
                adaptor_on = False     # (the next adapting run would have to continue at i_first = n + 1: restart instead)
            sync_from_oracle(g, o)
            PU.reset_margin(o)
    finally:
        g.close()
        o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(N_SEQ))
def test_random_call_sequence(hip, oracle, i):
    run_sequence(draw_sequence(i), hip, oracle)
