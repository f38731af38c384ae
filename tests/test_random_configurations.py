"""Randomised configurations: HIP engine (through the C ABI) against the CPU oracle on combinations no hand-written test names.

The hand-written parity tests walk the configuration space along its axes (every geometry with one kernel, every kernel on one
geometry …).  This file draws whole configurations — element type × dimension × chains × target × metric × integrator × trajectory
sampler × termination criterion × momentum refreshment — from a seeded generator, so that the rare product of two features (a strict
U-turn criterion on a multi-wave chain with a tempered integrator and partial refreshment, a one-chain dense metric in Float32 …) meets the
oracle too.  The draw is deterministic (seed = case index): a failure names its case and reproduces.

The bar is the one of tests/test_gpu_parity.py: every discrete statistic identical on every chain unless the oracle took one of that
chain's decisions within the margin bound of a tie (tests/parity_util.py), continuous results to the tolerances of that file on the agreeing
chains.  A configuration the reference rejects must be rejected by both engines with the same error class.
"""
import os

import numpy as np
import pytest

import ahmc_amd as A
import parity_util as PU
from test_gpu_parity import RTOL, compare_transition_stats, make_target, realign, _spd

N_CASES = int(os.environ.get("AHMC_RANDOM_CASES", "96"))   # (a one-off hunt: AHMC_RANDOM_CASES=1000 pytest tests/test_random_configurations.py -m gpu)
DIMS = [1, 2, 3, 7, 16, 31, 33, 64, 65, 100, 129, 255, 256, 300, 513, 700, 1025, 2048]
CHAINS = [1, 2, 5, 63, 64, 65, 130, 257]


def draw_case(i):
    """configuration number i — everything a transition depends on"""
    rs = np.random.default_rng(1000 + i)
    c = {"i": i, "dtype": (np.float64, np.float64, np.float32)[rs.integers(3)]}
    c["D"] = D = int(rs.choice(DIMS))
    c["N"] = N = int(rs.choice(CHAINS if D <= 300 else CHAINS[:6]))
    targets = ["iso", "diag"] + (["funnel"] if D >= 2 else []) + (["hier"] if D >= 3 else []) + (["dense"] if D <= 129 else [])
    c["target"] = str(rs.choice(targets))
    metrics = ["unit", "diag_shared", "diag_chain"] + (["dense"] if D <= 129 else [])
    c["metric"] = str(rs.choice(metrics))
    # step size: small enough that trees grow, large enough that some turn early; per chain or one scalar
    base = (0.35 if c["target"] != "funnel" else 0.2) * D ** -0.25
    c["eps_per_chain"] = bool(rs.integers(2))
    c["eps"] = base * (0.5 + rs.random(N)) if c["eps_per_chain"] else float(base * (0.5 + rs.random()))
    c["integrator"] = str(rs.choice(["leapfrog", "leapfrog", "jittered", "tempered"]))
    c["jitter"] = float(rs.choice([0.1, 0.5]))
    c["alpha"] = float(rs.choice([1.02, 1.1]))
    c["nuts"] = bool(rs.integers(4))   # three in four dynamic
    if c["nuts"]:
        c["TS"] = str(rs.choice(["multinomial", "slice"]))
        c["TC"] = str(rs.choice(["generalised", "generalised", "classic", "strict"]))
        c["max_depth"] = int(rs.integers(1, 7 if D <= 300 else 6))
        c["delta_max"] = float(rs.choice([1000.0, 1000.0, 5.0]))
    else:
        c["TS"] = str(rs.choice(["endpoint", "multinomial"]))
        c["static"] = str(rs.choice(["nsteps", "nsteps", "time"]))
        c["L"] = int(rs.integers(1, 12))
        c["lam"] = float(base * rs.uniform(0.5, 6.0))
    c["refresh"] = float(rs.choice([0.0, 0.0, 0.3, 0.9]))
    c["n_transitions"] = 3
    c["seed"] = int(rs.integers(1, 1 << 30))
    return c


def describe(c):
    keys = ("dtype", "D", "N", "target", "metric", "integrator", "nuts", "TS", "TC", "max_depth", "static", "L", "refresh", "eps_per_chain")
    return f"case {c['i']}: " + " ".join(f"{k}={c[k].__name__ if k == 'dtype' else c[k]}" for k in keys if k in c)


def build(c, rng):
    D, N = c["D"], c["N"]
    if c["metric"] == "dense":
        metric = A.DenseEuclideanMetric(_spd(D, rng))
    elif c["metric"] == "unit":
        metric = A.UnitEuclideanMetric((D, N))
    elif c["metric"] == "diag_shared":
        metric = A.DiagEuclideanMetric(0.5 + rng.random(D))
    else:
        metric = A.DiagEuclideanMetric(np.asfortranarray(0.5 + rng.random((D, N))))
    target = A.DenseGaussian(_spd(D, rng, 3.0)) if c["target"] == "dense" else make_target(c["target"], D, rng)
    h = A.Hamiltonian(metric, target)
    lf = {"leapfrog": lambda: A.Leapfrog(c["eps"]), "jittered": lambda: A.JitteredLeapfrog(c["eps"], c["jitter"]),
          "tempered": lambda: A.TemperedLeapfrog(c["eps"], c["alpha"])}[c["integrator"]]()
    TS = {"endpoint": A.EndPointTS, "multinomial": A.MultinomialTS, "slice": A.SliceTS}[c["TS"]]
    if c["nuts"]:
        TC = {"generalised": A.GeneralisedNoUTurn, "classic": A.ClassicNoUTurn, "strict": A.StrictGeneralisedNoUTurn}[c["TC"]]
        term = TC(max_depth=c["max_depth"], delta_max=c["delta_max"])
    else:
        term = A.FixedNSteps(c["L"]) if c["static"] == "nsteps" else A.FixedIntegrationTime(c["lam"])
    refreshment = A.PartialMomentumRefreshment(c["refresh"]) if c["refresh"] else A.FullMomentumRefreshment()
    return h, lf, A.HMCKernel(refreshment, A.Trajectory(TS, lf, term))


def refused(c):
    """FixedIntegrationTime with a VECTOR of step sizes (src/trajectory.jl:241-243 takes one nominal ϵ); a vector of one chain's is that ϵ"""
    return (not c["nuts"]) and c.get("static") == "time" and c["eps_per_chain"] and c["N"] > 1


def advance(e, kernel):
    """one transition: Engine.transition, or — PartialMomentumRefreshment lives in the sample loop's kernel configuration — a
    one-iteration sample call (src/sampler.jl:159-248 with n_samples = 1)"""
    if kernel.refreshment.alpha and getattr(e, "_advance_by_run", False):
        e.run(kernel, 1, 0)
    else:
        e.transition(kernel)      # (Engine.transition hands a partial refreshment to one iteration of the sample loop itself)


def run_case(c, hip, oracle):
    rng = np.random.default_rng(c["seed"])
    dtype, D, N = c["dtype"], c["D"], c["N"]
    what = describe(c)
    h, lf, kernel = build(c, rng)
    th0 = 0.5 * rng.normal(size=(D, N))
    engines, errors = [], []
    for lib in (hip, oracle):
        try:
            e = A.Engine(h, N, dtype=dtype, rng=c["seed"] & 0xFFFF, lib=lib)
            engines.append(e)
            e.set_integrator(lf)
            e.set_position(th0)
            e.refresh()          # (partial refreshment mixes with the momentum the point holds: give it one)
            advance(e, kernel)
            errors.append(None)
        except A.AHMCError as ex:
            errors.append(type(ex))
    try:
        # the same configuration is valid on both sides, or refused by both with the same class of error
        assert errors[0] == errors[1], f"{what}: HIP engine {errors[0]}, oracle {errors[1]}"
        if errors[0] is not None:
            return "refused:" + errors[0].__name__
        g, o = engines
        rt = RTOL[dtype] * 100
        for it in range(c["n_transitions"]):
            if it:
                for e in (g, o):
                    advance(e, kernel)
            sg, so = g.stats(), o.stats()
            clear = PU.decision_margin(o, reset=False) >= PU.MARGIN_BOUND[np.dtype(dtype)]
            # Float32 on a trajectory that left the stable region (energy error beyond 20 on either side — on its way to the Δ_max test):
            # rounding is amplified exponentially along it, WHERE it crosses Δ_max is not a property any margin bounds.  Those chains are
            # held to "both sides saw the instability"; everywhere else, and in Float64 throughout, the margin rule applies
            sel = None
            if dtype == np.float32:
                wild = np.maximum(np.abs(sg["max_hamiltonian_energy_error"]), np.abs(so["max_hamiltonian_energy_error"])) > 20
                wild |= ~np.isfinite(sg["max_hamiltonian_energy_error"]) | ~np.isfinite(so["max_hamiltonian_energy_error"])
                both = (np.abs(sg["max_hamiltonian_energy_error"]) > 5) & (np.abs(so["max_hamiltonian_energy_error"]) > 5)
                both |= ~np.isfinite(sg["max_hamiltonian_energy_error"]) & ~np.isfinite(so["max_hamiltonian_energy_error"])
                assert (both | ~wild).all(), (what, "an unstable trajectory on one side only", np.flatnonzero(wild & ~both)[:8])
                sel = ~wild
            same = compare_transition_stats(sg, so, dtype, o, what, sel=sel)
            # (a near-tie in the CHOICE of the candidate leaves every statistic alone and moves θ: the continuous results are held to
            # the tolerance on the chains none of whose decisions was near a tie)
            same = same & clear
            zg, zo = g.phasepoint(), o.phasepoint()
            tol = 1e-8 if dtype == np.float64 else 2e-2   # (f32: trees of up to 63 single-precision leapfrogs)
            np.testing.assert_allclose(zg.theta[:, same], zo.theta[:, same], rtol=tol, atol=tol, err_msg=what + " theta")
            np.testing.assert_allclose(zg.lp.value[same], zo.lp.value[same], rtol=tol * 10, atol=tol * 10 * max(1.0, D / 16), err_msg=what + " lp")
            if not c["nuts"]:
                L = c["L"] if c["static"] == "nsteps" else max(1, int(np.floor(c["lam"] / float(np.ravel(c["eps"])[0]))))
                assert (sg["n_steps"] == L).all(), (what, L, sg["n_steps"][:8])
            realign(g, o, same)
        return "ran"
    finally:
        for e in engines:
            e.close()


FIRST = int(os.environ.get("AHMC_RANDOM_FIRST", "0"))    # (a hunt continues where the last one stopped)


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(FIRST, N_CASES))
def test_random_configuration(hip, oracle, i):
    c = draw_case(i)
    if refused(c):
        # FixedIntegrationTime needs ONE nominal step size (reference quirk Q6): both sides must refuse the vector
        assert run_case(c, hip, oracle) == "refused:ArgumentError", describe(c)
        return
    assert run_case(c, hip, oracle) == "ran", describe(c)


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(0, N_CASES, 2))
def test_random_configuration_bulk_equals_stepwise(hip, i):
    """the same configurations through ONE `ahmc_sample_from` call of 5 iterations against 5 calls of one: the launch schedule (batched
    NUTS launches, prefetched normals, the dense engine's epochs) must not move a single bit of any chain"""
    c = draw_case(i)
    if refused(c):
        pytest.skip("refused configuration (covered by test_random_configuration)")
    rng = np.random.default_rng(c["seed"])
    h, lf, kernel = build(c, rng)
    th0 = 0.5 * rng.normal(size=(c["D"], c["N"]))
    out = []
    for bulk in (True, False):
        e = A.Engine(h, c["N"], dtype=c["dtype"], rng=c["seed"] & 0xFFFF, lib=hip)
        try:
            e.set_integrator(lf)
            e.set_position(th0)
            e.refresh()
            if bulk:
                e.run(kernel, 5, 0)
            else:
                for _ in range(5):
                    e.run(kernel, 1, 0)
            z, st = e.phasepoint(), e.stats()
            out.append((z.theta.copy(), z.r.copy(), st["n_steps"].copy(), st["hamiltonian_energy"].copy()))
        finally:
            e.close()
    for a, b, name in zip(out[0], out[1], ("theta", "r", "n_steps", "hamiltonian_energy")):
        np.testing.assert_array_equal(a, b, err_msg=describe(c) + " " + name)


def draw_adaptation(i):
    rs = np.random.default_rng(50_000 + i)
    c = {"i": i, "dtype": (np.float64, np.float64, np.float32)[rs.integers(3)]}
    c["D"] = D = int(rs.choice([1, 2, 5, 16, 33, 64, 100, 257, 700]))
    c["N"] = int(rs.choice([1, 2, 17, 64, 130]))
    c["metric"] = str(rs.choice(["unit", "diag_chain", "diag_chain", "diag_shared"] + (["dense"] if D <= 100 else [])))
    kinds = ["stepsize", "stan", "naive", "massmatrix"] + (["nutpie"] if c["metric"] == "diag_chain" else [])
    if c["metric"] == "diag_shared":
        kinds = ["stepsize", "pooled"]
    c["kind"] = str(rs.choice(kinds))
    c["n_adapts"] = int(rs.integers(12, 220))
    c["buffers"] = (int(rs.integers(1, 90)), int(rs.integers(1, 60)), int(rs.integers(2, 40)))
    c["delta"] = float(rs.choice([0.65, 0.8, 0.95]))
    c["seed"] = int(rs.integers(1, 1 << 30))
    return c


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(max(24, N_CASES // 4)))
def test_random_adaptation(hip, oracle, i):
    """adapt!(…) on IDENTICAL inputs (θ, ∇ℓπ, α drawn at random per iteration — the dual-averaging loop is not contractive, see
    test_adaptation) for random adaptor kinds, metrics, sizes, Stan window parameters and n_adapts — including windows that do not
    fit n_adapts (src/adaptation/stan_adaptor.jl:13-50: the 15 % / 75 % / 10 % fallback): ϵ and M⁻¹ agree at every checkpoint and
    after finalize!"""
    c = draw_adaptation(i)
    rng = np.random.default_rng(c["seed"])
    D, N, dtype, n_adapts = c["D"], c["N"], c["dtype"], c["n_adapts"]
    metric = {"unit": lambda: A.UnitEuclideanMetric((D, N)), "diag_chain": lambda: A.DiagEuclideanMetric((D, N)),
              "diag_shared": lambda: A.DiagEuclideanMetric((D,)), "dense": lambda: A.DenseEuclideanMetric((D,))}[c["metric"]]()
    h = A.Hamiltonian(metric, A.IsoGaussian(D))
    lf = A.Leapfrog(np.full(N, 0.1)) if c["metric"] != "dense" else A.Leapfrog(0.1)
    ssa = A.StepSizeAdaptor(c["delta"], lf)
    pc = {"nutpie": A.NutpieVar, "pooled": A.PooledVar}.get(c["kind"], A.MassMatrixAdaptor)(metric)
    ib, tb, ws = c["buffers"]
    ad = {"stepsize": ssa, "massmatrix": pc, "naive": A.NaiveHMCAdaptor(pc, ssa)}.get(c["kind"], A.StanHMCAdaptor(pc, ssa, ib, tb, ws))
    what = f"adaptation case {i}: " + " ".join(f"{k}={getattr(v, '__name__', v)}" for k, v in c.items() if k not in ("i", "seed"))
    engines, errors = [], []
    for lib in (hip, oracle):
        try:
            e = A.Engine(h, N, dtype=dtype, rng=3, lib=lib)
            engines.append(e)
            e.set_integrator(lf)
            e.set_position(np.zeros((D, N)))
            e.adaptor_init(ad)
            errors.append(None)
        except A.AHMCError as ex:
            errors.append(type(ex))
    try:
        assert errors[0] == errors[1], f"{what}: HIP engine {errors[0]}, oracle {errors[1]}"
        if errors[0] is not None:
            return
        g, o = engines
        rt = 1e-9 if dtype == np.float64 else 2e-3
        scale = 0.5 + 2 * rng.random((D, 1))
        checks = {1, 2, n_adapts // 3, n_adapts // 2, n_adapts - 1, n_adapts, n_adapts + 3}
        for it in range(1, n_adapts + 4):
            th, gr, al = scale * rng.normal(size=(D, N)), rng.normal(size=(D, N)) / scale, np.clip(c["delta"] + 0.15 * rng.standard_normal(N), 0.0, 1.0)   # (around δ: ϵ neither collapses nor explodes)
            for e in (g, o):
                e.adapt(it, n_adapts, theta=th, alpha=al, grad=gr if c["kind"] == "nutpie" else None)
            if it in checks:
                np.testing.assert_allclose(g.get_stepsize(), o.get_stepsize(), rtol=rt, atol=1e3 * float(np.finfo(dtype).tiny), err_msg=f"{what}: ϵ at {it}")
                if c["metric"] != "unit":
                    Mo = o.get_metric()
                    np.testing.assert_allclose(g.get_metric(), Mo, rtol=rt * 10, atol=rt * 10 * float(np.abs(Mo).max()), err_msg=f"{what}: M⁻¹ at {it}")
    finally:
        for e in engines:
            e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(1, N_CASES, 4))
def test_random_configuration_checkpoint_resume(hip, i):
    """HMCState (src/abstractmcmc.jl:11-27) on random configurations WITH an adaptor running: iterations 1 … 4, `get_state`, 5 … 8 —
    against a fresh engine that is given the state and runs 5 … 8: bit for bit, whatever kernel, integrator, metric and adaptor"""
    c = draw_case(i)
    if refused(c) or c["refresh"]:
        pytest.skip("refused configuration / partial refreshment carries the momentum, which HMCState does not hold (the reference's neither)")
    rng = np.random.default_rng(c["seed"])
    h, lf, kernel = build(c, rng)
    th0 = 0.5 * rng.normal(size=(c["D"], c["N"]))
    adapt = c["integrator"] == "leapfrog" and c["metric"] != "dense"
    if not c["nuts"] and c["static"] == "time" and c["N"] > 1:
        adapt = False   # (per-chain adapted step sizes + FixedIntegrationTime: refused, Q6)
    def fresh():
        e = A.Engine(h, c["N"], dtype=c["dtype"], rng=c["seed"] & 0xFFFF, lib=hip)
        e.set_integrator(lf)
        e.set_position(th0)
        if adapt:
            pc = A.MassMatrixAdaptor(h.metric)
            e.adaptor_init(A.StanHMCAdaptor(pc, A.StepSizeAdaptor(0.8, lf), 2, 2, 2) if c["metric"] != "unit" else A.StepSizeAdaptor(0.8, lf))
        return e
    a = fresh()
    try:
        a.run(kernel, 4, 8 if adapt else 0)
        st = a.get_state()
        a.run(kernel, 8, 8 if adapt else 0, i_first=5)
        b = fresh()
        try:
            b.set_state(st)
            b.run(kernel, 8, 8 if adapt else 0, i_first=5)
            za, zb = a.phasepoint(), b.phasepoint()
            np.testing.assert_array_equal(za.theta, zb.theta, err_msg=describe(c))
            np.testing.assert_array_equal(a.get_stepsize(), b.get_stepsize(), err_msg=describe(c))
            if c["metric"] != "unit":
                np.testing.assert_array_equal(a.get_metric(), b.get_metric(), err_msg=describe(c))
            np.testing.assert_array_equal(a.stats()["n_steps"], b.stats()["n_steps"], err_msg=describe(c))
        finally:
            b.close()
    finally:
        a.close()


def draw_dense(i):
    rs = np.random.default_rng(90_000 + i)
    c = {"i": i, "dtype": (np.float64, np.float64, np.float32)[rs.integers(3)]}
    c["D"] = int(rs.choice([256, 384, 512, 512] if c["dtype"] == np.float64 else [256, 384, 512, 768, 1024]))
    c["N"] = int(rs.integers(33, 700 if c["D"] <= 512 else 300))
    c["chunk"] = rs.choice([None, None, "3", "5", "17"])
    c["TS"] = str(rs.choice(["multinomial", "multinomial", "slice"]))
    c["TC"] = str(rs.choice(["generalised", "generalised", "classic", "strict"]))
    c["temper"] = rs.choice([None, None, 1.03, 1.08])
    # (the column-tile shapes compiled beside the default, AHMC_EPOCH2_SHAPES: Float64 D = 512 both, Float32 D = 512 the 16-chain one)
    c["nct"] = rs.choice([None, None, "1", "2"] if c["dtype"] == np.float64 else [None, "1"]) if c["D"] == 512 else None
    c["max_depth"] = int(rs.integers(3, 11))
    c["rho"] = float(rs.choice([0.5, 0.9]))
    c["spd"] = bool(rs.integers(2))
    c["n"] = (int(rs.integers(1, 4)), int(rs.integers(1, 4)))   # adapting iterations, draws
    c["seed"] = int(rs.integers(1, 1 << 30))
    return c


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(max(12, N_CASES // 8)))
def test_random_dense_epoch(hip, monkeypatch, i):
    """the chain-complete dense kernels (`k_dense_epoch`, `k_dense_epoch2<T, NW, NCT, WPE, CRIT>`) against the step-synchronous kernels on the
    HIP engine at random shapes: D (4 … 16 waves per workgroup), a chain count that leaves the last workgroup ragged, epochs cut by random chunk
    lengths, sampler × criterion × TemperedLeapfrog, the column-tile shape, max_depth, the target's correlation, identity / full M⁻¹ — Float64: the same
    n_steps on every chain in every transition and the same draws to 1e-9, free-running; Float32 (they add r·v, θ·g, ρ·v in another order): one
    iteration at a time from the same state, ≥ 99 % of the chains with the same n_steps AND within 2e-3 per transition"""
    import torch

    c = draw_dense(i)
    rs = np.random.default_rng(c["seed"])
    D, N, dtype = c["D"], c["N"], c["dtype"]
    what = "dense case " + " ".join(f"{k}={getattr(v, '__name__', v)}" for k, v in c.items() if k != "seed")
    idx = np.arange(D)
    P = np.asfortranarray(np.linalg.inv(c["rho"] ** np.abs(idx[:, None] - idx[None, :])))
    if c["spd"]:
        Q, _ = np.linalg.qr(rs.normal(size=(D, D)))
        Minv = (Q * np.linspace(0.6, 2.0, D)) @ Q.T
        Minv = np.asfortranarray((Minv + Minv.T) / 2)
    else:
        Minv = np.eye(D, order="F")
    th0 = np.asfortranarray(rs.normal(size=(D, N)))
    eps0 = (0.12 if c["rho"] > 0.7 else 0.3) * (0.7 + 0.6 * rs.random(N))
    TS = {"multinomial": A.MultinomialTS, "slice": A.SliceTS}[c["TS"]]
    TC = {"generalised": A.GeneralisedNoUTurn, "classic": A.ClassicNoUTurn, "strict": A.StrictGeneralisedNoUTurn}[c["TC"]]
    n_adapts, n_draws = c["n"]

    def env(engine):
        monkeypatch.setenv("AHMC_DENSE_EPOCH", "1" if engine == "epoch" else "0")
        monkeypatch.setenv("AHMC_DENSE_EPOCH_MIN", "32")
        for var, val in (("AHMC_DENSE_CHUNK", c["chunk"]), ("AHMC_DENSE_EPOCH_NCT", c["nct"]), ("AHMC_DENSE_EPOCH_V", "2" if c["nct"] else None)):
            if val:
                monkeypatch.setenv(var, str(val))
            else:
                monkeypatch.delenv(var, raising=False)

    def make(engine):
        env(engine)
        lf = A.TemperedLeapfrog(eps0, float(c["temper"])) if c["temper"] else A.Leapfrog(eps0)
        k = A.HMCKernel(A.Trajectory(TS, lf, TC(max_depth=c["max_depth"], delta_max=1000.0)))
        g = A.Engine(A.Hamiltonian(A.DenseEuclideanMetric(Minv), A.DenseGaussian(P)), N, dtype=dtype, rng=A.PhiloxRNG(c["seed"] & 0xFFFF), lib=hip)
        g.set_integrator(lf)
        g.set_position(th0)
        g.adaptor_init(A.StepSizeAdaptor(0.8, lf))
        return g, k

    if dtype == np.float64:
        # free-running: the whole loop in one call on either engine
        out = {}
        for engine in ("step", "epoch"):
            g, k = make(engine)
            try:
                draws = torch.empty((n_draws, N, D), dtype=torch.float64, device="cuda")
                g.run(k, n_adapts + n_draws, n_adapts, drop_warmup=True, samples_out=draws.data_ptr())
                g.sync()
                st, acc = g.stats(), g.accum()
                out[engine] = (draws.cpu().numpy(), st["n_steps"].copy(), acc["total_n_steps"], g.get_stepsize().copy(), g.theta().copy(), g.info("dense_epoch_launches"))
            finally:
                g.close()
        a, b = out["step"], out["epoch"]
        assert a[5] == 0, what
        if b[5] == 0:
            pytest.skip("no chain-complete kernel for this shape (the engine keeps the step-synchronous kernels): " + what)
        np.testing.assert_array_equal(a[1], b[1], err_msg=what)
        assert a[2] == b[2], what
        np.testing.assert_allclose(a[3], b[3], rtol=1e-9, err_msg=what)
        # (free-running: up to five transitions of up to 1 023 leapfrogs with dual averaging in between — the two engines add r·v, θ·g, ρ·v in
        # another order; the 600-shape hunt's worst was 3.3e-9 at max_depth 10)
        np.testing.assert_allclose(a[0], b[0], rtol=2e-8, atol=2e-8, err_msg=what)
        np.testing.assert_allclose(a[4], b[4], rtol=2e-8, atol=2e-8, err_msg=what)
        return
    # Float32: ONE iteration at a time from the step engine's state (free-running, single-precision dual averaging between trees of up to 1 023
    # single-precision leapfrogs drifts apart without any decision being wrong): per transition at most 1 chain in 100 may take another number
    # of leapfrogs (a tie in single precision), every other chain ends within 2e-3 and adapts to the same step size
    (gs, k), (ge, _) = make("step"), make("epoch")
    try:
        for it in range(1, n_adapts + n_draws + 1):
            ge.set_state(gs.get_state())
            env("step")
            gs.run(k, it, n_adapts, i_first=it)
            env("epoch")
            ge.run(k, it, n_adapts, i_first=it)
            ns, ne = gs.stats()["n_steps"], ge.stats()["n_steps"]
            # (a tie in single precision changes the tree — another n_steps — or only the candidate the tree hands back: either way the chain
            # ends O(1) away; at most 1 chain in 100 per transition, or two)
            same = (ns == ne) & np.isclose(ge.theta(), gs.theta(), rtol=2e-3, atol=2e-3).all(axis=0)
            assert same.mean() >= 0.99 or (~same).sum() <= 2, (what, it, same.mean())
            np.testing.assert_allclose(ge.get_stepsize()[same], gs.get_stepsize()[same], rtol=1e-4, err_msg=f"{what} iteration {it}")
        assert gs.info("dense_epoch_launches") == 0, what
        if ge.info("dense_epoch_launches") == 0:
            pytest.skip("no chain-complete kernel for this shape (the engine keeps the step-synchronous kernels): " + what)
    finally:
        gs.close()
        ge.close()


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(max(12, N_CASES // 8)))
def test_random_reference_batch_exit(hip, oracle, i):
    """`ahmc_set_ref_compat` (the reference's matrix-mode early exit, src/integrator.jl:252-258 over all columns) at random
    sizes, metrics, step counts and directions, with one to three chains that blow up after a random number of steps: `step(lf, h, z, n)` and
    static EndPointTS transitions on the HIP engine against the oracle's literal form — on and off"""
    rs = np.random.default_rng(70_000 + i)
    D, N = int(rs.choice([1, 3, 16, 65, 200, 600])), int(rs.choice([2, 5, 64, 130, 257]))
    metric = str(rs.choice(["unit", "diag_chain", "diag_shared"]))
    compat = bool(rs.integers(4))            # three in four with the coupling on
    n = int(rs.integers(2, 12)) * (1 if rs.integers(2) else -1)
    bad = rs.choice(N, size=min(N - 1, int(rs.integers(1, 4))), replace=False)
    what = f"batch exit case {i}: D={D} N={N} metric={metric} compat={compat} n={n} bad={bad.tolist()}"
    rng = np.random.default_rng(int(rs.integers(1 << 30)))
    if metric == "unit":
        m = A.UnitEuclideanMetric((D, N))
    elif metric == "diag_shared":
        m = A.DiagEuclideanMetric(0.5 + rng.random(D))
    else:
        m = A.DiagEuclideanMetric(np.asfortranarray(0.5 + rng.random((D, N))))
    h = A.Hamiltonian(m, A.IsoGaussian(D))
    eps = 0.05 + 0.2 * rng.random(N)
    # a chain with ϵ = 1e80 … 1e160 leaves the finite numbers after one to three steps (θ ≈ ϵ·r, then ϵ²·…, the energy overflows)
    eps[bad] = 10.0 ** rng.choice([80, 110, 160], size=bad.size)
    th, r = rng.normal(size=(D, N)), rng.normal(size=(D, N))
    ok = np.ones(N, dtype=bool)
    ok[bad] = False
    engines = []
    try:
        for lib in (hip, oracle):
            engines.append(A.Engine(h, N, dtype=np.float64, rng=4 + i, lib=lib))
        g, o = engines
        for e in (g, o):
            e.set_integrator(A.Leapfrog(eps))
            e.set_ref_compat(compat)
            e.set_position(th, r)
            e.step(n)
        zg, zo = g.phasepoint(), o.phasepoint()
        np.testing.assert_allclose(zg.theta[:, ok], zo.theta[:, ok], rtol=1e-9, atol=1e-9, err_msg=what)
        np.testing.assert_allclose(zg.r[:, ok], zo.r[:, ok], rtol=1e-9, atol=1e-9, err_msg=what)
        np.testing.assert_array_equal(np.isfinite(zg.lp.value), np.isfinite(zo.lp.value), err_msg=what)
        np.testing.assert_array_equal(np.isfinite(zg.lk.value), np.isfinite(zo.lk.value), err_msg=what)
        kern = A.HMCKernel(A.Trajectory(A.EndPointTS, A.Leapfrog(eps), A.FixedNSteps(abs(n))))
        for e in (g, o):
            e.set_position(th)
        for _ in range(2):
            for e in (g, o):
                e.transition(kern)
            same = compare_transition_stats(g.stats(), o.stats(), np.float64, o, what)
            np.testing.assert_allclose(g.phasepoint().theta[:, same & ok], o.phasepoint().theta[:, same & ok], rtol=1e-9, atol=1e-9, err_msg=what)
            realign(g, o, same)
    finally:
        for e in engines:
            e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(2, N_CASES, 4))
def test_random_find_good_stepsize(hip, oracle, i):
    """find_good_stepsize per chain (src/trajectory.jl:768-837) on the random configurations' Hamiltonians and start points, from random initial
    step sizes: the powers of two and the direction of the search follow from decisions alone — identical unless the oracle had a near-tie"""
    c = draw_case(i)
    rng = np.random.default_rng(c["seed"])
    h, lf, kernel = build(c, rng)
    th0 = (0.5 if i % 8 < 6 else 3.0) * rng.normal(size=(c["D"], c["N"]))
    eps_init = float(10.0 ** rng.uniform(-3, 1))
    res, engines = [], []
    try:
        for lib in (hip, oracle):
            e = A.Engine(h, c["N"], dtype=c["dtype"], rng=c["seed"] & 0xFFFF, lib=lib)
            engines.append(e)
            e.set_position(th0)
            if lib is oracle:
                PU.reset_margin(e)
            res.append(e.find_good_stepsize(eps_init))
        PU.check_equal_or_near_tie(res[0], res[1], PU.decision_margin(engines[1]), c["dtype"], "find_good_stepsize " + describe(c) + f" from {eps_init:g}")
        # the point survives the search
        np.testing.assert_array_equal(engines[0].theta(), np.asarray(th0, dtype=c["dtype"]))
    finally:
        for e in engines:
            e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(3, N_CASES, 4))
def test_random_fused_warmup_equals_stepwise(hip, i):
    """the whole `sample` loop with an adaptor (adapt! inside the kernels, batched launches, windows, restarts, finalize!) in ONE call against
    transition + adapt! per iteration on the HIP engine — random configuration, random adaptor (kind, estimator, Stan buffers), n_adapts 8 … 70,
    a run that ends inside the warm-up or some draws after it: bit for bit"""
    c = draw_case(i)
    if refused(c) or c["metric"] == "dense" or (not c["nuts"] and c["static"] == "time" and c["N"] > 1):
        pytest.skip("refused configuration / DenseEuclideanMetric adapts through WelfordCov on its own path (test_dense_covariance_adaptation) / "
                    "per-chain adapted step sizes + FixedIntegrationTime (Q6)")
    rng = np.random.default_rng(c["seed"] + 1)
    h, lf, kernel = build(c, np.random.default_rng(c["seed"]))
    kinds = ["stepsize", "stan", "naive", "massmatrix"] + (["nutpie"] if c["metric"] == "diag_chain" else [])
    if c["metric"] == "diag_shared":
        kinds = ["stepsize", "pooled"]     # (ONE shared M⁻¹: the pooled estimator adapts it, the per-chain ones have nothing to write to)
    kind = str(rng.choice(kinds))
    ssa = A.StepSizeAdaptor(float(rng.choice([0.65, 0.8, 0.9])), lf)
    pc = {"nutpie": A.NutpieVar, "pooled": A.PooledVar}.get(kind, A.MassMatrixAdaptor)(h.metric)
    buffers = (int(rng.integers(1, 30)), int(rng.integers(1, 20)), int(rng.integers(2, 15)))
    ad = {"stepsize": ssa, "massmatrix": pc, "naive": A.NaiveHMCAdaptor(pc, ssa)}.get(kind, A.StanHMCAdaptor(pc, ssa, *buffers))
    n_adapts = int(rng.integers(8, 70))
    n = max(1, n_adapts + int(rng.integers(-5, 7)))
    what = describe(c) + f" adaptor={kind} buffers={buffers} n_adapts={n_adapts} n={n}"
    th0 = 0.5 * rng.normal(size=(c["D"], c["N"]))
    a = A.Engine(h, c["N"], dtype=c["dtype"], rng=c["seed"] & 0xFFFF, lib=hip)
    b = A.Engine(h, c["N"], dtype=c["dtype"], rng=c["seed"] & 0xFFFF, lib=hip)
    try:
        for e in (a, b):
            e.set_integrator(lf)
            e.set_position(th0)
            e.refresh()
            e.adaptor_init(ad)
        a.run(kernel, n, n_adapts)
        for it in range(1, n + 1):
            b.run(kernel, it, n_adapts, i_first=it)
        np.testing.assert_array_equal(a.theta(), b.theta(), err_msg=what)
        np.testing.assert_array_equal(a.get_stepsize(), b.get_stepsize(), err_msg=what)
        if c["metric"] != "unit":
            np.testing.assert_array_equal(a.get_metric(), b.get_metric(), err_msg=what)
        sa, sb = a.stats(), b.stats()
        for f in ("n_steps", "acceptance_rate", "hamiltonian_energy", "tree_depth", "is_accept"):
            np.testing.assert_array_equal(sa[f], sb[f], err_msg=what + " " + f)
    finally:
        a.close()
        b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("D,N,target,dtype", [(3, 100003, "funnel", np.float64), (32, 65537, "funnel", np.float64), (128, 4097, "iso", np.float64),
                                              (16, 70001, "hier", np.float32), (1, 131071, "iso", np.float64), (200, 1031, "diag", np.float32)])
def test_awkward_chain_counts_fused_equals_stepwise(hip, D, N, target, dtype):
    """chain counts that are multiples of nothing (a ragged last wave, a ragged last workgroup, more chains than one 16-bit dispatch-order key
    distinguishes, 2¹⁷ − 1 chains of one dimension): the whole loop — Stan warm-up inside the kernels, work-sorted launches of several transitions,
    the prefetched normals, the log-domain redo pass of the funnel's overflowing chains — in ONE call against one iteration per call, bit for bit"""
    rng = np.random.default_rng(D * 1000 + N)
    metric = A.DiagEuclideanMetric((D, N))
    h = A.Hamiltonian(metric, make_target(target, D, rng))
    lf = A.Leapfrog(np.full(N, 0.2))
    kernel = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=6)))
    ad = A.StanHMCAdaptor(A.MassMatrixAdaptor(metric), A.StepSizeAdaptor(0.8, lf), 4, 3, 5)
    th0 = rng.normal(size=(D, N))
    n_adapts, n = 22, 34
    a = A.Engine(h, N, dtype=dtype, rng=5, lib=hip)
    b = A.Engine(h, N, dtype=dtype, rng=5, lib=hip)
    try:
        for e in (a, b):
            e.set_integrator(lf)
            e.set_position(th0)
            e.adaptor_init(ad)
        a.run(kernel, n, n_adapts)
        for it in range(1, n + 1):
            b.run(kernel, it, n_adapts, i_first=it)
        np.testing.assert_array_equal(a.theta(), b.theta())
        np.testing.assert_array_equal(a.get_stepsize(), b.get_stepsize())
        np.testing.assert_array_equal(a.get_metric(), b.get_metric())
        sa, sb = a.stats(), b.stats()
        for f in ("n_steps", "acceptance_rate", "hamiltonian_energy", "tree_depth", "numerical_error"):
            np.testing.assert_array_equal(sa[f], sb[f], err_msg=f)
        assert sa["n_steps"].max() > 7 and np.isfinite(a.theta()).all()
        if target == "funnel":
            assert a.info("nuts_launches") + a.info("nuts_warm_launches") > 0
    finally:
        a.close()
        b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(max(12, N_CASES // 8)))
def test_random_diagnostics(hip, i):
    """output side at random sizes: the device-side running sums and reductions of a bulk run — moments, Σ n_steps,
    divergences, EBFMI (src/diagnosis.jl:1-3), ESS — against numpy on what the SAME chains produce one iteration per call (bulk == stepwise bit for
    bit, so every chain counts: no "most chains" threshold)"""
    import torch

    c = draw_case(7 * i + 3)
    if refused(c) or c["D"] > 300:
        pytest.skip("refused configuration / kept small: K draws of (D, N) travel to the host")
    rng = np.random.default_rng(c["seed"])
    h, lf, kernel = build(c, rng)
    D, N, dtype = c["D"], c["N"], c["dtype"]
    th0 = 0.5 * rng.normal(size=(D, N))
    K = int(rng.integers(8, 40))
    tdt = torch.float32 if dtype == np.float32 else torch.float64

    def fresh():
        e = A.Engine(h, N, dtype=dtype, rng=c["seed"] & 0xFFFF, lib=hip)
        e.set_integrator(lf)
        e.set_position(th0)
        e.refresh()
        return e

    a, b = fresh(), fresh()
    try:
        draws_d = torch.empty((K, N, D), dtype=tdt, device="cuda")
        a.run(kernel, K, 0, samples_out=draws_d.data_ptr())
        a.sync()
        draws = draws_d.cpu().numpy().astype(np.float64)
        E, TH, steps, ndiv = [], [], 0, 0
        for it in range(1, K + 1):
            b.run(kernel, it, 0, i_first=it)
            st = b.stats()
            E.append(st["hamiltonian_energy"].astype(np.float64))
            TH.append(b.theta().T.astype(np.float64))
            steps += int(st["n_steps"].sum())
            ndiv += int(st["numerical_error"].sum())
        what = describe(c) + f" K={K}"
        np.testing.assert_array_equal(draws, np.array(TH), err_msg=what)
        acc, g = a.accum(), a.gather_moments()
        assert acc["total_n_steps"] == steps and acc["n_divergent"] == ndiv and acc["n_transitions"] == K, what
        tol = 1e-9 if dtype == np.float64 else 2e-4
        mean = draws.mean(axis=(0, 1))
        np.testing.assert_allclose(g["mean"], mean, rtol=tol, atol=tol, err_msg=what)
        np.testing.assert_allclose(g["var"], (draws ** 2).mean(axis=(0, 1)) - mean ** 2, rtol=tol * 100, atol=tol * 100 * max(1.0, float((draws ** 2).mean())), err_msg=what)
        assert g["n_draws"] == K * N and g["total_n_steps"] == steps
        E = np.array(E)
        ok = np.var(E, axis=0, ddof=1) > 1e-12 * np.maximum(1.0, np.abs(E).max(axis=0)) ** 2     # (a chain whose energy never moved: 0/0)
        np.testing.assert_allclose(a.ebfmi()[ok], A.EBFMI(E)[ok], rtol=1e-6 if dtype == np.float64 else 5e-2, err_msg=what)
        got = a.ess(draws_d.data_ptr(), K)
        want = A.diagnostics.ess(draws_d.cpu().numpy(), axis=0).T
        fin = np.isfinite(want)
        np.testing.assert_allclose(got[fin], want[fin], rtol=1e-6 if dtype == np.float64 else 2e-2, err_msg=what)
    finally:
        a.close()
        b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(1, N_CASES, 4))
def test_random_sharding_invariance(hip, i):
    """multi-GPU by construction: a chain's stream depends on its GLOBAL index alone — the random configurations run as ONE engine of N
    chains and as two or three engines over contiguous blocks cut at random (`PhiloxRNG(seed, chain_offset=…)`, each with its slice of the step
    sizes, the metric and the start points, each adapting on its own): warm-up + draws, bit for bit the same chains"""
    c = draw_case(i)
    if refused(c) or c["N"] < 3 or c["metric"] == "dense" or (not c["nuts"] and c["static"] == "time"):
        pytest.skip("refused configuration / fewer than three chains / one shared dense M⁻¹ adapts from ALL chains (pooled: test_v3_state_gather) / Q6")
    rng = np.random.default_rng(c["seed"])
    D, N, dtype = c["D"], c["N"], c["dtype"]
    h, lf, kernel = build(c, rng)
    th0 = 0.5 * rng.normal(size=(D, N))
    cuts = sorted(set(int(x) for x in rng.integers(1, N, size=int(rng.integers(1, 3)))))
    blocks = list(zip([0] + cuts, cuts + [N]))
    seed = c["seed"] & 0xFFFF
    n_adapts, n = 6, 10
    adapt = c["metric"] != "diag_shared"     # (a shared (D,) M⁻¹ is adapted from all chains of an engine: per block it would differ)

    def sub(a, lo, hi):
        a = np.asarray(a)
        return a if a.ndim == 0 or a.shape[-1] != N else a[..., lo:hi]

    def engine(lo, hi):
        m = h.metric
        if isinstance(m, A.DiagEuclideanMetric) and np.ndim(m.Minv) == 2:
            m = A.DiagEuclideanMetric(np.asfortranarray(m.Minv[:, lo:hi]))
        elif isinstance(m, A.UnitEuclideanMetric):
            m = A.UnitEuclideanMetric((D, hi - lo))
        hh = A.Hamiltonian(m, h.target)
        l2 = type(lf)(sub(lf.eps, lo, hi)) if isinstance(lf, A.Leapfrog) else type(lf)(sub(lf.eps, lo, hi), lf.param)
        k2 = A.HMCKernel(kernel.refreshment, A.Trajectory(kernel.tau.TS, l2, kernel.tau.termination_criterion))
        e = A.Engine(hh, hi - lo, dtype=dtype, rng=A.PhiloxRNG(seed, chain_offset=lo), lib=hip)
        e.set_integrator(l2)
        e.set_position(th0[:, lo:hi])
        e.refresh()
        if adapt:
            e.adaptor_init(A.StanHMCAdaptor(A.MassMatrixAdaptor(m), A.StepSizeAdaptor(0.8, l2), 2, 1, 2) if c["metric"] == "diag_chain" else A.StepSizeAdaptor(0.8, l2))
        e.run(k2, n, n_adapts if adapt else 0)
        out = (e.theta().copy(), e.get_stepsize().copy(), e.stats()["n_steps"].copy())
        e.close()
        return out

    whole = engine(0, N)
    for lo, hi in blocks:
        part = engine(lo, hi)
        what = describe(c) + f" block {lo}:{hi} of {blocks}"
        np.testing.assert_array_equal(part[0], whole[0][:, lo:hi], err_msg=what)
        np.testing.assert_array_equal(part[1], whole[1][lo:hi], err_msg=what)
        np.testing.assert_array_equal(part[2], whole[2][lo:hi], err_msg=what)


def test_the_draw_covers_the_space():
    """(no GPU work) the generator reaches every value of every axis, and the rare products this file exists for"""
    cases = [draw_case(i) for i in range(96)]
    for key, want in (("dtype", 2), ("target", 5), ("metric", 4), ("integrator", 3), ("TS", 3), ("TC", 3), ("refresh", 3)):
        got = {str(c[key]) for c in cases if key in c}
        assert len(got) >= want, (key, got)
    assert sum(c["D"] > 512 for c in cases) >= 8                                               # multi-wave chains
    assert sum(c["nuts"] and c["TC"] != "generalised" and c["integrator"] != "leapfrog" for c in cases) >= 4
    assert sum(c["metric"] == "dense" and c["dtype"] == np.float32 for c in cases) >= 2
    assert sum(c["N"] == 1 for c in cases) >= 4
