"""GPU parity of the PIPELINES bench.py times on BASELINE configs[2] and configs[4] — the fused warm-up (adapt! inside
k_nuts, MODE 3 / 4) and the batched launches — on the geometries those configs run on:

  cfg5 (D = 2 048 hierarchical Gaussian): a chain spans 4 wavefronts of one workgroup, k_nuts<double,256,8,·,3>;
       (600, hier) is the 2-wave geometry (128,8);
  cfg3 (D = 32 Neal's funnel): 4 chains per wavefront in lockstep, k_nuts<double,16,2,·,2>, divergent paths.

Reference semantics: the `sample` loop `src/sampler.jl:182-228` with `adapt!` `src/sampler.jl:72-90`,
`src/adaptation/stan_adaptor.jl:137-159`, the dynamic transition `src/trajectory.jl:677-742`.

Two kinds of check, as for cfg2 (tests/test_gpu_parity.py::test_cfg2_pipeline_against_oracle):
  (i)  HIP == HIP bit for bit: `run(k, n, n_adapts)` in batches against `transition` + `adapt` one iteration at a time;
  (ii) HIP vs the CPU oracle in chunks of 10 iterations, both sides restarted from the oracle's complete state
       (θ, ϵ, M⁻¹, DAState, Welford, window counters) at every chunk — dual averaging amplifies a last-bit difference
       by ≈ 2× per iteration, so a free-running comparison is meaningful over short horizons only.
And every dispatch-schedule switch the library ships must leave the chains untouched (chains are independent: the
order in which workgroups start is not allowed to show in any result).
"""
import numpy as np
import pytest

import ahmc_amd as A
import parity_util as PU

pytestmark = pytest.mark.gpu


def _target(name, D):
    return {"hier": A.HierGaussian, "funnel": A.Funnel, "iso": A.IsoGaussian}[name](D)


def _setup(lib, D, N, target, seed, n_adapts_windows=None):
    metric = A.DiagEuclideanMetric(np.ones((D, N), order="F"))
    h = A.Hamiltonian(metric, _target(target, D))
    lf = A.Leapfrog(np.full(N, 0.1))
    e = A.Engine(h, N, dtype=np.float64, rng=A.PhiloxRNG(seed), lib=lib)
    e.set_integrator(lf)
    th0 = np.asfortranarray(np.random.default_rng(seed).random((D, N)))   # θ0 ~ U(0,1) as bench.py / test/sampler-vec.jl:7
    e.set_position(th0)
    k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=10, delta_max=1000.0)))
    # the reference's window schedule does not shrink with n_adapts (stan_adaptor.jl:13-50: 75 / 50 / 25 need n_adapts >= 150 for
    # one metric update), so the short runs here use StanHMCAdaptor(…; init_buffer = 9, term_buffer = 6, window_size = 15): with
    # n_adapts = 60 the window is 10…54 with splits at 24 and 54 — two metric updates + dual-averaging restarts, then finalize!
    ad = A.StanHMCAdaptor(A.MassMatrixAdaptor(metric), A.StepSizeAdaptor(0.8, lf), init_buffer=9, term_buffer=6, window_size=15)
    return e, k, ad


# (D, target, N, geometry the engine must pick)
PIPE = [(2048, "hier", 64, (256, 8)), (600, "hier", 96, (128, 8)), (32, "funnel", 512, (16, 2))]


@pytest.mark.parametrize("D,target,N,geom", PIPE)
def test_fused_warmup_equals_stepwise_on_bench_geometries(hip, D, target, N, geom):
    """(i) batched run == stepwise, bit for bit, through the whole Stan schedule (n_adapts = 60: init buffer 9, window
    splits at 24 and 54 — metric update + dual-averaging restart —, term buffer 6, finalize!) and 12 draws in
    one launch — the launches cfg5 / cfg3 spend their time in."""
    n_adapts, n = 60, 72
    a, k, ad = _setup(hip, D, N, target, 0x5EED0005)
    b, _, _ = _setup(hip, D, N, target, 0x5EED0005)
    real = hip.backend == "hip:gfx950"      # (False only in a dry run of this test code on the CPU checker, conftest.py)
    assert not real or (a.info("group_lanes"), a.info("elems_per_lane")) == geom
    ea, eb = a.find_good_stepsize(), b.find_good_stepsize()
    np.testing.assert_array_equal(ea, eb)
    for e in (a, b):
        e.adaptor_init(ad)
    l0 = a.info("nuts_launches") + a.info("nuts_warm_launches")
    out = np.zeros((D, N, n - n_adapts), order="F")
    a.run(k, n, n_adapts, drop_warmup=True, samples_out=out)
    a.sync()
    launches = a.info("nuts_launches") + a.info("nuts_warm_launches") - l0
    import os
    if real and not os.environ.get("AHMC_NUTS_LOGW"):   # (with the log-domain kernel alone there is no MODE 3 launch to count)
        assert a.info("nuts_warm_launches") >= 1 and launches <= 6, launches     # batches, not one launch per transition
    total, div = 0, 0
    for i in range(1, n + 1):
        b.transition(k)
        b.adapt(i, n_adapts)
        if i > n_adapts:
            np.testing.assert_array_equal(out[:, :, i - n_adapts - 1], b.theta(), err_msg=f"draw {i - n_adapts}")
            st = b.stats(["n_steps", "numerical_error"])
            total += int(st["n_steps"].sum())
            div += int(st["numerical_error"].sum())
    np.testing.assert_array_equal(a.theta(), b.theta())
    np.testing.assert_array_equal(a.get_stepsize(), b.get_stepsize())
    ma, mb = a.get_metric(), b.get_metric()
    np.testing.assert_array_equal(ma, mb)
    assert np.abs(ma - 1).max() > 0.05, "the window end must have updated M⁻¹"
    sa, sb = a.stats(), b.stats()
    for f in ("n_steps", "acceptance_rate", "hamiltonian_energy", "tree_depth", "numerical_error", "step_size"):
        np.testing.assert_array_equal(sa[f], sb[f], err_msg=f)
    acc = a.accum()
    assert acc["total_n_steps"] == total and acc["n_transitions"] == n - n_adapts and acc["n_divergent"] == div
    np.testing.assert_allclose(acc["sum_theta"], out.sum(axis=2), rtol=1e-12, atol=1e-12)
    a.close(); b.close()


@pytest.mark.parametrize("D,target,N,geom", PIPE)
def test_bench_pipeline_against_oracle_in_chunks(hip, oracle, D, target, N, geom):
    """(ii) the same pipeline against the oracle, chunk by chunk from the oracle's state: every iteration of the Stan
    schedule (init buffer, window, metric update + dual-averaging restart at 24 and at 54, term buffer, finalize! at 60) and the
    first draws, on multi-wave chains (cfg5) and on 4 chains per wave (cfg3)."""
    n_adapts, n_total = 60, 70
    g, k, ad = _setup(hip, D, N, target, 0x5EED0005)
    o, _, _ = _setup(oracle, D, N, target, 0x5EED0005)
    assert hip.backend != "hip:gfx950" or (g.info("group_lanes"), g.info("elems_per_lane")) == geom
    eg, eo = g.find_good_stepsize(), o.find_good_stepsize()
    PU.check_equal_or_near_tie(eg, eo, PU.decision_margin(o), np.float64, f"pipeline {target} D={D} find_good_stepsize")
    g.set_integrator(A.Leapfrog(eo))
    for e in (g, o):
        e.adaptor_init(ad)
    n_div, max_depth, n_stable_checked, n_short_div = 0, 0, 0, 0
    # Chunks: ONE iteration each through the warm-up, five for the draws.  From θ0 ~ U(0,1) the first iterations integrate
    # with step sizes that are still far too large (and again after each dual-averaging restart) — energy errors of 10³ … 10⁶⁹, hundreds of leapfrogs per tree: such a
    # trajectory is numerically unstable (that is what the divergence test detects), a last-bit difference between the two
    # sides grows by orders of magnitude INSIDE one transition, and where a chain is declared divergent can legitimately differ.
    # There the bar is applied to the chains whose transition was stable on the oracle (|ΔH|_max < 2); every chain is compared
    # again from the oracle's state at the next iteration, so all of the schedule's early part is still covered one step at a time.
    # The bar (round 6): a comparable chain may be off the oracle's track ONLY if the oracle took one of its decisions of the chunk
    # within 1e-9 (relative) of a tie — rounds 4–5 allowed 2 chains or 3 % per chunk without asking why.
    bounds, lo = [], 1
    while lo <= n_total:
        hi = min(lo if lo <= n_adapts else lo + 4, n_total)
        bounds.append((lo, hi))
        lo = hi + 1
    for lo, hi in bounds:
        g.set_state(o.get_state())
        for e in (g, o):
            e.run(k, hi, n_adapts, i_first=lo)
        sg, so = g.get_state(), o.get_state()
        assert sg["adaptor"] == so["adaptor"]
        on = np.isclose(sg["theta"], so["theta"], rtol=1e-7, atol=1e-7).all(axis=0)
        stg, sto = g.stats(), o.stats()
        # the chunk's LAST transition belongs to the track: same tree
        on &= (stg["n_steps"] == sto["n_steps"]) & (stg["tree_depth"] == sto["tree_depth"])
        margin = PU.decision_margin(o)
        what = f"pipeline {target} D={D} iterations {lo}..{hi}"
        if lo == hi:
            # Round 5: no iteration passes unchecked.  A transition is comparable when it was stable on the oracle (|ΔH|_max < 2) OR SHORT
            # (≤ 8 leapfrogs: the first iterations and the one after each dual-averaging restart diverge on the first or third leaf —
            # nothing has had the time to amplify a rounding, and a divergent first leaf must come out as n_steps = 1, θ unchanged, on
            # both sides).  By the oracle every iteration of the three pipelines has ≥ 15 such chains (iterations 1–3 and 56: all of them
            # short and divergent; iteration 4: 15 / 29 / 509); fewer than 8 is a failure, not a pass.
            sto1 = o.stats()
            stable = (np.abs(sto1["max_hamiltonian_energy_error"]) < 2.0) | (sto1["n_steps"] <= 8)
            n_stable_checked += int(stable.sum())
            assert stable.sum() >= 8, (lo, int(stable.sum()))
            PU.check_flips(on, margin, np.float64, what, sel=stable)
            n_short_div += int(((sto1["n_steps"] <= 8) & (sto1["numerical_error"] != 0)).sum())
        else:
            PU.check_flips(on, margin, np.float64, what)
        # the chunk's LAST transition, on the chains still on track: every statistic
        last_same = on
        # (the statistics of a trajectory that blew up — ΔH of 10³ … 10⁶⁹ in the first iterations from θ0 ~ U(0,1) — carry the
        # amplified rounding of both sides: they are compared on the transitions that stayed within |ΔH| < 50; the decisions,
        # the positions and the whole adaptation state are compared for every chain)
        both = on & last_same & (np.abs(sto["max_hamiltonian_energy_error"]) < 50) & (np.abs(stg["max_hamiltonian_energy_error"]) < 50)
        # (H − H0 cancels: at D = 2 048 the energies are 10³ … 10⁴, so the energy ERRORS are held to the energies' own tolerance —
        # 1e-7 of |H|, what `on` holds the positions to after up to ten dual-averaged iterations — not to their own magnitude)
        Habs = 1.0 + np.abs(sto["hamiltonian_energy"][both])
        np.testing.assert_allclose(stg["hamiltonian_energy"][both], sto["hamiltonian_energy"][both], rtol=1e-6, err_msg=f"H at iteration {hi}")
        np.testing.assert_allclose(stg["acceptance_rate"][both], sto["acceptance_rate"][both], rtol=1e-4, atol=1e-6, err_msg=f"α at iteration {hi}")
        for f in ("hamiltonian_energy_error", "max_hamiltonian_energy_error"):
            assert (np.abs(stg[f][both] - sto[f][both]) <= 1e-7 * Habs + 1e-6 * np.abs(sto[f][both])).all(), f"{f} at iteration {hi}"
        np.testing.assert_array_equal(stg["numerical_error"][both], sto["numerical_error"][both])
        n_div += int(sto["numerical_error"].sum())
        max_depth = max(max_depth, int(sto["tree_depth"].max()))
        # (a chunk of ten dual-averaged iterations doubles a last-bit difference ten times, and in the first iterations from
        # θ0 ~ U(0,1) at D = 2 048 the energy errors are 10³ … 10⁶⁹: the adaptation state is held to 1e-5 — a defect shows as O(1))
        np.testing.assert_allclose(sg["stepsize"][on], so["stepsize"][on], rtol=1e-5, err_msg=f"ϵ after iterations {lo}..{hi}")
        np.testing.assert_allclose(sg["metric"][:, on], so["metric"][:, on], rtol=1e-5, err_msg=f"M⁻¹ after {lo}..{hi}")
        if sg["da"] is not None:
            np.testing.assert_allclose(sg["da"][:, on], so["da"][:, on], rtol=1e-5, atol=1e-7, err_msg=f"DAState after {lo}..{hi}")
        if sg["welford"] is not None:
            w_g, w_o = sg["welford"][:, on, :], so["welford"][:, on, :]
            np.testing.assert_allclose(w_g, w_o, rtol=1e-5, atol=1e-6 * (1.0 + np.abs(w_o).max()), err_msg=f"Welford after {lo}..{hi}")
        if lo <= 24 <= hi:
            assert not np.allclose(so["metric"], 1.0), "the window split at iteration 24 must have updated M⁻¹"
    st = o.get_state()
    assert st["adaptor"]["adapting"] == 0 and st["adaptor"]["iteration"] == n_total
    assert max_depth >= 4, max_depth            # real trees: merges on several pending levels
    assert n_stable_checked >= 40 * N, n_stable_checked   # the one-iteration chunks compared most chains at most iterations
    assert n_short_div >= N, n_short_div                 # … among them the short divergent transitions of the first iterations
    if target == "funnel":
        assert n_div > 0, "the funnel's warm-up must contain divergent transitions"
    g.close(); o.close()


SCHEDULES = [
    {},                                                  # the default: every launch ordered by the work of the one before, length by measurement
    {"AHMC_NUTS_SCHED": "0"},                            # one launch length for all (round 3)
    {"AHMC_NUTS_SCHED": "0", "AHMC_NUTS_ORDER_REFRESH": "0"},   # … and the order from the run's totals (round 3's default)
    {"AHMC_NUTS_NO_ORDER": "1"},                         # chains in index order
    {"AHMC_NUTS_DRAW_BATCH": "7"},                       # ragged short launches in the sampling phase
    {"AHMC_NUTS_DRAW_BATCH": "2"},                       # the shortest re-sorted launch
    {"AHMC_NUTS_FIRST_BATCH": "5"},                      # a short first launch, then by measured work
    {"AHMC_NUTS_ORDER_REFRESH": "0", "AHMC_NUTS_DRAW_BATCH": "9", "AHMC_NUTS_FIRST_BATCH": "4"},
    {"AHMC_NUTS_BATCH": "6"},                            # warm-up and draws in short launches
    {"AHMC_NORMALS_PREFETCH": "0"},                      # round 6: the next launch's normals made in series again (the default makes them beside the launch)
    {"AHMC_NORMALS_PREFETCH_MAX_MB": "100000", "AHMC_NUTS_DRAW_BATCH": "5"},   # … prefetched for every launch length, ragged launches
]


@pytest.mark.parametrize("D,target,N", [(32, "funnel", 1024), (2048, "hier", 16)])
def test_dispatch_schedules_leave_the_chains_untouched(hip, monkeypatch, D, target, N):
    """Every schedule switch shipped in the library (launch length, dispatch order by step size / by measured work /
    refreshed per launch, short first launch) on a cfg3-shaped and a cfg5-shaped run: warm-up + draws, the draws, the
    statistics, the adapted step sizes and metric are identical to the default schedule's."""
    n_adapts, n = 40, 340          # 300 draws: the default schedule times groups of launches of 32, 16, … before it settles
    ref = None
    names = ("AHMC_NUTS_NO_ORDER", "AHMC_NUTS_ORDER_REFRESH", "AHMC_NUTS_DRAW_BATCH", "AHMC_NUTS_FIRST_BATCH", "AHMC_NUTS_BATCH", "AHMC_NUTS_SCHED",
             "AHMC_NORMALS_PREFETCH", "AHMC_NORMALS_PREFETCH_MAX_MB")
    for env in SCHEDULES:
        for v in names:
            monkeypatch.delenv(v, raising=False)
        for kk, vv in env.items():
            monkeypatch.setenv(kk, vv)
        e, k, ad = _setup(hip, D, N, target, 77)
        e.find_good_stepsize()
        e.adaptor_init(ad)
        out = np.zeros((D, N, n - n_adapts), order="F")
        l0 = e.info("nuts_launches")
        e.run(k, n, n_adapts, drop_warmup=True, samples_out=out)
        e.sync()
        res = (out, e.get_stepsize(), e.get_metric(), e.stats(), e.accum(), e.info("nuts_launches") - l0)
        e.close()
        if ref is None:
            ref = res
            continue
        np.testing.assert_array_equal(res[0], ref[0], err_msg=str(env))
        np.testing.assert_array_equal(res[1], ref[1], err_msg=str(env))
        np.testing.assert_array_equal(res[2], ref[2], err_msg=str(env))
        for f in ("n_steps", "acceptance_rate", "hamiltonian_energy", "tree_depth", "numerical_error"):
            np.testing.assert_array_equal(res[3][f], ref[3][f], err_msg=f"{f} {env}")
        assert res[4]["total_n_steps"] == ref[4]["total_n_steps"] and res[4]["n_divergent"] == ref[4]["n_divergent"]
        np.testing.assert_array_equal(res[4]["sum_theta"], ref[4]["sum_theta"], err_msg=str(env))
        if hip.backend == "hip:gfx950" and ("AHMC_NUTS_DRAW_BATCH" in env or "AHMC_NUTS_BATCH" in env or "AHMC_NUTS_SCHED" in env):
            assert res[5] != ref[5], (env, res[5], ref[5])   # the switch did change the launch plan
