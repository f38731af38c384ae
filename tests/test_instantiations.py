"""Every compiled instantiation of the NUTS kernel — thread geometry (G, E) × log-density family × its three drivers: the fused
warm-up (MODE 3 / 4, adapt! inside the kernel), the batched draws (MODE 0 / 1) and the general kernel (MODE 2: SliceTS +
StrictGeneralisedNoUTurn) — against the per-iteration path on the HIP engine, BIT FOR BIT.

Why this exists (round 4): each instantiation is a separate piece of machine code, and two classes of defect are per instantiation
and invisible to a test of one geometry: (1) the register allocator parking a spill under a narrowed exec mask (a memory fault of
k_nuts<double,8,2,3,1> on the MI355X; `isa_check.py` scans for the pattern at build time, this is the run-time net); (2) the
optimiser contracting a*b+c differently in two instantiations of one template, so that a warm-up run in one launch drifted from the
same warm-up run one iteration per call (fixed by -ffp-contract=on, advancedhmc.jl_amd/build.py).  The per-iteration path is
held to the oracle by tests/test_gpu_parity.py; equality with it carries that parity to every launch shape.
Reference semantics: src/sampler.jl:72-90,182-228, src/trajectory.jl:626-742."""
import numpy as np
import pytest

import ahmc_amd as A

pytestmark = pytest.mark.gpu

DS = [3, 5, 10, 24, 32, 50, 100, 128, 200, 300, 600, 1500, 2048, 4096]      # one D per compiled geometry (and two for some)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("tname", ["iso", "diag", "funnel", "hier"])
def test_every_instantiation_bulk_equals_stepwise(hip, tname, dtype):
    rng = np.random.default_rng(5)
    geoms = set()
    for D in DS:
        N = 96 if D <= 512 else 16
        tgt = {"iso": lambda: A.IsoGaussian(D), "diag": lambda: A.DiagGaussian(rng.normal(size=D), 0.5 + rng.random(D)),
               "funnel": lambda: A.Funnel(D), "hier": lambda: A.HierGaussian(D)}[tname]()
        metric = A.DiagEuclideanMetric(np.asfortranarray(0.5 + rng.random((D, N))))
        h = A.Hamiltonian(metric, tgt)
        lf = A.Leapfrog(np.full(N, 0.25 * D ** -0.25))
        th0 = 0.5 * rng.normal(size=(D, N))
        for mode in ("warm", "draw", "general"):
            if mode == "general":
                k = A.HMCKernel(A.Trajectory(A.SliceTS, lf, A.StrictGeneralisedNoUTurn(max_depth=6)))
            else:
                k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=8)))
            res = []
            for which in ("bulk", "step"):
                e = A.Engine(h, N, dtype=dtype, rng=A.PhiloxRNG(9), lib=hip)
                e.set_integrator(lf)
                e.set_position(th0)
                n, na = (14, 14) if mode == "warm" else (8, 0)
                if mode == "warm":   # Stan windows 3 / 2 / 4 of 14: window 4 … 12, split (and dual-averaging restart) at 12, finalize! at 14
                    e.adaptor_init(A.StanHMCAdaptor(A.MassMatrixAdaptor(metric), A.StepSizeAdaptor(0.8, lf), init_buffer=3, term_buffer=2, window_size=4))
                if which == "bulk":
                    e.run(k, n, na)
                else:
                    for i in range(1, n + 1):
                        e.transition(k)
                        if mode == "warm":
                            e.adapt(i, na)
                res.append((e.theta().copy(), e.stats(), e.get_stepsize().copy(), e.get_metric().copy()))
                geoms.add((e.info("group_lanes"), e.info("elems_per_lane")))
                e.close()
            (ta, sa, ea, ma), (tb, sb, eb, mb) = res
            what = f"{tname} D={D} {mode} {np.dtype(dtype).name}"
            assert np.isfinite(ta).all() and (sa["n_steps"] >= 1).all(), what
            np.testing.assert_array_equal(ta, tb, err_msg=what + ": θ")
            np.testing.assert_array_equal(ea, eb, err_msg=what + ": ϵ")
            np.testing.assert_array_equal(ma, mb, err_msg=what + ": M⁻¹")
            for f in ("n_steps", "acceptance_rate", "tree_depth", "hamiltonian_energy", "numerical_error"):
                np.testing.assert_array_equal(sa[f], sb[f], err_msg=what + ": " + f)
    if hip.backend == "hip:gfx950":
        assert len(geoms) >= 11, geoms     # every row of the geometry table was exercised
