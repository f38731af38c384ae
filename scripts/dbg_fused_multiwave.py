"""Where does the fused (in-kernel adapt!) warm-up of a multi-wave chain leave the stepwise path?  Three drivers of the same
iterations on the HIP engine — (A) run(k, i, n_adapts, i_first=i) one iteration per call (MODE 3 / 4, batch of 1),
(C) transition + adapt (MODE 0 / 1 + the host-launched adapt kernels) — compared after EVERY iteration, both restarted from
(C)'s state whenever they differ, so every difference is reported at its origin.   usage: dbg_fused_multiwave.py D target N"""
import sys

import numpy as np

sys.path.insert(0, ".")
import ahmc_amd as A  # noqa: E402

D, target, N = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
n_adapts, n = 60, 66
import os
hip = A.CLib(os.environ['AHMC_DBG_LIB']) if os.environ.get('AHMC_DBG_LIB') else A.load_hip_library()


def setup():
    metric = A.DiagEuclideanMetric(np.ones((D, N), order="F"))
    tgt = {"hier": A.HierGaussian, "funnel": A.Funnel, "iso": A.IsoGaussian}[target](D)
    h = A.Hamiltonian(metric, tgt)
    lf = A.Leapfrog(np.full(N, 0.1))
    e = A.Engine(h, N, dtype=np.float64, rng=A.PhiloxRNG(0x5EED0005), lib=hip)
    e.set_integrator(lf)
    e.set_position(np.asfortranarray(np.random.default_rng(0x5EED0005).random((D, N))))
    k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=10, delta_max=1000.0)))
    ad = A.StanHMCAdaptor(A.MassMatrixAdaptor(metric), A.StepSizeAdaptor(0.8, lf), init_buffer=9, term_buffer=6, window_size=15)
    e.find_good_stepsize()
    e.adaptor_init(ad)
    return e, k


a, k = setup()
c, _ = setup()
print("geometry", a.info("group_lanes"), a.info("elems_per_lane"))
for i in range(1, n + 1):
    a.run(k, i, n_adapts, i_first=i)
    c.transition(k)
    c.adapt(i, n_adapts)
    sa, sc = a.get_state(), c.get_state()
    sta, stc = a.stats(), c.stats()
    bad_th = ~(sa["theta"] == sc["theta"]).all(axis=0)
    bad_eps = sa["stepsize"] != sc["stepsize"]
    bad_m = ~(sa["metric"] == sc["metric"]).all(axis=0)
    bad_alpha = sta["acceptance_rate"] != stc["acceptance_rate"]
    bad_n = sta["n_steps"] != stc["n_steps"]
    bad_w = np.zeros(N, bool) if sa["welford"] is None else ~(sa["welford"] == sc["welford"]).all(axis=(0, 2))
    bad_da = np.zeros(N, bool) if sa["da"] is None else ~(sa["da"] == sc["da"]).all(axis=0)
    if bad_th.any() or bad_eps.any() or bad_m.any() or bad_alpha.any() or bad_w.any() or bad_da.any():
        j = np.flatnonzero(bad_alpha | bad_th | bad_eps | bad_m | bad_w | bad_da)
        print(f"iter {i}: theta {bad_th.sum()} eps {bad_eps.sum()} metric {bad_m.sum()} alpha {bad_alpha.sum()} n_steps {bad_n.sum()} welford {bad_w.sum()} da {bad_da.sum()}; chains {j[:8]}")
        for q in j[:3]:
            print(f"   chain {q}: alpha {sta['acceptance_rate'][q]!r} vs {stc['acceptance_rate'][q]!r}  n {sta['n_steps'][q]} vs {stc['n_steps'][q]}  "
                  f"eps {sa['stepsize'][q]!r} vs {sc['stepsize'][q]!r}  dH {sta['hamiltonian_energy_error'][q]:.3g} maxdH {stc['max_hamiltonian_energy_error'][q]:.3g}  "
                  f"max|dtheta| {np.abs(sa['theta'][:, q] - sc['theta'][:, q]).max():.3g}  max|dM| {np.abs(sa['metric'][:, q] - sc['metric'][:, q]).max():.3g}")
            if sa["da"] is not None:
                print("      da fused", sa["da"][:, q], "\n      da step ", sc["da"][:, q])
        a.set_state(sc)
    else:
        print(f"iter {i}: identical (mean n_steps {stc['n_steps'].mean():.1f}, max |maxdH| {np.abs(stc['max_hamiltonian_energy_error']).max():.3g})")
