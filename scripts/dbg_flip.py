#!/usr/bin/env python
"""Round 6: where does a chain of test_cfg2_pipeline_against_oracle leave the oracle's track although none of its decisions was a near-tie?
Replays chunk lo..hi of that test one iteration at a time on both engines (no re-sync inside the chunk) and prints, for every chain that is
off track at the end, the first iteration at which anything differs, the oracle's margin of that iteration and the state differences before it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ahmc_amd as A  # noqa: E402
import build_oracle  # noqa: E402
import parity_util as PU  # noqa: E402


def main(N=384, lo_target=11, hi_target=20, chunk=10):
    hip = A.load_hip_library()
    oracle = A.CLib(build_oracle.build())
    D, n_adapts, n_total = 128, 110, 120
    metric = A.DiagEuclideanMetric(np.ones((D, N), order="F"))
    h = A.Hamiltonian(metric, A.IsoGaussian(D))
    lf = A.Leapfrog(np.full(N, 0.1))
    k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=10, delta_max=1000.0)))
    th0 = np.asfortranarray(np.random.default_rng(2).random((D, N)))
    g = A.Engine(h, N, dtype=np.float64, rng=0x5EED0002, lib=hip)
    o = A.Engine(h, N, dtype=np.float64, rng=0x5EED0002, lib=oracle)
    for e in (g, o):
        e.set_integrator(lf)
        e.set_position(th0)
    eo = o.find_good_stepsize()
    g.find_good_stepsize()
    for e in (g, o):
        e.set_integrator(A.Leapfrog(eo))
        e.adaptor_init(A.StanHMCAdaptor(A.MassMatrixAdaptor(metric), A.StepSizeAdaptor(0.8, lf)))
    for lo in range(1, n_total + 1, chunk):
        hi = min(lo + chunk - 1, n_total)
        g.set_state(o.get_state())
        if lo != lo_target:
            o.run(k, hi, n_adapts, i_first=lo)
            continue
        PU.reset_margin(o)
        first = np.full(N, -1)
        rec = {}
        for i in range(lo, hi + 1):
            eps_g, eps_o = g.get_stepsize().copy(), o.get_stepsize().copy()
            thg, tho = g.theta().copy(), o.theta().copy()
            for e in (g, o):
                e.run(k, i, n_adapts, i_first=i)
            sg, so = g.stats(), o.stats()
            m = PU.decision_margin(o)
            diff = (sg["n_steps"] != so["n_steps"]) | (sg["tree_depth"] != so["tree_depth"]) | ~np.isclose(g.theta(), o.theta(), rtol=1e-7, atol=1e-7).all(axis=0)
            for c in np.flatnonzero(diff & (first < 0)):
                first[c] = i
                rec[c] = dict(iteration=i, margin=m[c], n_steps=(int(sg["n_steps"][c]), int(so["n_steps"][c])), depth=(int(sg["tree_depth"][c]), int(so["tree_depth"][c])),
                              dH_max=(float(sg["max_hamiltonian_energy_error"][c]), float(so["max_hamiltonian_energy_error"][c])),
                              H=(float(sg["hamiltonian_energy"][c]), float(so["hamiltonian_energy"][c])),
                              acc=(float(sg["acceptance_rate"][c]), float(so["acceptance_rate"][c])),
                              eps_before=(float(eps_g[c]), float(eps_o[c]), float(abs(eps_g[c] - eps_o[c]) / eps_o[c])),
                              theta_before_maxrel=float(np.max(np.abs(thg[:, c] - tho[:, c]) / (1e-300 + np.abs(tho[:, c])))),
                              numerr=(int(sg["numerical_error"][c]), int(so["numerical_error"][c])))
            print(f"iteration {i}: off track {int(diff.sum())}, max |eps_g/eps_o - 1| before it {np.max(np.abs(eps_g / eps_o - 1)):.3e}, "
                  f"max |dH|max oracle {np.abs(so['max_hamiltonian_energy_error']).max():.3g}, min margin {m.min():.3e}", flush=True)
        for c, r in rec.items():
            print("chain", c, r)
        break


if __name__ == "__main__":
    main()
