"""Every (thread geometry, log-density family) instantiation of k_nuts through its three drivers on the HIP engine — the fused warm-up
(MODE 3 / 4), the batched draws (MODE 0 / 1) and the general kernel (MODE 2: SliceTS + StrictGeneralisedNoUTurn) — against the
per-iteration path, bit for bit.  One process per target family (a device fault ends the process: the last line says where).
    python scripts/sweep_inst.py iso|diag|funnel|hier [dtype]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import ahmc_amd as A  # noqa: E402

hip = A.load_hip_library()
tname = sys.argv[1]
dtype = np.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else np.float64
import os
DS = [int(x) for x in os.environ["SWEEP_D"].split(",")] if os.environ.get("SWEEP_D") else [3, 5, 10, 24, 32, 50, 100, 128, 200, 300, 600, 1500, 2048, 4096]
rng = np.random.default_rng(5)
bad = 0
for D in DS:
    if tname == "hier" and D < 3:
        continue
    N = 96 if D <= 512 else 16
    tgt = {"iso": lambda: A.IsoGaussian(D), "diag": lambda: A.DiagGaussian(rng.normal(size=D), 0.5 + rng.random(D)),
           "funnel": lambda: A.Funnel(D), "hier": lambda: A.HierGaussian(D)}[tname]()
    metric = A.DiagEuclideanMetric(np.asfortranarray(0.5 + rng.random((D, N))))
    h = A.Hamiltonian(metric, tgt)
    eps = 0.25 * D ** -0.25
    lf = A.Leapfrog(np.full(N, eps))
    th0 = 0.5 * rng.normal(size=(D, N))
    for mode in ("warm", "draw", "general"):
        if mode == "general":
            k = A.HMCKernel(A.Trajectory(A.SliceTS, lf, A.StrictGeneralisedNoUTurn(max_depth=6)))
        else:
            k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=8)))
        print(f"{tname} D={D} {mode} ...", end=" ", flush=True)
        es = []
        for which in ("bulk", "step"):
            e = A.Engine(h, N, dtype=dtype, rng=A.PhiloxRNG(9), lib=hip)
            e.set_integrator(lf)
            e.set_position(th0)
            n, na = (14, 14) if mode == "warm" else (8, 0)
            if mode == "warm":
                e.adaptor_init(A.StanHMCAdaptor(A.MassMatrixAdaptor(metric), A.StepSizeAdaptor(0.8, lf), init_buffer=3, term_buffer=2, window_size=4))
            if which == "bulk":
                e.run(k, n, na)
            else:
                for i in range(1, n + 1):
                    e.transition(k)
                    if mode == "warm":
                        e.adapt(i, na)
            es.append((e.theta().copy(), e.stats(), e.get_stepsize().copy(), e.get_metric().copy(), (e.info("group_lanes"), e.info("elems_per_lane"))))
            e.close()
        (ta, sa, ea, ma, ga), (tb, sb, eb, mb, _) = es
        ok = np.array_equal(ta, tb) and np.array_equal(ea, eb) and np.array_equal(ma, mb) and all(np.array_equal(sa[f], sb[f]) for f in ("n_steps", "acceptance_rate", "tree_depth"))
        sane = np.isfinite(ta).all() and (sa["n_steps"] >= 1).all()
        bad += 0 if (ok and sane) else 1
        det = ""
        if not ok:
            dth = ~(ta == tb).all(axis=0)
            det = (f" chains: theta {dth.sum()} eps {(ea != eb).sum()} metric {(~(ma == mb).all(axis=0)).sum()} n_steps {(sa['n_steps'] != sb['n_steps']).sum()} "
                   f"alpha {(sa['acceptance_rate'] != sb['acceptance_rate']).sum()} max|dtheta| {np.abs(ta - tb).max():.2e} max rel deps {np.max(np.abs(ea - eb) / np.abs(eb)):.2e}")
        print(ga, "OK" if ok and sane else f"MISMATCH ok={ok} sane={sane}{det}", f"mean n_steps {sa['n_steps'].mean():.1f}", flush=True)
print(f"{tname}: {bad} bad")
