#!/usr/bin/env python
"""Stress of the multi-wave chain kernels (D > 512): every transition kind, several D / targets / seeds, engines created
and destroyed in one process, every result checked against the oracle (discrete statistics AND the candidate's
log-density, which is what a wrongly selected candidate shows up in).  Exit code 1 on the first mismatch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import ahmc_amd as A
import build_oracle
hip = A.load_hip_library(); oracle = A.CLib(build_oracle.build())
bad = 0
cases = [(D, t, s) for s in (1, 2) for D in (520, 600, 1000, 1024, 1100, 1500, 2048, 2500, 3000, 4096) for t in ("iso", "hier", "funnel")]
for D, tname, seed in cases:
    rng = np.random.default_rng(1000 * seed + D)
    N = 20
    tgt = {"iso": A.IsoGaussian, "hier": A.HierGaussian, "funnel": A.Funnel}[tname](D)
    h = A.Hamiltonian(A.DiagEuclideanMetric(np.asfortranarray(0.5 + rng.random((D, N)))), tgt)
    eps = (0.3 if tname != "funnel" else 0.15) * D ** -0.25
    lf = A.Leapfrog(np.full(N, eps) * (0.8 + 0.4 * rng.random(N)))
    g = A.Engine(h, N, rng=seed, lib=hip); o = A.Engine(h, N, rng=seed, lib=oracle)
    th = 0.5 * rng.normal(size=(D, N))
    for e in (g, o):
        e.set_integrator(lf); e.set_position(th)
    kernels = [A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=7))),
               A.HMCKernel(A.Trajectory(A.SliceTS, lf, A.GeneralisedNoUTurn(max_depth=6))),
               A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.FixedNSteps(5)))]
    for ik, k in enumerate(kernels):
        for it in range(3):
            g.transition(k); o.transition(k)
            sg, so = g.stats(), o.stats()
            same = (sg["n_steps"] == so["n_steps"]) & (sg["tree_depth"] == so["tree_depth"]) & (sg["is_accept"] == so["is_accept"])
            ld = np.isclose(sg["log_density"], so["log_density"], rtol=1e-7, atol=1e-7)
            acc = np.isclose(sg["acceptance_rate"], so["acceptance_rate"], rtol=1e-7, atol=1e-9)
            if same.mean() < 0.9 or (same & ~ld).any() or (same & ~acc).any():
                print(f"MISMATCH D={D} {tname} seed={seed} kernel {ik} it {it}: same {same.mean():.2f} logdens ok {ld.mean():.2f} acc ok {acc.mean():.2f}", flush=True)
                bad += 1
            t = o.phasepoint().theta
            g.set_position(t); o.set_position(t)
    # bulk path: fused batches incl. a short warm-up
    for e in (g, o):
        e.adaptor_init(A.StanHMCAdaptor(A.MassMatrixAdaptor(h.metric), A.StepSizeAdaptor(0.8, lf)))
        e.run(kernels[0], 12, 8)
    sg, so = g.stats(), o.stats()
    ok = np.isclose(g.theta(), o.theta(), rtol=1e-6, atol=1e-6).all(axis=0)
    if ok.mean() < 0.7:
        print(f"MISMATCH bulk D={D} {tname} seed={seed}: chains on track {ok.mean():.2f}", flush=True); bad += 1
    g.close(); o.close()
    print(f"ok D={D} {tname} seed={seed}", flush=True)
print("bad =", bad)
sys.exit(1 if bad else 0)
