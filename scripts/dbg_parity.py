"""debug: per-iteration HIP-vs-oracle match fraction over many transitions"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ahmc_amd as A
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import build_oracle
hip = A.load_hip_library(); oracle = A.CLib(build_oracle.build())
rng = np.random.default_rng(0)
adapt = len(sys.argv) > 1 and sys.argv[1] == "adapt"
D, N, n_adapts = 5, 256, 150
metric = A.DiagEuclideanMetric((D, N))
h = A.Hamiltonian(metric, A.DiagGaussian(np.zeros(D), np.array([0.5, 1.0, 2.0, 1.0, 0.3])))
lf = A.Leapfrog(np.full(N, 0.1))
k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=6)))
ad = A.StanHMCAdaptor(A.MassMatrixAdaptor(metric), A.StepSizeAdaptor(0.8, lf))
es = []
th = rng.normal(size=(D, N))
for lib in (hip, oracle):
    e = A.Engine(h, N, rng=17, lib=lib); e.set_integrator(lf); e.set_position(th)
    if adapt: e.adaptor_init(ad)
    es.append(e)
g, o = es
alive = np.ones(N, bool)
for i in range(1, 60):
    for e in (g, o):
        e.transition(k)
        if adapt: e.adapt(i, n_adapts)
    sg, so = g.stats(), o.stats()
    same = (sg["n_steps"] == so["n_steps"]) & (sg["tree_depth"] == so["tree_depth"])
    thg, tho = g.theta(), o.theta()
    close = np.all(np.isclose(thg, tho, rtol=1e-7, atol=1e-7), axis=0)
    newbad = alive & ~(same & close)
    if newbad.any():
        c = np.flatnonzero(newbad)[0]
        print(f"it {i}: {newbad.sum()} new mismatches; chain {c}: nsteps {sg['n_steps'][c]} vs {so['n_steps'][c]}, depth {sg['tree_depth'][c]} vs {so['tree_depth'][c]}, "
              f"eps {sg['step_size'][c]:.6g} vs {so['step_size'][c]:.6g}, acc {sg['acceptance_rate'][c]:.6g} vs {so['acceptance_rate'][c]:.6g}, Herr {sg['hamiltonian_energy_error'][c]:.6g} vs {so['hamiltonian_energy_error'][c]:.6g}, "
              f"maxHerr {sg['max_hamiltonian_energy_error'][c]:.6g} vs {so['max_hamiltonian_energy_error'][c]:.6g} numerr {sg['numerical_error'][c]} {so['numerical_error'][c]} thetaclose {close[c]}")
    alive &= same & close
    if i % 10 == 0: print(f"it {i}: alive {alive.mean():.4f}, mean depth {sg['tree_depth'].mean():.2f}")
