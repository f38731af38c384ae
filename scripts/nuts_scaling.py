"""time per NUTS transition vs leaves per transition (fixed step size, no adaptation): t = a + b * leaves"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ahmc_amd as A
D, N = int(os.environ.get("D", 128)), int(os.environ.get("N", 65536))
lib = A.load_hip_library()
rows = []
for eps in (0.8, 0.4, 0.2, 0.1, 0.05):
    h = A.Hamiltonian(A.DiagEuclideanMetric(np.ones((D, N), order="F")), A.IsoGaussian(D))
    lf = A.Leapfrog(np.full(N, eps))
    k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn()))
    e = A.Engine(h, N, rng=1, lib=lib)
    e.set_integrator(lf); e.set_position(np.random.default_rng(0).normal(size=(D, N)))
    e.run(k, 10); e.sync()
    t = time.perf_counter(); e.run(k, 20); e.sync(); dt = (time.perf_counter() - t) / 20
    acc = e.accum(False); leaves = acc["total_n_steps"] / (20 * N)
    rows.append((eps, leaves, dt * 1e3)); e.close()
    print(f"eps {eps}: {leaves:7.2f} leaves/transition, {dt*1e3:7.3f} ms/transition, {leaves*N/dt:.3e} leapfrog/s")
x = np.array([r[1] for r in rows]); y = np.array([r[2] for r in rows])
b, a = np.polyfit(x, y, 1)
print(f"fit: {a:.3f} ms fixed + {b*1e3:.2f} us per leaf-of-all-chains  -> asymptotic {N/b/1e-3:.3e} leapfrog/s")
