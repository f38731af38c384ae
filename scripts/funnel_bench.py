"""cfg3: D=32 Neal's funnel, DiagEuclideanMetric, NUTS(0.8, max_depth 10) + StanHMCAdaptor, 65 536 chains, f64"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ahmc_amd as A
D, N = 32, 65536
lib = A.load_hip_library()
metric = A.DiagEuclideanMetric(np.ones((D, N), order="F"))
h = A.Hamiltonian(metric, A.Funnel(D))
e = A.Engine(h, N, rng=A.PhiloxRNG(0x5EED0003), lib=lib)
lf = A.Leapfrog(np.full(N, 0.1)); e.set_integrator(lf)
e.set_position(np.asfortranarray(np.random.default_rng(3).random((D, N))))
e.find_good_stepsize()
e.adaptor_init(A.StanHMCAdaptor(A.MassMatrixAdaptor(metric), A.StepSizeAdaptor(0.8, lf)))
k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=10)))
t = time.perf_counter(); e.run(k, 200, 200); e.sync(); print("adaptation 200 steps: %.2f s" % (time.perf_counter() - t))
e.run(k, 16, 0); e.sync()
t = time.perf_counter(); e.run(k, 64, 0); e.sync(); dt = time.perf_counter() - t
acc = e.accum()
n = acc["n_transitions"] * N
m = acc["sum_theta"].sum(axis=1) / n; v = acc["sumsq_theta"].sum(axis=1) / n - m * m
print("cfg3 funnel: %.3e leapfrog/s, %.2f ms/transition, %.1f leapfrogs/transition, divergent fraction %.4f" % (
    acc["total_n_steps"] / dt, dt / 64 * 1e3, acc["total_n_steps"] / n, acc["n_divergent"] / n))
print("theta1: mean %.3f var %.3f (truth 0, 9); eps median %.3f" % (m[0], v[0], np.median(e.get_stepsize())))
