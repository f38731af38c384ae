"""cfg5 (one GPU's shard): D=2048 hierarchical Gaussian, DiagEuclideanMetric, NUTS(0.8) + StanHMCAdaptor,
262 144 chains / 8 GPUs = 32 768 chains per GPU, f64.  Multi-wave chains: (G,E) = (256,8), 4 waves per chain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ahmc_amd as A
D = int(os.environ.get("D", 2048)); N = int(os.environ.get("N", 32768))
n_adapt = int(os.environ.get("ADAPT", 100)); n_timed = int(os.environ.get("STEPS", 32))
lib = A.load_hip_library()
metric = A.DiagEuclideanMetric(np.ones((D, N), order="F"))
h = A.Hamiltonian(metric, A.HierGaussian(D))
e = A.Engine(h, N, rng=A.PhiloxRNG(0x5EED0005), lib=lib)
lf = A.Leapfrog(np.full(N, 0.1)); e.set_integrator(lf)
e.set_position(np.asfortranarray(np.random.default_rng(5).random((D, N))))
e.find_good_stepsize()
e.adaptor_init(A.StanHMCAdaptor(A.MassMatrixAdaptor(metric), A.StepSizeAdaptor(0.8, lf)))
k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=10)))
t = time.perf_counter(); e.run(k, n_adapt, n_adapt); e.sync(); print("adaptation %d steps: %.2f s" % (n_adapt, time.perf_counter() - t))
e.run(k, 16, 0); e.sync()
e.reset_accum()
t = time.perf_counter(); e.run(k, n_timed, 0); e.sync(); dt = time.perf_counter() - t
acc = e.accum()
n = acc["n_transitions"] * N
print("cfg5 hier D=%d N=%d: %.3e leapfrog/s, %.2f ms/transition, %.1f leapfrogs/transition, divergent fraction %.4f, eps median %.4f" % (
    D, N, acc["total_n_steps"] / dt, dt / n_timed * 1e3, acc["total_n_steps"] / n, acc["n_divergent"] / n, np.median(e.get_stepsize())))
print("algorithmic HBM floor: %.1f GB/s-equivalent at 4*D*8 B per leapfrog" % (acc["total_n_steps"] / dt * 4 * D * 8 / 1e9))
