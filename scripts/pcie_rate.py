"""cfg2 with the draws handed back to a HOST buffer (the reference's `sample` returns host matrices):
the PCIe-inclusive rate DESIGN.md §6 quotes beside bench.py's HBM-resident `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ahmc_amd as A
D, N, K = 128, 65536, 96
lib = A.load_hip_library()
metric = A.DiagEuclideanMetric(np.ones((D, N), order="F"))
h = A.Hamiltonian(metric, A.IsoGaussian(D))
e = A.Engine(h, N, rng=A.PhiloxRNG(0x5EED0002), lib=lib)
lf = A.Leapfrog(np.full(N, 0.1)); e.set_integrator(lf)
theta0 = np.asfortranarray(np.random.default_rng(2).random((D, N)))
t = time.perf_counter(); e.set_position(theta0); e.sync()
print("H2D of the initial state (%d MiB, pageable): %.2f ms" % (theta0.nbytes >> 20, (time.perf_counter() - t) * 1e3))
e.find_good_stepsize()
e.adaptor_init(A.StanHMCAdaptor(A.MassMatrixAdaptor(metric), A.StepSizeAdaptor(0.8, lf)))
k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=10)))
e.run(k, 200, 200); e.sync()
e.run(k, K, 0); e.sync()


def timed(label, out):
    e.reset_accum()
    t = time.perf_counter(); e.run(k, K, 0, samples_out=out); e.sync(); dt = time.perf_counter() - t
    acc = e.accum(moments=False)
    print("%-28s %.3e leapfrog/s, %.2f ms/transition" % (label, acc["total_n_steps"] / dt, dt / K * 1e3))


timed("draws stay in HBM:", None)
dev = torch.empty(K * D * N, dtype=torch.float64, device="cuda")
timed("draws -> device buffer:", dev)
del dev
pinned = torch.empty(K * D * N, dtype=torch.float64).pin_memory()
timed("draws -> pinned host:", pinned)
pageable = np.empty(K * D * N)
pageable[::512] = 0  # touch the pages first
timed("draws -> pageable host:", pageable)
