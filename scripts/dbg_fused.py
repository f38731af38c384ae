import sys, numpy as np
sys.path.insert(0, ".")
import ahmc_amd as A
hip = A.load_hip_library()
rng = np.random.default_rng(0)
D, N = 5, 512
sig = 0.5 + 2 * rng.random(D)
for est in (A.MassMatrixAdaptor, A.NutpieVar):
    metric = A.DiagEuclideanMetric((D, N))
    h = A.Hamiltonian(metric, A.DiagGaussian(np.zeros(D), sig))
    lf = A.Leapfrog(np.full(N, 0.1))
    e = A.Engine(h, N, rng=77, lib=hip)
    e.set_integrator(lf); e.set_position(rng.normal(size=(D, N)))
    e.adaptor_init(A.StanHMCAdaptor(est(metric), A.StepSizeAdaptor(0.8, lf)))
    k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn()))
    e.run(k, 300, 300)
    print(est.__name__, "metric median", np.median(e.get_metric(), axis=1), "truth", sig ** 2, "eps", np.median(e.get_stepsize()))
