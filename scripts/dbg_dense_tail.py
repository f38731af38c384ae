"""debug: after Stan (WelfordCov) adaptation on the dense engine, are max-depth trees real?  Replays one transition
from the HIP engine's state on the oracle (same Philox streams) and compares n_steps chain by chain."""
import os, sys
sys.path.insert(0, ".")
import numpy as np, ahmc_amd as A
from ahmc_amd import _capi
from oracle.build_oracle import build
hip = A.load_hip_library(); oracle = _capi.CLib(build())
D, N = int(os.environ.get("D", 64)), int(os.environ.get("N", 4096))
idx = np.arange(D); Sigma = 0.9 ** np.abs(idx[:, None] - idx[None, :])
P = np.asfortranarray(np.linalg.inv(Sigma))
def mk(lib):
    h = A.Hamiltonian(A.DenseEuclideanMetric(np.eye(D)), A.DenseGaussian(P))
    e = A.Engine(h, N, rng=A.PhiloxRNG(77), lib=lib)
    return e
g = mk(hip)
lf = A.Leapfrog(np.full(N, 0.05)); g.set_integrator(lf)
g.set_position(np.asfortranarray(np.random.default_rng(4).random((D, N))))
g.find_good_stepsize()
g.adaptor_init(A.StanHMCAdaptor(A.MassMatrixAdaptor(A.DenseEuclideanMetric(np.eye(D))), A.StepSizeAdaptor(0.8, lf)))
k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=10)))
n_ad = int(os.environ.get("ADAPT", 200))
g.run(k, n_ad, n_ad)
for rep in range(6):
    th = g.phasepoint().theta.copy(); eps = g.get_stepsize().copy(); M = g.get_metric().copy()
    it = g.info("iteration")
    g.transition(k)
    ns = g.stats()["n_steps"]
    bad = np.nonzero(ns >= 255)[0]
    print(f"rep {rep}: iteration {it}, n_steps median {np.median(ns)}, max {ns.max()}, chains >=255: {bad[:10]}", flush=True)
    if len(bad):
        o = mk(oracle)
        o.set_metric(A.DenseEuclideanMetric(M))
        o.set_integrator(A.Leapfrog(eps))
        o.set_position(th)
        o.seed(A.PhiloxRNG(77), iteration=it)
        o.transition(k)
        no = o.stats()["n_steps"]
        print("   oracle n_steps at those chains:", no[bad[:10]], " hip:", ns[bad[:10]], " agree overall:", np.mean(no == ns))
        sg, so = g.stats(), o.stats()
        print("   hip  H err:", sg["hamiltonian_energy_error"][bad[:5]], "acc", sg["acceptance_rate"][bad[:5]])
        print("   orac H err:", so["hamiltonian_energy_error"][bad[:5]], "acc", so["acceptance_rate"][bad[:5]])
        o.close()
        break
