"""k_dense_epoch against the step-synchronous kernels on the HIP engine: the same chains from the same state, a few iterations each;
prints where they part.   usage: dbg_dense_epoch.py [D] [N] [iters]   (AHMC_DENSE_EPOCH_MIN is set to 32 for the epoch run)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import ahmc_amd as A  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2304
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
hip = A.load_hip_library()
idx = np.arange(D)
P = np.asfortranarray(np.linalg.inv(0.9 ** np.abs(idx[:, None] - idx[None, :])))
rs = np.random.default_rng(2024)
Q, _ = np.linalg.qr(rs.normal(size=(D, D)))
Minv = (Q * np.linspace(0.6, 2.0, D)) @ Q.T
Minv = np.asfortranarray((Minv + Minv.T) / 2)
th0 = np.asfortranarray(rs.normal(size=(D, N)))
eps0 = 0.12 * (0.7 + 0.6 * rs.random(N))


def run(epoch):
    os.environ["AHMC_DENSE_EPOCH"] = "1" if epoch else "0"
    os.environ["AHMC_DENSE_EPOCH_MIN"] = "32"
    lf = A.Leapfrog(eps0)
    k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=10, delta_max=1000.0)))
    g = A.Engine(A.Hamiltonian(A.DenseEuclideanMetric(Minv), A.DenseGaussian(P)), N, rng=A.PhiloxRNG(77), lib=hip)
    g.set_integrator(lf)
    g.set_position(th0)
    g.adaptor_init(A.StepSizeAdaptor(0.8, lf))
    out = []
    for i in range(1, iters + 1):
        t0 = time.time()
        g.run(k, i, iters, i_first=i)
        g.sync()
        st = g.stats()
        out.append((g.theta().copy(), st["n_steps"].copy(), st["acceptance_rate"].copy(), g.get_stepsize().copy(), time.time() - t0))
    print("epoch" if epoch else "step ", "launches", g.info("dense_epoch_launches"), "gemm", g.info("dense_gemm_launches"), g.info("dense_gemm_small_launches"),
          "times", [round(o[4], 3) for o in out])
    g.close()
    return out


a = run(False)
b = run(True)
for i, (x, y) in enumerate(zip(a, b)):
    same = x[1] == y[1]
    dth = np.abs(x[0] - y[0]).max(axis=0)
    print(f"iter {i + 1}: n_steps equal {same.mean():.4f} (mean {x[1].mean():.1f} / {y[1].mean():.1f}); max|dθ| on equal chains {dth[same].max() if same.any() else -1:.3g}; "
          f"max|dα| {np.abs(x[2] - y[2])[same].max() if same.any() else -1:.3g}; max|dϵ| {np.abs(x[3] - y[3])[same].max() if same.any() else -1:.3g}; finite {np.isfinite(y[0]).all()}")
