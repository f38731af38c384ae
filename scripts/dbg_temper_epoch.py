import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ahmc_amd as A
hip = A.load_hip_library()
D, N = int(os.environ.get("DBG_D", "512")), 128
rs = np.random.default_rng(1)
idx = np.arange(D)
P = np.asfortranarray(np.linalg.inv(0.9 ** np.abs(idx[:, None] - idx[None, :])))
Q, _ = np.linalg.qr(rs.normal(size=(D, D)))
Mi = (Q * np.linspace(0.6, 2.0, D)) @ Q.T
Minv = np.asfortranarray((Mi + Mi.T) / 2)
th0 = np.asfortranarray(rs.normal(size=(D, N)))
eps0 = 0.12 * (0.7 + 0.6 * rs.random(N))
for engine in ("step", "epoch"):
    os.environ["AHMC_DENSE_EPOCH"] = "1" if engine == "epoch" else "0"
    os.environ["AHMC_DENSE_EPOCH_MIN"] = "32"
    lf = A.TemperedLeapfrog(eps0, 1.05)
    k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=10, delta_max=1000.0)))
    g = A.Engine(A.Hamiltonian(A.DenseEuclideanMetric(Minv), A.DenseGaussian(P)), N, rng=A.PhiloxRNG(78), lib=hip)
    g.set_integrator(lf)
    g.set_position(th0)
    for it in range(2):
        g.run(k, 1, 0)
        st = g.stats()
        print(engine, it, "n_steps", st["n_steps"][:10], "dHmax", st["max_hamiltonian_energy_error"][:4], "H", st["hamiltonian_energy"][:3], "numerr", st["numerical_error"][:10], "epoch launches", g.info("dense_epoch_launches"), flush=True)
    g.close()
