"""k_dgemm timing probe: dense metric + dense target, set_position = 2 GEMMs (P·θ, M⁻¹·r); run under rocprofv3"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ahmc_amd as A
D = int(os.environ.get("D", 512))
lib = A.load_hip_library()
idx = np.arange(D); S = 0.9 ** np.abs(idx[:, None] - idx[None, :])
for N in (8192, 4096, 2048, 512):
    h = A.Hamiltonian(A.DenseEuclideanMetric(np.asfortranarray(S)), A.DenseGaussian(np.asfortranarray(np.linalg.inv(S))))
    e = A.Engine(h, N, rng=1, lib=lib)
    th = np.asfortranarray(np.random.default_rng(0).normal(size=(D, N)))
    for _ in range(10):
        e.set_position(th, th)
    e.close()
