#!/usr/bin/env python
"""Where an ask / tell request spends its wall time (device closure): per-call averages over one NUTS transition."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ahmc_amd as A
LOG2PI = 1.8378770664093454835606594728112
D, N = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 16384
lib = A.load_hip_library()
rng = np.random.default_rng(1)
th0 = rng.normal(size=(D, N)); eps = np.full(N, 0.6 / D ** 0.25); lf = A.Leapfrog(eps)
kernel = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=10))); k = kernel.cfg()
e = A.Engine(A.Hamiltonian(A.UnitEuclideanMetric(D), A.ExternalTarget(D, lambda th: (-(LOG2PI * D + (th * th).sum(axis=0)) / 2, -th))), N, rng=3, lib=lib)
e.set_integrator(lf); e.set_position(th0)
buf = torch.empty((N, D), dtype=torch.float64, device="cuda"); n = C.c_int64()
T = {"pending": 0.0, "torch_launch": 0.0, "torch_sync": 0.0, "advance": 0.0}; req = 0
for rep in range(3):
    e._call("ahmc_ext_begin", C.byref(k), 1)
    while True:
        t0 = time.perf_counter(); e._call("ahmc_ext_pending", C.byref(n), None, buf.data_ptr()); t1 = time.perf_counter()
        if n.value == 0: break
        lp = -(LOG2PI * D + (buf * buf).sum(dim=1)) / 2; t2 = time.perf_counter()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        e._call("ahmc_ext_advance", lp.data_ptr(), buf.data_ptr()); t4 = time.perf_counter()
        if rep > 0:
            T["pending"] += t1 - t0; T["torch_launch"] += t2 - t1; T["torch_sync"] += t3 - t2; T["advance"] += t4 - t3; req += 1
print({k2: round(v / req * 1e6, 1) for k2, v in T.items()}, "us per request over", req, "requests, N =", N)
