#!/usr/bin/env python
"""cfg3-like per-transition work traces from the CPU oracle (CPU only, ≈ 1 minute): the loop of bench.py's cfg3 — D = 32 funnel,
per-chain Diag metric, NUTS(0.8), StanHMCAdaptor, 1 000 adapting transitions + 1 000 draws — on N (default 4 096) chains,
n_steps of every chain at every transition → $TRACE (default /tmp/cfg3_trace.npz).  Input of pack_model.py / queue_model.py."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ahmc_amd as A
import bench as B
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import build_oracle
lib = A.CLib(build_oracle.build())
cfg = B.CONFIGS[os.environ.get("CFG", "cfg3")]  # CFG=cfg2: the iso Gaussian of the headline config
N = int(os.environ.get("N", 4096)); n_ad = int(os.environ.get("ADAPT", 1000)); n_dr = int(os.environ.get("DRAWS", 1000))
eng, kernel = B.build_engine(A, lib, cfg, N, cfg["seed"], 0)
t = time.time()
Wa = np.zeros((n_ad, N), np.int32)
for i in range(1, n_ad + 1):
    eng.transition(kernel); eng.adapt(i, n_ad)
    Wa[i - 1] = eng.stats(["n_steps"])["n_steps"]
print("warm-up", time.time() - t, "s, mean lf/transition", Wa.mean())
eps = eng.get_stepsize()
Wd = np.zeros((n_dr, N), np.int32)
t = time.time()
for i in range(n_dr):
    eng.transition(kernel)
    Wd[i] = eng.stats(["n_steps"])["n_steps"]
print("draws", time.time() - t, "s, mean lf/transition", Wd.mean())
np.savez_compressed(os.environ.get('TRACE', '/tmp/cfg3_trace.npz'), Wa=Wa, Wd=Wd, eps=eps)
