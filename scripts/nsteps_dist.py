"""distribution of n_steps per transition in the bench configuration (after adaptation)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ahmc_amd as A, bench
lib = A.load_hip_library()
eng, k = bench.build_engine(A, lib, 128, 65536, 0x5EED0002, 0)
eng.run(k, 200, 200)
import time
for it in range(6):
    eng.sync(); t = time.perf_counter(); eng.run(k, 1, 0); eng.sync(); dt = time.perf_counter() - t
    n = eng.stats(["n_steps"])["n_steps"]
    vals, cnt = np.unique(n, return_counts=True)
    print(f"transition {it}: {dt*1e3:.3f} ms, mean {n.mean():.2f}, max {n.max()},", dict(zip(vals.tolist(), cnt.tolist())))
eps = eng.get_stepsize(); print("eps quantiles", np.quantile(eps, [0, 0.01, 0.5, 0.99, 1]))
