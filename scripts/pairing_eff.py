"""lockstep efficiency of 2/4/8 chains per wave under different chain orderings (bench configuration)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ahmc_amd as A, bench
lib = A.load_hip_library()
eng, k = bench.build_engine(A, lib, 128, 65536, 0x5EED0002, 0)
eng.run(k, 200, 200)
ns = []
for it in range(24):
    eng.run(k, 1, 0); ns.append(eng.stats(["n_steps"])["n_steps"].astype(np.int64))
ns = np.array(ns)            # (T, N)
eps = eng.get_stepsize()
def eff(x, order, cpw):
    y = x[:, order].reshape(x.shape[0], -1, cpw)
    return y.mean() / y.max(axis=2).mean()
ident = np.arange(ns.shape[1]); by_eps = np.argsort(eps); by_mean = np.argsort(ns[:8].mean(axis=0))
for cpw in (2, 4, 8):
    print(cpw, "chains/wave: lockstep efficiency  random %.3f  sorted-by-eps %.3f  sorted-by-past-mean (8 its, scored on the other 16) %.3f" % (
        eff(ns, ident, cpw), eff(ns, by_eps, cpw), eff(ns[8:], by_mean, cpw)))
print("corr(eps, mean n_steps) = %.3f" % np.corrcoef(eps, ns.mean(axis=0))[0, 1])
