"""wall time of the adaptation phase of cfg2 (fused warm-up batches, or AHMC_ADAPT_FUSED=0 for per-transition launches)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ahmc_amd as A
import bench
lib = A.load_hip_library()
eng, kernel = bench.build_engine(A, lib, 128, 65536, 0x5EED0002, 0)
n = int(os.environ.get("ADAPT", 200))
t = time.perf_counter(); eng.run(kernel, n, n); eng.sync(); dt = time.perf_counter() - t
print("cfg2 adaptation: %d transitions in %.3f s = %.2f ms per transition" % (n, dt, dt / n * 1e3))
