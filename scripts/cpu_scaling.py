"""CPU-oracle thread scaling on this host (informs bench.py's cpu_baseline sample)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, time, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "oracle"))
import numpy as np, ahmc_amd as A, bench, build_oracle
lib = A.CLib(build_oracle.build())
n = int(sys.argv[1])
eng, k = bench.build_engine(A, lib, 128, n, 1, 0)
eng.run(k, 60, 60)
t = time.perf_counter(); eng.run(k, 20, 0); dt = time.perf_counter() - t
print(eng.accum(False)["total_n_steps"] / dt)
''' % (ROOT, ROOT)
for th in (1, 8, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1): break
    env = dict(os.environ, OMP_NUM_THREADS=str(th), OMP_PROC_BIND="false")
    out = subprocess.run([sys.executable, "-c", code, str(max(256, 64 * th))], env=env, capture_output=True, text=True)
    print(th, "threads:", out.stdout.strip() or out.stderr[-300:])
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("cpu.max n/a")
