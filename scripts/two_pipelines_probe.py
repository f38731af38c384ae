#!/usr/bin/env python
"""Experiment (round 6): would two chain pipelines pay for the fused kernels?  The same N chains as ONE engine against TWO engines of N/2
(contiguous blocks, chain_offset) driven from two host threads on their own streams — the tail of one engine's launch runs beside the other's
bulk.  No engine change: an upper bound on what a two-pipeline scheduler inside `ahmc_sample` could gain.
    python scripts/two_pipelines_probe.py cfg3 [cfg2]"""
import sys, threading, time, json
import numpy as np
sys.path.insert(0, ".")
import torch
import ahmc_amd as A
import bench

lib = A.load_hip_library()
CFG = {"cfg2": dict(D=128, target="iso", metric="diag", adaptor="stan", N=65536),
       "cfg3": dict(D=32, target="funnel", metric="diag", adaptor="stan", N=65536)}


def run(engs, kernels, n_adapts, n_draws):
    for e in engs:
        e.sync()
    res = [None] * len(engs)

    def work(i):
        e, k = engs[i], kernels[i]
        t0 = time.perf_counter()
        e.run(k, n_adapts, n_adapts)
        e.sync()
        t1 = time.perf_counter()
        a = e.accum(moments=False)["total_n_steps"]
        e.run(k, n_draws, 0)
        e.sync()
        t2 = time.perf_counter()
        res[i] = (t0, t1, t2, a, e.accum(moments=False)["total_n_steps"])
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(engs))]
    T0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    T1 = time.perf_counter()
    la, ld = sum(r[3] for r in res), sum(r[4] for r in res)
    ta = max(r[1] for r in res) - min(r[0] for r in res)
    td = max(r[2] for r in res) - min(r[1] for r in res)
    return {"whole": (la + ld) / (T1 - T0), "warmup": la / ta, "draws_approx": ld / td, "seconds": T1 - T0}


for name in sys.argv[1:] or ["cfg3"]:
    cfg = CFG[name]; N = cfg["N"]
    for parts in (1, 2, 1, 2, 4):
        engs, ks = [], []
        for p in range(parts):
            e, k = bench.build_engine(A, lib, cfg, N // parts, 1234, p * (N // parts))
            engs.append(e); ks.append(k)
        r = run(engs, ks, 1000, 1000)
        print(name, "engines", parts, json.dumps({k: float("%.4g" % v) for k, v in r.items()}), flush=True)
        for e in engs: e.close()
