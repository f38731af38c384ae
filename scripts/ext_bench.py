#!/usr/bin/env python
"""User log-density through ask / tell (ahmc_ext_*) on one GPU: what a request costs.

    python scripts/ext_bench.py [--chains 65536] [--dim 128] [--transitions 8] [--host]

The density is the isotropic Gaussian of cfg2 evaluated OUTSIDE the engine: by torch on the device, reading θ in
place at ahmc_theta_ptr and handing device pointers back (default), or by numpy on the host through the (D,N) copies
of the protocol (--host: the PCIe-inclusive figure).  Reports chain-leapfrogs/s, requests, µs per request, and the
fused engine's figure for the same chains and step sizes as the yardstick.  One untimed transition first: the first
ahmc_ext_begin allocates the step-synchronous engine's buffers (round 2 measurement, 16 384 x 128: 0.22 ms per
request in steady state — 0.16 ms of it ahmc_ext_advance = ingest + k_d_tree + compaction + the 4-byte read-back —
against 2.0 ms when the allocations are averaged in; scripts/ext_trace.py splits a request into its calls)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LOG2PI = 1.8378770664093454835606594728112


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=65536)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--transitions", type=int, default=8)
    ap.add_argument("--host", action="store_true")
    args = ap.parse_args()
    import torch
    import ahmc_amd as A

    D, N = args.dim, args.chains
    lib = A.load_hip_library()
    rng = np.random.default_rng(1)
    th0 = rng.normal(size=(D, N))
    eps = np.full(N, 0.6 / D ** 0.25)
    lf = A.Leapfrog(eps)
    kernel = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=10)))
    k = kernel.cfg()

    def host_fn(theta):
        return -(LOG2PI * D + (theta * theta).sum(axis=0)) / 2, -theta

    e = A.Engine(A.Hamiltonian(A.UnitEuclideanMetric(D), A.ExternalTarget(D, host_fn)), N, rng=3, lib=lib)
    e.set_integrator(lf)
    e.set_position(th0)
    out = {"D": D, "N": N, "transitions": args.transitions, "closure": "host numpy" if args.host else "device torch"}
    requests = 0
    e._call("ahmc_ext_begin", C.byref(k), 1)  # untimed: allocations of the step-synchronous engine
    e._ext_drive()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if args.host:
        for _ in range(args.transitions):
            e._call("ahmc_ext_begin", C.byref(k), 1)
            requests += e._ext_drive()
    else:
        buf = torch.empty((N, D), dtype=torch.float64, device="cuda")  # row c = chain c: the (D,N) column-major layout
        n = C.c_int64()
        for _ in range(args.transitions):
            e._call("ahmc_ext_begin", C.byref(k), 1)
            while True:
                e._call("ahmc_ext_pending", C.byref(n), None, buf.data_ptr())  # device-to-device copy of θ (in-place views need a torch storage wrapper)
                if n.value == 0:
                    break
                lp = -(LOG2PI * D + (buf * buf).sum(dim=1)) / 2
                torch.cuda.synchronize()
                e._call("ahmc_ext_advance", lp.data_ptr(), buf.data_ptr())  # -∇ℓπ = θ
                requests += 1
    e.sync()
    dt = time.perf_counter() - t0
    leap = int(e.stats()["n_steps"].sum())  # last transition only; scale by the number of transitions for the estimate
    out.update(seconds=dt, requests=requests, us_per_request=dt / max(requests, 1) * 1e6,
               leapfrogs_per_s_estimate=leap * args.transitions / dt)
    e.close()
    # the fused engine on the same chains
    f = A.Engine(A.Hamiltonian(A.UnitEuclideanMetric(D), A.IsoGaussian(D)), N, rng=3, lib=lib)
    f.set_integrator(lf)
    f.set_position(th0)
    f.run(kernel, 4)
    f.sync()
    t0 = time.perf_counter()
    f.run(kernel, args.transitions)
    f.sync()
    dtf = time.perf_counter() - t0
    out["fused_leapfrogs_per_s"] = f.accum(moments=False)["total_n_steps"] / dtf
    f.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
