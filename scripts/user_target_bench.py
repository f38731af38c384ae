#!/usr/bin/env python
"""A user log-density on the device, three ways, next to the built-in family: 65 536 chains × D = 128 isotropic
Gaussian, per-chain Diag metric, NUTS(0.8) after a StanHMCAdaptor / StepSizeAdaptor warm-up, f64 — chain-leapfrogs per second of the draws.

  builtin   AHMC_TARGET_ISO_GAUSS                      (fused k_nuts)
  plugin    tests/user_targets/iso_gauss.hpp           (the same fused kernels with the user's device function inside)
  bitcode / object   tests/user_targets/iso_gauss_object.hip compiled to raw amdgcn bitcode / a relocatable device object and LINKED into
            the fused kernels (build_target_plugin_from_object: device LTO inlines it)
  kernel    tests/user_targets/kernels.hip iso_gauss_f64 as a hipFunction_t (the engine launches it once per global step)
  asktell   the same density evaluated by torch on the device through ahmc_ext_* (a host round trip per leapfrog)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ahmc_amd as A  # noqa: E402
from ahmc_amd import _capi as capi  # noqa: E402

D, N = int(os.environ.get("D", 128)), int(os.environ.get("N", 65536))
n_adapt, n_draw = int(os.environ.get("ADAPT", 100)), int(os.environ.get("DRAWS", 60))
UT = os.path.join(ROOT, "tests", "user_targets")
lib = A.load_hip_library()
which = sys.argv[1:] or ["builtin", "plugin", "bitcode", "object", "kernel"]
out = {}
for mode in which:
    metric = A.DiagEuclideanMetric(np.ones((D, N), order="F"))
    keep = None
    if mode == "builtin":
        target = A.IsoGaussian(D)
    elif mode == "plugin":
        target = A.PluginTarget(D, os.path.join(UT, "iso_gauss.hpp"))
    elif mode in ("object", "bitcode"):   # the same density as COMPILED device code, linked into the fused kernels (round 5)
        from ahmc_amd.build import build_device_object
        target = A.ObjectTarget(D, build_device_object(os.path.join(UT, "iso_gauss_object.hip"), bitcode=(mode == "bitcode")))
    elif mode == "kernel":
        from ahmc_amd.build import build_code_object
        from ahmc_amd.hipmod import Module
        keep = Module(build_code_object(os.path.join(UT, "kernels.hip")))
        target = A.KernelTarget(D, keep.function("iso_gauss_f64"), handle_kind=capi.KERNEL_HIP_FUNCTION, block_threads=256, chains_per_block=4)
    else:
        LOG2PI = 1.8378770664093454835606594728112

        def fn(th):  # θ (D, N) torch tensor on the device → (ℓπ, ∇ℓπ)
            return -(th * th).sum(dim=0) / 2 - D * LOG2PI / 2, -th
        target = A.ExternalTarget(D, fn)
    e = A.Engine(A.Hamiltonian(metric, target), N, rng=A.PhiloxRNG(0x5EED0002), lib=lib)
    lf = A.Leapfrog(np.full(N, 0.1))
    e.set_integrator(lf)
    e.set_position(np.asfortranarray(np.random.default_rng(2).random((D, N))))
    e.find_good_stepsize()
    fused = mode in ("builtin", "plugin", "object", "bitcode")
    e.adaptor_init(A.StanHMCAdaptor(A.MassMatrixAdaptor(metric), A.StepSizeAdaptor(0.8, lf)) if fused else A.StepSizeAdaptor(0.8, lf))
    k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=10)))
    t = time.perf_counter(); e.run(k, n_adapt, n_adapt); e.sync(); ta = time.perf_counter() - t
    e.reset_accum()
    t = time.perf_counter(); e.run(k, n_draw, 0); e.sync(); dt = time.perf_counter() - t
    acc = e.accum()
    out[mode] = {"leapfrogs_per_s": acc["total_n_steps"] / dt, "ms_per_transition": dt / n_draw * 1e3, "leapfrogs_per_transition": acc["total_n_steps"] / (n_draw * N),
                 "warmup_s": ta, "max_abs_mean": float(np.abs(acc["sum_theta"].sum(axis=1) / (n_draw * N)).max())}
    print(mode, json.dumps(out[mode]), flush=True)
    e.close()
print(json.dumps({"D": D, "N": N, "n_adapt": n_adapt, "n_draw": n_draw, "results": out}))
