"""Small committed fixtures for the GPU parity tests: the ORACLE's results on fixed seeded inputs, so that
tests/test_gpu_parity.py::test_against_committed_fixtures can check the HIP engine without rebuilding the oracle
(and so that a change of either side shows up as a diff of this file).

    python tests/golden/make_oracle_fixtures.py        # needs oracle/libahmc_oracle.so (python oracle/build_oracle.py)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import ahmc_amd as A  # noqa: E402
from ahmc_amd import _capi  # noqa: E402
from oracle.build_oracle import build  # noqa: E402

CASES = {
    # name: (target, D, N, metric, kernel factory, eps, transitions)
    "nuts_iso_diag": ("iso", 6, 32, "diag", lambda lf: A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=7))), 0.25, 3),
    "nuts_funnel_slice": ("funnel", 6, 32, "unit", lambda lf: A.HMCKernel(A.Trajectory(A.SliceTS, lf, A.GeneralisedNoUTurn(max_depth=6))), 0.3, 3),
    "hmc_endpoint": ("iso", 6, 32, "diag", lambda lf: A.HMCKernel(A.Trajectory(A.EndPointTS, lf, A.FixedNSteps(8))), 0.2, 3),
    "hmc_multinomial": ("iso", 6, 32, "unit", lambda lf: A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.FixedNSteps(8))), 0.2, 3),
}


def build_case(name, lib):
    target, D, N, metric, mk, eps, n = CASES[name]
    rng = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
    m = A.UnitEuclideanMetric((D, N)) if metric == "unit" else A.DiagEuclideanMetric(np.asfortranarray(0.5 + rng.random((D, N))))
    t = A.IsoGaussian(D) if target == "iso" else A.Funnel(D)
    lf = A.Leapfrog(np.full(N, eps))
    e = A.Engine(A.Hamiltonian(m, t), N, rng=12345, lib=lib)
    e.set_integrator(lf)
    th0 = 0.7 * rng.normal(size=(D, N))
    e.set_position(th0)
    return e, mk(lf), n


def run_case(name, lib, with_margin=False):
    """with_margin (the oracle only): per transition, the chains' smallest decision margin (tests/parity_util.py) — what lets the GPU
    test demand exact agreement except at the oracle's own near-ties"""
    e, k, n = build_case(name, lib)
    out = {}
    if with_margin:
        sys.path.insert(0, os.path.join(HERE, ".."))
        import parity_util as PU

        PU.reset_margin(e)
    for it in range(n):
        e.transition(k)
        s = e.stats()
        z = e.phasepoint()
        if with_margin:
            out[f"{name}/margin{it}"] = PU.decision_margin(e)
        out[f"{name}/theta{it}"] = z.theta.copy()
        out[f"{name}/n_steps{it}"] = s["n_steps"].copy()
        out[f"{name}/H{it}"] = s["hamiltonian_energy"].copy()
        out[f"{name}/acc{it}"] = s["acceptance_rate"].copy()
    e.close()
    return out


if __name__ == "__main__":
    lib = _capi.CLib(build())
    data = {}
    for name in CASES:
        data.update(run_case(name, lib, with_margin=True))
    path = os.path.join(HERE, "oracle_fixtures.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, os.path.getsize(path), "bytes")
