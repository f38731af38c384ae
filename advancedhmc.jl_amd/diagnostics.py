"""Output side of the hot path: what the callers do with the draws and the
stat arrays — `bundle_samples` (ext/AdvancedHMCMCMCChainsExt.jl:7-41), EBFMI (src/diagnosis.jl:1-3)
and an effective-sample-size estimator.

ESS: the reference never computes it itself (MCMCChains.jl does, and no reference test calls it:
parity unpinned).  Defined here as Geyer's initial-monotone-sequence estimator on the FFT
autocorrelation, per chain and dimension:  τ = −1 + 2 Σ_{t≥0} P̂_t with P̂_t = ρ̂_{2t} + ρ̂_{2t+1}
truncated at the first non-positive pair and made non-increasing;  ESS = n / τ."""
from __future__ import annotations

import numpy as np


def autocorrelation(x, axis=0):
    """normalised autocorrelation along `axis` by FFT (biased estimator, lag 0 = 1)"""
    x = np.moveaxis(np.asarray(x, dtype=np.float64), axis, 0)
    n = x.shape[0]
    xc = x - x.mean(axis=0, keepdims=True)
    m = 1 << (2 * n - 1).bit_length()
    f = np.fft.rfft(xc, n=m, axis=0)
    acov = np.fft.irfft(f * np.conj(f), n=m, axis=0)[:n] / n
    var = acov[0]
    with np.errstate(invalid="ignore", divide="ignore"):
        rho = acov / var
    return np.moveaxis(rho, 0, axis)


def ess(draws, axis=0):
    """Effective sample size of each series along `axis` (Geyer initial monotone sequence).
    draws: (n_draws, ...) → array of the remaining shape.  Constant series give n_draws."""
    x = np.moveaxis(np.asarray(draws, dtype=np.float64), axis, 0)
    n = x.shape[0]
    rho = autocorrelation(x, axis=0)
    rho = np.where(np.isfinite(rho), rho, 0.0)
    npair = n // 2
    P = rho[0:2 * npair:2] + rho[1:2 * npair:2]          # P_t = ρ_2t + ρ_2t+1
    positive = np.cumprod(P > 0, axis=0).astype(bool)      # stop at the first non-positive pair
    P = np.where(positive, P, 0.0)
    P = np.minimum.accumulate(P, axis=0)                   # initial monotone sequence
    tau = -1.0 + 2.0 * P.sum(axis=0)
    tau = np.maximum(tau, 1.0 / n)
    # a series that never moved (every transition rejected) has no autocorrelation to estimate: n_draws, as k_ess and the CPU checker
    # (round 6: this function returned n² there — found by tests/test_random_configurations.py::test_random_diagnostics)
    return np.where(x.max(axis=0) == x.min(axis=0), float(n), n / tau)


def EBFMI(energies, axis=0):
    """mean((E_{i+1} − E_i)²) / var(E) per chain (src/diagnosis.jl:1-3; var with the n−1 divisor as Statistics.var)"""
    e = np.moveaxis(np.asarray(energies, dtype=np.float64), axis, 0)
    d = np.diff(e, axis=0)
    return (d * d).mean(axis=0) / e.var(axis=0, ddof=1)


INTERNALS = ("n_steps", "is_accept", "acceptance_rate", "log_density", "hamiltonian_energy", "hamiltonian_energy_error",
             "max_hamiltonian_energy_error", "tree_depth", "numerical_error", "step_size", "nom_step_size", "is_adapt")


def bundle_samples(thetas, stats, param_names=None, discard_initial=0, thinning=1):
    """The Chains-shaped bundle of ext/AdvancedHMCMCMCChainsExt.jl:7-41 as plain arrays:
    `value` (n_samples, n_params + n_internals, n_chains), `names`, `internals` (names of the stat
    columns).  `thetas`: list of (D, N) (or (D,)) draws; `stats`: list of stat dicts of `sample`."""
    th = np.stack([np.asarray(t).reshape(np.asarray(t).shape[0], -1) for t in thetas])[discard_initial::thinning]
    st = stats[discard_initial::thinning]
    n, D, N = th.shape
    names = list(param_names) if param_names is not None else [f"param_{i + 1}" for i in range(D)]
    if len(names) != D:
        raise ValueError("param_names must have one entry per dimension")
    internals = [k for k in INTERNALS if st and k in st[0]]
    cols = [np.stack([np.broadcast_to(np.asarray(s[k], dtype=np.float64), (N,)) for s in st]) for k in internals]
    value = np.concatenate([th] + [c[:, None, :] for c in cols], axis=1) if cols else th
    return {"value": value, "names": names + internals, "params": names, "internals": internals}
