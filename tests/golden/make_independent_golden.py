"""An INDEPENDENT numpy restatement of the RNG-free pieces of the reference, written from the formulas in
SURVEY.md Appendix A (not from oracle/): leapfrog (src/integrator.jl:216-265), tempered leapfrog (:198-209),
dual averaging (src/adaptation/stepsize.jl:178-210), WelfordVar / NutpieVar (src/adaptation/massmatrix.jl:141-157,
:238-250).  Writes tests/golden/independent_golden.json in the schema of oracle/dump_golden.jl, so that
tests/test_oracle_golden.py replays the oracle against it exactly as it would against a Julia dump.

    python tests/golden/make_independent_golden.py
"""
import json
import os

import numpy as np

LOG2PI = float(np.log(2 * np.pi))


def lp(th):
    return -(th ** 2).sum(0) / 2 - th.shape[0] * LOG2PI / 2


def leapfrog(th, r, minv, eps, n, alpha=None):
    th, r = th.copy(), r.copy()
    e = eps if n > 0 else -eps
    g = th.copy()  # -∇ℓπ for the iso Gaussian
    for i in range(1, abs(n) + 1):
        if alpha is not None:
            r = r * np.sqrt(alpha) if 2 * (i - 1) + 1 <= abs(n) else r / np.sqrt(alpha)
        r = r - e / 2 * g
        th = th + e * (minv * r)
        g = th.copy()
        r = r - e / 2 * g
        if alpha is not None:
            r = r * np.sqrt(alpha) if 2 * (i - 1) + 2 <= abs(n) else r / np.sqrt(alpha)
    return th, r, lp(th), -(minv * r * r).sum(0) / 2, g


def main():
    out = {}
    D, N = 5, 4
    th = np.linspace(-1.0, 1.0, D * N).reshape(N, D).T
    r = np.linspace(0.5, -0.7, D * N).reshape(N, D).T
    Minv = np.linspace(0.5, 1.5, D * N).reshape(N, D).T
    eps = np.linspace(0.05, 0.2, N)
    vec = lambda a: a.T.reshape(-1).tolist()
    for name, minv in (("unit", np.ones((D, N))), ("diag", Minv)):
        t = {"theta": vec(th), "r": vec(r), "eps": eps.tolist(), "lp0": lp(th).tolist(), "lk0": (-(minv * r * r).sum(0) / 2).tolist()}
        for n in (7, -4):
            a, b, l1, l2, g = leapfrog(th, r, minv, eps, n)
            t[f"step{n}"] = {"theta": vec(a), "r": vec(b), "lp": l1.tolist(), "lk": l2.tolist(), "grad": vec(g)}
        if name == "diag":
            t["minv"] = vec(Minv)
        out["leapfrog_" + name] = t
    a, b, l1, l2, _ = leapfrog(th, r, np.ones((D, N)), np.full(N, 0.1), 6, alpha=1.05)
    out["tempered"] = {"theta": vec(a), "r": vec(b), "lp": l1.tolist(), "lk": l2.tolist()}
    # dual averaging: γ=0.05, t0=10, κ=0.75, μ=log(10 ϵ0)
    alphas = [0.3, 0.95, 0.6, 1.0, 0.05, 0.8, 0.8, 0.7, 0.99, 0.4]
    m, mu, xbar, Hbar, e = 0, np.log(10 * 0.1), 0.0, 0.0, 0.1
    eps_seq = []
    for al in alphas:
        m += 1
        eta = 1 / (m + 10)
        Hbar = (1 - eta) * Hbar + eta * (0.8 - min(1.0, al))
        x = mu - Hbar * np.sqrt(m) / 0.05
        xbar = (1 - m ** -0.75) * xbar + m ** -0.75 * x
        e = float(np.exp(x))
        eps_seq.append(e)
    out["dual_averaging"] = {"alpha": alphas, "eps": eps_seq, "final": float(np.exp(xbar))}
    k = np.arange(1, 16)[:, None]
    xs = np.sin(np.arange(1, 4)[None, :] * k) * np.array([1.0, 2.0, 0.5])
    gs = np.cos(np.arange(1, 4)[None, :] * k) / np.array([1.0, 4.0, 0.25])
    n = len(xs)
    est = lambda z: n / ((n + 5) * (n - 1)) * ((z - z.mean(0)) ** 2).sum(0) + 1e-3 * 5 / (n + 5)
    out["welford"] = {"x": xs.tolist(), "g": gs.tolist(), "var": est(xs).tolist(), "nutpie": np.sqrt(est(xs) / est(gs)).tolist()}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "independent_golden.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
