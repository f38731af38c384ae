"""AHMC_DENSE_SPLIT = 0 / 1 / 2 on the dense engine, in one process (the switch is read per call):
(1) the three schedules give bit-identical chains on a small problem, (2) their speed on the cfg4 shard.
Writes one line per result to gpurun_out/split_check.log as it goes (the GPU call may be cut short).

    python scripts/dense_split_check.py [ADAPT=24] [STEPS=8]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import ahmc_amd as A  # noqa: E402

os.makedirs("gpurun_out", exist_ok=True)
LOG = open("gpurun_out/split_check.log", "a")


def say(s):
    print(s, flush=True)
    LOG.write(s + "\n")
    LOG.flush()
    os.fsync(LOG.fileno())


lib = A.load_hip_library()
MODES = [m for m in os.environ.get("MODES", "1,2,0").split(",")]


def small(split):
    os.environ["AHMC_DENSE_SPLIT"] = split
    rng = np.random.default_rng(5)
    D, N = 48, 2304
    B = rng.normal(size=(D, D))
    P = np.asfortranarray(B @ B.T / D + np.eye(D))
    h = A.Hamiltonian(A.DenseEuclideanMetric(np.asfortranarray(np.linalg.inv(P) * 0.7 + 0.3 * np.eye(D))), A.DenseGaussian(P))
    lf = A.Leapfrog(np.full(N, 0.15) * (0.6 + 0.8 * rng.random(N)))
    k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=7)))
    e = A.Engine(h, N, rng=31, lib=lib)
    e.set_integrator(lf)
    e.set_position(0.5 * rng.normal(size=(D, N)))
    e.run(k, 5)
    z, s, a = e.phasepoint(), e.stats(), e.accum()
    e.close()
    return z.theta.copy(), z.r.copy(), s["n_steps"].copy(), a["total_n_steps"]


ref = None
for m in ([] if os.environ.get("SKIP_SMALL") else MODES):
    t = time.perf_counter()
    out = small(m)
    if ref is None:
        ref = out
        say("small split=%s: reference, total_n_steps %d (%.1f s)" % (m, out[3], time.perf_counter() - t))
    else:
        same = all(np.array_equal(a, b) for a, b in zip(out[:3], ref[:3])) and out[3] == ref[3]
        say("small split=%s: %s (%.1f s)" % (m, "IDENTICAL to the reference" if same else "DIFFERENT", time.perf_counter() - t))

D, N = 512, int(os.environ.get("N", 8192))
n_adapt, n_timed = int(os.environ.get("ADAPT", 24)), int(os.environ.get("STEPS", 8))
idx = np.arange(D)
Sigma = 0.9 ** np.abs(idx[:, None] - idx[None, :])
P = np.asfortranarray(np.linalg.inv(Sigma))
h = A.Hamiltonian(A.DenseEuclideanMetric(np.asfortranarray(np.eye(D))), A.DenseGaussian(P))
os.environ["AHMC_DENSE_SPLIT"] = MODES[0]
e = A.Engine(h, N, rng=A.PhiloxRNG(0x5EED0004), lib=lib)
lf = A.Leapfrog(np.full(N, 0.05))
e.set_integrator(lf)
e.set_position(np.asfortranarray(np.random.default_rng(4).random((D, N))))
e.find_good_stepsize()
e.adaptor_init(A.StepSizeAdaptor(0.8, lf))
k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=10)))
t = time.perf_counter()
e.run(k, n_adapt, n_adapt)
e.sync()
say("cfg4 shard: adaptation %d transitions %.2f s" % (n_adapt, time.perf_counter() - t))
for rep in range(2):
    for m in MODES:
        os.environ["AHMC_DENSE_SPLIT"] = m
        e.reset_accum()
        t = time.perf_counter()
        e.run(k, n_timed, 0)
        e.sync()
        dt = time.perf_counter() - t
        acc = e.accum()
        lfps = acc["total_n_steps"] / dt
        say("cfg4 shard split=%s rep %d: %.3e leapfrog/s = %.1f TFLOP/s (%.1f leapfrogs/transition, %.1f ms/transition)" % (
            m, rep, lfps, lfps * 4 * D * D / 1e12, acc["total_n_steps"] / (n_timed * N), dt / n_timed * 1e3))
