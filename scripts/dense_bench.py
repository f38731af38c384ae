"""cfg4 (one GPU's shard): D=512 correlated Gaussian Σ_ij = 0.9^|i-j| (ℓπ = -½θᵀΣ⁻¹θ, Σ⁻¹ applied as a GEMM),
DenseEuclideanMetric (M⁻¹ = I initially, shared by all chains), NUTS(0.8) with step-size adaptation,
32 768 chains / 4 GPUs = 8 192 chains per GPU, f64.  F_lf = 6·D² + 14·D flop per chain-leapfrog as built
(three D×D products: M⁻¹r twice — position update and kinetic energy — and Σ⁻¹θ)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ahmc_amd as A
D = int(os.environ.get("D", 512)); N = int(os.environ.get("N", 8192))
n_adapt = int(os.environ.get("ADAPT", 60)); n_timed = int(os.environ.get("STEPS", 16))
dtype = np.float32 if os.environ.get("DTYPE") == "f32" else np.float64
lib = A.load_hip_library()
idx = np.arange(D)
Sigma = 0.9 ** np.abs(idx[:, None] - idx[None, :])
P = np.asfortranarray(np.linalg.inv(Sigma))
minv = np.asfortranarray(Sigma if os.environ.get("MINV") == "sigma" else np.eye(D))
h = A.Hamiltonian(A.DenseEuclideanMetric(minv), A.DenseGaussian(P))
e = A.Engine(h, N, dtype=dtype, rng=A.PhiloxRNG(0x5EED0004), lib=lib)
# round 6: TC=classic|strict (src/trajectory.jl:551-557,579-617), TEMPER=α (src/integrator.jl:198-209) — the variants k_dense_epoch2 / k_d_tree2 took over
lf = A.TemperedLeapfrog(np.full(N, 0.05), float(os.environ["TEMPER"])) if os.environ.get("TEMPER") else A.Leapfrog(np.full(N, 0.05)); e.set_integrator(lf)
TC = {"classic": A.ClassicNoUTurn, "strict": A.StrictGeneralisedNoUTurn}.get(os.environ.get("TC", ""), A.GeneralisedNoUTurn)
e.set_position(np.asfortranarray(np.random.default_rng(4).random((D, N))))
e.find_good_stepsize()
if os.environ.get("ADAPTOR") == "stan":  # WelfordCov adaptation of the shared dense metric (M⁻¹ → Σ: short trees)
    e.adaptor_init(A.StanHMCAdaptor(A.MassMatrixAdaptor(A.DenseEuclideanMetric(minv)), A.StepSizeAdaptor(0.8, lf)))
else:
    e.adaptor_init(A.StepSizeAdaptor(0.8, lf))
k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, TC(max_depth=10)))
t = time.perf_counter(); e.run(k, n_adapt, n_adapt); e.sync(); print("adaptation %d steps: %.2f s" % (n_adapt, time.perf_counter() - t))
e.run(k, 4, 0); e.sync()
e.reset_accum()
t = time.perf_counter(); e.run(k, n_timed, 0); e.sync(); dt = time.perf_counter() - t
acc = e.accum()
n = acc["n_transitions"] * N
lfps = acc["total_n_steps"] / dt
print("cfg4 dense D=%d N=%d %s: %.3e leapfrog/s, %.2f ms/transition, %.1f leapfrogs/transition, divergent %.4f, eps median %.4f, acc %.3f" % (
    D, N, dtype.__name__, lfps, dt / n_timed * 1e3, acc["total_n_steps"] / n, acc["n_divergent"] / n, np.median(e.get_stepsize()),
    np.mean(e.stats()["acceptance_rate"])))
print("epoch launches %d, gemm launches %d, pool %d" % (e.info("dense_epoch_launches"), e.info("dense_gemm_launches"), e.info("dense_pool")))
print("MFMA: %.2f TFLOP/s at F_lf = 4·D² flop per chain-leapfrog (two D×D products per step; useful leapfrogs only)" % (lfps * 4 * D * D / 1e12))
if os.environ.get("EPS_STATS"):
    eps = e.get_stepsize(); ns = e.stats()["n_steps"]
    print("eps percentiles 0/0.1/1/50/99/100:", np.percentile(eps, [0, 0.1, 1, 50, 99, 100]))
    print("n_steps percentiles 50/99/99.9/100:", np.percentile(ns, [50, 99, 99.9, 100]), "chains with n_steps >= 511:", int((ns >= 511).sum()))
