/*
 * ahmc_hip.h — C ABI of the MI355X-native chain-batched HMC/NUTS trajectory engine.
 *
 * This header is the drop-in boundary.  The reference (AdvancedHMC.jl) has no
 * FFI: its boundary is Julia multiple dispatch.  Every entry point below therefore names the
 * reference method it stands in for (path:line under the AdvancedHMC.jl checkout); the Julia
 * package extension that `ccall`s them is julia/AdvancedHMCMI355XExt.jl.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.  Every function returns an
 *     int32 status (AHMC_OK == 0); the message for the last failure on a context is
 *     ahmc_last_error(ctx) (ahmc_last_error(NULL) for a failed ahmc_create).  Nothing throws
 *     or aborts across the ABI.  Numerical trouble is never an error: as in the reference it
 *     becomes -Inf energies, a rejection and a stat flag (src/hamiltonian.jl:95-104).
 *   - Arrays are column-major (D, N) exactly as a Julia Matrix{T} holds them: chain c owns the
 *     contiguous slice [c*D, (c+1)*D)  (src/integrator.jl:226-227, test/sampler-vec.jl:11).
 *     Per-chain scalars are length-N vectors.  The element type T is fixed at ahmc_create.
 *   - A `const void*`/`void*` array argument may be a host pointer or a device pointer of the
 *     context's device (unified addressing decides the copy direction).  The caller owns every
 *     pointer it passes; the library owns the context, its device state and its scratch.
 *   - A context is bound to one device and one stream and is not thread-safe; calls are
 *     asynchronous on the stream until ahmc_sync / any get_* (which synchronise).
 *   - Sign convention of the cached gradient follows the reference: `grad` holds
 *     -∇ℓπ(θ) = ∂H/∂θ  (src/hamiltonian.jl:45-48).  `lk` holds ℓκ = -K(r).
 *
 * The same ABI is implemented twice: advancedhmc.jl_amd/csrc (HIP, gfx950 — the product) and
 * oracle/ (scalar CPU restatement of the reference — test infrastructure only).
 */
#ifndef AHMC_HIP_H
#define AHMC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* v6 = v5 + ahmc_set_ref_compat, AHMC_INFO_STEPSIZE_SCALAR (round 6); v5 = v4 + ahmc_sample_reserve, ahmc_comm_info, AHMC_INFO_NUTS_DRAW_BATCH; v4 = v3 + the
 * device-side user log-densities (target plugin / kernel), the accumulator checkpoint, the dense engine's launch counters. */
#define AHMC_ABI_VERSION 6

typedef struct ahmc_ctx ahmc_ctx;

/* status codes */
enum {
  AHMC_OK = 0,
  AHMC_ERR_ARGUMENT = 1,    /* Julia ArgumentError / @argcheck  (src/hamiltonian.jl:55-57,94) */
  AHMC_ERR_UNSUPPORTED = 2, /* combination the engine has no kernel for                      */
  AHMC_ERR_RUNTIME = 3,     /* HIP runtime failure (message carries hipGetErrorString)        */
  AHMC_ERR_STATE = 4        /* call order violated (e.g. transition before set_position)      */
};

/* element type: Float32 / Float64 (src/AdvancedHMC.jl:37 default Float64) */
enum { AHMC_F32 = 0, AHMC_F64 = 1 };

/* metric: src/metric.jl:17-35 (Unit), :52-72 (Diag), :89-120 (Dense) */
enum { AHMC_METRIC_UNIT = 0, AHMC_METRIC_DIAG = 1, AHMC_METRIC_DENSE = 2 };

/* built-in log-density families evaluated inside the kernels (SURVEY.md §8d synthetic inputs);
 * AHMC_TARGET_EXTERNAL = the caller computes (ℓπ, ∇ℓπ): the `h.∂ℓπ∂θ(θ)` callback of
 * src/hamiltonian.jl:45-48 stays on the Julia side — single leapfrogs between ahmc_lf_pre and
 * ahmc_lf_post, whole transitions through the ask / tell calls ahmc_ext_*.                   */
enum {
  AHMC_TARGET_ISO_GAUSS = 0,  /* ℓπ = Σ -(log2π + θ²)/2           test/common.jl:40-44, m=0,s=1 */
  AHMC_TARGET_DIAG_GAUSS = 1, /* params: m[D], s[D]                test/common.jl:35-77           */
  AHMC_TARGET_FUNNEL = 2,     /* θ1~N(0,3²), θi~N(0,e^{θ1})        research/notebooks/geweke_test.ipynb cell 4 */
  AHMC_TARGET_HIER_GAUSS = 3, /* θ=(μ,logτ,x..): μ~N(0,1), logτ~N(0,1), xi~N(μ,τ²)  SURVEY §8d cfg5 */
  AHMC_TARGET_DENSE_GAUSS = 4,/* ℓπ = -½ θᵀPθ, params: P (D,D) column-major precision  SURVEY §8d cfg4 */
  AHMC_TARGET_EXTERNAL = 5,   /* the caller evaluates: ask / tell (ahmc_ext_*), ahmc_lf_pre / ahmc_lf_post              */
  AHMC_TARGET_PLUGIN = 6,     /* a user device FUNCTION compiled into the trajectory kernels (ahmc_set_target_plugin)     */
  AHMC_TARGET_KERNEL = 7      /* a user device KERNEL the engine launches itself (ahmc_set_target_kernel)                */
};

/* integrator: src/integrator.jl:71-74 (Leapfrog), :112-156 (Jittered), :174-209 (Tempered) */
enum { AHMC_INTEGRATOR_LEAPFROG = 0, AHMC_INTEGRATOR_JITTERED = 1, AHMC_INTEGRATOR_TEMPERED = 2 };

/* trajectory sampler: src/trajectory.jl:90 (EndPointTS), :102-119 (SliceTS), :129-136 (MultinomialTS) */
enum { AHMC_TS_ENDPOINT = 0, AHMC_TS_MULTINOMIAL = 1, AHMC_TS_SLICE = 2 };

/* dynamic termination criterion: src/trajectory.jl:414-449 */
enum { AHMC_TC_CLASSIC = 0, AHMC_TC_GENERALISED = 1, AHMC_TC_STRICT = 2 };

/* per-chain statistics of the last transition; names follow the reference's stat NamedTuple
 * (static: src/trajectory.jl:286-298; NUTS: :726-739; integrator: src/integrator.jl:58).
 * T-typed fields come back as T[N], integer fields as int32[N].                             */
enum {
  AHMC_STAT_N_STEPS = 0,                      /* int32 */
  AHMC_STAT_IS_ACCEPT = 1,                    /* int32 (0/1) */
  AHMC_STAT_ACCEPTANCE_RATE = 2,              /* T */
  AHMC_STAT_LOG_DENSITY = 3,                  /* T */
  AHMC_STAT_HAMILTONIAN_ENERGY = 4,           /* T */
  AHMC_STAT_HAMILTONIAN_ENERGY_ERROR = 5,     /* T */
  AHMC_STAT_MAX_HAMILTONIAN_ENERGY_ERROR = 6, /* T  (NUTS only) */
  AHMC_STAT_TREE_DEPTH = 7,                   /* int32 (NUTS only) */
  AHMC_STAT_NUMERICAL_ERROR = 8,              /* int32 (0/1) */
  AHMC_STAT_STEP_SIZE = 9,                    /* T */
  AHMC_STAT_NOM_STEP_SIZE = 10,               /* T */
  AHMC_STAT__COUNT = 11
};

/* adaptor kinds: src/adaptation/Adaptation.jl:41-64 (Naive), stan_adaptor.jl (Stan),
 * stepsize.jl (StepSizeAdaptor = NesterovDualAveraging), massmatrix.jl (WelfordVar)         */
enum {
  AHMC_ADAPT_NONE = 0,
  AHMC_ADAPT_STEPSIZE = 1,   /* StepSizeAdaptor(δ, integrator)                          */
  AHMC_ADAPT_MASSMATRIX = 2, /* MassMatrixAdaptor(metric) → WelfordVar, updated every step */
  AHMC_ADAPT_NAIVE = 3,      /* NaiveHMCAdaptor(pc, ssa)                                */
  AHMC_ADAPT_STAN = 4        /* StanHMCAdaptor(pc, ssa; 75/50/25)                       */
};

/* ------------------------------------------------------------------------------------------ */
/* lifetime                                                                                   */

/* Allocate a context for N chains of dimension D on `device`.  `stream` is a hipStream_t to
 * enqueue on (NULL → the library creates its own).  Replaces the implicit allocation done by
 * sample_init / resize (src/sampler.jl:25-46).                                               */
int32_t ahmc_create(int32_t device, int32_t dtype, int64_t D, int64_t N, void* stream,
                    ahmc_ctx** out);
int32_t ahmc_destroy(ahmc_ctx* ctx);
const char* ahmc_last_error(const ahmc_ctx* ctx);
int32_t ahmc_abi_version(void);
/* "hip:gfx950" for the product library, "cpu-oracle" for oracle/ */
const char* ahmc_backend(void);
int32_t ahmc_sync(ahmc_ctx* ctx);
/* the hipStream_t the context enqueues on (for HIP-event timing by the caller) */
void* ahmc_stream(ahmc_ctx* ctx);

/* ------------------------------------------------------------------------------------------ */
/* configuration (the fields of Hamiltonian / metric / integrator structs)                    */

/* Hamiltonian.ℓπ/∂ℓπ∂θ (src/hamiltonian.jl:1-20): choose a built-in family.  `params` holds
 * n_params elements of T (layout per family above) or NULL.                                  */
int32_t ahmc_set_target(ahmc_ctx* ctx, int32_t kind, const void* params, int64_t n_params);

/* A user log-density ON THE DEVICE — the `h.∂ℓπ∂θ(θ)` callable of src/hamiltonian.jl:45-48 / LogDensityProblems'
 * logdensity_and_gradient behind src/AdvancedHMC.jl:163-186 — without a host round trip.  Two forms (contract, helper
 * functions and an example: include/ahmc_user_target.h):
 *
 * ahmc_set_target_plugin: `plugin_so` is a shared object made from the user's device FUNCTION and the engine's own kernel
 *   sources (advancedhmc.jl_amd/build.py: build_target_plugin) for the context's element type and thread geometry
 *   (ahmc_get_info: AHMC_INFO_GROUP_LANES / _ELEMS_PER_LANE).  From then on every call of the built-in families' path —
 *   phasepoint, step, refresh, static and NUTS transitions, find_good_stepsize, ahmc_sample with the fused warm-up — runs
 *   the SAME fused kernels with the user's density inside.  `params`: n_params values of T the function reads (copied to
 *   device memory), or NULL.  Errors: AHMC_ERR_ARGUMENT if the file cannot be bound or was built for another element type /
 *   geometry / engine version.
 *
 * ahmc_set_target_kernel: `handle` is a device KERNEL the caller already owns — handle_kind AHMC_KERNEL_HIP_FUNCTION: a
 *   hipFunction_t (hipModuleGetFunction; what AMDGPU.jl compiles a Julia kernel to); AHMC_KERNEL_HIP_SYMBOL: the host
 *   address of a __global__ function linked into the process; AHMC_KERNEL_HOST (CPU checker only): a plain C function.
 *   Signature, for every kind (T = the context's element type):
 *       void f(const T* theta, T* lp, T* grad_neg, const int32_t* cols, int64_t n_cols, int32_t D, int64_t N, void* user)
 *   for k < n_cols: chain c = cols ? cols[k] : k; read theta[c·D .. c·D+D); write lp[c] = ℓπ(θ_c) and grad_neg[c·D + d] =
 *   −∂ℓπ/∂θ_d.  The engine launches it on the context's stream with ⌈n_cols / chains_per_block⌉ blocks of block_threads
 *   threads wherever the ask / tell protocol would hand the chains to the caller: whole transitions (static, NUTS),
 *   find_good_stepsize, phasepoint, step and ahmc_sample all work as for a built-in family, on the step-synchronous engine.
 *   A non-finite ℓπ becomes −Inf (src/hamiltonian.jl:95-104).                                                        */
enum { AHMC_KERNEL_HIP_FUNCTION = 0, AHMC_KERNEL_HIP_SYMBOL = 1, AHMC_KERNEL_HOST = 2 };
int32_t ahmc_set_target_plugin(ahmc_ctx* ctx, const char* plugin_so, const void* params, int64_t n_params);
int32_t ahmc_set_target_kernel(ahmc_ctx* ctx, int32_t handle_kind, void* handle, int32_t block_threads,
                               int32_t chains_per_block, void* user);

/* metric constructors + renew (src/metric.jl:31,61-69,104-117).  Unit: Minv ignored.
 * Diag: n == D (one M⁻¹ shared by all chains) or n == D*N (per-chain (D,N), F5 in SURVEY).
 * Dense: n == D*D (shared).  sqrt / Cholesky factors are recomputed here, as `renew` does.   */
int32_t ahmc_set_metric(ahmc_ctx* ctx, int32_t kind, const void* Minv, int64_t n);
int32_t ahmc_get_metric(ahmc_ctx* ctx, void* Minv_out, int64_t n);

/* Leapfrog(ϵ) with scalar (n==1) or per-chain (n==N) nominal step size
 * (src/integrator.jl:71-74, update_nom_step_size :60).                                       */
int32_t ahmc_set_stepsize(ahmc_ctx* ctx, const void* eps, int64_t n);
int32_t ahmc_get_stepsize(ahmc_ctx* ctx, void* eps_out /* T[N] */);
/* JitteredLeapfrog(ϵ0, jitter) / TemperedLeapfrog(ϵ, α): param = jitter or α               */
int32_t ahmc_set_integrator(ahmc_ctx* ctx, int32_t kind, double param);

/* Counter-based RNG (Philox4x32-10): key = seed, stream of local chain c =
 * chain_offset + chain_stride*c, per-transition counter = `iteration` (advanced by every
 * *_transition call).  Replaces the `rng` argument threaded through src/sampler.jl:159 etc.;
 * "identical seeds" (north_star) is defined on this stream, which the oracle shares
 * (SURVEY.md §8c(1)).  chain_stride = 1: independent chains (one shared `rng`, or a shard of a
 * multi-GPU run with chain_offset = first global chain).  chain_stride = 0: every chain draws
 * the same variates — the "vector of identically seeded RNGs" case of
 * test/sampler-vec.jl:69-80 and src/utilities.jl:12-23.                                      */
int32_t ahmc_seed(ahmc_ctx* ctx, uint64_t seed, uint64_t chain_offset, uint64_t chain_stride,
                  uint64_t iteration);

/* ------------------------------------------------------------------------------------------ */
/* phase point (src/hamiltonian.jl:88-139)                                                    */

/* phasepoint(h, θ, r): set θ (and r if non-NULL, else r = 0) and fill the caches ℓπ, -∇ℓπ, ℓκ
 * with the built-in target; non-finite ℓπ/ℓκ values become -Inf (:95-104).                  */
int32_t ahmc_set_position(ahmc_ctx* ctx, const void* theta, const void* r);
/* PhasePoint(θ, r, ℓπ, ℓκ) with caller-supplied caches (external target): lp T[N], grad (D,N)
 * holds -∇ℓπ.                                                                                */
int32_t ahmc_set_phasepoint(ahmc_ctx* ctx, const void* theta, const void* r, const void* lp,
                            const void* grad);
/* any output may be NULL.  theta,r,grad: (D,N); lp,lk: T[N]                                  */
int32_t ahmc_get_phasepoint(ahmc_ctx* ctx, void* theta, void* r, void* lp, void* grad, void* lk);

/* refresh(rng, FullMomentumRefreshment(), h, z) (src/hamiltonian.jl:213-220) with
 * rand_momentum (src/metric.jl:290-320); alpha in (0,1) selects PartialMomentumRefreshment(α)
 * (:243-254); alpha == 0 → full.  Uses and does NOT advance the iteration counter.           */
int32_t ahmc_refresh_momentum(ahmc_ctx* ctx, double alpha);

/* step(lf, h, z, n_steps; fwd = n_steps > 0) (src/integrator.jl:216-265) with the built-in
 * target, all steps fused in one launch; each chain stops at its own first non-finite point.  */
int32_t ahmc_leapfrog(ahmc_ctx* ctx, int64_t n_steps);

/* external-gradient split step (the two halves of src/integrator.jl:237-243 around the
 * ∂H∂θ(h, θ) callback at :241).  lf_pre: temper, r -= ϵ/2·g, θ += ϵ·M⁻¹r.  The caller then
 * evaluates (ℓπ, ∇ℓπ) at ahmc_theta_ptr() and hands them to lf_post: r -= ϵ/2·g, temper,
 * ℓκ, non-finite → -Inf.  `fwd` = 0 integrates backwards.                                    */
int32_t ahmc_lf_pre(ahmc_ctx* ctx, int32_t fwd, int64_t i, int64_t n_steps);
int32_t ahmc_lf_post(ahmc_ctx* ctx, int32_t fwd, int64_t i, int64_t n_steps, const void* lp,
                     const void* grad_neg /* -∇ℓπ (D,N) */);
/* device pointer of the context's θ (D,N) for zero-copy gradient evaluation by the caller    */
void* ahmc_theta_ptr(ahmc_ctx* ctx);

/* ------------------------------------------------------------------------------------------ */
/* transitions (src/sampler.jl:48-58 = jitter → refresh → trajectory transition)             */

/* Static HMC: transition(rng, h, HMCKernel(Trajectory{TS}(lf, FixedNSteps(L))), z)
 * (src/trajectory.jl:271-300, sample_phasepoint :336-390, mh_accept_ratio :855-880,
 * accept_phasepoint! :303-332).  TS ∈ {ENDPOINT, MULTINOMIAL}.  lambda > 0 selects
 * FixedIntegrationTime(λ): L = max(1, floor(λ/ϵ_nominal)) (:240-243; needs scalar ϵ).       */
int32_t ahmc_hmc_transition(ahmc_ctx* ctx, int64_t L, double lambda, int32_t sampler);

/* NUTS: transition(rng, h, HMCKernel(Trajectory{TS}(lf, TC(max_depth, Δ_max))), z) per chain
 * (src/trajectory.jl:677-742, build_tree :626-675), run for all N chains at once.
 * Domain: max_depth >= 1 (AHMC_ERR_ARGUMENT below; the reference takes any Int and would hand back the start point with 0/0
 * statistics).  An engine limit, not the reference's: max_depth <= 24 on the fused kernels, <= 17 on the dense / external-target
 * engine — AHMC_ERR_UNSUPPORTED beyond (the reference's default is 10).                        */
int32_t ahmc_nuts_transition(ahmc_ctx* ctx, int32_t max_depth, double delta_max,
                             int32_t criterion, int32_t sampler);

/* statistics of the last transition; `out` has N elements of the field's type               */
int32_t ahmc_get_stat(ahmc_ctx* ctx, int32_t field, void* out);

/* find_good_stepsize(rng, h, θ) (src/trajectory.jl:768-837) for every chain independently;
 * result becomes the per-chain nominal step size.                                            */
int32_t ahmc_find_good_stepsize(ahmc_ctx* ctx, double initial_step_size, int32_t max_n_iters);

/* ------------------------------------------------------------------------------------------ */
/* adaptation (src/adaptation/{stepsize,massmatrix,stan_adaptor}.jl, glue src/sampler.jl:3-22,72-90)                           */

/* construct the adaptor: kind above; δ target acceptance; (init_buffer, term_buffer,
 * window_size) = (75, 50, 25) for Stan (stan_adaptor.jl:94-103).                             */
int32_t ahmc_adaptor_init(ahmc_ctx* ctx, int32_t kind, double delta, int32_t init_buffer,
                          int32_t term_buffer, int32_t window_size);
/* adapt!(h, κ, adaptor, i, n_adapts, θ, α) (src/sampler.jl:72-90): updates the adaptor, the
 * metric and the nominal step size in place.  theta (D,N) / alpha T[N] are the position and the
 * acceptance rate to adapt on; NULL means "the context's current θ" / "the last transition's
 * acceptance_rate" (what the sample loop passes, src/sampler.jl:187).                        */
int32_t ahmc_adapt(ahmc_ctx* ctx, int64_t i, int64_t n_adapts, const void* theta,
                   const void* alpha);
/* The variance estimator behind MassMatrixAdaptor / NaiveHMCAdaptor / StanHMCAdaptor with a Diag
 * metric: WelfordVar (src/adaptation/massmatrix.jl:64-157) or NutpieVar (:160-250: Welford
 * estimators of the positions AND of the gradients, M⁻¹ = sqrt(var θ / var ∇)).  Call before
 * ahmc_adaptor_init.                                                                          */
/* AHMC_VAR_POOLED (no reference counterpart — in matrix mode the reference resizes the estimator
 * to (D,N), one per chain, massmatrix.jl:103-121): the metric stays ONE shared (D,) M⁻¹.  Every chain still runs its
 * own WelfordVar (:141-150); an update pools them (Chan's merge of equal-count partitions: μ = mean_c μ_c,
 * M = Σ_c M_c + n Σ_c (μ_c − μ)², n_tot = n·N) over the chains of this context and, when a communicator is set
 * (ahmc_comm_init / ahmc_set_comm), over all ranks — one all-gather of 2·D + 1 doubles per update (with
 * StanHMCAdaptor: per window end) — and applies get_estimation (:152-157) to the pooled (n_tot, M).              */
enum { AHMC_VAR_WELFORD = 0, AHMC_VAR_NUTPIE = 1, AHMC_VAR_POOLED = 2 };
int32_t ahmc_set_var_estimator(ahmc_ctx* ctx, int32_t estimator);
/* adapt! on an explicit phase point (src/adaptation/Adaptation.jl:24-26 PositionOrPhasePoint): as
 * ahmc_adapt plus grad (D,N) = z.ℓπ.gradient, which NutpieVar needs; with NutpieVar a theta
 * without grad is the reference's error "requires position and gradient information"
 * (massmatrix.jl:234-236).  NULLs mean the context's own state.                              */
int32_t ahmc_adapt_point(ahmc_ctx* ctx, int64_t i, int64_t n_adapts, const void* theta,
                         const void* grad, const void* alpha);
/* Stan window schedule for n_adapts (stan_adaptor.jl:13-50): writes up to cap split points,
 * returns their count in *n_splits and the window start/end.  Pure host logic.               */
int32_t ahmc_stan_windows(int32_t init_buffer, int32_t term_buffer, int32_t window_size,
                          int64_t n_adapts, int64_t* window_start, int64_t* window_end,
                          int64_t* splits, int32_t cap, int32_t* n_splits);

/* Checkpoint / resume of the adaptor (the reference's state is a value, HMCState(i, transition,
 * metric, κ, adaptor) src/abstractmcmc.jl:11-27; resuming = calling step again with it).  The metric and the nominal
 * step sizes are fields of h / κ: save and restore them with ahmc_get/set_metric and ahmc_get/set_stepsize, the
 * phase point with ahmc_get/set_phasepoint (or ahmc_set_position) — restore those FIRST, then the adaptor.  `da`:
 * T (5,N) = DAState m, ϵ, μ, x̄, H̄ (stepsize.jl:13-23; m as T); `welford`: T (n_welford, D, N) = WelfordVar μ, M, var
 * (massmatrix.jl:84-101) [+ NutpieVar's gradient estimator μ_g, M_g (:172-190)].  Either array may be NULL in get
 * (only the header is filled: call once to learn has_da / n_welford).  A run resumed this way continues bit for bit.
 * DenseEuclideanMetric adaptors (WelfordCov) are AHMC_ERR_UNSUPPORTED.                                           */
typedef struct {
  int32_t kind;           /* AHMC_ADAPT_*                                                    */
  int32_t var_estimator;  /* AHMC_VAR_*                                                      */
  int32_t init_buffer, term_buffer, window_size; /* StanHMCAdaptor(…; 75, 50, 25)            */
  int32_t adapting;       /* the adaptor has not seen iteration n_adapts yet                 */
  int32_t has_da;         /* `da` is meaningful                                              */
  int32_t n_welford;      /* 0, 3 or 5 (D,N) arrays in `welford`                             */
  double delta;           /* target acceptance rate                                          */
  int64_t stan_i;         /* StanHMCAdaptorState.i (stan_adaptor.jl:7-11)                    */
  int64_t n_adapts;       /* n_adapts the window schedule was built for (0: not yet)         */
  int64_t wv_n;           /* WelfordVar.n                                                    */
  int64_t iteration;      /* the Philox iteration counter (transitions done)                 */
} ahmc_adaptor_state;
int32_t ahmc_get_adaptor_state(ahmc_ctx* ctx, ahmc_adaptor_state* state, void* da, void* welford);
int32_t ahmc_set_adaptor_state(ahmc_ctx* ctx, const ahmc_adaptor_state* state, const void* da,
                               const void* welford);

/* ------------------------------------------------------------------------------------------ */
/* driver: sample(rng, h, κ, θ, n_samples, adaptor, n_adapts) (src/sampler.jl:159-248)        */

typedef struct {
  int32_t nuts;          /* 1: NUTS, 0: static HMC                                        */
  int32_t sampler;       /* AHMC_TS_*                                                      */
  int32_t criterion;     /* AHMC_TC_* (NUTS)                                               */
  int32_t max_depth;     /* NUTS (default 10)                                              */
  double delta_max;      /* NUTS (default 1000)                                            */
  int64_t L;             /* static HMC                                                     */
  double lambda;         /* static HMC FixedIntegrationTime, 0 = unused                    */
  double refresh_alpha;  /* 0 = FullMomentumRefreshment                                    */
} ahmc_kernel_cfg;

/* Runs n_samples transitions (+ adapt! for i <= n_adapts): the loop body of `sample`
 * (src/sampler.jl:182-228).  samples_out: NULL, or device/host buffer of (D, N, n_keep) receiving θ
 * after each kept transition (n_keep = n_samples - (drop_warmup ? n_adapts : 0)).
 * Accumulators (see ahmc_get_accum) are reset at the first kept transition — when that transition lies in this call
 * (ahmc_sample_from with i_first beyond it continues them).
 * Chains are independent, so NUTS transitions are issued in batches (AHMC_INFO_NUTS_BATCH per kernel
 * launch) — in the warm-up too, where adapt! then runs inside the kernel after every transition (step sizes
 * and per-chain Diag mass matrices never need another chain's data).  The results are bit-identical to
 * calling ahmc_nuts_transition + ahmc_adapt once per iteration.  After the call the ahmc_get_stat
 * arrays hold the LAST transition's statistics.  A HOST samples_out is filled through two device stages,
 * the D2H copy of one batch overlapping the next batch's kernel on a second stream; the copies are
 * ordered before anything enqueued on the context's stream afterwards, so ahmc_sync() covers them
 * (pinned host memory keeps the call asynchronous; a pageable buffer must stay valid until then too). */
int32_t ahmc_sample(ahmc_ctx* ctx, const ahmc_kernel_cfg* cfg, int64_t n_samples,
                    int64_t n_adapts, int32_t drop_warmup, void* samples_out);
/* The same loop from iteration i_first on (i = i_first … n_samples; ahmc_sample = i_first 1): the continuation of a
 * run that was checkpointed after iteration i_first − 1 (ahmc_get/set_adaptor_state) — also mid-warm-up, where the
 * window schedule and adapt!'s iteration argument depend on the absolute i.  samples_out is indexed by the absolute
 * iteration as in ahmc_sample (a resumed call hands in the full-size buffer or NULL).                             */
int32_t ahmc_sample_from(ahmc_ctx* ctx, const ahmc_kernel_cfg* cfg, int64_t i_first, int64_t n_samples,
                         int64_t n_adapts, int32_t drop_warmup, void* samples_out);

/* Announce a sampling run (no reference counterpart; the reference allocates inside `sample`, src/sampler.jl:159-181):
 * reserve ahead of time what ahmc_sample / ahmc_sample_from would otherwise allocate inside their first launch — the
 * momentum normals of a launch of min(AHMC_INFO_NUTS_BATCH, n_samples) NUTS transitions (up to 16 GiB, bounded by half of
 * the device memory that is free; if even that cannot be had the launch length is halved until it fits and
 * AHMC_INFO_NUTS_BATCH reports the shorter one).  Optional: without it the first call of a run reserves lazily, for its
 * own length.  Static-HMC kernels, the step-synchronous engine and ask / tell runs reserve nothing here.         */
int32_t ahmc_sample_reserve(ahmc_ctx* ctx, const ahmc_kernel_cfg* cfg, int64_t n_samples);
/* (ABI v6) The reference's MATRIX-MODE early exit, opt-in.  In vectorised mode `step` ends the integration of EVERY chain at the first
 * step after which ANY chain's phase point is non-finite — `!isfinite(z) && break` with isfinite over all columns,
 * src/integrator.jl:252-258, src/hamiltonian.jl:141-142.  By default each chain stops at its own first non-finite point
 * (the reference's scalar semantics, and what 65 536 independent chains want).  on != 0 reproduces the coupled behaviour for
 * ahmc_leapfrog and for ahmc_hmc_transition with EndPointTS — one launch per step and a flag read back after each: a mode for bit-level
 * comparisons with the sampler-vec path, not for throughput.  Not available with TemperedLeapfrog or on the dense engine
 * (AHMC_ERR_UNSUPPORTED at the call that would need it).                                                                              */
int32_t ahmc_set_ref_compat(ahmc_ctx* ctx, int32_t on);

/* running accumulators over kept transitions: Σ n_steps (all chains), number of kept
 * transitions, number of divergent transitions, per-chain Σθ and Σθ² (D,N) (may be NULL)     */
int32_t ahmc_get_accum(ahmc_ctx* ctx, int64_t* total_n_steps, int64_t* n_transitions,
                       int64_t* n_divergent, void* sum_theta, void* sumsq_theta);
int32_t ahmc_reset_accum(ahmc_ctx* ctx);
/* The accumulators as part of a checkpoint (with ahmc_get/set_adaptor_state and ahmc_get/set_phasepoint): n_transitions,
 * per-chain Σ n_steps and divergence counts (int64[N] each), Σθ, Σθ² ((D,N) of T) and the five running sums of the kept
 * transitions' energies behind ahmc_ebfmi ((5,N) of T).  Host or device pointers; any may be NULL.  A run resumed with
 * ahmc_sample_from(i_first > first kept iteration) CONTINUES the accumulators (it does not reset them), so after
 * set_accum_state in a new context ahmc_get_accum / ahmc_gather_moments / ahmc_ebfmi cover the whole run.              */
int32_t ahmc_get_accum_state(ahmc_ctx* ctx, int64_t* n_transitions, int64_t* n_steps, int64_t* n_divergent,
                             void* sum_theta, void* sumsq_theta, void* energy_sums);
int32_t ahmc_set_accum_state(ahmc_ctx* ctx, int64_t n_transitions, const int64_t* n_steps, const int64_t* n_divergent,
                             const void* sum_theta, const void* sumsq_theta, const void* energy_sums);

/* ------------------------------------------------------------------------------------------ */
/* whole transitions with an EXTERNAL target: ask / tell                                      */
/*
 * With AHMC_TARGET_EXTERNAL the user's log-density — the `h.∂ℓπ∂θ(θ)` callback of
 * src/hamiltonian.jl:45-48, i.e. LogDensityProblems.logdensity_and_gradient behind
 * src/AdvancedHMC.jl:163-186 — stays with the caller; everything else of
 * transition(rng, h, κ, z) (src/sampler.jl:48-58; static :271-300, NUTS :677-742) and of
 * find_good_stepsize (:768-837) runs in the engine.  The engine advances every running chain to
 * the point where its next leapfrog needs (ℓπ, ∇ℓπ), then hands control back:
 *
 *     ahmc_ext_begin(ctx, &cfg, n_trans);              // or ahmc_ext_find_good_stepsize_begin
 *     for (;;) {
 *       ahmc_ext_pending(ctx, &n, chains, theta);      // n == 0: finished
 *       if (n == 0) break;
 *       // caller: lp[c], grad_neg[:, c] = ℓπ(θ[:, c]), -∇ℓπ(θ[:, c]) for the n listed chains
 *       ahmc_ext_advance(ctx, lp, grad_neg);
 *     }
 *
 * Chains run asynchronously through the n_trans transitions (a chain that ends one starts its next
 * in the same call); the statistics, the phase point and the iteration counter are those of
 * n_trans calls of ahmc_*_transition.  The engine reuses the cached (ℓπ, ∇ℓπ) of the start point
 * where the reference's `refresh` re-evaluates it (src/hamiltonian.jl:213-220 → phasepoint :115-119):
 * identical for a deterministic log-density (the CPU oracle asks for that evaluation, as the
 * reference does).  HIP engine: runs on the step-synchronous engine of the dense metric — Unit /
 * Diag / Dense metric; Leapfrog / JitteredLeapfrog / TemperedLeapfrog; full or partial
 * refreshment (partial + NUTS: n_trans = 1); static EndPointTS / MultinomialTS; NUTS with MultinomialTS /
 * SliceTS and any of the three U-turn criteria; anything else is AHMC_ERR_UNSUPPORTED.
 * Any other call that changes the context between begin and the end of the loop is
 * AHMC_ERR_STATE; ahmc_ext_cancel abandons the run (the phase point is then unspecified: set it
 * again).                                                                                      */
int32_t ahmc_ext_begin(ahmc_ctx* ctx, const ahmc_kernel_cfg* cfg, int32_t n_trans);
/* find_good_stepsize(rng, h, θ) per chain (src/trajectory.jl:768-837) with the caller's ℓπ     */
int32_t ahmc_ext_find_good_stepsize_begin(ahmc_ctx* ctx, double initial_step_size,
                                          int32_t max_n_iters);
/* What the engine waits for.  *n_pending chains (0 = the run is complete); chains_out (int32[N],
 * host, may be NULL) receives their indices in unspecified order; theta_out ((D,N), host or device,
 * may be NULL) receives the positions to evaluate — only the listed columns are meaningful.  On
 * the HIP engine they are also in place at ahmc_theta_ptr(ctx) for zero-copy evaluation.       */
int32_t ahmc_ext_pending(ahmc_ctx* ctx, int64_t* n_pending, int32_t* chains_out, void* theta_out);
/* lp T[N], grad_neg (D,N) = -∇ℓπ (the sign convention of ahmc_lf_post): the entries of the pending
 * chains are read, the others ignored.  A non-finite ℓπ becomes -Inf (src/hamiltonian.jl:95-104). */
int32_t ahmc_ext_advance(ahmc_ctx* ctx, const void* lp, const void* grad_neg);
int32_t ahmc_ext_cancel(ahmc_ctx* ctx);

/* ------------------------------------------------------------------------------------------ */
/* multi-GPU: the final gather of a chain-sharded run                           */
/*
 * Chains are independent, so a sharded run exchanges nothing while sampling (each rank: its own context, Philox
 * chain_offset = its first global chain); the reference has no counterpart (no collective anywhere).
 * The only communication is at the end — and at the window ends of AHMC_VAR_POOLED — over RCCL / xGMI.
 * The communicator is either handed in (ahmc_set_comm: an ncclComm_t the host made with its own RCCL binding; the
 * library resolves ncclAllReduce / ncclAllGather from the RCCL copy already loaded in the process, so the handle and
 * the entry points match) or made here: rank 0 calls ahmc_comm_unique_id, the host broadcasts the 128 bytes by any
 * means it has, every rank calls ahmc_comm_init.  No communicator = a world of one.                            */
#define AHMC_UNIQUE_ID_BYTES 128
int32_t ahmc_comm_unique_id(void* id_out /* AHMC_UNIQUE_ID_BYTES */);
int32_t ahmc_comm_init(ahmc_ctx* ctx, const void* id, int32_t n_ranks, int32_t rank); /* owned by ctx */
int32_t ahmc_set_comm(ahmc_ctx* ctx, void* nccl_comm, int32_t n_ranks, int32_t rank);  /* caller's; NULL detaches */
/* What the communicator itself reports — measured by two small all-reduces when the communicator was attached
 * (ahmc_comm_init / ahmc_set_comm), not taken from the arguments: the number of ranks that joined (Σ 1), the chains of all
 * ranks (Σ N) and the smallest / largest N of any rank.  A run's record can so prove that n_ranks processes took part;
 * ahmc_gather_state needs min == max.  Without a communicator: 1, N, N, N.  Any pointer may be NULL.             */
int32_t ahmc_comm_info(ahmc_ctx* ctx, int64_t* ranks_seen, int64_t* chains_total, int64_t* chains_min,
                       int64_t* chains_max);
/* Pooled moments of the kept draws of ALL ranks (the accumulators of ahmc_sample): a device reduction over the
 * chains, one ncclAllReduce of 2·D + 3 doubles, then mean[D], var[D] (host doubles, may be NULL), the number of
 * draws, Σ n_steps and the number of divergent transitions.  Every rank receives the same values.              */
int32_t ahmc_gather_moments(ahmc_ctx* ctx, double* mean, double* var, int64_t* n_draws,
                            int64_t* total_n_steps, int64_t* n_divergent);
/* ncclAllGather of the positions: theta_all (DEVICE pointer, (D, N, n_ranks) elements of T) receives rank r's (D,N)
 * block at offset r·D·N.  All ranks must hold the same N.                                                       */
int32_t ahmc_gather_state(ahmc_ctx* ctx, void* theta_all);

/* ------------------------------------------------------------------------------------------ */
/* diagnostics on the device                                             */
/* EBFMI(Es) = mean(diff(Es).^2) / var(Es) (src/diagnosis.jl:1-3) per chain over the energies of the kept transitions
 * since the accumulators were last reset (ahmc_sample resets them at its first kept transition): out T[N], NaN for
 * fewer than two transitions.  Running sums kept by the transition kernels; no per-iteration read-back.         */
int32_t ahmc_ebfmi(ahmc_ctx* ctx, void* out);
/* Effective sample size of every (dimension, chain) series of `draws` = the (D, N, n_draws) buffer ahmc_sample fills
 * (device pointer): out T (D,N), host or device.  Geyer's initial monotone sequence on the autocovariances.  The
 * reference computes no ESS (MCMCChains.jl does; no reference test calls it): the definition is this engine's.  */
int32_t ahmc_ess(ahmc_ctx* ctx, const void* draws, int64_t n_draws, void* out);

/* Engine introspection (no reference counterpart; used by bench.py to price the roofline per launch
 * and by the tests to assert which thread geometry ran).                                        */
typedef enum {
  AHMC_INFO_GROUP_LANES = 0,      /* G: lanes per chain (G > 64: G/64 wavefronts per chain)        */
  AHMC_INFO_ELEMS_PER_LANE = 1,   /* E: contiguous dimensions per lane                           */
  AHMC_INFO_NUTS_LAUNCHES = 2,    /* launches of the dominant NUTS kernel since ahmc_create      */
  AHMC_INFO_NUTS_BATCH = 3,       /* transitions per NUTS launch in the sampling phase           */
  AHMC_INFO_ITERATION = 4,        /* transitions done (the Philox iteration counter)             */
  AHMC_INFO_NUTS_KERNEL_NS = 5,   /* Σ device time of those launches, ns (HIP events; synchronises) */
  AHMC_INFO_NUTS_WARM_LAUNCHES = 6,  /* the same two for the warm-up instantiation of the kernel (adapt! inside)  */
  AHMC_INFO_NUTS_WARM_KERNEL_NS = 7,
  AHMC_INFO_DENSE_GEMM_LAUNCHES = 8,        /* step-synchronous engine: launches of the 64×64-tile MFMA GEMM since ahmc_create   */
  AHMC_INFO_DENSE_GEMM_SMALL_LAUNCHES = 9,  /* … of the 64×16-tile GEMM (few columns: the tails of the batches)              */
  AHMC_INFO_DENSE_PIPELINES = 10,           /* chain pipelines (streams) of the last dense NUTS batch: 1 or 2                 */
  AHMC_INFO_DENSE_POOL = 11,                /* 1: the last dense NUTS batch ran on the point pool (k_d_tree2), 0: the copying kernel */
  AHMC_INFO_NUTS_DRAW_BATCH = 12,           /* transitions per launch the engine settled on for the sampling phase by timing its own
                                               launches (ahmc_sample, round 4); 0 while it has not settled (then AHMC_INFO_NUTS_BATCH) */
  AHMC_INFO_DENSE_EPOCH_LAUNCHES = 13,      /* launches of the chain-complete dense kernel (k_dense_epoch, round 4) since ahmc_create  */
  AHMC_INFO_STEPSIZE_SCALAR = 14            /* 1: the context holds ONE nominal step size (ahmc_set_stepsize with n = 1, not adapted per chain
                                               since) — ahmc_get_stepsize always fills N values, so a checkpoint asks here whether to hand
                                               back one (FixedIntegrationTime takes nothing else, src/trajectory.jl:241-243)          */
} ahmc_info;
int32_t ahmc_get_info(ahmc_ctx* ctx, int32_t what, int64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* AHMC_HIP_H */
