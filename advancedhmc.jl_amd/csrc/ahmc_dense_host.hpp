// ahmc_dense_host.hpp — host side of the step-synchronous dense engine (included by ahmc_api.hip
// after Ctx / make_kp / launch_fill_caches).  See ahmc_dense.hpp for the design.
#pragma once

template <class T>
bool dense_engine(const Ctx<T>* c) {
  return c->metric_kind == AHMC_METRIC_DENSE || c->target_kind == AHMC_TARGET_DENSE_GAUSS || c->target_kind == AHMC_TARGET_KERNEL;
}

// lp[c] ← sanitize(lp[c]) for the listed chains (PhasePoint: a non-finite ℓπ → −Inf, src/hamiltonian.jl:95-104)
template <class T>
__global__ __launch_bounds__(256) void k_u_sanitize(T* __restrict__ lp, int64_t n, const int* __restrict__ list) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int64_t c = list ? (int64_t)list[j] : j;
  lp[c] = sanitize(lp[c]);
}

// AHMC_TARGET_KERNEL: (ℓπ, g = −∇ℓπ) at θ of the listed chains by the user's device kernel, launched on the context's stream —
// the `h.∂ℓπ∂θ(θ)` call of src/hamiltonian.jl:45-48 without leaving the device (signature: include/ahmc_hip.h)
template <class T>
int dn_user_target(Ctx<T>* c, const int* list, int64_t n, bool sanitize_lp = true) {  // sanitize_lp = false: the caller's next kernel sanitises ℓπ as it reads it (k_d_tree2)
  if (n <= 0) return AHMC_OK;
  if (!c->uk_handle) return fail(c, AHMC_ERR_STATE, "AHMC_TARGET_KERNEL without a kernel (ahmc_set_target_kernel)");
  const T* th = c->th;
  T* lp = c->lp;
  T* g = c->g;
  const int32_t* cols = list;
  int64_t n_cols = n, N = c->N;
  int32_t D = (int32_t)c->D;
  void* user = c->uk_user;
  void* args[] = {&th, &lp, &g, &cols, &n_cols, &D, &N, &user};
  const unsigned grid = (unsigned)((n + c->uk_cpb - 1) / c->uk_cpb);
  if (c->uk_kind == AHMC_KERNEL_HIP_FUNCTION)
    HIPCHK(hipModuleLaunchKernel(static_cast<hipFunction_t>(c->uk_handle), grid, 1, 1, (unsigned)c->uk_block, 1, 1, 0, c->stream, args, nullptr));
  else
    HIPCHK(hipLaunchKernel(c->uk_handle, dim3(grid), dim3((unsigned)c->uk_block), args, 0, c->stream));
  if (sanitize_lp) hipLaunchKernelGGL((k_u_sanitize<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->lp, n, list);
  HIPCHK(hipGetLastError());
  return AHMC_OK;
}

// (ℓπ, g) of the listed chains for a target that is not the dense Gaussian: the user's kernel, or the built-in family's /
// the plugin's group kernel (which has no chain list: it evaluates every chain)
template <class T>
int dn_other_target(Ctx<T>* c, const int* list, int64_t n, bool sanitize_lp = true) {
  return c->target_kind == AHMC_TARGET_KERNEL ? dn_user_target(c, list, n, sanitize_lp) : launch_fill_caches_builtin(c);
}

template <class T>
int dn_gemm(Ctx<T>* c, const T* A, const T* X, T* Y, int64_t ncols, const int* list = nullptr, const T* A2 = nullptr, T* Y2 = nullptr,
            const int* ptidx = nullptr, int64_t xps = 0, int64_t yps = 0, int64_t xcs = 0, int64_t ycs = 0) {  // operands in the chains' pool points (k_dgemm)
  if (xcs == 0) xcs = c->D;  // (column stride of a plain (D,N) array)
  if (ycs == 0) ycs = c->D;
  if (ncols <= 0) return AHMC_OK;
  // few columns: the 64×16-tile kernel puts 4× as many workgroups on the chip (same arithmetic per column,
  // so results do not depend on which kernel ran).  A2 / Y2: a second product on the same X in the same launch.
  const int64_t row_blocks = (c->D + GB_M - 1) / GB_M * (A2 ? 2 : 1);
  static const int64_t small_below = getenv("AHMC_GEMM_SMALL_BELOW") ? atoll(getenv("AHMC_GEMM_SMALL_BELOW")) : 1;  // measured D=512: N=512 22 vs 38 µs, N=2048 38 vs 40, N=4096 67 vs 59
  if (row_blocks * ((ncols + GB_N - 1) / GB_N) < small_below * c->n_cu) {
    dim3 grid((unsigned)row_blocks, (unsigned)((ncols + 15) / 16));
    hipLaunchKernelGGL((k_dgemm_small<T>), grid, dim3(256), 0, c->stream, A, X, Y, (int)c->D, ncols, list, A2, Y2, ptidx, xps, yps, xcs, ycs);
    HIPCHK(hipGetLastError());
    c->dn_gemm_small += 1;
    return AHMC_OK;
  }
  const int64_t cb8 = ((ncols + GB_N - 1) / GB_N + 7) / 8 * 8;  // column blocks padded to the 8 XCDs (see k_dgemm)
  dim3 grid((unsigned)(row_blocks * cb8));
  hipLaunchKernelGGL((k_dgemm<T>), grid, dim3(256), 0, c->stream, A, X, Y, (int)c->D, ncols, list, A2, Y2, ptidx, xps, yps, xcs, ycs);
  HIPCHK(hipGetLastError());
  c->dn_gemm_big += 1;
  return AHMC_OK;
}

template <class T>
DP<T> make_dp(Ctx<T>* c) {
  DP<T> q;
  q.W = c->dn_W;
  q.S = c->dn_S;
  q.es = c->dn_es;
  q.RB = c->dn_RB;
  q.VB = c->dn_VB;
  q.n_trans = 1;
  q.n_active = c->dn_active;
  q.list = nullptr;
  q.n_list = c->N;
  q.dense_metric = c->metric_kind == AHMC_METRIC_DENSE ? 1 : 0;
  return q;
}

// C = M⁻¹·P for the dense metric + dense target pair: w′ = M⁻¹g′ = (M⁻¹P)θ′ is then a second product on the SAME
// θ′ as g′ = Pθ′, and one launch serves both (one X tile read, twice the workgroups — what matters when few
// chains are still running).  Rebuilt whenever the metric or the target changes.
template <class T>
int dn_refresh_fused(Ctx<T>* c) {
  c->dn_fused_ok = false;
  if (c->metric_kind != AHMC_METRIC_DENSE || c->target_kind != AHMC_TARGET_DENSE_GAUSS || !c->dn_minv || !c->tparams) return AHMC_OK;
  if (!c->dn_C) HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->dn_C), sizeof(T) * c->D * c->D));
  int rc = dn_gemm(c, c->dn_minv, c->tparams, c->dn_C, c->D);  // P's columns as the "chains"
  if (rc) return rc;
  c->dn_fused_ok = true;
  return AHMC_OK;
}

// workspace for trees of up to max_depth doublings
template <class T>
int dn_ensure(Ctx<T>* c, int max_depth, int criterion = AHMC_TC_GENERALISED) {
  const int nlev = max_depth > 1 ? max_depth - 1 : 1;
  const size_t per_level = criterion == AHMC_TC_STRICT ? DLevel<2>::STRIDE : DLevel<1>::STRIDE;
  const size_t slots = (size_t)DS_FIXED + per_level * nlev;
  if (slots > c->dn_slots) {
    if (c->dn_W) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->dn_W)); c->dn_W = nullptr; }
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->dn_W), slots * (size_t)c->D * (size_t)c->N * sizeof(T)));
    HIPCHK(hipMemsetAsync(c->dn_W, 0, slots * (size_t)c->D * (size_t)c->N * sizeof(T), c->stream));
    c->dn_slots = slots;
  }
  if (!c->dn_S) {
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->dn_S), (size_t)c->N * sizeof(DChain<T>)));
    HIPCHK(hipMemsetAsync(c->dn_S, 0, (size_t)c->N * sizeof(DChain<T>), c->stream));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->dn_es), (size_t)c->N * sizeof(T)));
    HIPCHK(hipMemsetAsync(c->dn_es, 0, (size_t)c->N * sizeof(T), c->stream));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->dn_active), 8 * sizeof(int)));                   // [0] batch counter, [1..4] compaction counts of the pipelines
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->dn_list), 4 * (size_t)c->N * sizeof(int)));       // two ping-pong lists per pipeline
  }
  return AHMC_OK;
}

// the point pool of k_d_tree2 / k_dense_epoch for trees of up to max_depth doublings: 2·max_depth + 2 points of 5 vectors per chain
// (+ 1: the speculative half-step of k_dense_epoch), and max_depth + 2 ρ vectors (cfg4's shard, D = 512, 8 192 chains, max_depth 10:
// 3.9 GB + 0.4 GB of the 288)
template <class T>
int dn_ensure_pool(Ctx<T>* c, int max_depth, int criterion = AHMC_TC_GENERALISED) {
  // (StrictGeneralisedNoUTurn holds one more point per pending level — its last-built leaf — and the edge a doubling grew from)
  const int npt = (criterion == AHMC_TC_STRICT ? 3 * max_depth + 4 : 2 * max_depth + 3), nrho = PR_LEVEL0 + (max_depth > 1 ? max_depth : 2);
  const size_t DN = (size_t)c->D * (size_t)c->N;
  if (npt > c->dn_npt) {
    if (c->dn_P) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->dn_P)); c->dn_P = nullptr; c->dn_npt = 0; }
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->dn_P), (size_t)npt * PV_COUNT * DN * sizeof(T)));
    HIPCHK(hipMemsetAsync(c->dn_P, 0, (size_t)npt * PV_COUNT * DN * sizeof(T), c->stream));
    c->dn_npt = npt;
  }
  if (nrho > c->dn_nrho) {
    if (c->dn_R) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->dn_R)); c->dn_R = nullptr; c->dn_nrho = 0; }
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->dn_R), (size_t)nrho * DN * sizeof(T)));
    HIPCHK(hipMemsetAsync(c->dn_R, 0, (size_t)nrho * DN * sizeof(T), c->stream));
    c->dn_nrho = nrho;
  }
  if (!c->dn_S2) {
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->dn_S2), (size_t)c->N * sizeof(DChain2<T>)));
    HIPCHK(hipMemsetAsync(c->dn_S2, 0, (size_t)c->N * sizeof(DChain2<T>), c->stream));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->dn_ptcur), (size_t)c->N * sizeof(int)));
    HIPCHK(hipMemsetAsync(c->dn_ptcur, 0, (size_t)c->N * sizeof(int), c->stream));
  }
  return AHMC_OK;
}

template <class T>
unsigned dn_grid_elems(Ctx<T>* c, int64_t n = -1) { return (unsigned)(((int64_t)c->D * (n < 0 ? c->N : n) + 255) / 256); }
template <class T>
unsigned dn_grid_chains(Ctx<T>* c, int64_t n = -1) { return (unsigned)(((n < 0 ? c->N : n) + 3) / 4); }  // one wave per chain, 4 per block

// v = ∂H∂r(r) into the CUR_V slot, ℓκ = −½ r·v (src/hamiltonian.jl:50-68,155-184)
template <class T>
int dn_velocity(Ctx<T>* c, const int* list = nullptr, int64_t n = -1) {
  if (n < 0) n = c->N;
  if (n == 0) return AHMC_OK;
  T* V = c->dn_W + (size_t)DS_CUR_V * c->D * c->N;
  if (c->metric_kind == AHMC_METRIC_DENSE) {
    int rc = dn_gemm(c, c->dn_minv, c->r, V, n, list);
    if (rc) return rc;
  } else {
    hipLaunchKernelGGL((k_d_vdiag<T>), dim3(dn_grid_elems(c, n)), dim3(256), 0, c->stream, c->r,
                       c->metric_kind == AHMC_METRIC_DIAG ? c->minv : nullptr, c->minv_per_chain ? 1 : 0, V, (int)c->D, n, list);
  }
  hipLaunchKernelGGL((k_d_coldot<T>), dim3(dn_grid_chains(c, n)), dim3(256), 0, c->stream, c->r, V, c->lk, T(-0.5), (int)c->D, n, list);
  HIPCHK(hipGetLastError());
  return AHMC_OK;
}

// (ℓπ, g = −∇ℓπ) at θ
template <class T>
int dn_target(Ctx<T>* c, const int* list = nullptr, int64_t n = -1) {
  if (n < 0) n = c->N;
  if (n == 0) return AHMC_OK;
  if (c->target_kind == AHMC_TARGET_DENSE_GAUSS) {
    int rc = dn_gemm(c, c->tparams, c->th, c->g, n, list);  // g = Pθ
    if (rc) return rc;
    hipLaunchKernelGGL((k_d_coldot<T>), dim3(dn_grid_chains(c, n)), dim3(256), 0, c->stream, c->th, c->g, c->lp, T(-0.5), (int)c->D, n, list);
    HIPCHK(hipGetLastError());
    return AHMC_OK;
  }
  // built-in family: the group kernel (it also writes a Unit/Diag ℓκ, overwritten by dn_velocity); or the user's kernel
  return dn_other_target(c, list, n);
}

template <class T>
int dn_fill_caches(Ctx<T>* c) {
  int rc = dn_ensure(c, 2);
  if (rc) return rc;
  rc = dn_target(c);
  if (rc) return rc;
  return dn_velocity(c);
}

// W = M⁻¹g of the current points (dense metric): what the recurrence of dn_step starts from
template <class T>
int dn_prepare_w(Ctx<T>* c) {
  if (c->metric_kind != AHMC_METRIC_DENSE) return AHMC_OK;
  return dn_gemm(c, c->dn_minv, c->g, c->dn_W + (size_t)DS_CUR_W * c->D * c->N, c->N);
}

// One leapfrog of every listed chain with its signed step es[c] (0 = motionless: caches only).
// Dense metric: v = M⁻¹r is carried by the recurrence v ← v − ϵ/2·w, w = M⁻¹g (linear in r), so a
// step costs TWO D×D products — g′ = Pθ′ and w′ = M⁻¹g′ — not three (M⁻¹r twice + Pθ).
template <class T>
int dn_step(Ctx<T>* c, const int* list = nullptr, int64_t n = -1) {
  if (n < 0) n = c->N;
  if (n == 0) return AHMC_OK;
  const bool dm = c->metric_kind == AHMC_METRIC_DENSE;
  T* V = c->dn_W + (size_t)DS_CUR_V * c->D * c->N;
  T* W = dm ? c->dn_W + (size_t)DS_CUR_W * c->D * c->N : nullptr;
  const T* minv = c->metric_kind == AHMC_METRIC_DIAG ? c->minv : nullptr;
  const int pc = c->minv_per_chain ? 1 : 0;
  hipLaunchKernelGGL((k_d_pre<T>), dim3(dn_grid_elems(c, n)), dim3(256), 0, c->stream, c->th, c->r, c->g, V, W, minv, pc, c->dn_es, (int)c->D, n, list);
  int rc;
  const bool dt = c->target_kind == AHMC_TARGET_DENSE_GAUSS;
  if (dt) rc = dn_gemm(c, c->tparams, c->th, c->g, n, list);  // g′ = Pθ′
  else rc = dn_other_target(c, list, n);                      // built-in family: (ℓπ, g′) by the group kernel; or the user's kernel
  if (rc) return rc;
  if (dm) {
    rc = dn_gemm(c, c->dn_minv, c->g, W, n, list);  // w′ = M⁻¹g′
    if (rc) return rc;
  }
  hipLaunchKernelGGL((k_d_post<T>), dim3(dn_grid_chains(c, n)), dim3(256), 0, c->stream, c->th, c->r, c->g, V, W, minv, pc, c->dn_es, c->lp, c->lk,
                     dt ? 1 : 0, (int)c->D, n, list);
  HIPCHK(hipGetLastError());
  return AHMC_OK;
}

// fresh momenta of n_trans consecutive transitions (iterations c->iteration + k):
// R_k = U⁻¹ Z_k (Dense; rand_momentum src/metric.jl:311-320) or Z_k ./ √M⁻¹; V_k = M⁻¹ R_k
template <class T>
int dn_momenta(Ctx<T>* c, int n_trans, T* R, T* V, uint32_t purpose = RNG_MOMENTUM) {
  const size_t need = (size_t)n_trans * (size_t)c->D * (size_t)c->N;
  if (need > c->znorm_elems) {
    if (c->znorm) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(c->znorm)); }
    c->znorm = nullptr;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->znorm), need * sizeof(T)));
    c->znorm_elems = need;
  }
  KP<T> p = make_kp(c);
  const int64_t pairs = ((c->D + 1) / 2) * c->N * (int64_t)n_trans;
  const unsigned grid = (unsigned)std::min<int64_t>((pairs + 255) / 256, (int64_t)c->n_cu * 32);
  hipLaunchKernelGGL((k_normals<T>), dim3(grid), dim3(256), 0, c->stream, p, c->znorm, n_trans, purpose);
  HIPCHK(hipGetLastError());
  const int64_t cols = (int64_t)n_trans * c->N;
  if (c->metric_kind == AHMC_METRIC_DENSE) {
    int rc = dn_gemm(c, c->dn_uinv, c->znorm, R, cols);
    if (rc) return rc;
    if (V) rc = dn_gemm(c, c->dn_minv, R, V, cols);
    return rc;
  }
  const T* sq = c->metric_kind == AHMC_METRIC_DIAG ? c->sqrt_minv : nullptr;
  hipLaunchKernelGGL((k_d_rdiag<T>), dim3((unsigned)((need + 255) / 256)), dim3(256), 0, c->stream, c->znorm, sq, c->minv_per_chain ? 1 : 0, R,
                     (int)c->D, c->N, (int64_t)need);
  if (V) {
    for (int k = 0; k < n_trans; ++k)
      hipLaunchKernelGGL((k_d_vdiag<T>), dim3(dn_grid_elems(c)), dim3(256), 0, c->stream, R + (size_t)k * c->D * c->N,
                         c->metric_kind == AHMC_METRIC_DIAG ? c->minv : nullptr, c->minv_per_chain ? 1 : 0, V + (size_t)k * c->D * c->N, (int)c->D, c->N,
                         (const int*)nullptr);
  }
  HIPCHK(hipGetLastError());
  return AHMC_OK;
}

// TemperedLeapfrog on the step-synchronous engine: the two temper calls of a leapfrog (src/integrator.jl:231,241) as
// element-wise launches around the step — for the chains moving forwards / backwards through n_pos / n_neg steps
template <class T>
int dn_temper(Ctx<T>* c, int64_t i, bool second_half, int64_t n_pos, int64_t n_neg) {
  if (c->integ_kind != AHMC_INTEGRATOR_TEMPERED) return AHMC_OK;
  T* V = c->dn_W + (size_t)DS_CUR_V * c->D * c->N;
  const T sa = (T)std::sqrt((T)c->integ_param);
  hipLaunchKernelGGL((k_d_temper<T>), dim3(dn_grid_elems(c)), dim3(256), 0, c->stream, c->r, V, c->lk, c->dn_es, sa, i, second_half ? 1 : 0, n_pos, n_neg,
                     (int)c->D, c->N);
  HIPCHK(hipGetLastError());
  return AHMC_OK;
}

template <class T>
int dn_check(Ctx<T>* c, const char* what, double refresh_alpha) {
  if (c->target_kind == AHMC_TARGET_EXTERNAL)
    return fail(c, AHMC_ERR_STATE, std::string(what) + ": with AHMC_TARGET_EXTERNAL the caller evaluates the log-density: use ahmc_set_phasepoint and the "
                                                       "ask / tell calls ahmc_ext_* (or ahmc_lf_pre / ahmc_lf_post for single leapfrogs)");
  if (refresh_alpha < 0 || refresh_alpha >= 1)
    return fail(c, AHMC_ERR_ARGUMENT, std::string(what) + ": PartialMomentumRefreshment needs 0 <= α < 1");
  return AHMC_OK;
}

// buffers for the fresh momenta of n_trans transitions (and their velocities)
template <class T>
int dn_ensure_batch(Ctx<T>* c, int n_trans) {
  const size_t need = (size_t)n_trans * (size_t)c->D * (size_t)c->N;
  if (need > c->dn_batch_elems) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->dn_RB) { HIPCHK(hipFree(c->dn_RB)); HIPCHK(hipFree(c->dn_VB)); }
    c->dn_RB = c->dn_VB = nullptr;
    c->dn_batch_elems = 0;
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->dn_RB), need * sizeof(T)));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->dn_VB), need * sizeof(T)));
    c->dn_batch_elems = need;
  }
  return AHMC_OK;
}

// refresh(rng, ref, h, z) of ONE transition into `out` (c->r itself, or a batch slot): the fresh draw
// rand_momentum (src/metric.jl:290-320), mixed with the current momentum for PartialMomentumRefreshment(α)
// (src/hamiltonian.jl:243-254).  α = 0: out = ξ.
template <class T>
int dn_fresh_momentum(Ctx<T>* c, double alpha, T* out, uint32_t purpose = RNG_MOMENTUM) {
  if (alpha == 0) return dn_momenta(c, 1, out, (T*)nullptr, purpose);
  int rc = dn_ensure_batch(c, 1);
  if (rc) return rc;
  T* xi = out == c->dn_VB ? c->dn_RB : c->dn_VB;  // a batch buffer that is not the destination
  rc = dn_momenta(c, 1, xi, (T*)nullptr, purpose);
  if (rc) return rc;
  const int64_t total = c->D * c->N;
  const T a = (T)alpha, s = std::sqrt(1 - a * a);
  hipLaunchKernelGGL((k_d_partial<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, out, c->r, xi, a, s, total);
  HIPCHK(hipGetLastError());
  return AHMC_OK;
}

// set the dense metric: M⁻¹ (D,D) column-major; U = chol(M⁻¹).U and U⁻¹ on the host in double
template <class T>
int dn_set_metric(Ctx<T>* c, const T* minv_in) {
  const int64_t D = c->D;
  std::vector<T> hm((size_t)D * D);
  HIPCHK(hipMemcpy(hm.data(), minv_in, sizeof(T) * D * D, hipMemcpyDefault));
  std::vector<double> U((size_t)D * D, 0.0), Ui((size_t)D * D, 0.0);
  for (int64_t j = 0; j < D; ++j) {  // upper Cholesky factor, UᵀU = M⁻¹ (src/metric.jl:104-109)
    for (int64_t i = 0; i <= j; ++i) {
      double s = (double)hm[i + j * D];
      for (int64_t k = 0; k < i; ++k) s -= U[k + i * D] * U[k + j * D];
      if (i == j) {
        if (!(s > 0)) return fail(c, AHMC_ERR_ARGUMENT, "PosDefException: M⁻¹ is not positive definite");
        U[i + j * D] = std::sqrt(s);
      } else {
        U[i + j * D] = s / U[i + i * D];
      }
    }
  }
  for (int64_t j = 0; j < D; ++j) {  // U⁻¹ (upper triangular) by back substitution on the unit vectors
    for (int64_t i = j; i >= 0; --i) {
      double s = (i == j) ? 1.0 : 0.0;
      for (int64_t k = i + 1; k <= j; ++k) s -= U[i + k * D] * Ui[k + j * D];
      Ui[i + j * D] = s / U[i + i * D];
    }
  }
  std::vector<T> hu((size_t)D * D);
  for (size_t i = 0; i < hu.size(); ++i) hu[i] = (T)Ui[i];
  if (!c->dn_minv) {
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->dn_minv), sizeof(T) * D * D));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->dn_uinv), sizeof(T) * D * D));
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(c->dn_minv, hm.data(), sizeof(T) * D * D, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->dn_uinv, hu.data(), sizeof(T) * D * D, hipMemcpyHostToDevice));
  c->metric_kind = AHMC_METRIC_DENSE;
  c->minv_per_chain = false;
  c->minv_n = D * D;
  return dn_refresh_fused(c);
}

template <class T>
int dn_refresh(Ctx<T>* c, double alpha) {
  int rc = dn_check(c, "refresh_momentum", alpha);
  if (rc) return rc;
  rc = dn_ensure(c, 2);
  if (rc) return rc;
  rc = dn_fresh_momentum(c, alpha, c->r);
  if (rc) return rc;
  KP<T> p = make_kp(c);
  hipLaunchKernelGGL((k_d_jitter<T>), dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, c->stream, p);
  return dn_fill_caches(c);
}

template <class T>
int dn_leapfrog(Ctx<T>* c, int64_t n_steps) {
  int rc = dn_check(c, "leapfrog", 0);
  if (rc) return rc;
  rc = dn_ensure(c, 2);
  if (rc) return rc;
  const int64_t n = n_steps < 0 ? -n_steps : n_steps;
  rc = dn_velocity(c);  // v = M⁻¹r of the point as it is now (the slot may hold another edge's)
  if (rc) return rc;
  rc = dn_prepare_w(c);
  if (rc) return rc;
  hipLaunchKernelGGL((k_d_set<T>), dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, c->stream, c->dn_es, c->eps_nom, T(n_steps > 0 ? 1 : -1), c->N);
  hipLaunchKernelGGL((k_d_freeze<T>), dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, c->stream, c->lp, c->lk, c->dn_es, c->N);
  for (int64_t i = 0; i < n; ++i) {
    rc = dn_temper(c, i + 1, false, n, n);
    if (rc) return rc;
    rc = dn_step(c);
    if (rc) return rc;
    rc = dn_temper(c, i + 1, true, n, n);
    if (rc) return rc;
    hipLaunchKernelGGL((k_d_freeze<T>), dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, c->stream, c->lp, c->lk, c->dn_es, c->N);
  }
  HIPCHK(hipGetLastError());
  return AHMC_OK;
}

template <class T>
int dn_hmc_multinomial(Ctx<T>* c, int64_t L, bool accum);  // ahmc_dense_mn_host.hpp

template <class T>
int dn_hmc_transition(Ctx<T>* c, int64_t L, int sampler, double refresh_alpha, bool accum) {
  int rc = dn_check(c, "hmc_transition", refresh_alpha);
  if (rc) return rc;
  rc = dn_ensure(c, 2);
  if (rc) return rc;
  rc = dn_fresh_momentum(c, refresh_alpha, c->r);  // refresh (src/sampler.jl:54-57)
  if (rc) return rc;
  rc = dn_fill_caches(c);
  if (rc) return rc;
  KP<T> p = make_kp(c);
  p.L = L;
  p.accum = accum ? 1 : 0;
  DP<T> q = make_dp(c);
  rc = dn_prepare_w(c);
  if (rc) return rc;
  hipLaunchKernelGGL((k_d_hmc_begin<T>), dim3(dn_grid_chains(c)), dim3(256), 0, c->stream, p, q);
  if (sampler == AHMC_TS_MULTINOMIAL) {  // energies-only passes + randcat + re-integration (ahmc_dense_mn.hpp)
    HIPCHK(hipGetLastError());
    rc = dn_hmc_multinomial(c, L, accum);
    if (rc) return rc;
    c->iteration += 1;
    return AHMC_OK;
  }
  for (int64_t i = 0; i < L; ++i) {
    rc = dn_temper(c, i + 1, false, L, L);
    if (rc) return rc;
    rc = dn_step(c);
    if (rc) return rc;
    rc = dn_temper(c, i + 1, true, L, L);
    if (rc) return rc;
    hipLaunchKernelGGL((k_d_freeze<T>), dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, c->stream, c->lp, c->lk, c->dn_es, c->N);
  }
  hipLaunchKernelGGL((k_d_hmc_end<T>), dim3(dn_grid_chains(c)), dim3(256), 0, c->stream, p, q);
  HIPCHK(hipGetLastError());
  c->iteration += 1;
  return AHMC_OK;
}

// RB / VB of a NUTS batch: the fresh momenta of its n_trans transitions and v = M⁻¹r of each (partial refreshment:
// n_trans = 1, mixed with the chains' current momenta)
template <class T>
int dn_nuts_batch_momenta(Ctx<T>* c, int n_trans, double refresh_alpha) {
  int rc = dn_ensure_batch(c, n_trans);
  if (rc) return rc;
  if (refresh_alpha == 0) return dn_momenta(c, n_trans, c->dn_RB, c->dn_VB);
  rc = dn_fresh_momentum(c, refresh_alpha, c->dn_RB);
  if (rc) return rc;
  if (c->metric_kind == AHMC_METRIC_DENSE) return dn_gemm(c, c->dn_minv, c->dn_RB, c->dn_VB, c->N);
  hipLaunchKernelGGL((k_d_vdiag<T>), dim3(dn_grid_elems(c)), dim3(256), 0, c->stream, c->dn_RB, c->metric_kind == AHMC_METRIC_DIAG ? c->minv : nullptr,
                     c->minv_per_chain ? 1 : 0, c->dn_VB, (int)c->D, c->N, (const int*)nullptr);
  HIPCHK(hipGetLastError());
  return AHMC_OK;
}

// one global step of the tree state machine with the kernel built for the criterion and the integrator (the default
// — GeneralisedNoUTurn, no tempering — is k_d_tree, the measured kernel)
template <class T>
void launch_d_tree(Ctx<T>* c, int criterion, unsigned grid, const KP<T>& p, const DP<T>& q, const T* minv_d, int per_chain, int dense_target, int do_post) {
  const bool temper = c->integ_kind == AHMC_INTEGRATOR_TEMPERED;
  const int dt = dt_threads_for(c->D);
#define AHMC_LAUNCH_TREE(K) hipLaunchKernelGGL(K, dim3(grid), dim3(dt), 0, c->stream, p, q, minv_d, per_chain, dense_target, do_post)
#define AHMC_TREE_BY_DT(CR, TP)                                                        \
  do {                                                                                 \
    if (dt == 64) AHMC_LAUNCH_TREE((k_d_tree_crit<T, CR, TP, 64>));                    \
    else if (dt == 128) AHMC_LAUNCH_TREE((k_d_tree_crit<T, CR, TP, 128>));             \
    else AHMC_LAUNCH_TREE((k_d_tree_crit<T, CR, TP, 256>));                            \
  } while (0)
  if (criterion == AHMC_TC_CLASSIC) {
    if (temper) AHMC_TREE_BY_DT(0, true); else AHMC_TREE_BY_DT(0, false);
  } else if (criterion == AHMC_TC_STRICT) {
    if (temper) AHMC_TREE_BY_DT(2, true); else AHMC_TREE_BY_DT(2, false);
  } else if (temper) {
    AHMC_TREE_BY_DT(1, true);
  } else {  // the default and the measured kernel
    if (dt == 64) AHMC_LAUNCH_TREE((k_d_tree<T, 64>));
    else if (dt == 128) AHMC_LAUNCH_TREE((k_d_tree<T, 128>));
    else AHMC_LAUNCH_TREE((k_d_tree<T, 256>));
  }
#undef AHMC_TREE_BY_DT
#undef AHMC_LAUNCH_TREE
}

// n_trans NUTS transitions of every chain (asynchronous chains, see ahmc_dense.hpp)
// k_dense_epoch2's compiled shapes: (D, element type) → column tiles per workgroup (16 chains each) and the waves per SIMD the kernel is
// compiled for (its register budget is 512 / WPE).  `want_nct` ≠ 0 picks the other shape where two are compiled (experiments).
// f64: 16 accumulators of 8 registers need the 256-register budget (one 8-wave workgroup per CU), 8 fit 128 and two workgroups share a
// CU; f32 accumulators are half as wide, so 32 chains per workgroup fit the 128-register budget as well.
// Measured (profiles/r6_experiments.md, cfg4's pipeline at 8 192 chains, TFLOP/s at 4D² per leapfrog; step-synchronous kernels in brackets):
//   f64  D = 384: (6,2,2) 30.5 [23.1];  D = 512: (8,2,2) 42.9 and (8,1,4) 25.8 against round 4's k_dense_epoch 43.9 — two 16-chain workgroups per CU
//        stream the matrices twice as often and serve half as many chains per latency-bound phase: kept as the record of that experiment;
//   f32  D = 256: (4,1,3) 35.1 [21.1];  384: (6,2,3) 41.8 [29.8];  512: (8,1,4) 51.5, (8,2,2) 50.9, (8,2,4) 30.5 [34.1];  768: (12,2,3) 54.2 [42.3];
//        1 024: (16,1,4) 55.2 [44.2].
//   f64  D = 768 / 1 024 run on the step-synchronous kernels (34.6 / 37.2 TFLOP/s = 0.44 / 0.47 of the peak): sixteen waves of 8 accumulators at 128
//        registers spill 536 VGPRs and lose to them (D = 1 024: 35.7), twelve at 170 registers are refused by the ISA scan.
#define AHMC_EPOCH2_SHAPES(X) \
  X(double, 6, 2, 2) X(double, 8, 1, 4) X(double, 8, 2, 2) \
  X(float, 4, 1, 3) X(float, 6, 2, 3) X(float, 8, 1, 4) X(float, 12, 2, 3) X(float, 16, 1, 4)
// … and for ClassicNoUTurn (0) / StrictGeneralisedNoUTurn (2): D = 512 in the one-workgroup shape (the tree phase's vector passes in chunks of four
// pairs: with eight the register allocator parks spills inside the divergent tree phase and isa_check.py refuses the kernels)
#define AHMC_EPOCH2_CRIT_SHAPES(X) X(double, 8, 2, 2, 0) X(double, 8, 2, 2, 2) X(float, 8, 2, 2, 0) X(float, 8, 2, 2, 2)
template <class T>
inline bool epoch2_shape(int D, int want_nct, int& nct, int& wpe, int want_wpe = 0, int criterion = AHMC_TC_GENERALISED) {
  if (D % DE2_RW != 0) return false;
  const int nw = D / DE2_RW;
  nct = wpe = 0;
  if (criterion != AHMC_TC_GENERALISED) {
#define AHMC_E2_PICKC(TT, NW_, NCT_, WPE_, CR_) \
  if (std::is_same<T, TT>::value && nw == NW_ && criterion == CR_) { nct = NCT_; wpe = WPE_; }
    AHMC_EPOCH2_CRIT_SHAPES(AHMC_E2_PICKC)
#undef AHMC_E2_PICKC
    return nct != 0;
  }
  // the default of a (T, NW) is its LAST listed shape unless `want_nct` / `want_wpe` (AHMC_DENSE_EPOCH_NCT / _WPE) name another
#define AHMC_E2_PICK(TT, NW_, NCT_, WPE_) \
  if (std::is_same<T, TT>::value && nw == NW_ && (want_nct == 0 || want_nct == NCT_) && (want_wpe == 0 || want_wpe == WPE_)) { nct = NCT_; wpe = WPE_; }
  AHMC_EPOCH2_SHAPES(AHMC_E2_PICK)
#undef AHMC_E2_PICK
  if (!nct && (want_nct || want_wpe)) return epoch2_shape<T>(D, 0, nct, wpe, 0);
  return nct != 0;
}
template <class T>
inline void launch_epoch2(int D, int nct, int wpe, unsigned grid, hipStream_t st, const KP<T>& p, const DP2<T>& q2, const T* Asw, int steps, int criterion = AHMC_TC_GENERALISED) {
  const int nw = D / DE2_RW;
  if (criterion != AHMC_TC_GENERALISED) {
#define AHMC_E2_LAUNCHC(TT, NW_, NCT_, WPE_, CR_) \
  if constexpr (std::is_same<T, TT>::value) { \
    if (nw == NW_ && nct == NCT_ && wpe == WPE_ && criterion == CR_) { hipLaunchKernelGGL((k_dense_epoch2<T, NW_, NCT_, WPE_, CR_>), dim3(grid), dim3(64 * NW_), 0, st, p, q2, Asw, steps); return; } \
  }
    AHMC_EPOCH2_CRIT_SHAPES(AHMC_E2_LAUNCHC)
#undef AHMC_E2_LAUNCHC
    return;
  }
#define AHMC_E2_LAUNCH(TT, NW_, NCT_, WPE_) \
  if constexpr (std::is_same<T, TT>::value) { \
    if (nw == NW_ && nct == NCT_ && wpe == WPE_) { \
      static const bool dbg_ = getenv("AHMC_DEBUG") != nullptr; \
      static bool said_ = false; \
      if (dbg_ && !said_) { \
        said_ = true; \
        int occ_ = 0; \
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ_, reinterpret_cast<const void*>(&k_dense_epoch2<T, NW_, NCT_, WPE_>), 64 * NW_, 0); \
        fprintf(stderr, "[ahmc] k_dense_epoch2<%s,%d,%d,%d>: %d workgroups of %d waves per CU, grid %u\n", sizeof(T) == 8 ? "double" : "float", NW_, NCT_, WPE_, occ_, NW_, grid); \
      } \
      hipLaunchKernelGGL((k_dense_epoch2<T, NW_, NCT_, WPE_>), dim3(grid), dim3(64 * NW_), 0, st, p, q2, Asw, steps); \
      return; \
    } \
  }
  AHMC_EPOCH2_SHAPES(AHMC_E2_LAUNCH)
#undef AHMC_E2_LAUNCH
}

template <class T>
int dn_nuts_transition(Ctx<T>* c, int max_depth, double delta_max, int criterion, int sampler, double refresh_alpha, bool accum,
                       int n_trans, T* samples_dev, int64_t adapt_i0 = -1, int64_t adapt_n = 0) {  // adapt_i0 >= 0: StepSizeAdaptor inside the kernel
  int rc = dn_check(c, "nuts_transition", refresh_alpha);
  if (rc) return rc;
  if (criterion < AHMC_TC_CLASSIC || criterion > AHMC_TC_STRICT) return fail(c, AHMC_ERR_ARGUMENT, "unknown termination criterion");
  if (max_depth > DN_MAXLEV + 1) return fail(c, AHMC_ERR_UNSUPPORTED, "nuts_transition: the dense engine supports max_depth <= 17");
  if (adapt_i0 >= 0 && refresh_alpha != 0)
    return fail(c, AHMC_ERR_UNSUPPORTED, "dense engine: the in-kernel StepSizeAdaptor needs full momentum refreshment (the momenta of a batch are drawn up front)");
  if (adapt_i0 >= 0 && getenv("AHMC_DENSE_POOL") && atoi(getenv("AHMC_DENSE_POOL")) == 0)  // (the caller decides from the same variable; checked again here
    return fail(c, AHMC_ERR_UNSUPPORTED, "dense engine: AHMC_DENSE_POOL=0 selects the copying tree kernel, which has no in-kernel StepSizeAdaptor");  // because a silent unadapted warm-up is the alternative)
  if (refresh_alpha != 0 && n_trans > 1) {
    // a partially refreshed momentum depends on the momentum the previous transition ended with, so the batch's
    // momenta cannot be drawn up front: one transition per batch
    for (int k = 0; k < n_trans; ++k) {
      rc = dn_nuts_transition(c, max_depth, delta_max, criterion, sampler, refresh_alpha, accum, 1, samples_dev ? samples_dev + (size_t)k * c->D * c->N : nullptr);
      if (rc) return rc;
    }
    return AHMC_OK;
  }
  rc = dn_ensure(c, max_depth, criterion);
  if (rc) return rc;
  rc = dn_nuts_batch_momenta(c, n_trans, refresh_alpha);
  if (rc) return rc;
  KP<T> p = make_kp(c);
  p.max_depth = max_depth;
  p.delta_max = (T)delta_max;
  p.criterion = criterion;
  p.sampler = sampler;
  p.accum = accum ? 1 : 0;
  p.samples_out = samples_dev;
  DP<T> q = make_dp(c);
  q.n_trans = n_trans;
  const bool dm = c->metric_kind == AHMC_METRIC_DENSE, dt = c->target_kind == AHMC_TARGET_DENSE_GAUSS;
  const T* minv_d = c->metric_kind == AHMC_METRIC_DIAG ? c->minv : nullptr;
  const int pc = c->minv_per_chain ? 1 : 0;
  T* Wcur = dm ? c->dn_W + (size_t)DS_CUR_W * c->D * c->N : nullptr;
  // The default NUTS (GeneralisedNoUTurn, untempered) runs on the point pool (k_d_tree2: no park / candidate / edge copies);
  // the other criteria and the TemperedLeapfrog on the copying kernel.  AHMC_DENSE_POOL=0 forces the latter (A/B, tests).
  const int pool_env = getenv("AHMC_DENSE_POOL") ? atoi(getenv("AHMC_DENSE_POOL")) : 1;
  const bool pool = pool_env != 0;   // (round 6: every criterion and the TemperedLeapfrog on the pool kernels; AHMC_DENSE_POOL=0: the copying kernels)
  c->dn_last_pool = pool ? 1 : 0;
  DP2<T> q2;
  memset(&q2, 0, sizeof(q2));
  const int64_t PS = (int64_t)PV_COUNT * c->D;  // pool stride between the points of a chain; between chains: CS (set below)
  int64_t CS = 0;
  T *Pth = nullptr, *Pg = nullptr, *Pw = nullptr;
  const int dtt = dt_threads_for(c->D);
  auto launch_tree2 = [&](unsigned grid, int do_post) {
#define AHMC_TREE2(CR)                                                                                                                        \
  do {                                                                                                                                        \
    if (dtt == 64) hipLaunchKernelGGL((k_d_tree2<T, 64, CR>), dim3(grid), dim3(64), 0, c->stream, p, q2, minv_d, pc, dt ? 1 : 0, do_post);     \
    else if (dtt == 128) hipLaunchKernelGGL((k_d_tree2<T, 128, CR>), dim3(grid), dim3(128), 0, c->stream, p, q2, minv_d, pc, dt ? 1 : 0, do_post); \
    else hipLaunchKernelGGL((k_d_tree2<T, 256, CR>), dim3(grid), dim3(256), 0, c->stream, p, q2, minv_d, pc, dt ? 1 : 0, do_post);             \
  } while (0)
    if (criterion == AHMC_TC_CLASSIC) AHMC_TREE2(0);
    else if (criterion == AHMC_TC_STRICT) AHMC_TREE2(2);
    else AHMC_TREE2(1);
#undef AHMC_TREE2
  };
  if (pool) {
    rc = dn_ensure_pool(c, max_depth, criterion);
    if (rc) return rc;
    q2.P = c->dn_P; q2.R = c->dn_R; q2.S = c->dn_S2; q2.ptcur = c->dn_ptcur; q2.es = c->dn_es; q2.RB = c->dn_RB; q2.VB = c->dn_VB;
    q2.n_trans = n_trans; q2.n_pt = c->dn_npt; q2.n_rho = c->dn_nrho; q2.n_active = c->dn_active; q2.list = nullptr; q2.n_list = c->N;
    CS = (int64_t)c->dn_npt * PS;
    q2.dense_metric = dm ? 1 : 0;
    q2.staged = dt ? 0 : 1;
    if (adapt_i0 >= 0) {
      q2.adapt_ss = 1;
      q2.i0 = adapt_i0;
      q2.n_adapts = adapt_n;
      q2.delta = (T)c->da_delta; q2.gamma = T(DA_GAMMA); q2.t0 = T(DA_T0); q2.kappa = T(DA_KAPPA);  // stepsize.jl:168-172
      q2.da_m = c->da_m; q2.da_eps = c->da_eps; q2.da_mu = c->da_mu; q2.da_xbar = c->da_xbar; q2.da_Hbar = c->da_Hbar;
      q2.da_tab = c->da_tab;
    }
    Pth = c->dn_P + (size_t)PV_TH * c->D;
    Pg = c->dn_P + (size_t)PV_G * c->D;
    Pw = c->dn_P + (size_t)PV_W * c->D;
    hipLaunchKernelGGL((k_d_tree2_reset<T>), dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, c->stream, c->dn_S2, c->dn_es, c->dn_ptcur, c->dn_active, c->N);
    launch_tree2((unsigned)c->N, 0);  // start of transition 0 (and, for Unit/Diag metrics, the first half of its first leapfrog)
  } else {
    hipLaunchKernelGGL((k_d_tree_reset<T>), dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, c->stream, c->dn_S, c->dn_es, c->dn_active, c->N);
    // start of transition 0 (and, for Unit/Diag metrics, the first half of its first leapfrog)
    launch_d_tree(c, criterion, (unsigned)c->N, p, q, minv_d, pc, dt ? 1 : 0, 0);
  }
  HIPCHK(hipGetLastError());
  // global steps until every chain has finished the batch.  Every CHUNK steps the list of chains
  // still running is compacted and its length read back, so the tail of the batch (few chains with
  // long trees left) costs GEMMs over those chains only.
  //
  // Two pipelines (round 2).  A global step is a GEMM (MFMA-bound, ≈190 µs at 8 192 chains) followed by the tree kernel
  // (a chain of dependent memory round trips, ≈75 µs) — each waits for the other, so the matrix pipe idles a quarter of
  // the time.  Chains are independent: the chain set is cut in two halves, each stepping through GEMM → tree on its
  // own stream with its own running-chain list, and the hardware overlaps one half's tree kernel with the other
  // half's GEMM.  (Dense target only: the built-in families' cache kernel has no chain list.)
  //
  // AHMC_DENSE_SPLIT=1 (default) gives each half its own stream.  Nothing then keeps the halves out of phase: both GEMMs
  // are enqueued together and share the matrix pipe, then both tree kernels share HBM — which is what the measured
  // +2.5 % says happened.  AHMC_DENSE_SPLIT=2 orders the work by KIND instead: every GEMM on the context's stream (A0 B0
  // A1 B1 …), every tree kernel on the second stream (A0 B0 A1 …), with an event per half in each direction (tree k
  // waits for GEMM k; the next GEMM of that half waits for its tree kernel).  Stream order then forces the phase shift:
  // GEMM B(s) can only run beside tree A(s), GEMM A(s+1) beside tree B(s).  Same kernels on the same data: bit-identical
  // results.  Measured in round 2 (profiles/r2_cfg4_timeline_split*.json): the phase shift happens as designed and both
  // kernels slow down by the factor they now share the chip — 18.5–19.6 TFLOP/s against 24.0 for =1 and 22.2 for =0.  Kept as a switch.
  const int chunk_env = getenv("AHMC_DENSE_CHUNK") ? atoi(getenv("AHMC_DENSE_CHUNK")) : 0;  // (experiments: global steps between two compactions)
  const int CHUNK_STEP = chunk_env >= 4 && chunk_env <= 256 ? chunk_env : 16;   // step-synchronous kernels
  const int CHUNK_EPOCH = chunk_env >= 4 && chunk_env <= 256 ? chunk_env : 64;  // k_dense_epoch: one launch per chunk (cfg4: 16 / 32 / 64 steps 35.7 / 36.4 / 36.8 TFLOP/s)
  const int64_t max_steps = (int64_t)n_trans * ((1ll << max_depth) - 1) + CHUNK_EPOCH;
  // Round 4 (AHMC_DENSE_EPOCH=0 switches it off): a pipeline with at least AHMC_DENSE_EPOCH_MIN running chains takes its chunk of
  // global steps in ONE launch of k_dense_epoch (chain-complete workgroups: both products, the second half-step, the trees and
  // the next first half-step of 32 chains per workgroup); fewer chains — the tail of a batch — keep the step-synchronous kernels.
  const int epoch_env = getenv("AHMC_DENSE_EPOCH") ? atoi(getenv("AHMC_DENSE_EPOCH")) : 1;
  const int64_t epoch_min = getenv("AHMC_DENSE_EPOCH_MIN") ? atoll(getenv("AHMC_DENSE_EPOCH_MIN")) : 2048;
  // Round 6: k_dense_epoch2 serves D = 256 / 384 / 512 / 768 / 1 024 in Float64 and Float32 (AHMC_DENSE_EPOCH_V=1: round 4's k_dense_epoch,
  // D = 256 / 512 in Float64 only — kept for A/B runs).  Its shape — column tiles per workgroup, waves per SIMD it is compiled for — comes from
  // epoch2_shape (AHMC_DENSE_EPOCH_NCT=1|2 picks among the compiled ones).
  // default: round 4's kernel where it exists (f64 D = 512: 43.9 against 42.9 TFLOP/s, D = 256: 28.4 against 18.0 for four 64-row waves —
  // profiles/r6_experiments.md), k_dense_epoch2 everywhere else
  const bool v1_has = sizeof(T) == 8 && (c->D == 512 || c->D == 256) && criterion == AHMC_TC_GENERALISED;
  const int epoch_v = getenv("AHMC_DENSE_EPOCH_V") ? atoi(getenv("AHMC_DENSE_EPOCH_V")) : (v1_has ? 1 : 2);
  int e2_nct = 0, e2_wpe = 0;
  const bool epoch2_ok = epoch_v >= 2 && epoch2_shape<T>((int)c->D, getenv("AHMC_DENSE_EPOCH_NCT") ? atoi(getenv("AHMC_DENSE_EPOCH_NCT")) : 0, e2_nct, e2_wpe,
                                                                 getenv("AHMC_DENSE_EPOCH_WPE") ? atoi(getenv("AHMC_DENSE_EPOCH_WPE")) : 0, criterion);
  const bool epoch1_ok = !epoch2_ok && v1_has;
  const bool epoch_ok = epoch_env != 0 && pool && dt && dm && c->dn_fused_ok && (epoch2_ok || epoch1_ok);   // (epoch2_shape knows the criteria it has kernels for)
  const int epoch_chains = epoch2_ok ? 16 * e2_nct : DE_CHAINS;
  q2.lazy_gw = (epoch_ok && (getenv("AHMC_DENSE_LAZY_GW") ? atoi(getenv("AHMC_DENSE_LAZY_GW")) : 1)) ? 1 : 0;  // (for the whole batch: the step-synchronous kernels of its tail must not trust a record the epoch kernel skipped)
  if (epoch_ok) {
    if (!c->dn_Asw) HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->dn_Asw), 2 * sizeof(T) * (size_t)c->D * (size_t)c->D));
    const int64_t tot = 2 * c->D * c->D;
    if (epoch2_ok) hipLaunchKernelGGL((k_dense_swizzle2<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, c->tparams, c->dn_C, c->dn_Asw, (int)c->D, (int)(c->D / DE2_RW));
    else hipLaunchKernelGGL((k_dense_swizzle<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, c->tparams, c->dn_C, c->dn_Asw, (int)c->D, (int)(c->D / 128));
    HIPCHK(hipGetLastError());
  }
#ifdef AHMC_EPOCH_PROF
  static unsigned long long* prof_dev = nullptr;
  if (epoch_ok && !prof_dev) {
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&prof_dev), 8 * sizeof(unsigned long long)));
    HIPCHK(hipMemset(prof_dev, 0, 8 * sizeof(unsigned long long)));
  }
  q2.prof = prof_dev;
#endif
  auto launch_epoch = [&](hipStream_t st, int steps) {
    const unsigned grid = (unsigned)((q2.n_list + epoch_chains - 1) / epoch_chains);
    if (epoch2_ok) {
      launch_epoch2<T>((int)c->D, e2_nct, e2_wpe, grid, st, p, q2, c->dn_Asw, steps, criterion);
      c->dn_epoch_launches += 1;
      return;
    }
    if constexpr (sizeof(T) == 8) {
      if (c->D == 256) hipLaunchKernelGGL((k_dense_epoch<T, 2>), dim3(grid), dim3(64 * DE_WAVES), 0, st, p, q2, c->dn_Asw, steps);
      else hipLaunchKernelGGL((k_dense_epoch<T, 4>), dim3(grid), dim3(64 * DE_WAVES), 0, st, p, q2, c->dn_Asw, steps);
      c->dn_epoch_launches += 1;
    }
  };
  const int split_env = getenv("AHMC_DENSE_SPLIT") ? atoi(getenv("AHMC_DENSE_SPLIT")) : 1;  // (read per call: the tests toggle it)
  // AHMC_DENSE_PIPES=3|4 (with AHMC_DENSE_SPLIT=1): more, smaller pipelines — more chances for one pipeline's memory-bound tree
  // kernel to run beside another's GEMM, smaller GEMM launches
  const int pipes_env = getenv("AHMC_DENSE_PIPES") ? atoi(getenv("AHMC_DENSE_PIPES")) : 2;
  const int NP = (split_env != 0 && dt && c->N >= 2048) ? ((split_env == 1 && pipes_env >= 2 && pipes_env <= 4 && c->N >= 512 * pipes_env) ? pipes_env : 2) : 1;
  c->dn_last_pipelines = NP;
  struct Pipe { hipStream_t s; const int* list; int64_t n_list; int pp; int* lists; int* cnt; int active; };
  Pipe pipes[4];
  const int64_t per = (c->N + NP - 1) / NP;  // capacity of one running-chain list of a pipeline (two per pipeline: 2·NP·per <= 4·N ints)
  if (NP >= 2 && !c->stream2) {
    HIPCHK(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->ev_split, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  }
  for (int k = 2; k < NP; ++k)
    if (!c->stream_x[k - 2]) {
      HIPCHK(hipStreamCreateWithFlags(&c->stream_x[k - 2], hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&c->ev_join_x[k - 2], hipEventDisableTiming));
    }
  const bool by_kind = NP == 2 && split_env == 2 && dm && c->dn_fused_ok;  // (one GEMM launch per half and step)
  if (by_kind && !c->ev_gemm[0]) {
    for (int k = 0; k < 2; ++k) {
      HIPCHK(hipEventCreateWithFlags(&c->ev_gemm[k], hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&c->ev_tree[k], hipEventDisableTiming));
    }
  }
  for (int k = 0; k < NP; ++k) {
    Pipe& h = pipes[k];
    h.s = k >= 2 ? c->stream_x[k - 2] : ((k == 0 && !by_kind) ? c->stream : c->stream2);  // (by kind: the stream of the half's tree kernel, compaction and read-back)
    h.lists = c->dn_list + (size_t)k * 2 * per;
    h.cnt = c->dn_active + 1 + k;
    h.pp = 0;
    h.active = 0;
    if (NP == 1) { h.list = nullptr; h.n_list = c->N; }
    else {
      // (no pipeline takes more than `per` chains — the capacity of its two ping-pong lists: with floor(N / NP) each and the
      // remainder on the last one, N % NP >= 2 overflowed the last pipeline's list into its own other half)
      const int64_t lo = std::min<int64_t>((int64_t)k * per, c->N), n = std::min<int64_t>(per, c->N - lo);
      hipLaunchKernelGGL((k_d_iota<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, h.lists + per, (int)lo, n);  // (second buffer: the first compaction writes the first)
      h.list = h.lists + per;
      h.n_list = n;
    }
  }
  HIPCHK(hipGetLastError());
  hipStream_t main_stream = c->stream;
  if (NP >= 2) {  // everything enqueued so far (momenta, start of transition 0, the lists) precedes the other pipelines
    HIPCHK(hipEventRecord(c->ev_split, main_stream));
    HIPCHK(hipStreamWaitEvent(c->stream2, c->ev_split, 0));
    for (int k = 2; k < NP; ++k) HIPCHK(hipStreamWaitEvent(c->stream_x[k - 2], c->ev_split, 0));
  }
  auto bail = [&](int code) { c->stream = main_stream; return code; };
  bool tree_recorded[2] = {false, false};
  auto any_running = [&]() {
    for (int k = 0; k < NP; ++k)
      if (pipes[k].n_list > 0) return true;
    return false;
  };
  for (int64_t done_steps = 0; done_steps < max_steps && any_running();) {
    bool any_epoch = false;
    for (int k = 0; k < NP; ++k) any_epoch = any_epoch || (epoch_ok && pipes[k].n_list >= epoch_min);
    const int CHUNK = any_epoch ? CHUNK_EPOCH : CHUNK_STEP;  // (a pipeline that has fallen below the threshold beside one that has not steps CHUNK_EPOCH times, too)
    for (int s = 0; s < CHUNK; ++s) {
      for (int k = 0; k < NP; ++k) {
        Pipe& h = pipes[k];
        if (h.n_list <= 0) continue;
        c->stream = h.s;  // (the helpers enqueue on the context's stream)
        if (epoch_ok && h.n_list >= epoch_min) {  // the whole chunk of this pipeline in one launch
          if (s == 0) {
            q2.list = h.list;
            q2.n_list = h.n_list;
            launch_epoch(h.s, CHUNK);
            if (hipGetLastError() != hipSuccess) return bail(fail(c, AHMC_ERR_RUNTIME, "k_dense_epoch launch failed"));
          }
          continue;
        }
        q.list = h.list;
        q.n_list = h.n_list;
        // one global step = g′ = Pθ′ (or the built-in family's kernel), w′ = M⁻¹g′, then the fused
        // second-half / tree / first-half kernel
        q2.list = h.list;
        q2.n_list = h.n_list;
        const int* pti = pool ? c->dn_ptcur : nullptr;
        const T* gX = pool ? Pth : c->th;   // θ′ of the leapfrogs in flight
        T* gY = pool ? Pg : c->g;           // g′
        T* gW = pool ? Pw : Wcur;           // w′
        const int64_t ps = pool ? PS : 0, cs = pool ? CS : 0;
        if (by_kind) {
          c->stream = main_stream;
          if (tree_recorded[k] && hipStreamWaitEvent(main_stream, c->ev_tree[k], 0) != hipSuccess) return bail(fail(c, AHMC_ERR_RUNTIME, "hipStreamWaitEvent failed"));
          rc = dn_gemm(c, c->tparams, gX, gY, h.n_list, h.list, c->dn_C, gW, pti, ps, ps, cs, cs);
          if (rc) return bail(rc);
          if (hipEventRecord(c->ev_gemm[k], main_stream) != hipSuccess || hipStreamWaitEvent(h.s, c->ev_gemm[k], 0) != hipSuccess)
            return bail(fail(c, AHMC_ERR_RUNTIME, "hipEventRecord / hipStreamWaitEvent failed"));
          c->stream = h.s;
        } else if (dt && dm && c->dn_fused_ok) {
          rc = dn_gemm(c, c->tparams, gX, gY, h.n_list, h.list, c->dn_C, gW, pti, ps, ps, cs, cs);  // g′ = Pθ′ and w′ = (M⁻¹P)θ′, one launch
          if (rc) return bail(rc);
        } else {
          // (a target that is not the dense Gaussian reads θ′ from / leaves g′ in the context's arrays: the pool is "staged")
          rc = dt ? dn_gemm(c, c->tparams, gX, gY, h.n_list, h.list, (const T*)nullptr, (T*)nullptr, pti, ps, ps, cs, cs) : dn_other_target(c, h.list, h.n_list, /*sanitize_lp=*/!pool);  // (the pool kernel sanitises ℓπ itself: one launch fewer per global step)
          if (rc) return bail(rc);
          if (dm) {
            rc = dt ? dn_gemm(c, c->dn_minv, (const T*)gY, gW, h.n_list, h.list, (const T*)nullptr, (T*)nullptr, pti, ps, ps, cs, cs)
                    : dn_gemm(c, c->dn_minv, (const T*)c->g, gW, h.n_list, h.list, (const T*)nullptr, (T*)nullptr, pti, (int64_t)0, ps, (int64_t)0, cs);
            if (rc) return bail(rc);
          }
        }
        if (pool) launch_tree2((unsigned)h.n_list, 1);
        else launch_d_tree(c, criterion, (unsigned)h.n_list, p, q, minv_d, pc, dt ? 1 : 0, 1);
        if (by_kind) {
          if (hipEventRecord(c->ev_tree[k], h.s) != hipSuccess) return bail(fail(c, AHMC_ERR_RUNTIME, "hipEventRecord failed"));
          tree_recorded[k] = true;
        }
      }
    }
    done_steps += CHUNK;
    c->dn_global_steps += CHUNK;
    for (int k = 0; k < NP; ++k) {
      Pipe& h = pipes[k];
      if (h.n_list <= 0) continue;
      c->stream = h.s;
      int* out = h.lists + (size_t)h.pp * (NP == 1 ? c->N : per);
      if (hipMemsetAsync(h.cnt, 0, sizeof(int), h.s) != hipSuccess) return bail(fail(c, AHMC_ERR_RUNTIME, "hipMemsetAsync failed"));
      if (pool) hipLaunchKernelGGL((k_d_compact2<T>), dim3((unsigned)((h.n_list + 255) / 256)), dim3(256), 0, h.s, c->dn_S2, h.list, h.n_list, out, h.cnt);
      else hipLaunchKernelGGL((k_d_compact<T>), dim3((unsigned)((h.n_list + 255) / 256)), dim3(256), 0, h.s, c->dn_S, h.list, h.n_list, out, h.cnt);
      if (hipMemcpyAsync(&h.active, h.cnt, sizeof(int), hipMemcpyDeviceToHost, h.s) != hipSuccess) return bail(fail(c, AHMC_ERR_RUNTIME, "hipMemcpyAsync failed"));
    }
    for (int k = 0; k < NP; ++k) {
      Pipe& h = pipes[k];
      if (h.n_list <= 0) continue;
      if (hipStreamSynchronize(h.s) != hipSuccess) return bail(fail(c, AHMC_ERR_RUNTIME, "hipStreamSynchronize failed"));
      c->dn_chain_steps += (int64_t)CHUNK * h.n_list;
      h.list = h.lists + (size_t)h.pp * (NP == 1 ? c->N : per);
      h.n_list = h.active;
      h.pp ^= 1;
    }
  }
  c->stream = main_stream;
  if (NP >= 2) {  // whatever is enqueued on the context's stream next comes after the other pipelines
    HIPCHK(hipEventRecord(c->ev_join, c->stream2));
    HIPCHK(hipStreamWaitEvent(main_stream, c->ev_join, 0));
    for (int k = 2; k < NP; ++k) {
      HIPCHK(hipEventRecord(c->ev_join_x[k - 2], c->stream_x[k - 2]));
      HIPCHK(hipStreamWaitEvent(main_stream, c->ev_join_x[k - 2], 0));
    }
  }
#ifdef AHMC_EPOCH_PROF
  if (epoch_ok) {
    unsigned long long h[8];
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(h, prof_dev, sizeof(h), hipMemcpyDeviceToHost));
    if (h[7]) fprintf(stderr, "[ahmc] k_dense_epoch cycles per workgroup-step (thread 0): columns %.0f, products %.0f, epilogue %.0f, barrier A %.0f, trees %.0f, barrier B %.0f (%llu workgroup-steps)\n",
                      (double)h[0] / h[7], (double)h[1] / h[7], (double)h[2] / h[7], (double)h[3] / h[7], (double)h[4] / h[7], (double)h[5] / h[7], h[7]);
  }
#endif
  static const bool dbg = getenv("AHMC_DEBUG") != nullptr;
  if (dbg) fprintf(stderr, "[ahmc] dense NUTS batch of %d: %lld global steps so far, %lld chain-slots stepped\n", n_trans, (long long)c->dn_global_steps, (long long)c->dn_chain_steps);
  c->iteration += (uint64_t)n_trans;
  return AHMC_OK;
}

// find_good_stepsize per chain (src/trajectory.jl:768-837, quirk Q3 kept) on the dense engine: every
// evaluation A(ϵ) = H after ONE leapfrog from the start point is one global step; k_d_fe_iter advances
// each chain's doubling / bisection and rewinds it to the start point.
template <class T>
int dn_find_eps(Ctx<T>* c, double init_eps, int max_iters) {
  int rc = dn_check(c, "find_good_stepsize", 0);
  if (rc) return rc;
  rc = dn_ensure(c, 2);
  if (rc) return rc;
  KP<T> p = make_kp(c);
  p.init_eps = (T)init_eps;
  p.max_iters = max_iters;
  DP<T> q = make_dp(c);
  hipLaunchKernelGGL((k_d_fe_save<T>), dim3(dn_grid_chains(c)), dim3(256), 0, c->stream, p, q);  // the caller's point survives the search
  rc = dn_momenta(c, 1, c->r, (T*)nullptr, (uint32_t)RNG_FINDEPS);
  if (rc) return rc;
  rc = dn_fill_caches(c);
  if (rc) return rc;
  rc = dn_prepare_w(c);
  if (rc) return rc;
  hipLaunchKernelGGL((k_d_fe_begin<T>), dim3(dn_grid_chains(c)), dim3(256), 0, c->stream, p, q, c->dn_active);
  HIPCHK(hipGetLastError());
  const int total = 2 * max_iters + 2;
  for (int it = 0; it < total; ++it) {
    rc = dn_step(c);
    if (rc) return rc;
    hipLaunchKernelGGL((k_d_fe_iter<T>), dim3(dn_grid_chains(c)), dim3(256), 0, c->stream, p, q);
    if ((it & 7) == 7) {
      int active = 0;
      HIPCHK(hipMemcpyAsync(&active, c->dn_active, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      if (active <= 0) break;
    }
  }
  hipLaunchKernelGGL((k_d_fe_end<T>), dim3(dn_grid_chains(c)), dim3(256), 0, c->stream, p, q);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->eps_nom, c->eps_cur, sizeof(T) * c->N, hipMemcpyDeviceToDevice, c->stream));
  c->eps_scalar = false;
  return AHMC_OK;
}

// ---- WelfordCov adaptation of the shared dense metric (see ahmc_dense.hpp) ----
template <class T>
int dn_cov_init(Ctx<T>* c) {
  const size_t D = (size_t)c->D;
  if (!c->wc_M) {
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->wc_M), sizeof(T) * D * D));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->wc_S), sizeof(T) * D * D));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->wc_cov), sizeof(T) * D * D));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->wc_mu), sizeof(T) * D * (2 + COV_SLICES)));
  }
  HIPCHK(hipMemsetAsync(c->wc_M, 0, sizeof(T) * D * D, c->stream));
  HIPCHK(hipMemsetAsync(c->wc_mu, 0, sizeof(T) * D, c->stream));
  c->wc_n = 0;
  return AHMC_OK;
}
template <class T>
int dn_cov_push(Ctx<T>* c, const T* th) {
  const int D = (int)c->D;
  T* mu = c->wc_mu;
  T* mb = c->wc_mu + D;
  T* partial = c->wc_mu + 2 * D;
  hipLaunchKernelGGL((k_d_colsum_partial<T>), dim3((unsigned)((D + 255) / 256), COV_SLICES), dim3(256), 0, c->stream, th, partial, D, c->N);
  hipLaunchKernelGGL((k_d_colsum_final<T>), dim3((unsigned)((D + 255) / 256)), dim3(256), 0, c->stream, partial, mb, D, c->N);
  const unsigned tb = (unsigned)((D + GB_M - 1) / GB_M);
  hipLaunchKernelGGL((k_dsyrk<T>), dim3(tb, tb), dim3(256), 0, c->stream, th, mb, c->wc_S, D, c->N);
  const T n = (T)c->wc_n, nb = (T)c->N;
  hipLaunchKernelGGL((k_d_cov_combine<T>), dim3((unsigned)(((int64_t)D * D + 255) / 256)), dim3(256), 0, c->stream, c->wc_M, mu, mb, c->wc_S, n, nb, D);
  hipLaunchKernelGGL((k_d_cov_mean<T>), dim3((unsigned)((D + 255) / 256)), dim3(256), 0, c->stream, mu, mb, n, nb, D);
  HIPCHK(hipGetLastError());
  c->wc_n += c->N;
  return AHMC_OK;
}
// update!(wc) + update(h, adaptor): M⁻¹ ← estimate, new Cholesky factor / U⁻¹ (host, at window ends)
template <class T>
int dn_cov_update(Ctx<T>* c) {
  if (c->wc_n < c->wv_nmin) return AHMC_OK;
  const int D = (int)c->D;
  hipLaunchKernelGGL((k_d_cov_estimate<T>), dim3((unsigned)(((int64_t)D * D + 255) / 256)), dim3(256), 0, c->stream, c->wc_M, c->wc_cov, (T)c->wc_n, D);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  return dn_set_metric(c, c->wc_cov);
}
