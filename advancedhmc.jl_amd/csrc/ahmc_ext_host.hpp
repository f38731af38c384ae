// ahmc_ext_host.hpp — whole transitions with an EXTERNAL target: the ask / tell calls ahmc_ext_* of
// include/ahmc_hip.h (included by ahmc_api.hip after ahmc_dense_host.hpp).
//
// The user's log-density (`h.∂ℓπ∂θ(θ)`, src/hamiltonian.jl:45-48) is evaluated by the caller, so a transition
// cannot be one fused kernel.  It does not have to be: the step-synchronous engine of ahmc_dense.hpp already
// advances every chain by ONE leapfrog per "global step" with the gradient coming from outside the tree kernel
// (there: a GEMM).  Here the outside is the caller.  One ahmc_ext_advance =
//     ingest (ℓπ, -∇ℓπ) of the pending chains  →  [dense metric: w′ = M⁻¹g′ on MFMA]  →
//     k_d_tree: second half of the leapfrog, one NUTS leaf + its merges (+ end / start of a transition),
//               first half of the next leapfrog  →  compaction of the running chains
// and the positions the next leapfrog needs are in c->th when it returns.  The kernels are those of the dense
// engine, unchanged (dense_target = 0: ℓπ is taken from the context, where ingest put it); Unit / Diag / Dense
// metric.  Static HMC (EndPointTS; MultinomialTS through the state machine of ahmc_dense_mn_host.hpp) and
// find_good_stepsize drive k_d_pre / k_d_post the same way.
#pragma once

// lp[c] ← sanitize(lp_in[c]) (PhasePoint: non-finite ℓπ → -Inf, src/hamiltonian.jl:95-104), g[:, c] ← g_in[:, c]
// for the listed chains (list == null: all n chains)
template <class T>
__global__ __launch_bounds__(256) void k_x_ingest(const T* __restrict__ lp_in, const T* __restrict__ g_in, T* __restrict__ lp, T* __restrict__ g,
                                                  int D, int64_t n, const int* __restrict__ list) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)D * n) return;
  const int64_t j = idx / D;
  const int d = (int)(idx - j * D);
  const int64_t c = list ? (int64_t)list[j] : j;
  g[c * D + d] = g_in[c * D + d];
  if (d == 0) lp[c] = sanitize(lp_in[c]);
}

template <class T>
int ext_common_checks(Ctx<T>* c, const char* what) {
  if (c->target_kind != AHMC_TARGET_EXTERNAL)
    return fail(c, AHMC_ERR_STATE, std::string(what) + ": the target is not AHMC_TARGET_EXTERNAL (built-in targets run through ahmc_*_transition / ahmc_sample)");
  if (!c->have_point) return fail(c, AHMC_ERR_STATE, std::string(what) + " before set_phasepoint");
  if (c->ext.mode != EXT_IDLE) return fail(c, AHMC_ERR_STATE, std::string(what) + ": a run is already in progress");
  return AHMC_OK;
}

// kernel arguments of the run in progress (c->iteration is constant during a NUTS batch / a find_good_stepsize
// search and advances per transition in the static-HMC mode, so they are rebuilt per call)
template <class T>
KP<T> ext_kp(Ctx<T>* c) {
  KP<T> p = make_kp(c);
  const ExtRun& x = c->ext;
  if (x.mode == EXT_NUTS) {
    p.max_depth = x.cfg.max_depth;
    p.delta_max = (T)x.cfg.delta_max;
    p.criterion = x.cfg.criterion;
    p.sampler = x.cfg.sampler;
  } else if (x.mode == EXT_HMC) {
    p.L = x.L;
  } else if (x.mode == EXT_FINDEPS) {
    p.init_eps = (T)x.fe_init;
    p.max_iters = x.fe_max;
  }
  p.accum = 0;
  p.samples_out = nullptr;
  return p;
}

template <class T>
DP<T> ext_dp(Ctx<T>* c) {
  DP<T> q = make_dp(c);
  q.n_trans = c->ext.mode == EXT_NUTS ? c->ext.n_trans : 1;
  q.list = c->ext.list;
  q.n_list = c->ext.n_list;
  return q;
}

// start of static-HMC transition c->iteration (dn_hmc_transition up to its first dn_step, with the target caches
// taken as they are: the previous transition — or set_phasepoint — left ℓπ and -∇ℓπ of θ in place)
template <class T>
int ext_hmc_start(Ctx<T>* c) {
  int rc = dn_fresh_momentum(c, c->ext.cfg.refresh_alpha, c->r);  // refresh (src/sampler.jl:54-57)
  if (rc) return rc;
  rc = dn_velocity(c);  // v = M⁻¹r, ℓκ
  if (rc) return rc;
  rc = dn_prepare_w(c);
  if (rc) return rc;
  KP<T> p = ext_kp(c);
  DP<T> q = ext_dp(c);
  hipLaunchKernelGGL((k_d_hmc_begin<T>), dim3(dn_grid_chains(c)), dim3(256), 0, c->stream, p, q);
  HIPCHK(hipGetLastError());
  c->ext.l = 0;
  if (c->ext.cfg.sampler == AHMC_TS_MULTINOMIAL) return mn_begin(c, c->ext.L, false);  // (does the first half-step itself)
  rc = dn_temper(c, 1, false, c->ext.L, c->ext.L);
  if (rc) return rc;
  return dn_pre_all(c);
}

template <class T>
int ext_begin(Ctx<T>* c, const ahmc_kernel_cfg* cfg, int n_trans) {
  if (!cfg) return fail(c, AHMC_ERR_ARGUMENT, "ext_begin: cfg is NULL");
  if (n_trans < 1) return fail(c, AHMC_ERR_ARGUMENT, "ext_begin: n_trans must be >= 1");
  int rc = ext_common_checks(c, "ext_begin");
  if (rc) return rc;
  if (cfg->refresh_alpha < 0 || cfg->refresh_alpha >= 1) return fail(c, AHMC_ERR_ARGUMENT, "ext_begin: PartialMomentumRefreshment needs 0 <= α < 1");
  if (cfg->refresh_alpha != 0 && cfg->nuts && n_trans > 1)
    return fail(c, AHMC_ERR_UNSUPPORTED, "ext_begin: with partial momentum refreshment a NUTS run is one transition (n_trans = 1): the next momentum "
                                         "depends on the one this transition ends with");
  ExtRun& x = c->ext;
  if (cfg->nuts) {
    if (cfg->sampler != AHMC_TS_MULTINOMIAL && cfg->sampler != AHMC_TS_SLICE) return fail(c, AHMC_ERR_ARGUMENT, "NUTS supports MultinomialTS and SliceTS");
    if (cfg->criterion < AHMC_TC_CLASSIC || cfg->criterion > AHMC_TC_STRICT) return fail(c, AHMC_ERR_ARGUMENT, "unknown termination criterion");
    if (cfg->max_depth < 1) return fail(c, AHMC_ERR_ARGUMENT, "ext_begin: max_depth must be >= 1");
    if (cfg->max_depth > DN_MAXLEV + 1) return fail(c, AHMC_ERR_UNSUPPORTED, "ext_begin: the step-synchronous engine supports max_depth <= 17");
    rc = dn_ensure(c, cfg->max_depth, cfg->criterion);
    if (rc) return rc;
    rc = dn_nuts_batch_momenta(c, n_trans, cfg->refresh_alpha);
    if (rc) return rc;
    x.mode = EXT_NUTS;
    x.cfg = *cfg;
    x.n_trans = n_trans;
    x.list = nullptr;
    x.n_list = c->N;
    x.pp = 0;
    x.steps = 0;
    // a transition is at most 2^max_depth − 1 leapfrogs (+ the motionless first step of the dense metric)
    x.max_steps = (int64_t)n_trans * ((1ll << cfg->max_depth) + 1) + 16;
    KP<T> p = ext_kp(c);
    DP<T> q = ext_dp(c);
    hipLaunchKernelGGL((k_d_tree_reset<T>), dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, c->stream, c->dn_S, c->dn_es, c->dn_active, c->N);
    const T* minv_d = c->metric_kind == AHMC_METRIC_DIAG ? c->minv : nullptr;
    // start of transition 0 of every chain (Unit / Diag metric: and the first half of its first leapfrog)
    launch_d_tree(c, cfg->criterion, (unsigned)c->N, p, q, minv_d, c->minv_per_chain ? 1 : 0, 0, 0);
    HIPCHK(hipGetLastError());
    return AHMC_OK;
  }
  // static HMC
  if (cfg->sampler != AHMC_TS_ENDPOINT && cfg->sampler != AHMC_TS_MULTINOMIAL)
    return fail(c, AHMC_ERR_ARGUMENT, "static HMC supports EndPointTS and MultinomialTS");
  int64_t L = cfg->L;
  if (cfg->lambda > 0) {  // nsteps(τ) for FixedIntegrationTime (src/trajectory.jl:241-243)
    rc = resolve_integration_time(c, cfg->lambda, L);
    if (rc) return rc;
  }
  if (L < 0) L = -L;
  if (L < 1) return fail(c, AHMC_ERR_ARGUMENT, "ext_begin: static HMC needs at least one leapfrog step");
  rc = dn_ensure(c, 2);
  if (rc) return rc;
  x.mode = EXT_HMC;
  x.cfg = *cfg;
  x.n_trans = n_trans;
  x.it = 0;
  x.L = L;
  x.list = nullptr;
  x.n_list = c->N;
  rc = ext_hmc_start(c);
  if (rc) x.mode = EXT_IDLE;
  return rc;
}

template <class T>
int ext_find_eps_begin(Ctx<T>* c, double init_eps, int max_iters) {
  int rc = ext_common_checks(c, "ext_find_good_stepsize_begin");
  if (rc) return rc;
  rc = dn_ensure(c, 2);
  if (rc) return rc;
  ExtRun& x = c->ext;
  x.mode = EXT_FINDEPS;
  x.fe_init = init_eps;
  x.fe_max = max_iters;
  x.fe_it = 0;
  x.fe_total = 2 * max_iters + 2;
  x.list = nullptr;
  x.n_list = c->N;
  auto bail = [&](int code) { x.mode = EXT_IDLE; return code; };
  KP<T> p = ext_kp(c);
  DP<T> q = ext_dp(c);
  hipLaunchKernelGGL((k_d_fe_save<T>), dim3(dn_grid_chains(c)), dim3(256), 0, c->stream, p, q);  // the caller's point survives the search
  rc = dn_momenta(c, 1, c->r, (T*)nullptr, (uint32_t)RNG_FINDEPS);
  if (rc) return bail(rc);
  rc = dn_velocity(c);  // (ℓπ, -∇ℓπ of θ are in place)
  if (rc) return bail(rc);
  rc = dn_prepare_w(c);
  if (rc) return bail(rc);
  hipLaunchKernelGGL((k_d_fe_begin<T>), dim3(dn_grid_chains(c)), dim3(256), 0, c->stream, p, q, c->dn_active);
  HIPCHK(hipGetLastError());
  rc = dn_pre_all(c);
  return rc ? bail(rc) : AHMC_OK;
}

template <class T>
int ext_pending(Ctx<T>* c, int64_t* n_pending, int32_t* chains_out, void* theta_out) {
  if (!n_pending) return fail(c, AHMC_ERR_ARGUMENT, "ext_pending: n_pending is NULL");
  const ExtRun& x = c->ext;
  if (x.mode == EXT_IDLE) {
    *n_pending = 0;
    return AHMC_OK;
  }
  if (chains_out) {
    if (x.list) HIPCHK(hipMemcpyAsync(chains_out, x.list, sizeof(int32_t) * (size_t)x.n_list, hipMemcpyDeviceToHost, c->stream));
    else
      for (int64_t i = 0; i < x.n_list; ++i) chains_out[i] = (int32_t)i;
  }
  if (theta_out) HIPCHK(hipMemcpyAsync(theta_out, c->th, sizeof(T) * c->D * c->N, hipMemcpyDefault, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));  // (a caller that evaluates on the device reads ahmc_theta_ptr after this)
  *n_pending = x.n_list;
  return AHMC_OK;
}

template <class T>
int ext_advance(Ctx<T>* c, const void* lp_in, const void* g_in) {
  ExtRun& x = c->ext;
  if (x.mode == EXT_IDLE) return fail(c, AHMC_ERR_STATE, "ext_advance: no run in progress");
  if (!lp_in || !g_in) return fail(c, AHMC_ERR_ARGUMENT, "ext_advance: NULL argument");
  // the caller's arrays may live on the host: stage them (persistent buffers), then ingest the pending chains
  if (!c->ext_gstage) {
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->ext_gstage), sizeof(T) * c->D * c->N));
    HIPCHK(hipMalloc(reinterpret_cast<void**>(&c->ext_lpstage), sizeof(T) * c->N));
  }
  auto on_device = [&](const void* ptr) {
    hipPointerAttribute_t at;
    const bool dev = hipPointerGetAttributes(&at, ptr) == hipSuccess && at.type == hipMemoryTypeDevice;
    (void)hipGetLastError();
    return dev;
  };
  const T* gsrc = static_cast<const T*>(g_in);
  const T* lsrc = static_cast<const T*>(lp_in);
  bool staged = false;
  if (!on_device(g_in)) {
    HIPCHK(hipMemcpyAsync(c->ext_gstage, g_in, sizeof(T) * c->D * c->N, hipMemcpyDefault, c->stream));
    gsrc = c->ext_gstage;
    staged = true;
  }
  if (!on_device(lp_in)) {
    HIPCHK(hipMemcpyAsync(c->ext_lpstage, lp_in, sizeof(T) * c->N, hipMemcpyDefault, c->stream));
    lsrc = c->ext_lpstage;
    staged = true;
  }
  // host arrays are the caller's again when this call returns: not every path below synchronises by itself
  if (staged) HIPCHK(hipStreamSynchronize(c->stream));
  hipLaunchKernelGGL((k_x_ingest<T>), dim3(dn_grid_elems(c, x.n_list)), dim3(256), 0, c->stream, lsrc, gsrc, c->lp, c->g, (int)c->D, x.n_list, x.list);
  HIPCHK(hipGetLastError());
  auto finish = [&]() {
    x.mode = EXT_IDLE;
    x.list = nullptr;
    x.n_list = 0;
  };
  int rc = AHMC_OK;
  if (x.mode == EXT_NUTS) {
    const bool dm = c->metric_kind == AHMC_METRIC_DENSE;
    const T* minv_d = c->metric_kind == AHMC_METRIC_DIAG ? c->minv : nullptr;
    T* Wcur = dm ? c->dn_W + (size_t)DS_CUR_W * c->D * c->N : nullptr;
    if (dm) {
      rc = dn_gemm(c, c->dn_minv, c->g, Wcur, x.n_list, x.list);  // w′ = M⁻¹g′
      if (rc) { finish(); return rc; }
    }
    KP<T> p = ext_kp(c);
    DP<T> q = ext_dp(c);
    launch_d_tree(c, x.cfg.criterion, (unsigned)x.n_list, p, q, minv_d, c->minv_per_chain ? 1 : 0, 0, 1);
    // the chains still running: the next request
    int* out = c->dn_list + (size_t)x.pp * c->N;
    int* cnt = c->dn_active + 1;
    HIPCHK(hipMemsetAsync(cnt, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL((k_d_compact<T>), dim3((unsigned)((x.n_list + 255) / 256)), dim3(256), 0, c->stream, c->dn_S, x.list, x.n_list, out, cnt);
    HIPCHK(hipGetLastError());
    int active = 0;
    HIPCHK(hipMemcpyAsync(&active, cnt, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->dn_global_steps += 1;
    c->dn_chain_steps += x.n_list;
    x.list = out;
    x.n_list = active;
    x.pp ^= 1;
    x.steps += 1;
    if (active == 0) {
      c->iteration += (uint64_t)x.n_trans;
      finish();
    } else if (x.steps > x.max_steps) {
      finish();
      return fail(c, AHMC_ERR_RUNTIME, "ext_advance: the NUTS batch did not terminate within its step bound");
    }
    return AHMC_OK;
  }
  if (x.mode == EXT_HMC) {
    rc = dn_post_all(c, false);
    if (rc) { finish(); return rc; }
    if (x.cfg.sampler == AHMC_TS_MULTINOMIAL) {
      rc = mn_after_step(c);  // records / counts the leapfrog and starts the next one, if any
      if (rc) { finish(); return rc; }
      if (c->mn.phase != MN_DONE) return AHMC_OK;
    } else {
      x.l += 1;
      rc = dn_temper(c, x.l, true, x.L, x.L);
      if (rc) { finish(); return rc; }
      hipLaunchKernelGGL((k_d_freeze<T>), dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, c->stream, c->lp, c->lk, c->dn_es, c->N);
      if (x.l < x.L) {
        rc = dn_temper(c, x.l + 1, false, x.L, x.L);
        if (!rc) rc = dn_pre_all(c);
        if (rc) finish();
        return rc;
      }
      KP<T> p = ext_kp(c);
      DP<T> q = ext_dp(c);
      hipLaunchKernelGGL((k_d_hmc_end<T>), dim3(dn_grid_chains(c)), dim3(256), 0, c->stream, p, q);
      HIPCHK(hipGetLastError());
    }
    c->iteration += 1;
    x.it += 1;
    if (x.it >= x.n_trans) {
      finish();
      return AHMC_OK;
    }
    rc = ext_hmc_start(c);
    if (rc) finish();
    return rc;
  }
  // find_good_stepsize: the leapfrog at the step size under test is complete; decide, rewind, next evaluation
  rc = dn_post_all(c, false);
  if (rc) { finish(); return rc; }
  KP<T> p = ext_kp(c);
  DP<T> q = ext_dp(c);
  hipLaunchKernelGGL((k_d_fe_iter<T>), dim3(dn_grid_chains(c)), dim3(256), 0, c->stream, p, q);
  HIPCHK(hipGetLastError());
  x.fe_it += 1;
  int active = 0;
  HIPCHK(hipMemcpyAsync(&active, c->dn_active, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (active <= 0 || x.fe_it >= x.fe_total) {
    hipLaunchKernelGGL((k_d_fe_end<T>), dim3(dn_grid_chains(c)), dim3(256), 0, c->stream, p, q);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->eps_nom, c->eps_cur, sizeof(T) * c->N, hipMemcpyDeviceToDevice, c->stream));
    c->eps_scalar = false;
    c->order_valid = false;
    finish();
    return AHMC_OK;
  }
  rc = dn_pre_all(c);
  if (rc) finish();
  return rc;
}
