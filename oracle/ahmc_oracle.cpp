// ahmc_oracle.cpp — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// A scalar, per-chain restatement of the hot path of AdvancedHMC.jl v0.8.6 (the reference,
// pure Julia).  It implements the C ABI of include/ahmc_hip.h so the same host code can drive
// either this checker or the HIP engine.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load it; the product package never does.
//
// PINNING.  Julia is absent from the build image, so the reference cannot be executed and no
// reference-generated numeric vectors exist ("parity unpinned" at the random-variate level:
// Julia's Xoshiro/ziggurat streams are replaced by the Philox4x32-10 streams defined below).
// What IS pinned: every known-answer / identity test the reference's own suite holds for this
// path is replayed against this file in tests/test_oracle_golden.py (fixtures in tests/golden/):
// window schedule (test/adaptation.jl:148-151), temper schedule (test/integrator.jl:89-106),
// BinaryTree combine (test/trajectory.jl:231-246), Termination table (:199-229), sampler
// combine (:143-177), energy identities (test/hamiltonian.jl:54-79), U-turn equivalences
// (test/trajectory.jl:249-325), step-loop ≡ step(n) (test/integrator.jl:17-32), the harmonic
// oscillator bound (:108-153), seed self-consistency (test/sampler-vec.jl:69-80) and the
// statistical checks (test/sampler-vec.jl:43,66).  An independent numpy restatement of the RNG-free
// formulas (tests/golden/make_independent_golden.py -> independent_golden.json) is replayed against
// this file by the same test module, and a second, independent transcription of the Julia source of the
// dynamic and static transitions (oracle/ahmc_ref.py, plain Python) must agree with this file bit for bit on
// the same Philox streams (tests/test_oracle_cross.py).
//
// Each function cites the reference lines it follows (paths relative to the AdvancedHMC.jl
// checkout).  Batch semantics: every chain is run through the reference's *scalar* (vector θ)
// code path independently; where the reference's matrix mode couples chains (Q1: whole-batch
// early exit, src/integrator.jl:252-258) the coupled behaviour is available via
// ahmco_set_ref_compat(ctx, 1) for documentation tests and is off by default.

#include "../include/ahmc_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include <sys/mman.h>   // lazily committed coroutine stacks
#include <ucontext.h>  // coroutines of the external-target ask / tell protocol (ahmc_ext_*)

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// Small-buffer vector for the per-chain D-vectors: the reference allocates a fresh array for every
// intermediate (src/integrator.jl:237-243); a malloc per leapfrog would dominate this port and make
// it scale badly over threads, so vectors up to 256 elements live inline (no heap traffic).
template <class T>
class Vec {
  static constexpr size_t INLINE = 256;
  size_t n_ = 0;
  T buf_[INLINE];
  std::vector<T> big_;
  T* p() { return n_ <= INLINE ? buf_ : big_.data(); }
  const T* p() const { return n_ <= INLINE ? buf_ : big_.data(); }

 public:
  Vec() {}
  explicit Vec(size_t n) { resize(n); }
  Vec(const Vec& o) : n_(o.n_) {
    if (n_ <= INLINE) std::copy(o.buf_, o.buf_ + n_, buf_); else big_ = o.big_;
  }
  Vec& operator=(const Vec& o) {
    if (this != &o) {
      n_ = o.n_;
      if (n_ <= INLINE) std::copy(o.buf_, o.buf_ + n_, buf_); else big_ = o.big_;
    }
    return *this;
  }
  void resize(size_t n) {
    if (n > INLINE) {
      if (n_ <= INLINE) big_.assign(buf_, buf_ + n_);
      big_.resize(n);
    } else if (n_ > INLINE) {
      std::copy(big_.begin(), big_.begin() + n, buf_);
    }
    n_ = n;
  }
  template <class It>
  void assign(It first, It last) {
    n_ = 0;
    resize((size_t)(last - first));
    std::copy(first, last, p());
  }
  size_t size() const { return n_; }
  T* data() { return p(); }
  const T* data() const { return p(); }
  T& operator[](size_t i) { return p()[i]; }
  const T& operator[](size_t i) const { return p()[i]; }
  T* begin() { return p(); }
  T* end() { return p() + n_; }
  const T* begin() const { return p(); }
  const T* end() const { return p() + n_; }
};

// ---------------------------------------------------------------------------------------------
// RNG specification shared with the HIP engine: Philox4x32-10 (Salmon et al., SC'11).
// counter = (chain, iteration, purpose, slot), key = (seed_lo, seed_hi).
// ---------------------------------------------------------------------------------------------
enum : uint32_t { RNG_MOMENTUM = 0, RNG_TRANSITION = 1, RNG_JITTER = 2, RNG_FINDEPS = 3 };
constexpr uint32_t COUPLED_CHAIN = 0xFFFFFFFFu;

struct Philox4 {
  uint32_t v[4];
};

inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                             uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  for (int round = 0; round < 10; ++round) {
    uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  return Philox4{{c0, c1, c2, c3}};
}

// 53-bit uniform in the open interval (0,1)
inline double u53(uint32_t hi, uint32_t lo) {
  uint64_t bits = ((uint64_t)(hi >> 5) << 26) | (uint64_t)(lo >> 6);
  return ((double)bits + 0.5) * (1.0 / 9007199254740992.0);
}

struct Rng {
  uint32_t k0, k1, chain, iter;
  Philox4 raw(uint32_t purpose, uint32_t slot) const {
    return philox4x32_10(chain, iter, purpose, slot, k0, k1);
  }
  double uniform(uint32_t purpose, uint32_t slot) const {
    Philox4 p = raw(purpose, slot);
    return u53(p.v[0], p.v[1]);
  }
  bool boolean(uint32_t purpose, uint32_t slot) const { return (raw(purpose, slot).v[0] >> 31) != 0; }
  double randexp(uint32_t purpose, uint32_t slot) const { return -std::log(uniform(purpose, slot)); }
  // sequential scalar draws of one NUTS transition: draw k = 32-bit word k & 3 of Philox block k >> 2;
  // uniform = (w + ½)·2⁻³² ∈ (0,1), boolean = top bit of the word
  double seq_uniform(uint32_t k) const {
    Philox4 p = raw(RNG_TRANSITION, k >> 2);
    return ((double)p.v[k & 3u] + 0.5) * 2.3283064365386962890625e-10;
  }
  bool seq_boolean(uint32_t k) const {
    Philox4 p = raw(RNG_TRANSITION, k >> 2);
    return (p.v[k & 3u] >> 31) != 0;
  }
  double seq_randexp(uint32_t k) const { return -std::log(seq_uniform(k)); }
  // standard normal for element d: Box–Muller on pair d/2
  double normal(uint32_t purpose, uint32_t d) const {
    Philox4 p = raw(purpose, d >> 1);
    double u1 = u53(p.v[0], p.v[1]), u2 = u53(p.v[2], p.v[3]);
    double rad = std::sqrt(-2.0 * std::log(u1));
    double ang = 6.283185307179586476925286766559 * u2;
    return (d & 1) ? rad * std::sin(ang) : rad * std::cos(ang);
  }
};

// Julia's min/max propagate NaN (Base.min(::Float64, ::Float64)); std::fmin does not.
template <class T>
inline T jl_min(T a, T b) {
  if (std::isnan(a) || std::isnan(b)) return std::numeric_limits<T>::quiet_NaN();
  return a < b ? a : b;
}
template <class T>
inline T jl_max(T a, T b) {
  if (std::isnan(a) || std::isnan(b)) return std::numeric_limits<T>::quiet_NaN();
  return a > b ? a : b;
}

// LogExpFunctions.logaddexp (dependency, compat "0.3, 1", not vendored):
//   Δ = x == y ? 0 : |x − y| ; max(x, y) + log1pexp(−Δ);  log1pexp(t) = log1p(exp(t)).
// Used at src/trajectory.jl:192,198.  Pinned by test/trajectory.jl:170 (log 100 ⊕ log 150 = log 250).
template <class T>
inline T logaddexp(T x, T y) {
  T d = (x == y) ? T(0) : std::abs(x - y);
  return jl_max(x, y) + std::log1p(std::exp(-d));
}

template <class T>
inline T neg_inf() {
  return -std::numeric_limits<T>::infinity();
}

// ---------------------------------------------------------------------------------------------
// Hamiltonian pieces for ONE chain
// ---------------------------------------------------------------------------------------------
template <class T>
struct Target {
  int kind = AHMC_TARGET_ISO_GAUSS;
  std::vector<T> params;
  // AHMC_TARGET_EXTERNAL: while an ahmc_ext_* run is in progress, the evaluation is a hand-over to the caller
  T (*ext_eval)(void* self, int64_t D, const T* th, T* g) = nullptr;
  void* ext_self = nullptr;
  // AHMC_TARGET_KERNEL, the checker's form (AHMC_KERNEL_HOST): the user's "kernel" is a plain C function with the kernel's
  // signature, called for one chain at a time (n_cols = 1, cols = NULL, the chain's θ / lp / grad_neg at offset 0)
  void (*kernel_host)(const T* theta, T* lp, T* grad_neg, const int32_t* cols, int64_t n_cols, int32_t D, int64_t N, void* user) = nullptr;
  void* kernel_user = nullptr;
};

// (ℓπ(θ), ∇ℓπ(θ)) — the user callback `h.∂ℓπ∂θ(θ)` of src/hamiltonian.jl:46.  `g` receives +∇ℓπ.
template <class T>
T logdensity_and_gradient(const Target<T>& tg, int64_t D, const T* th, T* g) {
  const T log2pi = T(1.8378770664093454835606594728112);
  switch (tg.kind) {
    case AHMC_TARGET_ISO_GAUSS: {
      // test/common.jl:40-44 with m = 0, s = 1: Σ -(log 2π + θ²)/2 ; gradient m .- x (:52-56)
      T v = 0;
      for (int64_t d = 0; d < D; ++d) {
        v += -(log2pi + th[d] * th[d]) / 2;
        g[d] = -th[d];
      }
      return v;
    }
    case AHMC_TARGET_DIAG_GAUSS: {
      // test/common.jl:40-44: -(log 2π + 2 log s + (m-x)²/s²)/2 ; exact gradient (m-x)/s²
      // (the test fixture's own gradient `m .- x` is only right for s = 1: SURVEY §4)
      const T* m = tg.params.data();
      const T* s = m + D;
      T v = 0;
      for (int64_t d = 0; d < D; ++d) {
        T diff = m[d] - th[d];
        v += -(log2pi + 2 * std::log(s[d]) + diff * diff / (s[d] * s[d])) / 2;
        g[d] = diff / (s[d] * s[d]);
      }
      return v;
    }
    case AHMC_TARGET_FUNNEL: {
      // Neal's funnel, research/notebooks/geweke_test.ipynb cell 4:
      //   θ1 ~ N(0, 3²);  θi ~ N(0, e^{θ1}) (variance e^{θ1}, i.e. std e^{θ1/2}), i = 2..D
      T y = th[0];
      T ss = 0;
      for (int64_t d = 1; d < D; ++d) ss += th[d] * th[d];
      T ey = std::exp(-y);  // 1/variance
      T nm1 = T(D - 1);
      T v = -(log2pi + 2 * std::log(T(3)) + y * y / 9) / 2 - nm1 * (log2pi + y) / 2 - ss * ey / 2;
      g[0] = -y / 9 - nm1 / 2 + ss * ey / 2;
      for (int64_t d = 1; d < D; ++d) g[d] = -th[d] * ey;
      return v;
    }
    case AHMC_TARGET_HIER_GAUSS: {
      // θ = (μ, log τ, x1..x_{D-2}); μ~N(0,1), logτ~N(0,1), xi~N(μ, τ²)
      T mu = th[0], lt = th[1];
      T itau2 = std::exp(-2 * lt);
      T s1 = 0, s2 = 0;
      for (int64_t d = 2; d < D; ++d) {
        T df = th[d] - mu;
        s1 += df;
        s2 += df * df;
      }
      T n = T(D - 2);
      T v = -(log2pi + mu * mu) / 2 - (log2pi + lt * lt) / 2 - n * (log2pi + 2 * lt) / 2 -
            s2 * itau2 / 2;
      g[0] = -mu + s1 * itau2;
      g[1] = -lt - n + s2 * itau2;
      for (int64_t d = 2; d < D; ++d) g[d] = -(th[d] - mu) * itau2;
      return v;
    }
    case AHMC_TARGET_DENSE_GAUSS: {
      // ℓπ = -½ θᵀPθ, ∇ = -Pθ, P (D,D) column-major symmetric precision
      const T* P = tg.params.data();
      T v = 0;
      for (int64_t i = 0; i < D; ++i) g[i] = 0;
      for (int64_t j = 0; j < D; ++j) {
        T tj = th[j];
        const T* col = P + j * D;
        for (int64_t i = 0; i < D; ++i) g[i] -= col[i] * tj;
      }
      for (int64_t i = 0; i < D; ++i) v += th[i] * g[i];
      return v / 2;
    }
    case AHMC_TARGET_KERNEL:
      if (tg.kernel_host) {  // h.∂ℓπ∂θ(θ) (src/hamiltonian.jl:45-48) as the user's function; it returns −∇ℓπ
        T lp = std::numeric_limits<T>::quiet_NaN();
        tg.kernel_host(th, &lp, g, nullptr, 1, (int32_t)D, 1, tg.kernel_user);
        for (int64_t d = 0; d < D; ++d) g[d] = -g[d];
        return lp;
      }
      [[fallthrough]];
    case AHMC_TARGET_EXTERNAL:
      if (tg.ext_eval) return tg.ext_eval(tg.ext_self, D, th, g);  // the caller's h.∂ℓπ∂θ(θ) (ask / tell)
      [[fallthrough]];  // outside an ahmc_ext_* run there is nobody to ask
    default:
      for (int64_t d = 0; d < D; ++d) g[d] = std::numeric_limits<T>::quiet_NaN();
      return std::numeric_limits<T>::quiet_NaN();
  }
}

// One chain's view of the metric (src/metric.jl:17-120).  Diag: minv/sqrt_minv point at this
// chain's D entries ((D,) shared or column c of (D,N)).  Dense: minv (D,D) col-major, chol = U
// with UᵀU = M⁻¹ (cholesky(Symmetric(M⁻¹)).U, :108).
template <class T>
struct MetricView {
  int kind;
  int64_t D;
  const T* minv;
  const T* sqrt_minv;
  const T* chol;
};

// ∂H∂r (src/hamiltonian.jl:50-68): Unit copy(r); Diag M⁻¹ .* r; Dense M⁻¹ * r
template <class T>
void dHdr(const MetricView<T>& m, const T* r, T* out) {
  const int64_t D = m.D;
  if (m.kind == AHMC_METRIC_UNIT) {
    for (int64_t d = 0; d < D; ++d) out[d] = r[d];
  } else if (m.kind == AHMC_METRIC_DIAG) {
    for (int64_t d = 0; d < D; ++d) out[d] = m.minv[d] * r[d];
  } else {
    for (int64_t i = 0; i < D; ++i) out[i] = 0;
    for (int64_t j = 0; j < D; ++j) {
      const T* col = m.minv + j * D;
      for (int64_t i = 0; i < D; ++i) out[i] += col[i] * r[j];
    }
  }
}

// neg_energy(h, r, θ) = ℓκ = -K(r) (src/hamiltonian.jl:155-184)
template <class T>
T neg_kinetic(const MetricView<T>& m, const T* r) {
  const int64_t D = m.D;
  T s = 0;
  if (m.kind == AHMC_METRIC_UNIT) {
    for (int64_t d = 0; d < D; ++d) s += r[d] * r[d];  // -sum(abs2, r)/2
  } else if (m.kind == AHMC_METRIC_DIAG) {
    for (int64_t d = 0; d < D; ++d) s += (r[d] * r[d]) * m.minv[d];  // -sum(abs2.(r) .* M⁻¹)/2
  } else {
    Vec<T> tmp(D);
    dHdr(m, r, tmp.data());  // mul!(_temp, M⁻¹, r); -dot(r, _temp)/2
    for (int64_t d = 0; d < D; ++d) s += r[d] * tmp[d];
  }
  return -s / 2;
}

// PhasePoint (src/hamiltonian.jl:88-107).  `g` is ℓπ.gradient = -∇ℓπ (:45-48).  ℓκ.gradient
// (∂H∂r) is never read after construction except by isfinite, so it is recomputed on demand.
template <class T>
struct PhasePoint {
  Vec<T> th, r, g;
  T lp = 0, lk = 0;
};

// the value sanitation of the PhasePoint constructor (:95-104): non-finite values → -Inf
template <class T>
inline T sanitize(T v) {
  return std::isfinite(v) ? v : neg_inf<T>();
}

// isfinite(z) (src/hamiltonian.jl:141-142): values AND gradients of ℓπ and ℓκ all finite
template <class T>
bool phasepoint_isfinite(const MetricView<T>& m, const PhasePoint<T>& z) {
  if (!std::isfinite(z.lp) || !std::isfinite(z.lk)) return false;
  for (T v : z.g)
    if (!std::isfinite(v)) return false;
  Vec<T> kr(m.D);
  dHdr(m, z.r.data(), kr.data());
  for (T v : kr)
    if (!std::isfinite(v)) return false;
  return true;
}

template <class T>
inline T energy(const PhasePoint<T>& z) {  // H = -(ℓπ + ℓκ)  (src/hamiltonian.jl:149,194)
  return -(z.lp + z.lk);
}

// phasepoint(h, θ, r) (src/hamiltonian.jl:115-119): evaluates ∂H∂θ and the kinetic cache
template <class T>
PhasePoint<T> make_phasepoint(const Target<T>& tg, const MetricView<T>& m, const T* th, const T* r) {
  PhasePoint<T> z;
  z.th.assign(th, th + m.D);
  z.r.assign(r, r + m.D);
  z.g.resize(m.D);
  T v = logdensity_and_gradient(tg, m.D, th, z.g.data());
  for (auto& x : z.g) x = -x;  // DualValue(res[1], -res[2])
  z.lp = sanitize(v);
  z.lk = sanitize(neg_kinetic(m, r));
  return z;
}

// integrator parameters of one chain (src/integrator.jl:71-74, :112-123, :174-179)
template <class T>
struct LeapfrogCfg {
  int kind = AHMC_INTEGRATOR_LEAPFROG;
  T eps = T(0.1);   // current (possibly jittered) step size
  T alpha = T(1);   // TemperedLeapfrog temperature
};

// temper (src/integrator.jl:198-209); identity for the other integrators (:52-56)
template <class T, class V>
inline void temper(const LeapfrogCfg<T>& lf, V& r, int64_t i, bool is_half, int64_t n_steps) {
  if (lf.kind != AHMC_INTEGRATOR_TEMPERED) return;
  int64_t i_temper = 2 * (i - 1) + 1 + (is_half ? 0 : 1);
  T s = std::sqrt(lf.alpha);
  if (i_temper <= n_steps)
    for (auto& x : r) x = x * s;
  else
    for (auto& x : r) x = x / s;
}

// step(lf, h, z, n_steps; fwd, full_trajectory) (src/integrator.jl:216-265), scalar chain.
// Returns the last point; if `traj` is non-null every intermediate point is appended to it.
template <class T>
PhasePoint<T> leapfrog_step(const LeapfrogCfg<T>& lf, const Target<T>& tg, const MetricView<T>& m,
                            const PhasePoint<T>& z0, int64_t n_steps_signed, bool fwd,
                            std::vector<PhasePoint<T>>* traj = nullptr, int64_t i0 = 0,
                            int64_t n_total = -1) {
  const int64_t D = m.D;
  int64_t n_steps = n_steps_signed < 0 ? -n_steps_signed : n_steps_signed;  // :220
  T eps = fwd ? lf.eps : -lf.eps;                                           // :222
  PhasePoint<T> z = z0;
  Vec<T> th = z0.th, r = z0.r, g = z0.g, kr(D);
  T value = z0.lp;
  // (i0, n_total) let a caller run one step of a longer tempered trajectory (ref_compat mode)
  const int64_t n_temper = n_total < 0 ? n_steps : n_total;
  for (int64_t i = 1; i <= n_steps; ++i) {
    temper(lf, r, i0 + i, true, n_temper);                         // :231
    for (int64_t d = 0; d < D; ++d) r[d] = r[d] - eps / 2 * g[d];  // :233
    dHdr(m, r.data(), kr.data());                                  // :235
    for (int64_t d = 0; d < D; ++d) th[d] = th[d] + eps * kr[d];   // :236
    value = logdensity_and_gradient(tg, D, th.data(), g.data());   // :238 ∂H∂θ
    for (int64_t d = 0; d < D; ++d) g[d] = -g[d];
    for (int64_t d = 0; d < D; ++d) r[d] = r[d] - eps / 2 * g[d];  // :239
    temper(lf, r, i0 + i, false, n_temper);                        // :241
    z.th = th; z.r = r; z.g = g;                                   // :243 phasepoint(h, θ, r; ℓπ)
    z.lp = sanitize(value);
    z.lk = sanitize(neg_kinetic(m, r.data()));
    if (traj) traj->push_back(z);
    if (!phasepoint_isfinite(m, z)) break;                         // :248-255
  }
  return z;
}

// rand_momentum (src/metric.jl:290-320) for one chain: z ~ N(0, I); Diag z ./ sqrtM⁻¹;
// Dense cholM⁻¹ \ z (upper-triangular back substitution)
template <class T>
void rand_momentum(const Rng& rng, uint32_t purpose, const MetricView<T>& m, T* r) {
  const int64_t D = m.D;
  for (int64_t d = 0; d < D; ++d) r[d] = (T)rng.normal(purpose, (uint32_t)d);
  if (m.kind == AHMC_METRIC_DIAG) {
    for (int64_t d = 0; d < D; ++d) r[d] = r[d] / m.sqrt_minv[d];
  } else if (m.kind == AHMC_METRIC_DENSE) {
    for (int64_t i = D - 1; i >= 0; --i) {
      T s = r[i];
      for (int64_t j = i + 1; j < D; ++j) s -= m.chol[i + j * D] * r[j];
      r[i] = s / m.chol[i + i * D];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// NUTS (src/trajectory.jl:454-742), literal recursive form
// ---------------------------------------------------------------------------------------------
struct Termination {  // :480-507
  bool dynamic = false, numerical = false;
};
inline Termination operator*(Termination a, Termination b) {
  return Termination{a.dynamic || b.dynamic, a.numerical || b.numerical};
}
inline bool isterminated(Termination t) { return t.dynamic || t.numerical; }

// ---- decision margins (round 6; the checker's own instrument, no counterpart in the reference) ----
// Every data-dependent decision of a transition is a comparison `a < b` of two floating-point quantities.  Another correct
// implementation (other summation order, FMA contraction, a last-ulp libm difference) may take the other branch ONLY where the two
// sides are within rounding of each other.  The oracle therefore records, per chain, the smallest RELATIVE distance |a − b| / scale
// any decision had since the record was last reset (`ahmco_decision_margin`), and the parity tests allow a chain to differ from
// the oracle only if that distance is below a stated bound — everything else must match exactly.  `scale` is the magnitude the
// rounding error of the comparison scales with: Σ|ρ_d·v_d| for a U-turn dot product, max(1, |H0|, |H′|) for energy tests,
// n for the slice sampler's integer rule.  A comparison with a non-finite side is decided by the non-finiteness: not recorded.
inline void note_margin(double* acc, double a, double b, double scale) {
  if (!acc || !std::isfinite(a) || !std::isfinite(b) || !std::isfinite(scale)) return;
  const double m = std::abs(a - b) / (scale > 1e-300 ? scale : 1e-300);
  if (m < *acc) *acc = m;
}

template <class T>
inline T maxabs(T a, T b) {  // :526
  return std::abs(a) > std::abs(b) ? a : b;
}

template <class T>
using PRef = std::shared_ptr<const PhasePoint<T>>;

template <class T>
struct BinaryTree {  // :512-520
  PRef<T> zleft, zright;  // references, like the Julia structs (no array copies on combine)
  Vec<T> rho;  // TurnStatistic (:454-467); unused (empty) for ClassicNoUTurn
  T sum_alpha = 0;
  int64_t n_alpha = 0;
  T dH_max = 0;
};

template <class T>
BinaryTree<T> combine(const BinaryTree<T>& l, const BinaryTree<T>& r) {  // :533-542
  BinaryTree<T> t;
  t.zleft = l.zleft;
  t.zright = r.zright;
  t.rho.resize(l.rho.size());
  for (size_t d = 0; d < l.rho.size(); ++d) t.rho[d] = l.rho[d] + r.rho[d];  // :467
  t.sum_alpha = l.sum_alpha + r.sum_alpha;
  t.n_alpha = l.n_alpha + r.n_alpha;
  t.dH_max = maxabs(l.dH_max, r.dH_max);
  return t;
}

// tree sampler state: SliceTS (ℓu, n) / MultinomialTS (ℓw)  (:102-136)
template <class T>
struct Sampler {
  PRef<T> zcand;
  T lw = 0;      // Multinomial: log total weight
  T lu = 0;      // Slice: log slice variable
  int64_t n = 0; // Slice: number of acceptable candidates
};

template <class T>
struct NutsCfg {
  int sampler = AHMC_TS_MULTINOMIAL;
  int criterion = AHMC_TC_GENERALISED;
  int max_depth = 10;
  T delta_max = T(1000);
};

template <class T>
struct NutsEnv {
  const LeapfrogCfg<T>* lf;
  const Target<T>* tg;
  const MetricView<T>* m;
  const NutsCfg<T>* cfg;
  const Rng* rng;
  uint32_t draw = 0;  // sequential index of the scalar draws of this transition
  double* margin = nullptr;  // per-chain decision-margin record (note_margin), or none
  double h_scale = 1.0;      // max(1, |H0|) of the transition: the scale of its energy comparisons' rounding error
};

template <class T>
inline T dot(const Vec<T>& a, const Vec<T>& b) {
  T s = 0;
  for (size_t d = 0; d < a.size(); ++d) s += a[d] * b[d];
  return s;
}

template <class T>
inline double absdot(const Vec<T>& a, const Vec<T>& b) {  // Σ|a_d·b_d|: what the rounding error of dot(a, b) scales with
  double s = 0;
  for (size_t d = 0; d < a.size(); ++d) s += std::abs((double)a[d] * (double)b[d]);
  return s;
}

template <class T>
inline bool generalised_uturn_criterion(const Vec<T>& rho, const Vec<T>& pm,
                                        const Vec<T>& pp, double* margin = nullptr) {  // :619-621
  const T dm = dot(rho, pm), dp = dot(rho, pp);
  note_margin(margin, (double)dm, 0.0, absdot(rho, pm));
  note_margin(margin, (double)dp, 0.0, absdot(rho, pp));
  return (dm <= 0) || (dp <= 0);
}

template <class T>
Vec<T> dHdr_vec(const MetricView<T>& m, const Vec<T>& r) {
  Vec<T> o(r.size());
  dHdr(m, r.data(), o.data());
  return o;
}

// isterminated(tc, h, t, tleft, tright) for the three criteria (:551-623)
template <class T>
Termination uturn(const NutsEnv<T>& e, const BinaryTree<T>& t, const BinaryTree<T>& tl,
                  const BinaryTree<T>& tr) {
  const MetricView<T>& m = *e.m;
  if (e.cfg->criterion == AHMC_TC_CLASSIC) {  // :551-557
    const int64_t D = m.D;
    Vec<T> dth(D), ndth(D), nr0(D);
    for (int64_t d = 0; d < D; ++d) {
      dth[d] = t.zright->th[d] - t.zleft->th[d];
      ndth[d] = -dth[d];
      nr0[d] = -t.zleft->r[d];
    }
    const Vec<T> v0 = dHdr_vec(m, nr0), v1 = dHdr_vec(m, t.zright->r);
    const T d0 = dot(dth, v0), d1 = dot(ndth, v1);
    note_margin(e.margin, (double)d0, 0.0, absdot(dth, v0));
    note_margin(e.margin, (double)d1, 0.0, absdot(ndth, v1));
    bool s = (d0 >= 0) || (d1 >= 0);
    return Termination{s, false};
  }
  bool s1 = generalised_uturn_criterion(t.rho, dHdr_vec(m, t.zleft->r), dHdr_vec(m, t.zright->r), e.margin);  // :566-570
  if (e.cfg->criterion == AHMC_TC_GENERALISED) return Termination{s1, false};
  // Strict (:579-617)
  Vec<T> rho2(m.D), rho3(m.D);
  for (int64_t d = 0; d < m.D; ++d) {
    rho2[d] = tl.rho[d] + tr.zleft->r[d];   // check_left_subtree :597-601
    rho3[d] = tl.zright->r[d] + tr.rho[d];  // check_right_subtree :609-615
  }
  bool s2 = generalised_uturn_criterion(rho2, dHdr_vec(m, t.zleft->r), dHdr_vec(m, tr.zleft->r), e.margin);
  bool s3 = generalised_uturn_criterion(rho3, dHdr_vec(m, tl.zright->r), dHdr_vec(m, t.zright->r), e.margin);
  return Termination{s1, false} * Termination{s2, false} * Termination{s3, false};
}

// sampler for a single-leaf tree (:163-176)
template <class T>
Sampler<T> leaf_sampler(const NutsEnv<T>& e, const Sampler<T>& s, T H0, const PRef<T>& z) {
  Sampler<T> o;
  o.zcand = z;
  if (e.cfg->sampler == AHMC_TS_SLICE) {
    o.lu = s.lu;
    o.n = (s.lu <= -energy(*z)) ? 1 : 0;  // Int(s.ℓu <= neg_energy(zcand))
    note_margin(e.margin, (double)s.lu, (double)-energy(*z), std::max(1.0, std::max(std::abs((double)H0), std::abs((double)s.lu))));
  } else {
    o.lw = H0 + (-energy(*z));  // H0 + neg_energy(zcand)
  }
  return o;
}

// combine(rng, s1, s2): uniform progressive sampling inside a subtree (:178-195)
template <class T>
Sampler<T> combine_rng(NutsEnv<T>& e, const Sampler<T>& s1, const Sampler<T>& s2) {
  Sampler<T> o;
  if (e.cfg->sampler == AHMC_TS_SLICE) {
    o.n = s1.n + s2.n;
    o.lu = s1.lu;
    T u = (T)e.rng->seq_uniform(e.draw++);
    o.zcand = (T(o.n) * u < T(s1.n)) ? s1.zcand : s2.zcand;
    if (o.n > 0) note_margin(e.margin, (double)(T(o.n) * u), (double)T(s1.n), (double)o.n);   // (n = 0: 0 < 0, exact on any implementation)
  } else {
    o.lw = logaddexp(s1.lw, s2.lw);
    T ex = (T)e.rng->seq_randexp(e.draw++);
    o.zcand = (o.lw < s1.lw + ex) ? s1.zcand : s2.zcand;
    // (log weights are energy DIFFERENCES H0 − H′: their rounding error scales with the energies, |H0| — passed in by the caller)
    note_margin(e.margin, (double)o.lw, (double)(s1.lw + ex), e.h_scale);
  }
  return o;
}

// Termination(sampler, nt, H0, H′): divergence test (:500-507)
template <class T>
Termination leaf_termination(const NutsEnv<T>& e, const Sampler<T>& s, T H0, T Hp) {
  const double sc = std::max(std::max(e.h_scale, std::abs((double)Hp)), std::abs((double)e.cfg->delta_max));
  if (e.cfg->sampler == AHMC_TS_SLICE) {
    note_margin(e.margin, (double)s.lu, (double)(e.cfg->delta_max + -Hp), sc);
    return Termination{false, !(s.lu < e.cfg->delta_max + -Hp)};
  }
  note_margin(e.margin, (double)-H0, (double)(e.cfg->delta_max + -Hp), sc);
  return Termination{false, !(-H0 < e.cfg->delta_max + -Hp)};
}

template <class T>
struct BuildResult {
  BinaryTree<T> tree;
  Sampler<T> sampler;
  Termination term;
};

// build_tree (:626-675)
template <class T>
BuildResult<T> build_tree(NutsEnv<T>& e, const PRef<T>& z, const Sampler<T>& sampler, int v,
                          int j, T H0) {
  if (j == 0) {
    // base case: one leapfrog step in direction v (:638-647)
    PRef<T> zpp = std::make_shared<PhasePoint<T>>(leapfrog_step(*e.lf, *e.tg, *e.m, *z, v, v > 0));
    const PhasePoint<T>& zp = *zpp;
    T Hp = energy(zp);
    T dH = Hp - H0;
    T alpha = std::exp(jl_min(T(0), -dH));
    BuildResult<T> out;
    out.sampler = leaf_sampler(e, sampler, H0, zpp);
    out.tree.zleft = zpp;
    out.tree.zright = zpp;
    if (e.cfg->criterion != AHMC_TC_CLASSIC) out.tree.rho = zp.r;  // TurnStatistic(tc, z′)
    out.tree.sum_alpha = alpha;
    out.tree.n_alpha = 1;
    out.tree.dH_max = dH;
    out.term = leaf_termination(e, out.sampler, H0, Hp);
    return out;
  }
  BuildResult<T> first = build_tree(e, z, sampler, v, j - 1, H0);  // :650
  if (!isterminated(first.term)) {
    BuildResult<T> second;
    const BinaryTree<T>*tl, *tr;
    if (v == -1) {
      second = build_tree(e, first.tree.zleft, sampler, v, j - 1, H0);  // :655-658
      tl = &second.tree;
      tr = &first.tree;
    } else {
      second = build_tree(e, first.tree.zright, sampler, v, j - 1, H0);  // :661-664
      tl = &first.tree;
      tr = &second.tree;
    }
    BuildResult<T> out;
    out.tree = combine(*tl, *tr);                                  // :666
    out.sampler = combine_rng(e, first.sampler, second.sampler);   // :667
    out.term = first.term * second.term * uturn(e, out.tree, *tl, *tr);  // :668-671
    return out;
  }
  return first;
}

// per-chain transition statistics (src/trajectory.jl:286-298, :726-739)
template <class T>
struct TStat {
  int32_t n_steps = 0, is_accept = 0, tree_depth = 0, numerical_error = 0;
  T acceptance_rate = 0, log_density = 0, hamiltonian_energy = 0, hamiltonian_energy_error = 0,
    max_hamiltonian_energy_error = 0;
};

// dynamic transition (:677-742)
template <class T>
PhasePoint<T> nuts_transition(NutsEnv<T>& e, const PhasePoint<T>& z0v, TStat<T>& st) {
  PRef<T> z0p = std::make_shared<PhasePoint<T>>(z0v);
  const PhasePoint<T>& z0 = *z0p;
  T H0 = energy(z0);
  e.h_scale = std::isfinite((double)H0) ? std::max(1.0, std::abs((double)H0)) : 1.0;
  BinaryTree<T> tree;
  tree.zleft = z0p;
  tree.zright = z0p;
  if (e.cfg->criterion != AHMC_TC_CLASSIC) tree.rho = z0.r;
  tree.sum_alpha = 0;
  tree.n_alpha = 0;
  tree.dH_max = 0;
  Sampler<T> sampler;  // TS(rng, z0) (:144-155)
  sampler.zcand = z0p;
  if (e.cfg->sampler == AHMC_TS_SLICE) {
    sampler.lu = -energy(z0) - (T)e.rng->seq_randexp(e.draw++);
    sampler.n = 1;
  } else {
    sampler.lw = 0;
  }
  Termination term;
  PRef<T> zcand = z0p;
  int j = 0;
  while (!isterminated(term) && j < e.cfg->max_depth) {
    bool vleft = e.rng->seq_boolean(e.draw++);  // :693
    BuildResult<T> sub;
    BinaryTree<T> tl, tr;
    if (vleft) {
      sub = build_tree(e, tree.zleft, sampler, -1, j, H0);
      tl = sub.tree;
      tr = tree;
    } else {
      sub = build_tree(e, tree.zright, sampler, 1, j, H0);
      tl = tree;
      tr = sub.tree;
    }
    if (!isterminated(sub.term)) {  // :708-713
      j = j + 1;
      bool acc;
      if (e.cfg->sampler == AHMC_TS_SLICE) {
        T u = (T)e.rng->seq_uniform(e.draw++);
        acc = T(sampler.n) * u < T(sub.sampler.n);  // :202
        if (sampler.n > 0) note_margin(e.margin, (double)(T(sampler.n) * u), (double)T(sub.sampler.n), (double)sampler.n);
      } else {
        T ex = (T)e.rng->seq_randexp(e.draw++);
        acc = sampler.lw < sub.sampler.lw + ex;  // :203-206
        note_margin(e.margin, (double)sampler.lw, (double)(sub.sampler.lw + ex), e.h_scale);
      }
      if (acc) zcand = sub.sampler.zcand;
    }
    tree = combine(tl, tr);  // :715
    // combine(zcand, sampler, sampler′) (:183-187, :197-200)
    if (e.cfg->sampler == AHMC_TS_SLICE)
      sampler.n = sampler.n + sub.sampler.n;
    else
      sampler.lw = logaddexp(sampler.lw, sub.sampler.lw);
    sampler.zcand = zcand;
    term = term * sub.term * uturn(e, tree, tl, tr);  // :719-722
  }
  T H = energy(*zcand);
  st.n_steps = (int32_t)tree.n_alpha;
  st.is_accept = 1;
  st.acceptance_rate = tree.sum_alpha / T(tree.n_alpha);
  st.log_density = zcand->lp;
  st.hamiltonian_energy = H;
  st.hamiltonian_energy_error = H - H0;
  st.max_hamiltonian_energy_error = tree.dH_max;
  st.tree_depth = j;
  st.numerical_error = term.numerical ? 1 : 0;
  return *zcand;
}

// ---------------------------------------------------------------------------------------------
// Stan window schedule (src/adaptation/stan_adaptor.jl:13-50)
// ---------------------------------------------------------------------------------------------
struct StanWindows {
  int64_t window_start = 0, window_end = 0;
  std::vector<int64_t> splits;
};

StanWindows stan_windows(int64_t init_buffer, int64_t term_buffer, int64_t window_size, int64_t n_adapts) {
  StanWindows w;
  w.window_start = init_buffer + 1;
  w.window_end = n_adapts - term_buffer;
  int64_t next_window = init_buffer + window_size;
  while (next_window <= w.window_end) {
    int64_t next_window_boundary = next_window + 2 * window_size;
    if (next_window_boundary > w.window_end) next_window = w.window_end;
    w.splits.push_back(next_window);
    window_size *= 2;
    next_window += window_size;
  }
  if (!w.splits.empty() && w.splits.back() == n_adapts) w.splits.pop_back();
  return w;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
struct CtxBase {
  virtual ~CtxBase() {}
  std::string err;
  int dtype = AHMC_F64;
};

static thread_local std::string g_create_err;

namespace {

template <class T>
struct Ctx : CtxBase {
  int64_t D = 0, N = 0;
  // phase point of every chain, (D,N) column-major
  std::vector<T> th, r, g, lp, lk;
  bool have_point = false;
  // Hamiltonian
  Target<T> target;
  int metric_kind = AHMC_METRIC_UNIT;
  bool metric_per_chain = false;
  std::vector<T> minv, sqrt_minv, chol;
  // integrator
  int integ_kind = AHMC_INTEGRATOR_LEAPFROG;
  T integ_param = 0;
  std::vector<T> eps_nom, eps_cur;  // (N,)
  bool eps_scalar = true;
  // rng
  uint64_t seed = 0, chain_offset = 0, chain_stride = 1, iteration = 0;
  // stats of the last transition
  std::vector<TStat<T>> stat;
  // smallest relative margin of any decision a chain took since the record was last reset (note_margin, ahmco_decision_margin)
  std::vector<double> margin;
  // adaptation
  int adapt_kind = AHMC_ADAPT_NONE;
  T da_delta = T(0.8), da_gamma = T(0.05), da_t0 = T(10), da_kappa = T(0.75);
  std::vector<int32_t> da_m;
  std::vector<T> da_eps, da_mu, da_xbar, da_Hbar;
  int64_t wv_n = 0, wv_nmin = 10;
  std::vector<T> wv_mu, wv_M, wv_var;
  int var_estimator = AHMC_VAR_WELFORD;  // WelfordVar or NutpieVar (massmatrix.jl:160-250)
  std::vector<T> wg_mu, wg_M;             // NutpieVar: the gradient estimator's Welford state
  // WelfordCov (massmatrix.jl:283-340) behind a DenseEuclideanMetric.  The metric is shared by all chains (this
  // engine's extension), so is the estimator: every adapt! pushes the N chains' positions one after another.
  int64_t wc_n = 0;
  std::vector<T> wc_mu, wc_M, wc_cov;
  int stan_init = 75, stan_term = 50, stan_window = 25;
  int64_t stan_i = 0;
  StanWindows windows;
  // accumulators
  int64_t acc_nsteps = 0, acc_ntrans = 0, acc_ndiv = 0;
  std::vector<int64_t> acc_nsteps_c, acc_ndiv_c;  // per chain (the checkpoint's form, ahmc_get/set_accum_state)
  std::vector<T> acc_sum, acc_sumsq;
  std::vector<T> acc_energy;  // (5,N): n, E_prev, Σ(ΔE)², mean(E), M2(E) over the kept transitions (ahmc_ebfmi)
  int64_t windows_n_adapts = 0;
  bool adapting = false;
  // cross-rank hook of the CPU checker (there is no RCCL here): the multi-rank tests hand in an all-gather over gloo
  int (*xgather)(const double* mine, double* all, int64_t count, void* user) = nullptr;
  void* xgather_user = nullptr;
  int x_ranks = 1, x_rank = 0;
  bool ref_compat = false;
  // External target, ask / tell (ahmc_ext_*): every chain runs the ordinary scalar code of this file inside its own
  // coroutine; the target evaluation of an AHMC_TARGET_EXTERNAL context parks the coroutine until the caller has
  // supplied (ℓπ, -∇ℓπ) for the position it published.  One chain runs at a time (no OpenMP in this mode).
  struct ExtCo {
    ucontext_t uc;
    void* stack = nullptr;  // mmap'd, committed page by page as the coroutine touches it (N stacks would not fit otherwise)
    size_t stack_bytes = 0;
    int state = 0;  // 0 runnable, 1 waiting for the caller's evaluation, 2 finished
    int k = 0;      // transition of the batch this chain is in
    ExtCo() = default;
    ExtCo(const ExtCo&) = delete;
    ExtCo& operator=(const ExtCo&) = delete;
    ~ExtCo() {
      if (stack) munmap(stack, stack_bytes);
    }
  };
  struct Ext {
    int mode = 0;  // 0 idle, 1 NUTS, 2 static HMC, 3 find_good_stepsize
    ahmc_kernel_cfg cfg{};
    int n_trans = 0;
    uint64_t iter0 = 0;
    double fe_init = 0;
    int fe_iters = 0;
    ucontext_t main_uc;
    std::unique_ptr<ExtCo[]> co;  // N coroutines; never moved once made (a ucontext_t points into itself)
    int64_t n_co = 0;
    int64_t cur = -1;
    std::vector<T> theta;  // (D,N): the positions the waiting chains want evaluated
    const T* in_lp = nullptr;
    const T* in_g = nullptr;
    std::vector<T> fe_out;
  } ext;

  MetricView<T> metric_view(int64_t c) const {
    MetricView<T> m;
    m.kind = metric_kind;
    m.D = D;
    m.minv = nullptr;
    m.sqrt_minv = nullptr;
    m.chol = nullptr;
    if (metric_kind == AHMC_METRIC_DIAG) {
      int64_t off = metric_per_chain ? c * D : 0;
      m.minv = minv.data() + off;
      m.sqrt_minv = sqrt_minv.data() + off;
    } else if (metric_kind == AHMC_METRIC_DENSE) {
      m.minv = minv.data();
      m.chol = chol.data();
    }
    return m;
  }
  Rng rng(int64_t c, uint64_t iter) const {
    Rng g;
    g.k0 = (uint32_t)seed;
    g.k1 = (uint32_t)(seed >> 32);
    g.chain = (uint32_t)(chain_offset + chain_stride * (uint64_t)c);
    g.iter = (uint32_t)iter;
    return g;
  }
  PhasePoint<T> load(int64_t c) const {
    PhasePoint<T> z;
    z.th.assign(th.begin() + c * D, th.begin() + (c + 1) * D);
    z.r.assign(r.begin() + c * D, r.begin() + (c + 1) * D);
    z.g.assign(g.begin() + c * D, g.begin() + (c + 1) * D);
    z.lp = lp[c];
    z.lk = lk[c];
    return z;
  }
  void store(int64_t c, const PhasePoint<T>& z) {
    std::copy(z.th.begin(), z.th.end(), th.begin() + c * D);
    std::copy(z.r.begin(), z.r.end(), r.begin() + c * D);
    std::copy(z.g.begin(), z.g.end(), g.begin() + c * D);
    lp[c] = z.lp;
    lk[c] = z.lk;
  }
  LeapfrogCfg<T> lfcfg(int64_t c) const {
    LeapfrogCfg<T> lf;
    lf.kind = integ_kind;
    lf.eps = eps_cur[c];
    lf.alpha = integ_kind == AHMC_INTEGRATOR_TEMPERED ? integ_param : T(1);
    return lf;
  }
};

template <class T>
int fail(Ctx<T>* c, int code, const std::string& msg) {
  c->err = msg;
  return code;
}

// renew(metric, M⁻¹) (src/metric.jl:61-63 sqrt.; :104-109 cholesky upper)
template <class T>
int set_metric(Ctx<T>* c, int kind, const T* minv, int64_t n) {
  const int64_t D = c->D, N = c->N;
  if (kind == AHMC_METRIC_UNIT) {
    c->metric_kind = kind;
    c->minv.clear(); c->sqrt_minv.clear(); c->chol.clear();
    return AHMC_OK;
  }
  if (!minv) return fail(c, AHMC_ERR_ARGUMENT, "set_metric: M⁻¹ pointer is NULL");
  if (kind == AHMC_METRIC_DIAG) {
    if (n != D && n != D * N)
      return fail(c, AHMC_ERR_ARGUMENT, "AxesMismatch: diagonal M⁻¹ must have D or D*N elements");
    c->metric_kind = kind;
    c->metric_per_chain = (n == D * N) && !(N == 1);
    c->minv.assign(minv, minv + n);
    c->sqrt_minv.resize(n);
    for (int64_t i = 0; i < n; ++i) c->sqrt_minv[i] = std::sqrt(c->minv[i]);
    return AHMC_OK;
  }
  if (kind == AHMC_METRIC_DENSE) {
    if (n != D * D) return fail(c, AHMC_ERR_ARGUMENT, "AxesMismatch: dense M⁻¹ must have D*D elements");
    c->metric_kind = kind;
    c->metric_per_chain = false;
    c->minv.assign(minv, minv + n);
    // upper Cholesky factor U, UᵀU = M⁻¹
    std::vector<T>& U = c->chol;
    U.assign(D * D, T(0));
    for (int64_t j = 0; j < D; ++j) {
      for (int64_t i = 0; i <= j; ++i) {
        T s = c->minv[i + j * D];
        for (int64_t k = 0; k < i; ++k) s -= U[k + i * D] * U[k + j * D];
        if (i == j) {
          if (!(s > 0)) return fail(c, AHMC_ERR_ARGUMENT, "PosDefException: M⁻¹ is not positive definite");
          U[i + j * D] = std::sqrt(s);
        } else {
          U[i + j * D] = s / U[i + i * D];
        }
      }
    }
    return AHMC_OK;
  }
  return fail(c, AHMC_ERR_ARGUMENT, "set_metric: unknown metric kind");
}

// jitter(rng, lf) (src/integrator.jl:140-156): ϵ = ϵ0 (1 + jitter (2u − 1)), per chain
template <class T>
void apply_jitter(Ctx<T>* c) {
  for (int64_t i = 0; i < c->N; ++i) {
    if (c->integ_kind == AHMC_INTEGRATOR_JITTERED) {
      T u = (T)c->rng(i, c->iteration).uniform(RNG_JITTER, 0);
      c->eps_cur[i] = c->eps_nom[i] * (1 + c->integ_param * (2 * u - 1));
    } else {
      c->eps_cur[i] = c->eps_nom[i];
    }
  }
}

// refresh (src/hamiltonian.jl:213-220 full; :243-254 partial) for one chain
template <class T>
PhasePoint<T> refresh(Ctx<T>* c, int64_t i, const PhasePoint<T>& z, T alpha) {
  MetricView<T> m = c->metric_view(i);
  std::vector<T> rn(c->D);
  rand_momentum(c->rng(i, c->iteration), RNG_MOMENTUM, m, rn.data());
  if (alpha != 0) {
    T s = std::sqrt(1 - alpha * alpha);
    for (int64_t d = 0; d < c->D; ++d) rn[d] = alpha * z.r[d] + s * rn[d];
  }
  return make_phasepoint(c->target, m, z.th.data(), rn.data());  // recomputes ℓπ, ∇ℓπ at θ
}

// mh_accept_ratio (src/trajectory.jl:855-880)
template <class T>
inline void mh_accept_ratio(const Rng& rng, uint32_t draw, T H, T Hp, bool& accept, T& alpha, double* margin = nullptr) {
  accept = Hp < H + (T)rng.randexp(RNG_TRANSITION, draw);
  note_margin(margin, (double)Hp, (double)(H + (T)rng.randexp(RNG_TRANSITION, draw)), std::max(1.0, std::max(std::abs((double)H), std::abs((double)Hp))));
  alpha = jl_min(T(1), std::exp(H - Hp));
}

// static transition for one chain (src/trajectory.jl:271-300), EndPointTS (:336-340) or
// MultinomialTS (:369-390 with the coupled n_steps_fwd of Q4)
template <class T>
void hmc_transition_chain(Ctx<T>* c, int64_t i, int64_t L, int sampler, int64_t n_fwd_coupled, T refresh_alpha,
                          int64_t stop_at /* ref_compat Q1: max steps, <0 = none */) {
  MetricView<T> m = c->metric_view(i);
  LeapfrogCfg<T> lf = c->lfcfg(i);
  Rng rng = c->rng(i, c->iteration);
  PhasePoint<T> z = refresh(c, i, c->load(i), refresh_alpha);
  T H0 = energy(z);
  PhasePoint<T> zp;
  bool is_accept;
  T alpha;
  TStat<T>& st = c->stat[i];
  if (sampler == AHMC_TS_ENDPOINT) {
    int64_t nst = (stop_at >= 0 && stop_at < L) ? stop_at : L;
    zp = leapfrog_step(lf, c->target, m, z, nst, true);
    mh_accept_ratio(rng, 0, energy(z), energy(zp), is_accept, alpha, &c->margin[i]);
  } else {
    std::vector<PhasePoint<T>> fwd, bwd;
    leapfrog_step(lf, c->target, m, z, n_fwd_coupled, true, &fwd);
    leapfrog_step(lf, c->target, m, z, L - n_fwd_coupled, false, &bwd);
    std::vector<const PhasePoint<T>*> zs;  // vcat(reverse(zs_bwd)..., z, zs_fwd...)
    for (auto it = bwd.rbegin(); it != bwd.rend(); ++it) zs.push_back(&*it);
    zs.push_back(&z);
    for (auto& p : fwd) zs.push_back(&p);
    // randcat(rng, zs, unnorm_ℓp) (:344-352, src/utilities.jl:51-59 scalar form)
    std::vector<T> lw(zs.size());
    T mx = neg_inf<T>();
    for (size_t k = 0; k < zs.size(); ++k) {
      lw[k] = -energy(*zs[k]);
      mx = jl_max(mx, lw[k]);
    }
    T se = 0;  // logsumexp
    for (T v : lw) se += std::exp(v - mx);
    T lse = mx + std::log(se);
    T u = (T)rng.uniform(RNG_TRANSITION, 0);
    T cum = 0;
    size_t idx = 0;
    // (margin: the cumulative sums live on [0, 1], each term exp(−H_k − lse) carries the ABSOLUTE rounding of its energy as a relative
    // error — the scale of the comparison is the transition's energy scale, as for every other weight comparison)
    const double hs = std::isfinite((double)H0) ? std::max(1.0, std::abs((double)H0)) : 1.0;
    while (cum < u && idx < zs.size()) {
      note_margin(&c->margin[i], (double)cum, (double)u, hs);
      cum += std::exp(lw[idx++] - lse);
    }
    if (idx < zs.size()) note_margin(&c->margin[i], (double)cum, (double)u, hs);   // the comparison that ended the scan
    if (idx < 1) idx = 1;
    zp = *zs[idx - 1];
    is_accept = true;
    T sa = 0;
    for (T v : lw) sa += std::exp(jl_min(T(0), -((-v) - energy(z))));  // α = exp(min(0, -ΔH))
    alpha = sa / T(lw.size());
  }
  // accept_phasepoint! (:303-332) then momentum flip (:283)
  PhasePoint<T> zn = is_accept ? zp : z;
  for (auto& x : zn.r) x = -x;
  T H = energy(zn), Hp = energy(zp);
  st.n_steps = (int32_t)L;
  st.is_accept = is_accept ? 1 : 0;
  st.acceptance_rate = alpha;
  st.log_density = zn.lp;
  st.hamiltonian_energy = H;
  st.hamiltonian_energy_error = H - H0;
  st.max_hamiltonian_energy_error = 0;
  st.tree_depth = 0;
  st.numerical_error = std::isfinite(Hp) ? 0 : 1;  // scalar-chain form of :295
  c->store(i, zn);
}

template <class T>
int64_t resolve_L(Ctx<T>* c, int64_t L, double lambda) {
  if (lambda > 0) {  // nsteps for FixedIntegrationTime (src/trajectory.jl:241-243)
    double e = (double)c->eps_nom[0];
    int64_t n = (int64_t)std::floor(lambda / e);
    return n < 1 ? 1 : n;
  }
  return L;
}

template <class T>
void accumulate(Ctx<T>* c) {
  if (c->acc_sum.empty()) {
    c->acc_sum.assign(c->D * c->N, T(0));
    c->acc_sumsq.assign(c->D * c->N, T(0));
  }
  c->acc_ntrans += 1;
  if (c->acc_energy.empty()) c->acc_energy.assign(5 * c->N, T(0));
  if (c->acc_nsteps_c.empty()) { c->acc_nsteps_c.assign(c->N, 0); c->acc_ndiv_c.assign(c->N, 0); }
  for (int64_t i = 0; i < c->N; ++i) {
    c->acc_nsteps += c->stat[i].n_steps;
    c->acc_ndiv += c->stat[i].numerical_error;
    c->acc_nsteps_c[i] += c->stat[i].n_steps;
    c->acc_ndiv_c[i] += c->stat[i].numerical_error;
    // running sums for EBFMI (src/diagnosis.jl:1-3), the recursion the HIP kernels use (accumulate_energy)
    T* ea = c->acc_energy.data();
    const T H = c->stat[i].hamiltonian_energy, n0 = ea[i], prev = ea[c->N + i], n = n0 + 1;
    if (n0 > 0) { const T d = H - prev; ea[2 * c->N + i] += d * d; }
    const T mean = ea[3 * c->N + i], delta = H - mean, mean2 = mean + delta / n;
    ea[3 * c->N + i] = mean2;
    ea[4 * c->N + i] += delta * (H - mean2);
    ea[i] = n;
    ea[c->N + i] = H;
  }
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < c->D * c->N; ++k) {
    c->acc_sum[k] += c->th[k];
    c->acc_sumsq[k] += c->th[k] * c->th[k];
  }
}

template <class T>
int hmc_transition(Ctx<T>* c, int64_t L, double lambda, int sampler, T refresh_alpha) {
  if (int rcb = check_builtin(c, "hmc_transition")) return rcb;
  if (!c->have_point) return fail(c, AHMC_ERR_STATE, "transition before set_position");
  if (sampler != AHMC_TS_ENDPOINT && sampler != AHMC_TS_MULTINOMIAL)
    return fail(c, AHMC_ERR_ARGUMENT, "static HMC supports EndPointTS and MultinomialTS");
  if (lambda > 0 && !c->eps_scalar && c->N != 1)  // (one chain: its step size IS the scalar)
    return fail(c, AHMC_ERR_ARGUMENT, "FixedIntegrationTime needs a scalar step size (src/trajectory.jl:241-243)");
  L = resolve_L(c, L, lambda);
  if (L < 0) L = -L;
  apply_jitter(c);
  int64_t n_fwd = 0;
  if (sampler == AHMC_TS_MULTINOMIAL) {
    // rand_coupled(rng, 0:n_steps) (src/trajectory.jl:373, src/utilities.jl:39-47): ONE draw
    Rng shared = c->rng(0, c->iteration);
    shared.chain = COUPLED_CHAIN;
    double u = shared.uniform(RNG_TRANSITION, 0);
    n_fwd = (int64_t)std::floor(u * (double)(L + 1));
    if (n_fwd > L) n_fwd = L;
  }
  int64_t stop_at = -1;
  if (c->ref_compat && sampler == AHMC_TS_ENDPOINT) {
    // Q1 (src/integrator.jl:252-258): in matrix mode the step loop breaks for ALL chains at the
    // first step where any chain is non-finite.  Dry-run to find that step.
    for (int64_t i = 0; i < c->N; ++i) {
      MetricView<T> m = c->metric_view(i);
      PhasePoint<T> z = refresh(c, i, c->load(i), refresh_alpha);
      std::vector<PhasePoint<T>> tr;
      leapfrog_step(c->lfcfg(i), c->target, m, z, L, true, &tr);
      if (!tr.empty() && !phasepoint_isfinite(m, tr.back())) {
        int64_t bs = (int64_t)tr.size();
        if (stop_at < 0 || bs < stop_at) stop_at = bs;
      }
    }
  }
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t i = 0; i < c->N; ++i) hmc_transition_chain(c, i, L, sampler, n_fwd, refresh_alpha, stop_at);
  c->iteration += 1;
  return AHMC_OK;
}

// (as the HIP engine's check_builtin: with AHMC_TARGET_EXTERNAL the caller evaluates the log-density — the calls that would evaluate it
// themselves are refused; whole transitions go through ahmc_ext_*, single leapfrogs through ahmc_lf_pre / ahmc_lf_post)
template <class T>
int check_builtin(Ctx<T>* c, const char* what) {
  if (c->target.kind == AHMC_TARGET_EXTERNAL)
    return fail(c, AHMC_ERR_STATE, std::string(what) + " needs a built-in target; with AHMC_TARGET_EXTERNAL the caller evaluates the log-density");
  return AHMC_OK;
}

template <class T>
int nuts_transition_all(Ctx<T>* c, int max_depth, double delta_max, int criterion, int sampler, T refresh_alpha) {
  if (int rcb = check_builtin(c, "nuts_transition")) return rcb;
  if (!c->have_point) return fail(c, AHMC_ERR_STATE, "transition before set_position");
  if (sampler != AHMC_TS_MULTINOMIAL && sampler != AHMC_TS_SLICE)
    return fail(c, AHMC_ERR_ARGUMENT, "NUTS supports MultinomialTS and SliceTS");
  if (criterion < AHMC_TC_CLASSIC || criterion > AHMC_TC_STRICT)
    return fail(c, AHMC_ERR_ARGUMENT, "unknown termination criterion");
  // (the boundary's domain, include/ahmc_hip.h: the reference takes any Int — max_depth <= 0 would return the start point with 0/0 statistics)
  if (max_depth < 1) return fail(c, AHMC_ERR_ARGUMENT, "max_depth must be >= 1");
  apply_jitter(c);
  NutsCfg<T> cfg;
  cfg.sampler = sampler;
  cfg.criterion = criterion;
  cfg.max_depth = max_depth;
  cfg.delta_max = (T)delta_max;
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t i = 0; i < c->N; ++i) {
    MetricView<T> m = c->metric_view(i);
    LeapfrogCfg<T> lf = c->lfcfg(i);
    Rng rng = c->rng(i, c->iteration);
    PhasePoint<T> z = refresh(c, i, c->load(i), refresh_alpha);
    NutsEnv<T> e{&lf, &c->target, &m, &cfg, &rng, 0};
    e.margin = &c->margin[i];
    PhasePoint<T> zc = nuts_transition(e, z, c->stat[i]);
    c->store(i, zc);
  }
  c->iteration += 1;
  return AHMC_OK;
}

// A(h, z, ϵ) + find_good_stepsize for one chain (src/trajectory.jl:753-837), incl. quirk Q3
template <class T>
T find_good_stepsize_chain(Ctx<T>* c, int64_t i, T init, int max_n_iters) {
  MetricView<T> m = c->metric_view(i);
  T eps = init, epsp = init;
  const T log_a_min = 2 * std::log(T(0.5)), log_a_cross = std::log(T(0.5)), log_a_max = std::log(T(0.75));
  const T d = 2, invd = T(1) / d;
  std::vector<T> r(c->D);
  rand_momentum(c->rng(i, c->iteration), RNG_FINDEPS, m, r.data());
  PhasePoint<T> z = make_phasepoint(c->target, m, c->th.data() + i * c->D, r.data());
  T H = energy(z);
  auto A = [&](T e) {
    LeapfrogCfg<T> lf;
    lf.kind = AHMC_INTEGRATOR_LEAPFROG;
    lf.eps = e;
    return energy(leapfrog_step(lf, c->target, m, z, 1, true));
  };
  double* mg = &c->margin[i];
  const double hs = std::isfinite((double)H) ? std::max(1.0, std::abs((double)H)) : 1.0;
  T Hp = A(eps);
  T dH = H - Hp;
  bool too_high = dH > log_a_cross;
  note_margin(mg, (double)dH, (double)log_a_cross, hs);
  for (int it = 0; it < max_n_iters; ++it) {
    epsp = too_high ? d * eps : invd * eps;
    Hp = A(eps);  // Q3: evaluated at ϵ, not ϵ′ (src/trajectory.jl:799-800)
    dH = H - Hp;
    note_margin(mg, (double)dH, (double)log_a_cross, hs);
    if (too_high != (dH > log_a_cross)) break;
    eps = epsp;
  }
  T lo = jl_min(eps, epsp), hi = jl_max(eps, epsp);  // minmax
  eps = lo; epsp = hi;
  for (int it = 0; it < max_n_iters; ++it) {
    T mid = eps / 2 + epsp / 2;  // Statistics.middle
    Hp = A(mid);
    dH = H - Hp;
    note_margin(mg, (double)dH, (double)log_a_max, hs);
    note_margin(mg, (double)dH, (double)log_a_min, hs);
    if (dH > log_a_max) eps = mid;
    else if (dH < log_a_min) epsp = mid;
    else { eps = mid; break; }
  }
  return eps;
}

template <class T>
void leapfrog_all(Ctx<T>* c, int64_t n_steps, bool fwd) {
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t i = 0; i < c->N; ++i)
    c->store(i, leapfrog_step(c->lfcfg(i), c->target, c->metric_view(i), c->load(i), n_steps, fwd));
}

template <class T>
std::vector<T> find_good_stepsize_all(Ctx<T>* c, T init, int max_n_iters) {
  std::vector<T> out(c->N);
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t i = 0; i < c->N; ++i) out[i] = find_good_stepsize_chain(c, i, init, max_n_iters);
  return out;
}


// ---------------------------------------------------------------------------------------------
// External target: ask / tell (include/ahmc_hip.h, ahmc_ext_*).  The transition code above is used
// unchanged; only the target evaluation differs (Target::ext_eval).
// ---------------------------------------------------------------------------------------------
thread_local void* g_ext_entry_arg = nullptr;

template <class T>
T ext_eval_thunk(void* self, int64_t D, const T* th, T* g) {
  auto* c = static_cast<Ctx<T>*>(self);
  auto& x = c->ext;
  const int64_t i = x.cur;
  std::copy(th, th + D, x.theta.begin() + i * D);
  x.co[i].state = 1;
  swapcontext(&x.co[i].uc, &x.main_uc);  // back to ahmc_ext_begin / ahmc_ext_advance
  // resumed by ahmc_ext_advance: the caller's arrays hold this chain's evaluation
  for (int64_t d = 0; d < D; ++d) g[d] = -x.in_g[i * D + d];  // in_g is -∇ℓπ; this function returns +∇ℓπ
  return x.in_lp[i];
}

template <class T>
void jitter_chain(Ctx<T>* c, int64_t i) {  // apply_jitter for one chain
  if (c->integ_kind == AHMC_INTEGRATOR_JITTERED) {
    T u = (T)c->rng(i, c->iteration).uniform(RNG_JITTER, 0);
    c->eps_cur[i] = c->eps_nom[i] * (1 + c->integ_param * (2 * u - 1));
  } else {
    c->eps_cur[i] = c->eps_nom[i];
  }
}

// everything chain i does in an ext run (c->iteration is set to the chain's own transition before every resume)
template <class T>
void ext_chain_body(Ctx<T>* c, int64_t i) {
  auto& x = c->ext;
  if (x.mode == 3) {
    x.fe_out[i] = find_good_stepsize_chain(c, i, (T)x.fe_init, x.fe_iters);
    return;
  }
  for (int k = 0; k < x.n_trans; ++k) {
    x.co[i].k = k;
    c->iteration = x.iter0 + (uint64_t)k;
    jitter_chain(c, i);
    if (x.mode == 1) {
      NutsCfg<T> cfg;
      cfg.sampler = x.cfg.sampler;
      cfg.criterion = x.cfg.criterion;
      cfg.max_depth = x.cfg.max_depth;
      cfg.delta_max = (T)x.cfg.delta_max;
      MetricView<T> m = c->metric_view(i);
      LeapfrogCfg<T> lf = c->lfcfg(i);
      Rng rng = c->rng(i, c->iteration);
      PhasePoint<T> z = refresh(c, i, c->load(i), (T)x.cfg.refresh_alpha);
      NutsEnv<T> e{&lf, &c->target, &m, &cfg, &rng, 0};
      e.margin = &c->margin[i];
      PhasePoint<T> zc = nuts_transition(e, z, c->stat[i]);
      c->store(i, zc);
    } else {
      int64_t L = resolve_L(c, x.cfg.L, x.cfg.lambda);
      if (L < 0) L = -L;
      int64_t n_fwd = 0;
      if (x.cfg.sampler == AHMC_TS_MULTINOMIAL) {  // rand_coupled: the same draw for every chain (hmc_transition)
        Rng shared = c->rng(0, c->iteration);
        shared.chain = COUPLED_CHAIN;
        double u = shared.uniform(RNG_TRANSITION, 0);
        n_fwd = (int64_t)std::floor(u * (double)(L + 1));
        if (n_fwd > L) n_fwd = L;
      }
      hmc_transition_chain(c, i, L, x.cfg.sampler, n_fwd, (T)x.cfg.refresh_alpha, -1);
    }
  }
}

template <class T>
void ext_co_entry() {
  auto* c = static_cast<Ctx<T>*>(g_ext_entry_arg);
  const int64_t i = c->ext.cur;
  ext_chain_body(c, i);
  c->ext.co[i].state = 2;
  // returning resumes uc_link = main_uc
}

template <class T>
void ext_resume(Ctx<T>* c, int64_t i) {
  auto& x = c->ext;
  x.cur = i;
  c->iteration = x.iter0 + (uint64_t)x.co[i].k;
  g_ext_entry_arg = c;
  x.co[i].state = 0;
  swapcontext(&x.main_uc, &x.co[i].uc);
  x.cur = -1;
}

template <class T>
void ext_finish_if_done(Ctx<T>* c) {
  auto& x = c->ext;
  for (int64_t i = 0; i < x.n_co; ++i)
    if (x.co[i].state != 2) return;
  if (x.mode == 3) {
    c->iteration = x.iter0;
    c->eps_nom = x.fe_out;
    c->eps_cur = x.fe_out;
    c->eps_scalar = false;
  } else {
    c->iteration = x.iter0 + (uint64_t)x.n_trans;
  }
  x.mode = 0;
  x.co.reset();
  x.n_co = 0;
  c->target.ext_eval = nullptr;
  c->target.ext_self = nullptr;
}

template <class T>
int ext_start(Ctx<T>* c, int mode) {
  auto& x = c->ext;
  x.mode = mode;
  x.iter0 = c->iteration;
  x.theta.assign(c->D * c->N, T(0));
  x.fe_out.assign(c->N, T(0));
  x.in_lp = nullptr;
  x.in_g = nullptr;
  c->target.ext_eval = &ext_eval_thunk<T>;
  c->target.ext_self = c;
  x.co.reset(new typename Ctx<T>::ExtCo[(size_t)c->N]);
  x.n_co = c->N;
  const size_t stack_bytes = (size_t)1 << 19;  // build_tree recurses max_depth deep; its vectors live on the heap
  for (int64_t i = 0; i < c->N; ++i) {
    auto& co = x.co[(size_t)i];
    void* st = mmap(nullptr, stack_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (st == MAP_FAILED) {
      x.co.reset();
      x.n_co = 0;
      x.mode = 0;
      c->target.ext_eval = nullptr;
      return fail(c, AHMC_ERR_RUNTIME, "ext_begin: cannot map the coroutine stacks");
    }
    co.stack = st;
    co.stack_bytes = stack_bytes;
    getcontext(&co.uc);
    co.uc.uc_stack.ss_sp = co.stack;
    co.uc.uc_stack.ss_size = co.stack_bytes;
    co.uc.uc_link = &x.main_uc;
    makecontext(&co.uc, (void (*)())&ext_co_entry<T>, 0);
  }
  for (int64_t i = 0; i < c->N; ++i) ext_resume(c, i);  // every chain runs to its first evaluation request
  ext_finish_if_done(c);
  return AHMC_OK;
}

// --- adaptation ------------------------------------------------------------------------------
template <class T>
void da_reset(Ctx<T>* c) {  // reset!(das) (src/adaptation/stepsize.jl:40-53)
  for (int64_t i = 0; i < c->N; ++i) {
    c->da_m[i] = 0;
    c->da_mu[i] = std::log(10 * c->da_eps[i]);
    c->da_xbar[i] = 0;
    c->da_Hbar[i] = 0;
  }
}

template <class T>
void da_adapt(Ctx<T>* c, const T* alpha_ext = nullptr) {  // adapt_stepsize! (src/adaptation/stepsize.jl:178-210), per chain
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < c->N; ++i) {
    T alpha = alpha_ext ? alpha_ext[i] : c->stat[i].acceptance_rate;
    int32_t m = c->da_m[i] + 1;
    T eta_H = T(1) / (T(m) + c->da_t0);
    T Hbar = (T(1) - eta_H) * c->da_Hbar[i] + eta_H * (c->da_delta - jl_min(T(1), alpha));
    T x = c->da_mu[i] - Hbar * (std::sqrt(T(m)) / c->da_gamma);
    T eta_x = std::pow(T(m), -c->da_kappa);
    T xbar = (T(1) - eta_x) * c->da_xbar[i] + eta_x * x;
    T eps = std::exp(x);
    if (!std::isfinite(eps)) continue;  // previous (m, ϵ, x̄, H̄) kept (:199-203)
    c->da_m[i] = m;
    c->da_eps[i] = eps;
    c->da_xbar[i] = xbar;
    c->da_Hbar[i] = Hbar;
  }
}

template <class T>
void wv_reset(Ctx<T>* c) {  // reset!(wv) (src/adaptation/massmatrix.jl:133-138)
  c->wv_n = 0;
  std::fill(c->wv_mu.begin(), c->wv_mu.end(), T(0));
  std::fill(c->wv_M.begin(), c->wv_M.end(), T(0));
  std::fill(c->wg_mu.begin(), c->wg_mu.end(), T(0));  // NutpieVar reset! (:232-236)
  std::fill(c->wg_M.begin(), c->wg_M.end(), T(0));
}

template <class T>
void wv_push(Ctx<T>* c, const T* th_ext = nullptr, const T* g_ext = nullptr) {  // push! (:141-149; NutpieVar :238-243)
  c->wv_n += 1;
  T n = T(c->wv_n);
  const bool nutpie = c->var_estimator == AHMC_VAR_NUTPIE;
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < c->D * c->N; ++k) {
    T delta = (th_ext ? th_ext[k] : c->th[k]) - c->wv_mu[k];
    c->wv_mu[k] = c->wv_mu[k] + delta / n;
    c->wv_M[k] = c->wv_M[k] + delta * delta * ((n - 1) / n);
    if (nutpie) {  // the same recursion on z.ℓπ.gradient
      T dg = (g_ext ? g_ext[k] : c->g[k]) - c->wg_mu[k];
      c->wg_mu[k] = c->wg_mu[k] + dg / n;
      c->wg_M[k] = c->wg_M[k] + dg * dg * ((n - 1) / n);
    }
  }
}

template <class T>
void wv_update(Ctx<T>* c) {  // update! (:60-62) + get_estimation (:152-157)
  if (c->wv_n < c->wv_nmin) return;
  T n = T(c->wv_n), e = T(1e-3);
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < c->D * c->N; ++k) {
    T est = n / ((n + 5) * (n - 1)) * c->wv_M[k] + e * (5 / (n + 5));
    if (c->var_estimator == AHMC_VAR_NUTPIE) {  // sqrt.(est(positions) ./ est(gradients)) (:246-250)
      T eg = n / ((n + 5) * (n - 1)) * c->wg_M[k] + e * (5 / (n + 5));
      est = std::sqrt(est / eg);
    }
    c->wv_var[k] = est;
  }
}

// AHMC_VAR_POOLED (include/ahmc_hip.h): the per-chain WelfordVar states pooled into one (D,) estimate — Chan's merge
// of equal-count partitions over the chains, then over the ranks (in rank order), then get_estimation (:152-157)
template <class T>
int wv_update_pooled(Ctx<T>* c) {
  if (c->wv_n < c->wv_nmin) return AHMC_OK;
  const int64_t D = c->D, N = c->N;
  const double n = (double)c->wv_n;
  std::vector<double> part((size_t)(2 * D + 1));
  for (int64_t d = 0; d < D; ++d) {
    double s_mu = 0, s_M = 0;
    for (int64_t ch = 0; ch < N; ++ch) { s_mu += (double)c->wv_mu[d + ch * D]; s_M += (double)c->wv_M[d + ch * D]; }
    const double mean = s_mu / (double)N;
    double s_d = 0;
    for (int64_t ch = 0; ch < N; ++ch) { const double df = (double)c->wv_mu[d + ch * D] - mean; s_d += df * df; }
    part[(size_t)d] = mean;
    part[(size_t)(D + d)] = s_M + n * s_d;
  }
  part[(size_t)(2 * D)] = n * (double)N;
  std::vector<double> all = part;
  int R = 1;
  if (c->xgather && c->x_ranks > 1) {
    R = c->x_ranks;
    all.assign(part.size() * (size_t)R, 0.0);
    if (c->xgather(part.data(), all.data(), (int64_t)part.size(), c->xgather_user) != 0) return fail(c, AHMC_ERR_RUNTIME, "pooled update: the all-gather hook failed");
  }
  const size_t stride = (size_t)(2 * D + 1);
  for (int64_t d = 0; d < D; ++d) {
    double na = all[(size_t)(2 * D)], mu = all[(size_t)d], M = all[(size_t)(D + d)];
    for (int p2 = 1; p2 < R; ++p2) {
      const double nb = all[p2 * stride + 2 * D], mub = all[p2 * stride + d], Mb = all[p2 * stride + D + d];
      const double delta = mub - mu, nt = na + nb;
      mu = mu + delta * (nb / nt);
      M = M + Mb + delta * delta * (na * nb / nt);
      na = nt;
    }
    c->wv_var[d] = (T)(na / ((na + 5) * (na - 1)) * M + 1e-3 * (5 / (na + 5)));
  }
  return AHMC_OK;
}

template <class T>
void wc_reset(Ctx<T>* c) {  // reset!(wc) (massmatrix.jl:316-321)
  c->wc_n = 0;
  std::fill(c->wc_mu.begin(), c->wc_mu.end(), T(0));
  std::fill(c->wc_M.begin(), c->wc_M.end(), T(0));
}
template <class T>
void wc_push(Ctx<T>* c, const T* th_ext) {  // push!(wc, s) (:323-331) for s = each chain's position in turn
  const int64_t D = c->D;
  std::vector<T> delta(D);
  for (int64_t ch = 0; ch < c->N; ++ch) {
    const T* s = (th_ext ? th_ext : c->th.data()) + ch * D;
    c->wc_n += 1;
    const T n = (T)c->wc_n;
    for (int64_t d = 0; d < D; ++d) {
      delta[d] = s[d] - c->wc_mu[d];
      c->wc_mu[d] = c->wc_mu[d] + delta[d] / n;
    }
    for (int64_t j = 0; j < D; ++j)
      for (int64_t i2 = 0; i2 < D; ++i2) c->wc_M[i2 + j * D] += (s[i2] - c->wc_mu[i2]) * delta[j];  // M + (s − μ) δᵀ
  }
}
template <class T>
void wc_update(Ctx<T>* c) {  // update! + get_estimation (:333-340): n/((n+5)(n−1))·M + 10⁻³·5/(n+5)·I
  if (c->wc_n < c->wv_nmin) return;
  const T n = (T)c->wc_n, e = T(1e-3);
  const int64_t D = c->D;
  for (int64_t j = 0; j < D; ++j)
    for (int64_t i2 = 0; i2 < D; ++i2) c->wc_cov[i2 + j * D] = n / ((n + 5) * (n - 1)) * c->wc_M[i2 + j * D] + (i2 == j ? e * (5 / (n + 5)) : T(0));
}

template <class T>
int adapt(Ctx<T>* c, int64_t i, int64_t n_adapts, const T* th_ext = nullptr, const T* alpha_ext = nullptr, const T* g_ext = nullptr) {  // src/sampler.jl:72-90
  if (c->adapt_kind == AHMC_ADAPT_NONE || i > n_adapts) return AHMC_OK;
  const bool has_ss = c->adapt_kind != AHMC_ADAPT_MASSMATRIX;
  const bool has_mm = c->adapt_kind != AHMC_ADAPT_STEPSIZE && c->metric_kind == AHMC_METRIC_DIAG;
  const bool has_cov = c->adapt_kind != AHMC_ADAPT_STEPSIZE && c->metric_kind == AHMC_METRIC_DENSE;
  if (has_mm && c->var_estimator == AHMC_VAR_NUTPIE && th_ext && !g_ext)  // massmatrix.jl:234-236
    return fail(c, AHMC_ERR_ARGUMENT, "`NutpieVar` adaptation requires position and gradient information!");
  if (c->adapt_kind == AHMC_ADAPT_STAN && (i == 1 || c->windows_n_adapts != n_adapts)) {  // initialize! (stan_adaptor.jl:105-115)
    c->windows = stan_windows(c->stan_init, c->stan_term, c->stan_window, n_adapts);
    c->windows_n_adapts = n_adapts;
  }
  if (i == n_adapts) c->adapting = false;
  const bool pooled = has_mm && c->var_estimator == AHMC_VAR_POOLED;
  if (c->adapt_kind == AHMC_ADAPT_STAN) {  // adapt!(tp::StanHMCAdaptor, ...) (:137-159)
    c->stan_i += 1;
    da_adapt(c, alpha_ext);
    bool in_window = c->stan_i >= c->windows.window_start && c->stan_i <= c->windows.window_end;
    bool window_end = std::find(c->windows.splits.begin(), c->windows.splits.end(), c->stan_i) != c->windows.splits.end();
    if (in_window && has_mm) {
      wv_push(c, th_ext, g_ext);
      if (window_end) {
        if (pooled) { int rc = wv_update_pooled(c); if (rc) return rc; } else wv_update(c);
      }
    }
    if (in_window && has_cov) {
      wc_push(c, th_ext);
      if (window_end) wc_update(c);
    }
    if (window_end) {
      da_reset(c);
      if (has_mm) wv_reset(c);
      if (has_cov) wc_reset(c);
    }
  } else {
    if (has_ss) da_adapt(c, alpha_ext);  // NaiveHMCAdaptor: ssa then pc (Adaptation.jl:52-60)
    if (has_mm) {
      wv_push(c, th_ext, g_ext);
      if (pooled) { int rc = wv_update_pooled(c); if (rc) return rc; } else wv_update(c);
    }
    if (has_cov) { wc_push(c, th_ext); wc_update(c); }
  }
  if (i == n_adapts && has_ss) {  // finalize! (stepsize.jl:55-62): ϵ = exp(x̄)
    for (int64_t k = 0; k < c->N; ++k) c->da_eps[k] = std::exp(c->da_xbar[k]);
  }
  // update(h, adaptor), update(κ, adaptor) (src/sampler.jl:3-22)
  if (has_mm) {
    int rc = set_metric(c, AHMC_METRIC_DIAG, c->wv_var.data(), pooled ? c->D : (int64_t)c->wv_var.size());
    if (rc) return rc;
  }
  if (has_cov) {  // update(h, adaptor): DenseEuclideanMetric(getM⁻¹) recomputes the Cholesky factor (metric.jl:104-109)
    std::vector<T> cov = c->wc_cov;
    int rc = set_metric(c, AHMC_METRIC_DENSE, cov.data(), (int64_t)cov.size());
    if (rc) return rc;
  }
  if (has_ss) {
    c->eps_nom = c->da_eps;
    c->eps_scalar = false;
  }
  return AHMC_OK;
}

template <class T>
int adaptor_init(Ctx<T>* c, int kind, double delta, int ib, int tb, int ws) {
  c->adapt_kind = kind;
  c->da_delta = (T)delta;
  c->stan_init = ib; c->stan_term = tb; c->stan_window = ws;
  c->stan_i = 0;
  c->windows_n_adapts = 0;
  c->adapting = kind != AHMC_ADAPT_NONE;
  if (kind != AHMC_ADAPT_NONE && kind != AHMC_ADAPT_STEPSIZE && c->metric_kind == AHMC_METRIC_DIAG && c->var_estimator == AHMC_VAR_POOLED &&
      c->metric_per_chain)
    return fail(c, AHMC_ERR_ARGUMENT, "adaptor_init: AHMC_VAR_POOLED adapts ONE shared (D,) M⁻¹: set a (D,) DiagEuclideanMetric");
  // NesterovDualAveraging(δ, ϵ) → DAState(ϵ) (stepsize.jl:25-33)
  c->da_eps = c->eps_nom;
  c->da_m.assign(c->N, 0);
  c->da_mu.resize(c->N); c->da_xbar.assign(c->N, T(0)); c->da_Hbar.assign(c->N, T(0));
  for (int64_t i = 0; i < c->N; ++i) c->da_mu[i] = std::log(10 * c->da_eps[i]);
  // WelfordVar{T}(size(metric); var = copy(M⁻¹)) per chain (src/AdvancedHMC.jl:113-115)
  if (c->metric_kind == AHMC_METRIC_DENSE) {  // WelfordCov{T}(size; cov = copy(M⁻¹)) (src/AdvancedHMC.jl:116-118)
    c->wc_n = 0;
    c->wc_mu.assign(c->D, T(0));
    c->wc_M.assign(c->D * c->D, T(0));
    c->wc_cov = c->minv;
  }
  if (c->metric_kind == AHMC_METRIC_DIAG) {
    c->wv_n = 0;
    c->wv_mu.assign(c->D * c->N, T(0));
    c->wv_M.assign(c->D * c->N, T(0));
    c->wv_var.resize(c->D * c->N);
    if (c->var_estimator == AHMC_VAR_NUTPIE) { c->wg_mu.assign(c->D * c->N, T(0)); c->wg_M.assign(c->D * c->N, T(0)); }
    for (int64_t i = 0; i < c->N; ++i)
      for (int64_t d = 0; d < c->D; ++d)
        c->wv_var[d + i * c->D] = c->minv[c->metric_per_chain ? d + i * c->D : d];
  }
  return AHMC_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
#define FOR_CTX(ctx, ...)                                                                      \
  do {                                                                                         \
    CtxBase* _b = reinterpret_cast<CtxBase*>(ctx);                                             \
    if (!_b) return AHMC_ERR_ARGUMENT;                                                         \
    if (_b->dtype == AHMC_F32) { using T = float; auto* c = static_cast<Ctx<T>*>(_b); __VA_ARGS__ } \
    else { using T = double; auto* c = static_cast<Ctx<T>*>(_b); __VA_ARGS__ }                 \
  } while (0)

// entry points that change the context: refused while an ahmc_ext_* run is in progress (the parked coroutines hold
// references into it)
#define FOR_CTX_MUT(ctx, ...)                                                                                        \
  FOR_CTX(ctx, {                                                                                                     \
    if (c->ext.mode != 0)                                                                                            \
      return fail(c, AHMC_ERR_STATE, std::string(__func__) + ": an ahmc_ext_* run is in progress (finish it or call ahmc_ext_cancel)"); \
    __VA_ARGS__                                                                                                      \
  })

extern "C" {

int32_t ahmc_abi_version(void) { return AHMC_ABI_VERSION; }
const char* ahmc_backend(void) { return "cpu-oracle"; }

int32_t ahmc_create(int32_t device, int32_t dtype, int64_t D, int64_t N, void* stream, ahmc_ctx** out) {
  (void)device; (void)stream;
  if (!out) { g_create_err = "ahmc_create: out is NULL"; return AHMC_ERR_ARGUMENT; }
  if (D < 1 || N < 1) { g_create_err = "ahmc_create: D and N must be >= 1"; return AHMC_ERR_ARGUMENT; }
  if (dtype != AHMC_F32 && dtype != AHMC_F64) { g_create_err = "ahmc_create: dtype must be AHMC_F32 or AHMC_F64"; return AHMC_ERR_ARGUMENT; }
  auto init = [&](auto* c) {
    using T = typename std::remove_reference<decltype(c->th[0])>::type;
    c->dtype = dtype; c->D = D; c->N = N;
    c->th.assign(D * N, T(0)); c->r.assign(D * N, T(0)); c->g.assign(D * N, T(0));
    c->lp.assign(N, T(0)); c->lk.assign(N, T(0));
    c->eps_nom.assign(N, T(0.1)); c->eps_cur.assign(N, T(0.1));
    c->stat.assign(N, TStat<T>());
    c->margin.assign(N, std::numeric_limits<double>::infinity());
  };
  if (dtype == AHMC_F32) { auto* c = new Ctx<float>(); init(c); *out = reinterpret_cast<ahmc_ctx*>(static_cast<CtxBase*>(c)); }
  else { auto* c = new Ctx<double>(); init(c); *out = reinterpret_cast<ahmc_ctx*>(static_cast<CtxBase*>(c)); }
  return AHMC_OK;
}

int32_t ahmc_destroy(ahmc_ctx* ctx) {
  delete reinterpret_cast<CtxBase*>(ctx);
  return AHMC_OK;
}

const char* ahmc_last_error(const ahmc_ctx* ctx) {
  if (!ctx) return g_create_err.c_str();
  return reinterpret_cast<const CtxBase*>(ctx)->err.c_str();
}

int32_t ahmc_sync(ahmc_ctx* ctx) { return ctx ? AHMC_OK : AHMC_ERR_ARGUMENT; }
void* ahmc_stream(ahmc_ctx*) { return nullptr; }

// oracle-only switch: reproduce the reference's matrix-mode batch coupling (Q1)
// oracle-only: number of OpenMP threads used by the batch loops (bench.py sizes it to the cgroup CPU quota)
int32_t ahmco_set_num_threads(int32_t n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  return 1;
#endif
}

int32_t ahmco_set_ref_compat(ahmc_ctx* ctx, int32_t on) {
  FOR_CTX(ctx, { c->ref_compat = on != 0; return AHMC_OK; });
}
// (ABI v6: the same switch as an entry point of the shared ABI — the HIP engine implements it too)
int32_t ahmc_set_ref_compat(ahmc_ctx* ctx, int32_t on) {
  FOR_CTX(ctx, { c->ref_compat = on != 0; return AHMC_OK; });
}

// oracle-only: out[N] = per chain, the smallest relative margin (note_margin) of any decision taken since the record was last
// reset — U-turn dot products against 0, `ℓw < ℓw₁ + e` (progressive sampling, both levels), the slice rules, the divergence test,
// the MH test, the multinomial index scan, the crossings of find_good_stepsize — +inf if none; reset != 0 starts a new record.
int32_t ahmco_decision_margin(ahmc_ctx* ctx, double* out, int32_t reset) {
  FOR_CTX(ctx, {
    for (int64_t i = 0; i < c->N; ++i) {
      if (out) out[i] = c->margin[(size_t)i];
      if (reset) c->margin[(size_t)i] = std::numeric_limits<double>::infinity();
    }
    return AHMC_OK;
  });
}

int32_t ahmc_set_target(ahmc_ctx* ctx, int32_t kind, const void* params, int64_t n_params) {
  FOR_CTX_MUT(ctx, {
    int64_t need = 0;
    switch (kind) {
      case AHMC_TARGET_ISO_GAUSS: case AHMC_TARGET_FUNNEL: case AHMC_TARGET_HIER_GAUSS: need = 0; break;
      case AHMC_TARGET_DIAG_GAUSS: need = 2 * c->D; break;
      case AHMC_TARGET_DENSE_GAUSS: need = c->D * c->D; break;
      case AHMC_TARGET_EXTERNAL: need = 0; break;
      default: return fail(c, AHMC_ERR_ARGUMENT, "set_target: unknown target kind");
    }
    if (kind == AHMC_TARGET_FUNNEL && c->D < 2) return fail(c, AHMC_ERR_ARGUMENT, "funnel needs D >= 2");
    if (kind == AHMC_TARGET_HIER_GAUSS && c->D < 3) return fail(c, AHMC_ERR_ARGUMENT, "hier_gauss needs D >= 3");
    if (n_params != need || (need > 0 && !params)) return fail(c, AHMC_ERR_ARGUMENT, "set_target: wrong parameter count for this family");
    c->target.kind = kind;
    const T* p = static_cast<const T*>(params);
    c->target.params.assign(p, p + need);
    c->have_point = false;
    return AHMC_OK;
  });
}

int32_t ahmc_set_target_plugin(ahmc_ctx* ctx, const char*, const void*, int64_t) {
  FOR_CTX_MUT(ctx, { return fail(c, AHMC_ERR_UNSUPPORTED, "set_target_plugin: a target plugin is device code of the HIP engine; the CPU checker takes the same "
                                                          "density as a host function (ahmc_set_target_kernel, AHMC_KERNEL_HOST) or through ask / tell"); });
}

int32_t ahmc_set_target_kernel(ahmc_ctx* ctx, int32_t handle_kind, void* handle, int32_t block_threads, int32_t chains_per_block, void* user) {
  FOR_CTX_MUT(ctx, {
    if (handle_kind != AHMC_KERNEL_HOST) return fail(c, AHMC_ERR_UNSUPPORTED, "set_target_kernel: the CPU checker takes AHMC_KERNEL_HOST (a plain C function) only");
    if (!handle) return fail(c, AHMC_ERR_ARGUMENT, "set_target_kernel: handle is NULL");
    if (block_threads < 1 || block_threads > 1024 || chains_per_block < 1) return fail(c, AHMC_ERR_ARGUMENT, "set_target_kernel: block_threads in 1..1024, chains_per_block >= 1");
    c->target.kind = AHMC_TARGET_KERNEL;
    c->target.params.clear();
    using F = void (*)(const T*, T*, T*, const int32_t*, int64_t, int32_t, int64_t, void*);
    c->target.kernel_host = reinterpret_cast<F>(handle);
    c->target.kernel_user = user;
    c->have_point = false;
    return AHMC_OK;
  });
}

int32_t ahmc_set_metric(ahmc_ctx* ctx, int32_t kind, const void* Minv, int64_t n) {
  FOR_CTX_MUT(ctx, { return set_metric(c, kind, static_cast<const T*>(Minv), n); });
}

int32_t ahmc_get_metric(ahmc_ctx* ctx, void* out, int64_t n) {
  FOR_CTX(ctx, {
    if (c->metric_kind == AHMC_METRIC_UNIT) return fail(c, AHMC_ERR_ARGUMENT, "get_metric: unit metric has no array");
    if (n != (int64_t)c->minv.size()) return fail(c, AHMC_ERR_ARGUMENT, "get_metric: size mismatch");
    std::memcpy(out, c->minv.data(), sizeof(T) * n);
    return AHMC_OK;
  });
}

int32_t ahmc_set_stepsize(ahmc_ctx* ctx, const void* eps, int64_t n) {
  FOR_CTX_MUT(ctx, {
    if (!eps || (n != 1 && n != c->N)) return fail(c, AHMC_ERR_ARGUMENT, "set_stepsize: need 1 or N step sizes");
    const T* e = static_cast<const T*>(eps);
    for (int64_t i = 0; i < c->N; ++i) c->eps_nom[i] = e[n == 1 ? 0 : i];
    c->eps_cur = c->eps_nom;
    c->eps_scalar = (n == 1);
    return AHMC_OK;
  });
}

int32_t ahmc_get_stepsize(ahmc_ctx* ctx, void* out) {
  FOR_CTX(ctx, { std::memcpy(out, c->eps_nom.data(), sizeof(T) * c->N); return AHMC_OK; });
}

int32_t ahmc_set_integrator(ahmc_ctx* ctx, int32_t kind, double param) {
  FOR_CTX_MUT(ctx, {
    if (kind < AHMC_INTEGRATOR_LEAPFROG || kind > AHMC_INTEGRATOR_TEMPERED) return fail(c, AHMC_ERR_ARGUMENT, "set_integrator: unknown kind");
    c->integ_kind = kind;
    c->integ_param = (T)param;
    return AHMC_OK;
  });
}

int32_t ahmc_seed(ahmc_ctx* ctx, uint64_t seed, uint64_t chain_offset, uint64_t chain_stride, uint64_t iteration) {
  FOR_CTX_MUT(ctx, { c->seed = seed; c->chain_offset = chain_offset; c->chain_stride = chain_stride; c->iteration = iteration; return AHMC_OK; });
}

int32_t ahmc_set_position(ahmc_ctx* ctx, const void* theta, const void* r) {
  FOR_CTX_MUT(ctx, {
    if (!theta) return fail(c, AHMC_ERR_ARGUMENT, "set_position: theta is NULL");
    if (c->target.kind == AHMC_TARGET_EXTERNAL) return fail(c, AHMC_ERR_STATE, "set_position needs a built-in target; use set_phasepoint");
    const T* th = static_cast<const T*>(theta);
    const T* rr = static_cast<const T*>(r);
    std::vector<T> zero(c->D, T(0));
    for (int64_t i = 0; i < c->N; ++i) {
      PhasePoint<T> z = make_phasepoint(c->target, c->metric_view(i), th + i * c->D, rr ? rr + i * c->D : zero.data());
      c->store(i, z);
    }
    c->have_point = true;
    return AHMC_OK;
  });
}

int32_t ahmc_set_phasepoint(ahmc_ctx* ctx, const void* theta, const void* r, const void* lp, const void* grad) {
  FOR_CTX_MUT(ctx, {
    if (!theta || !r || !lp || !grad) return fail(c, AHMC_ERR_ARGUMENT, "set_phasepoint: NULL argument");
    std::memcpy(c->th.data(), theta, sizeof(T) * c->D * c->N);
    std::memcpy(c->r.data(), r, sizeof(T) * c->D * c->N);
    std::memcpy(c->g.data(), grad, sizeof(T) * c->D * c->N);
    const T* l = static_cast<const T*>(lp);
    for (int64_t i = 0; i < c->N; ++i) {
      c->lp[i] = sanitize(l[i]);
      c->lk[i] = sanitize(neg_kinetic(c->metric_view(i), c->r.data() + i * c->D));
    }
    c->have_point = true;
    return AHMC_OK;
  });
}

int32_t ahmc_get_phasepoint(ahmc_ctx* ctx, void* theta, void* r, void* lp, void* grad, void* lk) {
  FOR_CTX(ctx, {
    size_t nb = sizeof(T) * c->D * c->N;
    if (theta) std::memcpy(theta, c->th.data(), nb);
    if (r) std::memcpy(r, c->r.data(), nb);
    if (grad) std::memcpy(grad, c->g.data(), nb);
    if (lp) std::memcpy(lp, c->lp.data(), sizeof(T) * c->N);
    if (lk) std::memcpy(lk, c->lk.data(), sizeof(T) * c->N);
    return AHMC_OK;
  });
}

int32_t ahmc_refresh_momentum(ahmc_ctx* ctx, double alpha) {
  FOR_CTX_MUT(ctx, {
    if (int rcb = check_builtin(c, "refresh_momentum")) return rcb;
    if (!c->have_point) return fail(c, AHMC_ERR_STATE, "refresh before set_position");
    for (int64_t i = 0; i < c->N; ++i) c->store(i, refresh(c, i, c->load(i), (T)alpha));
    return AHMC_OK;
  });
}

int32_t ahmc_leapfrog(ahmc_ctx* ctx, int64_t n_steps) {
  FOR_CTX_MUT(ctx, {
    if (int rcb = check_builtin(c, "leapfrog")) return rcb;
    if (!c->have_point) return fail(c, AHMC_ERR_STATE, "leapfrog before set_position");
    c->eps_cur = c->eps_nom;
    bool fwd = n_steps > 0;
    if (c->ref_compat) {
      // Q1: matrix-mode loop, all chains stop at the first step where any chain is non-finite
      int64_t n = n_steps < 0 ? -n_steps : n_steps;
      for (int64_t s = 1; s <= n; ++s) {
        bool all_finite = true;
        for (int64_t i = 0; i < c->N; ++i) {
          LeapfrogCfg<T> lf = c->lfcfg(i);
          MetricView<T> m = c->metric_view(i);
          PhasePoint<T> z = c->load(i);
          PhasePoint<T> zn = leapfrog_step(lf, c->target, m, z, 1, fwd, (std::vector<PhasePoint<T>>*)nullptr, s - 1, n);
          c->store(i, zn);
          all_finite = all_finite && phasepoint_isfinite(m, zn);
        }
        if (!all_finite) break;
      }
      return AHMC_OK;
    }
    leapfrog_all(c, n_steps, fwd);
    return AHMC_OK;
  });
}

int32_t ahmc_lf_pre(ahmc_ctx* ctx, int32_t fwd, int64_t i, int64_t n_steps) {
  FOR_CTX_MUT(ctx, {
    if (!c->have_point) return fail(c, AHMC_ERR_STATE, "lf_pre before set_phasepoint");
    for (int64_t k = 0; k < c->N; ++k) {
      LeapfrogCfg<T> lf = c->lfcfg(k);
      MetricView<T> m = c->metric_view(k);
      T eps = fwd ? lf.eps : -lf.eps;
      std::vector<T> r(c->r.begin() + k * c->D, c->r.begin() + (k + 1) * c->D), kr(c->D);
      temper(lf, r, i, true, n_steps);
      for (int64_t d = 0; d < c->D; ++d) r[d] = r[d] - eps / 2 * c->g[d + k * c->D];
      dHdr(m, r.data(), kr.data());
      for (int64_t d = 0; d < c->D; ++d) {
        c->th[d + k * c->D] = c->th[d + k * c->D] + eps * kr[d];
        c->r[d + k * c->D] = r[d];
      }
    }
    return AHMC_OK;
  });
}

int32_t ahmc_lf_post(ahmc_ctx* ctx, int32_t fwd, int64_t i, int64_t n_steps, const void* lp, const void* grad_neg) {
  FOR_CTX_MUT(ctx, {
    if (!lp || !grad_neg) return fail(c, AHMC_ERR_ARGUMENT, "lf_post: NULL argument");
    const T* l = static_cast<const T*>(lp);
    const T* gn = static_cast<const T*>(grad_neg);
    for (int64_t k = 0; k < c->N; ++k) {
      LeapfrogCfg<T> lf = c->lfcfg(k);
      MetricView<T> m = c->metric_view(k);
      T eps = fwd ? lf.eps : -lf.eps;
      std::vector<T> r(c->r.begin() + k * c->D, c->r.begin() + (k + 1) * c->D);
      for (int64_t d = 0; d < c->D; ++d) {
        c->g[d + k * c->D] = gn[d + k * c->D];
        r[d] = r[d] - eps / 2 * gn[d + k * c->D];
      }
      temper(lf, r, i, false, n_steps);
      std::copy(r.begin(), r.end(), c->r.begin() + k * c->D);
      c->lp[k] = sanitize(l[k]);
      c->lk[k] = sanitize(neg_kinetic(m, r.data()));
    }
    return AHMC_OK;
  });
}

void* ahmc_theta_ptr(ahmc_ctx* ctx) {
  CtxBase* b = reinterpret_cast<CtxBase*>(ctx);
  if (!b) return nullptr;
  if (b->dtype == AHMC_F32) return static_cast<Ctx<float>*>(b)->th.data();
  return static_cast<Ctx<double>*>(b)->th.data();
}

// ---- external target: ask / tell ----
int32_t ahmc_ext_begin(ahmc_ctx* ctx, const ahmc_kernel_cfg* cfg, int32_t n_trans) {
  FOR_CTX(ctx, {
    if (!cfg) return fail(c, AHMC_ERR_ARGUMENT, "ext_begin: cfg is NULL");
    if (n_trans < 1) return fail(c, AHMC_ERR_ARGUMENT, "ext_begin: n_trans must be >= 1");
    if (c->target.kind != AHMC_TARGET_EXTERNAL) return fail(c, AHMC_ERR_STATE, "ext_begin: the target is not AHMC_TARGET_EXTERNAL");
    if (!c->have_point) return fail(c, AHMC_ERR_STATE, "ext_begin before set_phasepoint");
    if (c->ext.mode != 0) return fail(c, AHMC_ERR_STATE, "ext_begin: a run is already in progress");
    if (cfg->refresh_alpha < 0 || cfg->refresh_alpha >= 1) return fail(c, AHMC_ERR_ARGUMENT, "ext_begin: PartialMomentumRefreshment needs 0 <= α < 1");
    if (cfg->nuts) {
      if (cfg->sampler != AHMC_TS_MULTINOMIAL && cfg->sampler != AHMC_TS_SLICE) return fail(c, AHMC_ERR_ARGUMENT, "NUTS supports MultinomialTS and SliceTS");
      if (cfg->criterion < AHMC_TC_CLASSIC || cfg->criterion > AHMC_TC_STRICT) return fail(c, AHMC_ERR_ARGUMENT, "unknown termination criterion");
      if (cfg->max_depth < 1) return fail(c, AHMC_ERR_ARGUMENT, "ext_begin: max_depth must be >= 1");
    } else {
      if (cfg->sampler != AHMC_TS_ENDPOINT && cfg->sampler != AHMC_TS_MULTINOMIAL)
        return fail(c, AHMC_ERR_ARGUMENT, "static HMC supports EndPointTS and MultinomialTS");
      if (cfg->lambda > 0 && !c->eps_scalar && c->N != 1)
        return fail(c, AHMC_ERR_ARGUMENT, "FixedIntegrationTime needs a scalar step size (src/trajectory.jl:241-243)");
      if (!(cfg->lambda > 0) && cfg->L == 0)   // (a run of no evaluations has nothing to ask the caller: the boundary's domain, as the HIP engine's)
        return fail(c, AHMC_ERR_ARGUMENT, "ext_begin: static HMC needs at least one leapfrog step");
    }
    c->ext.cfg = *cfg;
    c->ext.n_trans = n_trans;
    return ext_start(c, cfg->nuts ? 1 : 2);
  });
}

int32_t ahmc_ext_find_good_stepsize_begin(ahmc_ctx* ctx, double initial_step_size, int32_t max_n_iters) {
  FOR_CTX(ctx, {
    if (c->target.kind != AHMC_TARGET_EXTERNAL) return fail(c, AHMC_ERR_STATE, "ext_find_good_stepsize_begin: the target is not AHMC_TARGET_EXTERNAL");
    if (!c->have_point) return fail(c, AHMC_ERR_STATE, "ext_find_good_stepsize_begin before set_phasepoint");
    if (c->ext.mode != 0) return fail(c, AHMC_ERR_STATE, "ext_find_good_stepsize_begin: a run is already in progress");
    c->ext.fe_init = initial_step_size;
    c->ext.fe_iters = max_n_iters;
    c->ext.n_trans = 1;
    return ext_start(c, 3);
  });
}

int32_t ahmc_ext_pending(ahmc_ctx* ctx, int64_t* n_pending, int32_t* chains_out, void* theta_out) {
  FOR_CTX(ctx, {
    if (!n_pending) return fail(c, AHMC_ERR_ARGUMENT, "ext_pending: n_pending is NULL");
    int64_t n = 0;
    if (c->ext.mode != 0) {
      for (int64_t i = 0; i < c->N; ++i)
        if (c->ext.co[(size_t)i].state == 1) {
          if (chains_out) chains_out[n] = (int32_t)i;
          ++n;
        }
      if (theta_out && n > 0) std::memcpy(theta_out, c->ext.theta.data(), sizeof(T) * c->D * c->N);
    }
    *n_pending = n;
    return AHMC_OK;
  });
}

int32_t ahmc_ext_advance(ahmc_ctx* ctx, const void* lp, const void* grad_neg) {
  FOR_CTX(ctx, {
    if (c->ext.mode == 0) return fail(c, AHMC_ERR_STATE, "ext_advance: no run in progress");
    if (!lp || !grad_neg) return fail(c, AHMC_ERR_ARGUMENT, "ext_advance: NULL argument");
    c->ext.in_lp = static_cast<const T*>(lp);
    c->ext.in_g = static_cast<const T*>(grad_neg);
    std::vector<int64_t> waiting;
    for (int64_t i = 0; i < c->N; ++i)
      if (c->ext.co[(size_t)i].state == 1) waiting.push_back(i);
    for (int64_t i : waiting) ext_resume(c, i);
    c->ext.in_lp = nullptr;
    c->ext.in_g = nullptr;
    ext_finish_if_done(c);
    return AHMC_OK;
  });
}

int32_t ahmc_ext_cancel(ahmc_ctx* ctx) {
  FOR_CTX(ctx, {
    if (c->ext.mode != 0) {
      // the parked coroutines are simply dropped: what lives on their stacks are PhasePoint / tree vectors whose heap
      // blocks leak (bounded by the tree state of N chains); a test-infrastructure shortcut
      c->iteration = c->ext.iter0;
      c->ext.mode = 0;
      c->ext.co.reset();
      c->ext.n_co = 0;
      c->target.ext_eval = nullptr;
      c->target.ext_self = nullptr;
    }
    return AHMC_OK;
  });
}

int32_t ahmc_hmc_transition(ahmc_ctx* ctx, int64_t L, double lambda, int32_t sampler) {
  FOR_CTX_MUT(ctx, { return hmc_transition(c, L, lambda, sampler, T(0)); });
}

int32_t ahmc_nuts_transition(ahmc_ctx* ctx, int32_t max_depth, double delta_max, int32_t criterion, int32_t sampler) {
  FOR_CTX_MUT(ctx, { return nuts_transition_all(c, max_depth, delta_max, criterion, sampler, T(0)); });
}

int32_t ahmc_get_stat(ahmc_ctx* ctx, int32_t field, void* out) {
  FOR_CTX(ctx, {
    if (!out) return fail(c, AHMC_ERR_ARGUMENT, "get_stat: out is NULL");
    int32_t* oi = static_cast<int32_t*>(out);
    T* of = static_cast<T*>(out);
    for (int64_t i = 0; i < c->N; ++i) {
      const TStat<T>& s = c->stat[i];
      switch (field) {
        case AHMC_STAT_N_STEPS: oi[i] = s.n_steps; break;
        case AHMC_STAT_IS_ACCEPT: oi[i] = s.is_accept; break;
        case AHMC_STAT_ACCEPTANCE_RATE: of[i] = s.acceptance_rate; break;
        case AHMC_STAT_LOG_DENSITY: of[i] = s.log_density; break;
        case AHMC_STAT_HAMILTONIAN_ENERGY: of[i] = s.hamiltonian_energy; break;
        case AHMC_STAT_HAMILTONIAN_ENERGY_ERROR: of[i] = s.hamiltonian_energy_error; break;
        case AHMC_STAT_MAX_HAMILTONIAN_ENERGY_ERROR: of[i] = s.max_hamiltonian_energy_error; break;
        case AHMC_STAT_TREE_DEPTH: oi[i] = s.tree_depth; break;
        case AHMC_STAT_NUMERICAL_ERROR: oi[i] = s.numerical_error; break;
        case AHMC_STAT_STEP_SIZE: of[i] = c->eps_cur[i]; break;
        case AHMC_STAT_NOM_STEP_SIZE: of[i] = c->eps_nom[i]; break;
        default: return fail(c, AHMC_ERR_ARGUMENT, "get_stat: unknown field");
      }
    }
    return AHMC_OK;
  });
}

int32_t ahmc_find_good_stepsize(ahmc_ctx* ctx, double initial_step_size, int32_t max_n_iters) {
  FOR_CTX_MUT(ctx, {
    if (int rcb = check_builtin(c, "find_good_stepsize")) return rcb;
    if (!c->have_point) return fail(c, AHMC_ERR_STATE, "find_good_stepsize before set_position");
    std::vector<T> out = find_good_stepsize_all(c, (T)initial_step_size, max_n_iters);
    c->eps_nom = out;
    c->eps_cur = out;
    c->eps_scalar = false;
    return AHMC_OK;
  });
}

int32_t ahmc_adaptor_init(ahmc_ctx* ctx, int32_t kind, double delta, int32_t init_buffer, int32_t term_buffer, int32_t window_size) {
  FOR_CTX_MUT(ctx, {
    if (kind < AHMC_ADAPT_NONE || kind > AHMC_ADAPT_STAN) return fail(c, AHMC_ERR_ARGUMENT, "adaptor_init: unknown adaptor kind");
    return adaptor_init(c, kind, delta, init_buffer, term_buffer, window_size);
  });
}

int32_t ahmc_adapt(ahmc_ctx* ctx, int64_t i, int64_t n_adapts, const void* theta, const void* alpha) {
  FOR_CTX_MUT(ctx, { return adapt(c, i, n_adapts, static_cast<const T*>(theta), static_cast<const T*>(alpha)); });
}

int32_t ahmc_adapt_point(ahmc_ctx* ctx, int64_t i, int64_t n_adapts, const void* theta, const void* grad, const void* alpha) {
  FOR_CTX_MUT(ctx, { return adapt(c, i, n_adapts, static_cast<const T*>(theta), static_cast<const T*>(alpha), static_cast<const T*>(grad)); });
}

int32_t ahmc_set_var_estimator(ahmc_ctx* ctx, int32_t est) {
  FOR_CTX_MUT(ctx, {
    if (est != AHMC_VAR_WELFORD && est != AHMC_VAR_NUTPIE && est != AHMC_VAR_POOLED) return fail(c, AHMC_ERR_ARGUMENT, "set_var_estimator: unknown estimator");
    c->var_estimator = est;
    return AHMC_OK;
  });
}

int32_t ahmc_stan_windows(int32_t init_buffer, int32_t term_buffer, int32_t window_size, int64_t n_adapts,
                          int64_t* window_start, int64_t* window_end, int64_t* splits, int32_t cap, int32_t* n_splits) {
  StanWindows w = stan_windows(init_buffer, term_buffer, window_size, n_adapts);
  if (window_start) *window_start = w.window_start;
  if (window_end) *window_end = w.window_end;
  if (n_splits) *n_splits = (int32_t)w.splits.size();
  if (splits)
    for (int32_t k = 0; k < cap && k < (int32_t)w.splits.size(); ++k) splits[k] = w.splits[k];
  return AHMC_OK;
}

// (one internal body, reached without the PLT: a process may hold the HIP engine AND the CPU checker, both exporting
// these names — a call from one exported function to another could bind to the other library's)
static int32_t sample_from_impl(ahmc_ctx* ctx, const ahmc_kernel_cfg* cfg, int64_t i_first, int64_t n_samples, int64_t n_adapts, int32_t drop_warmup,
                                void* samples_out);

int32_t ahmc_sample_reserve(ahmc_ctx* ctx, const ahmc_kernel_cfg* cfg, int64_t n_samples) {
  FOR_CTX_MUT(ctx, {
    (void)n_samples;
    if (!cfg) return fail(c, AHMC_ERR_ARGUMENT, "sample_reserve: cfg is NULL");
    return AHMC_OK;  // the CPU checker draws its normals as it goes: nothing to reserve
  });
}

int32_t ahmc_sample(ahmc_ctx* ctx, const ahmc_kernel_cfg* cfg, int64_t n_samples, int64_t n_adapts, int32_t drop_warmup, void* samples_out) {
  return sample_from_impl(ctx, cfg, 1, n_samples, n_adapts, drop_warmup, samples_out);
}

int32_t ahmc_sample_from(ahmc_ctx* ctx, const ahmc_kernel_cfg* cfg, int64_t i_first, int64_t n_samples, int64_t n_adapts, int32_t drop_warmup,
                         void* samples_out) {
  return sample_from_impl(ctx, cfg, i_first, n_samples, n_adapts, drop_warmup, samples_out);
}

static int32_t sample_from_impl(ahmc_ctx* ctx, const ahmc_kernel_cfg* cfg, int64_t i_first, int64_t n_samples, int64_t n_adapts, int32_t drop_warmup,
                                void* samples_out) {
  FOR_CTX_MUT(ctx, {
    if (!cfg) return fail(c, AHMC_ERR_ARGUMENT, "sample: cfg is NULL");
    if (i_first < 1) return fail(c, AHMC_ERR_ARGUMENT, "sample_from: i_first must be >= 1");
    if (!c->have_point) return fail(c, AHMC_ERR_STATE, "sample before set_position");
    if (drop_warmup && c->adapt_kind == AHMC_ADAPT_NONE)
      return fail(c, AHMC_ERR_ARGUMENT, "Cannot drop warmup samples if there is no adaptation phase.");  // src/sampler.jl:172
    if (i_first > 1 && c->adapt_kind == AHMC_ADAPT_STAN && c->adapting && i_first <= n_adapts && c->stan_i != i_first - 1)
      return fail(c, AHMC_ERR_STATE, "sample_from: i_first = " + std::to_string(i_first) + " but the adaptor has seen " + std::to_string(c->stan_i) +
                                         " iterations (restore the checkpoint taken after iteration i_first - 1: ahmc_set_adaptor_state)");
    T* so = static_cast<T*>(samples_out);
    bool reset_done = i_first > (drop_warmup ? n_adapts + 1 : 1);  // a run resumed beyond its first kept transition continues the accumulators
    for (int64_t i = i_first; i <= n_samples; ++i) {  // src/sampler.jl:182-228
      int rc = cfg->nuts ? nuts_transition_all(c, cfg->max_depth, cfg->delta_max, cfg->criterion, cfg->sampler, (T)cfg->refresh_alpha)
                         : hmc_transition(c, cfg->L, cfg->lambda, cfg->sampler, (T)cfg->refresh_alpha);
      if (rc) return rc;
      rc = adapt(c, i, n_adapts);
      if (rc) return rc;
      if (!drop_warmup || i > n_adapts) {
        if (!reset_done) { c->acc_nsteps = c->acc_ntrans = c->acc_ndiv = 0; c->acc_sum.clear(); c->acc_sumsq.clear(); c->acc_energy.clear(); c->acc_nsteps_c.clear(); c->acc_ndiv_c.clear(); reset_done = true; }
        accumulate(c);
        int64_t j = i - (drop_warmup ? n_adapts : 0);
        if (so) std::memcpy(so + (j - 1) * c->D * c->N, c->th.data(), sizeof(T) * c->D * c->N);
      }
    }
    return AHMC_OK;
  });
}

int32_t ahmc_get_accum(ahmc_ctx* ctx, int64_t* total_n_steps, int64_t* n_transitions, int64_t* n_divergent, void* sum_theta, void* sumsq_theta) {
  FOR_CTX(ctx, {
    if (total_n_steps) *total_n_steps = c->acc_nsteps;
    if (n_transitions) *n_transitions = c->acc_ntrans;
    if (n_divergent) *n_divergent = c->acc_ndiv;
    size_t nb = sizeof(T) * c->D * c->N;
    if (sum_theta) { if (c->acc_sum.empty()) std::memset(sum_theta, 0, nb); else std::memcpy(sum_theta, c->acc_sum.data(), nb); }
    if (sumsq_theta) { if (c->acc_sumsq.empty()) std::memset(sumsq_theta, 0, nb); else std::memcpy(sumsq_theta, c->acc_sumsq.data(), nb); }
    return AHMC_OK;
  });
}

int32_t ahmc_reset_accum(ahmc_ctx* ctx) {
  FOR_CTX_MUT(ctx, { c->acc_nsteps = c->acc_ntrans = c->acc_ndiv = 0; c->acc_sum.clear(); c->acc_sumsq.clear(); c->acc_energy.clear(); c->acc_nsteps_c.clear(); c->acc_ndiv_c.clear(); return AHMC_OK; });
}

int32_t ahmc_get_accum_state(ahmc_ctx* ctx, int64_t* n_transitions, int64_t* n_steps, int64_t* n_divergent, void* sum_theta, void* sumsq_theta,
                             void* energy_sums) {
  FOR_CTX(ctx, {
    const size_t N = (size_t)c->N, nb = sizeof(T) * (size_t)c->D * N;
    if (n_transitions) *n_transitions = c->acc_ntrans;
    auto put = [&](void* dst, const void* src, size_t bytes, bool empty) { if (dst) { if (empty) std::memset(dst, 0, bytes); else std::memcpy(dst, src, bytes); } };
    put(n_steps, c->acc_nsteps_c.data(), sizeof(int64_t) * N, c->acc_nsteps_c.empty());
    put(n_divergent, c->acc_ndiv_c.data(), sizeof(int64_t) * N, c->acc_ndiv_c.empty());
    put(sum_theta, c->acc_sum.data(), nb, c->acc_sum.empty());
    put(sumsq_theta, c->acc_sumsq.data(), nb, c->acc_sumsq.empty());
    put(energy_sums, c->acc_energy.data(), sizeof(T) * 5 * N, c->acc_energy.empty());
    return AHMC_OK;
  });
}

int32_t ahmc_set_accum_state(ahmc_ctx* ctx, int64_t n_transitions, const int64_t* n_steps, const int64_t* n_divergent, const void* sum_theta,
                             const void* sumsq_theta, const void* energy_sums) {
  FOR_CTX_MUT(ctx, {
    if (n_transitions < 0) return fail(c, AHMC_ERR_ARGUMENT, "set_accum_state: n_transitions < 0");
    const size_t N = (size_t)c->N, DN = (size_t)c->D * N;
    c->acc_ntrans = n_transitions;
    if (n_steps) { c->acc_nsteps_c.assign(n_steps, n_steps + N); c->acc_nsteps = 0; for (int64_t v : c->acc_nsteps_c) c->acc_nsteps += v; }
    if (n_divergent) { c->acc_ndiv_c.assign(n_divergent, n_divergent + N); c->acc_ndiv = 0; for (int64_t v : c->acc_ndiv_c) c->acc_ndiv += v; }
    if (sum_theta) { const T* q = static_cast<const T*>(sum_theta); c->acc_sum.assign(q, q + DN); }
    if (sumsq_theta) { const T* q = static_cast<const T*>(sumsq_theta); c->acc_sumsq.assign(q, q + DN); }
    if (energy_sums) { const T* q = static_cast<const T*>(energy_sums); c->acc_energy.assign(q, q + 5 * N); }
    return AHMC_OK;
  });
}

// ---- adaptor checkpoint / resume, the final gather, device-side diagnostics (ABI v3) ----
int32_t ahmc_get_adaptor_state(ahmc_ctx* ctx, ahmc_adaptor_state* s, void* da_out, void* wv_out) {
  FOR_CTX(ctx, {
    if (!s) return fail(c, AHMC_ERR_ARGUMENT, "get_adaptor_state: state is NULL");
    const bool has_mm = c->adapt_kind != AHMC_ADAPT_NONE && c->adapt_kind != AHMC_ADAPT_STEPSIZE;
    if (has_mm && c->metric_kind == AHMC_METRIC_DENSE)
      return fail(c, AHMC_ERR_UNSUPPORTED, "get_adaptor_state: the WelfordCov state of a DenseEuclideanMetric adaptor does not round-trip yet");
    std::memset(s, 0, sizeof(*s));
    s->kind = c->adapt_kind; s->var_estimator = c->var_estimator;
    s->init_buffer = c->stan_init; s->term_buffer = c->stan_term; s->window_size = c->stan_window;
    s->adapting = c->adapting ? 1 : 0;
    s->delta = (double)c->da_delta;
    s->stan_i = c->stan_i; s->n_adapts = c->windows_n_adapts; s->wv_n = c->wv_n; s->iteration = (int64_t)c->iteration;
    s->n_welford = (has_mm && c->metric_kind == AHMC_METRIC_DIAG && !c->wv_mu.empty()) ? (c->var_estimator == AHMC_VAR_NUTPIE ? 5 : 3) : 0;
    s->has_da = (c->adapt_kind != AHMC_ADAPT_NONE && c->adapt_kind != AHMC_ADAPT_MASSMATRIX && !c->da_m.empty()) ? 1 : 0;
    const size_t N = (size_t)c->N, DN = (size_t)c->D * N;
    if (da_out && s->has_da) {
      T* o = static_cast<T*>(da_out);
      for (size_t i = 0; i < N; ++i) { o[i] = (T)c->da_m[i]; o[N + i] = c->da_eps[i]; o[2 * N + i] = c->da_mu[i]; o[3 * N + i] = c->da_xbar[i]; o[4 * N + i] = c->da_Hbar[i]; }
    }
    if (wv_out && s->n_welford) {
      T* o = static_cast<T*>(wv_out);
      const std::vector<T>* src[5] = {&c->wv_mu, &c->wv_M, &c->wv_var, &c->wg_mu, &c->wg_M};
      for (int k = 0; k < s->n_welford; ++k) std::memcpy(o + (size_t)k * DN, src[k]->data(), sizeof(T) * DN);
    }
    return AHMC_OK;
  });
}

int32_t ahmc_set_adaptor_state(ahmc_ctx* ctx, const ahmc_adaptor_state* s, const void* da_in, const void* wv_in) {
  FOR_CTX_MUT(ctx, {
    if (!s) return fail(c, AHMC_ERR_ARGUMENT, "set_adaptor_state: state is NULL");
    if (s->kind < AHMC_ADAPT_NONE || s->kind > AHMC_ADAPT_STAN) return fail(c, AHMC_ERR_ARGUMENT, "set_adaptor_state: unknown adaptor kind");
    if (s->has_da && !da_in) return fail(c, AHMC_ERR_ARGUMENT, "set_adaptor_state: the state has a dual-averaging part but da is NULL");
    if (s->n_welford && !wv_in) return fail(c, AHMC_ERR_ARGUMENT, "set_adaptor_state: the state has a variance estimator but welford is NULL");
    c->var_estimator = s->var_estimator;
    int rc = adaptor_init(c, s->kind, s->delta, s->init_buffer, s->term_buffer, s->window_size);
    if (rc) return rc;
    c->adapting = s->adapting != 0;
    c->stan_i = s->stan_i; c->wv_n = s->wv_n; c->iteration = (uint64_t)s->iteration; c->windows_n_adapts = s->n_adapts;
    if (s->kind == AHMC_ADAPT_STAN && s->n_adapts > 0) c->windows = stan_windows(c->stan_init, c->stan_term, c->stan_window, s->n_adapts);
    const size_t N = (size_t)c->N, DN = (size_t)c->D * N;
    if (s->has_da) {
      const T* in = static_cast<const T*>(da_in);
      for (size_t i = 0; i < N; ++i) { c->da_m[i] = (int32_t)in[i]; c->da_eps[i] = in[N + i]; c->da_mu[i] = in[2 * N + i]; c->da_xbar[i] = in[3 * N + i]; c->da_Hbar[i] = in[4 * N + i]; }
    }
    if (s->n_welford) {
      if (c->wv_mu.empty()) return fail(c, AHMC_ERR_STATE, "set_adaptor_state: a variance estimator needs a DiagEuclideanMetric (set the metric first)");
      if (s->n_welford == 5 && c->wg_mu.empty()) return fail(c, AHMC_ERR_STATE, "set_adaptor_state: NutpieVar state for a WelfordVar adaptor");
      const T* in = static_cast<const T*>(wv_in);
      std::vector<T>* dst[5] = {&c->wv_mu, &c->wv_M, &c->wv_var, &c->wg_mu, &c->wg_M};
      for (int k = 0; k < s->n_welford; ++k) std::memcpy(dst[k]->data(), in + (size_t)k * DN, sizeof(T) * DN);
    }
    return AHMC_OK;
  });
}

// the CPU checker has no RCCL: a world larger than one is wired by the test harness through ahmco_set_allgather
int32_t ahmc_comm_unique_id(void* id_out) {
  if (!id_out) return AHMC_ERR_ARGUMENT;
  std::memset(id_out, 0, AHMC_UNIQUE_ID_BYTES);
  return AHMC_OK;
}
int32_t ahmc_comm_init(ahmc_ctx* ctx, const void* id, int32_t n_ranks, int32_t rank) {
  FOR_CTX_MUT(ctx, {
    (void)id;
    if (n_ranks != 1 || rank != 0) return fail(c, AHMC_ERR_UNSUPPORTED, "comm_init: the CPU checker has no RCCL (multi-rank tests use ahmco_set_allgather)");
    c->x_ranks = 1; c->x_rank = 0;
    return AHMC_OK;
  });
}
int32_t ahmc_set_comm(ahmc_ctx* ctx, void* comm, int32_t n_ranks, int32_t rank) {
  FOR_CTX_MUT(ctx, {
    if (comm) return fail(c, AHMC_ERR_UNSUPPORTED, "set_comm: the CPU checker has no RCCL (multi-rank tests use ahmco_set_allgather)");
    (void)n_ranks; (void)rank;
    c->x_ranks = 1; c->x_rank = 0; c->xgather = nullptr;
    return AHMC_OK;
  });
}
int32_t ahmco_set_allgather(ahmc_ctx* ctx, int (*fn)(const double*, double*, int64_t, void*), void* user, int32_t n_ranks, int32_t rank) {
  FOR_CTX_MUT(ctx, {
    c->xgather = fn; c->xgather_user = user; c->x_ranks = fn ? n_ranks : 1; c->x_rank = fn ? rank : 0;
    return AHMC_OK;
  });
}

int32_t ahmc_comm_info(ahmc_ctx* ctx, int64_t* ranks_seen, int64_t* chains_total, int64_t* chains_min, int64_t* chains_max) {
  FOR_CTX(ctx, {
    // (the CPU checker's multi-rank world is the test harness's hook: it reports what it was told, for equal shards)
    if (ranks_seen) *ranks_seen = c->x_ranks;
    if (chains_total) *chains_total = c->N * (int64_t)c->x_ranks;
    if (chains_min) *chains_min = c->N;
    if (chains_max) *chains_max = c->N;
    return AHMC_OK;
  });
}

int32_t ahmc_gather_moments(ahmc_ctx* ctx, double* mean, double* var, int64_t* n_draws, int64_t* total_n_steps, int64_t* n_divergent) {
  FOR_CTX(ctx, {
    const int64_t D = c->D, N = c->N;
    std::vector<double> part((size_t)(2 * D + 3), 0.0);
    if (!c->acc_sum.empty())
      for (int64_t d = 0; d < D; ++d)
        for (int64_t ch = 0; ch < N; ++ch) { part[(size_t)d] += (double)c->acc_sum[d + ch * D]; part[(size_t)(D + d)] += (double)c->acc_sumsq[d + ch * D]; }
    part[(size_t)(2 * D)] = (double)c->acc_nsteps;
    part[(size_t)(2 * D + 1)] = (double)c->acc_ndiv;
    part[(size_t)(2 * D + 2)] = (double)c->acc_ntrans * (double)N;
    std::vector<double> tot = part;
    if (c->xgather && c->x_ranks > 1) {
      std::vector<double> all(part.size() * (size_t)c->x_ranks);
      if (c->xgather(part.data(), all.data(), (int64_t)part.size(), c->xgather_user) != 0) return fail(c, AHMC_ERR_RUNTIME, "gather_moments: the all-gather hook failed");
      std::fill(tot.begin(), tot.end(), 0.0);
      for (int r = 0; r < c->x_ranks; ++r)
        for (size_t k = 0; k < part.size(); ++k) tot[k] += all[(size_t)r * part.size() + k];
    }
    const double n = tot[(size_t)(2 * D + 2)];
    for (int64_t d = 0; d < D; ++d) {
      const double m = n > 0 ? tot[(size_t)d] / n : 0.0;
      if (mean) mean[d] = m;
      if (var) var[d] = n > 0 ? tot[(size_t)(D + d)] / n - m * m : 0.0;
    }
    if (n_draws) *n_draws = (int64_t)(n + 0.5);
    if (total_n_steps) *total_n_steps = (int64_t)(tot[(size_t)(2 * D)] + 0.5);
    if (n_divergent) *n_divergent = (int64_t)(tot[(size_t)(2 * D + 1)] + 0.5);
    return AHMC_OK;
  });
}

int32_t ahmc_gather_state(ahmc_ctx* ctx, void* theta_all) {
  FOR_CTX(ctx, {
    if (!theta_all) return fail(c, AHMC_ERR_ARGUMENT, "gather_state: theta_all is NULL");
    if (c->x_ranks > 1) return fail(c, AHMC_ERR_UNSUPPORTED, "gather_state: the CPU checker gathers positions in the test harness");
    std::memcpy(theta_all, c->th.data(), sizeof(T) * c->D * c->N);
    return AHMC_OK;
  });
}

int32_t ahmc_ebfmi(ahmc_ctx* ctx, void* out) {
  FOR_CTX(ctx, {
    if (!out) return fail(c, AHMC_ERR_ARGUMENT, "ebfmi: out is NULL");
    T* o = static_cast<T*>(out);
    for (int64_t i = 0; i < c->N; ++i) {
      if (c->acc_energy.empty() || c->acc_energy[i] < 2) { o[i] = std::numeric_limits<T>::quiet_NaN(); continue; }
      const T n = c->acc_energy[i], sd2 = c->acc_energy[2 * c->N + i], M2 = c->acc_energy[4 * c->N + i];
      o[i] = (sd2 / (n - 1)) / (M2 / (n - 1));
    }
    return AHMC_OK;
  });
}

int32_t ahmc_ess(ahmc_ctx* ctx, const void* draws_v, int64_t K, void* out_v) {
  FOR_CTX(ctx, {
    if (!draws_v || !out_v) return fail(c, AHMC_ERR_ARGUMENT, "ess: NULL argument");
    if (K < 4) return fail(c, AHMC_ERR_ARGUMENT, "ess: at least 4 draws per chain");
    const T* draws = static_cast<const T*>(draws_v);
    T* out = static_cast<T*>(out_v);
    const int64_t DN = c->D * c->N;
    _Pragma("omp parallel for schedule(static)")
    for (int64_t s2 = 0; s2 < DN; ++s2) {
      double mean = 0, lo = (double)draws[s2], hi = lo;
      for (int64_t k = 0; k < K; ++k) {
        const double x = (double)draws[k * DN + s2];
        mean += x;
        lo = x < lo ? x : lo;
        hi = x > hi ? x : hi;
      }
      mean /= (double)K;
      if (!(hi > lo)) { out[s2] = (T)K; continue; }   // a series that never moved: K (decided on the values, not on a γ₀ of rounding noise)
      auto gamma = [&](int64_t t) {
        double g = 0;
        for (int64_t k = 0; k + t < K; ++k) g += ((double)draws[k * DN + s2] - mean) * ((double)draws[(k + t) * DN + s2] - mean);
        return g / (double)K;
      };
      const double g0 = gamma(0);
      if (!(g0 > 0)) { out[s2] = (T)K; continue; }
      double tau = -1, prev = 1e300;
      for (int64_t t = 0; 2 * t + 1 < K; ++t) {
        double P = (gamma(2 * t) + gamma(2 * t + 1)) / g0;
        if (!(P > 0)) break;
        P = P < prev ? P : prev;
        prev = P;
        tau += 2 * P;
      }
      if (tau < 1.0 / (double)K) tau = 1.0 / (double)K;
      out[s2] = (T)((double)K / tau);
    }
    return AHMC_OK;
  });
}

int32_t ahmc_get_info(ahmc_ctx* ctx, int32_t what, int64_t* out) {
  FOR_CTX(ctx, {
    if (!out) return fail(c, AHMC_ERR_ARGUMENT, "get_info: out is NULL");
    switch (what) {  // the scalar oracle has no thread geometry: one "lane" holding the whole chain
      case AHMC_INFO_GROUP_LANES: *out = 1; break;
      case AHMC_INFO_ELEMS_PER_LANE: *out = c->D; break;
      case AHMC_INFO_NUTS_LAUNCHES: case AHMC_INFO_NUTS_KERNEL_NS: case AHMC_INFO_NUTS_WARM_LAUNCHES: case AHMC_INFO_NUTS_WARM_KERNEL_NS:
      case AHMC_INFO_DENSE_GEMM_LAUNCHES: case AHMC_INFO_DENSE_GEMM_SMALL_LAUNCHES: case AHMC_INFO_DENSE_PIPELINES: case AHMC_INFO_DENSE_POOL: case AHMC_INFO_NUTS_DRAW_BATCH: case AHMC_INFO_DENSE_EPOCH_LAUNCHES: *out = 0; break;
      case AHMC_INFO_NUTS_BATCH: *out = 1; break;
      case AHMC_INFO_ITERATION: *out = (int64_t)c->iteration; break;
      case AHMC_INFO_STEPSIZE_SCALAR: *out = c->eps_scalar ? 1 : 0; break;
      default: return fail(c, AHMC_ERR_ARGUMENT, "get_info: unknown key");
    }
    return AHMC_OK;
  });
}

// ---- oracle-only probes used by the golden tests (pieces of src/trajectory.jl) ---------------
double ahmco_logaddexp(double a, double b) { return logaddexp(a, b); }
void ahmco_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
  Philox4 p = philox4x32_10(c0, c1, c2, c3, k0, k1);
  std::memcpy(out, p.v, sizeof(p.v));
}
double ahmco_uniform(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t purpose, uint32_t slot) {
  Rng g{(uint32_t)seed, (uint32_t)(seed >> 32), chain, iter};
  return g.uniform(purpose, slot);
}
double ahmco_normal(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t purpose, uint32_t d) {
  Rng g{(uint32_t)seed, (uint32_t)(seed >> 32), chain, iter};
  return g.normal(purpose, d);
}
// BinaryTree combine statistics (test/trajectory.jl:231-246)
void ahmco_tree_combine(double sa1, int64_t n1, double dh1, double sa2, int64_t n2, double dh2, double* sa, int64_t* n, double* dh) {
  BinaryTree<double> a, b;
  a.sum_alpha = sa1; a.n_alpha = n1; a.dH_max = dh1;
  b.sum_alpha = sa2; b.n_alpha = n2; b.dH_max = dh2;
  BinaryTree<double> t = combine(a, b);
  *sa = t.sum_alpha; *n = t.n_alpha; *dh = t.dH_max;
}
// Termination algebra (test/trajectory.jl:199-229)
int32_t ahmco_termination_mul(int32_t d1, int32_t n1, int32_t d2, int32_t n2) {
  return isterminated(Termination{d1 != 0, n1 != 0} * Termination{d2 != 0, n2 != 0}) ? 1 : 0;
}
// temper schedule (test/integrator.jl:89-106): returns the factor applied to r
double ahmco_temper_factor(double alpha, int64_t i, int32_t is_half, int64_t n_steps) {
  LeapfrogCfg<double> lf;
  lf.kind = AHMC_INTEGRATOR_TEMPERED;
  lf.alpha = alpha;
  std::vector<double> r(1, 1.0);
  temper(lf, r, i, is_half != 0, n_steps);
  return r[0];
}
// U-turn criteria on an explicit (left, right, rho) triple, unit metric (test/trajectory.jl:249-325)
int32_t ahmco_uturn(int32_t criterion, int64_t D, const double* thl, const double* rl, const double* thr, const double* rr, const double* rho) {
  MetricView<double> m{AHMC_METRIC_UNIT, D, nullptr, nullptr, nullptr};
  NutsCfg<double> cfg;
  cfg.criterion = criterion;
  NutsEnv<double> e{nullptr, nullptr, &m, &cfg, nullptr, 0};
  BinaryTree<double> t;
  auto zl = std::make_shared<PhasePoint<double>>(), zr = std::make_shared<PhasePoint<double>>();
  zl->th.assign(thl, thl + D); zl->r.assign(rl, rl + D);
  zr->th.assign(thr, thr + D); zr->r.assign(rr, rr + D);
  t.zleft = zl; t.zright = zr;
  t.rho.assign(rho, rho + D);
  if (criterion != AHMC_TC_STRICT) return uturn(e, t, t, t).dynamic ? 1 : 0;
  // the left/right subtrees of test/trajectory.jl:84-93: (z0, z0, rho - z1.r) and (z1, z1, rho - z0.r)
  BinaryTree<double> tl, tr;
  tl.zleft = t.zleft; tl.zright = t.zleft; tr.zleft = t.zright; tr.zright = t.zright;
  tl.rho.resize(D); tr.rho.resize(D);
  for (int64_t d = 0; d < D; ++d) { tl.rho[d] = rho[d] - rr[d]; tr.rho[d] = rho[d] - rl[d]; }
  return uturn(e, t, tl, tr).dynamic ? 1 : 0;
}
// multinomial sampler combine: returns ℓw and writes whether the first candidate is kept
double ahmco_multinomial_combine(double lw1, double lw2, double randexp, int32_t* keep_first) {
  double lw = logaddexp(lw1, lw2);
  *keep_first = (lw < lw1 + randexp) ? 1 : 0;
  return lw;
}
int32_t ahmco_slice_combine(int64_t n1, int64_t n2, double u) { return (double)(n1 + n2) * u < (double)n1 ? 1 : 0; }

}  // extern "C"
