// A user log-density for ahmc_set_target_plugin (contract: include/ahmc_user_target.h): the isotropic Gaussian
// ℓπ = −½ Σ θ² − D/2·log 2π — the SAME arithmetic, in the same order, as the engine's built-in AHMC_TARGET_ISO_GAUSS, so the
// parity test can demand bit-identical chains from the plugin and the built-in family.
namespace ahmc_user {
template <class T, int G, int E>
__device__ __forceinline__ T logdensity(const T* /*params*/, int D, const T (&th)[E], T (&grad_neg)[E], int lane, int /*d0*/) {
  T ss = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    ss += th[e] * th[e];
    grad_neg[e] = th[e];  // −∂ℓπ/∂θ = θ (padding holds θ = 0, so its gradient is 0)
  }
  T part = -ss / 2;
  if (lane == 0) part -= (T)D * (T)1.8378770664093454835606594728112 / 2;  // the chain's constant, once
  return part;
}
}  // namespace ahmc_user
