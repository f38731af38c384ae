/* ahmc_user_target_object.h — a user log-density handed over as a RELOCATABLE DEVICE OBJECT / LLVM bitcode instead of a header.
 *
 * Reference surface replaced: `h.∂ℓπ∂θ(θ) -> (ℓπ, ∇ℓπ)` of /root/reference/src/hamiltonian.jl:45-48, built from a LogDensityProblems
 * object at src/AdvancedHMC.jl:163-186 — for a density that exists only as COMPILED device code: what GPUCompiler.jl / AMDGPU.jl emit
 * for a Julia closure (LLVM bitcode for amdgcn-amd-amdhsa), or `hipcc -fgpu-rdc -c` of a C++ function.
 *
 * The object defines ONE symbol with C linkage per element type it supports (this header DECLARES them — it is the ABI):
 *
 *     double ahmc_user_logdensity_f64(const double* params, int D, int E, const double* theta, double* grad_neg, int lane, int d0, int G);
 *     float  ahmc_user_logdensity_f32(const float*  params, int D, int E, const float*  theta, float*  grad_neg, int lane, int d0, int G);
 *
 *   theta[0..E)     the E consecutive elements d0 … d0+E−1 of θ this lane holds (private memory; padding d ≥ D holds 0)
 *   grad_neg[0..E)  OUT: −∂ℓπ/∂θ_{d0+e} (the sign of the reference's ∂H∂θ); MUST be 0 for padding elements
 *   returns         this lane's PARTIAL of ℓπ(θ): the engine sums the partials of the chain's G lanes
 *   params          the n_params values given to ahmc_set_target_plugin (device memory, shared by all chains), or NULL
 *   lane, G, E, d0  the thread geometry of the context (ahmc_get_info: AHMC_INFO_GROUP_LANES / AHMC_INFO_ELEMS_PER_LANE); d0 = lane·E
 * A density with cross-lane terms needs the engine's reductions and is written as a header plugin (ahmc_user_target.h) instead.
 *
 * `build_target_plugin_from_object` (advancedhmc.jl_amd/build.py) compiles the engine's trajectory kernels with this header as
 * their log-density family (TK = 4) under -fgpu-rdc and LINKS them with the object at build time: the device link runs LTO over
 * both, so a bitcode object is inlined into the leaf loop — the same kernels and the same arithmetic as a header plugin — and the
 * result is bound through the same launch table (ahmc_set_target_plugin).  A machine-code object (no bitcode) links as well; its
 * function is then CALLED per leapfrog with theta / grad_neg in scratch memory (correct, slower).
 */
#ifndef AHMC_USER_TARGET_OBJECT_H
#define AHMC_USER_TARGET_OBJECT_H
#ifdef __HIPCC__
#define AHMC_USER_DEVICE __device__
#else
#define AHMC_USER_DEVICE
#endif
#ifdef __cplusplus
extern "C" {
#endif
AHMC_USER_DEVICE double ahmc_user_logdensity_f64(const double* params, int D, int E, const double* theta, double* grad_neg, int lane, int d0, int G);
AHMC_USER_DEVICE float ahmc_user_logdensity_f32(const float* params, int D, int E, const float* theta, float* grad_neg, int lane, int d0, int G);
#ifdef __cplusplus
}
#endif

#if defined(__HIPCC__) && defined(AHMC_USER_TARGET_FROM_OBJECT)
/* the engine's side: ahmc_user::logdensity<T, G, E> (the header-plugin contract) forwards to the object's symbol */
namespace ahmc_user {
template <class T, int G, int E>
__device__ __forceinline__ T logdensity(const T* params, int D, const T (&theta)[E], T (&grad_neg)[E], int lane, int d0) {
  if constexpr (sizeof(T) == 8) return ahmc_user_logdensity_f64(params, D, E, theta, grad_neg, lane, d0, G);
  else return ahmc_user_logdensity_f32(params, D, E, theta, grad_neg, lane, d0, G);
}
}  // namespace ahmc_user
#endif
#endif
