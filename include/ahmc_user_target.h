/* ahmc_user_target.h — DOCUMENTATION ONLY: this header declares nothing (the user supplies the template it describes; the form with a
 * declared C symbol is ahmc_user_target_object.h).  How to hand the engine a log-density that runs ON THE DEVICE, inside the trajectory kernels.
 *
 * Reference surface replaced: the user callable `h.∂ℓπ∂θ(θ) -> (ℓπ, ∇ℓπ)` of /root/reference/src/hamiltonian.jl:45-48, built from a
 * LogDensityProblems object at src/AdvancedHMC.jl:163-186.  A Julia closure cannot run inside a HIP kernel behind a C ABI; the
 * three ways the engine takes a user density, fastest first:
 *
 *  1. TARGET PLUGIN (this header; ahmc_set_target_plugin): the density is a HIP device function.  `build_target_plugin`
 *     (advancedhmc.jl_amd/build.py; or by hand, see below) compiles the engine's own trajectory kernels — leapfrog, static HMC,
 *     NUTS incl. the fused warm-up, find_good_stepsize — with that function in place of a built-in family, for the element type
 *     and thread geometry of the context, into a small shared object the engine binds with dlopen.  Same kernels, same speed as
 *     the built-in families: no host round trip, no per-leapfrog launch.
 *  1b. TARGET OBJECT (ahmc_user_target_object.h; build_target_plugin_from_object): the same fused kernels for a density that exists only
 *     as COMPILED device code — a relocatable object or amdgcn LLVM bitcode (what GPUCompiler.jl emits for a Julia function) defining one C
 *     symbol; linked with the engine's kernels under device LTO (inlined), bound like a plugin.
 *  2. TARGET KERNEL (ahmc_set_target_kernel, ahmc_hip.h): the density is a device KERNEL the caller already has — a
 *     hipFunction_t (hipModuleGetFunction; what AMDGPU.jl compiles a Julia kernel to) or a __global__ symbol of the process —
 *     and the step-synchronous engine launches it itself between its tree kernels: one launch per leapfrog of all running
 *     chains, no host round trip.
 *  3. ASK / TELL (AHMC_TARGET_EXTERNAL, ahmc_ext_*): the caller evaluates (ℓπ, −∇ℓπ) wherever it likes; a host round trip per
 *     leapfrog.
 *
 * ---- the plugin contract --------------------------------------------------------------------------------------------------
 * A chain is a group of G consecutive lanes; lane `lane` (0 … G−1) holds the E consecutive elements d0 … d0+E−1 (d0 = lane·E) of
 * every D-vector of its chain in registers.  Elements with d ≥ D are padding: θ there is 0 and the gradient written there
 * MUST be 0.  Define, in namespace ahmc_user:
 *
 *     template <class T, int G, int E>
 *     __device__ T logdensity(const T* params, int D, const T (&theta)[E], T (&grad_neg)[E], int lane, int d0);
 *
 *   returns      this lane's PARTIAL of ℓπ(θ): the engine adds the partials of the G lanes (so a term that belongs to the chain
 *                as a whole goes into ONE lane's partial, e.g. `if (lane == 0) part += c;`)
 *   grad_neg[e]  = −∂ℓπ/∂θ_{d0+e}  (the sign the reference's `∂H∂θ` returns: DualValue(ℓπ, −∇ℓπ), src/hamiltonian.jl:45-48)
 *   params       the `n_params` values given to ahmc_set_target_plugin, in device memory (shared by all chains), or nullptr
 *
 * Cross-lane helpers (namespace ahmc, ahmc_device.hpp; all lanes of the chain must call them together):
 *     ahmc::group_sum1<G>(x)        Σ over the chain's lanes of x, result in every lane (bit-identical in all of them)
 *     ahmc::group_allsum<G>(v)      the same for a small array T v[K], in place, in one pass
 *     ahmc::group_bcast<G>(x, src)  lane src's x in every lane  (element d lives in lane d / E, slot d % E)
 * Non-finite ℓπ is handled by the engine as the reference does (→ −Inf, the point is rejected / divergent).
 *
 * Example (isotropic Gaussian, ℓπ = −½ Σ θ² − D/2·log 2π):
 *
 *     namespace ahmc_user {
 *     template <class T, int G, int E>
 *     __device__ T logdensity(const T*, int D, const T (&th)[E], T (&g)[E], int lane, int) {
 *       T ss = 0;
 *       for (int e = 0; e < E; ++e) { ss += th[e] * th[e]; g[e] = th[e]; }      // padding: θ = 0 → g = 0
 *       T part = -ss / 2;
 *       if (lane == 0) part -= (T)D * (T)0.91893853320467274178;                 // the chain's constant, once
 *       return part;
 *     }
 *     }
 *
 * By hand (what build_target_plugin runs; G, E from ahmc_get_info of the context):
 *     hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-gpu-rdc -I include \
 *           -DAHMC_INST_T=double -DAHMC_INST_TK=4 -DAHMC_PLUGIN_G=64 -DAHMC_PLUGIN_E=2 \
 *           -DAHMC_USER_TARGET_HEADER='"/abs/path/my_density.hpp"' -DAHMC_SOURCES_DIGEST='"<digest>"' \
 *           advancedhmc.jl_amd/csrc/ahmc_inst.hip -o libmy_density.so
 */
#ifndef AHMC_USER_TARGET_H
#define AHMC_USER_TARGET_H
#endif
