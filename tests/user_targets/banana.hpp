// A user log-density that is NOT a built-in family (ahmc_set_target_plugin; contract: include/ahmc_user_target.h):
// a "banana" (Rosenbrock-type) density with a coupling between neighbouring PAIRS of dimensions and two parameters,
//     ℓπ(θ) = − Σ_{k < D/2} [ (θ_{2k} − a)² / 2  +  b · (θ_{2k+1} − θ_{2k}²)² ]  −  (D odd: θ_{D−1}² / 2),     params = (a, b).
// Pairs never straddle two lanes when E is even (every default geometry for D > 4): no cross-lane traffic is needed.
namespace ahmc_user {
template <class T, int G, int E>
__device__ __forceinline__ T logdensity(const T* params, int D, const T (&th)[E], T (&grad_neg)[E], int /*lane*/, int d0) {
  static_assert(E % 2 == 0, "banana.hpp pairs dimensions (2k, 2k+1): it needs an even number of elements per lane");
  const T a = params[0], b = params[1];
  T part = 0;
#pragma unroll
  for (int e = 0; e < E; e += 2) {
    const int d = d0 + e;
    const T x = th[e], y = th[e + 1];
    if (d + 1 < D) {
      const T u = x - a, w = y - x * x;
      part -= u * u / 2 + b * w * w;
      grad_neg[e] = u - 4 * b * w * x;   // −∂ℓπ/∂x
      grad_neg[e + 1] = 2 * b * w;       // −∂ℓπ/∂y
    } else if (d < D) {                  // a last unpaired dimension: standard normal
      part -= x * x / 2;
      grad_neg[e] = x;
      grad_neg[e + 1] = 0;
    } else {
      grad_neg[e] = 0;
      grad_neg[e + 1] = 0;
    }
  }
  return part;
}
}  // namespace ahmc_user
