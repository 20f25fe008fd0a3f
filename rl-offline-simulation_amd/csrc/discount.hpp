// discount.hpp -- gamma**t for the discounted return G = G + gamma**t * R (offsim4rl/evaluators/psrs.py:262).
//
// The reference evaluates `gamma ** t` on the host (Python float ** int == libm pow), and Gs must match bit for bit,
// so the device never derives the factor itself: the caller passes a table gamma_pow[0..n) computed by the host's own
// pow.  Episodes are bounded by the log (t <= N), and for |gamma| < 1 the factor is exactly 0 from t ~ 7.4e4 on
// (gamma = 0.99), so a table that runs until the factor has become stationary covers every t:
//   t <  n                                  -> gamma_pow[t]
//   t >= n, table ends stationary           -> gamma_pow[n-1]   (last two entries equal and 0, +-inf or 1)
//   t >= n otherwise (caller's short table) -> device pow(): NOT guaranteed bit-identical to libm
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace offsim {

__device__ __noinline__ double discount_beyond_table(const double *__restrict__ gamma_pow, uint64_t n, double gamma, uint64_t t) {
    if (n >= 2) {
        const double a = gamma_pow[n - 1], b = gamma_pow[n - 2];
        if (a == b && (a == 0.0 || a == 1.0 || a == __builtin_inf() || a == -__builtin_inf())) return a;
    }
    return pow(gamma, (double)t);
}

__device__ __forceinline__ double discount_at(const double *__restrict__ gamma_pow, uint64_t n, double gamma, uint64_t t) {
    return t < n ? gamma_pow[t] : discount_beyond_table(gamma_pow, n, gamma, t);
}

}  // namespace offsim
