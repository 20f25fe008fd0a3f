// philox_dev.hpp -- the rocRAND provider of the rejection stream (OFFSIM_STREAM_PHILOX, include/offsim.h) on the device.
//
// Draw i of a rollout is rocrand_uniform_double of the engine (seed, subsequence 0) positioned on 32-bit output 2 i, i.e. u = k * 2^-53
// with k = (v1 | (v2 >> 11) << 32) + 1 in [1, 2^53] (rocrand_uniform.h): u in (0, 1].  Two forms:
//   philox_k53      the literal device API (rocrand_init / rocrand), k in [1, 2^53]: the generic kernels, which compare the double itself;
//   offsim_philox_* the same engine's rounds (rocrand_device::philox4x32_10_engine::ten_rounds) reached through a derived class, because
//                   the public pair indexes the engine state by a run-time sub-position (private memory: scratch loads in a loop that
//                   generates draws all the time).  Key and counter are set as rocrand_init(seed, 0, 4 m) sets them -- key = the seed's
//                   two halves, counter = m -- so block(seed, m) is the four 32-bit outputs 4 m .. 4 m + 3, i.e. draws 2 m and 2 m + 1.
//                   For the compiled-policy scans, which compare k with the 53-bit key T = the largest k with k * 2^-53 <= ratio
//                   (psrs.py:55-57 folded by offsim_compile_policy): "accept iff k <= T" is the reference's rule for every k below 2^53;
//                   k = 2^53 (u = 1.0 exactly, probability 2^-53 per draw) is looked at as 2^53 - 1, which differs from the rule only
//                   against an importance ratio of exactly 1 - 2^-53 (accepted here, rejected there).
// tests/test_gpu_round6.py holds the two forms against each other, against the oracle fed the same stream and against the reference's
// own PSRS.step under a replayed stream (tests/golden/philox_*.npz).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <rocrand/rocrand_kernel.h>

namespace offsim {

__device__ __forceinline__ uint64_t philox_k53(uint64_t seed, uint64_t i) {
    rocrand_state_philox4x32_10 st;
    rocrand_init(seed, 0ull, 2ull * i, &st);
    const uint32_t v1 = rocrand(&st), v2 = rocrand(&st);
    return ((uint64_t)v1 | ((uint64_t)(v2 >> 11) << 32)) + 1ull;
}

struct OffsimPhilox : rocrand_device::philox4x32_10_engine {
    __device__ __forceinline__ OffsimPhilox() {}
    __device__ __forceinline__ uint4 block(uint64_t seed, uint64_t m) {
        const uint2 key = {(unsigned int)seed, (unsigned int)(seed >> 32)};
        const uint4 ctr = {(unsigned int)m, (unsigned int)(m >> 32), 0u, 0u};
        return this->ten_rounds(ctr, key);
    }
};
__device__ __forceinline__ uint64_t offsim_philox_clamp(uint32_t v1, uint32_t v2) {
    const uint64_t k = ((uint64_t)v1 | ((uint64_t)(v2 >> 11) << 32)) + 1ull;
    return k > 0x1fffffffffffffull ? 0x1fffffffffffffull : k;
}
// both draws of one block: draws 2 m and 2 m + 1
__device__ __forceinline__ void offsim_philox_pair(uint64_t seed, uint64_t m, uint64_t &k0, uint64_t &k1) {
    OffsimPhilox e;
    const uint4 v = e.block(seed, m);
    k0 = offsim_philox_clamp(v.x, v.y);
    k1 = offsim_philox_clamp(v.z, v.w);
}
__device__ __forceinline__ uint64_t offsim_philox_k(uint64_t seed, uint64_t i) {
    uint64_t k0, k1;
    offsim_philox_pair(seed, i >> 1, k0, k1);
    return (i & 1ull) ? k1 : k0;
}

}  // namespace offsim
