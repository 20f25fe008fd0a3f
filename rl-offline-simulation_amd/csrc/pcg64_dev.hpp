// Device-side NumPy-compatible random streams for gfx950.
//
// np.random.default_rng(seed) == PCG64(SeedSequence(seed)); the reference draws its rejection
// uniforms and its queue shuffles from it (offsim4rl/evaluators/psrs.py:20,23,30,56).  rocRAND
// has no PCG64, so the generator is written out here: 128-bit LCG on two 64-bit halves, XSL-RR
// output, O(log k) jump-ahead so that lane k of a wavefront can own draw c+k.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace offsim {

struct U128 {
    uint64_t lo, hi;
};

__device__ __forceinline__ U128 u128(uint64_t hi, uint64_t lo) {
    U128 r;
    r.lo = lo;
    r.hi = hi;
    return r;
}
__device__ __forceinline__ U128 mul128(U128 a, U128 b) {
    U128 r;
    r.lo = a.lo * b.lo;
    r.hi = __umul64hi(a.lo, b.lo) + a.hi * b.lo + a.lo * b.hi;
    return r;
}
__device__ __forceinline__ U128 add128(U128 a, U128 b) {
    U128 r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + (r.lo < a.lo ? 1ull : 0ull);
    return r;
}

#define OFFSIM_PCG_MULT_HI 0x2360ED051FC65DA4ull
#define OFFSIM_PCG_MULT_LO 0x4385DF649FCCF645ull

__device__ __forceinline__ U128 pcg_mult() { return u128(OFFSIM_PCG_MULT_HI, OFFSIM_PCG_MULT_LO); }

// one LCG step
__device__ __forceinline__ U128 pcg_step(U128 s, U128 inc) { return add128(mul128(s, pcg_mult()), inc); }

// XSL-RR 128/64 output of a state (NumPy outputs the state AFTER the step)
__device__ __forceinline__ uint64_t pcg_output(U128 s) {
    uint64_t x = s.hi ^ s.lo;
    unsigned rot = (unsigned)(s.hi >> 58);
    return (x >> rot) | (x << ((64u - rot) & 63u));
}

// Affine map x -> mult*x + plus equal to `delta` LCG steps with increment inc.
struct Jump {
    U128 mult, plus;
};
__device__ inline Jump pcg_jump(U128 inc, uint64_t delta) {
    U128 acc_m = u128(0, 1), acc_p = u128(0, 0);
    U128 cur_m = pcg_mult(), cur_p = inc;
    while (delta > 0) {
        if (delta & 1) {
            acc_m = mul128(acc_m, cur_m);
            acc_p = add128(mul128(acc_p, cur_m), cur_p);
        }
        cur_p = mul128(add128(cur_m, u128(0, 1)), cur_p);
        cur_m = mul128(cur_m, cur_m);
        delta >>= 1;
    }
    Jump j;
    j.mult = acc_m;
    j.plus = acc_p;
    return j;
}
__device__ __forceinline__ U128 pcg_apply(Jump j, U128 s) { return add128(mul128(j.mult, s), j.plus); }

// SeedSequence(seed).generate_state(4, uint64) followed by PCG64 seeding.
struct PcgInit {
    U128 state, inc;
};
__device__ inline PcgInit pcg_seed(uint64_t seed) {
    const uint32_t MULT_A = 0x931e8875u, MULT_B = 0x58f38dedu, MIX_L = 0xca01f9ddu, MIX_R = 0x4973f715u;
    uint32_t hc = 0x43b0d7e5u;
    uint32_t pool[4];
    uint32_t e1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t v = (i == 0) ? (uint32_t)seed : ((i == 1 && e1 != 0) ? e1 : 0u);
        v ^= hc;
        hc *= MULT_A;
        v *= hc;
        v ^= v >> 16;
        pool[i] = v;
    }
#pragma unroll
    for (int s = 0; s < 4; s++) {
#pragma unroll
        for (int d = 0; d < 4; d++) {
            if (s != d) {
                uint32_t v = pool[s];
                v ^= hc;
                hc *= MULT_A;
                v *= hc;
                v ^= v >> 16;
                uint32_t m = MIX_L * pool[d] - MIX_R * v;
                m ^= m >> 16;
                pool[d] = m;
            }
        }
    }
    uint32_t hb = 0x8b51f9ddu;
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t v = pool[i & 3];
        v ^= hb;
        hb *= MULT_B;
        v *= hb;
        v ^= v >> 16;
        w[i] = v;
    }
    uint64_t q0 = (uint64_t)w[0] | ((uint64_t)w[1] << 32), q1 = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
    uint64_t q2 = (uint64_t)w[4] | ((uint64_t)w[5] << 32), q3 = (uint64_t)w[6] | ((uint64_t)w[7] << 32);
    PcgInit r;
    // inc = (initseq << 1) | 1 with initseq = q2:q3
    r.inc = u128((q2 << 1) | (q3 >> 63), (q3 << 1) | 1ull);
    U128 s = u128(0, 0);
    s = pcg_step(s, r.inc);
    s = add128(s, u128(q0, q1));
    s = pcg_step(s, r.inc);
    r.state = s;
    return r;
}

// Sequential generator with NumPy's buffered 32-bit halves (next_uint32): used by the shuffles.
struct PcgSeq {
    U128 state, inc;
    uint32_t hi32;
    bool has32;
    __device__ __forceinline__ void init(PcgInit p) {
        state = p.state;
        inc = p.inc;
        has32 = false;
        hi32 = 0;
    }
    __device__ __forceinline__ uint64_t next64() {
        state = pcg_step(state, inc);
        return pcg_output(state);
    }
    __device__ __forceinline__ uint32_t next32() {
        if (has32) {
            has32 = false;
            return hi32;
        }
        uint64_t n = next64();
        has32 = true;
        hi32 = (uint32_t)(n >> 32);
        return (uint32_t)n;
    }
    // random_interval(max) for max < 2**32: smallest all-ones mask >= max, redraw while above
    __device__ __forceinline__ uint32_t interval32(uint32_t max) {
        uint32_t mask = 0xffffffffu >> __clz((int)max);
        uint32_t v;
        do {
            v = next32() & mask;
        } while (v > max);
        return v;
    }
};

}  // namespace offsim
