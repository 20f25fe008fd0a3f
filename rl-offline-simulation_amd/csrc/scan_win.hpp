// scan_win.hpp -- the fast evalMC scan: compiled-policy keys + per-state candidate windows in LDS.
//
// Why: a PSRS rollout is one long dependent chain (state -> head of that state's queue -> accept ->
// next state).  Reading the queue head from HBM on every step costs a DRAM round trip (two with a
// per-rollout permutation) per simulated step.  Here the next few candidates of EVERY state's queue
// are kept in an LDS window per rollout, so the chain runs at LDS/ALU latency, and the HBM traffic
// (permutation indices + 8-byte candidate keys) is issued in lane-parallel refill phases every 64
// accepted steps, two phases ahead of use (software pipeline A: indices, B: keys, C: land in LDS).
//
// Compiled policy (offsim_compile_policy): for a fixed tabular policy pi the rejection test of
// psrs.py:53-57 depends only on the row, so it is folded into one 64-bit key per grouped row:
//     key = [T>>32 : 21 | done : 1 | z_next_slot : 10 | T & 0xffffffff : 32],  T = floor(thr * 2^53),
//     thr = pi[z][a]/p_log[a]/M
// and  reject  <=>  u > thr  <=>  k53 > T  with u = k53 * 2^-53 (exact; NaN/>=1 thr -> T = 2^53-1).
// Windows hold the high dword (the "digest": top 21 bits of T | done | z_next); a
// draw whose top 21 bits tie with the digest (p = 2^-21) is resolved exactly on the slow path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "discount.hpp"
#include "offsim.h"
#include "pcg64_dev.hpp"
#include "philox_dev.hpp"

namespace offsim {

typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef uint32_t scan_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) scan_u32x2 lds_u32x2;

#define OFFSIM_RING 128  // draws kept ahead per rollout (power of two >= 2*64)
#define OFFSIM_PH 64     // accepted steps per refill phase (lane i stages step i's reward)

__device__ __forceinline__ uint64_t pack_key(double thr, int done, int zn) {
    uint64_t T;
    if (thr != thr || thr >= 1.0) T = (1ull << 53) - 1;  // NaN compares false -> never rejected; u < 1 <= thr
    else if (thr < 0.0) T = 0;                          // cannot happen for probabilities
    else T = (uint64_t)floor(thr * 9007199254740992.0);  // exact power-of-two scaling
    return ((T >> 32) << 43) | ((uint64_t)(done ? 1 : 0) << 42) | ((uint64_t)(zn & 1023) << 32) | (T & 0xffffffffull);
}
__device__ __forceinline__ uint64_t key_T(uint64_t key) { return ((key >> 43) << 32) | (key & 0xffffffffull); }

// rare paths kept out of line so that they do not inflate the register budget of the chain loop
__device__ __forceinline__ uint64_t exact_draw53(U128 base, U128 inc, uint64_t n_steps) {
    return pcg_output(pcg_apply(pcg_jump(inc, n_steps), base)) >> 11;
}

template <typename PL>
__global__ void k_compile_policy(offsim_table t, const double *__restrict__ pi, uint64_t *__restrict__ keys) {
    extern __shared__ uint32_t seg_lds[];
    for (int i = threadIdx.x; i <= t.n_slots; i += blockDim.x) seg_lds[i] = t.seg_off[i];
    __syncthreads();
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= t.N) return;
    int lo = 0, hi = t.n_slots;  // slot with seg_off[slot] <= g < seg_off[slot+1]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (seg_lds[mid] <= (uint32_t)g) lo = mid;
        else hi = mid;
    }
    const double *pnew = pi + (size_t)lo * t.nA;
    const PL *plog = (const PL *)t.p_log;
    const int nA = t.nA, a = t.a[g];
    double M = -__builtin_inf();
    bool nan = false;
    for (int k = 0; k < nA; k++) {
        double q = pnew[k] / plog_f64<PL>(plog, g * nA + k);
        nan |= (q != q);
        M = q > M ? q : M;
    }
    if (nan) M = __builtin_nan("");
    double thr = pnew[a] / plog_f64<PL>(plog, g * nA + a) / M;
    keys[g] = pack_key(thr, t.done[g], t.z_next[g]);
}

template <int ROUNDS>
__device__ __forceinline__ uint32_t rd_lane(const uint32_t (&v)[ROUNDS], int q, int l) {
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < ROUNDS; i++) {
        uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)v[i], l);
        r = (q == i) ? x : r;
    }
    return r;
}
template <int ROUNDS>
__device__ __forceinline__ void wr_lane(uint32_t (&v)[ROUNDS], int q, int l, uint32_t val) {
    const int lane = threadIdx.x & 63;  // (this clang has no writelane builtin: predicated move instead)
#pragma unroll
    for (int i = 0; i < ROUNDS; i++) v[i] = (q == i && lane == l) ? val : v[i];
}

// W = window entries per state, ROUNDS = ceil(n_slots / 64), D = W/2 entries requested per state per phase.
//
// The chain loop is written for the scalar unit: one CU has ONE scalar ALU for its 16 rollouts, and the
// first version of this loop was bound by it (84 SALU instructions per simulated step, rocprofv3 PMC).
// So the per-state cursor lives in LDS (read as a per-lane value: its arithmetic is VALU), rewards and
// discount exponents are reconstructed at the phase boundary, and only what is inherently wave-uniform
// (ballot masks, the draw counter, the current state) stays scalar.
//
// LDS per block: [seg_off (n_slots+1) u32, shared] then per wave [win n_slots*W u32][ring 128 u32][meta n_slots x {cur, landed}]
// RNG (round 6): the provider of the rejection stream, as in the row-packed scan (OFFSIM_STREAM_PCG64: NumPy's default_rng, the
// reference's numbers; OFFSIM_STREAM_PHILOX: rocRAND's Philox4x32-10, csrc/philox_dev.hpp) -- the ring's fill, the exact look's draw
// and the stream state written back are all that differ.
template <int W, int ROUNDS, bool TRACE, int RNG = OFFSIM_STREAM_PCG64>
__global__ void __launch_bounds__(256, 4)
    k_eval_mc_win(offsim_table t, offsim_rollouts ro, const uint64_t *__restrict__ keys, double gamma,
                  const double *__restrict__ gamma_pow, int64_t n_gamma_pow64, int64_t max_episodes64, offsim_evalmc_out out) {
    constexpr uint32_t TICK = W > 16 ? 16u : 32u;  // accepted steps between refill passes (few busy states: refill sooner)
    constexpr uint32_t TICK_LOG = W > 16 ? 4u : 5u;
    constexpr int D = W > 8 ? 8 : 4;  // entries one request may bring (a visit consumes ~2-4 candidates; bigger requests cost more than they save)
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int waves = blockDim.x / 64;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64), lane = threadIdx.x & 63;  // wave id is uniform: keep it scalar
    const int n_slots = t.n_slots;
    // Every per-wave region starts on a 512-byte LDS address: the draw ring (512 B) and the window rows (W*4 B each) are
    // then naturally aligned, and the chain loop forms their addresses with one add-shift and one and-or.
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_byte *)lds_raw;
    const uint32_t lds_pad = (0u - lds_base) & 511u;
    uint32_t *seg = (uint32_t *)(lds_raw + lds_pad);
    const uint32_t seg_bytes = ((uint32_t)(n_slots + 1) * 4 + 511u) & ~511u;
    const uint32_t win_bytes = (uint32_t)n_slots * W * 4;
    const uint32_t log_bytes = OFFSIM_PH * 8u + (TRACE ? OFFSIM_PH * 4u : 0u);  // per-phase step log (+ candidates popped per step)
    const uint32_t wave_bytes = (OFFSIM_RING * 4 + win_bytes + (uint32_t)n_slots * 16 + log_bytes + 511u) & ~511u;
    const uint32_t ring_off = lds_base + lds_pad + seg_bytes + (uint32_t)wave * wave_bytes;  // LDS byte address of this wave's region
    uint32_t *ring = (uint32_t *)(lds_raw + lds_pad + seg_bytes + (size_t)wave * wave_bytes);
    uint32_t *win = ring + OFFSIM_RING;
    const uint32_t win_off = ring_off + OFFSIM_RING * 4;
    uint2 *meta = (uint2 *)((unsigned char *)win + win_bytes);  // .x = cur (candidates popped), .y = landed (window valid up to)
    const uint32_t meta_off = win_off + win_bytes;
    uint32_t *fillq = (uint32_t *)(meta + n_slots);  // queue position up to which entries have been requested
    uint32_t *claim = fillq + n_slots;               // refill ownership: which lane requests for a state this tick
    // step log of the current phase, entry i = {cursor behind the accepted candidate, digest with the z_next field
    // replaced by the state the step left}: written by the chain loop with one LDS store per step (keeping it in
    // per-lane registers cost five VALU instructions per step), read back lane = step by the refill pass and the flush
    const uint32_t log_off = meta_off + (uint32_t)n_slots * 16u;
    uint32_t *popq = (uint32_t *)(claim + n_slots) + OFFSIM_PH * 2;  // TRACE only
    for (int i = threadIdx.x; i <= n_slots; i += blockDim.x) seg[i] = t.seg_off[i];
    __syncthreads();
    const int r = blockIdx.x * waves + wave;
    if (r >= ro.R) return;

    const uint32_t *perm_row = ro.perm ? ro.perm + (int64_t)r * ro.perm_stride : nullptr;
    const uint32_t *init_row = ro.init_perm ? ro.init_perm + (int64_t)r * ro.init_stride : nullptr;
    uint32_t *cur_glb = ro.cursor + (int64_t)r * n_slots;
    const uint32_t *keys32 = (const uint32_t *)keys;
    const uint32_t N0 = (uint32_t)t.N0;
    const uint32_t n_gamma_pow = (uint32_t)(n_gamma_pow64 > 0x7fffffffll ? 0x7fffffffll : n_gamma_pow64);
    const uint32_t max_episodes = (uint32_t)(max_episodes64 > 0xffffffffll ? 0xffffffffll : max_episodes64);

    // ---- priming: every state's window is filled once, synchronously (lane = state) ----
#pragma unroll
    for (int q = 0; q < ROUNDS; q++) {
        const int s = q * 64 + lane;
        if (s < n_slots) {
            const uint32_t c0 = cur_glb[s], beg_s = seg[s], len_s = seg[s + 1] - beg_s;
            const uint32_t left = len_s - c0, want = left < (uint32_t)W ? left : (uint32_t)W;
#pragma unroll
            for (int e = 0; e < W; e++) {
                if ((uint32_t)e < want) {
                    const uint32_t p = beg_s + c0 + e;
                    const uint32_t g = perm_row ? perm_row[p] : p;
                    win[(uint32_t)s * W + (c0 + e) % W] = keys32[2 * (size_t)g + 1];
                }
            }
            meta[s] = make_uint2(c0, c0 + want);
            fillq[s] = c0 + want;
        }
    }

    // ---- refill pipeline, demand driven: the lane that logged a step also tops up the window of the state that step
    // left.  Every 32 accepted steps one pass runs all three stages for different lane groups: the lanes of two ticks
    // ago land their digests in LDS (C), the lanes of the previous tick gather digests through the indices that have
    // arrived (B), the lanes of this tick request indices (A).  A request lands 2 ticks later; HBM latency is ~10x
    // shorter than a tick, so no stage ever waits.
    uint32_t rq_state = 0, rq_slot = 0, rq_pos = 0, rq_cnt = 0;
    uint32_t idxA[D], digB[D];
#pragma unroll
    for (int e = 0; e < D; e++) idxA[e] = digB[e] = 0;

    // ---- rejection stream: ring of the top 21 bits of the next draws ----
    constexpr bool philox = RNG == OFFSIM_STREAM_PHILOX;
    U128 lane_state = u128(0, 0);
    Jump j64;
    j64.mult = u128(0, 1);
    j64.plus = u128(0, 0);
    const uint64_t ph_seed = philox ? ro.rng[4 * r + 0] : 0ull, ph_c0 = philox ? ro.rng[4 * r + 1] : 0ull;  // Philox: (seed, draws consumed so far, 0, 0)
    if (!philox) {
        const U128 base = u128(ro.rng[4 * r + 0], ro.rng[4 * r + 1]);
        const U128 inc = u128(ro.rng[4 * r + 2], ro.rng[4 * r + 3]);
        j64 = pcg_jump(inc, 64);
        lane_state = pcg_apply(pcg_jump(inc, (uint64_t)lane + 1), base);  // yields draw `lane`
    }
    uint32_t gen = 0, c = 0;  // draws generated / consumed since kernel start (every examined candidate = one draw)
    auto gen_block = [&]() {
        if constexpr (philox) {
            // lanes 0..31 each own one block = two draws (lanes 32..63 repeat them); a stream position of odd parity pairs the draws
            // across blocks: every lane then works out its two draws one by one (a rollout that was stepped before: rare)
            const uint64_t d0 = ph_c0 + (uint64_t)gen + 2u * ((uint32_t)lane & 31u);
            uint64_t k0, k1;
            if ((ph_c0 & 1ull) == 0ull) {
                offsim_philox_pair(ph_seed, d0 >> 1, k0, k1);
            } else {
                k0 = offsim_philox_k(ph_seed, d0);
                k1 = offsim_philox_k(ph_seed, d0 + 1ull);
            }
            const uint32_t at = (gen + 2u * ((uint32_t)lane & 31u)) & (OFFSIM_RING - 1);  // (gen is a multiple of 64: the pair never wraps)
            ring[at] = (uint32_t)(k0 >> 32) << 11;  // top 21 bits of the 53-bit draw, aligned with the digest's T21
            ring[at + 1u] = (uint32_t)(k1 >> 32) << 11;
        } else {
            ring[(gen + lane) & (OFFSIM_RING - 1)] = (uint32_t)(pcg_output(lane_state) >> 43) << 11;  // top 21 bits, aligned with the digest's T21
            lane_state = pcg_apply(j64, lane_state);
        }
        gen += 64;
    };
    gen_block();
    gen_block();
    auto exact53 = [&](uint32_t n_steps) -> uint64_t {  // the 53-bit draw that is the n_steps-th of this launch (n_steps >= 1)
        if constexpr (philox) {
            return offsim_philox_k(ph_seed, ph_c0 + (uint64_t)n_steps - 1ull);
        } else {
            const U128 base = u128(ro.rng[4 * r + 0], ro.rng[4 * r + 1]);
            const U128 inc = u128(ro.rng[4 * r + 2], ro.rng[4 * r + 3]);
            return pcg_output(pcg_apply(pcg_jump(inc, n_steps), base)) >> 11;
        }
    };

    // ---- initial-state prefetch: lane i holds the slot of init index ib+i ----
    uint32_t ic = ro.init_cursor[r], ib = ic;
    int init_reg = -1;
    auto load_init = [&]() {
        ib = ic;
        uint32_t k = ic + lane;
        int v = -1;
        if (k < N0) v = t.init_slot[init_row ? init_row[k] : k];
        init_reg = v;
    };
    load_init();

    int slot = ro.cur_slot[r];
    uint32_t ep = 0, ep_acc = 0, n_len = 0, steps = 0, len_acc = 0;
    uint32_t n_dry = 0, n_tie = 0, n_flush = 0;
    double sum_g = 0.0, G = 0.0;
    int status = OFFSIM_ST_OK;
    // per-phase log: lane i remembers where the phase's i-th accepted step came from
    uint32_t pos_log = 0, pop_log = 0, dig_log = 0;
    uint32_t nph = 0, pop_acc = 0;
    const bool r64 = t.r_dtype == OFFSIM_F64;

    // Reward pipeline, three phases deep so that no phase boundary waits on HBM:
    //   R1 (phase k ends):   issue g = perm[p] and gamma**t for the phase's accepted steps
    //   R2 (one phase later): issue r[g]        R3 (two phases later): accumulate returns in step order
    uint32_t g1 = 0;                     // R1 -> R2
    double gp1 = 0.0, gp2 = 0.0, rv2 = 0.0;  // R1 -> R2 -> R3
    uint32_t pop1 = 0;
    uint64_t dm1 = 0, dm2 = 0;
    uint32_t n1 = 0, n2 = 0, st1 = 0;
    uint32_t tt_chain = 0;               // accepted steps since the last episode end, as of the end of the last logged phase
    uint32_t ticks_done = 0;
    uint32_t slot_log = 0;  // (declared here: the refill pass reads the states the logged steps left)
    auto load_log = [&]() {  // lane i <- step i of the phase (valid for lanes < nph)
        const scan_u32x2 e = *(lds_u32x2 *)(log_off + (uint32_t)lane * 8u);
        pos_log = e.x - 1u;
        dig_log = e.y;
        slot_log = e.y & 1023u;
        if (TRACE) pop_log = popq[lane];
    };
    auto refill_tick = [&](uint32_t lo, uint32_t hi) {
        load_log();
        if (rq_state == 2) {  // C: land
            const uint2 mm = meta[rq_slot];
            const uint32_t wbase = rq_slot * W;
#pragma unroll
            for (int e = 0; e < D; e++) {
                const uint32_t pos = rq_pos + e;
                if ((uint32_t)e < rq_cnt && pos >= mm.x) win[wbase + pos % W] = digB[e];
            }
            const uint32_t end = rq_pos + rq_cnt;
            if (rq_pos <= mm.y && end > mm.y) meta[rq_slot].y = end;
            rq_state = 0;
        }
        if (rq_state == 1) {  // B: gather the digests (high dword of each key)
#pragma unroll
            for (int e = 0; e < D; e++)
                if ((uint32_t)e < rq_cnt) digB[e] = keys32[2 * (size_t)idxA[e] + 1];
            rq_state = 2;
        }
        const bool in_tick = (uint32_t)lane >= lo && (uint32_t)lane < hi;  // A: one request per state visited in this tick
        if (in_tick) claim[slot_log] = (uint32_t)lane;
        if (in_tick && claim[slot_log] == (uint32_t)lane) {
            const uint32_t s_ = slot_log;
            const uint32_t cur_s = meta[s_].x, beg_s = seg[s_], len_s = seg[s_ + 1] - beg_s;
            uint32_t f = fillq[s_];
            f = f < cur_s ? cur_s : f;
            const uint32_t have = f - cur_s;
            const uint32_t room = have < (uint32_t)W ? (uint32_t)W - have : 0u;
            const uint32_t left = len_s - f;
            uint32_t want = room < left ? room : left;
            want = want < (uint32_t)D ? want : (uint32_t)D;
            if (want) {
#pragma unroll
                for (int e = 0; e < D; e++) {
                    if ((uint32_t)e < want) {
                        const uint32_t p = beg_s + f + e;
                        idxA[e] = perm_row ? perm_row[p] : p;
                    }
                }
                rq_slot = s_;
                rq_pos = f;
                rq_cnt = want;
                rq_state = 1;
                fillq[s_] = f + want;
            }
        }
    };
    auto flush = [&]() {
        load_log();
        // ---- uses first: everything consumed here was requested at least one phase ago ----
        {   // R3: in-order discounted-return accumulation (psrs.py:262-269).  The sums must be sequential (bit-exact Gs),
            // so the inner loop is just two v_readlane and one v_add_f64 per step; episode ends (about one per phase)
            // are handled between runs of steps instead of being tested at every step.
            const double prod = gp2 * rv2;  // product first, then the running sum in step order
            uint32_t i = 0;
            uint64_t dm = dm2;
            while (i < n2) {
                const uint32_t e = dm ? (uint32_t)__ffsll((unsigned long long)dm) - 1u : n2;  // next episode end (index) or none
                const uint32_t run_end = e < n2 ? e + 1u : n2;
                len_acc += run_end - i;
                for (; i + 4u <= run_end; i += 4u) {  // (unrolled: the loop control costs as much as the additions)
                    G = G + readlane_f64(prod, (int)i);
                    G = G + readlane_f64(prod, (int)i + 1);
                    G = G + readlane_f64(prod, (int)i + 2);
                    G = G + readlane_f64(prod, (int)i + 3);
                }
                for (; i < run_end; i++) G = G + readlane_f64(prod, (int)i);
                if (e < n2) {
                    if (lane == 0) {
                        if (out.ep_g && (int64_t)ep_acc < out.ep_cap) out.ep_g[(int64_t)r * out.ep_cap + ep_acc] = G;
                        if (out.ep_len && (int64_t)n_len <= out.ep_cap) out.ep_len[(int64_t)r * (out.ep_cap + 1) + n_len] = (int32_t)len_acc;
                    }
                    sum_g += G;
                    ep_acc++;
                    n_len++;
                    G = 0.0;
                    len_acc = 0;
                    dm &= dm - 1ull;  // clear that done bit
                }
            }
        }
        // ---- then the new requests ----
        {   // R2
            double rv = 0.0;
            if ((uint32_t)lane < n1) {
                rv = r64 ? ((const double *)t.r)[g1] : (double)((const float *)t.r)[g1];
                if (TRACE) {
                    const uint32_t st = st1 + lane;
                    if (out.trace_row && (int64_t)st < out.trace_cap) out.trace_row[(int64_t)r * out.trace_cap + st] = t.orig_idx[g1];
                    if (out.trace_pop && (int64_t)st < out.trace_cap) out.trace_pop[(int64_t)r * out.trace_cap + st] = pop1;
                }
            }
            rv2 = rv;
            gp2 = gp1;
            dm2 = dm1;
            n2 = n1;
        }
        {   // R1
            const uint64_t done_mask = __ballot((uint32_t)lane < nph && ((dig_log >> 10) & 1u));  // episode ends of this phase
            uint32_t g = 0;
            double gp = 0.0;
            if ((uint32_t)lane < nph) {
                const uint32_t p = seg[slot_log] + pos_log;  // grouped position in this rollout's queue order
                g = perm_row ? perm_row[p] : p;              // accepted row, through the rollout's permutation
                const uint64_t below = done_mask & ((1ull << lane) - 1ull);  // episode ends earlier in this phase
                const uint32_t t_log = below ? (uint32_t)lane - 1u - (63u - (uint32_t)__clzll((long long)below)) : tt_chain + (uint32_t)lane;
                gp = discount_at(gamma_pow, n_gamma_pow, gamma, t_log);
            }
            g1 = g;
            gp1 = gp;
            pop1 = pop_log;
            dm1 = done_mask;
            n1 = nph;
            st1 = steps;
            tt_chain = done_mask ? nph - 1u - (63u - (uint32_t)__clzll((long long)done_mask)) : tt_chain + nph;
            steps += nph;
            nph = 0;
        }
        n_flush++;
    };


    // ---- the chain ----
    // Written as a straight-line fast loop (accept from the window) that leaves through ONE rarely-taken branch per
    // kind of event; a single wave pays ~10 cycles per dependent instruction and ~25 per taken branch, so the shape of
    // this loop, not memory, sets the kernel time.
    bool need_reset = true, dn = false;
    uint2 m = make_uint2(0u, 0u);
    uint32_t kt = 0;
    for (;;) {
        if (need_reset) {  // env.reset() at the start of every episode (psrs.py:249)
            if (ep >= max_episodes) break;
            if (ic >= N0) {  // psrs.py:33-35, 250-252
                status = OFFSIM_ST_NO_INIT;
                slot = -1;
                break;
            }
            if (ic - ib >= 64) load_init();
            slot = __builtin_amdgcn_readlane(init_reg, (int)(ic - ib));
            ic++;
            need_reset = false;
            dn = false;
            m = meta[slot];
            kt = ring[(c + lane) & (OFFSIM_RING - 1)];
        }
        // fast loop: one iteration = one accepted step served from the LDS window.  The four wavefronts of a SIMD share
        // its issue port and the sixteen of a CU its scalar unit, and the kernel time follows the NUMBER of instructions in
        // this loop, vector or scalar alike (measured: 47 -> 41 -> 45 instructions gave 1.90 -> 1.74 -> 1.88 s).  So: the
        // ring holds the draws pre-shifted (k21 << 11), which lets one compare against the whole digest find the first lane
        // that is not a clear reject; only that lane's tie test is done, on the scalar side; per-state addresses are kept in
        // vector registers; the step log is one LDS store; each kind of event has its own rarely taken exit.
        uint64_t many = 0;
        bool accepted;  // the last look accepted a candidate (the loop was left for an event)
        uint32_t vslot, vlog;
        asm("v_mov_b32 %0, %1" : "=v"(vslot) : "s"(slot));
        asm("v_mov_b32 %0, %1" : "=v"(vlog) : "s"(log_off + nph * 8u));  // LDS address of the log entry of the next step
        uint32_t vrow = win_off + vslot * (uint32_t)(W * 4), vmeta = meta_off + vslot * 8u;  // LDS addresses of its window row and meta entry
        uint32_t v1023, vringm;
        asm("v_mov_b32 %0, 0x3ff" : "=v"(v1023));
        asm("v_mov_b32 %0, %1" : "=v"(vringm) : "s"(OFFSIM_RING * 4u - 4u));
        uint32_t *plog = popq + nph;
        const uint32_t nph_in = nph;
        int tick_b = (int)((TICK - 1u) - (nph & (TICK - 1u)));  // goes negative when a multiple of TICK steps has been logged
        const int b_in = tick_b;
        const uint32_t gen_m64 = gen - 64u;
        for (;;) {
            // candidate `lane` of the current state sits at ring position (cur + lane) mod W of its row; entries that have
            // not landed (and lanes >= W) get an empty digest: they never win, kt <= 0 only as a tie
            const uint32_t v_avail = m.y - m.x;
            uint32_t dig = *(lds_u32 *)((((m.x + (uint32_t)lane) << 2) & (uint32_t)(W * 4 - 4)) | vrow);
            dig = (uint32_t)lane < (v_avail < (uint32_t)W ? v_avail : (uint32_t)W) ? dig : 0u;
            many = __ballot(kt <= dig);  // not a clear reject: k21 <= T21 (the low 11 bits of kt are zero)
            int f;
            asm("s_ff1_i32_b64 %0, %1" : "=s"(f) : "s"(many));  // -1 if none: lane 63 is read below, its digest is empty
            const uint32_t acc_dig = (uint32_t)__builtin_amdgcn_readlane((int)dig, f);
            const uint32_t acc_kt = (uint32_t)__builtin_amdgcn_readlane((int)kt, f);
            accepted = acc_kt < (acc_dig & 0xfffff800u);  // clear accept: k21 < T21
            if (__builtin_expect(!accepted, 0)) break;
            const uint32_t f1 = (uint32_t)f + 1u;
            const uint32_t cur1 = m.x + f1;
            *(lds_u32 *)vmeta = cur1;  // meta[vslot].x
            {
                scan_u32x2 e;
                e.x = cur1;
                asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(e.y) : "v"(v1023), "v"(vslot), "s"(acc_dig));  // digest with z_next replaced by the state left
                *(lds_u32x2 *)vlog = e;
                vlog += 8u;
            }
            c += f1;
            if (TRACE) {
                *plog++ = pop_acc + f1;
                pop_acc = 0;
            }
            tick_b -= 1;
            asm("v_bfe_u32 %0, %1, 0, 10" : "=v"(vslot) : "s"(acc_dig));  // next state
            vrow = win_off + vslot * (uint32_t)(W * 4);
            vmeta = meta_off + vslot * 8u;
            {
                uint32_t ka = (c + (uint32_t)lane) << 2;
                asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(ka) : "v"(ka), "v"(vringm), "s"(ring_off));
                kt = *(lds_u32 *)ka;  // fetched one step ahead: needs only the new draw count
            }
            {
                const scan_u32x2 mv = *(lds_u32x2 *)vmeta;  // and the next state
                m.x = mv.x;
                m.y = mv.y;
            }
            dn = (acc_dig >> 10) & 1u;
            if (__builtin_expect(dn, 0)) break;                                             // episode end
            if (__builtin_expect((int)((uint32_t)tick_b | (gen_m64 - c)) < 0, 0)) break;  // a multiple of 32 steps logged | fewer than 64 draws left
        }
        slot = (int)__builtin_amdgcn_readfirstlane(vslot);
        nph = nph_in + (uint32_t)(b_in - tick_b);
        if (accepted) goto ev_tail;  // left for a rare event (dn says whether the episode ended)
        dn = false;
        if (many == 0ull) {
            if (__builtin_amdgcn_readfirstlane(m.y - m.x) != 0u) {  // every window candidate rejected: consume them, look again
                const uint32_t v_avail = m.y - m.x;
                const uint32_t d = __builtin_amdgcn_readfirstlane(v_avail < (uint32_t)W ? v_avail : (uint32_t)W);
                meta[slot].x = m.x + d;
                c += d;
                if (TRACE) pop_acc += d;
                goto ev_tail;
            }
            n_dry++;
        } else {
            n_tie++;  // top-21-bit tie (or a draw of exactly 0 on an empty lane): exact compare needed
        }
        {  // dry window or tie: candidates straight from HBM with full keys
            const uint32_t cur_z = __builtin_amdgcn_readfirstlane(m.x), land_z = __builtin_amdgcn_readfirstlane(m.y);
            const uint32_t beg_z = seg[slot], len_z = seg[slot + 1] - beg_z;
            if (len_z == 0) {  // KeyError (psrs.py:44)
                status = OFFSIM_ST_KEYERROR;
                break;
            }
            const uint32_t rem = len_z - cur_z;
            if (rem == 0) {  // psrs.py:44-45
                status = OFFSIM_ST_EXHAUSTED;
                break;
            }
            // 16 candidates per direct read: enough that "none accepted" is a 1e-5 event, and 4x fewer random sectors
            // than a full wavefront of gathers (each 4..8-byte gather moves a 64-byte sector)
            constexpr uint32_t DIRECT = W > 16 ? (uint32_t)W + 16u : 16u;
            const uint32_t nv = rem < DIRECT ? rem : DIRECT;
            const bool valid = (uint32_t)lane < nv;
            const uint32_t p = beg_z + cur_z + (valid ? lane : 0);
            const uint32_t g = perm_row ? perm_row[p] : p;
            const uint64_t key = keys[g];
            const uint32_t Tt = (uint32_t)(key >> 43);
            const uint32_t k21 = kt >> 11;
            uint64_t macc = __ballot(valid && k21 < Tt), mamb = __ballot(valid && k21 == Tt);
            int f = -1;
            while (true) {
                const uint64_t mm = macc | mamb;
                if (mm == 0) break;
                const int ff = __ffsll((unsigned long long)mm) - 1;
                if ((mamb >> ff) & 1ull) {  // exact: k53 of draw c+ff against the full T
                    const uint64_t k53 = exact53(c + (uint32_t)ff + 1);
                    const uint32_t klo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, ff);
                    const uint32_t khi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), ff);
                    if (k53 > key_T(((uint64_t)khi << 32) | klo)) {
                        mamb &= ~(1ull << ff);
                        continue;
                    }
                }
                f = ff;
                break;
            }
            const uint32_t d = f < 0 ? nv : (uint32_t)f + 1u;
            // the candidates behind the consumed ones are already in registers: they become the new window
            const uint32_t keep_end = nv < d + (uint32_t)W ? nv : d + (uint32_t)W;  // lanes [d, keep_end) stay queued
            if ((uint32_t)lane >= d && (uint32_t)lane < keep_end) win[(uint32_t)slot * W + (cur_z + lane) % W] = (uint32_t)(key >> 32);
            const uint32_t new_land = cur_z + keep_end;
            meta[slot] = make_uint2(cur_z + d, land_z < new_land ? new_land : land_z);
            if (fillq[slot] < new_land) fillq[slot] = new_land;
            c += d;
            if (TRACE) pop_acc += d;
            if (f >= 0) {
                const uint32_t acc_dig = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), f);
                {
                    scan_u32x2 e;
                    e.x = cur_z + (uint32_t)f + 1u;
                    e.y = (acc_dig & ~1023u) | (uint32_t)slot;
                    *(lds_u32x2 *)(log_off + nph * 8u) = e;
                }
                if (TRACE) {
                    popq[nph] = pop_acc;
                    pop_acc = 0;
                }
                dn = (acc_dig >> 10) & 1u;
                nph++;
                slot = (int)(acc_dig & 1023u);
            }
        }
    ev_tail:
        // ---- common tail of every event: draws, phase boundary, episode end, refreshed prefetch ----
        while (gen < c + 64) gen_block();
        while (ticks_done < (nph >> TICK_LOG)) {
            refill_tick(ticks_done << TICK_LOG, (ticks_done << TICK_LOG) + TICK);
            ticks_done++;
        }
        if (nph == OFFSIM_PH) {
            flush();
            ticks_done = 0;
        }
        if (dn) {
            ep++;
            need_reset = true;
        } else {
            m = meta[slot];
            kt = ring[(c + lane) & (OFFSIM_RING - 1)];
        }
    }
    const bool mid_episode = (status == OFFSIM_ST_EXHAUSTED);  // the step loop only stops inside an episode
    flush();
    flush();  // drain the reward pipeline (R2, R3 of the last phases)
    flush();
    if (mid_episode) {  // psrs.py:265: the cut-short episode still logs its length
        if (lane == 0 && out.ep_len && (int64_t)n_len <= out.ep_cap) out.ep_len[(int64_t)r * (out.ep_cap + 1) + n_len] = (int32_t)len_acc;
        n_len++;
    }
    // ---- write the env state back ----
    __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
    for (int q = 0; q < ROUNDS; q++) {
        int s = q * 64 + lane;
        if (s < n_slots) cur_glb[s] = meta[s].x;
    }
    if (lane == 0) {
        ro.init_cursor[r] = ic;
        ro.cur_slot[r] = slot;
        if (c && philox) {
            ro.rng[4 * r + 1] = ph_c0 + c;  // (seed, draws consumed so far, 0, 0)
        } else if (c) {
            const U128 base = u128(ro.rng[4 * r + 0], ro.rng[4 * r + 1]);
            const U128 inc = u128(ro.rng[4 * r + 2], ro.rng[4 * r + 3]);
            U128 nb = pcg_apply(pcg_jump(inc, c), base);
            ro.rng[4 * r + 0] = nb.hi;
            ro.rng[4 * r + 1] = nb.lo;
        }
        out.sum_g[r] = sum_g;
        out.n_ep[r] = ep_acc;
        out.steps[r] = steps;
        out.cand[r] = c;
        out.n_len[r] = n_len;
        out.status[r] = status;
        if (out.dbg) {
            out.dbg[4 * (int64_t)r + 0] = n_dry;
            out.dbg[4 * (int64_t)r + 1] = n_tie;
            out.dbg[4 * (int64_t)r + 2] = n_flush;
            out.dbg[4 * (int64_t)r + 3] = gen / 64;
        }
    }
}

}  // namespace offsim
