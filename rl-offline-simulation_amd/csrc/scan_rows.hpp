// scan_rows.hpp -- the headline evalMC scan, row-packed: FOUR rollouts per wavefront, one 16-lane DPP row each.
//
// Why: a PSRS rollout is one dependent chain, so with R = 4096 rollouts the kernel time is
// (steps per rollout) x (time per chain step), and with one rollout per wavefront (scan_win.hpp) a step cost ~67
// instructions of which a compute unit could retire only ~1.3 per cycle for its 16 rollouts: the CU was at its
// instruction-issue ceiling.  Here one instruction stream serves four rollouts: every per-rollout "scalar" (state, draw
// counter, cursors) is a VGPR value that is uniform inside a 16-lane row, cross-lane steps are DPP row operations, and
// a CU runs its 16 rollouts on 4 wavefronts (one per SIMD) at single-wavefront latency.
//
// The step itself has ONE LDS round trip on the chain (scan_win.hpp: two -- cursor pair, then the window row):
//   * windows are HEAD-ALIGNED: entry j of a state's 8-entry row is the j-th candidate still queued (0 = none loaded).
//     Accepting entry k pushes entries k+1.. to the front with one ds_write (lane j stores to slot (j-k-1) mod 8, the
//     vacated slots get 0), so the next look at that state needs no cursor to find its candidates;
//   * lanes 0..7 of a row compare draw c+j (ring of pre-shifted 21-bit draws) with digest j in one v_sub_co; the winner
//     (first lane that is not a clear reject) and its payload (done, z_next, a "needs an exact look" field) are found by
//     a 4-step DPP min over key = (j+1) << 28 | clear << 11 | done << 10 | z_next (clear: digest - draw >= 2^15);
//   * anything that is not a clean accept -- tie on 21 bits, no candidate accepted, window dry, episode end, draws
//     running low -- is an event: every row that has one goes through handle() (exact, reads the stream directly) while
//     the rows without one commit their step; all rows therefore take exactly one accepted step per iteration and the
//     iteration / log index is wave-uniform (scalar).
// Candidates come from the per-rollout digest stream written by the sampler reset (offsim_shuffle_queues_keys): the
// refill reads 16 consecutive bytes of it per request instead of gathering 4-byte digests through a permutation
// (64-byte sector each).  Every 16 iterations one tick runs, per row with lane = step: land last tick's requests,
// request the top-ups of the states this tick left, the three-stage reward pipeline (row index -> reward -> in-order
// discounted sum, bit-exact Gs), the draw top-up and the initial-state ring.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "discount.hpp"
#include "offsim.h"
#include "pcg64_dev.hpp"
#include "scan_win.hpp"  // pack_key / key_T, lds_u32

namespace offsim {

#define ROWS_TICK 16u
#define ROWS_RING 128u
#define ROWS_W 8u
#define ROWS_EMPTY 0x400u  // window slot without a candidate: T21 = 0 and the done bit, so that it can win only as an event
// per-rollout LDS region (byte offsets); window rows are 32-byte aligned, the region a multiple of 512
#define RO_RING 0u       // 128 draws, (k21 << 11)
#define RO_JUNKROW 512u  // 32 B: the "window row" of lanes 8..15 (read and written, never meaningful)
#define RO_PAD 544u      // 0xffffffff: the "draw" of lanes 8..15, so that they never win
#define RO_INIT 576u     // 16 upcoming initial states (slot or -1)
#define RO_LOG 640u      // 16 x {cursor behind the accepted candidate, state left | done << 10}; reused as the prod scratch
#define RO_POP 768u      // 16 x candidates popped by the step (TRACE)
#define RO_WIN 1024u     // n_slots x 8 digests, then cons[n_slots], land[n_slots], claim[n_slots]

typedef __attribute__((address_space(3))) volatile uint32_t ldsv_u32;
typedef __attribute__((address_space(3))) volatile scan_u32x2 ldsv_u32x2;
typedef __attribute__((address_space(3))) volatile double ldsv_f64;
#define LV32(a) (*(ldsv_u32 *)(a))
#define LV64(a) (*(ldsv_u32x2 *)(a))

__host__ __device__ constexpr uint32_t rows_region_bytes(uint32_t n_slots) { return (RO_WIN + n_slots * 44u + 511u) & ~511u; }

template <int CTRL>
__device__ __forceinline__ uint32_t row_dpp(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, 0xf, 0xf, false);
}
// min / broadcast over the 16 lanes of a row (rarely used paths; the chain loop has its own fused sequence)
__device__ __forceinline__ uint32_t row_min16(uint32_t x) {
    uint32_t y = row_dpp<0xB1>(x);  // quad_perm [1,0,3,2]
    x = y < x ? y : x;
    y = row_dpp<0x4E>(x);  // quad_perm [2,3,0,1]
    x = y < x ? y : x;
    y = row_dpp<0x141>(x);  // row_half_mirror
    x = y < x ? y : x;
    y = row_dpp<0x128>(x);  // row_ror:8
    return y < x ? y : x;
}

// k53 of the draw that needs n_steps LCG steps from the stream's start (exact tie decisions only)
__device__ __noinline__ uint64_t rows_exact53(const uint64_t *__restrict__ rng4, uint64_t n_steps) {
    const U128 base = u128(rng4[0], rng4[1]), inc = u128(rng4[2], rng4[3]);
    return pcg_output(pcg_apply(pcg_jump(inc, n_steps), base)) >> 11;
}

template <bool TRACE>
__global__ void __launch_bounds__(256)
    k_eval_mc_rows(offsim_table t, offsim_rollouts ro, offsim_streams sm, const uint64_t *__restrict__ keys, double gamma,
                   const double *__restrict__ gamma_pow, int64_t n_gamma_pow64, int64_t max_episodes64, offsim_evalmc_out out,
                   uint32_t seg_bytes, uint32_t region_bytes) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t li = lane & 15u, rw = lane >> 4, li4 = li * 4u;
    const uint32_t n_slots = (uint32_t)t.n_slots;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_byte *)lds_raw;
    const uint32_t lds_pad = (0u - lds_base) & 511u;
    const uint32_t seg_a = lds_base + lds_pad;  // seg_off copy, shared by the block
    for (uint32_t i = threadIdx.x; i <= n_slots; i += blockDim.x) LV32(seg_a + i * 4u) = t.seg_off[i];
    const uint32_t rpb = blockDim.x >> 4;  // rollouts per block
    const uint32_t rid = wave * 4u + rw;
    const int64_t r = (int64_t)blockIdx.x * rpb + rid;
    const uint32_t rbase = seg_a + seg_bytes + rid * region_bytes;
    const uint32_t win_a = rbase + RO_WIN, cons_a = win_a + n_slots * 32u, land_a = cons_a + n_slots * 4u, claim_a = land_a + n_slots * 4u;
    uint32_t dead = r < (int64_t)ro.R ? 0u : 1u;  // 1: the row has stopped (or never ran)
    const int64_t rr = dead ? 0 : r;  // (rows past R read rollout 0's inputs and write nothing)

    // per-lane constants that neutralise lanes 8..15 of every row inside the chain loop without a predicate: their window
    // row is the junk row (state multiplier 0), their draw is the pad word, their shifted stores land in the junk row
    const bool lower = li < 8u;
    const uint32_t scale_l = lower ? 32u : 0u;
    const uint32_t win_rd_l = lower ? win_a + li4 : rbase + RO_JUNKROW + (li4 - 32u);
    const uint32_t win_w_l = lower ? win_a : rbase + RO_JUNKROW;
    const uint32_t rmask_l = lower ? (ROWS_RING * 4u - 4u) : 0u;
    const uint32_t rbase_l = lower ? rbase + RO_RING : rbase + RO_PAD;
    const uint32_t lifield = (li + 1u) << 28;

    if (li < 8u) LV32(rbase + RO_JUNKROW + li4) = ROWS_EMPTY;
    if (li == 8u) LV32(rbase + RO_PAD) = 0xffffffffu;
    __syncthreads();
    auto seg_at = [&](uint32_t s) -> uint32_t { return LV32(seg_a + s * 4u); };

    const uint32_t *dbase = sm.dig + rr * sm.dig_stride;
    const uint16_t *lbase = sm.loc ? sm.loc + rr * sm.loc_stride : nullptr;
    const uint32_t *init_row = ro.init_perm ? ro.init_perm + rr * ro.init_stride : nullptr;
    uint32_t *cur_glb = ro.cursor + rr * n_slots;
    const uint64_t *rng4 = ro.rng + 4 * rr;
    const uint32_t N0 = (uint32_t)t.N0;
    const uint64_t n_gamma_pow = (uint64_t)n_gamma_pow64;
    const uint32_t max_episodes = (uint32_t)(max_episodes64 > 0x7fffffffll ? 0x7fffffffll : max_episodes64);
    const bool r64 = t.r_dtype == OFFSIM_F64;

    // ---- priming: every state's window holds the next 8 candidates of its queue ----
    for (uint32_t s = li; s < n_slots; s += 16u) {
        const uint32_t c0 = cur_glb[s], beg = seg_at(s), len = seg_at(s + 1u) - beg;
        const uint32_t left = len - c0, want = left < ROWS_W ? left : ROWS_W;
#pragma unroll
        for (uint32_t e = 0; e < ROWS_W; e++) LV32(win_a + s * 32u + e * 4u) = e < want ? dbase[beg + c0 + e] : ROWS_EMPTY;
        LV32(cons_a + s * 4u) = c0;
        LV32(land_a + s * 4u) = c0 + want;
        LV32(claim_a + s * 4u) = 0u;
    }

    // ---- rejection stream: lane j of the row owns draws j, j + 16, ... (jump-ahead); ring of the top 21 bits, pre-shifted ----
    U128 lane_state;
    U128 plus16;
    {
        const U128 base = u128(rng4[0], rng4[1]), inc = u128(rng4[2], rng4[3]);
        plus16 = pcg_jump(inc, 16).plus;
        lane_state = pcg_apply(pcg_jump(inc, (uint64_t)li + 1), base);  // yields draw li
    }
    const U128 mult16 = u128(0xb6a4239f3b315f84ull, 0xf6ef6d3d288c03c1ull);  // PCG multiplier ** 16 mod 2**128
    uint32_t gen = 0, c = 0;  // draws generated / consumed since kernel start (every examined candidate = one draw)
    auto gen16 = [&]() {
        LV32(rbase + RO_RING + (((gen + li) & (ROWS_RING - 1u)) << 2)) = (uint32_t)(pcg_output(lane_state) >> 43) << 11;
        lane_state = add128(mul128(mult16, lane_state), plus16);
        gen += 16u;
    };
#pragma unroll 1
    for (int i = 0; i < 7; i++) gen16();  // 112 draws ahead

    // ---- initial states: ring of the next 16 entries of the shuffled init queue (psrs.py:22-23, 32-37) ----
    uint32_t ic = ro.init_cursor[rr], ib = ic, ep = 0;
    auto load_init = [&]() {
        ib = ic;
        const uint32_t k = ic + li;
        int v = -1;
        if (k < N0) v = t.init_slot[init_row ? init_row[k] : k];
        LV32(rbase + RO_INIT + li4) = (uint32_t)v;
    };
    load_init();

    // ---- per-row state ----
    uint32_t z = 0;  // current state slot
    int status = OFFSIM_ST_OK;
    uint32_t nlog_dead = 0;  // steps the row logged in the tick it stopped in
    uint32_t pop_acc = 0;    // candidates popped so far by the step in progress (TRACE)
    uint32_t n_dry = 0, n_tie = 0, n_tick = 0;
    // refill: one outstanding request per lane
    uint32_t rq_s = 0, rq_p = 0, rq_n = 0, rq_d0 = 0, rq_d1 = 0, rq_d2 = 0, rq_d3 = 0;
    // reward pipeline, three ticks deep (R1: row index + discount, R2: reward, R3: in-order sums)
    uint32_t loc1 = 0, rowb1 = 0, pop1 = 0, n1 = 0, n2 = 0, dm1 = 0, dm2 = 0, st1 = 0;
    uint64_t any1 = 0, any2 = 0;
    double gp1 = 0.0, gp2 = 0.0, rv2 = 0.0;
    uint32_t tt_chain = 0, steps = 0, ep_acc = 0, n_len = 0, len_acc = 0;
    double G = 0.0, sum_g = 0.0;

    // env.reset() (psrs.py:32-37, :249-252): the next initial state, or the rollout stops
    auto do_reset = [&](uint32_t logged) {
        if (ep >= max_episodes) {  // psrs.py:248
            dead = 1u;
            nlog_dead = logged;
            return;
        }
        if (ic >= N0) {  // psrs.py:33-35, 250-252
            status = OFFSIM_ST_NO_INIT;
            dead = 1u;
            nlog_dead = logged;
            z = 0xffffffffu;
            return;
        }
        if (ic - ib >= 16u) load_init();
        z = LV32(rbase + RO_INIT + ((ic - ib) << 2));
        ic++;
    };
    if (!dead) do_reset(0u);

    // chain-loop registers
    uint32_t vrow_rd = 0, vrow_w = 0, vcons = 0, dig = 0, kt = 0, cz = 0;
    auto issue_reads = [&]() {
        const uint32_t zz = dead ? 0u : z;
        vrow_rd = zz * scale_l + win_rd_l;
        vrow_w = zz * scale_l + win_w_l;
        vcons = cons_a + zz * 4u;
        const uint32_t ra = (((c << 2) + li4) & rmask_l) | rbase_l;
        dig = LV32(vrow_rd);
        kt = LV32(ra);
        cz = LV32(vcons);
    };

    // the step's bookkeeping for a clean accept of window entry k1-1 with digest payload `key`
    auto commit = [&](uint32_t key, uint32_t k1, uint32_t it) {
        c += k1;
        const uint32_t cz1 = cz + k1;
        LV32(vcons) = cz1;
        scan_u32x2 e;
        e.x = cz1;
        e.y = z | (key & 0x400u);
        LV64(rbase + RO_LOG + it * 8u) = e;
        if (TRACE) {
            LV32(rbase + RO_POP + it * 4u) = pop_acc + k1;
            pop_acc = 0;
        }
        const uint32_t k1x4 = k1 << 2;
        const uint32_t data = li4 < k1x4 ? ROWS_EMPTY : dig;
        LV32(((li4 - k1x4) & (lower ? 28u : 0u)) | vrow_w) = data;
        z = key & 0x3ffu;
    };

    // exact path: candidates of state z straight from the stream, starting at queue position cz, until one is accepted
    // (completes the step: log, cursor, window = the candidates behind it) or the queue ends (the rollout stops)
    auto direct = [&](uint32_t it) {
        for (;;) {
            const uint32_t beg = seg_at(z), len = seg_at(z + 1u) - beg;
            if (len == 0u) {  // KeyError (psrs.py:44)
                status = OFFSIM_ST_KEYERROR;
                dead = 1u;
                nlog_dead = it;
                return;
            }
            const uint32_t rem = len - cz;
            if (rem == 0u) {  // psrs.py:44-45
                status = OFFSIM_ST_EXHAUSTED;
                dead = 1u;
                nlog_dead = it;
                LV32(cons_a + z * 4u) = cz;
                return;
            }
            while (gen - c < 16u) gen16();
            const uint32_t nv = rem < 16u ? rem : 16u;
            const bool valid = li < nv;
            const uint32_t dg = valid ? dbase[beg + cz + li] : 0u;
            const uint32_t kk = LV32(rbase + RO_RING + (((c + li) & (ROWS_RING - 1u)) << 2));
            bool ok = valid && kk <= dg;
            if (ok && dg - kk < 2048u) {  // top-21-bit tie: k53 of draw c+li against the full T
                const uint32_t lc = lbase ? (uint32_t)lbase[beg + cz + li] : cz + li;
                ok = !(rows_exact53(rng4, (uint64_t)c + li + 1u) > key_T(keys[beg + lc]));
            }
            const uint32_t f = row_min16(ok ? li : 16u);
            if (f == 16u) {  // all of them rejected: consumed (one draw each)
                c += nv;
                cz += nv;
                if (TRACE) pop_acc += nv;
                continue;
            }
            const uint32_t acc = row_min16(li == f ? dg : 0xffffffffu);  // the accepted candidate's digest
            const uint32_t k1 = f + 1u;
            c += k1;
            const uint32_t cz1 = cz + k1;
            LV32(cons_a + z * 4u) = cz1;
            scan_u32x2 e;
            e.x = cz1;
            e.y = z | (acc & 0x400u);
            LV64(rbase + RO_LOG + it * 8u) = e;
            if (TRACE) {
                LV32(rbase + RO_POP + it * 4u) = pop_acc + k1;
                pop_acc = 0;
            }
            const uint32_t keep = nv - k1 < ROWS_W ? nv - k1 : ROWS_W;  // the candidates behind it become the window
            if (li < 8u) LV32(win_a + z * 32u + li4) = ROWS_EMPTY;
            if (li >= k1 && li < k1 + keep) LV32(win_a + z * 32u + ((li - k1) << 2)) = dg;
            LV32(land_a + z * 4u) = cz1 + keep;
            z = acc & 0x3ffu;
            if (acc & 0x400u) {
                ep++;
                do_reset(it + 1u);
            }
            return;
        }
    };

    // a row whose look was not a clean accept (key from the chain loop's reduction, valid in all 16 lanes)
    auto handle = [&](uint32_t key, uint32_t it) {
        const uint32_t k1 = key >> 28;
        const uint32_t v = LV32(land_a + z * 4u) - cz;  // candidates the window really holds (the others are ROWS_EMPTY)
        if (key != 0xffffffffu && (key & 0x800u) && k1 - 1u < v) {  // clean accept: the event is the episode end or low draws
            commit(key, k1, it);
            if (key & 0x400u) {
                ep++;
                do_reset(it + 1u);
            }
        } else {
            uint32_t nrej;  // clear rejects in front of the first candidate that needs the exact look
            if (key == 0xffffffffu || k1 - 1u >= v) {  // every candidate of the window rejected (an empty slot cannot be accepted)
                nrej = v < ROWS_W ? v : ROWS_W;
                n_dry++;
            } else {
                nrej = k1 - 1u;
                n_tie++;
            }
            c += nrej;
            cz += nrej;
            if (TRACE) pop_acc += nrej;
            direct(it);
        }
        while (!dead && gen - c < 48u) gen16();
    };

    // ---- once per 16 iterations; lane = step of the tick ----
    auto tick = [&]() {
        const uint32_t n = dead ? nlog_dead : ROWS_TICK;
        nlog_dead = 0;
        n_tick++;
        const scan_u32x2 le = LV64(rbase + RO_LOG + li * 8u);
        const bool mine = li < n;
        const uint32_t s_i = le.y & 0x3ffu, pos_i = le.x - 1u;
        const bool done_i = mine && (le.y & 0x400u);
        uint32_t pop_i = 0;
        if (TRACE) pop_i = LV32(rbase + RO_POP + li4);

        // C: land the digests requested one tick ago.  Entries are appended only at the window's current end: whatever a
        // direct read has covered meanwhile is skipped, whatever does not fit is requested again later.
        if (rq_n) {
            const uint32_t cs = LV32(cons_a + rq_s * 4u);
            uint32_t ld = LV32(land_a + rq_s * 4u);
            const uint32_t dd[4] = {rq_d0, rq_d1, rq_d2, rq_d3};
#pragma unroll
            for (uint32_t e = 0; e < 4u; e++) {
                if (e < rq_n && rq_p + e == ld && ld - cs < ROWS_W) {
                    LV32(win_a + rq_s * 32u + ((ld - cs) << 2)) = dd[e];
                    ld++;
                }
            }
            LV32(land_a + rq_s * 4u) = ld;
            rq_n = 0;
        }
        // A: one request per state left in this tick (the lane that logged the step tops the state up)
        if (mine) LV32(claim_a + s_i * 4u) = li;
        if (mine && LV32(claim_a + s_i * 4u) == li) {
            const uint32_t cs = LV32(cons_a + s_i * 4u), ld = LV32(land_a + s_i * 4u);
            const uint32_t beg = seg_at(s_i), len = seg_at(s_i + 1u) - beg;
            const uint32_t have = ld - cs, room = have < ROWS_W ? ROWS_W - have : 0u, left = len - ld;
            uint32_t want = room < left ? room : left;
            want = want < 4u ? want : 4u;
            if (want) {
                const uint32_t *src = dbase + beg + ld;
                rq_d0 = src[0];
                if (want > 1u) rq_d1 = src[1];
                if (want > 2u) rq_d2 = src[2];
                if (want > 3u) rq_d3 = src[3];
                rq_s = s_i;
                rq_p = ld;
                rq_n = want;
            }
        }

        // R3: in-order discounted-return accumulation (psrs.py:262-269) for the steps of two ticks ago: the products are
        // broadcast through LDS, every lane of the row runs the same sequential sum (bit-exact Gs)
        {
            const double prod = li < n2 ? gp2 * rv2 : 0.0;  // product first, then the running sum in step order (+0.0 changes nothing)
            *(ldsv_f64 *)(rbase + RO_LOG + li * 8u) = prod;
            double p[16];
#pragma unroll
            for (int i = 0; i < 16; i++) p[i] = *(ldsv_f64 *)(rbase + RO_LOG + (uint32_t)i * 8u);
            int32_t base_len = (int32_t)len_acc;  // length of the open episode minus the steps of this tick already counted
            if (any2 == 0ull) {
#pragma unroll
                for (int i = 0; i < 16; i++) G = G + p[i];
            } else {
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    G = G + p[i];
                    if ((any2 >> i) & 1ull) {      // (wave-uniform) some row ends an episode at step i
                        if ((dm2 >> i) & 1u) {     // this row does (psrs.py:265-269)
                            if (li == 0u) {
                                if (out.ep_g && (int64_t)ep_acc < out.ep_cap) out.ep_g[r * out.ep_cap + ep_acc] = G;
                                if (out.ep_len && (int64_t)n_len <= out.ep_cap) out.ep_len[r * (out.ep_cap + 1) + n_len] = base_len + i + 1;
                            }
                            sum_g += G;
                            ep_acc++;
                            n_len++;
                            G = 0.0;
                            base_len = -(i + 1);
                        }
                    }
                }
            }
            len_acc = (uint32_t)(base_len + (int32_t)n2);
        }
        // R2: rewards of the steps of one tick ago
        {
            double rv = 0.0;
            if (li < n1) {
                const uint32_t g = rowb1 + loc1;
                rv = r64 ? ((const double *)t.r)[g] : (double)((const float *)t.r)[g];
                if (TRACE) {
                    const uint32_t st = st1 + li;
                    if (out.trace_row && (int64_t)st < out.trace_cap) out.trace_row[r * out.trace_cap + st] = t.orig_idx[g];
                    if (out.trace_pop && (int64_t)st < out.trace_cap) out.trace_pop[r * out.trace_cap + st] = pop1;
                }
            }
            rv2 = rv;
            gp2 = gp1;
            dm2 = dm1;
            any2 = any1;
            n2 = n1;
        }
        // R1: row index (through the stream of local indices) and discount factor of this tick's steps
        {
            const uint64_t bal = __ballot(done_i);
            const uint32_t dmrow = (uint32_t)(bal >> (rw * 16u)) & 0xffffu;  // episode ends of this row's tick
            uint32_t lc = 0, rb = 0;
            double gp = 0.0;
            if (mine) {
                rb = seg_at(s_i);
                lc = lbase ? (uint32_t)lbase[rb + pos_i] : pos_i;
                const uint32_t below = dmrow & ((1u << li) - 1u);  // episode ends earlier in this tick
                const uint32_t t_i = below ? li - 1u - (31u - (uint32_t)__clz((int)below)) : tt_chain + li;
                gp = discount_at(gamma_pow, n_gamma_pow, gamma, (uint64_t)t_i);
            }
            loc1 = lc;
            rowb1 = rb;
            gp1 = gp;
            pop1 = pop_i;
            dm1 = dmrow;
            any1 = (bal | (bal >> 16) | (bal >> 32) | (bal >> 48)) & 0xffffull;
            n1 = n;
            st1 = steps;
            tt_chain = dmrow ? n - 1u - (31u - (uint32_t)__clz((int)dmrow)) : tt_chain + n;
            steps += n;
        }
        if (!dead) {
            if (ic - ib >= 8u) load_init();
            while (gen - c < 112u) gen16();
        }
    };

    // ---- the chain ----
    // One iteration = one accepted step of every live row.  The inner loop holds nothing but the clean-accept path and is
    // left through one wave-uniform branch as soon as ANY row has an event; that iteration is then redone row by row
    // (rows without an event commit, the others go through handle()), and the fast loop is entered again.
    const uint32_t c7ff = 0x7ffu;
    for (;;) {
        uint32_t it = 0;
        issue_reads();
        while (it < ROWS_TICK) {
            uint32_t key = 0, k1 = 0;
            bool ev = false;
            for (;;) {
                uint32_t tmp;
                // key = (first lane of the row that is not a clear reject + 1) << 28 | clear << 11 | done << 10 | z_next, where
                // clear = the digest exceeds the draw by at least 16 units of T21 (anything closer gets the exact look);
                // all ones if every lane is a clear reject.  The 4-step row minimum leaves it in all 16 lanes.
                asm volatile(
                    "v_sub_co_u32 %0, vcc, %2, %3\n\t"
                    "v_and_or_b32 %1, %2, %4, %5\n\t"
                    "v_lshrrev_b32 %0, 15, %0\n\t"
                    "v_min_u32 %0, 1, %0\n\t"
                    "v_lshl_or_b32 %0, %0, 11, %1\n\t"
                    "v_cndmask_b32_e64 %0, %0, -1, vcc\n\t"
                    "s_nop 1\n\t"
                    "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                    "s_nop 1\n\t"
                    "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                    "s_nop 1\n\t"
                    "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                    "s_nop 1\n\t"
                    "v_min_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf"
                    : "=&v"(key), "=&v"(tmp)
                    : "v"(dig), "v"(kt), "s"(c7ff), "v"(lifield)
                    : "vcc");
                k1 = key >> 28;
                const uint32_t low = (gen - 24u) - (c + k1);  // negative: fewer than 24 draws would be left
                ev = (((low & 0x80000000u) | (key & 0xc00u) | dead) != 0x800u);  // not {clear, episode goes on, draws left, row live}
                if (__builtin_expect(__ballot(ev) != 0ull, 0)) break;
                commit(key, k1, it);
                it++;
                if (it == ROWS_TICK) break;
                issue_reads();
            }
            if (it == ROWS_TICK) break;
            if (!dead) {
                if (!ev) commit(key, k1, it);
                else handle(key, it);
            }
            it++;
            if (it < ROWS_TICK) issue_reads();
        }
        tick();
        if (__ballot(!dead) == 0ull) break;
    }
    tick();  // drain the reward pipeline (R2, R3 of the last ticks)
    tick();
    if (status == OFFSIM_ST_EXHAUSTED) {  // psrs.py:265: the cut-short episode still logs its length
        if (li == 0u && r < (int64_t)ro.R && out.ep_len && (int64_t)n_len <= out.ep_cap) out.ep_len[r * (out.ep_cap + 1) + n_len] = (int32_t)len_acc;
        n_len++;
    }
    // ---- write the env state back ----
    if (r < (int64_t)ro.R) {
        for (uint32_t s = li; s < n_slots; s += 16u) cur_glb[s] = LV32(cons_a + s * 4u);
        if (li == 0u) {
            ro.init_cursor[r] = ic;
            ro.cur_slot[r] = (int32_t)z;
            if (c) {
                const U128 base = u128(rng4[0], rng4[1]), inc = u128(rng4[2], rng4[3]);
                const U128 nb = pcg_apply(pcg_jump(inc, c), base);
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
                out.dbg[4 * r + 0] = n_dry;
                out.dbg[4 * r + 1] = n_tie;
                out.dbg[4 * r + 2] = n_tick;
                out.dbg[4 * r + 3] = gen / 16u;
            }
        }
    }
}

}  // namespace offsim
