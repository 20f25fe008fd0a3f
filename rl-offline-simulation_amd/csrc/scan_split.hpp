// scan_split.hpp -- the evalMC scan of scan_win.hpp with every rollout served by TWO wavefronts:
//
//   chain wave   the dependent chain only: accept from the LDS windows, step log, draws, episode resets, direct reads of
//                dry windows / ties.  It owns the cursors (meta[].x), the step log, the draw ring and a mailbox.
//   helper wave  everything that is not on the chain: the demand-driven refill pass (it is the ONLY writer of the windows,
//                of meta[].y = landed and of fillq) and the reward pipeline with the in-order return accumulation (it
//                owns sum_g, episode outputs, traces).
//
// Why: a single in-order wavefront issues one instruction per ~8 cycles, and with four rollouts per SIMD the vector and
// scalar units sat at ~50 %.  Splitting a rollout takes ~25 % of the instructions off the chain wave and doubles the
// wavefronts a SIMD can interleave (8 per SIMD: the kernel must stay within 64 VGPRs).
//
// Protocol (all in the rollout's LDS region; the LDS unit executes a wavefront's DS instructions in issue order, so data
// written before a counter is visible before the counter):
//   NLOG   steps logged so far (chain -> helper).  Log entries live in a ring of 128; a run of the chain never crosses a
//          multiple of TICK, so it never wraps inside the fast loop.
//   HDONE  steps whose log entries the helper no longer needs (helper -> chain), advanced per flushed phase of 64.
//   FIN    status + 1 once the chain has stopped (chain -> helper).
//   mailbox  after a direct read the chain hands the digests behind the consumed candidates to the helper, which
//          installs them in the window (the chain itself never writes a window: no write can race the refill).
// A window entry is read by the chain only for positions in [cur, landed); the helper writes an entry before it
// publishes landed, and re-uses a slot for position p + W only after it has seen cur > p.  `landed` may lag behind the
// truth (fewer candidates visible, at worst an unnecessary direct read), never run ahead of it.
#pragma once
#include "scan_win.hpp"

namespace offsim {

#define OFFSIM_LOGRING 128u
enum { SC_NLOG = 0, SC_HDONE = 1, SC_FIN = 2, SC_MBSEQ = 3, SC_MBDONE = 4, SC_MBSLOT = 8, SC_MBFIRST = 9, SC_MBCNT = 10, SC_MBDIG = 16 };

typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32s;

template <int W>
constexpr uint32_t split_ctrl_bytes() { return 4u * (16u + (uint32_t)(W > 16 ? W : 16)); }
template <int W, bool TRACE>
constexpr uint32_t split_log_bytes() { return OFFSIM_LOGRING * 8u + (TRACE ? OFFSIM_LOGRING * 4u : 0u) + split_ctrl_bytes<W>(); }

template <int W, int ROUNDS, bool TRACE>
__global__ void __launch_bounds__(512, 8)
    k_eval_mc_split(offsim_table t, offsim_rollouts ro, const uint64_t *__restrict__ keys, double gamma,
                    const double *__restrict__ gamma_pow, int64_t n_gamma_pow64, int64_t max_episodes64, offsim_evalmc_out out) {
    constexpr uint32_t TICK = W > 16 ? 16u : 32u;  // accepted steps between refill passes
    constexpr int D = W > 8 ? 8 : 4;               // entries one request may bring
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / 64), lane = threadIdx.x & 63;
    const int ch = wv & 3;         // rollout of this block
    const bool helper = wv >= 4;   // waves 0..3 run the chains, waves 4..7 their helpers
    const int n_slots = t.n_slots;
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_byte *)lds_raw;
    const uint32_t lds_pad = (0u - lds_base) & 511u;
    uint32_t *seg = (uint32_t *)(lds_raw + lds_pad);
    const uint32_t seg_bytes = ((uint32_t)(n_slots + 1) * 4 + 511u) & ~511u;
    const uint32_t win_bytes = (uint32_t)n_slots * W * 4;
    const uint32_t wave_bytes = (OFFSIM_RING * 4 + win_bytes + (uint32_t)n_slots * 16 + split_log_bytes<W, TRACE>() + 511u) & ~511u;
    const uint32_t ring_off = lds_base + lds_pad + seg_bytes + (uint32_t)ch * wave_bytes;  // LDS byte address of this rollout's region
    uint32_t *ring = (uint32_t *)(lds_raw + lds_pad + seg_bytes + (size_t)ch * wave_bytes);
    volatile uint32_t *win = ring + OFFSIM_RING;  // written by the helper only (and by the priming)
    const uint32_t win_off = ring_off + OFFSIM_RING * 4;
    volatile uint2 *meta = (volatile uint2 *)((unsigned char *)win + win_bytes);  // .x = cur (chain), .y = landed (helper)
    const uint32_t meta_off = win_off + win_bytes;
    volatile uint32_t *fillq = (volatile uint32_t *)(meta + n_slots);  // helper: queue position up to which entries have been requested
    volatile uint32_t *claim = fillq + n_slots;                        // helper: which lane requests for a state this tick
    const uint32_t log_off = meta_off + (uint32_t)n_slots * 16u;       // step log ring: {cursor behind the accepted candidate, digest | state left}
    volatile uint32_t *popq = (volatile uint32_t *)(claim + n_slots) + OFFSIM_LOGRING * 2;  // TRACE only
    lds_vu32s *ctrl = (lds_vu32s *)(log_off + OFFSIM_LOGRING * 8u + (TRACE ? OFFSIM_LOGRING * 4u : 0u));
    for (int i = threadIdx.x; i <= n_slots; i += blockDim.x) seg[i] = t.seg_off[i];
    __syncthreads();
    const int r = blockIdx.x * 4 + ch;
    const bool live = r < ro.R;

    const uint32_t *perm_row = (live && ro.perm) ? ro.perm + (int64_t)r * ro.perm_stride : nullptr;
    const uint32_t *keys32 = (const uint32_t *)keys;

    if (live && !helper) {
        // ---- priming: every state's window is filled once, synchronously (lane = state) ----
        const uint32_t *cur_glb = ro.cursor + (int64_t)r * n_slots;
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
                meta[s].x = c0;
                meta[s].y = c0 + want;
                fillq[s] = c0 + want;
            }
        }
        if (lane < 16) ctrl[lane] = 0;
    }
    __syncthreads();
    if (!live) return;

    const uint32_t n_gamma_pow = (uint32_t)(n_gamma_pow64 > 0x7fffffffll ? 0x7fffffffll : n_gamma_pow64);
    const uint32_t max_episodes = (uint32_t)(max_episodes64 > 0xffffffffll ? 0xffffffffll : max_episodes64);

    if (!helper) {
        // =====================================================================================================
        // chain wave
        // =====================================================================================================
        const uint32_t *init_row = ro.init_perm ? ro.init_perm + (int64_t)r * ro.init_stride : nullptr;
        const uint32_t N0 = (uint32_t)t.N0;
        U128 lane_state;
        Jump j64;
        {
            const U128 base = u128(ro.rng[4 * r + 0], ro.rng[4 * r + 1]);
            const U128 inc = u128(ro.rng[4 * r + 2], ro.rng[4 * r + 3]);
            j64 = pcg_jump(inc, 64);
            lane_state = pcg_apply(pcg_jump(inc, (uint64_t)lane + 1), base);  // yields draw `lane`
        }
        uint32_t gen = 0, c = 0;  // draws generated / consumed since kernel start (every examined candidate = one draw)
        auto gen_block = [&]() {
            ring[(gen + lane) & (OFFSIM_RING - 1)] = (uint32_t)(pcg_output(lane_state) >> 43) << 11;  // top 21 bits, aligned with the digest's T21
            lane_state = pcg_apply(j64, lane_state);
            gen += 64;
        };
        gen_block();
        gen_block();
        auto exact53 = [&](uint32_t n_steps) -> uint64_t {  // k53 of the draw that needs n_steps LCG steps from the start
            const U128 base = u128(ro.rng[4 * r + 0], ro.rng[4 * r + 1]);
            const U128 inc = u128(ro.rng[4 * r + 2], ro.rng[4 * r + 3]);
            return pcg_output(pcg_apply(pcg_jump(inc, n_steps), base)) >> 11;
        };
        uint32_t ic = ro.init_cursor[r], ib = ic;
        int init_reg = -1;
        auto load_init = [&]() {  // lane i holds the slot of init index ib + i
            ib = ic;
            uint32_t k = ic + lane;
            int v = -1;
            if (k < N0) v = t.init_slot[init_row ? init_row[k] : k];
            init_reg = v;
        };
        load_init();

        int slot = ro.cur_slot[r];
        uint32_t ep = 0, n_dry = 0, n_tie = 0;
        uint32_t nlog = 0, hdone = 0, mb_seq = 0, pop_acc = 0;
        int status = OFFSIM_ST_OK;
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
                m.x = meta[slot].x;
                m.y = meta[slot].y;
                kt = ring[(c + lane) & (OFFSIM_RING - 1)];
            }
            // fast loop (see scan_win.hpp): one iteration = one accepted step served from the LDS window
            uint64_t many = 0;
            bool accepted;
            uint32_t vslot, vlog;
            asm("v_mov_b32 %0, %1" : "=v"(vslot) : "s"(slot));
            asm("v_mov_b32 %0, %1" : "=v"(vlog) : "s"(log_off + (nlog & (OFFSIM_LOGRING - 1u)) * 8u));
            uint32_t vrow = win_off + vslot * (uint32_t)(W * 4), vmeta = meta_off + vslot * 8u;
            uint32_t v1023, vringm;
            asm("v_mov_b32 %0, 0x3ff" : "=v"(v1023));
            asm("v_mov_b32 %0, %1" : "=v"(vringm) : "s"(OFFSIM_RING * 4u - 4u));
            volatile uint32_t *plog = popq + (nlog & (OFFSIM_LOGRING - 1u));
            const uint32_t nlog_in = nlog;
            int tick_b = (int)((TICK - 1u) - (nlog & (TICK - 1u)));  // goes negative when a multiple of TICK steps has been logged
            const int b_in = tick_b;
            const uint32_t gen_m64 = gen - 64u;
            for (;;) {
                // entries that have landed: landed may lag behind cur after a direct read (the helper catches up), hence signed
                const int v_avail = (int)(m.y - m.x);
                uint32_t dig = *(lds_u32 *)((((m.x + (uint32_t)lane) << 2) & (uint32_t)(W * 4 - 4)) | vrow);
                dig = lane < (v_avail < W ? v_avail : W) ? dig : 0u;
                many = __ballot(kt <= dig);
                int f;
                asm("s_ff1_i32_b64 %0, %1" : "=s"(f) : "s"(many));
                const uint32_t acc_dig = (uint32_t)__builtin_amdgcn_readlane((int)dig, f);
                const uint32_t acc_kt = (uint32_t)__builtin_amdgcn_readlane((int)kt, f);
                accepted = acc_kt < (acc_dig & 0xfffff800u);
                if (__builtin_expect(!accepted, 0)) break;
                const uint32_t f1 = (uint32_t)f + 1u;
                const uint32_t cur1 = m.x + f1;
                *(lds_u32 *)vmeta = cur1;  // meta[vslot].x
                {
                    scan_u32x2 e;
                    e.x = cur1;
                    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(e.y) : "v"(v1023), "v"(vslot), "s"(acc_dig));
                    *(lds_u32x2 *)vlog = e;
                    vlog += 8u;
                }
                c += f1;
                if (TRACE) {
                    *plog++ = pop_acc + f1;
                    pop_acc = 0;
                }
                tick_b -= 1;
                asm("v_bfe_u32 %0, %1, 0, 10" : "=v"(vslot) : "s"(acc_dig));
                vrow = win_off + vslot * (uint32_t)(W * 4);
                vmeta = meta_off + vslot * 8u;
                {
                    uint32_t ka = (c + (uint32_t)lane) << 2;
                    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(ka) : "v"(ka), "v"(vringm), "s"(ring_off));
                    kt = *(lds_u32 *)ka;
                }
                {
                    const scan_u32x2 mv = *(volatile lds_u32x2 *)vmeta;
                    m.x = mv.x;
                    m.y = mv.y;
                }
                dn = (acc_dig >> 10) & 1u;
                if (__builtin_expect(dn, 0)) break;
                if (__builtin_expect((int)((uint32_t)tick_b | (gen_m64 - c)) < 0, 0)) break;
            }
            slot = (int)__builtin_amdgcn_readfirstlane(vslot);
            nlog = nlog_in + (uint32_t)(b_in - tick_b);
            if (accepted) goto ev_tail;
            dn = false;
            if (many == 0ull) {
                const int av = (int)__builtin_amdgcn_readfirstlane(m.y - m.x);
                if (av > 0) {  // every window candidate rejected: consume them, look again
                    const uint32_t d = (uint32_t)(av < W ? av : W);
                    meta[slot].x = __builtin_amdgcn_readfirstlane(m.x) + d;
                    c += d;
                    if (TRACE) pop_acc += d;
                    goto ev_tail;
                }
                n_dry++;
            } else {
                n_tie++;
            }
            {  // dry window or tie: candidates straight from HBM with full keys
                const uint32_t cur_z = __builtin_amdgcn_readfirstlane(m.x);
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
                meta[slot].x = cur_z + d;
                // the candidates behind the consumed ones are already in registers: hand them to the helper, which installs
                // them as the new window (if the mailbox is still busy they are simply dropped)
                const uint32_t keep_end = nv < d + (uint32_t)W ? nv : d + (uint32_t)W;  // lanes [d, keep_end) stay queued
                if (keep_end > d && (uint32_t)__builtin_amdgcn_readfirstlane(ctrl[SC_MBDONE]) == mb_seq) {
                    if ((uint32_t)lane >= d && (uint32_t)lane < keep_end) ctrl[SC_MBDIG + ((uint32_t)lane - d)] = (uint32_t)(key >> 32);
                    ctrl[SC_MBSLOT] = (uint32_t)slot;
                    ctrl[SC_MBFIRST] = cur_z + d;
                    ctrl[SC_MBCNT] = keep_end - d;
                    mb_seq++;
                    ctrl[SC_MBSEQ] = mb_seq;
                }
                c += d;
                if (TRACE) pop_acc += d;
                if (f >= 0) {
                    const uint32_t acc_dig = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), f);
                    {
                        scan_u32x2 e;
                        e.x = cur_z + (uint32_t)f + 1u;
                        e.y = (acc_dig & ~1023u) | (uint32_t)slot;
                        *(lds_u32x2 *)(log_off + (nlog & (OFFSIM_LOGRING - 1u)) * 8u) = e;
                    }
                    if (TRACE) {
                        popq[nlog & (OFFSIM_LOGRING - 1u)] = pop_acc;
                        pop_acc = 0;
                    }
                    dn = (acc_dig >> 10) & 1u;
                    nlog++;
                    slot = (int)(acc_dig & 1023u);
                }
            }
        ev_tail:
            while (gen < c + 64) gen_block();
            __asm__ volatile("" ::: "memory");  // the log stores stay in front of the counter (the LDS keeps their order)
            ctrl[SC_NLOG] = nlog;
            while (nlog + TICK - hdone > OFFSIM_LOGRING) {  // the next run must not overwrite entries the helper still needs
                hdone = (uint32_t)__builtin_amdgcn_readfirstlane(ctrl[SC_HDONE]);
                if (nlog + TICK - hdone > OFFSIM_LOGRING) __builtin_amdgcn_s_sleep(2);
            }
            if (dn) {
                ep++;
                need_reset = true;
            } else {
                m.x = meta[slot].x;
                m.y = meta[slot].y;
                kt = ring[(c + lane) & (OFFSIM_RING - 1)];
            }
        }
        __asm__ volatile("" ::: "memory");
        ctrl[SC_NLOG] = nlog;
        ctrl[SC_FIN] = (uint32_t)status + 1u;
        // ---- write the env state back ----
        uint32_t *cur_out = ro.cursor + (int64_t)r * n_slots;
#pragma unroll
        for (int q = 0; q < ROUNDS; q++) {
            int s = q * 64 + lane;
            if (s < n_slots) cur_out[s] = meta[s].x;
        }
        if (lane == 0) {
            ro.init_cursor[r] = ic;
            ro.cur_slot[r] = slot;
            if (c) {
                const U128 base = u128(ro.rng[4 * r + 0], ro.rng[4 * r + 1]);
                const U128 inc = u128(ro.rng[4 * r + 2], ro.rng[4 * r + 3]);
                U128 nb = pcg_apply(pcg_jump(inc, c), base);
                ro.rng[4 * r + 0] = nb.hi;
                ro.rng[4 * r + 1] = nb.lo;
            }
            out.cand[r] = c;
            out.status[r] = status;
            if (out.dbg) {
                out.dbg[4 * (int64_t)r + 0] = n_dry;
                out.dbg[4 * (int64_t)r + 1] = n_tie;
                out.dbg[4 * (int64_t)r + 3] = gen / 64;
            }
        }
        return;
    }

    // =========================================================================================================
    // helper wave
    // =========================================================================================================
    uint32_t rq_state = 0, rq_slot = 0, rq_pos = 0, rq_cnt = 0;
    uint32_t idxA[D], digB[D];
#pragma unroll
    for (int e = 0; e < D; e++) idxA[e] = digB[e] = 0;
    uint32_t ep_acc = 0, n_len = 0, steps = 0, len_acc = 0, n_flush = 0;
    double sum_g = 0.0, G = 0.0;
    uint32_t pos_log = 0, pop_log = 0, dig_log = 0, slot_log = 0;
    uint32_t nph = 0;
    const bool r64 = t.r_dtype == OFFSIM_F64;
    uint32_t g1 = 0;
    double gp1 = 0.0, gp2 = 0.0, rv2 = 0.0;
    uint32_t pop1 = 0;
    uint64_t dm1 = 0, dm2 = 0;
    uint32_t n1 = 0, n2 = 0, st1 = 0;
    uint32_t tt_chain = 0;
    auto load_log = [&](uint32_t base) {  // lane i <- step base + i (ring index)
        const uint32_t idx = (base + (uint32_t)lane) & (OFFSIM_LOGRING - 1u);
        const scan_u32x2 e = *(volatile lds_u32x2 *)(log_off + idx * 8u);
        pos_log = e.x - 1u;
        dig_log = e.y;
        slot_log = e.y & 1023u;
        if (TRACE) pop_log = popq[idx];
    };
    auto refill_tick = [&](uint32_t base, uint32_t lo, uint32_t hi) {
        load_log(base);
        if (rq_state == 2) {  // C: land
            const uint32_t mx = meta[rq_slot].x, my = meta[rq_slot].y;
            const uint32_t wbase = rq_slot * W;
#pragma unroll
            for (int e = 0; e < D; e++) {
                const uint32_t pos = rq_pos + e;
                if ((uint32_t)e < rq_cnt && pos >= mx) win[wbase + pos % W] = digB[e];
            }
            const uint32_t end = rq_pos + rq_cnt;
            if (rq_pos <= my && end > my) meta[rq_slot].y = end;
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
            if ((int)(meta[s_].y - cur_s) < 0) meta[s_].y = cur_s;  // the chain ran past the window (direct read): nothing is valid
            uint32_t f = fillq[s_];
            f = (int)(f - cur_s) < 0 ? cur_s : f;
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
    auto flush = [&](uint32_t base) {
        load_log(base);
        {   // R3: in-order discounted-return accumulation (psrs.py:262-269)
            const double prod = gp2 * rv2;
            uint32_t i = 0;
            uint64_t dm = dm2;
            while (i < n2) {
                const uint32_t e = dm ? (uint32_t)__ffsll((unsigned long long)dm) - 1u : n2;
                const uint32_t run_end = e < n2 ? e + 1u : n2;
                len_acc += run_end - i;
                for (; i + 4u <= run_end; i += 4u) {
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
                    dm &= dm - 1ull;
                }
            }
        }
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
            const uint64_t done_mask = __ballot((uint32_t)lane < nph && ((dig_log >> 10) & 1u));
            uint32_t g = 0;
            double gp = 0.0;
            if ((uint32_t)lane < nph) {
                const uint32_t p = seg[slot_log] + pos_log;
                g = perm_row ? perm_row[p] : p;
                const uint64_t below = done_mask & ((1ull << lane) - 1ull);
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

    uint32_t processed = 0, mb_done = 0;
    for (;;) {
        const uint32_t fin = (uint32_t)__builtin_amdgcn_readfirstlane(ctrl[SC_FIN]);  // FIN first: then NLOG is final
        const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane(ctrl[SC_NLOG]);
        {   // mailbox: digests behind the candidates a direct read consumed become window entries
            const uint32_t seq = (uint32_t)__builtin_amdgcn_readfirstlane(ctrl[SC_MBSEQ]);
            if (seq != mb_done) {
                const uint32_t ms = (uint32_t)__builtin_amdgcn_readfirstlane(ctrl[SC_MBSLOT]);
                const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane(ctrl[SC_MBFIRST]);
                const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane(ctrl[SC_MBCNT]);
                const uint32_t mx = meta[ms].x;
                if ((uint32_t)lane < cnt) {
                    const uint32_t pos = first + (uint32_t)lane;
                    const uint32_t dg = ctrl[SC_MBDIG + lane];
                    if ((int)(pos - mx) >= 0) win[ms * W + pos % W] = dg;
                }
                const uint32_t new_land = first + cnt;
                if ((int)(new_land - meta[ms].y) > 0) meta[ms].y = new_land;
                if ((int)(new_land - fillq[ms]) > 0) fillq[ms] = new_land;
                mb_done = seq;
                ctrl[SC_MBDONE] = seq;
            }
        }
        if (processed + TICK <= n) {
            refill_tick(processed & 64u, processed & 63u, (processed & 63u) + TICK);
            processed += TICK;
            if ((processed & 63u) == 0u) {
                nph = 64u;
                flush(processed - 64u);
                ctrl[SC_HDONE] = processed;
            }
            continue;
        }
        if (fin) {
            const uint32_t phase_start = processed & ~63u;
            nph = n - phase_start;  // entries of the unfinished phase (0..63)
            flush(phase_start);
            flush(0);
            flush(0);  // drain the reward pipeline (R2, R3 of the last phases)
            const bool mid_episode = (int)(fin - 1u) == OFFSIM_ST_EXHAUSTED;  // the step loop only stops inside an episode
            if (mid_episode) {  // psrs.py:265: the cut-short episode still logs its length
                if (lane == 0 && out.ep_len && (int64_t)n_len <= out.ep_cap) out.ep_len[(int64_t)r * (out.ep_cap + 1) + n_len] = (int32_t)len_acc;
                n_len++;
            }
            if (lane == 0) {
                out.sum_g[r] = sum_g;
                out.n_ep[r] = ep_acc;
                out.steps[r] = steps;
                out.n_len[r] = n_len;
                if (out.dbg) out.dbg[4 * (int64_t)r + 2] = n_flush;
            }
            break;
        }
        __builtin_amdgcn_s_sleep(100);  // ~2.7 us: a tick is ~12 us away, and every poll costs issue slots the chain waves need
    }
}

}  // namespace offsim
