// shuffle_wave.hpp -- exact wave-parallel Fisher-Yates for PSRS.reset_sampler (psrs.py:22-23, 29-30).
//
// NumPy's Generator.shuffle of a list is   for i = n-1 .. 1:  j = random_interval(i);  swap(x[i], x[j])
// with random_interval = masked rejection on buffered 32-bit halves of PCG64 outputs.  The chain looks strictly
// sequential, and the first kernel ran it that way (one lane per chain, random 4-byte swaps in HBM: two 64-byte
// sectors moved per swap, ~2e10 swaps/s at the random-sector limit of the memory system).  But it factors:
//   1. the sequence j(n-1), j(n-2), ..., j(1) is a function of the random stream only.  64 consecutive draws are
//      classified at once: draw l is accepted iff v_l <= i - (#accepts before l).  Start from the optimistic set
//      {v_l <= i}, take prefix counts with ballot/mbcnt, and strike the first lane whose test fails (its prefix is
//      exact, so it is a true reject) until none fails -- usually zero rounds, a lane can only fail when v_l lies
//      within 64 of i.  The accepted values, compacted in order, ARE the j sequence;
//   2. 64 consecutive swaps (i, j(i)), i = t .. t-63, commute unless two of them touch a common position: either a
//      partner lies in the group's own range of i (j(i) == i' for a later i': plain arithmetic, the later swap is
//      lane t - j(i)), or two partners are equal (found by writing lane-id tags to the partner positions and reading
//      them back).  A group with neither is applied in one LDS round trip; otherwise it is applied in pieces, in
//      lane order, so the result is bit-identical to the sequential chain.
// So one workgroup owns one chain, keeps the queue segment in LDS (16-bit entries, <= 65536 rows: the whole chain
// runs at LDS latency and the only HBM traffic is the final coalesced write of the permutation), and splits the
// work over four wavefronts connected by LDS rings:
//     G  (two of them, alternate blocks) raw 32-bit draws: PCG64 jump-ahead, one 64-bit output per lane  -> draw ring
//     C  classifies 2 x 64 draws per iteration and appends the accepted values                             -> j ring
//     A  applies 64 steps per iteration to the segment
// Segments that do not fit (the init queue of a big log, states with > 65536 rows) use the same roles with the
// segment left in global memory (32-bit entries, in place).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pcg64_dev.hpp"

namespace offsim {

#define SHUF_RG 2048u  // raw draws in the ring (power of two, multiple of 128)
// partners in the j ring (power of two, >= 3 * 128), a template parameter (as a kernel argument it cost 3.5 %): 4096 for the class of the longest LDS-resident
// chains (deep enough that C rarely waits behind a slow group of A; there one chain fills a CU anyway), 1024 elsewhere (the
// short chains of a skewed table share a CU, and their occupancy is what the fixed part of the LDS costs)
// cuts of the keyed chains of the longest size class (see k_shuffle_wave).  Measured per pass of the headline job
// (tools/time_reset.py ... keyed): uncut 0.645 s; one cut at 4096 / 8192 / 16384 / 32768: 0.575 / 0.548 / 0.540 / 0.568 s;
// two cuts at 16384 + 4096: 0.528 s, 16384 + 2048: 0.529 s; three cuts at 16384 + 4096 + 1024: 0.535 s, 32768 + 8192 + 2048:
// 0.536 s, 16384 + 8192 + 2048: 0.538 s, 32768 + 16384 + 4096: 0.552 s
#ifndef SHUF_CUT_HI
#define SHUF_CUT_HI 16384u
#endif
#ifndef SHUF_CUT_LO
#define SHUF_CUT_LO 4096u
#endif
#define SHUF_SQ_BIG 4096u
#define SHUF_SQ_SMALL 1024u
#define SHUF_CAP16 65536u
enum { SH_GEN0 = 0, SH_GEN1 = 1, SH_CPUB = 2, SH_FILL = 3, SH_TAIL = 4, SH_DONE = 5, SH_ATOP = 6, SH_EM0 = 7, SH_EM1 = 8, SH_CSTOP = 9,
       SH_ABORT = 10 };  // words of the control block
__device__ int32_t g_async_fault = 0;  // fault bits of asynchronous kernels on this device (offsim_async_faults, include/offsim.h)
// Every wait of one role for another is bounded.  A role that has polled SHUF_SPIN_LIMIT times (with s_sleep: ~0.1 s, five orders of
// magnitude beyond any legitimate wait) raises SH_ABORT; a role that finds SH_ABORT raised -- it looks every 1024 polls of a wait, so
// nothing of this sits on a common path -- raises OFFSIM_FAULT_SHUFFLE and ENDS (a wavefront that has ended no longer counts at the
// workgroup's barriers).  A slip of the ring protocol then costs the call its result (offsim_async_faults), not the stream.
#define SHUF_SPIN_LIMIT (1u << 20)
__device__ __noinline__ void shuf_bound_check(volatile __attribute__((address_space(3))) uint32_t *ctrl, uint32_t polls) {
    if (polls > SHUF_SPIN_LIMIT) ctrl[SH_ABORT] = 1u;
    if (ctrl[SH_ABORT]) {
        if ((threadIdx.x & 63u) == 0u) atomicOr(&g_async_fault, OFFSIM_FAULT_SHUFFLE);
        __builtin_amdgcn_endpgm();
    }
}
__device__ __forceinline__ void shuf_bound(volatile __attribute__((address_space(3))) uint32_t *ctrl, uint32_t &polls) {
#ifndef SHUF_NO_BOUND  // (SHUF_NO_BOUND: timing experiment only -- what the bound costs)
    if (__builtin_expect((++polls & 1023u) == 0u, 0)) shuf_bound_check(ctrl, polls);  // (out of line: the waits keep their shape)
#endif
}
#define SHUF_CH 512u  // keyed emit: positions per chunk (one turn of a wavefront: 4 pairs per lane)

// explicit LDS address space: keeps every ring / segment access a ds_* instruction (a generic pointer would make
// them flat_* operations, which also tie up the vector-memory counter)
typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;
typedef __attribute__((address_space(3))) volatile uint16_t lds_vu16;
typedef __attribute__((address_space(3))) volatile uint64_t lds_vu64;

__device__ __forceinline__ uint32_t sh_rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane(v); }  // (readfirstlane returns int)
__device__ __forceinline__ uint32_t sh_ld(lds_vu32 *p) { return sh_rfl(*p); }
__device__ __forceinline__ void sh_st(lds_vu32 *p, uint32_t v) { *p = v; }
__device__ __forceinline__ uint64_t sh_lowmask(uint32_t k) { return k >= 64u ? ~0ull : ((1ull << k) - 1ull); }
__device__ __forceinline__ uint32_t sh_ff1(uint64_t m) { return (uint32_t)__ffsll((unsigned long long)m) - 1u; }
__device__ __forceinline__ int sh_rank(uint64_t m) {  // set bits of m in front of this lane
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// LDS layout: [ctrl 16 w][draw ring RG w][j ring SQ w + 64 w trash][64 w tag winners][segment]
constexpr uint32_t shuf_fixed_lds_bytes(uint32_t sq) { return 4u * (16u + SHUF_RG + sq + 64u + 64u); }

// LDS16 = true : segments of n_lo < n <= n_hi <= 65536 rows, kept in LDS as 16-bit local indices (one launch per size class,
//                so that short chains are not held to the occupancy of the longest)
// LDS16 = false: segments with n > n_lo rows, shuffled in place in global memory (32-bit)
// TOP = 0, STOP = 1: the whole chain.  Keyed chains of the longest size class are cut into launches instead.  The chance
// of a conflict in a group of 64 steps is ~6000 / i, so the low end of a chain is mostly conflicts and settles and costs far
// more than its share of the steps; and a step never touches a position above its own, so the part of the segment that
// still matters shrinks with i -- yet the chain holds a whole CU's LDS, one chain per CU, every role a single wavefront
// bound by its own latencies.  A launch <TOP, STOP> runs the steps TOP-1 .. STOP of every chain (TOP = 0: from n-1, on the
// identity): it loads the TOP low positions as the previous launch left them (16-bit local rows, from the loc stream) and
// the number of 32-bit draws used so far (left in the chain's first digest word), continues the random stream from that
// count, and writes out the positions below TOP -- final above STOP, as they stand below.  Cuts are powers of two: mask
// boundaries, where C's batches end exactly anyway.  With cuts at 16384 and 4096 the launches hold one, three and seven
// chains per CU; the chains of a CU hide each other's latencies.
template <bool LDS16, uint32_t SHUF_SQ, uint32_t TOP = 0, uint32_t STOP = 1>
__global__ void __launch_bounds__(256)
    k_shuffle_wave(const uint32_t *__restrict__ seg_off, int32_t n_slots, int64_t N, int64_t N0, const uint64_t *__restrict__ seeds,
                   int32_t n_perm, uint32_t *__restrict__ perm, uint32_t *__restrict__ init_perm, uint32_t n_lo, uint32_t n_hi,
                   const uint32_t *__restrict__ dig32, uint32_t *__restrict__ dig_out, uint16_t *__restrict__ loc_out, uint32_t a_xchg) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    lds_vu32 *ctrl = (lds_vu32 *)lds_raw;
    lds_vu32 *ring = ctrl + 16;
    lds_vu32 *jq = ring + SHUF_RG;  // partners in step order; [SQ .. SQ+63] takes the stores of rejected lanes
    lds_vu32 *win = jq + SHUF_SQ + 64u;  // A: "lane l won a tag" (all zero between uses)
    lds_vu16 *x16 = (lds_vu16 *)(win + 64u);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;

    // chain of this workgroup: the init queue first (usually the longest chain), then state by state
    const int32_t s_idx = (int32_t)(blockIdx.x / (uint32_t)n_perm);
    const int32_t r = (int32_t)(blockIdx.x - (uint32_t)s_idx * (uint32_t)n_perm);
    const int32_t s = s_idx == 0 ? n_slots : s_idx - 1;
    uint32_t n, base_val;
    uint32_t *xg;
    if (s < n_slots) {
        const uint32_t b = seg_off[s];
        n = seg_off[s + 1] - b;
        xg = perm + (int64_t)r * N + b;
        base_val = b;
    } else {
        n = (uint32_t)N0;
        xg = init_perm + (int64_t)r * N0;
        base_val = 0;
    }
    if (n == 0) return;
    if (n <= n_lo || (LDS16 && n > n_hi)) return;  // this launch serves the chains with n_lo < n <= n_hi (its LDS is sized for n_hi)
    volatile uint32_t *x32 = (volatile uint32_t *)xg;
    constexpr bool FROM_STREAM = TOP != 0u, CUT = STOP > 1u;
    static_assert(!(FROM_STREAM || CUT) || LDS16, "the cut form is for LDS-resident keyed chains");
    static_assert((TOP & (TOP - 1u)) == 0u && (STOP & (STOP - 1u)) == 0u && (TOP == 0u || TOP > STOP), "cuts are powers of two");
    if ((FROM_STREAM || CUT) && (dig_out == nullptr || s >= n_slots)) return;
    const uint32_t n_rows = n;                  // rows of the segment (local rows are < n_rows)
    if (FROM_STREAM) n = TOP;                   // this launch sees the chain's low TOP positions only
    const uint32_t stop_i = STOP;               // the chain runs while i >= stop_i

    // keyed form: the queue order goes out as {digest, 16-bit row} streams (offsim_shuffle_queues_keys)
    const bool keyed = LDS16 && dig_out != nullptr && s < n_slots;
    const uint32_t *dsrc = keyed ? dig32 + base_val : nullptr;
    uint32_t *dg = keyed ? dig_out + (int64_t)r * N + base_val : nullptr;
    uint16_t *lc = keyed ? loc_out + (int64_t)r * N + base_val : nullptr;
    const uint32_t n_chunks = (n + SHUF_CH - 1u) / SHUF_CH;
    // one chunk of the finished order, by one wavefront: positions [SHUF_CH * ch, SHUF_CH * (ch + 1)) -- the digest of the
    // candidate at every queue position (gathered from this segment's slice of dig32: L2-resident, the same few hundred KB for
    // every rollout's chain of this state) and its 16-bit row inside the segment
    auto emit_chunk = [&](uint32_t ch) {
        lds_vu32 *xw = (lds_vu32 *)x16;  // (volatile: read behind the progress word that says the chunk is final)
        const uint32_t pairs = (n + 1u) >> 1, k0 = ch * (SHUF_CH / 2u) + (uint32_t)lane;
        uint32_t two[4], d0[4], d1[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t k = k0 + 64u * (uint32_t)u;
            two[u] = k < pairs ? xw[k] : 0u;
        }
        if (CUT && ch < STOP / SHUF_CH) {  // positions below the cut are not final: only their rows go out, for the next launch
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t k = k0 + 64u * (uint32_t)u;
                lc[2u * k] = (uint16_t)(two[u] & 0xffffu);
                lc[2u * k + 1u] = (uint16_t)(two[u] >> 16);
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i1 = two[u] >> 16;  // (the slot behind an odd-length segment still holds its identity value n)
            d0[u] = dsrc[two[u] & 0xffffu];
            d1[u] = dsrc[i1 < n_rows ? i1 : 0u];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t k = k0 + 64u * (uint32_t)u;
            if (k < pairs) {
                dg[2u * k] = d0[u];
                lc[2u * k] = (uint16_t)(two[u] & 0xffffu);
                if (2u * k + 1u < n) {
                    dg[2u * k + 1u] = d1[u];
                    lc[2u * k + 1u] = (uint16_t)(two[u] >> 16);
                }
            }
        }
    };

    if (threadIdx.x < 16u) ctrl[threadIdx.x] = threadIdx.x == SH_ATOP ? n - 1u : (threadIdx.x == SH_EM0 || threadIdx.x == SH_EM1) ? n_chunks : 0u;
    if (threadIdx.x < 64u) win[threadIdx.x] = 0;
    uint32_t c_start = 0;  // 32-bit draws the launches before this one used
    if (FROM_STREAM) {  // the low positions as the previous launch left them, and its draw count
        c_start = dg[0];
        for (uint32_t k = threadIdx.x; k < TOP; k += 256u) x16[k] = lc[k];
    } else if (LDS16) {  // identity, two entries per lane and store
        __attribute__((address_space(3))) uint32_t *xw = (__attribute__((address_space(3))) uint32_t *)x16;
        for (uint32_t k = threadIdx.x; 2u * k < n; k += 256u) xw[k] = ((2u * k + 1u) << 16) | (2u * k);
    } else {
        for (uint32_t k = threadIdx.x; k < n; k += 256u) xg[k] = base_val + k;
    }
    __syncthreads();

    // Ring protocol.  The rings live in LDS, every role is one wavefront, and the LDS unit executes the DS
    // instructions of a wavefront in issue order: data written before a counter is visible before the counter, and a
    // read issued after a counter was seen comes after the data.  So the counters are plain volatile words (the
    // volatile qualifier keeps the compiler from reordering them) and no s_waitcnt is spent on publishing.
    // -DSHUF_PROF (tools/prof_shuffle.py): per role, clocks in its loop / waiting for its neighbour / in the named extra
    // activity; written over the first 13 digests of the chain's stream, so the results of such a build are not usable
#ifdef SHUF_PROF
    uint64_t pf_wait = 0, pf_w0 = 0, pf_extra = 0;
    const uint64_t pf_begin = __builtin_amdgcn_s_memtime();
#define SPW0() pf_w0 = __builtin_amdgcn_s_memtime()
#define SPW1() pf_wait += __builtin_amdgcn_s_memtime() - pf_w0
#define SPX1() pf_extra += __builtin_amdgcn_s_memtime() - pf_w0
#else
#define SPW0()
#define SPW1()
#define SPX1()
#endif
    if (n >= 2) {
        if (wave == 0 || wave == 3) {
            // ---------------- G (two wavefronts, alternate blocks of 128 draws): raw draws.  In block k lane l owns 64-bit
            // output 64*k + l = 32-bit draws 2*(64*k + l) and +1 (next_uint32 hands out the low half first, then the
            // buffered high half).
            const uint32_t g = wave == 0 ? 0u : 1u;
            const PcgInit p = pcg_seed(seeds[r]);
            const Jump j128 = pcg_jump(p.inc, 128);
            U128 st = pcg_apply(pcg_jump(p.inc, (uint64_t)(c_start >> 1) + 64ull * g + (uint64_t)lane + 1), p.state);
            uint32_t blk = g, done_blocks = 0, cpub = 0;  // blk = index of the block this wavefront writes next
            // Keyed form: a step of the chain never touches a position above its own, so the order is final from the top down
            // while the chain still runs.  The two G wavefronts are ahead of C most of the time; while they wait for room in
            // the draw ring they write the finished chunks out (even / odd chunks), one per poll; what is left when the chain
            // ends goes out with the whole workgroup.
            int32_t em_next = (int32_t)n_chunks - 1 - (int32_t)(((n_chunks - 1u) & 1u) != g);  // highest chunk of this parity
            for (;;) {
                bool stop = false;
                SPW0();
                uint32_t polls = 0;
                while ((blk + 1u) * 128u - cpub > SHUF_RG) {
                    shuf_bound(ctrl, polls);
                    if (sh_ld(ctrl + SH_DONE)) {
                        stop = true;
                        break;
                    }
                    if (keyed && em_next >= 0 && sh_ld(ctrl + SH_ATOP) < SHUF_CH * (uint32_t)em_next) {
#ifdef SHUF_PROF
                        const uint64_t e0 = __builtin_amdgcn_s_memtime();
                        emit_chunk((uint32_t)em_next);
                        pf_extra += __builtin_amdgcn_s_memtime() - e0;  // G: writing chunks out (inside its waiting time)
#else
                        emit_chunk((uint32_t)em_next);
#endif
                        sh_st(ctrl + (g ? SH_EM1 : SH_EM0), (uint32_t)em_next);  // this parity is out from here up
                        em_next -= 2;
                    } else {
                        __builtin_amdgcn_s_sleep(1);
                    }
                    cpub = sh_ld(ctrl + SH_CPUB);
                }
                SPW1();
                if (stop) break;
                const uint64_t o = pcg_output(st);
                st = pcg_apply(j128, st);
                const uint32_t idx = (blk * 128u + 2u * (uint32_t)lane) & (SHUF_RG - 1u);
                // (one aligned 64-bit store; two 32-bit stores time the same here, 0.486 against 0.484 s per reset: the 256 cycles
                // tools/micro/issue.hip reports for ds_write_b64 are those of a MISALIGNED access, its addresses are lane * 4)
                *(lds_vu64 *)(ring + idx) = o;  // low half first
                blk += 2u;
                done_blocks++;
                sh_st(ctrl + (g ? SH_GEN1 : SH_GEN0), done_blocks);
            }
        } else if (wave == 1) {
            // ---------------- C: the j sequence, 2 x 64 draws per iteration.  The second batch starts from i2 = i - accepts
            // of the first, so the wavefront has two independent instruction streams.
            uint32_t i = n - 1u, c = c_start & 1u, avail = 0, fill = 0, tail = 0;  // (an odd count: the high half of a 64-bit output is next)
            uint32_t mask = 0xffffffffu >> __builtin_clz(i);
            int lowpow = (int)((mask >> 1) + 1u);  // steps below this index use the next smaller mask
            auto wait_draws = [&](uint32_t upto) {
                if (__builtin_expect(upto <= avail, 1)) return;  // (marked for the block layout of C's loop, like the room check below: 0.472 -> 0.470 s)
                const bool waited = upto > avail;
                if (waited) SPW0();
                uint32_t polls = 0;
                while (upto > avail) {
                    const uint64_t gg = *(lds_vu64 *)(ctrl + SH_GEN0);
                    const uint32_t g0 = sh_rfl((uint32_t)gg), g1 = sh_rfl((uint32_t)(gg >> 32));
                    avail = 128u * (g0 <= g1 ? 2u * g0 : 2u * g1 + 1u);  // contiguous blocks
                    if (upto > avail) {
                        shuf_bound(ctrl, polls);
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                if (waited) SPW1();
            };
            auto wait_room = [&](uint32_t upto) {  // the j ring may hold entries [tail, tail + SQ)
                if (__builtin_expect(upto - tail <= SHUF_SQ, 1)) return;
                const bool waited = upto - tail > SHUF_SQ;
                if (waited) SPW0();
                uint32_t polls = 0;
                while (upto - tail > SHUF_SQ) {
                    tail = sh_ld(ctrl + SH_TAIL);
                    if (upto - tail > SHUF_SQ) {
                        shuf_bound(ctrl, polls);
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                if (waited) SPX1();  // C: waiting for room in the j ring (A behind)
            };
            // strike optimistic accepts that do not hold, first one first, until all hold
            auto settle = [&](uint64_t &bal, int &rk, uint32_t v, uint32_t ib) {
                uint64_t f = bal & __ballot((int)v > (int)ib - rk);
                while (f) {
                    bal &= ~(1ull << sh_ff1(f));
                    rk = sh_rank(bal);
                    f = bal & __ballot((int)v > (int)ib - rk);
                }
            };
            wait_draws(c + 128u);
            uint32_t ra = ring[c + (uint32_t)lane], rb = ring[c + 64u + (uint32_t)lane], rc = 0, rd = 0;
#ifndef SHUF_C_PAIRS
#define SHUF_C_PAIRS 2
#endif
            // One iteration: the pair of batches in (ra, rb) -- SEL: in (rc, rd) -- and the next pair prefetched into the other two, which is
            // where the draws of the NEXT iteration are when this one is over, on the boundary path too.  The loop below alternates, so that
            // no pair is copied, and only every second iteration publishes the draw counter (no test).  (SHUF_C_PAIRS 1: one per trip.)
            auto iter = [&](auto SEL, const bool publish) __attribute__((always_inline)) {
                constexpr bool S = decltype(SEL)::value;
                const uint32_t r1 = S ? rc : ra, r2 = S ? rd : rb;
                wait_draws(c + 256u);  // this pair and the prefetch of the next
                uint32_t p1 = ring[(c + 128u + (uint32_t)lane) & (SHUF_RG - 1u)];
                uint32_t p2 = ring[(c + 192u + (uint32_t)lane) & (SHUF_RG - 1u)];
                const uint32_t v1 = r1 & mask, v2 = r2 & mask;
                uint64_t bal1 = __ballot(v1 <= i);  // optimistic: accepted if no earlier lane of the batch had been accepted
                uint32_t n1 = (uint32_t)__popcll(bal1);
                uint32_t i2 = i - n1;
                uint64_t bal2 = __ballot((int)v2 <= (int)i2);
                int rk1 = sh_rank(bal1), rk2 = sh_rank(bal2);  // accepts in front of the lane: its draw serves step i - rk
                // one test for both batches (93 % of the pairs settle nothing; a test and a branch per batch: reset 0.471 against 0.468 s): the
                // second batch's optimistic ranks are only right if the first one's hold, so whenever either fails the pair is settled in order
                const uint64_t f1 = bal1 & __ballot((int)v1 > (int)i - rk1), f2 = bal2 & __ballot((int)v2 > (int)i2 - rk2);
                if (__builtin_expect((f1 | f2) != 0ull, 0)) {
                    if (f1) {
                        settle(bal1, rk1, v1, i);
                        n1 = (uint32_t)__popcll(bal1);
                        i2 = i - n1;
                        bal2 = __ballot((int)v2 <= (int)i2);
                        rk2 = sh_rank(bal2);
                    }
                    if ((bal2 & __ballot((int)v2 > (int)i2 - rk2)) != 0ull) settle(bal2, rk2, v2, i2);
                }
                const uint32_t n2 = (uint32_t)__popcll(bal2);
                const int i_new = (int)i2 - (int)n2;
                if (__builtin_expect(i_new >= lowpow, 1)) {
                    // after settling the accepted lanes are exactly those with v <= i - rk: append them in order; the
                    // others store into the trash words behind the ring
                    wait_room(fill + n1 + n2);
                    // (the settled ballots ARE the accepted lanes: one v_cndmask under the mask instead of the accept test once more per lane --
                    // 0.466 -> 0.4625 s)
                    uint32_t a1, a2;
                    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a1) : "v"(SHUF_SQ + (uint32_t)lane), "v"((fill + (uint32_t)rk1) & (SHUF_SQ - 1u)), "s"(bal1));
                    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(a2) : "v"(SHUF_SQ + (uint32_t)lane), "v"((fill + n1 + (uint32_t)rk2) & (SHUF_SQ - 1u)), "s"(bal2));
                    jq[a1] = v1;
                    jq[a2] = v2;
                    fill += n1 + n2;
                    sh_st(ctrl + SH_FILL, fill);
                    i = (uint32_t)i_new;
                    c += 128u;
                } else {
                    // a mask boundary (or the end of the chain) inside the pair: batch 1 only, and only the draws -- accepted
                    // or not -- of steps at or above the boundary
                    const uint64_t lowm = __ballot((int)i - rk1 < lowpow);
                    const uint64_t below = (lowm & (0ull - lowm)) - 1ull;  // lanes in front of the first such draw (all if none)
                    const uint64_t acc = bal1 & below;
                    const uint32_t na = (uint32_t)__popcll(acc);
                    wait_room(fill + na);
                    jq[((acc >> lane) & 1ull) ? ((fill + (uint32_t)rk1) & (SHUF_SQ - 1u)) : SHUF_SQ + (uint32_t)lane] = v1;
                    fill += na;
                    sh_st(ctrl + SH_FILL, fill);
                    i -= na;
                    c += (uint32_t)__popcll(below);
                    if ((int)i < lowpow && i >= 1u) {  // crossed a power of two: the mask shrinks
                        mask = 0xffffffffu >> __builtin_clz(i);
                        lowpow = (int)((mask >> 1) + 1u);
                    }
                    wait_draws(c + 128u);
                    p1 = ring[(c + (uint32_t)lane) & (SHUF_RG - 1u)];
                    p2 = ring[(c + 64u + (uint32_t)lane) & (SHUF_RG - 1u)];
                }
                if (S) {
                    ra = p1;
                    rb = p2;
                } else {
                    rc = p1;
                    rd = p2;
                }
                // (a store costs less than the test and the branch that published every 256 draws -- reset 0.485 -> 0.480 s)
                if (publish) sh_st(ctrl + SH_CPUB, c);
            };
#if SHUF_C_PAIRS == 1
            while (i >= stop_i) {
                iter(std::false_type{}, true);
                ra = rc;
                rb = rd;
            }
#else
            while (i >= stop_i) {
                iter(std::false_type{}, false);
                if (__builtin_expect(i < stop_i, 0)) break;
                iter(std::true_type{}, true);
            }
            sh_st(ctrl + SH_CPUB, c);
#endif
            if (CUT) sh_st(ctrl + SH_CSTOP, (c_start & ~1u) + c);  // (the cut is a mask boundary: i == STOP - 1 here; draws from the stream's start)
            sh_st(ctrl + SH_DONE, 1u);
        } else {
            // ---------------- A: apply, 64 consecutive steps i_top, i_top - 1, ... per iteration (lane l: step i_top - l).
            // One tag round finds every pair of equal partners; together with the arithmetic cuts that gives all the places
            // where the group has to be split, and the pieces are then plain conflict-free swaps.
            uint32_t i_top = n - 1u, done = 0, fill = 0;
            auto xrd = [&](uint32_t k) -> uint32_t { return LDS16 ? (uint32_t)x16[k] : x32[k]; };
            auto xwr = [&](uint32_t k, uint32_t val) {
                if (LDS16) x16[k] = (uint16_t)val;
                else x32[k] = val;
            };
            // the piecewise path: lanes < cnt hold (il, v); their tags are still in place, b = the values under the tags,
            // confl = lanes whose partner is a later step of the group, F = lanes that lost a tag
            auto piecewise = [&](uint32_t cnt, uint32_t i_first, uint32_t il, uint32_t v, uint32_t b, uint32_t tg, uint64_t confl, uint64_t F) {
                const bool in = (uint32_t)lane < cnt;
                if (in) xwr(v, b);  // take the tags back
                uint64_t cuts = 0;
                while (confl) {
                    cuts |= 1ull << (i_first - sh_rfl((uint32_t)__builtin_amdgcn_readlane((int)v, (int)sh_ff1(confl))));  // the lane that owns step v
                    confl &= confl - 1ull;
                }
                if (F) {  // every lane with a shared partner except the first of them: losers and the winners they saw
                    const bool lost = tg != (uint32_t)lane;
                    if (lost) win[tg] = 1u;
                    const uint32_t w = win[(uint32_t)lane];
                    if (lost) win[tg] = 0u;
                    const uint64_t D = F | __ballot(w != 0u);
                    cuts |= D & (D - 1ull);
                }
                // piece number of every lane = cuts at or in front of it; the pieces are applied in order
                const uint32_t pid = in ? (uint32_t)sh_rank(cuts) + (uint32_t)((cuts >> lane) & 1ull) : 0xffffffffu;
                const uint32_t n_pieces = (uint32_t)__popcll(cuts & sh_lowmask(cnt) & ~1ull) + 1u;
                for (uint32_t pc = (uint32_t)(cuts & 1ull); pc < n_pieces + (uint32_t)(cuts & 1ull); pc++) {
                    if (pid == pc) {
                        const uint32_t a2 = xrd(il), b2 = xrd(v);
                        xwr(il, b2);
                        xwr(v, a2);
                    }
                }
            };
            // full groups: no lane masks anywhere on the common path.  (Fetching the next group's partners early was
            // measured slower: this wavefront is bound by the instructions it issues, not by the LDS round trips.)
            const uint32_t lo = stop_i - 1u;  // the chain's steps are i_top .. lo + 1
#ifdef SHUF_FAULT_INJECT  // (test build: this role never starts, so the others run into their bounds -- and so does this wait)
            for (uint32_t polls = 0;; __builtin_amdgcn_s_sleep(1)) shuf_bound(ctrl, polls);
#endif
            // The loop is written for its block layout (round 4): what a group costs beyond its LDS round trips is scalar control flow, and the
            // straightforward `while` form of this loop came out with four TAKEN branches on its common path.  Cold exits are marked, the group
            // with a conflict is an early `continue`, the loop is rotated: two taken branches per group (reset 0.479 -> 0.472 s; a hand-written
            // form with one was not faster -- this wavefront re-reads the fill counter every other group and would leave the block as often; the
            // piecewise path as an out-of-line function: 0.515 s).
            // The loop is written for what its instructions cost the wavefront (round 4): what a group costs beyond its LDS round trips is
            // scalar control flow and 16-cycle stores, and the straightforward `while` form came out with four TAKEN branches on its common
            // path.  Cold exits are marked, the group with a conflict returns early, and the loop runs SHUF_A_GROUPS groups per trip, of which
            // only the last publishes the ring tail and the final-position word (C sizes its appends by the one, the G wavefronts write the
            // finished order out by the other: both may lag a few groups, and the wavefront publishes before it ever waits at the loop's
            // end) -- no test, no branch: reset 0.479 -> 0.472 s for the layout, 0.4645 -> 0.4515 s for two groups per trip.  (A hand-written
            // form with one taken branch per group was not faster; the piecewise path as an out-of-line function: 0.515 s.)
            auto group = [&](const bool publish) __attribute__((always_inline)) {
                if (__builtin_expect(fill - done < 64u, 0)) {
                    SPW0();
                    uint32_t polls = 0;
                    while (fill - done < 64u) {
                        fill = sh_ld(ctrl + SH_FILL);
                        if (fill - done < 64u) {
                            shuf_bound(ctrl, polls);
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    SPW1();
                }
                const uint32_t v = jq[(done + (uint32_t)lane) & (SHUF_SQ - 1u)];
                done += 64u;
                if (publish) sh_st(ctrl + SH_TAIL, done);
                const uint32_t i_first = i_top;
                const uint32_t il = i_first - (uint32_t)lane;
                i_top -= 64u;
                const uint64_t confl = __ballot(v < il) & __ballot(v > i_top);
                const uint32_t a = xrd(il), b = xrd(v);
                xwr(v, (uint32_t)lane);
                const uint32_t tg = xrd(v);
                const uint64_t F = __ballot(tg != (uint32_t)lane);
                if (__builtin_expect((confl | F) != 0ull, 0)) {
                    SPW0();
                    piecewise(64u, i_first, il, v, b, tg, confl, F);
                    SPX1();
                    sh_st(ctrl + SH_ATOP, i_top);
                    return;
                }
                xwr(il, b);
                xwr(v, a);
                if (publish) sh_st(ctrl + SH_ATOP, i_top);
            };
            // Round 5, the exchange form of a group (LDS-resident segments, a_xchg): step l of the group is swap(x[i_l], x[v_l]) with the
            // i_l distinct and above every partner of a LATER lane, so as long as no partner is a later step's own position (confl, plain
            // arithmetic -- those groups take the tag round and the pieces as before) the group is: a_l = x[i_l] read up front (it does
            // not wait for the partners: one LDS round trip together with them), ONE atomic exchange of the 16-bit entry x[v_l] <- a_l per
            // lane, x[i_l] <- what came back.  Lanes with EQUAL partners need no tags and no pieces: the LDS serves the lanes of one
            // instruction that hit the same word in ascending lane order (= step order), so a lane gets back exactly what the sequential
            // chain would find there (the property offsim_lds_order_ok checks on every device before this form is chosen).  The exchange of
            // a 16-bit half is ds_mskor_rtn_b32 (D = (D & ~mask) | data, returns the old word) on the entry's half of its dword; the other
            // half, which another lane may be exchanging in the same instruction, is preserved by the masks.  Four DS instructions and
            // two round trips per group instead of seven and three.
            typedef __attribute__((address_space(3))) unsigned char lds_b8;
            const uint32_t x16_a = (uint32_t)(uintptr_t)(lds_b8 *)x16;
            auto group_x = [&](const bool publish) __attribute__((always_inline)) {
                if (__builtin_expect(fill - done < 64u, 0)) {
                    SPW0();
                    uint32_t polls = 0;
                    while (fill - done < 64u) {
                        fill = sh_ld(ctrl + SH_FILL);
                        if (fill - done < 64u) {
                            shuf_bound(ctrl, polls);
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    SPW1();
                }
                const uint32_t i_first = i_top;
                const uint32_t il = i_first - (uint32_t)lane;
                const uint32_t v = jq[(done + (uint32_t)lane) & (SHUF_SQ - 1u)];
                const uint32_t a = (uint32_t)x16[il];
                done += 64u;
                if (publish) sh_st(ctrl + SH_TAIL, done);
                i_top -= 64u;
                const uint64_t confl = __ballot(v < il) & __ballot(v > i_top);
                if (__builtin_expect(confl != 0ull, 0)) {  // a partner is a later step's own position: the tag round and the pieces
                    SPW0();
                    const uint32_t b = (uint32_t)x16[v];
                    x16[v] = (uint16_t)lane;
                    const uint32_t tg = (uint32_t)x16[v];
                    piecewise(64u, i_first, il, v, b, tg, confl, __ballot(tg != (uint32_t)lane));
                    SPX1();
                    sh_st(ctrl + SH_ATOP, i_top);
                    return;
                }
                const uint32_t sh = (v & 1u) << 4;
                uint32_t old;
                asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)"
                             : "=v"(old)
                             : "v"(x16_a + ((v & ~1u) << 1)), "v"(0xffffu << sh), "v"(a << sh)
                             : "memory");
                x16[il] = (uint16_t)(old >> sh);
                if (publish) sh_st(ctrl + SH_ATOP, i_top);
            };
#ifndef SHUF_A_GROUPS
#define SHUF_A_GROUPS 2
#endif
            // The same groups as one hand-scheduled loop (SHUF_A_ASM, the default): the compiled form above waits for every LDS access
            // of the volatile segment in turn -- three round trips per group (partners + own entries, the exchange, the store of what came
            // back before the next group's reads may go).  The LDS executes one wavefront's DS instructions in issue order, so none of
            // these waits is needed for correctness; here the NEXT group's partners and own entries are requested as soon as this group's
            // exchange has been issued (its reads come behind the exchange in the LDS queue: a partner of this group that is one of the
            // next group's own positions is seen), the wavefront waits for the exchange only, stores what came back, and finds the next
            // group's operands nearly there.  Two copies of the group per trip, the second publishes the ring tail and the final-position
            // word.  Left with a code: 0 = the chain's full groups are done, 1 = the group at i_top has a partner among its own later
            // positions (nothing of it applied; v_cur holds its partners), 2 = the next group's partners are not in the ring yet.
            uint32_t v_cur = 0, a_cur = 0, il_cur = 0;
            auto run_x = [&]() __attribute__((always_inline)) -> uint32_t {
                uint32_t code, t_sh, t_ad, t_m, t_d, t_old, t_wa, t_ja;
                const uint32_t jq_lane = (uint32_t)(uintptr_t)(lds_b8 *)jq + 4u * (uint32_t)lane, ctrl_a = (uint32_t)(uintptr_t)(lds_b8 *)ctrl;
                il_cur = i_top - (uint32_t)lane;
#define SHUF_AX_GROUP(G, GW, WAIT, NEXT, PUBLISH)                                                                             \
                G ":\n\t"                                                                                                    \
                "s_waitcnt lgkmcnt(" WAIT ")\n\t"                            /* this group's partners and own entries (NOT the stores issued behind them: the LDS answers one wavefront in issue order) */ \
                GW ":\n\t"                                                                                                   \
                "s_sub_u32 s21, %[it], 64\n\t"                                                                               \
                "v_cmp_lt_u32_e32 vcc, %[v], %[il]\n\t"                                                                      \
                "v_cmp_lt_u32_e64 s[22:23], s21, %[v]\n\t"                                                                   \
                "v_lshlrev_b32_e32 %[sh], 4, %[v]\n\t"                      /* (v & 1) << 4 in its low five bits */         \
                "v_lshrrev_b32_e32 %[ad], 1, %[v]\n\t"                                                                       \
                "s_and_b64 s[22:23], s[22:23], vcc\n\t"                     /* partners among the group's own later positions */ \
                "s_cbranch_scc1 91f\n\t"                                                                                      \
                "v_lshl_add_u32 %[ad], %[ad], 2, %[x16a]\n\t"               /* the dword that holds entry v */               \
                "v_lshlrev_b32_e32 %[m], %[sh], %[ffff]\n\t"                                                                 \
                "v_lshlrev_b32_e32 %[d], %[sh], %[a]\n\t"                                                                    \
                "ds_mskor_rtn_b32 %[old], %[ad], %[m], %[d]\n\t"           /* x[v] <- a, lanes with one v in lane order */   \
                "v_lshl_add_u32 %[wa], %[il], 1, %[x16a]\n\t"                                                                \
                "s_add_u32 %[done], %[done], 64\n\t"                                                                         \
                "s_mov_b32 %[it], s21\n\t"                                                                                   \
                "v_subrev_u32_e32 %[il], 64, %[il]\n\t"                                                                      \
                "s_cmp_lt_u32 %[it], %[lo64]\n\t"                                                                            \
                "s_cbranch_scc1 92f\n\t"                                                                                      \
                "s_sub_u32 s20, %[fill], %[done]\n\t"                                                                        \
                "s_cmp_lt_u32 s20, 64\n\t"                                                                                   \
                "s_cbranch_scc1 93f\n\t"                                                                                      \
                "s_and_b32 s20, %[done], %[sqm]\n\t"                                                                         \
                "v_lshl_add_u32 %[ja], s20, 2, %[jql]\n\t"                                                                   \
                "ds_read_b32 %[v], %[ja]\n\t"                                /* the next group's partners ... */             \
                "v_lshl_add_u32 %[ja], %[il], 1, %[x16a]\n\t"                                                                \
                "ds_read_u16 %[a], %[ja]\n\t"                                /* ... and own entries, behind the exchange */  \
                "s_waitcnt lgkmcnt(2)\n\t"                                   /* the exchange */                              \
                "s_nop 0\n\t"                                                                                                \
                "v_lshrrev_b32_e32 %[old], %[sh], %[old]\n\t"                                                                \
                "ds_write_b16 %[wa], %[old]\n\t"                             /* x[i] <- what was at x[v] */                  \
                PUBLISH                                                                                                      \
                "s_branch " NEXT "\n\t"
#define SHUF_AX_PUBLISH                                                                                                       \
                "v_mov_b32_e32 %[m], %[done]\n\t"                                                                            \
                "v_mov_b32_e32 %[d], %[it]\n\t"                                                                              \
                "ds_write_b32 %[ctrl], %[m] offset:16\n\t"                   /* SH_TAIL */                                   \
                "ds_write_b32 %[ctrl], %[d] offset:24\n\t"                   /* SH_ATOP */
                asm volatile(
                    "s_and_b32 s20, %[done], %[sqm]\n\t"
                    "v_lshl_add_u32 %[ja], s20, 2, %[jql]\n\t"
                    "ds_read_b32 %[v], %[ja]\n\t"
                    "v_lshl_add_u32 %[ja], %[il], 1, %[x16a]\n\t"
                    "ds_read_u16 %[a], %[ja]\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    "s_branch 82f\n\t"
#if SHUF_A_GROUPS >= 4
                    SHUF_AX_GROUP("80", "82", "3", "84f", "")                /* behind the reads: the entry store and the two words of the last copy */
                    SHUF_AX_GROUP("84", "85", "1", "86f", "")
                    SHUF_AX_GROUP("86", "87", "1", "81f", "")
#else
                    SHUF_AX_GROUP("80", "82", "3", "81f", "")                /* behind the reads: the entry store and the two words of the copy below */
#endif
                    SHUF_AX_GROUP("81", "83", "1", "80b", SHUF_AX_PUBLISH)   /* behind the reads: the entry store of the copy above */
                    "91:\n\t"                                                /* a conflict: nothing of this group is applied */
                    "s_mov_b32 %[code], 1\n\t"
                    "s_branch 99f\n\t"
                    "92:\n\t"                                                /* the chain's last full group */
                    "s_mov_b32 %[code], 0\n\t"
                    "s_branch 94f\n\t"
                    "93:\n\t"                                                /* the ring does not hold the next group's partners yet */
                    "s_mov_b32 %[code], 2\n\t"
                    "94:\n\t"
                    "s_waitcnt lgkmcnt(0)\n\t"
                    "s_nop 0\n\t"
                    "v_lshrrev_b32_e32 %[old], %[sh], %[old]\n\t"
                    "ds_write_b16 %[wa], %[old]\n\t"
                    "99:\n\t"
                    "s_waitcnt lgkmcnt(0)"
                    : [it] "+s"(i_top), [done] "+s"(done), [il] "+v"(il_cur), [v] "+v"(v_cur), [a] "+v"(a_cur), [code] "=&s"(code), [sh] "=&v"(t_sh),
                      [ad] "=&v"(t_ad), [m] "=&v"(t_m), [d] "=&v"(t_d), [old] "=&v"(t_old), [wa] "=&v"(t_wa), [ja] "=&v"(t_ja)
                    : [fill] "s"(fill), [lo64] "s"(64u + lo), [sqm] "s"(SHUF_SQ - 1u), [x16a] "v"(x16_a), [jql] "v"(jq_lane), [ffff] "v"(0xffffu),
                      [ctrl] "v"(ctrl_a)
                    : "vcc", "scc", "memory", "s20", "s21", "s22", "s23");
                static_assert(SH_TAIL == 4 && SH_ATOP == 6, "the immediates of the applier's loop");
#undef SHUF_AX_GROUP
#undef SHUF_AX_PUBLISH
                return code;
            };
#ifndef SHUF_A_ASM
#define SHUF_A_ASM 1
#endif
            if (LDS16 && a_xchg && SHUF_A_ASM) {
                while (i_top >= 64u + lo) {
                    if (__builtin_expect(fill - done < 64u, 0)) {
                        SPW0();
                        uint32_t polls = 0;
                        while (fill - done < 64u) {
                            fill = sh_ld(ctrl + SH_FILL);
                            if (fill - done < 64u) {
                                shuf_bound(ctrl, polls);
                                __builtin_amdgcn_s_sleep(1);
                            }
                        }
                        SPW1();
                    }
                    const uint32_t code = run_x();
                    if (code == 1u) {  // the group at i_top: the tag round and the pieces, as in the compiled form
                        SPW0();
                        const uint32_t i_first = i_top, il = i_first - (uint32_t)lane, v = v_cur;
                        done += 64u;
                        i_top -= 64u;
                        const uint64_t confl = __ballot(v < il) & __ballot(v > i_top);
                        const uint32_t b = (uint32_t)x16[v];
                        x16[v] = (uint16_t)lane;
                        const uint32_t tg = (uint32_t)x16[v];
                        piecewise(64u, i_first, il, v, b, tg, confl, __ballot(tg != (uint32_t)lane));
                        SPX1();
                    }
                    sh_st(ctrl + SH_TAIL, done);
                    sh_st(ctrl + SH_ATOP, i_top);
                }
            } else if (LDS16 && a_xchg) {
                if (i_top >= 64u + lo) for (;;) {
                    group_x(false);
                    if (__builtin_expect(i_top < 64u + lo, 0)) break;
                    group_x(true);
                    if (__builtin_expect(i_top < 64u + lo, 0)) break;
                }
            } else
            // (written out, not as an inner loop: with `for (g ...) { group(false); if (...) goto out; }` the compiler's layout gave the gain away)
            if (i_top >= 64u + lo) for (;;) {
#if SHUF_A_GROUPS >= 4
                group(false);
                if (__builtin_expect(i_top < 64u + lo, 0)) break;
                group(false);
                if (__builtin_expect(i_top < 64u + lo, 0)) break;
#endif
#if SHUF_A_GROUPS >= 2
                group(false);
                if (__builtin_expect(i_top < 64u + lo, 0)) break;
#endif
                group(true);
                if (__builtin_expect(i_top < 64u + lo, 0)) break;
            }
            sh_st(ctrl + SH_TAIL, done);
            sh_st(ctrl + SH_ATOP, i_top);
            if (i_top > lo) {  // the last, partial group
                const uint32_t cnt = i_top - lo;
                uint32_t polls = 0;
                while (fill - done < cnt) {
                    fill = sh_ld(ctrl + SH_FILL);
                    if (fill - done < cnt) {
                        shuf_bound(ctrl, polls);
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                const uint32_t v = jq[(done + (uint32_t)lane) & (SHUF_SQ - 1u)];
                const uint32_t il = i_top - (uint32_t)lane;
                const bool in = (uint32_t)lane < cnt;
                const uint64_t confl = __ballot(in && v < il && v > lo);
                uint32_t b = 0, tg = (uint32_t)lane;
                if (in) {
                    b = xrd(v);
                    xwr(v, (uint32_t)lane);
                    tg = xrd(v);
                }
                piecewise(cnt, i_top, il, v, b, tg, confl, __ballot(tg != (uint32_t)lane));
            }
        }
    }
#ifdef SHUF_PROF
    const uint64_t pf_end = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
    if (LDS16) {  // the only HBM traffic of the chain: one coalesced write of the finished order
        __attribute__((address_space(3))) const uint32_t *xw = (__attribute__((address_space(3))) const uint32_t *)x16;
        if (keyed) {
            // the chunks the G wavefronts did not get to (the low end of the order)
            const uint32_t em0 = ctrl[SH_EM0], em1 = ctrl[SH_EM1];
            for (uint32_t ch = (uint32_t)wave; ch < n_chunks; ch += 4u)
                if (ch < ((ch & 1u) ? em1 : em0)) emit_chunk(ch);
        } else {
            for (uint32_t k = threadIdx.x; 2u * k < n; k += 256u) {
                const uint32_t two = xw[k];
                xg[2u * k] = base_val + (two & 0xffffu);
                if (2u * k + 1u < n) xg[2u * k + 1u] = base_val + (two >> 16);
            }
        }
    }
    if (CUT) {  // (behind the write-out of chunk 0, which holds this word)
        __syncthreads();
        if (threadIdx.x == 0) dg[0] = ctrl[SH_CSTOP];
    }
#ifdef SHUF_PROF
    __syncthreads();
    if (keyed && lane == 0 && n >= 64u) {  // units of 64 clocks
        dg[3 * wave + 0] = (uint32_t)((pf_end - pf_begin) >> 6);
        dg[3 * wave + 1] = (uint32_t)(pf_wait >> 6);
        dg[3 * wave + 2] = (uint32_t)(pf_extra >> 6);
        if (wave == 0) dg[12] = (uint32_t)((__builtin_amdgcn_s_memtime() - pf_begin) >> 6);
    }
#endif
}

}  // namespace offsim
