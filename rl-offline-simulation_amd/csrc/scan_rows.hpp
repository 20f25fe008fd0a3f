// scan_rows.hpp -- the headline evalMC scan, row-packed: FOUR rollouts per wavefront, one 16-lane DPP row each.
//
// Why: a PSRS rollout is one dependent chain, so with R = 4096 rollouts the kernel time is
// (steps per rollout) x (time per chain step), and with one rollout per wavefront (scan_win.hpp) a step cost ~67
// instructions of which a compute unit could retire only ~1.3 per cycle for its 16 rollouts: the CU was at its
// instruction-issue ceiling.  Here one instruction stream serves four rollouts: every per-rollout "scalar" (state, draw
// counter, cursors) is a VGPR value that is uniform inside a 16-lane row, cross-lane steps are DPP row operations, and
// a CU runs its 16 rollouts on 4 wavefronts (one per SIMD) at single-wavefront latency.
//
// The step itself has ONE LDS round trip on the chain (scan_win.hpp: two -- cursor pair, then the window row), and FOUR LDS
// instructions (entry read, draw read, row store, log store): on gfx950 a DS instruction holds the issuing wavefront for 16
// cycles (a dense 64-lane ds_read_b32: 8), an ordinary VALU / SALU instruction for 4.4, a not-taken branch for ~10
// (tools/micro/issue.hip), so the step is priced by its instruction mix, not by its dependent chain:
//   * windows are HEAD-ALIGNED: entry j of a state's 8-entry row is the j-th candidate still queued (0 = none loaded).
//     Accepting entry k pushes entries k+1.. to the front with one ds_write (lane j stores to slot (j-k-1) mod 8, the
//     vacated slots get 0), so the next look at that state needs no cursor to find its candidates -- and the chain keeps
//     NO cursor at all: a state's queue position is land[s] (the position behind its window, which only refills move)
//     minus the entries the window holds.  The step log is one dword (state left | done << 10 | candidates consumed), the
//     tick turns it into queue positions (a segmented prefix sum over the tick's steps, per state) for the reward pipeline;
//   * entries are stored BIASED, digest - 16 units of T21 (0 if that is negative), and the ring holds the draws as
//     k21 << 11 | 0x7ff: then "draw <= entry" (one v_sub_co, its borrow) is a CLEAR accept -- the candidate's threshold is
//     above the draw by 16 units or more -- and "entry < draw <= entry + 17 units" (one v_cmp on the difference) marks the
//     lanes that need the exact 53-bit look.  Lanes 0..7 of a row test draw c+j against entry j; the first clear accept
//     and its payload (done, z_next) come out of a 3-step DPP min over key = (j+1) << 26 | done << 10 | z_next (byte 3 of
//     the key is then 4 x the candidates consumed: the draw counter and the row shift take it as an SDWA operand);
//   * anything else -- a lane near a tie anywhere in the wavefront, no clear accept in a row, window dry, episode end,
//     draws running low -- is an event: the hand-scheduled loop is left, every row that has an event goes through
//     handle() (exact, reads the stream directly) while the rows without one commit their step; all rows therefore take
//     exactly one accepted step per iteration and the iteration / log index is wave-uniform (scalar).
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
#include "philox_dev.hpp"
#include "scan_win.hpp"  // pack_key / key_T, lds_u32
#include "shuffle_wave.hpp"  // g_async_fault

namespace offsim {

#define ROWS_TICK 16u
#define ROWS_RING 256u
#define ROWS_W 8u
#define ROWS_EMPTY 0u      // window slot without a candidate (a draw is never <= 0: ring entries have their low eleven bits set)
#define ROWS_BIAS 0x8000u  // window entries are digest - 16 units of T21: "draw <= entry" is then a CLEAR accept
#define ROWS_NOT_LANDED 0xffffffffu  // no digest has this value: the next-state field of a digest is < 0x3ff
#define ROWS_AMB 0xffff7800u  // entry - draw >= this (i.e. the draw exceeds the entry by at most 17 units): the exact look decides
// per-rollout LDS region (byte offsets); window rows are 32-byte aligned, the region a multiple of 512
#define RO_RING 0u       // 256 draws, k21 << 11 | 0x7ff: a tick (16 looks of <= 8 candidates) never runs out, so the chain loop does not check
#define RO_INIT 1024u    // ring of 32 upcoming initial states (slot or -1), entry k of the queue at (k - first) & 31
#define RO_LOG 1152u     // 16 step-log words: state left | done << 10 | candidates consumed (rows_log_k)
#define RO_LOG2 1216u    // second log buffer (HELPER: the chain fills one while the helper wavefront reads the other)
#define RO_SYNC 1280u    // HELPER: hand-off words between a rollout's chain wavefront and its helper wavefront (32 B), scratch words (64 B at +64)
#define RO_LOGH 1408u    // 2 x 16 halfwords beside the step logs: the high bits of the accepted candidate's local row (stream format B, exact-path steps)
enum { SY_TICK = 0, SY_HTICK = 4, SY_C = 8, SY_GEN = 12, SY_FIN = 16, SY_N0 = 20, SY_N1 = 24, SY_REQ = 28 };  // byte offsets in RO_SYNC
#define ROWS_LIF 26u      // the key's lane field: (lane + 1) << 26, so that byte 3 of a key = 4 x candidates consumed
// step-log word: bits 0..9 state left, bit 10 done, and the candidates the step consumed -- 1..8 in bits 26..29 (what the
// chain loop writes: its key, masked) or, with bit 31 set, any count in bits 11..30 (the exact path: a step can run through
// many rejected candidates)
#define ROWS_LOG_KMASK 0x3c000000u
__device__ __forceinline__ uint32_t rows_log_word(uint32_t s, uint32_t done_bit, uint32_t k) { return s | done_bit | (k << 11) | 0x80000000u; }
__device__ __forceinline__ uint32_t rows_log_k(uint32_t e) { return (e >> 31) ? ((e >> 11) & 0xfffffu) : (e >> 26); }
#define ROWS_SPIN_LIMIT (1u << 22)  // polls (with s_sleep) before a hand-off wait gives up: ~2 s, never reached unless the protocol is broken
#define RO_WIN 1472u     // n_slots x 8 window entries, then cons[n_slots] u32, land[n_slots] u32, claim[n_slots] u8

typedef __attribute__((address_space(3))) volatile uint32_t ldsv_u32;
typedef __attribute__((address_space(3))) volatile scan_u32x2 ldsv_u32x2;
typedef __attribute__((address_space(3))) volatile double ldsv_f64;
typedef uint32_t scan_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) volatile scan_u32x4 ldsv_u32x4;
#define LV32(a) (*(ldsv_u32 *)(a))
#define LV64(a) (*(ldsv_u32x2 *)(a))
#define LV128(a) (*(ldsv_u32x4 *)(a))
typedef __attribute__((address_space(3))) volatile uint8_t ldsv_u8;
typedef __attribute__((address_space(3))) volatile uint16_t ldsv_u16;
#define LV16(a) (*(ldsv_u16 *)(a))
__device__ __forceinline__ void lds_w16(uint32_t a, uint32_t v) { asm volatile("ds_write_b16 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ uint32_t lds_r16(uint32_t a) {
    uint32_t v;
    asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}
#define LV8(a) (*(ldsv_u8 *)(a))
// entries a head-aligned window row holds (the non-empty ones come first)
__device__ __forceinline__ uint32_t rows_held(scan_u32x4 h0, scan_u32x4 h1) {
    return (h0.x != 0u) + (h0.y != 0u) + (h0.z != 0u) + (h0.w != 0u) + (h1.x != 0u) + (h1.y != 0u) + (h1.z != 0u) + (h1.w != 0u);
}

// the DMA areas of a workgroup's wavefronts, rounded so that the rollout regions behind them stay 1024-byte aligned (the draw ring's address is formed with an OR)
__host__ __device__ constexpr uint32_t rows_dma_total(uint32_t n_chain) { return (n_chain * 7680u + 1023u) & ~1023u; }
__host__ __device__ constexpr uint32_t rows_region_bytes(uint32_t n_slots) { return (RO_WIN + n_slots * 41u + 1023u) & ~1023u; }  // 8 KiB up to 164 states

// Per-wavefront landing area of the tick's global loads.  They are issued as LDS-DMA (global_load_lds_dword: no VGPR
// destination, lane l's dword lands at slot base + 4 l) from inline asm, so that the compiler neither sees a pending result
// it would have to wait for at the loop back-edge (a whole HBM round trip per tick, 46 % of the kernel when measured) nor
// orders later LDS reads behind them; the next tick opens with s_waitcnt vmcnt(0) -- long satisfied -- and reads the slots.
// (the two request areas hold FOUR dwords per lane each, lane l's at base + 16 l: one global_load_lds_dwordx4 fills an area)
enum { DS_RQA = 0 /* four slots: digests 0..3 of every lane's request */, DS_RQB = 4 /* four slots: digests 4..7 */,
       DS_RQD = 8 /* the request each lane made: position | entries << 28 */, DS_RQS = 9 /* ... and its state */,
       DS_RQ_SET = 10 /* slots from one set of request areas to the other (ROWS_LAND_LAG 2): the second set is A, B, D, S again */,
       DS_LOC = 20, DS_GP = 21 /* four slots: the discount factor of every lane's step (and the table entry behind it) */, DS_RLO = 25, DS_RHI,
       DS_PROD /* two slots: 16 products per row */, DS_SLOTS = 29 };
// byte offsets of the four request areas of set 0 / 1 inside a wavefront's DMA area
struct RowsRq { uint32_t a, b, d, s; };
__device__ __forceinline__ RowsRq rows_rq(bool second) {
    const uint32_t o = second ? DS_RQ_SET * 256u : 0u;
    return RowsRq{o + DS_RQA * 256u, o + DS_RQB * 256u, o + DS_RQD * 256u, o + DS_RQS * 256u};
}
// Ticks between a request round and the landing of what it asked for.  1: the requests of tick t's steps land at the end of tick
// t + 1 -- one tick period minus the helper's ~1000 cycles, ~2.5 us at the round-4 loop.  2: two sets of request areas used in turn, a
// round lands two ticks later (~5.4 us).  Which one wins depends on how many requests are in flight: with a request for
// every entry a window lacks (rq_minroom 1) 8 % of the loads on the XCDs with the longer memory latency missed a one-tick landing at
// 4096 rollouts -- a top-up that is late is a top-up lost, 50 % more dry windows there, and the kernel ends with its slowest workgroup
// -- and lag 2 was the faster kernel (1.02 s against 1.04); with no request for a single entry (rq_minroom 2: a third fewer requests)
// 0.3 % are late at lag 1, on every XCD alike, and the windows are fuller when they are looked at: 0.930 s against 0.975 s.
#ifndef ROWS_LAND_LAG
#define ROWS_LAND_LAG 1
#endif
// entries a window must lack before it is topped up: a kernel argument (rq_minroom), chosen by the launcher from the load (below)
#ifndef ROWS_RQ_MAX
#define ROWS_RQ_MAX 8u  // entries a top-up asks for at most (4: the second request area stays unused)
#endif
#define ROWS_DMA_BYTES 7680u  // per wavefront: DS_SLOTS x 256 B, rounded to a multiple of 1024 (the rollout regions behind it stay 1024-byte aligned:
                              // the draw ring's address is formed with an OR)

// A window entry is never ROWS_EMPTY (the entries a window holds are counted: that is the chain's only cursor): a digest at or
// below the bias -- a candidate that is never a CLEAR accept -- becomes ROWS_NEVER, which no draw is <= (a ring entry's low
// eleven bits are all set, a payload's never are) and which still lies within the exact look's band of every draw it could accept.
#define ROWS_NEVER 0x200u
__device__ __forceinline__ uint32_t rows_bias(uint32_t dig, uint32_t bias) { return dig > bias ? dig - bias : ROWS_NEVER; }
// Two layouts of the 32-bit digest a stream position holds (offsim_streams.format):
//   A  [T21 | done | z_next 10]                          threshold to 21 bits; the local row is the 16-bit loc entry (segments <= 65536 rows)
//   B  [T16 | hi 6..2 | done | hi 1..0 | z_next 8]       threshold to 16 bits; hi = bits 16..22 of the local row, the loc entry its low 16
//                                                        bits (segments of up to 2^23 rows, <= 256 states): still 4 + 2 bytes per position
//   C  [T14 | hi 8..2 | done | hi 1..0 | z_next 8]       threshold to 14 bits; hi = bits 8..16 of the local row, the loc entry ONE byte
//                                                        (segments of up to 2^17 rows, <= 255 states): 4 + 1 bytes per position
// Everything that differs between them is a constant of the launch: where the threshold starts, which low bits travel with a key, the
// bias and the exact-look band (in threshold units: 16 and 17 of T21, 2 and 3 of T16 / T14), how a draw is laid down in the ring,
// how many bits of a local row the loc stream holds.
struct RowsFormat {
    uint32_t tshift, paymask, zmask, smask, bias, amb, emask, locbits;
};
__host__ __device__ constexpr RowsFormat rows_format(int fmt) {  // (the format is a template parameter of the kernel: these are immediates)
    // (round 5: formats B and C carry NO bias and a band of one unit.  "draw <= entry" on the untouched digest is already a clear accept --
    // the draw's top bits are below the threshold's, its low bits, all ones, only an upper bound -- and the one case that needs the exact
    // look is equal top bits; the two extra units of round 4's band sent three times as many looks to the compiled exact path: 9.0 k
    // per rollout on C4's shard at 14-bit thresholds.  Format A keeps its soaked 16 / 17 units of T21: 8e-6 of the looks either way.)
    // With no bias a digest's PAYLOAD (its bits below the threshold) must never be all ones: a draw with equal top bits is laid down as
    // top | paymask, and against such a digest the difference would be 0 -- "clear accept" -- where the exact look has to decide.  The
    // payload's low byte is the next state: formats B and C are refused for more than 255 states (next state <= 254) by
    // offsim_compile_digests, offsim_shuffle_queues_keys and offsim_eval_mc_streams.
    return fmt == OFFSIM_STREAMS_B   ? RowsFormat{16u, 0xffffu, 0x4ffu, 0xffu, 0u, 0u - (1u << 16), ROWS_LOG_KMASK | 0xfb00u, 16u}
           : fmt == OFFSIM_STREAMS_C ? RowsFormat{18u, 0x3ffffu, 0x4ffu, 0xffu, 0u, 0u - (1u << 18), ROWS_LOG_KMASK | 0x3fb00u, 8u}
                                     : RowsFormat{11u, 0x7ffu, 0x7ffu, 0x3ffu, ROWS_BIAS, ROWS_AMB, ROWS_LOG_KMASK, 16u};
}
// the upper bits of the local row out of a format-B / C PAYLOAD (a digest or key masked with the format's paymask): bits 8, 9 and 11..
__device__ __forceinline__ uint32_t rows_loc_hi(uint32_t pay) { return ((pay >> 8) & 3u) | (((pay >> 11) & 0x7fu) << 2); }

// Cache policy of the reward pipeline's loads (local row of a served candidate, its reward; one dword out of a sector that is not
// looked at again).  Which one is right depends on the top-up regime: with a two-tick landing and a request for every
// missing entry, " nt" took the kernel from 1.022 to 0.990 s (with the default policy these 128 B per accepted step pushed the digest
// sectors out of L2 on the XCDs with the longer memory latency: 477 cycles per iteration there against 453 on the others); with the
// one-tick landing and a third fewer requests of the final kernel the DEFAULT policy is the faster one, 0.932 -> 0.903 s (a
// non-temporal load takes longer to return, the helper's round gets longer, 13 k instead of 2 k top-ups per rollout arrive late).
// (nt on the digest requests themselves: 550 cycles per iteration.)
#ifndef ROWS_RW_CACHE
#define ROWS_RW_CACHE ""
#endif
__device__ __forceinline__ void lds_dma_dword(const void *gptr, uint32_t lds_dst_uniform) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %1, off" ROWS_RW_CACHE "\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gptr), "s"(lds_dst_uniform)
        : "memory");
}

#ifndef ROWS_RQ_CACHE
#define ROWS_RQ_CACHE ""  // cache policy of the digest requests (A/B builds: " nt", " sc1", ...)
#endif
// four consecutive dwords per lane: lane l's land at lds_dst_uniform + 16 l
__device__ __forceinline__ void lds_dma_x4(const void *gptr, uint32_t lds_dst_uniform) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off" ROWS_RQ_CACHE "\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gptr), "s"(lds_dst_uniform)
        : "memory");
}

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

// ---- the rocRAND provider (OFFSIM_STREAM_PHILOX): csrc/philox_dev.hpp ----
// the exact look's draw, by provider: index = draws of the stream before it
template <int RNG>
__device__ __forceinline__ uint64_t rows_exact_draw(const uint64_t *__restrict__ rng4, uint64_t index) {
    if constexpr (RNG == OFFSIM_STREAM_PHILOX) return offsim_philox_k(rng4[0], rng4[1] + index);
    else return rows_exact53(rng4, index + 1u);
}

// HELPER = false: one wavefront does everything for its four rollouts (also the TRACE build).
// HELPER = true : the workgroup has a second set of wavefronts, one per chain wavefront and (by the dispatch order of a
//   workgroup's waves) on the same SIMD: the helper owns the rejection stream (it fills the draw ring ahead of the chain) and
//   the whole reward pipeline (row index -> reward -> in-order discounted sums), which the chain feeds through the step log.
//   The chain keeps the look, the exact path and the window refill.  A single in-order wavefront retires an instruction per
//   ~7 cycles here, so a SIMD has room for both, and the chain's timeline loses ~45 % of its instructions.
//   Hand-off (all in the rollout's LDS region, RO_SYNC): the chain fills log buffer k & 1 during tick k, stores the step
//   count and its draw counter, waits for its LDS stores (s_waitcnt lgkmcnt(0)) and only then stores SY_TICK = k + 1; the
//   helper polls SY_TICK, reads the buffer, and stores SY_HTICK = k + 1 once its reads have returned (the chain does not
//   reuse that buffer before).  Draws: the helper writes ring entries, waits, then stores SY_GEN; the chain never looks
//   beyond the SY_GEN it has read, the helper never generates beyond SY_C + 240 of the 256 ring entries.  This relies on the
//   LDS executing the DS instructions of ONE wavefront in issue order (data before flag); every wait is bounded
//   (ROWS_SPIN_LIMIT) and ends the rollout with OFFSIM_ST_PROTOCOL instead of hanging the stream.
// RNG: the provider of the rejection stream (OFFSIM_STREAM_PCG64: NumPy's default_rng, the reference's numbers; OFFSIM_STREAM_PHILOX:
//   rocRAND's Philox4x32-10 device API).  Only the generation of ring entries, the exact look's draw and the stream state written back
//   differ: the ring holds the top 32 bits of the 53-bit draw either way.
template <bool TRACE, bool HELPER, int FMT, int RNG = OFFSIM_STREAM_PCG64>
__global__ void __launch_bounds__(HELPER ? 512 : 256)
    k_eval_mc_rows(offsim_table t, offsim_rollouts ro, offsim_streams sm, const uint64_t *__restrict__ keys, double gamma,
                   const double *__restrict__ gamma_pow, int64_t n_gamma_pow64, int64_t max_episodes64, offsim_evalmc_out out,
                   uint32_t seg_bytes, uint32_t region_bytes, uint32_t rq_minroom, uint32_t rows_used) {
    static_assert(!(TRACE && HELPER), "the TRACE build is the single-wavefront kernel");
    extern __shared__ __align__(16) unsigned char lds_raw[];
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_all = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t n_chain = HELPER ? (blockDim.x >> 7) : (blockDim.x >> 6);  // chain wavefronts of the workgroup
    const bool is_helper = HELPER && wave_all >= n_chain;
    const uint32_t wave = is_helper ? wave_all - n_chain : wave_all;  // the pair's index
    const uint32_t li = lane & 15u, rw = lane >> 4, li4 = li * 4u;
    const uint32_t n_slots = (uint32_t)t.n_slots;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_byte *)lds_raw;
    const uint32_t lds_pad = (0u - lds_base) & 1023u;
    const uint32_t seg_a = lds_base + lds_pad;  // seg_off copy, shared by the block
    for (uint32_t i = threadIdx.x; i <= n_slots; i += blockDim.x) LV32(seg_a + i * 4u) = t.seg_off[i];
    // rows_used (4, 2 or 1): the rows of a wavefront that carry a rollout.  A sparse launch leaves rows empty on purpose (the launcher,
    // offsim_eval_mc_streams): every event of a row -- an episode end, a row without a clear accept -- holds the whole wavefront, so
    // with CUs to spare two wavefronts of two rollouts are faster than one of four.  An empty row is a row that has stopped.
    const uint32_t rid = wave * 4u + rw;  // (the row's LDS region)
    const int64_t r = rw < rows_used ? ((int64_t)blockIdx.x * n_chain + wave) * rows_used + rw : (int64_t)ro.R;
    const uint32_t dma_a = (uint32_t)__builtin_amdgcn_readfirstlane((int)(seg_a + seg_bytes + wave * ROWS_DMA_BYTES));  // this pair's DMA slots
    const uint32_t rbase = seg_a + seg_bytes + rows_dma_total(n_chain) + rid * region_bytes;
    auto dma_slot = [&](uint32_t slot) __attribute__((always_inline)) -> uint32_t { return LV32(dma_a + slot * 256u + lane * 4u); };
#define ROWS_READ_A() LV128(dma_a + DS_RQA * 256u + lane * 16u)

    const uint32_t win_a = rbase + RO_WIN, cons_a = win_a + n_slots * 32u, land_a = cons_a + n_slots * 4u, claim_a = land_a + n_slots * 4u;
    const uint32_t sync_a = rbase + RO_SYNC;
    constexpr RowsFormat F = rows_format(FMT);
    constexpr bool fmt_b = FMT != OFFSIM_STREAMS_A;  // B, C: the digest carries the upper bits of the local row
    constexpr bool fmt_c = FMT == OFFSIM_STREAMS_C;
    uint32_t dead = r < (int64_t)ro.R ? 0u : 1u;  // 1: the row has stopped (or never ran)
    const int64_t rr = dead ? 0 : r;  // (rows past R read rollout 0's inputs and write nothing)
    if (!is_helper && li < 8u) LV32(sync_a + li4) = 0u;  // (without helpers SY_TICK stays 0: the dry-row handler takes the log buffer's parity from it)
    if (!is_helper) {  // no request has been made yet (the dry-row path of the loop looks at the descriptors of both sets)
        LV32(dma_a + DS_RQD * 256u + lane * 4u) = 0u;
        LV32(dma_a + (DS_RQ_SET + DS_RQD) * 256u + lane * 4u) = 0u;
    }

    // Lanes 8..15 of a row are exact duplicates of lanes 0..7 inside the chain (same window entry, same draw, same key, same
    // stores): the row minimum then needs only the three DPP steps that stay inside eight lanes, and nothing is predicated.
    const uint32_t li4w = (li & 7u) * 4u;
    const uint32_t ring_a = rbase + RO_RING;
    const uint32_t lifield = ((li & 7u) + 1u) << ROWS_LIF;

    __syncthreads();
    auto seg_at = [&](uint32_t s) __attribute__((always_inline)) -> uint32_t { return LV32(seg_a + s * 4u); };

    const uint32_t *dbase = sm.dig + rr * sm.dig_stride;
    constexpr uint32_t LOCB = F.locbits / 8u;  // bytes of a loc-stream entry
    const unsigned char *lbase = sm.loc ? (const unsigned char *)sm.loc + rr * sm.loc_stride * (int64_t)LOCB : nullptr;
    auto loc_at = [&](uint32_t idx) __attribute__((always_inline)) -> uint32_t {
        return LOCB == 1u ? (uint32_t)lbase[idx] : (uint32_t)((const uint16_t *)lbase)[idx];
    };
    const uint32_t *init_row = ro.init_perm ? ro.init_perm + rr * ro.init_stride : nullptr;
    uint32_t *cur_glb = ro.cursor + rr * n_slots;
    const uint64_t *rng4 = ro.rng + 4 * rr;
    const uint32_t N0 = (uint32_t)t.N0;
    const uint64_t n_gamma_pow = (uint64_t)n_gamma_pow64;
    const uint32_t max_episodes = (uint32_t)(max_episodes64 > 0x7fffffffll ? 0x7fffffffll : max_episodes64);
    const bool r64 = t.r_dtype == OFFSIM_F64;

    // ---- priming: every state's window holds the next 8 candidates of its queue ----
    if (!is_helper)
    for (uint32_t s = li; s < n_slots; s += 16u) {
        const uint32_t c0 = cur_glb[s], beg = seg_at(s), len = seg_at(s + 1u) - beg;
        const uint32_t left = len - c0, want = left < ROWS_W ? left : ROWS_W;
#pragma unroll
        for (uint32_t e = 0; e < ROWS_W; e++) LV32(win_a + s * 32u + e * 4u) = e < want ? rows_bias(dbase[beg + c0 + e], F.bias) : ROWS_EMPTY;
        LV32(cons_a + s * 4u) = c0;
        LV32(land_a + s * 4u) = c0 + want;
        LV8(claim_a + s) = 0u;
    }

    // ---- rejection stream: lane j of the row owns draws j, j + 16, ... (jump-ahead); ring of k21 << 11 | 0x7ff ----
    const bool owns_draws = HELPER ? is_helper : true;
    constexpr bool philox = RNG == OFFSIM_STREAM_PHILOX;
    U128 lane_state = u128(0, 0);
    U128 plus16 = u128(0, 0);
    // Philox: the rng row is (seed, draws consumed so far, 0, 0); lanes 0..7 of a row each own one block = two draws per round of 16
    // (lanes 8..15 repeat them: one instruction stream for the wavefront)
    const uint64_t ph_seed = philox ? rng4[0] : 0ull, ph_c0 = philox ? rng4[1] : 0ull;
    if (owns_draws && !philox) {
        const U128 base = u128(rng4[0], rng4[1]), inc = u128(rng4[2], rng4[3]);
        plus16 = pcg_jump(inc, 16).plus;
        lane_state = pcg_apply(pcg_jump(inc, (uint64_t)li + 1), base);  // yields draw li
    }
    const U128 mult16 = u128(0xb6a4239f3b315f84ull, 0xf6ef6d3d288c03c1ull);  // PCG multiplier ** 16 mod 2**128
    uint32_t gen = 0, c = 0;  // draws generated (HELPER chain: known to be generated) / consumed since kernel start
    auto gen16 = [&]() __attribute__((always_inline)) {
        if constexpr (philox) {
            // draws gen + 2 j and gen + 2 j + 1 of this launch (j = lane & 7) are draws ph_c0 + ... of the stream; an odd ph_c0 shifts
            // the pairing by one: the block of stream draw d is d >> 1, so every lane works out the two blocks its two draws lie in
            // only when the stream position is odd (a rollout that was stepped before: rare) -- the even case is one block
            const uint64_t d0 = ph_c0 + (uint64_t)gen + 2u * (li & 7u);
            uint64_t k0, k1;
            if ((ph_c0 & 1ull) == 0ull) {
                offsim_philox_pair(ph_seed, d0 >> 1, k0, k1);
            } else {
                k0 = offsim_philox_k(ph_seed, d0);
                k1 = offsim_philox_k(ph_seed, d0 + 1ull);
            }
            const uint32_t low = (1u << F.tshift) - 1u;
            const scan_u32x2 pr = {(uint32_t)(k0 >> 21) | low, (uint32_t)(k1 >> 21) | low};
            LV64(rbase + RO_RING + (((gen + 2u * (li & 7u)) & (ROWS_RING - 1u)) << 2)) = pr;  // (gen is a multiple of 16: the pair never wraps)
        } else {
            const uint32_t top = (uint32_t)(pcg_output(lane_state) >> 32);  // (the draw's top bits down to the threshold's resolution, ones below)
            LV32(rbase + RO_RING + (((gen + li) & (ROWS_RING - 1u)) << 2)) = top | ((1u << F.tshift) - 1u);
            lane_state = add128(mul128(mult16, lane_state), plus16);
        }
        gen += 16u;
    };
    if (owns_draws) {
#pragma unroll 1
        for (int i = 0; i < 15; i++) gen16();  // 240 draws ahead
    }
    // bounded wait of the hand-off protocol: polls `ready` until it holds; false if it gave up
    auto spin_until = [&](auto ready) __attribute__((always_inline)) -> bool {
        for (uint32_t n = 0; n < ROWS_SPIN_LIMIT; n++) {
            if (ready()) return true;
            __builtin_amdgcn_s_sleep(4);
        }
        return false;
    };
    // HELPER chain: at least m draws beyond c are in the ring (published by the helper), or the rollout stops
    int status = OFFSIM_ST_OK;
    uint32_t nlog_dead = 0;  // steps the row logged in the tick it stopped in
    auto need_draws = [&](uint32_t m, uint32_t logged) __attribute__((always_inline)) {
        if (HELPER) {
            if (gen - c >= m) return;
            LV32(sync_a + SY_C) = c;  // the helper generates up to 240 beyond what it knows to be consumed
            const bool ok = spin_until([&]() {
                gen = LV32(sync_a + SY_GEN);
                return gen - c >= m;
            });
            if (!ok) {
                status = OFFSIM_ST_PROTOCOL;
                dead = 1u;
                nlog_dead = logged;
            }
        } else {
            while (gen - c < m) gen16();
        }
    };

    // ---- initial states (psrs.py:22-23, 32-37): ring of 32 entries of the shuffled init queue, filled 16 at a time ----
    // The two dependent loads behind an initial state (queue entry -> row -> slot) never stall a tick: the next 16 entries
    // are fetched into registers one load per tick (pf_st 1: rows requested, 2: slots requested) and stored as soon as the
    // half of the ring they go to has been consumed; a rollout that runs into the end of what is stored (16 episode ends
    // within three ticks) finishes the fetch on the spot.
    const uint32_t ic0 = ro.init_cursor[rr], init_a = rbase + RO_INIT;
    uint32_t ic = ic0, filled = ic0, ep = 0, pf_st = 0;
    int pf_v = -1;
    auto prefetch_rows = [&]() __attribute__((always_inline)) {
        const uint32_t k = filled + li;
        pf_v = k < N0 ? (init_row ? (int)init_row[k] : (int)k) : -1;
        pf_st = 1u;
    };
    auto prefetch_step = [&]() __attribute__((always_inline)) {
        if (pf_st == 2u) {
            if ((int32_t)(filled - ic) <= 16) {  // the half these 16 go to has been consumed
                LV32(init_a + (((filled + li - ic0) & 31u) << 2)) = (uint32_t)pf_v;
                filled += 16u;
                prefetch_rows();
            }
        } else if (pf_st == 1u) {
            pf_v = pf_v >= 0 ? t.init_slot[pf_v] : -1;
            pf_st = 2u;
        }
    };
    if (!is_helper) {
        prefetch_rows();
        prefetch_step();
        prefetch_step();  // entries [ic0, ic0 + 16) are stored, the rows of the next 16 requested
    }

#ifdef OFFSIM_ROWS_PROF
    uint64_t pf_fast = 0, pf_slow = 0, pf_tick = 0, pf_nslow = 0, pf_t0 = 0, pf_ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pf_t1 = 0;
#define PF_START() pf_t0 = __builtin_amdgcn_s_memtime()
#define PF_ADD(x) x += __builtin_amdgcn_s_memtime() - pf_t0
#define PF_PH(k) { const uint64_t _n = __builtin_amdgcn_s_memtime(); pf_ph[k] += _n - pf_t1; pf_t1 = _n; }
#define PS_T1() pf_t1 = __builtin_amdgcn_s_memtime()  /* stamps of the chain's slow iterations: chain slots 3-6, 8, 9 (the helper's own use of them is in its own registers) */
#define PS(k) PF_PH(k)
#else
#define PF_PH(k)
#define PF_START()
#define PF_ADD(x)
#define PS_T1()
#define PS(k)
#endif
    // ---- per-row state ----
    uint32_t log_a = rbase + RO_LOG;  // the step log of the current tick (HELPER: alternates between RO_LOG and RO_LOG2)
    // The chain keeps the tick's step log in a REGISTER, lane = step (round 4): a step's word is merged into its lane under a lane mask
    // (one v_cndmask in the hand-scheduled loop instead of a 16-cycle LDS store per step), the tick hands the sixteen words over with one
    // store.  last_log: the word of the most recent step that was logged outside the loop (the loop's first copy writes the log of the
    // step before it from (key, state left) of that step; handed last_log in both, it rewrites the same word).
    uint32_t elog = 0, last_log = 0;
    auto log_step = [&](uint32_t it, uint32_t word) __attribute__((always_inline)) {
        elog = li == it ? word : elog;
        last_log = word;
    };
    uint32_t logh_a = rbase + RO_LOGH;  // ... and its side bytes (format B: local-row high bits of the steps logged in the long form)
    uint32_t z = 0;  // current state slot
    uint32_t n_dry = 0, n_tie = 0, n_late = 0, n_miss = 0, n_req = 0;
    // refill: one outstanding request per lane
    uint32_t rq_s = 0, rq_p = 0, rq_n = 0;
    // reward pipeline, three ticks deep (R1: row index + discount, R2: reward, R3: in-order sums)
    uint32_t lh1 = 0;
    uint32_t rowb1 = 0, half1 = 0, pop1 = 0, n1 = 0, n2 = 0, dm1 = 0, dm2 = 0, st1 = 0;
    uint64_t any1 = 0, any2 = 0;
    double gp2 = 0.0, gpx1 = 0.0;
    bool bx1 = false;
    uint32_t tt_chain = 0, steps = 0, ep_acc = 0, n_len = 0, len_acc = 0;
    double G = 0.0, sum_g = 0.0;

    // env.reset() (psrs.py:32-37, :249-252): the next initial state, or the rollout stops
    auto do_reset = [&](uint32_t logged) __attribute__((always_inline)) {
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
        while (ic >= filled) prefetch_step();  // (ic < N0: the fetch is never idle here; at most two rounds)
        z = LV32(init_a + (((ic - ic0) & 31u) << 2));
        ic++;
    };
    if (!dead && !is_helper) do_reset(0u);

    // chain registers: w = this lane's window entry of the current state (read from ra), kt = its draw
    uint32_t ra = 0, w = 0, kt = 0;
    auto issue_reads = [&]() __attribute__((always_inline)) {
        const uint32_t zz = dead ? 0u : z;
        ra = win_a + zz * 32u + li4w;
        w = LV32(ra);
        kt = LV32(ring_a + (((c << 2) + li4w) & (ROWS_RING * 4u - 4u)));
    };
    // entries the window of the row's current state holds (head-aligned: they come first), from the look's entries
    auto held = [&](uint32_t wv) __attribute__((always_inline)) -> uint32_t {
        const uint64_t b = __ballot(wv != ROWS_EMPTY);
        return (uint32_t)__popc((uint32_t)(b >> (rw * 16u)) & 0xffu);
    };
    // the look of the hand-scheduled loop, restated for the iterations that run outside it (some row of the wavefront has
    // stopped, or TRACE): key as there, amb = some lane of the row needs the exact look
    auto look = [&](uint32_t &key, bool &amb) __attribute__((always_inline)) {
        const uint32_t d = w - kt;
        key = row_min16(kt > w ? 0xffffffffu : ((w & F.paymask) | lifield));
        amb = row_min16(d >= F.amb ? 0u : 1u) == 0u;
    };

    // the step's bookkeeping for a clear accept of window entry k1-1 with payload `key`
    auto commit = [&](uint32_t key, uint32_t k1, uint32_t it) __attribute__((always_inline)) {
        c += k1;
        log_step(it, rows_log_word(z, key & 0x400u, k1));
        if (fmt_b) lds_w16(logh_a + it * 2u, rows_loc_hi(key));
        const uint32_t k1x4 = k1 << 2;
        LV32(((li4w - k1x4) & 28u) | (ra - li4w)) = li4w < k1x4 ? ROWS_EMPTY : w;  // (lanes 8..15 repeat the stores of lanes 0..7)
        z = key & F.smask;
    };

    // exact path: candidates of state z straight from the stream, starting at queue position cz (`popped` candidates of the
    // step are consumed already), until one is accepted (completes the step: log, window = the candidates behind it, land) or
    // the queue ends (the rollout stops)
    auto direct = [&](uint32_t it, uint32_t cz, uint32_t popped) __attribute__((always_inline)) {
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
                if (li < 8u) LV32(win_a + z * 32u + li4) = ROWS_EMPTY;  // everything the window held is consumed: cursor = land
                LV32(land_a + z * 4u) = cz;
                return;
            }
            need_draws(16u, it);
            if (dead) return;
            // candidates up to the end of the 64-byte sector the head lies in: the window's last top-up came out of that sector, so it
            // is still in L2 / the Infinity Cache (a read of 16 would nearly always reach into the next one: an HBM round trip for
            // candidates the step rarely gets to); the batches behind an all-rejected one are whole sectors
            const uint32_t in_sector = 16u - ((uint32_t)((uintptr_t)(dbase + beg + cz) >> 2) & 15u);
            const uint32_t nv = rem < in_sector ? rem : in_sector;
            const bool valid = li < nv;
            const uint32_t dg = valid ? dbase[beg + cz + li] : 0u;
            PS(5);
#ifdef OFFSIM_ROWS_PROF
            asm volatile("s_waitcnt vmcnt(0)" ::"v"(dg) : "memory");
#endif
            PS(6);
            const uint32_t k21 = LV32(rbase + RO_RING + (((c + li) & (ROWS_RING - 1u)) << 2)) >> F.tshift;
            bool ok = valid && k21 <= (dg >> F.tshift);
            if (ok && k21 == (dg >> F.tshift)) {  // tie at the digest's resolution: k53 of draw c+li against the full T
                const uint32_t lc = lbase ? (loc_at(beg + cz + li) | (fmt_b ? rows_loc_hi(dg & F.paymask) << F.locbits : 0u)) : cz + li;
                ok = !(rows_exact_draw<RNG>(rng4, (uint64_t)c + li) > key_T(keys[beg + lc]));
            }
            const uint32_t fk = row_min16(ok ? ((li << 20) | (dg & F.paymask)) : 0xffffffffu);  // first accepted lane and its digest's payload
            if (fk == 0xffffffffu) {  // all of them rejected: consumed (one draw each)
                c += nv;
                cz += nv;
                popped += nv;
                continue;
            }
            const uint32_t acc = fk & 0xfffffu;
            const uint32_t k1 = (fk >> 20) + 1u;
            c += k1;
            const uint32_t cz1 = cz + k1;
            log_step(it, rows_log_word(z, acc & 0x400u, popped + k1));
            if (fmt_b) lds_w16(logh_a + it * 2u, rows_loc_hi(acc));
            const uint32_t keep = nv - k1 < ROWS_W ? nv - k1 : ROWS_W;  // the candidates behind it become the window
            if (li < 8u) LV32(win_a + z * 32u + li4) = ROWS_EMPTY;
            if (li >= k1 && li < k1 + keep) LV32(win_a + z * 32u + ((li - k1) << 2)) = rows_bias(dg, F.bias);
            LV32(land_a + z * 4u) = cz1 + keep;
            z = acc & F.smask;
            if (acc & 0x400u) {
                ep++;
                do_reset(it + 1u);
            }
            return;
        }
    };

    // the look did not give the row a clear accept.  A lane near a tie: the exact look starts at the head of the queue, which is
    // land - (entries held); what the window rejected is rejected again, with the same draws.  No lane near a tie: every
    // candidate the window holds is a clear reject (or it holds none), they are consumed and the exact look starts behind them.
    auto exact_step = [&](uint32_t it, bool amb) __attribute__((always_inline)) {
        const uint32_t hv = held(w), ld = LV32(land_a + z * 4u);
        PS(4);
        if (amb) {
            n_tie++;
            direct(it, ld - hv, 0u);
        } else {
            n_dry++;
            c += hv;
            direct(it, ld, hv);
        }
    };

    // one iteration of a live row outside the hand-scheduled loop
    auto slow_step = [&](uint32_t key, bool amb, uint32_t it) __attribute__((always_inline)) {
        if (key != 0xffffffffu && !amb) {  // clear accept: the row's event, if any, is the episode end or low draws
            commit(key, (key >> ROWS_LIF) & 15u, it);
            if (key & 0x400u) {
                ep++;
                do_reset(it + 1u);
            }
        } else {
            exact_step(it, amb);
        }
        if (!dead) need_draws((ROWS_TICK - (it + 1u)) * 8u + 8u, it + 1u);  // enough for the rest of the tick (looks of <= 8)
    };

    // ---- reward pipeline, three ticks deep so that no tick waits on HBM; lane = step of the tick.  R1 (tick k): request the
    // local row (loc stream) and the discount factor of the steps of tick k; R2 (k+1): request their rewards; R3 (k+2): the
    // in-order discounted sums (bit-exact Gs).  Runs in the chain wavefront's tick, or in the helper wavefront (HELPER).
    uint32_t in_loc = 0, in_gplo = 0, in_gphi = 0;
    // first half: what the previous tick's loads brought, and R3 (no new load is issued here)
    auto rewards_a = [&]() __attribute__((always_inline)) {
        in_loc = dma_slot(DS_LOC);
        {
            const scan_u32x4 gpq = LV128(dma_a + DS_GP * 256u + lane * 16u);
            in_gplo = gpq.x;
            in_gphi = gpq.y;
        }
        const uint32_t in_rlo = dma_slot(DS_RLO), in_rhi = dma_slot(DS_RHI);
        PF_PH(3);
        // R3: in-order discounted-return accumulation (psrs.py:262-269) for the steps of two ticks ago: the products are
        // broadcast through LDS, every lane of the row runs the same sequential sum
        {
            double rv = 0.0;
            if (li < n2) rv = r64 ? __hiloint2double((int)in_rhi, (int)in_rlo) : (double)__uint_as_float(in_rlo);
            const double prod = li < n2 ? gp2 * rv : 0.0;  // product first, then the running sum in step order (+0.0 changes nothing)
            const uint32_t pa = dma_a + DS_PROD * 256u + rw * 128u;
            *(ldsv_f64 *)(pa + li * 8u) = prod;
            double p[16];
#pragma unroll
            for (int i = 0; i < 16; i += 2) {  // (two products per LDS read)
                const scan_u32x4 q = LV128(pa + (uint32_t)i * 8u);
                p[i] = __hiloint2double((int)q.y, (int)q.x);
                p[i + 1] = __hiloint2double((int)q.w, (int)q.z);
            }
            int32_t base_len = (int32_t)len_acc;  // length of the open episode minus the steps of this tick already counted
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (((any2 >> (4 * q)) & 0xfull) == 0ull) {  // (wave-uniform) no row ends an episode in these four steps
#pragma unroll
                    for (int i = 4 * q; i < 4 * q + 4; i++) G = G + p[i];
                } else {
#pragma unroll
                    for (int i = 4 * q; i < 4 * q + 4; i++) {
                        G = G + p[i];
                        if ((dm2 >> i) & 1u) {  // this row ends an episode at step i (psrs.py:265-269)
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
        PF_PH(4);
    };
    // Queue position of the candidate every step of the tick accepted, lane = step: the cursor of the state the step left, plus
    // the candidates the tick's EARLIER steps consumed in that state, plus its own, minus one -- and the cursors move on.  One LDS
    // atomic does all of it: the LDS applies the lanes of a ds_add_rtn_u32 that hit one address in ascending lane order (= step
    // order), so a lane gets back exactly the cursor as its step found it (tools/micro/lds_atomic_order.hip checks this on the
    // hardware; offsim_selftest_lds_atomic_order is the same check behind the C ABI).
    auto positions = [&](uint32_t n, uint32_t e) __attribute__((always_inline)) -> uint32_t {
        uint32_t pos = 0;
        if (li < n) {
            const uint32_t ca = cons_a + (e & F.smask) * 4u, k_i = rows_log_k(e);
            uint32_t pre;
            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(pre) : "v"(ca), "v"(k_i) : "memory");
            pos = pre + k_i - 1u;
        }
        return pos;
    };
    // second half: R2 and R1 issue this tick's loads
    auto rewards_b = [&](uint32_t n, uint32_t e_i, uint32_t pos_i, uint32_t lochi_i) __attribute__((always_inline)) {
        const bool mine = li < n;
        const uint32_t s_i = e_i & F.smask, pop_i = rows_log_k(e_i);
        const bool done_i = mine && (e_i & 0x400u);
        // R2: rewards of the steps of one tick ago (row = segment start + local row, the latter from the loc stream)
        {
            // (Every load of the pipeline is issued by every lane -- one without work reads the rollout's own stream state -- so
            // that a round issues a FIXED number of vector-memory instructions: the helper's waits count them, ROWS_VM_*.)
            const bool act = li < n1;
            const uint32_t lc = lbase ? (((in_loc >> half1) & ((1u << F.locbits) - 1u)) | (lh1 << F.locbits)) : in_loc;
            const uint32_t g = rowb1 + lc;
            if (r64) {
                const uint32_t *src = act ? (const uint32_t *)((const double *)t.r + g) : (const uint32_t *)rng4;
                lds_dma_dword(src, dma_a + DS_RLO * 256u);
                lds_dma_dword(src + 1, dma_a + DS_RHI * 256u);
            } else {
                lds_dma_dword(act ? (const void *)((const float *)t.r + g) : (const void *)rng4, dma_a + DS_RLO * 256u);
            }
            if (TRACE && act) {
                const uint32_t st = st1 + li;
                if (out.trace_row && (int64_t)st < out.trace_cap) out.trace_row[r * out.trace_cap + st] = t.orig_idx[g];
                if (out.trace_pop && (int64_t)st < out.trace_cap) out.trace_pop[r * out.trace_cap + st] = pop1;
            }
            gp2 = bx1 ? gpx1 : __hiloint2double((int)in_gphi, (int)in_gplo);
            dm2 = dm1;
            any2 = any1;
            n2 = n1;
        }
        PF_PH(5);
        // R1: local row (the aligned dword of the 16-bit loc stream that holds it) and discount factor of this tick's steps
        {
            const uint64_t bal = __ballot(done_i);
            const uint32_t dmrow = (uint32_t)(bal >> (rw * 16u)) & 0xffffu;  // episode ends of this row's tick
            uint32_t hf = 0;
            const uint32_t rb = mine ? seg_at(s_i) : 0u;
            if (lbase) {
                const uint64_t a16 = mine ? (uint64_t)(uintptr_t)(lbase + (uint64_t)(rb + pos_i) * LOCB) : (uint64_t)(uintptr_t)rng4;
                hf = ((uint32_t)a16 & 3u) * 8u;  // where the entry starts inside its dword, in bits
                lds_dma_dword((const void *)(uintptr_t)(a16 & ~3ull), dma_a + DS_LOC * 256u);
            } else if (mine) {
                LV32(dma_a + DS_LOC * 256u + lane * 4u) = pos_i;  // table order: the local row is the queue position
            }
            const uint32_t below = dmrow & ((1u << li) - 1u);  // episode ends earlier in this tick
            const uint32_t t_i = below ? li - 1u - (31u - (uint32_t)__clz((int)below)) : tt_chain + li;
            const bool in_table = mine && (uint64_t)t_i + 1u < n_gamma_pow;  // (one 16-byte load: the entry and the one behind it)
            lds_dma_x4(in_table ? (const void *)(gamma_pow + t_i) : (const void *)rng4, dma_a + DS_GP * 256u);
            bx1 = mine && !in_table;  // the table's last entry or beyond it (csrc/discount.hpp): worked out here, used in place of the slot
            if (bx1) gpx1 = discount_at(gamma_pow, n_gamma_pow, gamma, (uint64_t)t_i);
            rowb1 = rb;
            lh1 = fmt_b ? lochi_i : 0u;
            half1 = hf;
            pop1 = pop_i;
            dm1 = dmrow;
            any1 = (bal | (bal >> 16) | (bal >> 32) | (bal >> 48)) & 0xffffull;
            n1 = n;
            st1 = steps;
            tt_chain = dmrow ? n - 1u - (31u - (uint32_t)__clz((int)dmrow)) : tt_chain + n;
            steps += n;
        }
        PF_PH(6);
    };

    // A: one request per state left in the tick (the lane that holds the step tops the state up): the entries behind the
    // window's end that the window has room for NOW (it is counted as it stands: the chain may have been there again since the
    // step was logged), as one or two 16-byte LDS-DMA loads into this pair's request areas.  Returns what it asked for
    // (0 = nothing).  (The loads fetch whole groups of four; a request that would read beyond the table's last row is left to
    // the exact path.)
    // (rq: the set of request areas this round uses; rq_prev / odd: with two sets in turn (ROWS_LAND_LAG 2) the set of the round before,
    // whose requests have not landed yet, and the parity of this round)
    auto request = [&](bool mine, uint32_t s_i, uint32_t &q_s, uint32_t &q_p, uint32_t &q_n, const RowsRq rq, const RowsRq rq_prev, bool odd,
                       bool two_sets) __attribute__((always_inline)) {
        q_n = 0;
        const uint32_t sa = mine ? s_i : 0u;  // (a lane without a step looks at state 0 and asks for nothing)
        // The state's claim byte: the lane that tops the state up in this round.  With two sets the byte holds a lane per round parity
        // (low nibble: even rounds), so that this round finds the request the round BEFORE made for the state -- it lands a tick from
        // now, and a second request aimed at the same window end would be dropped when it lands in turn (9 % of all top-ups, measured).
        uint32_t cl_prev = 0;
        if (two_sets) {
            const uint32_t old = LV8(claim_a + sa);
            cl_prev = odd ? (old & 15u) : (old >> 4);
            if (mine) LV8(claim_a + s_i) = (uint8_t)(odd ? ((old & 15u) | (li << 4)) : ((old & 0xf0u) | li));
        } else if (mine) {
            LV8(claim_a + s_i) = (uint8_t)li;
        }
        // one batch of reads, one wait: who holds the state's claim, the window as it stands, its end, the segment (and what the lane
        // that held the claim in the round before asked for)
        uint32_t cl = LV8(claim_a + sa);
        const scan_u32x4 h0 = LV128(win_a + sa * 32u), h1 = LV128(win_a + sa * 32u + 16u);
        uint32_t ld = LV32(land_a + sa * 4u);
        const uint32_t beg = seg_at(sa), len = seg_at(sa + 1u) - beg;
        uint32_t pend = 0;
        if (two_sets) {
            cl = odd ? (cl >> 4) : (cl & 15u);
            const uint32_t pd = LV32(dma_a + rq_prev.d + (rw * 16u + cl_prev) * 4u), ps = LV32(dma_a + rq_prev.s + (rw * 16u + cl_prev) * 4u);
            pend = (ps == sa && (pd & 0xfffffffu) == ld) ? pd >> 28 : 0u;  // entries on their way to this window's end as it stands
        }
        const uint32_t held = rows_held(h0, h1) + pend;
        const uint32_t room = held < ROWS_W ? ROWS_W - held : 0u;
        ld += pend;
        const uint32_t left = len - ld;
        uint32_t want = room < left ? room : left;
        want = want < ROWS_RQ_MAX ? want : ROWS_RQ_MAX;
        // No top-up while the window lacks a single entry (rq_minroom 2, what the launcher passes when every CU is busy): a request costs a
        // whole 128-byte line of fabric traffic whatever it asks for, and one entry more or less rarely decides whether a window runs
        // dry; a third fewer requests is what lets a one-tick landing arrive in time (ROWS_LAND_LAG).  From three entries up the dry
        // rows win (lag 1: 0.975 s at 3, 1.07 s at 4, against 0.930 s).  With a quarter of the CUs idle or more, every entry is asked for.
        if (room < rq_minroom) want = 0u;
        // (every lane issues the loads -- one that asks for nothing reads the rollout's own stream state into its slot, which
        // nobody looks at: a round then issues a fixed number of vector-memory instructions, ROWS_VM_REQ)
        const bool ok = mine && cl == li && want && (int64_t)beg + ld + 8 <= t.N;
        const uint32_t *src = ok ? dbase + beg + ld : (const uint32_t *)rng4;
        lds_dma_x4(src, dma_a + rq.a);
        if (ROWS_RQ_MAX > 4u) lds_dma_x4(ok && want > 4u ? src + 4 : (const uint32_t *)rng4, dma_a + rq.b);
        if (ok) {
            q_s = s_i;
            q_p = ld;
            q_n = want;
        }
    };

    // ================================================ the helper wavefront (HELPER) ================================================
    if (is_helper) {
        // The helper runs AHEAD of its chain wavefront on their SIMD (issue priority).  The chain's period without an exact-path
        // stall (~5000 cycles) is barely longer than a helper round, and a helper that is still busy when a tick is published
        // asks for that tick's top-ups late: on the XCDs with the longer memory latency 8 % of them then missed the next tick
        // and the windows ran dry half again as often (measured; tools/clock_rows.py).  At priority the round is short enough
        // everywhere; it costs the chain ~25 cycles per iteration of issue slots, which the steadier top-ups more than return.
#ifndef ROWS_HELPER_PRIO
#define ROWS_HELPER_PRIO 2
#endif
        __builtin_amdgcn_s_setprio(ROWS_HELPER_PRIO);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        LV32(sync_a + SY_GEN) = gen;  // the first 240 draws are in the ring
        uint32_t fin = 0;
#ifdef OFFSIM_ROWS_PROF
        pf_t1 = __builtin_amdgcn_s_memtime();
#endif
        for (uint32_t k = 0;; k++) {
            // wait for the chain to publish tick k; meanwhile keep the ring topped up (the chain may run short inside a tick)
            uint32_t cp = 0;
            bool ok = false;
            for (uint32_t tries = 0; tries < ROWS_SPIN_LIMIT; tries++) {
                const uint32_t tk = LV32(sync_a + SY_TICK);
                if (__ballot(tk <= k) == 0ull) {  // (first of all: the tick's top-ups are waiting to be asked for)
                    ok = true;
                    break;
                }
                cp = LV32(sync_a + SY_C);
                if (gen - cp < 240u) {
                    while (gen - cp < 240u) gen16();
                    LV32(sync_a + SY_GEN) = gen;  // (behind the ring entries: the LDS runs one wavefront's DS instructions in issue order)
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (!ok) break;  // (the chain reports OFFSIM_ST_PROTOCOL when it is the one that gave up; here nothing more can be summed)
            PF_PH(8);
            // The request areas are free once the previous round's request loads have landed: they were that round's FIRST loads,
            // so at most the ones issued behind them -- the reward pipeline's: reward (one or two dwords), local row, discount
            // factor (one) -- may still be out.  (Vector-memory operations complete in issue order.)  Not waiting for those is
            // what keeps the top-ups of this tick from queueing behind a slow reward load of the last one.
            if (r64) {
                if (lbase) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            } else {
                if (lbase) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            }
            // The chain has read the RQ slots of the previous round (before it published this tick): mark them "not landed".
            // The helper never waits for the digests it requests; the chain lands what has arrived (it has, a tick later).
            const RowsRq rq = rows_rq(ROWS_LAND_LAG == 2 && (k & 1u));  // this round's set of request areas (the chain landed it before it published the tick)
            {
                // (first of all the set's descriptors go: the chain's dry-row path may look into the request areas at any time, and takes
                // staged digests only under a descriptor that is still the same behind the read of the data)
                LV32(dma_a + rq.d + lane * 4u) = 0u;
                const scan_u32x4 none = {ROWS_NOT_LANDED, ROWS_NOT_LANDED, ROWS_NOT_LANDED, ROWS_NOT_LANDED};
                LV128(dma_a + rq.a + lane * 16u) = none;
                LV128(dma_a + rq.b + lane * 16u) = none;
            }
            const uint32_t la = rbase + ((k & 1u) ? RO_LOG2 : RO_LOG);
            const uint32_t n = LV32(sync_a + ((k & 1u) ? SY_N1 : SY_N0));
            const uint32_t le = LV32(la + li4);
            // (format B: bits 16.. of the served candidate's local row -- in the log word itself, or beside it for a step logged in the long form)
            const uint32_t le_hi = !fmt_b ? 0u : (le >> 31) ? lds_r16(rbase + RO_LOGH + ((k & 1u) ? 32u : 0u) + li * 2u) : rows_loc_hi(le);
            fin = LV32(sync_a + SY_FIN);
            cp = LV32(sync_a + SY_C);
            LV32(sync_a + SY_HTICK) = k + 1u;  // (behind the reads of the buffer: the chain may reuse it)
            {   // the window top-ups this tick's steps call for, first of all (the chain lands them at the end of its next tick:
                // what has not arrived by then is lost)
                uint32_t q_s = 0, q_p = 0, q_n = 0;
                request(li < n, le & F.smask, q_s, q_p, q_n, rq, rows_rq(ROWS_LAND_LAG == 2 && !(k & 1u)), (k & 1u) != 0u, ROWS_LAND_LAG == 2 && k > 0u);  // (its LDS reads return before the first load is issued: the marks are in place)
                LV32(dma_a + rq.d + lane * 4u) = q_n ? (q_p | (q_n << 28)) : 0u;
                LV32(dma_a + rq.s + lane * 4u) = q_s;
            }
            LV32(sync_a + SY_REQ) = k + 1u;
            const uint32_t pos_i = positions(n, le);
            PF_PH(9);
            // the slots the reward pipeline reads were loaded by the previous round: everything but this round's request loads
            if (ROWS_RQ_MAX > 4u) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            rewards_a();
            rewards_b(n, le, pos_i, le_hi);
            if (gen - cp < 240u) {
                while (gen - cp < 240u) gen16();
                LV32(sync_a + SY_GEN) = gen;
            }
            PF_PH(10);
            if (__ballot(fin == 0u) == 0ull) break;  // every rollout of the wavefront has stopped: tick k was the last one
        }
        for (int dr = 0; dr < 2; dr++) {  // drain the pipeline (R2, R3 of the last ticks)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            rewards_a();
            rewards_b(0u, 0u, 0u, 0u);
        }
        if (fin == (uint32_t)OFFSIM_ST_EXHAUSTED + 1u) {  // psrs.py:265: the cut-short episode still logs its length
            if (li == 0u && r < (int64_t)ro.R && out.ep_len && (int64_t)n_len <= out.ep_cap) out.ep_len[r * (out.ep_cap + 1) + n_len] = (int32_t)len_acc;
            n_len++;
        }
        if (li == 0u && r < (int64_t)ro.R) {
            out.sum_g[r] = sum_g;
            out.n_ep[r] = ep_acc;
            out.n_len[r] = n_len;
#ifdef OFFSIM_ROWS_PROF
            if (out.dbg && out.ep_g && out.ep_cap >= 24)
                for (int k = 0; k < 12; k++) out.ep_g[r * out.ep_cap + 12 + k] = (double)pf_ph[k];
#endif
        }
        return;
    }

    // ---- the chain wavefront's tick, once per 16 iterations; lane = step of the tick ----
    uint32_t tick_k = 0;
    auto tick = [&]() __attribute__((always_inline)) {
#ifdef OFFSIM_ROWS_PROF
        pf_t1 = __builtin_amdgcn_s_memtime();
#endif
        const uint32_t n = dead ? nlog_dead : ROWS_TICK;
        nlog_dead = 0;
        scan_u32x4 in_a = {0u, 0u, 0u, 0u}, in_b = {0u, 0u, 0u, 0u};
        uint32_t v_ht = 0, v_gen = 0;
        uint32_t le = 0;
        if (HELPER) {
            // First what does not depend on the helper's request round -- the later that round is looked at, the more of its loads
            // have arrived (what has not by then is a top-up lost): the two counters the next tick needs, the initial-state ring.
            v_ht = LV32(sync_a + SY_HTICK);
            v_gen = LV32(sync_a + SY_GEN);
            const uint32_t v_req = LV32(sync_a + SY_REQ);  // (one batch of reads, one wait)
            if (!dead) {
                prefetch_step();
                // the buffer of the next tick was read by the helper two ticks ago, and the ring holds a tick's worth of draws
                // (format A: the tick's log lives in registers until the flush below, which overwrites the log of two ticks ago -- the helper
                // may be a whole tick behind; format B writes the side bytes of the NEXT tick's long-form steps as they happen, into the
                // half the helper reads for the tick before this one)
                const uint32_t want_h = fmt_b ? tick_k : (tick_k ? tick_k - 1u : 0u);
                if (__ballot(v_ht < want_h) != 0ull && !spin_until([&]() { return LV32(sync_a + SY_HTICK) >= want_h; })) {
                    status = OFFSIM_ST_PROTOCOL;
                    dead = 1u;
                }
                if (!dead) {
                    gen = v_gen;
                    need_draws(136u, 0u);
                }
            }
            PF_PH(7);
            // The flag of the helper's request round that lands now (it made the requests of tick tick_k - ROWS_LAND_LAG; with two sets
            // of request areas it used set tick_k & 1).  A flag that is not there yet is rare; only then a bounded wait.
            const RowsRq rq = rows_rq(ROWS_LAND_LAG == 2 && (tick_k & 1u));
            if (tick_k >= (uint32_t)ROWS_LAND_LAG) {
                const uint32_t want_r = tick_k + 1u - (uint32_t)ROWS_LAND_LAG;
                if (__ballot(v_req < want_r) != 0ull) {
                    if (!spin_until([&]() { return LV32(sync_a + SY_REQ) >= want_r; }) && !dead) {
                        status = OFFSIM_ST_PROTOCOL;
                        dead = 1u;
                    }
                }
            }
            // Landing and hand-off, hand-written (round 4; the compiled form of the same steps took ~1400 of the tick's 1950
            // cycles): lane = step of the round that lands.  One batch of reads (descriptor, state, the eight staged
            // digests), a second one (the state's window row and its end), the count of entries the head-aligned row holds, the
            // number k of staged digests that land -- the request was aimed at the window's end as it stands, its groups of four
            // have arrived, the row has room -- and k stores under the lane masks k > i; then the tick's log (lane = step), its
            // counters and the flag, behind the landing (one wavefront's DS instructions execute in issue order: data before flag).
            // Entries are stored biased like everywhere else: max(digest - bias, ROWS_NEVER) with the subtraction saturating (for a
            // digest within 0x200 of the bias this is ROWS_NEVER where rows_bias() keeps the difference: both are below every draw).
            {
                const uint32_t rqo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(dma_a + rq.a));
                // (wave-uniform by construction, but kept in VGPRs by the compiler: the scalar operands are read off lane 0)
                const uint32_t landon = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tick_k >= (uint32_t)ROWS_LAND_LAG ? 1u : 0u));
                const uint32_t na = sync_a + ((tick_k & 1u) ? SY_N1 : SY_N0);
                const uint32_t fin_v = (uint32_t)status + 1u, tk1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tick_k + 1u));
                asm volatile(
                    "s_mov_b64 s[34:35], exec\n\t"
                    "s_cmp_eq_u32 %[landon], 0\n\t"
                    "s_cbranch_scc1 5f\n\t"
                    "v_add_u32 v80, %[rqo], %[lane4]\n\t"
                    "v_lshl_add_u32 v83, %[lane4], 2, %[rqo]\n\t"
                    "ds_read_b32 v81, v80 offset:2048\n\t"                 // descriptor: position | entries << 28
                    "ds_read_b32 v82, v80 offset:2304\n\t"                 // ... and its state
                    "ds_read_b128 v[84:87], v83\n\t"                       // digests 0..3
                    "ds_read_b128 v[88:91], v83 offset:1024\n\t"           // digests 4..7
                    "s_waitcnt lgkmcnt(2)\n\t"
                    "v_lshrrev_b32 v92, 28, v81\n\t"                       // entries asked for
                    "v_and_b32 v81, 0xfffffff, v81\n\t"                    // the window end the request was aimed at
                    "v_cmp_ne_u32_e64 s[22:23], 0, v92\n\t"                // lanes with a request
                    "v_mov_b32 v104, 0\n\t"
                    "v_mov_b32 v105, 0\n\t"
                    "v_cndmask_b32_e64 v82, 0, v82, s[22:23]\n\t"          // (a lane without one looks at state 0)
                    "v_lshl_add_u32 v93, v82, 5, %[wina]\n\t"
                    "v_lshl_add_u32 v94, v82, 2, %[landa]\n\t"
                    "ds_read_b128 v[96:99], v93\n\t"                       // the window row as it stands
                    "ds_read_b128 v[100:103], v93 offset:16\n\t"
                    "ds_read_b32 v95, v94\n\t"                             // its end
                    "s_waitcnt lgkmcnt(3)\n\t"                             // the staged digests
                    "v_max3_u32 v106, v84, v85, v86\n\t"
                    "v_max3_u32 v107, v88, v89, v90\n\t"
                    "v_max_u32 v106, v106, v87\n\t"                        // all-ones: a group of four has not arrived
                    "v_max_u32 v107, v107, v91\n\t"
                    "v_sub_u32_e64 v84, v84, %[bias] clamp\n\t"
                    "v_sub_u32_e64 v85, v85, %[bias] clamp\n\t"
                    "v_sub_u32_e64 v86, v86, %[bias] clamp\n\t"
                    "v_sub_u32_e64 v87, v87, %[bias] clamp\n\t"
                    "v_sub_u32_e64 v88, v88, %[bias] clamp\n\t"
                    "v_sub_u32_e64 v89, v89, %[bias] clamp\n\t"
                    "v_sub_u32_e64 v90, v90, %[bias] clamp\n\t"
                    "v_sub_u32_e64 v91, v91, %[bias] clamp\n\t"
                    "v_max_u32 v84, 0x200, v84\n\t"
                    "v_max_u32 v85, 0x200, v85\n\t"
                    "v_max_u32 v86, 0x200, v86\n\t"
                    "v_max_u32 v87, 0x200, v87\n\t"
                    "v_max_u32 v88, 0x200, v88\n\t"
                    "v_max_u32 v89, 0x200, v89\n\t"
                    "v_max_u32 v90, 0x200, v90\n\t"
                    "v_max_u32 v91, 0x200, v91\n\t"
                    "v_cmp_ne_u32_e64 s[24:25], -1, v106\n\t"              // digests 0..3 have arrived
                    "v_cmp_ne_u32_e64 s[26:27], -1, v107\n\t"              // digests 4..7 have arrived
                    "s_waitcnt lgkmcnt(0)\n\t"
                    "v_min_u32 v96, 1, v96\n\t"                            // entries the row holds (head-aligned: they come first)
                    "v_min_u32 v97, 1, v97\n\t"
                    "v_min_u32 v98, 1, v98\n\t"
                    "v_min_u32 v99, 1, v99\n\t"
                    "v_min_u32 v100, 1, v100\n\t"
                    "v_min_u32 v101, 1, v101\n\t"
                    "v_min_u32 v102, 1, v102\n\t"
                    "v_min_u32 v103, 1, v103\n\t"
                    "v_add3_u32 v104, v96, v97, v98\n\t"
                    "v_add3_u32 v105, v99, v100, v101\n\t"
                    "v_add3_u32 v104, v104, v102, v103\n\t"
                    "v_add_u32 v104, v104, v105\n\t"
                    "v_cmp_eq_u32_e64 s[20:21], v81, v95\n\t"              // aimed at the end as it stands
                    "s_and_b64 s[28:29], s[24:25], s[26:27]\n\t"
                    "s_and_b64 s[20:21], s[20:21], s[22:23]\n\t"           // hit
                    "v_min_u32 v105, 4, v92\n\t"
                    "s_andn2_b64 s[30:31], s[22:23], s[20:21]\n\t"         // request that missed its window end
                    "v_cndmask_b32_e64 v105, 0, v105, s[24:25]\n\t"        // the first group, if it is there
                    "v_cndmask_b32_e64 v105, v105, v92, s[28:29]\n\t"      // everything, if both are
                    "v_sub_u32 v106, 8, v104\n\t"                          // room
                    "v_cndmask_b32_e64 v105, 0, v105, s[20:21]\n\t"
                    "v_lshl_add_u32 v93, v104, 2, v93\n\t"                 // the row's end
                    "v_min_u32 v105, v105, v106\n\t"                       // k: entries that land
                    "v_addc_co_u32_e64 %[nreq], s[36:37], 0, %[nreq], s[22:23]\n\t"
                    "v_addc_co_u32_e64 %[nmiss], s[36:37], 0, %[nmiss], s[30:31]\n\t"
                    "s_andn2_b64 s[30:31], s[20:21], s[24:25]\n\t"         // hit, but not arrived
                    "v_cmp_lt_u32_e32 vcc, 0, v105\n\t"
                    "v_addc_co_u32_e64 %[nlate], s[36:37], 0, %[nlate], s[30:31]\n\t"
                    "s_mov_b64 exec, vcc\n\t"
                    "v_add_u32 v95, v95, v105\n\t"
                    "ds_write_b32 v93, v84\n\t"
                    "ds_write_b32 v94, v95\n\t"                            // the window's new end
                    "v_cmp_lt_u32_e32 vcc, 1, v105\n\t"
                    "s_mov_b64 exec, vcc\n\t"
                    "ds_write_b32 v93, v85 offset:4\n\t"
                    "v_cmp_lt_u32_e32 vcc, 2, v105\n\t"
                    "s_mov_b64 exec, vcc\n\t"
                    "ds_write_b32 v93, v86 offset:8\n\t"
                    "v_cmp_lt_u32_e32 vcc, 3, v105\n\t"
                    "s_mov_b64 exec, vcc\n\t"
                    "ds_write_b32 v93, v87 offset:12\n\t"
                    "v_cmp_lt_u32_e32 vcc, 4, v105\n\t"
                    "s_cbranch_vccz 5f\n\t"
                    "s_mov_b64 exec, vcc\n\t"
                    "ds_write_b32 v93, v88 offset:16\n\t"
                    "v_cmp_lt_u32_e32 vcc, 5, v105\n\t"
                    "s_mov_b64 exec, vcc\n\t"
                    "ds_write_b32 v93, v89 offset:20\n\t"
                    "v_cmp_lt_u32_e32 vcc, 6, v105\n\t"
                    "s_mov_b64 exec, vcc\n\t"
                    "ds_write_b32 v93, v90 offset:24\n\t"
                    "v_cmp_lt_u32_e32 vcc, 7, v105\n\t"
                    "s_mov_b64 exec, vcc\n\t"
                    "ds_write_b32 v93, v91 offset:28\n\t"
                    "5:\n\t"
                    "s_mov_b64 exec, s[34:35]\n\t"
                    "ds_write_b32 %[loga], %[elog]\n\t"                     // the tick's sixteen log words, lane = step
                    "ds_write_b32 %[na], %[nv]\n\t"
                    "ds_write_b32 %[synca], %[cv] offset:8\n\t"            // SY_C
                    "v_cmp_ne_u32_e32 vcc, 0, %[dead]\n\t"
                    "v_mov_b32 v106, %[tk1]\n\t"
                    "s_mov_b64 exec, vcc\n\t"
                    "ds_write_b32 %[synca], %[fin] offset:16\n\t"          // SY_FIN: the rows that have stopped
                    "s_mov_b64 exec, s[34:35]\n\t"
                    "ds_write_b32 %[synca], v106\n\t"                       // SY_TICK
                    : [nreq] "+v"(n_req), [nmiss] "+v"(n_miss), [nlate] "+v"(n_late)
                    : [rqo] "s"(rqo), [lane4] "v"(lane * 4u), [wina] "v"(win_a), [landa] "v"(land_a), [loga] "v"(log_a + li4), [elog] "v"(elog),
                      [na] "v"(na), [nv] "v"(n), [synca] "v"(sync_a), [cv] "v"(c), [dead] "v"(dead), [fin] "v"(fin_v), [tk1] "s"(tk1),
                      [landon] "s"(landon), [bias] "s"(F.bias)
                    : "vcc", "scc", "memory", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s34", "s35", "s36",
                      "s37", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97",
                      "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");
                static_assert(SY_TICK == 0 && SY_C == 8 && SY_FIN == 16 && DS_RQD * 256 == 2048 && DS_RQS * 256 == 2304 && DS_RQB * 256 == 1024 && DS_RQA == 0,
                              "the immediates of the hand-written tick");
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the LDS-DMA loads of the previous tick (issued ~16 iterations ago)
            le = elog;  // (single wavefront: the log never leaves the registers)
            in_a = ROWS_READ_A(), in_b = LV128(dma_a + DS_RQB * 256u + lane * 16u);
        }
        PF_PH(0);
        if (!HELPER)
        {
        // C: land the requested digests.  Entries are appended only at the window's current end: whatever a direct read has
        // covered meanwhile is skipped, whatever does not fit -- or has not arrived -- is requested again later.  The window is
        // head-aligned, so its end is the number of entries it holds.
        uint32_t land_have = 0, land_ld = 0;
        if (rq_n) {
            const scan_u32x4 h0 = LV128(win_a + rq_s * 32u), h1 = LV128(win_a + rq_s * 32u + 16u);
            land_ld = LV32(land_a + rq_s * 4u);
            land_have = rows_held(h0, h1);
        }
        PF_PH(1);
        {
            // A request lands where it was aimed at -- the window's end; one that a direct read has overtaken meanwhile is dropped
            // (the state is topped up again when it is next left).  Of what was asked for, the groups of four that have arrived
            // land as far as the window has room; the other stores go to a scratch word.
            const bool hit = rq_n != 0u && rq_p == land_ld;
            n_req += rq_n != 0u;
            n_miss += rq_n != 0u && !hit;
            const bool got_a = in_a.x != ROWS_NOT_LANDED && in_a.y != ROWS_NOT_LANDED && in_a.z != ROWS_NOT_LANDED && in_a.w != ROWS_NOT_LANDED;
            uint32_t k = !hit ? 0u : !got_a ? 0u : rq_n < 4u ? rq_n : 4u;
            n_late += hit && !got_a;
            if (ROWS_RQ_MAX > 4u) {
                const bool got_b = in_b.x != ROWS_NOT_LANDED && in_b.y != ROWS_NOT_LANDED && in_b.z != ROWS_NOT_LANDED && in_b.w != ROWS_NOT_LANDED;
                k = (k == 4u && got_b) ? rq_n : k;
            }
            const uint32_t room = ROWS_W - land_have;
            k = k < room ? k : room;
            const uint32_t dst = win_a + rq_s * 32u + (land_have << 2);
            const uint32_t scratch = sync_a + 64u + li4;  // (the hand-off words end at 32)
            LV32(k > 0u ? dst : scratch) = rows_bias(in_a.x, F.bias);
            LV32(k > 1u ? dst + 4u : scratch) = rows_bias(in_a.y, F.bias);
            LV32(k > 2u ? dst + 8u : scratch) = rows_bias(in_a.z, F.bias);
            LV32(k > 3u ? dst + 12u : scratch) = rows_bias(in_a.w, F.bias);
            if (ROWS_RQ_MAX > 4u && __ballot(k > 4u) != 0ull) {
                LV32(k > 4u ? dst + 16u : scratch) = rows_bias(in_b.x, F.bias);
                LV32(k > 5u ? dst + 20u : scratch) = rows_bias(in_b.y, F.bias);
                LV32(k > 6u ? dst + 24u : scratch) = rows_bias(in_b.z, F.bias);
                LV32(k > 7u ? dst + 28u : scratch) = rows_bias(in_b.w, F.bias);
            }
            if (k) LV32(land_a + rq_s * 4u) = land_ld + k;
            rq_n = 0;
        }
        }
        PF_PH(2);
        if (HELPER) {
            steps += n;
            tick_k++;
            log_a = rbase + ((tick_k & 1u) ? RO_LOG2 : RO_LOG);
            logh_a = rbase + RO_LOGH + ((tick_k & 1u) ? 32u : 0u);
        } else {
            const uint32_t pos_i = positions(n, le);
            request(li < n, le & F.smask, rq_s, rq_p, rq_n, rows_rq(false), rows_rq(false), false, false);
            rewards_a();
            rewards_b(n, le, pos_i, !fmt_b ? 0u : (le >> 31) ? lds_r16(logh_a + li * 2u) : rows_loc_hi(le));
            if (!dead) {
                prefetch_step();
                while (gen - c < 240u) gen16();
            }
            PF_PH(7);
        }
    };

    // ---- the chain ----
    // One iteration = one accepted step of every live row.  fast_run() is the hand-scheduled loop: nothing but the
    // clear-accept path, left through one wave-uniform branch as soon as ANY lane of the wavefront sees something else; that
    // iteration is then redone row by row (rows without an event commit, the others take the exact path), and the loop is
    // entered again.  Rows that have stopped are masked out of the loop (exec), the others keep its pace; TRACE builds run every
    // iteration the second way.
    //
    // what the loop hands over when it is left for an event: the look's key, the lanes that need the exact look, and the
    // stores of the step as the loop had prepared them (valid for the rows whose look was a clear accept)
    uint32_t ex_key = 0, ex_d = 0, ex_slot = 0;
    uint64_t ex_amb = 0;
    auto fast_run = [&](uint32_t &it, uint64_t live) __attribute__((always_inline)) {
        // episode ends the loop may serve itself: after this many the episode cap, the end of the init queue or the end of
        // what is stored in the ring is reached and the C++ path has to look
        const uint32_t initp0 = init_a + (((ic - ic0) & 31u) << 2);
        uint32_t initp = initp0, left = filled - ic;
        {
            left = left < N0 - ic ? left : N0 - ic;  // (ic <= N0)
            const uint32_t cap = ep + 1u < max_episodes ? max_episodes - ep - 1u : 0u;
            left = left < cap ? left : cap;
        }
        uint32_t c4 = (c << 2) + li4w;  // draw counter, as the byte offset of this lane's ring slot
        uint32_t zz = z;
        uint32_t key = last_log, key2, d, tt, nrd, e, zn, rb, w2;
        uint64_t amb, ev;
        const uint32_t gen4 = gen << 2;  // draws known to be in the ring (x 4, like c4)
        // One copy of the step; the loop body is the sixteen steps of a tick, copy i logging step i, so there is neither a step
        // counter nor a back edge, and the loop is entered at copy `it` through a branch table.
        // What a step costs is the SUM of what its instructions hold the wavefront for (tools/micro/issue.hip: VALU / SALU 4.3
        // cycles, three-operand VALU 5.2, a DS instruction 16, a branch not taken ~7; tools/micro/step_loop.hip runs this very
        // loop in isolation and reproduces the sum to a few cycles): the step is issue-bound, so it is made of as few
        // instructions as the algorithm allows.  Round 4:
        //   * THREE DS instructions instead of four: the entry's read, the draw's read, the shifted row's store.  The log word
        //     goes into a register (elog, lane = step) under the lane mask of its step, in the DPP wait states of the NEXT copy
        //     (the state left and the key of a step survive until then: the register sets alternate);
        //   * the entry's read is issued as soon as the next state is known and the copy waits for the two reads only
        //     (lgkmcnt(1): the store behind them may still be on its way; one wavefront's DS instructions execute in issue
        //     order).  The read is therefore AHEAD of the row store: a row that stays in its state would see the row before the
        //     shift, so "same state" is one more event of the out-of-line path, which reads the entry again behind the store.
        //     One test covers it together with the episode end and the row without a clear accept: u = (next ^ this) - 1 is
        //     >= 0x3ff iff next == this or next carries the done bit / is the all-ones key, and < 0x100 otherwise (<= 256 states);
        //   * the wait states of the three DPP minima hold the exact-look test and the log of the previous step;
        //   * a row WITHOUT a clear accept (its window is dry, or all its entries were rejected: 2 % of the iterations) no longer
        //     sends the wavefront through the compiled exact path (~2800 cycles of divergent C++ per event): label 50 serves it
        //     here -- the held entries are consumed, the candidates up to the end of the head's 64-byte sector come straight from the stream, the same biased
        //     compare picks the first clear accept, the window becomes the candidates behind it -- and the loop is entered again
        //     at the next copy.  Whatever is not plain (a lane near a tie, the queue's last sixteen rows, draws running low,
        //     sixteen rejections in a row, no initial state left, stream format B) leaves for the C++ path with nothing committed.
        // Measured in isolation (step_loop.hip, alone / next to a busy partner wavefront): round 3's step 200 / 227 cycles, early
        // read 177 / 202, this one 162 / 185.
        // The chain keeps no cursor: a queue position is land - (entries held), worked out where it is needed (tick, exact
        // path).  The one branch of a step leaves for every event at once.
        // The state of a row (zz = state, ra = the address its entry was read from, w = the entry) is not copied at the end of a
        // step: even copies take it from (zz, ra, w) and leave the next one in (zn, rb, w2), odd copies the other way round, and
        // likewise the key (even copies: key, odd copies: key2); the entry code fills both sets, the exits put everything where
        // the C++ code expects it.  The first copy after the entry logs "the step before it" from what the entry code put into
        // its registers -- last_log, split so that the copy rebuilds exactly that word.
        // Registers private to the block (clobbered): v110 = this lane's window slot x 4, v111 = (slot + 1) << 26, v112..v116 =
        // the format's constants (payload mask, next-state mask, exact-look band, ring mask, log mask), v117..v127 temporaries.
#define ROWS_STEP(EPI, BACK, ZZ, RA, W, ZN, RN, WN, KEY, KEYP, LMSH)                                                          \
            BACK ":\n\t"                                                 /* (this look's entry and draw have arrived: waited for at the end of the copy before / at the entry / in ROWS_EPI) */ \
            "v_sub_co_u32 %[d], vcc, " W ", %[kt]\n\t"                   /* borrow: not a clear accept */                \
            "v_and_or_b32 " KEY ", " W ", v112, v111\n\t"                /* (lane + 1) << 26 | the digest's low bits: done << 10 | z_next (| local-row high bits) */ \
            "v_cmp_le_u32_e64 %[amb], v114, %[d]\n\t"                    /* the draw is above the entry by <= 17 units of T21: exact look */ \
            "v_cndmask_b32_e64 " KEY ", " KEY ", -1, vcc\n\t"            /* (gfx940+: two instructions between the VALU that wrote vcc and this read of it) */ \
            "v_and_or_b32 %[e], " KEYP ", v116, " ZN "\n\t"              /* wait state 1 of 2 ahead of the DPP read: log word of the PREVIOUS step: candidates consumed << 26 | done << 10 | state left (still in the set this copy is about to overwrite) */ \
            "s_lshl_b64 s[22:23], s[26:27], " LMSH "\n\t"                /* the lanes that hold the previous step's log word */ \
            "v_min_u32_dpp " KEY ", " KEY ", " KEY " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                  \
            "v_cndmask_b32_e64 %[elog], %[elog], %[e], s[22:23]\n\t"                                                      \
            "s_nop 0\n\t"                                                                                                 \
            "v_min_u32_dpp " KEY ", " KEY ", " KEY " quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"                  \
            "s_nop 1\n\t"                                                                                                 \
            "v_min_u32_dpp " KEY ", " KEY ", " KEY " row_half_mirror row_mask:0xf bank_mask:0xf\n\t"                      \
            "v_and_b32 " ZN ", v113, " KEY "\n\t"                        /* next state (| done << 10: an event) */      \
            "v_lshl_add_u32 " RN ", " ZN ", 5, v109\n\t"                 /* this lane's entry in the next state's row */ \
            "ds_read_b32 " WN ", " RN "\n\t"                             /* next look's entry (ahead of the store: wrong if it is this state -- an event) */ \
            "v_add_u32_sdwa %[c4], %[c4], " KEY " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t" /* + 4 x candidates consumed */ \
            "v_and_or_b32 %[nrd], %[c4], v115, %[ringa]\n\t"                                                              \
            "ds_read_b32 %[kt], %[nrd]\n\t"                              /* next look's draw */                          \
            "v_xad_u32 %[e], " ZN ", " ZZ ", -1\n\t"                      /* (next ^ this) - 1 */                         \
            "v_cmp_lt_u32_e64 %[ev], v115, %[e]\n\t"                     /* same state, episode end, or the all-ones key of a row without a clear accept */ \
            "v_sub_co_u32_sdwa %[tt], vcc, v110, " KEY " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t" /* slot (x4) of this lane's entry after the shift; borrow: it was consumed */ \
            "v_cndmask_b32_e64 %[d], " W ", 0, vcc\n\t"                                                                   \
            "v_bfi_b32 %[tt], 28, %[tt], " RA "\n\t"                     /* its address in this state's row */           \
            "s_or_b64 %[ev], %[ev], %[amb]\n\t"                                                                           \
            "s_cbranch_scc1 " EPI "f\n\t"                                                                                 \
            /* ---- no lane of the wavefront has an event: commit the step of all four rows ---- */                      \
            "ds_write_b32 %[tt], %[d]\n\t"                               /* the row, shifted */                          \
            "s_waitcnt lgkmcnt(1)\n\t"                                   /* the two reads */
        /* Out of line: the look's events are episode ends (psrs.py:249-269: env.reset() pops the shuffled init queue) and / or rows  */  \
        /* that stay in their state.  The row's next initial states wait in its LDS ring; `left` says how many the loop may take     */  \
        /* before the C++ path has to look (episode cap, init queue empty, ring to refill).  The step is committed (its key carries  */  \
        /* the done bit into the log word the next copy makes), the next state of the rows whose episode ended is their next initial */  \
        /* state, and every row's entry is read again behind the row store.  DRY: where a row without a clear accept goes.          */
#define ROWS_EPI(EPI, BACK, OUT, DRY, ZZ, RA, W, ZN, RN, WN, KEY, KEYP, LMSH)                                                 \
            EPI ":\n\t"                                                                                                    \
            "s_cmp_lg_u64 %[amb], 0\n\t"                                                                                   \
            "s_cbranch_scc1 " OUT "f\n\t"                               /* a lane needs the exact look */                  \
            "v_cmp_eq_u32_e32 vcc, -1, " KEY "\n\t"                     /* all-ones key: a row without a clear accept */   \
            "s_cbranch_vccnz " DRY "f\n\t"                                                                                 \
            "v_cmp_le_u32_e32 vcc, 0x400, " ZN "\n\t"                   /* vcc: the rows whose episode ends */             \
            "v_cmp_eq_u32_e64 %[ev], 0, %[left]\n\t"                                                                       \
            "s_and_b64 %[ev], %[ev], vcc\n\t"                                                                              \
            "s_cbranch_scc1 " OUT "f\n\t"                               /* a row may not take another reset here */       \
            "s_mov_b64 exec, vcc\n\t"                                                                                      \
            "ds_read_b32 " ZN ", %[initp]\n\t"                          /* the next initial state */                      \
            "v_subrev_u32 %[left], 1, %[left]\n\t"                                                                         \
            "v_add_u32 %[initp], 4, %[initp]\n\t"                                                                          \
            "v_and_b32 %[initp], 0xffffff7f, %[initp]\n\t"              /* the ring of 32 is 128 bytes at a 256-byte boundary: wrap */ \
            "s_mov_b64 exec, %[live]\n\t"                                                                                  \
            "ds_write_b32 %[tt], %[d]\n\t"                                                                                 \
            "s_waitcnt lgkmcnt(1)\n\t"                                  /* (the early entry read, the draw, the initial state) */ \
            "v_lshl_add_u32 " RN ", " ZN ", 5, v109\n\t"                                                                   \
            "ds_read_b32 " WN ", " RN "\n\t"                                                                               \
            "s_waitcnt lgkmcnt(0)\n\t"                                                                                     \
            "s_branch " BACK "b\n\t"
        // an event the loop does not serve: nothing of copy I is committed, the row state goes where the C++ code expects it
#define ROWS_OUT_A(OUT, DRY, I, DRYT) OUT ":\n\t" "s_movk_i32 %[it], " I "\n\t" "s_branch 2f\n\t" DRY ":\n\t" "s_movk_i32 %[it], " I "\n\t" "s_branch " DRYT "0f\n\t"
#define ROWS_OUT_B(OUT, DRY, I, DRYT) OUT ":\n\t" "s_movk_i32 %[it], " I "\n\t" "s_branch 22f\n\t" DRY ":\n\t" "s_movk_i32 %[it], " I "\n\t" "s_branch " DRYT "1f\n\t"
#define RA "%[zz]", "%[ra]", "%[w]", "%[zn]", "%[rb]", "%[w2]", "%[key]", "%[key2]"
#define RB "%[zn]", "%[rb]", "%[w2]", "%[zz]", "%[ra]", "%[w]", "%[key2]", "%[key]"
#define ROWS_STEP_(...) ROWS_STEP(__VA_ARGS__)
#define ROWS_EPI_(...) ROWS_EPI(__VA_ARGS__)
        // The dry-row handler (stream format A).  Label 51: the event came from an odd copy, its row state goes to (zz, ra, w, key)
        // first; label 50: the handler.  Temporaries: hv = %[nrd] (entries the window held), p = %[e] (queue position behind the
        // window), dg = %[w2] (sixteen candidates, lane = position), v[120:121] = the state's segment, v122 = compare, v123 = land
        // address / scratch, v[124:125] = address, v126, v127, v117..v119 scratch; s[30:31] = the rows without a clear accept,
        // s[42:43] = lanes.  Nothing of the iteration is committed before the last test has passed; what it leaves for the entry
        // code: the state and the reads of the next look, and in `key` the log word of this step (the first copy after an entry
        // writes "the step before it" from key).
/* sixteen candidates whatever the sector (0.903 -> 0.900 s; with non-temporal reward loads it was the other way round) */
#define ROWS_DRY_SECTOR "v_mov_b32 v121, 16\n\t"
// The candidates of a dry row come straight from the stream.  (Round 4 tried a look into the request areas first -- a top-up of the
// very window end is often already on chip -- under a seqlock on the descriptor: no gain, the event is ~1500 cycles either way, and an
// unresolved race; the experiment lives in `git log` only.)
#define ROWS_DRY_CANDIDATES ROWS_DRY_CANDIDATES_STREAM
#define ROWS_DRY_CANDIDATES_STREAM                                                                                            \
            "v_add3_u32 v124, v120, %[e], v118\n\t"                      /* grouped position of this lane's candidate */ \
            "v_mov_b32 v125, 0\n\t"                                                                                       \
            "v_lshl_add_u64 v[124:125], v[124:125], 2, %[dbase]\n\t"                                                      \
            /* Candidates up to the end of the 64-byte sector the queue's head lies in (the window's last top-up came out of it). */ \
            "v_lshrrev_b32 v121, 2, v124\n\t"                                                                             \
            "v_sub_u32 v121, v121, v118\n\t"                                                                              \
            "v_and_b32 v121, 15, v121\n\t"                                                                                \
            "v_sub_u32 v121, 16, v121\n\t"                             /* candidates in the sector, 1..16 (the same in every lane of the row) */ \
            ROWS_DRY_SECTOR                                                                                                \
            "v_cmp_lt_u32_e32 vcc, v118, v121\n\t"                                                                        \
            "s_mov_b64 exec, s[30:31]\n\t"                                                                                \
            "v_mov_b32 %[zn], -1\n\t"                                  /* (a lane without a candidate accepts nothing) */ \
            "s_and_b64 exec, exec, vcc\n\t"                                                                               \
            "global_load_dword %[w2], v[124:125], off\n\t"                                                                \
            "ds_read_b32 v117, v126\n\t"                                 /* draw c + hv + lane */                        \
            "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
// (SBIAS / SNBIAS: the format's window bias and its negation; LOGH: formats B and C leave the upper bits of the accepted candidate's local
// row beside the long-form log word, as the C++ path does)
#define ROWS_DRY_LOGH                                                                                                        \
            "ds_read_b32 v124, %[ringa] offset:1280\n\t"               /* SY_TICK: its parity names the tick's log buffer */ \
            "v_bfe_u32 v125, %[zn], 8, 2\n\t"                                                                           \
            "v_bfe_u32 v126, %[zn], 11, 7\n\t"                                                                          \
            "v_lshl_or_b32 v125, v126, 2, v125\n\t"                    /* rows_loc_hi(key) */                          \
            "s_lshl_b32 s42, %[it], 1\n\t"                                                                              \
            "s_waitcnt lgkmcnt(0)\n\t"                                                                                  \
            "v_and_b32 v124, 1, v124\n\t"                                                                               \
            "v_lshl_add_u32 v124, v124, 5, %[ringa]\n\t"                                                                \
            "v_add_u32 v124, s42, v124\n\t"                                                                             \
            "ds_write_b16 v124, v125 offset:1408\n\t"                  /* RO_LOGH + 32 x parity + 2 x it */
#define ROWS_DRY_HANDLER(SBIAS, SNBIAS, LOGH)                                                                                \
            "51:\n\t"                                                                                                     \
            "s_waitcnt lgkmcnt(0)\n\t"                                                                                    \
            "v_mov_b32 %[zz], %[zn]\n\t"                                                                                  \
            "v_mov_b32 %[ra], %[rb]\n\t"                                                                                  \
            "v_mov_b32 %[w], %[w2]\n\t"                                                                                   \
            "v_mov_b32 %[key], %[key2]\n\t"                                                                               \
            "50:\n\t"                                                                                                     \
            "s_waitcnt lgkmcnt(0)\n\t"                                   /* (the early read of the next entry is on its way into w2) */ \
            "v_cmp_eq_u32_e64 s[30:31], -1, %[key]\n\t"                  /* the rows without a clear accept */           \
            /* may every row that needs one take an initial state?  (a dry row's step may end an episode: asked of all of them) */ \
            "v_and_b32 v117, 0x400, %[key]\n\t"                                                                           \
            "v_cmp_ne_u32_e32 vcc, 0, v117\n\t"                                                                           \
            "v_cmp_eq_u32_e64 s[42:43], 0, %[left]\n\t"                                                                   \
            "s_and_b64 s[42:43], s[42:43], vcc\n\t"                      /* (the all-ones key carries the done bit: covers the dry rows) */ \
            "s_cbranch_scc1 2f\n\t"                                                                                       \
            /* entries the window held: lanes 0..7 (8..15 repeat them) */                                                \
            "v_cmp_ne_u32_e32 vcc, 0, %[w]\n\t"                                                                           \
            "v_lshl_add_u32 v123, %[zz], 2, %[landb]\n\t"                /* &land[state] */                              \
            "v_lshl_add_u32 v126, %[zz], 2, %[sega]\n\t"                 /* &seg_off[state] */                           \
            "v_cndmask_b32_e64 %[nrd], 0, 1, vcc\n\t"                                                                     \
            "s_nop 1\n\t"                                                                                                 \
            "v_add_u32_dpp %[nrd], %[nrd], %[nrd] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                     \
            "v_add_u32 v107, %[zz], %[claimb]\n\t"                                                                        \
            "ds_read_b32 %[e], v123\n\t"                                                                                  \
            "ds_read2_b32 v[120:121], v126 offset1:1\n\t"                                                                 \
            "v_add_u32_dpp %[nrd], %[nrd], %[nrd] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"                     \
            "v_lshrrev_b32 v118, 2, %[li4]\n\t"                          /* lane of the row, 0..15 */                    \
            "v_add_u32 v119, 4, %[li4]\n\t"                                                                               \
            "v_add_u32_dpp %[nrd], %[nrd], %[nrd] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"                         \
            "v_lshlrev_b32 v119, 24, v119\n\t"                           /* (lane + 1) << 26 */                          \
            "s_waitcnt lgkmcnt(0)\n\t"                                                                                    \
            "v_sub_u32 v122, v121, v120\n\t"                             /* rows of the state */                         \
            "v_sub_u32 v122, v122, %[e]\n\t"                             /* ... still queued behind the window */        \
            "v_cmp_gt_u32_e32 vcc, 16, v122\n\t"                         /* fewer than sixteen: the C++ path */          \
            "s_and_b64 vcc, vcc, s[30:31]\n\t"                                                                            \
            "s_cbranch_vccnz 2f\n\t"                                                                                      \
            /* draw counter of the dry rows behind the entries held: c4 - 255 (the all-ones key was added) + 4 hv, as a 16-lane slot */ \
            "v_add_u32 v127, 0xffffff01, %[c4]\n\t"                                                                       \
            "v_lshl_add_u32 v127, %[nrd], 2, v127\n\t"                                                                    \
            "v_sub_u32 v127, v127, v110\n\t"                             /* 4 x draws consumed */                        \
            "s_sub_u32 s42, 17, %[it]\n\t"                               /* this event's sixteen draws, and eight for every look left in the tick (the loop */ \
            "s_lshl_b32 s42, s42, 5\n\t"                                 /* itself never checks: a tick's worth is in the ring when it starts): 4 x (16 + 8 (15 - it)) */ \
            "v_add_u32 v126, s42, v127\n\t"                                                                               \
            "v_cmp_gt_u32_e32 vcc, v126, %[gen4]\n\t"                    /* ... are not known to be in the ring */     \
            "s_and_b64 vcc, vcc, s[30:31]\n\t"                                                                            \
            "s_cbranch_vccnz 2f\n\t"                                                                                      \
            "v_add_u32 v127, v127, %[li4]\n\t"                                                                            \
            "v_and_or_b32 v126, v127, v115, %[ringa]\n\t"                                                                 \
            ROWS_DRY_CANDIDATES                                                                                            \
            "v_cmp_lt_u32_e32 vcc, " SBIAS ", %[w2]\n\t"                 /* the window's bias (rows_bias) */             \
            "v_add_u32 v122, " SNBIAS ", %[w2]\n\t"                                                                       \
            "v_mov_b32 v125, 0x200\n\t"                                  /* ROWS_NEVER (a literal and vcc do not share the constant bus) */ \
            "s_nop 0\n\t"                                                                                                 \
            "v_cndmask_b32_e32 v122, v125, v122, vcc\n\t"                                                                 \
            "v_sub_co_u32 v126, vcc, v122, v117\n\t"                     /* borrow: not a clear accept */                \
            "v_and_or_b32 %[zn], %[w2], v112, v119\n\t"                                                                   \
            "v_cmp_le_u32_e64 s[42:43], v114, v126\n\t"                  /* a lane near a tie: the C++ path */           \
            "v_cndmask_b32_e64 %[zn], %[zn], -1, vcc\n\t"                                                                 \
            "s_mov_b64 exec, s[30:31]\n\t"                             /* (all sixteen lanes of the dry rows again) */  \
            "s_cmp_lg_u64 s[42:43], 0\n\t"                                                                                \
            "s_cbranch_scc1 52f\n\t"                                                                                      \
            "v_min_u32_dpp %[zn], %[zn], %[zn] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                        \
            "s_nop 1\n\t"                                                                                                 \
            "v_min_u32_dpp %[zn], %[zn], %[zn] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"                        \
            "s_nop 1\n\t"                                                                                                 \
            "v_min_u32_dpp %[zn], %[zn], %[zn] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"                            \
            "s_nop 1\n\t"                                                                                                 \
            "v_min_u32_dpp %[zn], %[zn], %[zn] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                                  \
            "s_nop 0\n\t"                                                                                                 \
            "v_cmp_eq_u32_e32 vcc, -1, %[zn]\n\t"                        /* sixteen rejections in a row: the C++ path */ \
            "s_cbranch_vccnz 52f\n\t"                                                                                     \
            /* ---- every test has passed: commit.  The dry rows first (exec): draw counter, log word, land ---- */       \
            "v_add_u32 %[c4], 0xffffff01, %[c4]\n\t"                                                                      \
            "v_lshl_add_u32 %[c4], %[nrd], 2, %[c4]\n\t"                                                                  \
            "v_add_u32_sdwa %[c4], %[c4], %[zn] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"   \
            "v_lshlrev_b32 v127, 2, %[nrd]\n\t"                                                                           \
            "v_add_u32_sdwa v127, v127, %[zn] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"     /* 4 x candidates the step consumed */ \
            "v_lshlrev_b32 v127, 9, v127\n\t"                                                                             \
            "v_and_b32 v126, 0x400, %[zn]\n\t"                                                                            \
            "v_or3_b32 v127, v127, v126, %[zz]\n\t"                                                                       \
            "v_or_b32 v127, 0x80000000, v127\n\t"                        /* rows_log_word(state left, done, candidates) */ \
            "v_lshrrev_b32 v125, 26, %[zn]\n\t"                          /* k1: the accepted candidate is the k1-th of the sixteen */ \
            "v_sub_u32 v126, v121, v125\n\t"                                                                              \
            "v_min_u32 v126, 8, v126\n\t"                                /* candidates behind it (of the sector's) that become the window */ \
            "v_add3_u32 v126, %[e], v125, v126\n\t"                                                                       \
            "ds_write_b32 v123, v126\n\t"                                /* land = the position behind the window */     \
            LOGH                                                                                                           \
            "v_add_u32 %[ndry], 1, %[ndry]\n\t"                                                                           \
            "s_mov_b64 exec, %[live]\n\t"                                                                                 \
            "ds_write_b32 %[tt], %[d]\n\t"                               /* the rows with a clear accept: shifted; the dry rows: emptied (every lane's slot: the all-ones key) */ \
            "s_mov_b64 exec, s[30:31]\n\t"                                                                                \
            "v_sub_u32_sdwa v126, %[li4], %[zn] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"   /* 4 x (lane - k1): this lane's slot in the new window */ \
            "v_sub_u32 v124, %[ra], v110\n\t"                                                                             \
            "v_cmp_gt_u32_e32 vcc, 32, v126\n\t"                                                                          \
            "v_add_u32 v124, v124, v126\n\t"                                                                              \
            "v_cmp_lt_u32_e64 s[42:43], v118, v121\n\t"                  /* (only the sector's candidates) */            \
            "s_and_b64 exec, exec, vcc\n\t"                                                                               \
            "s_and_b64 exec, exec, s[42:43]\n\t"                                                                          \
            "ds_write_b32 v124, v122\n\t"                                                                                 \
            "s_mov_b64 exec, %[live]\n\t"                                                                                 \
            /* ---- all rows: the key that counts, the log word, the next state ---- */                                  \
            "v_and_or_b32 v126, %[key], v116, %[zz]\n\t"                 /* log word of a row with a clear accept */     \
            "v_cndmask_b32_e64 %[zn], %[key], %[zn], s[30:31]\n\t"                                                        \
            "v_cndmask_b32_e64 %[key], v126, v127, s[30:31]\n\t"         /* -> last_log */                               \
            "v_and_b32 %[zz], v113, %[zn]\n\t"                                                                            \
            "v_cmp_le_u32_e32 vcc, 0x400, %[zz]\n\t"                     /* the rows whose episode ends */               \
            "s_mov_b64 exec, vcc\n\t"                                                                                     \
            "ds_read_b32 %[zz], %[initp]\n\t"                                                                             \
            "v_subrev_u32 %[left], 1, %[left]\n\t"                                                                        \
            "v_add_u32 %[initp], 4, %[initp]\n\t"                                                                         \
            "v_and_b32 %[initp], 0xffffff7f, %[initp]\n\t"                                                                \
            "s_mov_b64 exec, %[live]\n\t"                                                                                 \
            "v_and_or_b32 %[nrd], %[c4], v115, %[ringa]\n\t"                                                              \
            "ds_read_b32 %[kt], %[nrd]\n\t"                                                                               \
            "s_waitcnt lgkmcnt(1)\n\t"                                                                                    \
            "v_lshl_add_u32 %[ra], %[zz], 5, v109\n\t"                                                                    \
            "ds_read_b32 %[w], %[ra]\n\t"                                /* behind every store of the step */            \
            "s_add_u32 %[it], %[it], 1\n\t"                                                                               \
            "s_waitcnt lgkmcnt(0)\n\t"                                   /* (the entry code copies w and kt into the second register set) */ \
            "s_cmp_lt_u32 %[it], 16\n\t"                                                                                  \
            "s_cbranch_scc1 40b\n\t"                                     /* enter the loop again at the next copy */     \
            "s_lshl_b64 s[22:23], s[26:27], 15\n\t"                      /* it was the tick's last step: its log word */ \
            "s_nop 0\n\t"                                                                                                 \
            "v_cndmask_b32_e64 %[elog], %[elog], %[key], s[22:23]\n\t"                                                    \
            "s_branch 3f\n\t"                                                                                             \
            "52:\n\t"                                                                                                     \
            "s_mov_b64 exec, %[live]\n\t"                                                                                 \
            "s_branch 2f\n\t"
#define ROWS_FAST_ASM(SPAY, SZM, SAMB, SKM, DRYT, HANDLER)                                                                    \
        asm volatile(                                                                                                    \
            "s_mov_b64 s[24:25], exec\n\t"                               /* rows that have stopped sit the loop out: nothing of theirs is read or written */ \
            "s_mov_b64 exec, %[live]\n\t"                                                                                 \
            "s_mov_b32 s26, 0x10001\n\t"                                 /* s[26:27]: lane 0 of every row */              \
            "s_mov_b32 s27, 0x10001\n\t"                                                                                  \
            "v_and_b32 v110, 28, %[li4]\n\t"                                                                              \
            "v_add_u32 v109, 0x5c0, %[ringa]\n\t"                        /* RO_WIN behind RO_RING */                     \
            "v_mov_b32 v112, " SPAY "\n\t"                                                                                \
            "v_mov_b32 v113, " SZM "\n\t"                                                                                 \
            "v_add_u32 v111, 4, v110\n\t"                                                                                 \
            "v_mov_b32 v114, " SAMB "\n\t"                                                                                \
            "v_mov_b32 v115, 0x3fc\n\t"                                                                                   \
            "v_lshlrev_b32 v111, 24, v111\n\t"                                                                            \
            "v_add_u32 v109, v109, v110\n\t"                             /* this lane's entry of state 0's window row */ \
            "v_mov_b32 v116, " SKM "\n\t"                                                                                 \
            "40:\n\t"                                                                                                     \
            "v_mov_b32 %[zn], %[zz]\n\t"                                 /* both register sets hold the row state: any copy may be the first */ \
            "v_mov_b32 %[rb], %[ra]\n\t"                                                                                  \
            "v_mov_b32 %[w2], %[w]\n\t"                                                                                   \
            "v_mov_b32 %[key2], %[key]\n\t"                              /* (key arrives holding last_log: the first copy rebuilds that word, whichever its parity) */ \
            "v_bfi_b32 %[e], v116, 0, %[key]\n\t"                        /* last_log outside the key's part of a log word: what the first copy takes for "the state left" */ \
            "s_bitcmp1_b32 %[it], 0\n\t"                                                                                  \
            "s_cbranch_scc1 31f\n\t"                                                                                      \
            "v_mov_b32 %[zn], %[e]\n\t"                                  /* an even copy comes first: it reads the previous step's state from zn */ \
            "s_branch 32f\n\t"                                                                                            \
            "31:\n\t"                                                                                                     \
            "v_mov_b32 %[zz], %[e]\n\t"                                  /* an odd copy comes first (its own state is in zn, rb, w2) */ \
            "32:\n\t"                                                                                                     \
            "s_waitcnt lgkmcnt(0)\n\t"                                                                                    \
            "s_cmp_eq_u32 %[it], 0\n\t"                                                                                   \
            "s_cbranch_scc1 100f\n\t"                                                                                     \
            "s_getpc_b64 s[20:21]\n\t"                                   /* = the address of the next instruction; the table starts 20 bytes behind it */ \
            "s_lshl_b32 s22, %[it], 2\n\t"                                                                                \
            "s_add_u32 s22, s22, 20\n\t"                                                                                  \
            "s_add_u32 s20, s20, s22\n\t"                                                                                 \
            "s_addc_u32 s21, s21, 0\n\t"                                                                                  \
            "s_setpc_b64 s[20:21]\n\t"                                                                                    \
            "s_branch 100f\n\t" "s_branch 101f\n\t" "s_branch 102f\n\t" "s_branch 103f\n\t"                              \
            "s_branch 104f\n\t" "s_branch 105f\n\t" "s_branch 106f\n\t" "s_branch 107f\n\t"                              \
            "s_branch 108f\n\t" "s_branch 109f\n\t" "s_branch 110f\n\t" "s_branch 111f\n\t"                              \
            "s_branch 112f\n\t" "s_branch 113f\n\t" "s_branch 114f\n\t" "s_branch 115f\n\t"                              \
            ROWS_STEP_("200", "100", RA, "63")                                                                            \
            ROWS_STEP_("201", "101", RB, "0")                                                                             \
            ROWS_STEP_("202", "102", RA, "1")                                                                             \
            ROWS_STEP_("203", "103", RB, "2")                                                                             \
            ROWS_STEP_("204", "104", RA, "3")                                                                             \
            ROWS_STEP_("205", "105", RB, "4")                                                                             \
            ROWS_STEP_("206", "106", RA, "5")                                                                             \
            ROWS_STEP_("207", "107", RB, "6")                                                                             \
            ROWS_STEP_("208", "108", RA, "7")                                                                             \
            ROWS_STEP_("209", "109", RB, "8")                                                                             \
            ROWS_STEP_("210", "110", RA, "9")                                                                             \
            ROWS_STEP_("211", "111", RB, "10")                                                                            \
            ROWS_STEP_("212", "112", RA, "11")                                                                            \
            ROWS_STEP_("213", "113", RB, "12")                                                                            \
            ROWS_STEP_("214", "114", RA, "13")                                                                            \
            ROWS_STEP_("215", "115", RB, "14")                                                                            \
            "116:\n\t"                                                                                                    \
            "v_and_or_b32 %[e], %[key2], v116, %[zn]\n\t"                /* the tick is over: the last step's log word; copy 15 left the row state in (zz, ra, w) */ \
            "s_lshl_b64 s[22:23], s[26:27], 15\n\t"                                                                       \
            "s_movk_i32 %[it], 16\n\t"                                                                                    \
            "v_cndmask_b32_e64 %[elog], %[elog], %[e], s[22:23]\n\t"                                                      \
            "s_branch 3f\n\t"                                                                                             \
            ROWS_EPI_("200", "101", "300", "400", RA, "0")                                                                \
            ROWS_EPI_("201", "102", "301", "401", RB, "0")                                                                \
            ROWS_EPI_("202", "103", "302", "402", RA, "0")                                                                \
            ROWS_EPI_("203", "104", "303", "403", RB, "0")                                                                \
            ROWS_EPI_("204", "105", "304", "404", RA, "0")                                                                \
            ROWS_EPI_("205", "106", "305", "405", RB, "0")                                                                \
            ROWS_EPI_("206", "107", "306", "406", RA, "0")                                                                \
            ROWS_EPI_("207", "108", "307", "407", RB, "0")                                                                \
            ROWS_EPI_("208", "109", "308", "408", RA, "0")                                                                \
            ROWS_EPI_("209", "110", "309", "409", RB, "0")                                                                \
            ROWS_EPI_("210", "111", "310", "410", RA, "0")                                                                \
            ROWS_EPI_("211", "112", "311", "411", RB, "0")                                                                \
            ROWS_EPI_("212", "113", "312", "412", RA, "0")                                                                \
            ROWS_EPI_("213", "114", "313", "413", RB, "0")                                                                \
            ROWS_EPI_("214", "115", "314", "414", RA, "0")                                                                \
            ROWS_EPI_("215", "116", "315", "415", RB, "0")                                                                \
            ROWS_OUT_A("300", "400", "0", DRYT) ROWS_OUT_B("301", "401", "1", DRYT) ROWS_OUT_A("302", "402", "2", DRYT) ROWS_OUT_B("303", "403", "3", DRYT)  \
            ROWS_OUT_A("304", "404", "4", DRYT) ROWS_OUT_B("305", "405", "5", DRYT) ROWS_OUT_A("306", "406", "6", DRYT) ROWS_OUT_B("307", "407", "7", DRYT)  \
            ROWS_OUT_A("308", "408", "8", DRYT) ROWS_OUT_B("309", "409", "9", DRYT) ROWS_OUT_A("310", "410", "10", DRYT) ROWS_OUT_B("311", "411", "11", DRYT) \
            ROWS_OUT_A("312", "412", "12", DRYT) ROWS_OUT_B("313", "413", "13", DRYT) ROWS_OUT_A("314", "414", "14", DRYT) ROWS_OUT_B("315", "415", "15", DRYT) \
            HANDLER                                                                                                       \
            "60:\n\t"                                                    /* (stream format B: a row without a clear accept leaves for the C++ path) */ \
            "s_branch 2f\n\t"                                                                                             \
            "61:\n\t"                                                                                                     \
            "22:\n\t"                                                    /* event in an odd copy: its state is in (zn, rb, w2), its key in key2 */ \
            "s_waitcnt lgkmcnt(0)\n\t"                                   /* (the early read of the next entry is on its way into w) */ \
            "v_mov_b32 %[zz], %[zn]\n\t"                                                                                  \
            "v_mov_b32 %[ra], %[rb]\n\t"                                                                                  \
            "v_mov_b32 %[w], %[w2]\n\t"                                                                                   \
            "v_mov_b32 %[key], %[key2]\n\t"                                                                               \
            "2:\n\t"                                                                                                      \
            "v_sub_u32_sdwa %[c4], %[c4], %[key] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"  /* nothing of this iteration is committed */ \
            "3:\n\t"                                                                                                      \
            "s_mov_b64 exec, s[24:25]\n\t"                                                                                \
            "s_waitcnt lgkmcnt(0)"                                       /* no read of this loop outlives it */          \
            : [w] "+v"(w), [kt] "+v"(kt), [c4] "+v"(c4), [ra] "+v"(ra), [zz] "+v"(zz), [initp] "+v"(initp), [left] "+v"(left), [key] "+v"(key),         \
              [elog] "+v"(elog), [ndry] "+v"(n_dry), [key2] "=&v"(key2), [d] "=&v"(d), [tt] "=&v"(tt), [nrd] "=&v"(nrd), [e] "=&v"(e), [zn] "=&v"(zn),  \
              [rb] "=&v"(rb), [w2] "=&v"(w2), [amb] "=&s"(amb), [ev] "=&s"(ev), [it] "+s"(it)                                                           \
            : [ringa] "v"(ring_a), [claimb] "v"(claim_a), [dmaa] "s"(dma_a), [li4] "v"(li4), [landb] "v"(land_a), [gen4] "v"(gen4), [dbase] "v"(dbase), [sega] "s"(seg_a),  \
              [live] "s"(live)                                                                                                                           \
            : "vcc", "scc", "memory", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s30", "s31", "s42", "s43", "v110", "v111", "v112",        \
              "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v96", "v97",       \
              "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "s36", "s37", "s38", "s39", "s40", "s41")
        static_assert(RO_WIN - RO_RING == 0x5c0u && DS_RQD * 256u == 2048u && DS_RQS * 256u == 2304u && DS_RQ_SET * 256u == 2560u && DS_RQB * 256u == 1024u, "immediates of the loop");
        static_assert(rows_format(OFFSIM_STREAMS_A).paymask == 0x7ffu && rows_format(OFFSIM_STREAMS_A).zmask == 0x7ffu &&
                      rows_format(OFFSIM_STREAMS_A).amb == 0xffff7800u && (rows_format(OFFSIM_STREAMS_A).emask | 0x400u) == 0x3c000400u &&
                      rows_format(OFFSIM_STREAMS_A).bias == 0x8000u && rows_format(OFFSIM_STREAMS_A).smask == 0x3ffu, "the literals of the format-A loop");
        static_assert(rows_format(OFFSIM_STREAMS_B).paymask == 0xffffu && rows_format(OFFSIM_STREAMS_B).zmask == 0x4ffu &&
                      rows_format(OFFSIM_STREAMS_B).amb == 0xffff0000u && (rows_format(OFFSIM_STREAMS_B).emask | 0x400u) == 0x3c00ff00u, "the literals of the format-B loop");
        static_assert(rows_format(OFFSIM_STREAMS_C).paymask == 0x3ffffu && rows_format(OFFSIM_STREAMS_C).zmask == 0x4ffu &&
                      rows_format(OFFSIM_STREAMS_C).amb == 0xfffc0000u && (rows_format(OFFSIM_STREAMS_C).emask | 0x400u) == 0x3c03ff00u, "the literals of the format-C loop");
        static_assert(RO_SYNC - RO_RING == 1280u && SY_TICK == 0 && RO_LOGH - RO_RING == 1408u && rows_format(OFFSIM_STREAMS_B).bias == 0u &&
                      rows_format(OFFSIM_STREAMS_C).bias == 0u, "immediates of the dry-row handler");
        if constexpr (fmt_c) {
            ROWS_FAST_ASM("0x3ffff", "0x4ff", "0xfffc0000", "0x3c03ff00", "5", ROWS_DRY_HANDLER("0", "0", ROWS_DRY_LOGH));
        } else if constexpr (fmt_b) {
            ROWS_FAST_ASM("0xffff", "0x4ff", "0xffff0000", "0x3c00ff00", "5", ROWS_DRY_HANDLER("0", "0", ROWS_DRY_LOGH));
        } else {
            ROWS_FAST_ASM("0x7ff", "0x7ff", "0xffff7800", "0x3c000400", "5", ROWS_DRY_HANDLER("0x8000", "0xffff8000", ""));
        }
#undef ROWS_FAST_ASM
#undef ROWS_DRY_HANDLER
#undef ROWS_DRY_LOGH
#undef ROWS_STEP
#undef ROWS_EPI
#undef ROWS_OUT_A
#undef ROWS_OUT_B
#undef ROWS_STEP_
#undef ROWS_EPI_
#undef RA
#undef RB
        c = (c4 - li4w) >> 2;
        z = zz;
        {
            const uint32_t served = ((initp - initp0) >> 2) & 31u;  // episode ends the loop served (the pointer wraps)
            ic += served;
            ep += served;
        }
        ex_key = key;
        ex_d = d;
        ex_slot = tt;
        ex_amb = amb;
    };
    // the iteration the loop was left in, from what it handed over (no second look, no re-read of the window)
    auto resume_step = [&](uint32_t it) __attribute__((always_inline)) {
        PS_T1();
        const uint32_t key = ex_key;
        const bool amb = ((uint32_t)(ex_amb >> (rw * 16u)) & 0xffffu) != 0u;
        if (key != 0xffffffffu && !amb) {  // clear accept: the stores the loop had prepared, then the episode end if that was the event
            log_step(it, (key & (F.emask | 0x400u)) | z);
            LV32(ex_slot) = ex_d;
            c += (key >> ROWS_LIF) & 15u;
            z = key & F.smask;
            if (key & 0x400u) {
                ep++;
                do_reset(it + 1u);
            }
            PS(3);
        } else {
            exact_step(it, amb);
            PS(8);
        }
        if (!dead) need_draws((ROWS_TICK - (it + 1u)) * 8u + 8u, it + 1u);
        PS(9);
    };

    if (!dead) need_draws(136u, 0u);  // (HELPER: the helper wavefront has filled the ring)
    // shader cycles and 100 MHz ticks of the chain (out.dbg): the clock it ran at, and how the wavefronts' run times spread
    const uint64_t pf_c0 = out.dbg ? __builtin_amdgcn_s_memtime() : 0ull, pf_r0 = out.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    for (uint32_t drained = 0;;) {
        uint32_t it = 0;
        if (drained) it = ROWS_TICK;  // every row has stopped: only the reward pipeline is still draining
        while (it < ROWS_TICK) {
            issue_reads();
            const uint64_t live = __ballot(dead == 0u);
            if (!TRACE && live != 0ull) {  // (rows that have stopped are masked out of the loop)
                PF_START();
                fast_run(it, live);
                PF_ADD(pf_fast);
                if (it == ROWS_TICK) break;
                PF_START();
                if (!dead) resume_step(it);
            } else {
                PF_START();
                uint32_t key = 0;
                bool amb = false;
                look(key, amb);
                if (!dead) slow_step(key, amb, it);
            }
            it++;
            PF_ADD(pf_slow);
#ifdef OFFSIM_ROWS_PROF
            pf_nslow++;
#endif
        }
        PF_START();
        tick();
        PF_ADD(pf_tick);
        if (__ballot(!dead) == 0ull && (HELPER || ++drained == 3u)) break;  // (single wavefront: two more ticks drain the reward pipeline)
    }
    const uint64_t pf_c1 = out.dbg ? __builtin_amdgcn_s_memtime() - pf_c0 : 0ull, pf_r1 = out.dbg ? __builtin_amdgcn_s_memrealtime() - pf_r0 : 0ull;
#ifdef OFFSIM_ROWS_PROF
    pf_ph[10] = pf_c1;
    pf_ph[11] = pf_r1;
#endif
    if (!HELPER && status == OFFSIM_ST_EXHAUSTED) {  // psrs.py:265: the cut-short episode still logs its length
        if (li == 0u && r < (int64_t)ro.R && out.ep_len && (int64_t)n_len <= out.ep_cap) out.ep_len[r * (out.ep_cap + 1) + n_len] = (int32_t)len_acc;
        n_len++;
    }
    // (diagnostics: top-ups made, aimed at a window end that had moved, not arrived in time -- summed over the row's lanes)
    if (out.dbg) {
        for (int m = 1; m < 16; m <<= 1) {
            n_req += __shfl_xor(n_req, m);
            n_miss += __shfl_xor(n_miss, m);
            n_late += __shfl_xor(n_late, m);
        }
    }
    // ---- write the env state back ----
    if (r < (int64_t)ro.R) {
        for (uint32_t s = li; s < n_slots; s += 16u) {  // cursor = the position behind the window minus the entries it holds
            const scan_u32x4 h0 = LV128(win_a + s * 32u), h1 = LV128(win_a + s * 32u + 16u);
            const uint32_t hv = (h0.x != ROWS_EMPTY) + (h0.y != ROWS_EMPTY) + (h0.z != ROWS_EMPTY) + (h0.w != ROWS_EMPTY) +
                                (h1.x != ROWS_EMPTY) + (h1.y != ROWS_EMPTY) + (h1.z != ROWS_EMPTY) + (h1.w != ROWS_EMPTY);
            cur_glb[s] = LV32(land_a + s * 4u) - hv;
        }
        if (li == 0u) {
            ro.init_cursor[r] = ic;
            ro.cur_slot[r] = (int32_t)z;
            if (c && philox) {
                ro.rng[4 * r + 1] = ph_c0 + c;  // (seed, draws consumed so far, 0, 0)
            } else if (c) {
                const U128 base = u128(rng4[0], rng4[1]), inc = u128(rng4[2], rng4[3]);
                const U128 nb = pcg_apply(pcg_jump(inc, c), base);
                ro.rng[4 * r + 0] = nb.hi;
                ro.rng[4 * r + 1] = nb.lo;
            }
            if (!HELPER) {  // (HELPER: the helper wavefront owns the sums and writes them)
                out.sum_g[r] = sum_g;
                out.n_ep[r] = ep_acc;
                out.n_len[r] = n_len;
            }
            out.steps[r] = steps;
            out.cand[r] = c;
            out.status[r] = status;
            if (status == OFFSIM_ST_PROTOCOL) atomicOr(&g_async_fault, OFFSIM_FAULT_SCAN);
#ifdef OFFSIM_ROWS_PROF
            if (out.dbg) {
                out.dbg[4 * r + 0] = (int64_t)pf_fast;
                out.dbg[4 * r + 1] = (int64_t)pf_slow;
                out.dbg[4 * r + 2] = (int64_t)pf_tick;
                out.dbg[4 * r + 3] = (int64_t)(pf_nslow | ((uint64_t)n_dry << 32));
                if (out.ep_g && out.ep_cap >= 24)
                    for (int k = 0; k < 12; k++) out.ep_g[r * out.ep_cap + k] = (double)pf_ph[k];
            }
#else
            if (out.dbg) {
                out.dbg[4 * r + 0] = (int64_t)n_dry | ((int64_t)n_req << 32);
                out.dbg[4 * r + 1] = (int64_t)n_tie | ((int64_t)n_late << 16) | ((int64_t)n_miss << 40);
                out.dbg[4 * r + 2] = (int64_t)pf_c1;
                out.dbg[4 * r + 3] = (int64_t)pf_r1;
            }
#endif
        }
    }
}

}  // namespace offsim
