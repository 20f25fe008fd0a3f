// shuffle_chunk.hpp -- exact Fisher-Yates for queues that do not fit LDS (states of more than 65536 rows), without random
// accesses to global memory.
//
// The in-place variant of shuffle_wave.hpp keeps such a segment in global memory and pays a random 128-byte line fetched plus a
// 64-byte sector written back per swap (202 B per swap measured, 3 TB/s of random lines: the memory system's limit).  But two steps
// of the chain   for i = n-1 .. 1:  j = interval(i);  swap(x[i], x[j])   commute unless the EARLIER one's partner j is the later
// one's own position or partner: a step never touches a position above its own.  So the positions are cut into chunks of CB, and
// the chunks are processed top-down, one at a time in LDS:
//   phase I   the steps ABOVE this chunk whose partner lies in it have left a message each in the chunk's list -- in decreasing
//             order of their i, which is the order they have to be applied in -- holding (i, j, the value x[i] had at step i):
//             the chunk takes the value at j (that is x[i]'s final value: it goes into the reply list of i's chunk) and puts the
//             message's value there;
//   phase II  the chunk's own steps, hi-1 .. lo: a partner inside the chunk is an ordinary swap in LDS (the grouped apply of
//             shuffle_wave.hpp: 64 steps at a time, cut into pieces where two of them touch a common position); a partner below
//             the chunk sends (i, j, x[i]) to the list of j's chunk, and x[i] is final once the reply comes;
//   then the chunk is stored (4 B per position).  When the chain has run, every chunk is loaded once more, its replies are
//   scattered into it in LDS, and the finished order goes out as the candidate streams (digest | local-row bits 16.., local-row
//   low half).
// All global traffic is sequential: 8 + 8 B per message, 8 + 8 B per reply, 4 + 4 B per position and the 6 B of the streams, against
// 202 B per swap.  The lists keep their order without sorting: chunks run top-down, a chunk's steps run in decreasing i, and the
// slot of a message in its list is taken with ONE ds_add_rtn_u32 per group of 64 steps -- the LDS applies the lanes of one such
// instruction that hit the same address in ascending lane order (= decreasing i; the property scan_rows.hpp relies on as well,
// offsim_selftest_lds_atomic_order).
//
// One PERSISTENT workgroup per CU: it owns one slice of the workspace (message pool + reply pool) and takes (state, rollout) chains
// from a global counter, longest states first.  Roles as in shuffle_wave.hpp: G x 2 (raw draws), C (the j sequence), A (everything
// above).  List capacities: a chunk [lo, hi) receives Binomial-sum(m / (i + 1), i >= hi) messages, mean mu = m ln(n / hi); its list
// holds mu + 8 sqrt(mu) + 128 (a list that overflowed -- 6e-16 per list -- raises OFFSIM_FAULT_SHUFFLE: the call's orders are void).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pcg64_dev.hpp"
#include "shuffle_wave.hpp"

namespace offsim {

enum { SC_WORK = 11, SC_NCH = 12 };  // further words of the control block (SH_* 0..10)

// messages a chunk list of a segment of n rows has to hold (see above); host and device use the same expression
__host__ __device__ inline uint32_t shc_list_cap(uint32_t n, uint32_t lo, uint32_t hi) {
#ifdef SHC_TEST_SMALL_LISTS  // (test build: lists that overflow, so that the fault path runs)
    return 16u;
#endif
    if (hi >= n) return 128u;
    const float mu = (float)(hi - lo) * logf((float)n / (float)hi);
    return (uint32_t)(mu + 8.0f * sqrtf(mu)) + 128u;
}
// read a dword per lane into LDS at lds_dst_uniform + 4 * lane without a register for the data (brings the line into L2)
__device__ __forceinline__ void shc_touch(const void *gptr, uint32_t lds_dst_uniform) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gptr), "s"(lds_dst_uniform)
        : "memory");
}
// the next slot of a list (lanes of one instruction that take from the same counter get ascending slots in lane order)
__device__ __forceinline__ uint32_t shc_take(lds_vu32 *p) {
    return __hip_atomic_fetch_add((__attribute__((address_space(3))) uint32_t *)p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t shc_xchg(lds_vu32 *p, uint32_t v) {
    return __hip_atomic_exchange((__attribute__((address_space(3))) uint32_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// the upper bits h of a local row (bits 16.. in stream format B, bits 8.. in format C) where the digest word carries them: bits 8, 9 and 11..
__host__ __device__ inline uint32_t shc_hi_bits(uint32_t h) { return ((h & 3u) << 8) | ((h >> 2) << 11); }
// entry k of a loc stream of `loc_bits`-bit entries (16: formats A, B; 8: format C)
__device__ __forceinline__ void shc_loc_store(unsigned char *lc, uint32_t k, uint32_t v, uint32_t loc_bits) {
    if (loc_bits == 8u) lc[k] = (unsigned char)v;
    else ((uint16_t *)lc)[k] = (uint16_t)v;
}
__device__ __forceinline__ uint32_t shc_loc_load(const unsigned char *lc, uint32_t k, uint32_t loc_bits) {
    return loc_bits == 8u ? (uint32_t)lc[k] : (uint32_t)((const uint16_t *)lc)[k];
}
// message pool entries a workgroup needs for segments of up to n rows (host side: sizes the workspace)
inline uint64_t shc_pool_entries(uint32_t n, uint32_t cb) {
    uint64_t tot = 0;
    for (uint32_t lo = 0; lo < n; lo += cb) tot += (uint64_t)shc_list_cap(n, lo, lo + cb < n ? lo + cb : n) + 8u;
    return tot + tot / 64u + 1024u;  // (slack for float rounding between host and device and between segment sizes)
}

// LDS: [ctrl 16 w][draw ring RG w][j ring SQ + 64 w][moff: kcap + 1 w][mcnt, rcnt: kcap w each] (kcap = chunks of the table's longest state)[xd: CB + 64 w][xl: CB + 64 w]
// (RG raw draws in the ring, SQ partners in the j ring: powers of two, RG a multiple of 256, SQ >= 384)
// (PLAIN: every chain's records are plain row indices -- the orders go out as permutations -- and the second word of an entry does not exist)
template <uint32_t CB, uint32_t SHC_RG, uint32_t SHC_SQ, bool PLAIN>
constexpr uint32_t shc_lds_bytes(uint32_t kcap) { return 4u * (16u + SHC_RG + SHC_SQ + 64u + 3u * kcap + 16u) + (PLAIN ? 4u : 8u) * (CB + 64u); }

template <uint32_t CB, uint32_t SHC_RG, uint32_t SHC_SQ, bool PLAIN>
__global__ void __launch_bounds__(256)
    k_shuffle_chunked(const uint32_t *__restrict__ seg_off, int64_t N, const uint64_t *__restrict__ seeds, int32_t n_perm,
                      const uint32_t *__restrict__ work_seg, const uint32_t *__restrict__ n_work_seg, uint32_t *__restrict__ counter, uint64_t *__restrict__ ws,
                      int64_t ws_block_words, uint32_t msg_cap, uint32_t kcap, const uint32_t *__restrict__ dig32, uint32_t *__restrict__ dig_out,
                      void *__restrict__ loc_out, int32_t n_slots, int64_t N0, uint32_t *__restrict__ init_perm, uint32_t *__restrict__ perm_out,
                      int64_t lc_words, uint32_t loc_bits) {
    constexpr uint32_t JB = __builtin_ctz(CB);  // bits of a position inside its chunk
    static_assert((CB & (CB - 1u)) == 0u && JB <= 15u, "chunk size: a power of two, at most 32768 (a position inside its chunk travels in 15 bits)");
    extern __shared__ __align__(16) unsigned char lds_raw[];
    lds_vu32 *ctrl = (lds_vu32 *)lds_raw;
    lds_vu32 *ring = ctrl + 16;
    lds_vu32 *jq = ring + SHC_RG;
    lds_vu32 *moff = jq + SHC_SQ + 64u, *mcnt = moff + kcap + 16u, *rcnt = mcnt + kcap;  // (moff[K] = the end of the last list)
    // the chunk: one RECORD per position, in the streams' own layout -- xd = the digest word (digest | bits 16.. of the local row),
    // xl = the local row's low half; entries [CB .. CB+63]: one scratch entry per lane (lanes without a partner in the chunk)
    lds_vu32 *xd = rcnt + kcap;
    lds_vu32 *xl = xd + (PLAIN ? 0u : CB + 64u);  // (a 32-bit word per entry: phase I exchanges it; PLAIN: not there, never touched)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // roles: 0 and 3 = G, 1 = C, 2 = A
    // this workgroup's pools: messages as {record | j inside its chunk << 48} + {i}, replies as {record | i inside its chunk << 48}
    uint64_t *m64 = ws + (int64_t)blockIdx.x * ws_block_words;
    uint32_t *m32 = (uint32_t *)(m64 + msg_cap);
    uint64_t *rpool = m64 + msg_cap + (PLAIN ? 0u : (msg_cap + 1u) / 2u);  // (PLAIN: a message is one word: row 23 bits | j << 23 | i << (23 + JB))
    // (chains whose records are plain row indices -- the init queue, and every chain when the orders go out as permutations, perm_out != NULL --
    // have no use for the records' low halves: those are written to this scratch area)
    unsigned char *lc_plain = (unsigned char *)(m64 + ws_block_words - lc_words);
    const uint32_t n_work = n_work_seg[0] * (uint32_t)n_perm;

    for (;;) {
        __syncthreads();  // (the previous chain's last reads of the control block and the lists are done)
        if (threadIdx.x == 0) ctrl[SC_WORK] = atomicAdd(counter, 1u);
        __syncthreads();
        const uint32_t w = ctrl[SC_WORK];
        if (w >= n_work) return;
        const uint32_t s = work_seg[w / (uint32_t)n_perm];
        const int32_t r = (int32_t)(w % (uint32_t)n_perm);
        // (work item "state n_slots" = the rollout's init queue, psrs.py:22-23: plain row indices 0 .. N0-1, no digests)
        const bool initq = s == (uint32_t)n_slots, plain = initq || perm_out != nullptr;
        const uint32_t beg = initq ? 0u : seg_off[s], n = initq ? (uint32_t)N0 : seg_off[s + 1] - beg;
        const uint32_t K = (n + CB - 1u) / CB;  // (<= kcap: the launch sized the list arrays for the table's longest state)
        uint32_t *dg = initq ? init_perm + (int64_t)r * N0 : (perm_out ? perm_out : dig_out) + (int64_t)r * N + beg;
        unsigned char *lc = plain ? lc_plain : (unsigned char *)loc_out + ((int64_t)r * N + beg) * (int64_t)(loc_bits >> 3);
        const uint32_t base_val = initq ? 0u : beg;  // (a plain record = the grouped row, or the index into the init rows)
        const uint32_t *dsrc = plain ? seg_off : dig32 + beg;  // (plain records: never read)
        __syncthreads();  // (everyone has read SC_WORK)
        if (threadIdx.x < 16u) ctrl[threadIdx.x] = 0u;
        for (uint32_t k = threadIdx.x; k < K; k += 256u) {
            const uint32_t lo = k * CB, hi = lo + CB < n ? lo + CB : n;
            moff[k + 1u] = shc_list_cap(n, lo, hi);
            mcnt[k] = 0u;
            rcnt[k] = 0u;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t off = 0;
            for (uint32_t k = 0; k < K; k++) {  // capacities -> offsets
                const uint32_t cap = moff[k + 1u];
                moff[k] = off;
                off += cap;
            }
            moff[K] = off;
            if (off > msg_cap) {  // the workspace was sized for shorter segments
                ctrl[SH_ABORT] = 1u;
                atomicOr(&g_async_fault, OFFSIM_FAULT_SHUFFLE);
            }
        }
        __syncthreads();
        if (ctrl[SH_ABORT]) return;

        if (n >= 2u) {
            if (wave == 0 || wave == 3) {
                // ---------------- G (two wavefronts, alternate blocks of 128 draws), as in shuffle_wave.hpp
                const uint32_t g = wave == 0 ? 0u : 1u;
                const PcgInit p = pcg_seed(seeds[r]);
                const Jump j128 = pcg_jump(p.inc, 128);
                U128 st = pcg_apply(pcg_jump(p.inc, 64ull * g + (uint64_t)lane + 1), p.state);
                uint32_t blk = g, done_blocks = 0, cpub = 0;
                for (;;) {
                    bool stop = false;
                    uint32_t polls = 0;
                    while ((blk + 1u) * 128u - cpub > SHC_RG) {
                        shuf_bound(ctrl, polls);
                        if (sh_ld(ctrl + SH_DONE)) {
                            stop = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                        cpub = sh_ld(ctrl + SH_CPUB);
                    }
                    if (stop) break;
                    const uint64_t o = pcg_output(st);
                    st = pcg_apply(j128, st);
                    const uint32_t idx = (blk * 128u + 2u * (uint32_t)lane) & (SHC_RG - 1u);
#ifdef SHUF_G_B64  // (A/B: one 64-bit store -- 0.452 against 0.446 s per reset of a 25-state table, tools/ab_reset.sh)
                    *(lds_vu64 *)(ring + idx) = o;  // low half first
#else
                    ring[idx] = (uint32_t)o;  // low half first
                    ring[idx + 1u] = (uint32_t)(o >> 32);
#endif
                    blk += 2u;
                    done_blocks++;
                    sh_st(ctrl + (g ? SH_GEN1 : SH_GEN0), done_blocks);
                }
            } else if (wave == 1) {
                // ---------------- C: the j sequence, 2 x 64 draws per iteration, as in shuffle_wave.hpp
                uint32_t i = n - 1u, c = 0, avail = 0, fill = 0, tail = 0, c_pub = 0;
                uint32_t mask = 0xffffffffu >> __builtin_clz(i);
                int lowpow = (int)((mask >> 1) + 1u);
                auto wait_draws = [&](uint32_t upto) {
                    uint32_t polls = 0;
                    while (upto > avail) {
                        const uint64_t gg = *(lds_vu64 *)(ctrl + SH_GEN0);
                        const uint32_t g0 = sh_rfl((uint32_t)gg), g1 = sh_rfl((uint32_t)(gg >> 32));
                        avail = 128u * (g0 <= g1 ? 2u * g0 : 2u * g1 + 1u);
                        if (upto > avail) {
                            shuf_bound(ctrl, polls);
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                };
                auto wait_room = [&](uint32_t upto) {
                    uint32_t polls = 0;
                    while (upto - tail > SHC_SQ) {
                        tail = sh_ld(ctrl + SH_TAIL);
                        if (upto - tail > SHC_SQ) {
                            shuf_bound(ctrl, polls);
                            __builtin_amdgcn_s_sleep(2);
                        }
                    }
                };
                auto settle = [&](uint64_t &bal, int &rk, uint32_t v, uint32_t ib) {
                    uint64_t f = bal & __ballot((int)v > (int)ib - rk);
                    while (f) {
                        bal &= ~(1ull << sh_ff1(f));
                        rk = sh_rank(bal);
                        f = bal & __ballot((int)v > (int)ib - rk);
                    }
                };
                wait_draws(c + 128u);
                uint32_t r1 = ring[c + (uint32_t)lane], r2 = ring[c + 64u + (uint32_t)lane];
                while (i >= 1u) {
                    wait_draws(c + 256u);
                    const uint32_t p1 = ring[(c + 128u + (uint32_t)lane) & (SHC_RG - 1u)];
                    const uint32_t p2 = ring[(c + 192u + (uint32_t)lane) & (SHC_RG - 1u)];
                    const uint32_t v1 = r1 & mask, v2 = r2 & mask;
                    uint64_t bal1 = __ballot(v1 <= i);
                    uint32_t n1 = (uint32_t)__popcll(bal1);
                    uint32_t i2 = i - n1;
                    uint64_t bal2 = __ballot((int)v2 <= (int)i2);
                    int rk1 = sh_rank(bal1), rk2 = sh_rank(bal2);
                    if (__builtin_expect((bal1 & __ballot((int)v1 > (int)i - rk1)) != 0ull, 0)) {
                        settle(bal1, rk1, v1, i);
                        n1 = (uint32_t)__popcll(bal1);
                        i2 = i - n1;
                        bal2 = __ballot((int)v2 <= (int)i2);
                        rk2 = sh_rank(bal2);
                    }
                    if (__builtin_expect((bal2 & __ballot((int)v2 > (int)i2 - rk2)) != 0ull, 0)) settle(bal2, rk2, v2, i2);
                    const uint32_t n2 = (uint32_t)__popcll(bal2);
                    const int i_new = (int)i2 - (int)n2;
                    if (__builtin_expect(i_new >= lowpow, 1)) {
                        wait_room(fill + n1 + n2);
                        const uint32_t a1 = (int)v1 <= (int)i - rk1 ? ((fill + (uint32_t)rk1) & (SHC_SQ - 1u)) : SHC_SQ + (uint32_t)lane;
                        const uint32_t a2 = (int)v2 <= (int)i2 - rk2 ? ((fill + n1 + (uint32_t)rk2) & (SHC_SQ - 1u)) : SHC_SQ + (uint32_t)lane;
                        jq[a1] = v1;
                        jq[a2] = v2;
                        fill += n1 + n2;
                        sh_st(ctrl + SH_FILL, fill);
                        i = (uint32_t)i_new;
                        c += 128u;
                        r1 = p1;
                        r2 = p2;
                    } else {
                        const uint64_t lowm = __ballot((int)i - rk1 < lowpow);
                        const uint64_t below = (lowm & (0ull - lowm)) - 1ull;
                        const uint64_t acc = bal1 & below;
                        const uint32_t na = (uint32_t)__popcll(acc);
                        wait_room(fill + na);
                        jq[((acc >> lane) & 1ull) ? ((fill + (uint32_t)rk1) & (SHC_SQ - 1u)) : SHC_SQ + (uint32_t)lane] = v1;
                        fill += na;
                        sh_st(ctrl + SH_FILL, fill);
                        i -= na;
                        c += (uint32_t)__popcll(below);
                        if ((int)i < lowpow && i >= 1u) {
                            mask = 0xffffffffu >> __builtin_clz(i);
                            lowpow = (int)((mask >> 1) + 1u);
                        }
                        wait_draws(c + 128u);
                        r1 = ring[(c + (uint32_t)lane) & (SHC_RG - 1u)];
                        r2 = ring[(c + 64u + (uint32_t)lane) & (SHC_RG - 1u)];
                    }
                    if (c - c_pub >= 256u) {  // (a store every iteration instead: no difference here, 0.445 s either way)
                        c_pub = c;
                        sh_st(ctrl + SH_CPUB, c);
                    }
                }
                sh_st(ctrl + SH_DONE, 1u);
            } else {
                // ---------------- A: the chunks, top-down
                uint32_t done = 0, fill = 0;
#ifdef SHC_PROF
                uint64_t pf_i = 0, pf_ii = 0, pf_io = 0, pf_wait = 0, pf_t = __builtin_amdgcn_s_memtime();
#define SHC_PH(x) { const uint64_t _n = __builtin_amdgcn_s_memtime(); x += _n - pf_t; pf_t = _n; }
#else
#define SHC_PH(x)
#endif
                for (int32_t cc = (int32_t)K - 1; cc >= 0; cc--) {
                    const uint32_t c = (uint32_t)cc, lo = c * CB, hi = lo + CB < n ? lo + CB : n, m = hi - lo;
                    // identity: position lo + k holds the record of local row lo + k (its digest comes in sequentially: 4 B per row)
                    for (uint32_t k0 = (uint32_t)lane; k0 < m; k0 += 1024u) {
                        uint32_t dv[16];
#pragma unroll
                        for (int u = 0; u < 16; u++) {
                            const uint32_t k = k0 + 64u * (uint32_t)u;
                            dv[u] = (k < m && !plain) ? dsrc[lo + k] : 0u;
                        }
#pragma unroll
                        for (int u = 0; u < 16; u++) {
                            const uint32_t k = k0 + 64u * (uint32_t)u, loc = lo + k;
                            if (k < m) {
                                xd[k] = PLAIN ? loc : plain ? base_val + loc : (dv[u] | shc_hi_bits(loc >> loc_bits));  // (PLAIN: the base goes on at the stores)
                                if (!PLAIN) xl[k] = loc & ((1u << loc_bits) - 1u);
                            }
                        }
                    }
                    SHC_PH(pf_io);
                    // ---- phase I: the messages of the steps above, in list order.  Eight groups of 64 are in flight (one per register set).
                    {
                        const uint32_t cnt = sh_ld(mcnt + c), mo = sh_ld(moff + c);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (the pool was another chain's a moment ago: nothing stale from the L1)
                        const uint64_t *ml = m64 + mo;
                        const uint32_t *mi = m32 + mo;
                        uint64_t cur[8], nxt[8];
                        uint32_t curi[8], nxti[8];
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            const uint32_t e = (uint32_t)q * 64u + (uint32_t)lane;
                            cur[q] = e < cnt ? ml[e] : 0ull;
                            curi[q] = (e < cnt && !PLAIN) ? mi[e] : 0u;
                        }
                        for (uint32_t b0 = 0; b0 < cnt; b0 += 512u) {
#pragma unroll
                            for (int q = 0; q < 8; q++) {
                                const uint32_t e = b0 + 512u + (uint32_t)q * 64u + (uint32_t)lane;
                                nxt[q] = e < cnt ? ml[e] : 0ull;
                                nxti[q] = (e < cnt && !PLAIN) ? mi[e] : 0u;
                            }
#pragma unroll
                            for (int q = 0; q < 8; q++) {
                                const uint32_t g0 = b0 + (uint32_t)q * 64u;
                                if (g0 >= cnt) break;
                                const bool in = g0 + (uint32_t)lane < cnt;
                                const uint64_t msg = cur[q];
                                const uint32_t vd = PLAIN ? (uint32_t)msg & 0x7fffffu : (uint32_t)msg, vl = (uint32_t)(msg >> 32) & 0xffffu;
                                const uint32_t isrc = PLAIN ? (uint32_t)(msg >> (23u + JB)) : curi[q];
                                const uint32_t adr = !in ? CB + (uint32_t)lane : PLAIN ? (uint32_t)(msg >> 23) & (CB - 1u) : (uint32_t)(msg >> 48);
                                // One exchange per lane and word puts the message's record there and takes what was there.  Lanes of a group
                                // with the same j are served in ascending lane order -- the list's order -- by the LDS itself (the property
                                // offsim_selftest_lds_atomic_order checks): no tags, no pieces.
                                // (the slot of the reply in its list is taken in the same LDS round trip)
                                const uint32_t cs = isrc >> JB;
                                const uint32_t od = shc_xchg(xd + adr, vd), ol = PLAIN ? 0u : shc_xchg(xl + adr, vl);
                                const uint32_t slot = shc_take(in ? rcnt + cs : ctrl + 15);
                                if (in)  // what was there is the final entry of position isrc: the reply
                                    rpool[(uint64_t)cs * CB + slot] = (uint64_t)od | ((uint64_t)ol << 32) | ((uint64_t)(isrc & (CB - 1u)) << 48);
                            }
#pragma unroll
                            for (int q = 0; q < 8; q++) {
                                cur[q] = nxt[q];
                                curi[q] = nxti[q];
                            }
                        }
                    }
                    SHC_PH(pf_i);
                    // ---- phase II: the chunk's own steps hi-1 .. max(lo, 1), 64 at a time (lane l: step i_top - l)
                    const uint32_t lo_step = lo ? lo : 1u;
                    uint32_t i_top = hi - 1u;
                    while (i_top >= lo_step && hi > lo_step) {
                        const uint32_t left = i_top - lo_step + 1u, cnt = left < 64u ? left : 64u;
                        if (fill - done < cnt) {
                            SHC_PH(pf_ii);
                            uint32_t polls = 0;
                            while (fill - done < cnt) {
                                fill = sh_ld(ctrl + SH_FILL);
                                if (fill - done < cnt) {
                                    shuf_bound(ctrl, polls);
                                    __builtin_amdgcn_s_sleep(1);
                                }
                            }
                            SHC_PH(pf_wait);
                        }
                        const uint32_t v = jq[(done + (uint32_t)lane) & (SHC_SQ - 1u)];
                        done += cnt;
                        sh_st(ctrl + SH_TAIL, done);  // issued after the read: the entries may be overwritten
                        const bool in = (uint32_t)lane < cnt;
                        const uint32_t il = i_top - (uint32_t)lane;  // (meaningless beyond cnt)
                        const bool intl = in && v >= lo, ext = in && v < lo;
                        const uint32_t i_low = i_top - cnt;  // the group's steps are i_top .. i_low + 1
                        const uint32_t pa = in ? il - lo : CB + (uint32_t)lane;
                        // (one LDS round trip: the record at il, and for a partner in a lower chunk the slot of the message in that chunk's list --
                        // in step order: ascending lanes -- and the list's bounds)
                        const uint32_t d = ext ? v >> JB : 0u;
                        uint32_t ad = xd[pa], al = PLAIN ? 0u : xl[pa];
                        const uint32_t mo = moff[d], me = moff[d + 1u];
                        const uint32_t slot = shc_take(ext ? mcnt + d : ctrl + 15);
                        if (__ballot(intl) != 0ull) {  // (a group whose partners all lie below the chunk only sends)
                            // Swaps inside the chunk: the record at il goes to the partner's entry by an exchange, what was there comes back to
                            // il.  Lanes with the SAME partner need nothing more (the LDS serves them in lane order = step order: each gets what
                            // the one before it put there).  Only a partner that is the position of a LATER step of the group splits the
                            // group: that step has to read its entry after this swap.
                            const uint64_t confl = __ballot(intl && v < il && v > i_low);
                            const uint32_t adr = intl ? v - lo : CB + (uint32_t)lane;
                            if (__builtin_expect(confl == 0ull, 1)) {
                                const uint32_t bd = shc_xchg(xd + adr, ad), bl = PLAIN ? 0u : shc_xchg(xl + adr, al);
                                if (intl) {  // (a self-swap, v == il, gets its own record back)
                                    xd[pa] = bd;
                                    if (!PLAIN) xl[pa] = bl;
                                }
                            } else {
                                uint64_t cuts = 0, cf = confl;
                                while (cf) {  // cut in front of the lane that owns the step at the partner's position
                                    cuts |= 1ull << (i_top - sh_rfl((uint32_t)__builtin_amdgcn_readlane((int)v, (int)sh_ff1(cf))));
                                    cf &= cf - 1ull;
                                }
                                cuts &= sh_lowmask(cnt) & ~1ull;
                                const uint32_t n_pieces = (uint32_t)__popcll(cuts) + 1u;
                                const uint32_t pid = (uint32_t)sh_rank(cuts) + (uint32_t)((cuts >> lane) & 1ull);
                                for (uint32_t pc = 0; pc < n_pieces; pc++)
                                    if (pid == pc && in) {
                                        ad = xd[pa];
                                        if (!PLAIN) al = xl[pa];
                                        if (intl) {
                                            const uint32_t bd = shc_xchg(xd + adr, ad), bl = PLAIN ? 0u : shc_xchg(xl + adr, al);
                                            xd[pa] = bd;
                                            if (!PLAIN) xl[pa] = bl;
                                        }
                                    }
                            }
                        }
                        if (ext) {  // the partner lies in a lower chunk: (i, j, the record at i) goes to that chunk's list
                            if (slot < me - mo) {
                                if (PLAIN) {
                                    m64[mo + slot] = (uint64_t)ad | ((uint64_t)(v & (CB - 1u)) << 23) | ((uint64_t)il << (23u + JB));
                                } else {
                                    m64[mo + slot] = (uint64_t)ad | ((uint64_t)al << 32) | ((uint64_t)(v & (CB - 1u)) << 48);
                                    m32[mo + slot] = il;
                                }
                            } else {
                                ctrl[SH_ABORT] = 2u;
                            }
                        }
                        i_top -= cnt;  // (never below lo - 1, and chunk 0 ends at step 1: no wrap)
                        if (!plain && c > 0u && i_top - lo == CB / 4u - 1u) {  // a quarter of the chunk to go: bring the next chunk's digests into L2
                            // (one load per 64-byte sector, as LDS-DMA into the lanes' scratch entries: no register waits for the data)
                            const uint32_t sc_a = sh_rfl((uint32_t)(uintptr_t)(xd + CB));
#pragma unroll
                            for (uint32_t u = 0; u < CB / 1024u; u++) shc_touch(dsrc + (lo - CB) + ((uint32_t)lane + 64u * u) * 16u, sc_a);
                        }
                    }
                    SHC_PH(pf_ii);
                    if (__builtin_expect(sh_ld(ctrl + SH_ABORT) == 2u, 0)) {  // a list overflowed
                        if (lane == 0) atomicOr(&g_async_fault, OFFSIM_FAULT_SHUFFLE);
                        ctrl[SH_ABORT] = 1u;
                        __builtin_amdgcn_endpgm();
                    }
                    // the chunk goes out in the streams' own layout (positions that wait for a reply hold what they sent)
                    for (uint32_t k0 = (uint32_t)lane; k0 < m; k0 += 512u) {
                        uint32_t vd[8], vl[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            const uint32_t k = k0 + 64u * (uint32_t)u;
                            vd[u] = xd[k < m ? k : 0u];
                            vl[u] = PLAIN ? 0u : xl[k < m ? k : 0u];
                        }
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            const uint32_t k = k0 + 64u * (uint32_t)u;
                            if (k < m) {
                                dg[lo + k] = PLAIN ? base_val + vd[u] : vd[u];
                                if (!PLAIN) shc_loc_store(lc, lo + k, vl[u], loc_bits);
                            }
                        }
                    }
                    SHC_PH(pf_io);
                }
#ifdef SHC_PROF
                if (w == 0u && lane == 0) {
                    uint64_t *o = (uint64_t *)(counter + 16);
                    o[0] = pf_i, o[1] = pf_ii, o[2] = pf_io, o[3] = pf_wait;
                }
#endif
            }
        } else if (threadIdx.x == 0) {  // a state with a single row
            dg[0] = plain ? base_val : dsrc[0];  // (a single row)
            if (!PLAIN) shc_loc_store(lc, 0u, 0u, loc_bits);
        }
        __syncthreads();
#ifdef SHC_PROF
        const uint64_t pf_f0 = __builtin_amdgcn_s_memtime();
#endif
        // ---- the replies: every chunk that sent messages once more, its replies scattered into it in LDS.  Eight independent loads per
        // thread and turn.
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (the chunks and replies were stored by wavefront A: nothing stale from the L1)
        for (uint32_t c = 1; c < K; c++) {  // (chunk 0 has no partner below it)
            const uint32_t lo = c * CB, hi = lo + CB < n ? lo + CB : n, m = hi - lo;
            const uint32_t cnt = rcnt[c];
            if (cnt == 0u) continue;
            for (uint32_t k0 = threadIdx.x; k0 < m; k0 += 2048u) {
                uint32_t vd[8];
                uint32_t vl[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t k = k0 + 256u * (uint32_t)u;
                    vd[u] = k < m ? dg[lo + k] : 0u;
                    vl[u] = (k < m && !PLAIN) ? shc_loc_load(lc, lo + k, loc_bits) : 0u;
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t k = k0 + 256u * (uint32_t)u;
                    if (k < m) {
                        xd[k] = vd[u];
                        if (!PLAIN) xl[k] = vl[u];
                    }
                }
            }
            __syncthreads();
            const uint64_t *rl = rpool + (uint64_t)c * CB;
            for (uint32_t e0 = threadIdx.x; e0 < cnt; e0 += 2048u) {
                uint64_t rep[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t e = e0 + 256u * (uint32_t)u;
                    rep[u] = e < cnt ? rl[e] : 0ull;
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint32_t e = e0 + 256u * (uint32_t)u;
                    if (e < cnt) {
                        xd[(uint32_t)(rep[u] >> 48)] = PLAIN ? base_val + (uint32_t)rep[u] : (uint32_t)rep[u];
                        if (!PLAIN) xl[(uint32_t)(rep[u] >> 48)] = (uint32_t)(rep[u] >> 32) & 0xffffu;
                    }
                }
            }
            __syncthreads();
            for (uint32_t k = threadIdx.x; k < m; k += 256u) {
                dg[lo + k] = xd[k];
                if (!PLAIN) shc_loc_store(lc, lo + k, xl[k], loc_bits);
            }
            __syncthreads();
        }
#ifdef SHC_PROF
        if (w == 0u && threadIdx.x == 0) ((uint64_t *)(counter + 16))[4] = __builtin_amdgcn_s_memtime() - pf_f0;
#endif
    }
}

// The chains with more than `above` rows -- states, and "state n_slots" = the init queue (N0 rows) -- longest first (the persistent
// workgroups take them in that order), and the work counter.
__global__ void __launch_bounds__(256) k_chunk_worklist(const uint32_t *__restrict__ seg_off, int32_t n_slots, uint32_t n0, uint32_t above,
                                                        uint32_t *__restrict__ work_seg, uint32_t *__restrict__ n_work_seg, uint32_t *__restrict__ counter) {
    __shared__ uint32_t cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    auto len_of = [&](int32_t q) -> uint32_t { return q == n_slots ? n0 : seg_off[q + 1] - seg_off[q]; };
    for (int32_t s = (int32_t)threadIdx.x; s <= n_slots; s += 256) {
        const uint32_t len = len_of(s);
        if (len <= above) continue;
        uint32_t rank = 0;  // chains that come first: longer ones, and equally long ones with a smaller index
        for (int32_t q = 0; q <= n_slots; q++) {
            const uint32_t lq = len_of(q);
            rank += (lq > len || (lq == len && q < s)) ? 1u : 0u;
        }
        work_seg[rank] = (uint32_t)s;
        atomicAdd(&cnt, 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        n_work_seg[0] = cnt;
        counter[0] = 0u;
    }
}

}  // namespace offsim
