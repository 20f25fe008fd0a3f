// shuffle_wave.hpp -- exact wave-parallel Fisher-Yates for PSRS.reset_sampler (psrs.py:22-23, 29-30).
//
// NumPy's Generator.shuffle of a list is   for i = n-1 .. 1:  j = random_interval(i);  swap(x[i], x[j])
// with random_interval = masked rejection on buffered 32-bit halves of PCG64 outputs.  The chain looks strictly
// sequential, and the first kernel ran it that way (one lane per chain, random 4-byte swaps in HBM: two 64-byte
// sectors moved per swap, ~2e10 swaps/s at the random-sector limit of the memory system).  But nothing in it
// depends on the DATA except the swaps themselves:
//   - which draws are accepted, the step i each accepted draw serves and its partner j are functions of the random
//     stream only, and 64 consecutive draws can be classified at once: draw l is accepted iff v_l <= i - (#accepts
//     before l).  Start from the optimistic set {v_l <= i}, take prefix counts with ballot/mbcnt, and strike the
//     first lane whose test fails (its prefix is exact, so it is a true reject) until none fails: usually zero
//     rounds, since a lane can only fail when v_l lies within 64 of i;
//   - 64 consecutive swaps commute unless two of them touch a common position.  A partner that lies inside the
//     batch's own range of i (j_l == i_m, m later) is found by arithmetic and becomes a cut in front of lane m; two
//     equal partners are found when the swaps are applied, by writing lane-id tags to the partner positions and
//     reading them back.  Swaps are applied segment by segment in lane order, so the result is bit-identical to the
//     sequential chain.
// So one workgroup owns one chain, keeps the queue segment in LDS (16-bit entries, <= 65536 rows: the whole chain
// runs at LDS latency and the only HBM traffic is the final coalesced write of the permutation), and splits the
// work over four wavefronts connected by LDS queues:
//     G  (two of them, alternate blocks) raw 32-bit draws: PCG64 jump-ahead, one 64-bit output per lane  -> ring
//     C  classifies 2 x 64 draws per iteration: accepted masks, (i, j) per lane, cut masks                -> records
//     A  applies the swaps of each record to the segment
// Segments that do not fit (the init queue of a big log, states with > 65536 rows) use the same roles with the
// segment left in global memory (32-bit entries, in place).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pcg64_dev.hpp"

namespace offsim {

#define SHUF_RG 2048u  // raw draws in the ring (power of two, multiple of 128)
#define SHUF_QB 16u    // batch records in flight (power of two)
#define SHUF_CAP16 65472u  // rows of a segment kept in LDS (16-bit indices, 64 dummy entries behind it)
enum { SH_GEN0 = 0, SH_GEN1 = 1, SH_CPUB = 2, SH_QHEAD = 3, SH_QTAIL = 4, SH_DONE = 5 };  // words of the control block

// explicit LDS address space: keeps every queue / segment access a ds_* instruction (a generic pointer would make
// them flat_* operations, which also tie up the vector-memory counter)
typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;
typedef __attribute__((address_space(3))) volatile uint16_t lds_vu16;
typedef __attribute__((address_space(3))) volatile uint64_t lds_vu64;
typedef __attribute__((address_space(3))) volatile unsigned char lds_vu8;
typedef uint32_t sh_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t sh_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) volatile sh_u32x4 lds_vu32x4;
typedef __attribute__((address_space(3))) volatile sh_u32x2 lds_vu32x2;

__device__ __forceinline__ uint32_t sh_rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane(v); }  // (readfirstlane returns int)
__device__ __forceinline__ uint32_t sh_ld(lds_vu32 *p) { return sh_rfl(*p); }
__device__ __forceinline__ void sh_st(lds_vu32 *p, uint32_t v) { *p = v; }
__device__ __forceinline__ uint64_t sh_lowmask(uint32_t k) { return k >= 64u ? ~0ull : ((1ull << k) - 1ull); }
__device__ __forceinline__ uint32_t sh_ff1(uint64_t m) { return (uint32_t)__ffsll((unsigned long long)m) - 1u; }
__device__ __forceinline__ int sh_rank(uint64_t m) {  // set bits of m in front of this lane
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// LDS layout: [ctrl 16 w][ring RG w][record headers QB x 4 w][records QB x 128 w][segment]
constexpr uint32_t shuf_fixed_lds_bytes() { return 4u * (16u + SHUF_RG + SHUF_QB * 4u + SHUF_QB * 128u); }

// LDS16 = true : segments with 1 <= n <= cap16 rows, kept in LDS as 16-bit local indices
// LDS16 = false: segments with n > cap16 rows, shuffled in place in global memory (32-bit)
template <bool LDS16>
__global__ void __launch_bounds__(256)
    k_shuffle_wave(const uint32_t *__restrict__ seg_off, int32_t n_slots, int64_t N, int64_t N0, const uint64_t *__restrict__ seeds,
                   int32_t n_perm, uint32_t *__restrict__ perm, uint32_t *__restrict__ init_perm, uint32_t cap16, int dbg_mode) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    lds_vu32 *ctrl = (lds_vu32 *)lds_raw;
    lds_vu32 *ring = ctrl + 16;
    lds_vu32 *qhdr = ring + SHUF_RG;       // {accepted mask lo, hi, cut mask lo, hi} per record
    lds_vu32 *qrec = qhdr + SHUF_QB * 4u;  // per lane {position i, partner j}: byte offsets (LDS16) or indices
    lds_vu16 *x16 = (lds_vu16 *)(qrec + SHUF_QB * 128u);
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
    if (LDS16 ? (n > cap16) : (n <= cap16)) return;
    volatile uint32_t *x32 = (volatile uint32_t *)xg;
    const uint32_t dummy_idx = ((n + 1u) & ~1u) + (uint32_t)lane;  // LDS16: this lane's private entry behind the segment (< 65536)
    const uint32_t dummy_pk = (dummy_idx << 16) | dummy_idx;

    if (threadIdx.x < 16u) ctrl[threadIdx.x] = 0;
    if (LDS16) {  // identity, two entries per lane and store
        __attribute__((address_space(3))) uint32_t *xw = (__attribute__((address_space(3))) uint32_t *)x16;
        for (uint32_t k = threadIdx.x; 2u * k < n; k += 256u) xw[k] = ((2u * k + 1u) << 16) | (2u * k);
    } else {
        for (uint32_t k = threadIdx.x; k < n; k += 256u) xg[k] = base_val + k;
    }
    __syncthreads();

    // Queue protocol.  All queues live in LDS, every role is one wavefront, and the LDS unit executes the DS
    // instructions of a wavefront in issue order: data written before a counter is visible before the counter, and a
    // read issued after a counter was seen comes after the data.  So the counters are plain volatile words (the
    // volatile qualifier keeps the compiler from reordering them) and no s_waitcnt is spent on publishing.
    if (n >= 2) {
        if (wave == 0 || wave == 3) {
            // ---------------- G (two wavefronts, alternate blocks of 128 draws): raw draws.  In block k lane l owns 64-bit
            // output 64*k + l = 32-bit draws 2*(64*k + l) and +1 (next_uint32 hands out the low half first, then the
            // buffered high half).
            const uint32_t g = wave == 0 ? 0u : 1u;
            const PcgInit p = pcg_seed(seeds[r]);
            const Jump j128 = pcg_jump(p.inc, 128);
            U128 st = pcg_apply(pcg_jump(p.inc, 64ull * g + (uint64_t)lane + 1), p.state);
            uint32_t blk = g, done_blocks = 0, cpub = 0;  // blk = index of the block this wavefront writes next
            for (;;) {
                bool stop = false;
                while ((blk + 1u) * 128u - cpub > SHUF_RG) {
                    if (sh_ld(ctrl + SH_DONE)) {
                        stop = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    cpub = sh_ld(ctrl + SH_CPUB);
                }
                if (stop) break;
                const uint64_t o = pcg_output(st);
                st = pcg_apply(j128, st);
                const uint32_t idx = (blk * 128u + 2u * (uint32_t)lane) & (SHUF_RG - 1u);
                *(lds_vu64 *)(ring + idx) = o;  // low half first
                blk += 2u;
                done_blocks++;
                sh_st(ctrl + (g ? SH_GEN1 : SH_GEN0), done_blocks);
            }
        } else if (wave == 1) {
            // ---------------- C: classify, 2 x 64 draws per iteration.  The second batch starts from i2 = i - accepts of
            // the first, so the wavefront has two independent instruction streams; conflicts do not stop a batch (they
            // become cut masks for A), so every iteration consumes its 128 draws except at a mask boundary.
            uint32_t i = n - 1u, c = 0, avail = 0, qh = 0, qt = 0, c_pub = 0;
            uint32_t mask = 0xffffffffu >> __builtin_clz(i);
            int lowpow = (int)((mask >> 1) + 1u);  // steps below this index use the next smaller mask
            auto wait_draws = [&](uint32_t upto) {
                while (upto > avail) {
                    const uint64_t gg = *(lds_vu64 *)(ctrl + SH_GEN0);
                    const uint32_t g0 = sh_rfl((uint32_t)gg), g1 = sh_rfl((uint32_t)(gg >> 32));
                    avail = 128u * (g0 <= g1 ? 2u * g0 : 2u * g1 + 1u);  // contiguous blocks
                    if (upto > avail) __builtin_amdgcn_s_sleep(1);
                }
            };
            auto wait_room = [&](uint32_t slots) {
                while (qh + slots - qt > SHUF_QB) {
                    qt = sh_ld(ctrl + SH_QTAIL);
                    if (qh + slots - qt > SHUF_QB) __builtin_amdgcn_s_sleep(1);
                }
            };
            // strike optimistic accepts that do not hold, first one first, until all hold
            auto settle = [&](uint64_t &bal, int &il, uint32_t v, uint32_t ib) {
                uint64_t f = bal & __ballot((int)v > il);
                while (f) {
                    bal &= ~(1ull << sh_ff1(f));
                    il = (int)ib - sh_rank(bal);
                    f = bal & __ballot((int)v > il);
                }
            };
            // partner inside the batch's own range of i: the later swap (the lane whose i equals that partner) must see the
            // earlier one -> cut in front of it
            auto cut_mask = [&](uint64_t bal, int il, uint32_t v, uint32_t ib) -> uint64_t {
                uint64_t confl = bal & __ballot((int)v < il) & __ballot((int)v > (int)ib - (int)__popcll(bal));
                uint64_t cuts = 0;
                while (confl) {
                    const int vf = (int)__builtin_amdgcn_readlane((int)v, (int)sh_ff1(confl));
                    cuts |= bal & __ballot(il == vf);
                    confl &= confl - 1ull;
                }
                return cuts;
            };
            // LDS16 records are mask-free: one word (i << 16 | j) per lane, and a lane that has no swap in the record swaps a
            // private dummy entry behind the segment with itself; a batch with cuts is written as one record per piece.  So A
            // runs every record on all 64 lanes without masks or headers.  After settling, the accepted lanes are exactly
            // those with v <= il, so the common case needs no lane mask here either.  (The in-place global-memory variant
            // keeps masks: header {accepted, cuts}.)
            auto put16 = [&](uint32_t packed) {
                while (qh + 1u - qt > SHUF_QB) {
                    qt = sh_ld(ctrl + SH_QTAIL);
                    if (qh + 1u - qt > SHUF_QB) __builtin_amdgcn_s_sleep(1);
                }
                qrec[(qh & (SHUF_QB - 1u)) * 128u + (uint32_t)lane] = packed;
                qh++;
                sh_st(ctrl + SH_QHEAD, qh);
            };
            auto emit = [&](int il, uint32_t v, uint64_t accm, uint64_t cuts, bool whole) {  // whole: accm = every lane with v <= il
                if (LDS16) {
                    const uint32_t pk = ((uint32_t)il << 16) | v;
                    if (__builtin_expect(cuts == 0ull && whole, 1)) {
                        put16((int)v <= il ? pk : dummy_pk);
                    } else {
                        uint64_t rem = accm;
                        while (rem) {
                            uint64_t seg = rem;
                            const uint64_t cm = cuts & rem & (rem - 1ull);  // cuts in front of lanes other than the first remaining one
                            if (cm) seg = rem & ((cm & (0ull - cm)) - 1ull);
                            put16(((seg >> lane) & 1ull) ? pk : dummy_pk);
                            rem &= ~seg;
                        }
                    }
                } else {
                    const uint32_t slot = qh & (SHUF_QB - 1u);
                    sh_u32x2 rv;
                    rv.x = (uint32_t)il;
                    rv.y = v;
                    *(lds_vu32x2 *)(qrec + slot * 128u + 2u * (uint32_t)lane) = rv;
                    sh_u32x4 hv;
                    hv.x = (uint32_t)accm;
                    hv.y = (uint32_t)(accm >> 32);
                    hv.z = (uint32_t)cuts;
                    hv.w = (uint32_t)(cuts >> 32);
                    *(lds_vu32x4 *)(qhdr + slot * 4u) = hv;
                    qh++;
                    sh_st(ctrl + SH_QHEAD, qh);
                }
            };
            wait_draws(128u);
            uint32_t r1 = ring[(uint32_t)lane], r2 = ring[64u + (uint32_t)lane];
            while (i >= 1u) {
                wait_draws(c + 256u);  // this pair and the prefetch of the next
                const uint32_t p1 = ring[(c + 128u + (uint32_t)lane) & (SHUF_RG - 1u)];
                const uint32_t p2 = ring[(c + 192u + (uint32_t)lane) & (SHUF_RG - 1u)];
                const uint32_t v1 = r1 & mask, v2 = r2 & mask;
                uint64_t bal1 = __ballot(v1 <= i);  // optimistic: accepted if no earlier lane of the batch had been accepted
                uint32_t i2 = i - (uint32_t)__popcll(bal1);
                uint64_t bal2 = __ballot((int)v2 <= (int)i2);
                int il1 = (int)i - sh_rank(bal1);   // the step a lane's draw serves
                int il2 = (int)i2 - sh_rank(bal2);
                if (__builtin_expect((bal1 & __ballot((int)v1 > il1)) != 0ull, 0)) {
                    settle(bal1, il1, v1, i);
                    i2 = i - (uint32_t)__popcll(bal1);
                    bal2 = __ballot((int)v2 <= (int)i2);
                    il2 = (int)i2 - sh_rank(bal2);
                }
                if (__builtin_expect((bal2 & __ballot((int)v2 > il2)) != 0ull, 0)) settle(bal2, il2, v2, i2);
                const int i_new = (int)i2 - __popcll(bal2);
                if (__builtin_expect(i_new >= lowpow, 1)) {
                    const uint64_t cuts1 = cut_mask(bal1, il1, v1, i), cuts2 = cut_mask(bal2, il2, v2, i2);
                    if (!LDS16) wait_room(2u);
                    if (bal1) emit(il1, v1, bal1, cuts1, true);
                    if (bal2) emit(il2, v2, bal2, cuts2, true);
                    i = (uint32_t)i_new;
                    c += 128u;
                    r1 = p1;
                    r2 = p2;
                } else {
                    // a mask boundary (or the end of the chain) inside the pair: batch 1 only, and only the draws -- accepted
                    // or not -- of steps at or above the boundary
                    const uint64_t lowm = __ballot(il1 < lowpow);
                    const uint64_t below = (lowm & (0ull - lowm)) - 1ull;  // lanes in front of the first such draw (all if none)
                    const uint64_t acc = bal1 & below;
                    if (acc) {
                        const uint64_t cuts = cut_mask(acc, il1, v1, i);
                        if (!LDS16) wait_room(1u);
                        emit(il1, v1, acc, cuts, false);
                    }
                    i -= (uint32_t)__popcll(acc);
                    c += (uint32_t)__popcll(below);
                    if ((int)i < lowpow && i >= 1u) {  // crossed a power of two: the mask shrinks
                        mask = 0xffffffffu >> __builtin_clz(i);
                        lowpow = (int)((mask >> 1) + 1u);
                    }
                    wait_draws(c + 128u);
                    r1 = ring[(c + (uint32_t)lane) & (SHUF_RG - 1u)];
                    r2 = ring[(c + 64u + (uint32_t)lane) & (SHUF_RG - 1u)];
                }
                if (c - c_pub >= 256u) {  // the ring is 2048 draws deep: G does not need every step
                    c_pub = c;
                    sh_st(ctrl + SH_CPUB, c);
                }
            }
            if (!LDS16) {  // end marker: a record with an empty mask (LDS16: A stops when DONE is set and the queue is empty)
                wait_room(1u);
                *(lds_vu64 *)(qhdr + (qh & (SHUF_QB - 1u)) * 4u) = 0ull;
                qh++;
                sh_st(ctrl + SH_QHEAD, qh);
            }
            sh_st(ctrl + SH_DONE, 1u);
        } else {
            // ---------------- A: apply the records in order, each one segment by segment (cut masks) and, inside a
            // segment, as far as the partners are distinct (lane-id tags written to the partner positions and read back)
            uint32_t qt = 0, qh = 0;
            lds_vu8 *xb = (lds_vu8 *)x16;
            if (LDS16) {
                // mask-free records (see emit): all 64 lanes swap, lanes without a swap exchange their dummy entry with itself
                for (;;) {
                    bool fin = false;
                    while (qt == qh) {
                        qh = sh_ld(ctrl + SH_QHEAD);
                        if (qt != qh) break;
                        if (sh_ld(ctrl + SH_DONE)) {  // C sets DONE after its last record: look once more, then stop
                            qh = sh_ld(ctrl + SH_QHEAD);
                            fin = qt == qh;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (fin) break;
                    const uint32_t rv = qrec[(qt & (SHUF_QB - 1u)) * 128u + (uint32_t)lane];
                    qt++;
                    sh_st(ctrl + SH_QTAIL, qt);  // issued after the read of the record: its slot may be reused
                    if (dbg_mode == 1) continue;
                    const uint32_t il = (rv >> 15) & 0x1fffeu, v = (rv & 0xffffu) << 1;  // byte offsets
                    const uint32_t a = *(lds_vu16 *)(xb + il);
                    const uint32_t b = *(lds_vu16 *)(xb + v);
                    *(lds_vu16 *)(xb + v) = (uint16_t)lane;  // tag: two lanes with the same partner see one winner
                    const uint32_t tg = *(lds_vu16 *)(xb + v);
                    if (__builtin_expect(__ballot(tg != (uint32_t)lane) == 0ull, 1)) {
                        *(lds_vu16 *)(xb + il) = (uint16_t)b;
                        *(lds_vu16 *)(xb + v) = (uint16_t)a;
                        continue;
                    }
                    // equal partners somewhere: take the tags back and apply the record piecewise
                    *(lds_vu16 *)(xb + v) = (uint16_t)b;
                    uint64_t rem = __ballot((rv >> 16) != dummy_idx);
                    while (rem) {
                        const bool act = (rem >> lane) & 1ull;
                        uint32_t a2 = 0, b2 = 0, t2 = (uint32_t)lane;
                        if (act) {
                            a2 = *(lds_vu16 *)(xb + il);
                            b2 = *(lds_vu16 *)(xb + v);
                            *(lds_vu16 *)(xb + v) = (uint16_t)lane;
                            t2 = *(lds_vu16 *)(xb + v);
                        }
                        const uint64_t F = __ballot(act && t2 != (uint32_t)lane);
                        uint64_t proc = rem;
                        if (F) {  // stop in front of the second lane of the earliest group of equal partners
                            uint64_t Wn = 0, FF = F;
                            while (FF) {
                                Wn |= 1ull << (uint32_t)__builtin_amdgcn_readlane((int)t2, (int)sh_ff1(FF));
                                FF &= FF - 1ull;
                            }
                            const uint64_t D = F | Wn;
                            const uint64_t D2 = D & (D - 1ull);
                            proc = rem & sh_lowmask(sh_ff1(D2));
                            if (act && !((proc >> lane) & 1ull)) *(lds_vu16 *)(xb + v) = (uint16_t)b2;  // not this time: take the tag back
                        }
                        if ((proc >> lane) & 1ull) {
                            *(lds_vu16 *)(xb + il) = (uint16_t)b2;
                            *(lds_vu16 *)(xb + v) = (uint16_t)a2;
                        }
                        rem &= ~proc;
                    }
                }
            } else
            for (;;) {
                while (qt == qh) {
                    qh = sh_ld(ctrl + SH_QHEAD);
                    if (qt == qh) __builtin_amdgcn_s_sleep(1);
                }
                const uint32_t slot = qt & (SHUF_QB - 1u);
                const sh_u32x4 hv = *(lds_vu32x4 *)(qhdr + slot * 4u);
                const sh_u32x2 rv = *(lds_vu32x2 *)(qrec + slot * 128u + 2u * (uint32_t)lane);
                const uint64_t accm = ((uint64_t)sh_rfl(hv.y) << 32) | sh_rfl(hv.x);
                const uint64_t cuts = ((uint64_t)sh_rfl(hv.w) << 32) | sh_rfl(hv.z);
                if (!accm) break;
                const uint32_t il = rv.x, v = rv.y;
                qt++;
                sh_st(ctrl + SH_QTAIL, qt);  // issued after the reads of the record: its slot may be reused
                uint64_t rem = dbg_mode == 1 ? 0ull : accm;
                while (rem) {
                    uint64_t seg = rem;
                    const uint64_t cm = cuts & rem & (rem - 1ull);  // cuts in front of lanes other than the first remaining one
                    if (__builtin_expect(cm != 0ull, 0)) seg = rem & ((cm & (0ull - cm)) - 1ull);
                    const bool act = (seg >> lane) & 1ull;
                    uint32_t a = 0, b = 0, tg = (uint32_t)lane;
                    if (act) {
                        if (LDS16) {
                            a = *(lds_vu16 *)(xb + il);
                            b = *(lds_vu16 *)(xb + v);
                            *(lds_vu16 *)(xb + v) = (uint16_t)lane;  // tag: two lanes with the same partner see one winner
                            tg = *(lds_vu16 *)(xb + v);
                        } else {
                            a = x32[il];
                            b = x32[v];
                            x32[v] = (uint32_t)lane;
                            tg = x32[v];
                        }
                    }
                    const uint64_t F = __ballot(act && tg != (uint32_t)lane);
                    uint64_t proc = seg;
                    if (__builtin_expect(F != 0ull, 0)) {  // equal partners: stop in front of the second lane of the earliest group
                        uint64_t Wn = 0, FF = F;
                        while (FF) {
                            Wn |= 1ull << (uint32_t)__builtin_amdgcn_readlane((int)tg, (int)sh_ff1(FF));
                            FF &= FF - 1ull;
                        }
                        const uint64_t D = F | Wn;
                        const uint64_t D2 = D & (D - 1ull);
                        proc = seg & sh_lowmask(sh_ff1(D2));
                        if (act && !((proc >> lane) & 1ull)) {  // not this time: take the tag back
                            if (LDS16) *(lds_vu16 *)(xb + v) = (uint16_t)b;
                            else x32[v] = b;
                        }
                    }
                    if ((proc >> lane) & 1ull) {
                        if (LDS16) {
                            *(lds_vu16 *)(xb + il) = (uint16_t)b;
                            *(lds_vu16 *)(xb + v) = (uint16_t)a;
                        } else {
                            x32[il] = b;
                            x32[v] = a;
                        }
                    }
                    rem &= ~proc;
                }
            }
        }
    }
    __syncthreads();
    if (LDS16) {  // the only HBM traffic of the chain: one coalesced write of the finished permutation
        __attribute__((address_space(3))) const uint32_t *xw = (__attribute__((address_space(3))) const uint32_t *)x16;
        for (uint32_t k = threadIdx.x; 2u * k < n; k += 256u) {
            const uint32_t two = xw[k];
            xg[2u * k] = base_val + (two & 0xffffu);
            if (2u * k + 1u < n) xg[2u * k + 1u] = base_val + (two >> 16);
        }
    }
}

}  // namespace offsim
