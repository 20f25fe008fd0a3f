// shuffle_wave.hpp -- exact wave-parallel Fisher-Yates for PSRS.reset_sampler (psrs.py:22-23, 29-30).
//
// NumPy's Generator.shuffle of a list is   for i = n-1 .. 1:  j = random_interval(i);  swap(x[i], x[j])
// with random_interval = masked rejection on buffered 32-bit halves of PCG64 outputs.  The chain looks strictly
// sequential, and the first kernel ran it that way (one lane per chain, random 4-byte swaps in HBM: two 64-byte
// sectors moved per swap, ~2e10 swaps/s at the random-sector limit of the memory system).  But nothing in it
// depends on the DATA except the swaps themselves:
//   - which draws are accepted, the step i each accepted draw belongs to and its partner j are functions of the
//     random stream only, and 64 consecutive draws can be classified at once (a draw is accepted iff
//     v <= i - (#accepts before it): ballot + prefix count, re-checked exactly; the first lane whose optimistic
//     classification flips ends the batch);
//   - 64 consecutive swaps commute unless two of them touch a common position, which for n ~ 6e4 happens in a few
//     percent of the batches and is detected exactly (partner inside the batch's own i-range: arithmetic; two equal
//     partners: lane-id tags written to the partner positions and read back).  A batch is cut in front of the later
//     swap of the first conflicting pair, so the result is bit-identical to the sequential chain.
// So one workgroup owns one chain, keeps the queue segment in LDS (16-bit entries, <= 65536 rows: the whole chain
// runs at LDS latency and the only HBM traffic is the final coalesced write of the permutation), and splits the
// work over four wavefronts connected by LDS queues:
//     G  (two of them) generates the raw 32-bit draws (PCG64 jump-ahead, one 64-bit output per lane)      -> ring
//     C  classifies 64 draws per batch: accepted mask, (i, j) per lane, batch cuts           -> record queue
//     A  applies the swaps of a record to the segment (dup detection by tags, sub-batches)
// Segments that do not fit (the init queue of a big log, states with > 65536 rows) use the same three roles with the
// segment left in global memory (32-bit entries, in place).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pcg64_dev.hpp"

namespace offsim {

#define SHUF_RG 1024u  // raw draws in the ring (power of two, multiple of 128)
#define SHUF_QB 16u    // batch records in flight (power of two)
#define SHUF_CAP16 65536u
enum { SH_GEN0 = 0, SH_GEN1 = 1, SH_CPUB = 2, SH_QHEAD = 3, SH_QTAIL = 4, SH_DONE = 5 };  // words of the control block

// explicit LDS address space: keeps every queue / segment access a ds_* instruction (a generic pointer would make
// them flat_* operations, which also tie up the vector-memory counter)
typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;
typedef __attribute__((address_space(3))) volatile uint16_t lds_vu16;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

typedef __attribute__((address_space(3))) volatile uint64_t lds_vu64;

__device__ __forceinline__ uint32_t sh_ld(lds_vu32 *p) { return __builtin_amdgcn_readfirstlane(*p); }
__device__ __forceinline__ void sh_st(lds_vu32 *p, uint32_t v) { *p = v; }
__device__ __forceinline__ uint64_t sh_lowmask(uint32_t k) { return k >= 64u ? ~0ull : ((1ull << k) - 1ull); }
__device__ __forceinline__ uint32_t sh_ff1(uint64_t m) { return (uint32_t)__ffsll((unsigned long long)m) - 1u; }

constexpr uint32_t shuf_fixed_lds_bytes() { return 64u + SHUF_RG * 4u + SHUF_QB * 8u + SHUF_QB * 128u * 4u; }

// LDS16 = true : segments with 1 <= n <= cap16 rows, kept in LDS as 16-bit local indices
// LDS16 = false: segments with n > cap16 rows, shuffled in place in global memory (32-bit)
template <bool LDS16>
__global__ void __launch_bounds__(256)
    k_shuffle_wave(const uint32_t *__restrict__ seg_off, int32_t n_slots, int64_t N, int64_t N0, const uint64_t *__restrict__ seeds,
                   int32_t n_perm, uint32_t *__restrict__ perm, uint32_t *__restrict__ init_perm, uint32_t cap16, int dbg_mode) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    lds_vu32 *ctrl = (lds_vu32 *)lds_raw;
    lds_vu32 *ring = ctrl + 16;
    lds_vu32 *qhdr = ring + SHUF_RG;
    lds_vu32 *qrec = qhdr + 2 * SHUF_QB;
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

    const uint64_t T0 = __builtin_readcyclecounter();
    if (threadIdx.x < 16) ctrl[threadIdx.x] = 0;
    if (LDS16) {
        for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) x16[k] = (uint16_t)k;
    } else {
        for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) xg[k] = base_val + k;
    }
    __syncthreads();
    const uint64_t T1 = __builtin_readcyclecounter();
    uint64_t w_a = 0, w_b = 0, sec1 = 0, sec2 = 0;  // cycles spent waiting (dbg)
    uint32_t n_fast = 0, n_iter = 0;

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
                const uint64_t tw = __builtin_readcyclecounter();
                while ((blk + 1u) * 128u - cpub > SHUF_RG) {
                    if (sh_ld(ctrl + SH_DONE)) {
                        stop = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    cpub = sh_ld(ctrl + SH_CPUB);
                }
                if (stop) break;
                w_a += __builtin_readcyclecounter() - tw;
                const uint64_t o = pcg_output(st);
                st = pcg_apply(j128, st);
                const uint32_t idx = (blk * 128u + 2u * (uint32_t)lane) & (SHUF_RG - 1u);
                *(lds_vu64 *)(ring + idx) = o;  // low half first
                blk += 2u;
                done_blocks++;
                sh_st(ctrl + (g ? SH_GEN1 : SH_GEN0), done_blocks);
            }
        } else if (wave == 1) {
            // ---------------- C: classify.  Two batches of 64 draws per iteration: the second one assumes that the first
            // runs to its end (i2 = i - accepts of the first), which gives the wavefront two independent instruction
            // streams; when either batch has a stop or a conflict (about a quarter of the pairs) the first batch is
            // redone exactly and the loop restarts behind it.
            uint32_t i = n - 1u, c = 0, avail = 0, qh = 0, qt = 0, c_pub = 0;
            uint32_t mask = 0xffffffffu >> __builtin_clz(i);
            int lowpow = (int)((mask >> 1) + 1u);  // steps below this index use the next smaller mask
            auto wait_draws = [&](uint32_t upto) {
                if (upto <= avail) return;
                const uint64_t tw = __builtin_readcyclecounter();
                while (upto > avail) {
                    const uint64_t gg = *(lds_vu64 *)(ctrl + SH_GEN0);
                    const uint32_t g0 = __builtin_amdgcn_readfirstlane((uint32_t)gg), g1 = __builtin_amdgcn_readfirstlane((uint32_t)(gg >> 32));
                    avail = 128u * (g0 <= g1 ? 2u * g0 : 2u * g1 + 1u);  // contiguous blocks
                    if (upto > avail) __builtin_amdgcn_s_sleep(1);
                }
                w_a += __builtin_readcyclecounter() - tw;
            };
            auto wait_room = [&](uint32_t slots) {
                if (qh + slots - qt <= SHUF_QB) return;
                const uint64_t tw = __builtin_readcyclecounter();
                while (qh + slots - qt > SHUF_QB) {
                    qt = sh_ld(ctrl + SH_QTAIL);
                    if (qh + slots - qt > SHUF_QB) __builtin_amdgcn_s_sleep(1);
                }
                w_b += __builtin_readcyclecounter() - tw;
            };
            auto emit = [&](int il, uint32_t v, uint64_t accm) {
                const uint32_t slot = qh & (SHUF_QB - 1u);
                if (LDS16) {
                    qrec[slot * 128u + lane] = ((uint32_t)il << 16) | (v & 0xffffu);
                } else {
                    qrec[slot * 128u + lane] = (uint32_t)il;
                    qrec[slot * 128u + 64u + lane] = v;
                }
                *(lds_vu64 *)(qhdr + slot * 2u) = accm;
                qh++;
                sh_st(ctrl + SH_QHEAD, qh);
            };
            wait_draws(128u);
            uint32_t r1 = ring[(uint32_t)lane], r2 = ring[64u + (uint32_t)lane];
            while (i >= 1u) {
                const uint64_t s0 = dbg_mode >= 8 ? __builtin_readcyclecounter() : 0;
                wait_draws(c + 256u);  // this pair and the prefetch of the next
                const uint32_t p1 = ring[(c + 128u + (uint32_t)lane) & (SHUF_RG - 1u)];
                const uint32_t p2 = ring[(c + 192u + (uint32_t)lane) & (SHUF_RG - 1u)];
                const uint32_t v1 = r1 & mask, v2 = r2 & mask;
                const uint64_t bal1 = __ballot(v1 <= i);  // accepted if no earlier lane of the batch had been accepted
                const uint32_t n1 = (uint32_t)__popcll(bal1);
                const uint32_t i2 = i - n1;               // (wraps when the chain ends inside batch 1: caught by low1)
                const uint64_t bal2 = __ballot((int)v2 <= (int)i2);
                const uint32_t n2 = (uint32_t)__popcll(bal2);
                const int il1 = (int)i - (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal1, 0u));
                const int il2 = (int)i2 - (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal2, 0u));
                // il = the step a lane's draw serves if every earlier optimistic accept holds.  Trouble, per batch:
                //   an optimistic accept that does not hold (v > il);  a draw -- accepted or not -- of a step below the mask
                //   boundary or below step 1 (il < lowpow);  a partner inside the batch's own range of i (v in (i - n, il))
                const uint64_t bad1 = (bal1 & __ballot((int)v1 > il1)) | __ballot(il1 < lowpow) |
                                      (bal1 & __ballot((int)v1 < il1) & __ballot((int)v1 > (int)i - (int)n1));
                const uint64_t bad2 = (bal2 & __ballot((int)v2 > il2)) | __ballot(il2 < lowpow) |
                                      (bal2 & __ballot((int)v2 < il2) & __ballot((int)v2 > (int)i2 - (int)n2));
                const uint64_t s1 = dbg_mode >= 8 ? __builtin_readcyclecounter() : 0;
                if (dbg_mode >= 8) sec1 += s1 - s0;
                if (__builtin_expect((bad1 | bad2) == 0ull && dbg_mode != 3, 1)) {
                    n_fast++;
                    if (dbg_mode != 2) {
                        wait_room(2u);
                        if (bal1) emit(il1, v1, bal1);
                        if (bal2) emit(il2, v2, bal2);
                    }
                    i = i2 - n2;
                    c += 128u;
                    r1 = p1;
                    r2 = p2;
                } else if (dbg_mode == 3) {  // measurement only: how fast do draws arrive
                    c += 128u;
                    i -= i > 96u ? 96u : i;
                    r1 = p1;
                    r2 = p2;
                } else {
                    // exact treatment of batch 1: stop in front of the first lane whose optimistic accept does not hold (its
                    // prefix is exact, so it is a true reject and will be lane 0 of the next batch), in front of the first
                    // draw that belongs to a step below the mask boundary, and in front of the later swap of a conflicting pair
                    const uint64_t okm = bal1 & ~__ballot((int)v1 > il1);
                    const uint64_t stopm = (bal1 & ~okm) | __ballot(il1 < lowpow);
                    uint64_t below = (stopm & (0ull - stopm)) - 1ull;  // lanes in front of the first stop bit (all if none)
                    uint64_t accm = okm & below;
                    const int n_acc = __popcll(accm);
                    uint64_t confl = accm & __ballot((int)v1 < il1) & __ballot((int)v1 > (int)i - n_acc);
                    while (confl) {
                        const uint32_t l = sh_ff1(confl);
                        const int vf = (int)__builtin_amdgcn_readlane((int)v1, (int)l);
                        const uint64_t mm = accm & __ballot(il1 <= vf);
                        if (mm) {
                            below &= (mm & (0ull - mm)) - 1ull;
                            accm &= below;
                        }
                        confl &= confl - 1ull;
                        confl &= below;
                    }
                    if (accm && dbg_mode != 2) {
                        wait_room(1u);
                        emit(il1, v1, accm);
                    }
                    i -= (uint32_t)__popcll(accm);
                    c += (uint32_t)__popcll(below);
                    wait_draws(c + 128u);
                    r1 = ring[(c + (uint32_t)lane) & (SHUF_RG - 1u)];
                    r2 = ring[(c + 64u + (uint32_t)lane) & (SHUF_RG - 1u)];
                }
                if (dbg_mode >= 8) sec2 += __builtin_readcyclecounter() - s1;
                n_iter++;
                if (__builtin_expect((int)i < lowpow && i >= 1u, 0)) {  // crossed a power of two: the mask shrinks
                    mask = 0xffffffffu >> __builtin_clz(i);
                    lowpow = (int)((mask >> 1) + 1u);
                }
                if (c - c_pub >= 256u) {  // the ring is 1024 draws deep: G does not need every step
                    c_pub = c;
                    sh_st(ctrl + SH_CPUB, c);
                }
            }
            // end marker: a record with an empty mask
            wait_room(1u);
            const uint32_t slot = qh & (SHUF_QB - 1u);
            *(lds_vu64 *)(qhdr + slot * 2u) = 0ull;
            qh++;
            sh_st(ctrl + SH_QHEAD, qh);
            sh_st(ctrl + SH_DONE, 1u);
        } else {
            // ---------------- A: apply
            uint32_t qt = 0, qh = 0;
            for (;;) {
                if (qt == qh) {
                    const uint64_t tw = __builtin_readcyclecounter();
                    while (qt == qh) {
                        qh = sh_ld(ctrl + SH_QHEAD);
                        if (qt == qh) __builtin_amdgcn_s_sleep(1);
                    }
                    w_a += __builtin_readcyclecounter() - tw;
                }
                const uint32_t slot = qt & (SHUF_QB - 1u);
                const uint64_t hv = *(lds_vu64 *)(qhdr + slot * 2u);
                uint32_t il, v;
                if (LDS16) {
                    const uint32_t rv = qrec[slot * 128u + lane];
                    il = rv >> 16;
                    v = rv & 0xffffu;
                } else {
                    il = qrec[slot * 128u + lane];
                    v = qrec[slot * 128u + 64u + lane];
                }
                const uint32_t a_lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)hv), a_hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(hv >> 32));
                const uint64_t accm = ((uint64_t)a_hi << 32) | a_lo;  // (readfirstlane returns int: widen through uint32_t)
                if (!accm) break;
                qt++;
                sh_st(ctrl + SH_QTAIL, qt);  // issued after the reads of the record: its slot may be reused
                uint64_t rem = dbg_mode == 1 ? 0ull : accm;
                while (rem) {
                    const bool act = (rem >> lane) & 1ull;
                    uint32_t a = 0, b = 0, t = (uint32_t)lane;
                    if (act) {
                        if (LDS16) {
                            a = x16[il];
                            b = x16[v];
                            x16[v] = (uint16_t)lane;  // tag: two lanes with the same partner see one winner
                            t = x16[v];
                        } else {
                            a = x32[il];
                            b = x32[v];
                            x32[v] = (uint32_t)lane;
                            t = x32[v];
                        }
                    }
                    const uint64_t F = __ballot(act && t != (uint32_t)lane);
                    uint64_t proc = rem;
                    if (__builtin_expect(F != 0ull, 0)) {  // equal partners: stop in front of the second lane of the earliest group
                        uint64_t Wn = 0, FF = F;
                        while (FF) {
                            const uint32_t f = sh_ff1(FF);
                            Wn |= 1ull << (uint32_t)__builtin_amdgcn_readlane((int)t, (int)f);
                            FF &= FF - 1ull;
                        }
                        const uint64_t D = F | Wn;
                        const uint64_t D2 = D & (D - 1ull);
                        proc = rem & sh_lowmask(sh_ff1(D2));
                        if (act && !((proc >> lane) & 1ull)) {  // not this time: take the tag back
                            if (LDS16) x16[v] = (uint16_t)b;
                            else x32[v] = b;
                        }
                    }
                    if ((proc >> lane) & 1ull) {
                        if (LDS16) {
                            x16[il] = (uint16_t)b;
                            x16[v] = (uint16_t)a;
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
    const uint64_t T2 = __builtin_readcyclecounter();
    __syncthreads();
    const uint64_t T3 = __builtin_readcyclecounter();
    if (LDS16) {
        for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) xg[k] = base_val + (uint32_t)x16[k];
    }
    if (dbg_mode >= 8 && blockIdx.x == 5000 && lane == 0) {
        const uint64_t T4 = __builtin_readcyclecounter();
        printf("wave %d n=%u init %llu roles %llu barrier %llu writeout %llu wait_a %llu wait_b %llu sec1 %llu sec2 %llu fast %u iter %u\n", wave, n, (unsigned long long)(T1 - T0),
               (unsigned long long)(T2 - T1), (unsigned long long)(T3 - T2), (unsigned long long)(T4 - T3), (unsigned long long)w_a, (unsigned long long)w_b, (unsigned long long)sec1, (unsigned long long)sec2, n_fast, n_iter);
    }
}

}  // namespace offsim
