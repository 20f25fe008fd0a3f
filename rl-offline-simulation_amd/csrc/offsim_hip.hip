// offsim_hip.hip -- HIP kernels (gfx950 / CDNA4) and the C ABI of include/offsim.h.
//
// Hot path: Per-State Rejection Sampling replay loop of offsim4rl
//   PSRS.reset_sampler / reset / step / _default_reject   offsim4rl/evaluators/psrs.py:19-57
//   evalMC_psrs                                           offsim4rl/evaluators/psrs.py:241-271
// Layout: logged transitions SoA in HBM, rows grouped by from-state (CSR), so the candidates of
// one state's queue are contiguous; per-rollout queue cursors live in LDS; one wavefront (64
// lanes) simulates one rollout and tests 64 consecutive candidates of the current queue per
// iteration, lane k with PCG64 draw c+k (jump-ahead), __ballot picks the first accepted one.
//
// No CPU fallback lives here: without a HIP device every entry point returns OFFSIM_EHIP.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "offsim.h"
#include "discount.hpp"
#include "pcg64_dev.hpp"
#include "philox_dev.hpp"  // rocRAND's Philox4x32-10 on the device (the OFFSIM_STREAM_PHILOX provider)

extern "C" int offsim_lds_order_ok(void);
#include "shuffle_wave.hpp"
#include "shuffle_chunk.hpp"

using namespace offsim;

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, const char *detail = "") {
    snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}
#define HIP_TRY(expr)                                                        \
    do {                                                                     \
        hipError_t _e = (expr);                                              \
        if (_e != hipSuccess) return fail(OFFSIM_EHIP, #expr ": %s", hipGetErrorString(_e)); \
    } while (0)
#define LAUNCH_CHECK()                                                       \
    do {                                                                     \
        hipError_t _e = hipGetLastError();                                   \
        if (_e != hipSuccess) return fail(OFFSIM_EHIP, "kernel launch: %s", hipGetErrorString(_e)); \
    } while (0)

extern "C" const char *offsim_last_error(void) { return g_err; }
extern "C" int offsim_version(void) { return 100; }
extern "C" int offsim_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(OFFSIM_EHIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

#define WAVE 64

// ------------------------------------------------------------------------------------------------
// Table construction: stable group-by-state (counting sort with per-chunk histograms)
// ------------------------------------------------------------------------------------------------
#define GRP_CHUNK 2048  // rows ranked by one wavefront, 64 at a time, in buffer order

// pass 1: hist[chunk][slot] = rows of `slot` inside the chunk
__global__ void k_group_hist(const int32_t *__restrict__ slot, int64_t N, int32_t n_slots, uint32_t *__restrict__ hist) {
    extern __shared__ uint32_t lds_cnt[];
    for (int s = threadIdx.x; s < n_slots; s += blockDim.x) lds_cnt[s] = 0;
    __syncthreads();
    int64_t base = (int64_t)blockIdx.x * GRP_CHUNK;
    for (int i = threadIdx.x; i < GRP_CHUNK; i += blockDim.x) {
        int64_t row = base + i;
        if (row < N) atomicAdd(&lds_cnt[slot[row]], 1u);
    }
    __syncthreads();
    uint32_t *h = hist + (int64_t)blockIdx.x * n_slots;
    for (int s = threadIdx.x; s < n_slots; s += blockDim.x) h[s] = lds_cnt[s];
}

// pass 2a: per slot, exclusive scan over chunks (in place) and slot totals
__global__ void k_group_scan_chunks(uint32_t *__restrict__ hist, int64_t n_chunks, int32_t n_slots, uint32_t *__restrict__ totals) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots) return;
    uint32_t run = 0;
    for (int64_t c = 0; c < n_chunks; c++) {
        uint32_t v = hist[c * n_slots + s];
        hist[c * n_slots + s] = run;
        run += v;
    }
    totals[s] = run;
}
// pass 2b: seg_off = exclusive scan of totals (one wavefront)
__global__ void k_group_scan_slots(const uint32_t *__restrict__ totals, int32_t n_slots, uint32_t *__restrict__ seg_off) {
    uint32_t carry = 0;
    int lane = threadIdx.x;
    for (int b = 0; b < n_slots; b += WAVE) {
        int s = b + lane;
        uint32_t v = s < n_slots ? totals[s] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            uint32_t o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (s < n_slots) seg_off[s] = carry + inc - v;
        carry += __shfl(inc, WAVE - 1);
    }
    if (lane == 0) seg_off[n_slots] = carry;
}
// pass 3: stable rank inside the chunk, one wavefront per chunk; order[dst] = row
__global__ void k_group_scatter(const int32_t *__restrict__ slot, int64_t N, int32_t n_slots,
                                const uint32_t *__restrict__ hist, const uint32_t *__restrict__ seg_off,
                                int32_t *__restrict__ order) {
    extern __shared__ uint32_t lds_pos_raw[];  // next grouped row of each slot for this chunk
    volatile uint32_t *lds_pos = lds_pos_raw;    // lanes hand values to each other through it
    const uint32_t *h = hist + (int64_t)blockIdx.x * n_slots;
    for (int s = threadIdx.x; s < n_slots; s += WAVE) lds_pos[s] = seg_off[s] + h[s];
    __syncthreads();
    int lane = threadIdx.x;
    int64_t base = (int64_t)blockIdx.x * GRP_CHUNK;
    for (int i = 0; i < GRP_CHUNK; i += WAVE) {
        int64_t row = base + i + lane;
        bool valid = row < N;
        int32_t key = valid ? slot[row] : -1;
        uint64_t todo = __ballot(valid);
        uint32_t dst = 0;
        while (todo) {  // one pass per distinct key among the 64 rows
            int leader = __ffsll((unsigned long long)todo) - 1;
            int32_t k = __shfl(key, leader);
            uint64_t same = __ballot(valid && key == k);
            if (valid && key == k) {
                uint32_t before = __popcll(same & ((1ull << lane) - 1ull));
                dst = lds_pos[k] + before;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): reads of lds_pos[k] done before its update
            if (lane == leader) lds_pos[k] += __popcll(same);
            todo &= ~same;
        }
        if (valid) order[dst] = (int32_t)row;
    }
}

extern "C" int64_t offsim_group_scratch_bytes(int64_t N, int32_t n_slots) {
    int64_t n_chunks = (N + GRP_CHUNK - 1) / GRP_CHUNK;
    if (n_chunks < 1) n_chunks = 1;
    return (n_chunks * n_slots + n_slots) * (int64_t)sizeof(uint32_t) + 256;
}

extern "C" int offsim_group_by_state(const int32_t *slot, int64_t N, int32_t n_slots, uint32_t *seg_off, int32_t *order,
                                     void *scratch, void *stream) {
    if (N < 0 || n_slots <= 0 || !seg_off || !scratch) return fail(OFFSIM_EINVAL, "group_by_state: bad argument%s");
    if ((int64_t)n_slots * 4 > 64 * 1024) return fail(OFFSIM_EUNSUPPORTED, "group_by_state: more than 16384 states%s");
    if (N >= (1ll << 31)) return fail(OFFSIM_EUNSUPPORTED, "group_by_state: N >= 2^31%s");
    hipStream_t st = (hipStream_t)stream;
    int64_t n_chunks = (N + GRP_CHUNK - 1) / GRP_CHUNK;
    uint32_t *hist = (uint32_t *)scratch;
    uint32_t *totals = hist + (n_chunks > 0 ? n_chunks : 1) * n_slots;
    if (n_chunks == 0) {
        HIP_TRY(hipMemsetAsync(seg_off, 0, sizeof(uint32_t) * (n_slots + 1), st));
        return OFFSIM_OK;
    }
    size_t lds = sizeof(uint32_t) * n_slots;
    hipLaunchKernelGGL(k_group_hist, dim3((unsigned)n_chunks), dim3(256), lds, st, slot, N, n_slots, hist);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_group_scan_chunks, dim3((n_slots + 63) / 64), dim3(64), 0, st, hist, n_chunks, n_slots, totals);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_group_scan_slots, dim3(1), dim3(WAVE), 0, st, totals, n_slots, seg_off);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_group_scatter, dim3((unsigned)n_chunks), dim3(WAVE), lds, st, slot, N, n_slots, hist, seg_off, order);
    LAUNCH_CHECK();
    return OFFSIM_OK;
}

// dst[g] = src[order[g]] for rows of row_bytes bytes (1,2,4,8 or a multiple of 4)
template <typename T>
__global__ void k_gather_elem(const T *__restrict__ src, const int32_t *__restrict__ order, int64_t N, T *__restrict__ dst) {
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < N) dst[g] = src[order[g]];
}
__global__ void k_gather_words(const uint32_t *__restrict__ src, const int32_t *__restrict__ order, int64_t N, int32_t words,
                               uint32_t *__restrict__ dst) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * words) return;
    int64_t g = i / words;
    int32_t w = (int32_t)(i - g * words);
    dst[i] = src[(int64_t)order[g] * words + w];
}
extern "C" int offsim_gather_rows(const void *src, const int32_t *order, int64_t N, int32_t row_bytes, void *dst, void *stream) {
    if (N < 0 || row_bytes <= 0) return fail(OFFSIM_EINVAL, "gather_rows: bad argument%s");
    if (N == 0) return OFFSIM_OK;
    hipStream_t st = (hipStream_t)stream;
    unsigned nb = (unsigned)((N + 255) / 256);
    if (row_bytes == 1) hipLaunchKernelGGL(k_gather_elem<uint8_t>, dim3(nb), dim3(256), 0, st, (const uint8_t *)src, order, N, (uint8_t *)dst);
    else if (row_bytes == 2) hipLaunchKernelGGL(k_gather_elem<uint16_t>, dim3(nb), dim3(256), 0, st, (const uint16_t *)src, order, N, (uint16_t *)dst);
    else if (row_bytes == 4) hipLaunchKernelGGL(k_gather_elem<uint32_t>, dim3(nb), dim3(256), 0, st, (const uint32_t *)src, order, N, (uint32_t *)dst);
    else if (row_bytes == 8) hipLaunchKernelGGL(k_gather_elem<uint64_t>, dim3(nb), dim3(256), 0, st, (const uint64_t *)src, order, N, (uint64_t *)dst);
    else if (row_bytes % 4 == 0) {
        int32_t words = row_bytes / 4;
        unsigned nbw = (unsigned)((N * words + 255) / 256);
        hipLaunchKernelGGL(k_gather_words, dim3(nbw), dim3(256), 0, st, (const uint32_t *)src, order, N, words, (uint32_t *)dst);
    } else if (row_bytes % 2 == 0) {
        return fail(OFFSIM_EUNSUPPORTED, "gather_rows: row_bytes must be 1, 2 or a multiple of 4%s");
    } else return fail(OFFSIM_EUNSUPPORTED, "gather_rows: row_bytes must be 1, 2 or a multiple of 4%s");
    LAUNCH_CHECK();
    return OFFSIM_OK;
}

// ------------------------------------------------------------------------------------------------
// Sampler: seeding and per-rollout queue shuffles
// ------------------------------------------------------------------------------------------------
__global__ void k_seed_streams(const uint64_t *__restrict__ seeds, int32_t R, uint64_t *__restrict__ out) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    PcgInit p = pcg_seed(seeds[r]);
    out[4 * r + 0] = p.state.hi;
    out[4 * r + 1] = p.state.lo;
    out[4 * r + 2] = p.inc.hi;
    out[4 * r + 3] = p.inc.lo;
}
extern "C" int offsim_seed_streams(const uint64_t *seeds, int32_t R, uint64_t *rng_out, void *stream) {
    if (R < 0 || !seeds || !rng_out) return fail(OFFSIM_EINVAL, "seed_streams: bad argument%s");
    if (R == 0) return OFFSIM_OK;
    hipLaunchKernelGGL(k_seed_streams, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, seeds, R, rng_out);
    LAUNCH_CHECK();
    return OFFSIM_OK;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device and sticks: set once per (kernel, device), not per launch
// (it sits on the single-step latency path otherwise)
static hipError_t allow_big_lds_fn(const void *fn, int bytes) {
    struct Seen {
        const void *fn;
        int max_bytes;  // the limit applied on the devices in `devs` (only ever raised)
        uint64_t devs;  // one bit per device id (ids >= 64 are set every time)
    };
    static thread_local Seen seen[64];
    static thread_local int n_seen = 0;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    Seen *hit = nullptr;
    for (int i = 0; i < n_seen; i++)
        if (seen[i].fn == fn) hit = &seen[i];
    if (hit && dev < 64 && bytes <= hit->max_bytes && ((hit->devs >> dev) & 1ull)) return hipSuccess;
    const int want = hit && hit->max_bytes > bytes ? hit->max_bytes : bytes;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, want);
    if (e != hipSuccess) return e;
    if (!hit && n_seen < 64) {
        hit = &seen[n_seen++];
        hit->fn = fn;
        hit->max_bytes = 0;
        hit->devs = 0;
    }
    if (hit) {
        if (want > hit->max_bytes) {
            hit->max_bytes = want;
            hit->devs = 0;
        }
        if (dev < 64) hit->devs |= 1ull << dev;
    }
    return hipSuccess;
}
#define allow_big_lds(kernel, bytes) allow_big_lds_fn((const void *)(kernel), (bytes))

// wave-parallel exact Fisher-Yates (shuffle_wave.hpp), one workgroup per chain.  dig_out == NULL: orders as permutations of grouped
// rows; otherwise the keyed form (digest stream + 16-bit local rows per queue position) for the state queues
static int lds_order_ok_on(void *stream);  // the LDS lane-order guard as the launch paths ask it (defined beside offsim_lds_order_ok)
static int launch_shuffle(const offsim_table *t, const uint64_t *seeds, int32_t n_perm, uint32_t *perm_out, uint32_t *init_perm_out,
                          const uint32_t *dig32, uint32_t *dig_out, uint16_t *loc_out, hipStream_t st, bool big_segments_elsewhere = false) {
    const uint32_t max_seg = t->max_seg > 0 ? (uint32_t)(t->max_seg > 0xffffffffll ? 0xffffffffll : t->max_seg) : 0xffffffffu;
    const uint32_t min_seg = t->min_seg > 0 ? (uint32_t)(t->min_seg > 0xffffffffll ? 0xffffffffll : t->min_seg) : 1u;
    const uint32_t n0 = (uint32_t)(t->N0 > 0xffffffffll ? 0xffffffffll : t->N0);
    const int64_t n_blocks = (int64_t)(t->n_slots + 1) * n_perm;
    if (n_blocks > 0x7fffffffll) return fail(OFFSIM_EUNSUPPORTED, "shuffle_queues: too many chains%s");
    HIP_TRY(allow_big_lds((k_shuffle_wave<true, SHUF_SQ_BIG>), 160 * 1024));
    HIP_TRY(allow_big_lds((k_shuffle_wave<true, SHUF_SQ_SMALL>), 160 * 1024));
    // the applier's exchange form (one ds_mskor_rtn_b32 per lane instead of the tag round) where the LDS serves same-address lanes in
    // lane order (offsim_lds_order_ok: once per device); elsewhere the tag form -- same orders
    static const bool no_xchg = getenv("OFFSIM_SHUFFLE_XCHG") && atoi(getenv("OFFSIM_SHUFFLE_XCHG")) == 0;
    const int okv = lds_order_ok_on(st);
    if (okv < 0) return okv;
    const uint32_t a_xchg = (okv == 1 && !no_xchg) ? 1u : 0u;
    if (!big_segments_elsewhere && (max_seg > SHUF_CAP16 || n0 > SHUF_CAP16)) {  // first: these chains are the long ones (or the chunked kernel has them)
        hipLaunchKernelGGL((k_shuffle_wave<false, SHUF_SQ_SMALL>), dim3((unsigned)n_blocks), dim3(256), shuf_fixed_lds_bytes(SHUF_SQ_SMALL), st, t->seg_off,
                           t->n_slots, t->N, t->N0, seeds, n_perm, perm_out, init_perm_out, SHUF_CAP16, 0xffffffffu, dig32, dig_out, loc_out, 0u);
        LAUNCH_CHECK();
    }
    // LDS-resident segments by size class, longest first: the LDS of a launch is sized for its class, so several short
    // chains share a CU instead of inheriting the one-chain-per-CU occupancy of a 60 k-row segment
    static const uint32_t bounds[] = {SHUF_CAP16, 32768u, 8192u, 2048u, 0u};
    for (int k = 0; k < 4; k++) {
        const uint32_t hi = bounds[k], lo = bounds[k + 1];
        const bool seg_in = min_seg <= hi && max_seg > lo, init_in = n0 > lo && n0 <= hi;
        if (!seg_in && !init_in) continue;
        uint32_t need = 0;
        if (seg_in) need = max_seg < hi ? max_seg : hi;
        if (init_in && n0 > need) need = n0;
        const uint32_t sq = k == 0 ? SHUF_SQ_BIG : SHUF_SQ_SMALL;
        const size_t lds16 = shuf_fixed_lds_bytes(sq) + (((size_t)need * 2 + 15) & ~(size_t)15) + 16;
        if (k == 0 && dig_out) {
            // keyed chains of the longest class in three launches (shuffle_wave.hpp, <TOP, STOP>): the steps above 16384 with the
            // whole segment in LDS, then 16383 .. 4096 with 32 KB per chain (three chains per CU), then the conflict-ridden low end
            // with 8 KB (seven per CU); the init queue (never keyed) keeps the one-launch form
#define SHUF_LAUNCH(SQ, TOP, STOP, LDSB)                                                                                              \
    do {                                                                                                                              \
        HIP_TRY(allow_big_lds((k_shuffle_wave<true, SQ, TOP, STOP>), 160 * 1024));                                                     \
        hipLaunchKernelGGL((k_shuffle_wave<true, SQ, TOP, STOP>), dim3((unsigned)n_blocks), dim3(256), (LDSB), st, t->seg_off, t->n_slots, \
                           t->N, t->N0, seeds, n_perm, perm_out, init_perm_out, lo, hi, dig32, dig_out, loc_out, a_xchg);             \
        LAUNCH_CHECK();                                                                                                               \
    } while (0)
#define SHUF_TAILB(TOP) (shuf_fixed_lds_bytes(SHUF_SQ_SMALL) + (size_t)(TOP) * 2 + 16)
#ifdef SHUF_PROF  // (tools/prof_shuffle.py: the stamps of the LAST launch survive in the streams -- OFFSIM_SHUFFLE_PROF_LAUNCHES = 1 / 2 stops after the first / second)
            const int prof_launches = getenv("OFFSIM_SHUFFLE_PROF_LAUNCHES") ? atoi(getenv("OFFSIM_SHUFFLE_PROF_LAUNCHES")) : 3;
#else
            const int prof_launches = 3;
#endif
            SHUF_LAUNCH(SHUF_SQ_BIG, 0, SHUF_CUT_HI, lds16);
            if (prof_launches >= 2) SHUF_LAUNCH(SHUF_SQ_SMALL, SHUF_CUT_HI, SHUF_CUT_LO, SHUF_TAILB(SHUF_CUT_HI));
            if (prof_launches >= 3) SHUF_LAUNCH(SHUF_SQ_SMALL, SHUF_CUT_LO, 1, SHUF_TAILB(SHUF_CUT_LO));
#undef SHUF_LAUNCH
#undef SHUF_TAILB
            if (init_in)  // (an init queue of this size class: its chains are not keyed)
                hipLaunchKernelGGL((k_shuffle_wave<true, SHUF_SQ_BIG>), dim3((unsigned)n_perm), dim3(256), lds16, st, t->seg_off, t->n_slots, t->N,
                                   t->N0, seeds, n_perm, perm_out, init_perm_out, lo, hi, dig32, nullptr, nullptr, a_xchg);
        } else if (k == 0)
            hipLaunchKernelGGL((k_shuffle_wave<true, SHUF_SQ_BIG>), dim3((unsigned)n_blocks), dim3(256), lds16, st, t->seg_off, t->n_slots, t->N, t->N0,
                               seeds, n_perm, perm_out, init_perm_out, lo, hi, dig32, dig_out, loc_out, a_xchg);
        else
            hipLaunchKernelGGL((k_shuffle_wave<true, SHUF_SQ_SMALL>), dim3((unsigned)n_blocks), dim3(256), lds16, st, t->seg_off, t->n_slots, t->N, t->N0,
                               seeds, n_perm, perm_out, init_perm_out, lo, hi, dig32, dig_out, loc_out, a_xchg);
        LAUNCH_CHECK();
    }
    return OFFSIM_OK;
}

// Fault bits raised by asynchronous kernels of this device since the last call (and cleared by it): OFFSIM_FAULT_SHUFFLE -- a role of
// the shuffle's ring protocol gave up a bounded wait (the orders of that call are invalid); OFFSIM_FAULT_SCAN -- the same in the scan
// (those rollouts also carry OFFSIM_ST_PROTOCOL).  Synchronise the stream the kernels ran on first.
__global__ void k_fault_exchange(int32_t *out) { *out = atomicExch(&offsim::g_async_fault, 0); }

extern "C" int offsim_async_faults(void) {
    // read and clear in ONE atomic exchange on the device (a separate read and clear could lose a bit raised in between); the word
    // is per device: callers that interleave several environments on one device attribute a fault to whatever they ran since their
    // last call (include/offsim.h)
    static int32_t *slot[64] = {nullptr};
    static std::mutex mu;
    std::lock_guard<std::mutex> hold(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(OFFSIM_EHIP, "async_faults: no device%s");
    if (!slot[dev] && hipMalloc((void **)&slot[dev], sizeof(int32_t)) != hipSuccess) return fail(OFFSIM_EHIP, "async_faults: allocation failed%s");
    hipLaunchKernelGGL(k_fault_exchange, dim3(1), dim3(1), 0, (hipStream_t)0, slot[dev]);
    if (hipGetLastError() != hipSuccess) return fail(OFFSIM_EHIP, "async_faults: launch failed%s");
    int32_t v = 0;
    if (hipMemcpy(&v, slot[dev], sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return fail(OFFSIM_EHIP, "async_faults: read failed%s");
    return v;
}

extern "C" int offsim_shuffle_queues(const offsim_table *t, const uint64_t *seeds, int32_t n_perm, uint32_t *perm_out,
                                     uint32_t *init_perm_out, void *stream) {
    if (!t || !seeds || n_perm < 0 || !perm_out || !init_perm_out) return fail(OFFSIM_EINVAL, "shuffle_queues: bad argument%s");
    if (n_perm == 0) return OFFSIM_OK;
    return launch_shuffle(t, seeds, n_perm, perm_out, init_perm_out, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

// Keyed form of the chains that do not fit LDS (states of more than 65536 rows, stream format B): the global-memory variant has shuffled
// grouped-row indices in place in the state's slice of dig_out; every position becomes {digest | local-row bits 16.., local-row low half}.
__global__ void __launch_bounds__(256) k_big_keys(const uint32_t *__restrict__ seg_off, int64_t N, const uint32_t *__restrict__ dig32,
                                                  uint32_t *__restrict__ dig_out, uint16_t *__restrict__ loc_out, uint32_t r0) {
    const uint32_t s = blockIdx.y, r = r0 + blockIdx.z;
    const uint32_t beg = seg_off[s], len = seg_off[s + 1] - beg;
    if (len <= SHUF_CAP16) return;
    for (uint32_t k = blockIdx.x * 1024u + threadIdx.x; k < len && k < (blockIdx.x + 1u) * 1024u; k += 256u) {
        const int64_t p = (int64_t)r * N + beg + k;
        const uint32_t row = dig_out[p], loc = row - beg, h = loc >> 16;
        dig_out[p] = dig32[row] | ((h & 3u) << 8) | ((h >> 2) << 11);
        loc_out[p] = (uint16_t)loc;
    }
}

// Workspace of the chunked shuffle (shuffle_chunk.hpp): header (work counter, work list) + one message pool and one reply pool per
// persistent workgroup, sized for the table's longest state.
// Chunk size: OFFSIM_SHUFFLE_CHUNK = 4096 (default; measured at C2 / C3: 2048: 0.077 / 1.22 s per pass, 4096: 0.071 / 1.08, 8192: 0.095 / 1.26),
// 2048, 8192 or 16384 positions; as many persistent workgroups per CU as its LDS holds (three to five at 4096).
#define SHC_HEADER_BYTES 4096
static uint32_t shc_cb() {
    static const int v = getenv("OFFSIM_SHUFFLE_CHUNK") ? atoi(getenv("OFFSIM_SHUFFLE_CHUNK")) : 0;
    return v == 16384 ? 16384u : v == 8192 ? 8192u : v == 2048 ? 2048u : 4096u;
}
// rows of the longest chain the chunked kernel serves: states and the init queue, as far as they exceed the LDS capacity (0: none)
static uint32_t shc_longest(const offsim_table *t) {
    const int64_t a = t->max_seg > (int64_t)SHUF_CAP16 ? t->max_seg : 0, b = t->N0 > (int64_t)SHUF_CAP16 ? t->N0 : 0;
    return (uint32_t)(a > b ? a : b);
}
static int64_t shc_block_words(const offsim_table *t, uint32_t *msg_cap_out, int64_t *lc_words_out = nullptr, bool plain = false) {
    const uint32_t n = shc_longest(t), cb = shc_cb();
    const uint64_t msg = shc_pool_entries(n, cb), rep = (uint64_t)((n + cb - 1u) / cb) * cb;
    if (msg_cap_out) *msg_cap_out = (uint32_t)msg;
    if (lc_words_out) *lc_words_out = plain ? 0 : (int64_t)((n + 3u) / 4u);  // (16-bit scratch for the low halves of an init queue's plain records)
    if (plain) return (int64_t)(msg + rep);  // (orders as permutations: a message is one 64-bit word, and so is a reply)
    return (int64_t)(msg + (msg + 1u) / 2u + rep + (n + 3u) / 4u);  // messages: a 64-bit and a 32-bit word each; replies: one 64-bit word
}
extern "C" int64_t offsim_shuffle_workspace_bytes(const offsim_table *t, int32_t n_blocks) {
    if (!t || n_blocks < 1 || shc_longest(t) == 0 || t->max_seg > (1ll << 23) || t->N0 > (1ll << 23)) return 0;
    return SHC_HEADER_BYTES + (int64_t)n_blocks * shc_block_words(t, nullptr) * 8;
}

// The chunked kernel for the chains that do not fit LDS, if the caller lent a workspace that holds at least one workgroup's pools:
// *n_wg_out = the persistent workgroups it will run with (0: not applicable -- the in-place form has to serve those chains).
// perm_out != NULL: the orders go out as permutations (plain records); otherwise as the streams dig_out / loc_out.
struct ShcPlan {
    int64_t n_wg = 0, words = 0, lc_words = 0;
    uint32_t msg_cap = 0, kcap = 0, lds_b = 0;
};
static ShcPlan shc_plan(const offsim_table *t, void *workspace, int64_t workspace_bytes, bool plain) {
    ShcPlan p;
    if (shc_longest(t) == 0 || t->max_seg > (1ll << 23) || t->N0 > (1ll << 23) || !workspace || ((uintptr_t)workspace & 7u) != 0) return p;
    if (t->n_slots + 1 > (SHC_HEADER_BYTES / 4 - 64)) return p;  // (the work list lives in the header)
    p.words = shc_block_words(t, &p.msg_cap, &p.lc_words, plain);
    p.n_wg = (workspace_bytes - SHC_HEADER_BYTES) / (p.words * 8);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    p.kcap = (shc_longest(t) + shc_cb() - 1u) / shc_cb();
#define SHC_LDSB(PL)                                                                                                                    \
    (shc_cb() == 16384u ? shc_lds_bytes<16384u, 2048u, 1024u, PL>(p.kcap) : shc_cb() == 4096u ? shc_lds_bytes<4096u, 1024u, 512u, PL>(p.kcap) \
     : shc_cb() == 2048u ? shc_lds_bytes<2048u, 1024u, 512u, PL>(p.kcap) : shc_lds_bytes<8192u, 1024u, 512u, PL>(p.kcap))
    p.lds_b = plain ? SHC_LDSB(true) : SHC_LDSB(false);
#undef SHC_LDSB
    const int64_t per_cu = (160 * 1024) / (int64_t)((p.lds_b + 1023u) & ~1023u);  // persistent workgroups per CU (what their chunks leave of its LDS)
    p.n_wg = p.n_wg > cus * per_cu ? cus * per_cu : p.n_wg;
    if (p.n_wg < 1) p.n_wg = 0;
    return p;
}
static int shc_launch(const ShcPlan &p, const offsim_table *t, const uint64_t *seeds, int32_t n_perm, const uint32_t *dig32, uint32_t *dig_out,
                      void *loc_out, uint32_t *init_perm_out, uint32_t *perm_out, void *workspace, hipStream_t st, uint32_t loc_bits = 16u,
                      uint32_t chains_above = SHUF_CAP16) {
    // the chunked kernel applies a group of messages with one ds_wrxchg_rtn_b32 per lane and relies on the LDS serving same-address
    // lanes in lane order: asked once per device (offsim_lds_order_ok)
    { const int okv = lds_order_ok_on(st); if (okv < 0) return okv; if (okv == 0) return fail(OFFSIM_EUNSUPPORTED, "chunked shuffle: this device's LDS does not serve same-address lanes of one instruction in lane order (offsim_lds_order_ok): lend no workspace -- the in-place shuffle does not need it%s"); }
    uint32_t *hdr = (uint32_t *)workspace;  // [0] work counter, [1] number of long chains, [64 ..] their indices, longest first
    hipLaunchKernelGGL(k_chunk_worklist, dim3(1), dim3(256), 0, st, t->seg_off, t->n_slots, (uint32_t)(t->N0 > 0xffffffffll ? 0xffffffffll : t->N0), chains_above,
                       hdr + 64, hdr + 1, hdr);
    LAUNCH_CHECK();
#define SHC_LAUNCH_(CB, RG, SQ, PL)                                                                                                     \
    do {                                                                                                                                \
        HIP_TRY(allow_big_lds((k_shuffle_chunked<CB, RG, SQ, PL>), 160 * 1024));                                                         \
        hipLaunchKernelGGL((k_shuffle_chunked<CB, RG, SQ, PL>), dim3((unsigned)p.n_wg), dim3(256), p.lds_b, st, t->seg_off, t->N, seeds, n_perm,  \
                           hdr + 64, hdr + 1, hdr, (uint64_t *)((char *)workspace + SHC_HEADER_BYTES), p.words, p.msg_cap, p.kcap, dig32,   \
                           dig_out, loc_out, t->n_slots, t->N0, init_perm_out, perm_out, p.lc_words, loc_bits);                         \
    } while (0)
#define SHC_LAUNCH(CB, RG, SQ)                                                                                                          \
    do {                                                                                                                                \
        if (perm_out) SHC_LAUNCH_(CB, RG, SQ, true);                                                                                    \
        else SHC_LAUNCH_(CB, RG, SQ, false);                                                                                            \
    } while (0)
    if (shc_cb() == 16384u) SHC_LAUNCH(16384u, 2048u, 1024u);
    else if (shc_cb() == 4096u) SHC_LAUNCH(4096u, 1024u, 512u);
    else if (shc_cb() == 2048u) SHC_LAUNCH(2048u, 1024u, 512u);
    else SHC_LAUNCH(8192u, 1024u, 512u);
#undef SHC_LAUNCH
#undef SHC_LAUNCH_
    LAUNCH_CHECK();
    return OFFSIM_OK;
}

extern "C" int offsim_shuffle_queues_ws(const offsim_table *t, const uint64_t *seeds, int32_t n_perm, uint32_t *perm_out, uint32_t *init_perm_out,
                                        void *workspace, int64_t workspace_bytes, void *stream) {
    if (!t || !seeds || n_perm < 0 || !perm_out || !init_perm_out) return fail(OFFSIM_EINVAL, "shuffle_queues: bad argument%s");
    if (n_perm == 0) return OFFSIM_OK;
    const ShcPlan p = t->max_seg > 0 ? shc_plan(t, workspace, workspace_bytes, true) : ShcPlan();  // (max_seg unknown: the in-place form)
    int rc = launch_shuffle(t, seeds, n_perm, perm_out, init_perm_out, nullptr, nullptr, nullptr, (hipStream_t)stream, p.n_wg >= 1);
    if (rc || p.n_wg < 1) return rc;
    return shc_launch(p, t, seeds, n_perm, nullptr, nullptr, nullptr, init_perm_out, perm_out, workspace, (hipStream_t)stream);
}

extern "C" int offsim_shuffle_queues_keys_ws(const offsim_table *t, const uint64_t *seeds, int32_t n_perm, const uint32_t *dig32, int32_t format,
                                             uint32_t *dig_out, void *loc_out, uint32_t *init_perm_out, void *workspace, int64_t workspace_bytes,
                                             void *stream) {
    if (!t || !seeds || n_perm < 0 || !init_perm_out || (t->N > 0 && (!dig32 || !dig_out || !loc_out)))
        return fail(OFFSIM_EINVAL, "shuffle_queues_keys: bad argument%s");
    if (format != OFFSIM_STREAMS_A && format != OFFSIM_STREAMS_B && format != OFFSIM_STREAMS_C) return fail(OFFSIM_EINVAL, "shuffle_queues_keys: bad format%s");
    if (t->max_seg <= 0 && t->N > 0) return fail(OFFSIM_EINVAL, "shuffle_queues_keys: offsim_table.max_seg must be set%s");
    const bool big = t->max_seg > (int64_t)SHUF_CAP16;
    if (format == OFFSIM_STREAMS_C) {
        // one byte of the local row beside the digest: every chain runs on the chunked kernel (the LDS-resident kernel hands its low
        // positions from launch to launch as 16-bit rows in the loc stream; these streams have no room for them)
        if (t->max_seg > (1ll << 17) || t->n_slots > 255 || t->N0 > (1ll << 23))
            return fail(OFFSIM_EUNSUPPORTED, "shuffle_queues_keys: format C holds 2^17 rows per state, 255 states%s");
        if (n_perm == 0 || t->N == 0) return OFFSIM_OK;
        offsim_table tc = *t;  // (the pools are sized for the longest chain, whatever its length)
        if (tc.max_seg <= (int64_t)SHUF_CAP16) tc.max_seg = (int64_t)SHUF_CAP16 + 1;
        const ShcPlan pc = shc_plan(&tc, workspace, workspace_bytes, false);
        if (pc.n_wg < 1) return fail(OFFSIM_EUNSUPPORTED, "shuffle_queues_keys: format C needs a workspace (offsim_shuffle_workspace_bytes)%s");
        return shc_launch(pc, &tc, seeds, n_perm, dig32, dig_out, loc_out, init_perm_out, nullptr, workspace, (hipStream_t)stream, 8u, 0u);
    }
    if (big && format != OFFSIM_STREAMS_B)
        return fail(OFFSIM_EUNSUPPORTED, "shuffle_queues_keys: a state has more than 65536 rows: stream format B (or offsim_shuffle_queues)%s");
    if (big && (t->max_seg > (1ll << 23) || t->n_slots > 255)) return fail(OFFSIM_EUNSUPPORTED, "shuffle_queues_keys: format B holds 2^23 rows per state, 255 states%s");
    if (n_perm == 0) return OFFSIM_OK;
    hipStream_t st = (hipStream_t)stream;
    // the chains that do not fit LDS: the chunked kernel when the caller lent a workspace that holds at least one workgroup's pools,
    // otherwise in place in global memory (a state: in its slice of dig_out) and one more pass that turns the order into streams
    const ShcPlan p = shc_plan(t, workspace, workspace_bytes, false);
    const bool chunked = p.n_wg >= 1;
    int rc = launch_shuffle(t, seeds, n_perm, big && !chunked ? dig_out : nullptr, init_perm_out, dig32, dig_out, (uint16_t *)loc_out, st, chunked);
    if (rc || (!big && !chunked)) return rc;
    if (chunked) return shc_launch(p, t, seeds, n_perm, dig32, dig_out, loc_out, init_perm_out, nullptr, workspace, st);
    for (int32_t r0 = 0; r0 < n_perm; r0 += 65535) {  // (gridDim.z holds at most 65535 orders)
        const int32_t nz = n_perm - r0 < 65535 ? n_perm - r0 : 65535;
        hipLaunchKernelGGL(k_big_keys, dim3((unsigned)((t->max_seg + 1023) / 1024), (unsigned)t->n_slots, (unsigned)nz), dim3(256), 0, st,
                           t->seg_off, t->N, dig32, dig_out, (uint16_t *)loc_out, (uint32_t)r0);
        LAUNCH_CHECK();
    }
    return OFFSIM_OK;
}

extern "C" int offsim_shuffle_queues_keys(const offsim_table *t, const uint64_t *seeds, int32_t n_perm, const uint32_t *dig32, int32_t format,
                                          uint32_t *dig_out, void *loc_out, uint32_t *init_perm_out, void *stream) {
    return offsim_shuffle_queues_keys_ws(t, seeds, n_perm, dig32, format, dig_out, loc_out, init_perm_out, nullptr, 0, stream);
}

// ------------------------------------------------------------------------------------------------
// env reset / set_state
// ------------------------------------------------------------------------------------------------
__global__ void k_env_reset(offsim_table t, offsim_rollouts ro, const uint8_t *__restrict__ mask, int32_t *__restrict__ out_row) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ro.R) return;
    if (mask && !mask[r]) return;
    uint32_t ic = ro.init_cursor[r];
    if ((int64_t)ic >= t.N0) {  // psrs.py:33-35: self.s = None
        ro.cur_slot[r] = -1;
        if (out_row) out_row[r] = -1;
        return;
    }
    uint32_t k = ro.init_perm ? ro.init_perm[(int64_t)r * ro.init_stride + ic] : ic;
    ro.init_cursor[r] = ic + 1;
    ro.cur_slot[r] = t.init_slot[k];
    if (out_row) out_row[r] = t.init_orig[k];
}
extern "C" int offsim_env_reset(const offsim_table *t, offsim_rollouts *ro, const uint8_t *mask, int32_t *out_init_row,
                                void *stream) {
    if (!t || !ro || ro->R < 0) return fail(OFFSIM_EINVAL, "env_reset: bad argument%s");
    if (ro->R == 0) return OFFSIM_OK;
    hipLaunchKernelGGL(k_env_reset, dim3((ro->R + 255) / 256), dim3(256), 0, (hipStream_t)stream, *t, *ro, mask, out_init_row);
    LAUNCH_CHECK();
    return OFFSIM_OK;
}
__global__ void k_env_set_state(offsim_rollouts ro, const int32_t *__restrict__ slot, const uint8_t *__restrict__ mask) {
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ro.R) return;
    if (mask && !mask[r]) return;
    ro.cur_slot[r] = slot[r];
}
// payload of a batched step / reset (offsim.h: offsim_vector_gather): one thread per environment
struct VecCols {
    offsim_column c[8];
};
__global__ void k_vector_gather(const int32_t *__restrict__ row, const int32_t *__restrict__ status, const uint8_t *__restrict__ mask, int32_t R,
                                VecCols cols, int32_t n_cols, uint8_t *__restrict__ alive) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= R) return;
    const int32_t rw = row[k];
    bool ok;
    if (status) {
        ok = status[k] == OFFSIM_ST_OK && rw >= 0;
        if (alive) alive[k] = (uint8_t)(alive[k] && ok);
    } else {
        const bool m = mask ? mask[k] != 0 : true;
        ok = m && rw >= 0;
        if (alive && m) alive[k] = (uint8_t)(rw >= 0);
    }
    for (int c = 0; c < n_cols; c++) {
        const int64_t nb = cols.c[c].row_bytes;
        unsigned char *d = (unsigned char *)cols.c[c].dst + (int64_t)k * nb;
        if (ok) {
            const unsigned char *sp = (const unsigned char *)cols.c[c].src + (int64_t)rw * nb;
            if (((nb | (int64_t)(uintptr_t)sp | (int64_t)(uintptr_t)d) & 3) == 0) {
                for (int64_t b = 0; b < nb; b += 4) *(uint32_t *)(d + b) = *(const uint32_t *)(sp + b);
            } else {
                for (int64_t b = 0; b < nb; b++) d[b] = sp[b];
            }
        } else if (cols.c[c].zero_if_not_ok) {
            for (int64_t b = 0; b < nb; b++) d[b] = 0;
        }
    }
}

extern "C" int offsim_vector_gather(const int32_t *row, const int32_t *status, const uint8_t *mask, int32_t R, const offsim_column *cols,
                                    int32_t n_cols, uint8_t *alive, void *stream) {
    if (!row || R < 0 || n_cols < 0 || n_cols > 8 || (n_cols > 0 && !cols)) return fail(OFFSIM_EINVAL, "vector_gather: bad argument%s");
    VecCols vc;
    memset(&vc, 0, sizeof(vc));
    for (int c = 0; c < n_cols; c++) {
        if (!cols[c].src || !cols[c].dst || cols[c].row_bytes <= 0) return fail(OFFSIM_EINVAL, "vector_gather: bad column%s");
        vc.c[c] = cols[c];
    }
    if (R == 0) return OFFSIM_OK;
    hipLaunchKernelGGL(k_vector_gather, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, (hipStream_t)stream, row, status, mask, R, vc, n_cols,
                       alive);
    LAUNCH_CHECK();
    return OFFSIM_OK;
}

extern "C" int offsim_env_set_state(offsim_rollouts *ro, const int32_t *slot, const uint8_t *mask, void *stream) {
    if (!ro || !slot || ro->R < 0) return fail(OFFSIM_EINVAL, "env_set_state: bad argument%s");
    if (ro->R == 0) return OFFSIM_OK;
    hipLaunchKernelGGL(k_env_set_state, dim3((ro->R + 255) / 256), dim3(256), 0, (hipStream_t)stream, *ro, slot, mask);
    LAUNCH_CHECK();
    return OFFSIM_OK;
}

// ------------------------------------------------------------------------------------------------
// The replay loop.  One wavefront = one rollout.
// ------------------------------------------------------------------------------------------------
template <typename PL>
__device__ __forceinline__ double plog_f64(const PL *p, int64_t i);
template <>
__device__ __forceinline__ double plog_f64<float>(const float *p, int64_t i) { return (double)p[i]; }
template <>
__device__ __forceinline__ double plog_f64<double>(const double *p, int64_t i) { return p[i]; }
template <>
__device__ __forceinline__ double plog_f64<__half>(const __half *p, int64_t i) { return (double)__half2float(p[i]); }

// psrs.py:53-57 for one candidate.  k53 = 53-bit draw, u = k53 * 2**-53.
// F64: divisions and comparison in double (p_log widened exactly).
template <typename PL>
__device__ __forceinline__ bool rejects_f64(const PL *__restrict__ plog, int64_t g, int a, const double *pnew, int nA, uint64_t k53) {
    double M = -__builtin_inf();
    bool nan = false;
    for (int k = 0; k < nA; k++) {
        double q = pnew[k] / plog_f64<PL>(plog, g * nA + k);
        nan |= (q != q);
        M = q > M ? q : M;
    }
    if (nan) M = __builtin_nan("");  // ndarray.max() propagates NaN
    double thr = pnew[a] / plog_f64<PL>(plog, g * nA + a) / M;
    double u = (double)k53 * (1.0 / 9007199254740992.0);
    return u > thr;
}
// F32: p_new and p_log both float32 -> NumPy divides in float32 and (NumPy 2 / torch 0-d) rounds u to float32.
__device__ __forceinline__ bool rejects_f32(const float *__restrict__ plog, int64_t g, int a, const float *pnew, int nA, uint64_t k53) {
    float M = -__builtin_inff();
    bool nan = false;
    for (int k = 0; k < nA; k++) {
        float q = pnew[k] / plog[g * nA + k];
        nan |= (q != q);
        M = q > M ? q : M;
    }
    if (nan) M = __builtin_nanf("");
    float thr = pnew[a] / plog[g * nA + a] / M;
    double u = (double)k53 * (1.0 / 9007199254740992.0);
    return (float)u > thr;
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    long long b = __double_as_longlong(v);
    int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
    int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

struct StepResult {
    int status;       // OFFSIM_ST_OK (accepted) | EXHAUSTED | KEYERROR
    uint32_t popped;  // candidates consumed
    int32_t g;        // accepted grouped row
    int32_t z_next;
    double r;
    bool done;
};

// Per-wave jump table in LDS: entry d (1..64) advances the rollout's PCG64 stream by d draws.
struct WaveRng {
    U128 lane_state;  // PCG64: state that yields draw (c + lane)
    Jump *table;      // PCG64: LDS, 65 entries
    int kind;         // OFFSIM_STREAM_*
    uint64_t seed, c; // Philox: the rollout's seed, draws consumed so far (lane k looks at draw c + k)
};

// One PSRS.step (psrs.py:39-51).  All arguments wave-uniform except what lanes load.
// PROB = double (F64 mode, any p_log type) or float (F32 mode, p_log float).
template <typename PL, typename PROB>
__device__ __forceinline__ StepResult psrs_step(const offsim_table &t, const uint32_t *__restrict__ seg_off,
                                                 const uint32_t *__restrict__ perm_row, int slot, uint32_t *cursor,
                                                 const PROB *pnew, int reject_mode, uint32_t max_pop, WaveRng &rng,
                                                 uint64_t &consumed, uint32_t width = WAVE) {
    // (width: candidates looked at per round -- the step server looks at the few whose lines it has touched ahead of the request)
    const int lane = threadIdx.x & (WAVE - 1);
    StepResult res;
    res.popped = 0;
    res.g = -1;
    res.z_next = -1;
    res.r = 0.0;
    res.done = false;
    if (slot < 0 || slot >= t.n_slots) {
        res.status = OFFSIM_ST_KEYERROR;
        return res;
    }
    const uint32_t beg = seg_off[slot];
    const uint32_t len = seg_off[slot + 1] - beg;
    if (len == 0) {  // z never occurs as a from-state: self.queues[z] raises KeyError (psrs.py:44)
        res.status = OFFSIM_ST_KEYERROR;
        return res;
    }
    uint32_t cur = cursor[slot];
    const PL *plog = (const PL *)t.p_log;
    const int nA = t.nA;
    for (;;) {
        uint32_t rem = len - cur;
        if (rem == 0 || (max_pop && res.popped >= max_pop)) {  // psrs.py:44-45
            res.status = OFFSIM_ST_EXHAUSTED;
            break;
        }
        uint32_t nv = rem < width ? rem : width;
        if (reject_mode == OFFSIM_REJECT_NEVER) nv = 1;
        if (max_pop && nv > max_pop - res.popped) nv = max_pop - res.popped;
        const bool valid = (uint32_t)lane < nv;
        uint32_t g = beg + cur + (valid ? lane : 0);
        if (perm_row) g = perm_row[g];
        // candidate stream (p_log row, a) and, speculatively, the accept-only stream
        const int a = t.a[g];
        const int32_t zn = t.z_next[g];
        const uint8_t dn = t.done[g];
        const double rv = t.r_dtype == OFFSIM_F64 ? ((const double *)t.r)[g] : (double)((const float *)t.r)[g];
        bool acc;
        if (reject_mode == OFFSIM_REJECT_NEVER) {
            acc = valid;
        } else {
            const uint64_t k53 = rng.kind == OFFSIM_STREAM_PHILOX ? philox_k53(rng.seed, rng.c + (uint64_t)lane) : pcg_output(rng.lane_state) >> 11;
            bool rej;
            if constexpr (sizeof(PROB) == 4) rej = rejects_f32((const float *)plog, (int64_t)g, a, (const float *)pnew, nA, k53);
            else rej = rejects_f64<PL>(plog, (int64_t)g, a, (const double *)pnew, nA, k53);
            acc = valid && !rej;
        }
        uint64_t m = __ballot(acc);
        uint32_t d;  // candidates consumed by this iteration
        int f = -1;
        if (m == 0) d = nv;
        else {
            f = __ffsll((unsigned long long)m) - 1;
            d = (uint32_t)f + 1;
        }
        cur += d;
        res.popped += d;
        if (reject_mode != OFFSIM_REJECT_NEVER) {  // every examined candidate consumed exactly one draw (psrs.py:56)
            if (rng.kind == OFFSIM_STREAM_PHILOX) {
                rng.c += d;
            } else {
                Jump j = rng.table[d];
                rng.lane_state = pcg_apply(j, rng.lane_state);
            }
            consumed += d;
        }
        if (f >= 0) {
            res.status = OFFSIM_ST_OK;
            res.g = __builtin_amdgcn_readlane((int)g, f);
            res.z_next = __builtin_amdgcn_readlane(zn, f);
            res.done = __builtin_amdgcn_readlane((int)dn, f) != 0;
            res.r = readlane_f64(rv, f);
            break;
        }
    }
    cursor[slot] = cur;
    return res;
}

// Fill the per-wave jump table and position lane k on draw k of the stream that starts at `base`.
__device__ __forceinline__ void wave_rng_init(WaveRng &rng, Jump *table, U128 base, U128 inc, int kind = OFFSIM_STREAM_PCG64) {
    const int lane = threadIdx.x & (WAVE - 1);
    rng.table = table;
    rng.kind = kind;
    rng.seed = base.hi;  // (Philox: the rng row is seed, draws consumed, 0, 0)
    rng.c = base.lo;
    if (kind == OFFSIM_STREAM_PHILOX) return;
    Jump mine = pcg_jump(inc, (uint64_t)lane + 1);
    table[lane + 1] = mine;
    if (lane == 0) {
        Jump id;
        id.mult = u128(0, 1);
        id.plus = u128(0, 0);
        table[0] = id;
    }
    rng.lane_state = pcg_apply(mine, base);  // draw k is the output after k+1 steps
}

// ---- step_batch: one PSRS.step per rollout, cursors stay in global memory ----
template <typename PL, typename PROB>
__global__ void __launch_bounds__(256) k_step_batch(offsim_table t, offsim_rollouts ro, const PROB *__restrict__ p_new,
                                                    int reject_mode, int advance, uint32_t max_pop,
                                                    int32_t *__restrict__ out_row, int32_t *__restrict__ out_status,
                                                    uint32_t *__restrict__ out_popped) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), lane = threadIdx.x & (WAVE - 1);  // uniform -> SGPR
    const int r = blockIdx.x * (blockDim.x / WAVE) + wave;
    if (r >= ro.R) return;
    int slot = ro.cur_slot[r];
    if (slot < 0) {
        if (lane == 0) {
            if (out_row) out_row[r] = -1;
            if (out_status) out_status[r] = OFFSIM_ST_INACTIVE;
            if (out_popped) out_popped[r] = 0;
        }
        return;
    }
    Jump *table = (Jump *)lds_raw + wave * (WAVE + 1);
    WaveRng rng;
    U128 base = u128(ro.rng[4 * r + 0], ro.rng[4 * r + 1]);
    U128 inc = u128(ro.rng[4 * r + 2], ro.rng[4 * r + 3]);
    rng.kind = ro.rng_kind;
    if (reject_mode != OFFSIM_REJECT_NEVER) wave_rng_init(rng, table, base, inc, ro.rng_kind);
    uint64_t consumed = 0;
    const uint32_t *perm_row = ro.perm ? ro.perm + (int64_t)r * ro.perm_stride : nullptr;
    StepResult s = psrs_step<PL, PROB>(t, t.seg_off, perm_row, slot, ro.cursor + (int64_t)r * t.n_slots,
                                       p_new + (int64_t)r * t.nA, reject_mode, max_pop, rng, consumed);
    if (lane == 0) {
        if (consumed && ro.rng_kind == OFFSIM_STREAM_PHILOX) {
            ro.rng[4 * r + 1] = base.lo + consumed;
        } else if (consumed) {
            U128 nb = pcg_apply(pcg_jump(inc, consumed), base);
            ro.rng[4 * r + 0] = nb.hi;
            ro.rng[4 * r + 1] = nb.lo;
        }
        if (s.status == OFFSIM_ST_OK && advance) ro.cur_slot[r] = s.z_next;  // psrs.py:49-50
        if (out_row) out_row[r] = s.status == OFFSIM_ST_OK ? t.orig_idx[s.g] : -1;
        if (out_status) out_status[r] = s.status;
        if (out_popped) out_popped[r] = s.popped;
    }
}

// ---- eval_mc: the whole evalMC_psrs loop (psrs.py:241-271) on device, cursors in LDS ----
// TD = true adds the tabular learner of qlearn_psrs / expSARSA_psrs (psrs.py:119-239) to the loop: the rollout's
// Q[n_slots,nA] lives in LDS and is updated after every accepted step, in step order.
// ---- NumPy's global (legacy) MT19937 stream on the device, for np.random.choice among tied maxima (offsim4rl/agents/tabular.py:4-5).
// The state lives in LDS: 624 words + the position, as np.random.get_state() returns them.  Every lane runs the same code on the same
// values; the regeneration of the 624 words is spread over the wavefront's lanes.
__device__ __forceinline__ void mt_twist(volatile uint32_t *mt, int lane) {
    // new[k] = new-or-old[(k + 397) % 624] ^ twist(old[k], old-or-new[(k + 1) % 624]); in chunks of 64 consecutive k every lane reads
    // before any lane writes (one instruction each), later chunks see the words earlier chunks wrote -- the order the scalar code has
    for (int base = 0; base < 624; base += 64) {
        const int k = base + lane;
        uint32_t v = 0;
        if (k < 624) {
            const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            v = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        if (k < 624) mt[k] = v;
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}
__device__ __forceinline__ uint32_t mt_next(volatile uint32_t *mt, uint32_t &pos, int lane) {
    if (pos >= 624u) {
        mt_twist(mt, lane);
        pos = 0u;
    }
    uint32_t y = mt[pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
// RandomState.randint(0, rng + 1) (numpy/random/_bounded_integers: masked rejection on 32-bit words; nothing is drawn for rng = 0)
__device__ __forceinline__ uint32_t mt_bounded(volatile uint32_t *mt, uint32_t &pos, int lane, uint32_t rng) {
    if (rng == 0u) return 0u;
    uint32_t mask = rng;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    uint32_t v;
    do {
        v = mt_next(mt, pos, lane) & mask;
    } while (v > rng);
    return v;
}

template <typename PL, typename PROB, bool TD>
__global__ void __launch_bounds__(256, 4) k_eval_mc(offsim_table t, offsim_rollouts ro, const PROB *__restrict__ pi,
                                                 int reject_mode, double gamma, const double *__restrict__ gamma_pow,
                                                 int64_t n_gamma_pow, int64_t max_episodes, offsim_evalmc_out out, offsim_td td) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int waves = blockDim.x / WAVE;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), lane = threadIdx.x & (WAVE - 1);  // uniform -> SGPR
    const int n_slots = t.n_slots, nA = t.nA;
    // LDS carve: [jump tables][Q per wave (TD only)][pi][seg_off][cursors per wave]
    Jump *tables = (Jump *)lds_raw;
    double *q_lds = (double *)(tables + waves * (WAVE + 1)) + (TD ? (size_t)wave * n_slots * nA : 0);
    PROB *pi_lds = (PROB *)((double *)(tables + waves * (WAVE + 1)) + (TD ? (size_t)waves * n_slots * nA : 0));
    uint32_t *seg_lds = (uint32_t *)(pi_lds + (size_t)n_slots * nA);
    uint32_t *cur_lds = seg_lds + (n_slots + 1) + (size_t)wave * n_slots;
    // TD with an epsilon-greedy behaviour policy: this rollout's action distribution in its current state, rebuilt from its Q row before every step
    double *beh_lds = (double *)(((uintptr_t)(seg_lds + (n_slots + 1) + (size_t)waves * n_slots) + 7) & ~(uintptr_t)7) + (size_t)wave * nA;
    // TD: this rollout's copy of the tie-breaking MT19937 stream (624 words; the position is kept in a register)
    volatile uint32_t *mt_lds = (volatile uint32_t *)(((double *)(((uintptr_t)(seg_lds + (n_slots + 1) + (size_t)waves * n_slots) + 7) & ~(uintptr_t)7)) + (size_t)waves * nA) + (size_t)wave * 624;
    for (int i = threadIdx.x; i < n_slots * nA; i += blockDim.x) pi_lds[i] = pi[i];
    for (int i = threadIdx.x; i <= n_slots; i += blockDim.x) seg_lds[i] = t.seg_off[i];
    __syncthreads();
    const int r = blockIdx.x * waves + wave;
    if (r >= ro.R) return;
    uint32_t *cur_glb = ro.cursor + (int64_t)r * n_slots;
    for (int s = lane; s < n_slots; s += WAVE) cur_lds[s] = cur_glb[s];
    uint32_t mt_pos = 624u;
    if (TD) {
        for (int i = lane; i < n_slots * nA; i += WAVE) q_lds[i] = td.q[(int64_t)r * n_slots * nA + i];
        if (td.tie_mt) {
            for (int i = lane; i < 624; i += WAVE) mt_lds[i] = td.tie_mt[(int64_t)r * 625 + i];
            mt_pos = td.tie_mt[(int64_t)r * 625 + 624];
        }
    }

    WaveRng rng;
    U128 base = u128(ro.rng[4 * r + 0], ro.rng[4 * r + 1]);
    U128 inc = u128(ro.rng[4 * r + 2], ro.rng[4 * r + 3]);
    wave_rng_init(rng, tables + wave * (WAVE + 1), base, inc, ro.rng_kind);
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): table and cursors visible to the whole wave

    const uint32_t *perm_row = ro.perm ? ro.perm + (int64_t)r * ro.perm_stride : nullptr;
    const uint32_t *init_row = ro.init_perm ? ro.init_perm + (int64_t)r * ro.init_stride : nullptr;
    uint32_t ic = ro.init_cursor[r];
    int slot = ro.cur_slot[r];
    int64_t ep = 0, n_len = 0, steps = 0, cand = 0;
    uint64_t consumed = 0;
    double sum_g = 0.0;
    int status = OFFSIM_ST_OK;
    bool terminate = false;
    while (ep < max_episodes && !terminate) {
        // env.reset()  (psrs.py:32-37, :249-252)
        if ((int64_t)ic >= t.N0) {
            status = OFFSIM_ST_NO_INIT;
            slot = -1;
            break;
        }
        uint32_t k = init_row ? init_row[ic] : ic;
        ic++;
        slot = t.init_slot[k];
        double G = 0.0;
        int64_t tt = 0;
        bool done = false;
        while (!done) {
            const double gp = discount_at(gamma_pow, (uint64_t)n_gamma_pow, gamma, (uint64_t)tt);  // issued ahead of the step
            const PROB *p_step = pi_lds + (size_t)slot * nA;
            if (TD && td.behaviour != OFFSIM_BEHAVIOUR_FIXED) {
                // the learner's own behaviour policy on its Q row (offsim4rl/agents/tabular.py), rebuilt in LDS before every step
                const double *qs = q_lds + (size_t)slot * nA;
                double mx = qs[0];
                for (int k = 1; k < nA; k++) mx = qs[k] > mx ? qs[k] : mx;
                __builtin_amdgcn_s_waitcnt(0xc07f);
                if (td.behaviour == OFFSIM_BEHAVIOUR_EPS_GREEDY) {
                    // epsilon_greedy_policy (tabular.py:24-32; epsilon = 0: greedy_policy, :11-16): epsilon / nA everywhere and
                    // 1 - epsilon + epsilon / nA at _random_argmax -- np.random.choice(np.where(x == np.max(x))[0]), tabular.py:4-5:
                    // one bounded integer from the tie stream when several actions hold the maximum, none otherwise
                    int n_max = 0;
                    for (int k = 0; k < nA; k++) n_max += qs[k] == mx ? 1 : 0;
                    uint32_t pick = 0;
                    if (n_max > 1 && td.tie_mt) pick = mt_bounded(mt_lds, mt_pos, lane, (uint32_t)n_max - 1u);
                    int best = 0, seen = 0;
                    for (int k = 0; k < nA; k++) {
                        if (qs[k] == mx) {
                            if ((uint32_t)seen == pick) best = k;
                            seen++;
                        }
                    }
                    const double eps = td.epsilon_ep ? td.epsilon_ep[ep < td.n_sched ? ep : td.n_sched - 1] : td.epsilon;
                    const double lo = eps / (double)nA, hi = 1.0 - eps + lo;
                    for (int k = lane; k < nA; k += WAVE) beh_lds[k] = k == best ? hi : lo;
                    if (lane == 0 && td.beh_arg && steps < td.td_cap) td.beh_arg[(int64_t)r * td.td_cap + steps] = best;
                } else {
                    // soft_greedy_policy (tabular.py:18-22): uniform over np.isclose(Q[s], max) (rtol 1e-5, atol 1e-8)
                    const double tol = 1e-8 + 1e-5 * fabs(mx);
                    int n_close = 0;
                    for (int k = 0; k < nA; k++) n_close += fabs(qs[k] - mx) <= tol ? 1 : 0;
                    const double w = 1.0 / (double)n_close;
                    for (int k = lane; k < nA; k += WAVE) beh_lds[k] = fabs(qs[k] - mx) <= tol ? w : 0.0;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                p_step = (const PROB *)beh_lds;
            }
            StepResult s = psrs_step<PL, PROB>(t, seg_lds, perm_row, slot, cur_lds, p_step, reject_mode, 0u, rng, consumed);
            cand += s.popped;
            if (s.status != OFFSIM_ST_OK) {  // :257-259 (None) or KeyError
                status = s.status;
                terminate = true;
                break;
            }
            if (lane == 0) {
                if (out.trace_row && steps < out.trace_cap) out.trace_row[(int64_t)r * out.trace_cap + steps] = t.orig_idx[s.g];
                if (out.trace_pop && steps < out.trace_cap) out.trace_pop[(int64_t)r * out.trace_cap + steps] = s.popped;
            }
            if (TD) {  // psrs.py:165-168 (Q-learning) / :223 (expected SARSA); every lane computes the same update
                const int A = t.a[s.g];
                const double q_sa = q_lds[slot * nA + A];
                const double *qn = q_lds + (size_t)s.z_next * nA;
                double nxt;
                if (td.mode == OFFSIM_TD_QLEARN) {
                    nxt = qn[0];
                    for (int k = 1; k < nA; k++) nxt = qn[k] > nxt ? qn[k] : nxt;
                } else {  // Q[S_] @ pi[S_]: left-to-right here; NumPy hands it to BLAS, so parity is to rounding only
                    const PROB *pn = pi_lds + (size_t)s.z_next * nA;
                    nxt = 0.0;
                    for (int k = 0; k < nA; k++) nxt = nxt + qn[k] * (double)pn[k];
                }
                const double td_err = s.r + gamma * nxt - q_sa;
                if (lane == 0 && td.td_err && steps < td.td_cap) td.td_err[(int64_t)r * td.td_cap + steps] = td_err;
                const double alpha = td.alpha_ep ? td.alpha_ep[ep < td.n_sched ? ep : td.n_sched - 1] : td.alpha;  // alpha(episode), psrs.py:168
                __builtin_amdgcn_s_waitcnt(0xc07f);
                q_lds[slot * nA + A] = q_sa + alpha * td_err;
                if (td.q_snap && steps % td.snap_stride == 0 && steps / td.snap_stride < td.snap_cap) {  // save_Q (psrs.py:172-173)
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    double *dst = td.q_snap + ((int64_t)r * td.snap_cap + steps / td.snap_stride) * n_slots * nA;
                    for (int i = lane; i < n_slots * nA; i += WAVE) dst[i] = q_lds[i];
                }
            }
            G = G + gp * s.r;  // :262 (no FMA contraction: built with -ffp-contract=off)
            tt++;
            steps++;
            slot = s.z_next;
            done = s.done;
        }
        if (status == OFFSIM_ST_KEYERROR) break;  // the reference raises out of evalMC_psrs here
        if (lane == 0 && out.ep_len && n_len <= out.ep_cap) out.ep_len[(int64_t)r * (out.ep_cap + 1) + n_len] = (int32_t)tt;
        n_len++;  // :265 lengths.append(t) always
        if (done) {  // :266-269
            if (lane == 0 && out.ep_g && ep < out.ep_cap) out.ep_g[(int64_t)r * out.ep_cap + ep] = G;
            sum_g += G;
            ep++;
        } else if (lane == 0 && out.ep_g && ep < out.ep_cap) {
            // return of the episode cut short by exhaustion: evalMC_psrs drops it (:266), qlearn_psrs / expSARSA_psrs
            // append it (psrs.py:177, :232); stored past the n_ep completed ones
            out.ep_g[(int64_t)r * out.ep_cap + ep] = G;
        }
    }
    // write back the env state so that a later call continues where this one stopped
    __builtin_amdgcn_s_waitcnt(0xc07f);
    for (int s = lane; s < n_slots; s += WAVE) cur_glb[s] = cur_lds[s];
    if (TD) {
        for (int i = lane; i < n_slots * nA; i += WAVE) td.q[(int64_t)r * n_slots * nA + i] = q_lds[i];
        if (td.tie_mt) {
            for (int i = lane; i < 624; i += WAVE) td.tie_mt[(int64_t)r * 625 + i] = mt_lds[i];
            if (lane == 0) td.tie_mt[(int64_t)r * 625 + 624] = mt_pos;
        }
    }
    if (lane == 0) {
        ro.init_cursor[r] = ic;
        ro.cur_slot[r] = slot;
        if (consumed && ro.rng_kind == OFFSIM_STREAM_PHILOX) {
            ro.rng[4 * r + 1] = base.lo + consumed;
        } else if (consumed) {
            U128 nb = pcg_apply(pcg_jump(inc, consumed), base);
            ro.rng[4 * r + 0] = nb.hi;
            ro.rng[4 * r + 1] = nb.lo;
        }
        out.sum_g[r] = sum_g;
        out.n_ep[r] = ep;
        out.steps[r] = steps;
        out.cand[r] = cand;
        out.n_len[r] = n_len;
        out.status[r] = status;
    }
}

static size_t evalmc_lds_bytes(int waves, int n_slots, int nA, size_t prob_bytes, bool td = false) {
    return (size_t)waves * (WAVE + 1) * sizeof(Jump) + (td ? (size_t)waves * n_slots * nA * 8 : 0) + (size_t)n_slots * nA * prob_bytes +
           (size_t)(n_slots + 1) * 4 + (size_t)waves * n_slots * 4 + (td ? 8 + (size_t)waves * nA * 8 + (size_t)waves * 624 * 4 : 0);
}

static int check_table(const offsim_table *t) {
    if (!t) return fail(OFFSIM_EINVAL, "table is NULL%s");
    if (t->N < 0 || t->n_slots <= 0 || t->nA <= 0) return fail(OFFSIM_EINVAL, "table: bad N / n_slots / nA%s");
    if (t->N > 0 && (!t->seg_off || !t->p_log || !t->a || !t->r || !t->z_next || !t->done || !t->orig_idx))
        return fail(OFFSIM_EINVAL, "table: NULL column%s");
    if (t->plog_dtype != OFFSIM_F32 && t->plog_dtype != OFFSIM_F64 && t->plog_dtype != OFFSIM_F16)
        return fail(OFFSIM_EINVAL, "table: bad plog_dtype%s");
    if (t->r_dtype != OFFSIM_F32 && t->r_dtype != OFFSIM_F64) return fail(OFFSIM_EINVAL, "table: bad r_dtype%s");
    return OFFSIM_OK;
}

extern "C" int offsim_step_batch(const offsim_table *t, offsim_rollouts *ro, const void *p_new, int32_t prob_mode,
                                 int32_t reject_mode, int32_t advance, int32_t *out_row, int32_t *out_status,
                                 uint32_t *out_popped, void *stream) {
    int rc = check_table(t);
    if (rc) return rc;
    if (!ro || ro->R < 0 || !p_new) return fail(OFFSIM_EINVAL, "step_batch: bad argument%s");
    if (ro->R == 0) return OFFSIM_OK;
    if (prob_mode == OFFSIM_PROB_F32 && t->plog_dtype != OFFSIM_F32)
        return fail(OFFSIM_EINVAL, "step_batch: OFFSIM_PROB_F32 needs an f32 p_log%s");
    hipStream_t st = (hipStream_t)stream;
    const int waves = 4;
    dim3 grid((ro->R + waves - 1) / waves), block(waves * WAVE);
    size_t lds = (size_t)waves * (WAVE + 1) * sizeof(Jump);
    uint32_t max_pop = 0;
    if (advance == 0 && reject_mode == OFFSIM_REJECT_NEVER) max_pop = 1;  // "pop one candidate" primitive
#define LAUNCH_STEP(PL, PROB)                                                                                      \
    hipLaunchKernelGGL((k_step_batch<PL, PROB>), grid, block, lds, st, *t, *ro, (const PROB *)p_new, reject_mode, \
                       advance, max_pop, out_row, out_status, out_popped)
    if (prob_mode == OFFSIM_PROB_F32) LAUNCH_STEP(float, float);
    else if (t->plog_dtype == OFFSIM_F32) LAUNCH_STEP(float, double);
    else if (t->plog_dtype == OFFSIM_F64) LAUNCH_STEP(double, double);
    else LAUNCH_STEP(__half, double);
#undef LAUNCH_STEP
    LAUNCH_CHECK();
    return OFFSIM_OK;
}

// ------------------------------------------------------------------------------------------------
// Step server: PSRS.step for ONE environment without a kernel launch per call (per_state_rejection.py:85-95 is one Python call per
// simulated step; a launch + stream synchronise is ~27 us).  One wavefront stays resident and serves requests posted through a mailbox in
// host-coherent pinned memory (include/offsim.h: offsim_step_mailbox): the host writes p_new and the command, then the request number;
// the wavefront polls the request number (system-scope loads over PCIe), runs exactly the step k_step_batch runs -- same psrs_step, same
// stream arithmetic, state written back to the rollout's rows after every step -- and answers with row / status / popped and the request
// number.  It ends on command, or by itself after `idle_polls` polls without a request, so that a device-wide synchronise elsewhere in
// the process waits milliseconds at most; the host starts it again on demand.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T sys_load(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
template <typename T>
__device__ __forceinline__ void sys_store(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// The lines the next step of state `slot` will read -- segment bounds, cursor, the order entries and table rows of its next `width`
// candidates -- touched while the server has nothing to do: the request then finds them in the CU's vector cache / L2 instead of
// walking four dependent trips to memory (~0.7 us each, tools/micro/memlat.hip).  Values are not kept: the step reads them again.
#define SERVER_LOOK 8u
template <typename PL>
__device__ __forceinline__ void server_touch_ahead(const offsim_table &t, const uint32_t *__restrict__ perm_row, int slot, const uint32_t *cursor) {
    const int lane = threadIdx.x & (WAVE - 1);
    if (slot < 0 || slot >= t.n_slots) return;
    const uint32_t beg = t.seg_off[slot], len = t.seg_off[slot + 1] - beg, cur = cursor[slot];
    if (cur >= len) return;
    const uint32_t rem = len - cur, nv = rem < SERVER_LOOK ? rem : SERVER_LOOK;
    uint32_t g = beg + cur + ((uint32_t)lane < nv ? lane : 0);
    if (perm_row) g = perm_row[g];
    const PL *plog = (const PL *)t.p_log;
    uint32_t x = (uint32_t)t.a[g] + (uint32_t)t.z_next[g] + t.done[g] + (uint32_t)t.orig_idx[g];
    x += t.r_dtype == OFFSIM_F64 ? ((const uint32_t *)t.r)[2 * (int64_t)g] : ((const uint32_t *)t.r)[g];
    const unsigned char *row = (const unsigned char *)(plog + (int64_t)g * t.nA);
    x += row[0] + row[(int64_t)t.nA * sizeof(PL) - 1];  // (a row of <= 24 probabilities spans two lines at most)
    asm volatile("" ::"v"(x));
}

template <typename PL, typename PROB>
__global__ void __launch_bounds__(64) k_step_server(offsim_table t, offsim_rollouts ro, offsim_step_mailbox *mb, uint32_t idle_polls) {
    __shared__ Jump table[WAVE + 1];
    __shared__ PROB p_sh[OFFSIM_MAILBOX_MAX_ACTIONS];
    const int lane = threadIdx.x & (WAVE - 1);
    uint32_t last = sys_load(&mb->seq_out);
    if (lane == 0) sys_store(&mb->state, (uint32_t)OFFSIM_SERVER_RUNNING);
    // the environment's state stays in registers between requests (and is written through to its rows after every one: whoever stops
    // the server finds them current)
    int slot = ro.cur_slot[0];
    U128 base = u128(ro.rng[0], ro.rng[1]);
    const U128 inc = u128(ro.rng[2], ro.rng[3]);
    const uint32_t *head = (const uint32_t *)mb;
    const bool pcg = ro.rng_kind != OFFSIM_STREAM_PHILOX;
    if (pcg) {  // the jump table depends on the stream's increment only: built once, not per request
        table[lane + 1] = pcg_jump(inc, (uint64_t)lane + 1);
        if (lane == 0) {
            Jump id;
            id.mult = u128(0, 1);
            id.plus = u128(0, 0);
            table[0] = id;
        }
    }
    server_touch_ahead<PL>(t, ro.perm, slot, ro.cursor);
    for (;;) {
        // One poll = ONE read of the mailbox's first 64 bytes (lane i: dword i): request number, command, reject mode, the first
        // probabilities, and the request number once more in the line's last dword -- the host writes that one before the first, so a
        // snapshot that shows both holds the payload between them, whatever order its halves were read in.
        // FOUR polls are in flight (loads return in order): a request is seen a quarter of a PCIe read after it is posted, on average,
        // instead of half of one.
        uint32_t seq = last, polls = 0, w = 0;
#define SERVER_POLL() (lane < 16 ? sys_load(head + lane) : 0u)
        uint32_t w0 = SERVER_POLL(), w1 = SERVER_POLL(), w2 = SERVER_POLL(), w3 = SERVER_POLL();
#define SERVER_CHECK(wk)                                                                                                  \
    {                                                                                                                     \
        w = wk;                                                                                                           \
        seq = (uint32_t)__builtin_amdgcn_readlane((int)w, 0);                                                             \
        if ((seq != last && (uint32_t)__builtin_amdgcn_readlane((int)w, 15) == seq) || ++polls > idle_polls) break;       \
        wk = SERVER_POLL();                                                                                               \
    }
        for (;;) {
            SERVER_CHECK(w0)
            SERVER_CHECK(w1)
            SERVER_CHECK(w2)
            SERVER_CHECK(w3)
        }
#undef SERVER_CHECK
#undef SERVER_POLL
        if (seq == last || (uint32_t)__builtin_amdgcn_readlane((int)w, 15) != seq) break;  // idle: end (the host starts the server again with its next request)
        const uint32_t cmd = (uint32_t)__builtin_amdgcn_readlane((int)w, 1);
        const int reject_mode = __builtin_amdgcn_readlane((int)w, 2);
        if (cmd == OFFSIM_SERVER_CMD_EXIT) {
            if (lane == 0) sys_store(&mb->seq_out, seq);
            break;
        }
        int32_t row = -1, status = OFFSIM_ST_INACTIVE;
        uint32_t popped = 0;
        if (cmd == OFFSIM_SERVER_CMD_RESET) {  // PSRS.reset (psrs.py:32-37), as k_env_reset
            const uint32_t ic = ro.init_cursor[0];
            status = OFFSIM_ST_OK;
            if ((int64_t)ic >= t.N0) {
                slot = -1;
            } else {
                const uint32_t k = ro.init_perm ? ro.init_perm[ic] : ic;
                if (lane == 0) ro.init_cursor[0] = ic + 1;
                slot = t.init_slot[k];
                row = t.init_orig[k];
            }
            if (lane == 0) ro.cur_slot[0] = slot;
        } else {
            // probabilities: the first five doubles (ten floats) came with the poll, the others are read now
            constexpr int in_head = sizeof(PROB) == 8 ? 5 : 10;
            if (sizeof(PROB) == 8) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane * 2 + 4) & 63) * 4, (int)w);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane * 2 + 5) & 63) * 4, (int)w);
                if (lane < in_head && lane < t.nA) ((double *)p_sh)[lane] = __hiloint2double((int)hi, (int)lo);
            } else {
                const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane + 4) & 63) * 4, (int)w);
                if (lane < in_head && lane < t.nA) ((float *)p_sh)[lane] = __uint_as_float(v);
            }
            if (lane >= in_head && lane < t.nA) p_sh[lane] = sys_load((const PROB *)mb->p_tail + (lane - in_head));
            __builtin_amdgcn_s_waitcnt(0xc07f);
            const int advance = cmd == OFFSIM_SERVER_CMD_STEP ? 1 : 0;
            const uint32_t max_pop = cmd == OFFSIM_SERVER_CMD_POP_ONE ? 1u : 0u;
            if (slot >= 0) {
                WaveRng rng;
                rng.kind = ro.rng_kind;
                rng.table = table;
                rng.seed = base.hi;  // (as wave_rng_init, without filling the table again)
                rng.c = base.lo;
                if (pcg) rng.lane_state = pcg_apply(table[lane + 1], base);
                uint64_t consumed = 0;
                const StepResult s = psrs_step<PL, PROB>(t, t.seg_off, ro.perm, slot, ro.cursor, p_sh, reject_mode, max_pop, rng, consumed, SERVER_LOOK);
                if (consumed && !pcg) {
                    base.lo += consumed;
                    if (lane == 0) ro.rng[1] = base.lo;
                } else if (consumed) {
                    base = pcg_apply(consumed <= WAVE ? table[consumed] : pcg_jump(inc, consumed), base);
                    if (lane == 0) {
                        ro.rng[0] = base.hi;
                        ro.rng[1] = base.lo;
                    }
                }
                if (s.status == OFFSIM_ST_OK && advance) {  // psrs.py:49-50
                    slot = s.z_next;
                    if (lane == 0) ro.cur_slot[0] = slot;
                }
                row = s.status == OFFSIM_ST_OK ? t.orig_idx[s.g] : -1;
                status = s.status;
                popped = s.popped;
            }
        }
        if (lane == 0) {  // the answer: one 16-byte store (request number, row, status, popped)
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 ans = {seq, (uint32_t)row, (uint32_t)status, popped};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(&mb->seq_out), "v"(ans) : "memory");
        }
        server_touch_ahead<PL>(t, ro.perm, slot, ro.cursor);  // (behind the answer: the caller is busy with it for a few microseconds)
        last = seq;
    }
    __threadfence_system();
    if (lane == 0) sys_store(&mb->state, (uint32_t)OFFSIM_SERVER_EXITED);
}

extern "C" int offsim_host_alloc(int64_t bytes, void **host_ptr) {
    if (bytes <= 0 || !host_ptr) return fail(OFFSIM_EINVAL, "host_alloc: bad argument%s");
    void *p = nullptr;
    HIP_TRY(hipHostMalloc(&p, (size_t)bytes, hipHostMallocCoherent | hipHostMallocMapped));
    memset(p, 0, (size_t)bytes);
    *host_ptr = p;
    return OFFSIM_OK;
}

extern "C" int offsim_host_free(void *host_ptr) {
    if (host_ptr) HIP_TRY(hipHostFree(host_ptr));
    return OFFSIM_OK;
}

// Host side of one request to the resident step server (the protocol of include/offsim.h, in C: a Python caller pays one foreign
// call per PSRS.step instead of a dozen attribute stores and a polling loop in the interpreter).
extern "C" int offsim_step_server_call(offsim_step_mailbox *mb, const void *p_new, int32_t n_actions, int32_t prob_mode, uint32_t cmd,
                                       int32_t reject_mode, uint64_t max_spins, int32_t *out3) {
    if (!mb || !out3 || n_actions < 0 || n_actions > OFFSIM_MAILBOX_MAX_ACTIONS) return fail(OFFSIM_EINVAL, "step_server_call: bad argument%s");
    volatile offsim_step_mailbox *m = mb;
    if (p_new && n_actions > 0) {
        const size_t item = prob_mode == OFFSIM_PROB_F32 ? 4 : 8, head_bytes = sizeof(mb->p_head), total = item * (size_t)n_actions;
        memcpy((void *)mb->p_head, p_new, total < head_bytes ? total : head_bytes);
        if (total > head_bytes) memcpy((void *)mb->p_tail, (const char *)p_new + head_bytes, total - head_bytes);
    }
    if (m->cmd != cmd) m->cmd = cmd;
    if (m->reject_mode != reject_mode) m->reject_mode = reject_mode;
    const uint32_t seq = m->seq_in + 1u;
    __atomic_store_n(&mb->seq_in2, seq, __ATOMIC_RELEASE);  // behind the payload
    __atomic_store_n(&mb->seq_in, seq, __ATOMIC_RELEASE);   // ... and last
    // The wait is bounded by polls (max_spins) AND by wall-clock time, looked at every 65536 polls: OFFSIM_SERVER_ANSWER_SECONDS, or what
    // the environment variable of that name says (0: no bound).  The clock runs only while the server is RUNNING: a launch that is still
    // queued behind other work of a shared device (state STARTING, or not yet started) is not a dead server.
    static const double answer_s = getenv("OFFSIM_SERVER_ANSWER_SECONDS") ? atof(getenv("OFFSIM_SERVER_ANSWER_SECONDS")) : OFFSIM_SERVER_ANSWER_SECONDS;
    uint64_t spins = 0;
    struct timespec t_start = {0, 0};
    while (__atomic_load_n(&mb->seq_out, __ATOMIC_ACQUIRE) != seq) {
        __builtin_ia32_pause();
        if ((++spins & 1023u) == 0) {
            const uint32_t state = m->state;
            if (state == OFFSIM_SERVER_EXITED && __atomic_load_n(&mb->seq_out, __ATOMIC_ACQUIRE) != seq) return OFFSIM_SERVER_GONE;
            if (max_spins && spins > max_spins) return fail(OFFSIM_EHIP, "step_server_call: the resident step server does not answer%s");
            if ((spins & 65535u) == 0 && answer_s > 0.0) {
                struct timespec now;
                clock_gettime(CLOCK_MONOTONIC, &now);
                if (state != OFFSIM_SERVER_RUNNING || (t_start.tv_sec == 0 && t_start.tv_nsec == 0)) t_start = now;
                else if ((double)(now.tv_sec - t_start.tv_sec) + 1e-9 * (double)(now.tv_nsec - t_start.tv_nsec) > answer_s)
                    return fail(OFFSIM_EHIP, "step_server_call: the resident step server did not answer in time%s");
            }
        }
    }
    out3[0] = m->row;
    out3[1] = m->status;
    out3[2] = (int32_t)m->popped;
    return OFFSIM_OK;
}

extern "C" int offsim_step_server_start(const offsim_table *t, offsim_rollouts *ro, offsim_step_mailbox *mailbox, int32_t prob_mode,
                                        uint32_t idle_polls, void *stream) {
    int rc = check_table(t);
    if (rc) return rc;
    if (!ro || ro->R != 1 || !mailbox) return fail(OFFSIM_EINVAL, "step_server_start: one rollout and a mailbox%s");
    if (ro->perm && ro->perm_stride < 0) return fail(OFFSIM_EINVAL, "step_server_start: bad perm%s");
    if (t->nA > OFFSIM_MAILBOX_MAX_ACTIONS) return fail(OFFSIM_EUNSUPPORTED, "step_server_start: more actions than the mailbox holds (use offsim_step_batch)%s");
    if (prob_mode == OFFSIM_PROB_F32 && t->plog_dtype != OFFSIM_F32) return fail(OFFSIM_EINVAL, "step_server_start: OFFSIM_PROB_F32 needs an f32 p_log%s");
    void *dev_mb = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dev_mb, mailbox, 0));
    mailbox->state = OFFSIM_SERVER_STARTING;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_SERVER(PL, PROB) hipLaunchKernelGGL((k_step_server<PL, PROB>), dim3(1), dim3(WAVE), 0, st, *t, *ro, (offsim_step_mailbox *)dev_mb, idle_polls)
    if (prob_mode == OFFSIM_PROB_F32) LAUNCH_SERVER(float, float);
    else if (t->plog_dtype == OFFSIM_F32) LAUNCH_SERVER(float, double);
    else if (t->plog_dtype == OFFSIM_F64) LAUNCH_SERVER(double, double);
    else LAUNCH_SERVER(__half, double);
#undef LAUNCH_SERVER
    LAUNCH_CHECK();
    return OFFSIM_OK;
}

extern "C" int offsim_eval_mc(const offsim_table *t, offsim_rollouts *ro, const void *pi, int32_t prob_mode,
                              int32_t reject_mode, double gamma, const double *gamma_pow, int64_t n_gamma_pow,
                              int64_t max_episodes, const offsim_evalmc_out *out, void *stream) {
    int rc = check_table(t);
    if (rc) return rc;
    if (!ro || ro->R < 0 || !pi || !out) return fail(OFFSIM_EINVAL, "eval_mc: bad argument%s");
    if (!out->sum_g || !out->n_ep || !out->steps || !out->cand || !out->n_len || !out->status)
        return fail(OFFSIM_EINVAL, "eval_mc: required output is NULL%s");
    if (ro->R == 0) return OFFSIM_OK;
    if (prob_mode == OFFSIM_PROB_F32 && t->plog_dtype != OFFSIM_F32)
        return fail(OFFSIM_EINVAL, "eval_mc: OFFSIM_PROB_F32 needs an f32 p_log%s");
    if (n_gamma_pow > 0 && !gamma_pow) return fail(OFFSIM_EINVAL, "eval_mc: gamma_pow is NULL%s");
    hipStream_t st = (hipStream_t)stream;
    size_t pb = prob_mode == OFFSIM_PROB_F32 ? 4 : 8;
    int waves = 4;
    while (waves > 1 && evalmc_lds_bytes(waves, t->n_slots, t->nA, pb) > 64 * 1024) waves >>= 1;
    size_t lds = evalmc_lds_bytes(waves, t->n_slots, t->nA, pb);
    if (lds > 160 * 1024) return fail(OFFSIM_EUNSUPPORTED, "eval_mc: per-state cursors and policy exceed 160 KiB of LDS%s");
    dim3 grid((ro->R + waves - 1) / waves), block(waves * WAVE);
#define LAUNCH_MC(PL, PROB)                                                                                          \
    do {                                                                                                             \
        if (lds > 64 * 1024)                                                                                         \
            HIP_TRY(allow_big_lds((k_eval_mc<PL, PROB, false>), (int)lds)); \
        hipLaunchKernelGGL((k_eval_mc<PL, PROB, false>), grid, block, lds, st, *t, *ro, (const PROB *)pi, reject_mode, gamma, \
                           gamma_pow, n_gamma_pow, max_episodes, *out, offsim_td{});                                 \
    } while (0)
    if (prob_mode == OFFSIM_PROB_F32) LAUNCH_MC(float, float);
    else if (t->plog_dtype == OFFSIM_F32) LAUNCH_MC(float, double);
    else if (t->plog_dtype == OFFSIM_F64) LAUNCH_MC(double, double);
    else LAUNCH_MC(__half, double);
#undef LAUNCH_MC
    LAUNCH_CHECK();
    return OFFSIM_OK;
}

// ---- one driver iteration of the batched evaluator in ONE launch (VectorPSRS.step_and_reset): PSRS.step with the environment's own
// p_new (per_state_rejection.py:85-95), the payload of the served row (action, next observation, reward, done -> the caller's step
// buffers), and for the environments whose episode ended PSRS.reset (psrs.py:32-37) with the initial observation written over the next
// one.  One wavefront per environment, as k_step_batch; the lanes share the copies.
__device__ __forceinline__ void vec_copy_cols(const VecCols &cols, int n_cols, int64_t src_row, int64_t dst_row, bool ok, int lane) {
    for (int c = 0; c < n_cols; c++) {
        const int64_t nb = cols.c[c].row_bytes;
        unsigned char *d = (unsigned char *)cols.c[c].dst + dst_row * nb;
        if (ok) {
            const unsigned char *sp = (const unsigned char *)cols.c[c].src + src_row * nb;
            if (((nb | (int64_t)(uintptr_t)sp | (int64_t)(uintptr_t)d) & 3) == 0) {
                for (int64_t b = 4 * lane; b < nb; b += 4 * WAVE) *(uint32_t *)(d + b) = *(const uint32_t *)(sp + b);
            } else {
                for (int64_t b = lane; b < nb; b += WAVE) d[b] = sp[b];
            }
        } else if (cols.c[c].zero_if_not_ok) {
            for (int64_t b = lane; b < nb; b += WAVE) d[b] = 0;
        }
    }
}

template <typename PL, typename PROB>
__global__ void __launch_bounds__(256) k_vector_step(offsim_table t, offsim_rollouts ro, const PROB *__restrict__ p_new, int reject_mode,
                                                     VecCols step_cols, int n_step, VecCols reset_cols, int n_reset,
                                                     uint8_t *__restrict__ alive, int32_t *__restrict__ out_row, int32_t *__restrict__ out_status) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), lane = threadIdx.x & (WAVE - 1);
    const int r = blockIdx.x * (blockDim.x / WAVE) + wave;
    if (r >= ro.R) return;
    const int slot = ro.cur_slot[r];
    if (slot < 0) {  // (no current state: reset() returned None earlier, or the environment was never reset)
        if (lane == 0) {
            if (out_row) out_row[r] = -1;
            if (out_status) out_status[r] = OFFSIM_ST_INACTIVE;
            if (alive) alive[r] = 0;
        }
        vec_copy_cols(step_cols, n_step, 0, r, false, lane);
        return;
    }
    Jump *table = (Jump *)lds_raw + wave * (WAVE + 1);
    WaveRng rng;
    const U128 base = u128(ro.rng[4 * r + 0], ro.rng[4 * r + 1]), inc = u128(ro.rng[4 * r + 2], ro.rng[4 * r + 3]);
    rng.kind = ro.rng_kind;
    if (reject_mode != OFFSIM_REJECT_NEVER) wave_rng_init(rng, table, base, inc, ro.rng_kind);
    uint64_t consumed = 0;
    const uint32_t *perm_row = ro.perm ? ro.perm + (int64_t)r * ro.perm_stride : nullptr;
    const StepResult s = psrs_step<PL, PROB>(t, t.seg_off, perm_row, slot, ro.cursor + (int64_t)r * t.n_slots, p_new + (int64_t)r * t.nA,
                                             reject_mode, 0u, rng, consumed);
    const bool ok = s.status == OFFSIM_ST_OK;
    const int32_t row = ok ? t.orig_idx[s.g] : -1;
    vec_copy_cols(step_cols, n_step, row, r, ok, lane);
    bool live = ok;
    int32_t next_slot = ok ? s.z_next : slot;  // (a None step leaves the state where it was, psrs.py:44-45)
    if (ok && s.done) {  // the caller's loop: `if done: obs = env.reset()` (examples/cartpole/psrs_from_expert_heuristic.py:76-80)
        const uint32_t ic = ro.init_cursor[r];
        if ((int64_t)ic >= t.N0) {
            next_slot = -1;
            live = false;
        } else {
            const uint32_t k = ro.init_perm ? ro.init_perm[(int64_t)r * ro.init_stride + ic] : ic;
            if (lane == 0) ro.init_cursor[r] = ic + 1;
            next_slot = t.init_slot[k];
            vec_copy_cols(reset_cols, n_reset, t.init_orig[k], r, true, lane);
        }
    }
    if (lane == 0) {
        if (consumed && ro.rng_kind == OFFSIM_STREAM_PHILOX) {
            ro.rng[4 * r + 1] = base.lo + consumed;
        } else if (consumed) {
            const U128 nb = pcg_apply(pcg_jump(inc, consumed), base);
            ro.rng[4 * r + 0] = nb.hi;
            ro.rng[4 * r + 1] = nb.lo;
        }
        ro.cur_slot[r] = next_slot;
        if (alive) alive[r] = (uint8_t)(alive[r] && live);
        if (out_row) out_row[r] = row;
        if (out_status) out_status[r] = s.status;
    }
}

extern "C" int offsim_vector_step(const offsim_table *t, offsim_rollouts *ro, const void *p_new, int32_t prob_mode, int32_t reject_mode,
                                  const offsim_column *step_cols, int32_t n_step_cols, const offsim_column *reset_cols, int32_t n_reset_cols,
                                  uint8_t *alive, int32_t *out_row, int32_t *out_status, void *stream) {
    int rc = check_table(t);
    if (rc) return rc;
    if (!ro || ro->R < 0 || !p_new || n_step_cols < 0 || n_step_cols > 8 || n_reset_cols < 0 || n_reset_cols > 8 ||
        (n_step_cols > 0 && !step_cols) || (n_reset_cols > 0 && !reset_cols))
        return fail(OFFSIM_EINVAL, "vector_step: bad argument%s");
    if (prob_mode == OFFSIM_PROB_F32 && t->plog_dtype != OFFSIM_F32) return fail(OFFSIM_EINVAL, "vector_step: OFFSIM_PROB_F32 needs an f32 p_log%s");
    VecCols sc, rcols;
    memset(&sc, 0, sizeof(sc));
    memset(&rcols, 0, sizeof(rcols));
    for (int c = 0; c < n_step_cols; c++) {
        if (!step_cols[c].src || !step_cols[c].dst || step_cols[c].row_bytes <= 0) return fail(OFFSIM_EINVAL, "vector_step: bad column%s");
        sc.c[c] = step_cols[c];
    }
    for (int c = 0; c < n_reset_cols; c++) {
        if (!reset_cols[c].src || !reset_cols[c].dst || reset_cols[c].row_bytes <= 0) return fail(OFFSIM_EINVAL, "vector_step: bad column%s");
        rcols.c[c] = reset_cols[c];
    }
    if (ro->R == 0) return OFFSIM_OK;
    hipStream_t st = (hipStream_t)stream;
    const int waves = 4;
    dim3 grid((ro->R + waves - 1) / waves), block(waves * WAVE);
    const size_t lds = (size_t)waves * (WAVE + 1) * sizeof(Jump);
#define LAUNCH_VS(PL, PROB)                                                                                                         \
    hipLaunchKernelGGL((k_vector_step<PL, PROB>), grid, block, lds, st, *t, *ro, (const PROB *)p_new, reject_mode, sc, n_step_cols, rcols, \
                       n_reset_cols, alive, out_row, out_status)
    if (prob_mode == OFFSIM_PROB_F32) LAUNCH_VS(float, float);
    else if (t->plog_dtype == OFFSIM_F32) LAUNCH_VS(float, double);
    else if (t->plog_dtype == OFFSIM_F64) LAUNCH_VS(double, double);
    else LAUNCH_VS(__half, double);
#undef LAUNCH_VS
    LAUNCH_CHECK();
    return OFFSIM_OK;
}

// ---- PSRS_Exo.step (psrs.py:99-117): two queue families, endogenous state s and exogenous state x.  Every candidate
// pops the head of BOTH s_queues[s] and x_queues[x]; the rejection test reads the s-row (a, p_log); the accepted
// candidate's s-row gives (r, s', done) and its x-row gives x'.  Lane k tests candidate k of both queues with draw c+k.
template <typename PL, typename PROB>
__global__ void __launch_bounds__(256) k_step_exo(offsim_table ts, offsim_table tx, offsim_rollouts rs, offsim_rollouts rx,
                                                  const PROB *__restrict__ p_new, int32_t *__restrict__ out_row_s,
                                                  int32_t *__restrict__ out_row_x, int32_t *__restrict__ out_status,
                                                  uint32_t *__restrict__ out_popped) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), lane = threadIdx.x & (WAVE - 1);
    const int r = blockIdx.x * (blockDim.x / WAVE) + wave;
    if (r >= rs.R) return;
    const int ss = rs.cur_slot[r], xs = rx.cur_slot[r];
    int status = OFFSIM_ST_OK;
    uint32_t popped = 0;
    int32_t row_s = -1, row_x = -1;
    if (ss < 0 || xs < 0) status = OFFSIM_ST_INACTIVE;
    else if (ss >= ts.n_slots || xs >= tx.n_slots) status = OFFSIM_ST_KEYERROR;
    if (status == OFFSIM_ST_OK) {
        const uint32_t beg_s = ts.seg_off[ss], len_s = ts.seg_off[ss + 1] - beg_s;
        const uint32_t beg_x = tx.seg_off[xs], len_x = tx.seg_off[xs + 1] - beg_x;
        if (len_s == 0 || len_x == 0) status = OFFSIM_ST_KEYERROR;  // self.s_queues[s] / self.x_queues[x] raise KeyError
        else {
            Jump *table = (Jump *)lds_raw + wave * (WAVE + 1);
            WaveRng rng;
            const U128 base = u128(rs.rng[4 * r + 0], rs.rng[4 * r + 1]);
            const U128 inc = u128(rs.rng[4 * r + 2], rs.rng[4 * r + 3]);
            wave_rng_init(rng, table, base, inc);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            uint32_t cur_s = rs.cursor[(int64_t)r * ts.n_slots + ss], cur_x = rx.cursor[(int64_t)r * tx.n_slots + xs];
            const uint32_t *perm_s = rs.perm ? rs.perm + (int64_t)r * rs.perm_stride : nullptr;
            const uint32_t *perm_x = rx.perm ? rx.perm + (int64_t)r * rx.perm_stride : nullptr;
            const PL *plog = (const PL *)ts.p_log;
            const PROB *pn = p_new + (int64_t)r * ts.nA;
            uint64_t consumed = 0;
            int32_t ns = -1, nx = -1;
            for (;;) {
                const uint32_t rem_s = len_s - cur_s, rem_x = len_x - cur_x;
                const uint32_t rem = rem_s < rem_x ? rem_s : rem_x;
                if (rem == 0) {  // psrs.py:104-107
                    status = OFFSIM_ST_EXHAUSTED;
                    break;
                }
                const uint32_t nv = rem < WAVE ? rem : WAVE;
                const bool valid = (uint32_t)lane < nv;
                uint32_t gs = beg_s + cur_s + (valid ? lane : 0), gx = beg_x + cur_x + (valid ? lane : 0);
                if (perm_s) gs = perm_s[gs];
                if (perm_x) gx = perm_x[gx];
                const int a = ts.a[gs];
                const uint64_t k53 = pcg_output(rng.lane_state) >> 11;
                bool rej;
                if constexpr (sizeof(PROB) == 4) rej = rejects_f32((const float *)plog, (int64_t)gs, a, (const float *)pn, ts.nA, k53);
                else rej = rejects_f64<PL>(plog, (int64_t)gs, a, (const double *)pn, ts.nA, k53);
                const uint64_t m = __ballot(valid && !rej);
                const int f = m ? __ffsll((unsigned long long)m) - 1 : -1;
                const uint32_t d = f < 0 ? nv : (uint32_t)f + 1;
                cur_s += d;
                cur_x += d;
                popped += d;
                consumed += d;
                rng.lane_state = pcg_apply(rng.table[d], rng.lane_state);
                if (f >= 0) {
                    const int gsf = __builtin_amdgcn_readlane((int)gs, f), gxf = __builtin_amdgcn_readlane((int)gx, f);
                    row_s = ts.orig_idx[gsf];
                    row_x = tx.orig_idx[gxf];
                    ns = ts.z_next[gsf];
                    nx = tx.z_next[gxf];
                    break;
                }
            }
            if (lane == 0) {
                rs.cursor[(int64_t)r * ts.n_slots + ss] = cur_s;
                rx.cursor[(int64_t)r * tx.n_slots + xs] = cur_x;
                if (consumed) {
                    U128 nb = pcg_apply(pcg_jump(inc, consumed), base);
                    rs.rng[4 * r + 0] = nb.hi;
                    rs.rng[4 * r + 1] = nb.lo;
                }
                if (status == OFFSIM_ST_OK) {  // psrs.py:116
                    rs.cur_slot[r] = ns;
                    rx.cur_slot[r] = nx;
                }
            }
        }
    }
    if (lane == 0) {
        if (out_row_s) out_row_s[r] = status == OFFSIM_ST_OK ? row_s : -1;
        if (out_row_x) out_row_x[r] = status == OFFSIM_ST_OK ? row_x : -1;
        if (out_status) out_status[r] = status;
        if (out_popped) out_popped[r] = popped;
    }
}

extern "C" int offsim_step_exo(const offsim_table *ts, const offsim_table *tx, offsim_rollouts *rs, offsim_rollouts *rx,
                               const void *p_new, int32_t prob_mode, int32_t *out_row_s, int32_t *out_row_x,
                               int32_t *out_status, uint32_t *out_popped, void *stream) {
    int rc = check_table(ts);
    if (rc) return rc;
    rc = check_table(tx);
    if (rc) return rc;
    if (!rs || !rx || rs->R != rx->R || rs->R < 0 || !p_new) return fail(OFFSIM_EINVAL, "step_exo: bad argument%s");
    if (rs->R == 0) return OFFSIM_OK;
    if (prob_mode == OFFSIM_PROB_F32 && ts->plog_dtype != OFFSIM_F32) return fail(OFFSIM_EINVAL, "step_exo: OFFSIM_PROB_F32 needs an f32 p_log%s");
    hipStream_t st = (hipStream_t)stream;
    const int waves = 4;
    dim3 grid((rs->R + waves - 1) / waves), block(waves * WAVE);
    size_t lds = (size_t)waves * (WAVE + 1) * sizeof(Jump);
#define LAUNCH_EXO(PL, PROB) \
    hipLaunchKernelGGL((k_step_exo<PL, PROB>), grid, block, lds, st, *ts, *tx, *rs, *rx, (const PROB *)p_new, out_row_s, out_row_x, out_status, out_popped)
    if (prob_mode == OFFSIM_PROB_F32) LAUNCH_EXO(float, float);
    else if (ts->plog_dtype == OFFSIM_F32) LAUNCH_EXO(float, double);
    else if (ts->plog_dtype == OFFSIM_F64) LAUNCH_EXO(double, double);
    else LAUNCH_EXO(__half, double);
#undef LAUNCH_EXO
    LAUNCH_CHECK();
    return OFFSIM_OK;
}

// qlearn_psrs / expSARSA_psrs (psrs.py:119-239) with a state-independent behaviour policy: evalMC's loop plus the
// tabular TD update, all R rollouts in one launch (generic 64-candidate step; f64 probabilities).
extern "C" int offsim_eval_td(const offsim_table *t, offsim_rollouts *ro, const double *pi, int32_t reject_mode, double gamma,
                              const double *gamma_pow, int64_t n_gamma_pow, int64_t max_episodes, const offsim_evalmc_out *out,
                              const offsim_td *td, void *stream) {
    int rc = check_table(t);
    if (rc) return rc;
    if (!ro || ro->R < 0 || !pi || !out || !td) return fail(OFFSIM_EINVAL, "eval_td: bad argument%s");
    if (!out->sum_g || !out->n_ep || !out->steps || !out->cand || !out->n_len || !out->status)
        return fail(OFFSIM_EINVAL, "eval_td: required output is NULL%s");
    if ((td->mode != OFFSIM_TD_QLEARN && td->mode != OFFSIM_TD_EXPSARSA) || !td->q) return fail(OFFSIM_EINVAL, "eval_td: bad td mode or NULL q%s");
    if (td->behaviour != OFFSIM_BEHAVIOUR_FIXED && td->behaviour != OFFSIM_BEHAVIOUR_EPS_GREEDY && td->behaviour != OFFSIM_BEHAVIOUR_SOFT_GREEDY)
        return fail(OFFSIM_EINVAL, "eval_td: bad behaviour%s");
    if ((td->alpha_ep || td->epsilon_ep) && td->n_sched <= 0) return fail(OFFSIM_EINVAL, "eval_td: schedules need n_sched > 0%s");
    if (td->q_snap && (td->snap_stride <= 0 || td->snap_cap < 0)) return fail(OFFSIM_EINVAL, "eval_td: bad snapshot stride / capacity%s");
    if (n_gamma_pow > 0 && !gamma_pow) return fail(OFFSIM_EINVAL, "eval_td: gamma_pow is NULL%s");
    if (ro->R == 0) return OFFSIM_OK;
    hipStream_t st = (hipStream_t)stream;
    int waves = 4;
    while (waves > 1 && evalmc_lds_bytes(waves, t->n_slots, t->nA, 8, true) > 64 * 1024) waves >>= 1;
    size_t lds = evalmc_lds_bytes(waves, t->n_slots, t->nA, 8, true);
    if (lds > 160 * 1024) return fail(OFFSIM_EUNSUPPORTED, "eval_td: Q table, cursors and policy exceed 160 KiB of LDS%s");
    dim3 grid((ro->R + waves - 1) / waves), block(waves * WAVE);
#define LAUNCH_TD(PL)                                                                                                   \
    do {                                                                                                                \
        if (lds > 64 * 1024)                                                                                            \
            HIP_TRY(allow_big_lds((k_eval_mc<PL, double, true>), (int)lds)); \
        hipLaunchKernelGGL((k_eval_mc<PL, double, true>), grid, block, lds, st, *t, *ro, pi, reject_mode, gamma, gamma_pow, \
                           n_gamma_pow, max_episodes, *out, *td);                                                       \
    } while (0)
    if (t->plog_dtype == OFFSIM_F32) LAUNCH_TD(float);
    else if (t->plog_dtype == OFFSIM_F64) LAUNCH_TD(double);
    else LAUNCH_TD(__half);
#undef LAUNCH_TD
    LAUNCH_CHECK();
    return OFFSIM_OK;
}

// ------------------------------------------------------------------------------------------------
// Fast evalMC scan: compiled-policy keys + LDS candidate windows (scan_win.hpp)
// ------------------------------------------------------------------------------------------------
#include "scan_win.hpp"

extern "C" int offsim_compile_policy(const offsim_table *t, const double *pi, uint64_t *keys_out, void *stream) {
    int rc = check_table(t);
    if (rc) return rc;
    if (!pi || (t->N > 0 && !keys_out)) return fail(OFFSIM_EINVAL, "compile_policy: bad argument%s");
    if (t->n_slots > 1024) return fail(OFFSIM_EUNSUPPORTED, "compile_policy: more than 1024 states do not fit the 10-bit next-state field%s");
    if (t->N == 0) return OFFSIM_OK;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((t->N + 255) / 256)), block(256);
    size_t lds = sizeof(uint32_t) * (t->n_slots + 1);
    if (t->plog_dtype == OFFSIM_F32) hipLaunchKernelGGL(k_compile_policy<float>, grid, block, lds, st, *t, pi, keys_out);
    else if (t->plog_dtype == OFFSIM_F64) hipLaunchKernelGGL(k_compile_policy<double>, grid, block, lds, st, *t, pi, keys_out);
    else hipLaunchKernelGGL(k_compile_policy<__half>, grid, block, lds, st, *t, pi, keys_out);
    LAUNCH_CHECK();
    return OFFSIM_OK;
}

// Name of the kernel offsim_eval_mc_keys launches (measurement label).  (Round 1 also had a two-wavefront variant of this
// kernel; the row-packed scan of scan_rows.hpp, with its helper wavefronts, has taken its place.)
extern "C" const char *offsim_eval_mc_keys_kernel(int32_t n_slots, int32_t R) {
    (void)R;
    return n_slots > 256 ? "" : "k_eval_mc_win";
}

extern "C" int offsim_eval_mc_keys(const offsim_table *t, offsim_rollouts *ro, const uint64_t *keys, double gamma,
                                   const double *gamma_pow, int64_t n_gamma_pow, int64_t max_episodes,
                                   const offsim_evalmc_out *out, void *stream) {
    int rc = check_table(t);
    if (rc) return rc;
    if (!ro || ro->R < 0 || !out || (t->N > 0 && !keys)) return fail(OFFSIM_EINVAL, "eval_mc_keys: bad argument%s");
    if (!out->sum_g || !out->n_ep || !out->steps || !out->cand || !out->n_len || !out->status)
        return fail(OFFSIM_EINVAL, "eval_mc_keys: required output is NULL%s");
    if (t->n_slots > 256) return fail(OFFSIM_EUNSUPPORTED, "eval_mc_keys: candidate windows support at most 256 states%s");
    if (ro->rng_kind != OFFSIM_STREAM_PCG64 && ro->rng_kind != OFFSIM_STREAM_PHILOX) return fail(OFFSIM_EINVAL, "eval_mc_keys: unknown stream provider%s");
    if (t->N >= 0xffffffffll) return fail(OFFSIM_EUNSUPPORTED, "eval_mc_keys: queue positions, candidate and step counters are 32-bit (N < 2^32)%s");
    if (n_gamma_pow > 0 && !gamma_pow) return fail(OFFSIM_EINVAL, "eval_mc_keys: gamma_pow is NULL%s");
    if (ro->R == 0) return OFFSIM_OK;
    hipStream_t st = (hipStream_t)stream;
    const int waves = 4;
    dim3 grid((ro->R + waves - 1) / waves), block(waves * 64);
    const bool trace = out->trace_row || out->trace_pop;
    const bool philox = ro->rng_kind == OFFSIM_STREAM_PHILOX;  // (the provider of the rejection stream is a template parameter, as in the row-packed scan)
    const int rounds = (t->n_slots + 63) / 64;
#define LAUNCH_WIN(W, ROUNDS)                                                                                         \
    do {                                                                                                              \
        size_t lds = 512 + (((size_t)(t->n_slots + 1) * 4 + 511) & ~(size_t)511) +                                      \
                     (size_t)waves * ((OFFSIM_RING * 4 + (size_t)t->n_slots * (W) * 4 + (size_t)t->n_slots * 16 + OFFSIM_PH * 8 + (trace ? OFFSIM_PH * 4 : 0) + 511) & ~(size_t)511); \
        if (trace && philox)                                                                                          \
            hipLaunchKernelGGL((k_eval_mc_win<W, ROUNDS, true, OFFSIM_STREAM_PHILOX>), grid, block, lds, st, *t, *ro, keys, gamma, gamma_pow, \
                               n_gamma_pow, max_episodes, *out);                                                      \
        else if (trace)                                                                                               \
            hipLaunchKernelGGL((k_eval_mc_win<W, ROUNDS, true, OFFSIM_STREAM_PCG64>), grid, block, lds, st, *t, *ro, keys, gamma, gamma_pow, \
                               n_gamma_pow, max_episodes, *out);                                                      \
        else if (philox)                                                                                              \
            hipLaunchKernelGGL((k_eval_mc_win<W, ROUNDS, false, OFFSIM_STREAM_PHILOX>), grid, block, lds, st, *t, *ro, keys, gamma, gamma_pow, \
                               n_gamma_pow, max_episodes, *out);                                                      \
        else                                                                                                          \
            hipLaunchKernelGGL((k_eval_mc_win<W, ROUNDS, false, OFFSIM_STREAM_PCG64>), grid, block, lds, st, *t, *ro, keys, gamma, gamma_pow, \
                               n_gamma_pow, max_episodes, *out);                                                      \
    } while (0)
    // window depth by state count so that 16 rollouts (4 blocks) fit one CU's 160 KiB of LDS
    if (rounds == 1) LAUNCH_WIN(32, 1);
    else if (rounds == 2) LAUNCH_WIN(8, 2);
    else if (rounds == 3) LAUNCH_WIN(8, 3);
    else LAUNCH_WIN(8, 4);
#undef LAUNCH_WIN
    LAUNCH_CHECK();
    return OFFSIM_OK;
}

// ------------------------------------------------------------------------------------------------
// Headline scan on per-rollout candidate streams (scan_rows.hpp)
// ------------------------------------------------------------------------------------------------
#include "scan_rows.hpp"

__global__ void k_key_digests(const uint64_t *__restrict__ keys, int64_t N, int format, uint32_t *__restrict__ out) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    const uint32_t hi = (uint32_t)(keys[g] >> 32);  // [T >> 32 : 21 | done | z_next : 10]
    const uint32_t ts = format == OFFSIM_STREAMS_B ? 16u : 18u;  // (formats B, C: a coarser threshold, the local row's upper bits left zero here)
    out[g] = format == OFFSIM_STREAMS_A ? hi : ((hi >> ts) << ts) | (hi & 0x400u) | (hi & 0xffu);
}

extern "C" int offsim_compile_digests(const offsim_table *t, const uint64_t *keys, int32_t format, uint32_t *dig32_out, void *stream) {
    int rc = check_table(t);
    if (rc) return rc;
    if (t->N > 0 && (!keys || !dig32_out)) return fail(OFFSIM_EINVAL, "compile_digests: bad argument%s");
    if (format != OFFSIM_STREAMS_A && format != OFFSIM_STREAMS_B && format != OFFSIM_STREAMS_C) return fail(OFFSIM_EINVAL, "compile_digests: bad format%s");
    if (format == OFFSIM_STREAMS_B && t->n_slots > 255) return fail(OFFSIM_EUNSUPPORTED, "compile_digests: format B holds 255 states (a payload of all ones is not a digest, csrc/scan_rows.hpp rows_format)%s");
    if (format == OFFSIM_STREAMS_C && t->n_slots > 255) return fail(OFFSIM_EUNSUPPORTED, "compile_digests: format C holds 255 states%s");
    if (t->N == 0) return OFFSIM_OK;
    hipLaunchKernelGGL(k_key_digests, dim3((unsigned)((t->N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, keys, t->N, format, dig32_out);
    LAUNCH_CHECK();
    return OFFSIM_OK;
}

extern "C" int offsim_eval_mc_streams(const offsim_table *t, offsim_rollouts *ro, const offsim_streams *sm, const uint64_t *keys,
                                      double gamma, const double *gamma_pow, int64_t n_gamma_pow, int64_t max_episodes,
                                      const offsim_evalmc_out *out, void *stream) {
    int rc = check_table(t);
    if (rc) return rc;
    if (!ro || ro->R < 0 || !out || !sm || (t->N > 0 && (!keys || !sm->dig))) return fail(OFFSIM_EINVAL, "eval_mc_streams: bad argument%s");
    if (!out->sum_g || !out->n_ep || !out->steps || !out->cand || !out->n_len || !out->status)
        return fail(OFFSIM_EINVAL, "eval_mc_streams: required output is NULL%s");
    if (t->n_slots > 256) return fail(OFFSIM_EUNSUPPORTED, "eval_mc_streams: candidate windows support at most 256 states%s");
    if (ro->rng_kind != OFFSIM_STREAM_PCG64 && ro->rng_kind != OFFSIM_STREAM_PHILOX) return fail(OFFSIM_EINVAL, "eval_mc_streams: unknown stream provider%s");
    if (n_gamma_pow > 0 && !gamma_pow) return fail(OFFSIM_EINVAL, "eval_mc_streams: gamma_pow is NULL%s");
    if (t->N >= 0xffffffffll) return fail(OFFSIM_EUNSUPPORTED, "eval_mc_streams: queue positions are 32-bit (N < 2^32)%s");
    // positions inside a state's queue travel in 17-bit fields of the request descriptors and as 16-bit local rows (loc)
    if (sm->format != OFFSIM_STREAMS_A && sm->format != OFFSIM_STREAMS_B && sm->format != OFFSIM_STREAMS_C) return fail(OFFSIM_EINVAL, "eval_mc_streams: bad stream format%s");
    if (t->N > 0 && (t->max_seg <= 0 || t->max_seg > (sm->format == OFFSIM_STREAMS_B ? (1ll << 23) : sm->format == OFFSIM_STREAMS_C ? (1ll << 17) : 65536ll)))
        return fail(OFFSIM_EUNSUPPORTED, "eval_mc_streams: needs offsim_table.max_seg set and <= 65536 rows per state (format A) / 2^23 (B) / 2^17 (C)%s");
    if (sm->format != OFFSIM_STREAMS_A && !sm->loc) return fail(OFFSIM_EINVAL, "eval_mc_streams: formats B and C need the loc stream%s");
    if (sm->format != OFFSIM_STREAMS_A && t->n_slots > 255) return fail(OFFSIM_EUNSUPPORTED, "eval_mc_streams: formats B and C hold 255 states%s");
    if (ro->R == 0) return OFFSIM_OK;
    // the tick takes its queue positions with one ds_add_rtn_u32 per rollout, lane = step, and relies on the LDS serving same-address
    // lanes in ascending lane order: checked once per device, refused (not silently wrong) where the property is absent
    { const int okv = lds_order_ok_on(stream); if (okv < 0) return okv; if (okv == 0) return fail(OFFSIM_EUNSUPPORTED, "eval_mc_streams: this device's LDS does not serve same-address lanes of one instruction in lane order (offsim_lds_order_ok): use offsim_eval_mc_keys on permutations%s"); }
    hipStream_t st = (hipStream_t)stream;
    const bool trace = out->trace_row || out->trace_pop;
    // four rollouts per wavefront; as many wavefronts per workgroup (<= 4) as the CU's 160 KiB of LDS hold regions for
    const uint32_t region = rows_region_bytes((uint32_t)t->n_slots);
    const uint32_t seg_bytes = (((uint32_t)t->n_slots + 1u) * 4u + 1023u) & ~1023u;
    int waves = 4;  // (the DMA areas of the workgroup are rounded up together: rows_dma_total)
    while (waves >= 1 && 1024u + seg_bytes + rows_dma_total((uint32_t)waves) + (uint32_t)waves * 4u * region > 160u * 1024u) waves--;
    if (waves < 1) return fail(OFFSIM_EUNSUPPORTED, "eval_mc_streams: LDS too small for this state count%s");
    int cus = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    // OFFSIM_ROWS_HELPER=0 runs the single-wavefront form of the kernel (what the TRACE build always is)
    static const bool helper = !(getenv("OFFSIM_ROWS_HELPER") && atoi(getenv("OFFSIM_ROWS_HELPER")) == 0);
    // A launch that does not need every CU at four chain wavefronts of four rollouts each is SPREAD, in two ways (10 M rows, scan seconds,
    // tools/sweep_launch_shape.sh and sweep_launch_shape2.sh, profiles/r04_launch_shape_sweep*.txt):
    //  * as few chain wavefronts (+ helpers) per workgroup as still fit the device in one wave of workgroups, the workgroup asking for more
    //    than half of the CU's LDS so that no two share a CU: the kernel's time is the chain's latency whatever the number of rollouts, and
    //    a chain wavefront that shares neither its SIMD with a helper nor the LDS pipeline with three other pairs is the faster chain
    //    (2048 rollouts: two pairs per CU 0.782 s, packed four to a CU 0.853 s);
    //  * at four rollouts per CU or fewer, two rollouts to a chain wavefront instead of four (two of its rows left empty), at one per CU
    //    one: every event of a row -- an episode end, a row without a clear accept -- holds its whole wavefront (1024 rollouts: 0.696 s
    //    as two pairs of two against 0.745 as one pair of four; 512: 0.684 against 0.744; 256: 0.655 with one rollout per wavefront).
    //    At eight per CU four pairs of two and two pairs of four are level (0.773 / 0.768): four rows stay.
    // OFFSIM_ROWS_WAVES = 1..4 and OFFSIM_ROWS_PER_WAVE = 1 | 2 | 4 force a shape (A/B runs, the variant matrix of the test suite).
    static const int waves_env = getenv("OFFSIM_ROWS_WAVES") ? atoi(getenv("OFFSIM_ROWS_WAVES")) : 0;
    static const int rpw_env = getenv("OFFSIM_ROWS_PER_WAVE") ? atoi(getenv("OFFSIM_ROWS_PER_WAVE")) : 0;
    const int waves_fit = waves;
    int rpw = 4;
    const bool rpw_forced = rpw_env == 1 || rpw_env == 2 || rpw_env == 4;
    if (rpw_forced) rpw = rpw_env;
    if (waves_env > 0) {
        waves = waves_env < waves_fit ? waves_env : waves_fit;
    } else if (helper && !trace) {
        if (!rpw_forced) rpw = ro->R <= (int64_t)cus ? 1 : ro->R <= 4ll * cus ? 2 : 4;
        int w = (int)((ro->R + (int64_t)rpw * cus - 1) / ((int64_t)rpw * cus));  // chain wavefronts per CU when the rollouts are dealt out evenly
        waves = w < 1 ? 1 : w < waves_fit ? w : waves_fit;
    }
    const int rpb = waves * rpw;  // rollouts per workgroup (its LDS holds four regions per chain wavefront whatever their use)
    size_t lds = 1024 + seg_bytes + (size_t)rows_dma_total((uint32_t)waves) + (size_t)waves * 4u * region;
    if (waves < waves_fit && lds < 81u * 1024u) lds = 81u * 1024u;  // (a workgroup per CU)
    dim3 grid((unsigned)((ro->R + rpb - 1) / rpb));
    // Entries a window must lack before the helper asks for its top-up (scan_rows.hpp, request()): 2 when the launch fills the device
    // -- a third fewer requests in flight is what lets them land within one tick -- and 1 when a quarter of the CUs or more stay idle
    // (measured at 10 M rows, scan seconds with 1 / 2: 512 rollouts 0.840 / 0.865, 3072: 0.855 / 0.881, 3584: 0.895 / 0.917, 4096: 1.074 /
    // 0.899: the cliff is at the full device).  OFFSIM_ROWS_MINROOM overrides (A/B runs).
    static const int minroom_env = getenv("OFFSIM_ROWS_MINROOM") ? atoi(getenv("OFFSIM_ROWS_MINROOM")) : 0;
    const uint32_t rq_minroom = minroom_env > 0 ? (uint32_t)minroom_env : (int64_t)ro->R <= 12ll * cus ? 1u : 2u;  // (rollouts, whatever the shape of the launch)
    // (the provider of the rejection stream is a template parameter: PCG64 = the reference's numbers, PHILOX = rocRAND's device API)
#define LAUNCH_ROWS_RNG(TR, HL, FMT, RNG, THREADS)                                                                                 \
    do {                                                                                                                           \
        HIP_TRY(allow_big_lds((k_eval_mc_rows<TR, HL, FMT, RNG>), 160 * 1024));                                                     \
        hipLaunchKernelGGL((k_eval_mc_rows<TR, HL, FMT, RNG>), grid, dim3((unsigned)(THREADS)), lds, st, *t, *ro, *sm, keys, gamma, gamma_pow, \
                           n_gamma_pow, max_episodes, *out, seg_bytes, region, rq_minroom, (uint32_t)rpw);                                        \
    } while (0)
#define LAUNCH_ROWS(TR, HL, FMT, THREADS)                                                                                          \
    do {                                                                                                                           \
        if (ro->rng_kind == OFFSIM_STREAM_PHILOX) LAUNCH_ROWS_RNG(TR, HL, FMT, OFFSIM_STREAM_PHILOX, THREADS);                      \
        else LAUNCH_ROWS_RNG(TR, HL, FMT, OFFSIM_STREAM_PCG64, THREADS);                                                            \
    } while (0)
    if (sm->format == OFFSIM_STREAMS_B) {
        if (trace) LAUNCH_ROWS(true, false, OFFSIM_STREAMS_B, waves * 64);
        else if (!helper) LAUNCH_ROWS(false, false, OFFSIM_STREAMS_B, waves * 64);
        else LAUNCH_ROWS(false, true, OFFSIM_STREAMS_B, waves * 128);  // a helper wavefront per chain wavefront
    } else if (sm->format == OFFSIM_STREAMS_C) {
        if (trace) LAUNCH_ROWS(true, false, OFFSIM_STREAMS_C, waves * 64);
        else if (!helper) LAUNCH_ROWS(false, false, OFFSIM_STREAMS_C, waves * 64);
        else LAUNCH_ROWS(false, true, OFFSIM_STREAMS_C, waves * 128);
    } else {
        if (trace) LAUNCH_ROWS(true, false, OFFSIM_STREAMS_A, waves * 64);
        else if (!helper) LAUNCH_ROWS(false, false, OFFSIM_STREAMS_A, waves * 64);
        else LAUNCH_ROWS(false, true, OFFSIM_STREAMS_A, waves * 128);
    }
#undef LAUNCH_ROWS
#undef LAUNCH_ROWS_RNG
    LAUNCH_CHECK();
    return OFFSIM_OK;
}

// Self-test of the one hardware property csrc/scan_rows.hpp relies on beyond the ISA manual: the LDS applies the lanes of ONE
// ds_add_rtn_u32 that hit the same address in ascending lane order, so that lane i gets back base + the addends of the lower
// lanes with its address (the tick's queue positions are taken that way).  Random address patterns with 1 .. 162 distinct
// addresses, lanes sitting out at random; *mismatches (device, int64) receives the number of lanes that got anything else.
__global__ void k_selftest_lds_order(uint32_t seed0, int trials, uint32_t n_addr, unsigned long long *bad) {
    __shared__ uint32_t cell[256];
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long nbad = 0;
    uint32_t h = seed0 * 2654435761u + blockIdx.x * 97u + 1u;
    for (int t = 0; t < trials; t++) {
        for (uint32_t i = lane; i < 256; i += 64) cell[i] = 1000u * i;
        __syncthreads();
        h = h * 1664525u + 1013904223u;
        const uint32_t hl = (h ^ (lane * 0x9e3779b9u)) * 2246822519u;
        const uint32_t a = (hl >> 8) % n_addr, kk = 1u + ((hl >> 20) & 7u);
        const bool active = ((hl >> 28) & 7u) != 0u;
        uint32_t got = 0;
        if (active) {
            const uint32_t addr = (uint32_t)(uintptr_t)(offsim::lds_u32 *)&cell[a];
            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(got) : "v"(addr), "v"(kk) : "memory");
        }
        uint32_t want = 1000u * a;
        for (uint32_t j = 0; j < 64; j++) {
            const uint32_t aj = __shfl(a, j), kj = __shfl(kk, j);
            const bool actj = __shfl((int)active, j);
            if (j < lane && actj && aj == a) want += kj;
        }
        if (active && got != want) nbad++;
        __syncthreads();
        // the same for ds_wrxchg_rtn_b32 (csrc/shuffle_chunk.hpp applies a group of messages with one exchange per lane): a lane gets
        // back what the previous lane with its address stored (or the cell's initial value), and the last lane's value stays
        for (uint32_t i = lane; i < 256; i += 64) cell[i] = 1000u * i;
        __syncthreads();
        if (active) {
            const uint32_t addr = (uint32_t)(uintptr_t)(offsim::lds_u32 *)&cell[a];
            asm volatile("ds_wrxchg_rtn_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(got) : "v"(addr), "v"(lane + 1u) : "memory");
        }
        want = 1000u * a;
        uint32_t last = 1000u * a;
        for (uint32_t j = 0; j < 64; j++) {
            const uint32_t aj = __shfl(a, j);
            const bool actj = __shfl((int)active, j);
            if (actj && aj == a) {
                if (j < lane) want = j + 1u;
                last = j + 1u;
            }
        }
        __syncthreads();
        if (active && (got != want || cell[a] != last)) nbad++;
        __syncthreads();
        // ... and for ds_mskor_rtn_b32 on one HALF of a dword (round 5: the LDS-resident shuffle's applier exchanges a 16-bit entry with
        // it -- the word's other half, which another lane of the same instruction may be exchanging, must survive): a lane gets back the
        // half as the previous lane with its (word, half) left it, the last such lane's value stays, the other half is untouched by it
        for (uint32_t i = lane; i < 256; i += 64) cell[i] = (7000u + i) << 16 | (3000u + i);
        __syncthreads();
        const uint32_t half = (hl >> 5) & 1u, sh = half * 16u;
        if (active) {
            const uint32_t addr = (uint32_t)(uintptr_t)(offsim::lds_u32 *)&cell[a];
            asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=v"(got) : "v"(addr), "v"(0xffffu << sh), "v"((lane + 1u) << sh) : "memory");
            got = (got >> sh) & 0xffffu;
        }
        want = half ? 7000u + a : 3000u + a;
        last = want;
        uint32_t last_other = half ? 3000u + a : 7000u + a;  // the word's OTHER half: its initial value, or what the last lane that exchanged it stored
        for (uint32_t j = 0; j < 64; j++) {
            const uint32_t aj = __shfl(a, j), hj = __shfl(half, j);
            const bool actj = __shfl((int)active, j);
            if (actj && aj == a && hj == half) {
                if (j < lane) want = j + 1u;
                last = j + 1u;
            }
            if (actj && aj == a && hj != half) last_other = j + 1u;
        }
        __syncthreads();
        if (active && (got != want || ((cell[a] >> sh) & 0xffffu) != last || ((cell[a] >> (16u - sh)) & 0xffffu) != last_other)) nbad++;
        __syncthreads();
    }
    if (nbad) atomicAdd(bad, nbad);
}

extern "C" int offsim_selftest_lds_atomic_order(int64_t *mismatches, void *stream) {
    if (!mismatches) return fail(OFFSIM_EINVAL, "selftest: NULL output%s");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(mismatches, 0, sizeof(int64_t), st));
    const uint32_t n_addr[] = {1u, 2u, 3u, 5u, 16u, 40u, 162u};
    for (uint32_t n : n_addr) hipLaunchKernelGGL(k_selftest_lds_order, dim3(256), dim3(64), 0, st, 12345u + n, 500, n, (unsigned long long *)mismatches);
    LAUNCH_CHECK();
    return OFFSIM_OK;
}

// The same self-test as a RUNTIME GUARD (include/offsim.h): run once per device, short (3 launches of 64 wavefronts x 100 trials x the two
// instructions, ~1.2e6 lane operations), the verdict cached.  offsim_eval_mc_streams and the chunked shuffle ask it before their first
// launch on a device and refuse (OFFSIM_EUNSUPPORTED) when the LDS of this part does not serve same-address lanes in lane order, instead
// of returning wrong numbers; the host mirror routes such a device to the window kernel and the in-place shuffle.
// OFFSIM_FORCE_LDS_ORDER_MISMATCH=1 makes the verdict "absent" (tests).
static int lds_order_verdict(hipStream_t caller, bool have_caller) {
    static int verdict[64];
    static bool init = false;
    static std::mutex mu;
    std::lock_guard<std::mutex> hold(mu);
    if (!init) {
        for (int &v : verdict) v = -1;
        init = true;
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(OFFSIM_EHIP, "lds_order_ok: no device%s");
    if (verdict[dev] >= 0) return verdict[dev];
    // The first call on a device allocates, launches on a stream of its own and synchronises: not something an asynchronous entry point
    // may do while its caller's stream is being captured into a graph (the allocation and the synchronisation would invalidate the
    // capture).  Such a caller is told to ask once beforehand.
    if (have_caller) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(caller, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail(OFFSIM_EINVAL, "the stream is capturing and this device's LDS lane-order guard has not run yet: call offsim_lds_order_ok() once outside the capture%s");
    }
    const char *force = getenv("OFFSIM_FORCE_LDS_ORDER_MISMATCH");
    if (force && atoi(force) == 1) return verdict[dev] = 0;
    unsigned long long *bad = nullptr, host = 0;
    hipStream_t st = nullptr;
    if (hipMalloc((void **)&bad, sizeof(*bad)) != hipSuccess) return fail(OFFSIM_EHIP, "lds_order_ok: allocation failed%s");
    // (its own stream: the caller's stream may be capturing, and nothing of the caller's is waited for)
    bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess && hipMemsetAsync(bad, 0, sizeof(*bad), st) == hipSuccess;
    if (ok) {
        const uint32_t n_addr[] = {1u, 5u, 162u};
        for (uint32_t n : n_addr) hipLaunchKernelGGL(k_selftest_lds_order, dim3(64), dim3(64), 0, st, 777u + n, 100, n, bad);
        ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(&host, bad, sizeof(host), hipMemcpyDeviceToHost, st) == hipSuccess &&
             hipStreamSynchronize(st) == hipSuccess;
    }
    if (st) (void)hipStreamDestroy(st);
    (void)hipFree(bad);
    if (!ok) return fail(OFFSIM_EHIP, "lds_order_ok: the self-test did not run%s");
    return verdict[dev] = host == 0 ? 1 : 0;
}
extern "C" int offsim_lds_order_ok(void) { return lds_order_verdict(nullptr, false); }
// ... as the launch paths ask it: cached verdict, or the self-test now unless `stream` is capturing
static int lds_order_ok_on(void *stream) { return lds_order_verdict((hipStream_t)stream, true); }

// ------------------------------------------------------------------------------------------------
// Encoders
// ------------------------------------------------------------------------------------------------
// CartpoleBoxEncoder.get_box (heuristic.py:19-60).  float32 observations compared against the
// float32-rounded literals (what NumPy 2 does for np.float32 < python float; see oracle header).
__global__ void k_encode_box(const float4 *__restrict__ obs, int64_t N, int32_t *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float4 o = obs[i];
    const float ONE = 0.0174532f, SIX = 0.1047192f, TWELVE = 0.2094384f, FIFTY = 0.87266f;
    float x = o.x, xd = o.y, th = o.z, thd = o.w;
    int box;
    if (x < -2.4f || x > 2.4f || th < -TWELVE || th > TWELVE) {
        box = -1;
    } else {
        box = x < -0.8f ? 0 : (x < 0.8f ? 1 : 2);
        box += xd < -0.5f ? 0 : (xd < 0.5f ? 3 : 6);
        box += th < -SIX ? 0 : (th < -ONE ? 9 : (th < 0.f ? 18 : (th < ONE ? 27 : (th < SIX ? 36 : 45))));
        box += thd < -FIFTY ? 0 : (thd < FIFTY ? 54 : 108);
    }
    out[i] = box;
}
extern "C" int offsim_encode_box(const float *obs, int64_t N, int32_t *out_z, void *stream) {
    if (N < 0 || (N > 0 && (!obs || !out_z))) return fail(OFFSIM_EINVAL, "encode_box: bad argument%s");
    if (N == 0) return OFFSIM_OK;
    hipLaunchKernelGGL(k_encode_box, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4 *)obs, N, out_z);
    LAUNCH_CHECK();
    return OFFSIM_OK;
}

// HOMER obs_encoder forward + argmax, one row per lane, weights in LDS, k-ordered fmaf chains
// (bit-identical to an f32 MFMA accumulation chain; see cdna guide "FP32-input MFMA").
template <typename XT>
__device__ __forceinline__ float x_f32(const XT *x, int64_t i);
template <>
__device__ __forceinline__ float x_f32<float>(const float *x, int64_t i) { return x[i]; }
template <>
__device__ __forceinline__ float x_f32<__half>(const __half *x, int64_t i) { return __half2float(x[i]); }

#define MLP_MAX_H 128
template <typename XT>
__global__ void __launch_bounds__(256) k_encode_mlp(const XT *__restrict__ x, int64_t N, int dO, const float *__restrict__ W1,
                                                    const float *__restrict__ b1, int H, const float *__restrict__ W2,
                                                    const float *__restrict__ b2, int nZ, int32_t *__restrict__ out_z,
                                                    float *__restrict__ out_logits) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    float *w1 = (float *)lds_raw;  // [H][dO]
    float *bb1 = w1 + H * dO;      // [H]
    float *w2 = bb1 + H;           // [nZ][H]
    float *bb2 = w2 + nZ * H;      // [nZ]
    for (int i = threadIdx.x; i < H * dO; i += blockDim.x) w1[i] = W1[i];
    for (int i = threadIdx.x; i < H; i += blockDim.x) bb1[i] = b1[i];
    for (int i = threadIdx.x; i < nZ * H; i += blockDim.x) w2[i] = W2[i];
    for (int i = threadIdx.x; i < nZ; i += blockDim.x) bb2[i] = b2[i];
    __syncthreads();
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < N; row += stride) {
        float h[MLP_MAX_H];
#pragma unroll 4
        for (int j = 0; j < H; j++) {
            float acc = 0.f;
            for (int k = 0; k < dO; k++) acc = fmaf(x_f32<XT>(x, row * dO + k), w1[j * dO + k], acc);
            acc += bb1[j];
            h[j] = acc > 0.f ? acc : 0.01f * acc;  // LeakyReLU(0.01)
        }
        int best = 0;
        float bv = -__builtin_inff();
        for (int c = 0; c < nZ; c++) {
            float acc = 0.f;
            for (int k = 0; k < H; k++) acc = fmaf(h[k], w2[c * H + k], acc);
            acc += bb2[c];
            if (out_logits) out_logits[row * nZ + c] = acc;
            if (acc > bv) {  // first maximal index, as torch.max(dim=1)
                bv = acc;
                best = c;
            }
        }
        out_z[row] = best;
    }
}
#include "encode_mfma.hpp"

template <typename XT>
static int launch_mlp_mfma(const XT *x, int64_t N, int dO, const float *W1, const float *b1, int H, const float *W2, const float *b2,
                           int nZ, int32_t *out_z, float *out_logits, hipStream_t st) {
    const int HT = (H + 31) / 32, ZT = (nZ + 31) / 32;
    const int dOp = (dO + 1) & ~1;
    {   // weights in registers (encode_mfma.hpp, second kernel): the observation widths of the reference's configurations, hidden layer
        // <= 64, rows 16-byte aligned where they are read as 16-byte words
        const bool al = ((uintptr_t)x & 15) == 0;
        int64_t groups = (N + 31) / 32;
        unsigned nb = (unsigned)((groups + 3) / 4);
        if (nb > 256) nb = 256;  // one workgroup (a wavefront per SIMD) per CU
        if (nb < 1) nb = 1;
        // OFFSIM_ENCODER_F32=1: the exact-f32 products (round 3's kernel); default: the same kernel on bf16 x 3 (encode_mfma.hpp)
        static const bool f32_products = getenv("OFFSIM_ENCODER_F32") && atoi(getenv("OFFSIM_ENCODER_F32")) != 0;
#define LAUNCH_REG(DOc, HTc, ZTc, WPEc)                                                                                                \
    do {                                                                                                                               \
        if (f32_products)                                                                                                              \
            hipLaunchKernelGGL((k_encode_mlp_mfma_reg<XT, DOc, HTc, ZTc, WPEc>), dim3(nb * WPEc), dim3(256), 0, st, x, N, W1, b1, H, W2, b2, nZ, \
                               out_z, out_logits);                                                                                     \
        else                                                                                                                           \
            hipLaunchKernelGGL((k_encode_mlp_mfma_split<XT, DOc, HTc, ZTc, WPEc>), dim3(nb * WPEc), dim3(256), 0, st, x, N, W1, b1, H, W2, b2, nZ, \
                               out_z, out_logits);                                                                                     \
        LAUNCH_CHECK();                                                                                                                \
        return OFFSIM_OK;                                                                                                              \
    } while (0)
        if (HT == 2 && al) {  // (narrow observations: a tile is short, two wavefronts per SIMD fill each other's gaps)
            if (dO == 128 && ZT == 2) LAUNCH_REG(128, 2, 2, 1);
            if (dO == 128 && ZT == 1) LAUNCH_REG(128, 2, 1, 1);
            if (dO == 2 && ZT == 1) LAUNCH_REG(2, 2, 1, 2);
            if (dO == 2 && ZT == 2) LAUNCH_REG(2, 2, 2, 2);
            if (dO == 4 && ZT == 1) LAUNCH_REG(4, 2, 1, 2);
            if (dO == 4 && ZT == 2) LAUNCH_REG(4, 2, 2, 2);
        }
#undef LAUNCH_REG
    }
    size_t lds = sizeof(float) * ((size_t)HT * 32 * (dOp + 1) + (size_t)ZT * 32 * (HT * 32 + 1) + HT * 32 + ZT * 32);
    if (lds > 160 * 1024) return fail(OFFSIM_EUNSUPPORTED, "encode_mlp: weights exceed LDS%s");
    int64_t groups = (N + 31) / 32;
    unsigned nb = (unsigned)((groups + 3) / 4);
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
#define LAUNCH_MFMA(HTc, ZTc)                                                                                              \
    do {                                                                                                                   \
        if (lds > 64 * 1024)                                                                                               \
            HIP_TRY(allow_big_lds((k_encode_mlp_mfma<XT, HTc, ZTc>), (int)lds)); \
        hipLaunchKernelGGL((k_encode_mlp_mfma<XT, HTc, ZTc>), dim3(nb), dim3(256), lds, st, x, N, dO, W1, b1, H, W2, b2, nZ, out_z, out_logits); \
    } while (0)
    if (HT == 1 && ZT == 1) LAUNCH_MFMA(1, 1);
    else if (HT == 2 && ZT == 1) LAUNCH_MFMA(2, 1);
    else if (HT == 2 && ZT == 2) LAUNCH_MFMA(2, 2);
    else if (HT == 1 && ZT == 2) LAUNCH_MFMA(1, 2);
    else if (HT == 4 && ZT == 1) LAUNCH_MFMA(4, 1);
    else if (HT == 4 && ZT == 2) LAUNCH_MFMA(4, 2);
    else return 1;  // shape outside the MFMA instantiations: caller falls back to the VALU kernel
#undef LAUNCH_MFMA
    LAUNCH_CHECK();
    return OFFSIM_OK;
}

extern "C" int offsim_encode_mlp(const void *x, int32_t x_dtype, int64_t N, int32_t dO, const float *W1, const float *b1,
                                 int32_t H, const float *W2, const float *b2, int32_t nZ, int32_t *out_z, float *out_logits,
                                 void *stream) {
    if (N < 0 || dO <= 0 || H <= 0 || nZ <= 0 || !W1 || !b1 || !W2 || !b2) return fail(OFFSIM_EINVAL, "encode_mlp: bad argument%s");
    if (H > MLP_MAX_H) return fail(OFFSIM_EUNSUPPORTED, "encode_mlp: hidden size > 128%s");
    if (N == 0) return OFFSIM_OK;
    {   // matrix-core path for the shapes the reference uses (H <= 128, nZ <= 64); VALU kernel otherwise
        int rc = 1;
        if (x_dtype == OFFSIM_F32) rc = launch_mlp_mfma<float>((const float *)x, N, dO, W1, b1, H, W2, b2, nZ, out_z, out_logits, (hipStream_t)stream);
        else if (x_dtype == OFFSIM_F16) rc = launch_mlp_mfma<__half>((const __half *)x, N, dO, W1, b1, H, W2, b2, nZ, out_z, out_logits, (hipStream_t)stream);
        if (rc <= 0) return rc;
    }
    size_t lds = sizeof(float) * ((size_t)H * dO + H + (size_t)nZ * H + nZ);
    if (lds > 160 * 1024) return fail(OFFSIM_EUNSUPPORTED, "encode_mlp: weights exceed LDS%s");
    unsigned nb = (unsigned)((N + 255) / 256);
    if (nb > 2048) nb = 2048;
    hipStream_t st = (hipStream_t)stream;
    if (x_dtype == OFFSIM_F32) {
        if (lds > 64 * 1024) HIP_TRY(allow_big_lds((k_encode_mlp<float>), (int)lds));
        hipLaunchKernelGGL(k_encode_mlp<float>, dim3(nb), dim3(256), lds, st, (const float *)x, N, dO, W1, b1, H, W2, b2, nZ, out_z, out_logits);
    } else if (x_dtype == OFFSIM_F16) {
        if (lds > 64 * 1024) HIP_TRY(allow_big_lds((k_encode_mlp<__half>), (int)lds));
        hipLaunchKernelGGL(k_encode_mlp<__half>, dim3(nb), dim3(256), lds, st, (const __half *)x, N, dO, W1, b1, H, W2, b2, nZ, out_z, out_logits);
    } else return fail(OFFSIM_EINVAL, "encode_mlp: x_dtype must be OFFSIM_F32 or OFFSIM_F16%s");
    LAUNCH_CHECK();
    return OFFSIM_OK;
}
