// Round 6, time-boxed: would the chain of csrc/scan_rows.hpp step faster with its windows in VGPRs instead of LDS?
//
// The row-packed scan keeps, per rollout, one 8-entry head-aligned window per state in LDS; a step of the chain reads the entry of the
// state it arrives in (one ds_read_b32 for the wavefront's four rollouts: every 16-lane row has its own state, i.e. its own address) and
// stores the shifted row back (one ds_write_b32).  The review of round 5 asked whether 163 states x one VGPR each (a 16-lane row = 16
// entries of one rollout's window of that state) would take the dependent LDS round trip out of the step.  Register-relative addressing on
// gfx9 is wave-uniform (s_set_gpr_idx_on / M0): four rollouts in four different states need four indexed moves, each under its row's
// exec mask, each behind a v_readlane of that row's state -- and the accept shifts the row by a per-row amount, which DPP (immediate
// shift counts only) does as a binary ladder.  This program prices exactly those building blocks, on one wavefront per SIMD, alone and
// beside a busy partner wavefront (the scan's helper), against the LDS form of the same two operations:
//   lds      ds_read_b32 of the next state's entry (lane-varying address) -> next state; ds_write_b32 of the shifted row
//   vread    4 x { v_readlane, s_mov exec, s_set_gpr_idx_on, v_mov (indexed source), s_set_gpr_idx_off } -> next state; no write-back
//   vboth    vread + the write-back: a 3-stage DPP shift ladder (row_shr 1 / 2 / 4 under v_cndmask) and 4 indexed stores
// Every variant is a DEPENDENT chain (the next state comes out of the entry just read), like the scan's.  Printed: cycles per step.
//   hipcc --offload-arch=gfx950 -O3 vgpr_win.hip -o vgpr_win && ./vgpr_win [steps]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(3))) volatile uint32_t ldsv_u32;
#define LV32(a) (*(ldsv_u32 *)(a))
#define NS 160u  // states: v[64 .. 223] hold the windows of the VGPR variants

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// clobber list v64 .. v223
#define C8(a) "v" #a "0", "v" #a "1", "v" #a "2", "v" #a "3", "v" #a "4", "v" #a "5", "v" #a "6", "v" #a "7", "v" #a "8", "v" #a "9"
#define WIN_CLOBBERS "v64", "v65", "v66", "v67", "v68", "v69", C8(7), C8(8), C8(9), C8(10), C8(11), C8(12), C8(13), C8(14), C8(15), C8(16), C8(17), C8(18), C8(19), \
                     C8(20), C8(21), "v220", "v221", "v222", "v223"

// one indexed read of row K's state: z (row-uniform VGPR) -> s20 -> v[64 + s20] under the row's lanes
#define VREAD_ROW(K, LANE, MLO, MHI)                                                                                           \
    "v_readlane_b32 s20, %[z], " LANE "\n\t"                                                                                  \
    "s_mov_b32 exec_lo, " MLO "\n\t"                                                                                          \
    "s_mov_b32 exec_hi, " MHI "\n\t"                                                                                          \
    "s_set_gpr_idx_on s20, 0x1\n\t"                              /* src0 relative */                                         \
    "v_mov_b32 %[w], v64\n\t"                                                                                                 \
    "s_set_gpr_idx_off\n\t"
#define VWRITE_ROW(K, LANE, MLO, MHI)                                                                                          \
    "v_readlane_b32 s20, %[zo], " LANE "\n\t"                                                                                 \
    "s_mov_b32 exec_lo, " MLO "\n\t"                                                                                          \
    "s_mov_b32 exec_hi, " MHI "\n\t"                                                                                          \
    "s_set_gpr_idx_on s20, 0x8\n\t"                              /* dst relative */                                          \
    "v_mov_b32 v64, %[sh]\n\t"                                                                                                \
    "s_set_gpr_idx_off\n\t"

template <int VAR>
__global__ void __launch_bounds__(512) k_win(uint64_t *out, int steps, int partner, uint32_t seed) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t base = (uint32_t)(uintptr_t)(lds_byte *)lds_raw;
    const uint32_t li = lane & 15u, rw = lane >> 4;
    if (threadIdx.x == 0) LV32(base) = 0u;
    __syncthreads();
    if (wave >= 4) {  // the partner wavefront of each SIMD (the scan's helper): idle, or busy ~40 % of the time
        if (!partner) return;
        __builtin_amdgcn_s_setprio(2);
        const uint32_t mine = base + 1024u + 16u * 8192u + (wave - 4u) * 512u;
        uint32_t v = lane, acc = 0;
        for (uint32_t n = 0; n < (1u << 24); n++) {
            if (LV32(base) == 0xdeadu) break;
            if (partner == 2 && (n & 63u) < 26u) {
                for (int k = 0; k < 12; k++) {
                    v = v * 3u + acc;
                    acc += LV32(mine + ((v >> 7) & 60u));
                    v ^= v >> 5;
                    LV32(mine + 64u + li * 4u) = v;
                    for (int q = 0; q < 10; q++) v = v * 0x9e3779b9u + (uint32_t)q;
                }
            } else {
                acc += LV32(mine);
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (acc == 0x12345u) out[63] = v;
        return;
    }
    // windows: entry = [random high bits | next state]; LDS copy for the lds variant (per rollout region of 8 KiB: 160 states x 32 B)
    const uint32_t rid = wave * 4u + rw, win_a = base + 1024u + rid * 8192u;
    for (uint32_t s = 0; s < NS; s++)
        if (li < 8u) LV32(win_a + s * 32u + li * 4u) = (mix(seed + rid * 7919u + s * 131u + li) & 0xffffff00u) | (mix(seed * 5u + rid * 31u + s * 17u + li) % NS);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t z = mix(seed + rid) % NS, w = 0, zo = z, sh = 0, kamt = 1u + (li & 3u);
    uint32_t cnt = (uint32_t)steps;
    uint64_t t0 = 0, t1 = 0;
    if (VAR == 0) {
        const uint32_t rd = win_a + (li & 7u) * 4u;
        uint32_t a = rd + z * 32u, t;
        t0 = __builtin_amdgcn_s_memtime();
        asm volatile(
            "1:\n\t"
            "ds_read_b32 %[w], %[a]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_and_b32 %[t], 0xff, %[w]\n\t"                        /* next state */
            "ds_write_b32 %[a], %[w]\n\t"                           /* the row store (same instruction count as the shifted store) */
            "v_lshl_add_u32 %[a], %[t], 5, %[rd]\n\t"
            "s_sub_u32 %[cnt], %[cnt], 1\n\t"
            "s_cmp_lg_u32 %[cnt], 0\n\t"
            "s_cbranch_scc1 1b\n\t"
            "s_waitcnt lgkmcnt(0)"
            : [w] "+v"(w), [a] "+v"(a), [t] "=&v"(t), [cnt] "+s"(cnt)
            : [rd] "v"(rd)
            : "scc", "memory");
        t1 = __builtin_amdgcn_s_memtime();
        z = (a - rd) >> 5;
    } else {
        // fill v64 .. v223 from the LDS copy (lane = rollout row x entry: lanes 8..15 of a row repeat 0..7)
        const uint32_t rd = win_a + (li & 7u) * 4u;
        asm volatile(
            "s_mov_b32 s21, 0\n\t"
            "v_mov_b32 %[w], %[rd]\n\t"
            "2:\n\t"
            "ds_read_b32 %[sh], %[w]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_set_gpr_idx_on s21, 0x8\n\t"
            "v_mov_b32 v64, %[sh]\n\t"
            "s_set_gpr_idx_off\n\t"
            "v_add_u32 %[w], 32, %[w]\n\t"
            "s_add_u32 s21, s21, 1\n\t"
            "s_cmp_lt_u32 s21, 160\n\t"
            "s_cbranch_scc1 2b\n\t"
            : [w] "+v"(w), [sh] "+v"(sh)
            : [rd] "v"(rd)
            : "scc", "memory", "s21", WIN_CLOBBERS);
        t0 = __builtin_amdgcn_s_memtime();
        if (VAR == 1) {
            asm volatile(
                "s_mov_b64 s[22:23], exec\n\t"
                "1:\n\t"
                VREAD_ROW(0, "0", "0xffff", "0") VREAD_ROW(1, "16", "0xffff0000", "0") VREAD_ROW(2, "32", "0", "0xffff") VREAD_ROW(3, "48", "0", "0xffff0000")
                "s_mov_b64 exec, s[22:23]\n\t"
                "v_and_b32 %[z], 0xff, %[w]\n\t"                     /* next state (row-uniform after the scan's DPP minimum; here lane-wise: the readlane takes the row's first lane) */
                "s_sub_u32 %[cnt], %[cnt], 1\n\t"
                "s_cmp_lg_u32 %[cnt], 0\n\t"
                "s_cbranch_scc1 1b\n\t"
                : [w] "+v"(w), [z] "+v"(z), [cnt] "+s"(cnt)
                :
                : "scc", "memory", "s20", "s22", "s23", WIN_CLOBBERS);
        } else {
            uint32_t b1 = kamt & 1u, b2 = kamt & 2u, b4 = kamt & 4u, t;
            asm volatile(
                "s_mov_b64 s[22:23], exec\n\t"
                "v_cmp_ne_u32_e64 s[24:25], 0, %[b1]\n\t"
                "v_cmp_ne_u32_e64 s[26:27], 0, %[b2]\n\t"
                "v_cmp_ne_u32_e64 s[28:29], 0, %[b4]\n\t"
                "1:\n\t"
                VREAD_ROW(0, "0", "0xffff", "0") VREAD_ROW(1, "16", "0xffff0000", "0") VREAD_ROW(2, "32", "0", "0xffff") VREAD_ROW(3, "48", "0", "0xffff0000")
                "s_mov_b64 exec, s[22:23]\n\t"
                "v_mov_b32 %[zo], %[z]\n\t"
                "v_and_b32 %[z], 0xff, %[w]\n\t"
                /* the accepted entry leaves: shift the row towards lane 0 by the row's own amount (binary ladder; vacated lanes take 0) */
                "v_mov_b32 %[sh], %[w]\n\t"
                "s_nop 1\n\t"
                "v_mov_b32_dpp %[t], %[sh] row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                "v_cndmask_b32_e64 %[sh], %[sh], %[t], s[24:25]\n\t"
                "s_nop 1\n\t"
                "v_mov_b32_dpp %[t], %[sh] row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                "v_cndmask_b32_e64 %[sh], %[sh], %[t], s[26:27]\n\t"
                "s_nop 1\n\t"
                "v_mov_b32_dpp %[t], %[sh] row_shl:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                "v_cndmask_b32_e64 %[sh], %[sh], %[t], s[28:29]\n\t"
                VWRITE_ROW(0, "0", "0xffff", "0") VWRITE_ROW(1, "16", "0xffff0000", "0") VWRITE_ROW(2, "32", "0", "0xffff") VWRITE_ROW(3, "48", "0", "0xffff0000")
                "s_mov_b64 exec, s[22:23]\n\t"
                "s_sub_u32 %[cnt], %[cnt], 1\n\t"
                "s_cmp_lg_u32 %[cnt], 0\n\t"
                "s_cbranch_scc1 1b\n\t"
                : [w] "+v"(w), [z] "+v"(z), [zo] "+v"(zo), [sh] "+v"(sh), [t] "=&v"(t), [cnt] "+s"(cnt)
                : [b1] "v"(b1), [b2] "v"(b2), [b4] "v"(b4)
                : "scc", "vcc", "memory", "s20", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", WIN_CLOBBERS);
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    if (lane == 0) {
        out[wave] = t1 - t0;
        out[8 + wave] = z + w;
        if (wave == 3) LV32(base) = 0xdeadu;  // (the four chain wavefronts end together: the partners may stop)
    }
}

int main(int argc, char **argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 200000;
    uint64_t *d, h[64];
    hipMalloc(&d, sizeof(h));
    typedef void (*fn_t)(uint64_t *, int, int, uint32_t);
    fn_t fns[3] = {k_win<0>, k_win<1>, k_win<2>};
    const char *names[3] = {"lds (read + row store)", "vread (4 indexed reads)", "vboth (+ shift ladder, 4 indexed stores)"};
    const size_t lds = 1024 + 16 * 8192 + 4 * 512 + 1024;
    for (auto f : fns) hipFuncSetAttribute((const void *)f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    printf("cycles per step of a dependent chain, one wavefront per SIMD, four rollouts per wavefront, %d steps, %u states\n", steps, NS);
    printf("%-44s %10s %12s %12s\n", "variant", "alone", "idle partner", "busy partner");
    for (int v = 0; v < 3; v++) {
        printf("%-44s", names[v]);
        for (int p = 0; p < 3; p++) {
            hipMemset(d, 0, sizeof(h));
            hipLaunchKernelGGL(fns[v], dim3(1), dim3(512), lds, 0, d, steps, p, 12345u);
            if (hipDeviceSynchronize() != hipSuccess) {
                printf(" launch failed\n");
                return 1;
            }
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            double c = 0;
            for (int w = 0; w < 4; w++) c += (double)h[w] / steps / 4.0;
            printf(" %12.1f", c);
        }
        printf("\n");
    }
    return 0;
}
