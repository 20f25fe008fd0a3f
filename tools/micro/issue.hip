// Issue cost of a single wavefront's instruction stream on gfx950 (one wave per SIMD, or with a polling partner on the same SIMD):
// what the hand-scheduled chain loop of csrc/scan_rows.hpp can be made of.   hipcc --offload-arch=gfx950 -O3 issue.hip -o issue
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;

// every case: 64 x BODY per loop iteration, `iters` iterations, cycles from s_memtime (wave 0 of the block reports)
#define CASE(NAME, N_PER_BODY, BODY, ...)                                                                              \
    __global__ void __launch_bounds__(512) NAME(uint64_t *out, int iters, int partner) {                              \
        __shared__ uint32_t buf[4096];                                                                                 \
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = (i * 4 + 64) & 4095;                            \
        __syncthreads();                                                                                               \
        const int wave = threadIdx.x >> 6;                                                                             \
        uint32_t v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, v4 = 4, v5 = 5, v6 = ((threadIdx.x & 63) >> 4) * 512 + 1024, v7 = 7, a0 = (threadIdx.x & 63) * (partner >= 16 ? 8 : 4); \
        uint32_t s0 = 1, s1 = 2; uint64_t p64 = threadIdx.x;                                                                                       \
        (void)s0; (void)s1;                                                                                            \
        if (wave >= 4) { /* partner waves (same SIMDs as waves 0..3): poll LDS with s_sleep like the scan's helper */  \
            if (!partner) return;                                                                                      \
            lds_vu32 *b = (lds_vu32 *)buf;                                                                             \
            while (b[4095] != 0xdeadu) {                                                                               \
                if (partner == 2) { for (int k = 0; k < 32; k++) v1 = v1 * 3u + v0; }                                 \
                __builtin_amdgcn_s_sleep(4);                                                                           \
            }                                                                                                          \
            if (v1 == 0x12345u) out[63] = v1;                                                                          \
            return;                                                                                                    \
        }                                                                                                              \
        const int em = partner >> 8; partner &= 255; \
        if (em == 1 && (threadIdx.x & 63) >= 32) return; \
        if (em == 2 && (threadIdx.x & 15) >= 8) return; \
        if (em == 3 && (threadIdx.x & 15) >= 1) return; \
        uint64_t t0 = __builtin_amdgcn_s_memtime();                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
        for (int i = 0; i < iters; i++) {                                                                              \
            asm volatile(REP64(BODY)                                                                                   \
                         : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(a0), "+s"(s0), "+s"(s1), [p] "+v"(p64) \
                         : : "vcc", "scc", "memory", "s20", "s21", "s22", "s23");                                      \
        }                                                                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
        uint64_t t1 = __builtin_amdgcn_s_memtime();                                                                    \
        if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + a0 + s0 + s1 + (uint32_t)p64; }     \
        if (threadIdx.x == 0) ((lds_vu32 *)buf)[4095] = 0xdeadu;                                                              \
    }

CASE(k_valu_dep, 1, "v_add_u32 %0, %0, %1\n\t")
CASE(k_valu_ind2, 2, "v_add_u32 %0, %0, %1\n\tv_add_u32 %2, %2, %1\n\t")
CASE(k_valu_ind4, 4, "v_add_u32 %0, %0, %1\n\tv_add_u32 %2, %2, %1\n\tv_add_u32 %3, %3, %1\n\tv_add_u32 %4, %4, %1\n\t")
CASE(k_vop3_dep, 1, "v_lshl_add_u32 %0, %0, 1, %1\n\t")
CASE(k_vop3_ind4, 4, "v_lshl_add_u32 %0, %0, 1, %1\n\tv_and_or_b32 %2, %2, %1, %3\n\tv_lshl_add_u32 %4, %4, 1, %1\n\tv_and_or_b32 %5, %5, %1, %3\n\t")
CASE(k_salu_dep, 1, "s_add_u32 %9, %9, 1\n\t")
CASE(k_valu_salu, 2, "v_add_u32 %0, %0, %1\n\ts_add_u32 %9, %9, 1\n\t")
CASE(k_valu2_salu, 3, "v_add_u32 %0, %0, %1\n\tv_add_u32 %2, %2, %1\n\ts_add_u32 %9, %9, 1\n\t")
CASE(k_nop0, 1, "s_nop 0\n\t")
CASE(k_valu_nop, 2, "v_add_u32 %0, %0, %1\n\ts_nop 0\n\t")
CASE(k_dpp_dep_nop1, 2, "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t")
CASE(k_dpp_dep_fill2, 3, "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_u32 %2, %2, %1\n\tv_add_u32 %3, %3, %1\n\t")
CASE(k_dpp_dep_fill1s, 3, "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_u32 %2, %2, %1\n\ts_add_u32 %9, %9, 1\n\t")
CASE(k_dpp_ind, 2, "v_min_u32_dpp %0, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_min_u32_dpp %3, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t")
CASE(k_cmp_vcc_cnd, 2, "v_sub_co_u32 %2, vcc, %0, %1\n\tv_cndmask_b32_e64 %0, %0, -1, vcc\n\t")
CASE(k_cmp_sgpr_sor, 3, "v_cmp_le_u32_e64 s[20:21], %1, %0\n\tv_add_u32 %0, %0, %1\n\ts_or_b64 s[22:23], s[20:21], s[20:21]\n\t")
CASE(k_lds_chase, 1, "ds_read_b32 %8, %8\n\ts_waitcnt lgkmcnt(0)\n\t")
CASE(k_lds_read_ind, 1, "ds_read_b32 %2, %8\n\t")
CASE(k_lds_write_ind, 1, "ds_write_b32 %8, %1\n\t")
CASE(k_lds_rw_valu, 4, "ds_read_b32 %2, %8\n\tv_add_u32 %0, %0, %1\n\tds_write_b32 %8, %1 offset:256\n\tv_add_u32 %3, %3, %1\n\t")
CASE(k_waitcnt_sat, 2, "v_add_u32 %0, %0, %1\n\ts_waitcnt lgkmcnt(4)\n\t")
CASE(k_branch_nt, 3, "v_add_u32 %0, %0, %1\n\ts_cmp_eq_u32 %9, 0\n\ts_cbranch_scc1 0f\n\t0:\n\t")
CASE(k_branch_t, 2, "v_add_u32 %0, %0, %1\n\ts_branch 0f\n\t0:\n\t")
CASE(k_readlane, 2, "v_readfirstlane_b32 s20, %0\n\tv_add_u32 %0, s20, %1\n\t")
CASE(k_valu_dep_fill_salu, 2, "v_add_u32 %0, %0, %1\n\ts_add_u32 %9, %9, 1\n\t")
CASE(k_chain3_fill, 4, "v_add_u32 %0, %0, %1\n\tv_add_u32 %2, %2, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %3, %3, %1\n\t")
CASE(k_ds_add_rtn, 1, "ds_add_rtn_u32 %2, %8, %1\n\t")
CASE(k_bperm, 2, "ds_bpermute_b32 %0, %8, %0\n\ts_waitcnt lgkmcnt(0)\n\t")


#define EXEC_SET(M) "s_mov_b64 exec, " M "\n\t"
CASE(k_w32_fill3, 4, "ds_write_b32 %8, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %2, %2, %1\n\tv_add_u32 %3, %3, %1\n\t")
CASE(k_r32_fill3, 4, "ds_read_b32 %4, %8\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %2, %2, %1\n\tv_add_u32 %3, %3, %1\n\t")
CASE(k_w32_fill7, 8, "ds_write_b32 %8, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %2, %2, %1\n\tv_add_u32 %3, %3, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %2, %2, %1\n\tv_add_u32 %3, %3, %1\n\tv_add_u32 %5, %5, %1\n\t")
CASE(k_w32_half, 1, "ds_write_b32 %8, %1\n\t")   /* launched with exec = low 32 lanes (see EXECMODE) */
CASE(k_w2x32, 1, "ds_write2_b32 %8, %1, %2 offset0:0 offset1:64\n\t")
CASE(k_w64, 1, "ds_write_b64 %8, %[p]\n\t")
CASE(k_r64, 1, "ds_read_b64 %[p], %8\n\t")
CASE(k_r2x32, 1, "ds_read2_b32 %[p], %8 offset0:0 offset1:64\n\t")
CASE(k_w8, 1, "ds_write_b8 %8, %1\n\t")
CASE(k_w16, 1, "ds_write_b16 %8, %1\n\t")
CASE(k_vcmp_vccnz, 2, "v_cmp_eq_u32_e32 vcc, 0, %0\n\ts_cbranch_vccnz 0f\n\t0:\n\t")
CASE(k_vcmp_f4_vccnz, 6, "v_cmp_eq_u32_e32 vcc, 0, %0\n\tv_add_u32 %2, %2, %1\n\tv_add_u32 %3, %3, %1\n\tv_add_u32 %4, %4, %1\n\tv_add_u32 %5, %5, %1\n\ts_cbranch_vccnz 0f\n\t0:\n\t")
CASE(k_scmp_f2_scc, 4, "s_cmp_eq_u32 %9, 0\n\tv_add_u32 %2, %2, %1\n\tv_add_u32 %3, %3, %1\n\ts_cbranch_scc1 0f\n\t0:\n\t")
CASE(k_sor_scc, 2, "s_or_b64 s[22:23], s[20:21], s[20:21]\n\ts_cbranch_scc0 0f\n\t0:\n\t")
CASE(k_vcmp_f4_sor, 6, "v_cmp_le_u32_e64 s[20:21], %1, %0\n\tv_add_u32 %2, %2, %1\n\tv_add_u32 %3, %3, %1\n\tv_add_u32 %4, %4, %1\n\tv_add_u32 %5, %5, %1\n\ts_or_b64 s[22:23], s[20:21], s[20:21]\n\t")
CASE(k_vcmp_f2_sor, 4, "v_cmp_le_u32_e64 s[20:21], %1, %0\n\tv_add_u32 %2, %2, %1\n\tv_add_u32 %3, %3, %1\n\ts_or_b64 s[22:23], s[20:21], s[20:21]\n\t")
CASE(k_execz, 2, "v_add_u32 %0, %0, %1\n\ts_cbranch_execz 0f\n\t0:\n\t")
CASE(k_vcmpx, 2, "v_add_u32 %0, %0, %1\n\tv_cmp_le_u32_e64 s[20:21], %1, %0\n\t")
CASE(k_w32_same16, 1, "ds_write_b32 %6, %1\n\t")  /* v6 = row-uniform address */
CASE(k_r32_same16, 1, "ds_read_b32 %2, %6\n\t")
CASE(k_add_nortn, 1, "ds_add_u32 %8, %1\n\t")

struct C { const char *name; void (*fn)(uint64_t *, int, int); int n; };
#define E(NAME, N) {#NAME, NAME, N}
int main() {
    uint64_t *d;
    hipMalloc(&d, 64 * 8);
    C cases[] = {E(k_valu_dep, 1), E(k_valu_ind2, 2), E(k_valu_ind4, 4), E(k_vop3_dep, 1), E(k_vop3_ind4, 4), E(k_salu_dep, 1), E(k_valu_salu, 2),
                 E(k_valu2_salu, 3), E(k_nop0, 1), E(k_valu_nop, 2), E(k_dpp_dep_nop1, 2), E(k_dpp_dep_fill2, 3), E(k_dpp_dep_fill1s, 3), E(k_dpp_ind, 2),
                 E(k_cmp_vcc_cnd, 2), E(k_cmp_sgpr_sor, 3), E(k_lds_chase, 1), E(k_lds_read_ind, 1), E(k_lds_write_ind, 1), E(k_lds_rw_valu, 4),
                 E(k_waitcnt_sat, 2), E(k_branch_nt, 3), E(k_branch_t, 2), E(k_readlane, 2), E(k_valu_dep_fill_salu, 2), E(k_chain3_fill, 4),
                 E(k_ds_add_rtn, 1), E(k_bperm, 2),
                 E(k_w32_fill3, 4), E(k_r32_fill3, 4), E(k_w32_fill7, 8), E(k_w2x32, 1), E(k_w64, 1), E(k_r64, 1), E(k_r2x32, 1), E(k_w8, 1), E(k_w16, 1),
                 E(k_vcmp_vccnz, 2), E(k_vcmp_f4_vccnz, 6), E(k_scmp_f2_scc, 4), E(k_sor_scc, 2), E(k_vcmp_f4_sor, 6), E(k_vcmp_f2_sor, 4), E(k_execz, 2),
                 E(k_vcmpx, 2), E(k_w32_same16, 1), E(k_r32_same16, 1), E(k_add_nortn, 1)};
    const int iters = 2000;
    printf("%-24s %10s %10s %10s %10s   (cycles per BODY / per instruction; active lanes: all | 0..31 | 8 of every 16 | 1 of every 16)\n", "case", "all 64", "low 32", "8 per row", "1 per row");
    for (auto &c : cases) {
        double r[4];
        const int modes[4] = {0, 1 << 8, 2 << 8, 3 << 8};
        for (int p = 0; p < 4; p++) {
            uint64_t h[2] = {0, 0};
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(c.fn, dim3(1), dim3(512), 0, 0, d, iters, modes[p]);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            r[p] = (double)h[0] / (iters * 64.0);
        }
        printf("%-24s %5.1f/%4.1f %5.1f/%4.1f %5.1f/%4.1f %5.1f/%4.1f\n", c.name, r[0], r[0] / c.n, r[1], r[1] / c.n, r[2], r[2] / c.n, r[3], r[3] / c.n);
    }
    return 0;
}
