// Calibration of rocprofv3 FETCH_SIZE for the access shape of the scan's window top-ups (csrc/scan_rows.hpp, request()): LDS-DMA loads of
// 16 bytes per lane (global_load_lds_dwordx4) from SCATTERED, 4-byte-aligned addresses of a buffer far larger than the Infinity Cache -- a
// shape tools/calib_fetch.py (4-byte gathers, wide copies) does not cover.  Every lane of every wavefront issues LOADS loads from
// pseudo-random positions; the program prints the byte counts to compare the counter with:
//   useful bytes (16 per load), 64-byte sectors touched (1 or 2 per load), 128-byte lines touched (1 or 2 per load).
//   hipcc --offload-arch=gfx950 -O3 calib_dma.hip -o calib_dma
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- ./calib_dma [aligned]
// "aligned": the same loads from 16-byte-aligned positions (never more than one sector).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define LOADS 256

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ void __launch_bounds__(256) k_calib_dma(const uint32_t *buf, uint64_t n_words, int aligned, unsigned long long *counts) {
    __shared__ __align__(16) uint32_t land[4 * 256 * 4];  // 1 KiB per wavefront
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)&land[wave * 256];
    const uint32_t lds_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds);
    unsigned long long sectors = 0, lines = 0;
    uint32_t h = mix(blockIdx.x * 256u + threadIdx.x + 1u);
    for (int i = 0; i < LOADS; i++) {
        h = mix(h + 0x9e3779b9u * (uint32_t)(i + 1));
        uint64_t w = ((uint64_t)h * (uint64_t)mix(h ^ 0x5bd1e995u)) % (n_words - 8u);
        if (aligned) w &= ~3ull;
        const uint32_t *src = buf + w;
        const uint64_t b0 = (uint64_t)(uintptr_t)src, b1 = b0 + 15u;
        sectors += 1u + ((b0 >> 6) != (b1 >> 6));
        lines += 1u + ((b0 >> 7) != (b1 >> 7));
        asm volatile(
            "s_mov_b32 m0, %1\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %0, off\n\t"
            :
            : "v"(src), "s"(lds_u)
            : "memory");
        if ((i & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    atomicAdd(&counts[0], sectors);
    atomicAdd(&counts[1], lines);
    if (land[threadIdx.x] == 0x12345u) counts[2] = 1;
}

int main(int argc, char **argv) {
    const int aligned = argc > 1 && !strcmp(argv[1], "aligned");
    const uint64_t n_words = 1ull << 30;  // 4 GiB
    uint32_t *buf;
    unsigned long long *counts, h[3];
    if (hipMalloc(&buf, n_words * 4) != hipSuccess || hipMalloc(&counts, sizeof(h)) != hipSuccess) return 1;
    (void)hipMemset(buf, 1, n_words * 4);
    (void)hipMemset(counts, 0, sizeof(h));
    const int blocks = 4096;
    hipLaunchKernelGGL(k_calib_dma, dim3(blocks), dim3(256), 0, 0, buf, n_words, aligned, counts);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    (void)hipMemcpy(h, counts, sizeof(h), hipMemcpyDeviceToHost);
    const double loads = (double)blocks * 256 * LOADS;
    printf("k_calib_dma (%s): %.0f loads of 16 B: useful %.4e B; 64-B sectors touched %.4e B; 128-B lines touched %.4e B\n",
           aligned ? "16-byte aligned" : "4-byte aligned", loads, loads * 16, (double)h[0] * 64, (double)h[1] * 128);
    return 0;
}
