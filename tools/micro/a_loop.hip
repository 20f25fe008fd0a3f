// What does the apply loop of shuffle_wave.hpp cost per record, alone and next to three busy wavefronts?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;
typedef __attribute__((address_space(3))) volatile uint16_t lds_vu16;
typedef __attribute__((address_space(3))) volatile unsigned char lds_vu8;

__global__ void __launch_bounds__(256) k_apply(uint64_t *out, int iters, int busy_waves, int variant) {
    extern __shared__ __align__(16) unsigned char raw[];
    lds_vu32 *rec = (lds_vu32 *)raw;            // 16 slots x 64
    lds_vu8 *xb = (lds_vu8 *)(raw + 16 * 64 * 4);  // 61774 u16
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t n = 61774;
    for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) *(lds_vu16 *)(xb + 2 * k) = (uint16_t)k;
    uint32_t h = threadIdx.x * 2654435761u + 12345u;
    for (int sl = wave; sl < 16; sl += 4) {
        h = h * 1664525u + 1013904223u;
        uint32_t il = 60000u - sl * 64 - lane, v = (h >> 8) % 50000u;
        rec[sl * 64 + lane] = (il << 16) | v;
    }
    __syncthreads();
    if (wave == 0) {
        uint64_t t0 = __builtin_readcyclecounter();
        uint32_t qt = 0;
        uint64_t nflag = 0;
        for (int it = 0; it < iters; it++) {
            const uint32_t rv = rec[(qt & 15u) * 64 + lane];
            qt++;
            rec[1024 + 0] ;  // nothing
            const uint32_t il = (rv >> 15) & 0x1fffeu, v = (rv & 0xffffu) << 1;
            const uint32_t a = *(lds_vu16 *)(xb + il);
            const uint32_t b = *(lds_vu16 *)(xb + v);
            if (variant == 0) {
                *(lds_vu16 *)(xb + v) = (uint16_t)lane;
                const uint32_t tg = *(lds_vu16 *)(xb + v);
                if (__ballot(tg != (uint32_t)lane) == 0ull) {
                    *(lds_vu16 *)(xb + il) = (uint16_t)b;
                    *(lds_vu16 *)(xb + v) = (uint16_t)a;
                } else {
                    *(lds_vu16 *)(xb + v) = (uint16_t)b;
                    nflag++;
                }
            } else {
                *(lds_vu16 *)(xb + il) = (uint16_t)b;
                *(lds_vu16 *)(xb + v) = (uint16_t)a;
            }
        }
        uint64_t t1 = __builtin_readcyclecounter();
        if (lane == 0) { out[0] = t1 - t0; out[1] = nflag; }
    } else if (wave <= busy_waves) {  // busy neighbours: VALU + LDS traffic
        uint32_t x = lane;
        for (int it = 0; it < iters * 4; it++) {
            x = x * 1664525u + 1013904223u;
            rec[1024 + wave * 64 + lane] = x;
            x ^= rec[1024 + wave * 64 + ((lane + 1) & 63)];
        }
        if (x == 0x12345) out[2] = x;
    }
}
int main() {
    uint64_t *d, h[3];
    hipMalloc(&d, 24);
    hipFuncSetAttribute((const void *)k_apply, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int it = 20000;
    for (int variant = 0; variant < 2; variant++)
        for (int busy = 0; busy <= 3; busy += 3) {
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(k_apply, dim3(1), dim3(256), 140000, 0, d, it, busy, variant);
                hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
            }
            printf("variant %d busy %d: %.1f cycles/record (dup flags %llu)\n", variant, busy, (double)h[0] / it, (unsigned long long)h[1]);
        }
    return 0;
}
