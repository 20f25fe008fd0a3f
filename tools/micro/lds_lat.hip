// Micro-benchmarks for the shuffle pipeline design: LDS dependent-read latency, dependent VALU/SALU chains, and
// wave-to-wave ping-pong through LDS flags.   hipcc --offload-arch=gfx950 -O3 lds_lat.hip -o lds_lat
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) volatile uint32_t lds_vu32;

__global__ void k_chase(uint64_t *out, int iters) {
    __shared__ uint32_t buf[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) buf[i] = (i * 17 + 5) & 1023;
    __syncthreads();
    lds_vu32 *b = (lds_vu32 *)buf;
    uint32_t p = threadIdx.x & 63;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) p = b[p];
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = p; }
}
__global__ void k_valu(uint64_t *out, int iters) {
    uint32_t x = threadIdx.x, y = 3;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) x = x * 3u + y;  // v_mad dependent chain (mul+add -> 1-2 instr)
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = x; }
}
__global__ void k_ballot(uint64_t *out, int iters) {  // VALU -> SGPR -> SALU -> VALU round trip
    uint32_t x = threadIdx.x;
    uint32_t s = 77;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            uint64_t m = __ballot(x < s);
            s = s + (uint32_t)__popcll(m) + 1u;
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = s; }
}
__global__ void k_pingpong(uint64_t *out, int iters) {  // two waves alternate incrementing a flag in LDS
    __shared__ uint32_t flag[4];
    if (threadIdx.x == 0) flag[0] = 0;
    __syncthreads();
    lds_vu32 *f = (lds_vu32 *)flag;
    const int wave = threadIdx.x >> 6;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
        const uint32_t want = 2u * i + (uint32_t)wave;
        while ((uint32_t)__builtin_amdgcn_readfirstlane(f[0]) != want) {}
        f[0] = want + 1u;
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = f[0]; }
}
int main() {
    uint64_t *d, h[2];
    hipMalloc(&d, 16);
    const int it = 20000;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, 0, d, it); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        if (rep) printf("LDS dependent read: %.1f cycles/iter\n", (double)h[0] / it);
        hipLaunchKernelGGL(k_valu, dim3(1), dim3(64), 0, 0, d, it); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        if (rep) printf("dependent VALU mad: %.1f cycles/op\n", (double)h[0] / it / 16);
        hipLaunchKernelGGL(k_ballot, dim3(1), dim3(64), 0, 0, d, it); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        if (rep) printf("v_cmp->s_bcnt->s_add->v_cmp loop: %.1f cycles/iter\n", (double)h[0] / it / 8);
        hipLaunchKernelGGL(k_pingpong, dim3(1), dim3(128), 0, 0, d, it); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        if (rep) printf("LDS ping-pong: %.1f cycles per handoff (one way)\n", (double)h[0] / it / 2);
        hipLaunchKernelGGL(k_pingpong, dim3(1), dim3(256), 0, 0, d, it); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    }
    return 0;
}
