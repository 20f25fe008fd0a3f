// Dependent-load latency against the span of memory the loads fall into (is the latency of the scan's stream reads a matter of TLB reach?).
//   hipcc --offload-arch=gfx950 -O3 memlat.hip -o memlat && ./memlat
// One lane per wavefront follows a chain of pseudo-random 64-byte-aligned addresses inside a span of S bytes (the address of hop k+1
// depends on the value hop k returns: no two loads of a wavefront overlap); `waves` wavefronts per CU run at once (1: unloaded).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k_chase(const uint32_t *buf, uint64_t span_words, int hops, uint64_t *out, uint32_t seed) {
    const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if ((threadIdx.x & 63) != 0) return;
    uint64_t x = (uint64_t)(w + 1) * 0x9e3779b97f4a7c15ull + seed;
    uint32_t acc = 0;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int h = 0; h < hops; h++) {
        x = x * 6364136223846793005ull + 1442695040888963407ull + acc;
        const uint64_t idx = ((x >> 20) % (span_words >> 4)) << 4;  // 64-byte aligned
        acc = __builtin_nontemporal_load(buf + idx) & 1u;  // (buffer is zero: acc stays 0, but the next address waits for it)
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    out[w] = (t1 - t0) + acc;
}
int main() {
    size_t total = 200ull << 30;
    uint32_t *buf = nullptr;
    while (hipMalloc(&buf, total) != hipSuccess) total -= 8ull << 30;
    hipMemset(buf, 0, total);
    uint64_t *out;
    hipMalloc(&out, 8 * 65536);
    printf("buffer %zu GiB; cycles of s_memtime (100 MHz ticks x clock ratio are not needed: s_memtime counts shader clocks) per dependent 64-byte load\n", total >> 30);
    printf("%10s %14s %14s %14s\n", "span", "1 wave/CU", "8 waves/CU", "32 waves/CU");
    const size_t spans[] = {64ull << 20, 1ull << 30, 8ull << 30, 32ull << 30, 128ull << 30, total};
    for (size_t s : spans) {
        if (s > total) continue;
        double r[3];
        const int wv[3] = {1, 8, 32};
        for (int m = 0; m < 3; m++) {
            const int blocks = 256 * (wv[m] > 4 ? wv[m] / 4 : 1), threads = wv[m] > 4 ? 256 : 64 * wv[m];
            const int hops = 2000;
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(k_chase, dim3(blocks), dim3(threads), 0, 0, buf, (uint64_t)(s / 4), hops, out, 12345u + rep);
                hipDeviceSynchronize();
            }
            const int nw = blocks * (threads / 64);
            uint64_t *h = (uint64_t *)malloc(8 * nw);
            hipMemcpy(h, out, 8 * nw, hipMemcpyDeviceToHost);
            double sum = 0;
            for (int i = 0; i < nw; i++) sum += (double)h[i];
            r[m] = sum / nw / hops;
            free(h);
        }
        printf("%7zu MiB %14.0f %14.0f %14.0f\n", s >> 20, r[0], r[1], r[2]);
    }
    return 0;
}
