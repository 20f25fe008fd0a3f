// In what order does the LDS apply the lanes of ONE ds_add_rtn_u32 that hit the same address?  csrc/scan_rows.hpp (positions of a
// tick's accepted candidates) relies on ascending lane order: lane i gets back the sum of the addends of the lower lanes with its
// address.  This checks it exhaustively over random address patterns.   hipcc --offload-arch=gfx950 -O3 lds_atomic_order.hip -o lds_atomic_order
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void k(uint32_t seed0, int trials, uint64_t *bad, uint32_t n_addr) {
    __shared__ uint32_t cell[256];
    const uint32_t lane = threadIdx.x & 63u;
    uint64_t nbad = 0;
    uint32_t h = seed0 * 2654435761u + blockIdx.x * 97u + 1u;
    for (int t = 0; t < trials; t++) {
        for (uint32_t i = lane; i < 256; i += 64) cell[i] = 1000u * i;
        __syncthreads();
        h = h * 1664525u + 1013904223u;
        const uint32_t hl = (h ^ (lane * 0x9e3779b9u)) * 2246822519u;
        const uint32_t a = (hl >> 8) % n_addr, kk = 1u + ((hl >> 20) & 7u);
        const bool active = ((hl >> 28) & 7u) != 0u;  // some lanes sit out
        uint32_t got = 0;
        if (active) {
            typedef __attribute__((address_space(3))) uint32_t lds_u32;
            const uint32_t addr = (uint32_t)(uintptr_t)(lds_u32 *)&cell[a];
            asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(got) : "v"(addr), "v"(kk) : "memory");
        }
        // expected: base + sum of kk over lower ACTIVE lanes with the same address
        uint32_t want = 1000u * a;
        for (uint32_t j = 0; j < 64; j++) {
            const uint32_t aj = __shfl(a, j), kj = __shfl(kk, j);
            const bool actj = __shfl((int)active, j);
            if (j < lane && actj && aj == a) want += kj;
        }
        if (active && got != want) nbad++;
        __syncthreads();
    }
    if (nbad) atomicAdd((unsigned long long *)bad, (unsigned long long)nbad);
}
int main() {
    uint64_t *d, h = 0;
    hipMalloc(&d, 8);
    hipMemset(d, 0, 8);
    for (uint32_t n_addr : {1u, 2u, 3u, 5u, 16u, 40u, 162u}) {
        hipLaunchKernelGGL(k, dim3(512), dim3(64), 0, 0, 12345u + n_addr, 4000, d, n_addr);
        hipDeviceSynchronize();
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("addresses %3u: mismatches so far %llu (of %llu lane-operations)\n", n_addr, (unsigned long long)h, 512ull * 4000 * 64);
    }
    return h != 0;
}
