// encode_mfma.hpp -- HOMER obs_encoder forward + argmax on the matrix cores (gfx950).
//
// z = argmax(W2 * leaky_relu(W1 * x + b1, 0.01) + b2)   (offsim4rl/encoders/homer.py:159-168, models.py:15-19)
//
// Both layers are computed TRANSPOSED with v_mfma_f32_32x32x2_f32 (exact f32 products, f32 accumulate):
//   Hid^T [H x 32 rows]  = W1 [H x dO] * X^T [dO x 32 rows]      A = W1 tile, B = X^T
//   Log^T [nZ x 32 rows] = W2 [nZ x H] * Hid^T [H x 32 rows]     A = W2 tile, B = Hid^T
// In the C/D layout of the 32x32 MFMA the lane index is the COLUMN (= data row here) and the registers
// are matrix rows (hidden units), which is exactly the B-operand layout of the next product up to a
// permutation of k -- so the hidden layer never leaves the registers (no LDS transpose).  The price is a
// fixed, non-natural order of the k summation; the encoder's parity criterion is tolerance based
// (logits within 1e-5, argmax equal where the top-2 gap > 1e-4), which this meets.
// One wave = 32 observations per iteration; weights staged once per block in LDS (row stride padded by 1).
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace offsim {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// MFMA 32x32 C/D: register g of lane l is matrix row (g&3) + 8*(g>>2) + 4*(l>>5), column l&31
__device__ __forceinline__ int cd_row(int g, int hi) { return (g & 3) + 8 * (g >> 2) + 4 * hi; }

template <typename XT, int HT /* hidden tiles of 32 */, int ZT /* latent tiles of 32 */>
__global__ void __launch_bounds__(256) k_encode_mlp_mfma(const XT *__restrict__ x, int64_t N, int dO, const float *__restrict__ W1,
                                                         const float *__restrict__ b1, int H, const float *__restrict__ W2,
                                                         const float *__restrict__ b2, int nZ, int32_t *__restrict__ out_z,
                                                         float *__restrict__ out_logits) {
    extern __shared__ __align__(16) float lds_w[];
    const int dOp = (dO + 1) & ~1;           // k extent of layer 1, padded to the MFMA's K = 2
    const int s1 = dOp + 1, s2 = HT * 32 + 1;  // padded LDS row strides (bank-conflict-free column reads)
    float *w1 = lds_w;                        // [HT*32][s1], rows >= H and cols >= dO are zero
    float *w2 = w1 + HT * 32 * s1;            // [ZT*32][s2], rows >= nZ and cols >= H are zero
    float *bb1 = w2 + ZT * 32 * s2;           // [HT*32]
    float *bb2 = bb1 + HT * 32;               // [ZT*32]
    for (int i = threadIdx.x; i < HT * 32 * s1; i += blockDim.x) {
        int u = i / s1, k = i - u * s1;
        w1[i] = (u < H && k < dO) ? W1[u * dO + k] : 0.f;
    }
    for (int i = threadIdx.x; i < ZT * 32 * s2; i += blockDim.x) {
        int zc = i / s2, k = i - zc * s2;
        w2[i] = (zc < nZ && k < H) ? W2[zc * H + k] : 0.f;
    }
    for (int i = threadIdx.x; i < HT * 32; i += blockDim.x) bb1[i] = i < H ? b1[i] : 0.f;
    for (int i = threadIdx.x; i < ZT * 32; i += blockDim.x) bb2[i] = i < nZ ? b2[i] : 0.f;
    __syncthreads();

    const int lane = threadIdx.x & 63, col = lane & 31, hi = lane >> 5;
    const int wave_global = (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * (blockDim.x >> 6);
    const int half = dOp >> 1;  // lanes with hi = 0 sum k in [0, half), hi = 1 sum k in [half, dOp): contiguous reads per lane
    for (int64_t row0 = (int64_t)wave_global * 32; row0 < N; row0 += (int64_t)n_waves * 32) {
        const int64_t row = row0 + col;
        const bool live = row < N;
        const XT *xr = x + (live ? row : 0) * dO + hi * half;
        // ---- layer 1 ----
        f32x16 acc1[HT];
#pragma unroll
        for (int t = 0; t < HT; t++)
#pragma unroll
            for (int g = 0; g < 16; g++) acc1[t][g] = 0.f;
        for (int s = 0; s < half; s++) {
            const int k = hi * half + s;
            float xb = 0.f;
            if (live && k < dO) {
                if constexpr (sizeof(XT) == 2) xb = __half2float(xr[s]);
                else xb = xr[s];
            }
#pragma unroll
            for (int t = 0; t < HT; t++) {
                const float wa = w1[(t * 32 + col) * s1 + k];  // A[i = unit][k]
                acc1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, xb, acc1[t], 0, 0, 0);
            }
        }
        // bias + LeakyReLU(0.01); lane now holds hidden units cd_row(g, hi) + 32 t of data row `col`
#pragma unroll
        for (int t = 0; t < HT; t++)
#pragma unroll
            for (int g = 0; g < 16; g++) {
                float v = acc1[t][g] + bb1[t * 32 + cd_row(g, hi)];
                acc1[t][g] = v > 0.f ? v : 0.01f * v;
            }
        // ---- layer 2: k-step (t, g) pairs unit cd_row(g,0)+32t (hi = 0 lanes) with cd_row(g,1)+32t (hi = 1 lanes) ----
        f32x16 acc2[ZT];
#pragma unroll
        for (int zt = 0; zt < ZT; zt++)
#pragma unroll
            for (int g = 0; g < 16; g++) acc2[zt][g] = 0.f;
#pragma unroll
        for (int t = 0; t < HT; t++)
#pragma unroll
            for (int g = 0; g < 16; g++) {
                const int unit = t * 32 + cd_row(g, hi);
                const float hb = acc1[t][g];  // B[k = unit][j = data row]
#pragma unroll
                for (int zt = 0; zt < ZT; zt++) {
                    const float wa = w2[(zt * 32 + col) * s2 + unit];  // A[i = z][k = unit]
                    acc2[zt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, hb, acc2[zt], 0, 0, 0);
                }
            }
        // ---- bias, optional logits, argmax over z (first maximal index, as torch.max(dim=1)) ----
        float bv = -__builtin_inff();
        int bz = 0x7fffffff;
#pragma unroll
        for (int zt = 0; zt < ZT; zt++)
#pragma unroll
            for (int g = 0; g < 16; g++) {
                const int zc = zt * 32 + cd_row(g, hi);
                const float v = acc2[zt][g] + bb2[zc];
                if (zc < nZ) {
                    if (out_logits && live) out_logits[row * nZ + zc] = v;
                    if (v > bv || (v == bv && zc < bz)) {
                        bv = v;
                        bz = zc;
                    }
                }
            }
        // the two lane halves hold disjoint z sets of the same data row: combine
        const float ov = __shfl_xor(bv, 32);
        const int oz = __shfl_xor(bz, 32);
        if (ov > bv || (ov == bv && oz < bz)) {
            bv = ov;
            bz = oz;
        }
        if (live && hi == 0) out_z[row] = bz;
    }
}

// ---- the same forward with the WEIGHTS IN REGISTERS (shapes with a compile-time observation width) ----
// A wavefront of the kernel above spends its time fetching operands: one ds_read per MFMA for the weight and, for the observation,
// one two- or four-byte load per k-step with a lane stride of a whole row (measured: 26 % / 36 % of the f32 matrix peak at C5's and
// C3's shapes).  But every MFMA of a tile reads the same weight element in the same lane as the corresponding MFMA of the last tile:
// lane (col, hi) of MFMA (t, s) of layer 1 always needs W1[32 t + col][hi half + s], of MFMA (t, g, zt) of layer 2 always
// W2[32 zt + col][32 t + cd_row(g, hi)].  So each lane keeps its HT half + ZT HT 16 weights (192 at 128-64-50) and the biases in
// registers for the whole launch -- one wavefront per SIMD has 512 of them --, an observation row's half (k in [hi half, hi half +
// half): contiguous) arrives as a few 16-byte loads issued a tile ahead, and the loop is MFMAs with conversions in their shadow.
// The order of the k summation is the kernel's above, so the logits are the same bit for bit.
template <typename XT, int DO, int HT, int ZT, int WPE /* wavefronts per SIMD the registers are budgeted for */>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
    k_encode_mlp_mfma_reg(const XT *__restrict__ x, int64_t N, const float *__restrict__ W1, const float *__restrict__ b1, int H,
                          const float *__restrict__ W2, const float *__restrict__ b2, int nZ, int32_t *__restrict__ out_z,
                          float *__restrict__ out_logits) {
    constexpr int DOP = (DO + 1) & ~1, HALF = DOP / 2;
    constexpr int XB = HALF * (int)sizeof(XT);                  // bytes of an observation row one lane reads
    constexpr int XW = XB % 16 == 0 ? XB / 16 : 0;              // ... as 16-byte words (0: element by element)
    const int lane = threadIdx.x & 63, col = lane & 31, hi = lane >> 5;
    // ---- this lane's weights and biases ----
    float w1r[HT][HALF], w2r[ZT][HT][16], bb1[HT][16], bb2[ZT][16];
#pragma unroll
    for (int t = 0; t < HT; t++)
#pragma unroll
        for (int s2 = 0; s2 < HALF; s2++) {
            const int u = t * 32 + col, k = hi * HALF + s2;
            w1r[t][s2] = (u < H && k < DO) ? W1[(int64_t)u * DO + k] : 0.f;
        }
#pragma unroll
    for (int zt = 0; zt < ZT; zt++)
#pragma unroll
        for (int t = 0; t < HT; t++)
#pragma unroll
            for (int g = 0; g < 16; g++) {
                const int zc = zt * 32 + col, unit = t * 32 + cd_row(g, hi);
                w2r[zt][t][g] = (zc < nZ && unit < H) ? W2[(int64_t)zc * H + unit] : 0.f;
            }
#pragma unroll
    for (int t = 0; t < HT; t++)
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const int unit = t * 32 + cd_row(g, hi);
            bb1[t][g] = unit < H ? b1[unit] : 0.f;
        }
#pragma unroll
    for (int zt = 0; zt < ZT; zt++)
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const int zc = zt * 32 + cd_row(g, hi);
            bb2[zt][g] = zc < nZ ? b2[zc] : 0.f;
        }
    const int64_t wave_global = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    // PF tiles are fetched at once, a whole group of PF ahead: a tile of narrow observations (34 MFMAs at 2-64-25) is shorter than a
    // memory round trip
    constexpr int XR = XW ? XW : 1, PF = XB >= 64 ? 1 : 2;
    u32x4 xq[PF][XR];   // the group in flight
    XT xe[PF][XW ? 1 : HALF];
    auto fetch = [&](int64_t grp0) {
#pragma unroll
        for (int f = 0; f < PF; f++) {
            const int64_t row = grp0 + (int64_t)f * 32 + col;
            const XT *xr = x + (row < N ? row : 0) * DO + hi * HALF;  // (rows beyond the end read row 0: computed, never stored)
            if constexpr (XW > 0) {
#pragma unroll
                for (int q = 0; q < XW; q++) xq[f][q] = ((const u32x4 *)xr)[q];
            } else {
#pragma unroll
                for (int s2 = 0; s2 < HALF; s2++) xe[f][s2] = (hi * HALF + s2 < DO) ? xr[s2] : (XT)0;
            }
        }
    };
    int64_t grp0 = wave_global * 32 * PF;
    if (grp0 < N) fetch(grp0);
    for (; grp0 < N; grp0 += n_waves * 32 * PF) {
      // this group's observations as f32, then the next group's loads go out
      float xf[PF][HALF];
#pragma unroll
      for (int f = 0; f < PF; f++) {
        if constexpr (XW > 0) {
#pragma unroll
            for (int q = 0; q < XW; q++) {
                if constexpr (sizeof(XT) == 2) {
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const uint32_t wv = xq[f][q][e];
                        xf[f][q * 8 + e * 2] = __half2float(__ushort_as_half((unsigned short)(wv & 0xffffu)));
                        xf[f][q * 8 + e * 2 + 1] = __half2float(__ushort_as_half((unsigned short)(wv >> 16)));
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; e++) xf[f][q * 4 + e] = __uint_as_float(xq[f][q][e]);
                }
            }
        } else {
#pragma unroll
            for (int s2 = 0; s2 < HALF; s2++) {
                if constexpr (sizeof(XT) == 2) xf[f][s2] = __half2float(xe[f][s2]);
                else xf[f][s2] = (float)xe[f][s2];
            }
        }
      }
      if (grp0 + n_waves * 32 * PF < N) fetch(grp0 + n_waves * 32 * PF);
#pragma unroll
      for (int f = 0; f < PF; f++) {
        const int64_t row0 = grp0 + (int64_t)f * 32;
        if (row0 >= N) break;
        const int64_t row = row0 + col;
        const bool live = row < N;
        // ---- layer 1 (the accumulators start from the bias; LeakyReLU(0.01) = max(v, 0.01 v)) ----
        f32x16 acc1[HT];
#pragma unroll
        for (int t = 0; t < HT; t++)
#pragma unroll
            for (int g = 0; g < 16; g++) acc1[t][g] = bb1[t][g];
#pragma unroll
        for (int s2 = 0; s2 < HALF; s2++) {
#pragma unroll
            for (int t = 0; t < HT; t++) acc1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1r[t][s2], xf[f][s2], acc1[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < HT; t++)
#pragma unroll
            for (int g = 0; g < 16; g++) acc1[t][g] = fmaxf(acc1[t][g], 0.01f * acc1[t][g]);
        // ---- layer 2 (from the bias) ----
        f32x16 acc2[ZT];
#pragma unroll
        for (int zt = 0; zt < ZT; zt++)
#pragma unroll
            for (int g = 0; g < 16; g++) acc2[zt][g] = bb2[zt][g];
#pragma unroll
        for (int t = 0; t < HT; t++)
#pragma unroll
            for (int g = 0; g < 16; g++) {
#pragma unroll
                for (int zt = 0; zt < ZT; zt++) acc2[zt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2r[zt][t][g], acc1[t][g], acc2[zt], 0, 0, 0);
            }
        // ---- optional logits, argmax over z (first maximal index, as torch.max(dim=1): the z of a lane are walked downwards, >= wins) ----
        float bv = -__builtin_inff();
        int bz = 0x7fffffff;
#pragma unroll
        for (int zt = ZT - 1; zt >= 0; zt--)
#pragma unroll
            for (int g = 15; g >= 0; g--) {
                const int zc = zt * 32 + cd_row(g, hi);  // (ascending in g for a fixed zt and hi)
                const float v = acc2[zt][g];
                if (zc < nZ) {
                    if (out_logits && live) out_logits[row * nZ + zc] = v;
                    if (v >= bv) {
                        bv = v;
                        bz = zc;
                    }
                }
            }
        const float ov = __shfl_xor(bv, 32);
        const int oz = __shfl_xor(bz, 32);
        if (ov > bv || (ov == bv && oz < bz)) {
            bv = ov;
            bz = oz;
        }
        if (live && hi == 0) out_z[row] = bz;
      }
    }
}

}  // namespace offsim
