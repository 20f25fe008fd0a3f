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
// (Round 4: the register-resident kernel below exists in a second form on bf16 x 3 products, the default for its shapes.)
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


// ---- bf16 x 3 (round 4) -------------------------------------------------------------------------------------------------------
// The f32 MFMA above retires 2 k per 64 cycles; v_mfma_f32_32x32x16_bf16 retires 16 k per 32 cycles (MI355X_MICROARCH.md), and an f32
// value IS three bf16 values: h = the top 8 significant bits (the f32 word truncated to its high half), m = the top 8 of v - h, l = v - h
// - m (what is left has at most 8): the split is exact, every difference is exact, and a product of two bf16 is exact in f32.  Of the
// nine partial products the three smallest (m l, l m, l l: below 2^-24 of h h) are dropped, the other six are accumulated in f32 by the
// matrix core, smallest first: 6 x 32 cycles per 16 k = 12 cycles per k against 32 -- the logits differ from the f32 kernel's in the
// last bits (tolerance-based parity, as above: 1e-5).  fp16 observations are two bf16 values (11 significant bits).
// Operand layout of the 32x32x16 product: lane l holds A[m = l & 31][k = 8 (l >> 5) + i] and B[k = 8 (l >> 5) + i][n = l & 31], i = 0..7,
// as four dwords of two bf16 each (even i in the low half); C/D as for the f32 product.  The hidden layer again never leaves the
// registers: registers 8 c .. 8 c + 7 of a C tile (c = 0, 1) ARE a B operand -- hidden units 8 (2 c + i / 4) + 4 hi + i % 4 of the tile --
// and the W2 operand is laid down in that order of k once, at the top of the kernel.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t enc_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 enc_bf(enc_u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
// the high halves of two f32 words as a pair of bf16: [b : a]
__device__ __forceinline__ uint32_t enc_pack_hi(float a, float b) {
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float enc_trunc_bf16(float v) { return __uint_as_float(__float_as_uint(v) & 0xffff0000u); }
// (a, b) -> the pairs of their h, m, l parts
struct EncParts { uint32_t h, m, l; };
__device__ __forceinline__ EncParts enc_split3(float a, float b) {
    EncParts p;
    p.h = enc_pack_hi(a, b);
    const float ra = a - enc_trunc_bf16(a), rb = b - enc_trunc_bf16(b);
    p.m = enc_pack_hi(ra, rb);
    p.l = enc_pack_hi(ra - enc_trunc_bf16(ra), rb - enc_trunc_bf16(rb));
    return p;
}
#define ENC_SPLIT3(A, B, H, M, L) do { const EncParts _p = enc_split3(A, B); H = _p.h; M = _p.m; L = _p.l; } while (0)
#define ENC_MFMA_BF16(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_bf16(enc_bf(A), enc_bf(B), C, 0, 0, 0)

template <typename XT, int DO, int HT, int ZT, int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
    k_encode_mlp_mfma_split(const XT *__restrict__ x, int64_t N, const float *__restrict__ W1, const float *__restrict__ b1, int H,
                            const float *__restrict__ W2, const float *__restrict__ b2, int nZ, int32_t *__restrict__ out_z,
                            float *__restrict__ out_logits) {
    constexpr int DOP = (DO + 1) & ~1, HALF = DOP / 2;
    constexpr int XB = HALF * (int)sizeof(XT);
    constexpr int XW = XB % 16 == 0 ? XB / 16 : 0;
    constexpr bool L1_SPLIT = HALF % 8 == 0 && XW > 0;  // layer 1 on the bf16 product: a lane's share of a row is whole groups of eight
    constexpr int S1 = L1_SPLIT ? HALF / 8 : 1;          // ... that many products per hidden tile
    constexpr int XPARTS = sizeof(XT) == 2 ? 2 : 3;      // bf16 parts of an observation
    const int lane = threadIdx.x & 63, col = lane & 31, hi = lane >> 5;
    // ---- this lane's weights and biases ----
    float w1r[L1_SPLIT ? 1 : HT][L1_SPLIT ? 1 : HALF];
    enc_u32x4 w1s[L1_SPLIT ? HT : 1][S1][3], w2s[ZT][HT][2][3];
    float bb1[HT][16], bb2[ZT][16];
    if constexpr (L1_SPLIT) {
#pragma unroll
        for (int t = 0; t < HT; t++)
#pragma unroll
            for (int s = 0; s < S1; s++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int u = t * 32 + col, k = hi * HALF + 8 * s + 2 * q;
                    const float a = (u < H && k < DO) ? W1[(int64_t)u * DO + k] : 0.f, b = (u < H && k + 1 < DO) ? W1[(int64_t)u * DO + k + 1] : 0.f;
                    ENC_SPLIT3(a, b, w1s[t][s][0][q], w1s[t][s][1][q], w1s[t][s][2][q]);
                }
    } else {
#pragma unroll
        for (int t = 0; t < HT; t++)
#pragma unroll
            for (int s2 = 0; s2 < HALF; s2++) {
                const int u = t * 32 + col, k = hi * HALF + s2;
                w1r[t][s2] = (u < H && k < DO) ? W1[(int64_t)u * DO + k] : 0.f;
            }
    }
#pragma unroll
    for (int zt = 0; zt < ZT; zt++)
#pragma unroll
        for (int t = 0; t < HT; t++)
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int zc = zt * 32 + col;
                    const int i0 = 2 * q, i1 = 2 * q + 1;
                    const int u0 = t * 32 + 8 * (2 * c + i0 / 4) + 4 * hi + i0 % 4, u1 = t * 32 + 8 * (2 * c + i1 / 4) + 4 * hi + i1 % 4;
                    const float a = (zc < nZ && u0 < H) ? W2[(int64_t)zc * H + u0] : 0.f, b = (zc < nZ && u1 < H) ? W2[(int64_t)zc * H + u1] : 0.f;
                    ENC_SPLIT3(a, b, w2s[zt][t][c][0][q], w2s[zt][t][c][1][q], w2s[zt][t][c][2][q]);
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
    constexpr int XR = XW ? XW : 1, PF = XB >= 64 ? 1 : 2;
    enc_u32x4 xq[PF][XR];
    XT xe[PF][XW ? 1 : HALF];
    auto fetch = [&](int64_t grp0) {
#pragma unroll
        for (int f = 0; f < PF; f++) {
            const int64_t row = grp0 + (int64_t)f * 32 + col;
            const XT *xr = x + (row < N ? row : 0) * DO + hi * HALF;
            if constexpr (XW > 0) {
#pragma unroll
                for (int q = 0; q < XW; q++) xq[f][q] = ((const enc_u32x4 *)xr)[q];
            } else {
#pragma unroll
                for (int s2 = 0; s2 < HALF; s2++) xe[f][s2] = (hi * HALF + s2 < DO) ? xr[s2] : (XT)0;
            }
        }
    };
    int64_t grp0 = wave_global * 32 * PF;
    if (grp0 < N) fetch(grp0);
    for (; grp0 < N; grp0 += n_waves * 32 * PF) {
      // this group's observations: as f32 where layer 1 runs on the f32 product (narrow observations); otherwise the raw words are kept
      // and converted eight at a time (64 registers fewer at 128 fp16 values per lane); then the next group's loads go out
      float xf[L1_SPLIT ? 1 : PF][L1_SPLIT ? 1 : HALF];
      enc_u32x4 xc[L1_SPLIT ? PF : 1][L1_SPLIT ? XR : 1];
      if constexpr (L1_SPLIT) {
#pragma unroll
          for (int f = 0; f < PF; f++)
#pragma unroll
              for (int q = 0; q < XR; q++) xc[f][q] = xq[f][q];
      } else {
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
      }
      if (grp0 + n_waves * 32 * PF < N) fetch(grp0 + n_waves * 32 * PF);
#pragma unroll
      for (int f = 0; f < PF; f++) {
        const int64_t row0 = grp0 + (int64_t)f * 32;
        if (row0 >= N) break;
        const int64_t row = row0 + col;
        const bool live = row < N;
        // ---- layer 1 (from the bias; LeakyReLU(0.01) = max(v, 0.01 v)) ----
        f32x16 acc1[HT];
#pragma unroll
        for (int t = 0; t < HT; t++)
#pragma unroll
            for (int g = 0; g < 16; g++) acc1[t][g] = bb1[t][g];
        if constexpr (L1_SPLIT) {
#pragma unroll
            for (int s = 0; s < S1; s++) {
                enc_u32x4 xh, xm, xl = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    float a, b;  // elements 8 s + 2 q and 8 s + 2 q + 1 of this lane's share of the row
                    if constexpr (sizeof(XT) == 2) {
                        const uint32_t wv = xc[f][s][q];
                        a = __half2float(__ushort_as_half((unsigned short)(wv & 0xffffu)));
                        b = __half2float(__ushort_as_half((unsigned short)(wv >> 16)));
                    } else {
                        a = __uint_as_float(xc[f][2 * s + q / 2][(2 * q) & 3]);
                        b = __uint_as_float(xc[f][2 * s + q / 2][(2 * q + 1) & 3]);
                    }
                    if constexpr (XPARTS == 2) {  // fp16: 11 significant bits = two bf16, exactly
                        xh[q] = enc_pack_hi(a, b);
                        xm[q] = enc_pack_hi(a - enc_trunc_bf16(a), b - enc_trunc_bf16(b));
                    } else {
                        ENC_SPLIT3(a, b, xh[q], xm[q], xl[q]);
                    }
                }
#pragma unroll
                for (int t = 0; t < HT; t++) {
                    if constexpr (XPARTS == 3) acc1[t] = ENC_MFMA_BF16(w1s[t][s][0], xl, acc1[t]);
                    acc1[t] = ENC_MFMA_BF16(w1s[t][s][2], xh, acc1[t]);
                    acc1[t] = ENC_MFMA_BF16(w1s[t][s][1], xm, acc1[t]);
                    acc1[t] = ENC_MFMA_BF16(w1s[t][s][0], xm, acc1[t]);
                    acc1[t] = ENC_MFMA_BF16(w1s[t][s][1], xh, acc1[t]);
                    acc1[t] = ENC_MFMA_BF16(w1s[t][s][0], xh, acc1[t]);
                }
            }
        } else {
#pragma unroll
            for (int s2 = 0; s2 < HALF; s2++) {
#pragma unroll
                for (int t = 0; t < HT; t++) acc1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1r[t][s2], xf[f][s2], acc1[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < HT; t++)
#pragma unroll
            for (int g = 0; g < 16; g++) acc1[t][g] = fmaxf(acc1[t][g], 0.01f * acc1[t][g]);
        // ---- layer 2 (from the bias): registers 8 c .. 8 c + 7 of a hidden tile are the B operand of one product ----
        f32x16 acc2[ZT];
#pragma unroll
        for (int zt = 0; zt < ZT; zt++)
#pragma unroll
            for (int g = 0; g < 16; g++) acc2[zt][g] = bb2[zt][g];
#pragma unroll
        for (int t = 0; t < HT; t++)
#pragma unroll
            for (int c = 0; c < 2; c++) {
                enc_u32x4 hh, hm, hl;
#pragma unroll
                for (int q = 0; q < 4; q++) ENC_SPLIT3(acc1[t][8 * c + 2 * q], acc1[t][8 * c + 2 * q + 1], hh[q], hm[q], hl[q]);
#pragma unroll
                for (int zt = 0; zt < ZT; zt++) {
                    acc2[zt] = ENC_MFMA_BF16(w2s[zt][t][c][2], hh, acc2[zt]);
                    acc2[zt] = ENC_MFMA_BF16(w2s[zt][t][c][0], hl, acc2[zt]);
                    acc2[zt] = ENC_MFMA_BF16(w2s[zt][t][c][1], hm, acc2[zt]);
                    acc2[zt] = ENC_MFMA_BF16(w2s[zt][t][c][1], hh, acc2[zt]);
                    acc2[zt] = ENC_MFMA_BF16(w2s[zt][t][c][0], hm, acc2[zt]);
                    acc2[zt] = ENC_MFMA_BF16(w2s[zt][t][c][0], hh, acc2[zt]);
                }
            }
        // ---- optional logits, argmax over z (first maximal index, as torch.max(dim=1)) ----
        float bv = -__builtin_inff();
        int bz = 0x7fffffff;
#pragma unroll
        for (int zt = ZT - 1; zt >= 0; zt--)
#pragma unroll
            for (int g = 15; g >= 0; g--) {
                const int zc = zt * 32 + cd_row(g, hi);
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
#undef ENC_MFMA_BF16
#undef ENC_SPLIT3

}  // namespace offsim
