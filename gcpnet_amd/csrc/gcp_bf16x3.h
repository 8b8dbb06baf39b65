// fp32 products on the bf16 matrix pipe, exact to fp32 round-off.
//
// v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (64 FLOP / clk / SIMD); v_mfma_f32_32x32x16_bf16 at 16 x that.  An
// fp32 number is the exact sum of three bf16 numbers, x = h + m + l (8 significant bits each, taken by truncation: the
// residuals x - h and x - h - m are exact in fp32).  Of the nine products of two such sums the six with total weight >= 2^-16,
//     hh,  hm, mh,  hl, lh, mm,
// are kept (bf16 x bf16 is exact in fp32; the MFMA accumulates in fp32); the dropped ml, lm, ll are <= 3 * 2^-24 |a b|, i.e.
// fp32 round-off.  Measured (tools/ubench/bf16x6.hip, K = 128, against float64): 1.8e-7 of sum |a b| against 3.0e-7 for the
// fp32 MFMA chain; 2.2 - 2.4 x its throughput including the on-the-fly split of one operand.
// The other operand (weights) is split once per optimizer step by the pack kernels.
//
// Operand layout of v_mfma_f32_32x32x16_bf16 (lane l = 32 hi + e): A[i = e][k = 8 hi .. 8 hi + 7], B[k = 8 hi .. + 7][j = e], eight
// bf16 per lane in four VGPRs, element 2 d in the low half of dword d; C/D as for every 32 x 32 MFMA.
#pragma once
#include "common.h"

typedef __bf16 gcp_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned gcp_u32x4 __attribute__((ext_vector_type(4)));

// (bf16 trunc(x0), bf16 trunc(x1)) in one register
__device__ __forceinline__ unsigned gcp_bf16_pack_hi(float x0, float x1) {
    return __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
}

// eight fp32 values -> three bf16x8 terms
__device__ __forceinline__ void gcp_bf16x3_split8(const float (&x)[8], gcp_u32x4& h, gcp_u32x4& m, gcp_u32x4& l) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = x[2 * j], b = x[2 * j + 1];
        const unsigned hp = gcp_bf16_pack_hi(a, b);
        const float ra = a - __uint_as_float(hp << 16), rb = b - __uint_as_float(hp & 0xffff0000u);
        const unsigned mp = gcp_bf16_pack_hi(ra, rb);
        const float sa = ra - __uint_as_float(mp << 16), sb = rb - __uint_as_float(mp & 0xffff0000u);
        h[j] = hp; m[j] = mp; l[j] = gcp_bf16_pack_hi(sa, sb);
    }
}

// term t (0 = h, 1 = m, 2 = l) of one value, as the upper 16 bits of its fp32 pattern
__host__ __device__ inline unsigned gcp_bf16x3_term(float x, int t) {
    union { float f; unsigned u; } c;
    for (int k = 0; k < t; ++k) {
        c.f = x;
        c.u &= 0xffff0000u;
        x -= c.f;
    }
    c.f = x;
    return c.u >> 16;
}

__device__ __forceinline__ f32x16 gcp_mfma_bf16(gcp_u32x4 a, gcp_u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gcp_bf16x8, a), __builtin_bit_cast(gcp_bf16x8, b), c, 0, 0, 0);
}

// c += (ah + am + al) (bh + bm + bl), the six kept products, small terms first
__device__ __forceinline__ f32x16 gcp_mfma_bf16x6(const gcp_u32x4 (&a)[3], gcp_u32x4 bh, gcp_u32x4 bm, gcp_u32x4 bl, f32x16 c) {
#ifdef GCP_X3_PROBE  // (measurement build: three of the six products -- what halving the matrix work is worth to a launch; results imprecise)
    c = gcp_mfma_bf16(a[1], bh, c);
    c = gcp_mfma_bf16(a[0], bm, c);
    c = gcp_mfma_bf16(a[0], bh, c);
    return c;
#endif
    c = gcp_mfma_bf16(a[2], bh, c);
    c = gcp_mfma_bf16(a[0], bl, c);
    c = gcp_mfma_bf16(a[1], bm, c);
    c = gcp_mfma_bf16(a[1], bh, c);
    c = gcp_mfma_bf16(a[0], bm, c);
    c = gcp_mfma_bf16(a[0], bh, c);
    return c;
}
