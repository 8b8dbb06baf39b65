// fp32 products on the fp16 matrix pipe with THREE MFMAs per 16 k-elements (round 6).
//
// The six-product bf16 form (gcp_bf16x3.h) is exact to 3 * 2^-24 |a b| and costs six v_mfma_f32_32x32x16_bf16 per product block.
// fp16 carries 11 significant bits: x = h + l + e with h = RN16(x), l = RN16(x - h) (the residual is exact in fp32), |e| <= 2^-22 |x|.
// Of the four products the three with weight >= 2^-11 are kept,
//     hh,  hl, lh
// (fp16 x fp16 is exact in fp32; v_mfma_f32_32x32x16_f16 accumulates in fp32 and runs at the bf16 rate): the dropped ll and the two
// e terms are <= 3 * 2^-22 |a b| = 7e-7 -- HALF the matrix-pipe time for two more bits of round-off, still an order of magnitude inside
// the 1e-5 parity bound.
//
// fp16 has five exponent bits, so each operand is scaled by a power of two (exact) that is CONSTANT ALONG THE SUMMED INDEX:
//   * activations / gradients (split on the fly, one row of the tile per lane pair): 2^pa per ROW from the row's largest magnitude, so
//     that it lands in [2^14, 2^15); elements below 2^-28 of their row's maximum lose relative precision (absolute error 2^-39 of the
//     maximum: nothing a dot product can see);
//   * weights (split once per optimizer step by the pack kernels): the fixed factor 2^GCP_F16_WEXP -- full 22-bit precision for
//     |w| in [2^-9 .. 2^9), absolute error <= 2^-31 below, saturation (never inf) above;
// the accumulators are multiplied by 2^(pa + GCP_F16_WEXP) before the products are added and by its inverse afterwards (both exact:
// pa is clamped to [-60, 60] so that no sane value over- or underflows), so bias, residual state and the fp32-MFMA parts of a sum keep their bits.
//
// Operand layout of v_mfma_f32_32x32x16_f16: that of the bf16 instruction (gcp_bf16x3.h).
#pragma once
#include "common.h"
#include "gcp_bf16x3.h"

typedef _Float16 gcp_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gcp_f16x2 __attribute__((ext_vector_type(2)));

#define GCP_F16_WEXP 6

__device__ __forceinline__ f32x16 gcp_mfma_f16(gcp_u32x4 a, gcp_u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gcp_f16x8, a), __builtin_bit_cast(gcp_f16x8, b), c, 0, 0, 0);
}

// c += (ah + al) (bh + bl), the three kept products, small terms first
__device__ __forceinline__ f32x16 gcp_mfma_f16x3(const gcp_u32x4 (&a)[2], gcp_u32x4 bh, gcp_u32x4 bl, f32x16 c) {
    c = gcp_mfma_f16(a[1], bh, c);
    c = gcp_mfma_f16(a[0], bl, c);
    c = gcp_mfma_f16(a[0], bh, c);
    return c;
}

__device__ __forceinline__ unsigned gcp_f16_pack_rn(float x0, float x1) {
    const gcp_f16x2 v = {(_Float16)x0, (_Float16)x1};  // (round to nearest even)
    return __builtin_bit_cast(unsigned, v);
}

// eight fp32 values times `scale` (a power of two) -> two f16x8 terms
__device__ __forceinline__ void gcp_f16x2_split8(const float (&x)[8], float scale, gcp_u32x4& h, gcp_u32x4& l) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = x[2 * j] * scale, b = x[2 * j + 1] * scale;
        const gcp_f16x2 hp = {(_Float16)a, (_Float16)b};
        h[j] = __builtin_bit_cast(unsigned, hp);
        l[j] = gcp_f16_pack_rn(a - (float)hp[0], b - (float)hp[1]);
    }
}

// Row scale: the power of two that brings `m` (the largest magnitude of the row, >= 0) into [2^14, 2^15), clamped to [2^-60, 2^60]
// (rows below 2^-46 = 1.4e-14 keep an ABSOLUTE error of 2^-91 |w| per term instead; accumulators up to 2^60 survive the scaling);
// an all-zero (or subnormal) row: 2^0.  Returns the exponent pa (scale = 2^pa)
__device__ __forceinline__ int gcp_f16_row_exp(float m) {
    const int eb = (int)((__float_as_uint(m) >> 23) & 0xffu);  // biased exponent: floor(log2 m) + 127 for normal m, 0 for zero / subnormal
    return eb == 0 ? 0 : min(max(14 + 127 - eb, -60), 60);
}
__device__ __forceinline__ float gcp_exp2i(int p) { return __uint_as_float((unsigned)(p + 127) << 23); }  // 2^p, p in [-126, 127]

// term t (0 = h, 1 = l) of one WEIGHT (scaled by 2^GCP_F16_WEXP, saturated to the fp16 range), as 16 bits
__host__ __device__ inline unsigned gcp_f16x2_wterm(float w, int t) {
    float x = w * (float)(1 << GCP_F16_WEXP);
    x = x > 65504.f ? 65504.f : (x < -65504.f ? -65504.f : x);
    const _Float16 h = (_Float16)x;
    const _Float16 v = t == 0 ? h : (_Float16)(x - (float)h);
    unsigned short bits;
    __builtin_memcpy(&bits, &v, 2);
    return bits;
}
