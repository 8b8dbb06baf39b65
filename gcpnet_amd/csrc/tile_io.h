// Wave-private tile movers for the 32-row GCP kernels.  The point of this file is memory-level parallelism: a wave
// issues a batch of 16-byte loads (up to 8 per lane, 8 KiB per wave) before it touches any of the results, instead of
// one dependent row at a time; a tile is then on chip after about one HBM latency.
#pragma once
#include "common.h"

__device__ __forceinline__ bool gcp_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// tile[e * stride + coff + j] = src[(idx ? idx[r0 + e] : r0 + e) * dim + j]   for e < 32, j < dim; zeros past `rows`.
__device__ __forceinline__ void gcp_load_segment(const float* __restrict__ base, const int32_t* __restrict__ idx, int dim,
                                                 int r0, int rows, float* tile, int stride, int coff, int lane) {
    const int q = dim >> 2;  // 16-byte pieces per row
    if ((dim & 3) == 0 && q <= 64 && gcp_aligned16(base)) {
        const int rpi = 64 / q;  // rows covered by one wave-wide load instruction
        const int sub = lane / q, c4 = lane - sub * q;
        const bool lane_on = sub < rpi;
        constexpr int B = 8;
        for (int e0 = 0; e0 < GCP_TILE_ROWS; e0 += rpi * B) {
            float4 buf[B];
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const int e = e0 + b * rpi + sub;
                buf[b] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (lane_on && e < GCP_TILE_ROWS && r0 + e < rows) {
                    const int64_t src = idx ? (int64_t)idx[r0 + e] : (int64_t)(r0 + e);
                    buf[b] = *reinterpret_cast<const float4*>(base + src * dim + 4 * c4);
                }
            }
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const int e = e0 + b * rpi + sub;
                if (lane_on && e < GCP_TILE_ROWS) {
                    float* d = tile + e * stride + coff + 4 * c4;
                    d[0] = buf[b].x; d[1] = buf[b].y; d[2] = buf[b].z; d[3] = buf[b].w;
                }
            }
        }
        return;
    }
    // generic widths: one row per pass, dword accesses
#pragma unroll 4
    for (int e = 0; e < GCP_TILE_ROWS; ++e) {
        const int r = r0 + e;
        float* dst = tile + e * stride + coff;
        if (r < rows) {
            const int64_t src = idx ? (int64_t)idx[r] : (int64_t)r;
            const float* rowp = base + src * dim;
            for (int j = lane; j < dim; j += GCP_WAVE) dst[j] = rowp[j];
        } else {
            for (int j = lane; j < dim; j += GCP_WAVE) dst[j] = 0.f;
        }
    }
}

__device__ __forceinline__ void gcp_load_concat_tile(const gcp_concat_t& c, int mult, int r0, int rows, float* tile,
                                                     int stride, int lane) {
    int coff = 0;
    for (int sg = 0; sg < c.n; ++sg) {
        const int dim = c.dim[sg] * mult;
        gcp_load_segment(c.ptr[sg], c.idx[sg], dim, r0, rows, tile, stride, coff, lane);
        coff += dim;
    }
}

// dst[(r0 + e) * ld + j] = tile[e * stride + j] (+ res[(r0 + e) * ld + j])   for e < 32 with r0 + e < rows, j < width.
__device__ __forceinline__ void gcp_store_tile(float* __restrict__ dst, int64_t ld, int col0, int width, int r0, int rows,
                                               const float* tile, int stride, int lane) {
    const int q = width >> 2;
    if ((width & 3) == 0 && (ld & 3) == 0 && (col0 & 3) == 0 && q <= 64 && q > 0 && gcp_aligned16(dst)) {
        const int rpi = 64 / q;
        const int sub = lane / q, c4 = lane - sub * q;
        if (sub < rpi) {
            for (int e = sub; e < GCP_TILE_ROWS && r0 + e < rows; e += rpi) {
                const float* s = tile + e * stride + 4 * c4;
                *reinterpret_cast<float4*>(dst + (int64_t)(r0 + e) * ld + col0 + 4 * c4) = make_float4(s[0], s[1], s[2], s[3]);
            }
        }
        return;
    }
    for (int e = 0; e < GCP_TILE_ROWS && r0 + e < rows; ++e)
        for (int j = lane; j < width; j += GCP_WAVE) dst[(int64_t)(r0 + e) * ld + col0 + j] = tile[e * stride + j];
}
