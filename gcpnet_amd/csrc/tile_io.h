// Wave-private tile movers for the 32-row GCP kernels.  The point of this file is memory-level parallelism: a wave
// issues a batch of 16-byte loads (up to 8 per lane, 8 KiB per wave) before it touches any of the results, instead of
// one dependent row at a time; a tile is then on chip after about one HBM latency.
#pragma once
#include <type_traits>
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
        constexpr int B = 16;
        const int rmax = min(rows, r0 + GCP_TILE_ROWS) - 1;  // loads are UNCONDITIONAL (clamped address + select): a guarded
        const int c4c = lane_on ? c4 : 0;                    // load makes hipcc branch around it and wait vmcnt(0) per element
        for (int e0 = 0; e0 < GCP_TILE_ROWS; e0 += rpi * B) {
            float4 buf[B];
            if (idx) {
                int src[B];
#pragma unroll
                for (int b = 0; b < B; ++b) src[b] = idx[min(r0 + e0 + b * rpi + sub, rmax)];
#pragma unroll
                for (int b = 0; b < B; ++b) buf[b] = *reinterpret_cast<const float4*>(base + (int64_t)src[b] * dim + 4 * c4c);
            } else {
#pragma unroll
                for (int b = 0; b < B; ++b)
                    buf[b] = *reinterpret_cast<const float4*>(base + (int64_t)min(r0 + e0 + b * rpi + sub, rmax) * dim + 4 * c4c);
            }
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const int e = e0 + b * rpi + sub;
                if (lane_on && e < GCP_TILE_ROWS) {
                    const bool ok = r0 + e < rows;
                    float* d = tile + e * stride + coff + 4 * c4;
                    d[0] = ok ? buf[b].x : 0.f; d[1] = ok ? buf[b].y : 0.f; d[2] = ok ? buf[b].z : 0.f; d[3] = ok ? buf[b].w : 0.f;
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

// Deferred variant: `issue` puts the whole segment (<= B load instructions) in flight and returns; `commit` writes
// it to the tile.  Several segments (scalars, vectors, gradients ...) can thus share ONE memory round trip.
template <int B>
struct GcpSegBuf {
    float4 v[B];
    int rpi, sub, c4, nvalid;
    bool deferred, lane_on;
};

template <int B>
__device__ __forceinline__ void gcp_seg_issue(GcpSegBuf<B>& sb, const float* __restrict__ base,
                                              const int32_t* __restrict__ idx, int dim, int r0, int rows, float* tile,
                                              int stride, int coff, int lane) {
    const int q = dim >> 2;
    sb.deferred = false;
    if ((dim & 3) == 0 && q > 0 && q <= 64 && gcp_aligned16(base) && gcp_cdiv(GCP_TILE_ROWS, 64 / q) <= B) {
        sb.deferred = true;
        sb.rpi = 64 / q;
        sb.sub = lane / q;
        sb.c4 = lane - sb.sub * q;
        sb.lane_on = sb.sub < sb.rpi;
        sb.nvalid = min(rows - r0, GCP_TILE_ROWS);
        const int rmax = r0 + sb.nvalid - 1;
        const int c4c = sb.lane_on ? sb.c4 : 0;  // unconditional loads from clamped addresses; `commit` zeroes the rest
        if (idx) {
            int src[B];
#pragma unroll
            for (int b = 0; b < B; ++b) src[b] = idx[min(r0 + b * sb.rpi + sb.sub, rmax)];
#pragma unroll
            for (int b = 0; b < B; ++b) sb.v[b] = *reinterpret_cast<const float4*>(base + (int64_t)src[b] * dim + 4 * c4c);
        } else {
#pragma unroll
            for (int b = 0; b < B; ++b)
                sb.v[b] = *reinterpret_cast<const float4*>(base + (int64_t)min(r0 + b * sb.rpi + sb.sub, rmax) * dim + 4 * c4c);
        }
    } else {
        gcp_load_segment(base, idx, dim, r0, rows, tile, stride, coff, lane);
    }
}

template <int B>
__device__ __forceinline__ void gcp_seg_commit(const GcpSegBuf<B>& sb, float* tile, int stride, int coff) {
    if (!sb.deferred) return;
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const int e = b * sb.rpi + sb.sub;
        if (sb.lane_on && e < GCP_TILE_ROWS) {
            const bool ok = e < sb.nvalid;
            float* d = tile + e * stride + coff + 4 * sb.c4;
            d[0] = ok ? sb.v[b].x : 0.f; d[1] = ok ? sb.v[b].y : 0.f; d[2] = ok ? sb.v[b].z : 0.f; d[3] = ok ? sb.v[b].w : 0.f;
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
        float* out = dst + col0 + 4 * c4;
        if (rpi >= 2) {  // fixed trip count: all LDS reads of the tile are issued before the stores need them
            float4 v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int e = min(sub + i * rpi, GCP_TILE_ROWS - 1);
                const float* s = tile + e * stride + 4 * c4;
                v[i] = make_float4(s[0], s[1], s[2], s[3]);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int e = sub + i * rpi;
                if (sub < rpi && e < GCP_TILE_ROWS && r0 + e < rows) *reinterpret_cast<float4*>(out + (int64_t)(r0 + e) * ld) = v[i];
            }
        } else {
#pragma unroll 8
            for (int e = 0; e < GCP_TILE_ROWS; ++e) {
                const float* s = tile + e * stride + 4 * c4;
                if (r0 + e < rows) *reinterpret_cast<float4*>(out + (int64_t)(r0 + e) * ld) = make_float4(s[0], s[1], s[2], s[3]);
            }
        }
        return;
    }
    for (int e = 0; e < GCP_TILE_ROWS && r0 + e < rows; ++e)
        for (int j = lane; j < width; j += GCP_WAVE) dst[(int64_t)(r0 + e) * ld + col0 + j] = tile[e * stride + j];
}

// Frames of rows [r0, r0 + 32): one contiguous 1152-byte block, all five loads of a lane in flight together.
__device__ __forceinline__ void gcp_load_frames(const float* __restrict__ frames, int r0, int rows, float* fr, int lane) {
    const int n = min(GCP_TILE_ROWS, rows - r0) * 9;
    const float* src = frames + (int64_t)r0 * 9;
    float t[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) t[k] = src[min(lane + 64 * k, n - 1)];  // unconditional, clamped
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int i = lane + 64 * k;
        if (i < GCP_TILE_ROWS * 9) fr[i] = i < n ? t[k] : 0.f;
    }
}

// The small vector weights of a GCP2 block (vector_down [H, vi], vector_down_frames [3, vi], vector_up [vo, H]) are
// copied once into a wave-private LDS area; the per-row loops then read them with wave-uniform (broadcast) ds_reads
// instead of one global load per multiply.
struct GcpSmallW {
    const float* wd;
    const float* wf;
    const float* wu;
};

__host__ __device__ inline int gcp_small_w_floats(int vi, int H, int vo, int nf) {
    return vi > 0 ? (H * vi + (nf ? 3 * vi : 0) + vo * H + 4) : 0;
}

__device__ __forceinline__ void gcp_copy_to_lds(const float* __restrict__ src, float* dst, int n, int lane) {
    for (int i0 = 0; i0 < n; i0 += 64 * 4) {
        float t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            t[k] = src[min(i0 + lane + 64 * k, n - 1)];  // unconditional, clamped
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + lane + 64 * k;
            if (i < n) dst[i] = t[k];
        }
    }
}

// Small weights above this many floats are not copied: the loops read them from global memory (L2-resident, wave-uniform
// addresses) -- slower per access, but the LDS budget of wide-vector shapes (V_in = 68, H = 68: 28 KB) goes to the tiles.
#define GCP_SMALL_W_LDS_MAX 2048
__host__ __device__ inline int gcp_small_w_lds_floats(int vi, int H, int vo, int nf) {
    const int n = gcp_small_w_floats(vi, H, vo, nf);
    return n <= GCP_SMALL_W_LDS_MAX ? n : 0;
}

__device__ __forceinline__ GcpSmallW gcp_stage_small_weights(const gcp2_weights_t& w, int H, int nf, float* area, int lane) {
    GcpSmallW r;
    if (gcp_small_w_floats(w.vi, H, w.vo, nf) > GCP_SMALL_W_LDS_MAX) {
        r.wd = w.w_down; r.wf = w.w_frames; r.wu = w.w_up;
        return r;
    }
    r.wd = area;
    r.wf = area + H * w.vi;
    r.wu = r.wf + (nf ? 3 * w.vi : 0);
    if (w.vi > 0) {
        gcp_copy_to_lds(w.w_down, area, H * w.vi, lane);
        if (nf) gcp_copy_to_lds(w.w_frames, area + H * w.vi, 3 * w.vi, lane);
        if (w.vo > 0) gcp_copy_to_lds(w.w_up, area + H * w.vi + (nf ? 3 * w.vi : 0), w.vo * H, lane);
    }
    return r;
}

// 4 consecutive columns j0..j0+3 of row `row` of a [rows, ld] matrix (zeros outside): the accumulator-layout access.
__device__ __forceinline__ float4 gcp_load4(const float* base, int64_t row, int ld, int j0, bool ok, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec) {  // wave-uniform.  The load itself is unconditional (clamped address) and the result is selected: a guarded
                // load would make hipcc branch around it and wait vmcnt(0) for each one
        const bool in = ok && j0 + 3 < ld;
        const float4 t = *reinterpret_cast<const float4*>(base + (ok ? row : 0) * ld + min(j0, ld - 4));
        v.x = in ? t.x : 0.f; v.y = in ? t.y : 0.f; v.z = in ? t.z : 0.f; v.w = in ? t.w : 0.f;
        return v;
    }
    if (!ok) return v;
    const float* p = base + row * ld + j0;
    if (j0 + 0 < ld) v.x = p[0];
    if (j0 + 1 < ld) v.y = p[1];
    if (j0 + 2 < ld) v.z = p[2];
    if (j0 + 3 < ld) v.w = p[3];
    return v;
}

// Batched forms of gcp_load4 for the accumulator layout.  gcp_load4 with a RUN-TIME `vec` compiles to a branch around each
// load whose two sides merge right behind it, so hipcc waits (vmcnt(0)) for every load where it is issued: a loop of N calls
// costs N memory round trips.  Here the wave-uniform test is made once and all the loads of a tile (or of all NT tiles) sit in
// one basic block, requested together.
//   v[q] = base[row, c0 + 8 q + 4 hi .. + 3], q < 4: the 16 columns of one 32-wide tile this lane holds.
__device__ __forceinline__ void gcp_load_tile4(const float* __restrict__ base, int64_t row, int ld, int c0, int hi, bool ok, bool vec,
                                               float4 (&v)[4]) {
    if (vec) {
        const float* rp = base + (ok ? row : 0) * (int64_t)ld;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4*>(rp + min(c0 + 8 * q + 4 * hi, ld - 4));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool in = ok && c0 + 8 * q + 4 * hi + 3 < ld;
            v[q] = make_float4(in ? v[q].x : 0.f, in ? v[q].y : 0.f, in ? v[q].z : 0.f, in ? v[q].w : 0.f);
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = gcp_load4(base, row, ld, c0 + 8 * q + 4 * hi, ok, false);
    }
}
//   x[t][r] (+)= base[row, col0 + 32 t + crow(r, hi)] for all NT tiles of an accumulator-layout register set.
template <int NT, bool ADD>
__device__ __forceinline__ void gcp_load_acc_layout(const float* __restrict__ base, int64_t row, int ld, int col0, int hi, bool ok,
                                                    bool vec, f32x16 (&x)[NT]) {
    float4 v[NT][4];
    if (vec) {  // raw (clamped, unconditional) requests only; the out-of-range selects come after all of them
        const float* rp = base + (ok ? row : 0) * (int64_t)ld;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[t][q] = *reinterpret_cast<const float4*>(rp + min(col0 + 32 * t + 8 * q + 4 * hi, ld - 4));
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[t][q] = gcp_load4(base, row, ld, col0 + 32 * t + 8 * q + 4 * hi, ok, false);
    }
    __builtin_amdgcn_sched_barrier(0);  // all requests first: hipcc otherwise sinks each load to its use through one temporary
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool in = !vec || (ok && col0 + 32 * t + 8 * q + 4 * hi + 3 < ld);
            const float4 w = make_float4(in ? v[t][q].x : 0.f, in ? v[t][q].y : 0.f, in ? v[t][q].z : 0.f, in ? v[t][q].w : 0.f);
            if constexpr (ADD) {
                x[t][4 * q] += w.x; x[t][4 * q + 1] += w.y; x[t][4 * q + 2] += w.z; x[t][4 * q + 3] += w.w;
            } else {
                x[t][4 * q] = w.x; x[t][4 * q + 1] = w.y; x[t][4 * q + 2] = w.z; x[t][4 * q + 3] = w.w;
            }
        }
}

// Split form for loads that should stay in flight under other work: gcp_request_acc_layout issues the raw (clamped,
// unconditional) requests straight into the destination registers -- nothing consumes them, so neither the wave-uniform
// `vec` branch nor anything else makes hipcc wait -- and gcp_mask_acc_layout, called where the values are first needed,
// zeroes what was out of range.
template <int NT>
__device__ __forceinline__ void gcp_request_acc_layout(const float* __restrict__ base, int64_t row, int ld, int col0, int hi, bool ok,
                                                       bool vec, f32x16 (&x)[NT]) {
    if (vec) {
        const float* rp = base + (ok ? row : 0) * (int64_t)ld;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(rp + min(col0 + 32 * t + 8 * q + 4 * hi, ld - 4));
                x[t][4 * q] = v.x; x[t][4 * q + 1] = v.y; x[t][4 * q + 2] = v.z; x[t][4 * q + 3] = v.w;
            }
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = gcp_load4(base, row, ld, col0 + 32 * t + 8 * q + 4 * hi, ok, false);
                x[t][4 * q] = v.x; x[t][4 * q + 1] = v.y; x[t][4 * q + 2] = v.z; x[t][4 * q + 3] = v.w;
            }
    }
}
template <int NT>
__device__ __forceinline__ void gcp_mask_acc_layout(int ld, int col0, int hi, bool ok, bool vec, f32x16 (&x)[NT]) {
    if (!vec) return;  // (the scalar path has masked already)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool in = ok && col0 + 32 * t + 8 * q + 4 * hi + 3 < ld;
#pragma unroll
            for (int i = 0; i < 4; ++i) x[t][4 * q + i] = in ? x[t][4 * q + i] : 0.f;
        }
}

__device__ __forceinline__ void gcp_store4(float* base, int64_t row, int ld, int j0, float4 v, bool ok, bool vec) {
    if (!ok) return;
    float* p = base + row * ld + j0;
    if (vec && j0 + 3 < ld) { *reinterpret_cast<float4*>(p) = v; return; }
    if (j0 + 0 < ld) p[0] = v.x;
    if (j0 + 1 < ld) p[1] = v.y;
    if (j0 + 2 < ld) p[2] = v.z;
    if (j0 + 3 < ld) p[3] = v.w;
}

// Stores NT 32x32 accumulator tiles (C/D layout: register r of lane (row = lane & 31, half) holds column crow(r, half)) to
// dst[(r0 + row) * ld + col0 + 32 t + column], through a wave-private 32 x 36 LDS staging tile, one 32-column tile at a time.
// Why not store the registers directly (16 bytes per lane at column 8 q + 4 half): each such instruction writes 32-byte
// pieces of 32 different rows, i.e. a quarter of a 128-byte line per request, and HBM write throughput drops to ~2.4 TB/s
// (tools/ubench/tile_access.hip); after the LDS transposition every instruction writes eight full 128-byte lines (4.5 TB/s).
#define GCP_ACC_STAGE_FLOATS (32 * 36)
template <int NT>
__device__ __forceinline__ void gcp_store_acc_rows(float* __restrict__ dst, int ld, int col0, int width, int r0, int rows,
                                                   const f32x16 (&acc)[NT], float* stage, int lane) {
    const int e = lane & 31, hi = lane >> 5;
    const int sub = lane >> 3, c4 = 4 * (lane & 7);
    const bool vec = (ld & 3) == 0 && (col0 & 3) == 0 && gcp_aligned16(dst);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous tile's reads are done
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(stage + e * 36 + 8 * q + 4 * hi) =
                make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float4 w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const float4*>(stage + (8 * j + sub) * 36 + c4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = r0 + 8 * j + sub, c = 32 * t + c4;
            if (row < rows) {
                float* p = dst + (int64_t)row * ld + col0 + c;
                if (vec && c + 3 < width) {
                    *reinterpret_cast<float4*>(p) = w[j];
                } else {
                    if (c + 0 < width) p[0] = w[j].x;
                    if (c + 1 < width) p[1] = w[j].y;
                    if (c + 2 < width) p[2] = w[j].z;
                    if (c + 3 < width) p[3] = w[j].w;
                }
            }
        }
    }
}

// gcp_store_acc_rows for a destination the caller knows to be 16-byte aligned with ld % 4 == 0: one wave-uniform test for a
// tile without rows or columns past the end (then: no branch per store), the general form otherwise.
template <int NT>
__device__ __forceinline__ void gcp_store_acc_rows_dense(float* __restrict__ dst, int ld, int width, int r0, int rows,
                                                         const f32x16 (&acc)[NT], float* stage, int lane) {
    if (!(r0 + 32 <= rows && width == 32 * NT)) {  // (wave-uniform)
        gcp_store_acc_rows<NT>(dst, ld, 0, width, r0, rows, acc, stage, lane);
        return;
    }
    const int e = lane & 31, hi = lane >> 5;
    const int sub = lane >> 3, c4 = 4 * (lane & 7);
    float* p0 = dst + (int64_t)(r0 + sub) * ld + c4;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous tile's reads are done
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(stage + e * 36 + 8 * q + 4 * hi) =
                make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float4 w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const float4*>(stage + (8 * j + sub) * 36 + c4);
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(p0 + (int64_t)8 * j * ld + 32 * t) = w[j];
    }
}

// gcp_store_acc_rows_half for a destination the caller knows to be 16-byte aligned with ld % 4 == 0 and width == ld == 32 NT:
// no per-piece column tests, and one wave-uniform test for a tile without rows past the end instead of a branch per store (the
// general form compiles to ~8 branches per 32-column tile; inside an unrolled kernel body every one of them is a scheduling fence).
// PAD: ld (a multiple of 4) may be less than 32 NT -- the 16-byte pieces past column ld are dropped (one lane test per piece).
template <int NT, bool PAD = false>
__device__ __forceinline__ void gcp_store_acc_rows_half_dense(float* __restrict__ dst, int ld, int r0, int rows, const f32x16 (&acc)[NT],
                                                              float* stage, int lane, float scale = 1.f) {  // (scale: per lane = per row)
    const int e = lane & 31, hi = lane >> 5;
    const int sub = lane >> 2, c4 = 4 * (lane & 3);
    const bool full = r0 + 32 <= rows;  // wave-uniform
    float* p0 = dst + (int64_t)(r0 + sub) * ld + c4;
    float* p1 = p0 + (int64_t)16 * ld;
    const bool ok0 = r0 + sub < rows, ok1 = r0 + 16 + sub < rows;
    auto pieces = [&](auto full_tag) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous piece's reads are done
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    *reinterpret_cast<float4*>(stage + e * 20 + 8 * q + 4 * hi) =
                        make_float4(acc[t][8 * h + 4 * q] * scale, acc[t][8 * h + 4 * q + 1] * scale, acc[t][8 * h + 4 * q + 2] * scale,
                                    acc[t][8 * h + 4 * q + 3] * scale);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const float4 w0 = *reinterpret_cast<const float4*>(stage + sub * 20 + c4);
                const float4 w1 = *reinterpret_cast<const float4*>(stage + (16 + sub) * 20 + c4);
                const int c = 32 * t + 16 * h;
                const bool in = !PAD || c + c4 + 3 < ld;
                if ((decltype(full_tag)::value || ok0) && in) *reinterpret_cast<float4*>(p0 + c) = w0;
                if ((decltype(full_tag)::value || ok1) && in) *reinterpret_cast<float4*>(p1 + c) = w1;
            }
    };
    if (full) pieces(std::true_type{});
    else pieces(std::false_type{});
}

// Same through a 32 x 20 staging tile (2.5 KB): half tiles at a time, 64-byte pieces per row, 16 rows per store instruction
// (4.1 TB/s in tools/ubench/tile_access.hip) -- for kernels whose occupancy is bound by LDS.
#define GCP_ACC_STAGE_HALF_FLOATS (32 * 20)
template <int NT>
__device__ __forceinline__ void gcp_store_acc_rows_half(float* __restrict__ dst, int ld, int col0, int width, int r0, int rows,
                                                        const f32x16 (&acc)[NT], float* stage, int lane) {
    const int e = lane & 31, hi = lane >> 5;
    const int sub = lane >> 2, c4 = 4 * (lane & 3);
    const bool vec = (ld & 3) == 0 && (col0 & 3) == 0 && gcp_aligned16(dst);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous piece's reads are done
#pragma unroll
            for (int q = 0; q < 2; ++q)
                *reinterpret_cast<float4*>(stage + e * 20 + 8 * q + 4 * hi) =
                    make_float4(acc[t][8 * h + 4 * q], acc[t][8 * h + 4 * q + 1], acc[t][8 * h + 4 * q + 2], acc[t][8 * h + 4 * q + 3]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float4 w[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) w[j] = *reinterpret_cast<const float4*>(stage + (16 * j + sub) * 20 + c4);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = r0 + 16 * j + sub, c = 32 * t + 16 * h + c4;
                if (row < rows) {
                    float* p = dst + (int64_t)row * ld + col0 + c;
                    if (vec && c + 3 < width) {
                        *reinterpret_cast<float4*>(p) = w[j];
                    } else {
                        if (c + 0 < width) p[0] = w[j].x;
                        if (c + 1 < width) p[1] = w[j].y;
                        if (c + 2 < width) p[2] = w[j].z;
                        if (c + 3 < width) p[3] = w[j].w;
                    }
                }
            }
        }
}
