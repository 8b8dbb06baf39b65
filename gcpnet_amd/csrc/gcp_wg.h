// Multi-wave workgroup GCP2 kernels (gcp_wg_fwd.hip / gcp_wg_bwd.hip): shared shapes, packed-weight layout and helpers.
//
// One WORKGROUP of NW wavefronts owns a 32-row tile (edges or nodes).  Every dense Linear of the block is computed
// transposed, out^T[cols, 32 rows] = W[cols, K] . in^T[K, 32 rows], with the OUTPUT columns split across the waves: wave w
// holds the 32-column output tiles ot = w, w + NW, ...; the reduction operand (the tile's merged input [s | norms | frame
// scalars]) sits ONCE in LDS as X[32][KS] and every wave reads its B fragments from there with one ds_read_b128 per four
// k-pair steps; the A fragments (weights) stream from L2 in a fragment-ordered image, 16 bytes per lane per four steps.
// A wave therefore carries 16 accumulator registers per output tile and almost nothing else, so 3-4 waves share a SIMD and
// the VALU / LDS / store phases of one workgroup run under the MFMAs of the others.
//
// k-pair steps: step 4 g + i of group g pairs the columns (8 g + i, 8 g + 4 + i) for the lane halves (hi = 0, 1): a lane
// (row e, half hi) reads X[e][8 g + 4 hi .. + 3] -- which is also the layout of an accumulator register quad
// (column 8 q + 4 hi + i of row e), so accumulators feed the next GEMM without any data movement.
#pragma once
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "gcp_bf16x3.h"
#include "gcp_f16x2.h"

// GCPNET_DEBUG_UNSUPPORTED=1: the host entry points of the workgroup kernels say on stderr WHY a launch was refused (the caller
// then takes the wave-per-tile kernels: a routing question one otherwise answers by reading profiles)
inline int wg_unsupported(const char* file, int line, const char* why) {
    static const bool on = getenv("GCPNET_DEBUG_UNSUPPORTED") != nullptr;
    if (on) fprintf(stderr, "gcpnet: %s:%d refuses the launch: %s\n", file, line, why);
    return GCPNET_E_UNSUPPORTED;
}
#define WG_UNSUPPORTED(why) wg_unsupported(__FILE__, __LINE__, why)

// (GCP_WG_MAX_BLOCKS = 9: head + 8 residual blocks, include/gcpnet_hip.h)

// Packed image of one block's dense weights (floats; all sections are [..][64 lanes][4]):
//   A1 [NT][KG]   forward scalar_out:       W[32 ot + m][8 g + 4 hi + i]
//   G1 [GM][4 NT] forward gate Linear:      Wg[32 gm + m][8 g + 4 hi + i]         (gm < GM = ceil(vo / 32) tiles of gate outputs,
//                                                                                reduction over so)
//   A2 [NKT][4 NT] backward-data scalar_out: W[8 g + 4 hi + i][32 kt + m]          (reduction over so, output tile kt of K)
//   G2 [NT][VG]   backward gate Linear:     Wg[8 g + 4 hi + i][32 ot + m]         (reduction over vo)
// with m = lane & 31, hi = lane >> 5, zeros outside the matrices.
struct WgShape {
    int si, vi, so, vo, H, nf;
    int K, KG, KP;   // merged width, groups of 8 columns, KG * 8
    int NT;          // 32-wide tiles of so
    int NKT;         // 32-wide tiles of K
    int VG;          // groups of 8 gate outputs
    int GM;          // 32-row tiles of gate outputs (forward gate Linear): 1 or 2 (vo <= 64)
    int gated;
    // A2b [NKT][2 NT slabs][term][64][4 x 2 bf16]: section A2 once more as three bf16 terms per weight in the operand layout
    // of v_mfma_f32_32x32x16_bf16 (gcp_bf16x3.h); element i' < 8 of lane (m, hi) in slab j is W[32 (j / 2) + 16 (j % 2) +
    // 8 (i' / 4) + 4 hi + i' % 4][32 kt + m] -- the column set of eight consecutive accumulator registers
    // A1b [NT][NSLf = ceil(KG / 2) slabs][term][64][4]: section A1 as three bf16 terms; element i' of lane (m, hi) in slab j is
    // W[32 ot + m][16 j + 8 hi + i'] (natural column order: the forward's operand planes are split from the fp32 tile X)
    int64_t offA1, offG1, offA2, offG2, offA2b, offA1b, total;
    int NSLf;
};

__host__ __device__ inline WgShape wg_shape(int si, int vi, int so, int vo, int H, int use_frames, int gated) {
    WgShape s;
    s.si = si; s.vi = vi; s.so = so; s.vo = vo; s.H = vi > 0 ? H : 0;
    s.nf = (vi > 0 && use_frames) ? 9 : 0;
    s.K = si + s.H + s.nf;
    s.KG = gcp_cdiv(s.K, 8);
    s.KP = s.KG * 8;
    s.NT = gcp_cdiv(so, 32);
    s.NKT = gcp_cdiv(s.K, 32);
    s.VG = gcp_cdiv(vo, 8);
    s.GM = vo > 32 ? gcp_cdiv(vo, 32) : 1;
    s.gated = (gated && vo > 0 && vi > 0) ? 1 : 0;
    s.offA1 = 0;
    s.offG1 = s.offA1 + (int64_t)s.NT * s.KG * 256;
    s.offA2 = s.offG1 + (s.gated ? (int64_t)s.GM * 4 * s.NT * 256 : 0);
    s.offG2 = s.offA2 + (int64_t)s.NKT * 4 * s.NT * 256;
    s.offA2b = s.offG2 + (s.gated ? (int64_t)s.NT * s.VG * 256 : 0);
    // (GCP_W6_TERMS terms per element: two fp16 terms by default -- gcp_f16x2.h --, three bf16 terms with -DGCP_ARITH_F16X2=0)
    s.offA1b = s.offA2b + (int64_t)s.NKT * 2 * s.NT * GCP_W6_TERMS * 256;
    s.NSLf = gcp_cdiv(s.KG, 2);
    s.total = s.offA1b + (int64_t)s.NT * s.NSLf * GCP_W6_TERMS * 256;
    return s;
}

// LDS row stride (floats) of a [32][width] tile read with ds_read_b128 by lanes (row e, half hi): a multiple of 4 (16-byte
// alignment) that is an ODD multiple of 4, so that the 16 lanes of one LDS access group (distinct e mod 16) hit distinct
// 4-bank groups of the 64 banks.
__host__ __device__ inline int wg_stride(int width) {
    int q = gcp_cdiv(width, 4);
    return 4 * (q | 1);
}

// Workgroup barrier for LDS hand-offs: waits for this wave's DS operations only (__syncthreads() would also drain vmcnt, i.e.
// every outstanding global store and prefetched load of the wave).
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ bool wg_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// i / d for 0 <= i < 2^16, 1 < d < 2^16 without the ~35-instruction run-time division: magic = ceil(2^32 / d) from the host
// (wg_magic); magic == 0 (d == 1, or "not given") falls back to the division.
__host__ __device__ inline unsigned wg_magic(int d) { return d > 1 ? (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d) : 0u; }
__device__ __forceinline__ int wg_div(int i, int d, unsigned magic) { return magic ? (int)__umulhi((unsigned)i, magic) : i / d; }

// tile[r * stride + c] = src[r * width + c] for r < 32, c < width (rows of a 32-row tile are contiguous in memory: a flat,
// coalesced copy); rows >= nvalid read as zero.  Loads are unconditional (clamped) and batched four deep per thread.
// `vec`: width % 4 == 0, stride % 4 == 0 and src 16-byte aligned -> 16-byte accesses.
template <int NTH>
__device__ __forceinline__ void wg_tile_load(float* tile, int stride, const float* __restrict__ src, int width, int nvalid, int tid,
                                             bool vec) {
    if (vec) {
        const int q = width >> 2, n4 = nvalid * q, tot = 32 * q;
        for (int i0 = 0; i0 < tot; i0 += 4 * NTH) {
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f32x4*>(src + 4 * (int64_t)min(i0 + tid + k * NTH, n4 - 1));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + tid + k * NTH;
                if (i < tot) {
                    const int r = i / q, c4 = i - r * q;
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4*>(tile + r * stride + 4 * c4) = i < n4 ? v[k] : z;
                }
            }
        }
    } else {
        const int n1 = nvalid * width, tot = 32 * width;
        for (int i0 = 0; i0 < tot; i0 += 4 * NTH) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = src[min(i0 + tid + k * NTH, n1 - 1)];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + tid + k * NTH;
                if (i < tot) {
                    const int r = i / width, c = i - r * width;
                    tile[r * stride + c] = i < n1 ? v[k] : 0.f;
                }
            }
        }
    }
}

// Split form of wg_tile_load for tiles of up to B * NTH 16-byte pieces: `request` puts every load of the tile in flight (no
// branches around them: clamped addresses), `commit` writes them to LDS.  Several tiles requested back to back share ONE
// memory round trip instead of paying one each.
template <int B>
struct WgTileReq {
    f32x4 v[B];
    bool ok;  // the tile fits this form (16-byte pieces, B per thread); otherwise commit() falls back to wg_tile_load
};
template <int NTH, int B>
__device__ __forceinline__ void wg_tile_request(WgTileReq<B>& rq, const float* __restrict__ src, int width, int nvalid, int tid, bool vec) {
    const int q = width >> 2, n4 = nvalid * q;
    rq.ok = vec && q > 0 && 32 * q <= B * NTH;
#pragma unroll
    for (int k = 0; k < B; ++k) rq.v[k] = *reinterpret_cast<const f32x4*>(src + 4 * (int64_t)max(min(tid + k * NTH, n4 - 1), 0));
}
template <int NTH, int B>
__device__ __forceinline__ void wg_tile_commit(const WgTileReq<B>& rq, float* tile, int stride, const float* __restrict__ src, int width,
                                               int nvalid, int tid, bool vec, unsigned mq = 0) {
    if (!rq.ok) {
        wg_tile_load<NTH>(tile, stride, src, width, nvalid, tid, vec);
        return;
    }
    const int q = width >> 2, n4 = nvalid * q, tot = 32 * q;
#pragma unroll
    for (int k = 0; k < B; ++k) {
        const int i = tid + k * NTH;
        if (i < tot) {
            const int r = wg_div(i, q, mq), c4 = i - r * q;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(tile + r * stride + 4 * c4) = i < n4 ? rq.v[k] : z;
        }
    }
}

// dst[r * width + c] = tile[r * stride + c] for r < nvalid (flat, coalesced).
template <int NTH>
__device__ __forceinline__ void wg_tile_store(float* __restrict__ dst, const float* tile, int stride, int width, int nvalid, int tid,
                                              bool vec, unsigned mq = 0) {
    if (vec) {
        const int q = width >> 2, n4 = nvalid * q;
        for (int i = tid; i < n4; i += NTH) {
            const int r = wg_div(i, q, mq), c4 = i - r * q;
            *reinterpret_cast<f32x4*>(dst + 4 * (int64_t)i) = *reinterpret_cast<const f32x4*>(tile + r * stride + 4 * c4);
        }
    } else {
        const int n1 = nvalid * width;
        for (int i = tid; i < n1; i += NTH) {
            const int r = i / width, c = i - r * width;
            dst[i] = tile[r * stride + c];
        }
    }
}

// One 32 x 32 accumulator tile (C/D layout) -> dst[(r0 + row) * ld + c0 + col] as full 128-byte row pieces, through a
// wave-private 32 x 36 LDS staging tile (why: tile_io.h, gcp_store_acc_rows).  ld % 4 == 0, c0 % 32 == 0, dst 16-byte aligned.
#define WG_STAGE_FLOATS (32 * 36)
__device__ __forceinline__ void wg_store_acc(float* __restrict__ dst, int ld, int c0, int width, int r0, int nvalid, const f32x16& a,
                                             float* st, int lane) {
    const int e = lane & 31, hi = lane >> 5;
    const int sub = lane >> 3, c4 = 4 * (lane & 7);
    gcp_wave_lds_sync();  // (the staging tile's previous reads are done)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = {a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]};
        *reinterpret_cast<f32x4*>(st + e * 36 + 8 * q + 4 * hi) = v;
    }
    gcp_wave_lds_sync();
    f32x4 wv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wv[j] = *reinterpret_cast<const f32x4*>(st + (8 * j + sub) * 36 + c4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 8 * j + sub, c = c0 + c4;
        if (r < nvalid && c < width) *reinterpret_cast<f32x4*>(dst + (int64_t)(r0 + r) * ld + c) = wv[j];
    }
}
