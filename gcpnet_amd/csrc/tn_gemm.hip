// Weight-gradient GEMMs on gfx950: out[m, n] = sum_r A[r, m] * B[r, n], a "TN" GEMM whose reduction axis (rows =
// edges or nodes, 1e4..1e6) is huge and whose output (a weight matrix, <= 1024 x 1100) is small.
//
// The reference gets these from autograd through nn.Linear (src/models/components/gcpnet.py:303-324); here they are explicit.
// The row axis is split across workgroups (an even number of splits, 16- or 32-row chunks dealt round-robin); per-split partial
// sums go to scratch and a second kernel reduces them in a fixed order (deterministic; no atomics).  Products run on the bf16
// matrix pipe as three-term splits with six products kept (gcp_bf16x3.h: fp32 round-off).  Three kernels, in the order the host
// function tries them:
//   * tn_pipe_kernel (round 5; the long comment in front of it): rows HBM -> registers -> ONE split per operand element -> planes
//     in LDS -> products; 128 x 160 (three workgroups per CU) and 256 x 288 (every operand row read once) instantiations.  Takes
//     every problem whose operands are plain or tile-blocked segments (+ a column of ones for the bias gradient);
//   * tn_gemm_dma_kernel (rounds 2 - 4): four waves, 128 x 160 block, rows DMA'd into LDS 32 at a time (global_load_lds_dwordx4,
//     double buffered), every wave splits what it reads.  Serves operands with ROW GATHERS ([h_row | e | h_col | ...] is never
//     materialised: each 16-byte piece's source address is computed per lane) or an ACTIVATION on load, and, as MODE 0, plain
//     fp32-MFMA arithmetic for the tests (GCPNET_TN_FP32);
//   * tn_gemm_kernel: register-staged, any widths / strides / alignments (what the DMA cannot address).
#include <mutex>
#include <cstdlib>
#include <cstdio>
#include <type_traits>

#include "common.h"
#include "gcp_bf16x3.h"
#include "gcp_f16x2.h"

namespace {

// -DGCP_TN_TIMING: every wave sums s_memtime deltas of the four phases of its chunk loop (0 = wait + barrier, 1 = DMA issue + index
// requests, 2 = products, 3 = rest) and writes them OVER its split's partial sums (element 4 wave + phase of row 0): after the
// reduction out[0][4 w + k] is the sum over the splits, tools/tn_phase_timing.py prints it.  A measurement build, never shipped.
#ifdef GCP_TN_TIMING
#define TN_T_DECL unsigned long long tn_t_[4] = {0, 0, 0, 0}, tn_t0_ = __builtin_amdgcn_s_memtime()
#define TN_T_MARK(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tn_t_[k] += t_ - tn_t0_; tn_t0_ = t_; } while (0)
#define TN_T_STORE(part, wave, lane) do { if ((lane) == 0) for (int k_ = 0; k_ < 4; ++k_) (part)[4 * (wave) + k_] = (float)tn_t_[k_]; } while (0)
#else
#define TN_T_DECL
#define TN_T_MARK(k)
#define TN_T_STORE(part, wave, lane)
#endif

#ifndef GCP_TN_RK
#define GCP_TN_RK 32  // rows per staged chunk (16 or 32: -DGCP_TN_RK=16 halves the staging buffers; measured, see the header of x3_step)
#endif
constexpr int TN_BM = 128, TN_BN = 160, TN_RK = GCP_TN_RK;
// Rows per split: 128 splits for the big (edge-row) problems, and never fewer than 64 rows (node-row problems).  Alone a launch is
// fastest with 256 - 512 splits (every CU busy); inside the training step, where these launches share the chip with the caller's
// stream, HALF of that wins -- fewer resident workgroups, half the partial sums: configs[1] 11.64 -> 11.37 ms, configs[4] 213.8 ->
// 212.9 ms, c4 under hipGraph replay 4.37 -> 4.30 ms, configs[3] equal (64 and 96 splits: no further gain;
// profiles/r03_tn_bf16x3.txt).
constexpr int TN_TARGET_SPLITS = 128, TN_MIN_ROWS_PER_SPLIT = 64;
// host side: rows per split for the target split count (GCPNET_TN_SPLITS overrides the 128: a tuning knob)
inline int tn_rows_per_split_host(int rows) {
    static const int target = getenv("GCPNET_TN_SPLITS") && atoi(getenv("GCPNET_TN_SPLITS")) > 0 ? atoi(getenv("GCPNET_TN_SPLITS")) : TN_TARGET_SPLITS;
    const int r = gcp_round_up(gcp_cdiv(rows > 0 ? rows : 1, target), TN_RK);
    return r < TN_MIN_ROWS_PER_SPLIT ? TN_MIN_ROWS_PER_SPLIT : r;
}
constexpr int TN_LDA = TN_BM + 1, TN_LDB = TN_BN + 1;  // generic path: padded strides
constexpr int TN_DMA_LDS_FLOATS = 2 * TN_RK * (TN_BM + TN_BN);

struct TnArgs {
    int n;
    gcp_tn_problem_t p[GCP_TN_MAX_PROBLEMS];
    int M[GCP_TN_MAX_PROBLEMS], N[GCP_TN_MAX_PROBLEMS];
    int mb[GCP_TN_MAX_PROBLEMS], nb[GCP_TN_MAX_PROBLEMS];
    int block_start[GCP_TN_MAX_PROBLEMS + 1];
    int debug;   // measurement knob GCPNET_TN_DEBUG: bit 0 = no products, bit 1 = no DMA after the first chunk (results are then wrong)
    unsigned long long* stamps;  // profiling hook (gcpnet_debug_set_phase_timing): per workgroup of the pipelined kernels start / end time, HW_ID, XCC_ID
    long long stamp_cap;
};

__host__ __device__ inline int operand_width(const gcp_operand_t& o) {
    int w = o.ones ? 1 : 0;
    for (int k = 0; k < o.n; ++k) w += o.dim[k];
    return w;
}

// Tile-blocked segments (include/gcpnet_hip.h, gcp2_chain_item_t): float offset of the 16-byte piece at column c (a multiple of 4)
// of row 0 inside a tile, and the offset of row r's copy of that piece: (r / 32) tiles of 32 wp floats + (r % 32) pieces of 4.
__host__ __device__ inline int tb_col_offset(int c) { return (((c >> 5) * 4 + ((c & 31) >> 3)) * 64 + ((c >> 2) & 1) * 32) * 4; }
__host__ __device__ inline int64_t tb_row_offset(int64_t r, int wp) { return (r >> 5) * (int64_t)(32 * wp) + (r & 31) * 4; }

struct BlockWork {
    int pi, split, m0, n0, mw, nw, ntiles;
    int r_first, r_step, r_end, nchunks;  // chunk c covers rows r_first + c * r_step .. + 31, cut at r_end
};

// Rows of a split: 32-row chunks dealt round-robin, so that the workgroups of a launch, which advance in lockstep, read ONE contiguous
// window of the operands at any time (splits x 32 rows: every HBM channel busy) instead of `splits` windows a fixed stride apart
// (which camp on a few channels; the contiguous-range form of round 3 is gone).
__device__ __forceinline__ void split_rows(int rows, int splits, int split, int& r_first, int& r_step, int& r_end, int& nchunks) {
    const int total = gcp_cdiv(rows, TN_RK);
    r_first = split * TN_RK; r_step = splits * TN_RK; r_end = rows;
    nchunks = total > split ? gcp_cdiv(total - split, splits) : 0;
}

__device__ __forceinline__ BlockWork locate(const TnArgs& a) {
    BlockWork w;
    int pi = 0;
    while (pi + 1 < a.n && (int)blockIdx.x >= a.block_start[pi + 1]) ++pi;
    w.pi = pi;
    const gcp_tn_problem_t& P = a.p[pi];
    int b = blockIdx.x - a.block_start[pi];
    w.split = b % P.splits; b /= P.splits;
    const int nbi = b % a.nb[pi], mbi = b / a.nb[pi];
    w.m0 = mbi * TN_BM; w.n0 = nbi * TN_BN;
    w.mw = min(TN_BM, a.M[pi] - w.m0);
    w.nw = min(TN_BN, a.N[pi] - w.n0);
    w.ntiles = gcp_cdiv(w.nw, 32);
    split_rows(P.rows, P.splits, w.split, w.r_first, w.r_step, w.r_end, w.nchunks);
    return w;
}

// Output tiles (32x32) of a block are dealt round-robin to the four waves: tile id = wave + 4 i, i < 5, with
// (m-tile, n-tile) = (id / ntiles, id % ntiles).  This keeps all waves busy for skinny problems (M = 16 or N = 16).
__device__ __forceinline__ void store_partial(const gcp_tn_problem_t& P, const BlockWork& w, int M, int N, const f32x16* acc,
                                              int wave, int col, int hi) {
    float* part = P.partial + (int64_t)w.split * M * N;
    const int mtiles = gcp_cdiv(w.mw, 32), total = mtiles * w.ntiles;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int id = wave + 4 * i;
        if (id < total) {
            const int mi = id / w.ntiles, ni = id - mi * w.ntiles;
            const int n = w.n0 + 32 * ni + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = w.m0 + 32 * mi + gcp_crow(r, hi);
                if (m < M && n < N) part[(int64_t)m * N + n] = acc[i][r];
            }
        }
    }
}

// ---- generic path: register-staged, any widths / strides ---------------------------------------------------------
__device__ __forceinline__ void stage_operand(const gcp_operand_t& op, int c0, int width, int r0, int rows, float* S,
                                              int LD, int wave, int lane) {
    for (int rr = wave; rr < TN_RK; rr += 4) {
        const int r = r0 + rr;
        const bool valid = r < rows;
        float* dst = S + rr * LD;
        int cbase = 0;
        for (int sg = 0; sg < op.n; ++sg) {
            const int dim = op.dim[sg];
            const int lo = max(c0, cbase), hi = min(c0 + width, cbase + dim);
            if (lo < hi) {
                if (valid) {
                    const int64_t src = op.idx[sg] ? (int64_t)op.idx[sg][r] : (int64_t)r;
                    if (op.tb[sg]) {
                        const float* tp = op.ptr[sg] + tb_row_offset(src, gcp_round_up(dim, 32));
                        for (int c = lo + lane; c < hi; c += GCP_WAVE) dst[c - c0] = gcp_act(op.act, tp[tb_col_offset((c - cbase) & ~3) + ((c - cbase) & 3)], op.slope);
                    } else {
                    const float* rowp = op.ptr[sg] + src * op.ld[sg] - cbase;
                    for (int c = lo + lane; c < hi; c += GCP_WAVE) dst[c - c0] = gcp_act(op.act, rowp[c], op.slope);
                    }
                } else {
                    for (int c = lo + lane; c < hi; c += GCP_WAVE) dst[c - c0] = 0.f;
                }
            }
            cbase += dim;
        }
        if (op.ones) {
            if (cbase >= c0 && cbase < c0 + width && lane == 0) dst[cbase - c0] = valid ? 1.f : 0.f;
            cbase += 1;
        }
        for (int c = max(cbase, c0) + lane; c < c0 + width; c += GCP_WAVE) dst[c - c0] = 0.f;
    }
}

__global__ __launch_bounds__(256) void tn_gemm_kernel(TnArgs a) {
    __shared__ float As[TN_RK * TN_LDA];
    __shared__ float Bs[TN_RK * TN_LDB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int col = lane & 31, hi = lane >> 5;
    const BlockWork w = locate(a);
    const gcp_tn_problem_t& P = a.p[w.pi];
    const int total_tiles = gcp_cdiv(w.mw, 32) * w.ntiles;
    const int my_tiles = __builtin_amdgcn_readfirstlane(wave < total_tiles ? (total_tiles - wave + 3) / 4 : 0);  // wave-uniform
    const bool wave_active = my_tiles > 0;
    int aoff[5], boff[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int id = min(wave + 4 * i, total_tiles - 1), mi = id / w.ntiles;
        aoff[i] = 32 * mi; boff[i] = 32 * (id - mi * w.ntiles);
    }
    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int c = 0; c < w.nchunks; ++c) {
        const int r0 = w.r_first + c * w.r_step;
        stage_operand(P.a, w.m0, gcp_round_up(w.mw, 32), r0, w.r_end, As, TN_LDA, wave, lane);
        stage_operand(P.b, w.n0, w.ntiles * 32, r0, w.r_end, Bs, TN_LDB, wave, lane);
        __syncthreads();
        if (wave_active) {
#pragma unroll 4
            for (int ss = 0; ss < TN_RK / 2; ++ss) {
                const float* arow = As + (2 * ss + hi) * TN_LDA + col;
                const float* brow = Bs + (2 * ss + hi) * TN_LDB + col;
#pragma unroll
                for (int i = 0; i < 5; ++i)
                    if (i < my_tiles) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[aoff[i]], brow[boff[i]], acc[i], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (wave_active) store_partial(P, w, a.M[w.pi], a.N[w.pi], acc, wave, col, hi);
}

// ---- fast path: HBM -> LDS DMA, double buffered -------------------------------------------------------------------
struct Slot {
    const float* base;   // segment pointer + column offset of this 16-byte piece
    const int32_t* idx;  // optional row gather
    int ld;
    int row;             // row inside the 32-row chunk
    bool on;             // piece belongs to a real column group of the operand
};  // (ld < 0: a tile-blocked segment of padded width -ld; base then includes the piece's column offset inside a tile)

template <int NSLOT, int LD, int NTH = 256>
__device__ __forceinline__ void make_slots(const gcp_operand_t& op, int c0, Slot* s, int tid) {
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
        const int f = tid + NTH * k;
        const bool piece = f < TN_RK * LD / 4;  // (a thread count that does not divide the pieces: the slots past the end stay off)
        const int row = piece ? f / (LD / 4) : 0, c = c0 + 4 * (f % (LD / 4));
        s[k].row = row; s[k].on = false; s[k].base = nullptr; s[k].idx = nullptr; s[k].ld = 0;
        int cbase = 0;
        for (int sg = 0; sg < op.n; ++sg) {
            if (piece && c >= cbase && c < cbase + op.dim[sg]) {
                s[k].base = op.ptr[sg] + (op.tb[sg] ? tb_col_offset(c - cbase) : (c - cbase));
                s[k].idx = op.idx[sg];
                s[k].ld = op.tb[sg] ? -gcp_round_up(op.dim[sg], 32) : op.ld[sg];
                s[k].on = true;
            }
            cbase += op.dim[sg];
        }
    }
}

// Source rows of the pieces of one chunk.  Gather indices are ordinary global loads: they are requested one chunk
// ahead, so that no load result is consumed between DMA issues (hipcc drains vmcnt(0) at such a use, which would
// serialise the DMAs).
template <int NSLOT>
__device__ __forceinline__ void fetch_issue(const Slot* s, int* v, int r0, int r_last, const int32_t* any_idx) {
    if (any_idx) {  // wave-uniform: the operand has at least one gathered segment
#pragma unroll
        for (int k = 0; k < NSLOT; ++k) {  // unconditional loads (a per-piece branch would make hipcc wait per piece)
            const int32_t* ip = s[k].idx ? s[k].idx : any_idx;
            v[k] = ip[min(r0 + s[k].row, r_last)];
        }
    }
}

template <int NSLOT>
__device__ __forceinline__ void fetch_finish(const Slot* s, const int* v, int64_t* src, int r0, int r_last,
                                             const int32_t* any_idx) {
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
        const int r = min(r0 + s[k].row, r_last);
        src[k] = (any_idx && s[k].idx) ? (int64_t)v[k] : (int64_t)r;
    }
}

template <int NSLOT, int NTH = 256>
__device__ __forceinline__ void issue_dma(const Slot* s, const int64_t* src, float* buf, int tid) {
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
        float* dst = buf + (NTH * k + (tid & ~63)) * 4;  // wave-uniform LDS base; the hardware adds lane * 16 bytes
        if (s[k].on)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(s[k].base + (s[k].ld < 0 ? tb_row_offset(src[k], -s[k].ld) : src[k] * s[k].ld)),
                (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

__device__ __forceinline__ const int32_t* any_gather(const gcp_operand_t& op) {
    const int32_t* g = nullptr;
    for (int k = 0; k < op.n; ++k)
        if (op.idx[k]) g = op.idx[k];
    return g;
}

// activation on the landed tile: real data columns of valid rows only (never the ones column or padding)
__device__ __forceinline__ void act_in_lds(float* buf, int LD, int data_cols, int nvalid, int act, float slope, int tid, int nth = 256) {
    for (int i = tid; i < nvalid * data_cols; i += nth) {
        const int r = i / data_cols, c = i - r * data_cols;
        buf[r * LD + c] = gcp_act(act, buf[r * LD + c], slope);
    }
}

// One 16-row step of the three-term bf16 form for a wave's MT x (up to five) tiles.  As / Bs point at the wave's first row (8 hi)
// and column of the step; a fragment is eight rows of one column.  Per n-tile: split its fragment (44 VALU instructions), six
// products per m-tile; the rows of the next tile are requested before the products of this one.
// What was measured on the way (tools/ubench/x3_overlap.hip, profiles/r03_x3_overlap_ubench.txt; tools/tn_phase_timing.py): the
// split costs ~1.6 cycles an instruction and hides under the MFMAs wherever it is placed; a v_mfma_f32_32x32x16_bf16 stream of
// this shape runs at ~45 cycles a product (not 32) whether two or ten accumulators rotate; placing the split units, or the next
// chunk's DMA pieces, BETWEEN the products by hand (sched_barrier) bought nothing, the DMA pieces there cost more than in a phase
// of their own, and a (m half, row half) wave layout with 2 x 5 tiles per wave was slower in the step than this form.  16-row
// chunks (-DGCP_TN_RK=16: half the staging LDS, up to four workgroups per CU): 0.093 against 0.082 ms alone, 11.97 against 11.74 ms
// per configs[1] step, 213.7 against 210.0 ms per configs[4] step -- 32 rows stay.
template <int MT, int LDA, int LDB>
__device__ __forceinline__ void x3_step(const float* As, const float* Bs, const int (&boff)[5], int nt, f32x16 (&acc)[MT][5]) {
    gcp_u32x4 a3[MT][3];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        float x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = As[q * LDA + 32 * j];
        gcp_bf16x3_split8(x, a3[j][0], a3[j][1], a3[j][2]);
    }
    float y[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) y[q] = Bs[q * LDB + boff[0]];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        float yn[8];
        if (i + 1 < 5) {  // the next tile's rows before this tile's products (offsets of absent tiles are clamped to valid ones)
#pragma unroll
            for (int q = 0; q < 8; ++q) yn[q] = Bs[q * LDB + boff[i + 1 < 5 ? i + 1 : 4]];
        }
        if (i < nt) {  // (wave-uniform)
            gcp_u32x4 bh, bm, bl;
            gcp_bf16x3_split8(y, bh, bm, bl);
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[j][i] = gcp_mfma_bf16x6(a3[j], bh, bm, bl, acc[j][i]);
        }
        if (i + 1 < 5) {
#pragma unroll
            for (int q = 0; q < 8; ++q) y[q] = yn[q];
        }
    }
}

// X3: the products run on the bf16 pipe as three-term splits (gcp_bf16x3.h: six v_mfma_f32_32x32x16_bf16 per 16-row step, fp32
// round-off).  A lane's fragment is then eight ROWS of one column (the reduction axis is the row axis, and a sum does not care in
// which order the rows sit in the k slots: lane half `hi` takes rows 8 hi .. 8 hi + 7 of the step for both operands), read with
// eight ds_read_b32 and split in registers.  To split every A fragment once, a wave owns ONE m-tile and up to five n-tiles:
// waves are laid out mt_c (1, 2 or 4) along m and G = 4 / mt_c along n, n-tiles g, g + G, ... going to group g.
struct X3Tiles {
    int mi, g, G, my_n;
};
__device__ __forceinline__ X3Tiles x3_tiles(const BlockWork& w, int wave) {
    X3Tiles t;
    const int mtiles = gcp_cdiv(w.mw, 32);
    const int mt_c = mtiles > 2 ? 4 : mtiles;
    t.G = 4 / mt_c;
    t.mi = wave % mt_c;
    t.g = wave / mt_c;
    t.my_n = (t.mi < mtiles && t.g < w.ntiles) ? (w.ntiles - t.g + t.G - 1) / t.G : 0;
    return t;
}
__device__ __forceinline__ void store_partial_x3(const gcp_tn_problem_t& P, const BlockWork& w, int M, int N, const f32x16* acc,
                                                 const X3Tiles& t, int col, int hi) {
    float* part = P.partial + (int64_t)w.split * M * N;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        if (i < t.my_n) {
            const int n = w.n0 + 32 * (t.g + t.G * i) + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = w.m0 + 32 * t.mi + gcp_crow(r, hi);
                if (m < M && n < N) part[(int64_t)m * N + n] = acc[i][r];
            }
        }
    }
}

// MODE 0: fp32 MFMA, tiles dealt round-robin, four waves.  MODE 1: three-term bf16, one m-tile per wave (X3Tiles), four waves.
// (Since round 5 this kernel serves the operands the pipelined form does not take -- row gathers, activations on load -- and, as
// MODE 0, the tests that hold the bf16 forms against plain fp32 MFMA arithmetic.  The eight-wave, "planes" and big-block variants
// of rounds 3 - 4 are gone: each was faster alone and slower inside the step; profiles/r03_tn_bf16x3.txt, r04_tn_planes_experiment.txt.)
template <int MODE>
__global__ __launch_bounds__(256) void tn_gemm_dma_kernel(TnArgs a) {
    constexpr bool X3 = MODE >= 1;
    constexpr int MT = 1;
    constexpr int NTH = 256;
    constexpr int A_SLOTS = (TN_RK * TN_BM / 4 + NTH - 1) / NTH, B_SLOTS = (TN_RK * TN_BN / 4 + NTH - 1) / NTH;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int col = lane & 31, hi = lane >> 5;
    const BlockWork w = locate(a);
    const gcp_tn_problem_t& P = a.p[w.pi];
    const int total_tiles = gcp_cdiv(w.mw, 32) * w.ntiles;
    X3Tiles xt = x3_tiles(w, wave);
    xt.my_n = __builtin_amdgcn_readfirstlane(xt.my_n);
    const int my_tiles = X3 ? xt.my_n : __builtin_amdgcn_readfirstlane(wave < total_tiles ? (total_tiles - wave + 3) / 4 : 0);  // wave-uniform
    const bool wave_active = my_tiles > 0;
    int aoff[5], boff[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        if (X3) {
            aoff[i] = 32 * xt.mi; boff[i] = 32 * min(xt.g + xt.G * i, w.ntiles - 1);
        } else {
            const int id = min(wave + 4 * i, total_tiles - 1), mi = id / w.ntiles;
            aoff[i] = 32 * mi; boff[i] = 32 * (id - mi * w.ntiles);
        }
    }
    auto Abuf = [&](int b) { return lds + b * (TN_RK * (TN_BM + TN_BN)); };
    auto Bbuf = [&](int b) { return lds + b * (TN_RK * (TN_BM + TN_BN)) + TN_RK * TN_BM; };

    Slot sa[A_SLOTS], sb[B_SLOTS];
    make_slots<A_SLOTS, TN_BM, NTH>(P.a, w.m0, sa, tid);
    make_slots<B_SLOTS, TN_BN, NTH>(P.b, w.n0, sb, tid);
    const int32_t* ga = any_gather(P.a);
    const int32_t* gb = any_gather(P.b);
    // the ones column (bias gradients) is written by hand; its position inside the tile, or -1
    const int a_data = operand_width(P.a) - (P.a.ones ? 1 : 0), b_data = operand_width(P.b) - (P.b.ones ? 1 : 0);
    const int a_ones = P.a.ones ? (a_data - w.m0) : -1, b_ones = P.b.ones ? (b_data - w.n0) : -1;
    const bool a_has_ones = a_ones >= 0 && a_ones < TN_BM, b_has_ones = b_ones >= 0 && b_ones < TN_BN;
    const int a_cols = max(0, min(TN_BM, a_data - w.m0)), b_cols = max(0, min(TN_BN, b_data - w.n0));

    for (int i = tid; i < TN_DMA_LDS_FLOATS; i += NTH) lds[i] = 0.f;  // columns no DMA piece covers stay zero
    __syncthreads();

    f32x16 acc[MT][5];
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;

    const int r_last = w.r_end - 1;
    const int nchunks = w.nchunks;
    auto row0 = [&](int chunk) { return w.r_first + chunk * w.r_step; };
    int64_t ra[A_SLOTS], rb[B_SLOTS];
    auto stage = [&](int chunk, int b) {  // rows of `chunk` are in ra / rb
        const int r0 = row0(chunk);
        issue_dma<A_SLOTS, NTH>(sa, ra, Abuf(b), tid);
        issue_dma<B_SLOTS, NTH>(sb, rb, Bbuf(b), tid);
        if (tid < TN_RK) {
            const float one = (r0 + tid <= r_last) ? 1.f : 0.f;
            if (a_has_ones) Abuf(b)[tid * TN_BM + a_ones] = one;
            if (b_has_ones) Bbuf(b)[tid * TN_BN + b_ones] = one;
        }
    };
    int va[A_SLOTS], vb[B_SLOTS];
    fetch_issue<A_SLOTS>(sa, va, row0(0), r_last, ga);
    fetch_issue<B_SLOTS>(sb, vb, row0(0), r_last, gb);
    fetch_finish<A_SLOTS>(sa, va, ra, row0(0), r_last, ga);
    fetch_finish<B_SLOTS>(sb, vb, rb, row0(0), r_last, gb);
    // every source row resolved BEFORE the first piece goes out: hipcc otherwise sinks each select into the (conditional) issue
    // of its piece and drains vmcnt(0) -- the previous piece -- in front of every one
#pragma unroll
    for (int k = 0; k < A_SLOTS; ++k) asm volatile("" : "+v"(ra[k]));
#pragma unroll
    for (int k = 0; k < B_SLOTS; ++k) asm volatile("" : "+v"(rb[k]));
    stage(0, 0);
    fetch_issue<A_SLOTS>(sa, va, row0(1), r_last, ga);
    fetch_issue<B_SLOTS>(sb, vb, row0(1), r_last, gb);
    fetch_finish<A_SLOTS>(sa, va, ra, row0(1), r_last, ga);
    fetch_finish<B_SLOTS>(sb, vb, rb, row0(1), r_last, gb);
    TN_T_DECL;
    for (int c = 0; c < nchunks; ++c) {
        const int cur = c & 1;
        TN_T_MARK(3);
        __syncthreads();  // vmcnt(0) + barrier: this chunk has landed, and every wave is done with the other buffer
        TN_T_MARK(0);
        const int nvalid = min(TN_RK, w.r_end - row0(c));
        if (nvalid < TN_RK) {  // last chunk of the split: rows past the end were clamped duplicates, zero them
            for (int i = tid; i < (TN_RK - nvalid) * TN_BM; i += NTH) Abuf(cur)[nvalid * TN_BM + i] = 0.f;
            for (int i = tid; i < (TN_RK - nvalid) * TN_BN; i += NTH) Bbuf(cur)[nvalid * TN_BN + i] = 0.f;
            __syncthreads();
        }
        if (P.a.act || P.b.act) {
            if (P.a.act) act_in_lds(Abuf(cur), TN_BM, a_cols, nvalid, P.a.act, P.a.slope, tid, NTH);
            if (P.b.act) act_in_lds(Bbuf(cur), TN_BN, b_cols, nvalid, P.b.act, P.b.slope, tid, NTH);
            __syncthreads();
        }
        const int rn = row0(c + 2);  // gather indices of the chunk after next: requested now,
        if (c + 1 < nchunks) {                        // consumed after this chunk's MFMAs
            if (!(a.debug & 2)) stage(c + 1, cur ^ 1);
            fetch_issue<A_SLOTS>(sa, va, rn, r_last, ga);
            fetch_issue<B_SLOTS>(sb, vb, rn, r_last, gb);
        }
        TN_T_MARK(1);
        if (X3 && wave_active && !(a.debug & 1)) {
            const float* As = Abuf(cur) + col + 8 * hi * TN_BM + aoff[0];
            const float* Bs = Bbuf(cur) + col + 8 * hi * TN_BN;
#pragma unroll 1
            for (int ks = 0; ks < TN_RK / 16; ++ks) x3_step<MT, TN_BM, TN_BN>(As + 16 * ks * TN_BM, Bs + 16 * ks * TN_BN, boff, my_tiles, acc);
        }
        if (!X3 && wave_active) {
            // fragments of step ss + 1 are read (unconditionally: the offsets are clamped to valid tiles) before the MFMAs of
            // step ss, so that an MFMA never waits for its own ds_read
            const float* As = Abuf(cur) + col + hi * TN_BM;
            const float* Bs = Bbuf(cur) + col + hi * TN_BN;
            float fa[5], fb[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) { fa[i] = As[aoff[i]]; fb[i] = Bs[boff[i]]; }
#pragma unroll
            for (int ss = 0; ss < TN_RK / 2; ++ss) {
                float na[5], nb[5];
                if (ss + 1 < TN_RK / 2) {
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        na[i] = As[2 * (ss + 1) * TN_BM + aoff[i]];
                        nb[i] = Bs[2 * (ss + 1) * TN_BN + boff[i]];
                    }
                }
#pragma unroll
                for (int i = 0; i < 5; ++i)
                    if (i < my_tiles) acc[0][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[i], acc[0][i], 0, 0, 0);
                if (ss + 1 < TN_RK / 2) {
#pragma unroll
                    for (int i = 0; i < 5; ++i) { fa[i] = na[i]; fb[i] = nb[i]; }
                }
            }
        }
        TN_T_MARK(2);
        if (c + 1 < nchunks) {
            fetch_finish<A_SLOTS>(sa, va, ra, rn, r_last, ga);
            fetch_finish<B_SLOTS>(sb, vb, rb, rn, r_last, gb);
        }
    }
    if (X3) {
        if (wave_active) store_partial_x3(P, w, a.M[w.pi], a.N[w.pi], acc[0], xt, col, hi);
        TN_T_MARK(3);
        TN_T_STORE(P.partial + (int64_t)w.split * a.M[w.pi] * a.N[w.pi], wave, lane);
    } else if (wave_active) {
        store_partial(P, w, a.M[w.pi], a.N[w.pi], acc[0], wave, col, hi);
    }
}

// ---- pipelined form (round 5): rows HBM -> registers -> ONE split per operand element -> planes in LDS -> products ---------------
// What bounds a weight-gradient GEMM is its operand stream, not the matrix pipe: an output block of 128 x 160 needs 213 bf16 FLOP
// per operand byte with all six products of the three-term form, i.e. 11.7 TB/s at the bf16 peak -- twice what HBM delivers; a
// 256 x 288 block is balanced.  So the kernel to build keeps the operand rows streaming and hides everything else under them.
// The forms above do neither: every wave re-splits every B fragment it reads (8.8 VALU instructions per MFMA: the products phase
// runs at 2.4 x its MFMA time) and every chunk is waited for with vmcnt(0) one chunk after it was requested.  Here
//   * a thread requests one FRAGMENT-LANE per pass -- eight rows of one column, the k-octet a lane of v_mfma_f32_32x32x16_bf16
//     holds -- straight into registers, two chunks ahead of its use; hipcc counts these waits itself, exactly, because no load
//     sits behind a branch (tp_load) -- there is no staging buffer, no DMA and no hand-counted vmcnt;
//   * the lane is split into its three bf16 terms ONCE per workgroup and stored as three 16-byte entries of operand-ordered
//     planes (double buffered), which every wave reads with ds_read_b128: one barrier per 16-row chunk;
//   * a wave owns MT m-tiles x the n-tiles of its group and walks them UT at a time, the next B fragments requested from LDS
//     before the products of the current ones: >= 2 independent accumulators per product round, no dependent MFMA chain;
//   * passes (split + request) and product groups alternate inside a chunk, so that the matrix pipe is fed while the vector
//     ALU splits.
// Two instantiations: 4 waves x (1 x 5) tiles = 128 x 160, 54 KB of LDS, two workgroups per CU; and 4 waves x (2 x 9) tiles =
// 256 x 288 with one wave per SIMD (~400 registers), 102 KB: the (256,32) message GCPs' 256 x 277 gradients read every operand
// row ONCE.  Row gathers and activations on load stay with the kernels above (stream_ok).
constexpr int TS_RK = 16;

// Arithmetic of the pipelined kernels: 1 (default, with GCP_ARITH_F16X2) = two fp16 terms per operand element and three MFMAs per
// product block (gcp_f16x2.h) instead of three bf16 terms and six.  The summed index here is the ROW, so the power-of-two scale of an
// operand must be constant down a column WHILE products are added with it: every 32-column fragment carries a CURRENT exponent, owned
// by the wave that loads it -- set from the first non-zero chunk's largest magnitude (with GCP_TN_HEAD bits of headroom), lowered when a
// later chunk would overflow fp16, raised again when the data falls more than 2^GCP_TN_WINDOW below it --, published with the chunk's
// planes; a wave whose accumulators were scaled with another exponent multiplies them by the (exact) power of two in between, in
// either direction.  The partial sums leave multiplied by 2^-(ea + eb).  Elements far below the maximum of their 16-row x 32-column
// chunk lose relative precision exactly as in gcp_f16x2.h; the result stays within 3 * 2^-22 of sum |a b| + 2^-38 x (chunk maxima).
#ifndef GCP_TN_F16X2
#define GCP_TN_F16X2 GCP_ARITH_F16X2
#endif
#define GCP_TN_HEAD 2
#define GCP_TN_WINDOW 8
constexpr int TP_UNSET = 1000;  // exponent of a fragment that has only seen zeros

// largest value of the wave (v >= 0), wave-uniform
__device__ __forceinline__ float tp_wave_max(float v) {
#define TP_DPP_MAX(ctrl, rmask)                                                                                                   \
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, rmask, 0xf, false)))
    TP_DPP_MAX(0x111, 0xf);  // row_shr:1
    TP_DPP_MAX(0x112, 0xf);  // row_shr:2
    TP_DPP_MAX(0x114, 0xf);  // row_shr:4
    TP_DPP_MAX(0x118, 0xf);  // row_shr:8   -> lane 15 of every row of 16: the row's maximum
    TP_DPP_MAX(0x142, 0xa);  // row_bcast:15 -> rows 1, 3
    TP_DPP_MAX(0x143, 0xc);  // row_bcast:31 -> rows 2, 3: lane 63 holds the wave's maximum
#undef TP_DPP_MAX
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

struct TpLane {
    const float* base;  // column of this lane in row 0 of its segment (tile-blocked: its piece + element); a valid address for every kind
    int ld;             // floats per row; tile-blocked: padded width
    int kind;           // 0 = data, 1 = the ones column, 2 = padding; | 4 = tile-blocked segment
};

__device__ __forceinline__ TpLane tp_lane(const gcp_operand_t& op, int c) {  // c: column of the operand ([segments | ones | padding])
    TpLane L;
    L.base = op.ptr[0]; L.ld = 0; L.kind = 2;
    int cbase = 0;
    for (int sg = 0; sg < op.n; ++sg) {
        if (c >= cbase && c < cbase + op.dim[sg]) {
            const int k = c - cbase;
            L.base = op.ptr[sg] + (op.tb[sg] ? tb_col_offset(k & ~3) + (k & 3) : k);
            L.ld = op.tb[sg] ? gcp_round_up(op.dim[sg], 32) : op.ld[sg];
            L.kind = op.tb[sg] ? 4 : 0;
        }
        cbase += op.dim[sg];
    }
    if (op.ones && c == cbase) L.kind = 1;
    return L;
}
__device__ __forceinline__ int64_t tp_row_offset(const TpLane& L, int64_t r) { return (L.kind & 4) ? tb_row_offset(r, L.ld) : r * L.ld; }

// NBUF: plane buffers.  2: passes and product groups of a chunk interleave, one barrier per chunk.  1: products, barrier, passes,
// barrier -- half the LDS and fewer registers (no second set of B fragments): three workgroups per CU instead of two.
template <int NW, int MT, int NT, int UT, int NBUF = 2, bool F16 = GCP_TN_F16X2 && NW == 8>
struct TpCfg {
    static constexpr int NTH = 64 * NW, AT = MT * NW, BM = 32 * AT, BN = 32 * NT;
    static constexpr int PLANES = (AT + NT) * (F16 ? 2 : 3) * 256;  // floats per buffer: [fragment][term][64 lanes][4 dwords]
    static constexpr int NFR = AT + NT;                            // fragments (32-column tiles of A, then of B) per chunk
    // (two-term form: behind the planes, per buffer, the fragments' exponents of the chunk + the chunk index of the last change)
    static constexpr int LDS_FLOATS = NBUF * PLANES + (F16 ? NBUF * (NFR + 1) : 0);
    static constexpr int APASS = MT, BPASS = (NT + NW - 1) / NW, NPASS = APASS + BPASS;  // fragment-lanes per thread and chunk
    static constexpr int NG = (NT + UT - 1) / UT;                                          // product groups per chunk
};

template <int NW, int MT, int NT, int UT, int NBUF = 2, bool F16 = GCP_TN_F16X2 && NW == 8>
__global__ __launch_bounds__(64 * NW, NW == 5 ? 2 : (NBUF == 1 ? 3 : 2)) void tn_pipe_kernel(TnArgs a) {
    using C = TpCfg<NW, MT, NT, UT, NBUF, F16>;
    constexpr int AT = C::AT, BM = C::BM, BN = C::BN, NPASS = C::NPASS, APASS = C::APASS, NG = C::NG;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int col = lane & 31, hi = lane >> 5;
    int pi = 0;
    while (pi + 1 < a.n && (int)blockIdx.x >= a.block_start[pi + 1]) ++pi;
    const gcp_tn_problem_t& P = a.p[pi];
    const int M = a.M[pi], N = a.N[pi];
    int b = blockIdx.x - a.block_start[pi];
    const int split = b % P.splits; b /= P.splits;
    const int nbi = b % a.nb[pi], mbi = b / a.nb[pi];
    const int m0 = mbi * BM, n0 = nbi * BN;
    const int mtiles = gcp_cdiv(min(BM, M - m0), 32), ntiles = gcp_cdiv(min(BN, N - n0), 32);
    // rows: the FULL 16-row chunks dealt round-robin to the splits (the workgroups of a launch read one contiguous window of the
    // operands); the ragged last chunk, if any, goes to one split after its loop.  splits is even (gcpnet_tn_splits), so a split's
    // chunks keep their position inside the 32-row tiles of a tile-blocked segment and every lane advances by a constant.
    const int full_chunks = P.rows / TS_RK;
    const int nchunks = full_chunks > split ? gcp_cdiv(full_chunks - split, P.splits) : 0;
    const bool ragged = (P.rows % TS_RK) != 0 && (full_chunks % P.splits) == split;
    // tiles of this wave: m-groups of MT tiles, mt_c waves along m, G groups along n (n-tiles g, g + G, ...)
    const int mgroups = gcp_cdiv(mtiles, MT);
    // smallest power of two >= mgroups (<= NW); the five-wave form (160 x 160 blocks): one tile of M per wave
    const int mt_c = NW == 5 ? 5 : (mgroups > 4 ? 8 : (mgroups > 2 ? 4 : (mgroups > 1 ? 2 : 1)));
    const int G = NW / mt_c, mg = wave % mt_c, g = wave / mt_c;
    const int my_n = __builtin_amdgcn_readfirstlane((mg < mgroups && g < ntiles) ? (ntiles - g + G - 1) / G : 0);
    gcp_u32x4* const planes = reinterpret_cast<gcp_u32x4*>(lds);
    constexpr int PL4 = C::PLANES / 4;  // 16-byte entries per buffer
    constexpr int NTM = F16 ? 2 : 3;
    [[maybe_unused]] int* const pexp = reinterpret_cast<int*>(lds + NBUF * C::PLANES);  // [buf][NFR + 1]: exponents, then the change mark
    if (a.stamps && tid == 0 && (long long)blockIdx.x < a.stamp_cap) {
        unsigned long long* sp = a.stamps + (long long)blockIdx.x * GCP_MAX_STAMPS;
        sp[0] = __builtin_amdgcn_s_memtime();
        sp[2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
        sp[3] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    }

    // this thread's fragment-lanes: passes 0 .. MT - 1 = A tiles wave + NW j, the others B tiles wave + NW j (an absent tile re-reads
    // a valid one and is dropped)
    bool on[NPASS];
    int frag[NPASS];
    TpLane L[NPASS];
    const float* ptr[NPASS];  // rows of the chunk the next request of the pass goes to
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const bool isA = ps < APASS;
        const int t = wave + NW * (isA ? ps : ps - APASS);
        on[ps] = t < (isA ? mtiles : ntiles);
        frag[ps] = isA ? t : AT + t;
        L[ps] = isA ? tp_lane(P.a, m0 + 32 * min(t, mtiles - 1) + col) : tp_lane(P.b, n0 + 32 * min(t, ntiles - 1) + col);
        ptr[ps] = L[ps].base + tp_row_offset(L[ps], (int64_t)min(split, max(full_chunks - 1, 0)) * TS_RK + 8 * hi);
    }
    float x[NPASS][8];
    [[maybe_unused]] int pe[NPASS];  // (two-term form) running exponent of the pass's fragment: wave-uniform
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) pe[ps] = TP_UNSET;
    int issued = 0;  // chunks requested so far (all passes of a chunk advance together; the pointer stops at the split's last chunk)
    auto load_pass = [&](int ps) {
        // (the LAST pass is real for one wave only when the B tiles do not divide by the waves -- the ninth / fifth tile: the others
        // skip its requests.  A branch around loads makes hipcc's later counted waits assume the path without them; behind the last
        // pass that only lets the one wave with the extra requests wait a little too long)
        if (ps == NPASS - 1 && NPASS > 2 && !on[ps]) return;
        const int step = (L[ps].kind & 4) ? 4 : L[ps].ld;
#pragma unroll
        for (int q = 0; q < 8; ++q) x[ps][q] = ptr[ps][q * step];
    };
    auto advance = [&]() {  // after the last pass of a chunk's requests
        ++issued;
        if (issued < nchunks) {  // (wave-uniform; pointer arithmetic only)
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) ptr[ps] += (int64_t)(P.splits * TS_RK) * L[ps].ld;  // (tile-blocked: splits / 2 tiles of 32 ld floats)
        }
    };
    auto split_pass = [&](int ps, int buf, int nvalid, [[maybe_unused]] int chunk) {  // nvalid: valid rows of this lane's eight (8 inside the loop); chunk: index the planes belong to
        if (!on[ps]) return;  // (wave-uniform)
        const int kind = L[ps].kind & 3;
        if (__builtin_amdgcn_ballot_w64(kind != 0)) {  // (wave-uniform) a tile with the ones column or padding in it
#pragma unroll
            for (int q = 0; q < 8; ++q) x[ps][q] = kind == 0 ? x[ps][q] : (kind == 1 ? 1.f : 0.f);
        }
        if (nvalid < 8) {
#pragma unroll
            for (int q = 0; q < 8; ++q) x[ps][q] = q < nvalid ? x[ps][q] : 0.f;
        }
        if constexpr (F16) {
        float mx = fmaxf(fmaxf(fmaxf(fabsf(x[ps][0]), fabsf(x[ps][1])), fmaxf(fabsf(x[ps][2]), fabsf(x[ps][3]))),
                         fmaxf(fmaxf(fabsf(x[ps][4]), fabsf(x[ps][5])), fmaxf(fabsf(x[ps][6]), fabsf(x[ps][7]))));
        mx = tp_wave_max(mx);
        const int eb_ = (int)((__float_as_uint(mx) >> 23) & 0xffu);
        int* const ex = pexp + (buf & (NBUF - 1)) * (C::NFR + 1);
        if (eb_ != 0) {  // (wave-uniform; zero / subnormal chunks leave the exponent alone)
            const int pn = min(max(14 + 127 - eb_, -100), 100);  // brings the chunk's maximum into [2^14, 2^15)
            // the first data of the fragment, or more than the current exponent can hold (lower it, with headroom), or a chunk more than
            // 2^GCP_TN_WINDOW below what it was chosen for (raise it again: a spike in one row must not cost the rows behind it their
            // second term) -- either way the consumers rescale their accumulator tiles by an exact power of two
            if (pn < pe[ps] || pn > pe[ps] + GCP_TN_HEAD + GCP_TN_WINDOW) {
                pe[ps] = pn - GCP_TN_HEAD;
                if (lane == 0) ex[C::NFR] = chunk + 1;  // (several waves may write the same mark)
            }
        }
        const int pe_eff = pe[ps] == TP_UNSET ? 0 : pe[ps];
        if (lane == 0) ex[frag[ps]] = pe_eff;
        gcp_u32x4 th, tl3;
        gcp_f16x2_split8(x[ps], gcp_exp2i(pe_eff), th, tl3);
        gcp_u32x4* dst = planes + (buf & (NBUF - 1)) * PL4 + (frag[ps] * NTM) * 64 + lane;
        dst[0] = th; dst[64] = tl3;
        } else {
        gcp_u32x4 th, tm, tl3;
        gcp_bf16x3_split8(x[ps], th, tm, tl3);
        gcp_u32x4* dst = planes + (buf & (NBUF - 1)) * PL4 + (frag[ps] * 3) * 64 + lane;
        dst[0] = th; dst[64] = tm; dst[128] = tl3;
        }
    };
    // (two-term form) exponents the wave's accumulators carry: ea for its MT tiles of A, eb for its tiles g, g + G, ... of B
    [[maybe_unused]] int ea[MT], eb[NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) ea[m] = 0;
#pragma unroll
    for (int i = 0; i < NT; ++i) eb[i] = 0;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

    // the kept products, small terms first: three of two fp16 terms (lh, hl, hh: gcp_f16x2.h) or six of three bf16 terms
    constexpr int NPR = F16 ? 3 : 6;
    constexpr int TA[6] = {F16 ? 1 : 2, 0, F16 ? 0 : 1, 1, 0, 0}, TB[6] = {0, F16 ? 1 : 2, F16 ? 0 : 1, 0, 1, 0};
    auto read_b = [&](const gcp_u32x4* pl, int i0, gcp_u32x4 (&bt)[UT][NTM]) {
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            const gcp_u32x4* pb = pl + ((AT + min(g + G * min(i0 + u, NT - 1), ntiles - 1)) * NTM) * 64 + lane;
#pragma unroll
            for (int tm = 0; tm < NTM; ++tm) bt[u][tm] = pb[64 * tm];
        }
    };
    auto read_a = [&](const gcp_u32x4* pl, gcp_u32x4 (&a3)[MT][NTM]) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const gcp_u32x4* pa = pl + (min(mg * MT + m, mtiles - 1) * NTM) * 64 + lane;
#pragma unroll
            for (int tm = 0; tm < NTM; ++tm) a3[m][tm] = pa[64 * tm];
        }
    };
    auto mfma_t = [&](gcp_u32x4 av, gcp_u32x4 bv, f32x16 c) -> f32x16 {
        if constexpr (F16) {
        return gcp_mfma_f16(av, bv, c);
        } else {
        return gcp_mfma_bf16(av, bv, c);
        }
    };
    auto mul_group = [&](int k, const gcp_u32x4 (&a3)[MT][NTM], const gcp_u32x4 (&bt)[UT][NTM]) {
        if (k * UT < my_n) {  // (wave-uniform)
#pragma unroll
            for (int p = 0; p < NPR; ++p)
#pragma unroll
                for (int u = 0; u < UT; ++u)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        if (k * UT + u < NT) acc[m][k * UT + u] = mfma_t(a3[m][TA[p]], bt[u][TB[p]], acc[m][k * UT + u]);  // (tiles past my_n: computed on a valid fragment, never stored)
        }
    };
    // (two-term form) a chunk whose planes carry a change mark: the exponents of this wave's fragments again, and every accumulator
    // tile scaled with an older one multiplied by the exact power of two in between
    [[maybe_unused]] auto rescale = [&](int buf, int chunk) {
        if constexpr (F16) {
        const int* ex = pexp + (buf & (NBUF - 1)) * (C::NFR + 1);
        if (__builtin_amdgcn_readfirstlane(ex[C::NFR]) != chunk + 1) return;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int na = __builtin_amdgcn_readfirstlane(ex[min(mg * MT + m, mtiles - 1)]);
            if (na != ea[m]) {
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][i][r] = __builtin_ldexpf(acc[m][i][r], na - ea[m]);
                ea[m] = na;
            }
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int nb = __builtin_amdgcn_readfirstlane(ex[AT + min(g + G * i, ntiles - 1)]);
            if (nb != eb[i]) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][i][r] = __builtin_ldexpf(acc[m][i][r], nb - eb[i]);
                eb[i] = nb;
            }
        }
        }
    };

    // One tile with ONE set of B fragment registers: the products are ordered so that a term's registers retire early (bl after the
    // first product, bm after the third) and take the next tile's term at once -- the LDS latency of the next tile's fragments is
    // spent under this tile's remaining MFMAs without a second register set (UT == 1; the sums still run small terms first)
    auto mul_rot = [&](int k, const gcp_u32x4 (&a3)[MT][NTM], gcp_u32x4 (&bt)[NTM], const gcp_u32x4* pl, bool more) {
        if (k < my_n) {  // (wave-uniform)
            const gcp_u32x4* pn = pl + ((AT + min(g + G * min(k + 1, NT - 1), ntiles - 1)) * NTM) * 64 + lane;
            if constexpr (F16) {
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][k] = gcp_mfma_f16(a3[m][0], bt[1], acc[m][k]);
            if (more) { bt[1] = pn[64]; __builtin_amdgcn_sched_barrier(0); }  // (pinned, as below)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][k] = gcp_mfma_f16(a3[m][1], bt[0], acc[m][k]);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][k] = gcp_mfma_f16(a3[m][0], bt[0], acc[m][k]);
            if (more) bt[0] = pn[0];
            } else {
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][k] = gcp_mfma_bf16(a3[m][0], bt[2], acc[m][k]);
            if (more) { bt[2] = pn[128]; __builtin_amdgcn_sched_barrier(0); }  // (pinned: hipcc otherwise sinks the request behind the tile's last product)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][k] = gcp_mfma_bf16(a3[m][1], bt[1], acc[m][k]);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][k] = gcp_mfma_bf16(a3[m][0], bt[1], acc[m][k]);
            if (more) { bt[1] = pn[64]; __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][k] = gcp_mfma_bf16(a3[m][2], bt[0], acc[m][k]);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][k] = gcp_mfma_bf16(a3[m][1], bt[0], acc[m][k]);
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][k] = gcp_mfma_bf16(a3[m][0], bt[0], acc[m][k]);
            if (more) bt[0] = pn[0];
            }
        }
    };

    // prologue: chunk 0 requested, split into planes[0]; chunk 1 requested
    if constexpr (F16) {
    if (tid < NBUF) pexp[tid * (C::NFR + 1) + C::NFR] = 0;  // no change mark yet
    __syncthreads();
    }
    if (nchunks > 0) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) load_pass(ps);
        advance();
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            split_pass(ps, 0, 8, 0);
            load_pass(ps);
        }
        advance();
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    TN_T_DECL;
    for (int c = 0; c < nchunks; ++c) {
        const gcp_u32x4* pl = planes + (c & (NBUF - 1)) * PL4;
        constexpr bool DBUF = NBUF == 2 && NT * MT <= 5;  // two sets of B fragments only where the registers are there (the wide form holds 144 accumulators)
        gcp_u32x4 a3[MT][NTM], bt[DBUF ? 2 : 1][UT][NTM];
        rescale(c, c);
        read_a(pl, a3);
        read_b(pl, 0, bt[0]);
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            if (NBUF == 2) {
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {  // passes whose turn it is: pass ps runs ahead of product group (ps NG) / NPASS
                    if ((ps * NG) / NPASS == k) {
                        TN_T_MARK(3);
                        if (c + 1 < nchunks) split_pass(ps, (c + 1) & 1, 8, c + 1);
                        TN_T_MARK(0);
                        load_pass(ps);  // chunk c + 2 (unconditional -- past the end: the split's last chunk once more, dropped)
                        if (ps == NPASS - 1) advance();
                        TN_T_MARK(1);
                    }
                }
            }
            if (DBUF) {
                if (k + 1 < NG) read_b(pl, (k + 1) * UT, bt[(k + 1) & 1]);
                mul_group(k, a3, bt[DBUF ? (k & 1) : 0]);
            } else if (UT == 1) {
                mul_rot(k, a3, bt[0][0], pl, k + 1 < NG);
            } else {
                mul_group(k, a3, bt[0]);
                if (k + 1 < NG) read_b(pl, (k + 1) * UT, bt[0]);  // (behind the group's last MFMA issue; the SIMD's other waves cover the LDS latency)
            }
        }
        TN_T_MARK(3);
        if (NBUF == 1) {  // everyone is done with the planes: the next chunk goes into them
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                if (c + 1 < nchunks) split_pass(ps, 0, 8, c + 1);
                load_pass(ps);
            }
            advance();
            TN_T_MARK(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // planes of chunk c + 1 complete (NBUF = 2: and everyone is done with those of chunk c)
        TN_T_MARK(2);
    }
    if (ragged) {  // (workgroup-uniform) the operand's last, partial chunk: requested row by row with the row index clamped, not pipelined
        const int r0 = full_chunks * TS_RK + 8 * hi, r_last = P.rows - 1;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
            for (int q = 0; q < 8; ++q) x[ps][q] = L[ps].base[tp_row_offset(L[ps], min(r0 + q, r_last))];
            split_pass(ps, 0, P.rows - r0, nchunks);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        gcp_u32x4 a3[MT][NTM], bt[UT][NTM];
        rescale(0, nchunks);
        read_a(planes, a3);
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            read_b(planes, k * UT, bt);
            mul_group(k, a3, bt);
        }
    }
    float* part = P.partial + (int64_t)split * M * N;
    if (my_n > 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (mg * MT + m < mtiles) {
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    if (i < my_n) {
                        const int n = n0 + 32 * (g + G * i) + col;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int mm = m0 + 32 * (mg * MT + m) + gcp_crow(r, hi);
                            if constexpr (F16) {
                            if (mm < M && n < N) part[(int64_t)mm * N + n] = __builtin_ldexpf(acc[m][i][r], -(ea[m] + eb[i]));
                            } else {
                            if (mm < M && n < N) part[(int64_t)mm * N + n] = acc[m][i][r];
                            }
                        }
                    }
                }
            }
        }
    }
    TN_T_STORE(part, wave, lane);
    if (a.stamps && tid == 0 && (long long)blockIdx.x < a.stamp_cap) a.stamps[(long long)blockIdx.x * GCP_MAX_STAMPS + 1] = __builtin_amdgcn_s_memtime();
}

// GCPNET_TN_MID=1: the five-wave 160 x 160 form and, with it, both weight gradients of a gated (128,16) block as ONE product.
// OFF by default -- measured (same box, profiles/r06_tn_mid_ab.txt): the fused product reads 34 % fewer operand bytes, but the seven-
// block job list of a configs[1] layer takes 0.58 ms alone against 0.52 ms as two products per block (five waves on four SIMDs: one SIMD
// carries two waves' 30 MFMAs per chunk; 155 registers = two workgroups per CU), and the step does not move (10.02 - 10.08 ms against
// 10.03 - 10.04).  The kernel is bound by issue / latency at low occupancy, not by HBM bytes.
inline bool tn_mid_enabled() {  // (read per call, like GCPNET_TN_PIPE / GCPNET_TN_FP32: tests switch it inside one process)
    const char* e = getenv("GCPNET_TN_MID");
    return e && e[0] == '1';
}

inline bool stream_ok(const gcp_operand_t& o) {
    if (o.act) return false;
    for (int k = 0; k < o.n; ++k)
        if (o.idx[k]) return false;
    return true;
}

// Deterministic reduction of the per-split partial sums.  A workgroup handles 64 consecutive output elements; its four
// waves take the splits k = w, w+4, ... (coalesced 256-byte rows, eight loads in flight per lane) and are combined in a
// fixed order through LDS.
__global__ __launch_bounds__(256) void tn_reduce_kernel(TnArgs a) {
    __shared__ float red[4][64];
    const int pi = blockIdx.y;
    const gcp_tn_problem_t& P = a.p[pi];
    const int M = a.M[pi], N = a.N[pi];
    const int64_t full = (int64_t)M * N;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t base = (int64_t)blockIdx.x * 64; base < full; base += (int64_t)gridDim.x * 64) {
        const int64_t i = base + lane;
        const bool ok = i < full;
        const int m = ok ? (int)(i / N) : 0, n = ok ? (int)(i % N) : 0;
        const float* src = P.partial + (int64_t)m * N + n;
        float acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = 0.f;
        for (int k0 = w; k0 < P.splits; k0 += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + 4 * u;
                acc[u] += src[(int64_t)min(k, P.splits - 1) * full] * (k < P.splits ? 1.f : 0.f);
            }
        }
        red[w][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        __syncthreads();
        if (w == 0 && ok && m < P.out_m) {  // gradients leave in their final layouts: weight block, bias column; padding dropped
            const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
            if (P.m_split > 0 && m >= P.m_split) {  // (the second gradient of a fused product)
                if (n < P.out_n) P.out_b[(m - P.m_split) * P.out_b_sm + n * P.out_sn] = v;
                else if (P.out2_b && n == P.out2_n) P.out2_b[m - P.m_split] = v;
            } else if (n < P.out_n) P.out[m * P.out_sm + n * P.out_sn] = v;
            else if (P.out2 && n == P.out2_n) P.out2[m] = v;
        }
        __syncthreads();
    }
}

// Column sums of parts[n_parts, width] in two deterministic levels: groups of RP_GROUP parts, then the group sums.
constexpr int RP_GROUP = 64;
struct ReduceArgs {
    gcp_reduce_job_t j[GCP_REDUCE_MAX_JOBS];
};
template <bool SECOND>
__global__ __launch_bounds__(256) void reduce_partials_kernel(ReduceArgs a) {
    const gcp_reduce_job_t& J = a.j[blockIdx.y];
    const int width = J.width;
    const int groups = gcp_cdiv(J.n_parts, RP_GROUP);
    const float* __restrict__ in = SECOND ? J.tmp : J.parts;
    float* __restrict__ out = SECOND ? J.out : J.tmp;
    const int n_parts = SECOND ? groups : J.n_parts;
    const int group = SECOND ? groups : RP_GROUP;
    if (SECOND ? blockIdx.x > 0 : (int)blockIdx.x >= groups) return;
    const int p0 = blockIdx.x * group, p1 = min(n_parts, p0 + group);
    for (int c = threadIdx.x; c < width; c += 256) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int q = p0;
        // sixteen rows in flight (the additions keep the order of the four-row loop below: bit-identical sums) -- with four, a group
        // of 64 parts was sixteen dependent round trips per column: 162 us per launch for the 31 250 per-tile partials of a
        // (256,32) block on 10^6 rows
        for (; q + 15 < p1; q += 16) {
            float x[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) x[u] = in[(int64_t)(q + u) * width + c];
#pragma unroll
            for (int u = 0; u < 16; u += 4) { a0 += x[u]; a1 += x[u + 1]; a2 += x[u + 2]; a3 += x[u + 3]; }
        }
        for (; q + 3 < p1; q += 4) {
            a0 += in[(int64_t)q * width + c];
            a1 += in[(int64_t)(q + 1) * width + c];
            a2 += in[(int64_t)(q + 2) * width + c];
            a3 += in[(int64_t)(q + 3) * width + c];
        }
        for (; q < p1; ++q) a0 += in[(int64_t)q * width + c];
        out[(int64_t)blockIdx.x * width + c] = (a0 + a1) + (a2 + a3);
    }
}

inline bool dma_ok(const gcp_operand_t& o) {
    for (int k = 0; k < o.n; ++k)
        if ((o.dim[k] & 3) || (!o.tb[k] && (o.ld[k] & 3)) || (reinterpret_cast<uintptr_t>(o.ptr[k]) & 15)) return false;
    return true;
}

}  // namespace

// resident workgroups per CU of the two pipelined kernels as the runtime sees them (include/gcpnet_hip.h; tools/tn_occupancy.py)
extern "C" int gcpnet_debug_tn_occupancy(int wide) {
    int n = -1;
    const hipError_t err = wide ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, tn_pipe_kernel<8, 1, 9, 1>, 512, TpCfg<8, 1, 9, 1>::LDS_FLOATS * sizeof(float))
                                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, tn_pipe_kernel<4, 1, 5, 1, 1>, 256, TpCfg<4, 1, 5, 1, 1>::LDS_FLOATS * sizeof(float));
    return err == hipSuccess ? n : -(int)err;
}

extern "C" int gcpnet_tn_splits(int rows, int M, int N) {
    // The row-split count to launch a [rows, M]^T [rows, N] problem with (the caller sizes `partial` by it; gcpnet_tn_gemm takes any
    // count).  Even: a split's 16-row chunks then keep their position inside the 32-row tiles of a tile-blocked operand
    // (tn_pipe_kernel).  By rows only: fewer splits for a problem of several output blocks (fewer M x N partials to write and sum)
    // win when eight such problems share a launch (10^4 rows of (512,144) / (128,532): 0.188 -> 0.126 ms,
    // profiles/r05_tn_splits_sweep.txt) and lose in the step, where the feed-forward GCPs' launches carry one or two problems and
    // 32 splits x 4 blocks leave half the CUs without a workgroup (configs[4] 191.6 -> 192.7 ms, same box)
    (void)M; (void)N;
    return rows <= 0 ? 2 : gcp_round_up(gcp_cdiv(rows, tn_rows_per_split_host(rows)), 2);
}

extern "C" int gcpnet_tn_gemm(int n_problems, const gcp_tn_problem_t* problems, void* stream) {
    if (n_problems <= 0 || n_problems > GCP_TN_MAX_PROBLEMS || !problems) return GCPNET_E_BADARG;
    TnArgs a;
    a.n = n_problems;
    a.debug = 0;
    a.stamps = g_gcp_phase_buf; a.stamp_cap = g_gcp_phase_cap;
#ifdef GCP_DEBUG_KNOBS
    // Measurement knobs that CHANGE RESULTS exist only in a -DGCP_DEBUG_KNOBS build (never shipped; gcpnet_debug_knobs_compiled()
    // reports it and bench.py refuses such a library).  GCPNET_DEBUG_SKIP_TN: no launch, the gradients stay unwritten ("what would
    // the step cost without these"); GCPNET_TN_DEBUG: bit 0 = no products, bit 1 = no DMA after the first chunk.
    static const bool skip_env = getenv("GCPNET_DEBUG_SKIP_TN") != nullptr;
    if (skip_env) return 0;
    a.debug = getenv("GCPNET_TN_DEBUG") ? atoi(getenv("GCPNET_TN_DEBUG")) : 0;
#endif
    int blocks = 0, max_mn = 0;
    bool dma = true;
    for (int i = 0; i < n_problems; ++i) {
        const gcp_tn_problem_t& P = problems[i];
        if (P.rows < 0 || P.a.n < 0 || P.a.n > GCP_TN_MAX_SEG || P.b.n < 0 || P.b.n > GCP_TN_MAX_SEG || !P.out || !P.partial)
            return GCPNET_E_BADARG;
        if (P.splits < 1 || P.splits > 4096) return GCPNET_E_BADARG;  // (the caller's choice, gcpnet_tn_splits recommends; an odd count takes the earlier kernels)
        for (int k = 0; k < P.a.n; ++k)
            if (P.a.tb[k] && P.a.idx[k]) return GCPNET_E_BADARG;  // (a tile-blocked segment is addressed by its own row number)
        for (int k = 0; k < P.b.n; ++k)
            if (P.b.tb[k] && P.b.idx[k]) return GCPNET_E_BADARG;
        a.p[i] = P;
        a.M[i] = operand_width(P.a);
        a.N[i] = operand_width(P.b);
        if (a.M[i] <= 0 || a.N[i] <= 0) return GCPNET_E_BADARG;
        if (P.out_m < 0 || P.out_m > a.M[i] || P.out_n < 0 || P.out_n > a.N[i] || (P.out2 && (P.out2_n < 0 || P.out2_n >= a.N[i])))
            return GCPNET_E_BADARG;
        if (P.m_split < 0 || P.m_split > P.out_m || (P.m_split > 0 && !P.out_b) || (P.out2_b && !P.out2)) return GCPNET_E_BADARG;
        dma = dma && dma_ok(P.a) && dma_ok(P.b) && P.rows > 0;
        a.mb[i] = gcp_cdiv(a.M[i], TN_BM);
        a.nb[i] = gcp_cdiv(a.N[i], TN_BN);
        a.block_start[i] = blocks;
        blocks += a.mb[i] * a.nb[i] * P.splits;
        max_mn = max(max_mn, a.M[i] * a.N[i]);
    }
    a.block_start[n_problems] = blocks;
    hipStream_t st = (hipStream_t)stream;
    static const bool trace_env = getenv("GCPNET_TN_TRACE") != nullptr;  // (one line per problem on stderr: what a step asks of this call)
    if (trace_env)
        for (int i = 0; i < n_problems; ++i) {
            const gcp_tn_problem_t& P = problems[i];
            bool gather = !(stream_ok(P.a) && stream_ok(P.b));
            fprintf(stderr, "tn_gemm[%d/%d] rows %d M %d N %d splits %d act %d/%d ones %d/%d gather %d dma %d tb %d%d%d%d|%d%d%d%d\n", i, n_problems, P.rows, a.M[i], a.N[i],
                    P.splits, P.a.act, P.b.act, P.a.ones, P.b.ones, (int)gather, (int)(dma_ok(P.a) && dma_ok(P.b)), P.a.tb[0], P.a.tb[1], P.a.tb[2], P.a.tb[3],
                    P.b.tb[0], P.b.tb[1], P.b.tb[2], P.b.tb[3]);
        }
    const TnArgs all = a;  // (the reduction at the end runs over every problem, whichever kernel produced its partial sums)
    const int n_all = n_problems, max_mn_all = max_mn;
    // the pipelined kernels take every problem without a row gather / activation on load whose columns they can address
    // (GCPNET_TN_PIPE=0: the earlier kernels, for A/B runs)
    const bool pipe_env = !(getenv("GCPNET_TN_PIPE") && getenv("GCPNET_TN_PIPE")[0] == '0');
    const bool x3_ = getenv("GCPNET_TN_FP32") == nullptr;
    if (pipe_env && x3_) {
        // narrow: ONE plane buffer, three workgroups per CU (measured against two buffers / two workgroups: eight-problem launches of
        // the configs[1] / configs[2] shapes 0.405 -> 0.345 ms, 0.332 -> 0.292 ms; the thin gate problems gain most, 0.220 -> 0.173 ms)
        using Narrow = TpCfg<4, 1, 5, 1, 1>;
        // (measured and dropped: the wide form as two independent four-wave workgroups per CU, 128 x 288 each -- 256 registers with
        // spills, 7.36 against 6.75 ms on eight 256 x 276 problems of 10^6 rows, configs[4] step 196.9 against 190.3 ms)
        using Wide = TpCfg<8, 1, 9, 1>;
        // mid (round 6): FIVE waves x (1 x 5) tiles = 160 x 160, the narrow form's structure with one more tile of M: a (128,16) block's
        // [ds_pre | dgate]^T [s | norms | frame scalars | 1] is 144 x 145 -- both weight gradients of the block in ONE pass over its operands
        using Mid = TpCfg<5, 1, 5, 1, 1>;
        TnArgs narrow, wide, rest, mid;
        narrow.n = wide.n = rest.n = mid.n = 0;
        narrow.debug = wide.debug = rest.debug = mid.debug = a.debug;
        narrow.stamps = wide.stamps = rest.stamps = mid.stamps = a.stamps;
        narrow.stamp_cap = wide.stamp_cap = rest.stamp_cap = mid.stamp_cap = a.stamp_cap;
        int nblocks = 0, wblocks = 0, rblocks_ = 0, rest_mn = 0, mblocks = 0;
        const bool mid_env = tn_mid_enabled();
        bool rest_dma = true;
        for (int i = 0; i < n_problems; ++i) {
            const gcp_tn_problem_t& P = problems[i];
            const bool ok = P.rows > 0 && (P.splits & 1) == 0 && P.a.n > 0 && P.b.n > 0 && stream_ok(P.a) && stream_ok(P.b);  // (n > 0: a lane without a column of its own reads segment 0)
            const bool is_mid = ok && mid_env && a.M[i] > Narrow::BM && a.M[i] <= Mid::BM && a.N[i] <= Mid::BN;
            const bool is_wide = ok && !is_mid && a.M[i] > Narrow::BM;  // (M <= 128 with a wide N: column blocks of the narrow kernel, the thin A re-read)
            TnArgs& d = !ok ? rest : (is_mid ? mid : (is_wide ? wide : narrow));
            const int k = d.n++;
            d.p[k] = a.p[i]; d.M[k] = a.M[i]; d.N[k] = a.N[i];
            if (!ok) {
                d.mb[k] = a.mb[i]; d.nb[k] = a.nb[i];
                d.block_start[k] = rblocks_; rblocks_ += a.mb[i] * a.nb[i] * P.splits;
                rest_mn = max(rest_mn, a.M[i] * a.N[i]);
                rest_dma = rest_dma && dma_ok(P.a) && dma_ok(P.b) && P.rows > 0;
            } else if (is_mid) {
                d.mb[k] = 1; d.nb[k] = 1;
                d.block_start[k] = mblocks; mblocks += P.splits;
            } else if (is_wide) {
                d.mb[k] = gcp_cdiv(a.M[i], Wide::BM); d.nb[k] = gcp_cdiv(a.N[i], Wide::BN);
                d.block_start[k] = wblocks; wblocks += d.mb[k] * d.nb[k] * P.splits;
            } else {
                d.mb[k] = 1; d.nb[k] = gcp_cdiv(a.N[i], Narrow::BN);
                d.block_start[k] = nblocks; nblocks += d.nb[k] * P.splits;
            }
        }
        narrow.block_start[narrow.n] = nblocks; wide.block_start[wide.n] = wblocks; rest.block_start[rest.n] = rblocks_;
        mid.block_start[mid.n] = mblocks;
        constexpr size_t n_lds = (size_t)Narrow::LDS_FLOATS * sizeof(float), w_lds = (size_t)Wide::LDS_FLOATS * sizeof(float);
        // GCPNET_TN_F16=0 (read per call): the 256 x 288 form with three bf16 terms and six products again -- for data whose magnitudes
        // span more than ~2^15 INSIDE a 32-column fragment over a row split, where the shared running exponent of the two-term form costs
        // the small columns bits (tests/test_tn_gemm.py, `spikes`)
        using Wide6 = TpCfg<8, 1, 9, 1, 2, false>;
        constexpr size_t w6_lds = (size_t)Wide6::LDS_FLOATS * sizeof(float);
        const char* f16_env = getenv("GCPNET_TN_F16");
        const bool wide_f16 = GCP_TN_F16X2 && !(f16_env && f16_env[0] == '0');
        // (the attribute is per device: one flag per device ordinal, written once under a mutex)
        static std::mutex tp_mu;
        static bool tp_configured_dev[64] = {};
        int tp_dev = 0;
        if (hipGetDevice(&tp_dev) != hipSuccess || tp_dev < 0 || tp_dev >= 64) tp_dev = 0;
        std::lock_guard<std::mutex> tp_lock(tp_mu);
        bool& tp_configured = tp_configured_dev[tp_dev];
        if (!tp_configured) {
            hipError_t err = hipFuncSetAttribute((const void*)tn_pipe_kernel<4, 1, 5, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)n_lds);
            if (err == hipSuccess)
                err = hipFuncSetAttribute((const void*)tn_pipe_kernel<8, 1, 9, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)w_lds);
            if (err == hipSuccess && GCP_TN_F16X2)
                err = hipFuncSetAttribute((const void*)tn_pipe_kernel<8, 1, 9, 1, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)w6_lds);
            if (err != hipSuccess) return (int)err;
            tp_configured = true;
        }
        if (wide.n) {
            if (wide_f16 || !GCP_TN_F16X2) hipLaunchKernelGGL((tn_pipe_kernel<8, 1, 9, 1>), dim3(wblocks), dim3(Wide::NTH), w_lds, st, wide);
            else hipLaunchKernelGGL((tn_pipe_kernel<8, 1, 9, 1, 2, false>), dim3(wblocks), dim3(Wide::NTH), w6_lds, st, wide);
            GCP_HIP_CHECK_LAUNCH();
        }
        if (mid.n) {  // (30 KB of LDS: below the default limit, no attribute)
            hipLaunchKernelGGL((tn_pipe_kernel<5, 1, 5, 1, 1>), dim3(mblocks), dim3(Mid::NTH), (size_t)Mid::LDS_FLOATS * sizeof(float), st, mid);
            GCP_HIP_CHECK_LAUNCH();
        }
        if (narrow.n) {
            hipLaunchKernelGGL((tn_pipe_kernel<4, 1, 5, 1, 1>), dim3(nblocks), dim3(256), n_lds, st, narrow);
            GCP_HIP_CHECK_LAUNCH();
        }
        if (rest.n == 0) {
            hipLaunchKernelGGL(tn_reduce_kernel, dim3(min(1024, gcp_cdiv(max_mn_all, 64)), n_all), dim3(256), 0, st, all);
            GCP_HIP_CHECK_LAUNCH();
            return 0;
        }
        a = rest; blocks = rblocks_; max_mn = rest_mn; n_problems = rest.n; dma = rest_dma;
    }
    // what is left (row gathers, activations on load, odd split counts, GCPNET_TN_FP32 / GCPNET_TN_PIPE=0): the four-wave DMA kernel
    // where the operands can be DMA'd (widths / strides multiples of four floats, 16-byte aligned), the register-staged one otherwise
    // (GCPNET_TN_FP32: plain fp32 MFMA arithmetic, for the tests that hold the bf16 forms against it)
    const bool x3 = getenv("GCPNET_TN_FP32") == nullptr;
    if (dma) {
        static bool configured = false;
        const size_t lds_bytes = (size_t)TN_DMA_LDS_FLOATS * sizeof(float);
        if (!configured) {
            hipError_t err = hipFuncSetAttribute((const void*)tn_gemm_dma_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (err == hipSuccess)
                err = hipFuncSetAttribute((const void*)tn_gemm_dma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (err != hipSuccess) return (int)err;
            configured = true;
        }
        if (x3) hipLaunchKernelGGL(tn_gemm_dma_kernel<1>, dim3(blocks), dim3(256), lds_bytes, st, a);
        else hipLaunchKernelGGL(tn_gemm_dma_kernel<0>, dim3(blocks), dim3(256), lds_bytes, st, a);
    } else {
        hipLaunchKernelGGL(tn_gemm_kernel, dim3(blocks), dim3(256), 0, st, a);
    }
    GCP_HIP_CHECK_LAUNCH();
    (void)max_mn; (void)n_problems;
    const int rblocks = min(1024, gcp_cdiv(max_mn_all, 64));
    hipLaunchKernelGGL(tn_reduce_kernel, dim3(rblocks, n_all), dim3(256), 0, st, all);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int gcpnet_reduce_partials_groups(int n_parts) { return n_parts <= 0 ? 1 : gcp_cdiv(n_parts, RP_GROUP); }

extern "C" int gcpnet_reduce_partials(int n_jobs, const gcp_reduce_job_t* jobs, void* stream) {
    if (n_jobs <= 0 || n_jobs > GCP_REDUCE_MAX_JOBS || !jobs) return GCPNET_E_BADARG;
    ReduceArgs a;
    int max_groups = 1;
    for (int i = 0; i < n_jobs; ++i) {
        const gcp_reduce_job_t& J = jobs[i];
        if (J.n_parts <= 0 || J.width <= 0 || !J.parts || !J.tmp || !J.out) return GCPNET_E_BADARG;
        a.j[i] = J;
        max_groups = max(max_groups, gcp_cdiv(J.n_parts, RP_GROUP));
    }
    hipStream_t st = (hipStream_t)stream;
    if (max_groups == 1) {  // every job fits one group (<= 64 parts: small graphs): the first level writes the result itself
        for (int i = 0; i < n_jobs; ++i) a.j[i].tmp = a.j[i].out;
        hipLaunchKernelGGL(reduce_partials_kernel<false>, dim3(1, n_jobs), dim3(256), 0, st, a);
        GCP_HIP_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(reduce_partials_kernel<false>, dim3(max_groups, n_jobs), dim3(256), 0, st, a);
    GCP_HIP_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_partials_kernel<true>, dim3(1, n_jobs), dim3(256), 0, st, a);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

// ---- weight gradients of n GCP2 blocks in one call (include/gcpnet_hip.h, gcp2_wgrad_job_t) ------------------------------------------
namespace {

struct WgradDims {
    int nf, EP, VOP, si, n1;
    bool has_vec, gated;
};

inline WgradDims wgrad_dims(const gcp2_wgrad_job_t& J) {
    WgradDims d;
    d.has_vec = J.vi > 0;
    d.nf = (J.use_frames && J.vi > 0) ? 9 : 0;
    d.EP = gcp_round_up(J.hidden + d.nf, 4);
    d.VOP = gcp_round_up(J.vo, 4);
    d.si = 0;
    for (int k = 0; k < J.s_in.n; ++k) d.si += J.s_in.dim[k];
    d.n1 = d.si + (d.has_vec ? d.EP : 0) + 1;
    d.gated = J.gated && J.vi > 0 && J.vo > 0;
    return d;
}

inline bool wgrad_job_ok(const gcp2_wgrad_job_t& J) {
    if (J.rows <= 0 || J.so <= 0 || J.s_in.n < 0 || J.s_in.n + (J.vi > 0 ? 1 : 0) > GCP_TN_MAX_SEG || !J.ds_pre || !J.d_w_scalar || !J.d_b_scalar) return false;
    if (J.vi > 0 && (!J.ext || (J.w_part && (J.n_parts <= 0 || J.w_width <= 0 || !J.d_w_small)))) return false;
    if (J.gated && J.vi > 0 && J.vo > 0 && (!J.dgate || !J.d_w_gate || !J.d_b_gate)) return false;
    if (J.gated && J.vi > 0 && J.vo > 0 && (J.gate_lin ? (J.act_v != GCP_ACT_NONE || !J.w_scalar || !J.b_scalar) : !J.s_pre)) return false;
    return true;
}

// gate_lin: d vector_out_scale.weight[o, j] = sum_k G[o, k] W[j, k] + db[o] b[j] for the jobs of a call (G = dgate^T [s | ext] from the
// GEMM, db = its ones column): vo * so outputs of K multiply-adds per job -- one launch for all of them
struct GateFinishArgs {
    const float* G[GCP_TN_MAX_PROBLEMS * 2];
    const float* W[GCP_TN_MAX_PROBLEMS * 2];
    const float* b[GCP_TN_MAX_PROBLEMS * 2];
    const float* db[GCP_TN_MAX_PROBLEMS * 2];
    float* out[GCP_TN_MAX_PROBLEMS * 2];
    int vo[GCP_TN_MAX_PROBLEMS * 2], so[GCP_TN_MAX_PROBLEMS * 2], K[GCP_TN_MAX_PROBLEMS * 2];
};
__global__ __launch_bounds__(256) void gate_finish_kernel(GateFinishArgs a) {
    // a WAVE per output column j (sixteen columns per workgroup): its lanes read row j of W as one contiguous piece (k = lane, lane + 64,
    // lane + 128), multiply it with the <= 32 rows of G held in LDS and reduce each product over the wave.  (One thread per output with
    // W[j, :] read straight from memory -- a 564-byte stride between lanes -- took 54 us per launch: 0.55 ms of a configs[1] step.)
    __shared__ float Gs[32 * 193];
    const int q = blockIdx.y, vo = a.vo[q], so = a.so[q], K = a.K[q];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (vo > 32 || K > 192) {  // (outside the tile sizes: the plain form)
        for (int i = blockIdx.x * 256 + tid; i < vo * so; i += gridDim.x * 256) {
            const int o = i / so, j = i - o * so;
            float acc = 0.f;
            for (int k = 0; k < K; ++k) acc = fmaf(a.G[q][(int64_t)o * K + k], a.W[q][(int64_t)j * K + k], acc);
            a.out[q][i] = fmaf(a.db[q][o], a.b[q][j], acc);
        }
        return;
    }
    for (int i = tid; i < vo * K; i += 256) Gs[(i / K) * 193 + i % K] = a.G[q][i];
    __syncthreads();
    for (int jj = 0; jj < 4; ++jj) {
        const int j = blockIdx.x * 16 + w * 4 + jj;
        if (j >= so) break;  // (wave-uniform)
        const float* wr = a.W[q] + (int64_t)j * K;
        float wv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) wv[c] = lane + 64 * c < K ? wr[lane + 64 * c] : 0.f;
        const float bj = a.b[q][j];
        for (int o = 0; o < vo; ++o) {
            const float* g = Gs + o * 193;
            float p = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) p = fmaf(lane + 64 * c < K ? g[lane + 64 * c] : 0.f, wv[c], p);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) p += __shfl_xor(p, off);
            if (lane == 0) a.out[q][(int64_t)o * so + j] = fmaf(a.db[q][o], bj, p);
        }
    }
}

}  // namespace

extern "C" int64_t gcpnet_gcp2_weight_grads_workspace(int n, const gcp2_wgrad_job_t* jobs) {
    if (n <= 0 || !jobs) return 0;
    int64_t fl = 0;
    for (int i = 0; i < n; ++i) {
        const gcp2_wgrad_job_t& J = jobs[i];
        const WgradDims d = wgrad_dims(J);
        const int64_t splits = gcpnet_tn_splits(J.rows, 0, 0);
        fl += splits * J.so * d.n1;
        if (d.gated) fl += J.gate_lin ? splits * d.VOP * d.n1 + (int64_t)J.vo * d.n1 : splits * d.VOP * (J.so + 1);
        if (d.has_vec && J.w_part) fl += (int64_t)gcpnet_reduce_partials_groups(J.n_parts) * J.w_width;
    }
    return fl;
}

extern "C" int gcpnet_gcp2_weight_grads(int n, const gcp2_wgrad_job_t* jobs, float* workspace, void* stream) {
    if (n <= 0 || !jobs || !workspace) return GCPNET_E_BADARG;
    for (int i = 0; i < n; ++i)
        if (!wgrad_job_ok(jobs[i])) return GCPNET_E_BADARG;
    gcp_tn_problem_t probs[GCP_TN_MAX_PROBLEMS];
    gcp_reduce_job_t reds[GCP_REDUCE_MAX_JOBS];
    GateFinishArgs fin;
    int np = 0, nr = 0, nfin = 0, fin_blocks = 1;
    float* ws = workspace;
    auto flush_f = [&]() -> int {
        if (nfin) {
            hipLaunchKernelGGL(gate_finish_kernel, dim3((unsigned)fin_blocks, (unsigned)nfin), dim3(256), 0, (hipStream_t)stream, fin);
            GCP_HIP_CHECK_LAUNCH();
        }
        nfin = 0; fin_blocks = 1;
        return 0;
    };
    auto flush_p = [&]() -> int {
        const int rc = np ? gcpnet_tn_gemm(np, probs, stream) : 0;
        np = 0;
        return rc;
    };
    auto flush_r = [&]() -> int {
        const int rc = nr ? gcpnet_reduce_partials(nr, reds, stream) : 0;
        nr = 0;
        return rc;
    };
    auto plain = [](gcp_operand_t& o, const float* ptr, int width, int tb) {
        o = gcp_operand_t{};
        o.n = 1;
        o.ptr[0] = ptr; o.dim[0] = width; o.ld[0] = width; o.tb[0] = tb;
    };
    // every GEMM problem first (the reductions' launches follow, as the block-by-block host code ordered them)
    for (int i = 0; i < n; ++i) {
        const gcp2_wgrad_job_t& J = jobs[i];
        const WgradDims d = wgrad_dims(J);
        const int splits = gcpnet_tn_splits(J.rows, 0, 0);
        // both gradients of a gated block in ONE product when [ds_pre | dgate] fits a block of rows of the pipelined kernels (128, or 160
        // with the five-wave form when the second operand fits 160 too): the second operand -- the larger one -- is then read once
        const int mf = J.so + d.VOP;
        const bool fuse = d.gated && J.gate_lin && (mf <= 128 || (mf <= 160 && d.n1 <= 160 && tn_mid_enabled()));
        {   // d scalar_out.weight | bias: ds_pre^T [s segments | ext | 1]
            gcp_tn_problem_t& P = probs[np++];
            P = gcp_tn_problem_t{};
            P.rows = J.rows;
            plain(P.a, J.ds_pre, J.so, J.ds_pre_tb);
            P.b = J.s_in;
            P.b.act = 0; P.b.slope = J.slope; P.b.ones = 1;
            if (d.has_vec) {
                const int k = P.b.n++;
                P.b.ptr[k] = J.ext; P.b.idx[k] = nullptr; P.b.dim[k] = d.EP; P.b.ld[k] = d.EP; P.b.tb[k] = 0;
            }
            const int K = d.si + (d.has_vec ? J.hidden + d.nf : 0);
            P.out = J.d_w_scalar; P.out_sm = K; P.out_sn = 1; P.out_m = J.so; P.out_n = K;
            P.out2 = J.d_b_scalar; P.out2_n = d.n1 - 1;
            P.splits = splits;
            P.partial = ws;
            ws += (int64_t)splits * J.so * d.n1;
            if (fuse) {  // rows so .. so + vo - 1 of the product: G | d gate bias
                const int k = P.a.n++;
                P.a.ptr[k] = J.dgate; P.a.idx[k] = nullptr; P.a.dim[k] = d.VOP; P.a.ld[k] = d.VOP; P.a.tb[k] = 0;
                ws += (int64_t)splits * d.VOP * d.n1;  // (the partial sums: so + VOP rows per split)
                float* G = ws;
                ws += (int64_t)J.vo * d.n1;
                P.m_split = J.so; P.out_m = J.so + J.vo;
                P.out_b = G; P.out_b_sm = K; P.out2_b = J.d_b_gate;
                fin.G[nfin] = G; fin.W[nfin] = J.w_scalar; fin.b[nfin] = J.b_scalar; fin.db[nfin] = J.d_b_gate; fin.out[nfin] = J.d_w_gate;
                fin.vo[nfin] = J.vo; fin.so[nfin] = J.so; fin.K[nfin] = K;
                fin_blocks = max(fin_blocks, gcp_cdiv(J.so, 16));
                ++nfin;
            }
            if (np == GCP_TN_MAX_PROBLEMS) { const int rc = flush_p(); if (rc) return rc; }
            if (nfin == GCP_TN_MAX_PROBLEMS * 2) {
                int rc = flush_p(); if (rc) return rc;
                rc = flush_f(); if (rc) return rc;
            }
        }
        if (fuse) {
        } else if (d.gated && J.gate_lin) {  // G | d bias = dgate^T [s segments | ext | 1]: the first product's second operand again, no s_pre
            gcp_tn_problem_t& P = probs[np++];
            P = gcp_tn_problem_t{};
            P.rows = J.rows;
            plain(P.a, J.dgate, d.VOP, 0);
            P.b = J.s_in;
            P.b.act = 0; P.b.slope = J.slope; P.b.ones = 1;
            P.a.slope = J.slope;
            if (d.has_vec) {
                const int k = P.b.n++;
                P.b.ptr[k] = J.ext; P.b.idx[k] = nullptr; P.b.dim[k] = d.EP; P.b.ld[k] = d.EP; P.b.tb[k] = 0;
            }
            const int K = d.si + (d.has_vec ? J.hidden + d.nf : 0);
            float* G = ws;
            ws += (int64_t)J.vo * d.n1;
            P.out = G; P.out_sm = K; P.out_sn = 1; P.out_m = J.vo; P.out_n = K;
            P.out2 = J.d_b_gate; P.out2_n = d.n1 - 1;
            P.splits = splits;
            P.partial = ws;
            ws += (int64_t)splits * d.VOP * d.n1;
            fin.G[nfin] = G; fin.W[nfin] = J.w_scalar; fin.b[nfin] = J.b_scalar; fin.db[nfin] = J.d_b_gate; fin.out[nfin] = J.d_w_gate;
            fin.vo[nfin] = J.vo; fin.so[nfin] = J.so; fin.K[nfin] = K;
            fin_blocks = max(fin_blocks, gcp_cdiv(J.so, 16));
            ++nfin;
            // (the finishing launch reads what the GEMM launches of its jobs wrote: flush both together)
            // (the finishing launch reads what the GEMM launches wrote: it follows the last of them, or comes when its table is full)
            if (np == GCP_TN_MAX_PROBLEMS) { const int rc = flush_p(); if (rc) return rc; }
            if (nfin == GCP_TN_MAX_PROBLEMS * 2) {
                int rc = flush_p(); if (rc) return rc;
                rc = flush_f(); if (rc) return rc;
            }
        } else if (d.gated) {  // d vector_out_scale.weight | bias: dgate^T [act_v(s_pre) | 1]
            gcp_tn_problem_t& P = probs[np++];
            P = gcp_tn_problem_t{};
            P.rows = J.rows;
            plain(P.a, J.dgate, d.VOP, 0);
            plain(P.b, J.s_pre, J.so, J.s_pre_tb);
            P.a.slope = P.b.slope = J.slope;
            P.b.act = J.act_v; P.b.ones = 1;
            P.out = J.d_w_gate; P.out_sm = J.so; P.out_sn = 1; P.out_m = J.vo; P.out_n = J.so;
            P.out2 = J.d_b_gate; P.out2_n = J.so;
            P.splits = splits;
            P.partial = ws;
            ws += (int64_t)splits * d.VOP * (J.so + 1);
            if (np == GCP_TN_MAX_PROBLEMS) { const int rc = flush_p(); if (rc) return rc; }
        }
    }
    { const int rc = flush_p(); if (rc) return rc; }
    { const int rc = flush_f(); if (rc) return rc; }
    for (int i = 0; i < n; ++i) {
        const gcp2_wgrad_job_t& J = jobs[i];
        if (!(J.vi > 0 && J.w_part)) continue;
        gcp_reduce_job_t& R = reds[nr++];
        R.parts = J.w_part; R.n_parts = J.n_parts; R.width = J.w_width; R.tmp = ws; R.out = J.d_w_small;
        ws += (int64_t)gcpnet_reduce_partials_groups(J.n_parts) * J.w_width;
        if (nr == GCP_REDUCE_MAX_JOBS) { const int rc = flush_r(); if (rc) return rc; }
    }
    return flush_r();
}

extern "C" int gcpnet_stream_wait_stream(void* to, void* from) {
    // a ring of events: a wait enqueued on `to` refers to the record that preceded it, so an event may be recorded again as soon as
    // that wait has been enqueued; 256 slots keep re-use far from any call still being processed by the runtime
    constexpr int RING = 256, MAX_DEV = 16;
    static hipEvent_t ring[MAX_DEV][RING];
    static int made_[MAX_DEV] = {0}, next_[MAX_DEV] = {0};
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return GCPNET_E_BADARG;  // (events belong to the current device)
    std::lock_guard<std::mutex> lock(mu);
    int &made = made_[dev], &next = next_[dev];
    if (made < RING && next == made) {
        const hipError_t err = hipEventCreateWithFlags(&ring[dev][made], hipEventDisableTiming);
        if (err != hipSuccess) return (int)err;
        ++made;
    }
    hipEvent_t ev = ring[dev][next];
    next = (next + 1) % RING;
    hipError_t err = hipEventRecord(ev, (hipStream_t)from);
    if (err == hipSuccess) err = hipStreamWaitEvent((hipStream_t)to, ev, 0);
    return (int)err;
}
