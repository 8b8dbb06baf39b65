// Backward (data path) of a chain of residual GCP2 blocks, x_k = x_{k-1} + GCP_k(x_{k-1}) (ResGCP, reference
// src/models/components/gcpnet.py:921-924; the reference gets this from autograd), one launch for the whole chain.
//
// The gradient of the chain state never leaves the registers between blocks:
//   * d(s) of a 32-row tile lives in the 32x32 MFMA accumulator layout.  The scalar_out adjoint W^T ds_pre is
//     accumulated ON TOP of it (the ResGCP pass-through is the accumulator's initial value), so the result is directly
//     the next block's d(s_out);
//   * d(V) lives in the same layout, one register per (channel, xyz): it is what the vector-epilogue adjoint reads and what
//     the vector-prologue adjoint produces (vec_mfma.h).
// All Linears -- scalar_out^T, the gate adjoint, and the small vector ones (vector_down / vector_up forward recompute and
// their adjoints) -- run on the matrix cores with B fragments taken straight from registers wherever the producer is an MFMA;
// the per-row nonlinear pieces in between are element-wise register code.  LDS (~17 KB per wave, two waves per SIMD) only
// holds the block's input vectors and the row-major copies the per-tile weight-gradient partial sums need.
// Per block the wave reads what the forward saved (s_pre, the gates, the block's input vectors) and writes what the
// weight-gradient GEMM needs (ds_pre, dgate, norms / frame scalars, the per-tile small-weight partial sums).  Those reads are
// requested one block ahead: the s_pre tile of block k-1 goes into the registers the data GEMM of block k has just
// released, gates and vectors into spare registers -- a block's HBM latency is spent under the previous block's arithmetic.
#include "common.h"
#include "tile_io.h"
#include "vec_mfma.h"
#include "gcp_bf16x3.h"
#include "gcp_f16x2.h"

#include <algorithm>
#include <cstdlib>
#include <functional>
#include <map>
#include <mutex>
#include <queue>
#include <tuple>
#include <type_traits>
#include <vector>

// GCP_CB_X: measurement builds whose RESULTS ARE WRONG (tools/cb_variants.sh): bits remove one cost each so that its share of the
// launch time can be read under real contention.  1: step E's fragments loaded once; 2: no partial-sum passes; 4: no steps A - C;
// 8: no step F; 16: no gate MFMAs in D; 32: no ds_pre store; 64: no s_pre load; 128: no step E
#ifndef GCP_CB_X
#define GCP_CB_X 0
#endif
#ifndef GCP_CB_TNC
#define GCP_CB_TNC 3  // independent accumulator chains of a pass (1..3)
#endif
#ifndef GCP_CB_TNB
#define GCP_CB_TNB 6  // operand pairs of a small weight-gradient pass read from LDS per batch (3, 6, 12 or 24)
#endif

int gcp2_chain_bwd_registers(int rows, const float* frames, int n, const gcp2_chain_bwd_item_t* items, const float* d_s_out,
                             const float* d_v_out, const int32_t* out_idx, const float* out_scale, float* d_s_in, float* d_v_in,
                             unsigned* flags, int n_flags, hipStream_t st);

namespace {

struct ChainItemB {
    const float* pack;
    const float* v_in;
    const float* s_pre;
    const float* gate;
    float* ds_pre;
    float* dgate;
    float* ext;
    float* w_part;
    int act_s, act_v;
    int tb;  // s_pre / ds_pre in the tile-blocked layout (include/gcpnet_hip.h, gcp2_chain_item_t)
    const unsigned* sign;  // sign mask of s_pre (SGN instantiations: read instead of s_pre)
};

struct ChainBwdParams {
    int rows;
    const float* frames;
    gcp2_opts_t o;  // shared: slope, vmode, vector_residual, e3
    int n;
    ChainItemB it[GCP_MAX_CHAIN];
    const float* d_s_out;
    const float* d_v_out;
    // gathered form: the incoming gradient of row r is out_scale[j] * d_*_out[j], j = out_idx[r] -- the adjoint of a segment
    // mean / sum over the chain's output (the aggregation, gcpnet.py:939-947) read straight from the node-level table instead of
    // from a materialised [rows, .] copy; both NULL: d_*_out are per-row tensors
    const int32_t* out_idx;
    const float* out_scale;
    float* d_s_in;
    float* d_v_in;
    unsigned long long* stamps;
    long long stamp_cap;
    GcpShape sh;
    // Tail split (cb_plan_split below): workgroups [0, n_split) run only the blocks n - 1 .. k_split of the tiles [0, n_split) and
    // hand their state over through d_s_in / d_v_in; workgroups [tiles, tiles + n_split) finish those tiles (blocks k_split - 1 .. 0).
    // flags[tile]: 0 = untouched, 1 = first half claimed, 2 = first half done, 3 = the second-half workgroup took the whole tile
    // over (it found the flag at 0: no dispatch order is assumed, only used).  n_split == 0: every workgroup runs a whole tile.
    int tiles, n_split, k_split;
    int rev;  // test hook: workgroup indices reversed (second halves dispatched FIRST: they take tiles over or wait for a running first half)
    unsigned* flags;
};

struct CbLds {
    int VS, HS, FS, DS;
    int o_vt, o_x, o_vht, o_fr, o_dext, o_e3, o_stage, total;
};

__host__ __device__ inline CbLds cb_lds(const GcpShape& s) {
    CbLds l;
    l.VS = gcp_odd(3 * s.vi);
    l.HS = gcp_odd(3 * s.H);
    l.FS = gcp_odd(3 * s.HF);
    l.DS = gcp_odd(s.H + s.nf);
    l.o_vt = 0;
    l.o_x = l.o_vt + 32 * l.VS;
    l.o_vht = l.o_x + 32 * (l.VS > l.FS ? l.VS : l.FS);
    l.o_fr = l.o_vht + 32 * l.HS;
    l.o_dext = l.o_fr + 32 * 9;
    l.o_e3 = l.o_dext + 32 * l.DS;
    l.o_stage = l.o_e3 + 32 * 3;
    l.total = l.o_stage + GCP_ACC_STAGE_HALF_FLOATS;
    return l;
}

template <int N>
struct WFragC;
template <>
struct WFragC<1> {
    float v[1];
    __device__ __forceinline__ void load(const float* p) { v[0] = p[0]; }
};
template <>
struct WFragC<2> {
    float v[2];
    __device__ __forceinline__ void load(const float* p) {
        float2 t = *reinterpret_cast<const float2*>(p);
        v[0] = t.x; v[1] = t.y;
    }
};
template <>
struct WFragC<4> {
    float v[4];
    __device__ __forceinline__ void load(const float* p) {
        float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
};

// NTG: 32-wide tiles of the scalar state (si == so == 32 * NTG); VQ: register quads of the vector state (vi == vo <= 8 * VQ);
// PWL: all activations are identity / relu / leakyrelu.
// HC: hidden vector channels as a compile-time constant (0 = run-time value): with it the tests "channel < H" of the element-wise
// register code fold, and only the registers that can hold a [vh | vf] channel are walked.
// sigmoid(gate) of a row in the register-quad layout (channels 8 q + 4 hi .. + 3, q < VQ): one wave-uniform test, all VQ
// requests together, then the out-of-range selects (tile_io.h, gcp_load_tile4: a gcp_load4 call per quad is waited for on the
// spot).  Consumed here -- keeping raw requests in flight into step C costs the backward kernel 90 more spilled registers.
template <int VQ, int NV>
__device__ __forceinline__ void gcp_load_gate(const float* __restrict__ gate, int row, int vo, int hi, bool ok, bool vec,
                                              float (&sg)[NV]) {
    static_assert(NV >= 4 * VQ, "one register quad per 8 channels");
    float4 g[VQ];
    if (!gate) {
#pragma unroll
        for (int q = 0; q < VQ; ++q) g[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else if (vec) {
        const float* rp = gate + (int64_t)(ok ? row : 0) * vo;
#pragma unroll
        for (int q = 0; q < VQ; ++q) g[q] = *reinterpret_cast<const float4*>(rp + min(8 * q + 4 * hi, vo - 4));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < VQ; ++q) {
            const bool in = ok && 8 * q + 4 * hi + 3 < vo;
            g[q] = make_float4(in ? g[q].x : 0.f, in ? g[q].y : 0.f, in ? g[q].z : 0.f, in ? g[q].w : 0.f);
        }
    } else {
#pragma unroll
        for (int q = 0; q < VQ; ++q) g[q] = gcp_load4(gate, row, vo, 8 * q + 4 * hi, ok, false);
    }
#pragma unroll
    for (int q = 0; q < VQ; ++q) { sg[4 * q] = g[q].x; sg[4 * q + 1] = g[q].y; sg[4 * q + 2] = g[q].z; sg[4 * q + 3] = g[q].w; }
}

// The block's input vectors, a 32-row tile of [rows, 3 vi] (vi % 4 == 0, 16-byte aligned: host checks) = 32 * q contiguous
// 16-byte pieces (q = 3 vi / 4 <= 12), six per lane: all requests first (clamped: no branches), one guarded LDS write per piece
// later.  (The general segment loader of tile_io.h carries its fallback paths and ~60 branches into the unrolled block body.)
struct CbVin {
    float4 v[6];
};
__device__ __forceinline__ void cb_vin_issue(CbVin& b, const float* __restrict__ v_in, int vi, int r0, int rows, int lane) {
    const int q = (3 * vi) >> 2, n4 = min(rows - r0, GCP_TILE_ROWS) * q;
    const float4* src = reinterpret_cast<const float4*>(v_in + (int64_t)r0 * 3 * vi);
#pragma unroll
    for (int k = 0; k < 6; ++k) b.v[k] = src[min(lane + 64 * k, n4 - 1)];
}
__device__ __forceinline__ void cb_vin_commit(const CbVin& b, float* vt, int VS, int vi, int r0, int rows, int lane) {
    const int q = (3 * vi) >> 2, n4 = min(rows - r0, GCP_TILE_ROWS) * q, tot = GCP_TILE_ROWS * q;
    const unsigned magic = (unsigned)(((1ull << 32) + (unsigned)q - 1) / (unsigned)q);  // idx / q for idx < 2^16
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int idx = lane + 64 * k;
        if (idx < tot) {
            const int r = (int)__umulhi((unsigned)idx, magic), c4 = idx - r * q;
            const bool ok = idx < n4;
            float* d = vt + r * VS + 4 * c4;
            d[0] = ok ? b.v[k].x : 0.f; d[1] = ok ? b.v[k].y : 0.f; d[2] = ok ? b.v[k].z : 0.f; d[3] = ok ? b.v[k].w : 0.f;
        }
    }
}

// B6: W^T ds_pre on the bf16 matrix pipe, both operands as three bf16 terms, six products (gcp_bf16x3.h: exact to fp32 round-off)
// PAD: si == so is a multiple of 4 below 32 NTG (LBA: 100): the tile loads already return zeros past column so, the weight images
// are zero-padded, so the padding columns of d(s) stay zero through the chain; only the two full-line stores need the column test.
// SGN: act'(s_pre) from the forward's sign mask (one bit per element, gcp2_chain_item_t.s_sign) instead of s_pre itself: piecewise-
// linear activations only.  Two words per lane and block instead of sixteen 16-byte pieces -- 16 KB less HBM traffic per tile and
// block, and no 64-register request in flight between steps C and D (measured with GCP_CB_X & 64: the s_pre request costs 8.6 % of
// the launch where it is, and cannot move earlier: its 64 destination registers do not fit beside steps A - C).
template <int NTG, int VQ, bool PWL, int HC, bool B6, bool PAD = false, bool SGN = false>
__global__ __launch_bounds__(GCP_WAVE, 2) void gcp2_chain_bwd_kernel(ChainBwdParams p_kernarg) {
    static_assert(!SGN || (PWL && NTG % 2 == 0), "sign masks: piecewise-linear activations, whole words");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NV = 4 * VQ;  // registers per xyz component of a vector-channel quantity
    constexpr int NX = HC ? 4 * ((HC + 3 + 7) / 8) : 8;  // registers per xyz component of a [vh | vf] quantity (H + 3 <= 16)
    // The parameters are read through the kernarg segment pointer, laundered at every section boundary: uniform values are then
    // re-loaded (s_load) by the section that uses them instead of staying live in SGPRs across the whole block loop, where
    // they did not fit (~250 SGPRs spilled into VGPR lanes, a v_readlane_b32 per use).
    typedef const __attribute__((address_space(4))) ChainBwdParams* Karg;
    Karg kp = (Karg)__builtin_amdgcn_kernarg_segment_ptr();
// (generic view of the laundered constant-address-space pointer: the address space is inferred back, loads stay s_load)
#define p (*(const ChainBwdParams*)kp)
    int lane = threadIdx.x;
    int e = lane & 31, hi = lane >> 5;
    // which tile and which blocks of it (tail split, see ChainBwdParams): wave-uniform, decided before anything is loaded
    const int wg = p.rev ? (int)(gridDim.x - 1 - blockIdx.x) : (int)blockIdx.x;
    int tile = wg, k_hi = p.n - 1, k_lo = 0;
    bool handed = false;   // the incoming gradient is the state another workgroup left in d_s_in / d_v_in
    // (k_lo > 0: this workgroup leaves such a state behind)
    if (p.n_split > 0) {
        typedef __attribute__((address_space(1))) unsigned gu32;
        if (wg < p.n_split) {
            gu32* fl = (gu32*)(p.flags + tile);
            unsigned old = 0;
            if (lane == 0) old = __hip_atomic_compare_exchange_strong(fl, &old, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 0u : old;
            if (__builtin_amdgcn_readfirstlane(old) != 0) return;  // taken over by the workgroup of the second half
            k_lo = p.k_split;
        } else if (wg >= p.tiles) {
            tile = wg - p.tiles;
            gu32* fl = (gu32*)(p.flags + tile);
            unsigned f = 0;
            if (lane == 0) {
                for (;;) {
                    f = __hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (f == 0) {  // nobody has started this tile: run all of it here
                        unsigned exp0 = 0;
                        if (__hip_atomic_compare_exchange_strong(fl, &exp0, 3u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { f = 3; break; }
                        continue;
                    }
                    if (f == 2) break;
                    __builtin_amdgcn_s_sleep(32);  // (f == 1: the first half is running on some CU and will finish)
                }
            }
            f = __builtin_amdgcn_readfirstlane(f);
            if (f == 2) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // one buffer_inv sc1 after the match, then plain loads
                k_hi = p.k_split - 1;
                handed = true;
            }
        }
    }
    const int r0 = tile * GCP_TILE_ROWS;
    int row = r0 + e;
    GcpShape S;
    CbLds L;
    ChainItemB it;
    int kcur = 0;
    int rows, so, vi, H, HF, SVB, SVD, EP, VOP, NUG, xg, xs;
    float slope, ns_s, ns_v;
    bool scalar_gate, vec_vo, row_ok;
    float *vt, *xt, *vht, *fr, *dext, *e3t, *stage;
    // vt: the block's input vectors, [row][channel][xyz]; xt: row-major copy of d(vector_up output), later of [d vh | d vf]
    // (weight-gradient partials); vht: row-major copy of vector_down(v); dext: d(norms | frame scalars), from the lanes of the
    // scalar_out adjoint to the vh / vf channels' lanes; e3t: signs of the x_cross projections (e3 variant only); stage:
    // transposition tile of the row-wise stores (tile_io.h, gcp_store_acc_rows)
#define CB_RELOAD()                                                                                                       \
    do {                                                                                                                  \
        S = p.sh;                                                                                                         \
        it = p.it[kcur];                                                                                                  \
        L = cb_lds(S);                                                                                                    \
        rows = p.rows;                                                                                                    \
        row_ok = row < rows;                                                                                              \
        vt = lds + L.o_vt; xt = lds + L.o_x; vht = lds + L.o_vht; fr = lds + L.o_fr; dext = lds + L.o_dext;                \
        e3t = lds + L.o_e3; stage = lds + L.o_stage;                                                                      \
        so = S.so; vi = S.vi; /* si == so, vo == vi */                                                                    \
        H = HC ? HC : S.H; HF = HC ? HC + 3 : S.HF; /* (frames are in use whenever HC is given) */                         \
        SVB = HC ? 4 * ((HC + 7) / 8) : S.SVB; SVD = HC ? 4 * ((HC + 3 + 7) / 8) : S.SVD;                                  \
        EP = HC ? gcp_round_up(HC + 9, 4) : gcp_round_up(S.H + S.nf, 4); VOP = gcp_round_up(vi, 4);                                                      \
        slope = p.o.slope;                                                                                                \
        scalar_gate = p.o.vmode == GCP_VMODE_SCALAR_GATE;                                                                 \
        NUG = S.NUG;                                                                                                      \
        xg = NTG / NUG; xs = NTG - xg * NUG; /* group / slot of the 32-wide tile holding the norms and frame scalars */   \
        vec_vo = true; /* vi % 4 == 0 and 16-byte aligned gates: checked by the host */                                                                                           \
        ns_s = gcp_neg_slope(it.act_s, slope); ns_v = gcp_neg_slope(it.act_v, slope);                                     \
    } while (0)
    // (the per-lane addresses of the loop body would be hoisted out of the loop and spilled, too: lane indices laundered with it)
#define CB_LAUNDER()                                                                                                      \
    do {                                                                                                                  \
        asm volatile("" : "+v"(lane), "+v"(e), "+v"(hi), "+s"(kp));                                                       \
        row = r0 + e;                                                                                                     \
        CB_RELOAD();                                                                                                      \
    } while (0)
    kcur = k_hi;
    CB_RELOAD();

    gcp_stamp(p.stamps, p.stamp_cap, 0, lane);
    f32x16 dyr[NTG];
    // (two-term fp16 form of step E, gcp_f16x2.h: d(s) stays multiplied by the power of two the last step E gave it -- dyr = true d(s) /
    // dsc_inv, per lane = per row --; step D and the final store fold the inverse into a factor they apply anyway)
    [[maybe_unused]] float dsc_inv = 1.f;
    float sg[NV];  // sigmoid(gate) of the current block: channel crow(r, hi) of row e
    unsigned sm[NTG / 2 > 0 ? NTG / 2 : 1];  // SGN: sign words of the current block's s_pre (bit 16 t + r = register r of tile t)
    // ---- prologue: everything the LAST block needs, plus the incoming gradients, in one memory round trip ----------
    {
        CbVin vb;
        cb_vin_issue(vb, it.v_in, vi, r0, rows, lane);
        if (S.nf) gcp_load_frames(p.frames, r0, rows, fr, lane);
        gcp_load_gate<VQ>(scalar_gate ? it.gate : nullptr, row, vi, hi, row_ok, vec_vo, sg);
        if constexpr (SGN) {
#pragma unroll
            for (int w = 0; w < NTG / 2; ++w) sm[w] = it.sign[((int64_t)(r0 >> 5) * (NTG / 2) + w) * 64 + lane];
        }
        int64_t orow0 = row;
        float osc0 = 1.f;
        if (p.out_idx && !handed) {  // (wave-uniform) gathered incoming gradient: source row and weight of this lane's row
            orow0 = p.out_idx[row_ok ? row : rows - 1];
            if (p.out_scale) osc0 = p.out_scale[orow0];
        }
        const float* ds_src = handed ? p.d_s_in : p.d_s_out;
#pragma unroll
        for (int t = 0; t < NTG; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j0 = 32 * t + 8 * q + 4 * hi;
                const float4 b = gcp_load4(ds_src, orow0, so, j0, row_ok, true);
                dyr[t][4 * q] = osc0 * b.x; dyr[t][4 * q + 1] = osc0 * b.y; dyr[t][4 * q + 2] = osc0 * b.z; dyr[t][4 * q + 3] = osc0 * b.w;
            }
        cb_vin_commit(vb, vt, L.VS, vi, r0, rows, lane);
    }
    gcp_stamp(p.stamps, p.stamp_cap, 1, lane);

    // (`gathered`: state_src is the node-level table of the incoming gradient, see ChainBwdParams::out_idx; the source row and its
    // weight are looked up again where they are needed rather than carried in registers across the blocks)
    auto load_state = [&](const float* state_src, float(&out)[3][NV], bool gathered) {
        int64_t srow = row;
        float sc = 1.f;
        if (gathered) {  // (wave-uniform)
            srow = p.out_idx[row_ok ? row : rows - 1];
            if (p.out_scale) sc = p.out_scale[srow];
        }
#pragma unroll
        for (int q = 0; q < VQ; ++q) {
            const int o0 = 8 * q + 4 * hi;
            const bool on = row_ok && o0 < vi;
            const float4* sp = reinterpret_cast<const float4*>(state_src + (on ? srow * 3 * vi + 3 * o0 : 0));
            float t[12];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float4 x = sp[j];
                t[4 * j] = on ? sc * x.x : 0.f; t[4 * j + 1] = on ? sc * x.y : 0.f; t[4 * j + 2] = on ? sc * x.z : 0.f; t[4 * j + 3] = on ? sc * x.w : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int d = 0; d < 3; ++d) out[d][4 * q + i] = t[3 * i + d];
        }
    };
    float dvs[3][NV];  // d(V) state entering the current block: channel crow(r, hi) of row e.  Loaded here for the last block;
    load_state(handed ? p.d_v_in : p.d_v_out, dvs, p.out_idx != nullptr && !handed);  // afterwards carried over in registers from the end of the block before, where it is computed
    // sums of the second partial-sum pass of the block before and where they go: they leave with the NEXT block's small stores (end
    // of its step C), so that the coming block's requests do not queue behind them
    f32x4 tn2 = f32x4{0.f, 0.f, 0.f, 0.f};
    float* tn2_part = nullptr;
    for (int k = k_hi; k >= k_lo; --k) {
        kcur = k;
        CB_LAUNDER();
        const bool stamp_here = k == 0;
        // d(V) chain state: channel crow(r, hi) of row e, 12 consecutive floats per register quad.  It travels through the
        // output buffer d_v_in between blocks (each lane re-reads exactly the 48 bytes it wrote itself a whole block earlier,
        // so ordinary single-thread memory ordering applies; the lines come back from L2) -- 6 KB of LDS per wave less, which
        // is what lets eight waves share a CU.
        gcp_wave_lds_sync();
        if (stamp_here) gcp_stamp(p.stamps, p.stamp_cap, 2, lane);

        // ---- A. recompute [vh | vf] = [vector_down ; vector_down_frames] v on the matrix cores; norms and frame scalars
        //         (element-wise) go to the weight-gradient GEMM's operand `ext` -- through the LDS tile `dext` (free until the end of
        //         step E), from which they leave as full rows behind the s_pre requests at the end of step C: no store in step A -----
        // GATE_PRE (sign-mask instantiations at width 128): the gate adjoint's weight fragments -- the four output tiles of a step are
        // 16 contiguous bytes per lane -- requested in front of step C as eight 16-byte loads; step D then starts without a memory
        // round trip (its 32 4-byte requests, issued tile by tile inside step D, cost 8.5 % of the launch with their MFMAs, GCP_CB_X & 16)
#ifdef GCP_CB_NO_GATE_PRE
        constexpr bool GATE_PRE = false;
#else
        constexpr bool GATE_PRE = SGN && NTG == 4;
#endif
        float4 gq[GATE_PRE ? NV : 1];
        float dgr[NV];
        if constexpr (GCP_CB_X & 4) {
#pragma unroll
            for (int r = 0; r < NV; ++r) dgr[r] = sg[r] + dvs[0][r];
        } else {
            gcp_xyz_acc u;
            gcp_vmm_down<10>(it.pack + S.offVA + lane, S.SVA, vi, vt + e * L.VS, hi, u);
            float f[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) f[i] = fr[e * 9 + i];  // (unconditional: only the channels [H, HF) use them, and they exist only with frames)
#pragma unroll
            for (int r = 0; r < NX; ++r) {
                const int x = gcp_crow(r, hi);
                const float u0 = u[0][r], u1 = u[1][r], u2 = u[2][r];
                if (x < H) {
                    const float nr = sqrtf(u0 * u0 + u1 * u1 + u2 * u2 + 1e-8f);
                    vht[e * L.HS + 3 * x + 0] = u0; vht[e * L.HS + 3 * x + 1] = u1; vht[e * L.HS + 3 * x + 2] = u2;
                    dext[e * L.DS + x] = nr + 1e-8f;
                } else if (x < HF) {
                    const int kk = x - H;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        float pr = f[3 * a + 0] * u0 + f[3 * a + 1] * u1 + f[3 * a + 2] * u2;
                        if (p.o.e3 && a == 1) {
                            e3t[e * 3 + kk] = pr < 0.f ? -1.f : 1.f;  // sign for the adjoint of |.|
                            pr = fabsf(pr);
                        }
                        dext[e * L.DS + H + 3 * kk + a] = pr;
                    }
                }
            }
            // ---- B. vu = vector_up(vh), B fragments = the registers just produced ------------------------------------
            gcp_xyz_acc vu;
            gcp_xyz_zero(vu);
            gcp_vmm_regs<NX>(it.pack + S.offVB + lane, SVB, u, vu);
            if constexpr (GATE_PRE) {
                const float* wg = it.pack + S.offD + (int64_t)lane * NTG;
#pragma unroll
                for (int r = 0; r < NV; ++r) gq[r] = *reinterpret_cast<const float4*>(wg + (int64_t)r * 64 * NTG);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- C. adjoint of the vector epilogue (gcpnet.py:364-391), element-wise: d(vector_up output), d(gate) ------
            const bool vres = p.o.vector_residual != 0;
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                const int o = gcp_crow(r, hi);
                const bool on = o < vi;
                const int oc = on ? o : 0;
                float u0 = vu[0][r], u1 = vu[1][r], u2 = vu[2][r];
                {  // (no branch around three LDS reads: read, then select)
                    const float t0 = vt[e * L.VS + 3 * oc + 0], t1 = vt[e * L.VS + 3 * oc + 1], t2 = vt[e * L.VS + 3 * oc + 2];
                    u0 += vres ? t0 : 0.f; u1 += vres ? t1 : 0.f; u2 += vres ? t2 : 0.f;
                }
                const float g0 = dvs[0][r], g1 = dvs[1][r], g2 = dvs[2][r];
                float d0 = g0, d1 = g1, d2 = g2, dg = 0.f;
                const float dot = g0 * u0 + g1 * u1 + g2 * u2;
                if (scalar_gate) {
                    const float s1 = sg[r];
                    d0 = g0 * s1; d1 = g1 * s1; d2 = g2 * s1;
                    dg = dot * s1 * (1.f - s1);
                } else if (p.o.vmode == GCP_VMODE_SELF_GATE) {
                    const float rs = sqrtf(u0 * u0 + u1 * u1 + u2 * u2 + 1e-8f);
                    const float nn = rs + 1e-8f;
                    const float a = gcp_actf<PWL>(it.act_v, ns_v, slope, nn), da = gcp_dactf<PWL>(it.act_v, ns_v, slope, nn);
                    const float coef = dot * da / rs;
                    d0 = g0 * a + coef * u0; d1 = g1 * a + coef * u1; d2 = g2 * a + coef * u2;
                }
                dgr[r] = (on && row_ok) ? dg : 0.f;
                if (on) {  // row-major copy: operand of the vector_up weight-gradient partial sums, re-read in step F
                    xt[e * L.VS + 3 * o + 0] = d0; xt[e * L.VS + 3 * o + 1] = d1; xt[e * L.VS + 3 * o + 2] = d2;
                }
            }
        }
        gcp_wave_lds_sync();
        // per-tile partial sums of the small vector weight gradients (v_mfma_f32_16x16x4_f32 over the tile's 96 (row, xyz)
        // pairs, operands = the row-major LDS copies); reduced over tiles by gcpnet_reduce_partials
        const int l16 = lane & 15, kq = lane >> 4;
        // (step st = 3 u + v covers the reduction index 12 u + 4 v + kq: row 4 u + (4 v + kq) / 3, component (4 v + kq) % 3 -- three
        // per-lane (row, component) pairs for the whole pass; the 24 operand pairs are read from LDS in batches of GCP_CB_TNB and
        // accumulated in three independent chains: the pass is a latency chain, 6.6 k -> ~2 k cycles with this)
        // One 16 x 16 output tile (M, N <= 16: vi <= 16 and H + 3 <= 16 in this kernel); the sums stay in registers and leave with
        // the next batch of stores (small_tn_store)
        auto small_tn = [&](const float* A, int ars, int ams, int ads, int M, const float* B, int brs, int bms, int bds, int N) -> f32x4 {
            int aoff[3], boff[3], rv[3];
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const int kk = 4 * v + kq, rr = kk / 3, d = kk - 3 * rr;
                rv[v] = rr; aoff[v] = rr * ars + d * ads; boff[v] = rr * brs + d * bds;
            }
            const bool mok = l16 < M, nok = l16 < N;
            const float* ap = A + (mok ? l16 : 0) * ams;
            const float* bp = B + (nok ? l16 : 0) * bms;
            f32x4 acc[3];
#pragma unroll
            for (int v = 0; v < 3; ++v) acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int part4 = 0; part4 < 24 / GCP_CB_TNB; ++part4) {
                constexpr int NB = GCP_CB_TNB, UB = NB / 3;
                float av[NB], bv[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int u = UB * part4 + i / 3, v = i % 3;
                    av[i] = ap[4 * u * ars + aoff[v]];
                    bv[i] = bp[4 * u * brs + boff[v]];
                }
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int u = UB * part4 + i / 3, v = i % 3;
                    const float a = (mok && r0 + 4 * u + rv[v] < rows) ? av[i] : 0.f;
                    acc[v % GCP_CB_TNC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, nok ? bv[i] : 0.f, acc[v % GCP_CB_TNC], 0, 0, 0);
                }
            }
            f32x4 r4;
#pragma unroll
            for (int r = 0; r < 4; ++r) r4[r] = (acc[0][r] + acc[1][r]) + acc[2][r];
            return r4;
        };
        auto small_tn_store = [&](const f32x4& v, int M, int N, float* out, bool transposed) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * kq + r;
                if (i < M && l16 < N) out[transposed ? l16 * M + i : i * N + l16] = v[r];
            }
        };
        // s_pre of this block: requested here, where few registers are live, and in flight under the first partial-sum pass
        f32x16 spr[NTG];
        if constexpr (SGN) {
        } else if constexpr (GCP_CB_X & 64) {
#pragma unroll
            for (int t = 0; t < NTG; ++t) spr[t] = dyr[t];
        } else {
            // (one address select per request instead of two code paths: a wave-uniform branch around loads makes hipcc wait for
            // them where the sides merge.  Tile-blocked: piece (t, q) of this lane, 1 KB per instruction; rows of [rows, so] otherwise)
            const int64_t off_rm = (int64_t)(row_ok ? row : 0) * so, off_tb = (int64_t)r0 * (32 * NTG) + 4 * lane;
            const bool tb = it.tb != 0;
#pragma unroll
            for (int t = 0; t < NTG; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j0 = 32 * t + 8 * q + 4 * hi;
                    const float4 a = *reinterpret_cast<const float4*>(it.s_pre + (tb ? off_tb + (t * 4 + q) * 256 : off_rm + min(j0, so - 4)));
                    const bool in = row_ok && j0 + 3 < so;
                    spr[t][4 * q] = in ? a.x : 0.f; spr[t][4 * q + 1] = in ? a.y : 0.f; spr[t][4 * q + 2] = in ? a.z : 0.f; spr[t][4 * q + 3] = in ? a.w : 0.f;
                }
        }
        // E_PRE (sign-mask instantiations): step E's first three stages of weight fragments are requested HERE, in front of the block's small
        // stores (requested at the top of step E they queue behind those stores)
#ifdef GCP_CB_NO_E_PRE
        constexpr bool E_PRE = false;
#else
        constexpr bool E_PRE = SGN && B6;
#endif
        constexpr int NTM = GCP_W6_TERMS;  // terms per weight element of the B6 image (two fp16 / three bf16: common.h)
        gcp_u32x4 EA0[NTM], EA1[NTM], EA2[NTM];
        if constexpr (E_PRE) {
            const float* wq0 = it.pack + S.offB6 + (int64_t)lane * 4;
#pragma unroll
            for (int tm = 0; tm < NTM; ++tm) {
                EA0[tm] = *reinterpret_cast<const gcp_u32x4*>(wq0 + tm * 256);
                EA1[tm] = *reinterpret_cast<const gcp_u32x4*>(wq0 + NTM * 256 + tm * 256);
                EA2[tm] = *reinterpret_cast<const gcp_u32x4*>(wq0 + 2 * NTM * 256 + tm * 256);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ... and BEHIND the requests the block's small stores.  (vmcnt retires loads and stores in issue order, and a store is only
        // retired once L2 has acknowledged it: a load requested behind a batch of stores cannot be used before all of them are
        // through.  The per-element stores of the norms that sat inside step A cost a full acknowledgement round trip each -- hipcc
        // waits vmcnt(0) at the merge points of that divergent code -- and another one in front of vector_up's fragments.)  The norms /
        // frame scalars leave as whole rows of [rows, EP] (a tile is one contiguous piece of 32 EP floats; columns past H + nf are the
        // stride padding: zeros), d(gate) from the registers, and the second partial-sum pass's sums of the block before
        {
            const int q4 = EP >> 2, n4 = min(rows - r0, GCP_TILE_ROWS) * q4, hn = H + S.nf;
            const unsigned magic = (unsigned)(((1ull << 32) + (unsigned)q4 - 1) / (unsigned)q4);  // idx / q4 for idx < 2^16
            float4* dst = reinterpret_cast<float4*>(it.ext + (int64_t)r0 * EP);
            float4 xv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // (EP <= 32: at most 256 pieces per tile)
                xv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (64 * j >= GCP_TILE_ROWS * q4) continue;  // (wave-uniform; a compile-time test in the HC instantiations)
                const int idx = min(lane + 64 * j, GCP_TILE_ROWS * q4 - 1);
                const int rr = (int)__umulhi((unsigned)idx, magic), c = 4 * (idx - rr * q4);
                const float* s4 = dext + rr * L.DS;
                const float a0 = s4[min(c, hn - 1)], a1 = s4[min(c + 1, hn - 1)], a2 = s4[min(c + 2, hn - 1)], a3 = s4[min(c + 3, hn - 1)];
                xv[j] = make_float4(c < hn ? a0 : 0.f, c + 1 < hn ? a1 : 0.f, c + 2 < hn ? a2 : 0.f, c + 3 < hn ? a3 : 0.f);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (64 * j < GCP_TILE_ROWS * q4 && lane + 64 * j < n4) dst[lane + 64 * j] = xv[j];
            if (scalar_gate) {
#pragma unroll
                for (int q = 0; q < VQ; ++q)
                    if (row_ok && 8 * q + 4 * hi < VOP)  // (VOP % 4 == 0, dgate 16-byte aligned: one guarded 16-byte store)
                        *reinterpret_cast<float4*>(it.dgate + (int64_t)row * VOP + 8 * q + 4 * hi) =
                            make_float4(dgr[4 * q], dgr[4 * q + 1], dgr[4 * q + 2], dgr[4 * q + 3]);
            }
        }
        float* part = it.w_part ? it.w_part + (int64_t)(r0 >> 5) * (vi * H + vi * HF) : nullptr;  // (GCP_TILE_ROWS == 32)
#ifdef GCP_CB_FINE2
        if (stamp_here) gcp_stamp(p.stamps, p.stamp_cap, 1, lane);
#endif
        f32x4 tn1 = f32x4{0.f, 0.f, 0.f, 0.f};
        if ((GCP_CB_X & 2) == 0 && part) { tn1 = small_tn(xt, L.VS, 3, 1, vi, vht, L.HS, 3, 1, H); small_tn_store(tn1, vi, H, part, false); }  // d vector_up[o, h] = sum dvu[row, o, d] vh[row, h, d]
        if (stamp_here) gcp_stamp(p.stamps, p.stamp_cap, 3, lane);
        CB_LAUNDER();

        // ---- D. ds_pre = d(s_out) * act_s'(s_pre) + act_v'(s_pre) * (Wg^T dgate), in the s_pre registers; the gate adjoint's
        //         B fragments are the d(gate) registers (section D is packed for that pairing) ----------------------------
        {
            // (NV == S.NOO by construction: both are 4 * ceil(vo / 8)); one output tile at a time, the next tile's NV weight
            // fragments requested before this tile's MFMAs
            const float* wg = it.pack + S.offD + (int64_t)lane * NTG;
            float ga[2][NV];
            if (scalar_gate && !GATE_PRE) {
#pragma unroll
                for (int r = 0; r < NV; ++r) ga[0][r] = wg[(int64_t)r * 64 * NTG];
            }
#pragma unroll
            for (int t = 0; t < NTG; ++t) {
                f32x16 gacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) gacc[r] = 0.f;
                if constexpr (GATE_PRE) {
                    if (scalar_gate && !(GCP_CB_X & 16)) {
#pragma unroll
                        for (int r = 0; r < NV; ++r) {
                            const float a = t == 0 ? gq[r].x : (t == 1 ? gq[r].y : (t == 2 ? gq[r].z : gq[r].w));
                            gacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dgr[r], gacc, 0, 0, 0);
                        }
                    }
                } else if (scalar_gate && !(GCP_CB_X & 16)) {
                    if (t + 1 < NTG) {
#pragma unroll
                        for (int r = 0; r < NV; ++r) ga[(t + 1) & 1][r] = wg[(int64_t)r * 64 * NTG + t + 1];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < NV; ++r) gacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[t & 1][r], dgr[r], gacc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float d;
                    if constexpr (SGN) {
                        const bool pos = ((sm[(16 * t + r) >> 5] >> ((16 * t + r) & 31)) & 1u) != 0;
                        d = dyr[t][r] * (pos ? dsc_inv : ns_s * dsc_inv);
                        if (scalar_gate) d += (pos ? 1.f : ns_v) * gacc[r];
                    } else {
                        const float sp = spr[t][r];
                        d = (dyr[t][r] * dsc_inv) * gcp_dactf<PWL>(it.act_s, ns_s, slope, sp);
                        if (scalar_gate) d += gcp_dactf<PWL>(it.act_v, ns_v, slope, sp) * gacc[r];
                    }
                    spr[t][r] = row_ok ? d : 0.f;
                }
            }
        }
#ifdef GCP_CB_FINE3  // (measurement build: stamps 0 / 1 of block 0 = end of step D / end of the ds_pre store)
        if (stamp_here) gcp_stamp(p.stamps, p.stamp_cap, 0, lane);
#endif
        // ds_pre leaves for the weight-gradient GEMM.  STORE_LATE (the sign-mask instantiations): behind step E -- the registers are step
        // E's B operands until then anyway -- and behind the requests of step F, so that no load of the block queues behind these sixteen
        // stores (vmcnt retires loads and stores in issue order; in front of step E they cost 12 % of the launch, GCP_CB_X & 32)
#ifdef GCP_CB_STORE_EARLY
        constexpr bool STORE_LATE = false;
#else
        constexpr bool STORE_LATE = SGN && B6;
#endif
        auto store_ds_pre = [&]() {
            if constexpr (GCP_CB_X & 32) {
            } else if (it.tb) {  // tile-blocked: sixteen 1 KB pieces straight from the registers (rows past the end are zeros, the buffer holds whole tiles)
                float4* bp = reinterpret_cast<float4*>(it.ds_pre + (int64_t)r0 * (32 * NTG)) + lane;
#pragma unroll
                for (int t = 0; t < NTG; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) bp[(t * 4 + q) * 64] = make_float4(spr[t][4 * q], spr[t][4 * q + 1], spr[t][4 * q + 2], spr[t][4 * q + 3]);
            } else {
                gcp_store_acc_rows_half_dense<NTG, PAD>(it.ds_pre, so, r0, rows, spr, stage, lane);  // (so == 32 NTG or PAD; 16-byte aligned: host checks)
            }
        };
        if constexpr (!STORE_LATE) store_ds_pre();
#ifdef GCP_CB_FINE3
        if (stamp_here) gcp_stamp(p.stamps, p.stamp_cap, 1, lane);
#endif
        CB_LAUNDER();

        // STORE_LATE: what step F requests from memory -- the two small products' weight fragments and the block's incoming d(V) state --
        // is requested inside step E's last stages, ahead of the ds_pre stores
        float vc_f[NV], vd_f[NX], stf[3][NV];
        auto request_f = [&]() {
            gcp_vmm_frags<NV>(it.pack + S.offVC + lane, S.SVC, vc_f);
            gcp_vmm_frags<NX>(it.pack + S.offVD + lane, SVD, vd_f);
        };
        // ---- E. d(s) += W^T ds_pre: 16 * NTG k-pair steps whose B operands are the ds_pre registers; the weight
        //         fragments rotate through three batches of 4 steps, requested two batches ahead and pinned there ------
        auto data_gemm = [&](auto nu_tag, const float* wq, f32x16* acc) {
            constexpr int NU = decltype(nu_tag)::value;
            WFragC<NU> A0[4], A1[4], A2[4];
            auto ld = [&](WFragC<NU>(&a)[4], int st0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u].load(wq + (int64_t)min(st0 + u, NTG * 16 - 1) * 64 * NUG);
            };
            ld(A0, 0);
            ld(A1, 4);
            ld(A2, 8);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < NTG * 4; ++b) {
                WFragC<NU>(&a)[4] = (b % 3 == 0) ? A0 : ((b % 3 == 1) ? A1 : A2);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int st = b * 4 + u;
#pragma unroll
                    for (int uu = 0; uu < NU; ++uu)
                        acc[uu] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].v[uu], spr[st / 16][st % 16], acc[uu], 0, 0, 0);
                }
                if ((b + 3) * 4 < NTG * 16) ld(a, (b + 3) * 4);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (GCP_CB_X & 128) {
#pragma unroll
            for (int t = 0; t < NTG; ++t) dyr[t] += spr[t];
        } else if constexpr (B6) {
            // 2 NTG slabs of K = 16 (eight ds_pre registers each, split into three bf16 terms on the fly) x (NTG + 1) output tiles
            // of the merged axis; one stage = (slab, tile) = three 16-byte weight fragments per lane and six MFMAs, fragments
            // requested three stages ahead
            constexpr int NKT = NTG + 1, NST = 2 * NTG * NKT;
            const float* wq = it.pack + S.offB6 + (int64_t)lane * 4;
            f32x16 accx;
#pragma unroll
            for (int r = 0; r < 16; ++r) accx[r] = 0.f;
#ifdef GCP_CB_E4  // (experiment: four fragment buffers, requests four stages ahead)
            constexpr int ENB = 4;
#else
            constexpr int ENB = 3;
#endif
            gcp_u32x4 A0[NTM], A1[NTM], A2[NTM], A3[NTM];
            auto ld = [&](gcp_u32x4(&a)[NTM], int sg) {
                const float* q = wq + (int64_t)(sg < NST ? sg : NST - 1) * (NTM * 256);
#pragma unroll
                for (int tm = 0; tm < NTM; ++tm) a[tm] = *reinterpret_cast<const gcp_u32x4*>(q + tm * 256);
            };
#if GCP_ARITH_F16X2
            // two fp16 terms (gcp_f16x2.h): this lane pair's ds_pre row scaled by 2^pa into fp16's range; d(s) -- which the products are
            // added to -- by 2^(pa + GCP_F16_WEXP) for the duration of the stages (exact both ways)
            float e_sc, e_isc;
            {
                float m = 0.f;
#pragma unroll
                for (int t = 0; t < NTG; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(spr[t][r]));
                m = fmaxf(m, __shfl_xor(m, 32));
                const int pa = gcp_f16_row_exp(m);
                e_sc = gcp_exp2i(pa); e_isc = gcp_exp2i(-(pa + GCP_F16_WEXP));
                const float up = gcp_exp2i(pa + GCP_F16_WEXP) * dsc_inv;  // (from the previous block's scale to this one's: exact)
                dsc_inv = e_isc;
#pragma unroll
                for (int t = 0; t < NTG; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dyr[t][r] *= up;
            }
#endif
            if constexpr (E_PRE) {
#pragma unroll
                for (int tm = 0; tm < NTM; ++tm) { A0[tm] = EA0[tm]; A1[tm] = EA1[tm]; A2[tm] = EA2[tm]; }
            } else {
                ld(A0, 0);
                ld(A1, 1);
                ld(A2, 2);
            }
            if constexpr (ENB == 4) ld(A3, 3);
            __builtin_amdgcn_sched_barrier(0);
            [[maybe_unused]] gcp_u32x4 bh, bm, bl;
#pragma unroll
            for (int sg = 0; sg < NST; ++sg) {
                const int j = sg / NKT, uu = sg % NKT;
                if (uu == 0) {
                    float x[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) x[i] = spr[j / 2][8 * (j % 2) + i];
#if GCP_ARITH_F16X2
                    gcp_f16x2_split8(x, e_sc, bh, bl);
#else
                    gcp_bf16x3_split8(x, bh, bm, bl);
#endif
                }
                gcp_u32x4(&a)[NTM] = ENB == 4 ? ((sg % 4 == 0) ? A0 : ((sg % 4 == 1) ? A1 : ((sg % 4 == 2) ? A2 : A3)))
                                               : ((sg % 3 == 0) ? A0 : ((sg % 3 == 1) ? A1 : A2));
#if GCP_ARITH_F16X2
                if (uu < NTG) dyr[uu < NTG ? uu : 0] = gcp_mfma_f16x3(a, bh, bl, dyr[uu < NTG ? uu : 0]);
                else accx = gcp_mfma_f16x3(a, bh, bl, accx);
#else
                if (uu < NTG) dyr[uu < NTG ? uu : 0] = gcp_mfma_bf16x6(a, bh, bm, bl, dyr[uu < NTG ? uu : 0]);
                else accx = gcp_mfma_bf16x6(a, bh, bm, bl, accx);
#endif
                if ((GCP_CB_X & 1) == 0 && sg + ENB < NST) ld(a, sg + ENB);
                if constexpr (STORE_LATE) {
                    if (sg == NST - ENB) request_f();  // (the first fragment buffer has retired)
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (STORE_LATE) {
                // (the state too -- into the registers the fragment buffers have just left -- so that step F reads nothing behind the stores)
                load_state(k == p.n - 1 ? p.d_v_out : p.d_v_in, stf, k == p.n - 1 && p.out_idx != nullptr);  // (a handed-over tile starts below n - 1)
                __builtin_amdgcn_sched_barrier(0);
                store_ds_pre();
            }
#if GCP_ARITH_F16X2
#pragma unroll
            for (int r = 0; r < 16; ++r) accx[r] *= e_isc;  // (d(s) keeps its factor: dsc_inv)
#endif
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int x = gcp_crow(r, hi);
                if (x < H + S.nf) dext[e * L.DS + x] = accx[r];
            }
        } else {
        data_gemm(std::integral_constant<int, NTG>{}, it.pack + S.offB + (int64_t)lane * NUG, dyr);
        {  // the tile of the merged axis that holds the norms and frame scalars
            f32x16 accx[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) accx[0][r] = 0.f;
            data_gemm(std::integral_constant<int, 1>{}, it.pack + S.offB + ((int64_t)xg * S.NS * 64 + lane) * NUG + xs, accx);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int x = gcp_crow(r, hi);
                if (x < H + S.nf) dext[e * L.DS + x] = accx[0][r];
            }
        }
        }
        if (stamp_here) gcp_stamp(p.stamps, p.stamp_cap, 4, lane);
        CB_LAUNDER();

        if (k == k_lo) gcp_store_acc_rows_half_dense<NTG, PAD>(p.d_s_in, so, r0, rows, dyr, stage, lane, dsc_inv);  // d(s) leaves the chip (or waits for the second half)
        gcp_wave_lds_sync();  // dext is visible; the first partial-sum pass is done with xt

        // ---- F. adjoint of the vector prologue: d[vh | vf] = Wu^T dvu + (norm and frame-scalar terms), d(V) += Wdf^T d[vh | vf] ---
        if constexpr (GCP_CB_X & 8) {
        } else {
            float dvu[3][NV];  // d(vector_up output), back from its row-major copy
#pragma unroll
            for (int r = 0; r < NV; ++r) {
                const int o = gcp_crow(r, hi);
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const float t = xt[e * L.VS + 3 * min(o, vi - 1) + d];
                    dvu[d][r] = o < vi ? t : 0.f;
                }
            }
            gcp_wave_lds_sync();  // ... before xt is overwritten below
            gcp_xyz_acc dacc;
            gcp_xyz_zero(dacc);
            if constexpr (STORE_LATE) gcp_vmm_arr_pre<NV>(vc_f, S.SVC, dvu, dacc);
            else gcp_vmm_arr<NV>(it.pack + S.offVC + lane, S.SVC, dvu, dacc);
            float f[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) f[i] = fr[e * 9 + i];  // (unconditional: only the channels [H, HF) use them, and they exist only with frames)
#pragma unroll
            for (int r = 0; r < NX; ++r) {
                const int x = gcp_crow(r, hi);
                if (x < H) {  // norm term: d|vh| / |vh| * vh
                    const float v0 = vht[e * L.HS + 3 * x + 0], v1 = vht[e * L.HS + 3 * x + 1], v2 = vht[e * L.HS + 3 * x + 2];
                    const float dn = dext[e * L.DS + x] / sqrtf(v0 * v0 + v1 * v1 + v2 * v2 + 1e-8f);
                    dacc[0][r] += dn * v0; dacc[1][r] += dn * v1; dacc[2][r] += dn * v2;
                } else if (x < HF) {
                    const int kk = x - H;
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        float ds = dext[e * L.DS + H + 3 * kk + a];
                        if (p.o.e3 && a == 1) ds *= e3t[e * 3 + kk];
                        a0 = fmaf(f[3 * a + 0], ds, a0);
                        a1 = fmaf(f[3 * a + 1], ds, a1);
                        a2 = fmaf(f[3 * a + 2], ds, a2);
                    }
                    dacc[0][r] = a0; dacc[1][r] = a1; dacc[2][r] = a2;
                }
                if (x < HF) {  // row-major copy for the vector_down(.frames) weight-gradient partial sums
                    xt[e * L.FS + 0 * HF + x] = dacc[0][r]; xt[e * L.FS + 1 * HF + x] = dacc[1][r]; xt[e * L.FS + 2 * HF + x] = dacc[2][r];
                }
            }
            gcp_xyz_acc dv;
            gcp_xyz_zero(dv);
            if constexpr (STORE_LATE) gcp_vmm_regs_pre<NX>(vd_f, SVD, dacc, dv);
            else gcp_vmm_regs<NX>(it.pack + S.offVD + lane, SVD, dacc, dv);
            float st[3][NV];  // ResGCP pass-through + this block's contribution -> the new state
            if constexpr (STORE_LATE) {
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int r = 0; r < NV; ++r) st[d][r] = stf[d][r];
            } else {
                load_state(k == p.n - 1 ? p.d_v_out : p.d_v_in, st, k == p.n - 1 && p.out_idx != nullptr);  // (a handed-over tile starts below n - 1)
            }
#pragma unroll
            for (int q = 0; q < VQ; ++q) {
                const int o0 = 8 * q + 4 * hi;
                float t[12];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int d = 0; d < 3; ++d)
                        t[3 * i + d] = st[d][4 * q + i] + dv[d][4 * q + i] + (p.o.vector_residual ? dvu[d][4 * q + i] : 0.f);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int d = 0; d < 3; ++d) dvs[d][4 * q + i] = (row_ok && o0 < vi) ? t[3 * i + d] : 0.f;  // -> the next block
                if (row_ok && o0 < vi) {  // (vi % 4 == 0: the lane's 48 bytes are three aligned 16-byte pieces, one guard for all)
                    float4* dp = reinterpret_cast<float4*>(p.d_v_in + (int64_t)row * 3 * vi + 3 * o0);
#pragma unroll
                    for (int j = 0; j < 3; ++j) dp[j] = make_float4(t[4 * j], t[4 * j + 1], t[4 * j + 2], t[4 * j + 3]);
                }
            }
        }
        if (stamp_here) gcp_stamp(p.stamps, p.stamp_cap, 5, lane);
        CB_LAUNDER();
        // ---- G. the second partial-sum pass, then the next block's (k-1) gates and vectors in ONE batch of requests.  (Requested
        //         before the pass, to fly under it, hipcc spilled the 24 destination registers -- and waits vmcnt(0) in front of
        //         each spill: six serial memory round trips per block.) ------------------------------------------------------------
        gcp_wave_lds_sync();
#ifdef GCP_CB_FINE2
        if (stamp_here) gcp_stamp(p.stamps, p.stamp_cap, 0, lane);
#endif
        // d [vector_down ; vector_down_frames][x, c] = sum v[row, c, d] [dvh | dvf][row, d, x], stored as [H + 3, vi]
        if ((GCP_CB_X & 2) == 0 && part) { tn2 = small_tn(vt, L.VS, 3, 1, vi, xt, L.FS, 1, HF, HF); small_tn_store(tn2, vi, HF, part + vi * H, true); }
        gcp_wave_lds_sync();
        CB_LAUNDER();
        if (k > k_lo) {
            CbVin vb;
            cb_vin_issue(vb, p.it[k - 1].v_in, vi, r0, rows, lane);
            gcp_load_gate<VQ>(scalar_gate ? p.it[k - 1].gate : nullptr, row, vi, hi, row_ok, vec_vo, sg);
            if constexpr (SGN) {
#pragma unroll
                for (int w = 0; w < NTG / 2; ++w) sm[w] = p.it[k - 1].sign[((int64_t)(r0 >> 5) * (NTG / 2) + w) * 64 + lane];
            }
            cb_vin_commit(vb, vt, L.VS, vi, r0, rows, lane);
        }
        if (stamp_here) gcp_stamp(p.stamps, p.stamp_cap, 6, lane);
    }
    if (tn2_part) {
        const int l16 = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * kq + r;
            if (i < vi && l16 < HF) tn2_part[l16 * vi + i] = tn2[r];
        }
    }
    gcp_stamp(p.stamps, p.stamp_cap, 7, lane);
    if (k_lo > 0) {
        // the state this workgroup leaves in d_s_in / d_v_in for the workgroup of the second half (cdna_hip_programming.md, Guideline
        // 16): every store of the wave retired, one agent-scope release, the flag from one lane
        typedef __attribute__((address_space(1))) unsigned gu32;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store((gu32*)(p.flags + (r0 >> 5)), 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

#undef p
#undef CB_RELOAD
#undef CB_LAUNDER

template <int NTG, int VQ, bool PWL>
int launch_cb(const ChainBwdParams& p, size_t lds_bytes, bool sgn, bool need_sgn, hipStream_t st) {
    // W^T ds_pre on the bf16 pipe (three-term split, six products) unless GCPNET_CHAIN_BWD_FP32_MFMA / gcpnet_debug_set_fp32_mfma select the
    // fp32 MFMA form of the same product (A/B switch)
    static const bool b6_env = getenv("GCPNET_CHAIN_BWD_FP32_MFMA") == nullptr;
    const bool pad = p.sh.so != 32 * NTG;  // (padded widths: bf16 form only)
    const bool b6 = pad || (g_gcp_fp32_mfma < 0 ? b6_env : g_gcp_fp32_mfma == 0);
    const dim3 grid((unsigned)(p.tiles + p.n_split));
    // need_sgn: a block whose forward stored only the sign mask (s_pre NULL) -- only the sign-mask instantiations (piecewise-linear
    // activations, bf16 form) can run it; the fp32 A/B switch with such a chain is the caller's error, never a read of a missing s_pre
    if (need_sgn && !(PWL && sgn && b6)) return GCPNET_E_BADARG;
    if constexpr (PWL) {
        if (sgn && b6) {  // sign masks instead of s_pre (the forward wrote them for every block)
            if (p.sh.H == 4 && p.sh.nf) {
                if (pad) hipLaunchKernelGGL((gcp2_chain_bwd_kernel<NTG, VQ, true, 4, true, true, true>), grid, dim3(GCP_WAVE), lds_bytes, st, p);
                else hipLaunchKernelGGL((gcp2_chain_bwd_kernel<NTG, VQ, true, 4, true, false, true>), grid, dim3(GCP_WAVE), lds_bytes, st, p);
            } else {
                if (pad) hipLaunchKernelGGL((gcp2_chain_bwd_kernel<NTG, VQ, true, 0, true, true, true>), grid, dim3(GCP_WAVE), lds_bytes, st, p);
                else hipLaunchKernelGGL((gcp2_chain_bwd_kernel<NTG, VQ, true, 0, true, false, true>), grid, dim3(GCP_WAVE), lds_bytes, st, p);
            }
            GCP_HIP_CHECK_LAUNCH();
            return 0;
        }
    }
    if (p.sh.H == 4 && p.sh.nf) {  // the shipped shape (V = 16, bottleneck 4): hidden channel count known at compile time
        if (pad) hipLaunchKernelGGL((gcp2_chain_bwd_kernel<NTG, VQ, PWL, 4, true, true>), grid, dim3(GCP_WAVE), lds_bytes, st, p);
        else if (b6) hipLaunchKernelGGL((gcp2_chain_bwd_kernel<NTG, VQ, PWL, 4, true>), grid, dim3(GCP_WAVE), lds_bytes, st, p);
        else hipLaunchKernelGGL((gcp2_chain_bwd_kernel<NTG, VQ, PWL, 4, false>), grid, dim3(GCP_WAVE), lds_bytes, st, p);
        GCP_HIP_CHECK_LAUNCH();
        return 0;
    }
    if (pad) hipLaunchKernelGGL((gcp2_chain_bwd_kernel<NTG, VQ, PWL, 0, true, true>), grid, dim3(GCP_WAVE), lds_bytes, st, p);
    else if (b6) hipLaunchKernelGGL((gcp2_chain_bwd_kernel<NTG, VQ, PWL, 0, true>), grid, dim3(GCP_WAVE), lds_bytes, st, p);
    else hipLaunchKernelGGL((gcp2_chain_bwd_kernel<NTG, VQ, PWL, 0, false>), grid, dim3(GCP_WAVE), lds_bytes, st, p);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

template <int NTG>
int launch_cb2(const ChainBwdParams& p, size_t lds_bytes, bool pwl, bool sgn, bool need_sgn, hipStream_t st) {
    if (p.sh.vi <= 8) return pwl ? launch_cb<NTG, 1, true>(p, lds_bytes, sgn, need_sgn, st) : launch_cb<NTG, 1, false>(p, lds_bytes, false, need_sgn, st);
    return pwl ? launch_cb<NTG, 2, true>(p, lds_bytes, sgn, need_sgn, st) : launch_cb<NTG, 2, false>(p, lds_bytes, false, need_sgn, st);
}

}  // namespace

// ---- tail split -------------------------------------------------------------------------------------------------------------
// A tile's chain is one wave's serial job of n blocks, two waves per SIMD: `slots` tiles run at a time.  tiles = F slots + R leaves a
// last round of R < slots waves whose CUs are half empty while it lasts a whole chain (measured, tools/chain_rows_sweep.py: 4 998
// tiles on 2 048 slots take the time of 3 rounds, not 2.44).  The launch therefore cuts x tiles in two -- blocks n - 1 .. a first,
// dispatched ahead of everything, the rest of those tiles last -- so that the pieces fill the slots the last round would leave idle.
// (x, a) by simulating the in-order dispatch (workgroup costs in block units + a fixed cost per workgroup); the order is only what
// makes it fast: the hand-over protocol (ChainBwdParams::flags) is correct under any dispatch order.
static double cb_simulate(int tiles, int n, int slots, int x, int a) {
    const double c0 = 0.4;  // prologue / epilogue of a workgroup, in block units
    std::priority_queue<double, std::vector<double>, std::greater<double>> free_at;
    for (int i = 0; i < slots; ++i) free_at.push(0.0);
    std::vector<double> first_done((size_t)x);
    double end = 0.0;
    auto run = [&](double len, double not_before) {
        double t = free_at.top();
        free_at.pop();
        if (t < not_before) t = not_before;
        t += len;
        free_at.push(t);
        if (t > end) end = t;
        return t;
    };
    for (int i = 0; i < x; ++i) first_done[(size_t)i] = run(c0 + (n - a), 0.0);
    for (int i = x; i < tiles; ++i) run(c0 + n, 0.0);
    for (int i = 0; i < x; ++i) run(c0 + a, first_done[(size_t)i]);
    return end;
}

// (n_split, k_split) for a launch; n_split == 0: no split
static void cb_plan_split(int tiles, int n, int slots, int* n_split, int* k_split) {
    *n_split = 0; *k_split = 0;
    if (n < 2 || tiles <= slots || slots <= 0) return;
    static std::mutex mu;
    static std::map<std::tuple<int, int, int>, std::pair<int, int>> memo;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_tuple(tiles, n, slots);
    auto hit = memo.find(key);
    if (hit == memo.end()) {
        const int R = tiles % slots;
        double best = cb_simulate(tiles, n, slots, 0, 0) * 0.97;  // a split must buy at least 3 %
        std::pair<int, int> pick(0, 0);
        if (R > 0)
            for (int x : {R, R / 2, std::min(tiles, R + slots / 2)})
                for (int a = 1; a < n && x > 0; ++a) {  // a = blocks left for the second half (k_split)
                    const double t = cb_simulate(tiles, n, slots, x, a);
                    if (t < best) { best = t; pick = std::make_pair(x, a); }
                }
        hit = memo.emplace(key, pick).first;
    }
    *n_split = hit->second.first; *k_split = hit->second.second;
}

// test hook (gcpnet_debug_force_chain_split): n_split < 0 = plan as above; otherwise the first min(n_split, tiles) tiles are cut at
// k_split whatever the size, `rev` != 0 reverses the workgroup order
static int g_cb_force[3] = {-1, 0, 0};
extern "C" void gcpnet_debug_force_chain_split(int n_split, int k_split, int rev) {
    g_cb_force[0] = n_split; g_cb_force[1] = k_split; g_cb_force[2] = rev;
}

static int cb_slots(size_t lds_bytes) {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, c = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
        cus = c;
    }
    const int by_lds = lds_bytes ? (int)((size_t)160 * 1024 / lds_bytes) : 8;
    return cus * std::min(8, std::max(1, by_lds));  // two waves per SIMD (__launch_bounds__), LDS permitting
}

static bool chain_bwd_shape_ok(const gcp2_weights_t& w0, const GcpShape& S) {
    if (S.NG != 1 || w0.si != w0.so || w0.vi != w0.vo || w0.vi <= 0 || (w0.si & 3) || S.NTG < 2 || S.NTS != S.NTG) return false;
    // register budget of the kernel: vi == vo <= 16 (two register quads per xyz component), H + 3 <= 16, H + 9 <= 32
    if ((w0.vi & 3) || w0.vi > 16 || !S.vmm || S.HF > 16 || S.H + S.nf > 32) return false;
    if (S.NUG < S.NTG) return false;
    return (size_t)cb_lds(S).total * sizeof(float) <= 64 * 1024;
}

// Returns GCPNET_E_UNSUPPORTED when the chain does not fit this kernel; the caller then runs the blocks one by one
// (gcpnet_gcp2_backward), passing the state through HBM.
int gcp2_chain_bwd_registers(int rows, const float* frames, int n, const gcp2_chain_bwd_item_t* items, const float* d_s_out,
                             const float* d_v_out, const int32_t* out_idx, const float* out_scale, float* d_s_in, float* d_v_in,
                             unsigned* flags, int n_flags, hipStream_t st) {
    const gcp2_weights_t& w0 = items[0].w;
    const GcpShape S = gcp_shape(w0.si, w0.vi, w0.so, w0.vo, w0.hidden, w0.use_frames);
    if (!chain_bwd_shape_ok(w0, S)) return GCPNET_E_UNSUPPORTED;
    ChainBwdParams p;
    p.rows = rows; p.frames = frames; p.o = items[0].o; p.n = n;
    p.d_s_out = d_s_out; p.d_v_out = d_v_out; p.d_s_in = d_s_in; p.d_v_in = d_v_in;
    p.out_idx = out_idx; p.out_scale = out_idx ? out_scale : nullptr;
    bool pwl = true, sgn = true, need_sgn = false;
    for (int k = 0; k < n; ++k) {
        const gcp2_chain_bwd_item_t& c = items[k];
        const gcp2_opts_t& o = c.o;
        if (c.w.si != w0.si || c.w.vi != w0.vi || c.w.so != w0.so || c.w.vo != w0.vo || c.w.hidden != w0.hidden ||
            c.w.use_frames != w0.use_frames || o.vmode != p.o.vmode || o.vector_residual != p.o.vector_residual ||
            o.e3 != p.o.e3 || o.slope != p.o.slope)
            return GCPNET_E_UNSUPPORTED;
        ChainItemB& it = p.it[k];
        it.pack = c.w.pack;
        it.v_in = c.v_in; it.s_pre = c.s_pre; it.gate = c.gate;
        it.ds_pre = c.sc.ds_pre; it.dgate = c.sc.dgate; it.ext = c.sc.ext; it.w_part = c.sc.w_part;
        it.act_s = o.act_s; it.act_v = o.act_v;
        it.tb = c.tb;
        it.sign = c.s_sign;
        sgn = sgn && c.s_sign != nullptr && (reinterpret_cast<uintptr_t>(c.s_sign) & 3) == 0;
        need_sgn = need_sgn || c.s_pre == nullptr;
        pwl = pwl && gcp_is_pwl(o.act_s) && gcp_is_pwl(o.act_v);
    }
    sgn = sgn && pwl;
    p.stamps = g_gcp_phase_buf; p.stamp_cap = g_gcp_phase_cap;
    p.sh = S;
    const size_t lds_bytes = (size_t)cb_lds(S).total * sizeof(float);
    if (lds_bytes > 64 * 1024) return GCPNET_E_UNSUPPORTED;
    p.tiles = gcp_cdiv(rows, GCP_TILE_ROWS);
    p.n_split = 0; p.k_split = 0; p.flags = nullptr; p.rev = 0;
    if (flags && n_flags > 0) {
        if (g_cb_force[0] >= 0) {
            p.n_split = n >= 2 ? std::min(g_cb_force[0], p.tiles) : 0;
            p.k_split = std::min(std::max(g_cb_force[1], 1), n - 1);
            p.rev = g_cb_force[2];
        } else {
            cb_plan_split(p.tiles, n, cb_slots(lds_bytes), &p.n_split, &p.k_split);
        }
        if (p.n_split > n_flags) p.n_split = 0;
        if (p.n_split > 0) {
            p.flags = flags;
            hipError_t err = hipMemsetAsync(flags, 0, (size_t)p.n_split * sizeof(unsigned), st);  // (a memset node under capture: zero at every replay)
            if (err != hipSuccess) return (int)err;
        }
    }
#ifdef GCP_CB_ONLY_SHIPPED  // (development builds: only the instantiations configs[1] runs -- tools/kres.py / the ISA tools in seconds)
    if (need_sgn && !sgn) return GCPNET_E_BADARG;
    if (sgn) hipLaunchKernelGGL((gcp2_chain_bwd_kernel<4, 2, true, 4, true, false, true>), dim3((unsigned)(p.tiles + p.n_split)), dim3(GCP_WAVE), lds_bytes, st, p);
    else hipLaunchKernelGGL((gcp2_chain_bwd_kernel<4, 2, true, 4, true>), dim3((unsigned)(p.tiles + p.n_split)), dim3(GCP_WAVE), lds_bytes, st, p);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
#else
    if (S.NTG == 2) return launch_cb2<2>(p, lds_bytes, pwl, sgn, need_sgn, st);
    return launch_cb2<4>(p, lds_bytes, pwl, sgn, need_sgn, st);
#endif
}

// How many flag words a launch of this chain on `rows` rows would use for its tail split (0: it would not split): the caller passes
// a buffer of at least that many to gcpnet_gcp2_chain_backward_split.
extern "C" int gcpnet_gcp2_chain_backward_flags(int rows, int n, int si, int vi, int so, int vo, int hidden, int use_frames) {
    if (rows <= 0 || n < 2) return 0;
    gcp2_weights_t w0{};
    w0.si = si; w0.vi = vi; w0.so = so; w0.vo = vo; w0.hidden = hidden; w0.use_frames = use_frames;
    const GcpShape S = gcp_shape(si, vi, so, vo, hidden, use_frames);
    if (!chain_bwd_shape_ok(w0, S)) return 0;
    int x = 0, a = 0;
    if (g_cb_force[0] >= 0) return std::min(g_cb_force[0], gcp_cdiv(rows, GCP_TILE_ROWS));
    cb_plan_split(gcp_cdiv(rows, GCP_TILE_ROWS), n, cb_slots((size_t)cb_lds(S).total * sizeof(float)), &x, &a);
    return x;
}

extern "C" int gcpnet_gcp2_chain_backward_ok(int si, int vi, int so, int vo, int hidden, int use_frames) {
    gcp2_weights_t w0{};
    w0.si = si; w0.vi = vi; w0.so = so; w0.vo = vo; w0.hidden = hidden; w0.use_frames = use_frames;
    return chain_bwd_shape_ok(w0, gcp_shape(si, vi, so, vo, hidden, use_frames)) ? 1 : 0;
}

static int chain_backward_checked(int rows, const float* frames, int n, const gcp2_chain_bwd_item_t* items, const float* d_s_out,
                                  const float* d_v_out, const int32_t* out_idx, const float* out_scale, float* d_s_in, float* d_v_in,
                                  void* stream, unsigned* flags = nullptr, int n_flags = 0) {
    if (rows < 0 || n < 1 || n > GCP_MAX_CHAIN || !items || !d_s_out || !d_v_out || !d_s_in || !d_v_in) return GCPNET_E_BADARG;
    for (int k = 0; k < n; ++k) {
        const gcp2_chain_bwd_item_t& c = items[k];
        if (!c.w.pack || !c.w.w_down || !c.w.w_up || !c.v_in || (!c.s_pre && !c.s_sign) || !c.sc.ds_pre || !c.sc.ext) return GCPNET_E_BADARG;
        if (c.w.use_frames && (!frames || !c.w.w_frames)) return GCPNET_E_BADARG;
        if (c.o.vmode == GCP_VMODE_SCALAR_GATE && (!c.gate || !c.sc.dgate)) return GCPNET_E_BADARG;
    }
    if (rows == 0) return 0;
    auto misaligned = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) != 0; };
    for (int k = 0; k < n; ++k)  // the deferred tile loads and the accumulator-layout accesses are 16-byte accesses
        if (misaligned(items[k].v_in) || misaligned(items[k].s_pre) || misaligned(items[k].gate) || misaligned(items[k].sc.ds_pre) ||
            misaligned(items[k].sc.dgate) || misaligned(items[k].sc.ext))
            return GCPNET_E_UNSUPPORTED;
    if (misaligned(d_s_out) || misaligned(d_v_out) || misaligned(d_s_in) || misaligned(d_v_in)) return GCPNET_E_UNSUPPORTED;
    if (flags && (reinterpret_cast<uintptr_t>(flags) & 3)) return GCPNET_E_BADARG;
    return gcp2_chain_bwd_registers(rows, frames, n, items, d_s_out, d_v_out, out_idx, out_scale, d_s_in, d_v_in, flags, n_flags,
                                    (hipStream_t)stream);
}

extern "C" int gcpnet_gcp2_chain_backward(int rows, const float* frames, int n, const gcp2_chain_bwd_item_t* items,
                                          const float* d_s_out, const float* d_v_out, float* d_s_in, float* d_v_in,
                                          void* stream) {
    return chain_backward_checked(rows, frames, n, items, d_s_out, d_v_out, nullptr, nullptr, d_s_in, d_v_in, stream);
}

extern "C" int gcpnet_gcp2_chain_backward_gathered(int rows, const float* frames, int n, const gcp2_chain_bwd_item_t* items,
                                                   const float* d_s_tab, const float* d_v_tab, const int32_t* out_idx,
                                                   const float* out_scale, float* d_s_in, float* d_v_in, void* stream) {
    if (!out_idx) return GCPNET_E_BADARG;
    return chain_backward_checked(rows, frames, n, items, d_s_tab, d_v_tab, out_idx, out_scale, d_s_in, d_v_in, stream);
}

// Both forms above with the tail split (see cb_plan_split): `flags` = n_flags >= gcpnet_gcp2_chain_backward_flags(...) words of device
// memory that only this launch touches while it runs (zeroed here, on the stream); out_idx NULL = the plain form.  Same results.
extern "C" int gcpnet_gcp2_chain_backward_split(int rows, const float* frames, int n, const gcp2_chain_bwd_item_t* items,
                                                const float* d_s_out, const float* d_v_out, const int32_t* out_idx,
                                                const float* out_scale, float* d_s_in, float* d_v_in, uint32_t* flags, int n_flags,
                                                void* stream) {
    return chain_backward_checked(rows, frames, n, items, d_s_out, d_v_out, out_idx, out_scale, d_s_in, d_v_in, stream, flags, n_flags);
}
