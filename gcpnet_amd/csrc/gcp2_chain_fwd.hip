// Register-resident chain of residual GCP2 blocks, x_k = x_{k-1} + GCP_k(x_{k-1}) (ResGCP, reference
// src/models/components/gcpnet.py:921-924), forward, one launch for the whole chain.
//
// The scalar state of a 32-row tile lives in the 32x32 MFMA ACCUMULATOR LAYOUT for the whole chain: register (t, r) of
// lane half `hi` holds column 32t + (r&3) + 8(r>>2) + 4hi of row lane&31.  That layout is at the same time
//   * the B fragment of a k-pair step over columns (j0, j0+4) -> scalar_out reads its input straight from registers
//     (weights packed for that pairing, section F of the packed image),
//   * the layout scalar_out produces -> x += act(s_pre) is a plain register add,
//   * the B fragment of the vector-gate GEMM,
// so the scalar path needs NO LDS at all.  The small vector Linears run on the matrix cores as well (vec_mfma.h): the vector
// state sits in a 32 x 3V LDS tile (B fragments of vector_down), everything downstream of it is register arithmetic in the C/D
// layout, and the H norms and 9 frame scalars -- the only inputs of scalar_out that are not state -- go through a 32 x 16 LDS
// tile as ordinary B fragments.  ~17 KB of LDS per wave (vector tile, frames, extras tile, a 4.6 KB transposition tile for
// full-line stores) lets two waves share a SIMD: one wave's VALU / LDS / store phases run under the other's MFMAs.
#include <cstdlib>

#include "common.h"
#include "tile_io.h"
#include "vec_mfma.h"
#include "gcp_bf16x3.h"
#include "gcp_f16x2.h"

// GCP_CF_X: measurement builds whose RESULTS ARE WRONG (tools/cf_variants.sh): bits remove one cost each so that its share of the
// launch time can be read under real contention.  1: scalar_out's fragments loaded once; 2: no s_pre store; 4: no s_out store;
// 8: no gate GEMM; 16: no vector prologue MFMAs; 32: no vector epilogue; 64: no v_out store; 128: no scalar_out MFMAs; 256: no sign words
#ifndef GCP_CF_X
#define GCP_CF_X 0
#endif

namespace {

struct ChainItemF {
    const float* pack;
    const float* b_scalar;
    const float* b_gate;
    float* s_out;
    float* v_out;
    float* s_pre;
    float* gate;
    int act_s, act_v;
    int s_out_tb, s_pre_tb;  // tile-blocked outputs (include/gcpnet_hip.h, gcp2_chain_item_t)
    unsigned* s_sign;        // optional sign mask of s_pre (include/gcpnet_hip.h)
};

// Optional head block in front of the chain: the first message GCP after project-then-gather, (se, vi0) -> (s, V) with the
// gathered sources entering as addend tables (gcpnet_gcp2_forward's s_add / v_add).  Not residual: its output IS the chain's
// initial state, so fusing it saves the state's round trip through HBM; with n == 0 the kernel is that block alone.
struct HeadParams {
    const float* e_in;   // [rows, se]
    const float* xi_in;  // [rows, vi0, 3]
    gcp_concat_t s_add, v_add;
    ChainItemF it;
    GcpShape sh;
};

struct ChainParams {
    int rows;
    const float* s0;
    const float* v0;
    const float* frames;
    gcp2_opts_t o;
    int n;
    ChainItemF it[GCP_MAX_CHAIN];
    unsigned long long* stamps;
    long long stamp_cap;
    GcpShape sh;
};

struct ChainParamsH : ChainParams {  // kernel argument of the HEAD instantiations
    HeadParams hd;
};

struct ChainLds {
    int VS, XS;
    int o_vt, o_fr, o_ext, o_ust, o_stage, total;
};

__host__ __device__ inline ChainLds chain_lds(const GcpShape& s, const GcpShape* h = nullptr) {
    ChainLds l;
    const int vmax = h && h->vi > s.vi ? h->vi : s.vi;
    l.VS = gcp_odd(3 * vmax);
    int xw = gcp_round_up(s.H + s.nf, 2);
    if (h && h->KP - h->si > xw) xw = h->KP - h->si;  // the head reads its k padding from the extras tile too
    l.XS = gcp_odd(xw);
    l.o_vt = 0;
    l.o_fr = l.o_vt + 32 * l.VS;
    l.o_ext = l.o_fr + 32 * 9;
    l.o_ust = l.o_ext + 32 * l.XS;
    const int svb = h && h->SVB > s.SVB ? h->SVB : s.SVB;
    l.o_stage = l.o_ust + svb * 3 * 64;  // (ust: vector_down outputs parked between the vector prologue and epilogue)
    int stage = GCP_ACC_STAGE_FLOATS;  // the head's scalar input tile shares the transposition tile's space
    if (h && 32 * gcp_odd(h->si) > stage) stage = 32 * gcp_odd(h->si);
    l.total = l.o_stage + stage;
    return l;
}

template <int N>
struct WF;
template <>
struct WF<2> {
    float v[2];
    __device__ __forceinline__ void load(const float* p) {
        float2 t = *reinterpret_cast<const float2*>(p);
        v[0] = t.x; v[1] = t.y;
    }
};
template <>
struct WF<4> {
    float v[4];
    __device__ __forceinline__ void load(const float* p) {
        float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
};

// s_pre / s_out rows of a tile: the branch-free form when the destination allows 16-byte stores (tile_io.h)
template <int NT>
__device__ __forceinline__ void gcp_store_acc_rows_any(float* __restrict__ dst, int so, int r0, int rows, const f32x16 (&acc)[NT],
                                                       float* stage, int lane) {
    if ((so & 3) == 0 && gcp_aligned16(dst)) gcp_store_acc_rows_dense<NT>(dst, so, so, r0, rows, acc, stage, lane);
    else gcp_store_acc_rows<NT>(dst, so, 0, so, r0, rows, acc, stage, lane);
}

// The same tile in the tile-blocked layout: register quad (t, q) of every lane is one 16-byte piece, the 64 pieces of a (t, q) are
// 1 KB contiguous -- one store instruction = eight full lines, no LDS transposition, no row or column test (the buffer holds whole
// tiles of padded width 32 NT).
template <int NT>
__device__ __forceinline__ void gcp_store_acc_tb(float* __restrict__ dst, int r0, const f32x16 (&acc)[NT], int lane) {
    float4* bp = reinterpret_cast<float4*>(dst + (int64_t)r0 * (32 * NT)) + lane;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) bp[(t * 4 + q) * 64] = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
}

// NT = 32-wide tiles of the scalar state (so <= 32 * NT); PWL as in gcp2_fwd.hip.
template <bool HEAD>
struct ChainArg { typedef ChainParams type; };
template <>
struct ChainArg<true> { typedef ChainParamsH type; };

// (head parameters of a kernel argument; the plain instantiations never evaluate the result)
__device__ __forceinline__ const HeadParams& head_of(const ChainParamsH& p) { return p.hd; }
__device__ __forceinline__ const HeadParams& head_of(const ChainParams& p) { return *reinterpret_cast<const HeadParams*>(&p); }

// HEAD: a head block (HeadParams) runs first; ONLY: ... and nothing else (n == 0), so that the compiler sees one block shape.
// HC: hidden vector channels of the chain blocks as a compile-time constant (0 = run-time; plain instantiations only).
// F6: scalar_out over the state and the gate Linear on the bf16 matrix pipe, both operands as three bf16 terms, six products
// (gcp_bf16x3.h: exact to fp32 round-off); plain instantiations only.
template <int NT, bool PWL, bool HEAD, bool ONLY = false, int HC = 0, bool F6 = false>
__global__ __launch_bounds__(GCP_WAVE, 2) void gcp2_chain_fwd_kernel(typename ChainArg<HEAD>::type p_kernarg) {
    static_assert(!(HEAD && HC), "the head block has its own shape");
    // The parameters are read through the kernarg segment pointer, laundered at the top of every block (as in gcp2_chain_bwd.hip):
    // the uniform values are re-loaded (s_load) per block instead of staying live in SGPRs across the whole chain loop, where
    // they did not fit -- 169 SGPRs spilled into VGPR lanes, 947 v_readlane_b32 among the 4 461 VALU instructions of a block.
    typedef typename ChainArg<HEAD>::type ParamT;
    typedef const __attribute__((address_space(4))) ParamT* Karg;
    Karg kp = (Karg)__builtin_amdgcn_kernarg_segment_ptr();
#define p (*(const ParamT*)kp)
    static_assert(!(HEAD && F6), "the bf16 forms exist for the chain blocks");
    constexpr int NXR = HC ? 4 * ((HC + 3 + 7) / 8) : 16;  // registers that can hold a [vh | vf] channel
    constexpr int NVR = HC ? 8 : 16;  // registers that can hold an output vector channel (the HC instantiations: vo <= 16)
    extern __shared__ __attribute__((aligned(16))) float lds[];
#define HD (head_of(p))
    GcpShape S;
    ChainLds L;
    int lane = threadIdx.x;
    int e = lane & 31, hi = lane >> 5;
    const int r0 = blockIdx.x * GCP_TILE_ROWS;
    int rows, row;
    bool row_ok;
    float *vt, *fr, *ext, *ust, *stage;
    int vi, so, vo, H, NX;
    float slope;
    bool scalar_gate, vec_so;
#define CF_RELOAD()                                                                                               \
    do {                                                                                                          \
        S = p.sh;                                                                                                 \
        L = chain_lds(S, HEAD ? &HD.sh : nullptr);                                                                \
        rows = p.rows;                                                                                            \
        row = r0 + e;                                                                                             \
        row_ok = row < rows;                                                                                      \
        vt = lds + L.o_vt; fr = lds + L.o_fr; ext = lds + L.o_ext; ust = lds + L.o_ust; stage = lds + L.o_stage;  \
        vi = S.vi; so = S.so; vo = S.vo; H = HC ? HC : S.H;                                                       \
        NX = gcp_round_up(H + S.nf, 2) / 2; /* k-pair steps over the norms / frame scalars */                     \
        slope = p.o.slope;                                                                                        \
        scalar_gate = p.o.vmode == GCP_VMODE_SCALAR_GATE;                                                         \
        vec_so = (so & 3) == 0;                                                                                   \
    } while (0)
    CF_RELOAD();
    // ---- the tile: vectors + frames into LDS, scalars straight into the accumulator layout ---------------------------
    f32x16 xs[NT];
    float* et = stage;  // HEAD: the head's scalar inputs [32][ES] (consumed before the first row-wise store needs the tile)
    int ES = HEAD ? gcp_odd(HD.sh.si) : 1;
    if constexpr (HEAD) {
        GcpSegBuf<8> vb0, eb0;
        gcp_seg_issue(vb0, HD.xi_in, nullptr, 3 * HD.sh.vi, r0, rows, vt, L.VS, 0, lane);
        gcp_seg_issue(eb0, HD.e_in, nullptr, HD.sh.si, r0, rows, et, ES, 0, lane);
        if (S.nf) gcp_load_frames(p.frames, r0, rows, fr, lane);
        gcp_seg_commit(vb0, vt, L.VS, 0);
        gcp_seg_commit(eb0, et, ES, 0);
    } else {
        GcpSegBuf<8> vb0;
        gcp_seg_issue(vb0, p.v0, nullptr, 3 * vi, r0, rows, vt, L.VS, 0, lane);
        if (S.nf) gcp_load_frames(p.frames, r0, rows, fr, lane);
        gcp_load_acc_layout<NT, false>(p.s0, row, so, 0, hi, row_ok, vec_so, xs);
        gcp_seg_commit(vb0, vt, L.VS, 0);
    }
    for (int i = H + S.nf + hi; i < 2 * NX; i += 2) ext[e * L.XS + i] = 0.f;

    const int n_blocks = ONLY ? 0 : p.n;
    // PRE (the compile-time-hidden-width instantiations of the chain blocks): the weight fragments of the two small vector products
    // are requested AHEAD of the block's s_pre / s_out / v_out stores -- vector_up's before this block's stores, the next block's
    // vector_down ones with them -- because vmcnt retires loads and stores in issue order: a fragment requested behind a batch of
    // stores is usable only once L2 has acknowledged every one of them (the same reordering of the backward
    // kernel was worth 3 %)
#ifdef GCP_CF_NO_PRE  // (measurement build: the requests where they were, behind the stores)
    constexpr bool PRE = false;
#else
    constexpr bool PRE = HC > 0 && !HEAD;
#endif
    constexpr int NVA = PRE ? 8 : 4, NVB = PRE ? 4 * ((HC + 7) / 8) : 4;  // k-pair steps of vector_down (vi <= 16), registers of vector_up's input
    float va_pre[NVA], vb_pre[NVB];
    if constexpr (PRE) gcp_vmm_frags<NVA>(p.it[0].pack + S.offVA + lane, S.SVA, va_pre);
    for (int ci = HEAD ? -1 : 0; ci < n_blocks; ++ci) {
        asm volatile("" : "+v"(lane), "+v"(e), "+v"(hi), "+s"(kp));  // keep per-lane addresses AND the uniform parameters from being hoisted and spilled
        CF_RELOAD();
        et = stage;
        ES = HEAD ? gcp_odd(HD.sh.si) : 1;
        const bool head = HEAD && (ONLY || ci < 0);  // (compile-time false in the plain instantiations, true with ONLY)
        const ChainItemF& it = head ? HD.it : p.it[ci];
        const GcpShape& B = head ? HD.sh : S;  // the block's own shape: dims, step counts, section offsets of its pack
        const int Hb = HC ? HC : B.H, vib = B.vi;
        const int HFb = HC ? HC + 3 : B.HF, SVBb = HC ? 4 * ((HC + 7) / 8) : B.SVB;
        const float ns_s = gcp_neg_slope(it.act_s, slope), ns_v = gcp_neg_slope(it.act_v, slope);
        if (ci == n_blocks - 1) gcp_stamp(p.stamps, p.stamp_cap, 0, lane);
        gcp_wave_lds_sync();  // the previous block's vector tile update

        // HEAD: the rows of the pre-projected tables this row gathers are requested first (accumulator layout for the scalar
        // addends, four channels x xyz at a time for the vector ones) and added to the products below
        f32x16 acc[NT];
        float qa[4][12];
        // scalar_out's bias in the accumulator layout (columns 32 t + 8 q + 4 hi .. + 3 of a lane are contiguous: one 16-byte load
        // per register quad).  Plain blocks request it here, a whole phase before the GEMM that starts from it; the head block,
        // whose accumulators already collect the gathered addends, where it is added.
        f32x16 bias[NT];
        auto load_bias = [&]() {
            const bool vec_b = vec_so && ((reinterpret_cast<uintptr_t>(it.b_scalar) & 15) == 0);
            gcp_request_acc_layout<NT>(it.b_scalar, 0, so, 0, hi, true, vec_b, bias);
        };
        if constexpr (!HEAD) load_bias();
        if (head) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 12; ++j) qa[q][j] = 0.f;
            const int rc = min(row, rows - 1);
            for (int k = 0; k < HD.s_add.n; ++k) {
                const int32_t* ix = HD.s_add.idx[k];
                const int64_t src = ix ? (int64_t)ix[rc] : (int64_t)rc;
                gcp_load_acc_layout<NT, true>(HD.s_add.ptr[k], src, so, 0, hi, true, vec_so, acc);
            }
            const int HFP = gcp_round_up(B.HF, 4);
            for (int k = 0; k < HD.v_add.n; ++k) {
                const int32_t* ix = HD.v_add.idx[k];
                const float* trow = HD.v_add.ptr[k] + (ix ? (int64_t)ix[rc] : (int64_t)rc) * 3 * HFP;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int x0 = 8 * q + 4 * hi;
                    const bool on = x0 < B.HF;
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        const float4 t4 = *reinterpret_cast<const float4*>(trow + d * HFP + (on ? x0 : 0));
                        qa[q][4 * d + 0] += on ? t4.x : 0.f; qa[q][4 * d + 1] += on ? t4.y : 0.f;
                        qa[q][4 * d + 2] += on ? t4.z : 0.f; qa[q][4 * d + 3] += on ? t4.w : 0.f;
                    }
                }
            }
        }

        // ---- vector prologue on the matrix cores: [vh | vf] = [vector_down ; vector_down_frames] v, then (element-wise,
        //      in registers) the norms of vh and the projections of vf onto the row's frame -> the 32 x XS extras tile ------
        {
            gcp_xyz_acc u;
            if constexpr (GCP_CF_X & 16) gcp_xyz_zero(u);
            else if constexpr (PRE) gcp_vmm_down_pre<NVA>(va_pre, B.SVA, vib, vt + e * L.VS, hi, u);
            else gcp_vmm_down<16>(it.pack + B.offVA + lane, B.SVA, vib, vt + e * L.VS, hi, u);
            if (head) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int d = 0; d < 3; ++d)
#pragma unroll
                        for (int i = 0; i < 4; ++i) u[d][4 * q + i] += qa[q][4 * d + i];
                for (int i = Hb + B.nf + hi; i < B.KP - B.si; i += 2) ext[e * L.XS + i] = 0.f;  // the head's k padding
            }
            if (ci == n_blocks - 1) gcp_stamp(p.stamps, p.stamp_cap, 1, lane);
            float f[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) f[i] = S.nf ? fr[e * 9 + i] : 0.f;
#pragma unroll
            for (int r = 0; r < NXR; ++r) {
                const int x = gcp_crow(r, hi);
                const float u0 = u[0][r], u1 = u[1][r], u2 = u[2][r];
                if (r < SVBb) {  // parked for vector_up in the epilogue (wave-uniform guard)
                    ust[(r * 3 + 0) * 64 + lane] = u0; ust[(r * 3 + 1) * 64 + lane] = u1; ust[(r * 3 + 2) * 64 + lane] = u2;
                }
                if (x < Hb) {
                    ext[e * L.XS + x] = sqrtf(u0 * u0 + u1 * u1 + u2 * u2 + 1e-8f) + 1e-8f;
                } else if (x < HFb) {
                    const int k = x - Hb;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        float pr = f[3 * a + 0] * u0 + f[3 * a + 1] * u1 + f[3 * a + 2] * u2;
                        if (p.o.e3 && a == 1) pr = fabsf(pr);
                        ext[e * L.XS + Hb + 3 * k + a] = pr;
                    }
                }
            }
        }
        gcp_wave_lds_sync();

        if (ci == n_blocks - 1) gcp_stamp(p.stamps, p.stamp_cap, 2, lane);
        // ---- scalar_out: acc = b + W[:, state] x^T + W[:, extras] ext^T --------------------------------------------------
        f32x16 gacc;     // vector-gate pre-activations (initialised with the bias while the extras tile is multiplied)
        float gwa[16];   // first batch of the gate GEMM's weight fragments
        gcp_u32x4 G0[3], G1[3];  // (F6: the first two slabs' fragments)
#pragma unroll
        for (int r = 0; r < 16; ++r) { gacc[r] = 0.f; gwa[r] = 0.f; }
        if constexpr (HEAD) load_bias();
        gcp_mask_acc_layout<NT>(so, 0, hi, true, vec_so && ((reinterpret_cast<uintptr_t>(it.b_scalar) & 15) == 0), bias);
        // F6 with two fp16 terms (gcp_f16x2.h): the state row of this lane pair scaled by 2^pa into fp16's range, the accumulators by
        // 2^(pa + GCP_F16_WEXP) while the state products are added (the bias enters scaled -- exact --, the inverse follows the stages)
        float xsc = 1.f, acc_sc = 1.f, acc_isc = 1.f;
        if constexpr (F6 && GCP_ARITH_F16X2) {
            float m = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(xs[t][r]));
            m = fmaxf(m, __shfl_xor(m, 32));  // (the other half of the row's columns)
            const int pa = gcp_f16_row_exp(m);
            xsc = gcp_exp2i(pa); acc_sc = gcp_exp2i(pa + GCP_F16_WEXP); acc_isc = gcp_exp2i(-(pa + GCP_F16_WEXP));
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = head ? acc[t][r] + bias[t][r] : bias[t][r] * acc_sc;
#ifdef GCP_FWD_FINE
        if (ci == n_blocks - 1) gcp_stamp(p.stamps, p.stamp_cap, 3, lane);
#endif
        if (head) {
            // the head's own scalar inputs: B fragments from its LDS tile, weights from section A of its pack
            const float* wa0 = it.pack + B.offA + (int64_t)lane * NT;
            const int KE = B.si / 2;
            for (int k0 = 0; k0 < KE; k0 += 8) {
                WF<NT> a[8];
                float bv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int kk = min(k0 + u, KE - 1);
                    a[u].load(wa0 + (int64_t)kk * 64 * NT);
                    bv[u] = et[e * ES + 2 * kk + hi];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (k0 + u < KE) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].v[t], bv[u], acc[t], 0, 0, 0);
                    }
            }
        }
        {
            const float* wf = it.pack + S.offF + (int64_t)lane * NT;  // [step][64][NT]
            constexpr int U = 4;
            if constexpr (F6) {
                // 2 NT slabs of K = 16 (eight state registers each, split once per slab) x NT output tiles; one stage = three
                // 16-byte weight fragments per lane and six MFMAs, fragments requested three stages ahead
                constexpr int NSTG = 2 * NT * NT, NTM = GCP_W6_TERMS;
                const float* wq = it.pack + S.offF6 + (int64_t)lane * 4;
                gcp_u32x4 A0[NTM], A1[NTM], A2[NTM];
                auto ld6 = [&](gcp_u32x4(&a)[NTM], int sg) {
                    const float* q = wq + (int64_t)(sg < NSTG ? sg : NSTG - 1) * (NTM * 256);
#pragma unroll
                    for (int tm = 0; tm < NTM; ++tm) a[tm] = *reinterpret_cast<const gcp_u32x4*>(q + tm * 256);
                };
                ld6(A0, 0);
                ld6(A1, 1);
                ld6(A2, 2);
                __builtin_amdgcn_sched_barrier(0);
                [[maybe_unused]] gcp_u32x4 bh, bm, bl;
#pragma unroll
                for (int sg = 0; sg < NSTG; ++sg) {
                    const int j2 = sg / NT, t = sg % NT;
                    if (t == 0) {
                        float x[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) x[i] = xs[j2 / 2][8 * (j2 % 2) + i];
#if GCP_ARITH_F16X2
                        gcp_f16x2_split8(x, xsc, bh, bl);
#else
                        gcp_bf16x3_split8(x, bh, bm, bl);
#endif
                    }
                    gcp_u32x4(&a)[NTM] = (sg % 3 == 0) ? A0 : ((sg % 3 == 1) ? A1 : A2);
                    if constexpr ((GCP_CF_X & 128) == 0) {
#if GCP_ARITH_F16X2
                        acc[t] = gcp_mfma_f16x3(a, bh, bl, acc[t]);
#else
                        acc[t] = gcp_mfma_bf16x6(a, bh, bm, bl, acc[t]);
#endif
                    }
                    if ((GCP_CF_X & 1) == 0 && sg + 3 < NSTG) ld6(a, sg + 3);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (GCP_ARITH_F16X2) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[t][r] *= acc_isc;
                }
            } else if (!head) {
            WF<NT> A0[U], A1[U], A2[U];
            auto ld = [&](WF<NT>(&a)[U], int st0) {
#pragma unroll
                for (int u = 0; u < U; ++u) a[u].load(wf + (int64_t)min(st0 + u, NT * 16 - 1) * 64 * NT);
            };
#ifdef GCP_FWD_FINE
            if (ci == n_blocks - 1) gcp_stamp(p.stamps, p.stamp_cap, 4, lane);
#endif
            ld(A0, 0);
            ld(A1, U);
            ld(A2, 2 * U);
            __builtin_amdgcn_sched_barrier(0);
            // NT * 16 steps, fully unrolled so that the state registers are addressed statically; the weight fragments
            // rotate through three batches of U steps (requested two batches ahead)
#pragma unroll
            for (int b = 0; b < NT * 16 / U; ++b) {
                WF<NT>(&a)[U] = (b % 3 == 0) ? A0 : ((b % 3 == 1) ? A1 : A2);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int st = b * U + u;
                    const float bv = xs[st / 16][st % 16];
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].v[t], bv, acc[t], 0, 0, 0);
                }
                if ((b + 3) * U < NT * 16) ld(a, (b + 3) * U);
                __builtin_amdgcn_sched_barrier(0);  // keep the prefetch HERE: hipcc otherwise sinks each load to its use
#ifdef GCP_FWD_FINE
                if (b == 0 && ci == n_blocks - 1) gcp_stamp(p.stamps, p.stamp_cap, 5, lane);
                if (b == 7 && ci == n_blocks - 1) gcp_stamp(p.stamps, p.stamp_cap, 6, lane);
                if (b == 15 && ci == n_blocks - 1) gcp_stamp(p.stamps, p.stamp_cap, 7, lane);
#endif
            }
            }
#ifndef GCP_FWD_FINE
            if (ci == n_blocks - 1) gcp_stamp(p.stamps, p.stamp_cap, 3, lane);
#endif
            // (the gate GEMM's first weight fragments and its bias are requested here, one phase ahead)
            if (scalar_gate) {
                const float* wg0 = it.pack + B.offC + lane;
                if constexpr (F6) {
                    const float* q = it.pack + S.offC6 + (int64_t)lane * 4;
#pragma unroll
                    for (int tm = 0; tm < 3; ++tm) {
                        G0[tm] = *reinterpret_cast<const gcp_u32x4*>(q + tm * 256);
                        G1[tm] = *reinterpret_cast<const gcp_u32x4*>(q + 768 + tm * 256);
                    }
                }
                float bgv[16];  // unconditional (clamped) loads, ALL requested before the first select (a guarded load, or one
#pragma unroll          //  selected right behind its request, is waited for on the spot: sixteen serial round trips to L2)
                for (int r = 0; r < 16; ++r) {
                    if constexpr (!F6) gwa[r] = wg0[(int64_t)r * 64];
                    bgv[r] = it.b_gate[min(gcp_crow(r, hi), vo - 1)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 16; ++r) gacc[r] = gcp_crow(r, hi) < vo ? bgv[r] : 0.f;
            }
            // norms and frame scalars: ordinary B fragments from the 32 x 16 LDS tile, weights from section A
            const float* wa = it.pack + B.offA + (int64_t)lane * NT;
            const int kk0 = B.si / 2;
            const int NXb = head ? (B.KP - B.si) / 2 : NX;
            for (int x0 = 0; x0 < NXb; x0 += 8) {  // fragments of eight steps requested together (one L2 round trip, not eight)
                WF<NT> a[8];
                float bv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int x = min(x0 + u, NXb - 1);
                    a[u].load(wa + (int64_t)(kk0 + x) * 64 * NT);
                    bv[u] = ext[e * L.XS + 2 * x + hi];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (x0 + u < NXb) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].v[t], bv[u], acc[t], 0, 0, 0);
                    }
            }
        }

#ifndef GCP_FWD_FINE
        if (ci == n_blocks - 1) gcp_stamp(p.stamps, p.stamp_cap, 4, lane);
#endif
        // ---- vector gate Linear, B fragments = the accumulator registers ---------------------------------------------------
        if (F6 && scalar_gate && !(GCP_CF_X & 8)) {
            const float* q6 = it.pack + S.offC6 + (int64_t)lane * 4;
#pragma unroll
            for (int j2 = 0; j2 < 2 * NT; ++j2) {
                gcp_u32x4(&g)[3] = (j2 & 1) ? G1 : G0;
                float x[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) x[i] = gcp_actf<PWL>(it.act_v, ns_v, slope, acc[j2 / 2][8 * (j2 % 2) + i]);
                [[maybe_unused]] gcp_u32x4 bh, bm, bl;
                gcp_bf16x3_split8(x, bh, bm, bl);
                gacc = gcp_mfma_bf16x6(g, bh, bm, bl, gacc);
                if (j2 + 2 < 2 * NT) {
#pragma unroll
                    for (int tm = 0; tm < 3; ++tm) g[tm] = *reinterpret_cast<const gcp_u32x4*>(q6 + (int64_t)(j2 + 2) * 768 + tm * 256);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (scalar_gate) {
            const float* wg = it.pack + B.offC + lane;
            float(&wa)[16] = gwa;
            float wb[16];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float(&cur)[16] = (t & 1) ? wb : wa;
                float(&nxt)[16] = (t & 1) ? wa : wb;
                if (t + 1 < NT) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) nxt[r] = wg[(int64_t)((t + 1) * 16 + r) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);  // next tile's fragments stay in flight under this tile's MFMAs
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    gacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[r], gcp_actf<PWL>(it.act_v, ns_v, slope, acc[t][r]), gacc, 0, 0, 0);
            }
        }
#ifndef GCP_FWD_FINE
        if (ci == n_blocks - 1) gcp_stamp(p.stamps, p.stamp_cap, 5, lane);
#endif
        if constexpr (PRE) {  // (clamped index, no branch around the requests: a divergent merge point would cost a vmcnt(0))
            gcp_vmm_frags<NVB>(it.pack + B.offVB + lane, SVBb, vb_pre);
            gcp_vmm_frags<NVA>(p.it[min(ci + 1, n_blocks - 1)].pack + S.offVA + lane, S.SVA, va_pre);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- s_pre (saved for the backward) and the new state x += act(s_pre), both straight from / in registers ----------
        if ((GCP_CF_X & 2) == 0 && it.s_pre) {
            if (it.s_pre_tb) gcp_store_acc_tb<NT>(it.s_pre, r0, acc, lane);
            else gcp_store_acc_rows_any<NT>(it.s_pre, so, r0, rows, acc, stage, lane);
        }
        if constexpr ((NT % 2) == 0) {
            if ((GCP_CF_X & 256) == 0 && it.s_sign) {  // (wave-uniform) where s_pre is positive, one bit per register element: all the backward needs of it
                unsigned* sp = it.s_sign + ((int64_t)blockIdx.x * (NT / 2)) * 64 + lane;
#pragma unroll
                for (int w = 0; w < NT / 2; ++w) {
                    // (x > 0 for a finite x  <=>  its bit pattern, read as a signed integer, is > 0: one v_med3_i32 to 0 / 1 and one
                    // v_lshl_or_b32 per element instead of compare + select + or; +0.0 and every negative value give 0, as x > 0 does)
                    unsigned m = 0;
#pragma unroll
                    for (int b = 0; b < 32; ++b) {
                        const int bits = __float_as_int(acc[2 * w + b / 16][b % 16]);
                        m |= (unsigned)min(max(bits, 0), 1) << b;
                    }
                    sp[w * 64] = m;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {  // (the head is not residual: its output starts the state)
                const float y = gcp_actf<PWL>(it.act_s, ns_s, slope, acc[t][r]);
                xs[t][r] = head ? y : xs[t][r] + y;
            }
        if ((GCP_CF_X & 4) == 0 && it.s_out) {
            if (it.s_out_tb) gcp_store_acc_tb<NT>(it.s_out, r0, xs, lane);
            else gcp_store_acc_rows_any<NT>(it.s_out, so, r0, rows, xs, stage, lane);
        }

#ifndef GCP_FWD_FINE
        if (ci == n_blocks - 1) gcp_stamp(p.stamps, p.stamp_cap, 6, lane);
#endif
        // ---- vector epilogue: vector_up on the matrix cores (B fragments = the parked vector_down outputs), then sigmoid
        //      gate, gating and residual element-wise in registers; the vector tile is updated in place ------------------
        if constexpr ((GCP_CF_X & 32) == 0) {
            gcp_xyz_acc uin, vu;
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int d = 0; d < 3; ++d) uin[d][r] = r < SVBb ? ust[(r * 3 + d) * 64 + lane] : 0.f;
            gcp_xyz_zero(vu);
            if constexpr (PRE) gcp_vmm_regs_pre<NVB>(vb_pre, SVBb, uin, vu);
            else gcp_vmm_regs<16>(it.pack + B.offVB + lane, SVBb, uin, vu);
            float sg[NVR], x[NVR][3];
#pragma unroll
            for (int r = 0; r < NVR; ++r) {
                const int o = min(gcp_crow(r, hi), vo - 1);
                sg[r] = scalar_gate ? gcp_sigmoid(gacc[r]) : 1.f;
#pragma unroll
                for (int d = 0; d < 3; ++d) x[r][d] = head ? 0.f : vt[e * L.VS + 3 * o + d];
            }
#pragma unroll
            for (int r = 0; r < NVR; ++r) {
                const int o = gcp_crow(r, hi);
                if (o < vo) {
                    float u0 = vu[0][r], u1 = vu[1][r], u2 = vu[2][r];
                    if (p.o.vector_residual) { u0 += x[r][0]; u1 += x[r][1]; u2 += x[r][2]; }
                    float sc = sg[r];
                    if (p.o.vmode == GCP_VMODE_SELF_GATE)
                        sc = gcp_actf<PWL>(it.act_v, ns_v, slope, sqrtf(u0 * u0 + u1 * u1 + u2 * u2 + 1e-8f) + 1e-8f);
                    vt[e * L.VS + 3 * o + 0] = x[r][0] + u0 * sc;
                    vt[e * L.VS + 3 * o + 1] = x[r][1] + u1 * sc;
                    vt[e * L.VS + 3 * o + 2] = x[r][2] + u2 * sc;
                }
            }
            if (scalar_gate && it.gate) {  // saved for the backward, straight from the registers (accumulator-layout rows)
#pragma unroll
                for (int q = 0; q < NVR / 4; ++q)
                    if (8 * q < vo)
                        gcp_store4(it.gate, row, vo, 8 * q + 4 * hi, make_float4(sg[4 * q], sg[4 * q + 1], sg[4 * q + 2], sg[4 * q + 3]),
                                   row_ok, (vo & 3) == 0);
            }
        }
        if (head)  // back to the chain blocks' extras layout: their k padding must read as zero
            for (int i = H + S.nf + hi; i < 2 * NX; i += 2) ext[e * L.XS + i] = 0.f;
        gcp_wave_lds_sync();
        if ((GCP_CF_X & 64) == 0 && it.v_out) gcp_store_tile(it.v_out, 3 * vo, 0, 3 * vo, r0, rows, vt, L.VS, lane);
#ifndef GCP_FWD_FINE
        if (ci == n_blocks - 1) gcp_stamp(p.stamps, p.stamp_cap, 7, lane);
#endif
    }
}
#undef p
#undef HD
#undef CF_RELOAD

template <int NT, bool PWL, bool HEAD = false, bool ONLY = false>
int launch_chain(const typename ChainArg<HEAD>::type& p, size_t lds_bytes, hipStream_t st) {
    if constexpr (!HEAD) {
        // scalar_out / gate Linear on the bf16 pipe (three-term split, six products) unless GCPNET_CHAIN_FWD_FP32_MFMA is set
        static const bool f6_env = getenv("GCPNET_CHAIN_FWD_FP32_MFMA") == nullptr;
        const bool f6 = g_gcp_fp32_mfma < 0 ? f6_env : g_gcp_fp32_mfma == 0;
        const dim3 grid((unsigned)gcp_cdiv(p.rows, GCP_TILE_ROWS));
        if (p.sh.H == 4 && p.sh.nf && p.sh.vo <= 16) {  // the shipped shape (V = 16, bottleneck 4)
            if (f6) hipLaunchKernelGGL((gcp2_chain_fwd_kernel<NT, PWL, false, false, 4, true>), grid, dim3(GCP_WAVE), lds_bytes, st, p);
            else hipLaunchKernelGGL((gcp2_chain_fwd_kernel<NT, PWL, false, false, 4, false>), grid, dim3(GCP_WAVE), lds_bytes, st, p);
            GCP_HIP_CHECK_LAUNCH();
            return 0;
        }
        if (f6) {
            hipLaunchKernelGGL((gcp2_chain_fwd_kernel<NT, PWL, false, false, 0, true>), grid, dim3(GCP_WAVE), lds_bytes, st, p);
            GCP_HIP_CHECK_LAUNCH();
            return 0;
        }
    }
    hipLaunchKernelGGL((gcp2_chain_fwd_kernel<NT, PWL, HEAD, ONLY>), dim3((unsigned)gcp_cdiv(p.rows, GCP_TILE_ROWS)), dim3(GCP_WAVE),
                       lds_bytes, st, p);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

}  // namespace

static int fill_chain(ChainParams& p, const GcpShape& S, int n, const gcp2_chain_item_t* items, bool& pwl) {
    for (int k = 0; k < n; ++k) {
        const gcp2_chain_item_t& c = items[k];
        ChainItemF& it = p.it[k];
        it.pack = c.w.pack; it.b_scalar = c.w.b_scalar;
        it.b_gate = c.w.b_gate; it.s_out = c.s_out; it.v_out = c.v_out; it.s_pre = c.s_pre;
        it.gate = c.gate; it.act_s = c.o.act_s; it.act_v = c.o.act_v;
        it.s_out_tb = c.s_out_tb; it.s_pre_tb = c.s_pre_tb;
        it.s_sign = c.s_sign;
        pwl = pwl && gcp_is_pwl(c.o.act_s) && gcp_is_pwl(c.o.act_v);
    }
    (void)S;
    return 0;
}

static bool chain_shape_ok(const GcpShape& S, const gcp2_weights_t& w0) {
    if (S.NG != 1 || S.NTG < 2 || (w0.si & 1) || w0.vi <= 0 || w0.vo <= 0 || w0.vo > 32 || S.GT != 1) return false;
    if (S.NTS != S.NTG || !S.vmm || w0.vi > 32) return false;
    return true;
}

// Returns GCPNET_E_UNSUPPORTED when the shape does not fit the register-resident kernel; the caller then uses the
// LDS-resident chain of gcp2_fwd.hip.
int gcp2_chain_fwd_registers(int rows, const float* s0, const float* v0, const float* frames, int n,
                             const gcp2_chain_item_t* items, hipStream_t st) {
    const gcp2_weights_t& w0 = items[0].w;
    const GcpShape S = gcp_shape(w0.si, w0.vi, w0.so, w0.vo, w0.hidden, w0.use_frames);
    if (!chain_shape_ok(S, w0)) return GCPNET_E_UNSUPPORTED;
    ChainParams p;
    p.rows = rows; p.s0 = s0; p.v0 = v0; p.frames = frames;
    p.o = items[0].o;
    p.n = n;
    bool pwl = true;
    fill_chain(p, S, n, items, pwl);
    p.stamps = g_gcp_phase_buf; p.stamp_cap = g_gcp_phase_cap;
    p.sh = S;
    const size_t lds_bytes = (size_t)chain_lds(S).total * sizeof(float);
    if (lds_bytes > 64 * 1024) return GCPNET_E_UNSUPPORTED;
#ifdef GCP_CF_ONLY_SHIPPED  // (development builds: only the instantiation configs[1] runs)
    hipLaunchKernelGGL((gcp2_chain_fwd_kernel<4, true, false, false, 4, true>), dim3((unsigned)gcp_cdiv(p.rows, GCP_TILE_ROWS)), dim3(GCP_WAVE), lds_bytes, st, p);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
#else
    if (S.NTG == 2) return pwl ? launch_chain<2, true>(p, lds_bytes, st) : launch_chain<2, false>(p, lds_bytes, st);
    return pwl ? launch_chain<4, true>(p, lds_bytes, st) : launch_chain<4, false>(p, lds_bytes, st);
#endif
}

// 1 if a chain of residual blocks of this shape runs in the register-resident kernel above (callers that have another fast
// route -- the workgroup kernel -- ask before choosing; gcpnet_gcp2_chain_forward itself falls back to the LDS-resident chain).
extern "C" int gcpnet_gcp2_chain_forward_registers_ok(int si, int vi, int so, int vo, int hidden, int use_frames) {
    if (si != so || vi != vo) return 0;
    gcp2_weights_t w0{};
    w0.si = si; w0.vi = vi; w0.so = so; w0.vo = vo; w0.hidden = hidden; w0.use_frames = use_frames;
    const GcpShape S = gcp_shape(si, vi, so, vo, hidden, use_frames);
    if (!chain_shape_ok(S, w0)) return 0;
    return (size_t)chain_lds(S).total * sizeof(float) <= 64 * 1024 ? 1 : 0;
}

// The first message GCP after project-then-gather (gcp2_head_t), alone (n == 0) or fused in front of the chain of residual
// blocks it feeds (its outputs are then still written: the backward needs them).
extern "C" int gcpnet_gcp2_headchain_forward(int rows, const gcp2_head_t* head, const float* frames, int n,
                                             const gcp2_chain_item_t* items, void* stream) {
    if (rows < 0 || !head || n < 0 || n > GCP_MAX_CHAIN || (n > 0 && !items)) return GCPNET_E_BADARG;
    const gcp2_weights_t& hw = head->w;
    if (!head->e_in || !head->xi_in || !hw.pack || !hw.b_scalar || !head->s_out || !head->v_out) return GCPNET_E_BADARG;
    if (head->s_add.n < 0 || head->s_add.n > GCP_MAX_SEG || head->v_add.n < 0 || head->v_add.n > GCP_MAX_SEG) return GCPNET_E_BADARG;
    const GcpShape S0 = gcp_shape(hw.si, hw.vi, hw.so, hw.vo, hw.hidden, hw.use_frames);
    // the head must look like a chain block from its outputs' side, with a small plain input
    if (S0.NG != 1 || S0.NTG < 2 || hw.so != 32 * S0.NTG || !S0.vmm || hw.vo > 32 || (hw.vo & 3) || S0.GT != 1) return GCPNET_E_UNSUPPORTED;
    if ((hw.si & 3) || hw.si > 64 || hw.vi <= 0 || ((3 * hw.vi) & 3) || hw.vi > 20 || !hw.use_frames) return GCPNET_E_UNSUPPORTED;
    if (head->o.vector_residual) return GCPNET_E_UNSUPPORTED;
    if (head->o.vmode == GCP_VMODE_SCALAR_GATE && !hw.b_gate) return GCPNET_E_BADARG;
    auto misaligned = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) != 0; };
    if (misaligned(head->e_in) || misaligned(head->xi_in) || misaligned(head->s_out) || misaligned(head->s_pre) ||
        misaligned(head->gate))
        return GCPNET_E_UNSUPPORTED;
    const int hfp = gcp_round_up(S0.HF, 4);
    for (int k = 0; k < head->s_add.n; ++k)
        if (!head->s_add.ptr[k] || head->s_add.dim[k] != hw.so || misaligned(head->s_add.ptr[k])) return GCPNET_E_BADARG;
    for (int k = 0; k < head->v_add.n; ++k)
        if (!head->v_add.ptr[k] || head->v_add.dim[k] != hfp || misaligned(head->v_add.ptr[k])) return GCPNET_E_BADARG;
    if (rows == 0) return 0;
    ChainParamsH p;
    GcpShape S;
    bool pwl = gcp_is_pwl(head->o.act_s) && gcp_is_pwl(head->o.act_v);
    if (n > 0) {
        const gcp2_weights_t& w0 = items[0].w;
        S = gcp_shape(w0.si, w0.vi, w0.so, w0.vo, w0.hidden, w0.use_frames);
        if (!chain_shape_ok(S, w0) || w0.so != hw.so || w0.vo != hw.vo || S.nf != S0.nf) return GCPNET_E_UNSUPPORTED;
        for (int k = 0; k < n; ++k) {
            const gcp2_weights_t& w = items[k].w;
            if (!w.pack || !w.b_scalar || w.si != w0.si || w.vi != w0.vi || w.so != w0.so || w.vo != w0.vo ||
                w.hidden != w0.hidden || w.use_frames != w0.use_frames || items[k].o.vmode != head->o.vmode ||
                items[k].o.vector_residual != 0 || items[k].o.e3 != head->o.e3 || items[k].o.slope != head->o.slope)
                return GCPNET_E_UNSUPPORTED;
        }
        fill_chain(p, S, n, items, pwl);
    } else {  // head alone: the "chain" shape only provides output dims and LDS strides
        S = gcp_shape(hw.so, hw.vo, hw.so, hw.vo, hw.hidden < hw.vo ? hw.hidden : hw.vo, hw.use_frames);
        if (S.NTS != S.NTG || !S.vmm) return GCPNET_E_UNSUPPORTED;
    }
    p.rows = rows; p.s0 = nullptr; p.v0 = nullptr; p.frames = frames;
    p.o = head->o;
    p.n = n;
    p.stamps = g_gcp_phase_buf; p.stamp_cap = g_gcp_phase_cap;
    p.sh = S;
    p.hd.e_in = head->e_in; p.hd.xi_in = head->xi_in;
    p.hd.s_add = head->s_add; p.hd.v_add = head->v_add;
    p.hd.sh = S0;
    ChainItemF& it = p.hd.it;
    it.pack = hw.pack; it.b_scalar = hw.b_scalar; it.b_gate = hw.b_gate;
    it.s_out = head->s_out; it.v_out = head->v_out; it.s_pre = head->s_pre; it.gate = head->gate;
    it.act_s = head->o.act_s; it.act_v = head->o.act_v;
    it.s_out_tb = it.s_pre_tb = 0;
    it.s_sign = nullptr;
    const size_t lds_bytes = (size_t)chain_lds(S, &S0).total * sizeof(float);
    if (lds_bytes > 64 * 1024) return GCPNET_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        if (S.NTG == 2) return pwl ? launch_chain<2, true, true, true>(p, lds_bytes, st) : launch_chain<2, false, true, true>(p, lds_bytes, st);
        return pwl ? launch_chain<4, true, true, true>(p, lds_bytes, st) : launch_chain<4, false, true, true>(p, lds_bytes, st);
    }
    if (S.NTG == 2) return pwl ? launch_chain<2, true, true>(p, lds_bytes, st) : launch_chain<2, false, true>(p, lds_bytes, st);
    return pwl ? launch_chain<4, true, true>(p, lds_bytes, st) : launch_chain<4, false, true>(p, lds_bytes, st);
}
