// GCP2 forward for a 32-row tile per WORKGROUP (NW wavefronts), see gcp_wg.h: one block (any dims: embeddings, feed-forward
// and position-update GCPs on node rows, the first message GCP after project-then-gather) or a whole chain
// x_k = x_{k-1} + GCP_k(x_{k-1}) (ResGCP, reference src/models/components/gcpnet.py:921-924) behind an optional non-residual
// first block, in ONE launch with the (s, V) state of the tile on chip.  Replaces GCP2.forward (gcpnet.py:394-468).
//
// Per block and tile:
//   prologue (VALU, thread = (row, hidden channel)): [vh | vf] = [vector_down ; vector_down_frames] v (+ gathered addends),
//             norms of vh and frame projections of vf -> the extras columns of X; vh parked in LDS            -> barrier B1
//   GEMM (MFMA, wave = 32-column output tiles): s_pre = b + W [s | norms | frame scalars] (+ gathered addends);
//             gate partial = Wg[:, own columns] act_v(s_pre) -> LDS; s_pre and the new scalars leave through a wave-private
//             staging tile as full 64-byte row pieces                                                           -> barrier B2
//   epilogue (VALU, thread = (row, output channel)): vector_up, gate = sigmoid(sum of partials), gating, residual; the
//             wave's slice of the new scalar state goes back to X                                              -> barrier B3
#include <cstdlib>

#include "gcp_wg.h"

namespace {

struct WgBlk {
    const float* pk;
    const float* b_scalar;
    const float* b_gate;
    const float* w_down;
    const float* w_frames;
    const float* w_up;
    float* s_out;
    float* v_out;
    float* s_pre;
    float* gate;
    int64_t offG1, offA1b;
    int act_s, act_v;
    int si, vi, H, K, KG;
    int residual;
    int tb;  // bit 0: s_out, bit 1: s_pre written tile-blocked (so % 32 == 0)
};

struct WgFwdParams {
    int rows;
    const float* s_in;
    const float* v_in;
    const float* frames;
    gcp_concat_t s_add, v_add;  // block 0 only
    int so, vo, nf, e3, vmode, vres;
    float slope;
    int NT, NG;
    int n;
    int KS, VS, HS, GS;
    int o_x, o_v, o_vh, o_fr, o_ws, o_st, ws_floats, st_floats;
    int o_xp;  // (B6) operand planes of X: [slab][term][64 lanes][16 bytes]
    unsigned long long* stamps;  // profiling hook (gcpnet_debug_set_phase_timing): s_memtime stamps of wave 0, last block
    long long stamp_cap;
    WgBlk blk[GCP_WG_MAX_BLOCKS];
};



// Compile-time shapes (template parameter SHP; 0 = run-time shape): a launch of n > 1 IDENTICAL residual message GCPs
// (s, V) -> (s, V) with frames and a scalar gate -- the ResGCP chain of BASELINE configs[1] (128, 16, hidden 4) and configs[4]
// (256, 32, hidden 8).  One constexpr function gives every integer of such a launch; the host takes the instantiation only when the
// parameters it computed the general way are exactly these (wg_fwd_is_shape), and in the kernel they fold into immediates.
#define WG_FWD_DIMS(X) X(so) X(vo) X(nf) X(NT) X(NG) X(KS) X(VS) X(HS) X(GS) X(o_x) X(o_v) X(o_vh) X(o_fr) X(o_ws) X(o_st) X(ws_floats) X(st_floats) X(o_xp)
#define WG_FWD_BLK_DIMS(X) X(si) X(vi) X(H) X(K) X(KG)
struct WgFwdDims {
#define X(f) int f;
    WG_FWD_DIMS(X)
    WG_FWD_BLK_DIMS(X)
#undef X
    int lds_floats;
};
constexpr int cf_cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int cf_rup(int x, int m) { return (x + m - 1) / m * m; }
constexpr int cf_max(int a, int b) { return a > b ? a : b; }
constexpr int cf_stride(int width) { return 4 * (cf_cdiv(width, 4) | 1); }  // == wg_stride
constexpr WgFwdDims wg_fwd_dims(int S, int V, int HID, int NW, int MT, int b6 = 0) {
    WgFwdDims d{};
    d.so = S; d.vo = V; d.nf = 9; d.si = S; d.vi = V; d.H = HID;
    d.K = S + HID + 9; d.KG = cf_cdiv(d.K, 8);
    d.NT = cf_cdiv(S, 32); d.NG = cf_cdiv(d.NT, NW * MT);
    const int kmax = cf_max(8 * d.KG, cf_rup(S, 4));
    d.KS = cf_stride(kmax); d.VS = cf_stride(3 * V); d.HS = cf_stride(3 * cf_max(HID, 1)); d.GS = cf_stride(cf_rup(cf_max(V, 1), 8));
    const int HF = HID + 3;
    d.ws_floats = cf_rup(cf_rup(HF * cf_stride(V) + V * cf_stride(cf_max(HID, 1)) + V, 4) + cf_rup(S, 4), 4);
    int off = 0;
    d.o_x = off; off += 32 * d.KS;
    d.st_floats = cf_max(WG_STAGE_FLOATS, 32 * d.GS);
    d.o_st = off; off += NW * d.st_floats;
    d.o_ws = off; off += 2 * d.ws_floats;
    d.o_v = off; off += cf_rup(32 * d.VS, 4);
    d.o_vh = off; off += cf_rup(32 * d.HS, 4);
    d.o_fr = off; off += 32 * 9;
    d.o_xp = off; off += b6 ? cf_cdiv(d.KG, 2) * 768 : 0;
    d.lds_floats = off;
    return d;
}
template <int SHP> struct WgFwdShape { static constexpr int S = 0, V = 0, HID = 0; };
template <> struct WgFwdShape<1> { static constexpr int S = 128, V = 16, HID = 4; };
template <> struct WgFwdShape<2> { static constexpr int S = 256, V = 32, HID = 8; };

// B6 (MT == 1): the K loop on the bf16 matrix pipe -- once the tile X of a block is complete every thread splits a share of
// it into three bf16 terms (operand-ordered planes in LDS, shared by all waves), the weights come pre-split from section A1b;
// six products, fp32 accumulation (gcp_bf16x3.h: exact to fp32 round-off)
// G2: 32 < vo <= 64 with a scalar gate -- two 32-row tiles of gate outputs per wave (sixteen more accumulator registers; these
// shapes, the feed-forward GCP (s, V) -> (4 s, 2 V) of BASELINE configs[4], hold one workgroup per CU by their LDS anyway)
template <int NW, int MT, bool PWL, int SHP, bool B6 = false, bool G2 = false>
__global__ __launch_bounds__(64 * NW, (G2 ? 1 : (NW == 4 ? 3 : 2))) void gcp_wg_fwd_kernel(const WgFwdParams p_kernarg) {
    static_assert(!B6 || MT == 1, "bf16 form: one output tile per wave");
    static_assert(!G2 || (SHP == 0 && !B6), "two gate tiles: run-time shapes, fp32 MFMA form");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NTH = 64 * NW, TPR = NTH / 32, U = 4 / MT;
    // Parameters are read through the kernarg segment pointer, laundered at the phase boundaries: the uniform values are
    // re-loaded (s_load) by the phase that uses them instead of living in SGPRs across the whole block loop (where ~90 of them
    // were spilled into VGPR lanes).  The generic view of the pointer is inferred back to the constant address space.
    typedef const __attribute__((address_space(4))) WgFwdParams* Karg;
    Karg kp = (Karg)__builtin_amdgcn_kernarg_segment_ptr();
#define p (*(const WgFwdParams*)kp)
    constexpr WgFwdDims CF = wg_fwd_dims(WgFwdShape<SHP>::S, WgFwdShape<SHP>::V, WgFwdShape<SHP>::HID, NW, MT, B6);
#define DM(f) (SHP ? CF.f : p.f)       // a launch-wide shape value: immediate for the compile-time shapes
#define DB(blk, f) (SHP ? CF.f : (blk).f)  // a per-block one
    int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lane = tid & 63, e = lane & 31, hi = lane >> 5;
    const int r0 = blockIdx.x * 32;
    int prow = tid / TPR, psub = tid - prow * TPR;  // (row, sub-index) of the VALU phases
    int rows, nvalid, KS, VS, HS, GS, so, vo, nf, NT;
    float *X, *V, *VH, *FR, *ST;
    bool scalar_gate;
    float slope;
#define WG_RELOAD()                                                                                                      \
    do {                                                                                                                 \
        rows = p.rows;                                                                                                   \
        nvalid = min(32, rows - r0);                                                                                     \
        X = lds + DM(o_x); V = lds + DM(o_v); VH = lds + DM(o_vh); FR = lds + DM(o_fr);                                  \
        ST = lds + DM(o_st) + w * DM(st_floats); /* wave-private staging tile; waves 0..3: also their gate-partial slot */ \
        KS = DM(KS); VS = DM(VS); HS = DM(HS); GS = DM(GS);                                                              \
        so = DM(so); vo = DM(vo); nf = DM(nf); NT = DM(NT);                                                              \
        scalar_gate = SHP ? true : (p.vmode == GCP_VMODE_SCALAR_GATE && vo > 0);                                         \
        slope = p.slope;                                                                                                 \
    } while (0)
    // per-lane addresses are invariant over the block loop as well: hipcc hoists them out of it and spills them; laundering
    // the lane indices at the phase boundaries makes every phase recompute the few it needs
#define WG_LAUNDER()                                                                                                     \
    do {                                                                                                                 \
        asm volatile("" : "+v"(tid), "+v"(lane), "+v"(e), "+v"(hi), "+v"(prow), "+v"(psub), "+s"(kp));                    \
        WG_RELOAD();                                                                                                     \
    } while (0)
    WG_RELOAD();

    // small weights of block b -> LDS buffer b & 1: [vector_down ; vector_down_frames] rows (stride wg_stride(vi)), vector_up rows
    // (stride wg_stride(H): 16-byte reads), gate bias, scalar_out bias.  Split in a request (global loads into registers, issued before a block's GEMM) and a commit
    // (LDS writes, after it): the round trip to L2 stays under the MFMAs.
    constexpr int WSR = 4;
    float wsr[WSR];
    auto ws_src = [&](const WgBlk& B, int i, int& dst) -> const float* {  // flat index over [down ; frames | up | gate bias]
        const int H = DB(B, H), vi = DB(B, vi), HF = H + (nf ? 3 : 0), WSV = wg_stride(vi), WSU = wg_stride(H);
        const int n1 = HF * vi, n2 = n1 + vo * H;
        if (i < n1) {
            const int x = i / vi, c = i - x * vi;
            dst = x * WSV + c;
            return x < H ? B.w_down + i : B.w_frames + (i - H * vi);
        }
        if (i < n2) {
            const int j = i - n1, o = j / H, h = j - o * H;
            dst = HF * WSV + o * WSU + h;
            return B.w_up + j;
        }
        const int n3 = n2 + (scalar_gate ? vo : 0);
        if (i < n3) {
            dst = HF * WSV + vo * WSU + (i - n2);
            return B.b_gate + (i - n2);
        }
        dst = gcp_round_up(HF * WSV + vo * WSU + vo, 4) + (i - n3);
        return B.b_scalar + (i - n3);
    };
    auto ws_count = [&](const WgBlk& B) { return (DB(B, H) + (nf ? 3 : 0)) * DB(B, vi) + vo * DB(B, H) + (scalar_gate ? vo : 0) + so; };
    auto ws_request = [&](int b) {
        const WgBlk& B = p.blk[b];
        const int n = ws_count(B);
#pragma unroll
        for (int k = 0; k < WSR; ++k) {
            int dst;
            wsr[k] = *ws_src(B, min(tid + k * NTH, n - 1), dst);
        }
    };
    auto ws_commit = [&](int b) {
        const WgBlk& B = p.blk[b];
        float* ws = lds + DM(o_ws) + (b & 1) * DM(ws_floats);
        const int n = ws_count(B);
#pragma unroll
        for (int k = 0; k < WSR; ++k) {
            int dst;
            const int i = tid + k * NTH;
            ws_src(B, min(i, n - 1), dst);
            if (i < n) ws[dst] = wsr[k];
        }
        for (int i = tid + WSR * NTH; i < n; i += NTH) {  // (shapes with more than WSR * NTH small weights)
            int dst;
            const float* s = ws_src(B, i, dst);
            ws[dst] = *s;
        }
    };

    // ---- the tile: scalars -> X[:, 0:si0), vectors -> V, frames -> FR (flat, coalesced copies: tile rows are contiguous) ----
    {
        const WgBlk& B0 = p.blk[0];
        const int si0 = DB(B0, si), vw = 3 * DB(B0, vi);
        const float* src = p.s_in + (int64_t)r0 * si0;
        if ((si0 & 3) == 0 && wg_aligned16(p.s_in)) {
            const int q = si0 >> 2, n4 = nvalid * q;
            for (int i0 = 0; i0 < 32 * q; i0 += 4 * NTH) {
                f32x4 v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f32x4*>(src + 4 * (int64_t)min(i0 + tid + k * NTH, n4 - 1));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + tid + k * NTH;
                    if (i < 32 * q) {
                        const int r = i / q, c4 = i - r * q;
                        f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        *reinterpret_cast<f32x4*>(X + r * KS + 4 * c4) = i < n4 ? v[k] : z;
                    }
                }
            }
        } else {
            const int n1 = nvalid * si0;
            for (int i0 = 0; i0 < 32 * si0; i0 += 4 * NTH) {
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = src[min(i0 + tid + k * NTH, n1 - 1)];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = i0 + tid + k * NTH;
                    if (i < 32 * si0) {
                        const int r = i / si0, c = i - r * si0;
                        X[r * KS + c] = i < n1 ? v[k] : 0.f;
                    }
                }
            }
        }
        wg_tile_load<NTH>(V, VS, p.v_in + (int64_t)r0 * vw, vw, nvalid, tid, (vw & 3) == 0 && wg_aligned16(p.v_in));
        if (nf) {
            const float* fsrc = p.frames + (int64_t)r0 * 9;
            for (int i = tid; i < 32 * 9; i += NTH) FR[i] = fsrc[min(i, nvalid * 9 - 1)];
        }
        ws_request(0);
        ws_commit(0);
    }
    wg_barrier();

    auto run_block = [&](auto first_tag, const int b) {
        constexpr bool FIRST = decltype(first_tag)::value;
#define B (p.blk[b])
        auto stamp = [&](int k) {
            if (b == p.n - 1 && w == 0) gcp_stamp(p.stamps, p.stamp_cap, k, lane);
        };
        stamp(0);
        const float *ws, *wu, *bg, *bs;
        int H, vi, si, HF, WSV, WSU, KG, K, KP;
        float ns_s, ns_v;
#define RB_LAUNDER()                                                                                                     \
    do {                                                                                                                 \
        WG_LAUNDER();                                                                                                    \
        ws = lds + DM(o_ws) + (b & 1) * DM(ws_floats);                                                                   \
        H = DB(B, H); vi = DB(B, vi); si = DB(B, si); HF = H + (nf ? 3 : 0); WSV = wg_stride(vi); WSU = wg_stride(H);    \
        wu = ws + HF * WSV;                                                                                              \
        bg = wu + vo * WSU;                                                                                              \
        bs = ws + gcp_round_up(HF * WSV + vo * WSU + vo, 4);                                                             \
        KG = DB(B, KG); K = DB(B, K); KP = 8 * KG;                                                                       \
        ns_s = gcp_neg_slope(B.act_s, slope); ns_v = gcp_neg_slope(B.act_v, slope);                                      \
    } while (0)

        RB_LAUNDER();
        // ---- prologue ----------------------------------------------------------------------------------------------------
        {
            const float* vrow = V + prow * VS;
            const int grow = min(r0 + prow, rows - 1);
            const int HFP = gcp_round_up(HF, 4);
#pragma unroll 1
            for (int x = psub; x < HF; x += TPR) {
                float q0 = 0.f, q1 = 0.f, q2 = 0.f;
                if constexpr (FIRST) {  // shares of the pre-projected (gathered) vector sources
                    for (int k = 0; k < p.v_add.n; ++k) {
                        const int32_t* ix = p.v_add.idx[k];
                        const float* t = p.v_add.ptr[k] + (int64_t)(ix ? ix[grow] : grow) * 3 * HFP;
                        q0 += t[x]; q1 += t[HFP + x]; q2 += t[2 * HFP + x];
                    }
                }
                const float* wr = ws + x * WSV;
                float u0 = 0.f, u1 = 0.f, u2 = 0.f;
                if ((vi & 3) == 0) {  // four channels per step: one 16-byte read of the weights, three of the vectors
#pragma unroll 2
                    for (int c = 0; c < vi; c += 4) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + c);
                        const f32x4 a = *reinterpret_cast<const f32x4*>(vrow + 3 * c);
                        const f32x4 bq = *reinterpret_cast<const f32x4*>(vrow + 3 * c + 4);
                        const f32x4 d = *reinterpret_cast<const f32x4*>(vrow + 3 * c + 8);
                        u0 = fmaf(wv[0], a[0], u0); u1 = fmaf(wv[0], a[1], u1); u2 = fmaf(wv[0], a[2], u2);
                        u0 = fmaf(wv[1], a[3], u0); u1 = fmaf(wv[1], bq[0], u1); u2 = fmaf(wv[1], bq[1], u2);
                        u0 = fmaf(wv[2], bq[2], u0); u1 = fmaf(wv[2], bq[3], u1); u2 = fmaf(wv[2], d[0], u2);
                        u0 = fmaf(wv[3], d[1], u0); u1 = fmaf(wv[3], d[2], u1); u2 = fmaf(wv[3], d[3], u2);
                    }
                } else {
                    for (int c = 0; c < vi; ++c) {
                        const float wv = wr[c];
                        u0 = fmaf(wv, vrow[3 * c + 0], u0);
                        u1 = fmaf(wv, vrow[3 * c + 1], u1);
                        u2 = fmaf(wv, vrow[3 * c + 2], u2);
                    }
                }
                u0 += q0; u1 += q1; u2 += q2;
                if (x < H) {
                    VH[prow * HS + 3 * x + 0] = u0; VH[prow * HS + 3 * x + 1] = u1; VH[prow * HS + 3 * x + 2] = u2;
                    X[prow * KS + si + x] = sqrtf(u0 * u0 + u1 * u1 + u2 * u2 + 1e-8f) + 1e-8f;
                } else {
                    const int k = x - H;
                    const float* f = FR + prow * 9;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        float pr = f[3 * a + 0] * u0 + f[3 * a + 1] * u1 + f[3 * a + 2] * u2;
                        if (p.e3 && a == 1) pr = fabsf(pr);
                        X[prow * KS + si + H + 3 * k + a] = pr;
                    }
                }
            }
            const int npad = KP - K;  // k padding of this block's merged axis reads as zero
            for (int i = tid; i < 32 * npad; i += NTH) {
                const int r = i / npad, c = i - r * npad;
                X[r * KS + K + c] = 0.f;
            }
        }
#if GCP_ARITH_F16X2
        // (two-term fp16 form: the largest magnitude of every row of X, collected after B1 -- zeroed here, behind B3 of the block before,
        // whose readers are done.  32 words behind the operand planes: the planes use two thirds of their 768 floats per slab)
        if (B6 && tid < 32) reinterpret_cast<unsigned*>(lds + DM(o_xp))[gcp_cdiv(KG, 2) * (GCP_W6_TERMS * 256) + tid] = 0u;
#endif
        stamp(1);
        wg_barrier();  // B1
        if constexpr (B6) {
            // X is complete: eight consecutive columns of a row at a time -> three bf16 terms -> the lane's 16 bytes of the
            // slab's planes (lane = 32 (group & 1) + row: consecutive threads write consecutive 16-byte pieces)
            const int NG2 = 2 * gcp_cdiv(KG, 2);
            gcp_u32x4* XP = reinterpret_cast<gcp_u32x4*>(lds + DM(o_xp));
#if GCP_ARITH_F16X2
            // gcp_f16x2.h: a power-of-two scale per ROW of X (constant along the summed columns), from the row's largest magnitude
            unsigned* rmax = reinterpret_cast<unsigned*>(lds + DM(o_xp)) + gcp_cdiv(KG, 2) * (GCP_W6_TERMS * 256);
            for (int u = tid; u < 32 * KG; u += NTH) {
                const int r = u & 31, g = u >> 5;
                const float* xr = X + r * KS + 8 * g;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(xr), hi4 = *reinterpret_cast<const f32x4*>(xr + 4);
                const float m = fmaxf(fmaxf(fmaxf(fabsf(lo[0]), fabsf(lo[1])), fmaxf(fabsf(lo[2]), fabsf(lo[3]))),
                                      fmaxf(fmaxf(fabsf(hi4[0]), fabsf(hi4[1])), fmaxf(fabsf(hi4[2]), fabsf(hi4[3]))));
                atomicMax(rmax + r, __float_as_uint(m));  // (non-negative floats order like their bit patterns)
            }
            wg_barrier();
#endif
            for (int u = tid; u < 32 * NG2; u += NTH) {
                const int r = u & 31, g = u >> 5;
                const bool in = g < KG;
                const float* xr = X + r * KS + 8 * (in ? g : 0);
                const f32x4 lo = *reinterpret_cast<const f32x4*>(xr), hi4 = *reinterpret_cast<const f32x4*>(xr + 4);
                const float x8[8] = {in ? lo[0] : 0.f, in ? lo[1] : 0.f, in ? lo[2] : 0.f, in ? lo[3] : 0.f,
                                     in ? hi4[0] : 0.f, in ? hi4[1] : 0.f, in ? hi4[2] : 0.f, in ? hi4[3] : 0.f};
                gcp_u32x4* q = XP + (g >> 1) * (GCP_W6_TERMS * 64) + 32 * (g & 1) + r;
#if GCP_ARITH_F16X2
                gcp_u32x4 th, tl;
                gcp_f16x2_split8(x8, gcp_exp2i(gcp_f16_row_exp(__uint_as_float(rmax[r]))), th, tl);
                q[0] = th; q[64] = tl;
#else
                gcp_u32x4 th, tm, tl;
                gcp_bf16x3_split8(x8, th, tm, tl);
                q[0] = th; q[64] = tm; q[128] = tl;
#endif
            }
            wg_barrier();
        }
        stamp(2);
        if (b + 1 < p.n) ws_request(b + 1);

        RB_LAUNDER();
        // ---- scalar_out (+ gate partial) per output group ---------------------------------------------------------------
        f32x16 ynew[MT];  // the wave's slice of the block output (accumulator layout); the chain state when NG == 1
        f32x16 gacc;      // this wave's partial gate pre-activations (its columns of the reduction over so)
        f32x16 gacc2;     // (G2) ... of the gate outputs 32 .. 63
        float zero = 0.f;  // (laundered: hipcc otherwise hoists the zero-initialised accumulators above the VALU phases in
        asm volatile("" : "+v"(zero));  //  front of them and spills sixteen registers of zeros across each)
        bool gacc_set = false;
        for (int og = 0; og < DM(NG); ++og) {
            const int ot0 = og * NW * MT + w;
            if (ot0 >= NT) continue;  // (wave-uniform) nothing for this wave in this group
            int otc[MT];
            bool tv[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                tv[t] = ot0 + NW * t < NT;
                otc[t] = min(ot0 + NW * t, NT - 1);
            }
            f32x16 acc[MT];
            asm volatile("" : "+v"(zero));
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = zero;
            // K loop in batches of U groups of 8 columns, fragments requested two batches ahead.  With one tile per wave the
            // batch BEHIND the last one carries the four weight fragments of the gate Linear for this wave's columns: they
            // arrive under the reduction's MFMAs instead of costing a round trip to L2 afterwards.
            f32x4 a0[U][MT], a1[U][MT], b0[U], b1[U];
            int nb_;
            if constexpr (B6) {
                const int NSLf = gcp_cdiv(KG, 2);
                constexpr int NTM = GCP_W6_TERMS, SLAB = NTM * 64;  // 16-byte entries per slab: [term][64 lanes]
                const gcp_u32x4* pa6 = reinterpret_cast<const gcp_u32x4*>(B.pk + B.offA1b) + (int64_t)otc[0] * NSLf * SLAB + lane;
                const gcp_u32x4* pb6 = reinterpret_cast<const gcp_u32x4*>(lds + DM(o_xp)) + lane;
                auto lda = [&](gcp_u32x4(&a)[NTM], int sj) {
                    const gcp_u32x4* q = pa6 + (int64_t)min(sj, NSLf - 1) * SLAB;
#pragma unroll
                    for (int tm = 0; tm < NTM; ++tm) a[tm] = q[64 * tm];
                };
                gcp_u32x4 f0[NTM], f1[NTM], f2[NTM], f3[NTM];
                lda(f0, 0); lda(f1, 1); lda(f2, 2); lda(f3, 3);
                if (scalar_gate) {  // the gate Linear's four fragments for this wave's columns arrive under the reduction
                    const float* pkG = B.pk + B.offG1 + (int64_t)lane * 4;
#pragma unroll
                    for (int u = 0; u < U; ++u) a0[u][0] = *reinterpret_cast<const f32x4*>(pkG + (int64_t)(4 * otc[0] + u) * 256);
                }
                nb_ = 0;  // (the gate Linear below takes its fragments from a0)
                __builtin_amdgcn_sched_barrier(0);
                for (int sj = 0; sj < NSLf; sj += 4) {
#if GCP_ARITH_F16X2
#define WG_B6_STEP(A, S)                                                                                       \
    if ((S) < NSLf) {                                                                                          \
        const gcp_u32x4* qb = pb6 + (int64_t)(S) * SLAB;                                                       \
        const gcp_u32x4 bh = qb[0], bl = qb[64];                                                               \
        acc[0] = gcp_mfma_f16x3(A, bh, bl, acc[0]);                                                            \
        lda(A, (S) + 4);                                                                                       \
    }                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);
#else
#define WG_B6_STEP(A, S)                                                                                       \
    if ((S) < NSLf) {                                                                                          \
        const gcp_u32x4* qb = pb6 + (int64_t)(S) * SLAB;                                                       \
        const gcp_u32x4 bh = qb[0], bm = qb[64], bl = qb[128];                                                 \
        acc[0] = gcp_mfma_bf16x6(A, bh, bm, bl, acc[0]);                                                       \
        lda(A, (S) + 4);                                                                                       \
    }                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);
#endif
                    WG_B6_STEP(f0, sj)
                    WG_B6_STEP(f1, sj + 1)
                    WG_B6_STEP(f2, sj + 2)
                    WG_B6_STEP(f3, sj + 3)
#undef WG_B6_STEP
                }
#if GCP_ARITH_F16X2
                {   // back from 2^(pa + GCP_F16_WEXP): this lane's accumulators are all row e's
                    const unsigned* rmax = reinterpret_cast<const unsigned*>(lds + DM(o_xp)) + NSLf * (GCP_W6_TERMS * 256);
                    const float isc = gcp_exp2i(-(gcp_f16_row_exp(__uint_as_float(rmax[e])) + GCP_F16_WEXP));
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[0][r] *= isc;
                }
#endif
            } else {
                const float* pkA = B.pk + (int64_t)lane * 4;
                const float* pkG = B.pk + B.offG1 + (int64_t)lane * 4;
                const float* xb = X + e * KS + 4 * hi;
                const int nb = gcp_cdiv(KG, U);
                const bool gate_pf = MT == 1 && scalar_gate;
                auto ldA = [&](f32x4(&a)[U][MT], f32x4(&bb)[U], int bi) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int g = min(U * bi + u, KG - 1);
                        bb[u] = *reinterpret_cast<const f32x4*>(xb + 8 * g);
#pragma unroll
                        for (int t = 0; t < MT; ++t) {
                            const float* src = pkA + ((int64_t)otc[t] * KG + g) * 256;
                            if (MT == 1 && U == 4) src = (gate_pf && bi == nb) ? pkG + (int64_t)(4 * otc[0] + u) * 256 : src;
                            a[u][t] = *reinterpret_cast<const f32x4*>(src);
                        }
                    }
                };
                auto mm = [&](f32x4(&a)[U][MT], f32x4(&bb)[U], int bi) {
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (U * bi + u < KG) {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
#pragma unroll
                                for (int t = 0; t < MT; ++t)
                                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t][i], bb[u][i], acc[t], 0, 0, 0);
                        }
                };
                nb_ = nb;
                ldA(a0, b0, 0);
                for (int bi = 0; bi < nb; bi += 2) {
                    ldA(a1, b1, bi + 1);
                    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch HERE (hipcc otherwise sinks loads to their use)
                    mm(a0, b0, bi);
                    ldA(a0, b0, bi + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(a1, b1, bi + 1);
                }
            }
            stamp(3);
            // (lane indices only: inside the group loop the kernarg pointer must stay provably uniform)
            asm volatile("" : "+v"(tid), "+v"(lane), "+v"(e), "+v"(hi), "+v"(prow), "+v"(psub));
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool in = 32 * otc[t] + 8 * q + 4 * hi + 3 < so;
                    const f32x4 bq = *reinterpret_cast<const f32x4*>(bs + min(32 * otc[t] + 8 * q + 4 * hi, so - 4));
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[t][4 * q + i] += in ? bq[i] : 0.f;
                }
            if constexpr (FIRST) {
                // shares of the pre-projected (gathered) scalar sources: rows of [n_src, so] tables in the accumulator layout.
                // Requested only now (one exposed round trip per tile, first block only) -- held across the reduction they
                // would cost 16 registers per table.
                const int grow = min(r0 + e, rows - 1);
                for (int k = 0; k < p.s_add.n; ++k) {
                    const int32_t* ix = p.s_add.idx[k];
                    const float* base = p.s_add.ptr[k] + (int64_t)(ix ? ix[grow] : grow) * so;
                    f32x4 ad[MT][4];
#pragma unroll
                    for (int t = 0; t < MT; ++t)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            ad[t][q] = *reinterpret_cast<const f32x4*>(base + min(32 * otc[t] + 8 * q + 4 * hi, so - 4));
#pragma unroll
                    for (int t = 0; t < MT; ++t)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const bool in = 32 * otc[t] + 8 * q + 4 * hi + 3 < so;
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc[t][4 * q + i] += in ? ad[t][q][i] : 0.f;
                        }
                }
            }

            __builtin_amdgcn_sched_barrier(0);  // (phases kept apart: interleaving them only raises register pressure)
            // gate partial over this wave's columns: B fragments = act_v(s_pre) straight from the accumulators
            if (scalar_gate) {
                if (!gacc_set) {  // (initialised here, not in front of the reduction: sixteen live registers less in its loop)
                    asm volatile("" : "+v"(zero));
#pragma unroll
                    for (int r = 0; r < 16; ++r) gacc[r] = zero;
                    if constexpr (G2) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) gacc2[r] = zero;
                    }
                    gacc_set = true;
                }
                auto gate_mm = [&](const f32x16& av, auto frag, int otile) {
                    f32x4 ag2[4];
                    if constexpr (G2) {  // the second tile's fragments: section G1, tile gm = 1
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            ag2[q] = *reinterpret_cast<const f32x4*>(B.pk + B.offG1 + ((int64_t)(4 * NT + 4 * otile + q) * 64 + lane) * 4);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 ag = frag(q);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float bv = gcp_actf<PWL>(B.act_v, ns_v, slope, av[4 * q + i]);
                            gacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ag[i], bv, gacc, 0, 0, 0);
                            if constexpr (G2) gacc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ag2[q][i], bv, gacc2, 0, 0, 0);
                        }
                    }
                };
                if constexpr (MT == 1 && U == 4) {  // the prefetched batch nb sits in a0 when nb is even, in a1 when it is odd
                    if (nb_ & 1) gate_mm(acc[0], [&](int q) { return a1[q][0]; }, otc[0]);
                    else gate_mm(acc[0], [&](int q) { return a0[q][0]; }, otc[0]);
                } else {
#pragma unroll
                    for (int t = 0; t < MT; ++t)
                        if (tv[t]) {
                            f32x4 ag[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                ag[q] = *reinterpret_cast<const f32x4*>(B.pk + B.offG1 + ((int64_t)(4 * otc[t] + q) * 64 + lane) * 4);
                            gate_mm(acc[t], [&](int q) { return ag[q]; }, otc[t]);
                        }
                }
            }

            __builtin_amdgcn_sched_barrier(0);
            // full 128-byte row pieces through the wave-private staging tile (wg_store_acc)
            auto store_acc = [&](float* dst, const f32x16& av, int otile) { wg_store_acc(dst, so, 32 * otile, so, r0, nvalid, av, ST, lane); };
            // tile-blocked (gcp2_chain_item_t's layout): a register quad of the wave is 1 KB of whole lines, no staging tile
            auto store_tb = [&](float* dst, const f32x16& av, int otile) {
                float* d = dst + (int64_t)r0 * so + (int64_t)otile * 1024 + lane * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = {av[4 * q], av[4 * q + 1], av[4 * q + 2], av[4 * q + 3]};
                    *reinterpret_cast<f32x4*>(d + q * 256) = v;
                }
            };
            if (B.s_pre) {
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    if (tv[t]) {
                        if (B.tb & 2) store_tb(B.s_pre, acc[t], otc[t]);
                        else store_acc(B.s_pre, acc[t], otc[t]);
                    }
            }
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 old = {0.f, 0.f, 0.f, 0.f};  // ResGCP: the block's input slice is still in X (until B2)
                    if (B.residual) old = *reinterpret_cast<const f32x4*>(X + e * KS + min(32 * otc[t] + 8 * q + 4 * hi, so - 4));
#pragma unroll
                    for (int i = 0; i < 4; ++i) ynew[t][4 * q + i] = old[i] + gcp_actf<PWL>(B.act_s, ns_s, slope, acc[t][4 * q + i]);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (B.s_out) {
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    if (tv[t]) {
                        if (B.tb & 1) store_tb(B.s_out, ynew[t], otc[t]);
                        else store_acc(B.s_out, ynew[t], otc[t]);
                    }
            }
        }
        stamp(4);
        RB_LAUNDER();
        if (scalar_gate) {
            // partial gate pre-activations -> GP[w & 3][32][vo]; with eight waves the upper four add theirs in a second step
            // (half the LDS: the partials of a (256+, 32+) block would not fit otherwise)
            auto gp_quad = [&](int q) {
                f32x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) {  // (static register index)
                    float sacc = 0.f;
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) sacc = qq == q ? gacc[4 * qq + i] : sacc;
                    if constexpr (G2) {
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) sacc = qq + 4 == q ? gacc2[4 * qq + i] : sacc;
                    }
                    v[i] = sacc;
                }
                return v;
            };
            if (!gacc_set) {  // (a wave without output tiles contributes zeros)
#pragma unroll
                for (int r = 0; r < 16; ++r) gacc[r] = 0.f;
                if constexpr (G2) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) gacc2[r] = 0.f;
                }
            }
            gcp_wave_lds_sync();  // (the slot doubles as this wave's staging tile: its reads are done)
            if (w < 4)
                for (int q = 0; 8 * q < vo; ++q) *reinterpret_cast<f32x4*>(ST + e * GS + 8 * q + 4 * hi) = gp_quad(q);
            if constexpr (NW == 8) {
                wg_barrier();
                if (w >= 4)
                    for (int q = 0; 8 * q < vo; ++q) {
                        float* gp = lds + DM(o_st) + (w - 4) * DM(st_floats) + e * GS + 8 * q + 4 * hi;
                        const f32x4 v = gp_quad(q);
                        f32x4 o = *reinterpret_cast<const f32x4*>(gp);
                        o[0] += v[0]; o[1] += v[1]; o[2] += v[2]; o[3] += v[3];
                        *reinterpret_cast<f32x4*>(gp) = o;
                    }
            }
        }
        if (b + 1 < p.n) ws_commit(b + 1);
        wg_barrier();  // B2: every wave is done reading X; the gate partials are complete
        stamp(5);

        if (b + 1 < p.n && w < NT) {  // new scalar state -> X (B operand of the next block)
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int ot = w + NW * t;
                if (ot < NT) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = 32 * ot + 8 * q + 4 * hi;
                        if (c + 3 < so) {
                            f32x4 v = {ynew[t][4 * q], ynew[t][4 * q + 1], ynew[t][4 * q + 2], ynew[t][4 * q + 3]};
                            *reinterpret_cast<f32x4*>(X + e * KS + c) = v;
                        }
                    }
                }
            }
        }
        RB_LAUNDER();
        // ---- epilogue -----------------------------------------------------------------------------------------------------
        if (vo > 0) {
#pragma unroll 1
            for (int o = psub; o < vo; o += TPR) {
                float g = 1.f;
                if (scalar_gate) {
                    float s = bg[o];
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) s += lds[DM(o_st) + ww * DM(st_floats) + prow * GS + o];
                    g = gcp_sigmoid(s);
                }
                float u0 = 0.f, u1 = 0.f, u2 = 0.f;
                if ((H & 3) == 0) {
                    const float* vh = VH + prow * HS;
                    for (int h = 0; h < H; h += 4) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(wu + o * WSU + h);
                        const f32x4 a = *reinterpret_cast<const f32x4*>(vh + 3 * h);
                        const f32x4 bq = *reinterpret_cast<const f32x4*>(vh + 3 * h + 4);
                        const f32x4 d = *reinterpret_cast<const f32x4*>(vh + 3 * h + 8);
                        u0 = fmaf(wv[0], a[0], u0); u1 = fmaf(wv[0], a[1], u1); u2 = fmaf(wv[0], a[2], u2);
                        u0 = fmaf(wv[1], a[3], u0); u1 = fmaf(wv[1], bq[0], u1); u2 = fmaf(wv[1], bq[1], u2);
                        u0 = fmaf(wv[2], bq[2], u0); u1 = fmaf(wv[2], bq[3], u1); u2 = fmaf(wv[2], d[0], u2);
                        u0 = fmaf(wv[3], d[1], u0); u1 = fmaf(wv[3], d[2], u1); u2 = fmaf(wv[3], d[3], u2);
                    }
                } else {
                    for (int h = 0; h < H; ++h) {
                        const float wv = wu[o * WSU + h];
                        u0 = fmaf(wv, VH[prow * HS + 3 * h + 0], u0);
                        u1 = fmaf(wv, VH[prow * HS + 3 * h + 1], u1);
                        u2 = fmaf(wv, VH[prow * HS + 3 * h + 2], u2);
                    }
                }
                float* vp = V + prow * VS + 3 * o;
                float x0 = 0.f, x1 = 0.f, x2 = 0.f;
                if (p.vres || B.residual) { x0 = vp[0]; x1 = vp[1]; x2 = vp[2]; }
                if (p.vres) { u0 += x0; u1 += x1; u2 += x2; }
                if (p.vmode == GCP_VMODE_SELF_GATE)
                    g = gcp_actf<PWL>(B.act_v, ns_v, slope, sqrtf(u0 * u0 + u1 * u1 + u2 * u2 + 1e-8f) + 1e-8f);
                u0 *= g; u1 *= g; u2 *= g;
                if (B.residual) { u0 += x0; u1 += x1; u2 += x2; }
                vp[0] = u0; vp[1] = u1; vp[2] = u2;
                if (scalar_gate && B.gate && prow < nvalid) B.gate[(int64_t)(r0 + prow) * vo + o] = g;
            }
        }
        stamp(6);
        wg_barrier();  // B3: the vector tile is updated
        if (vo > 0 && B.v_out)
            wg_tile_store<NTH>(B.v_out + (int64_t)r0 * 3 * vo, V, VS, 3 * vo, nvalid, tid, ((3 * vo) & 3) == 0 && wg_aligned16(B.v_out));
        stamp(7);
    };

    run_block(std::true_type{}, 0);
    for (int b = 1; b < p.n; ++b) run_block(std::false_type{}, b);
}
#undef B
#undef p
#undef DM
#undef DB
#undef RB_LAUNDER
#undef WG_RELOAD
#undef WG_LAUNDER

// The scalar_out weight as the pack kernel reads it: logical W'[r][c], r < so, c < K, is W[r * ld + col(c)] (or, transposed,
// W[col(c) * ld + r]) with col(c) running through up to three column ranges of the stored matrix -- so that column slices of a
// Linear (project-then-gather keeps [e | norms | frame scalars] of scalar_out's columns; a projection uses one source's
// columns) and transposed weights (the input gradient of a Linear is a Linear with W^T) are packed without copies.
struct WgPackView {
    const float* W;
    int ld, trans, nseg;
    int start[3], len[3];
};
__device__ __forceinline__ float wg_view_at(const WgPackView& v, int r, int c) {
    int pc = -1;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k < v.nseg && pc < 0) {
            if (c < v.len[k]) pc = v.start[k] + c;
            else c -= v.len[k];
        }
    }
    return v.trans ? v.W[(int64_t)pc * v.ld + r] : v.W[(int64_t)r * v.ld + pc];
}

__device__ __forceinline__ void wg_pack_element(const WgShape& S, const WgPackView& view, const float* __restrict__ Wg,
                                                float* __restrict__ out, int64_t idx) {
    if (idx >= S.total) return;
    const int i = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    const int m = lane & 31, hi = lane >> 5;
    const int G4 = 4 * S.NT;
    float v = 0.f;
    if (idx < S.offG1) {
        const int64_t blk = idx >> 8;
        const int ot = (int)(blk / S.KG), g = (int)(blk - (int64_t)ot * S.KG);
        const int r = 32 * ot + m, c = 8 * g + 4 * hi + i;
        if (r < S.so && c < S.K) v = wg_view_at(view, r, c);
    } else if (idx < S.offA2) {
        const int gall = (int)((idx - S.offG1) >> 8);
        const int gm = gall / G4, g = gall - gm * G4;
        const int c = 8 * g + 4 * hi + i, o = 32 * gm + m;
        if (o < S.vo && c < S.so) v = Wg[(int64_t)o * S.so + c];
    } else if (idx >= S.offA1b) {
        int64_t blk = (idx - S.offA1b) >> 8;  // (ot, slab, term)
        const int term = (int)(blk % GCP_W6_TERMS); blk /= GCP_W6_TERMS;
        const int ot = (int)(blk / S.NSLf), j = (int)(blk - (int64_t)ot * S.NSLf);
        const int r = 32 * ot + m;
        unsigned bits = 0;
        for (int h2 = 0; h2 < 2; ++h2) {
            const int c = 16 * j + 8 * hi + 2 * i + h2;
            const float wv = (r < S.so && c < S.K) ? wg_view_at(view, r, c) : 0.f;
            bits |= (GCP_ARITH_F16X2 ? gcp_f16x2_wterm(wv, term) : gcp_bf16x3_term(wv, term)) << (16 * h2);
        }
        v = __uint_as_float(bits);
    } else if (idx >= S.offA2b) {
        int64_t blk = (idx - S.offA2b) >> 8;  // (kt, slab, term)
        const int term = (int)(blk % GCP_W6_TERMS); blk /= GCP_W6_TERMS;
        const int NSL = 2 * S.NT;
        const int kt = (int)(blk / NSL), j = (int)(blk - (int64_t)kt * NSL);
        const int c = 32 * kt + m;
        unsigned bits = 0;
        for (int h2 = 0; h2 < 2; ++h2) {
            const int ip = 2 * i + h2;  // element of the lane's eight
            const int r = 32 * (j >> 1) + 16 * (j & 1) + 8 * (ip >> 2) + 4 * hi + (ip & 3);
            const float wv = (r < S.so && c < S.K) ? wg_view_at(view, r, c) : 0.f;
            bits |= (GCP_ARITH_F16X2 ? gcp_f16x2_wterm(wv, term) : gcp_bf16x3_term(wv, term)) << (16 * h2);
        }
        v = __uint_as_float(bits);
    } else if (idx < S.offG2) {
        const int64_t blk = (idx - S.offA2) >> 8;
        const int kt = (int)(blk / G4), g = (int)(blk - (int64_t)kt * G4);
        const int r = 8 * g + 4 * hi + i, c = 32 * kt + m;
        if (r < S.so && c < S.K) v = wg_view_at(view, r, c);
    } else {
        const int64_t blk = (idx - S.offG2) >> 8;
        const int ot = (int)(blk / S.VG), g = (int)(blk - (int64_t)ot * S.VG);
        const int r = 8 * g + 4 * hi + i, c = 32 * ot + m;
        if (r < S.vo && c < S.so) v = Wg[(int64_t)r * S.so + c];
    }
    out[idx] = v;
}

__global__ __launch_bounds__(256) void wg_pack_kernel(WgShape S, WgPackView view, const float* __restrict__ Wg,
                                                      float* __restrict__ out) {
    wg_pack_element(S, view, Wg, out, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

// Several images in one launch (gcpnet_wg_pack_multi): the 256-element blocks of all jobs in one 1-D grid; a block finds its job in
// the prefix sums of the block counts and re-derives the job's shape from its seven dimensions.
constexpr int WG_PACK_MULTI_MAX = 24;
struct WgPackMultiArgs {
    int n;
    int block_start[WG_PACK_MULTI_MAX + 1];
    int dims[WG_PACK_MULTI_MAX][7];  // si, vi, so, vo, hidden, use_frames, gated
    WgPackView view[WG_PACK_MULTI_MAX];
    const float* Wg[WG_PACK_MULTI_MAX];
    float* out[WG_PACK_MULTI_MAX];
};
__global__ __launch_bounds__(256) void wg_pack_multi_kernel(WgPackMultiArgs a) {
    int j = 0;
    while (j + 1 < a.n && (int)blockIdx.x >= a.block_start[j + 1]) ++j;
    const int* d = a.dims[j];
    const WgShape S = wg_shape(d[0], d[1], d[2], d[3], d[4], d[5], d[6]);
    wg_pack_element(S, a.view[j], a.Wg[j], a.out[j], (int64_t)((int)blockIdx.x - a.block_start[j]) * 256 + threadIdx.x);
}

template <int SHP, int NW, int MT, bool B6 = false>
bool wg_fwd_is_shape(const WgFwdParams& p, size_t lds_bytes) {
    constexpr WgFwdDims CF = wg_fwd_dims(WgFwdShape<SHP>::S, WgFwdShape<SHP>::V, WgFwdShape<SHP>::HID, NW, MT, B6);
    if (p.n < 2 || p.vmode != GCP_VMODE_SCALAR_GATE || lds_bytes != (size_t)CF.lds_floats * sizeof(float)) return false;
    bool same = true;
#define X(f) same = same && p.f == CF.f;
    WG_FWD_DIMS(X)
#undef X
    for (int b = 0; b < p.n; ++b) {
#define X(f) same = same && p.blk[b].f == CF.f;
        WG_FWD_BLK_DIMS(X)
#undef X
    }
    return same;
}

template <int NW, int MT, int SHP = 0, bool B6 = false, bool G2 = false>
int launch_fwd(const WgFwdParams& p, bool pwl, size_t lds_bytes, hipStream_t st) {
    auto go = [&](auto kern) -> int {
        if (lds_bytes > 64 * 1024) {
            hipError_t err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (err != hipSuccess) return (int)err;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)gcp_cdiv(p.rows, 32)), dim3(64 * NW), lds_bytes, st, p);
        GCP_HIP_CHECK_LAUNCH();
        return 0;
    };
    if constexpr (SHP != 0) return go(gcp_wg_fwd_kernel<NW, MT, true, SHP, B6>);  // (compile-time shapes: PWL activations only)
    return pwl ? go(gcp_wg_fwd_kernel<NW, MT, true, 0, B6, G2>) : go(gcp_wg_fwd_kernel<NW, MT, false, 0, B6, G2>);
}

}  // namespace

extern "C" int64_t gcpnet_wg_pack_floats(int si, int vi, int so, int vo, int hidden, int use_frames, int gated) {
    return wg_shape(si, vi, so, vo, hidden, use_frames, gated).total;
}

extern "C" int gcpnet_wg_pack_view(const gcp2_weights_t* w, int gated, const float* W, int ld, int trans, int nseg,
                                   const int* start, const int* len, float* out, void* stream) {
    if (!w || !out || !W || nseg < 1 || nseg > 3 || !start || !len || ld < 1) return GCPNET_E_BADARG;
    const WgShape S = wg_shape(w->si, w->vi, w->so, w->vo, w->hidden, w->use_frames, gated);
    if (S.gated && !w->w_gate) return GCPNET_E_BADARG;
    WgPackView v;
    v.W = W; v.ld = ld; v.trans = trans; v.nseg = nseg;
    int tot = 0;
    for (int k = 0; k < 3; ++k) {
        v.start[k] = k < nseg ? start[k] : 0;
        v.len[k] = k < nseg ? len[k] : 0;
        if (v.start[k] < 0 || v.len[k] < 0) return GCPNET_E_BADARG;
        tot += v.len[k];
    }
    if (tot != S.K) return GCPNET_E_BADARG;
    hipLaunchKernelGGL(wg_pack_kernel, dim3((unsigned)gcp_cdiv((int)S.total, 256)), dim3(256), 0, (hipStream_t)stream, S, v, w->w_gate,
                       out);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int gcpnet_wg_pack_multi(int n, const gcp_wg_pack_job_t* jobs, void* stream) {
    if (n <= 0 || !jobs) return GCPNET_E_BADARG;
    for (int i0 = 0; i0 < n; i0 += WG_PACK_MULTI_MAX) {
        WgPackMultiArgs a;
        a.n = min(WG_PACK_MULTI_MAX, n - i0);
        int blocks = 0;
        for (int j = 0; j < a.n; ++j) {
            const gcp_wg_pack_job_t& J = jobs[i0 + j];
            const gcp2_weights_t& w = J.w;
            if (!J.out || !J.W || J.nseg < 1 || J.nseg > 3 || J.ld < 1) return GCPNET_E_BADARG;
            const WgShape S = wg_shape(w.si, w.vi, w.so, w.vo, w.hidden, w.use_frames, J.gated);
            if (S.gated && !w.w_gate) return GCPNET_E_BADARG;
            WgPackView& v = a.view[j];
            v.W = J.W; v.ld = J.ld; v.trans = J.trans; v.nseg = J.nseg;
            int tot = 0;
            for (int k = 0; k < 3; ++k) {
                v.start[k] = k < J.nseg ? J.start[k] : 0;
                v.len[k] = k < J.nseg ? J.len[k] : 0;
                if (v.start[k] < 0 || v.len[k] < 0) return GCPNET_E_BADARG;
                tot += v.len[k];
            }
            if (tot != S.K) return GCPNET_E_BADARG;
            const int dims[7] = {w.si, w.vi, w.so, w.vo, w.hidden, w.use_frames, J.gated};
            for (int k = 0; k < 7; ++k) a.dims[j][k] = dims[k];
            a.Wg[j] = w.w_gate;
            a.out[j] = J.out;
            a.block_start[j] = blocks;
            blocks += gcp_cdiv((int)S.total, 256);
        }
        a.block_start[a.n] = blocks;
        hipLaunchKernelGGL(wg_pack_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
        GCP_HIP_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int gcpnet_wg_pack(const gcp2_weights_t* w, int gated, float* out, void* stream) {
    if (!w || !w->w_scalar) return GCPNET_E_BADARG;
    const WgShape S = wg_shape(w->si, w->vi, w->so, w->vo, w->hidden, w->use_frames, gated);
    const int start = 0, len = S.K;
    return gcpnet_wg_pack_view(w, gated, w->w_scalar, S.K, 0, 1, &start, &len, out, stream);
}

// Waves per workgroup / output tiles per wave for an output width (shared with the Python side through gcpnet_wg_config).
static void wg_pick(int so, int& NW, int& MT) {
    if (so <= 128) { NW = 4; MT = 1; }
    else if (so <= 256) { NW = 8; MT = 1; }
    else { NW = 8; MT = 2; }
}

extern "C" int gcpnet_wg_forward(int rows, const float* s_in, const float* v_in, const float* frames, const gcp_concat_t* s_add,
                                 const gcp_concat_t* v_add, int n, const gcp_wg_block_t* blocks, void* stream) {
    if (rows < 0 || n <= 0 || n > GCP_WG_MAX_BLOCKS || !blocks || !s_in) return GCPNET_E_BADARG;
    const gcp2_weights_t& w0 = blocks[0].w;
    const int so = w0.so, vo = w0.vo;
    if (w0.vi < 0 || (w0.vi == 0 && (vo > 0 || n > 1))) return GCPNET_E_UNSUPPORTED;  // (vi == 0: one scalar-only Linear block)
    if (w0.vi > 0 && !v_in) return GCPNET_E_BADARG;
    if ((so & 3) || so < 4) return GCPNET_E_UNSUPPORTED;
    if (vo > 64 && blocks[0].o.vmode == GCP_VMODE_SCALAR_GATE) return WG_UNSUPPORTED("scalar gate with more than 64 output vectors");  // (two 32-row tiles of gate outputs)
    WgFwdParams p;
    p.rows = rows; p.s_in = s_in; p.v_in = v_in; p.frames = frames;
    p.s_add.n = 0; p.v_add.n = 0;
    if (s_add) p.s_add = *s_add;
    if (v_add) p.v_add = *v_add;
    if (p.s_add.n < 0 || p.s_add.n > GCP_MAX_SEG || p.v_add.n < 0 || p.v_add.n > GCP_MAX_SEG) return GCPNET_E_BADARG;
    const gcp2_opts_t& o0 = blocks[0].o;
    p.so = so; p.vo = vo; p.nf = (w0.use_frames && w0.vi > 0 ? 9 : 0); p.e3 = o0.e3; p.vmode = vo > 0 ? o0.vmode : GCP_VMODE_NONE;
    p.vres = o0.vector_residual; p.slope = o0.slope;
    if (p.nf && !frames) return GCPNET_E_BADARG;
    int NW, MT;
    wg_pick(so, NW, MT);
    // A single non-residual block with few reduction columns on many rows -- the first message GCP of BASELINE configs[4] after
    // project-then-gather: K = 58, so = 256, 10^6 rows -- is bound by the per-tile latencies of its gathers and barriers, not by its
    // MFMAs: four waves with two output tiles each need half the staging LDS, so that two workgroups share a CU
    // (GCPNET_WG_FWD_HEAD4=0 keeps the eight-wave form)
    static const bool head4 = !(getenv("GCPNET_WG_FWD_HEAD4") && getenv("GCPNET_WG_FWD_HEAD4")[0] == '0');
    static const int head4_k = getenv("GCPNET_WG_FWD_HEAD4_K") ? atoi(getenv("GCPNET_WG_FWD_HEAD4_K")) : 96;  // (tuning knob: K limit)
    const bool small_k = n == 1 && !blocks[0].residual && so > 128 && so <= 256 && vo <= 32 && w0.si + w0.hidden + 9 <= head4_k;
    if (head4 && small_k) { NW = 4; MT = 2; }
    p.NT = gcp_cdiv(so, 32);
    p.NG = gcp_cdiv(p.NT, NW * MT);
    p.n = n;
    const bool gated = p.vmode == GCP_VMODE_SCALAR_GATE;
    bool pwl = true;
    int kmax = 0, vmax = vo, hmax = 1, wsmax = 0, nslf_max = 0;
    auto misaligned = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) != 0; };
    for (int b = 0; b < n; ++b) {
        const gcp_wg_block_t& c = blocks[b];
        const gcp2_weights_t& w = c.w;
        if (!w.pack || !w.b_scalar || (w.vi > 0 && !w.w_down) || (w.vo > 0 && !w.w_up)) return GCPNET_E_BADARG;
        if (w.so != so || w.vo != vo || (w.vi > 0) != (w0.vi > 0) || (w.use_frames && w.vi > 0 ? 9 : 0) != p.nf) return GCPNET_E_UNSUPPORTED;
        if (p.nf && !w.w_frames) return GCPNET_E_BADARG;
        if (c.o.vmode != o0.vmode || c.o.e3 != o0.e3 || c.o.vector_residual != o0.vector_residual || c.o.slope != o0.slope)
            return GCPNET_E_UNSUPPORTED;
        if (b > 0 && (w.si != so || w.vi != vo)) return GCPNET_E_UNSUPPORTED;  // later blocks run on the state
        if (c.residual && (w.si != so || w.vi != vo)) return GCPNET_E_BADARG;
        if (o0.vector_residual && w.vi != vo) return GCPNET_E_BADARG;
        if (gated && !w.b_gate) return GCPNET_E_BADARG;
        if (misaligned(c.s_out) || misaligned(c.s_pre) || misaligned(w.pack) || misaligned(w.b_scalar)) return GCPNET_E_UNSUPPORTED;
        const WgShape S = wg_shape(w.si, w.vi, so, vo, w.hidden, w.use_frames, gated);
        WgBlk& k = p.blk[b];
        k.pk = w.pack; k.b_scalar = w.b_scalar; k.b_gate = w.b_gate; k.w_down = w.w_down; k.w_frames = w.w_frames; k.w_up = w.w_up;
        k.s_out = c.s_out; k.v_out = c.v_out; k.s_pre = c.s_pre; k.gate = c.gate;
        k.offG1 = S.offG1; k.offA1b = S.offA1b;
        nslf_max = max(nslf_max, S.NSLf);
        k.act_s = c.o.act_s; k.act_v = c.o.act_v;
        k.si = w.si; k.vi = w.vi; k.H = S.H; k.K = S.K; k.KG = S.KG;
        k.residual = c.residual;
        k.tb = (c.s_out_tb ? 1 : 0) | (c.s_pre_tb ? 2 : 0);
        if (k.tb && (so & 31)) return GCPNET_E_BADARG;  // tile-blocked tensors: whole 32-column tiles
        pwl = pwl && gcp_is_pwl(c.o.act_s) && gcp_is_pwl(c.o.act_v);
        kmax = max(kmax, S.KP);
        vmax = max(vmax, w.vi);
        hmax = max(hmax, S.H);
        const int HF = S.H + (p.nf ? 3 : 0);
        wsmax = max(wsmax, gcp_round_up(HF * wg_stride(w.vi) + vo * wg_stride(max(S.H, 1)) + vo, 4) + gcp_round_up(so, 4));
    }
    if ((n > 1 || blocks[0].residual) && p.NG != 1) return GCPNET_E_UNSUPPORTED;
    if (n > 1) kmax = max(kmax, gcp_round_up(so, 4));
    for (int k = 0; k < p.s_add.n; ++k)
        if (!p.s_add.ptr[k] || p.s_add.dim[k] != so || misaligned(p.s_add.ptr[k])) return GCPNET_E_BADARG;
    {
        const WgShape S0 = wg_shape(w0.si, w0.vi, so, vo, w0.hidden, w0.use_frames, gated);
        const int hfp = gcp_round_up(S0.H + (p.nf ? 3 : 0), 4);
        for (int k = 0; k < p.v_add.n; ++k)
            if (!p.v_add.ptr[k] || p.v_add.dim[k] != hfp) return GCPNET_E_BADARG;
    }
    if (rows == 0) return 0;
    p.KS = wg_stride(kmax);
    p.VS = wg_stride(3 * vmax);
    p.HS = wg_stride(3 * hmax);
    p.GS = wg_stride(gcp_round_up(max(vo, 1), 8));
    p.ws_floats = gcp_round_up(wsmax, 4);
    int off = 0;
    p.o_x = off; off += 32 * p.KS;
    p.st_floats = max(WG_STAGE_FLOATS, gated ? 32 * p.GS : 0);
    p.o_st = off; off += NW * p.st_floats;
    p.o_ws = off; off += (n > 1 ? 2 : 1) * p.ws_floats;
    p.o_v = off; off += gcp_round_up(32 * p.VS, 4);
    p.o_vh = off; off += gcp_round_up(32 * p.HS, 4);
    p.o_fr = off; off += 32 * 9;
    // K loop on the bf16 pipe: one output tile per wave, by default for the 8-wave shapes (one workgroup per CU anyway; at 4
    // waves the operand planes cost the third workgroup of a CU: GCPNET_WG_FWD_B6=all), if the planes fit
    static const char* b6_env = getenv("GCPNET_WG_FWD_B6");  // "0" = off, "all" = also 4-wave shapes
    const bool g2 = gated && vo > 32;  // two tiles of gate outputs: the run-time-shape fp32 form
    bool b6 = !g2 && MT == 1 && p.NG == 1 && (g_gcp_fp32_mfma < 0 ? !(b6_env && b6_env[0] == '0') : g_gcp_fp32_mfma == 0) &&
              (NW == 8 || (b6_env && b6_env[0] == 'a'));
    if (b6 && (size_t)(off + nslf_max * 768) * sizeof(float) > 160 * 1024) b6 = false;
    p.o_xp = off; off += b6 ? nslf_max * 768 : 0;
    const size_t lds_bytes = (size_t)off * sizeof(float);
    if (lds_bytes > 160 * 1024) return WG_UNSUPPORTED("the tile set does not fit in 160 KB of LDS");
    p.stamps = g_gcp_phase_buf; p.stamp_cap = g_gcp_phase_cap;
    hipStream_t st = (hipStream_t)stream;
    if (g2) {
        if (NW == 4 && MT == 1) return launch_fwd<4, 1, 0, false, true>(p, pwl, lds_bytes, st);
        if (NW == 4) return WG_UNSUPPORTED("two gate tiles in the four-wave / two-tile form");
        if (MT == 1) return launch_fwd<8, 1, 0, false, true>(p, pwl, lds_bytes, st);
        return launch_fwd<8, 2, 0, false, true>(p, pwl, lds_bytes, st);
    }
    if (pwl && !getenv("GCPNET_WG_FWD_NOSHAPE")) {
        if (NW == 4 && !b6 && wg_fwd_is_shape<1, 4, 1>(p, lds_bytes)) return launch_fwd<4, 1, 1>(p, pwl, lds_bytes, st);
        if (NW == 4 && b6 && wg_fwd_is_shape<1, 4, 1, true>(p, lds_bytes)) return launch_fwd<4, 1, 1, true>(p, pwl, lds_bytes, st);
        if (NW == 8 && MT == 1 && !b6 && wg_fwd_is_shape<2, 8, 1>(p, lds_bytes)) return launch_fwd<8, 1, 2>(p, pwl, lds_bytes, st);
        if (NW == 8 && MT == 1 && b6 && wg_fwd_is_shape<2, 8, 1, true>(p, lds_bytes)) return launch_fwd<8, 1, 2, true>(p, pwl, lds_bytes, st);
    }
    if (b6) return NW == 4 ? launch_fwd<4, 1, 0, true>(p, pwl, lds_bytes, st) : launch_fwd<8, 1, 0, true>(p, pwl, lds_bytes, st);
    if (NW == 4) return MT == 1 ? launch_fwd<4, 1>(p, pwl, lds_bytes, st) : launch_fwd<4, 2>(p, pwl, lds_bytes, st);
    if (MT == 1) return launch_fwd<8, 1>(p, pwl, lds_bytes, st);
    return launch_fwd<8, 2>(p, pwl, lds_bytes, st);
}
