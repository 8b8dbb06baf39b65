// GCP2 backward (data path) on gfx950.  There is no reference code for this: the reference gets it from autograd
// over src/models/components/gcpnet.py:394-468; this kernel is the hand-derived adjoint of gcp2_fwd.hip.
//
// Same wave-autonomous tiling as the forward (32 rows per wavefront, no inter-wave barriers).  The two GEMM-shaped
// steps run on v_mfma_f32_32x32x2_f32 in transposed form:
//   * gate adjoint      ds_pre_gate^T[so, rows] = Wg^T[so, vo]  x dgate^T[vo, rows]   (B from LDS)
//   * scalar_out adjoint dmerged^T[K, rows]     = W^T[K, so]    x ds_pre^T[so, rows]
//     where the B operand is the ds_pre accumulator tile ITSELF: a 32x32 C/D register (t, r) of lane half `hi` holds
//     column j = 32t + (r&3) + 8(r>>2) + 4hi of row lane&31, so using it as the B fragment of a k-pair step whose two
//     reduction indices are (j0, j0+4) needs no data movement at all; the packed A image is built for that pairing.
// Weight gradients are reductions over rows and are left to gcpnet_tn_gemm, fed by the per-row scratch written here.
#include "common.h"
#include "tile_io.h"

#include <type_traits>

namespace {

struct BwdParams {
    int rows;
    gcp_concat_t v_in;
    gcp_concat_t v_add;  // pre-projected vector inputs, as in the forward (gcp2_fwd.hip)
    const float* frames;
    gcp2_weights_t w;
    gcp2_opts_t o;
    const float* s_pre;
    const float* gate;
    const float* d_s_out;
    const float* d_v_out;
    float* d_s_in;
    float* d_v_in;
    gcp2_bwd_scratch_t sc;
    unsigned long long* stamps;
    long long stamp_cap;
    GcpShape sh;
};

struct BwdLds {
    int VS, HS, NS_, US, GS2, DS, FS;
    int o_vt, o_vht, o_rn, o_dvut, o_dvot, o_dgt, o_dext, o_dvhf, o_fr, o_sw, total;
};

__host__ __device__ inline BwdLds bwd_lds(const GcpShape& s) {
    BwdLds l;
    l.VS = gcp_odd(3 * s.vi);
    l.HS = gcp_odd(3 * s.H);
    l.NS_ = gcp_odd(s.H);
    l.US = gcp_odd(3 * s.vo);
    l.GS2 = gcp_odd(2 * s.NOO);
    l.DS = gcp_odd(s.H + 9);
    l.FS = gcp_odd(3 * (s.H + 3));
    l.o_vt = 0;
    l.o_vht = l.o_vt + 32 * l.VS;
    l.o_rn = l.o_vht + 32 * l.HS;
    l.o_dvut = l.o_rn + 32 * l.NS_;
    l.o_dvot = l.o_dvut + 32 * l.US;
    l.o_dgt = l.o_dvot + 32 * l.US;
    l.o_dext = l.o_dgt + 32 * l.GS2;
    l.o_dvhf = l.o_dext + 32 * l.DS;
    l.o_fr = l.o_dvhf + 32 * l.FS;
    l.o_sw = l.o_fr + 32 * 9;
    l.total = l.o_sw + gcp_small_w_lds_floats(s.vi, s.H, s.vo, s.nf);
    return l;
}

// -DGCP_BWD_FINE: the 8 stamp slots subdivide phases 1-2 instead of marking the 5 phase boundaries (tools/phase_timing.py)
#ifdef GCP_BWD_FINE
#define STAMP(k) ((void)0)
#define FSTAMP(k) gcp_stamp(p.stamps, p.stamp_cap, k, lane)
#else
#define STAMP(k) gcp_stamp(p.stamps, p.stamp_cap, k, lane)
#define FSTAMP(k) ((void)0)
#endif
#define load4 gcp_load4
#define store4 gcp_store4

template <int N>
struct WFragB;
template <>
struct WFragB<1> {
    float v[1];
    __device__ __forceinline__ void load(const float* p) { v[0] = p[0]; }
};
template <>
struct WFragB<2> {
    float v[2];
    __device__ __forceinline__ void load(const float* p) {
        float2 t = *reinterpret_cast<const float2*>(p);
        v[0] = t.x; v[1] = t.y;
    }
};
template <>
struct WFragB<4> {
    float v[4];
    __device__ __forceinline__ void load(const float* p) {
        float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
};

// SINGLE: one output group (so <= 128) -> straight-line fast path with the tile's s_pre / d(s_out) held in registers.
template <int NTG, int NUG, bool PWL, bool SINGLE>
__global__ __launch_bounds__(GCP_WAVE, 2) void gcp2_bwd_kernel(BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const GcpShape& S = p.sh;
    const BwdLds L = bwd_lds(S);
    int lane = threadIdx.x;
    int e = lane & 31, hi = lane >> 5;
    const int r0 = blockIdx.x * GCP_TILE_ROWS;
    const int rows = p.rows;
    int row = r0 + e;
    bool row_ok = row < rows;
    float* vt = lds + L.o_vt;
    float* vht = lds + L.o_vht;
    float* rn = lds + L.o_rn;
    float* dvut = lds + L.o_dvut;
    float* dvot = lds + L.o_dvot;  // d(v_out) tile, staged once with coalesced loads
    float* dgt = lds + L.o_dgt;    // sigmoid(gate) tile on entry, overwritten in place by d(gate)
    float* dext = lds + L.o_dext;
    float* dvhf = lds + L.o_dvhf;
    float* fr = lds + L.o_fr;
    const int si = S.si, vi = S.vi, so = S.so, vo = S.vo, H = S.H, HF = S.H + 3;
    // row strides of the per-row scratch handed to the weight-gradient GEMMs: multiples of 4 floats (16-byte DMA pieces)
    const int EP = gcp_round_up(S.H + S.nf, 4), VOP = gcp_round_up(S.vo, 4);
    const float slope = p.o.slope;
    const float ns_s = gcp_neg_slope(p.o.act_s, slope), ns_v = gcp_neg_slope(p.o.act_v, slope);
    const bool scalar_gate = (p.o.vmode == GCP_VMODE_SCALAR_GATE) && vo > 0 && vi > 0;
    const bool has_vec = vi > 0;
    const bool has_vout = has_vec && vo > 0;

    STAMP(0);
    FSTAMP(0);
    const bool vec_so = (so & 3) == 0, vec_si = (si & 3) == 0;
    constexpr bool single = SINGLE;
    const float* __restrict__ sp_ptr = p.s_pre;
    const float* __restrict__ dso_ptr = p.d_s_out;
    float* __restrict__ dsp_ptr = p.sc.ds_pre;
    f32x16 spr[NTG], dyr[NTG];
    // rows of the pre-projected vector tables this lane's row gathers ([n_src, 3, HF'] each): the gather indices are requested
    // first of all, so that they arrive with the tile
    const int HFPQ = gcp_round_up(HF, 4);
    const float* vq[GCP_MAX_SEG];
#pragma unroll
    for (int k = 0; k < GCP_MAX_SEG; ++k) {
        vq[k] = nullptr;
        if (k < p.v_add.n) {
            const int rc = min(row, rows - 1);
            vq[k] = p.v_add.ptr[k] + (p.v_add.idx[k] ? (int64_t)p.v_add.idx[k][rc] : (int64_t)rc) * 3 * HFPQ;
        }
    }
    // ---- 1. stage vectors / frames, recompute vh, its norms and the frame scalars ---------------------------
    if (has_vec) {  // inputs, upstream vector gradients, gates and frames: one memory round trip
        GcpSegBuf<8> vb0, gb0, tb0;
        gcp_seg_issue(vb0, p.v_in.ptr[0], p.v_in.idx[0], 3 * p.v_in.dim[0], r0, rows, vt, L.VS, 0, lane);
        if (has_vout) gcp_seg_issue(gb0, p.d_v_out, nullptr, 3 * vo, r0, rows, dvot, L.US, 0, lane);
        if (scalar_gate) gcp_seg_issue(tb0, p.gate, nullptr, vo, r0, rows, dgt, L.GS2, 0, lane);
        if (S.nf) gcp_load_frames(p.frames, r0, rows, fr, lane);
        gcp_seg_commit(vb0, vt, L.VS, 0);
        if (has_vout) gcp_seg_commit(gb0, dvot, L.US, 0);
        if (scalar_gate) gcp_seg_commit(tb0, dgt, L.GS2, 0);
        int coff = 3 * p.v_in.dim[0];
        for (int sg = 1; sg < p.v_in.n; ++sg) {
            gcp_load_segment(p.v_in.ptr[sg], p.v_in.idx[sg], 3 * p.v_in.dim[sg], r0, rows, vt, L.VS, coff, lane);
            coff += 3 * p.v_in.dim[sg];
        }
    }
    FSTAMP(1);
    const GcpSmallW sw = gcp_stage_small_weights(p.w, H, S.nf, lds + L.o_sw, lane);
    FSTAMP(2);
    for (int i = vo + hi; i < 2 * S.NOO; i += 2) dgt[e * L.GS2 + i] = 0.f;  // zero the gate-adjoint k padding
    // Single output group (so <= 128): s_pre and d(s_out) of the tile are requested here (after every load
    // phase 1 has to wait for: vmcnt retires in order), in the accumulator layout, and stay in flight
    // under phases 1-2; d(s_out) later doubles as the ResGCP pass-through term of the accumulator.
    // Shares of the pre-projected (gathered) sources in [vh | vf]: every value this lane will need (channels hi, hi + 2, ..),
    // requested in ONE go -- loads inside the run-time channel loops below would each be waited for on the spot, one gather
    // round trip per channel.  Up to two tables and 2 * QH channels (HF <= 16); anything else takes the loads in the loops.
    constexpr int QH = 8;
    const bool q_fast = p.v_add.n > 0 && p.v_add.n <= 2 && HF <= 2 * QH;  // wave-uniform
    float qa[2][QH][3], qf[2][2][3];
    if (q_fast) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float* t = vq[k < p.v_add.n ? k : 0];
#pragma unroll
            for (int i = 0; i < QH; ++i) {
                const int x = min(hi + 2 * i, HFPQ - 1);  // (clamped: always inside the row)
#pragma unroll
                for (int d = 0; d < 3; ++d) qa[k][i][d] = t[d * HFPQ + x];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int x = min(H + hi + 2 * j, HFPQ - 1);
#pragma unroll
                for (int d = 0; d < 3; ++d) qf[k][j][d] = t[d * HFPQ + x];
            }
        }
    }
    if constexpr (SINGLE) {
        gcp_request_acc_layout<NTG>(sp_ptr, row, so, 0, hi, row_ok, vec_so, spr);
        gcp_request_acc_layout<NTG>(dso_ptr, row, so, 0, hi, row_ok, vec_so, dyr);
    }
    gcp_wave_lds_sync();
    FSTAMP(3);
    // (the gathered shares of the pre-projected vector tables were requested before s_pre / d(s_out): they are older in the
    // in-order vmcnt queue, so consuming them here does not wait for those)
    float qsum[QH][3], qfsum[2][3];  // vector_down channels hi + 2 i; frame channels H + hi + 2 j
    if (q_fast) {
        const bool two = p.v_add.n > 1;
#pragma unroll
        for (int i = 0; i < QH; ++i)
#pragma unroll
            for (int d = 0; d < 3; ++d) qsum[i][d] = qa[0][i][d] + (two ? qa[1][i][d] : 0.f);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int d = 0; d < 3; ++d) qfsum[j][d] = qf[0][j][d] + (two ? qf[1][j][d] : 0.f);
    }
    if (has_vec) {
        const float* vrow = vt + e * L.VS;
#pragma unroll 1
        for (int h = hi, hidx = 0; h < H; h += 2, ++hidx) {
            const float* wd = sw.wd + h * vi;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            for (int c = 0; c < vi; ++c) {
                const float w = wd[c];
                a0 = fmaf(w, vrow[3 * c + 0], a0);
                a1 = fmaf(w, vrow[3 * c + 1], a1);
                a2 = fmaf(w, vrow[3 * c + 2], a2);
            }
            if (q_fast) {
#pragma unroll
                for (int i = 0; i < QH; ++i)  // (static register index: select)
                    if (i == hidx) { a0 += qsum[i][0]; a1 += qsum[i][1]; a2 += qsum[i][2]; }
            } else {
                for (int k = 0; k < p.v_add.n; ++k) {
                    a0 += vq[k][0 * HFPQ + h]; a1 += vq[k][1 * HFPQ + h]; a2 += vq[k][2 * HFPQ + h];
                }
            }
            vht[e * L.HS + 3 * h + 0] = a0;
            vht[e * L.HS + 3 * h + 1] = a1;
            vht[e * L.HS + 3 * h + 2] = a2;
            const float nr = sqrtf(a0 * a0 + a1 * a1 + a2 * a2 + 1e-8f);
            rn[e * L.NS_ + h] = 1.0f / nr;
            if (row_ok) p.sc.ext[(int64_t)row * EP + h] = nr + 1e-8f;
        }
        if (row_ok && hi == 0) {  // zero the stride padding
            for (int c = H + S.nf; c < EP; ++c) p.sc.ext[(int64_t)row * EP + c] = 0.f;
            if (scalar_gate)
                for (int c = vo; c < VOP; ++c) p.sc.dgate[(int64_t)row * VOP + c] = 0.f;
        }
        FSTAMP(4);
        if (S.nf) {
            const float* f = fr + e * 9;
            for (int k = hi; k < 3; k += 2) {
                const float* wf = sw.wf + k * vi;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f;
                for (int c = 0; c < vi; ++c) {
                    const float w = wf[c];
                    a0 = fmaf(w, vrow[3 * c + 0], a0);
                    a1 = fmaf(w, vrow[3 * c + 1], a1);
                    a2 = fmaf(w, vrow[3 * c + 2], a2);
                }
                if (q_fast) {  // frame channel k = hi + 2 j
                    const int j = (k - hi) >> 1;
                    a0 += j ? qfsum[1][0] : qfsum[0][0]; a1 += j ? qfsum[1][1] : qfsum[0][1]; a2 += j ? qfsum[1][2] : qfsum[0][2];
                } else {
                    for (int q = 0; q < p.v_add.n; ++q) {
                        a0 += vq[q][0 * HFPQ + H + k]; a1 += vq[q][1 * HFPQ + H + k]; a2 += vq[q][2 * HFPQ + H + k];
                    }
                }
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    float pr = f[3 * a + 0] * a0 + f[3 * a + 1] * a1 + f[3 * a + 2] * a2;
                    if (p.o.e3 && a == 1) {
                        // remember the sign for the adjoint of |.| in the dvhf tile's spare slot
                        dvhf[e * L.FS + 0 * HF + H + k] = pr < 0.f ? -1.f : 1.f;
                        pr = fabsf(pr);
                    }
                    if (row_ok) p.sc.ext[(int64_t)row * EP + H + 3 * k + a] = pr;
                }
            }
        }
        FSTAMP(5);
    }
    gcp_wave_lds_sync();
    STAMP(1);
    FSTAMP(6);

    asm volatile("" : "+v"(lane), "+v"(e), "+v"(hi));
    row = r0 + e;
    row_ok = row < rows;
    // ---- 2. adjoint of the vector epilogue (gcpnet.py:364-391) ------------------------------------------------
    if (has_vout) {
        for (int oc0 = hi; oc0 < vo; oc0 += 16) {  // 8 channels per lane per pass: all LDS reads, then the writes
            float du[8][3], dgv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int oc = oc0 + 2 * i;
                du[i][0] = du[i][1] = du[i][2] = 0.f; dgv[i] = 0.f;
                if (oc < vo) {
                    const float* wu = sw.wu + oc * H;
                    float u0 = 0.f, u1 = 0.f, u2 = 0.f;
                    for (int h = 0; h < H; ++h) {
                        const float w = wu[h];
                        u0 = fmaf(w, vht[e * L.HS + 3 * h + 0], u0);
                        u1 = fmaf(w, vht[e * L.HS + 3 * h + 1], u1);
                        u2 = fmaf(w, vht[e * L.HS + 3 * h + 2], u2);
                    }
                    if (p.o.vector_residual) {
                        u0 += vt[e * L.VS + 3 * oc + 0];
                        u1 += vt[e * L.VS + 3 * oc + 1];
                        u2 += vt[e * L.VS + 3 * oc + 2];
                    }
                    const float g0 = dvot[e * L.US + 3 * oc + 0], g1 = dvot[e * L.US + 3 * oc + 1], g2 = dvot[e * L.US + 3 * oc + 2];
                    float du0 = g0, du1 = g1, du2 = g2;
                    const float dot = g0 * u0 + g1 * u1 + g2 * u2;
                    if (scalar_gate) {
                        const float sg = dgt[e * L.GS2 + oc];
                        du0 = g0 * sg; du1 = g1 * sg; du2 = g2 * sg;
                        dgv[i] = dot * sg * (1.f - sg);
                    } else if (p.o.vmode == GCP_VMODE_SELF_GATE) {
                        const float rs = sqrtf(u0 * u0 + u1 * u1 + u2 * u2 + 1e-8f);
                        const float n = rs + 1e-8f;
                        const float a = gcp_actf<PWL>(p.o.act_v, ns_v, slope, n), da = gcp_dactf<PWL>(p.o.act_v, ns_v, slope, n);
                        const float coef = dot * da / rs;
                        du0 = g0 * a + coef * u0; du1 = g1 * a + coef * u1; du2 = g2 * a + coef * u2;
                    }
                    du[i][0] = du0; du[i][1] = du1; du[i][2] = du2;
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int oc = oc0 + 2 * i;
                if (oc < vo) {
                    dvut[e * L.US + 3 * oc + 0] = du[i][0];
                    dvut[e * L.US + 3 * oc + 1] = du[i][1];
                    dvut[e * L.US + 3 * oc + 2] = du[i][2];
                    if (scalar_gate) dgt[e * L.GS2 + oc] = dgv[i];
                    if (row_ok && scalar_gate) p.sc.dgate[(int64_t)row * VOP + oc] = dgv[i];
                }
            }
        }
    }
    gcp_wave_lds_sync();
    STAMP(2);
    FSTAMP(7);

    // ---- 3. ds_pre = d_s_out * act_s'(s_pre) + act_v'(s_pre) * (Wg^T dgate)  (per output group) ----------------
    // ---- 4. dmerged = W^T ds_pre, accumulated over the output groups, per merged-axis group ---------------------
    // keep the address arithmetic of the phases below from being hoisted above (and spilled across) phases 1-2
    asm volatile("" : "+v"(lane), "+v"(e), "+v"(hi));
    if constexpr (SINGLE) {  // first use of the s_pre / d(s_out) requests made before phase 1: zero what was out of range
        const bool okr = r0 + e < rows;
        gcp_mask_acc_layout<NTG>(so, 0, hi, okr, vec_so, spr);
        gcp_mask_acc_layout<NTG>(so, 0, hi, okr, vec_so, dyr);
    }
    row = r0 + e;
    row_ok = row < rows;
    // epilogue of one merged-axis group: d_s_in columns go to HBM, the vector extras (norm / frame-scalar adjoints) to LDS
    auto merged_epilogue = [&](auto nu_tag, int ug, f32x16(&acc2)[NUG]) {
        constexpr int NU = decltype(nu_tag)::value;
#pragma unroll
        for (int uu = 0; uu < NU; ++uu)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k0 = 32 * (ug * NUG + uu) + 8 * q + 4 * hi;
                if (k0 + 3 < si) {
                    store4(p.d_s_in, row, si, k0,
                           make_float4(acc2[uu][4 * q], acc2[uu][4 * q + 1], acc2[uu][4 * q + 2], acc2[uu][4 * q + 3]), row_ok, vec_si);
                } else {
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        const int k = k0 + x;
                        const float val = acc2[uu][4 * q + x];
                        if (k < si) {
                            if (row_ok) p.d_s_in[(int64_t)row * si + k] = val;
                        } else if (k < S.K) {
                            dext[e * L.DS + (k - si)] = val;
                        }
                    }
                }
            }
    };
    // scalar_out adjoint over ALL 16 * NTG k-pair steps of one output group, B operands = the ds_pre registers;
    // weight fragments rotate through three batches of 4 steps, requested two batches ahead and pinned there
    // (NU = live tiles of the group: the last merged-axis group is usually mostly padding, e.g. 1 of 4 for K = 153)
    auto data_gemm = [&](auto nu_tag, const float* wq, f32x16(&acc2)[NUG], f32x16(&ds)[NTG]) {
        constexpr int NU = decltype(nu_tag)::value;
        WFragB<NU> A0[4], A1[4], A2[4];
        auto ld = [&](WFragB<NU>(&a)[4], int st0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u].load(wq + (int64_t)min(st0 + u, NTG * 16 - 1) * 64 * NUG);
        };
        ld(A0, 0);
        ld(A1, 4);
        ld(A2, 8);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < NTG * 4; ++b) {
            WFragB<NU>(&a)[4] = (b % 3 == 0) ? A0 : ((b % 3 == 1) ? A1 : A2);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int st = b * 4 + u;
#pragma unroll
                for (int uu = 0; uu < NU; ++uu)
                    acc2[uu] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].v[uu], ds[st / 16][st % 16], acc2[uu], 0, 0, 0);
            }
            if ((b + 3) * 4 < NTG * 16) ld(a, (b + 3) * 4);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto tail_group = [&](auto nu_tag, int ug) {
        f32x16 acc2[NUG];
#pragma unroll
        for (int uu = 0; uu < NUG; ++uu)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[uu][r] = 0.f;
        data_gemm(nu_tag, p.w.pack + S.offB + ((int64_t)ug * S.NS * 64 + lane) * NUG, acc2, spr);
        merged_epilogue(nu_tag, ug, acc2);
    };

    if constexpr (SINGLE) {
        // ---- fast path (so <= 128): straight-line, so that register lifetimes are visible to the compiler:
        //      gate adjoint -> ds_pre (overwrites the s_pre registers) -> accumulator := d(s_out) (ResGCP) -> W^T ds_pre
        {
            f32x16 gacc[NTG];
#pragma unroll
            for (int t = 0; t < NTG; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) gacc[t][r] = 0.f;
            if (scalar_gate) {
                const float* wg = p.w.pack + S.offD + (int64_t)lane * NTG;
                for (int oo0 = 0; oo0 < S.NOO; oo0 += 8) {
                    WFragB<NTG> a[8];
                    float b[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int oo = min(oo0 + u, S.NOO - 1);
                        a[u].load(wg + (int64_t)oo * S.NG * 64 * NTG);
                        b[u] = dgt[e * L.GS2 + gcp_crow(oo, hi)];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (oo0 + u < S.NOO)
#pragma unroll
                            for (int t = 0; t < NTG; ++t)
                                gacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].v[t], b[u], gacc[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < NTG; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sp = spr[t][r];
                    float d = dyr[t][r] * gcp_dactf<PWL>(p.o.act_s, ns_s, slope, sp);
                    if (scalar_gate) d += gcp_dactf<PWL>(p.o.act_v, ns_v, slope, sp) * gacc[t][r];
                    spr[t][r] = row_ok ? d : 0.f;
                }
        }
#pragma unroll
        for (int t = 0; t < NTG; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                store4(dsp_ptr, row, so, 32 * t + 8 * q + 4 * hi,
                       make_float4(spr[t][4 * q], spr[t][4 * q + 1], spr[t][4 * q + 2], spr[t][4 * q + 3]), row_ok, vec_so);
        {
            f32x16 acc2[NUG];
#pragma unroll
            for (int uu = 0; uu < NUG; ++uu)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[uu][r] = (p.o.fused_residual && uu < NTG) ? dyr[uu < NTG ? uu : 0][r] : 0.f;
            data_gemm(std::integral_constant<int, NUG>{}, p.w.pack + S.offB + (int64_t)lane * NUG, acc2, spr);
            merged_epilogue(std::integral_constant<int, NUG>{}, 0, acc2);
        }
        for (int ug = 1; ug < S.NGK; ++ug) {
            const int live = min(NUG, gcp_cdiv(S.K, 32) - ug * NUG);  // wave-uniform
            if (NUG >= 4 && live == 1) tail_group(std::integral_constant<int, 1>{}, ug);
            else if (NUG >= 4 && live == 2) tail_group(std::integral_constant<int, (NUG >= 2 ? 2 : 1)>{}, ug);
            else tail_group(std::integral_constant<int, NUG>{}, ug);
        }
    } else {
    f32x16(&dsr)[NTG] = spr;  // ds_pre overwrites s_pre in place, tile by tile
    for (int ug = 0; ug < S.NGK; ++ug) {
        f32x16 acc2[NUG];
        // ResGCP: d(x) = d(out) + GCP^T d(out); the pass-through term initialises the accumulator (columns < si)
        if (single && ug == 0 && NUG == NTG) {
#pragma unroll
            for (int uu = 0; uu < NUG; ++uu)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[uu][r] = p.o.fused_residual ? dyr[uu < NTG ? uu : 0][r] : 0.f;
        } else {
            if (p.o.fused_residual) {  // (si == so there: columns past si read as zero)
                gcp_load_acc_layout<NUG, false>(dso_ptr, row, so, 32 * ug * NUG, hi, row_ok, vec_so, acc2);
            } else {
#pragma unroll
                for (int uu = 0; uu < NUG; ++uu)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[uu][r] = 0.f;
            }
        }
        for (int g = 0; g < S.NG; ++g) {
            if (ug == 0) {  // first pass over this output group: build ds_pre and keep a copy in HBM
                f32x16 gacc[NTG];
#pragma unroll
                for (int t = 0; t < NTG; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) gacc[t][r] = 0.f;
                if (scalar_gate) {
                    const float* wg = p.w.pack + S.offD + ((int64_t)g * 64 + lane) * NTG;
                    for (int oo0 = 0; oo0 < S.NOO; oo0 += 8) {  // fragments of 8 k-pair steps requested together
                        WFragB<NTG> a[8];
                        float b[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int oo = min(oo0 + u, S.NOO - 1);
                            a[u].load(wg + (int64_t)oo * S.NG * 64 * NTG);
                            b[u] = dgt[e * L.GS2 + gcp_crow(oo, hi)];
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (oo0 + u < S.NOO)
#pragma unroll
                                for (int t = 0; t < NTG; ++t)
                                    gacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].v[t], b[u], gacc[t], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int t = 0; t < NTG; ++t) {
                    float4 sp[4], dy[4];  // one tile's loads in flight together
                    if (single) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            sp[q] = make_float4(spr[t][4 * q], spr[t][4 * q + 1], spr[t][4 * q + 2], spr[t][4 * q + 3]);
                            dy[q] = make_float4(dyr[t][4 * q], dyr[t][4 * q + 1], dyr[t][4 * q + 2], dyr[t][4 * q + 3]);
                        }
                    } else {
                        gcp_load_tile4(sp_ptr, row, so, 32 * (g * NTG + t), hi, row_ok, vec_so, sp);
                        gcp_load_tile4(dso_ptr, row, so, 32 * (g * NTG + t), hi, row_ok, vec_so, dy);
                    }
                    float4 d[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        d[q].x = dy[q].x * gcp_dactf<PWL>(p.o.act_s, ns_s, slope, sp[q].x);
                        d[q].y = dy[q].y * gcp_dactf<PWL>(p.o.act_s, ns_s, slope, sp[q].y);
                        d[q].z = dy[q].z * gcp_dactf<PWL>(p.o.act_s, ns_s, slope, sp[q].z);
                        d[q].w = dy[q].w * gcp_dactf<PWL>(p.o.act_s, ns_s, slope, sp[q].w);
                        if (scalar_gate) {
                            d[q].x += gcp_dactf<PWL>(p.o.act_v, ns_v, slope, sp[q].x) * gacc[t][4 * q + 0];
                            d[q].y += gcp_dactf<PWL>(p.o.act_v, ns_v, slope, sp[q].y) * gacc[t][4 * q + 1];
                            d[q].z += gcp_dactf<PWL>(p.o.act_v, ns_v, slope, sp[q].z) * gacc[t][4 * q + 2];
                            d[q].w += gcp_dactf<PWL>(p.o.act_v, ns_v, slope, sp[q].w) * gacc[t][4 * q + 3];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int j0 = 32 * (g * NTG + t) + 8 * q + 4 * hi;
                        if (!row_ok) d[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                        dsr[t][4 * q + 0] = d[q].x; dsr[t][4 * q + 1] = d[q].y;
                        dsr[t][4 * q + 2] = d[q].z; dsr[t][4 * q + 3] = d[q].w;
                        store4(dsp_ptr, row, so, j0, d[q], row_ok, vec_so);
                    }
                }
            } else if (!single) {  // later merged-axis groups: re-read this lane's own ds_pre stores
                gcp_load_acc_layout<NTG, false>(dsp_ptr, row, so, 32 * g * NTG, hi, row_ok, vec_so, dsr);
            }
            // scalar_out adjoint for this (merged group, output group): 16 * NTG k-pair steps
            const float* wq = p.w.pack + S.offB + (((int64_t)ug * S.NS + (int64_t)g * NTG * 16) * 64 + lane) * NUG;
            {
                // 16 * NTG steps, fully unrolled (the B operands are registers); weight fragments rotate through three
                // batches of 4 steps, requested two batches ahead and pinned there (hipcc would sink them to their use)
                WFragB<NUG> A0[4], A1[4], A2[4];
                auto ld = [&](WFragB<NUG>(&a)[4], int st0) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) a[u].load(wq + (int64_t)min(st0 + u, NTG * 16 - 1) * 64 * NUG);
                };
                ld(A0, 0);
                ld(A1, 4);
                ld(A2, 8);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < NTG * 4; ++b) {
                    WFragB<NUG>(&a)[4] = (b % 3 == 0) ? A0 : ((b % 3 == 1) ? A1 : A2);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int st = b * 4 + u;
#pragma unroll
                        for (int uu = 0; uu < NUG; ++uu)
                            acc2[uu] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].v[uu], dsr[st / 16][st % 16], acc2[uu], 0, 0, 0);
                    }
                    if ((b + 3) * 4 < NTG * 16) ld(a, (b + 3) * 4);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // epilogue of this merged-axis group: d_s_in columns go to HBM, the vector extras to LDS
#pragma unroll
        for (int uu = 0; uu < NUG; ++uu)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k0 = 32 * (ug * NUG + uu) + 8 * q + 4 * hi;
                if (k0 + 3 < si) {
                    float4 v = make_float4(acc2[uu][4 * q], acc2[uu][4 * q + 1], acc2[uu][4 * q + 2], acc2[uu][4 * q + 3]);
                    store4(p.d_s_in, row, si, k0, v, row_ok, vec_si);
                } else {
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        const int k = k0 + x;
                        const float val = acc2[uu][4 * q + x];
                        if (k < si) {
                            if (row_ok) p.d_s_in[(int64_t)row * si + k] = val;
                        } else if (k < S.K) {
                            dext[e * L.DS + (k - si)] = val;
                        }
                    }
                }
            }
    }
    }
    STAMP(3);
    if (!has_vec) return;
    gcp_wave_lds_sync();
    asm volatile("" : "+v"(lane), "+v"(e), "+v"(hi));
    row = r0 + e;
    row_ok = row < rows;

    // ---- 5. adjoint of the vector prologue: d vh, d vf, then d v_in --------------------------------------------
    for (int h = hi; h < H; h += 2) {
        const float dn = dext[e * L.DS + h] * rn[e * L.NS_ + h];
        const float* wu = sw.wu;  // [vo, H]
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        if (has_vout)
            for (int oc = 0; oc < vo; ++oc) {
                const float w = wu[oc * H + h];
                a0 = fmaf(w, dvut[e * L.US + 3 * oc + 0], a0);
                a1 = fmaf(w, dvut[e * L.US + 3 * oc + 1], a1);
                a2 = fmaf(w, dvut[e * L.US + 3 * oc + 2], a2);
            }
        dvhf[e * L.FS + 0 * HF + h] = a0 + dn * vht[e * L.HS + 3 * h + 0];
        dvhf[e * L.FS + 1 * HF + h] = a1 + dn * vht[e * L.HS + 3 * h + 1];
        dvhf[e * L.FS + 2 * HF + h] = a2 + dn * vht[e * L.HS + 3 * h + 2];
    }
    for (int k = hi; k < 3; k += 2) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        if (S.nf) {
            const float* f = fr + e * 9;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float ds = dext[e * L.DS + H + 3 * k + a];
                if (p.o.e3 && a == 1) ds *= dvhf[e * L.FS + 0 * HF + H + k];
                a0 = fmaf(f[3 * a + 0], ds, a0);
                a1 = fmaf(f[3 * a + 1], ds, a1);
                a2 = fmaf(f[3 * a + 2], ds, a2);
            }
        }
        dvhf[e * L.FS + 0 * HF + H + k] = a0;
        dvhf[e * L.FS + 1 * HF + H + k] = a1;
        dvhf[e * L.FS + 2 * HF + H + k] = a2;
    }
    gcp_wave_lds_sync();
    if (p.sc.dvhf && row_ok)  // d[vh | vf] per row, [3, HF'] xyz-major: the gradient of the pre-projected vector tables
        for (int i = hi; i < 3 * HFPQ; i += 2) {
            const int d = i / HFPQ, x = i - d * HFPQ;
            p.sc.dvhf[(int64_t)row * 3 * HFPQ + i] = x < HF ? dvhf[e * L.FS + d * HF + x] : 0.f;
        }
    if (row_ok) {
        for (int c = hi; c < vi; c += 2) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            for (int h = 0; h < H; ++h) {
                const float w = sw.wd[h * vi + c];
                a0 = fmaf(w, dvhf[e * L.FS + 0 * HF + h], a0);
                a1 = fmaf(w, dvhf[e * L.FS + 1 * HF + h], a1);
                a2 = fmaf(w, dvhf[e * L.FS + 2 * HF + h], a2);
            }
            if (S.nf)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float w = sw.wf[k * vi + c];
                    a0 = fmaf(w, dvhf[e * L.FS + 0 * HF + H + k], a0);
                    a1 = fmaf(w, dvhf[e * L.FS + 1 * HF + H + k], a1);
                    a2 = fmaf(w, dvhf[e * L.FS + 2 * HF + H + k], a2);
                }
            if (p.o.vector_residual && has_vout) {
                a0 += dvut[e * L.US + 3 * c + 0];
                a1 += dvut[e * L.US + 3 * c + 1];
                a2 += dvut[e * L.US + 3 * c + 2];
            }
            if (p.o.fused_residual) {
                a0 += dvot[e * L.US + 3 * c + 0]; a1 += dvot[e * L.US + 3 * c + 1]; a2 += dvot[e * L.US + 3 * c + 2];
            }
            float* dp = p.d_v_in + ((int64_t)row * vi + c) * 3;
            dp[0] = a0; dp[1] = a1; dp[2] = a2;
        }
    }
    // ---- 6. this tile's share of the small vector weight gradients, on v_mfma_f32_16x16x4_f32 with the reduction
    //         running over the tile's 96 (row, xyz) pairs; all four operand tiles are still in LDS:
    //           d vector_up[o, h]                    = sum dvu[row, o, d] * vh[row, h, d]
    //           d [vector_down ; vector_down_frames][x, c] = sum v[row, c, d] * [dvh | dvf][row, d, x]
    //         The per-tile sums go to sc.w_part[tile, :] and are reduced over tiles by gcpnet_reduce_partials.
    if (p.sc.w_part) {
        asm volatile("" : "+v"(lane));
        const int l16 = lane & 15, kq = lane >> 4;
        float* part = p.sc.w_part + (int64_t)blockIdx.x * (vo * H + vi * HF);
        auto small_tn = [&](const float* A, int ars, int ams, int ads, int M, const float* B, int brs, int bms, int bds, int N,
                            float* out, bool transposed) {
            for (int mt = 0; mt < M; mt += 16)
                for (int nt = 0; nt < N; nt += 16) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    const int m = mt + l16, n = nt + l16;
                    const bool mok = m < M, nok = n < N;
                    const float* ap = A + (mok ? m : 0) * ams;
                    const float* bp = B + (nok ? n : 0) * bms;
#pragma unroll 8
                    for (int st = 0; st < 24; ++st) {
                        const int kidx = 4 * st + kq, rr = kidx / 3, d = kidx - 3 * rr;
                        float a = ap[rr * ars + d * ads], b = bp[rr * brs + d * bds];
                        a = (mok && r0 + rr < rows) ? a : 0.f;
                        b = nok ? b : 0.f;
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = mt + 4 * kq + r;
                        if (i < M && nok) out[transposed ? n * M + i : i * N + n] = acc[r];
                    }
                }
        };
        if (has_vout) small_tn(dvut, L.US, 3, 1, vo, vht, L.HS, 3, 1, H, part, false);
        small_tn(vt, L.VS, 3, 1, vi, dvhf, L.FS, 1, HF, HF, part + vo * H, true);  // stored as [H + 3, vi]
    }
    STAMP(4);
}

template <int NTG, int NUG, bool PWL, bool SINGLE>
int launch4(const BwdParams& p, size_t lds_bytes, hipStream_t st) {
    static size_t cur_max = 64 * 1024;
    if (lds_bytes > cur_max) {
        hipError_t err = hipFuncSetAttribute((const void*)gcp2_bwd_kernel<NTG, NUG, PWL, SINGLE>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (err != hipSuccess) return (int)err;
        cur_max = lds_bytes;
    }
    hipLaunchKernelGGL((gcp2_bwd_kernel<NTG, NUG, PWL, SINGLE>), dim3((unsigned)gcp_cdiv(p.rows, GCP_TILE_ROWS)), dim3(GCP_WAVE),
                       lds_bytes, st, p);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

template <int NTG, int NUG, bool PWL>
int launch3(const BwdParams& p, size_t lds_bytes, hipStream_t st) {
    if (p.sh.NG == 1) return launch4<NTG, NUG, PWL, true>(p, lds_bytes, st);
    return launch4<NTG, NUG, PWL, false>(p, lds_bytes, st);
}

template <int NTG, int NUG>
int launch(const BwdParams& p, size_t lds_bytes, hipStream_t st) {
    if (gcp_is_pwl(p.o.act_s) && gcp_is_pwl(p.o.act_v)) return launch3<NTG, NUG, true>(p, lds_bytes, st);
    return launch3<NTG, NUG, false>(p, lds_bytes, st);
}

template <int NTG>
int launch_ntg(const BwdParams& p, size_t lds_bytes, hipStream_t st) {
    switch (p.sh.NUG) {
        case 1: return launch<NTG, 1>(p, lds_bytes, st);
        case 2: return launch<NTG, 2>(p, lds_bytes, st);
        default: return launch<NTG, 4>(p, lds_bytes, st);
    }
}

}  // namespace

extern "C" int gcpnet_gcp2_bwd_tiles(int rows) { return rows <= 0 ? 0 : gcp_cdiv(rows, GCP_TILE_ROWS); }

extern "C" int gcpnet_gcp2_backward(int rows, const gcp_concat_t* s_in, const gcp_concat_t* v_in, const float* frames,
                                    const gcp2_weights_t* w, const gcp2_opts_t* opts, const gcp_concat_t* v_add, const float* s_pre,
                                    const float* gate, const float* d_s_out, const float* d_v_out, float* d_s_in,
                                    float* d_v_in, const gcp2_bwd_scratch_t* sc, void* stream) {
    (void)s_in;
    if (rows < 0 || !w || !opts || !sc || !s_pre || !d_s_out || !d_s_in || !w->pack || !sc->ds_pre) return GCPNET_E_BADARG;
    if (rows == 0) return 0;
    const bool has_vec = w->vi > 0;
    if (has_vec) {
        if (!v_in || v_in->n < 1 || !w->w_down || !d_v_in || !sc->ext) return GCPNET_E_BADARG;
        if (w->use_frames && (!frames || !w->w_frames)) return GCPNET_E_BADARG;
        if (w->vo > 0 && (!d_v_out || !w->w_up)) return GCPNET_E_BADARG;
    }
    if (w->vo > 64) return GCPNET_E_UNSUPPORTED;
    if (opts->fused_residual && (w->si != w->so || w->vi != w->vo)) return GCPNET_E_BADARG;
    BwdParams p;
    p.rows = rows;
    if (has_vec) p.v_in = *v_in; else p.v_in.n = 0;
    p.v_add.n = 0;
    if (v_add && v_add->n > 0) {
        if (!has_vec || v_add->n > GCP_MAX_SEG || !sc->dvhf) return GCPNET_E_BADARG;
        for (int k = 0; k < v_add->n; ++k)
            if (!v_add->ptr[k] || v_add->dim[k] != gcp_round_up(w->hidden + (w->use_frames ? 3 : 0), 4)) return GCPNET_E_BADARG;
        p.v_add = *v_add;
    }
    p.frames = frames;
    p.w = *w;
    p.o = *opts;
    if (!has_vec) p.o.vmode = GCP_VMODE_NONE;
    if (p.o.vmode == GCP_VMODE_SCALAR_GATE && w->vo > 0 && has_vec && (!gate || !sc->dgate || !w->w_gate)) return GCPNET_E_BADARG;
    p.s_pre = s_pre; p.gate = gate; p.d_s_out = d_s_out; p.d_v_out = d_v_out;
    p.d_s_in = d_s_in; p.d_v_in = d_v_in;
    p.sc = *sc;
    p.stamps = g_gcp_phase_buf; p.stamp_cap = g_gcp_phase_cap;
    p.sh = gcp_shape(w->si, w->vi, w->so, w->vo, w->hidden, w->use_frames);
    const BwdLds L = bwd_lds(p.sh);
    const size_t lds_bytes = (size_t)L.total * sizeof(float);
    if (lds_bytes > 160 * 1024) return GCPNET_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    switch (p.sh.NTG) {
        case 1: return launch_ntg<1>(p, lds_bytes, st);
        case 2: return launch_ntg<2>(p, lds_bytes, st);
        default: return launch_ntg<4>(p, lds_bytes, st);
    }
}
