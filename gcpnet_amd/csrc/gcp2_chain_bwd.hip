// Backward (data path) of a chain of residual GCP2 blocks, x_k = x_{k-1} + GCP_k(x_{k-1}) (ResGCP, reference
// src/models/components/gcpnet.py:921-924; the reference gets this from autograd), one launch for the whole chain.
//
// The gradient of the chain state never leaves the chip between blocks:
//   * d(s) of a 32-row tile lives in the 32x32 MFMA accumulator layout.  The scalar_out adjoint W^T ds_pre is
//     accumulated ON TOP of it (the ResGCP pass-through is the accumulator's initial value), so the result is directly
//     the next block's d(s_out);
//   * d(V) lives in a wave-private LDS tile that the vector-prologue adjoint updates in place.
// Per block the wave only reads what the forward saved (s_pre, the gates, the block's input vectors) and writes what the
// weight-gradient GEMM needs (ds_pre, dgate, norms / frame scalars, the per-tile small-weight partial sums).  Those reads
// are requested one block ahead: as soon as the data GEMM of block k has consumed its registers, the s_pre tile of block
// k-1 is requested into them, and the vectors / gates / small weights of block k-1 are requested into registers and
// committed to LDS when block k is finished -- a block's HBM latency is spent under the previous block's arithmetic.
// The arithmetic of one block is the one of gcp2_bwd.hip (single-output-group path).
#include "common.h"
#include "tile_io.h"

#include <type_traits>

int gcp2_chain_bwd_registers(int rows, const float* frames, int n, const gcp2_chain_bwd_item_t* items, const float* d_s_out,
                             const float* d_v_out, float* d_s_in, float* d_v_in, hipStream_t st);

namespace {

struct ChainItemB {
    const float* pack;
    const float* w_down;
    const float* w_frames;
    const float* w_up;
    const float* v_in;
    const float* s_pre;
    const float* gate;
    float* ds_pre;
    float* dgate;
    float* ext;
    float* w_part;
    int act_s, act_v;
};

struct ChainBwdParams {
    int rows;
    const float* frames;
    gcp2_opts_t o;  // shared: slope, vmode, vector_residual, e3
    int n;
    ChainItemB it[GCP_MAX_CHAIN];
    const float* d_s_out;
    const float* d_v_out;
    float* d_s_in;
    float* d_v_in;
    unsigned long long* stamps;
    long long stamp_cap;
    GcpShape sh;
};

struct CbLds {
    int VS, HS, NS_, GS2, DS, FS;
    int o_vt, o_vht, o_rn, o_dvut, o_dvt, o_dgt, o_dext, o_dvhf, o_fr, o_sw, total;
};

__host__ __device__ inline CbLds cb_lds(const GcpShape& s) {
    CbLds l;
    l.VS = gcp_odd(3 * s.vi);
    l.HS = gcp_odd(3 * s.H);
    l.NS_ = gcp_odd(s.H);
    l.GS2 = gcp_odd(2 * s.NOO);
    l.DS = gcp_odd(s.H + 9);
    l.FS = gcp_odd(3 * (s.H + 3));
    l.o_vt = 0;
    l.o_vht = l.o_vt + 32 * l.VS;
    l.o_rn = l.o_vht + 32 * l.HS;
    l.o_dvut = l.o_rn + 32 * l.NS_;
    l.o_dvt = l.o_dvut + 32 * l.VS;
    l.o_dgt = l.o_dvt + 32 * l.VS;
    l.o_dext = l.o_dgt + 32 * l.GS2;
    l.o_dvhf = l.o_dext + 32 * l.DS;
    l.o_fr = l.o_dvhf + 32 * l.FS;
    l.o_sw = l.o_fr + 32 * 9;
    l.total = l.o_sw + gcp_small_w_floats(s.vi, s.H, s.vo, s.nf);
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

// The three small vector weights of a block, requested into registers (4 clamped loads per array and lane: <= 256 floats
// each) and written to the LDS area later.
struct SmallWRegs {
    float d[4], f[4], u[4];
};
__device__ __forceinline__ void small_w_issue(SmallWRegs& r, const ChainItemB& it, int nd, int nf3, int nu, int lane) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        r.d[k] = it.w_down[min(lane + 64 * k, nd - 1)];
        r.f[k] = nf3 ? it.w_frames[min(lane + 64 * k, nf3 - 1)] : 0.f;
        r.u[k] = it.w_up[min(lane + 64 * k, nu - 1)];
    }
}
__device__ __forceinline__ void small_w_commit(const SmallWRegs& r, float* area, int nd, int nf3, int nu, int lane) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane + 64 * k;
        if (i < nd) area[i] = r.d[k];
        if (i < nf3) area[nd + i] = r.f[k];
        if (i < nu) area[nd + nf3 + i] = r.u[k];
    }
}

// NTG: 32-wide tiles of the scalar state (si == so == 32 * NTG); PWL: all activations are identity / relu / leakyrelu.
template <int NTG, bool PWL>
__global__ __launch_bounds__(GCP_WAVE, 1) void gcp2_chain_bwd_kernel(ChainBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const GcpShape& S = p.sh;
    const CbLds L = cb_lds(S);
    int lane = threadIdx.x;
    int e = lane & 31, hi = lane >> 5;
    const int r0 = blockIdx.x * GCP_TILE_ROWS;
    const int rows = p.rows;
    int row = r0 + e;
    bool row_ok = row < rows;
    float* vt = lds + L.o_vt;      // the block's input vectors
    float* vht = lds + L.o_vht;    // vector_down(v)
    float* rn = lds + L.o_rn;      // 1 / |vh|
    float* dvut = lds + L.o_dvut;  // d(vector_up output)
    float* dvt = lds + L.o_dvt;    // d(V) chain state
    float* dgt = lds + L.o_dgt;    // sigmoid(gate) on entry of a block, overwritten in place by d(gate)
    float* dext = lds + L.o_dext;  // d(norms | frame scalars)
    float* dvhf = lds + L.o_dvhf;  // [d vh | d vf]
    float* fr = lds + L.o_fr;
    float* swa = lds + L.o_sw;
    const int so = S.so, vi = S.vi, H = S.H, HF = S.H + 3;  // si == so, vo == vi
    const int EP = gcp_round_up(S.H + S.nf, 4), VOP = gcp_round_up(vi, 4);
    const int nd = H * vi, nf3 = S.nf ? 3 * vi : 0, nu = vi * H;
    const float* swd = swa;
    const float* swf = swa + nd;
    const float* swu = swa + nd + nf3;
    const float slope = p.o.slope;
    const bool scalar_gate = p.o.vmode == GCP_VMODE_SCALAR_GATE;
    const int NUG = S.NUG;
    const int xg = NTG / NUG, xs = NTG - xg * NUG;  // group / slot of the 32-wide tile holding the norms and frame scalars

    gcp_stamp(p.stamps, p.stamp_cap, 0, lane);
    f32x16 spr[NTG], dyr[NTG];
    // ---- prologue: everything the LAST block needs, plus the incoming gradients, in one memory round trip ----------
    {
        const ChainItemB& it = p.it[p.n - 1];
        GcpSegBuf<8> vb, gb;
        GcpSegBuf<4> tb;
        SmallWRegs swr;
        gcp_seg_issue(vb, it.v_in, nullptr, 3 * vi, r0, rows, vt, L.VS, 0, lane);
        gcp_seg_issue(gb, p.d_v_out, nullptr, 3 * vi, r0, rows, dvt, L.VS, 0, lane);
        if (scalar_gate) gcp_seg_issue(tb, it.gate, nullptr, vi, r0, rows, dgt, L.GS2, 0, lane);
        if (S.nf) gcp_load_frames(p.frames, r0, rows, fr, lane);
        small_w_issue(swr, it, nd, nf3, nu, lane);
        gcp_seg_commit(vb, vt, L.VS, 0);
        gcp_seg_commit(gb, dvt, L.VS, 0);
        if (scalar_gate) gcp_seg_commit(tb, dgt, L.GS2, 0);
        small_w_commit(swr, swa, nd, nf3, nu, lane);
        for (int i = vi + hi; i < 2 * S.NOO; i += 2) dgt[e * L.GS2 + i] = 0.f;  // zero the gate-adjoint k padding (stays zero)
#pragma unroll
        for (int t = 0; t < NTG; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j0 = 32 * t + 8 * q + 4 * hi;
                const float4 a = gcp_load4(it.s_pre, row, so, j0, row_ok, true);
                const float4 b = gcp_load4(p.d_s_out, row, so, j0, row_ok, true);
                spr[t][4 * q] = a.x; spr[t][4 * q + 1] = a.y; spr[t][4 * q + 2] = a.z; spr[t][4 * q + 3] = a.w;
                dyr[t][4 * q] = b.x; dyr[t][4 * q + 1] = b.y; dyr[t][4 * q + 2] = b.z; dyr[t][4 * q + 3] = b.w;
            }
    }
    gcp_stamp(p.stamps, p.stamp_cap, 1, lane);

    for (int k = p.n - 1; k >= 0; --k) {
        // Opaque to the optimiser: otherwise the per-lane addresses of the loop body are hoisted out of the loop and spilled.
        asm volatile("" : "+v"(lane), "+v"(e), "+v"(hi));
        row = r0 + e;
        row_ok = row < rows;
        const ChainItemB& it = p.it[k];
        const float ns_s = gcp_neg_slope(it.act_s, slope), ns_v = gcp_neg_slope(it.act_v, slope);
        const bool stamp_here = k == 0;
        gcp_wave_lds_sync();
        if (stamp_here) gcp_stamp(p.stamps, p.stamp_cap, 2, lane);
        // ---- 1. recompute vh, its norms and the frame scalars (the latter two also go to the weight-gradient GEMM) ---
        {
            const float* vrow = vt + e * L.VS;
            for (int h = hi; h < H; h += 2) {
                const float* wd = swd + h * vi;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f;
                for (int c = 0; c < vi; ++c) {
                    const float w = wd[c];
                    a0 = fmaf(w, vrow[3 * c + 0], a0);
                    a1 = fmaf(w, vrow[3 * c + 1], a1);
                    a2 = fmaf(w, vrow[3 * c + 2], a2);
                }
                vht[e * L.HS + 3 * h + 0] = a0;
                vht[e * L.HS + 3 * h + 1] = a1;
                vht[e * L.HS + 3 * h + 2] = a2;
                const float nr = sqrtf(a0 * a0 + a1 * a1 + a2 * a2 + 1e-8f);
                rn[e * L.NS_ + h] = 1.0f / nr;
                if (row_ok) it.ext[(int64_t)row * EP + h] = nr + 1e-8f;
            }
            if (row_ok && hi == 0) {  // zero the stride padding
                for (int c = H + S.nf; c < EP; ++c) it.ext[(int64_t)row * EP + c] = 0.f;
                if (scalar_gate)
                    for (int c = vi; c < VOP; ++c) it.dgate[(int64_t)row * VOP + c] = 0.f;
            }
            if (S.nf) {
                const float* f = fr + e * 9;
                for (int kk = hi; kk < 3; kk += 2) {
                    const float* wf = swf + kk * vi;
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
                    for (int c = 0; c < vi; ++c) {
                        const float w = wf[c];
                        a0 = fmaf(w, vrow[3 * c + 0], a0);
                        a1 = fmaf(w, vrow[3 * c + 1], a1);
                        a2 = fmaf(w, vrow[3 * c + 2], a2);
                    }
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        float pr = f[3 * a + 0] * a0 + f[3 * a + 1] * a1 + f[3 * a + 2] * a2;
                        if (p.o.e3 && a == 1) {
                            dvhf[e * L.FS + 0 * HF + H + kk] = pr < 0.f ? -1.f : 1.f;  // sign for the adjoint of |.|
                            pr = fabsf(pr);
                        }
                        if (row_ok) it.ext[(int64_t)row * EP + H + 3 * kk + a] = pr;
                    }
                }
            }
        }
        gcp_wave_lds_sync();
        if (stamp_here) gcp_stamp(p.stamps, p.stamp_cap, 3, lane);
        asm volatile("" : "+v"(lane), "+v"(e), "+v"(hi));
        row = r0 + e;
        row_ok = row < rows;
        // ---- 2. adjoint of the vector epilogue (gcpnet.py:364-391): d(vector_up output), d(gate) ---------------------
        for (int oc0 = hi; oc0 < vi; oc0 += 16) {  // 8 channels per lane per pass: all LDS reads, then the writes
            float du[8][3], dgv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int oc = oc0 + 2 * i;
                du[i][0] = du[i][1] = du[i][2] = 0.f; dgv[i] = 0.f;
                if (oc < vi) {
                    const float* wu = swu + oc * H;
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
                    const float g0 = dvt[e * L.VS + 3 * oc + 0], g1 = dvt[e * L.VS + 3 * oc + 1], g2 = dvt[e * L.VS + 3 * oc + 2];
                    float du0 = g0, du1 = g1, du2 = g2;
                    const float dot = g0 * u0 + g1 * u1 + g2 * u2;
                    if (scalar_gate) {
                        const float sg = dgt[e * L.GS2 + oc];
                        du0 = g0 * sg; du1 = g1 * sg; du2 = g2 * sg;
                        dgv[i] = dot * sg * (1.f - sg);
                    } else if (p.o.vmode == GCP_VMODE_SELF_GATE) {
                        const float rs = sqrtf(u0 * u0 + u1 * u1 + u2 * u2 + 1e-8f);
                        const float nn = rs + 1e-8f;
                        const float a = gcp_actf<PWL>(it.act_v, ns_v, slope, nn), da = gcp_dactf<PWL>(it.act_v, ns_v, slope, nn);
                        const float coef = dot * da / rs;
                        du0 = g0 * a + coef * u0; du1 = g1 * a + coef * u1; du2 = g2 * a + coef * u2;
                    }
                    du[i][0] = du0; du[i][1] = du1; du[i][2] = du2;
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int oc = oc0 + 2 * i;
                if (oc < vi) {
                    dvut[e * L.VS + 3 * oc + 0] = du[i][0];
                    dvut[e * L.VS + 3 * oc + 1] = du[i][1];
                    dvut[e * L.VS + 3 * oc + 2] = du[i][2];
                    if (scalar_gate) {
                        dgt[e * L.GS2 + oc] = dgv[i];
                        if (row_ok) it.dgate[(int64_t)row * VOP + oc] = dgv[i];
                    }
                }
            }
        }
        gcp_wave_lds_sync();
        if (stamp_here) gcp_stamp(p.stamps, p.stamp_cap, 4, lane);
        asm volatile("" : "+v"(lane), "+v"(e), "+v"(hi));
        row = r0 + e;
        row_ok = row < rows;

        // ---- 3. ds_pre = d(s_out) * act_s'(s_pre) + act_v'(s_pre) * (Wg^T dgate), in the s_pre registers -------------
        {
            f32x16 gacc[NTG];
#pragma unroll
            for (int t = 0; t < NTG; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) gacc[t][r] = 0.f;
            if (scalar_gate) {
                const float* wg = it.pack + S.offD + (int64_t)lane * NTG;
                for (int oo0 = 0; oo0 < S.NOO; oo0 += 8) {
                    WFragC<NTG> a[8];
                    float b[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int oo = min(oo0 + u, S.NOO - 1);
                        a[u].load(wg + (int64_t)oo * 64 * NTG);
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
                    float d = dyr[t][r] * gcp_dactf<PWL>(it.act_s, ns_s, slope, sp);
                    if (scalar_gate) d += gcp_dactf<PWL>(it.act_v, ns_v, slope, sp) * gacc[t][r];
                    spr[t][r] = row_ok ? d : 0.f;
                }
        }
#pragma unroll
        for (int t = 0; t < NTG; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                gcp_store4(it.ds_pre, row, so, 32 * t + 8 * q + 4 * hi,
                           make_float4(spr[t][4 * q], spr[t][4 * q + 1], spr[t][4 * q + 2], spr[t][4 * q + 3]), row_ok, true);

        // ---- 4. d(s) += W^T ds_pre: 16 * NTG k-pair steps whose B operands are the ds_pre registers; the weight
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
        data_gemm(std::integral_constant<int, NTG>{}, it.pack + S.offB + (int64_t)lane * NUG, dyr);
        {  // the tile of the merged axis that holds the norms and frame scalars: their adjoints go to LDS
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
        if (stamp_here) gcp_stamp(p.stamps, p.stamp_cap, 5, lane);

        // ---- requests for the next block (k-1): its s_pre tile into the registers the data GEMM has just released, its
        //      vectors / gates / small weights into registers that are committed to LDS at the end of this block ---------
        GcpSegBuf<8> vb;
        GcpSegBuf<4> tb;
        SmallWRegs swr;
        if (k > 0) {
            const ChainItemB& nx = p.it[k - 1];
#pragma unroll
            for (int t = 0; t < NTG; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 a = gcp_load4(nx.s_pre, row, so, 32 * t + 8 * q + 4 * hi, row_ok, true);
                    spr[t][4 * q] = a.x; spr[t][4 * q + 1] = a.y; spr[t][4 * q + 2] = a.z; spr[t][4 * q + 3] = a.w;
                }
            gcp_seg_issue(vb, nx.v_in, nullptr, 3 * vi, r0, rows, vt, L.VS, 0, lane);
            if (scalar_gate) gcp_seg_issue(tb, nx.gate, nullptr, vi, r0, rows, dgt, L.GS2, 0, lane);
            small_w_issue(swr, nx, nd, nf3, nu, lane);
            __builtin_amdgcn_sched_barrier(0);
        } else {  // first block of the chain: d(s) leaves the chip
#pragma unroll
            for (int t = 0; t < NTG; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    gcp_store4(p.d_s_in, row, so, 32 * t + 8 * q + 4 * hi,
                               make_float4(dyr[t][4 * q], dyr[t][4 * q + 1], dyr[t][4 * q + 2], dyr[t][4 * q + 3]), row_ok, true);
        }
        gcp_wave_lds_sync();

        // ---- 5. adjoint of the vector prologue: d vh, d vf, then d(V) updated in place ----------------------------------
        for (int h = hi; h < H; h += 2) {
            const float dn = dext[e * L.DS + h] * rn[e * L.NS_ + h];
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            for (int oc = 0; oc < vi; ++oc) {
                const float w = swu[oc * H + h];
                a0 = fmaf(w, dvut[e * L.VS + 3 * oc + 0], a0);
                a1 = fmaf(w, dvut[e * L.VS + 3 * oc + 1], a1);
                a2 = fmaf(w, dvut[e * L.VS + 3 * oc + 2], a2);
            }
            dvhf[e * L.FS + 0 * HF + h] = a0 + dn * vht[e * L.HS + 3 * h + 0];
            dvhf[e * L.FS + 1 * HF + h] = a1 + dn * vht[e * L.HS + 3 * h + 1];
            dvhf[e * L.FS + 2 * HF + h] = a2 + dn * vht[e * L.HS + 3 * h + 2];
        }
        for (int kk = hi; kk < 3; kk += 2) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            if (S.nf) {
                const float* f = fr + e * 9;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    float ds = dext[e * L.DS + H + 3 * kk + a];
                    if (p.o.e3 && a == 1) ds *= dvhf[e * L.FS + 0 * HF + H + kk];
                    a0 = fmaf(f[3 * a + 0], ds, a0);
                    a1 = fmaf(f[3 * a + 1], ds, a1);
                    a2 = fmaf(f[3 * a + 2], ds, a2);
                }
            }
            dvhf[e * L.FS + 0 * HF + H + kk] = a0;
            dvhf[e * L.FS + 1 * HF + H + kk] = a1;
            dvhf[e * L.FS + 2 * HF + H + kk] = a2;
        }
        gcp_wave_lds_sync();
        for (int c = hi; c < vi; c += 2) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            for (int h = 0; h < H; ++h) {
                const float w = swd[h * vi + c];
                a0 = fmaf(w, dvhf[e * L.FS + 0 * HF + h], a0);
                a1 = fmaf(w, dvhf[e * L.FS + 1 * HF + h], a1);
                a2 = fmaf(w, dvhf[e * L.FS + 2 * HF + h], a2);
            }
            if (S.nf)
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) {
                    const float w = swf[kk * vi + c];
                    a0 = fmaf(w, dvhf[e * L.FS + 0 * HF + H + kk], a0);
                    a1 = fmaf(w, dvhf[e * L.FS + 1 * HF + H + kk], a1);
                    a2 = fmaf(w, dvhf[e * L.FS + 2 * HF + H + kk], a2);
                }
            if (p.o.vector_residual) {
                a0 += dvut[e * L.VS + 3 * c + 0];
                a1 += dvut[e * L.VS + 3 * c + 1];
                a2 += dvut[e * L.VS + 3 * c + 2];
            }
            a0 += dvt[e * L.VS + 3 * c + 0]; a1 += dvt[e * L.VS + 3 * c + 1]; a2 += dvt[e * L.VS + 3 * c + 2];  // ResGCP pass-through
            dvt[e * L.VS + 3 * c + 0] = a0; dvt[e * L.VS + 3 * c + 1] = a1; dvt[e * L.VS + 3 * c + 2] = a2;
            if (k == 0 && row_ok) {
                float* dp = p.d_v_in + ((int64_t)row * vi + c) * 3;
                dp[0] = a0; dp[1] = a1; dp[2] = a2;
            }
        }
        if (stamp_here) gcp_stamp(p.stamps, p.stamp_cap, 6, lane);

        // ---- 6. this tile's share of the small vector weight gradients (see gcp2_bwd.hip, step 6) -------------------------
        if (it.w_part) {
            asm volatile("" : "+v"(lane));
            const int l16 = lane & 15, kq = lane >> 4;
            float* part = it.w_part + (int64_t)blockIdx.x * (vi * H + vi * HF);
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
            small_tn(dvut, L.VS, 3, 1, vi, vht, L.HS, 3, 1, H, part, false);
            small_tn(vt, L.VS, 3, 1, vi, dvhf, L.FS, 1, HF, HF, part + vi * H, true);  // stored as [H + 3, vi]
        }
        gcp_wave_lds_sync();
        if (k > 0) {  // the next block's tiles are (long) on chip: move them into place
            gcp_seg_commit(vb, vt, L.VS, 0);
            if (scalar_gate) gcp_seg_commit(tb, dgt, L.GS2, 0);
            small_w_commit(swr, swa, nd, nf3, nu, lane);
        }
    }
    gcp_stamp(p.stamps, p.stamp_cap, 7, lane);
}

template <int NTG, bool PWL>
int launch_cb(const ChainBwdParams& p, size_t lds_bytes, hipStream_t st) {
    hipLaunchKernelGGL((gcp2_chain_bwd_kernel<NTG, PWL>), dim3((unsigned)gcp_cdiv(p.rows, GCP_TILE_ROWS)), dim3(GCP_WAVE),
                       lds_bytes, st, p);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// Returns GCPNET_E_UNSUPPORTED when the chain does not fit this kernel; the caller then runs the blocks one by one
// (gcpnet_gcp2_backward), passing the state through HBM.
int gcp2_chain_bwd_registers(int rows, const float* frames, int n, const gcp2_chain_bwd_item_t* items, const float* d_s_out,
                             const float* d_v_out, float* d_s_in, float* d_v_in, hipStream_t st) {
    const gcp2_weights_t& w0 = items[0].w;
    const GcpShape S = gcp_shape(w0.si, w0.vi, w0.so, w0.vo, w0.hidden, w0.use_frames);
    if (S.NG != 1 || w0.si != w0.so || w0.vi != w0.vo || w0.vi <= 0 || w0.si != 32 * S.NTG) return GCPNET_E_UNSUPPORTED;
    if ((w0.vi & 3) || w0.vi > 20 || S.H + S.nf > 32 || S.H * w0.vi > 256) return GCPNET_E_UNSUPPORTED;
    if (S.NUG < S.NTG) return GCPNET_E_UNSUPPORTED;
    ChainBwdParams p;
    p.rows = rows; p.frames = frames; p.o = items[0].o; p.n = n;
    p.d_s_out = d_s_out; p.d_v_out = d_v_out; p.d_s_in = d_s_in; p.d_v_in = d_v_in;
    bool pwl = true;
    for (int k = 0; k < n; ++k) {
        const gcp2_chain_bwd_item_t& c = items[k];
        const gcp2_opts_t& o = c.o;
        if (c.w.si != w0.si || c.w.vi != w0.vi || c.w.so != w0.so || c.w.vo != w0.vo || c.w.hidden != w0.hidden ||
            c.w.use_frames != w0.use_frames || o.vmode != p.o.vmode || o.vector_residual != p.o.vector_residual ||
            o.e3 != p.o.e3 || o.slope != p.o.slope)
            return GCPNET_E_UNSUPPORTED;
        ChainItemB& it = p.it[k];
        it.pack = c.w.pack; it.w_down = c.w.w_down; it.w_frames = c.w.w_frames; it.w_up = c.w.w_up;
        it.v_in = c.v_in; it.s_pre = c.s_pre; it.gate = c.gate;
        it.ds_pre = c.sc.ds_pre; it.dgate = c.sc.dgate; it.ext = c.sc.ext; it.w_part = c.sc.w_part;
        it.act_s = o.act_s; it.act_v = o.act_v;
        pwl = pwl && gcp_is_pwl(o.act_s) && gcp_is_pwl(o.act_v);
    }
    p.stamps = g_gcp_phase_buf; p.stamp_cap = g_gcp_phase_cap;
    p.sh = S;
    const size_t lds_bytes = (size_t)cb_lds(S).total * sizeof(float);
    if (lds_bytes > 64 * 1024) return GCPNET_E_UNSUPPORTED;
    switch (S.NTG) {
        case 1: return pwl ? launch_cb<1, true>(p, lds_bytes, st) : launch_cb<1, false>(p, lds_bytes, st);
        case 2: return pwl ? launch_cb<2, true>(p, lds_bytes, st) : launch_cb<2, false>(p, lds_bytes, st);
        default: return pwl ? launch_cb<4, true>(p, lds_bytes, st) : launch_cb<4, false>(p, lds_bytes, st);
    }
}

extern "C" int gcpnet_gcp2_chain_backward(int rows, const float* frames, int n, const gcp2_chain_bwd_item_t* items,
                                          const float* d_s_out, const float* d_v_out, float* d_s_in, float* d_v_in,
                                          void* stream) {
    if (rows < 0 || n < 1 || n > GCP_MAX_CHAIN || !items || !d_s_out || !d_v_out || !d_s_in || !d_v_in) return GCPNET_E_BADARG;
    for (int k = 0; k < n; ++k) {
        const gcp2_chain_bwd_item_t& c = items[k];
        if (!c.w.pack || !c.w.w_down || !c.w.w_up || !c.v_in || !c.s_pre || !c.sc.ds_pre || !c.sc.ext) return GCPNET_E_BADARG;
        if (c.w.use_frames && (!frames || !c.w.w_frames)) return GCPNET_E_BADARG;
        if (c.o.vmode == GCP_VMODE_SCALAR_GATE && (!c.gate || !c.sc.dgate)) return GCPNET_E_BADARG;
    }
    if (rows == 0) return 0;
    auto misaligned = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) != 0; };
    for (int k = 0; k < n; ++k)  // the deferred tile loads and the accumulator-layout accesses are 16-byte accesses
        if (misaligned(items[k].v_in) || misaligned(items[k].s_pre) || misaligned(items[k].gate) || misaligned(items[k].sc.ds_pre))
            return GCPNET_E_UNSUPPORTED;
    if (misaligned(d_s_out) || misaligned(d_v_out) || misaligned(d_s_in)) return GCPNET_E_UNSUPPORTED;
    return gcp2_chain_bwd_registers(rows, frames, n, items, d_s_out, d_v_out, d_s_in, d_v_in, (hipStream_t)stream);
}
