// GCP2 forward on gfx950 -- replaces GCP2.forward (reference src/models/components/gcpnet.py:394-468).
//
// Design (MI355X-first):
//   * one 64-lane wavefront owns a tile of 32 rows (edges or nodes) and is fully autonomous: no inter-wave
//     barriers, workgroup = 1 wave, so the CU's 4 SIMDs interleave independent tiles and one wave's VALU/LDS
//     phases hide under another wave's MFMA phase;
//   * the scalar Linear (scalar_out) is computed TRANSPOSED, s_pre^T[so, 32 rows] = W[so, K] x merged^T[K, 32 rows],
//     with v_mfma_f32_32x32x2_f32 (exact fp32): A fragments are pre-packed weights streamed from L2 with one
//     16-byte load per k-pair, the B fragment is one conflict-free ds_read_b32 from the wave-private merged tile
//     (odd row stride), and every row's result lands in just two lanes, which is what the per-row epilogue wants;
//   * the tiny vector Linears (vector_down, vector_down_frames, vector_up: K = 4..36) run on the VALU, two lanes
//     per row; the vector gate Linear (vector_out_scale) runs on v_mfma_f32_16x16x4_f32 from the staged tile;
//   * the concatenation [h_row | e | h_col] is never materialised: the tile loader gathers each source.
#include "common.h"
#include "tile_io.h"

namespace {

struct FwdParams {
    int rows;
    gcp_concat_t s_in, v_in;
    const float* frames;
    gcp2_weights_t w;
    gcp2_opts_t o;
    const float* res_s;
    const float* res_v;
    float* s_out;
    float* v_out;
    float* s_pre;
    float* gate;
    int fused_res;  // res_s / res_v are the (single, ungathered) input itself
    unsigned long long* stamps;
    long long stamp_cap;
    GcpShape sh;
};

// Wave-private LDS layout (floats).  All row strides are odd so that 32 rows hit 32 distinct banks.
struct FwdLds {
    int KS, SS, VS, HS, GS;
    int o_mrg, o_stg, o_vt, o_vht, o_gt, o_fr, o_sw, total;
};

__host__ __device__ inline FwdLds fwd_lds(const GcpShape& s) {
    FwdLds l;
    l.KS = gcp_odd(s.KP);
    l.SS = 32 * s.NTG + 1;
    l.VS = gcp_odd(3 * (s.vi > s.vo ? s.vi : s.vo));
    l.HS = gcp_odd(3 * s.H);
    l.GS = gcp_odd(s.vo);
    int mrg = 32 * l.KS, stg = 32 * l.SS;
    l.o_mrg = 0;
    if (s.NG == 1) {  // single output group: the merged tile doubles as the staging tile (same row stride)
        if (l.KS < l.SS) { l.KS = l.SS; mrg = 32 * l.KS; }
        l.o_stg = 0;
        l.o_vt = mrg;
    } else {
        l.o_stg = mrg;
        l.o_vt = mrg + stg;
    }
    l.o_vht = l.o_vt + 32 * l.VS;
    l.o_gt = l.o_vht + 32 * l.HS;
    l.o_fr = l.o_gt + 32 * l.GS;
    l.o_sw = l.o_fr + 32 * 9;
    l.total = l.o_sw + gcp_small_w_floats(s.vi, s.H, s.vo, s.nf);
    return l;
}

template <int NTG>
struct WFrag;
template <>
struct WFrag<1> {
    float v[1];
    __device__ __forceinline__ void load(const float* p) { v[0] = p[0]; }
};
template <>
struct WFrag<2> {
    float v[2];
    __device__ __forceinline__ void load(const float* p) {
        float2 t = *reinterpret_cast<const float2*>(p);
        v[0] = t.x; v[1] = t.y;
    }
};
template <>
struct WFrag<4> {
    float v[4];
    __device__ __forceinline__ void load(const float* p) {
        float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
};

template <int NTG, int MOT, bool PWL>
__global__ __launch_bounds__(GCP_WAVE, 2) void gcp2_fwd_kernel(FwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const GcpShape& S = p.sh;
    const FwdLds L = fwd_lds(S);
    const int lane = threadIdx.x;
    const int e = lane & 31, hi = lane >> 5;
    const int r0 = blockIdx.x * GCP_TILE_ROWS;
    const int rows = p.rows;
    const int row = r0 + e;
    const bool row_ok = row < rows;
    float* mrg = lds + L.o_mrg;
    float* stg = lds + L.o_stg;
    float* vt = lds + L.o_vt;
    float* vht = lds + L.o_vht;
    float* gt = lds + L.o_gt;
    float* fr = lds + L.o_fr;
    const int si = S.si, vi = S.vi, so = S.so, vo = S.vo, H = S.H;
    const float slope = p.o.slope;
    const float ns_s = gcp_neg_slope(p.o.act_s, slope), ns_v = gcp_neg_slope(p.o.act_v, slope);

    gcp_stamp(p.stamps, p.stamp_cap, 0, lane);
    // ---- 1. stage the tile: scalars into the merged tile, vectors, frames --------------------------------
    // the first scalar and vector segments, the frames and the small weights go out in ONE memory round trip
    {
        GcpSegBuf<16> sb0;
        GcpSegBuf<8> vb0;
        gcp_seg_issue(sb0, p.s_in.ptr[0], p.s_in.idx[0], p.s_in.dim[0], r0, rows, mrg, L.KS, 0, lane);
        if (vi > 0) gcp_seg_issue(vb0, p.v_in.ptr[0], p.v_in.idx[0], 3 * p.v_in.dim[0], r0, rows, vt, L.VS, 0, lane);
        if (vi > 0 && S.nf) gcp_load_frames(p.frames, r0, rows, fr, lane);
        gcp_seg_commit(sb0, mrg, L.KS, 0);
        if (vi > 0) gcp_seg_commit(vb0, vt, L.VS, 0);
        int coff = p.s_in.dim[0];
        for (int sg = 1; sg < p.s_in.n; ++sg) {
            gcp_load_segment(p.s_in.ptr[sg], p.s_in.idx[sg], p.s_in.dim[sg], r0, rows, mrg, L.KS, coff, lane);
            coff += p.s_in.dim[sg];
        }
        coff = vi > 0 ? 3 * p.v_in.dim[0] : 0;
        for (int sg = 1; sg < p.v_in.n; ++sg) {
            gcp_load_segment(p.v_in.ptr[sg], p.v_in.idx[sg], 3 * p.v_in.dim[sg], r0, rows, vt, L.VS, coff, lane);
            coff += 3 * p.v_in.dim[sg];
        }
    }
    const GcpSmallW sw = gcp_stage_small_weights(p.w, H, S.nf, lds + L.o_sw, lane);
    for (int k = S.K + hi; k < S.KP; k += 2) mrg[e * L.KS + k] = 0.f;  // zero the k padding
    gcp_wave_lds_sync();
    gcp_stamp(p.stamps, p.stamp_cap, 1, lane);

    // ---- 2. vector prologue on the VALU: two lanes per row ----------------------------------------------
    if (vi > 0) {
        const float* vrow = vt + e * L.VS;
        for (int h = hi; h < H; h += 2) {  // vector_down + safe_norm over xyz (gcpnet.py:420-421)
            const float* wd = sw.wd + h * vi;
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
            mrg[e * L.KS + si + h] = sqrtf(a0 * a0 + a1 * a1 + a2 * a2 + 1e-8f) + 1e-8f;
        }
        if (S.nf) {  // vector_down_frames + scalarize (gcpnet.py:426-435, components/__init__.py:302,312)
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
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    float pr = f[3 * a + 0] * a0 + f[3 * a + 1] * a1 + f[3 * a + 2] * a2;
                    if (p.o.e3 && a == 1) pr = fabsf(pr);
                    mrg[e * L.KS + si + H + 3 * k + a] = pr;
                }
            }
        }
    }
    gcp_wave_lds_sync();
    gcp_stamp(p.stamps, p.stamp_cap, 2, lane);

    // ---- 3. scalar_out on the matrix cores, one group of NTG 32-wide output tiles at a time ---------------
    const int NOT = S.NOT;
    f32x4 gacc[MOT][2];  // MOT = compile-time bound on the 16-wide gate output tiles (vo <= 16 * MOT)
#pragma unroll
    for (int a = 0; a < MOT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) gacc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool scalar_gate = (p.o.vmode == GCP_VMODE_SCALAR_GATE) && vo > 0;
    const bool single = S.NG == 1;

    for (int g = 0; g < S.NG; ++g) {
        f32x16 acc[NTG];
#pragma unroll
        for (int t = 0; t < NTG; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = 32 * (g * NTG + t) + gcp_crow(r, hi);
                acc[t][r] = j < so ? p.w.b_scalar[j] : 0.f;
            }
        const float* wp = p.w.pack + S.offA + ((int64_t)g * S.KK * 64 + lane) * NTG;
        const float* bp = mrg + e * L.KS + hi;
        constexpr int U = 4;
        // software pipeline over k-pair steps: three rotating batches of U steps, so a batch's weight fragments (one
        // 16-byte L2 load per step) are requested two batches (2 * U * NTG MFMAs ~ 2k cycles) before they are used
        WFrag<NTG> A0[U], A1[U], A2[U];
        float B0[U], B1[U], B2[U];
        const int KK = S.KK;
        auto ld = [&](WFrag<NTG>(&a)[U], float(&b)[U], int kk0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kk = min(kk0 + u, KK - 1);
                a[u].load(wp + (int64_t)kk * 64 * NTG);
                b[u] = bp[2 * kk];
            }
        };
        auto mm = [&](WFrag<NTG>(&a)[U], float(&b)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int t = 0; t < NTG; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].v[t], b[u], acc[t], 0, 0, 0);
        };
        ld(A0, B0, 0);
        ld(A1, B1, U);
        ld(A2, B2, 2 * U);
        for (int kk0 = 0; kk0 < KK; kk0 += 3 * U) {  // KK is a multiple of U
            mm(A0, B0);
            ld(A0, B0, kk0 + 3 * U);
            if (kk0 + U < KK) mm(A1, B1);
            ld(A1, B1, kk0 + 4 * U);
            if (kk0 + 2 * U < KK) mm(A2, B2);
            ld(A2, B2, kk0 + 5 * U);
        }
        gcp_stamp(p.stamps, p.stamp_cap, 3, lane);
        const int c0 = g * 32 * NTG;
        const int gw = min(32 * NTG, so - c0);
        // ---- 4. epilogue of this group ----------------------------------------------------------------------
        // Staging tile: a separate region when there are several output groups; with a single group it is the merged
        // tile itself, same addressing, so each lane overwrites exactly the elements it has just consumed (its own
        // residual inputs) and the MFMA loop no longer needs them.
        float* st = single ? mrg : stg;
        const int sst = single ? L.KS : L.SS;
        gcp_wave_lds_sync();  // every B read of the merged tile by the MFMA loop has completed
        // gate fragments of this group: requested now, consumed after the staging below (one L2 latency, hidden)
        float wgf[8 * NTG];
        if constexpr (MOT == 1) {
            if (scalar_gate) {
                const float* wg0 = p.w.pack + S.offC + ((int64_t)g * 8 * NTG) * 64 + lane;
#pragma unroll
                for (int jj = 0; jj < 8 * NTG; ++jj) wgf[jj] = wg0[(int64_t)jj * 64];
            }
        }
        // (a) s_out = act(s_pre) (+ residual): with the fused residual x is read from the LDS tile, not from HBM.
        // Reads are batched per 32-wide tile before the writes (they alias for the compiler, which would otherwise
        // serialise every read behind the previous write).
#pragma unroll
        for (int t = 0; t < NTG; ++t) {
            float x[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jl = 32 * t + gcp_crow(r, hi);
                x[r] = (p.fused_res && c0 + jl < so) ? mrg[e * L.KS + c0 + jl] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                st[e * sst + 32 * t + gcp_crow(r, hi)] = gcp_actf<PWL>(p.o.act_s, ns_s, slope, acc[t][r]) + x[r];
        }
        gcp_wave_lds_sync();
        if (p.res_s && !p.fused_res) {  // separate residual tensor: slow path, re-read from HBM
            for (int ee = 0; ee < GCP_TILE_ROWS && r0 + ee < rows; ++ee)
                for (int j = lane; j < gw; j += GCP_WAVE) st[ee * sst + j] += p.res_s[(int64_t)(r0 + ee) * so + c0 + j];
            gcp_wave_lds_sync();
        }
        gcp_store_tile(p.s_out, so, c0, gw, r0, rows, st, sst, lane);
        gcp_stamp(p.stamps, p.stamp_cap, 4, lane);
        // (b) the pre-activations: saved for the backward, and B operand of the gate GEMM
        if (p.s_pre || scalar_gate) {
            gcp_wave_lds_sync();
#pragma unroll
            for (int t = 0; t < NTG; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[e * sst + 32 * t + gcp_crow(r, hi)] = acc[t][r];
            gcp_wave_lds_sync();
            if (p.s_pre) gcp_store_tile(p.s_pre, so, c0, gw, r0, rows, st, sst, lane);
            if (scalar_gate) {  // vector_out_scale(act_v(s_pre)) accumulated over this group's columns (gcpnet.py:386)
                const float* wg = p.w.pack + S.offC + ((int64_t)g * 8 * NTG) * 64 + lane;
                const int e16 = lane & 15, q = lane >> 4;
#pragma unroll
                for (int jj = 0; jj < 8 * NTG; ++jj) {
                    float b0 = gcp_actf<PWL>(p.o.act_v, ns_v, slope, st[e16 * sst + 4 * jj + q]);
                    float b1 = gcp_actf<PWL>(p.o.act_v, ns_v, slope, st[(16 + e16) * sst + 4 * jj + q]);
                    if (c0 + 4 * jj + q >= so) { b0 = 0.f; b1 = 0.f; }
                    if constexpr (MOT == 1) {
                        gacc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wgf[jj], b0, gacc[0][0], 0, 0, 0);
                        gacc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wgf[jj], b1, gacc[0][1], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int ot = 0; ot < MOT; ++ot) {
                            if (ot < NOT) {
                                const float a = wg[((int64_t)ot * S.NJ4 + jj) * 64];
                                gacc[ot][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, gacc[ot][0], 0, 0, 0);
                                gacc[ot][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, gacc[ot][1], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
        gcp_wave_lds_sync();
    }

    gcp_stamp(p.stamps, p.stamp_cap, 5, lane);
    if (vo == 0) return;

    // ---- 5. vector epilogue: sigmoid gate, vector_up, gating, residual (gcpnet.py:364-391) -----------------
    if (scalar_gate) {
        const int e16 = lane & 15, q = lane >> 4;
#pragma unroll
        for (int ot = 0; ot < MOT; ++ot) {
            if (ot < NOT) {
#pragma unroll
                for (int eh = 0; eh < 2; ++eh)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int oo = 16 * ot + 4 * q + r;
                        if (oo < vo) gt[(16 * eh + e16) * L.GS + oo] = gcp_sigmoid(gacc[ot][eh][r] + p.w.b_gate[oo]);
                    }
            }
        }
    }
    gcp_wave_lds_sync();
    if (vi == 0) {  // create_zero_vector (gcpnet.py:447-449)
        if (row_ok)
            for (int i = hi; i < 3 * vo; i += 2) {
                const int64_t off = (int64_t)row * 3 * vo + i;
                p.v_out[off] = p.res_v ? p.res_v[off] : 0.f;
            }
        return;
    }
    for (int oc0 = hi; oc0 < vo; oc0 += 16) {  // 8 channels per lane per pass: all reads, then the in-place writes
        float y[8][3];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int oc = oc0 + 2 * i;
            y[i][0] = y[i][1] = y[i][2] = 0.f;
            if (oc < vo) {
                const float* wu = sw.wu + oc * H;
                float u0 = 0.f, u1 = 0.f, u2 = 0.f;
                for (int h = 0; h < H; ++h) {
                    const float w = wu[h];
                    u0 = fmaf(w, vht[e * L.HS + 3 * h + 0], u0);
                    u1 = fmaf(w, vht[e * L.HS + 3 * h + 1], u1);
                    u2 = fmaf(w, vht[e * L.HS + 3 * h + 2], u2);
                }
                float x0 = 0.f, x1 = 0.f, x2 = 0.f;  // this channel of the input (vector residual / ResGCP residual)
                if (p.o.vector_residual || p.fused_res) {
                    x0 = vt[e * L.VS + 3 * oc + 0]; x1 = vt[e * L.VS + 3 * oc + 1]; x2 = vt[e * L.VS + 3 * oc + 2];
                }
                if (p.o.vector_residual) { u0 += x0; u1 += x1; u2 += x2; }
                float sc = 1.f;
                if (scalar_gate) {
                    sc = gt[e * L.GS + oc];
                } else if (p.o.vmode == GCP_VMODE_SELF_GATE) {
                    sc = gcp_actf<PWL>(p.o.act_v, ns_v, slope, sqrtf(u0 * u0 + u1 * u1 + u2 * u2 + 1e-8f) + 1e-8f);
                }
                float y0 = u0 * sc, y1 = u1 * sc, y2 = u2 * sc;
                if (p.fused_res) { y0 += x0; y1 += x1; y2 += x2; }
                if (p.res_v && !p.fused_res && row_ok) {
                    const int64_t off = ((int64_t)row * vo + oc) * 3;
                    y0 += p.res_v[off]; y1 += p.res_v[off + 1]; y2 += p.res_v[off + 2];
                }
                y[i][0] = y0; y[i][1] = y1; y[i][2] = y2;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int oc = oc0 + 2 * i;
            // in place: this lane is the only reader of channel oc of its row (oc < vi whenever the input is read)
            if (oc < vo) { vt[e * L.VS + 3 * oc + 0] = y[i][0]; vt[e * L.VS + 3 * oc + 1] = y[i][1]; vt[e * L.VS + 3 * oc + 2] = y[i][2]; }
        }
    }
    gcp_wave_lds_sync();
    gcp_store_tile(p.v_out, 3 * vo, 0, 3 * vo, r0, rows, vt, L.VS, lane);
    if (scalar_gate && p.gate) gcp_store_tile(p.gate, vo, 0, vo, r0, rows, gt, L.GS, lane);
    gcp_stamp(p.stamps, p.stamp_cap, 6, lane);
}

__global__ void pack_gcp2_kernel(gcp2_weights_t w, GcpShape S, float* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S.total) return;
    float val = 0.f;
    if (i < S.offB) {  // A: forward fragments [NG][KK][64][NTG] = W[j, k]
        int64_t x = i - S.offA;
        const int t = x % S.NTG; x /= S.NTG;
        const int lane = x % 64; x /= 64;
        const int kk = x % S.KK;
        const int g = x / S.KK;
        const int j = 32 * (g * S.NTG + t) + (lane & 31), k = 2 * kk + (lane >> 5);
        if (j < S.so && k < S.K) val = w.w_scalar[(int64_t)j * S.K + k];
    } else if (i < S.offC) {  // B: backward-data fragments [NGK][NS][64][NUG] = W[j(step, half), k]
        int64_t x = i - S.offB;
        const int uu = x % S.NUG; x /= S.NUG;
        const int lane = x % 64; x /= 64;
        const int st = x % S.NS;
        const int ug = x / S.NS;
        const int j = 32 * (st / 16) + (((st % 16) & 3) + 8 * ((st % 16) >> 2) + 4 * (lane >> 5));
        const int k = 32 * (ug * S.NUG + uu) + (lane & 31);
        if (j < S.so && k < S.K) val = w.w_scalar[(int64_t)j * S.K + k];
    } else if (i < S.offD) {  // C: gate forward fragments (16x16x4) [NOT][NJ4][64] = Wg[o, j]
        int64_t x = i - S.offC;
        const int lane = x % 64; x /= 64;
        const int jj = x % S.NJ4;
        const int ot = x / S.NJ4;
        const int o = 16 * ot + (lane & 15), j = 4 * jj + (lane >> 4);
        if (w.w_gate && o < S.vo && j < S.so) val = w.w_gate[(int64_t)o * S.so + j];
    } else {  // D: gate backward fragments (32x32x2) [NOO][NG][64][NTG] = Wg[o, j]
        int64_t x = i - S.offD;
        const int t = x % S.NTG; x /= S.NTG;
        const int lane = x % 64; x /= 64;
        const int g = x % S.NG;
        const int oo = x / S.NG;
        const int o = 2 * oo + (lane >> 5), j = 32 * (g * S.NTG + t) + (lane & 31);
        if (w.w_gate && o < S.vo && j < S.so) val = w.w_gate[(int64_t)o * S.so + j];
    }
    out[i] = val;
}

template <int NTG, int MOT, bool PWL>
int launch_fwd3(const FwdParams& p, dim3 grid, size_t lds_bytes, hipStream_t st) {
    static size_t cur_max = 64 * 1024;  // dynamic LDS above 64 KiB needs an explicit opt-in, once per size
    if (lds_bytes > cur_max) {
        hipError_t err = hipFuncSetAttribute((const void*)gcp2_fwd_kernel<NTG, MOT, PWL>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (err != hipSuccess) return (int)err;
        cur_max = lds_bytes;
    }
    hipLaunchKernelGGL((gcp2_fwd_kernel<NTG, MOT, PWL>), grid, dim3(GCP_WAVE), lds_bytes, st, p);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

template <int NTG, int MOT>
int launch_fwd2(const FwdParams& p, dim3 grid, size_t lds_bytes, hipStream_t st) {
    if (gcp_is_pwl(p.o.act_s) && gcp_is_pwl(p.o.act_v)) return launch_fwd3<NTG, MOT, true>(p, grid, lds_bytes, st);
    return launch_fwd3<NTG, MOT, false>(p, grid, lds_bytes, st);
}

template <int NTG>
int launch_fwd(const FwdParams& p, dim3 grid, size_t lds_bytes, hipStream_t st) {
    if (p.sh.NOT <= 1) return launch_fwd2<NTG, 1>(p, grid, lds_bytes, st);
    if (p.sh.NOT <= 2) return launch_fwd2<NTG, 2>(p, grid, lds_bytes, st);
    return launch_fwd2<NTG, 4>(p, grid, lds_bytes, st);
}

int check_concat(const gcp_concat_t* c, int total) {
    if (!c || c->n < 0 || c->n > GCP_MAX_SEG) return GCPNET_E_BADARG;
    int sum = 0;
    for (int k = 0; k < c->n; ++k) {
        if (!c->ptr[k] || c->dim[k] <= 0) return GCPNET_E_BADARG;
        sum += c->dim[k];
    }
    return sum == total ? 0 : GCPNET_E_BADARG;
}

}  // namespace

extern "C" int gcpnet_abi_version(void) { return GCPNET_ABI_VERSION; }

extern "C" int64_t gcpnet_gcp2_pack_floats(int si, int vi, int so, int vo, int hidden, int use_frames) {
    return gcp_shape(si, vi, so, vo, hidden, use_frames).total;
}

extern "C" int gcpnet_pack_gcp2_weights(const gcp2_weights_t* w, float* pack_out, void* stream) {
    if (!w || !pack_out || !w->w_scalar) return GCPNET_E_BADARG;
    const GcpShape S = gcp_shape(w->si, w->vi, w->so, w->vo, w->hidden, w->use_frames);
    const int threads = 256;
    const int64_t blocks = (S.total + threads - 1) / threads;
    hipLaunchKernelGGL(pack_gcp2_kernel, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream, *w, S, pack_out);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int gcpnet_gcp2_forward(int rows, const gcp_concat_t* s_in, const gcp_concat_t* v_in, const float* frames,
                                   const gcp2_weights_t* w, const gcp2_opts_t* opts, const float* res_s,
                                   const float* res_v, float* s_out, float* v_out, float* s_pre, float* gate,
                                   void* stream) {
    if (rows < 0 || !w || !opts || !s_out || !w->pack || !w->b_scalar) return GCPNET_E_BADARG;
    if (rows == 0) return 0;
    if (check_concat(s_in, w->si)) return GCPNET_E_BADARG;
    if (w->vi > 0) {
        if (check_concat(v_in, w->vi) || !w->w_down) return GCPNET_E_BADARG;
        if (w->use_frames && (!frames || !w->w_frames)) return GCPNET_E_BADARG;
    }
    if (w->vo > 0 && !v_out) return GCPNET_E_BADARG;
    if (w->vo > 0 && w->vi > 0 && !w->w_up) return GCPNET_E_BADARG;
    if (w->vo > 64) return GCPNET_E_UNSUPPORTED;
    if (opts->vmode == GCP_VMODE_SCALAR_GATE && w->vo > 0 && w->vi > 0 && (!w->w_gate || !w->b_gate))
        return GCPNET_E_BADARG;
    if (opts->vector_residual && w->vi != w->vo) return GCPNET_E_BADARG;

    FwdParams p;
    p.rows = rows;
    p.s_in = *s_in;
    if (w->vi > 0) p.v_in = *v_in; else p.v_in.n = 0;
    p.frames = frames;
    p.w = *w;
    p.o = *opts;
    if (w->vi == 0) p.o.vmode = GCP_VMODE_NONE;  // zero vectors: nothing to gate (gcpnet.py:447-449)
    p.res_s = res_s; p.res_v = res_v;
    p.s_out = s_out; p.v_out = v_out; p.s_pre = s_pre; p.gate = gate;
    p.fused_res = (res_s && s_in->n == 1 && !s_in->idx[0] && res_s == s_in->ptr[0] && w->si == w->so &&
                   (w->vo == 0 || (res_v && w->vi == w->vo && v_in->n == 1 && !v_in->idx[0] && res_v == v_in->ptr[0])))
                      ? 1 : 0;
    p.stamps = g_gcp_phase_buf; p.stamp_cap = g_gcp_phase_cap;
    p.sh = gcp_shape(w->si, w->vi, w->so, w->vo, w->hidden, w->use_frames);
    const FwdLds L = fwd_lds(p.sh);
    const size_t lds_bytes = (size_t)L.total * sizeof(float);
    if (lds_bytes > 160 * 1024) return GCPNET_E_UNSUPPORTED;
    const dim3 grid((unsigned)gcp_cdiv(rows, GCP_TILE_ROWS));
    hipStream_t st = (hipStream_t)stream;
    switch (p.sh.NTG) {
        case 1: return launch_fwd<1>(p, grid, lds_bytes, st);
        case 2: return launch_fwd<2>(p, grid, lds_bytes, st);
        default: return launch_fwd<4>(p, grid, lds_bytes, st);
    }
}
