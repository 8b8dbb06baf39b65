// GCP2 forward on gfx950 -- replaces GCP2.forward (reference src/models/components/gcpnet.py:394-468), and chains of
// residual GCP2 blocks (ResGCP, gcpnet.py:921-924) in one launch.
//
// Design (MI355X-first):
//   * one 64-lane wavefront owns a tile of 32 rows (edges or nodes) and is fully autonomous: workgroup = 1 wave, no
//     s_barrier, LDS hand-offs need only lgkmcnt(0); the CU's SIMDs interleave independent tiles;
//   * scalar_out is computed TRANSPOSED, s_pre^T[so, 32 rows] = W[so, K] x merged^T[K, 32 rows], with
//     v_mfma_f32_32x32x2_f32 (exact fp32): A fragments are pre-packed weights streamed from L2 (one 16-byte load per
//     k-pair, three batches in flight), the B fragment is one conflict-free ds_read_b32 from the wave-private merged
//     tile (odd row stride), and every row's result lands in just two lanes;
//   * the vector gate Linear (vector_out_scale) is fed STRAIGHT FROM THE ACCUMULATOR REGISTERS: a 32x32 C/D register
//     (t, r) of lane half `hi` holds column j = 32t + (r&3) + 8(r>>2) + 4hi of row lane&31, which is exactly a B fragment
//     of a k-pair step over (j0, j0+4); the gate weights are packed for that pairing, so no staging is needed;
//   * s_pre is stored from the registers (16-byte pieces), the new state x + act(s_pre) is written IN PLACE over the
//     x tile (each lane overwrites the elements it has just consumed) and leaves as full 512-byte rows;
//   * the tiny vector Linears (K = 4..36) run on the VALU, two lanes per row, with their weights in LDS;
//   * concatenations [h_row | e | h_col] are never materialised: the tile loader gathers each source;
//   * chain mode: after a block the tile IS the next block's input -- no reload, and the stores of block k drain under
//     the MFMAs of block k+1.
#include "common.h"
#include "tile_io.h"
#include "vec_mfma.h"
#include "gcp_bf16x3.h"
#include "gcp_f16x2.h"

int gcp2_chain_fwd_registers(int rows, const float* s0, const float* v0, const float* frames, int n,
                             const gcp2_chain_item_t* items, hipStream_t st);

namespace {

struct ChainItem {
    const float* pack;
    const float* b_scalar;
    const float* w_down;
    const float* w_frames;
    const float* w_up;
    const float* b_gate;
    float* s_out;
    float* v_out;
    float* s_pre;
    float* gate;
    int act_s, act_v;
};

struct FwdParams {
    int rows;
    gcp_concat_t s_in, v_in;
    const float* frames;
    gcp2_opts_t o;  // shared: slope, vmode, vector_residual, e3 (activations are per item)
    int n;
    ChainItem it[GCP_MAX_CHAIN];
    gcp_concat_t v_add;  // pre-projected vector inputs: rows of [n_src, 3, HFP] tables gathered and added to [vh | vf] (n == 1)
    gcp_concat_t s_add;  // pre-projected scalar inputs: rows of [n_src, so] tables gathered and added to s_pre (n == 1, NG == 1)
    const float* res_s;  // separate residual tensors (n == 1 only)
    const float* res_v;
    int fused_res;  // out = x + GCP(x) with x the tile itself
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
    int mrg = 32 * l.KS;
    const int stg = 32 * l.SS;
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
    // vector_down outputs between the vector prologue and epilogue: row-major (VALU path) or parked registers (MFMA path)
    const int vht = 32 * l.HS, ust = s.vmm ? s.SVB * 3 * 64 : 0;
    l.o_gt = l.o_vht + (vht > ust ? vht : ust);
    l.o_fr = l.o_gt + 32 * l.GS;
    l.o_sw = l.o_fr + 32 * 9;
    l.total = l.o_sw + gcp_small_w_lds_floats(s.vi, s.H, s.vo, s.nf);
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

// NTG: 32-wide output tiles per accumulator group; GT: 32-wide tiles of the gate outputs (vo <= 32 * GT);
// PWL: both activations are identity / relu / leakyrelu.
template <int NTG, int GT, bool PWL>
__global__ __launch_bounds__(GCP_WAVE, 1) void gcp2_fwd_kernel(FwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const GcpShape& S = p.sh;
    const FwdLds L = fwd_lds(S);
    int lane = threadIdx.x;
    int e = lane & 31, hi = lane >> 5;
    const int r0 = blockIdx.x * GCP_TILE_ROWS;
    const int rows = p.rows;
    int row = r0 + e;
    bool row_ok = row < rows;
    float* mrg = lds + L.o_mrg;
    float* stg = lds + L.o_stg;
    float* vt = lds + L.o_vt;
    float* vht = lds + L.o_vht;
    float* gt = lds + L.o_gt;
    float* fr = lds + L.o_fr;
    const int si = S.si, vi = S.vi, so = S.so, vo = S.vo, H = S.H;
    const float slope = p.o.slope;
    const bool scalar_gate = (p.o.vmode == GCP_VMODE_SCALAR_GATE) && vo > 0;
    const bool single = S.NG == 1;
    const bool vec_so = (so & 3) == 0;

    gcp_stamp(p.stamps, p.stamp_cap, 0, lane);
    // ---- 1. stage the tile: the first scalar and vector segments and the frames go out in ONE memory round trip ---
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
    for (int k = S.K + hi; k < S.KP; k += 2) mrg[e * L.KS + k] = 0.f;  // zero the k padding (stays zero along a chain)
    gcp_stamp(p.stamps, p.stamp_cap, 1, lane);

    for (int ci = 0; ci < p.n; ++ci) {
        // Opaque to the optimiser: otherwise every per-lane address of the loop body (hundreds of values) is hoisted
        // out of the chain loop as loop-invariant and then spilled; recomputing them per block costs a few VALU ops.
        asm volatile("" : "+v"(lane), "+v"(e), "+v"(hi));
        row = r0 + e;
        row_ok = row < rows;
        const ChainItem& it = p.it[ci];
        const float ns_s = gcp_neg_slope(it.act_s, slope), ns_v = gcp_neg_slope(it.act_v, slope);
        gcp2_weights_t wsm;
        wsm.vi = vi; wsm.vo = vo; wsm.w_down = it.w_down; wsm.w_frames = it.w_frames; wsm.w_up = it.w_up;
        gcp_wave_lds_sync();  // the previous block is done with the small-weight area
        const bool vmm = S.vmm != 0;  // the small vector Linears run on the matrix cores (vec_mfma.h); else on the VALU
        GcpSmallW sw;
        sw.wd = sw.wf = sw.wu = nullptr;
        if (!vmm) sw = gcp_stage_small_weights(wsm, H, S.nf, lds + L.o_sw, lane);
        gcp_wave_lds_sync();

        // ---- 2. vector prologue: vh = vector_down(v) and its norms over xyz (gcpnet.py:420-421), vector_down_frames +
        //         scalarize (gcpnet.py:426-435, components/__init__.py:302,312) -> columns si.. of the merged tile -----------
        if (vmm) {
            // pre-projected vector inputs (node-level products done by the caller): gathered rows of [n_src, 3, HFP] tables,
            // requested before the product below and added to its result -- channels crow(r, hi) of this lane, four at a time
            float qa[4][12];
            const bool has_vadd = p.v_add.n > 0;  // wave-uniform
            if (has_vadd) {
                const int HFP = gcp_round_up(S.HF, 4);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < 12; ++j) qa[q][j] = 0.f;
                for (int k = 0; k < p.v_add.n; ++k) {
                    const int32_t* ix = p.v_add.idx[k];
                    const int rc = min(row, rows - 1);
                    const float* trow = p.v_add.ptr[k] + (ix ? (int64_t)ix[rc] : (int64_t)rc) * 3 * HFP;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int x0 = 8 * q + 4 * hi;
                        const bool on = x0 < S.HF;
#pragma unroll
                        for (int d = 0; d < 3; ++d) {
                            const float4 t = *reinterpret_cast<const float4*>(trow + d * HFP + (on ? x0 : 0));
                            qa[q][4 * d + 0] += on ? t.x : 0.f; qa[q][4 * d + 1] += on ? t.y : 0.f;
                            qa[q][4 * d + 2] += on ? t.z : 0.f; qa[q][4 * d + 3] += on ? t.w : 0.f;
                        }
                    }
                }
            }
            gcp_xyz_acc u;
            gcp_vmm_down_loop(it.pack + S.offVA + lane, S.SVA, vi, vt + e * L.VS, hi, u);
            if (has_vadd) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int d = 0; d < 3; ++d)
#pragma unroll
                        for (int i = 0; i < 4; ++i) u[d][4 * q + i] += qa[q][4 * d + i];
            }
            float f[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) f[i] = S.nf ? fr[e * 9 + i] : 0.f;
            float* ust = vht;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int x = gcp_crow(r, hi);
                const float u0 = u[0][r], u1 = u[1][r], u2 = u[2][r];
                if (r < S.SVB) {  // parked for vector_up in the epilogue (wave-uniform guard)
                    ust[(r * 3 + 0) * 64 + lane] = u0; ust[(r * 3 + 1) * 64 + lane] = u1; ust[(r * 3 + 2) * 64 + lane] = u2;
                }
                if (x < H) {
                    mrg[e * L.KS + si + x] = sqrtf(u0 * u0 + u1 * u1 + u2 * u2 + 1e-8f) + 1e-8f;
                } else if (x < S.HF) {
                    const int k = x - H;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        float pr = f[3 * a + 0] * u0 + f[3 * a + 1] * u1 + f[3 * a + 2] * u2;
                        if (p.o.e3 && a == 1) pr = fabsf(pr);
                        mrg[e * L.KS + si + H + 3 * k + a] = pr;
                    }
                }
            }
        } else if (vi > 0) {
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
        if (ci == 0) gcp_stamp(p.stamps, p.stamp_cap, 2, lane);

        // ---- 3. scalar_out on the matrix cores, one group of NTG 32-wide output tiles at a time -----------------
        f32x16 gacc[GT];
#pragma unroll
        for (int a = 0; a < GT; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) gacc[a][r] = 0.f;

        const bool vec_b = vec_so && ((reinterpret_cast<uintptr_t>(it.b_scalar) & 15) == 0);
        for (int g = 0; g < S.NG; ++g) {
            f32x16 acc[NTG];
            // (a lane's columns 8 q + 4 hi .. + 3 of a tile are contiguous: one 16-byte load per register quad)
            gcp_load_acc_layout<NTG, false>(it.b_scalar, 0, so, 32 * g * NTG, hi, true, vec_b, acc);
            // Pre-projected inputs (node-level GEMMs done by the caller): their gathered rows are requested here, in the
            // accumulator layout, and added after the k loop -- the gather latency is spent under the MFMAs.
            f32x16 pre[NTG];
            const bool has_add = p.s_add.n > 0;  // wave-uniform; n == 1 only
            if (has_add) {
#pragma unroll
                for (int t = 0; t < NTG; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) pre[t][r] = 0.f;
                for (int k = 0; k < p.s_add.n; ++k) {
                    const int32_t* ix = p.s_add.idx[k];
                    const int rc = min(row, rows - 1);
                    const int64_t src = ix ? (int64_t)ix[rc] : (int64_t)rc;
                    gcp_load_acc_layout<NTG, true>(p.s_add.ptr[k], src, so, 32 * g * NTG, hi, true, vec_so, pre);
                }
            }
            const float* wp = it.pack + S.offA + ((int64_t)g * S.KK * 64 + lane) * NTG;
            const float* bp = mrg + e * L.KS + hi;
            constexpr int U = 4;
            // software pipeline over k-pair steps: three rotating batches of U steps, so a batch's weight fragments are
            // requested two batches (2 * U * NTG MFMAs ~ 2k cycles) before they are used
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
            __builtin_amdgcn_sched_barrier(0);
            for (int kk0 = 0; kk0 < KK; kk0 += 3 * U) {  // KK is a multiple of U
                // the sched_barriers pin each prefetch where it is written: hipcc otherwise sinks the loads to their use
                mm(A0, B0);
                ld(A0, B0, kk0 + 3 * U);
                __builtin_amdgcn_sched_barrier(0);
                if (kk0 + U < KK) mm(A1, B1);
                ld(A1, B1, kk0 + 4 * U);
                __builtin_amdgcn_sched_barrier(0);
                if (kk0 + 2 * U < KK) mm(A2, B2);
                ld(A2, B2, kk0 + 5 * U);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (has_add) {
#pragma unroll
                for (int t = 0; t < NTG; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] += pre[t][r];
            }
            if (ci == 0) gcp_stamp(p.stamps, p.stamp_cap, 3, lane);
            const int c0 = g * 32 * NTG;
            const int gw = min(32 * NTG, so - c0);

            // ---- 4a. vector gate Linear on the matrix cores, B fragments = the accumulator registers (gcpnet.py:386) --
            if (scalar_gate) {
#pragma unroll
                for (int a = 0; a < GT; ++a) {
                    const float* wg = it.pack + S.offC + (((int64_t)a * S.NS + (int64_t)(g * NTG) * 16) * 64 + lane);
                    float wa[16], wb[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) wa[r] = wg[(int64_t)r * 64];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < NTG; ++t) {
                        float(&cur)[16] = (t & 1) ? wb : wa;
                        float(&nxt)[16] = (t & 1) ? wa : wb;
                        if (t + 1 < NTG) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) nxt[r] = wg[(int64_t)((t + 1) * 16 + r) * 64];
                        }
                        __builtin_amdgcn_sched_barrier(0);  // next tile's fragments stay in flight under these MFMAs
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float b = gcp_actf<PWL>(it.act_v, ns_v, slope, acc[t][r]);
                            gacc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[r], b, gacc[a], 0, 0, 0);
                        }
                    }
                }
            }
            // ---- 4b. pre-activations saved for the backward: straight from the registers -----------------------------
            if (it.s_pre) {
#pragma unroll
                for (int t = 0; t < NTG; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int j0 = c0 + 32 * t + 8 * q + 4 * hi;
                        gcp_store4(it.s_pre, row, so, j0,
                                   make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]), row_ok, vec_so);
                    }
            }
            // ---- 4c. s_out = act(s_pre) (+ residual).  Staging tile: a separate region when there are several output
            // groups; with a single group it is the merged tile itself, same addressing, so each lane overwrites exactly
            // the residual inputs it has just consumed -- and the tile then holds the next block's input.
            float* st = single ? mrg : stg;
            const int sst = single ? L.KS : L.SS;
            gcp_wave_lds_sync();  // every B read of the merged tile by the MFMA loop has completed
#pragma unroll
            for (int t = 0; t < NTG; ++t) {  // reads batched before writes (they alias for the compiler)
                float x[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int jl = 32 * t + gcp_crow(r, hi);
                    x[r] = (p.fused_res && c0 + jl < so) ? mrg[e * L.KS + c0 + jl] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    st[e * sst + 32 * t + gcp_crow(r, hi)] = gcp_actf<PWL>(it.act_s, ns_s, slope, acc[t][r]) + x[r];
            }
            gcp_wave_lds_sync();
            if (p.res_s && !p.fused_res) {  // separate residual tensor: slow path, re-read from HBM
                for (int ee = 0; ee < GCP_TILE_ROWS && r0 + ee < rows; ++ee)
                    for (int j = lane; j < gw; j += GCP_WAVE) st[ee * sst + j] += p.res_s[(int64_t)(r0 + ee) * so + c0 + j];
                gcp_wave_lds_sync();
            }
            if (it.s_out) gcp_store_tile(it.s_out, so, c0, gw, r0, rows, st, sst, lane);
            if (ci == 0) gcp_stamp(p.stamps, p.stamp_cap, 4, lane);
        }
        if (vo == 0) continue;

        // ---- 5. vector epilogue: sigmoid gate, vector_up, gating, residual (gcpnet.py:364-391) -------------------
        if (scalar_gate && !vmm) {
#pragma unroll
            for (int a = 0; a < GT; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int oo = 32 * a + gcp_crow(r, hi);
                    const float bg = it.b_gate[min(oo, vo - 1)];  // unconditional (clamped) load: a guarded one is waited for on the spot
                    if (oo < vo) gt[e * L.GS + oo] = gcp_sigmoid(gacc[a][r] + bg);
                }
        }
        gcp_wave_lds_sync();
        if (ci == 0) gcp_stamp(p.stamps, p.stamp_cap, 5, lane);
        if (vi == 0) {  // create_zero_vector (gcpnet.py:447-449)
            if (row_ok)
                for (int i = hi; i < 3 * vo; i += 2) {
                    const int64_t off = (int64_t)row * 3 * vo + i;
                    it.v_out[off] = p.res_v ? p.res_v[off] : 0.f;
                }
            continue;
        }
        if (vmm) {
            // vector_up on the matrix cores (B fragments = the parked vector_down outputs); sigmoid gate, gating and the
            // residuals element-wise in registers; the vector tile is updated in place (channel crow(r, hi) of row e)
            gcp_xyz_acc uin, vu;
            const float* ust = vht;
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int d = 0; d < 3; ++d) uin[d][r] = r < S.SVB ? ust[(r * 3 + d) * 64 + lane] : 0.f;
            gcp_xyz_zero(vu);
            gcp_vmm_regs<16>(it.pack + S.offVB + lane, S.SVB, uin, vu);
            float sg[16], x[16][3];
            const bool need_x = p.o.vector_residual || p.fused_res;
            float bg[16];  // the gate bias, requested in one go (a load inside the conditional below would be waited for on the spot)
            if (scalar_gate) {
#pragma unroll
                for (int r = 0; r < 16; ++r) bg[r] = it.b_gate[min(gcp_crow(r, hi), vo - 1)];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = min(gcp_crow(r, hi), vo - 1);
                sg[r] = scalar_gate ? gcp_sigmoid(gacc[0][r] + bg[r]) : 1.f;
#pragma unroll
                for (int d = 0; d < 3; ++d) x[r][d] = need_x ? vt[e * L.VS + 3 * o + d] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = gcp_crow(r, hi);
                if (o < vo) {
                    float u0 = vu[0][r], u1 = vu[1][r], u2 = vu[2][r];
                    if (p.o.vector_residual) { u0 += x[r][0]; u1 += x[r][1]; u2 += x[r][2]; }
                    float sc = sg[r];
                    if (p.o.vmode == GCP_VMODE_SELF_GATE)
                        sc = gcp_actf<PWL>(it.act_v, ns_v, slope, sqrtf(u0 * u0 + u1 * u1 + u2 * u2 + 1e-8f) + 1e-8f);
                    float y0 = u0 * sc, y1 = u1 * sc, y2 = u2 * sc;
                    if (p.fused_res) { y0 += x[r][0]; y1 += x[r][1]; y2 += x[r][2]; }
                    if (p.res_v && !p.fused_res && row_ok) {
                        const int64_t off = ((int64_t)row * vo + o) * 3;
                        y0 += p.res_v[off]; y1 += p.res_v[off + 1]; y2 += p.res_v[off + 2];
                    }
                    vt[e * L.VS + 3 * o + 0] = y0; vt[e * L.VS + 3 * o + 1] = y1; vt[e * L.VS + 3 * o + 2] = y2;
                }
            }
            if (scalar_gate && it.gate) {  // saved for the backward, straight from the registers
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (8 * q < vo)
                        gcp_store4(it.gate, row, vo, 8 * q + 4 * hi, make_float4(sg[4 * q], sg[4 * q + 1], sg[4 * q + 2], sg[4 * q + 3]),
                                   row_ok, (vo & 3) == 0);
            }
        } else
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
                        sc = gcp_actf<PWL>(it.act_v, ns_v, slope, sqrtf(u0 * u0 + u1 * u1 + u2 * u2 + 1e-8f) + 1e-8f);
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
        if (it.v_out) gcp_store_tile(it.v_out, 3 * vo, 0, 3 * vo, r0, rows, vt, L.VS, lane);
        if (!vmm && scalar_gate && it.gate) gcp_store_tile(it.gate, vo, 0, vo, r0, rows, gt, L.GS, lane);
        if (ci == 0) gcp_stamp(p.stamps, p.stamp_cap, 6, lane);
    }
}

__device__ __forceinline__ void pack_gcp2_element(const gcp2_weights_t& w, const GcpShape& S, float* out, int64_t i);

__global__ void pack_gcp2_kernel(gcp2_weights_t w, GcpShape S, float* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S.total) return;
    pack_gcp2_element(w, S, out, i);
}

// the blocks of a ResGCP chain (one shape) in ONE launch: blockIdx.y = block
struct PackMultiArgs {
    gcp2_weights_t w[GCP_MAX_CHAIN];
    float* out[GCP_MAX_CHAIN];
};
__global__ void pack_gcp2_multi_kernel(PackMultiArgs a, GcpShape S) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S.total) return;
    pack_gcp2_element(a.w[blockIdx.y], S, a.out[blockIdx.y], i);
}

__device__ __forceinline__ void pack_gcp2_element(const gcp2_weights_t& w, const GcpShape& S, float* out, int64_t i) {
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
        const int j = 32 * (st / 16) + gcp_crow(st % 16, lane >> 5);
        const int k = 32 * (ug * S.NUG + uu) + (lane & 31);
        if (j < S.so && k < S.K) val = w.w_scalar[(int64_t)j * S.K + k];
    } else if (i < S.offD) {  // C: gate forward fragments [GT][NS][64] = Wg[o, j(step, half)]
        int64_t x = i - S.offC;
        const int lane = x % 64; x /= 64;
        const int st = x % S.NS;
        const int a = x / S.NS;
        const int o = 32 * a + (lane & 31), j = 32 * (st / 16) + gcp_crow(st % 16, lane >> 5);
        if (w.w_gate && o < S.vo && j < S.so) val = w.w_gate[(int64_t)o * S.so + j];
    } else if (i >= S.offF6) {  // F6 / C6: forward weights over the register-resident state / gate Linear as three bf16 terms
        const bool gate = i >= S.offC6;
        int64_t x = i - (gate ? S.offC6 : S.offF6);
        const int d = x % 4; x /= 4;
        const int lane = x % 64; x /= 64;
        const int nterm = gate ? 3 : GCP_W6_TERMS;  // (F6: two fp16 terms by default, gcp_f16x2.h; the gate image: three bf16 terms)
        const int term = x % nterm; x /= nterm;
        const int t = gate ? 0 : (int)(x % S.NTG);
        const int j = (int)(gate ? x : x / S.NTG);
        const int orow = 32 * t + (lane & 31);  // output column of scalar_out / gate channel
        unsigned bits = 0;
        for (int h2 = 0; h2 < 2; ++h2) {
            const int r = 8 * (j & 1) + 2 * d + h2;
            const int k = 32 * (j >> 1) + gcp_crow(r, lane >> 5);  // state column / s_pre column
            float wv = 0.f;
            if (gate) { if (w.w_gate && orow < S.vo && k < S.so) wv = w.w_gate[(int64_t)orow * S.so + k]; }
            else if (orow < S.so && k < S.si) wv = w.w_scalar[(int64_t)orow * S.K + k];
            bits |= ((!gate && GCP_ARITH_F16X2) ? gcp_f16x2_wterm(wv, term) : gcp_bf16x3_term(wv, term)) << (16 * h2);
        }
        val = __uint_as_float(bits);
    } else if (i >= S.offB6) {  // B6: backward-data weights as three bf16 terms, [slab][tile of K][term][64][4 x 2 bf16] (gcp_bf16x3.h)
        int64_t x = i - S.offB6;
        const int d = x % 4; x /= 4;
        const int lane = x % 64; x /= 64;
        const int term = x % GCP_W6_TERMS; x /= GCP_W6_TERMS;
        const int uu = x % S.NKT;
        const int j = (int)(x / S.NKT);
        const int kp = 32 * uu + (lane & 31);  // padded merged axis -> column of w_scalar: scalars, then (from tile NTS on) the rest
        const int k = uu < S.NTS ? (kp < S.si ? kp : S.K) : S.si + (kp - 32 * S.NTS);
        unsigned bits = 0;
        for (int h2 = 0; h2 < 2; ++h2) {
            const int r = 8 * (j & 1) + 2 * d + h2;  // accumulator register of tile j / 2 that is element 2 d + h2 of the slab
            const int c = 32 * (j >> 1) + gcp_crow(r, lane >> 5);
            const float wv = (c < S.so && k < S.K) ? w.w_scalar[(int64_t)c * S.K + k] : 0.f;
            bits |= (GCP_ARITH_F16X2 ? gcp_f16x2_wterm(wv, term) : gcp_bf16x3_term(wv, term)) << (16 * h2);
        }
        val = __uint_as_float(bits);
    } else if (i >= S.offVA) {  // V: the small vector Linears as MFMA A fragments (vec_mfma.h), all [step][64]
        // Wdf = [vector_down ; vector_down_frames] is [HF, vi]; Wu = vector_up is [vo, H]
        auto wdf = [&](int x, int c) -> float {
            if (x < S.H) return w.w_down[(int64_t)x * S.vi + c];
            return w.w_frames[(int64_t)(x - S.H) * S.vi + c];
        };
        if (i < S.offVB) {  // VA[s][lane] = Wdf[x = lane & 31, c = 2 s + half]
            const int64_t x0 = i - S.offVA;
            const int lane = x0 % 64, s = (int)(x0 / 64);
            const int x = lane & 31, c = 2 * s + (lane >> 5);
            if (x < S.HF && c < S.vi) val = wdf(x, c);
        } else if (i < S.offVC) {  // VB[r][lane] = Wu[o = lane & 31, h = crow(r, half)]
            const int64_t x0 = i - S.offVB;
            const int lane = x0 % 64, r = (int)(x0 / 64);
            const int o = lane & 31, h = gcp_crow(r, lane >> 5);
            if (o < S.vo && h < S.H) val = w.w_up[(int64_t)o * S.H + h];
        } else if (i < S.offVD) {  // VC[r][lane] = Wu[o = crow(r, half), h = lane & 31]
            const int64_t x0 = i - S.offVC;
            const int lane = x0 % 64, r = (int)(x0 / 64);
            const int h = lane & 31, o = gcp_crow(r, lane >> 5);
            if (o < S.vo && h < S.H) val = w.w_up[(int64_t)o * S.H + h];
        } else {  // VD[ct][r][lane] = Wdf[x = crow(r, half), c = 32 ct + (lane & 31)]
            int64_t x0 = i - S.offVD;
            const int lane = x0 % 64; x0 /= 64;
            const int r = x0 % S.SVD, ct = (int)(x0 / S.SVD);
            const int c = 32 * ct + (lane & 31), x = gcp_crow(r, lane >> 5);
            if (x < S.HF && c < S.vi) val = wdf(x, c);
        }
    } else if (i >= S.offF) {  // F: forward fragments for a register-resident state [NTS*16][64][NTG] = W[j, k(step, half)]
        int64_t x = i - S.offF;
        const int t = x % S.NTG; x /= S.NTG;
        const int lane = x % 64; x /= 64;
        const int st = (int)x;
        const int j = 32 * t + (lane & 31), k = 32 * (st / 16) + gcp_crow(st % 16, lane >> 5);
        if (j < S.so && k < S.si) val = w.w_scalar[(int64_t)j * S.K + k];
    } else {  // D: gate backward fragments (32x32x2) [NOO][NG][64][NTG] = Wg[o, j]
        int64_t x = i - S.offD;
        const int t = x % S.NTG; x /= S.NTG;
        const int lane = x % 64; x /= 64;
        const int g = x % S.NG;
        const int oo = x / S.NG;
        const int o = gcp_crow(oo, lane >> 5), j = 32 * (g * S.NTG + t) + (lane & 31);
        if (w.w_gate && o < S.vo && j < S.so) val = w.w_gate[(int64_t)o * S.so + j];
    }
    out[i] = val;
}

template <int NTG, int GT, bool PWL>
int launch_fwd3(const FwdParams& p, dim3 grid, size_t lds_bytes, hipStream_t st) {
    static size_t cur_max = 64 * 1024;  // dynamic LDS above 64 KiB needs an explicit opt-in, once per size
    if (lds_bytes > cur_max) {
        hipError_t err = hipFuncSetAttribute((const void*)gcp2_fwd_kernel<NTG, GT, PWL>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (err != hipSuccess) return (int)err;
        cur_max = lds_bytes;
    }
    hipLaunchKernelGGL((gcp2_fwd_kernel<NTG, GT, PWL>), grid, dim3(GCP_WAVE), lds_bytes, st, p);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

template <int NTG, int GT>
int launch_fwd2(const FwdParams& p, dim3 grid, size_t lds_bytes, hipStream_t st) {
    bool pwl = true;
    for (int k = 0; k < p.n; ++k) pwl = pwl && gcp_is_pwl(p.it[k].act_s) && gcp_is_pwl(p.it[k].act_v);
    if (pwl) return launch_fwd3<NTG, GT, true>(p, grid, lds_bytes, st);
    return launch_fwd3<NTG, GT, false>(p, grid, lds_bytes, st);
}

int launch_fwd(const FwdParams& p, hipStream_t st) {
    const FwdLds L = fwd_lds(p.sh);
    const size_t lds_bytes = (size_t)L.total * sizeof(float);
    if (lds_bytes > 160 * 1024) return GCPNET_E_UNSUPPORTED;
    if (p.sh.GT > 2) return GCPNET_E_UNSUPPORTED;
    const dim3 grid((unsigned)gcp_cdiv(p.rows, GCP_TILE_ROWS));
    const bool g2 = p.sh.GT == 2;
    switch (p.sh.NTG) {
        case 1: return g2 ? launch_fwd2<1, 2>(p, grid, lds_bytes, st) : launch_fwd2<1, 1>(p, grid, lds_bytes, st);
        case 2: return g2 ? launch_fwd2<2, 2>(p, grid, lds_bytes, st) : launch_fwd2<2, 1>(p, grid, lds_bytes, st);
        default: return g2 ? launch_fwd2<4, 2>(p, grid, lds_bytes, st) : launch_fwd2<4, 1>(p, grid, lds_bytes, st);
    }
}

int check_concat(const gcp_concat_t* c, int total) {
    if (!c || c->n < 1 || c->n > GCP_MAX_SEG) return GCPNET_E_BADARG;
    int sum = 0;
    for (int k = 0; k < c->n; ++k) {
        if (!c->ptr[k] || c->dim[k] <= 0) return GCPNET_E_BADARG;
        sum += c->dim[k];
    }
    return sum == total ? 0 : GCPNET_E_BADARG;
}

int check_weights(const gcp2_weights_t* w, const gcp2_opts_t* o, bool need_vout) {
    if (!w || !o || !w->pack || !w->b_scalar) return GCPNET_E_BADARG;
    if (w->vi > 0 && !w->w_down) return GCPNET_E_BADARG;
    if (w->vi > 0 && w->use_frames && !w->w_frames) return GCPNET_E_BADARG;
    if (need_vout && w->vi > 0 && !w->w_up) return GCPNET_E_BADARG;
    if (w->vo > 64) return GCPNET_E_UNSUPPORTED;
    if (o->vmode == GCP_VMODE_SCALAR_GATE && w->vo > 0 && w->vi > 0 && !w->b_gate) return GCPNET_E_BADARG;
    if (o->vector_residual && w->vi != w->vo) return GCPNET_E_BADARG;
    return 0;
}

void fill_item(ChainItem& it, const gcp2_weights_t& w, const gcp2_opts_t& o, float* s_out, float* v_out, float* s_pre,
               float* gate) {
    it.pack = w.pack; it.b_scalar = w.b_scalar; it.w_down = w.w_down; it.w_frames = w.w_frames; it.w_up = w.w_up;
    it.b_gate = w.b_gate; it.s_out = s_out; it.v_out = v_out; it.s_pre = s_pre; it.gate = gate;
    it.act_s = o.act_s; it.act_v = o.act_v;
}

}  // namespace

extern "C" int gcpnet_abi_version(void) { return GCPNET_ABI_VERSION; }
extern "C" int64_t gcpnet_tb_floats(int rows, int width) {
    return (int64_t)gcp_cdiv(rows > 0 ? rows : 0, GCP_TILE_ROWS) * GCP_TILE_ROWS * gcp_round_up(width > 0 ? width : 0, 32);
}
extern "C" int64_t gcpnet_tb_sign_words(int rows, int width) {
    const int nt = gcp_cdiv(width > 0 ? width : 0, 32);
    if (nt & 1) return 0;  // (two 32-wide tiles = 32 register elements per lane = one word)
    return (int64_t)gcp_cdiv(rows > 0 ? rows : 0, GCP_TILE_ROWS) * (nt / 2) * 64;
}
extern "C" int gcpnet_debug_knobs_compiled(void) {
#ifdef GCP_DEBUG_KNOBS
    return 1;
#else
    return 0;
#endif
}

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

extern "C" int gcpnet_pack_gcp2_weights_multi(int n, const gcp2_weights_t* w, float* const* pack_out, void* stream) {
    if (n < 1 || n > GCP_MAX_CHAIN || !w || !pack_out) return GCPNET_E_BADARG;
    PackMultiArgs a;
    for (int k = 0; k < n; ++k) {
        if (!pack_out[k] || !w[k].w_scalar) return GCPNET_E_BADARG;
        if (w[k].si != w[0].si || w[k].vi != w[0].vi || w[k].so != w[0].so || w[k].vo != w[0].vo || w[k].hidden != w[0].hidden ||
            w[k].use_frames != w[0].use_frames)
            return GCPNET_E_BADARG;  // (one shape: the blocks of a chain)
        a.w[k] = w[k];
        a.out[k] = pack_out[k];
    }
    const GcpShape S = gcp_shape(w[0].si, w[0].vi, w[0].so, w[0].vo, w[0].hidden, w[0].use_frames);
    const int threads = 256;
    const int64_t blocks = (S.total + threads - 1) / threads;
    hipLaunchKernelGGL(pack_gcp2_multi_kernel, dim3((unsigned)blocks, n), dim3(threads), 0, (hipStream_t)stream, a, S);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t gcpnet_gcp2_forward_lds_bytes(int si, int vi, int so, int vo, int hidden, int use_frames) {
    return (int64_t)fwd_lds(gcp_shape(si, vi, so, vo, hidden, use_frames)).total * (int64_t)sizeof(float);
}

extern "C" int gcpnet_gcp2_forward(int rows, const gcp_concat_t* s_in, const gcp_concat_t* v_in, const float* frames,
                                   const gcp2_weights_t* w, const gcp2_opts_t* opts, const gcp_concat_t* s_add,
                                   const gcp_concat_t* v_add, const float* res_s, const float* res_v, float* s_out, float* v_out, float* s_pre,
                                   float* gate, void* stream) {
    if (rows < 0 || !s_out) return GCPNET_E_BADARG;
    if (int rc = check_weights(w, opts, w && w->vo > 0)) return rc;
    if (s_add && s_add->n > 0) {
        if (s_add->n > GCP_MAX_SEG) return GCPNET_E_BADARG;
        for (int k = 0; k < s_add->n; ++k)
            if (!s_add->ptr[k] || s_add->dim[k] != w->so) return GCPNET_E_BADARG;
    }
    if (rows == 0) return 0;
    if (check_concat(s_in, w->si)) return GCPNET_E_BADARG;
    if (w->vi > 0) {
        if (check_concat(v_in, w->vi)) return GCPNET_E_BADARG;
        if (w->use_frames && !frames) return GCPNET_E_BADARG;
    }
    if (w->vo > 0 && !v_out) return GCPNET_E_BADARG;

    FwdParams p;
    p.rows = rows;
    p.s_in = *s_in;
    if (w->vi > 0) p.v_in = *v_in; else p.v_in.n = 0;
    p.frames = frames;
    p.o = *opts;
    if (w->vi == 0) p.o.vmode = GCP_VMODE_NONE;  // zero vectors: nothing to gate (gcpnet.py:447-449)
    p.n = 1;
    fill_item(p.it[0], *w, *opts, s_out, v_out, s_pre, gate);
    if (s_add && s_add->n > 0) p.s_add = *s_add; else p.s_add.n = 0;
    p.v_add.n = 0;
    if (v_add && v_add->n > 0) {
        const GcpShape sv = gcp_shape(w->si, w->vi, w->so, w->vo, w->hidden, w->use_frames);
        if (v_add->n > GCP_MAX_SEG) return GCPNET_E_BADARG;
        if (!sv.vmm) return GCPNET_E_UNSUPPORTED;  // the addends enter the MFMA form of the vector prologue
        for (int k = 0; k < v_add->n; ++k)
            if (!v_add->ptr[k] || v_add->dim[k] != gcp_round_up(sv.HF, 4) || (reinterpret_cast<uintptr_t>(v_add->ptr[k]) & 15))
                return GCPNET_E_BADARG;
        p.v_add = *v_add;
    }
    p.res_s = res_s; p.res_v = res_v;
    p.fused_res = (res_s && s_in->n == 1 && !s_in->idx[0] && res_s == s_in->ptr[0] && w->si == w->so &&
                   (w->vo == 0 || (res_v && w->vi == w->vo && v_in->n == 1 && !v_in->idx[0] && res_v == v_in->ptr[0])))
                      ? 1 : 0;
    p.stamps = g_gcp_phase_buf; p.stamp_cap = g_gcp_phase_cap;
    p.sh = gcp_shape(w->si, w->vi, w->so, w->vo, w->hidden, w->use_frames);
    return launch_fwd(p, (hipStream_t)stream);
}

extern "C" int gcpnet_gcp2_chain_forward(int rows, const float* s0, const float* v0, const float* frames, int n,
                                         const gcp2_chain_item_t* items, void* stream) {
    if (rows < 0 || n < 1 || n > GCP_MAX_CHAIN || !items || !s0) return GCPNET_E_BADARG;
    const gcp2_weights_t& w0 = items[0].w;
    for (int k = 0; k < n; ++k) {
        const gcp2_weights_t& w = items[k].w;
        if (int rc = check_weights(&w, &items[k].o, w.vo > 0)) return rc;
        if (w.si != w0.si || w.vi != w0.vi || w.so != w0.so || w.vo != w0.vo || w.hidden != w0.hidden ||
            w.use_frames != w0.use_frames || w.si != w.so || w.vi != w.vo)
            return GCPNET_E_BADARG;
        if (items[k].o.vmode != items[0].o.vmode || items[k].o.vector_residual != items[0].o.vector_residual ||
            items[k].o.e3 != items[0].o.e3 || items[k].o.slope != items[0].o.slope)
            return GCPNET_E_BADARG;
        if (k == n - 1 && (!items[k].s_out || (w.vo > 0 && !items[k].v_out))) return GCPNET_E_BADARG;
    }
    if (w0.vi > 0 && (!v0 || (w0.use_frames && !frames))) return GCPNET_E_BADARG;
    if (rows == 0) return 0;
    {  // preferred: the register-resident kernel (gcp2_chain_fwd.hip)
        const int rc = gcp2_chain_fwd_registers(rows, s0, v0, frames, n, items, (hipStream_t)stream);
        if (rc != GCPNET_E_UNSUPPORTED) return rc;
    }
    for (int k = 0; k < n; ++k)  // (the tile-blocked outputs exist in the register-resident kernel only)
        if (items[k].s_out_tb || items[k].s_pre_tb || items[k].s_sign) return GCPNET_E_UNSUPPORTED;
    FwdParams p;
    p.rows = rows;
    p.s_in.n = 1; p.s_in.ptr[0] = s0; p.s_in.idx[0] = nullptr; p.s_in.dim[0] = w0.si;
    p.v_in.n = w0.vi > 0 ? 1 : 0; p.v_in.ptr[0] = v0; p.v_in.idx[0] = nullptr; p.v_in.dim[0] = w0.vi;
    p.frames = frames;
    p.o = items[0].o;
    if (w0.vi == 0) p.o.vmode = GCP_VMODE_NONE;
    p.n = n;
    for (int k = 0; k < n; ++k) fill_item(p.it[k], items[k].w, items[k].o, items[k].s_out, items[k].v_out, items[k].s_pre, items[k].gate);
    p.s_add.n = 0;
    p.v_add.n = 0;
    p.res_s = nullptr; p.res_v = nullptr;
    p.fused_res = 1;
    p.stamps = g_gcp_phase_buf; p.stamp_cap = g_gcp_phase_cap;
    p.sh = gcp_shape(w0.si, w0.vi, w0.so, w0.vo, w0.hidden, w0.use_frames);
    if (p.sh.NG != 1) return GCPNET_E_UNSUPPORTED;  // the state must fit one accumulator group (so <= 128)
    return launch_fwd(p, (hipStream_t)stream);
}
