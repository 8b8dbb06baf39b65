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
#include "common.h"
#include "tile_io.h"
#include "vec_mfma.h"

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

struct ChainLds {
    int VS, XS;
    int o_vt, o_fr, o_ext, o_ust, o_stage, total;
};

__host__ __device__ inline ChainLds chain_lds(const GcpShape& s) {
    ChainLds l;
    l.VS = gcp_odd(3 * s.vi);
    l.XS = gcp_odd(gcp_round_up(s.H + s.nf, 2));
    l.o_vt = 0;
    l.o_fr = l.o_vt + 32 * l.VS;
    l.o_ext = l.o_fr + 32 * 9;
    l.o_ust = l.o_ext + 32 * l.XS;
    l.o_stage = l.o_ust + s.SVB * 3 * 64;  // (ust: vector_down outputs parked between the vector prologue and epilogue)
    l.total = l.o_stage + GCP_ACC_STAGE_FLOATS;
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

// NT = 32-wide tiles of the scalar state (so <= 32 * NT); PWL as in gcp2_fwd.hip.
template <int NT, bool PWL>
__global__ __launch_bounds__(GCP_WAVE, 2) void gcp2_chain_fwd_kernel(ChainParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const GcpShape& S = p.sh;
    const ChainLds L = chain_lds(S);
    int lane = threadIdx.x;
    int e = lane & 31, hi = lane >> 5;
    const int r0 = blockIdx.x * GCP_TILE_ROWS;
    const int rows = p.rows;
    int row = r0 + e;
    bool row_ok = row < rows;
    float* vt = lds + L.o_vt;
    float* fr = lds + L.o_fr;
    float* ext = lds + L.o_ext;
    float* ust = lds + L.o_ust;
    float* stage = lds + L.o_stage;
    const int si = S.si, vi = S.vi, so = S.so, vo = S.vo, H = S.H;
    const int NX = gcp_round_up(H + S.nf, 2) / 2;  // k-pair steps over the norms / frame scalars
    const float slope = p.o.slope;
    const bool scalar_gate = p.o.vmode == GCP_VMODE_SCALAR_GATE;
    const bool vec_so = (so & 3) == 0;
    // ---- the tile: vectors + frames into LDS, scalars straight into the accumulator layout ---------------------------
    f32x16 xs[NT];
    {
        GcpSegBuf<8> vb0;
        gcp_seg_issue(vb0, p.v0, nullptr, 3 * vi, r0, rows, vt, L.VS, 0, lane);
        if (S.nf) gcp_load_frames(p.frames, r0, rows, fr, lane);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = gcp_load4(p.s0, row, so, 32 * t + 8 * q + 4 * hi, row_ok, vec_so);
                xs[t][4 * q] = v.x; xs[t][4 * q + 1] = v.y; xs[t][4 * q + 2] = v.z; xs[t][4 * q + 3] = v.w;
            }
        gcp_seg_commit(vb0, vt, L.VS, 0);
    }
    for (int i = H + S.nf + hi; i < 2 * NX; i += 2) ext[e * L.XS + i] = 0.f;

    for (int ci = 0; ci < p.n; ++ci) {
        asm volatile("" : "+v"(lane), "+v"(e), "+v"(hi));  // keep per-lane addresses from being hoisted and spilled
        row = r0 + e;
        row_ok = row < rows;
        const ChainItemF& it = p.it[ci];
        const float ns_s = gcp_neg_slope(it.act_s, slope), ns_v = gcp_neg_slope(it.act_v, slope);
        if (ci == p.n - 1) gcp_stamp(p.stamps, p.stamp_cap, 0, lane);
        gcp_wave_lds_sync();  // the previous block's vector tile update

        // ---- vector prologue on the matrix cores: [vh | vf] = [vector_down ; vector_down_frames] v, then (element-wise,
        //      in registers) the norms of vh and the projections of vf onto the row's frame -> the 32 x XS extras tile ------
        {
            gcp_xyz_acc u;
            gcp_vmm_down<16>(it.pack + S.offVA + lane, S.SVA, vi, vt + e * L.VS, hi, u);
            if (ci == p.n - 1) gcp_stamp(p.stamps, p.stamp_cap, 1, lane);
            float f[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) f[i] = S.nf ? fr[e * 9 + i] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int x = gcp_crow(r, hi);
                const float u0 = u[0][r], u1 = u[1][r], u2 = u[2][r];
                if (r < S.SVB) {  // parked for vector_up in the epilogue (wave-uniform guard)
                    ust[(r * 3 + 0) * 64 + lane] = u0; ust[(r * 3 + 1) * 64 + lane] = u1; ust[(r * 3 + 2) * 64 + lane] = u2;
                }
                if (x < H) {
                    ext[e * L.XS + x] = sqrtf(u0 * u0 + u1 * u1 + u2 * u2 + 1e-8f) + 1e-8f;
                } else if (x < S.HF) {
                    const int k = x - H;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        float pr = f[3 * a + 0] * u0 + f[3 * a + 1] * u1 + f[3 * a + 2] * u2;
                        if (p.o.e3 && a == 1) pr = fabsf(pr);
                        ext[e * L.XS + H + 3 * k + a] = pr;
                    }
                }
            }
        }
        gcp_wave_lds_sync();

        if (ci == p.n - 1) gcp_stamp(p.stamps, p.stamp_cap, 2, lane);
        // ---- scalar_out: acc = b + W[:, state] x^T + W[:, extras] ext^T --------------------------------------------------
        f32x16 acc[NT];
        f32x16 gacc;     // vector-gate pre-activations (initialised with the bias while the extras tile is multiplied)
        float gwa[16];   // first batch of the gate GEMM's weight fragments
#pragma unroll
        for (int r = 0; r < 16; ++r) { gacc[r] = 0.f; gwa[r] = 0.f; }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = 32 * t + gcp_crow(r, hi);
                acc[t][r] = j < so ? it.b_scalar[j] : 0.f;
            }
        {
            const float* wf = it.pack + S.offF + (int64_t)lane * NT;  // [step][64][NT]
            constexpr int U = 4;
            WF<NT> A0[U], A1[U], A2[U];
            auto ld = [&](WF<NT>(&a)[U], int st0) {
#pragma unroll
                for (int u = 0; u < U; ++u) a[u].load(wf + (int64_t)min(st0 + u, NT * 16 - 1) * 64 * NT);
            };
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
            }
            if (ci == p.n - 1) gcp_stamp(p.stamps, p.stamp_cap, 3, lane);
            // (the gate GEMM's first weight fragments and its bias are requested here, one phase ahead)
            if (scalar_gate) {
                const float* wg0 = it.pack + S.offC + lane;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    gwa[r] = wg0[(int64_t)r * 64];
                    gacc[r] = gcp_crow(r, hi) < vo ? it.b_gate[min(gcp_crow(r, hi), vo - 1)] : 0.f;
                }
            }
            // norms and frame scalars: ordinary B fragments from the 32 x 16 LDS tile, weights from section A
            const float* wa = it.pack + S.offA + (int64_t)lane * NT;
            const int kk0 = si / 2;
            for (int x0 = 0; x0 < NX; x0 += 8) {  // fragments of eight steps requested together (one L2 round trip, not eight)
                WF<NT> a[8];
                float bv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int x = min(x0 + u, NX - 1);
                    a[u].load(wa + (int64_t)(kk0 + x) * 64 * NT);
                    bv[u] = ext[e * L.XS + 2 * x + hi];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (x0 + u < NX) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].v[t], bv[u], acc[t], 0, 0, 0);
                    }
            }
        }

        if (ci == p.n - 1) gcp_stamp(p.stamps, p.stamp_cap, 4, lane);
        // ---- vector gate Linear, B fragments = the accumulator registers ---------------------------------------------------
        if (scalar_gate) {
            const float* wg = it.pack + S.offC + lane;
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
        if (ci == p.n - 1) gcp_stamp(p.stamps, p.stamp_cap, 5, lane);
        // ---- s_pre (saved for the backward) and the new state x += act(s_pre), both straight from / in registers ----------
        if (it.s_pre) gcp_store_acc_rows<NT>(it.s_pre, so, 0, so, r0, rows, acc, stage, lane);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) xs[t][r] += gcp_actf<PWL>(it.act_s, ns_s, slope, acc[t][r]);
        if (it.s_out) gcp_store_acc_rows<NT>(it.s_out, so, 0, so, r0, rows, xs, stage, lane);

        if (ci == p.n - 1) gcp_stamp(p.stamps, p.stamp_cap, 6, lane);
        // ---- vector epilogue: vector_up on the matrix cores (B fragments = the parked vector_down outputs), then sigmoid
        //      gate, gating and residual element-wise in registers; the vector tile is updated in place ------------------
        {
            gcp_xyz_acc uin, vu;
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int d = 0; d < 3; ++d) uin[d][r] = r < S.SVB ? ust[(r * 3 + d) * 64 + lane] : 0.f;
            gcp_xyz_zero(vu);
            gcp_vmm_regs<16>(it.pack + S.offVB + lane, S.SVB, uin, vu);
            float sg[16], x[16][3];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = min(gcp_crow(r, hi), vo - 1);
                sg[r] = scalar_gate ? gcp_sigmoid(gacc[r]) : 1.f;
#pragma unroll
                for (int d = 0; d < 3; ++d) x[r][d] = vt[e * L.VS + 3 * o + d];
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
                    vt[e * L.VS + 3 * o + 0] = x[r][0] + u0 * sc;
                    vt[e * L.VS + 3 * o + 1] = x[r][1] + u1 * sc;
                    vt[e * L.VS + 3 * o + 2] = x[r][2] + u2 * sc;
                }
            }
            if (scalar_gate && it.gate) {  // saved for the backward, straight from the registers (accumulator-layout rows)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (8 * q < vo)
                        gcp_store4(it.gate, row, vo, 8 * q + 4 * hi, make_float4(sg[4 * q], sg[4 * q + 1], sg[4 * q + 2], sg[4 * q + 3]),
                                   row_ok, (vo & 3) == 0);
            }
        }
        gcp_wave_lds_sync();
        if (it.v_out) gcp_store_tile(it.v_out, 3 * vo, 0, 3 * vo, r0, rows, vt, L.VS, lane);
        if (ci == p.n - 1) gcp_stamp(p.stamps, p.stamp_cap, 7, lane);
    }
}

template <int NT, bool PWL>
int launch_chain(const ChainParams& p, size_t lds_bytes, hipStream_t st) {
    hipLaunchKernelGGL((gcp2_chain_fwd_kernel<NT, PWL>), dim3((unsigned)gcp_cdiv(p.rows, GCP_TILE_ROWS)), dim3(GCP_WAVE),
                       lds_bytes, st, p);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// Returns GCPNET_E_UNSUPPORTED when the shape does not fit the register-resident kernel; the caller then uses the
// LDS-resident chain of gcp2_fwd.hip.
int gcp2_chain_fwd_registers(int rows, const float* s0, const float* v0, const float* frames, int n,
                             const gcp2_chain_item_t* items, hipStream_t st) {
    const gcp2_weights_t& w0 = items[0].w;
    const GcpShape S = gcp_shape(w0.si, w0.vi, w0.so, w0.vo, w0.hidden, w0.use_frames);
    if (S.NG != 1 || S.NTG < 2 || (w0.si & 1) || w0.vi <= 0 || w0.vo <= 0 || w0.vo > 32 || S.GT != 1) return GCPNET_E_UNSUPPORTED;
    if (S.NTS != S.NTG || !S.vmm || w0.vi > 32) return GCPNET_E_UNSUPPORTED;
    ChainParams p;
    p.rows = rows; p.s0 = s0; p.v0 = v0; p.frames = frames;
    p.o = items[0].o;
    p.n = n;
    bool pwl = true;
    for (int k = 0; k < n; ++k) {
        const gcp2_chain_item_t& c = items[k];
        ChainItemF& it = p.it[k];
        it.pack = c.w.pack; it.b_scalar = c.w.b_scalar;
        it.b_gate = c.w.b_gate; it.s_out = c.s_out; it.v_out = c.v_out; it.s_pre = c.s_pre;
        it.gate = c.gate; it.act_s = c.o.act_s; it.act_v = c.o.act_v;
        pwl = pwl && gcp_is_pwl(c.o.act_s) && gcp_is_pwl(c.o.act_v);
    }
    p.stamps = g_gcp_phase_buf; p.stamp_cap = g_gcp_phase_cap;
    p.sh = S;
    const size_t lds_bytes = (size_t)chain_lds(S).total * sizeof(float);
    if (lds_bytes > 64 * 1024) return GCPNET_E_UNSUPPORTED;
    if (S.NTG == 2) return pwl ? launch_chain<2, true>(p, lds_bytes, st) : launch_chain<2, false>(p, lds_bytes, st);
    return pwl ? launch_chain<4, true>(p, lds_bytes, st) : launch_chain<4, false>(p, lds_bytes, st);
}
