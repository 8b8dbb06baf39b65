// GCP2 backward for a 32-row tile per WORKGROUP (NW wavefronts), the adjoint of gcp_wg_fwd.hip's block: given d(s_out),
// d(v_out) and the saved s_pre / gate it writes d(s_in), d(v_in) and the weight gradients (autograd through GCP2.forward,
// reference src/models/components/gcpnet.py:394-468; with `residual` the block was x + GCP(x), gcpnet.py:921-924).
//
// Persistent workgroups loop over the tiles.  Phases per tile (barriers between them):
//   load   v_in, d(v_out), gate, frames (and s_in when the weight gradient is fused) -> LDS; s_pre / d(s_out) slices requested
//   P1     (VALU, thread = (row, hidden channel)) recompute [vh | vf], norms, frame scalars
//   P2     (VALU, thread = (row, output channel)) adjoint of the vector epilogue: d(vector_up out), d(gate pre-activation)
//   P3     (MFMA, wave = its 32-column tiles of so) gate adjoint Wg^T dgate, ds_pre -> LDS tile DS
//   P4     (MFMA, wave = 32-column tiles of the merged axis K) d[s | norms | frame scalars] = W^T ds_pre; a lone last tile is
//          reduced split-K across the waves
//   P5/P6  (MFMA, fused mode) dW[so, K + 1] += ds_pre^T [s | norms | frame scalars | 1], dWg[vo, so] += dgate^T act_v(s_pre):
//          the reduction runs over the tile's rows; the accumulators stay in registers across ALL tiles of the workgroup and
//          leave once, as per-workgroup partial sums (summed in a fixed order by wg_reduce_kernel: deterministic, no atomics)
//   P7/P8  (VALU) adjoint of the vector prologue: d vh, d vf, d(v_in)
//   P9     (VALU) per-thread partial sums of the small vector weight gradients
// Without the fusion (node rows: few tiles, wide blocks) ds_pre / norms / dgate go to HBM for gcpnet_tn_gemm instead.
#include <cstdlib>

#include "gcp_wg.h"

// GCP_WG_X: measurement builds whose RESULTS ARE WRONG (tools/wg_variants.sh): bits remove one cost each so that its share of the
// launch time can be read under real contention.  1: P4's fragments from 6 KB (= GCP_WG_EXP1); 2: no P7; 4: no P8; 8: no P9;
// 16: no P1; 32: no P2; 64: no gate MFMAs in P3; 128: no ds_pre store; 256: no P4 MFMAs; 512: no s_pre / d(s_out) requests of the later
// output tiles in P3
#ifndef GCP_WG_X
#define GCP_WG_X 0
#endif
#if (GCP_WG_X & 1) && !defined(GCP_WG_EXP1)
#define GCP_WG_EXP1
#endif

namespace {

// Everything about a block's shape that the kernel needs as integers: dimensions, tile counts, LDS strides and offsets.  One
// constexpr function computes it for the host (which fills WgBwdParams from it) and for the compile-time-shape instantiations
// of the kernel (SHP below), where all of it folds into immediates, loops unroll and the uniform branches disappear.
#define WG_BWD_DIMS(X)                                                                                                      \
    X(si) X(vi) X(so) X(vo) X(H) X(nf) X(K) X(NT) X(NKT) X(SG) X(VG) X(HF) X(split) X(LW) X(KW) X(NNT) X(EP) X(VOP) X(HFP)  \
    X(KS) X(DSS) X(VS) X(US) X(HS) X(FS) X(DGS) X(EXS) X(EPS) X(WSV) X(WSU) X(WTV) X(WTU)                                   \
    X(o_x) X(o_ds) X(o_v) X(o_dvo) X(o_dvu) X(o_vh) X(o_dvhf) X(o_fr) X(o_dg) X(o_rn) X(o_sgn) X(o_dext) X(o_epart) X(o_ws) \
    X(n_up) X(n_sm) X(sm_tiles) X(sm_up_tiles) X(sm_nu) X(sm_nd) X(npass)
#define WG_BWD_MAGICS(X) X(mg_v) X(mg_o) X(mg_g) X(mg_x) X(mg_xpad) X(mg_epad) X(mg_ns) X(mg_wdt) X(mg_hfp) X(mg_ep) X(mg_vop)

struct WgBwdDims {
#define X(f) int f;
    WG_BWD_DIMS(X)
#undef X
#define X(f) unsigned f;
    WG_BWD_MAGICS(X)
#undef X
    int KTn, fused, lds_floats;
};

constexpr int c_cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int c_rup(int x, int m) { return (x + m - 1) / m * m; }
constexpr int c_max(int a, int b) { return a > b ? a : b; }
constexpr int c_min(int a, int b) { return a < b ? a : b; }
constexpr int c_stride(int width) { return 4 * (c_cdiv(width, 4) | 1); }  // == wg_stride
constexpr unsigned c_magic(int d) { return d > 1 ? (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d) : 0u; }  // == wg_magic

// (si, vi, so, vo, hidden: as in gcp2_weights_t; gated: scalar gate in use; want_fused: the caller asked for fused weight
// gradients; NW: waves per workgroup)
// npass: the ds_pre tile holds the so columns of ONE pass (1 / npass of the wave's output tiles), P3 -> P4 run npass times with
// the W^T ds_pre accumulators kept across the passes (plain fp32 form only): the (s,V) -> (4s,2V) feed-forward GCP of BASELINE
// configs[4] has so = 1024, its full ds_pre tile would be 131 KB
constexpr WgBwdDims wg_bwd_dims(int si, int vi, int so, int vo, int hidden, int use_frames, int gated, int want_fused, int NW,
                                int b6 = 0, int npass = 1) {
    WgBwdDims d{};
    d.npass = npass;
    d.si = si; d.vi = vi; d.so = so; d.vo = vo;
    d.H = vi > 0 ? hidden : 0;
    d.nf = (vi > 0 && use_frames) ? 9 : 0;
    d.K = si + d.H + d.nf;
    d.NT = c_cdiv(so, 32);
    d.NKT = c_cdiv(d.K, 32);
    d.SG = 4 * d.NT;
    d.VG = c_cdiv(vo, 8);
    d.HF = d.H + (d.nf ? 3 : 0);
    const int rem = d.NKT % NW;
    d.split = (rem == 1 && d.NKT > 1) ? 1 : 0;
    const int NFT = d.split ? d.NKT - 1 : d.NKT;
    d.KTn = c_cdiv(NFT, NW);
    d.LW = d.K - 32 * (d.NKT - 1);
    d.KW = d.K + 1;
    d.NNT = c_cdiv(d.KW, 32);
#ifdef GCP_WG_FUSE_FENCE  // (the round-2 envelope: vo <= 16, si % 4 == 0; kept as a debugging switch)
    d.fused = (want_fused && d.NT <= NW && d.KTn == 1 && d.NNT <= 5 && vo <= 16 && (si & 3) == 0) ? 1 : 0;
#else
    // (the fused gate weight gradient holds ONE 32-row tile of gate outputs per wave: gated blocks with vo > 32 take the plain form)
    d.fused = (want_fused && d.NT <= NW && d.KTn == 1 && d.NNT <= 5 && !(gated && vo > 32)) ? 1 : 0;
#endif
    d.EP = c_rup(d.H + d.nf, 4);
    d.VOP = c_rup(vo, 4);
    d.HFP = c_rup(d.HF, 4);
    d.n_up = vo * d.H;
    d.n_sm = vo * d.H + d.HF * vi;
    d.sm_nu = c_max(c_cdiv(d.H, 16), 1);
    d.sm_nd = c_cdiv(vi, 16);
    d.sm_up_tiles = vo > 0 ? c_cdiv(vo, 16) * d.sm_nu : 0;
    d.sm_tiles = d.sm_up_tiles + c_cdiv(d.HF, 16) * d.sm_nd;
    const int xw = d.fused ? 8 * c_cdiv(d.KW, 8) : 0;
    d.KS = c_stride(c_max(xw, 4));
    d.DSS = c_stride(32 * c_cdiv(d.NT, npass));
    d.VS = c_stride(3 * vi); d.US = c_stride(3 * c_max(vo, 1)); d.HS = c_stride(3 * c_max(d.H, 1)); d.FS = c_stride(3 * d.HF);
    d.DGS = c_stride(c_rup(c_max(vo, 1), 8));
    d.EXS = c_stride(c_max(d.EP, d.K - si));
    d.EPS = c_max(20, c_stride(c_rup(d.LW, 8)));
    d.WSV = c_stride(vi); d.WTV = c_stride(d.HF); d.WSU = c_stride(c_max(d.H, 1)); d.WTU = c_stride(c_max(vo, 1));
    int off = 0;
    d.o_x = off; off += d.fused ? 32 * d.KS : 0;
    // (b6: the ds_pre tile as three bf16 planes in MFMA operand order, [2 NT slabs][term][64 lanes][16 bytes], instead of fp32 rows)
    d.o_ds = off; off += b6 ? c_max(32 * d.DSS, 2 * d.NT * 768) : 32 * d.DSS;
    d.o_epart = off; off += NW * 32 * d.EPS;
    d.o_v = off; off += 32 * d.VS;
    d.o_dvu = off; off += 32 * d.US;
    d.o_vh = off; off += 32 * d.HS;
    d.o_fr = off; off += 32 * 9;
    d.o_dg = off; off += 32 * d.DGS;
    d.o_rn = off; off += c_rup(32 * (d.H | 1), 4);
    d.o_sgn = off; off += 32 * 3 + 32;
    d.o_ws = off; off += d.HF * d.WSV + vi * d.WTV + vo * d.WSU + d.H * d.WTU;
    off = c_rup(off, 4);
    // d(v_out) is dead after P2; its space then holds d[vh | vf] and (not fused: from P1 on) the extras tile.  The extras tile
    // must not overlap d(v_out) when it is written in P1, so it sits behind it in that case.
    d.o_dvo = off;
    const int dvo = 32 * d.US, dq = 32 * d.FS, dx = 32 * d.EXS;
    d.o_dvhf = off;
    if (d.fused) { d.o_dext = off + dq; off += c_max(dvo, dq + dx); }
    else { d.o_dext = off + c_max(dvo, dq); off += c_max(dvo, dq) + dx; }
    d.lds_floats = off;
    // (the fused form adds the 32 x (K + 1) input tile X: when that pushes the tile set past the 160 KB of a CU -- the second
    // feed-forward GCP of BASELINE configs[4] after its leading columns were split off, (128,64) -> (256,32) -- take the plain form)
    if (d.fused && (long long)off * 4 > 160 * 1024) return wg_bwd_dims(si, vi, so, vo, hidden, use_frames, gated, 0, NW, b6, npass);
    // plain fp32 form that does not fit: two, then four passes over the output tiles (whole tiles per wave and pass)
    if (!d.fused && !b6 && d.KTn == 1 && (long long)off * 4 > 160 * 1024 && npass < 4 && c_cdiv(d.NT, NW) % (2 * npass) == 0 &&
        d.NT % (NW * 2 * npass) == 0)
        return wg_bwd_dims(si, vi, so, vo, hidden, use_frames, gated, 0, NW, 0, 2 * npass);
    d.mg_v = c_magic(3 * vi / 4); d.mg_o = c_magic(3 * vo / 4); d.mg_g = c_magic(vo / 4); d.mg_x = c_magic(si / 4);
    d.mg_xpad = c_magic(8 * c_cdiv(d.KW, 8) - d.K); d.mg_epad = c_magic(d.EP - (d.H + d.nf));
    d.mg_ns = c_magic(c_min(si, d.K) - 32 * (d.NKT - 1)); d.mg_wdt = c_magic(3 * d.HFP); d.mg_hfp = c_magic(d.HFP);
    d.mg_ep = c_magic(d.EP / 4); d.mg_vop = c_magic(d.VOP / 4);
    return d;
}

// Compile-time shapes (template parameter SHP of the kernel): the residual message GCPs of the shipped configurations,
// (s, V) -> (s, V) with frames and a scalar gate.  0 = run-time shape.
template <int SHP> struct WgBwdShape { static constexpr int S = 0, V = 0, HID = 0; };
template <> struct WgBwdShape<1> { static constexpr int S = 128, V = 16, HID = 4; };  // BASELINE configs[1]: (128, 16), bottleneck 4
template <> struct WgBwdShape<2> { static constexpr int S = 256, V = 32, HID = 8; };  // BASELINE configs[4]: (256, 32)

struct WgBwdParams {
    int rows, ntiles;
    const float* s_in;
    const float* v_in;
    const float* frames;
    const float* s_pre;
    const float* gate;
    const float* d_s_out;
    const float* d_v_out;
    gcp_concat_t v_add;
    float* d_s_in;
    float* d_v_in;
    const float* pk;
    int64_t offA2, offG2, offA2b;
    const float* w_down;
    const float* w_frames;
    const float* w_up;
    float* ds_pre;
    float* dvhf;
    float* ext;
    float* dgate;
    float* dw_part;
    float* dwg_part;
    float* wsm_part;
    int si, vi, so, vo, H, nf, K, NT, NKT, SG, VG, HF;
    int act_s, act_v, vmode, vres, e3, residual;
    float slope;
    int split, LW;
    int KW, NNT;
    int EP, VOP, HFP;
    int KS, DSS, VS, US, HS, FS, DGS, EXS, EPS, WSV, WSU, WTV, WTU;
    int o_x, o_ds, o_v, o_dvo, o_dvu, o_vh, o_dvhf, o_fr, o_dg, o_rn, o_sgn, o_dext, o_epart, o_ws;
    int n_up, n_sm;  // small weight gradients: vector_up entries, all entries
    unsigned mg_v, mg_o, mg_g, mg_x, mg_xpad, mg_epad, mg_ns, mg_wdt, mg_hfp, mg_ep, mg_vop;  // wg_magic of the tile-copy divisors
    int sm_tiles, sm_up_tiles, sm_nu, sm_nd;  // their 16 x 16 tiles: all, those of vector_up, tiles along N (up / down)
    int npass;                                // passes of P3 -> P4 over the output tiles (MP instantiations)
    int tb;  // tile-blocked layouts (include/gcpnet_hip.h, gcp_wg_bwd_args_t.tb): bit 0 s_pre, 1 d_s_out, 2 d_s_in, 3 ds_pre
    unsigned long long* stamps;  // profiling hook: s_memtime stamps of wave 0 at the phase boundaries (last tile of the workgroup)
    long long stamp_cap;
};

constexpr int NSW = 2;  // 16 x 16 tiles of the small vector weight gradients per wave

// NW waves; KT = full K tiles per wave in P4; FN = 32-wide tiles of the fused weight gradient's K + 1 columns (0 = not fused);
// SHP: compile-time shape (WgBwdShape), 0 = every dimension from the parameters
// B6 (not fused, KT == 1): P4 on the bf16 matrix pipe -- ds_pre is split into three bf16 terms where it is produced (P3, in
// registers) and kept in LDS as operand-ordered planes, the weights come pre-split from section A2b; six products, fp32
// accumulation (gcp_bf16x3.h: exact to fp32 round-off)
// (fused with at most two tiles of K + 1 columns -- the first message GCP after project-then-gather: K + 1 = 51 -- carries 32
// instead of 80 persistent accumulator registers: a third workgroup per CU hides more of the per-tile phase latencies)
// MP: P3 -> P4 in `npass` passes over the output tiles with a ds_pre tile of one pass's columns (plain fp32 form, run-time shapes)
template <int NW, int KT, int FN, bool PWL, int SHP, bool B6 = false, bool MP = false>
__global__ __launch_bounds__(64 * NW, (FN > 0 && FN <= 2 && NW == 4) ? 3 : 2) void gcp_wg_bwd_kernel(const WgBwdParams p_kernarg) {
    static_assert(!B6 || (KT == 1 && FN == 0), "bf16 form: plain mode, one K tile per wave");
    static_assert(!MP || (FN == 0 && !B6 && SHP == 0), "passes over so: plain fp32 form, run-time shapes");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NTH = 64 * NW, TPR = NTH / 32;
    constexpr bool FUSED = FN > 0;
    constexpr int FNR = FUSED ? FN : 1;
    // The parameters are read through the kernarg segment pointer, and that pointer is laundered at every phase boundary: the
    // ~80 uniform values (and everything uniform derived from them) are then re-loaded (s_load, scalar cache) by the phase
    // that needs them instead of staying live in SGPRs across the whole persistent tile loop -- where they do not fit:
    // hipcc spilled ~360 of them into VGPR lanes, ~1 100 v_readlane_b32 per tile and wave.
    typedef const __attribute__((address_space(4))) WgBwdParams* Karg;
    Karg kp = (Karg)__builtin_amdgcn_kernarg_segment_ptr();
#define p (*kp)
    int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lane = tid & 63, e = lane & 31, hi = lane >> 5;
    int prow = tid / TPR, psub = tid - prow * TPR;
    constexpr WgBwdDims CD = wg_bwd_dims(WgBwdShape<SHP>::S, WgBwdShape<SHP>::V, WgBwdShape<SHP>::S, WgBwdShape<SHP>::V,
                                         WgBwdShape<SHP>::HID, 1, 1, FN > 0, NW, B6);
#define DM(f) (SHP ? CD.f : p.f)  // a shape value: immediate for the compile-time shapes
    int rows, KS, DSS, VS, US, HS, FS, DGS, EXS, EPS, si, vi, so, vo, H, nf, K, HF, NT, SG, exs;
    int NKT, VG, split, LW, KW, NNT, EP, VOP, HFP, WSV, WSU, WTV, WTU, n_up, n_sm, sm_tiles, sm_up_tiles, sm_nu, sm_nd, npass;
    float *X, *DS, *V, *DVO, *DVU, *VH, *DVHF, *FR, *DG, *RN, *SGN, *DEXT, *EPART, *ST, *extp;
    const float *WD, *WDT, *WU, *WUT;
#define WG_RELOAD()                                                                                                        \
    do {                                                                                                                   \
        rows = p.rows;                                                                                                     \
        KS = DM(KS); DSS = DM(DSS); VS = DM(VS); US = DM(US); HS = DM(HS); FS = DM(FS); DGS = DM(DGS); EXS = DM(EXS);      \
        EPS = DM(EPS); si = DM(si); vi = DM(vi); so = DM(so); vo = DM(vo); H = DM(H); nf = DM(nf); K = DM(K); HF = DM(HF); \
        NT = DM(NT); SG = DM(SG); NKT = DM(NKT); VG = DM(VG); split = DM(split); LW = DM(LW); KW = DM(KW); NNT = DM(NNT);  \
        EP = DM(EP); VOP = DM(VOP); HFP = DM(HFP); WSV = DM(WSV); WSU = DM(WSU); WTV = DM(WTV); WTU = DM(WTU);             \
        n_up = DM(n_up); n_sm = DM(n_sm); sm_tiles = DM(sm_tiles); sm_up_tiles = DM(sm_up_tiles); sm_nu = DM(sm_nu);       \
        sm_nd = DM(sm_nd); npass = DM(npass);                                                                              \
        X = lds + DM(o_x); DS = lds + DM(o_ds); V = lds + DM(o_v); DVO = lds + DM(o_dvo); DVU = lds + DM(o_dvu);           \
        VH = lds + DM(o_vh); DVHF = lds + DM(o_dvhf); FR = lds + DM(o_fr); DG = lds + DM(o_dg); RN = lds + DM(o_rn);       \
        SGN = lds + DM(o_sgn); DEXT = lds + DM(o_dext); EPART = lds + DM(o_epart);                                         \
        ST = EPART + w * 32 * EPS; /* wave-private staging (half tiles, 32 x 20) shares the wave's split-K partial slot */ \
        WD = lds + DM(o_ws);       /* [HF][WSV]   [vector_down ; vector_down_frames] */                                    \
        WDT = WD + HF * WSV;       /* [vi][WTV]   transposed */                                                            \
        WU = WDT + vi * WTV;       /* [vo][WSU]   vector_up */                                                             \
        WUT = WU + vo * WSU;       /* [H][WTU]    transposed */                                                            \
        /* where P1 leaves [norms | frame scalars]: X's extras columns, or (not fused) a tile of its own that goes to HBM  \
           (it doubles as the d(extras) tile later) */                                                                     \
        extp = FUSED ? X + si : DEXT;                                                                                      \
        exs = FUSED ? KS : EXS;                                                                                            \
    } while (0)
    // Per-lane addresses are loop-invariant in the persistent tile loop too: hipcc hoists them all out of it and spills
    // them.  Laundering the lane indices at every phase boundary makes each phase recompute the few it needs.
#define WG_LAUNDER()                                                                                                       \
    do {                                                                                                                   \
        asm volatile("" : "+v"(tid), "+v"(lane), "+v"(e), "+v"(hi), "+v"(prow), "+v"(psub), "+s"(kp));                      \
        WG_RELOAD();                                                                                                       \
    } while (0)
    WG_RELOAD();
    const bool gated = p.vmode == GCP_VMODE_SCALAR_GATE && vo > 0;
    const float slope = p.slope;
    const float ns_s = gcp_neg_slope(p.act_s, slope), ns_v = gcp_neg_slope(p.act_v, slope);

    // ---- once per workgroup: small weights (both orientations) -> LDS, zero the never-written paddings ------------------
    {
        float* wd = lds + DM(o_ws);
        float* wdt = wd + HF * WSV;
        float* wu = wdt + vi * WTV;
        float* wut = wu + vo * WSU;
        for (int i = tid; i < HF * vi; i += NTH) {
            const int x = i / vi, c = i - x * vi;
            const float v = x < H ? p.w_down[i] : p.w_frames[i - H * vi];
            wd[x * WSV + c] = v;
            wdt[c * WTV + x] = v;
        }
        for (int i = tid; i < vo * H; i += NTH) {
            const int o = i / H, h = i - o * H;
            const float v = p.w_up[i];
            wu[o * WSU + h] = v;
            wut[h * WTU + o] = v;
        }
        for (int i = tid; i < 32 * DGS; i += NTH) DG[i] = 0.f;
        if constexpr (!B6)
            for (int i = tid; i < 32 * DSS; i += NTH) DS[i] = 0.f;
    }
    // persistent accumulators
    f32x16 dW[FNR], dWg;
#pragma unroll
    for (int n = 0; n < FNR; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) dW[n][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) dWg[r] = 0.f;
    f32x4 wsm[NSW];
#pragma unroll
    for (int sl = 0; sl < NSW; ++sl) wsm[sl] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbg = 0.f;
    wg_barrier();

    const bool vec_v = (vi & 3) == 0, vec_o = (vo & 3) == 0, vec_h = (H & 3) == 0;
    // Tile loads, one tile ahead: every global load of a tile is requested (clamped addresses, no branches) before P7 of the
    // previous tile and written to LDS at the top of its own iteration -- the memory round trip runs under P7-P9.
    WgTileReq<2> rv, ro;
    WgTileReq<1> rg;
    WgTileReq<4> rx;
    float frv[2];
    f32x4 spq[4], dyq[4];  // s_pre / d(s_out) slices of this wave's first tile of so (used in P3)
    auto request = [&](int t) {
        const int tr0 = t * 32, tnv = min(32, rows - tr0);
        const int64_t trow = min(tr0 + e, rows - 1);
        const int ot = min(w, NT - 1);
        // (tile-blocked: the lanes' 16-byte pieces of one register quad are 1 KB of whole lines; an address select, no branch)
        const int64_t tbo = (int64_t)t * 32 * so + (int64_t)ot * 1024 + lane * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = min(32 * ot + 8 * q + 4 * hi, so - 4);
            spq[q] = *reinterpret_cast<const f32x4*>(p.s_pre + ((p.tb & 1) ? tbo + q * 256 : trow * so + c));
            dyq[q] = *reinterpret_cast<const f32x4*>(p.d_s_out + ((p.tb & 2) ? tbo + q * 256 : trow * so + c));
        }
        // A tile that is absent (no output vectors, no gate, no frames) or not in the 16-byte form (its commit loads it itself)
        // still issues its requests -- no branches around loads -- but from `safe`, one row of s_pre (so >= 4: 16 valid bytes),
        // never from another tile's address with this tile's width: that over-read the last tile of v_in by up to
        // 32 (max(vo, 9) - 3 vi) floats, past the end of the caller's allocation when vi is small
        const float* safe = p.s_pre + (int64_t)tr0 * so;
        const float* vsrc = p.v_in + (int64_t)tr0 * 3 * vi;
        const bool v_vec = ((3 * vi) & 3) == 0 && wg_aligned16(p.v_in);
        const bool o_vec = vo > 0 && ((3 * vo) & 3) == 0 && wg_aligned16(p.d_v_out);
        const float* osrc = o_vec ? p.d_v_out + (int64_t)tr0 * 3 * vo : safe;
        const bool g_vec = gated && vec_o && wg_aligned16(p.gate);
        const float* gsrc = g_vec ? p.gate + (int64_t)tr0 * vo : safe;
        wg_tile_request<NTH, 2>(rv, v_vec ? vsrc : safe, 3 * vi, v_vec ? tnv : 0, tid, v_vec);
        wg_tile_request<NTH, 2>(ro, osrc, 3 * max(vo, 1), o_vec ? tnv : 0, tid, o_vec);
        wg_tile_request<NTH, 1>(rg, gsrc, max(vo, 1), g_vec ? tnv : 0, tid, g_vec);
        if constexpr (FUSED) {
            const bool x_vec = (si & 3) == 0 && wg_aligned16(p.s_in);
            const float* xsrc = x_vec ? p.s_in + (int64_t)tr0 * si : safe;
            wg_tile_request<NTH, 4>(rx, xsrc, si, x_vec ? tnv : 0, tid, x_vec);
        }
        const float* fsrc = nf ? p.frames + (int64_t)tr0 * 9 : safe;
#pragma unroll
        for (int k = 0; k < 2; ++k) frv[k] = fsrc[nf ? min(tid + k * NTH, tnv * 9 - 1) : 0];
    };
    request(blockIdx.x);
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        WG_LAUNDER();
        auto stamp = [&](int k) {  // (a tile from the middle of the workgroup's range: steady state, not the drained tail)
#if defined(GCP_WG_STAMP_TAIL) && GCP_WG_STAMP_TAIL == 4
            return;
#endif
            if (w == 0 && (tile - (int)blockIdx.x) / (int)gridDim.x == p.ntiles / (int)gridDim.x / 2) gcp_stamp(p.stamps, p.stamp_cap, k, lane);
        };
        stamp(0);
        const int r0 = tile * 32;
        const int nvalid = min(32, rows - r0);
        const bool row_ok = e < nvalid;
        const int64_t rowc = min(r0 + e, rows - 1);
        {
            const float* vsrc = p.v_in + (int64_t)r0 * 3 * vi;
            const bool v_vec = ((3 * vi) & 3) == 0 && wg_aligned16(p.v_in);
            const float* osrc = vo > 0 ? p.d_v_out + (int64_t)r0 * 3 * vo : vsrc;
            const bool o_vec = vo > 0 && ((3 * vo) & 3) == 0 && wg_aligned16(p.d_v_out);
            const float* gsrc = gated ? p.gate + (int64_t)r0 * vo : vsrc;
            const bool g_vec = gated && vec_o && wg_aligned16(p.gate);
            wg_tile_commit<NTH, 2>(rv, V, VS, vsrc, 3 * vi, nvalid, tid, v_vec, DM(mg_v));
            if (vo > 0) {
                wg_tile_commit<NTH, 2>(ro, DVO, US, osrc, 3 * vo, nvalid, tid, o_vec, DM(mg_o));
                if (gated) wg_tile_commit<NTH, 1>(rg, DG, DGS, gsrc, vo, nvalid, tid, g_vec, DM(mg_g));
            }
            if (nf) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (tid + k * NTH < 32 * 9) FR[tid + k * NTH] = frv[k];
            }
            if constexpr (FUSED) {
                const float* xsrc = p.s_in + (int64_t)r0 * si;
                wg_tile_commit<NTH, 4>(rx, X, KS, xsrc, si, nvalid, tid, (si & 3) == 0 && wg_aligned16(p.s_in), DM(mg_x));
                const int npad = 8 * gcp_cdiv(KW, 8) - K;  // ones column (bias gradient) + zero padding
                for (int i = tid; i < 32 * npad; i += NTH) {
                    const int r = wg_div(i, npad, DM(mg_xpad)), c = i - r * npad;
                    X[r * KS + K + c] = (c == 0 && r < nvalid) ? 1.f : 0.f;
                }
            }
        }
        wg_barrier();
        stamp(1);

        WG_LAUNDER();
        // ---- P1: recompute [vh | vf], norms, frame scalars ---------------------------------------------------------------
        {
            const float* vrow = V + prow * VS;
            const int grow = min(r0 + prow, rows - 1);
            for (int x = psub; x < ((GCP_WG_X & 16) ? 0 : HF); x += TPR) {
                float q0 = 0.f, q1 = 0.f, q2 = 0.f;
                for (int k = 0; k < p.v_add.n; ++k) {  // shares of the pre-projected (gathered) vector sources
                    const int32_t* ix = p.v_add.idx[k];
                    const float* t = p.v_add.ptr[k] + (int64_t)(ix ? ix[grow] : grow) * 3 * HFP;
                    q0 += t[x]; q1 += t[HFP + x]; q2 += t[2 * HFP + x];
                }
                const float* wr = WD + x * WSV;
                float u0 = 0.f, u1 = 0.f, u2 = 0.f;
                if (vec_v) {
                    for (int c = 0; c < vi; c += 4) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + c);
                        const f32x4 a = *reinterpret_cast<const f32x4*>(vrow + 3 * c);
                        const f32x4 b = *reinterpret_cast<const f32x4*>(vrow + 3 * c + 4);
                        const f32x4 d = *reinterpret_cast<const f32x4*>(vrow + 3 * c + 8);
                        u0 = fmaf(wv[0], a[0], u0); u1 = fmaf(wv[0], a[1], u1); u2 = fmaf(wv[0], a[2], u2);
                        u0 = fmaf(wv[1], a[3], u0); u1 = fmaf(wv[1], b[0], u1); u2 = fmaf(wv[1], b[1], u2);
                        u0 = fmaf(wv[2], b[2], u0); u1 = fmaf(wv[2], b[3], u1); u2 = fmaf(wv[2], d[0], u2);
                        u0 = fmaf(wv[3], d[1], u0); u1 = fmaf(wv[3], d[2], u1); u2 = fmaf(wv[3], d[3], u2);
                    }
                } else {
                    for (int c = 0; c < vi; ++c) {
                        const float wv = wr[c];
                        u0 = fmaf(wv, vrow[3 * c + 0], u0); u1 = fmaf(wv, vrow[3 * c + 1], u1); u2 = fmaf(wv, vrow[3 * c + 2], u2);
                    }
                }
                u0 += q0; u1 += q1; u2 += q2;
                if (x < H) {
                    VH[prow * HS + 3 * x + 0] = u0; VH[prow * HS + 3 * x + 1] = u1; VH[prow * HS + 3 * x + 2] = u2;
                    const float nr = sqrtf(u0 * u0 + u1 * u1 + u2 * u2 + 1e-8f);
                    RN[prow * (H | 1) + x] = 1.0f / nr;
                    extp[prow * exs + x] = nr + 1e-8f;
                } else {
                    const int k = x - H;
                    const float* f = FR + prow * 9;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        float pr = f[3 * a + 0] * u0 + f[3 * a + 1] * u1 + f[3 * a + 2] * u2;
                        if (p.e3 && a == 1) {
                            SGN[prow * 3 + k] = pr < 0.f ? -1.f : 1.f;
                            pr = fabsf(pr);
                        }
                        extp[prow * exs + H + 3 * k + a] = pr;
                    }
                }
            }
            if constexpr (!FUSED) {
                const int npad = EP - (H + nf);  // stride padding of the ext rows that go to HBM
                for (int i = tid; i < 32 * npad; i += NTH) {
                    const int r = wg_div(i, npad, DM(mg_epad)), c = i - r * npad;
                    DEXT[r * EXS + H + nf + c] = 0.f;
                }
            }
        }
        // P2 reads, of what P1 wrote, only the vh rows of its own thread's row -- threads of a row sit in one wave (row = tid / TPR):
        // a wave-level LDS sync suffices unless the extras tile leaves for HBM here (not fused)
        if constexpr (FUSED) {
            gcp_wave_lds_sync();
        } else {
            wg_barrier();
            if (p.ext) wg_tile_store<NTH>(p.ext + (int64_t)r0 * EP, DEXT, EXS, EP, nvalid, tid, wg_aligned16(p.ext), DM(mg_ep));
        }

        stamp(2);
        WG_LAUNDER();
        // ---- P2: adjoint of the vector epilogue (gcpnet.py:364-391) --------------------------------------------------------
        if (vo > 0 && !(GCP_WG_X & 32)) {
            for (int o = psub; o < vo; o += TPR) {
                const float* wu = WU + o * WSU;
                const float* vh = VH + prow * HS;
                float u0 = 0.f, u1 = 0.f, u2 = 0.f;
                if (vec_h) {
                    for (int h = 0; h < H; h += 4) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(wu + h);
                        const f32x4 a = *reinterpret_cast<const f32x4*>(vh + 3 * h);
                        const f32x4 b = *reinterpret_cast<const f32x4*>(vh + 3 * h + 4);
                        const f32x4 d = *reinterpret_cast<const f32x4*>(vh + 3 * h + 8);
                        u0 = fmaf(wv[0], a[0], u0); u1 = fmaf(wv[0], a[1], u1); u2 = fmaf(wv[0], a[2], u2);
                        u0 = fmaf(wv[1], a[3], u0); u1 = fmaf(wv[1], b[0], u1); u2 = fmaf(wv[1], b[1], u2);
                        u0 = fmaf(wv[2], b[2], u0); u1 = fmaf(wv[2], b[3], u1); u2 = fmaf(wv[2], d[0], u2);
                        u0 = fmaf(wv[3], d[1], u0); u1 = fmaf(wv[3], d[2], u1); u2 = fmaf(wv[3], d[3], u2);
                    }
                } else {
                    for (int h = 0; h < H; ++h) {
                        const float wv = wu[h];
                        u0 = fmaf(wv, vh[3 * h + 0], u0); u1 = fmaf(wv, vh[3 * h + 1], u1); u2 = fmaf(wv, vh[3 * h + 2], u2);
                    }
                }
                if (p.vres) { u0 += V[prow * VS + 3 * o + 0]; u1 += V[prow * VS + 3 * o + 1]; u2 += V[prow * VS + 3 * o + 2]; }
                const float g0 = DVO[prow * US + 3 * o + 0], g1 = DVO[prow * US + 3 * o + 1], g2 = DVO[prow * US + 3 * o + 2];
                float d0 = g0, d1 = g1, d2 = g2;
                const float dot = g0 * u0 + g1 * u1 + g2 * u2;
                if (gated) {
                    const float sg = DG[prow * DGS + o];
                    d0 = g0 * sg; d1 = g1 * sg; d2 = g2 * sg;
                    DG[prow * DGS + o] = dot * sg * (1.f - sg);
                } else if (p.vmode == GCP_VMODE_SELF_GATE) {
                    const float rs = sqrtf(u0 * u0 + u1 * u1 + u2 * u2 + 1e-8f);
                    const float n = rs + 1e-8f;
                    const float a = gcp_actf<PWL>(p.act_v, ns_v, slope, n), da = gcp_dactf<PWL>(p.act_v, ns_v, slope, n);
                    const float coef = dot * da / rs;
                    d0 = g0 * a + coef * u0; d1 = g1 * a + coef * u1; d2 = g2 * a + coef * u2;
                }
                DVU[prow * US + 3 * o + 0] = d0; DVU[prow * US + 3 * o + 1] = d1; DVU[prow * US + 3 * o + 2] = d2;
            }
        }
#if GCP_ARITH_F16X2
        // (two-term fp16 form of P4: the per-row maxima of ds_pre that P3 collects -- 32 words behind the operand planes, zeroed here:
        // P4 of the tile before is behind a barrier, P3 of this one in front of the next)
        if (B6 && tid < 32) reinterpret_cast<unsigned*>(DS)[2 * NT * (GCP_W6_TERMS * 256) + tid] = 0u;
#endif
        wg_barrier();
        if constexpr (!FUSED) {
            if (gated && p.dgate) wg_tile_store<NTH>(p.dgate + (int64_t)r0 * VOP, DG, DGS, VOP, nvalid, tid, wg_aligned16(p.dgate), DM(mg_vop));
        }

        stamp(3);
        WG_LAUNDER();
        // (MP: the W^T ds_pre accumulators of P4 live across the passes; otherwise they are declared here and first touched in P4)
        f32x16 acc2[KT], acc3;
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
        const int NP = MP ? npass : 1;
        for (int hp = 0; hp < NP; ++hp) {
        const int tpp = MP ? gcp_cdiv(gcp_cdiv(NT, NW), NP) : 4;  // output tiles per wave and pass
        const int t_lo = MP ? hp * tpp : 0, ot_base = NW * t_lo;     // the pass covers the tiles [ot_base, ot_base + NW * tpp)
        const int gb = 4 * ot_base, ge = MP ? min(SG, gb + 4 * NW * tpp) : SG;  // ... = these groups of eight columns of so
        // ---- P3: ds_pre = d(s_out) act_s'(s_pre) + act_v'(s_pre) (Wg^T dgate), this wave's tiles of so -> DS ---------------
        f32x16 spa;  // act_v(s_pre) of the wave's tile (fused: B operand of the gate weight gradient)
        // B6: the first four slabs of P4's weight fragments are requested HERE, in front of P3's ds_pre stores (requested at the top of
        // P4 they queue behind those stores: vmcnt retires loads and stores in issue order)
#ifndef GCP_WG_P4_PRE  // (measured, round 6: 2.349 against 2.376 ms per (256,32) launch, inside the box's noise; a build option)
        constexpr bool P4_PRE = false;
#else
        constexpr bool P4_PRE = B6;
#endif
        constexpr int NTM = GCP_W6_TERMS, SLAB = NTM * 64;  // terms per element of the operand planes / weight image; 16-byte entries per slab
        gcp_u32x4 pre0[NTM], pre1[NTM], pre2[NTM], pre3[NTM];
        if constexpr (P4_PRE) {
            const int NFT0 = split ? NKT - 1 : NKT, NSL0 = 2 * NT;
            const gcp_u32x4* q0 = reinterpret_cast<const gcp_u32x4*>(p.pk + p.offA2b) + (int64_t)min(w, NFT0 - 1) * NSL0 * SLAB + lane;
#pragma unroll
            for (int tm = 0; tm < NTM; ++tm) {
                pre0[tm] = q0[tm * 64];
                pre1[tm] = q0[(int64_t)min(1, NSL0 - 1) * SLAB + tm * 64];
                pre2[tm] = q0[(int64_t)min(2, NSL0 - 1) * SLAB + tm * 64];
                pre3[tm] = q0[(int64_t)min(3, NSL0 - 1) * SLAB + tm * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        for (int t = t_lo; t < t_lo + tpp && w + NW * t < NT; ++t) {
            const int ot = w + NW * t;
            if (t > 0 && !(GCP_WG_X & 512)) {
                const int64_t tbo = (int64_t)tile * 32 * so + (int64_t)ot * 1024 + lane * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = min(32 * ot + 8 * q + 4 * hi, so - 4);
                    spq[q] = *reinterpret_cast<const f32x4*>(p.s_pre + ((p.tb & 1) ? tbo + q * 256 : rowc * so + c));
                    dyq[q] = *reinterpret_cast<const f32x4*>(p.d_s_out + ((p.tb & 2) ? tbo + q * 256 : rowc * so + c));
                }
            }
            f32x16 gacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) gacc[r] = 0.f;
            if (gated && !(GCP_WG_X & 64)) {
                const float* pg = p.pk + p.offG2 + ((int64_t)ot * VG * 64 + lane) * 4;
                for (int g = 0; g < VG; ++g) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(pg + (int64_t)g * 256);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(DG + e * DGS + 8 * g + 4 * hi);
#pragma unroll
                    for (int i = 0; i < 4; ++i) gacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], gacc, 0, 0, 0);
                }
            }
            f32x16 dsp;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool in = row_ok && 32 * ot + 8 * q + 4 * hi + 3 < so;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float sp = spq[q][i];
                    float d = dyq[q][i] * gcp_dactf<PWL>(p.act_s, ns_s, slope, sp);
                    if (gated) d += gcp_dactf<PWL>(p.act_v, ns_v, slope, sp) * gacc[4 * q + i];
                    dsp[4 * q + i] = in ? d : 0.f;
                    spa[4 * q + i] = in ? gcp_actf<PWL>(p.act_v, ns_v, slope, sp) : 0.f;
                }
                if constexpr (!B6) {
                    const f32x4 v = {dsp[4 * q], dsp[4 * q + 1], dsp[4 * q + 2], dsp[4 * q + 3]};
                    *reinterpret_cast<f32x4*>(DS + e * DSS + 32 * (ot - ot_base) + 8 * q + 4 * hi) = v;
                }
            }
            if constexpr (B6) {  // the tile's two K = 16 slabs: eight registers each, split here, once for every wave's P4
                gcp_u32x4* pl = reinterpret_cast<gcp_u32x4*>(DS) + (int64_t)(2 * ot) * SLAB + lane;
#if GCP_ARITH_F16X2
                // two fp16 terms (gcp_f16x2.h) need the ROW's largest |ds_pre| over all of so -- the other waves' tiles too: this tile's
                // share goes to rmax[e] (LDS, behind the planes), the values wait as fp32 in the lane's own 64 bytes of the planes and
                // are split behind the barrier that ends P3
                float m = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) m = fmaxf(m, fabsf(dsp[i]));
                atomicMax(reinterpret_cast<unsigned*>(DS) + 2 * NT * (GCP_W6_TERMS * 256) + e, __float_as_uint(m));
#pragma unroll
                for (int jh = 0; jh < 2; ++jh)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm) {
                        const f32x4 v = {dsp[8 * jh + 4 * tm], dsp[8 * jh + 4 * tm + 1], dsp[8 * jh + 4 * tm + 2], dsp[8 * jh + 4 * tm + 3]};
                        *reinterpret_cast<f32x4*>(pl + jh * SLAB + tm * 64) = v;
                    }
#else
#pragma unroll
                for (int jh = 0; jh < 2; ++jh) {
                    float x8[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) x8[i] = dsp[8 * jh + i];
                    gcp_u32x4 th, tm, tl;
                    gcp_bf16x3_split8(x8, th, tm, tl);
                    pl[jh * 192] = th; pl[jh * 192 + 64] = tm; pl[jh * 192 + 128] = tl;
                }
#endif
            }
            if (GCP_WG_X & 128) {
            } else if (p.ds_pre && (p.tb & 8)) {  // tile-blocked operand of gcpnet_tn_gemm: straight from the registers (zeros in the rows past the end)
                float* dst = p.ds_pre + (int64_t)tile * 32 * so + (int64_t)ot * 1024 + lane * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = {dsp[4 * q], dsp[4 * q + 1], dsp[4 * q + 2], dsp[4 * q + 3]};
                    *reinterpret_cast<f32x4*>(dst + q * 256) = v;
                }
            } else if (p.ds_pre) {  // (head block: summed per source node afterwards; not fused: operand of gcpnet_tn_gemm)
                const int sub = lane >> 2, c4 = 4 * (lane & 3);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    gcp_wave_lds_sync();
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const f32x4 v = {dsp[8 * h + 4 * q], dsp[8 * h + 4 * q + 1], dsp[8 * h + 4 * q + 2], dsp[8 * h + 4 * q + 3]};
                        *reinterpret_cast<f32x4*>(ST + e * 20 + 8 * q + 4 * hi) = v;
                    }
                    gcp_wave_lds_sync();
                    f32x4 wv[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) wv[j] = *reinterpret_cast<const f32x4*>(ST + (16 * j + sub) * 20 + c4);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int r = 16 * j + sub, c = 32 * ot + 16 * h + c4;
                        if (r < nvalid && c < so) *reinterpret_cast<f32x4*>(p.ds_pre + (int64_t)(r0 + r) * so + c) = wv[j];
                    }
                }
            }
            if constexpr (FUSED) {
                // ---- (P6) dWg[vo][own 32 columns of so] += dgate^T act_v(s_pre): reduction over the tile's rows.  The B operand is
                //      the transposed act_v(s_pre) tile: through the wave-private staging tile, 16 columns at a time (no workgroup
                //      barrier, no second pass over DS) ---------------------------------------------------------------------------
                if (gated) {
                    float bt[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        gcp_wave_lds_sync();
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const f32x4 v = {spa[8 * h + 4 * q], spa[8 * h + 4 * q + 1], spa[8 * h + 4 * q + 2], spa[8 * h + 4 * q + 3]};
                            *reinterpret_cast<f32x4*>(ST + e * 20 + 8 * q + 4 * hi) = v;
                        }
                        gcp_wave_lds_sync();
                        const float* sb = ST + hi * 20 + (e & 15);
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const float x = sb[2 * j * 20];
                            bt[j] = (e >> 4) == h ? x : bt[j];
                        }
                    }
                    const float* ga = DG + hi * DGS + min(e, DGS - 1);
                    float at[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) at[j] = ga[2 * j * DGS];
#pragma unroll
                    for (int j = 0; j < 16; ++j) dWg = __builtin_amdgcn_mfma_f32_32x32x2f32(e < vo ? at[j] : 0.f, bt[j], dWg, 0, 0, 0);
                }
            }
        }
        if constexpr (FUSED) {
            if (gated && tid < 8 * vo) {  // gate bias gradient: column sums of dgate, 8 threads (4 rows each) per channel
                const int o = tid >> 3, part = tid & 7;
                float sacc = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) sacc += DG[(4 * part + k) * DGS + o];
                sacc += __shfl_xor(sacc, 1);
                sacc += __shfl_xor(sacc, 2);
                sacc += __shfl_xor(sacc, 4);
                dbg += sacc;
            }
        }
        wg_barrier();
#if GCP_ARITH_F16X2
        [[maybe_unused]] float p4_isc = 1.f;  // 2^-(pa + GCP_F16_WEXP) of this lane's row: what P4's sums are multiplied by
        if constexpr (B6) {
            const float rm = __uint_as_float(reinterpret_cast<const unsigned*>(DS)[2 * NT * (GCP_W6_TERMS * 256) + e]);
            const int pa = gcp_f16_row_exp(rm);
            const float sc = gcp_exp2i(pa);
            p4_isc = gcp_exp2i(-(pa + GCP_F16_WEXP));
            for (int t = t_lo; t < t_lo + tpp && w + NW * t < NT; ++t) {  // (the wave's own tiles, its lanes' own 64 bytes: no other reader yet)
                gcp_u32x4* pl = reinterpret_cast<gcp_u32x4*>(DS) + (int64_t)(2 * (w + NW * t)) * SLAB + lane;
#pragma unroll
                for (int jh = 0; jh < 2; ++jh) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(pl + jh * SLAB), v1 = *reinterpret_cast<const f32x4*>(pl + jh * SLAB + 64);
                    const float x8[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    gcp_u32x4 th, tl;
                    gcp_f16x2_split8(x8, sc, th, tl);
                    pl[jh * SLAB] = th; pl[jh * SLAB + 64] = tl;
                }
            }
            wg_barrier();
        }
#endif

        stamp(4);
        WG_LAUNDER();
        // ---- P4: d[s | norms | frame scalars]^T = W^T ds_pre^T ------------------------------------------------------------
        {
            const int NFT = split ? NKT - 1 : NKT;  // tiles handed out whole, round-robin
            if (w < NFT) {
                int ktc[KT];
#pragma unroll
                for (int j = 0; j < KT; ++j) ktc[j] = min(w + NW * j, NFT - 1);
                f32x4 res[KT][4];  // ResGCP pass-through d(out) (requested now, added after the reduction)
                if (p.residual) {
#pragma unroll
                    for (int j = 0; j < KT; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            res[j][q] = *reinterpret_cast<const f32x4*>(
                                p.d_s_out + ((p.tb & 2) ? (int64_t)tile * 32 * so + (int64_t)min(ktc[j], NT - 1) * 1024 + q * 256 + lane * 4
                                                        : rowc * so + min(32 * ktc[j] + 8 * q + 4 * hi, so - 4)));
                }
                if constexpr (B6) {
                    const int NSL = 2 * NT;  // slabs of the reduction over so
                    const gcp_u32x4* pa6 = reinterpret_cast<const gcp_u32x4*>(p.pk + p.offA2b) + (int64_t)ktc[0] * NSL * SLAB + lane;
                    const gcp_u32x4* pb6 = reinterpret_cast<const gcp_u32x4*>(DS) + lane;
                    auto lda = [&](gcp_u32x4(&a)[NTM], int sj) {
#ifdef GCP_WG_EXP1  // (measurement build, wrong results: every slab's fragments from slabs 0 / 1 of tile 0 -- 6 KB, L1-resident)
                        const gcp_u32x4* q = reinterpret_cast<const gcp_u32x4*>(p.pk + p.offA2b) + lane + (int64_t)(sj & 1) * SLAB;
#else
                        const gcp_u32x4* q = pa6 + (int64_t)min(sj, NSL - 1) * SLAB;
#endif
#pragma unroll
                        for (int tm = 0; tm < NTM; ++tm) a[tm] = q[64 * tm];
                    };
                    gcp_u32x4 a0[NTM], a1[NTM], a2[NTM], a3[NTM];
                    if constexpr (P4_PRE) {
#pragma unroll
                        for (int tm = 0; tm < NTM; ++tm) { a0[tm] = pre0[tm]; a1[tm] = pre1[tm]; a2[tm] = pre2[tm]; a3[tm] = pre3[tm]; }
                    } else {
                        lda(a0, 0); lda(a1, 1); lda(a2, 2); lda(a3, 3);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    for (int sj = 0; sj < NSL; sj += 4) {  // (NSL = 2 NT; a trailing pair is guarded)
#if GCP_ARITH_F16X2
#define WG_B6_STEP(A, S)                                                                                       \
    if ((S) < NSL) {                                                                                           \
        const gcp_u32x4* qb = pb6 + (int64_t)(S) * SLAB;                                                       \
        const gcp_u32x4 bh = qb[0], bl = qb[64];                                                               \
        if constexpr ((GCP_WG_X & 256) == 0) acc2[0] = gcp_mfma_f16x3(A, bh, bl, acc2[0]);                     \
        lda(A, (S) + 4);                                                                                       \
    }                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);
#else
#define WG_B6_STEP(A, S)                                                                                       \
    if ((S) < NSL) {                                                                                           \
        const gcp_u32x4* qb = pb6 + (int64_t)(S) * SLAB;                                                       \
        const gcp_u32x4 bh = qb[0], bm = qb[64], bl = qb[128];                                                 \
        if constexpr ((GCP_WG_X & 256) == 0) acc2[0] = gcp_mfma_bf16x6(A, bh, bm, bl, acc2[0]);                \
        lda(A, (S) + 4);                                                                                       \
    }                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);
#endif
                        WG_B6_STEP(a0, sj)
                        WG_B6_STEP(a1, sj + 1)
                        WG_B6_STEP(a2, sj + 2)
                        WG_B6_STEP(a3, sj + 3)
#undef WG_B6_STEP
                    }
#if GCP_ARITH_F16X2
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[0][r] *= p4_isc;
#endif
                } else {
                constexpr int U = KT == 1 ? 4 : 1;
                const float* pa = p.pk + p.offA2 + (int64_t)lane * 4;
                const float* db = DS + e * DSS + 4 * hi;
                auto ld = [&](f32x4(&a)[U][KT], f32x4(&bb)[U], int g0) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int g = min(g0 + u, ge - 1);
                        bb[u] = *reinterpret_cast<const f32x4*>(db + 8 * (g - gb));
#pragma unroll
                        for (int j = 0; j < KT; ++j) a[u][j] = *reinterpret_cast<const f32x4*>(pa + ((int64_t)ktc[j] * SG + g) * 256);
                    }
                };
                auto mm = [&](f32x4(&a)[U][KT], f32x4(&bb)[U], int g0) {
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (g0 + u < ge) {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
#pragma unroll
                                for (int j = 0; j < KT; ++j)
                                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][j][i], bb[u][i], acc2[j], 0, 0, 0);
                        }
                };
                f32x4 a0[U][KT], a1[U][KT], b0[U], b1[U];
                ld(a0, b0, gb);
                for (int g0 = gb; g0 < ge; g0 += 2 * U) {
                    ld(a1, b1, g0 + U);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(a0, b0, g0);
                    ld(a0, b0, g0 + 2 * U);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(a1, b1, g0 + U);
                }
                }
                if (hp == NP - 1) {  // (the reduction over so is complete)
#pragma unroll
                for (int j = 0; j < KT; ++j) {
                    const int kt = w + NW * j;
                    if (kt < NFT) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int c = 32 * kt + 8 * q + 4 * hi;
                            f32x4 v = {acc2[j][4 * q], acc2[j][4 * q + 1], acc2[j][4 * q + 2], acc2[j][4 * q + 3]};
                            if (p.residual && c + 3 < so) { v[0] += res[j][q][0]; v[1] += res[j][q][1]; v[2] += res[j][q][2]; v[3] += res[j][q][3]; }
                            if (c + 3 < si && (p.tb & 4)) {  // (si % 32 == 0: whole tiles; zeros in the rows past the end)
                                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                                *reinterpret_cast<f32x4*>(p.d_s_in + (int64_t)tile * 32 * si + (int64_t)kt * 1024 + q * 256 + lane * 4) = row_ok ? v : z;
                            } else if (c + 3 < si) {
                                if (row_ok) *reinterpret_cast<f32x4*>(p.d_s_in + (int64_t)(r0 + e) * si + c) = v;
                            } else {
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const int cc = c + i;
                                    if (cc < si) {
                                        if (row_ok) p.d_s_in[(int64_t)(r0 + e) * si + cc] = v[i];
                                    } else if (cc < K) {
                                        DEXT[e * EXS + cc - si] = v[i];
                                    }
                                }
                            }
                        }
                    }
                }
                }
            }
            WG_LAUNDER();
            if (split) {  // the last tile of K: every wave reduces over its share of so, partial sums -> EPART[w]
                const int kt = NKT - 1;
                const int gs = gcp_cdiv(ge - gb, NW), g_lo = gb + w * gs, g_hi = min(ge, g_lo + gs);
                const float* pa = p.pk + p.offA2 + ((int64_t)kt * SG * 64 + lane) * 4;
                const float* db = DS + e * DSS + 4 * hi;
                if constexpr (B6) {
                    const int NSL = 2 * NT, ss = gcp_cdiv(NSL, NW), s_lo = w * ss, s_hi = min(NSL, s_lo + ss);
                    const gcp_u32x4* pa6 = reinterpret_cast<const gcp_u32x4*>(p.pk + p.offA2b) + (int64_t)kt * NSL * SLAB + lane;
                    const gcp_u32x4* pb6 = reinterpret_cast<const gcp_u32x4*>(DS) + lane;
                    for (int sj = s_lo; sj < s_hi; ++sj) {
                        const gcp_u32x4* qa = pa6 + (int64_t)sj * SLAB;
                        const gcp_u32x4* qb = pb6 + (int64_t)sj * SLAB;
#if GCP_ARITH_F16X2
                        const gcp_u32x4 a6[2] = {qa[0], qa[64]};
                        acc3 = gcp_mfma_f16x3(a6, qb[0], qb[64], acc3);
#else
                        const gcp_u32x4 a6[3] = {qa[0], qa[64], qa[128]};
                        acc3 = gcp_mfma_bf16x6(a6, qb[0], qb[64], qb[128], acc3);
#endif
                    }
#if GCP_ARITH_F16X2
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc3[r] *= p4_isc;
#endif
                } else
                for (int g0 = g_lo; g0 < g_hi; g0 += 4) {
                    f32x4 a[4], bb[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int g = max(min(g0 + u, g_hi - 1), gb);  // (a wave whose share of the pass is empty reads, and skips, group gb)
                        a[u] = *reinterpret_cast<const f32x4*>(pa + (int64_t)g * 256);
                        bb[u] = *reinterpret_cast<const f32x4*>(db + 8 * (g - gb));
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (g0 + u < g_hi) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], bb[u][i], acc3, 0, 0, 0);
                        }
                }
                gcp_wave_lds_sync();  // (ST shares this slot: its reads are done)
                if (hp == NP - 1)
                for (int q = 0; 8 * q < LW; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float sacc = 0.f;
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) sacc = qq == q ? acc3[4 * qq + i] : sacc;
                        v[i] = sacc;
                    }
                    *reinterpret_cast<f32x4*>(EPART + (w * 32 + e) * EPS + 8 * q + 4 * hi) = v;
                }
            }
        }
        wg_barrier();  // P4 is done with the ds_pre tile (last pass: its results are in place)
        if (hp + 1 < NP) WG_LAUNDER();
        }  // passes

#ifndef GCP_WG_STAMP_TAIL  // (-DGCP_WG_STAMP_TAIL: stamps 5 and 6 move behind P7 and P8, to split the tail of the tile for tools/wg_phase_timing.py)
        stamp(5);
#endif
        WG_LAUNDER();
        if constexpr (FUSED) {
            // ---- P5: dW[own 32 rows of so][K + 1] += ds_pre^T [s | norms | frame scalars | 1] (reduction over the 32 rows) ---
            if (w < NT) {
                const float* da = DS + hi * DSS + 32 * w + e;
                const float* xb = X + hi * KS + e;
                float a_n = da[0];
                float b_n[FN];
#pragma unroll
                for (int n = 0; n < FN; ++n) b_n[n] = xb[32 * min(n, NNT - 1)];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const float a_c = a_n;
                    float b_c[FN];
#pragma unroll
                    for (int n = 0; n < FN; ++n) b_c[n] = b_n[n];
                    if (j + 1 < 16) {
                        a_n = da[2 * (j + 1) * DSS];
#pragma unroll
                        for (int n = 0; n < FN; ++n) b_n[n] = xb[2 * (j + 1) * KS + 32 * min(n, NNT - 1)];
                    }
#pragma unroll
                    for (int n = 0; n < FN; ++n)
                        if (n < NNT) dW[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_c, b_c[n], dW[n], 0, 0, 0);
                }
            }
        }

#ifndef GCP_WG_STAMP_TAIL
        stamp(6);
#endif
        WG_LAUNDER();
        // the next tile's loads (this tile's again if it is the last: harmless).  Fused: requested after P8 instead -- next to
        // the 100+ persistent accumulator registers the 70 request registers do not fit alongside P7 / P8
        if constexpr (!FUSED) request(min(tile + (int)gridDim.x, p.ntiles - 1));
        // ResGCP pass-through of d(v_out) for P8 (the LDS copy has been recycled; L2 still has the tile): requested here, one
        // phase ahead, for the thread's first two channels
        float gpre[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        if (p.residual && vo > 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float* g = p.d_v_out + ((int64_t)min(r0 + prow, rows - 1) * vo + min(psub + j * TPR, vo - 1)) * 3;
                gpre[j][0] = g[0]; gpre[j][1] = g[1]; gpre[j][2] = g[2];
            }
        }
#if defined(GCP_WG_STAMP_TAIL) && GCP_WG_STAMP_TAIL >= 2 && GCP_WG_STAMP_TAIL != 4  // (variants 2, 3: stamp 5 behind the next tile's requests, stamp 6 behind P7)
        stamp(5);
#endif
        // ---- P7: adjoint of the vector prologue: d vh, d vf ---------------------------------------------------------------
        {
#if defined(GCP_WG_STAMP_TAIL) && GCP_WG_STAMP_TAIL == 3  // (variant 3: P7 without its arithmetic -- what the phase costs empty; results wrong)
            for (int x = psub; x < 0; x += TPR) {
#else
            for (int x = psub; x < ((GCP_WG_X & 2) ? 0 : HF); x += TPR) {
#endif
                float a0 = 0.f, a1 = 0.f, a2 = 0.f;
                // d(extras) of this thread's entry (one column for a hidden channel, three for a frame row), REQUESTED here -- all
                // partial sums of the split tile at once -- and used behind the vector_up^T loop: written as dext(x) calls at the
                // point of use they compile to one LDS round trip per partial sum, 8 - 24 serial waits per thread (8 k cycles per
                // tile at (256,32), tools/wg_phase_timing.py with -DGCP_WG_STAMP_TAIL=2)
                float dxp[3][NW];
                const int xcol0 = x < H ? x : H + 3 * (x - H), xn = x < H ? 1 : 3;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const int xc = xcol0 + (a < xn ? a : 0);       // extras column (a >= xn: a duplicate, not used)
                    const int cc = si + xc - 32 * (NKT - 1);        // its column inside the split tile, if it lies there
                    const bool sp = split && cc >= 0;
#pragma unroll
                    for (int ww = 0; ww < NW; ++ww)
                        dxp[a][ww] = sp ? EPART[(ww * 32 + prow) * EPS + cc] : (ww == 0 ? DEXT[prow * EXS + xc] : 0.f);
                }
                auto dsum = [&](int a) -> float {  // fixed order, as dext() had it
                    float sacc = 0.f;
#pragma unroll
                    for (int ww = 0; ww < NW; ++ww) sacc += dxp[a][ww];
                    return sacc;
                };
                if (x < H) {
                    if (vo > 0) {
                        const float* wt = WUT + x * WTU;
                        const float* du = DVU + prow * US;
                        if (vec_o) {
                            for (int o = 0; o < vo; o += 4) {
                                const f32x4 wv = *reinterpret_cast<const f32x4*>(wt + o);
                                const f32x4 a = *reinterpret_cast<const f32x4*>(du + 3 * o);
                                const f32x4 b = *reinterpret_cast<const f32x4*>(du + 3 * o + 4);
                                const f32x4 d = *reinterpret_cast<const f32x4*>(du + 3 * o + 8);
                                a0 = fmaf(wv[0], a[0], a0); a1 = fmaf(wv[0], a[1], a1); a2 = fmaf(wv[0], a[2], a2);
                                a0 = fmaf(wv[1], a[3], a0); a1 = fmaf(wv[1], b[0], a1); a2 = fmaf(wv[1], b[1], a2);
                                a0 = fmaf(wv[2], b[2], a0); a1 = fmaf(wv[2], b[3], a1); a2 = fmaf(wv[2], d[0], a2);
                                a0 = fmaf(wv[3], d[1], a0); a1 = fmaf(wv[3], d[2], a1); a2 = fmaf(wv[3], d[3], a2);
                            }
                        } else {
                            for (int o = 0; o < vo; ++o) {
                                const float wv = wt[o];
                                a0 = fmaf(wv, du[3 * o + 0], a0); a1 = fmaf(wv, du[3 * o + 1], a1); a2 = fmaf(wv, du[3 * o + 2], a2);
                            }
                        }
                    }
                    const float dn = dsum(0) * RN[prow * (H | 1) + x];
                    a0 += dn * VH[prow * HS + 3 * x + 0]; a1 += dn * VH[prow * HS + 3 * x + 1]; a2 += dn * VH[prow * HS + 3 * x + 2];
                } else if (nf) {
                    const int k = x - H;
                    const float* f = FR + prow * 9;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        float ds = dsum(a);
                        if (p.e3 && a == 1) ds *= SGN[prow * 3 + k];
                        a0 = fmaf(f[3 * a + 0], ds, a0); a1 = fmaf(f[3 * a + 1], ds, a1); a2 = fmaf(f[3 * a + 2], ds, a2);
                    }
                }
                DVHF[prow * FS + 3 * x + 0] = a0; DVHF[prow * FS + 3 * x + 1] = a1; DVHF[prow * FS + 3 * x + 2] = a2;
            }
            if (split) {  // scalar-input columns inside the split tile: summed over the waves' partials and stored
                const int c0 = 32 * (NKT - 1), ns = min(si, K) - c0;  // (> 0 only when the last tile holds scalar columns)
                for (int i = tid; i < 32 * max(ns, 0); i += NTH) {
                    const int r = wg_div(i, ns, DM(mg_ns)), c = i - r * ns;
                    float sacc = 0.f;
#pragma unroll
                    for (int ww = 0; ww < NW; ++ww) sacc += EPART[(ww * 32 + r) * EPS + c];
                    if (p.residual) sacc += r < nvalid ? p.d_s_out[(int64_t)(r0 + r) * so + c0 + c] : 0.f;
                    if (r < nvalid) p.d_s_in[(int64_t)(r0 + r) * si + c0 + c] = sacc;
                }
            }
        }
#if defined(GCP_WG_STAMP_TAIL) && GCP_WG_STAMP_TAIL == 4  // (variant 4: slot w = arrival of wave w at the barrier behind P7; slot layout of the tool does not apply)
        if ((tile - (int)blockIdx.x) / (int)gridDim.x == p.ntiles / (int)gridDim.x / 2) gcp_stamp(p.stamps, p.stamp_cap, w, lane);
#endif
        wg_barrier();
#if defined(GCP_WG_STAMP_TAIL) && GCP_WG_STAMP_TAIL >= 2 && GCP_WG_STAMP_TAIL != 4
        stamp(6);
#elif defined(GCP_WG_STAMP_TAIL) && GCP_WG_STAMP_TAIL == 1
        stamp(5);
#endif

        WG_LAUNDER();
        // ---- P8: d(v_in) = [vector_down ; vector_down_frames]^T d[vh | vf] (+ pass-through terms) ----------------------------
        // d(v_in) leaves through an LDS tile (the split-K partial slots, free since the barrier behind P7) as whole 16-byte pieces of
        // its rows at the end of the tile: written from here as three 4-byte stores per (row, channel) with a 12-byte lane stride,
        // P8 cost 9 % of the (256,32) launch (GCP_WG_X & 4) -- most of it those stores
#ifndef GCP_WG_STAGE_DVIN  // (measured, round 6: 2.349 against 2.355 ms per (256,32) launch -- not the stores; left as a build option)
        const bool stage_dv = false;
#else
        const bool stage_dv = ((3 * vi) & 3) == 0 && wg_aligned16(p.d_v_in) && NW * EPS >= 3 * vi + 4;
#endif
        const int DVS = 4 * ((3 * vi) / 4 | 1);  // row stride of the staged tile (an odd number of 16-byte pieces)
        auto p8 = [&](int c, bool pre, float h0, float h1, float h2) {
            const float* wt = WDT + c * WTV;
            const float* dq = DVHF + prow * FS;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
            // four hidden channels per step, 16-byte LDS reads (both rows are 16-byte aligned: WTV, FS are multiples of 4 floats;
            // a last partial group reads past the row's entries, inside the LDS allocation, and does not use them)
            for (int x0 = 0; x0 < HF; x0 += 4) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wt + x0);
                const f32x4 qa = *reinterpret_cast<const f32x4*>(dq + 3 * x0);
                const f32x4 qb = *reinterpret_cast<const f32x4*>(dq + 3 * x0 + 4);
                const f32x4 qc = *reinterpret_cast<const f32x4*>(dq + 3 * x0 + 8);
                a0 = fmaf(wv[0], qa[0], a0); a1 = fmaf(wv[0], qa[1], a1); a2 = fmaf(wv[0], qa[2], a2);
                if (x0 + 1 < HF) { a0 = fmaf(wv[1], qa[3], a0); a1 = fmaf(wv[1], qb[0], a1); a2 = fmaf(wv[1], qb[1], a2); }
                if (x0 + 2 < HF) { a0 = fmaf(wv[2], qb[2], a0); a1 = fmaf(wv[2], qb[3], a1); a2 = fmaf(wv[2], qc[0], a2); }
                if (x0 + 3 < HF) { a0 = fmaf(wv[3], qc[1], a0); a1 = fmaf(wv[3], qc[2], a1); a2 = fmaf(wv[3], qc[3], a2); }
            }
            if (p.vres && vo > 0) { a0 += DVU[prow * US + 3 * c + 0]; a1 += DVU[prow * US + 3 * c + 1]; a2 += DVU[prow * US + 3 * c + 2]; }
            if (prow < nvalid) {
                if (p.residual) {
                    if (pre) {
                        a0 += h0; a1 += h1; a2 += h2;
                    } else {
                        const float* g = p.d_v_out + ((int64_t)(r0 + prow) * vo + c) * 3;
                        a0 += g[0]; a1 += g[1]; a2 += g[2];
                    }
                }
                if (stage_dv) {
                    float* dp = EPART + prow * DVS + 3 * c;
                    dp[0] = a0; dp[1] = a1; dp[2] = a2;
                } else {
                    float* dp = p.d_v_in + ((int64_t)(r0 + prow) * vi + c) * 3;
                    dp[0] = a0; dp[1] = a1; dp[2] = a2;
                }
            }
        };
        if constexpr ((GCP_WG_X & 4) == 0) {
        if (psub < vi) p8(psub, true, gpre[0][0], gpre[0][1], gpre[0][2]);
        if (psub + TPR < vi) p8(psub + TPR, true, gpre[1][0], gpre[1][1], gpre[1][2]);
        for (int c = psub + 2 * TPR; c < vi; c += TPR) p8(c, false, 0.f, 0.f, 0.f);
        }
        if (p.dvhf) {  // d[vh | vf] per row, [3, HF'] xyz-major: the gradient of the pre-projected vector tables' gathered rows
            const int wdt = 3 * HFP;
            for (int i = tid; i < nvalid * wdt; i += NTH) {
                const int r = wg_div(i, wdt, DM(mg_wdt)), j = i - r * wdt, d = wg_div(j, HFP, DM(mg_hfp)), x = j - d * HFP;
                p.dvhf[(int64_t)r0 * wdt + i] = x < HF ? DVHF[r * FS + 3 * x + d] : 0.f;
            }
        }
#if defined(GCP_WG_STAMP_TAIL) && GCP_WG_STAMP_TAIL < 2
        stamp(6);
#endif
        WG_LAUNDER();
        if constexpr (FUSED) request(min(tile + (int)gridDim.x, p.ntiles - 1));
        // ---- P9: small vector weight gradients on the matrix cores (v_mfma_f32_16x16x4_f32): 16 x 16 output tiles of
        //          d vector_up[o, h] = sum dvu[row, o, :] . vh[row, h, :] and
        //          d [vector_down ; vector_down_frames][x, c] = sum d[vh | vf][row, x, :] . v[row, c, :],
        //          reduction over the tile's 96 (row, xyz) pairs (slot (st, kq) = row 4 (st & 7) + kq, component st >> 3);
        //          tiles round-robin over the waves, accumulators persistent across the workgroup's tiles -----------------
        {
            const int l16 = lane & 15, kq = lane >> 4;
#pragma unroll
            for (int sl = 0; sl < ((GCP_WG_X & 8) ? 0 : NSW); ++sl) {
                const int t = w + NW * sl;
                if (t < sm_tiles) {
                    const bool up = t < sm_up_tiles;
                    const int tt = up ? t : t - sm_up_tiles;
                    const int nn = up ? sm_nu : sm_nd;  // tiles along N
                    const int mt = tt / nn, nt = tt - mt * nn;
                    const int M = up ? vo : HF, N = up ? H : vi;
                    const int m = 16 * mt + l16, n = 16 * nt + l16;
                    const int sa = up ? US : FS, sb = up ? HS : VS;
                    const float* pa = (up ? DVU : DVHF) + 3 * min(m, M - 1) + kq * sa;
                    const float* pb = (up ? VH : V) + 3 * min(n, N - 1) + kq * sb;
                    const bool mok = m < M, nok = n < N;
                    f32x4 acc = wsm[sl];
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        float av[8], bv[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            av[k] = pa[4 * k * sa + d];
                            bv[k] = pb[4 * k * sb + d];
                        }
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(mok ? av[k] : 0.f, nok ? bv[k] : 0.f, acc, 0, 0, 0);
                    }
                    wsm[sl] = acc;
                }
            }
        }
        wg_barrier();  // the tiles are free for the next iteration's loads
        if (stage_dv)  // (the staged d(v_in) rows; the slots are next written in P3 / P4 of the following tile, two barriers from here)
            wg_tile_store<NTH>(p.d_v_in + (int64_t)r0 * 3 * vi, EPART, DVS, 3 * vi, nvalid, tid, true, DM(mg_v));
        stamp(7);
    }

    WG_LAUNDER();
    // ---- partial weight gradients of this workgroup ---------------------------------------------------------------------------
    if constexpr (FUSED) {
        if (w < NT) {
            float* out = p.dw_part + (int64_t)blockIdx.x * so * KW;
#pragma unroll
            for (int n = 0; n < FN; ++n)
                if (n < NNT) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = 32 * w + gcp_crow(r, hi), c = 32 * n + e;
                        if (m < so && c < KW) out[(int64_t)m * KW + c] = dW[n][r];
                    }
                }
            if (gated) {
                float* og = p.dwg_part + (int64_t)blockIdx.x * vo * (so + 1);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = gcp_crow(r, hi), c = 32 * w + e;
                    if (o < vo && c < so) og[(int64_t)o * (so + 1) + c] = dWg[r];
                }
            }
        }
        if (gated && tid < 8 * vo && (tid & 7) == 0)
            p.dwg_part[(int64_t)blockIdx.x * vo * (so + 1) + (int64_t)(tid >> 3) * (so + 1) + so] = dbg;
    }
    if (p.wsm_part) {
        const int l16 = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int sl = 0; sl < NSW; ++sl) {
            const int t = w + NW * sl;
            if (t < sm_tiles) {
                const bool up = t < sm_up_tiles;
                const int tt = up ? t : t - sm_up_tiles;
                const int nn = up ? sm_nu : sm_nd;
                const int mt = tt / nn, nt = tt - mt * nn;
                const int M = up ? vo : HF, N = up ? H : vi;
                const int n = 16 * nt + l16;
                float* out = p.wsm_part + (int64_t)blockIdx.x * n_sm + (up ? 0 : n_up);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * mt + 4 * kq + r;
                    if (i < M && n < N) out[i * N + n] = wsm[sl][r];
                }
            }
        }
    }
}

#undef p
#undef DM
#undef WG_RELOAD
#undef WG_LAUNDER

// out[...] = sum over parts, fixed order.  parts[g][r * C + c] for r < R, c < C; columns c < CW of row r go to
// out_w[r * CW + c], column CW (when C == CW + 1) to out_b[r].  A block sums 64 consecutive entries: its 64 x WR_SL threads are
// WR_SL slices of the parts axis (16 loads in flight each), combined through LDS in a fixed order -- a few hundred parts of a small
// matrix are otherwise one dependent load chain per thread.  (Four slices = 256 threads: a 16-slice form, 1 024 threads per block,
// was tried in round 4 -- faster alone, but this kernel runs on the caller's stream beside the weight-gradient GEMMs, and a block
// that needs sixteen free wave slots of one CU waits for them: 1.16 ms instead of 0.40 ms per launch inside the configs[4] step.)
struct WgReduceArgs {
    gcp_wg_reduce_job_t j[GCP_WG_REDUCE_MAX_JOBS];
};
constexpr int WR_SL = 4;

__global__ __launch_bounds__(64 * WR_SL) void wg_reduce_kernel(WgReduceArgs a) {
    __shared__ float red[WR_SL][64];
    const gcp_wg_reduce_job_t& J = a.j[blockIdx.y];
    const float* __restrict__ parts = J.parts;
    const int G = J.n_parts, R = J.R, C = J.C, CW = J.CW;
    float* __restrict__ out_w = J.out_w;
    float* __restrict__ out_b = J.out_b;
    const int64_t n = (int64_t)R * C;
    if ((int64_t)blockIdx.x * 64 >= n) return;  // (the grid is sized for the largest job of the launch; uniform per block)
    const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + col;
    const int per = (G + WR_SL - 1) / WR_SL, g0 = min(G, sl * per), g1 = min(G, g0 + per);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < n) {
        int g = g0;
        for (; g + 15 < g1; g += 16) {
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = parts[(int64_t)(g + k) * n + i];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k & 7] += v[k];
        }
        for (; g + 3 < g1; g += 4) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = parts[(int64_t)(g + k) * n + i];
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += v[k];
        }
        for (; g < g1; ++g) acc[0] += parts[(int64_t)g * n + i];
    }
    red[sl][col] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (sl == 0 && i < n) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < WR_SL; k += 4) v += (red[k][col] + red[k + 1][col]) + (red[k + 2][col] + red[k + 3][col]);
        const int r = (int)(i / C), c = (int)(i - (int64_t)r * C);
        if (c < CW) out_w[(int64_t)r * CW + c] = v;
        else if (out_b) out_b[r] = v;
    }
}

int g_wg_cus = 0;

template <int NW, int KT, int FN, int SHP = 0, bool B6 = false, bool MP = false>
int launch_bwd(const WgBwdParams& p, bool pwl, int grid, size_t lds_bytes, hipStream_t st) {
    auto go = [&](auto kern) -> int {
        if (lds_bytes > 64 * 1024) {
            hipError_t err = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (err != hipSuccess) return (int)err;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * NW), lds_bytes, st, p);
        GCP_HIP_CHECK_LAUNCH();
        return 0;
    };
    if constexpr (SHP != 0) return go(gcp_wg_bwd_kernel<NW, KT, FN, true, SHP, B6>);  // (the compile-time shapes exist for PWL only)
    return pwl ? go(gcp_wg_bwd_kernel<NW, KT, FN, true, 0, B6, MP>) : go(gcp_wg_bwd_kernel<NW, KT, FN, false, 0, B6, MP>);
}

// True if the launch's shape is exactly the compile-time shape SHP (every integer the kernel would otherwise read).
template <int SHP, int NW, int FN, bool B6 = false>
bool wg_bwd_is_shape(const WgBwdParams& p, const gcp2_weights_t& w, bool gated) {
    if (!(w.si == WgBwdShape<SHP>::S && w.so == WgBwdShape<SHP>::S && w.vi == WgBwdShape<SHP>::V && w.vo == WgBwdShape<SHP>::V &&
          w.hidden == WgBwdShape<SHP>::HID && w.use_frames && gated))
        return false;
    constexpr WgBwdDims CD = wg_bwd_dims(WgBwdShape<SHP>::S, WgBwdShape<SHP>::V, WgBwdShape<SHP>::S, WgBwdShape<SHP>::V,
                                         WgBwdShape<SHP>::HID, 1, 1, FN > 0, NW, B6);
    bool same = true;
#define X(f) same = same && p.f == CD.f;
    WG_BWD_DIMS(X)
    WG_BWD_MAGICS(X)
#undef X
    return same;
}

}  // namespace

extern "C" int gcpnet_wg_reduce_multi(int n_jobs, const gcp_wg_reduce_job_t* jobs, void* stream) {
    if (n_jobs <= 0 || n_jobs > GCP_WG_REDUCE_MAX_JOBS || !jobs) return GCPNET_E_BADARG;
    WgReduceArgs a;
    int blocks = 1;
    for (int k = 0; k < n_jobs; ++k) {
        const gcp_wg_reduce_job_t& J = jobs[k];
        if (!J.parts || J.n_parts <= 0 || J.R <= 0 || J.C <= 0 || J.CW < 0 || J.CW > J.C || !J.out_w) return GCPNET_E_BADARG;
        a.j[k] = J;
        blocks = max(blocks, gcp_cdiv(J.R * J.C, 64));
    }
    hipLaunchKernelGGL(wg_reduce_kernel, dim3((unsigned)blocks, (unsigned)n_jobs), dim3(64 * WR_SL), 0, (hipStream_t)stream, a);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int gcpnet_wg_reduce(const float* parts, int n_parts, int R, int C, int CW, float* out_w, float* out_b, void* stream) {
    gcp_wg_reduce_job_t J;
    J.parts = parts; J.n_parts = n_parts; J.R = R; J.C = C; J.CW = CW; J.out_w = out_w; J.out_b = out_b;
    return gcpnet_wg_reduce_multi(1, &J, stream);
}

// Plan of one backward launch: workgroup count (= rows of the partial buffers), fused or not, scratch widths.
extern "C" int gcpnet_wg_backward_plan(int rows, const gcp2_weights_t* w, const gcp2_opts_t* o, int want_fused, gcp_wg_bwd_plan_t* plan) {
    if (!w || !o || !plan) return GCPNET_E_BADARG;
    if (w->vi <= 0 || (w->so & 3) || w->so < 4) return WG_UNSUPPORTED("no vector input, or so not a multiple of 4");
    const bool gated = o->vmode == GCP_VMODE_SCALAR_GATE && w->vo > 0;
    if (gated && w->vo > 64) return WG_UNSUPPORTED("scalar gate with more than 64 output vectors");  // (the request of the gate tile: 16 pieces of 16 bytes per row)
    const WgShape S = wg_shape(w->si, w->vi, w->so, w->vo, w->hidden, w->use_frames, gated);
    int NW = (w->so > 160 || S.K > 160) ? 8 : 4;
    // (experiment: few reduction columns and 128 < so <= 256 -- the first message GCP of configs[4] -- with four waves, plain form)
    if (getenv("GCPNET_WG_BWD_HEAD4") && w->so > 128 && w->so <= 256 && S.K <= 96) NW = 4;
    if (!g_wg_cus) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        g_wg_cus = cus;
    }
    // About one tile per CU (node rows of configs[1] / configs[3]: 313 / 238 tiles on 256 CUs): an eight-wave workgroup holds a CU
    // alone, tiles past the first round cost a whole second one and a round leaves half of every CU's wave slots to latency; four-wave
    // workgroups start two per CU.  Measured (same box, GCPNET_WG_BWD_SMALL_NW8 = the old choice): configs[1] 11.36 -> 11.20 ms,
    // configs[3] 26.23 -> 26.01 ms.  NOT for a handful of tiles (the 16 / 63 node tiles of configs[0] / configs[3]'s NMS batches: most
    // CUs idle, a tile's latency is what counts and eight waves per tile shorten it: captured c1 2.53 vs 2.60 ms, c4 4.02 vs 4.12).
    static const bool small_nw4 = getenv("GCPNET_WG_BWD_SMALL_NW8") == nullptr;
    if (NW == 8 && small_nw4 && 2 * gcp_cdiv(rows, 32) > g_wg_cus && gcp_cdiv(rows, 32) <= 2 * g_wg_cus && gcp_cdiv(S.NT, 4) <= 4) {
        const WgBwdDims D4 = wg_bwd_dims(w->si, w->vi, w->so, w->vo, w->hidden, w->use_frames, gated,
                                         want_fused && !getenv("GCPNET_WG_BWD_NOFUSE"), 4);
        if (D4.KTn <= 4 && D4.sm_tiles <= NSW * 4 && D4.npass == 1 && (size_t)D4.lds_floats * sizeof(float) <= 160 * 1024) NW = 4;
    }
    if (const char* ev = getenv("GCPNET_WG_BWD_NW")) {  // (tuning knob: 4 or 8 waves per workgroup)
        if (ev[0] == '4') NW = 4;
        if (ev[0] == '8') NW = 8;
    }
    if (gcp_cdiv(S.NT, NW) > 4) return WG_UNSUPPORTED("more than four output tiles per wave");
    const WgBwdDims D = wg_bwd_dims(w->si, w->vi, w->so, w->vo, w->hidden, w->use_frames, gated,
                                    want_fused && !getenv("GCPNET_WG_BWD_NOFUSE"), NW);
    if (D.KTn > 4) return WG_UNSUPPORTED("more than four K tiles per wave");
    if (D.sm_tiles > NSW * NW) return WG_UNSUPPORTED("small vector weight gradients: more 16 x 16 tiles than the waves hold");
    const int KTn = D.KTn, fused = D.fused, KW = D.KW, n_sm = D.n_sm, split = D.split;
    const int ntiles = gcp_cdiv(rows, 32);
    plan->nw = NW;
    plan->kt = KTn == 1 ? 1 : 4;
    plan->fused = fused;
    const int per_cu = (fused && NW == 4 && D.NNT <= 2 && getenv("GCPNET_WG_BWD_FN2")) ? 3 : 2;  // resident workgroups per CU of the instantiation that will run
    static const int per_cu_env = getenv("GCPNET_WG_BWD_PER_CU") ? atoi(getenv("GCPNET_WG_BWD_PER_CU")) : 0;  // (tuning knob: persistent workgroups per CU)
    plan->grid = ntiles <= 0 ? 1 : min(ntiles, (per_cu_env > 0 ? per_cu_env : per_cu) * g_wg_cus);
    plan->kw = KW;
    plan->n_small = n_sm;
    plan->ext_w = gcp_round_up(S.H + S.nf, 4);
    plan->dgate_w = gcp_round_up(w->vo, 4);
    plan->split = split;
    return 0;
}

extern "C" int gcpnet_wg_backward(int rows, const gcp_wg_bwd_args_t* a, void* stream) {
    if (rows < 0 || !a) return GCPNET_E_BADARG;
    const gcp2_weights_t& w = a->w;
    const gcp2_opts_t& o = a->o;
    gcp_wg_bwd_plan_t pl;
    if (int rc = gcpnet_wg_backward_plan(rows, &w, &o, a->dw_part != nullptr, &pl)) return rc;
    if ((a->dw_part != nullptr) != (pl.fused != 0)) return GCPNET_E_BADARG;  // the caller sizes its buffers from the same plan
    if (!w.pack || !w.w_down || (w.vo > 0 && !w.w_up) || !a->v_in || !a->s_pre || !a->d_s_out || !a->d_s_in || !a->d_v_in)
        return GCPNET_E_BADARG;
    if (w.vo > 0 && !a->d_v_out) return GCPNET_E_BADARG;
    const bool gated = o.vmode == GCP_VMODE_SCALAR_GATE && w.vo > 0;
    if (gated && !a->gate) return GCPNET_E_BADARG;
    if (a->residual && (w.si != w.so || w.vi != w.vo)) return GCPNET_E_BADARG;
    if (o.vector_residual && w.vi != w.vo) return GCPNET_E_BADARG;
    auto misaligned = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) != 0; };
    if (misaligned(a->s_pre) || misaligned(a->d_s_out) || misaligned(a->d_s_in) || misaligned(w.pack) || misaligned(a->ds_pre))
        return WG_UNSUPPORTED("a scalar tensor is not 16-byte aligned");
    if (a->residual && (w.si & 3)) return WG_UNSUPPORTED("residual block with si not a multiple of 4");
    if (a->tb & ~15) return GCPNET_E_BADARG;
    if (((a->tb & 11) && (w.so & 31)) || ((a->tb & 4) && (w.si & 31)) || ((a->tb & 8) && !a->ds_pre))
        return GCPNET_E_BADARG;  // tile-blocked tensors: whole 32-column tiles
    const WgShape S = wg_shape(w.si, w.vi, w.so, w.vo, w.hidden, w.use_frames, gated);
    if (S.nf && (!a->frames || !w.w_frames)) return GCPNET_E_BADARG;
    if (pl.fused && (!a->s_in || (gated && !a->dwg_part))) return GCPNET_E_BADARG;
    if (rows == 0) return 0;
    WgBwdParams p;
    p.rows = rows; p.ntiles = gcp_cdiv(rows, 32);
    p.s_in = a->s_in; p.v_in = a->v_in; p.frames = a->frames; p.s_pre = a->s_pre; p.gate = a->gate;
    p.d_s_out = a->d_s_out; p.d_v_out = a->d_v_out;
    p.v_add.n = 0;
    if (a->v_add) p.v_add = *a->v_add;
    if (p.v_add.n < 0 || p.v_add.n > GCP_MAX_SEG) return GCPNET_E_BADARG;
    p.d_s_in = a->d_s_in; p.d_v_in = a->d_v_in;
    p.tb = a->tb;
    p.pk = w.pack; p.offA2 = S.offA2; p.offG2 = S.offG2; p.offA2b = S.offA2b;
    p.w_down = w.w_down; p.w_frames = w.w_frames; p.w_up = w.w_up;
    p.ds_pre = a->ds_pre; p.dvhf = a->dvhf; p.ext = a->ext; p.dgate = a->dgate;
    p.dw_part = a->dw_part; p.dwg_part = a->dwg_part; p.wsm_part = a->wsm_part;
    const int NW = pl.nw;
    // P4 on the bf16 pipe (plain mode, one K tile per wave, the operand planes fit in LDS) unless GCPNET_WG_BWD_FP32_MFMA /
    // gcpnet_debug_set_fp32_mfma select the fp32 MFMA form
    static const bool b6_env = getenv("GCPNET_WG_BWD_FP32_MFMA") == nullptr;
    bool b6 = !pl.fused && pl.kt == 1 && (g_gcp_fp32_mfma < 0 ? b6_env : g_gcp_fp32_mfma == 0);
    if (b6 && (size_t)wg_bwd_dims(w.si, w.vi, w.so, w.vo, w.hidden, w.use_frames, gated, pl.fused, NW, 1).lds_floats * sizeof(float) > 160 * 1024)
        b6 = false;
    const WgBwdDims D = wg_bwd_dims(w.si, w.vi, w.so, w.vo, w.hidden, w.use_frames, gated, pl.fused, NW, b6);
    if (D.fused != pl.fused || D.split != pl.split || D.KW != pl.kw) return GCPNET_E_BADARG;  // (same function: cannot differ)
#define X(f) p.f = D.f;
    WG_BWD_DIMS(X)
    WG_BWD_MAGICS(X)
#undef X
    p.act_s = o.act_s; p.act_v = o.act_v; p.vmode = w.vo > 0 ? o.vmode : GCP_VMODE_NONE; p.vres = o.vector_residual; p.e3 = o.e3;
    p.residual = a->residual; p.slope = o.slope;
    for (int k = 0; k < p.v_add.n; ++k)
        if (!p.v_add.ptr[k] || p.v_add.dim[k] != p.HFP) return GCPNET_E_BADARG;
    const size_t lds_bytes = (size_t)D.lds_floats * sizeof(float);
    if (lds_bytes > 160 * 1024) {
        char why[200];
        snprintf(why, sizeof why, "the tile set (%zu bytes: (%d,%d)->(%d,%d), hidden %d, %d waves, K tiles per wave %d) does not fit in 160 KB of LDS",
                 lds_bytes, w.si, w.vi, w.so, w.vo, w.hidden, NW, D.KTn);
        return WG_UNSUPPORTED(why);
    }
    const bool pwl = gcp_is_pwl(o.act_s) && gcp_is_pwl(o.act_v);
    p.stamps = g_gcp_phase_buf; p.stamp_cap = g_gcp_phase_cap;
    hipStream_t st = (hipStream_t)stream;
    const int grid = pl.grid;
#ifdef GCP_WG_ONLY_SHIPPED  // (development builds: only the instantiation of the configs[4] chain)
    if (pwl && !pl.fused && NW == 8 && pl.kt == 1 && b6 && wg_bwd_is_shape<2, 8, 0, true>(p, w, gated))
        return launch_bwd<8, 1, 0, 2, true>(p, pwl, grid, lds_bytes, st);
    return GCPNET_E_UNSUPPORTED;
#else
    // compile-time shapes: the residual message GCPs of BASELINE configs[1] (fused, 4 waves) and configs[4] (8 waves)
    if (pwl && !getenv("GCPNET_WG_BWD_NOSHAPE")) {
        if (pl.fused && NW == 4 && wg_bwd_is_shape<1, 4, 5>(p, w, gated)) return launch_bwd<4, 1, 5, 1>(p, pwl, grid, lds_bytes, st);
        if (!pl.fused && NW == 8 && pl.kt == 1 && b6 && wg_bwd_is_shape<2, 8, 0, true>(p, w, gated))
            return launch_bwd<8, 1, 0, 2, true>(p, pwl, grid, lds_bytes, st);
        if (!pl.fused && NW == 8 && pl.kt == 1 && !b6 && wg_bwd_is_shape<2, 8, 0>(p, w, gated))
            return launch_bwd<8, 1, 0, 2>(p, pwl, grid, lds_bytes, st);
    }
    if (D.npass > 1)
        return NW == 4 ? launch_bwd<4, 1, 0, 0, false, true>(p, pwl, grid, lds_bytes, st) : launch_bwd<8, 1, 0, 0, false, true>(p, pwl, grid, lds_bytes, st);
    if (b6) return NW == 4 ? launch_bwd<4, 1, 0, 0, true>(p, pwl, grid, lds_bytes, st) : launch_bwd<8, 1, 0, 0, true>(p, pwl, grid, lds_bytes, st);
    // (FN = 2, three workgroups per CU: measured in round 3 -- 429 us against 267 us per launch for the first message GCP of
    // configs[1], same step time; opt-in for experiments)
    if (pl.fused && NW == 4 && D.NNT <= 2 && getenv("GCPNET_WG_BWD_FN2")) return launch_bwd<4, 1, 2>(p, pwl, grid, lds_bytes, st);
    if (pl.fused) return NW == 4 ? launch_bwd<4, 1, 5>(p, pwl, grid, lds_bytes, st) : launch_bwd<8, 1, 5>(p, pwl, grid, lds_bytes, st);
    if (NW == 4) return pl.kt == 1 ? launch_bwd<4, 1, 0>(p, pwl, grid, lds_bytes, st) : launch_bwd<4, 4, 0>(p, pwl, grid, lds_bytes, st);
    return pl.kt == 1 ? launch_bwd<8, 1, 0>(p, pwl, grid, lds_bytes, st) : launch_bwd<8, 4, 0>(p, pwl, grid, lds_bytes, st);
#endif
}
