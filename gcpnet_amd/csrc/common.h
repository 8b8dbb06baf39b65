// Device-side helpers shared by the gfx950 kernels of libgcpnet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gcpnet_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GCP_WAVE 64
#define GCP_TILE_ROWS 32  // rows (edges or nodes) owned by one wavefront: the N dimension of v_mfma_f32_32x32x2_f32

// ---- activations (models/__init__.py:42-57) and their derivatives w.r.t. the pre-activation ----------------
#define GCP_SELU_ALPHA 1.6732632423543772f
#define GCP_SELU_SCALE 1.0507009873554805f

__device__ __forceinline__ float gcp_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ float gcp_act(int act, float x, float slope) {
    switch (act) {
        case GCP_ACT_RELU: return x > 0.f ? x : 0.f;
        case GCP_ACT_LEAKYRELU: return x > 0.f ? x : slope * x;
        case GCP_ACT_SELU: return GCP_SELU_SCALE * (x > 0.f ? x : GCP_SELU_ALPHA * (__expf(x) - 1.0f));
        case GCP_ACT_SILU: return x * gcp_sigmoid(x);
        case GCP_ACT_SIGMOID: return gcp_sigmoid(x);
        default: return x;
    }
}

__device__ __forceinline__ float gcp_act_grad(int act, float x, float slope) {
    switch (act) {
        case GCP_ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case GCP_ACT_LEAKYRELU: return x > 0.f ? 1.f : slope;
        case GCP_ACT_SELU: return GCP_SELU_SCALE * (x > 0.f ? 1.f : GCP_SELU_ALPHA * __expf(x));
        case GCP_ACT_SILU: {
            float s = gcp_sigmoid(x);
            return s * (1.f + x * (1.f - s));
        }
        case GCP_ACT_SIGMOID: {
            float s = gcp_sigmoid(x);
            return s * (1.f - s);
        }
        default: return 1.f;
    }
}

// Activations inside the kernels.  identity / relu / leakyrelu (every shipped GCP2 config) are one branch-free
// piecewise-linear formula with a wave-uniform negative-side slope (1, 0, slope); kernels are instantiated with
// PWL = true for them.  selu / silu / sigmoid go through the generic switch in a separate instantiation (PWL = false), so
// that the common path does not carry exp / divide code inside its unrolled loops.
__host__ __device__ inline bool gcp_is_pwl(int act) { return act == GCP_ACT_NONE || act == GCP_ACT_RELU || act == GCP_ACT_LEAKYRELU; }
__host__ __device__ inline float gcp_neg_slope(int act, float slope) {
    return act == GCP_ACT_NONE ? 1.f : (act == GCP_ACT_RELU ? 0.f : slope);
}
template <bool PWL>
__device__ __forceinline__ float gcp_actf(int act, float ns, float slope, float x) {
    if constexpr (PWL) return x > 0.f ? x : (ns == 0.f ? 0.f : ns * x);
    else return gcp_act(act, x, slope);
}
template <bool PWL>
__device__ __forceinline__ float gcp_dactf(int act, float ns, float slope, float x) {
    if constexpr (PWL) return x > 0.f ? 1.f : ns;
    else return gcp_act_grad(act, x, slope);
}

// Workgroup = one wavefront in the GCP kernels, so cross-lane hand-offs through LDS need no s_barrier and, above all,
// no vmcnt(0): __syncthreads() would also wait for every outstanding global STORE of the wave to retire (vmcnt counts
// stores on CDNA), which costs ~10 us per staging step under load.  DS operations of one wave complete in order;
// waiting for lgkmcnt(0) (and stopping the compiler from moving memory operations across) is sufficient.
__device__ __forceinline__ void gcp_wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Row index inside a 32x32 MFMA C/D tile held by (register r, lane half hi): the column is lane & 31.
__device__ __forceinline__ int gcp_crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__host__ __device__ __forceinline__ int gcp_round_up(int x, int m) { return (x + m - 1) / m * m; }
__host__ __device__ __forceinline__ int gcp_cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ __forceinline__ int gcp_odd(int x) { return x | 1; }

// ---- derived shape constants of one GCP2 block, shared by host packing code and kernels ---------------------
// Arithmetic of the big products of the wave-per-tile chain kernels (scalar_out forward, its adjoint): 1 (default) = two fp16 terms per
// operand, three MFMAs per product block (gcp_f16x2.h: 3 * 2^-22 |a b|); 0 (-DGCP_ARITH_F16X2=0) = three bf16 terms, six MFMAs
// (gcp_bf16x3.h: 3 * 2^-24 |a b|), the form of rounds 3 - 5.  Compile-time: the packed weight images differ.
#ifndef GCP_ARITH_F16X2
#define GCP_ARITH_F16X2 1
#endif
#define GCP_W6_TERMS (GCP_ARITH_F16X2 ? 2 : 3)

struct GcpShape {
    int si, vi, so, vo, H, nf;  // nf = 9 frame scalars or 0
    int K;    // merged width  = si (+ H + nf when vi > 0)
    int KP;   // K rounded up to 8 (MFMA k-pairs, 4-deep software pipeline)
    int KK;   // KP / 2 forward steps
    int NTG;  // 32-wide output tiles per accumulator group (1, 2 or 4)
    int NG;   // output groups
    int NUG;  // 32-wide tiles of the merged axis per backward accumulator group
    int NGK;  // merged-axis groups
    int NS;   // backward-data reduction steps = NG * NTG * 16
    int GT;   // 32-wide tiles of the vector-gate outputs (forward gate GEMM, fed from the accumulator registers)
    int NOO;  // gate backward steps: k-pair step r pairs the gate outputs (crow(r, 0), crow(r, 1)) = (j0, j0 + 4)
    int NTS;  // 32-wide tiles of the scalar input (register-resident chain kernel: the state is the B operand)
    // small vector Linears on the matrix cores (vec_mfma.h); vmm = the shape fits (H + 3 <= 32, vo <= 32)
    int vmm, HF;             // HF = H + (nf ? 3 : 0): rows of [vector_down ; vector_down_frames]
    int SVA, SVB, SVC, SVD;  // k-pair steps of the four products, CT = 32-wide tiles of vi
    int CT;
    int64_t offA, offB, offC, offD, offF, offVA, offVB, offVC, offVD, total;  // section offsets (floats) inside the packed image
    // B6: the backward-data weights once more as THREE bf16 terms (w = h + m + l exactly, by truncation) in the operand layout of
    // v_mfma_f32_32x32x16_bf16, for the chain backward kernel's fp32-exact product on the bf16 matrix pipe (gcp_bf16x3.h):
    // [slab j < 2 NTG][tile uu < NKT of the (padded) merged axis][term][64 lanes][4 dwords of two bf16]; 0 floats when the block
    // cannot run in that kernel
    // (round 6, GCP_ARITH_F16X2, the default: B6 and F6 hold TWO fp16 terms of 2^GCP_F16_WEXP w instead -- gcp_f16x2.h, three MFMAs
    // per product block; `GCP_W6_TERMS` terms per element either way.  The gate image C6 stays three bf16 terms.)
    int NKT;
    int64_t offB6;
    // F6 / C6: the same three-term bf16 images of the forward scalar_out weights over a register-resident state
    // ([slab j < 2 NTG][output tile t < NTG][term][64][4]) and of the gate Linear ([slab][term][64][4]), for the wave-per-tile
    // chain forward kernel; 0 floats when the block cannot run there
    int64_t offF6, offC6;
};

__host__ __device__ inline GcpShape gcp_shape(int si, int vi, int so, int vo, int H, int use_frames) {
    GcpShape s;
    s.si = si; s.vi = vi; s.so = so; s.vo = vo; s.H = vi > 0 ? H : 0;
    s.nf = (vi > 0 && use_frames) ? 9 : 0;
    s.K = si + s.H + s.nf;
    s.KP = gcp_round_up(s.K, 8);
    s.KK = s.KP / 2;
    s.NTG = so <= 32 ? 1 : (so <= 64 ? 2 : 4);
    s.NG = gcp_cdiv(gcp_cdiv(so, 32), s.NTG);
    s.NUG = s.K <= 32 ? 1 : (s.K <= 64 ? 2 : 4);
    s.NGK = gcp_cdiv(gcp_cdiv(s.K, 32), s.NUG);
    s.NS = s.NG * s.NTG * 16;
    s.GT = gcp_cdiv(vo, 32);
    s.NOO = 4 * gcp_cdiv(vo, 8);
    s.offA = 0;
    s.offB = s.offA + (int64_t)s.NG * s.KK * 64 * s.NTG;
    s.offC = s.offB + (int64_t)s.NGK * s.NS * 64 * s.NUG;
    s.offD = s.offC + (int64_t)s.GT * s.NS * 64;
    s.NTS = gcp_cdiv(si, 32);
    s.offF = s.offD + (int64_t)s.NOO * s.NG * 64 * s.NTG;
    // F (only used when the block can run in the register-resident chain kernel: one output group)
    s.offVA = s.offF + (s.NG == 1 ? (int64_t)s.NTS * 16 * 64 * s.NTG : 0);
    // V: [vector_down ; vector_down_frames] forward, vector_up forward, vector_up^T and [down ; frames]^T (backward)
    s.HF = s.H + (s.nf ? 3 : 0);
    s.vmm = (vi > 0 && vo > 0 && s.HF <= 32 && vo <= 32) ? 1 : 0;
    s.SVA = gcp_cdiv(vi, 2);
    s.SVB = 4 * gcp_cdiv(s.H, 8);
    s.SVC = s.NOO;
    s.SVD = 4 * gcp_cdiv(s.HF, 8);
    s.CT = gcp_cdiv(vi, 32);
    s.offVB = s.offVA + (s.vmm ? (int64_t)s.SVA * 64 : 0);
    s.offVC = s.offVB + (s.vmm ? (int64_t)s.SVB * 64 : 0);
    s.offVD = s.offVC + (s.vmm ? (int64_t)s.SVC * 64 : 0);
    s.offB6 = s.offVD + (s.vmm ? (int64_t)s.CT * s.SVD * 64 : 0);
    // (the merged axis of the B6 image is PADDED: scalars in tiles 0 .. NTS - 1 -- columns past si are zero --, norms / frame scalars
    // in tile NTS, as the chain backward kernel walks it; identical to the raw axis when si is a multiple of 32)
    s.NKT = s.NTS + gcp_cdiv(s.H + s.nf, 32);
    const bool chainable = s.NG == 1 && si == so && vi == vo && vi > 0 && (si & 3) == 0 && s.NTS == s.NTG && s.NKT == s.NTG + 1;
    s.offF6 = s.offB6 + (chainable ? (int64_t)2 * s.NTG * s.NKT * GCP_W6_TERMS * 256 : 0);
    const bool fwd6 = s.NG == 1 && s.NTG >= 2 && s.NTS == s.NTG && s.GT == 1 && vi > 0 && vo > 0;
    s.offC6 = s.offF6 + (fwd6 ? (int64_t)2 * s.NTG * s.NTG * GCP_W6_TERMS * 256 : 0);
    s.total = s.offC6 + (fwd6 ? (int64_t)2 * s.NTG * 3 * 256 : 0);
    return s;
}

// ---- optional phase timing (profiling hook): when a buffer is registered with gcpnet_debug_set_phase_timing(), every
// wave-tile writes up to GCP_MAX_STAMPS s_memtime stamps to buf[tile * GCP_MAX_STAMPS + k] ---------------------------
#define GCP_MAX_STAMPS 8
extern unsigned long long* g_gcp_phase_buf;  // host-side copy of the registered device pointer (graph_ops.hip)
extern long long g_gcp_phase_cap;
// gcpnet_debug_set_fp32_mfma: -1 = not set (each launcher's environment variable decides), 0 = bf16 x 6, 1 = fp32 MFMA
extern int g_gcp_fp32_mfma;
__device__ __forceinline__ void gcp_stamp(unsigned long long* buf, long long cap, int k, int lane) {
    if (buf && lane == 0 && (long long)blockIdx.x < cap) buf[(long long)blockIdx.x * GCP_MAX_STAMPS + k] = __builtin_amdgcn_s_memtime();
}

#define GCP_HIP_CHECK_LAUNCH()                         \
    do {                                               \
        hipError_t e__ = hipGetLastError();            \
        if (e__ != hipSuccess) return (int)e__;        \
    } while (0)
