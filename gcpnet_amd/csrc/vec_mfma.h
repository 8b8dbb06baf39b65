// The small vector Linears of a GCP2 block (vector_down [H, vi], vector_down_frames [3, vi], vector_up [vo, H]; reference
// src/models/components/gcpnet.py:303-324,420-435,364-391) on the matrix cores, in the same transposed, rows-on-lanes form as
// scalar_out: out^T[channels, 32 rows] = W[channels, k] x in^T[k, 32 rows], one product per xyz component.
//
// Every intermediate lives in the 32x32 C/D layout of v_mfma_f32_32x32x2_f32: register r of lane (row = lane & 31, half =
// lane >> 5) holds channel gcp_crow(r, half) of that row.  Two consequences the kernels rely on:
//   * the per-row nonlinear steps between the Linears (norms, frame projections, gating, their adjoints) are plain
//     element-wise register arithmetic -- a row's xyz components of one channel sit in the SAME register index of three
//     accumulators;
//   * a C/D register is directly the B fragment of a k-pair step over channels (j0, j0 + 4) (j0 = crow(r, 0)), so chained
//     Linears (vector_down -> vector_up, and all the adjoints) need no LDS round trip; the weights are packed for that
//     pairing (sections VB / VC / VD of the packed image, pack_gcp2_kernel).
// Only the first product reads LDS (the row's input vectors, one conflict-free ds_read_b32 per xyz component and step).
#pragma once
#include "common.h"

typedef f32x16 gcp_xyz_acc[3];

__device__ __forceinline__ void gcp_xyz_zero(gcp_xyz_acc& u) {
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) u[d][r] = 0.f;
}

// u[d][x] = sum_c Wdf[x, c] v[row, c, d]: A fragments from section VA (`wa` already offset by the lane), B fragments from the
// row's vector tile `vrow` ([c][xyz] floats).  MAXS = compile-time bound on the k-pair steps (ceil(vi / 2)).
template <int MAXS>
__device__ __forceinline__ void gcp_vmm_down(const float* __restrict__ wa, int steps, int vi, const float* vrow, int hi,
                                             gcp_xyz_acc& u) {
    float a[MAXS];
#pragma unroll
    for (int s = 0; s < MAXS; ++s) a[s] = wa[(int64_t)min(s, steps - 1) * 64];
    gcp_xyz_zero(u);
#pragma unroll
    for (int s0 = 0; s0 < MAXS; s0 += 2)
        if (s0 < steps) {  // wave-uniform, one test per two steps (a step past the end multiplies by a zero weight)
#pragma unroll
            for (int s = s0; s < s0 + 2 && s < MAXS; ++s) {
                const float* b = vrow + 3 * min(2 * s + hi, vi - 1);  // (an odd vi pads with a zero weight)
                const float as = s < steps ? a[s] : 0.f;
#pragma unroll
                for (int d = 0; d < 3; ++d) u[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(as, b[d], u[d], 0, 0, 0);
            }
        }
}

// Same for any vi: the k-pair steps go in chunks of eight (runtime loop), each chunk's weight fragments requested together.
__device__ __forceinline__ void gcp_vmm_down_loop(const float* __restrict__ wa, int steps, int vi, const float* vrow, int hi,
                                                  gcp_xyz_acc& u) {
    gcp_xyz_zero(u);
    for (int s0 = 0; s0 < steps; s0 += 8) {
        float a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = wa[(int64_t)min(s0 + i, steps - 1) * 64];
#pragma unroll
        for (int i0 = 0; i0 < 8; i0 += 2)
            if (s0 + i0 < steps) {  // wave-uniform
#pragma unroll
                for (int i = i0; i < i0 + 2; ++i) {
                    const int s = s0 + i;
                    const float* b = vrow + 3 * min(2 * s + hi, vi - 1);
                    const float as = s < steps ? a[i] : 0.f;
#pragma unroll
                    for (int d = 0; d < 3; ++d) u[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(as, b[d], u[d], 0, 0, 0);
                }
            }
    }
}

// out[d] += W x in[d], the reduction running over the channels held by registers r < steps of `in` (k-pair (crow(r, 0),
// crow(r, 1)) per step); `w` = section VB / VC / VD already offset by the lane.
template <int MAXR>
__device__ __forceinline__ void gcp_vmm_regs(const float* __restrict__ w, int steps, const gcp_xyz_acc& in, gcp_xyz_acc& out) {
    float a[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; ++r) a[r] = w[(int64_t)min(r, steps - 1) * 64];
    static_assert(MAXR % 4 == 0, "steps come in groups of four registers");
#pragma unroll
    for (int r0 = 0; r0 < MAXR; r0 += 4)
        if (r0 < steps) {  // wave-uniform; the step counts are multiples of 4
#pragma unroll
            for (int r = r0; r < r0 + 4; ++r)
#pragma unroll
                for (int d = 0; d < 3; ++d) out[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], in[d][r], out[d], 0, 0, 0);
        }
}

// Same with the input held in plain register arrays in[d][r] (r < NR).
template <int NR>
__device__ __forceinline__ void gcp_vmm_arr(const float* __restrict__ w, int steps, const float (&in)[3][NR], gcp_xyz_acc& out) {
    float a[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) a[r] = w[(int64_t)min(r, steps - 1) * 64];
    static_assert(NR % 4 == 0, "steps come in groups of four registers");
#pragma unroll
    for (int r0 = 0; r0 < NR; r0 += 4)
        if (r0 < steps) {  // wave-uniform; the step counts are multiples of 4
#pragma unroll
            for (int r = r0; r < r0 + 4; ++r)
#pragma unroll
                for (int d = 0; d < 3; ++d) out[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], in[d][r], out[d], 0, 0, 0);
        }
}

// Forms of gcp_vmm_regs / gcp_vmm_arr whose weight fragments were requested earlier by the caller (gcp_vmm_frags): a kernel that has
// stores in flight issues these requests BEFORE the stores -- vmcnt retires loads and stores in issue order, so a fragment
// requested behind a batch of stores is only usable once every one of those stores has been acknowledged by L2.
template <int NR>
__device__ __forceinline__ void gcp_vmm_frags(const float* __restrict__ w, int steps, float (&a)[NR]) {
#pragma unroll
    for (int r = 0; r < NR; ++r) a[r] = w[(int64_t)min(r, steps - 1) * 64];
}
template <int MAXR>
__device__ __forceinline__ void gcp_vmm_regs_pre(const float (&a)[MAXR], int steps, const gcp_xyz_acc& in, gcp_xyz_acc& out) {
    static_assert(MAXR % 4 == 0, "steps come in groups of four registers");
#pragma unroll
    for (int r0 = 0; r0 < MAXR; r0 += 4)
        if (r0 < steps) {
#pragma unroll
            for (int r = r0; r < r0 + 4; ++r)
#pragma unroll
                for (int d = 0; d < 3; ++d) out[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], in[d][r], out[d], 0, 0, 0);
        }
}
template <int NR>
__device__ __forceinline__ void gcp_vmm_arr_pre(const float (&a)[NR], int steps, const float (&in)[3][NR], gcp_xyz_acc& out) {
    static_assert(NR % 4 == 0, "steps come in groups of four registers");
#pragma unroll
    for (int r0 = 0; r0 < NR; r0 += 4)
        if (r0 < steps) {
#pragma unroll
            for (int r = r0; r < r0 + 4; ++r)
#pragma unroll
                for (int d = 0; d < 3; ++d) out[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], in[d][r], out[d], 0, 0, 0);
        }
}

// gcp_vmm_down with fragments requested earlier (gcp_vmm_frags<MAXS>(wa, steps, a)).
template <int MAXS>
__device__ __forceinline__ void gcp_vmm_down_pre(const float (&a)[MAXS], int steps, int vi, const float* vrow, int hi, gcp_xyz_acc& u) {
    gcp_xyz_zero(u);
#pragma unroll
    for (int s0 = 0; s0 < MAXS; s0 += 2)
        if (s0 < steps) {  // wave-uniform, one test per two steps (a step past the end multiplies by a zero weight)
#pragma unroll
            for (int s = s0; s < s0 + 2 && s < MAXS; ++s) {
                const float* b = vrow + 3 * min(2 * s + hi, vi - 1);  // (an odd vi pads with a zero weight)
                const float as = s < steps ? a[s] : 0.f;
#pragma unroll
                for (int d = 0; d < 3; ++d) u[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(as, b[d], u[d], 0, 0, 0);
            }
        }
}
