// Small HBM-bound pieces around the GCP kernels: dropout (components/__init__.py:97-135), fused multi-tensor Adam.
#include "common.h"

namespace {

// Counter-based uniform in [0, 1): a 64-bit mix (splitmix64 finaliser) of (seed, group index).  The mask is a pure function of
// (seed, index), so the backward recomputes it instead of storing it, and results do not depend on the launch geometry.
__device__ __forceinline__ float gcp_uniform(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// y[g * group + j] = keep(g) ? x[..] / keep_prob : 0, keep(g) = uniform(seed, g) < keep_prob.  group = 1: nn.Dropout on the
// scalars; group = 3: VectorDropout, one Bernoulli draw per 3-vector (components/__init__.py:113-114).
__global__ __launch_bounds__(256) void dropout_kernel(int64_t n_groups, int group, const float* __restrict__ x, float keep_prob,
                                                      uint64_t seed, float* __restrict__ y) {
    const float inv = 1.0f / keep_prob;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < n_groups; g += (int64_t)gridDim.x * 256) {
        const float m = gcp_uniform(seed, (uint64_t)g) < keep_prob ? inv : 0.f;
        for (int j = 0; j < group; ++j) y[g * group + j] = x[g * group + j] * m;
    }
}

struct AdamArgs {
    gcp_adam_tensor_t t[GCP_ADAM_MAX_TENSORS];
    int n;
    float lr, beta1, beta2, eps, weight_decay, bias1, bias2;  // bias_k = 1 - beta_k^step
};

// torch.optim.Adam (amsgrad = False, maximize = False), one launch for up to GCP_ADAM_MAX_TENSORS parameter tensors:
// blockIdx.y = tensor.  (The reference trains with Adam, configs/model/gcpnet_*.yaml: optimizer; a GCPNet layer has ~90
// small parameter tensors, i.e. hundreds of tiny launches per step through the per-tensor path.)
__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
    const gcp_adam_tensor_t& T = a.t[blockIdx.y];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < T.n; i += (int64_t)gridDim.x * 256) {
        float g = T.grad[i];
        const float p = T.param[i];
        if (a.weight_decay != 0.f) g = fmaf(a.weight_decay, p, g);
        const float m = a.beta1 * T.exp_avg[i] + (1.f - a.beta1) * g;
        const float v = a.beta2 * T.exp_avg_sq[i] + (1.f - a.beta2) * g * g;
        T.exp_avg[i] = m;
        T.exp_avg_sq[i] = v;
        const float denom = sqrtf(v) / sqrtf(a.bias2) + a.eps;
        T.param[i] = p - (a.lr / a.bias1) * (m / denom);
    }
}

// The same with the step count in DEVICE memory (a captured step: hipGraph replays would freeze a host-side count and its bias
// corrections): every thread reads *step, which adam_step_inc_kernel -- launched behind the last adam launch of the call -- advances.
__global__ __launch_bounds__(256) void adam_dev_kernel(AdamArgs a, const int64_t* __restrict__ step) {
    const gcp_adam_tensor_t& T = a.t[blockIdx.y];
    const double st = (double)(*step + 1);
    const float bias1 = (float)(1.0 - pow((double)a.beta1, st)), bias2 = (float)(1.0 - pow((double)a.beta2, st));
    const float rs2 = 1.f / sqrtf(bias2), lr1 = a.lr / bias1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < T.n; i += (int64_t)gridDim.x * 256) {
        float g = T.grad[i];
        const float p = T.param[i];
        if (a.weight_decay != 0.f) g = fmaf(a.weight_decay, p, g);
        const float m = a.beta1 * T.exp_avg[i] + (1.f - a.beta1) * g;
        const float v = a.beta2 * T.exp_avg_sq[i] + (1.f - a.beta2) * g * g;
        T.exp_avg[i] = m;
        T.exp_avg_sq[i] = v;
        const float denom = sqrtf(v) * rs2 + a.eps;
        T.param[i] = p - lr1 * (m / denom);
    }
}
__global__ void adam_step_inc_kernel(int64_t* step) { *step += 1; }

// y = act(x) / dx = g * act'(x), element-wise (models/__init__.py:42-57); used where an activation sits between separately launched
// pieces (the frame-gate path of GCP2, components/gcpnet.py:369-384).
__global__ __launch_bounds__(256) void act_kernel(int64_t n, const float* __restrict__ x, const float* __restrict__ g, int act, float slope,
                                                  float* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        y[i] = g ? g[i] * gcp_act_grad(act, x[i], slope) : gcp_act(act, x[i], slope);
}

// Frame gate of GCP2 (components/gcpnet.py:369-384 with vectorize, components/__init__.py:329-378, which is linear in the frame:
// for node rows the mean out-edge frame does what gather / vectorize / scatter-mean do).  Per row, with g the 9 gate scalars,
// F the row's frame, Wf = vector_up_frames.weight [vo, 3]:
//   gv[c, :] = sum_a g[3 c + a] F[a, :];  gvr[o, :] = sum_c Wf[o, c] gv[c, :];  n[o] = sqrt(|gvr[o]|^2 + 1e-8) + 1e-8;
//   out[o, :] = vu[o, :] * act_v(n[o]).
__device__ __forceinline__ void fg_gv(const float* g, const float* F, float (&gv)[3][3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int d = 0; d < 3; ++d) gv[c][d] = g[3 * c] * F[d] + g[3 * c + 1] * F[3 + d] + g[3 * c + 2] * F[6 + d];
}

__global__ __launch_bounds__(256) void frame_gate_fwd_kernel(int64_t rows, int vo, const float* __restrict__ g, int ldg,
                                                             const float* __restrict__ frames, const float* __restrict__ wf,
                                                             const float* __restrict__ vu, int act, float slope, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * vo; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / vo;
        const int o = (int)(i - r * vo);
        float gv[3][3];
        fg_gv(g + r * ldg, frames + r * 9, gv);
        float q[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) q[d] = wf[3 * o] * gv[0][d] + wf[3 * o + 1] * gv[1][d] + wf[3 * o + 2] * gv[2][d];
        const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + 1e-8f) + 1e-8f;
        const float a = gcp_act(act, n, slope);
#pragma unroll
        for (int d = 0; d < 3; ++d) out[3 * i + d] = vu[3 * i + d] * a;
    }
}

// One thread per row: d vu, d g (row-local sums over the output channels), and this wave's share of d Wf [vo, 3] in
// part[wave, vo * 3] (wave-level butterfly; summed over waves by gcpnet_reduce_partials).
__global__ __launch_bounds__(256) void frame_gate_bwd_kernel(int64_t rows, int vo, const float* __restrict__ g, int ldg,
                                                             const float* __restrict__ frames, const float* __restrict__ wf,
                                                             const float* __restrict__ vu, int act, float slope,
                                                             const float* __restrict__ d_out, float* __restrict__ d_vu,
                                                             float* __restrict__ d_g, float* __restrict__ part) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool ok = r < rows;
    const int64_t rc = ok ? r : 0;
    float gv[3][3], dgv[3][3];
    fg_gv(g + rc * ldg, frames + rc * 9, gv);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int d = 0; d < 3; ++d) dgv[c][d] = 0.f;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    for (int o = 0; o < vo; ++o) {
        const float w0 = wf[3 * o], w1 = wf[3 * o + 1], w2 = wf[3 * o + 2];
        float q[3], dq[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) q[d] = w0 * gv[0][d] + w1 * gv[1][d] + w2 * gv[2][d];
        const float rs = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + 1e-8f), n = rs + 1e-8f;
        const float a = gcp_act(act, n, slope), da_dn = gcp_act_grad(act, n, slope);
        const int64_t j = (rc * vo + o) * 3;
        float da = 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float go = ok ? d_out[j + d] : 0.f;
            if (ok) d_vu[j + d] = go * a;
            da += go * vu[j + d];
        }
        const float coef = da * da_dn / rs;
        float dw[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            dq[d] = coef * q[d];
            dgv[0][d] += w0 * dq[d]; dgv[1][d] += w1 * dq[d]; dgv[2][d] += w2 * dq[d];
            dw[0] += dq[d] * gv[0][d]; dw[1] += dq[d] * gv[1][d]; dw[2] += dq[d] * gv[2][d];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = dw[c];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
            if ((threadIdx.x & 63) == 0) part[wave * vo * 3 + 3 * o + c] = v;
        }
    }
    if (ok) {
        const float* F = frames + r * 9;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int a = 0; a < 3; ++a) d_g[r * ldg + 3 * c + a] = dgv[c][0] * F[3 * a] + dgv[c][1] * F[3 * a + 1] + dgv[c][2] * F[3 * a + 2];
        for (int k = 9; k < ldg; ++k) d_g[r * ldg + k] = 0.f;
    }
}

}  // namespace

extern "C" int gcpnet_activation(int64_t n, const float* x, const float* grad, int act, float slope, float* y, void* stream) {
    if (n < 0 || !x || !y || act < 0 || act > GCP_ACT_SIGMOID) return GCPNET_E_BADARG;
    if (n == 0) return 0;
    const int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(act_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, (hipStream_t)stream, n, x, grad, act, slope, y);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int gcpnet_frame_gate_bwd_parts(int64_t rows) { return (int)((((rows + 255) / 256) * 256) / 64); }

extern "C" int gcpnet_frame_gate_forward(int64_t rows, int vo, const float* g, int ldg, const float* frames, const float* w_up_frames,
                                         const float* vu, int act, float slope, float* out, void* stream) {
    if (rows < 0 || vo < 1 || ldg < 9 || !g || !frames || !w_up_frames || !vu || !out) return GCPNET_E_BADARG;
    if (rows == 0) return 0;
    const int64_t nb = (rows * vo + 255) / 256;
    hipLaunchKernelGGL(frame_gate_fwd_kernel, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, (hipStream_t)stream, rows, vo, g, ldg,
                       frames, w_up_frames, vu, act, slope, out);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int gcpnet_frame_gate_backward(int64_t rows, int vo, const float* g, int ldg, const float* frames, const float* w_up_frames,
                                          const float* vu, int act, float slope, const float* d_out, float* d_vu, float* d_g, float* part,
                                          void* stream) {
    if (rows < 0 || vo < 1 || ldg < 9 || !g || !frames || !w_up_frames || !vu || !d_out || !d_vu || !d_g || !part) return GCPNET_E_BADARG;
    if (rows == 0) return 0;
    hipLaunchKernelGGL(frame_gate_bwd_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, vo, g, ldg,
                       frames, w_up_frames, vu, act, slope, d_out, d_vu, d_g, part);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int gcpnet_dropout(int64_t n_groups, int group, const float* x, float keep_prob, uint64_t seed, float* y, void* stream) {
    if (n_groups < 0 || group < 1 || !x || !y || !(keep_prob > 0.f) || keep_prob > 1.f) return GCPNET_E_BADARG;
    if (n_groups == 0) return 0;
    const int64_t nb = (n_groups + 255) / 256;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, (hipStream_t)stream, n_groups, group, x,
                       keep_prob, seed, y);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

// ---- many small strided 2-D copies / fills in ONE launch: the glue around the kernels of a block -- column slices and padded /
// transposed forms of the small vector weights, the pieces of an assembled weight gradient -- is a dozen ~4 us launches per block as
// ATen cat / pad / clone / copy_; blockIdx.y = job ------------------------------------------------------------------------------
struct Copy2dArgs {
    gcp_copy2d_job_t j[GCP_COPY2D_MAX_JOBS];
};
__global__ __launch_bounds__(256) void copy2d_multi_kernel(Copy2dArgs a) {
    const gcp_copy2d_job_t& J = a.j[blockIdx.y];
    const int64_t n = (int64_t)J.rows * J.cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / J.cols, c = i - r * J.cols;
        J.dst[r * J.dst_rs + c * J.dst_cs] = J.src ? J.src[r * J.src_rs + c * J.src_cs] : 0.f;
    }
}

extern "C" int gcpnet_copy2d_multi(int n, const gcp_copy2d_job_t* jobs, void* stream) {
    if (n < 0 || (n > 0 && !jobs)) return GCPNET_E_BADARG;
    for (int i0 = 0; i0 < n; i0 += GCP_COPY2D_MAX_JOBS) {
        Copy2dArgs a;
        const int m = n - i0 < GCP_COPY2D_MAX_JOBS ? n - i0 : GCP_COPY2D_MAX_JOBS;
        int64_t nmax = 0;
        for (int k = 0; k < m; ++k) {
            a.j[k] = jobs[i0 + k];
            if (!a.j[k].dst || a.j[k].rows < 0 || a.j[k].cols < 0) return GCPNET_E_BADARG;
            const int64_t e = (int64_t)a.j[k].rows * a.j[k].cols;
            nmax = e > nmax ? e : nmax;
        }
        if (nmax == 0) continue;
        const int64_t nb = (nmax + 255) / 256;
        hipLaunchKernelGGL(copy2d_multi_kernel, dim3((unsigned)(nb < 256 ? nb : 256), m), dim3(256), 0, (hipStream_t)stream, a);
        GCP_HIP_CHECK_LAUNCH();
    }
    return 0;
}

// adjoint of y = a + clamp(alpha b, lo, hi) w.r.t. b (the position update, components/gcpnet.py:1156-1158): alpha g inside the clamp
__global__ __launch_bounds__(256) void axpy_clamp_bwd_kernel(int64_t n, const float* __restrict__ g, const float* __restrict__ b, float alpha,
                                                             int clamp, float lo, float hi, float* __restrict__ gb) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float u = b[i] * alpha;
        gb[i] = (!clamp || (u >= lo && u <= hi)) ? g[i] * alpha : 0.f;
    }
}

extern "C" int gcpnet_axpy_clamp_backward(int64_t n, const float* g, const float* b, float alpha, int clamp, float lo, float hi, float* gb,
                                          void* stream) {
    if (n < 0 || (n > 0 && (!g || !b || !gb))) return GCPNET_E_BADARG;
    if (n == 0) return 0;
    const int64_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(axpy_clamp_bwd_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, (hipStream_t)stream, n, g, b, alpha, clamp,
                       lo, hi, gb);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

static int adam_launch(int n, const gcp_adam_tensor_t* tensors, float lr, float beta1, float beta2, float eps, float weight_decay,
                       int step, int64_t* step_dev, void* stream);

extern "C" int gcpnet_adam_step(int n, const gcp_adam_tensor_t* tensors, float lr, float beta1, float beta2, float eps,
                                float weight_decay, int step, void* stream) {
    if (step < 1) return GCPNET_E_BADARG;
    return adam_launch(n, tensors, lr, beta1, beta2, eps, weight_decay, step, nullptr, stream);
}

extern "C" int gcpnet_adam_step_dev(int n, const gcp_adam_tensor_t* tensors, float lr, float beta1, float beta2, float eps,
                                    float weight_decay, int64_t* step_dev, void* stream) {
    if (!step_dev) return GCPNET_E_BADARG;
    if (int rc = adam_launch(n, tensors, lr, beta1, beta2, eps, weight_decay, 1, step_dev, stream)) return rc;
    hipLaunchKernelGGL(adam_step_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

static int adam_launch(int n, const gcp_adam_tensor_t* tensors, float lr, float beta1, float beta2, float eps, float weight_decay,
                       int step, int64_t* step_dev, void* stream) {
    if (n < 0 || (n > 0 && !tensors)) return GCPNET_E_BADARG;
    for (int i0 = 0; i0 < n; i0 += GCP_ADAM_MAX_TENSORS) {
        AdamArgs a;
        a.n = n - i0 < GCP_ADAM_MAX_TENSORS ? n - i0 : GCP_ADAM_MAX_TENSORS;
        int64_t nmax = 0;
        for (int k = 0; k < a.n; ++k) {
            a.t[k] = tensors[i0 + k];
            if (!a.t[k].param || !a.t[k].grad || !a.t[k].exp_avg || !a.t[k].exp_avg_sq || a.t[k].n < 0) return GCPNET_E_BADARG;
            nmax = a.t[k].n > nmax ? a.t[k].n : nmax;
        }
        a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
        // (in double, as torch.optim.Adam computes them: 1 - 0.999^step in fp32 is off by ~1e-4 relative at small step counts)
        a.bias1 = (float)(1.0 - pow((double)beta1, (double)step));
        a.bias2 = (float)(1.0 - pow((double)beta2, (double)step));
        if (nmax == 0) continue;
        const int64_t nb = (nmax + 255) / 256;
        if (step_dev) hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)(nb < 64 ? nb : 64), a.n), dim3(256), 0, (hipStream_t)stream, a, step_dev);
        else hipLaunchKernelGGL(adam_kernel, dim3((unsigned)(nb < 64 ? nb : 64), a.n), dim3(256), 0, (hipStream_t)stream, a);
        GCP_HIP_CHECK_LAUNCH();
    }
    return 0;
}
