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

}  // namespace

extern "C" int gcpnet_dropout(int64_t n_groups, int group, const float* x, float keep_prob, uint64_t seed, float* y, void* stream) {
    if (n_groups < 0 || group < 1 || !x || !y || !(keep_prob > 0.f) || keep_prob > 1.f) return GCPNET_E_BADARG;
    if (n_groups == 0) return 0;
    const int64_t nb = (n_groups + 255) / 256;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, (hipStream_t)stream, n_groups, group, x,
                       keep_prob, seed, y);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int gcpnet_adam_step(int n, const gcp_adam_tensor_t* tensors, float lr, float beta1, float beta2, float eps,
                                float weight_decay, int step, void* stream) {
    if (n < 0 || (n > 0 && !tensors) || step < 1) return GCPNET_E_BADARG;
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
        a.bias1 = 1.f - powf(beta1, (float)step);
        a.bias2 = 1.f - powf(beta2, (float)step);
        if (nmax == 0) continue;
        const int64_t nb = (nmax + 255) / 256;
        hipLaunchKernelGGL(adam_kernel, dim3((unsigned)(nb < 64 ? nb : 64), a.n), dim3(256), 0, (hipStream_t)stream, a);
        GCP_HIP_CHECK_LAUNCH();
    }
    return 0;
}
