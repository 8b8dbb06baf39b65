// Weight-gradient GEMMs on gfx950: out[m, n] = sum_r A[r, m] * B[r, n], a "TN" GEMM whose reduction axis (rows =
// edges or nodes, 1e4..1e6) is huge and whose output (a weight matrix, <= 1024 x 1100) is small.
//
// The reference gets these from autograd through nn.Linear (src/models/components/gcpnet.py:303-324); here they
// are explicit.  Decomposition: the row axis is split across workgroups (1024 rows each); a workgroup of 4 waves
// owns a 128 x 160 block of the output, wave w holding m-tile w and up to five 32x32 fp32 accumulators
// (v_mfma_f32_32x32x2_f32, reduction over row pairs).  Both operands are staged through LDS 32 rows at a time in
// row-major order, which is bank-conflict free for both the A and the B fragment reads.  Operands are
// concatenations of gathered sources (so [h_row | e | h_col | norms | frame scalars] is never materialised), can be
// passed through an activation on load, and can carry a column of ones (bias gradients).  Per-split partial sums
// go to scratch and a second kernel reduces them in a fixed order (deterministic; no atomics).
#include "common.h"

namespace {

constexpr int TN_BM = 128, TN_BN = 160, TN_RK = 32, TN_ROWS_PER_SPLIT = 1024;
constexpr int TN_LDA = TN_BM + 1, TN_LDB = TN_BN + 1;  // +1: two row-halves of a fragment read never collide

struct TnArgs {
    int n;
    gcp_tn_problem_t p[GCP_TN_MAX_PROBLEMS];
    int M[GCP_TN_MAX_PROBLEMS], N[GCP_TN_MAX_PROBLEMS];
    int mb[GCP_TN_MAX_PROBLEMS], nb[GCP_TN_MAX_PROBLEMS];
    int block_start[GCP_TN_MAX_PROBLEMS + 1];
};

__host__ __device__ inline int operand_width(const gcp_operand_t& o) {
    int w = o.ones ? 1 : 0;
    for (int k = 0; k < o.n; ++k) w += o.dim[k];
    return w;
}

// Stage columns [c0, c0 + width) of rows [r0, r0 + 32) of an operand into S[32][LD].
__device__ __forceinline__ void stage_operand(const gcp_operand_t& op, int c0, int width, int r0, int rows, float* S,
                                              int LD, int wave, int lane) {
    for (int rr = wave; rr < TN_RK; rr += 4) {
        const int r = r0 + rr;
        const bool valid = r < rows;
        float* dst = S + rr * LD;
        int cbase = 0;
        for (int sg = 0; sg < op.n; ++sg) {
            const int dim = op.dim[sg];
            const int lo = max(c0, cbase), hi = min(c0 + width, cbase + dim);
            if (lo < hi) {
                if (valid) {
                    const int64_t src = op.idx[sg] ? (int64_t)op.idx[sg][r] : (int64_t)r;
                    const float* rowp = op.ptr[sg] + src * op.ld[sg] - cbase;
                    for (int c = lo + lane; c < hi; c += GCP_WAVE) dst[c - c0] = gcp_act(op.act, rowp[c], op.slope);
                } else {
                    for (int c = lo + lane; c < hi; c += GCP_WAVE) dst[c - c0] = 0.f;
                }
            }
            cbase += dim;
        }
        if (op.ones) {
            if (cbase >= c0 && cbase < c0 + width && lane == 0) dst[cbase - c0] = valid ? 1.f : 0.f;
            cbase += 1;
        }
        for (int c = max(cbase, c0) + lane; c < c0 + width; c += GCP_WAVE) dst[c - c0] = 0.f;
    }
}

__global__ __launch_bounds__(256) void tn_gemm_kernel(TnArgs a) {
    __shared__ float As[TN_RK * TN_LDA];
    __shared__ float Bs[TN_RK * TN_LDB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int col = lane & 31, hi = lane >> 5;
    int pi = 0;
    while (pi + 1 < a.n && (int)blockIdx.x >= a.block_start[pi + 1]) ++pi;
    const gcp_tn_problem_t& P = a.p[pi];
    const int M = a.M[pi], N = a.N[pi];
    int b = blockIdx.x - a.block_start[pi];
    const int split = b % P.splits; b /= P.splits;
    const int nbi = b % a.nb[pi];
    const int mbi = b / a.nb[pi];
    const int m0 = mbi * TN_BM, n0 = nbi * TN_BN;
    const int mw = min(TN_BM, M - m0), nw = min(TN_BN, N - n0);
    const int ntiles = gcp_cdiv(nw, 32);
    const bool wave_active = wave * 32 < mw;
    const int r_begin = split * TN_ROWS_PER_SPLIT;
    const int r_end = min(P.rows, r_begin + TN_ROWS_PER_SPLIT);

    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int r0 = r_begin; r0 < r_end; r0 += TN_RK) {
        stage_operand(P.a, m0, gcp_round_up(mw, 32), r0, r_end, As, TN_LDA, wave, lane);
        stage_operand(P.b, n0, ntiles * 32, r0, r_end, Bs, TN_LDB, wave, lane);
        __syncthreads();
        if (wave_active) {
#pragma unroll 4
            for (int ss = 0; ss < TN_RK / 2; ++ss) {
                const float av = As[(2 * ss + hi) * TN_LDA + wave * 32 + col];
                const float* brow = Bs + (2 * ss + hi) * TN_LDB + col;
#pragma unroll
                for (int t = 0; t < 5; ++t)
                    if (t < ntiles) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, brow[32 * t], acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if (!wave_active) return;
    float* part = P.partial + (int64_t)split * M * N;
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        if (t < ntiles) {
            const int n = n0 + 32 * t + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wave * 32 + gcp_crow(r, hi);
                if (m < M && n < N) part[(int64_t)m * N + n] = acc[t][r];
            }
        }
    }
}

__global__ void tn_reduce_kernel(TnArgs a) {
    const int pi = blockIdx.y;
    const gcp_tn_problem_t& P = a.p[pi];
    const int M = a.M[pi], N = a.N[pi];
    const int64_t total = (int64_t)M * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < P.splits; ++k) s += P.partial[(int64_t)k * total + i];
        const int m = (int)(i / N), n = (int)(i % N);
        P.out[m * P.out_sm + n * P.out_sn] = s;
    }
}

}  // namespace

extern "C" int gcpnet_tn_splits(int rows, int M, int N) {
    (void)M; (void)N;
    return rows <= 0 ? 1 : gcp_cdiv(rows, TN_ROWS_PER_SPLIT);
}

extern "C" int gcpnet_tn_gemm(int n_problems, const gcp_tn_problem_t* problems, void* stream) {
    if (n_problems <= 0 || n_problems > GCP_TN_MAX_PROBLEMS || !problems) return GCPNET_E_BADARG;
    TnArgs a;
    a.n = n_problems;
    int blocks = 0, max_mn = 0;
    for (int i = 0; i < n_problems; ++i) {
        const gcp_tn_problem_t& P = problems[i];
        if (P.rows < 0 || P.a.n < 0 || P.a.n > GCP_TN_MAX_SEG || P.b.n < 0 || P.b.n > GCP_TN_MAX_SEG || !P.out || !P.partial)
            return GCPNET_E_BADARG;
        if (P.splits != gcpnet_tn_splits(P.rows, 0, 0)) return GCPNET_E_BADARG;
        a.p[i] = P;
        a.M[i] = operand_width(P.a);
        a.N[i] = operand_width(P.b);
        if (a.M[i] <= 0 || a.N[i] <= 0) return GCPNET_E_BADARG;
        a.mb[i] = gcp_cdiv(a.M[i], TN_BM);
        a.nb[i] = gcp_cdiv(a.N[i], TN_BN);
        a.block_start[i] = blocks;
        blocks += a.mb[i] * a.nb[i] * P.splits;
        max_mn = max(max_mn, a.M[i] * a.N[i]);
    }
    a.block_start[n_problems] = blocks;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(tn_gemm_kernel, dim3(blocks), dim3(256), 0, st, a);
    GCP_HIP_CHECK_LAUNCH();
    const int rblocks = min(256, gcp_cdiv(max_mn, 256));
    hipLaunchKernelGGL(tn_reduce_kernel, dim3(rblocks, n_problems), dim3(256), 0, st, a);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}
