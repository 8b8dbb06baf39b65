// Input side of the hot path: the NMS featuriser and the radius-graph builder, on the GPU.
//   * edge features  (src/datamodules/components/nms_dataset.py:23-45, helper.py:16-45): e = [edge_attr | 16 RBF(|x_row - x_col|)],
//     xi = unit(x_row - x_col), nan_to_num'ed;
//   * node features  (nms_dataset.py:48-61, helper.py:52-59): h = |vel|, chi = [vel, forward orientation, backward orientation],
//     orientations along the node order of each graph;
//   * radius graph   (atom3d_dataset.py:110-112 recipe: r = 4.5, at most K neighbours, no self loops): for every target node the K
//     nearest nodes of the same graph within r, in ascending distance -- the order gcpnet_amd.synthetic.radius_graph (scipy
//     cKDTree) produces, so edge lists compare bit for bit.  Cell list: nodes are sorted by (graph, cell) on the host side of the
//     ABI (index preprocessing, like the CSR plans); this kernel scans the 27 cells around a node.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void nms_edge_features_kernel(int64_t E, const float* __restrict__ x, const int32_t* __restrict__ row,
                                                                const int32_t* __restrict__ col, const float* __restrict__ attr, int A,
                                                                float d_max, int n_rbf, float* __restrict__ e_out,
                                                                float* __restrict__ xi_out) {
    const int W = A + n_rbf;
    const float sigma = d_max / (float)n_rbf;  // (D_max - D_min) / D_count with D_min = 0
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < E; i += (int64_t)gridDim.x * 256) {
        const int r = row[i], c = col[i];
        const float dx = x[3 * r] - x[3 * c], dy = x[3 * r + 1] - x[3 * c + 1], dz = x[3 * r + 2] - x[3 * c + 2];
        const float d = sqrtf(dx * dx + dy * dy + dz * dz);
        float* eo = e_out + i * W;
        for (int a = 0; a < A; ++a) {
            const float v = attr[i * A + a];
            eo[a] = isnan(v) ? 0.f : v;
        }
        for (int k = 0; k < n_rbf; ++k) {
            const float mu = n_rbf > 1 ? d_max * (float)k / (float)(n_rbf - 1) : 0.f;  // torch.linspace(0, D_max, D_count)
            const float z = (d - mu) / sigma;
            eo[A + k] = __expf(-z * z);
        }
        // _normalize + nan_to_num: 0 / 0 -> nan -> 0
        const float inv = d > 0.f ? 1.0f / d : 0.f;
        xi_out[3 * i] = dx * inv; xi_out[3 * i + 1] = dy * inv; xi_out[3 * i + 2] = dz * inv;
    }
}

__global__ __launch_bounds__(256) void nms_node_features_kernel(int64_t N, const float* __restrict__ vel, const float* __restrict__ x,
                                                                const int32_t* __restrict__ batch, float* __restrict__ h_out,
                                                                float* __restrict__ chi_out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        const float vx = vel[3 * i], vy = vel[3 * i + 1], vz = vel[3 * i + 2];
        h_out[i] = sqrtf(vx * vx + vy * vy + vz * vz);
        float* c = chi_out + 9 * i;
        c[0] = vx; c[1] = vy; c[2] = vz;
        const int g = batch ? batch[i] : 0;
        // forward: unit(X[i+1] - X[i]), zero for the last node of a graph; backward: unit(X[i-1] - X[i]), zero for the first
        const bool has_next = i + 1 < N && (batch ? batch[i + 1] : 0) == g;
        const bool has_prev = i > 0 && (batch ? batch[i - 1] : 0) == g;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bool ok = s == 0 ? has_next : has_prev;
            const int64_t j = s == 0 ? i + 1 : i - 1;
            float dx = 0.f, dy = 0.f, dz = 0.f;
            if (ok) { dx = x[3 * j] - x[3 * i]; dy = x[3 * j + 1] - x[3 * i + 1]; dz = x[3 * j + 2] - x[3 * i + 2]; }
            const float d = sqrtf(dx * dx + dy * dy + dz * dz);
            const float inv = d > 0.f ? 1.0f / d : 0.f;
            c[3 + 3 * s] = dx * inv; c[4 + 3 * s] = dy * inv; c[5 + 3 * s] = dz * inv;
        }
    }
}

__global__ __launch_bounds__(256) void orientations_kernel(int64_t N, const float* __restrict__ x, const int32_t* __restrict__ batch,
                                                           float* __restrict__ chi_out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        const int g = batch ? batch[i] : 0;
        const bool has_next = i + 1 < N && (batch ? batch[i + 1] : 0) == g;
        const bool has_prev = i > 0 && (batch ? batch[i - 1] : 0) == g;
        float* c = chi_out + 6 * i;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bool ok = s == 0 ? has_next : has_prev;
            const int64_t j = s == 0 ? i + 1 : i - 1;
            float dx = 0.f, dy = 0.f, dz = 0.f;
            if (ok) { dx = x[3 * j] - x[3 * i]; dy = x[3 * j + 1] - x[3 * i + 1]; dz = x[3 * j + 2] - x[3 * i + 2]; }
            const float d = sqrtf(dx * dx + dy * dy + dz * dz);
            const float inv = d > 0.f ? 1.0f / d : 0.f;
            c[3 * s] = dx * inv; c[3 * s + 1] = dy * inv; c[3 * s + 2] = dz * inv;
        }
    }
}

constexpr int RG_MAX_K = 64;

// One thread per target node (in cell-sorted order).  cell_start[c] .. cell_start[c + 1] are the sorted positions of the nodes
// of cell c (cells of all graphs concatenated: graph g owns cells [g * ncell, (g + 1) * ncell)).  Distances in double from the
// float coordinates: the same exact values, hence the same order, as scipy's cKDTree on the float32 array.
// FIRST: torch_cluster 1.6.0's selection instead of the nearest K (its CUDA kernel `radius_kernel`, csrc/cuda/radius_cuda.cu: for each
// target the candidates of its graph are visited in ascending INDEX order, a candidate with dist^2 < r^2 -- strict -- is taken, the
// walk stops at the cap; `radius_graph(loop=False)`, torch_cluster/radius.py, calls it with cap K + 1 WITH the node itself among the
// candidates and removes the self loop afterwards): the K lowest ids of {nodes within r, self included}, self dropped -- a node
// whose K + 1 lowest in-range ids are all below its own keeps K + 1 neighbours, as upstream does.  Here K is that cap (K + 1 of the
// caller) and the list is kept ascending by id.
template <bool FIRST>
__global__ __launch_bounds__(128) void radius_graph_kernel(int N, const float* __restrict__ xs /* sorted coordinates */,
                                                           const int32_t* __restrict__ order /* sorted position -> node id */,
                                                           const int32_t* __restrict__ cell_of /* sorted position -> global cell */,
                                                           const int32_t* __restrict__ cell_start, int nx, int ny, int nz, double r2,
                                                           int K, int32_t* __restrict__ nbr /* [N, K] by node id */,
                                                           int32_t* __restrict__ cnt /* [N] by node id */) {
    const int p = blockIdx.x * 128 + threadIdx.x;
    if (p >= N) return;
    const int ncell = nx * ny * nz;
    const int gc = cell_of[p], g = gc / ncell, lc = gc - g * ncell;
    const int cz = lc / (nx * ny), cy = (lc - cz * nx * ny) / nx, cx = lc - cz * nx * ny - cy * nx;
    const double px = xs[3 * p], py = xs[3 * p + 1], pz = xs[3 * p + 2];
    double bd[RG_MAX_K];
    int bi[RG_MAX_K];
    int n = 0;
    for (int dz = -1; dz <= 1; ++dz) {
        const int z = cz + dz;
        if (z < 0 || z >= nz) continue;
        for (int dy = -1; dy <= 1; ++dy) {
            const int y = cy + dy;
            if (y < 0 || y >= ny) continue;
            for (int dx = -1; dx <= 1; ++dx) {
                const int xq = cx + dx;
                if (xq < 0 || xq >= nx) continue;
                const int c = g * ncell + (z * ny + y) * nx + xq;
                for (int q = cell_start[c]; q < cell_start[c + 1]; ++q) {
                    if (!FIRST && q == p) continue;
                    const double ex = xs[3 * q] - px, ey = xs[3 * q + 1] - py, ez = xs[3 * q + 2] - pz;
                    double d2 = ex * ex + ey * ey + ez * ez;
                    if (FIRST) {
                        // torch_cluster 1.6.0's radius kernel decides in the coordinates' own precision: dist accumulated over x, y, z
                        // in float (`dist += d * d`, contracted to fused multiply-adds by its compiler) and compared with (float)(r * r)
                        // -- a pair within a float rounding error of the cutoff must fall on the same side here
                        const float fx = xs[3 * q] - xs[3 * p], fy = xs[3 * q + 1] - xs[3 * p + 1], fz = xs[3 * q + 2] - xs[3 * p + 2];
                        const float dist = __fmaf_rn(fz, fz, __fmaf_rn(fy, fy, fx * fx));
                        if (!(dist < (float)r2)) continue;
                    } else if (d2 > r2) continue;
                    const int id = order[q];
                    if (FIRST) d2 = (double)id;  // the sort key of this mode
                    // insertion into the ascending list (ties: lower node id first), keeping at most K
                    int pos = n < K ? n : K;
                    while (pos > 0 && (bd[pos - 1] > d2 || (bd[pos - 1] == d2 && bi[pos - 1] > id))) --pos;
                    if (pos >= K) continue;
                    const int last = n < K ? n : K - 1;
                    for (int t = last; t > pos; --t) { bd[t] = bd[t - 1]; bi[t] = bi[t - 1]; }
                    bd[pos] = d2; bi[pos] = id;
                    if (n < K) ++n;
                }
            }
        }
    }
    const int me = order[p];
    if (FIRST) {  // drop the self loop (if it made the list), keep the order
        int m = 0;
        for (int t = 0; t < n; ++t)
            if (bi[t] != me) bi[m++] = bi[t];
        n = m;
    }
    cnt[me] = n;
    for (int t = 0; t < K; ++t) nbr[(int64_t)me * K + t] = t < n ? bi[t] : -1;
}

}  // namespace

extern "C" int gcpnet_nms_edge_features(int64_t E, const float* x, const int32_t* row, const int32_t* col, const float* edge_attr,
                                        int n_attr, float d_max, int n_rbf, float* e_out, float* xi_out, void* stream) {
    if (E < 0 || !x || !row || !col || (n_attr > 0 && !edge_attr) || n_attr < 0 || n_rbf < 1 || !(d_max > 0.f) || !e_out || !xi_out)
        return GCPNET_E_BADARG;
    if (E == 0) return 0;
    const int64_t nb = (E + 255) / 256;
    hipLaunchKernelGGL(nms_edge_features_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, (hipStream_t)stream, E, x, row, col,
                       edge_attr, n_attr, d_max, n_rbf, e_out, xi_out);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int gcpnet_nms_node_features(int64_t N, const float* vel, const float* x, const int32_t* batch, float* h_out, float* chi_out,
                                        void* stream) {
    if (N < 0 || !vel || !x || !h_out || !chi_out) return GCPNET_E_BADARG;
    if (N == 0) return 0;
    const int64_t nb = (N + 255) / 256;
    hipLaunchKernelGGL(nms_node_features_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, (hipStream_t)stream, N, vel, x, batch,
                       h_out, chi_out);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int gcpnet_orientations(int64_t N, const float* x, const int32_t* batch, float* chi_out, void* stream) {
    if (N < 0 || !x || !chi_out) return GCPNET_E_BADARG;
    if (N == 0) return 0;
    const int64_t nb = (N + 255) / 256;
    hipLaunchKernelGGL(orientations_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, (hipStream_t)stream, N, x, batch, chi_out);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int gcpnet_radius_graph(int N, const float* x_sorted, const int32_t* order, const int32_t* cell_of, const int32_t* cell_start,
                                   int nx, int ny, int nz, double radius, int max_neighbors, int32_t* nbr, int32_t* count, void* stream) {
    if (N < 0 || !x_sorted || !order || !cell_of || !cell_start || nx < 1 || ny < 1 || nz < 1 || !(radius > 0.0) || max_neighbors < 1 ||
        max_neighbors > RG_MAX_K || !nbr || !count)
        return GCPNET_E_BADARG;
    if (N == 0) return 0;
    hipLaunchKernelGGL(radius_graph_kernel<false>, dim3((unsigned)((N + 127) / 128)), dim3(128), 0, (hipStream_t)stream, N, x_sorted, order,
                       cell_of, cell_start, nx, ny, nz, radius * radius, max_neighbors, nbr, count);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int gcpnet_radius_graph_first(int N, const float* x_sorted, const int32_t* order, const int32_t* cell_of,
                                         const int32_t* cell_start, int nx, int ny, int nz, double radius, int max_neighbors, int32_t* nbr,
                                         int32_t* count, void* stream) {
    if (N < 0 || !x_sorted || !order || !cell_of || !cell_start || nx < 1 || ny < 1 || nz < 1 || !(radius > 0.0) || max_neighbors < 1 ||
        max_neighbors + 1 > RG_MAX_K || !nbr || !count)
        return GCPNET_E_BADARG;
    if (N == 0) return 0;
    hipLaunchKernelGGL(radius_graph_kernel<true>, dim3((unsigned)((N + 127) / 128)), dim3(128), 0, (hipStream_t)stream, N, x_sorted, order,
                       cell_of, cell_start, nx, ny, nz, radius * radius, max_neighbors + 1, nbr, count);
    GCP_HIP_CHECK_LAUNCH();
    return 0;
}
