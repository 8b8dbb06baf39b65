// Does the operand split (VALU) run UNDER the bf16 MFMAs of the same wave?  One wave per SIMD (4 waves / workgroup, 1 workgroup / CU),
// a 2 x 5 tile step as in tn_gemm.hip (60 MFMAs, 7 fragment splits of 44 VALU instructions) repeated ITER times on LDS data.
//   mode 0: MFMAs only    1: splits only    2: both, split units between the MFMAs    3: both, all splits first, then all MFMAs
//   mode 4: as 2 but the split units only after every SECOND MFMA pair (bigger VALU runs)
//   hipcc --offload-arch=gfx950 -O3 -I../../gcpnet_amd/csrc -I../../include x3_overlap.hip -o x3_overlap && ./x3_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "gcp_bf16x3.h"

__device__ __forceinline__ void pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned hp = gcp_bf16_pack_hi(a, b);
    const float ra = a - __uint_as_float(hp << 16), rb = b - __uint_as_float(hp & 0xffff0000u);
    const unsigned mp = gcp_bf16_pack_hi(ra, rb);
    const float sa = ra - __uint_as_float(mp << 16), sb = rb - __uint_as_float(mp & 0xffff0000u);
    h = hp; m = mp; l = gcp_bf16_pack_hi(sa, sb);
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* in, float* out, unsigned long long* cyc, int iters) {
    __shared__ float L[32 * 288];
    const int tid = threadIdx.x, lane = tid & 63, col = lane & 31, hi = lane >> 5;
    for (int i = tid; i < 32 * 288; i += 256) L[i] = in[i];
    __syncthreads();
    f32x16 acc[2][5];
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 5; ++i) for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
    const float* As = L + 8 * hi * 288 + col;
    const float* Bs = L + 8 * hi * 288 + 128 + col;
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        gcp_u32x4 a3[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = As[q * 288 + 32 * j + (it & 1)];
            gcp_bf16x3_split8(x, a3[j][0], a3[j][1], a3[j][2]);
        }
        float y[2][8];
        gcp_u32x4 c3[2][3];
#pragma unroll
        for (int q = 0; q < 8; ++q) y[0][q] = Bs[q * 288];
#pragma unroll
        for (int q = 0; q < 8; ++q) y[1][q] = Bs[q * 288 + 32];
        gcp_bf16x3_split8(y[0], c3[0][0], c3[0][1], c3[0][2]);
        if (MODE == 5 || MODE == 6) {  // term-major over all ten accumulators; 6: with every split (up front)
            gcp_u32x4 all[5][3];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                if (MODE == 6 || i == 0) {
                    float z[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) z[q] = Bs[q * 288 + 32 * i];
                    gcp_bf16x3_split8(z, all[i][0], all[i][1], all[i][2]);
                } else {
#pragma unroll
                    for (int t = 0; t < 3; ++t) all[i][t] = all[0][t];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[j][i] = gcp_mfma_bf16(a3[j][TA[p]], all[i][TB[p]], acc[j][i]);
            __builtin_amdgcn_sched_barrier(0);
            continue;
        }
        if (MODE == 3) {  // all splits, then all MFMAs
            gcp_u32x4 all[5][3];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                float z[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) z[q] = Bs[q * 288 + 32 * i];
                gcp_bf16x3_split8(z, all[i][0], all[i][1], all[i][2]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[j][i] = gcp_mfma_bf16(a3[j][TA[p]], all[i][TB[p]], acc[j][i]);
            __builtin_amdgcn_sched_barrier(0);
            continue;
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int cb = i & 1, nb = cb ^ 1;
            if (i + 2 < 5) {
#pragma unroll
                for (int q = 0; q < 8; ++q) y[cb][q] = Bs[q * 288 + 32 * (i + 2)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                if (MODE != 1) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[j][i] = gcp_mfma_bf16(a3[j][TA[p]], c3[MODE == 0 ? 0 : cb][TB[p]], acc[j][i]);
                }
                if (MODE != 0 && i + 1 < 5) {
                    if (MODE == 4) {
                        if (p == 1 || p == 3) {
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                const int q = (p - 1) + u;
                                unsigned th, tm, tl;
                                pair(y[nb][2 * q], y[nb][2 * q + 1], th, tm, tl);
                                c3[nb][0][q] = th; c3[nb][1][q] = tm; c3[nb][2][q] = tl;
                            }
                        }
                    } else if (p >= 1 && p < 5) {
                        unsigned th, tm, tl;
                        pair(y[nb][2 * (p - 1)], y[nb][2 * (p - 1) + 1], th, tm, tl);
                        c3[nb][0][p - 1] = th; c3[nb][1][p - 1] = tm; c3[nb][2][p - 1] = tl;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MODE == 1) {  // keep the splits alive
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[0][0][t] += __uint_as_float(c3[0][t][0] ^ c3[1][t][1] ^ a3[0][t][2] ^ a3[1][t][3]);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 5; ++i) for (int r = 0; r < 16; ++r) s += acc[j][i][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* what, const float* in, float* out, unsigned long long* cyc, int blocks) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= blocks;
    printf("%-44s %8.1f cycles / step (60 MFMAs = 1920)   %7.3f ms wall -> %.2f GHz\n", what, avg / iters, ms, avg / (ms * 1e6));
}

int main() {
    const int blocks = 256;
    float *in, *out;
    unsigned long long* cyc;
    hipMalloc(&in, 32 * 288 * 4); hipMalloc(&out, 2 * blocks * 256 * 4); hipMalloc(&cyc, 2 * blocks * 8);
    std::vector<float> h(32 * 288);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0>("MFMAs only", in, out, cyc, blocks);
    run<1>("splits only", in, out, cyc, blocks);
    run<2>("both, one split unit after each MFMA pair", in, out, cyc, blocks);
    run<4>("both, two split units after every second pair", in, out, cyc, blocks);
    run<3>("both, all splits then all MFMAs", in, out, cyc, blocks);
    run<5>("MFMAs only, term-major over 10 accumulators", in, out, cyc, blocks);
    run<6>("all splits, then term-major MFMAs", in, out, cyc, blocks);
    printf("two workgroups per CU (two waves per SIMD):\n");
    run<0>("MFMAs only", in, out, cyc, 2 * blocks);
    run<2>("both, one split unit after each MFMA pair", in, out, cyc, 2 * blocks);
    run<5>("MFMAs only, term-major over 10 accumulators", in, out, cyc, 2 * blocks);
    run<6>("all splits, then term-major MFMAs", in, out, cyc, 2 * blocks);
    return 0;
}
