// Microbenchmark + accuracy check: an fp32 product C[32,32] = A[32,K] . B[K,32] on the bf16 matrix pipe, operands split into
// three bf16 terms each (x = h + m + l exactly, by truncation) and six products kept (hh, hm, mh, hl, lh, mm; the dropped
// ml, lm, ll are below 2^-23 relative), fp32 accumulation inside v_mfma_f32_32x32x16_bf16 -- against the exact-fp32
// v_mfma_f32_32x32x2_f32 chain and a float64 host reference.
//   hipcc --offload-arch=gfx950 -O3 bf16x6.hip -o bf16x6 && ./bf16x6
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_hi(float x0, float x1) {  // (bf16 trunc(x0), bf16 trunc(x1)) in one register
    return __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
}
// eight fp32 values -> three bf16x8 terms (truncation; x = h + m + l exactly unless the low term underflows)
__device__ __forceinline__ void split8(const float (&x)[8], u32x4& h, u32x4& m, u32x4& l) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = x[2 * j], b = x[2 * j + 1];
        const unsigned hp = pack_hi(a, b);
        const float ra = a - __uint_as_float(hp << 16), rb = b - __uint_as_float(hp & 0xffff0000u);
        const unsigned mp = pack_hi(ra, rb);
        const float sa = ra - __uint_as_float(mp << 16), sb = rb - __uint_as_float(mp & 0xffff0000u);
        h[j] = hp; m[j] = mp; l[j] = pack_hi(sa, sb);
    }
}
__device__ __forceinline__ f32x16 mm(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// one wave: C = A B with K a multiple of 16.  mode 0: fp32 MFMA, mode 1: bf16 x 6
__global__ __launch_bounds__(64) void gemm(const float* A, const float* B, float* C, int K, int mode) {
    const int lane = threadIdx.x, e = lane & 31, hi = lane >> 5;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (mode == 0) {
        for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[e * K + k + hi], B[(k + hi) * 32 + e], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            float a[8], b[8];
            for (int i = 0; i < 8; ++i) { a[i] = A[e * K + k + 8 * hi + i]; b[i] = B[(k + 8 * hi + i) * 32 + e]; }
            u32x4 ah, am, al, bh, bm, bl;
            split8(a, ah, am, al);
            split8(b, bh, bm, bl);
            acc = mm(al, bh, acc); acc = mm(ah, bl, acc); acc = mm(am, bm, acc);  // small terms first
            acc = mm(am, bh, acc); acc = mm(ah, bm, acc); acc = mm(ah, bh, acc);
        }
    }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + e] = acc[r];
}

// issue-rate: per K = 16 slab and NT output tiles, 8 fp32 MFMAs per tile vs 6 bf16 MFMAs per tile (+ the split of the B side)
template <int NT, int MODE>
__global__ __launch_bounds__(64) void rate(float* out, unsigned long long* t, int iters) {
    f32x16 acc[NT];
    for (int a = 0; a < NT; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 1.0f + threadIdx.x * 1e-3f + i * 0.37f;
    u32x4 wa = {0x3f803f80u, 0x3f813f82u, 0x3f833f84u, 0x3f853f86u};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int a = 0; a < NT; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[i], x[(i + 1) & 7], acc[a], 0, 0, 0);
        } else {
            u32x4 bh, bm, bl;
            split8(x, bh, bm, bl);
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                acc[a] = mm(wa, bh, acc[a]); acc[a] = mm(wa, bm, acc[a]); acc[a] = mm(wa, bl, acc[a]);
                acc[a] = mm(wa, bh, acc[a]); acc[a] = mm(wa, bm, acc[a]); acc[a] = mm(wa, bh, acc[a]);
            }
        }
        for (int i = 0; i < 8; ++i) x[i] += acc[0][i] * 1e-30f;  // (keeps the split inside the loop)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < NT; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

template <int NT, int MODE>
void run_rate(int waves_per_cu) {
    int blocks = 256 * waves_per_cu, iters = 2000;
    float* out; unsigned long long* t;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&t, blocks * 8);
    rate<NT, MODE><<<blocks, 64>>>(out, t, iters);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    rate<NT, MODE><<<blocks, 64>>>(out, t, iters);
    hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), t, blocks * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto v : h) sum += v;
    printf("%s, %d output tiles, %d waves/CU: %.0f ticks per K=16 slab per wave, kernel %.3f ms, %.1f fp32-equivalent TFLOP/s\n",
           MODE ? "bf16 x 6" : "fp32 mfma", NT, waves_per_cu, sum / blocks / iters, ms,
           2.0 * 32 * 32 * 16 * NT * (double)iters * blocks / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(t);
}

int main() {
    const int K = 128;
    std::vector<float> A(32 * K), B(K * 32), C0(1024), C1(1024);
    srand(1);
    for (auto& v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    for (auto& v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f * (rand() % 7 == 0 ? 1e-3f : 1.f);
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    gemm<<<1, 64>>>(dA, dB, dC, K, 0); hipMemcpy(C0.data(), dC, 4096, hipMemcpyDeviceToHost);
    gemm<<<1, 64>>>(dA, dB, dC, K, 1); hipMemcpy(C1.data(), dC, 4096, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, scale = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double r = 0, ab = 0;
            for (int k = 0; k < K; ++k) { r += (double)A[i * K + k] * B[k * 32 + j]; ab += fabs((double)A[i * K + k] * B[k * 32 + j]); }
            e0 = fmax(e0, fabs(C0[i * 32 + j] - r) / ab); e1 = fmax(e1, fabs(C1[i * 32 + j] - r) / ab); scale = fmax(scale, ab);
        }
    printf("max |C - C64| / sum|a b|:  fp32 mfma %.3e   bf16 x 6 %.3e   (fp32 eps 5.96e-8)\n", e0, e1);
    run_rate<5, 0>(4); run_rate<5, 1>(4); run_rate<5, 0>(8); run_rate<5, 1>(8);
    return 0;
}
