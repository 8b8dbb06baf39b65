// Microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 in s_memtime ticks, independent accumulators vs one dependent
// chain, with 1 or 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* t, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x * 1e-4f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

template <int NACC>
void run(int waves_per_cu, const char* name) {
    int ncu = 256, blocks = ncu * waves_per_cu, iters = 2000;
    float* out; unsigned long long* t;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&t, blocks * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<NACC><<<blocks, 64>>>(out, t, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<NACC><<<blocks, 64>>>(out, t, iters);
    hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), t, blocks * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto v : h) sum += v;
    double per = sum / blocks / (iters * (double)NACC);
    double tflops = 2.0 * 32 * 32 * 2 * (double)iters * NACC * blocks / (ms * 1e-3) / 1e12;
    printf("%-28s waves/CU %2d: %.1f memtime ticks per MFMA per wave, kernel %.3f ms, %.1f TFLOP/s\n", name, waves_per_cu, per, ms, tflops);
}

int main() {
    run<4>(4, "4 independent accumulators");
    run<4>(8, "4 independent accumulators");
    run<1>(4, "1 dependent chain");
    run<1>(8, "1 dependent chain");
    run<2>(4, "2 accumulators");
    return 0;
}
