// Microbenchmark: v_mfma_f32_32x32x2_f32 fed by A fragments streamed from L2 (the scalar_out loops of the GCP kernels:
// one 16-byte load per lane and k step, four accumulators), against the same MFMA stream with constant operands.
// Reports s_memtime ticks per MFMA per wave for 1 and 2 waves per SIMD and several prefetch shapes.
//   hipcc --offload-arch=gfx950 -O3 mfma_stream.hip -o mfma_stream && ./mfma_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

// MODE 0: no loads.  MODE 1: three rotating batches of U steps (as in the kernels).  MODE 2: same, loads through a
// wave-uniform base (SGPR) + lane offset (buffer-style addressing: no per-step 64-bit VALU address arithmetic).
template <int MODE, int U>
__global__ __launch_bounds__(64) void k(const float* __restrict__ w, int steps, int iters, float* out, unsigned long long* t) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float b = 1.0f + lane * 1e-4f;
    const float4* wp = reinterpret_cast<const float4*>(w) + lane;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
            float4 a = make_float4(1.f, 2.f, 3.f, 4.f);
            for (int s = 0; s < steps; ++s) {
                acc[0] = MF(a.x, b, acc[0]); acc[1] = MF(a.y, b, acc[1]); acc[2] = MF(a.z, b, acc[2]); acc[3] = MF(a.w, b, acc[3]);
            }
        } else {
            float4 A0[U], A1[U], A2[U];
            auto ld = [&](float4(&a)[U], int s0) {
#pragma unroll
                for (int u = 0; u < U; ++u) a[u] = wp[(size_t)min(s0 + u, steps - 1) * 64];
            };
            auto mm = [&](float4(&a)[U]) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    acc[0] = MF(a[u].x, b, acc[0]); acc[1] = MF(a[u].y, b, acc[1]); acc[2] = MF(a[u].z, b, acc[2]); acc[3] = MF(a[u].w, b, acc[3]);
                }
            };
            ld(A0, 0); ld(A1, U); ld(A2, 2 * U);
            __builtin_amdgcn_sched_barrier(0);
            for (int s0 = 0; s0 < steps; s0 += 3 * U) {
                mm(A0); ld(A0, s0 + 3 * U); __builtin_amdgcn_sched_barrier(0);
                mm(A1); ld(A1, s0 + 4 * U); __builtin_amdgcn_sched_barrier(0);
                mm(A2); ld(A2, s0 + 5 * U); __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 64 + lane] = s + lds[lane];
    if (lane == 0) t[blockIdx.x] = t1 - t0;
}

template <int MODE, int U>
void run(const float* w, int waves_per_cu, int steps, const char* name) {
    const int blocks = 256 * waves_per_cu, iters = 40;
    float* out; unsigned long long* t;
    (void)hipMalloc(&out, blocks * 64 * 4); (void)hipMalloc(&t, blocks * 8);
    const size_t lds = waves_per_cu <= 4 ? 36 * 1024 : 18 * 1024;  // pins the residency to 4 or 8 one-wave workgroups per CU
    (void)hipFuncSetAttribute((const void*)k<MODE, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k<MODE, U><<<blocks, 64, lds>>>(w, steps, iters, out, t);
    (void)hipDeviceSynchronize();
    k<MODE, U><<<blocks, 64, lds>>>(w, steps, iters, out, t);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), t, blocks * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto v : h) sum += v;
    const int st = MODE == 0 ? steps : (steps + 3 * U - 1) / (3 * U) * 3 * U;
    printf("%-44s waves/CU %d: %.1f ticks per MFMA per wave\n", name, waves_per_cu, sum / blocks / ((double)iters * st * 4));
    (void)hipFree(out); (void)hipFree(t);
}

int main() {
    const int steps = 72;  // 72 k steps x 1 KiB = 72 KiB of fragments per pass (one GCP block's scalar_out), L2 resident
    float* w;
    (void)hipMalloc(&w, (size_t)(steps + 64) * 1024);
    (void)hipMemset(w, 0, (size_t)(steps + 64) * 1024);
    for (int wpc : {4, 8}) {
        run<0, 4>(w, wpc, steps, "constant operands (no loads)");
        run<1, 4>(w, wpc, steps, "streamed A, 3 batches x 4 steps in flight");
        run<1, 8>(w, wpc, steps, "streamed A, 3 batches x 8 steps in flight");
        run<1, 2>(w, wpc, steps, "streamed A, 3 batches x 2 steps in flight");
    }
    return 0;
}
