// Microbenchmark: how fast can a 64-lane wave move a 32 x 128 fp32 tile between HBM and its registers, for the access
// patterns used by the GCP kernels?  One wave per workgroup, one tile per wave (like the kernels), 160 000 rows.
//   acc-load   : MFMA accumulator layout straight from global: lane (row = lane & 31, half) reads 16 B at column
//                32 t + 8 q + 4 half, 16 instructions per tile, each touching 32 rows x 32 B
//   row-load   : coalesced: each instruction reads two full 512-byte rows, 16 instructions per tile
//   acc-store / row-store : same for stores
//   lds-store  : accumulator layout -> LDS, one 32 x 32 tile at a time -> stores of full 128-byte lines (8 rows per instruction)
//   lds-load   : row-load + write to an LDS tile (stride 132) + read back in the accumulator layout (ds_read_b128)
// hipcc --offload-arch=gfx950 -O3 tile_access.hip -o tile_access && ./tile_access
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int LD = 128;

template <int MODE>
__global__ __launch_bounds__(64, 2) void k(const float* __restrict__ in, float* __restrict__ out, int rows) {
    __shared__ __attribute__((aligned(16))) float tile[32 * 132];
    const int lane = threadIdx.x, e = lane & 31, hi = lane >> 5;
    const int r0 = blockIdx.x * 32;
    float4 v[16];
    if (MODE == 0 || MODE == 2 || MODE == 5 || MODE == 6) {  // accumulator layout
        const int row = min(r0 + e, rows - 1);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[4 * t + q] = *reinterpret_cast<const float4*>(in + (int64_t)row * LD + 32 * t + 8 * q + 4 * hi);
    } else if (MODE == 1 || MODE == 3) {  // two full rows per instruction
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = min(r0 + 2 * i + hi, rows - 1);
            v[i] = *reinterpret_cast<const float4*>(in + (int64_t)row * LD + 4 * e);
        }
    } else {  // MODE 4: rows -> LDS -> accumulator layout
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = min(r0 + 2 * i + hi, rows - 1);
            v[i] = *reinterpret_cast<const float4*>(in + (int64_t)row * LD + 4 * e);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) *reinterpret_cast<float4*>(tile + (2 * i + hi) * 132 + 4 * e) = v[i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[4 * t + q] = *reinterpret_cast<const float4*>(tile + e * 132 + 32 * t + 8 * q + 4 * hi);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { v[i].x += 1.f; v[i].y += 1.f; v[i].z += 1.f; v[i].w += 1.f; }
    if (MODE == 2) {  // accumulator-layout store
        const int row = r0 + e;
        if (row < rows)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(out + (int64_t)row * LD + 32 * t + 8 * q + 4 * hi) = v[4 * t + q];
    } else if (MODE == 5) {  // accumulator layout -> one 32 x 32 tile at a time through LDS -> full 128-byte lines per row
        float* st = tile;  // [32][36]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(st + e * 36 + 8 * q + 4 * hi) = v[4 * t + q];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float4 w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const float4*>(st + (8 * j + (lane >> 3)) * 36 + 4 * (lane & 7));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = r0 + 8 * j + (lane >> 3);
                if (row < rows) *reinterpret_cast<float4*>(out + (int64_t)row * LD + 32 * t + 4 * (lane & 7)) = w[j];
            }
        }
    } else if (MODE == 6) {  // same through a 32 x 20 tile: half tiles, 64-byte pieces per row (16 rows per instruction)
        float* st = tile;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int q = 0; q < 2; ++q) *reinterpret_cast<float4*>(st + e * 20 + 8 * q + 4 * hi) = v[4 * t + 2 * h + q];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                float4 w[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) w[j] = *reinterpret_cast<const float4*>(st + (16 * j + (lane >> 2)) * 20 + 4 * (lane & 3));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = r0 + 16 * j + (lane >> 2);
                    if (row < rows) *reinterpret_cast<float4*>(out + (int64_t)row * LD + 32 * t + 16 * h + 4 * (lane & 3)) = w[j];
                }
            }
    } else if (MODE == 3) {  // coalesced store
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = r0 + 2 * i + hi;
            if (row < rows) *reinterpret_cast<float4*>(out + (int64_t)row * LD + 4 * e) = v[i];
        }
    } else {  // reduce to one float per lane so that the loads are not dead
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
        out[(int64_t)blockIdx.x * 64 + lane] = s;
    }
}

template <int MODE>
void run(const char* name, const float* in, float* out, int rows, double bytes) {
    const int blocks = (rows + 31) / 32;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) k<MODE><<<blocks, 64>>>(in, out, rows);
    hipDeviceSynchronize();
    hipEventRecord(a);
    const int iters = 20;
    for (int i = 0; i < iters; ++i) k<MODE><<<blocks, 64>>>(in, out, rows);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("%-10s %8.1f us per launch  %7.0f GB/s\n", name, ms / iters * 1e3, bytes / (ms / iters * 1e-3) / 1e9);
}

int main() {
    const int rows = 160000 * 8;  // 655 MB per array: well past the 256 MB Infinity Cache
    float *in, *out;
    hipMalloc(&in, (size_t)rows * LD * 4);
    hipMalloc(&out, (size_t)rows * LD * 4);
    hipMemset(in, 0, (size_t)rows * LD * 4);
    const double rd = (double)rows * LD * 4;
    run<0>("acc-load", in, out, rows, rd);
    run<1>("row-load", in, out, rows, rd);
    run<4>("lds-load", in, out, rows, rd);
    run<2>("acc-store", in, out, rows, 2 * rd);
    run<3>("row-store", in, out, rows, 2 * rd);
    run<5>("lds-store", in, out, rows, 2 * rd);
    run<6>("lds-store64", in, out, rows, 2 * rd);
    return 0;
}
