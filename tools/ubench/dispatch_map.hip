// Microbenchmark: where do the one-wave workgroups of a launch go?  Each workgroup records HW_REG_HW_ID and XCC_ID at its
// start, so that (XCD, SE, CU, SIMD, wave slot) can be tabulated against blockIdx.
//   hipcc --offload-arch=gfx950 -O3 dispatch_map.hip -o dispatch_map && ./dispatch_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void k(unsigned* out, unsigned long long* t, int spin) {
    extern __shared__ float lds[];
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID, 32 bits
    unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);  // HW_REG_XCC_ID bits 3:0
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
    lds[threadIdx.x] = x;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; t[blockIdx.x] = t0; }
}
int main() {
    const int blocks = 4096;
    unsigned* out; unsigned long long* t;
    hipMalloc(&out, blocks * 8); hipMalloc(&t, blocks * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 20 * 1024);
    k<<<blocks, 64, 20 * 1024>>>(out, t, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(2 * blocks); std::vector<unsigned long long> ht(blocks);
    hipMemcpy(h.data(), out, blocks * 8, hipMemcpyDeviceToHost);
    hipMemcpy(ht.data(), t, blocks * 8, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull; for (auto v : ht) if (v < tmin) tmin = v;
    printf("block  xcc se sh cu simd wave  start\n");
    for (int b = 0; b < blocks; ++b) {
        if (b < 48 || (b >= 1024 && b < 1040) || (b >= 2048 && b < 2064)) {
            unsigned hw = h[2 * b];
            printf("%5d  %3u %2u %2u %2u %4u %4u  %llu\n", b, h[2 * b + 1] & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15,
                   (hw >> 4) & 3, hw & 15, ht[b] - tmin);
        }
    }
    return 0;
}
