// Which hardware wave slots do two co-resident 256-thread workgroups (71 KB LDS each, 2 per CU) occupy on a SIMD?
// Decides whether HW_ID.wave_id can tell the two co-resident waves of a SIMD apart (edge-kernel priority experiment).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void k(unsigned *out, int spin) {
    extern __shared__ float smem[];
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float acc = threadIdx.x;
    for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;
    smem[threadIdx.x] = acc;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
        out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = (unsigned)(t0 >> 8);
    }
    if (smem[(threadIdx.x + 1) & 255] == 12345.f) out[0] = 0;
}
int main() {
    const int blocks = 2048;
    unsigned *d;
    hipMalloc(&d, blocks * 8 * sizeof(unsigned));
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 71 * 1024, 0, d, 20000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(blocks * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, int> slot, simd;
    std::map<unsigned, std::map<unsigned, int>> per_wave_slot;
    int mixed = 0;
    for (int b = 0; b < blocks; ++b) {
        unsigned s0 = h[(b * 4) * 2] & 0xF;
        bool mix = false;
        for (int w = 0; w < 4; ++w) {
            unsigned hw = h[(b * 4 + w) * 2];
            slot[hw & 0xF]++;
            simd[(hw >> 4) & 3]++;
            per_wave_slot[w][(hw >> 4) & 3]++;
            if ((hw & 0xF) != s0) mix = true;
        }
        mixed += mix;
    }
    printf("wave_id histogram:");
    for (auto &kv : slot) printf(" [%u]=%d", kv.first, kv.second);
    printf("\nsimd_id histogram:");
    for (auto &kv : simd) printf(" [%u]=%d", kv.first, kv.second);
    printf("\nblocks whose 4 waves have different wave_id: %d of %d\n", mixed, blocks);
    for (int b = 0; b < 6; ++b) {
        printf("block %d:", b);
        for (int w = 0; w < 4; ++w) printf(" %08x", h[(b * 4 + w) * 2]);
        printf("\n");
    }
    return 0;
}
