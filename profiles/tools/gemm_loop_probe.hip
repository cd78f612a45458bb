// What does the k-loop of the 64-row fp32-MFMA tile GEMM (mfma_core.h, gemm_rows64_t) cost per MFMA, ingredient by ingredient?
// One workgroup of 4 waves per CU (or two: second argument), A tile in LDS, packed weights in L2; each workgroup repeats the
// 33-group GEMM `reps` times back to back.  Variants: 0 bare MFMAs (operands never reloaded), 1 + LDS A reads, 2 + global B loads,
// 3 both (= production loop), 4 production loop with the scalar-base addressing variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../keypoint-diffusion_amd/csrc/mfma_core.h"
using namespace kpd;

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const float *__restrict__ Wp, float *out, unsigned long long *cyc, int reps) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < TM * SA; i += 256) smem[i] = 0.001f * (float)((i * 7 + blockIdx.x) % 97);
    __syncthreads();
    f32x16 acc[2][2];
    acc_zero(acc);
    const int r = lane & 31, h = lane >> 5;
    const float *a0p = smem + r * SA + 4 * h;
    const float *a1p = smem + (32 + r) * SA + 4 * h;
    gf32x4 *bp = as_global(reinterpret_cast<const f32x4 *>(Wp) + (wave * 64 + lane) * 2);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        f32x4 xa0, xa1, xb0, xb1, ya0, ya1, yb0, yb1;
        KPD_GEMM_LOAD(xa0, xa1, xb0, xb1, 0)
        KPD_GEMM_LOAD(ya0, ya1, yb0, yb1, 1)
#pragma unroll 1
        for (int p = 0; p < NG / 2; ++p) {
            const int g = 2 * p;
            __builtin_amdgcn_sched_barrier(0);
            KPD_GEMM_STEP(xa0, xa1, xb0, xb1)
            __builtin_amdgcn_sched_barrier(0);
            const int g2 = g + 2 < NG ? g + 2 : NG - 1;
            if (MODE & 1) {
                xa0 = *reinterpret_cast<const f32x4 *>(a0p + 8 * g2);
                xa1 = *reinterpret_cast<const f32x4 *>(a1p + 8 * g2);
            }
            if (MODE & 2) {
                xb0 = bp[g2 * 512];
                xb1 = bp[g2 * 512 + 1];
            }
            __builtin_amdgcn_sched_barrier(0);
            KPD_GEMM_STEP(ya0, ya1, yb0, yb1)
            __builtin_amdgcn_sched_barrier(0);
            const int g3 = g + 3 < NG ? g + 3 : NG - 1;
            if (MODE & 1) {
                ya0 = *reinterpret_cast<const f32x4 *>(a0p + 8 * g3);
                ya1 = *reinterpret_cast<const f32x4 *>(a1p + 8 * g3);
            }
            if (MODE & 2) {
                yb0 = bp[g3 * 512];
                yb1 = bp[g3 * 512 + 1];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        KPD_GEMM_STEP(xa0, xa1, xb0, xb1)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[0][0][i] + acc[0][1][i] + acc[1][0][i] + acc[1][1][i];
    out[blockIdx.x * 256 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE>
void run(const float *W, float *out, unsigned long long *cyc, int blocks, int reps, int lds_pad) {
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int lds = TM * SA * 4 + lds_pad;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, W, out, cyc, 2);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, W, out, cyc, reps);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
    const double mfma = (double)reps * NG * 16;
    printf("mode %d (%s%s) blocks %d lds_pad %d: %.3f ms, %.1f cycles per MFMA per wave (s_memtime), %.1f TFLOP/s\n", MODE,
           (MODE & 1) ? "LDS-A " : "", (MODE & 2) ? "global-B" : "", blocks, lds_pad, ms, avg / mfma,
           (double)blocks * 4 * mfma * 4096 / (ms * 1e-3) / 1e12);
}

int main(int argc, char **argv) {
    const int reps = 40;
    float *W, *out; unsigned long long *cyc;
    hipMalloc(&W, WP_FLOATS * 4 + 4096); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 4 * 8);
    std::vector<float> hw(WP_FLOATS);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.01f * (float)((i * 13) % 31 - 15);
    hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    for (int two = 0; two < 2; ++two) {
        const int blocks = two ? 512 : 256, pad = two ? 0 : 24000;
        run<0>(W, out, cyc, blocks, reps, pad);
        run<1>(W, out, cyc, blocks, reps, pad);
        run<2>(W, out, cyc, blocks, reps, pad);
        run<3>(W, out, cyc, blocks, reps, pad);
    }
    return 0;
}
