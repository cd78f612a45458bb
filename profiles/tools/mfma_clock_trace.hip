// Which fp32-MFMA rate does this part SUSTAIN, and at which clock?  (round-3 question: the guide quotes 155 TFLOP/s measured for
// v_mfma_f32_32x32x2_f32, round 2's loop probe saw 132.)  Bare back-to-back fp32 MFMAs on every SIMD of every CU (no memory
// traffic at all), launched as a series of kernels after an idle pause; per launch: wall time (HIP events), the shader-clock
// cycles the waves counted (s_memtime) => effective clock = cycles / time, and the rate.  A burst that starts at the boost clock
// and settles lower is DVFS under the MFMA load, not a property of the instruction.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_clock_trace mfma_clock_trace.hip && ./mfma_clock_trace
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 2) void k_mfma(float *out, unsigned long long *cyc, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.0f;
    const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {          // four independent accumulators: 16 MFMAs per iteration, none waits on another
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[3], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main(int argc, char **argv) {
    const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 1;       // 1: one workgroup of 4 waves per CU; 2: two
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, blocks = cus * waves_per_simd;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&cyc, (size_t)blocks * 4 * 8);
    std::vector<unsigned long long> h(blocks * 4);
    printf("# %s, %d CUs, clockRate %d kHz; %d wave(s) per SIMD\n", prop.name, cus, prop.clockRate, waves_per_simd);
    printf("# launch  t_start_ms  kernel_ms  MFMA_per_wave  cycles_per_MFMA  eff_clock_MHz  TFLOP/s\n");
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    sleep(3);                                                         // let the part idle down first
    const auto T0 = std::chrono::steady_clock::now();
    // 40 short launches (~0.25 ms each), then 40 long ones (~25 ms each), then 20 short ones again
    for (int l = 0; l < 100; ++l) {
        const int iters = (l >= 40 && l < 80) ? 60000 : 600;
        const double t_start = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= h.size();
        const double mfma = 16.0 * iters;
        printf("%3d %10.2f %9.4f %8.0f %8.2f %8.0f %8.1f\n", l, t_start, ms, mfma, avg / mfma, avg / (ms * 1e3),
               (double)blocks * 4 * mfma * 4096 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
