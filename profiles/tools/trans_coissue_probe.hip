// Does the transcendental unit (v_exp_f32 / v_rcp_f32, quarter rate) run beside the fp32 MFMA on gfx950?  Round 2 established that
// plain VALU work does not (coissue_probe.hip: the f32 MFMA IS the vector pipe); SiLU costs this path 537 trans instructions per
// thread and 64-edge tile (exp2 + rcp per element of the A and T tiles), ~9 % of the edge kernel, so the question decides whether
// interleaving them with the k-loop could hide them.  Modes: MFMA only / helper only / both, with the helper (a) in a second wave
// of the same SIMD, (b) interleaved into the MFMA wave's own instruction stream.  Helper kinds: fma (control), exp, rcp.
// Launches last tens of milliseconds and are repeated: the clock is settled (profiles/r03_mfma_clock_trace.txt).
//   hipcc -O3 --offload-arch=gfx950 -o trans_coissue_probe trans_coissue_probe.hip && ./trans_coissue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>
__device__ __forceinline__ float help4(float v) {
    if (KIND == 0) return fmaf(v, 1.0001f, 0.5f);
    if (KIND == 1) return __builtin_amdgcn_exp2f(v) * 0.5f;        // v_exp_f32 (+ a mul to keep the value bounded)
    return __builtin_amdgcn_rcpf(v) + 1.0f;                         // v_rcp_f32 (+ an add)
}

// 512 threads: waves 0-3 MFMA role (one per SIMD), waves 4-7 helper role (one per SIMD)
template <int KIND>
__global__ __launch_bounds__(512) void k_two_waves(float *out, int mode, int n_mfma, int n_help) {
    const int tid = threadIdx.x, wave = tid >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
            const float x = tid * 1e-3f, y = 1.0f + tid * 1e-4f;
            for (int i = 0; i < n_mfma; i += 4) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
            }
            r = a0[0] + a1[1] + a2[2] + a3[3];
        }
    } else if (mode & 2) {
        float c0 = 1.0f + tid * 1e-3f, c1 = 1.1f, c2 = 1.2f, c3 = 1.3f;
        for (int i = 0; i < n_help; i += 4) { c0 = help4<KIND>(c0); c1 = help4<KIND>(c1); c2 = help4<KIND>(c2); c3 = help4<KIND>(c3); }
        r = c0 + c1 + c2 + c3;
    }
    out[blockIdx.x * 512 + tid] = r;
}

// 256 threads: every wave issues 4 MFMAs then PER independent helper ops, same instruction stream
template <int KIND, int PER>
__global__ __launch_bounds__(256) void k_same_wave(float *out, int mode, int n_mfma) {
    const int tid = threadIdx.x;
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    const float x = tid * 1e-3f, y = 1.0f + tid * 1e-4f;
    float c[8] = {1.0f + tid * 1e-3f, 1.1f, 1.2f, 1.3f, 1.4f, 1.5f, 1.6f, 1.7f};
    for (int i = 0; i < n_mfma; i += 4) {
        if (mode & 1) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
        }
        if (mode & 2) {
#pragma unroll
            for (int j = 0; j < PER; ++j) c[j & 7] = help4<KIND>(c[j & 7]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float r = a0[0] + a1[1] + a2[2] + a3[3];
    for (int j = 0; j < 8; ++j) r += c[j];
    out[blockIdx.x * 256 + tid] = r;
}

template <typename F>
float time_ms(F launch) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 3;
}

template <int KIND>
void run(float *out, const char *name, int n_help) {
    const int n_mfma = 400000;                  // 400 k x 64 cycles = 25.6 M cycles ~ 11 ms
    float t[3];
    for (int mode = 1; mode <= 3; ++mode) t[mode - 1] = time_ms([&] { hipLaunchKernelGGL(k_two_waves<KIND>, dim3(256), dim3(512), 0, 0, out, mode, n_mfma, n_help); });
    printf("second wave, helper %-4s: mfma %.2f ms  helper %.2f ms  both %.2f ms  (sum %.2f, max %.2f)\n", name, t[0], t[1], t[2], t[0] + t[1],
           t[0] > t[1] ? t[0] : t[1]);
    for (int mode = 1; mode <= 3; ++mode) t[mode - 1] = time_ms([&] { hipLaunchKernelGGL((k_same_wave<KIND, 4>), dim3(256), dim3(256), 0, 0, out, mode, n_mfma); });
    printf("same wave  , helper %-4s (4 per 4 MFMAs): mfma %.2f ms  helper %.2f ms  both %.2f ms  (sum %.2f)\n", name, t[0], t[1], t[2], t[0] + t[1]);
    for (int mode = 1; mode <= 3; ++mode) t[mode - 1] = time_ms([&] { hipLaunchKernelGGL((k_same_wave<KIND, 8>), dim3(256), dim3(256), 0, 0, out, mode, n_mfma); });
    printf("same wave  , helper %-4s (8 per 4 MFMAs): mfma %.2f ms  helper %.2f ms  both %.2f ms  (sum %.2f)\n", name, t[0], t[1], t[2], t[0] + t[1]);
}

int main() {
    float *out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    run<0>(out, "fma", 6400000);
    run<1>(out, "exp", 1600000);
    run<2>(out, "rcp", 1600000);
    return 0;
}
