// Probe: does a wave issuing back-to-back v_mfma_f32_32x32x2_f32 leave issue slots for a second
// wave on the same SIMD (VALU / LDS / global-load work)?  512-thread workgroups, 1 per CU:
// waves 0-3 = MFMA role, waves 4-7 = "helper" role.  mode bit0: MFMA role active, bit1: helper active.
// helper kind: 0 = dependent v_fma chain, 1 = LDS reads, 2 = global loads (L2 resident).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void probe(float *out, const float *g, int mode, int kind, int n_mfma, int n_help, int prio) {
    __shared__ float lds[4096];
    const int tid = threadIdx.x, wave = tid >> 6;
    lds[tid] = tid; lds[tid + 512] = 1.0f;
    __syncthreads();
    float r = 0.f;
    if (wave < 4) {
        if (mode & 1) {
            f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
            float x = tid * 1e-3f, y = 1.0f + tid * 1e-4f;
            for (int i = 0; i < n_mfma; i += 4) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
            }
            r = a0[0] + a1[1] + a2[2] + a3[3];
        }
    } else if (mode & 2) {
        if (prio) __builtin_amdgcn_s_setprio(3);
        if (kind == 0) {
            float a = tid * 1e-3f, b = 1.0001f, c0 = 0, c1 = 1, c2 = 2, c3 = 3;
            for (int i = 0; i < n_help; i += 4) { c0 = fmaf(a, b, c0); c1 = fmaf(a, b, c1); c2 = fmaf(a, b, c2); c3 = fmaf(a, b, c3); }
            r = c0 + c1 + c2 + c3;
        } else if (kind == 1) {
            int idx = tid & 1023;
            for (int i = 0; i < n_help; ++i) { float v = lds[idx]; idx = (idx + (int)v + 64) & 1023; r += v; }
        } else {
            size_t idx = (size_t)blockIdx.x * 4096 + (tid & 255) * 4;
            for (int i = 0; i < n_help; ++i) { float4 v = *(const float4 *)(g + idx); r += v.x; idx = (idx + 1024 + (int)(v.y)) & ((1u << 22) - 1); }
        }
    }
    out[blockIdx.x * 512 + tid] = r;
}
int main() {
    float *out, *g;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&g, (size_t)(1 << 22) * 4 + 65536); hipMemset(g, 0, (size_t)(1 << 22) * 4 + 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n_mfma = 20000;
    int n_help[3] = {400000, 8000, 1500};
    for (int prio = 0; prio < 2; ++prio)
    for (int kind = 0; kind < 3; ++kind)
        for (int mode = 1; mode <= 3; ++mode) {
            probe<<<256, 512>>>(out, g, mode, kind, n_mfma, n_help[kind], prio);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            probe<<<256, 512>>>(out, g, mode, kind, n_mfma, n_help[kind], prio);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("prio %d kind %d (%s) mode %d (%s): %.3f ms\n", prio, kind, kind == 0 ? "valu fma" : kind == 1 ? "lds dep-read" : "global dep-load",
                   mode, mode == 1 ? "mfma only" : mode == 2 ? "helper only" : "both", ms);
        }
    return 0;
}
