// Probe of the register-chained GEMM inner loop (chain_core.h) on gfx950: what does each ingredient of the chunk loop
// cost per v_mfma_f32_16x16x4_f32?  256-thread workgroups, `bpc` workgroups per CU (1 or 2 waves per SIMD).
//   level 0: bare MFMA stream, 16 accumulators, 64 MFMAs per "chunk"
//   level 1: + one ds_read_b128 per 4 MFMAs (A fragments from LDS, software-pipelined one batch ahead)
//   level 2: + one workgroup barrier per chunk
//   level 3: + LDS-DMA refill of one 16-KB chunk per chunk (global_load_lds_dwordx4, three-buffer ring, counted vmcnt)
// Build: hipcc -O3 --offload-arch=gfx950 chain_probe.hip -o chain_probe.   Output: cycles per MFMA at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// PIN: pin the LDS reads of batch g + 1 ahead of the MFMAs of batch g with sched_barrier (as chain_core.h chunk_gemm)
#ifndef PIN
#define PIN 0
#endif
template <int LEVEL>
__global__ __launch_bounds__(256, 2) void probe(float *out, const float *wts, int n_chunks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    v4f *ring = reinterpret_cast<v4f *>(smem);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 3 * 1024; i += 256) ring[i] = v4f{1e-3f * i, 1.f, 0.5f, 0.25f};
    __syncthreads();
    v4f acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
    v4f x = {1.0f + lane * 1e-3f, 0.5f, 0.25f, 0.125f};
    const v4f *src = reinterpret_cast<const v4f *>(wts) + (size_t)(blockIdx.x & 63) * 1024 + tid;
    if (LEVEL >= 3) {
        for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds((glb_void *)(src + 256 * j), (lds_void *)(ring + 64 * wave + 256 * j), 16, 0, 0);
        for (int j = 0; j < 4; ++j) __builtin_amdgcn_global_load_lds((glb_void *)(src + 256 * j), (lds_void *)(ring + 1024 + 64 * wave + 256 * j), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
#pragma unroll 1
    for (int c = 0; c < n_chunks; ++c) {
        const int b = c % 3;
        if (LEVEL >= 3) {
            const int b2 = (c + 2) % 3;
            const v4f *g = src + (size_t)((c + 2) & 15) * 65536;
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_global_load_lds((glb_void *)(g + 256 * j), (lds_void *)(ring + b2 * 1024 + 64 * wave + 256 * j), 16, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        const v4f *wp = ring + b * 1024 + lane;
        if (LEVEL >= 1) {
            v4f w[2][4];
#pragma unroll
            for (int m = 0; m < 4; ++m) w[0][m] = wp[m * 64];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g + 1 < 4) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) w[(g + 1) & 1][m] = wp[(4 * (g + 1) + m) * 64];
                }
                if (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        acc[4 * g + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[g & 1][m][r], x[r], acc[4 * g + m], 0, 0, 0);
                if (PIN) __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int m = 0; m < 4; ++m) acc[4 * g + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[(r + m) & 3], x[r], acc[4 * g + m], 0, 0, 0);
        }
        if (LEVEL >= 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if (LEVEL >= 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (LEVEL >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + tid] = r;
}

template <int LEVEL>
static void run(float *out, const float *wts, int bpc) {
    const int n_chunks = 2000, blocks = 256 * bpc;
    const size_t lds = bpc == 1 ? 100 * 1024 : 49152;     // bpc 1: pad LDS so that only one workgroup fits a CU
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe<LEVEL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<LEVEL><<<blocks, 256, lds>>>(out, wts, n_chunks);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<LEVEL><<<blocks, 256, lds>>>(out, wts, n_chunks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: bpc waves x n_chunks x 64 MFMAs
    const double mfma_per_simd = (double)bpc * n_chunks * 64;
    printf("level %d, %d workgroup(s)/CU: %.3f ms, %.1f cycles per MFMA per SIMD at 2.4 GHz (32 = peak)\n", LEVEL, bpc, ms,
           ms * 1e-3 * 2.4e9 / mfma_per_simd);
}

int main() {
    float *out, *wts;
    hipMalloc(&out, 512 * 256 * 4);
    hipMalloc(&wts, (size_t)(16 * 65536 + 65536) * 16);
    hipMemset(wts, 0, (size_t)(16 * 65536 + 65536) * 16);
    for (int bpc = 1; bpc <= 2; ++bpc) {
        run<0>(out, wts, bpc);
        run<1>(out, wts, bpc);
        run<2>(out, wts, bpc);
        run<3>(out, wts, bpc);
    }
    return 0;
}
