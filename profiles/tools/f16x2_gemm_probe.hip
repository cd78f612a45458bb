// Unit check of the f16x2 tile GEMM (mfma_core.h gemm_rows64_h + pack.hip pack_f16_split) against the exact fp32 tile GEMM
// (gemm_rows64_t) and a double-precision host product, on one 64 x 264 tile with realistic magnitudes.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../../keypoint-diffusion_amd/csrc/pack.hip"
#include "../../keypoint-diffusion_amd/csrc/mfma_core.h"
using namespace kpd;

__global__ __launch_bounds__(256) void k_probe(const float *A, const float *Wp, const void *Wh, const float *wx, float *out32, float *outh, float *outx) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // fp32 tile
    for (int i = tid; i < TM * SA; i += 256) smem[i] = 0.0f;
    __syncthreads();
    for (int i = tid; i < TM * KP; i += 256) smem[(i / KP) * SA + (i % KP)] = A[i];
    __syncthreads();
    f32x16 acc[2][2];
    acc_zero(acc);
    gemm_rows64_t<NG, SA>(smem, Wp, acc, wave, lane);
    for (int mt = 0; mt < 2; ++mt) for (int nt = 0; nt < 2; ++nt) for (int r = 0; r < 16; ++r)
        out32[acc_row(mt, r, lane) * 256 + acc_col(nt, wave, lane)] = acc[mt][nt][r];
    __syncthreads();
    // f16 planes
    _Float16 *Ah = reinterpret_cast<_Float16 *>(smem);
    for (int i = tid; i < TM * 272 / 2; i += 256) {
        const int r = (2 * i) / 272, c = (2 * i) % 272;
        const float a = c < KP ? H_SCALE_A * A[r * KP + c] : 0.0f, b = c + 1 < KP ? H_SCALE_A * A[r * KP + c + 1] : 0.0f;
        unsigned hi, lo;
        split_pair(a, b, hi, lo);
        *reinterpret_cast<unsigned *>(Ah + r * SAH + c) = hi;
        *reinterpret_cast<unsigned *>(Ah + PLANE_H + r * SAH + c) = lo;
    }
    __syncthreads();
    float *wxs = smem + TM * SAH + 64;
    for (int i = tid; i < 272; i += 256) wxs[i] = i < KP ? H_SCALE_W * wx[i] : 0.0f;
    __syncthreads();
    const float ex = row_dot_h(Ah, wxs, tid);
    if ((tid & 3) == 0) outx[tid >> 2] = ex * H_UNSCALE;
    {   // packed-f16 dot-product form
        _Float16 *wxh = reinterpret_cast<_Float16 *>(wxs + 272);
        for (int i = tid; i < 272; i += 256) {
            const float w = wxs[i];
            const _Float16 hi = (_Float16)w;
            wxh[i] = hi;
            wxh[272 + i] = (_Float16)(w - (float)hi);
        }
        __syncthreads();
        const float e2 = row_dot_h2<4>(Ah, wxh, tid);
        if ((tid & 3) == 0) outx[128 + (tid >> 2)] = e2 * H_UNSCALE;
    }
    {   // the same dot element by element
        const int row = tid >> 2, q = tid & 3;
        float sx = 0.0f;
        for (int kx = q; kx < 272; kx += 4) sx = fmaf((float)Ah[row * SAH + kx] + (float)Ah[PLANE_H + row * SAH + kx], wxs[kx], sx);
        sx += __shfl_xor(sx, 1);
        sx += __shfl_xor(sx, 2);
        if ((tid & 3) == 0) outx[64 + (tid >> 2)] = sx * H_UNSCALE;
    }
    acc_zero(acc);
    gemm_rows64_h(Ah, Wh, acc, wave, lane);
    for (int mt = 0; mt < 2; ++mt) for (int nt = 0; nt < 2; ++nt) for (int r = 0; r < 16; ++r)
        outh[acc_row(mt, r, lane) * 256 + acc_col(nt, wave, lane)] = acc[mt][nt][r] * H_UNSCALE;
}


// Timing form: every workgroup repeats the tile GEMM `iters` times on planes it builds once; cycles by s_memtime on wave 0.
__global__ __launch_bounds__(256, 2) void k_time(const void *Wh, int iters, int mode, unsigned long long *cyc, float *sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    _Float16 *Ah = reinterpret_cast<_Float16 *>(smem);
    for (int i = tid; i < 2 * PLANE_H; i += 256) Ah[i] = (_Float16)(0.001f * (float)((i * 7 + blockIdx.x) & 255));
    __syncthreads();
    f32x16 acc[2][2];
    acc_zero(acc);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        gemm_rows64_h(Ah, Wh, acc, wave, lane, (mode & 2) ? (int)((blockIdx.x * 5u + it) % KH_STEPS) : 0);
        if (mode & 1) __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    float v = 0.f;
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) v += acc[m][n][r];
    if (v == 12345.678f) sink[tid] = v;
}

static void time_gemm(const void *dWh, int blocks, int lds, int iters, int mode, const char *what) {
    unsigned long long *cyc; float *sink;
    hipMalloc(&cyc, blocks * 8); hipMalloc(&sink, 1024);
    hipFuncSetAttribute((const void *)k_time, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_time, dim3(blocks), dim3(256), lds, 0, dWh, 2, mode, cyc, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_time, dim3(blocks), dim3(256), lds, 0, dWh, iters, mode, cyc, sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto c : h) sum += (double)c;
    const double per_gemm = sum / blocks / iters;
    const double mfma = 17.0 * 12.0;
    printf("%-44s %4d blocks: %8.0f ticks per tile GEMM (%5.1f per MFMA; floor 32)   kernel %.3f ms -> %.0f useful TFLOP/s\n", what, blocks,
           per_gemm, per_gemm / mfma, ms, (double)blocks * iters * 2.0 * 64 * 264 * 256 / (ms * 1e-3) / 1e12);
    hipFree(cyc); hipFree(sink);
}

// Two tiles per workgroup: waves 0-3 and 4-7 run the same tile GEMM on their own A planes and the SAME weight fragments, in step
// (workgroup barriers) -- does the second reader of a weight line hit in the CU's L1?
__global__ __launch_bounds__(512, 2) void k_time2(const void *Wh, int iters, int mode, unsigned long long *cyc, float *sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = wave >> 2;
    _Float16 *Ah = reinterpret_cast<_Float16 *>(smem) + half * 2 * PLANE_H;
    for (int i = tid & 255; i < 2 * PLANE_H; i += 256) Ah[i] = (_Float16)(0.001f * (float)((i * 7 + blockIdx.x + half) & 255));
    __syncthreads();
    f32x16 acc[2][2];
    acc_zero(acc);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        gemm_rows64_h(Ah, Wh, acc, wave & 3, lane);
        if (mode == 1) __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    float v = 0.f;
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) v += acc[m][n][r];
    if (v == 12345.678f) sink[tid] = v;
}

static void time_gemm2(const void *dWh, int iters, int mode, const char *what) {
    const int blocks = 256;
    unsigned long long *cyc; float *sink;
    hipMalloc(&cyc, blocks * 8); hipMalloc(&sink, 4096);
    hipFuncSetAttribute((const void *)k_time2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_time2, dim3(blocks), dim3(512), 150 * 1024, 0, dWh, 2, mode, cyc, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_time2, dim3(blocks), dim3(512), 150 * 1024, 0, dWh, iters, mode, cyc, sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto c : h) sum += (double)c;
    const double per_gemm = sum / blocks / iters;
    printf("%-44s %4d blocks: %8.0f ticks per PAIR of tile GEMMs (%5.1f per MFMA and SIMD; floor 32)   kernel %.3f ms -> %.0f useful TFLOP/s\n", what,
           blocks, per_gemm, per_gemm / (2 * 17.0 * 12.0), ms, 2.0 * blocks * iters * 2.0 * 64 * 264 * 256 / (ms * 1e-3) / 1e12);
    hipFree(cyc); hipFree(sink);
}

int main() {
    std::vector<float> A(TM * KP), W(257 * KP);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto &v : A) v = 3.0f * rnd();
    for (int r = 0; r < TM; ++r) for (int c = 258; c < KP; ++c) A[r * KP + c] = c == 260 ? 1.0f : 0.0f;
    for (auto &v : W) v = 0.25f * rnd();
    float *dA, *dW, *dWp, *dWx, *o32, *oh, *ox; void *dWh;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dW, W.size() * 4); hipMalloc(&dWp, WP_FLOATS * 4); hipMalloc(&dWx, KP * 4 + 64);
    hipMalloc(&dWh, WH_HALVES * 2); hipMalloc(&o32, TM * 256 * 4); hipMalloc(&oh, TM * 256 * 4); hipMalloc(&ox, 3 * TM * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    pack_gemm_weight(dW, 257, KP, 0, KP, dWp, dWx, nullptr);
    pack_f16_split(dWp, dWh, nullptr);
    hipFuncSetAttribute((const void *)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(256), 75 * 1024, 0, dA, dWp, dWh, dWx, o32, oh, ox);
    hipDeviceSynchronize();
    std::vector<float> r32(TM * 256), rh(TM * 256);
    hipMemcpy(r32.data(), o32, r32.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(rh.data(), oh, rh.size() * 4, hipMemcpyDeviceToHost);
    double e32 = 0, eh = 0, mx = 0;
    for (int r = 0; r < TM; ++r) for (int n = 0; n < 256; ++n) {
        double ref = 0;
        for (int k = 0; k < KP; ++k) ref += (double)A[r * KP + k] * (double)W[n * KP + k];
        mx = fmax(mx, fabs(ref));
        e32 = fmax(e32, fabs(r32[r * 256 + n] - ref));
        eh = fmax(eh, fabs(rh[r * 256 + n] - ref));
    }
    time_gemm(dWh, 256, 150 * 1024, 200, 0, "one workgroup per CU (one wave per SIMD)");
    time_gemm(dWh, 512, 75 * 1024, 200, 0, "two workgroups per CU (two waves per SIMD)");
    time_gemm(dWh, 512, 75 * 1024, 200, 1, "two per CU, barrier after every GEMM");
    time_gemm(dWh, 256, 150 * 1024, 200, 2, "one per CU, k-steps rotated per workgroup");
    time_gemm(dWh, 512, 75 * 1024, 200, 2, "two per CU, k-steps rotated per workgroup");
    time_gemm2(dWh, 200, 0, "tile pair per workgroup, free running");
    time_gemm2(dWh, 200, 1, "tile pair per workgroup, barrier per GEMM");
    printf("max |ref| %.4f   fp32 MFMA max err %.3e (%.2e rel)   f16x2 max err %.3e (%.2e rel)\n", mx, e32, e32 / mx, eh, eh / mx);
    std::vector<float> rx(3 * TM);
    hipMemcpy(rx.data(), ox, 3 * TM * 4, hipMemcpyDeviceToHost);
    for (int r = 0; r < 4; ++r) {
        double ref = 0;
        for (int k = 0; k < KP; ++k) ref += (double)A[r * KP + k] * (double)W[256 * KP + k];
        printf("row %d: chunked %.7f  elementwise %.7f  dot2 %.7f  host %.7f\n", r, rx[r], rx[64 + r], rx[128 + r], ref);
    }
    double exe = 0, exm = 0;
    for (int r = 0; r < TM; ++r) {
        double ref = 0;
        for (int k = 0; k < KP; ++k) ref += (double)A[r * KP + k] * (double)W[256 * KP + k];
        exm = fmax(exm, fabs(ref));
        exe = fmax(exe, fabs(rx[r] - ref));
    }
    printf("column 256 (row_dot_h): max |ref| %.4f  max err %.3e (%.2e rel)\n", exm, exe, exe / exm);
    double e2m = 0;
    for (int r = 0; r < TM; ++r) {
        double ref = 0;
        for (int k = 0; k < KP; ++k) ref += (double)A[r * KP + k] * (double)W[256 * KP + k];
        e2m = fmax(e2m, fabs(rx[128 + r] - ref));
    }
    printf("column 256 (row_dot_h2, v_dot2_f32_f16): max err %.3e (%.2e rel)\n", e2m, e2m / exm);
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
