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

// Can VALU work ride in the gaps of this GEMM loop?  The production loop with FILL independent VALU instructions (half v_exp_f32,
// half v_fma_f32, eight independent chains) behind every MFMA of the same wave.
template <int FILL>
__device__ __forceinline__ void gemm_fill(const _Float16 *__restrict__ Ah, const void *__restrict__ Wh, f32x16 (&acc)[2][2], int wave, int lane,
                                          float (&f)[8]) {
    const int r = lane & 31, h = lane >> 5;
    const _Float16 *a0p = Ah + r * SAH + 8 * h;
    const _Float16 *a1p = Ah + (32 + r) * SAH + 8 * h;
    gf32x4 *bp = as_global(reinterpret_cast<const f32x4 *>(Wh) + wave * 256 + lane);
    f32x4 b[H_B_DEPTH][4], a[3][4];
#pragma unroll
    for (int i = 0; i < H_B_DEPTH - 1; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) b[i][j] = bp[i * 1024 + j * 64];
    a[0][0] = *reinterpret_cast<const f32x4 *>(a0p); a[0][1] = *reinterpret_cast<const f32x4 *>(a1p);
    a[0][2] = *reinterpret_cast<const f32x4 *>(a0p + PLANE_H); a[0][3] = *reinterpret_cast<const f32x4 *>(a1p + PLANE_H);
#define FILLER(I)                                                                   \
    _Pragma("unroll") for (int q = 0; q < FILL; ++q) {                              \
        const int c = ((I) * FILL + q) & 7;                                         \
        if (q & 1) f[c] = __builtin_amdgcn_exp2f(f[c]);                             \
        else f[c] = fmaf(f[c], 0.999f, 0.001f);                                     \
    }
#pragma unroll
    for (int s = 0; s < KH_STEPS; ++s) {
        const int cb = s % H_B_DEPTH, ca = s % 3, can = (s + 1) % 3;
        if (s + 1 < KH_STEPS) {
            a[can][0] = *reinterpret_cast<const f32x4 *>(a0p + 16 * (s + 1)); a[can][1] = *reinterpret_cast<const f32x4 *>(a1p + 16 * (s + 1));
            a[can][2] = *reinterpret_cast<const f32x4 *>(a0p + PLANE_H + 16 * (s + 1)); a[can][3] = *reinterpret_cast<const f32x4 *>(a1p + PLANE_H + 16 * (s + 1));
        }
        if (s + H_B_DEPTH - 1 < KH_STEPS) {
            const int nb = (s + H_B_DEPTH - 1) % H_B_DEPTH;
#pragma unroll
            for (int j = 0; j < 4; ++j) b[nb][j] = bp[(s + H_B_DEPTH - 1) * 1024 + j * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        KPD_H_MFMA(acc[0][0], a[ca][2], b[cb][0]); FILLER(0) KPD_H_MFMA(acc[0][1], a[ca][2], b[cb][2]); FILLER(1)
        KPD_H_MFMA(acc[1][0], a[ca][3], b[cb][0]); FILLER(2) KPD_H_MFMA(acc[1][1], a[ca][3], b[cb][2]); FILLER(3)
        KPD_H_MFMA(acc[0][0], a[ca][0], b[cb][1]); FILLER(4) KPD_H_MFMA(acc[0][1], a[ca][0], b[cb][3]); FILLER(5)
        KPD_H_MFMA(acc[1][0], a[ca][1], b[cb][1]); FILLER(6) KPD_H_MFMA(acc[1][1], a[ca][1], b[cb][3]); FILLER(7)
        KPD_H_MFMA(acc[0][0], a[ca][0], b[cb][0]); FILLER(8) KPD_H_MFMA(acc[0][1], a[ca][0], b[cb][2]); FILLER(9)
        KPD_H_MFMA(acc[1][0], a[ca][1], b[cb][0]); FILLER(10) KPD_H_MFMA(acc[1][1], a[ca][1], b[cb][2]); FILLER(11)
        __builtin_amdgcn_sched_barrier(0);
    }
#undef FILLER
}

template <int FILL>
__global__ __launch_bounds__(256, 1) void k_time_fill(const void *Wh, int iters, unsigned long long *cyc, float *sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    _Float16 *Ah = reinterpret_cast<_Float16 *>(smem);
    for (int i = tid; i < 2 * PLANE_H; i += 256) Ah[i] = (_Float16)(0.001f * (float)((i * 7 + blockIdx.x) & 255));
    __syncthreads();
    f32x16 acc[2][2];
    acc_zero(acc);
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = 0.01f * (float)(lane + i);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) gemm_fill<FILL>(Ah, Wh, acc, wave, lane, f);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    float v = 0.f;
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) v += acc[m][n][r];
    for (int i = 0; i < 8; ++i) v += f[i];
    if (v == 12345.678f) sink[tid] = v;
}

template <int FILL>
static void time_fill(const void *dWh) {
    const int blocks = 256, iters = 200;
    unsigned long long *cyc; float *sink;
    hipMalloc(&cyc, blocks * 8); hipMalloc(&sink, 4096);
    hipFuncSetAttribute((const void *)k_time_fill<FILL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k_time_fill<FILL>, dim3(blocks), dim3(256), 150 * 1024, 0, dWh, 2, cyc, sink);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_time_fill<FILL>, dim3(blocks), dim3(256), 150 * 1024, 0, dWh, iters, cyc, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto c : h) sum += (double)c;
    const double per = sum / blocks / iters;
    printf("one workgroup per CU, %2d VALU fillers per MFMA (%5d per tile GEMM): %8.0f ticks per tile GEMM (%5.1f per MFMA)\n", FILL, FILL * 204, per, per / 204.0);
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
    time_fill<0>(dWh); time_fill<2>(dWh); time_fill<4>(dWh); time_fill<6>(dWh); time_fill<8>(dWh);
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
