// Shared pieces of the training engines (egnn_train.hip, gvp_train.hip): parameter table entries, row-major GEMM /
// GEMV wrappers over rocBLAS on the caller's stream, the split-K weight-gradient GEMM, and the generic elementwise,
// column-sum and reduction kernels.  Included by exactly those translation units; everything is file-local.
#pragma once
#include <rocblas/rocblas.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace kpd {
namespace {

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigm(x); }
__device__ __forceinline__ float silu_grad(float x) {
    const float s = sigm(x);
    return s * (1.0f + x * (1.0f - s));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// Y += b (kept: the pre-activation), A = SiLU(Y)
__global__ void k_bias_silu(float *__restrict__ Y, const float *__restrict__ b, long long total, int cols, int ld,
                            float *__restrict__ A) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
    const float v = Y[(size_t)r * ld + c] + b[c];
    Y[(size_t)r * ld + c] = v;
    A[(size_t)r * ld + c] = silu_f(v);
}

__global__ void k_bias_add(float *__restrict__ Y, const float *__restrict__ b, long long total, int cols, int ld) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
    Y[(size_t)r * ld + c] += b[c];
}

__global__ void k_copy_rows(const float *__restrict__ src, int lds, float *__restrict__ dst, int ldd, long long total, int cols) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
    dst[(size_t)r * ldd + c] = src[(size_t)r * lds + c];
}

__global__ void k_fill(float *__restrict__ p, float v, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---- backward kernels --------------------------------------------------------------------------------------------------
// dY *= SiLU'(pre), in place
__global__ void k_silu_bwd(float *__restrict__ dY, const float *__restrict__ pre, long long total, int cols, int ld) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
    dY[(size_t)r * ld + c] *= silu_grad(pre[(size_t)r * ld + c]);
}

// out[0] += sum of v[0..n): one workgroup, double accumulation, fixed order (a lone scalar gradient such as the attention
// bias is a heavily cancelling sum over all edges; float atomics in arbitrary order cost it two to three digits)
__global__ void k_sum_scalar(const float *__restrict__ v, int n, float *__restrict__ out) {
    __shared__ double part[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += (double)v[i];
#pragma unroll
    for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += part[w];
        out[0] += (float)t;
    }
}

__global__ void k_sub_inplace(float *__restrict__ a, const float *__restrict__ b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] -= b[i];
}

// y[c * incy] += sum over rows r of A[r][c] * (x ? x[r] : 1): bias gradients and the A^T x products of the heads.
// Tall-skinny (hundreds of thousands of rows, <= 512 columns): a row block per workgroup, columns across threads
// (coalesced row reads), one atomic per column and workgroup.
constexpr int COLSUM_ROWS = 256, COLSUM_THREADS = 320;       // 320 threads: the 257 columns of a layer in one pass
__global__ void k_colsum(const float *__restrict__ A, int lda, const float *__restrict__ x, int M, int K, float *__restrict__ y,
                         int incy, float *__restrict__ y2) {
    const int r0 = blockIdx.x * COLSUM_ROWS, r1 = min(M, r0 + COLSUM_ROWS);
    for (int c = threadIdx.x; c < K; c += blockDim.x) {
        float s0 = 0.0f, s1 = 0.0f, p0 = 0.0f, p1 = 0.0f;       // s: weighted by x (or plain), p: plain sums when both are wanted
        int r = r0;
        for (; r + 1 < r1; r += 2) {
            const float a0 = A[(size_t)r * lda + c], a1 = A[(size_t)(r + 1) * lda + c];
            s0 = fmaf(a0, x ? x[r] : 1.0f, s0);
            s1 = fmaf(a1, x ? x[r + 1] : 1.0f, s1);
            p0 += a0;
            p1 += a1;
        }
        if (r < r1) {
            const float a0 = A[(size_t)r * lda + c];
            s0 = fmaf(a0, x ? x[r] : 1.0f, s0);
            p0 += a0;
        }
        if (y) atomicAdd(&y[(size_t)c * incy], s0 + s1);
        if (y2) atomicAdd(&y2[c], p0 + p1);
    }
}

// dst[i] += sum over s of part[s][i]
__global__ void k_reduce_parts(const float *__restrict__ part, int n_parts, int rows, int cols, float *__restrict__ dst, int ldd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    float s = 0.0f;
    for (int k = 0; k < n_parts; ++k) s += part[(size_t)k * rows * cols + i];
    const int r = i / cols, c = i - r * cols;
    dst[(size_t)r * ldd + c] += s;
}

inline dim3 grid1(long long n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }

struct Param {
    const float *w = nullptr;
    float *g = nullptr;
    int rows = 0, cols = 0;
};

// what the wrappers need from an engine: the BLAS handle, the stream of the current call, split-K scratch, a ones vector
struct TrainCtx {
    rocblas_handle blas = nullptr;
    hipStream_t st = nullptr;
    float *part = nullptr;
    size_t part_floats = 0;
    float *ones = nullptr;
    std::map<std::string, Param> params;
};

#define KPD_BLAS(call)                                                                            \
    do {                                                                                          \
        rocblas_status s_ = (call);                                                               \
        if (s_ != rocblas_status_success) {                                                       \
            kpd::set_error("%s:%d: %s -> rocblas status %d", __FILE__, __LINE__, #call, (int)s_); \
            return KPD_ERR_HIP;                                                                   \
        }                                                                                         \
    } while (0)

// row-major C[M,N] = alpha op(A) op(B) + beta C
kpd_status gemm(TrainCtx *T, bool tA, bool tB, int M, int N, int K, const float *A, int lda, const float *B, int ldb,
                float beta, float *C, int ldc, float alpha = 1.0f) {
    if (M == 0 || N == 0) return KPD_OK;
    if (K == 0) {
        if (beta == 0.0f) KPD_HIP(hipMemset2DAsync(C, (size_t)ldc * 4, 0, (size_t)N * 4, M, T->st));
        return KPD_OK;
    }
    KPD_BLAS(rocblas_sgemm(T->blas, tB ? rocblas_operation_transpose : rocblas_operation_none,
                           tA ? rocblas_operation_transpose : rocblas_operation_none, N, M, K, &alpha, B, ldb, A, lda, &beta, C, ldc));
    return KPD_OK;
}

// y[M] (stride incy) = beta y + A[M,K] x (stride incx), A row-major
kpd_status gemv_n(TrainCtx *T, int M, int K, const float *A, int lda, const float *x, int incx, float beta, float *y,
                  int incy) {
    if (M == 0) return KPD_OK;
    const float alpha = 1.0f;
    KPD_BLAS(rocblas_sgemv(T->blas, rocblas_operation_transpose, K, M, &alpha, A, lda, x, incx, &beta, y, incy));
    return KPD_OK;
}

// y[K] (stride incy) += A[M,K]^T x[M] (x = nullptr: column sums), A row-major
kpd_status gemv_t_acc(TrainCtx *T, int M, int K, const float *A, int lda, const float *x, float *y, int incy) {
    if (M == 0 || !y) return KPD_OK;
    hipLaunchKernelGGL(k_colsum, dim3(cdiv(M, COLSUM_ROWS)), dim3(COLSUM_THREADS), 0, T->st, A, lda, x, M, K, y, incy, (float *)nullptr);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// y[K] (stride incy) += A^T x and y2[K] += column sums of A, in one pass over A
kpd_status gemv_t_colsum_acc(TrainCtx *T, int M, int K, const float *A, int lda, const float *x, float *y, int incy, float *y2) {
    if (M == 0 || (!y && !y2)) return KPD_OK;
    hipLaunchKernelGGL(k_colsum, dim3(cdiv(M, COLSUM_ROWS)), dim3(COLSUM_THREADS), 0, T->st, A, lda, x, M, K, y, incy, y2);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status colsum_acc(TrainCtx *T, int M, int K, const float *A, int lda, float *y) { return gemv_t_acc(T, M, K, A, lda, nullptr, y, 1); }

// weight gradient C[M,N] += A[K,M]^T B[K,N] with K = rows of a tall activation matrix: the output is a few tiles only, so
// K is split over GRAD_SPLIT batches (one strided-batched GEMM into partial sums) and the partials are reduced
constexpr int GRAD_SPLIT = 48;
kpd_status grad_gemm(TrainCtx *T, int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc) {
    if (!C || M == 0 || N == 0 || K == 0) return KPD_OK;
    const int chunk = K / GRAD_SPLIT;
    if (chunk < 128 || (size_t)M * N > T->part_floats / GRAD_SPLIT) return gemm(T, true, false, M, N, K, A, lda, B, ldb, 1.0f, C, ldc);
    const float one = 1.0f, zero = 0.0f;
    KPD_BLAS(rocblas_sgemm_strided_batched(T->blas, rocblas_operation_none, rocblas_operation_transpose, N, M, chunk, &one, B, ldb,
                                           (rocblas_stride)chunk * ldb, A, lda, (rocblas_stride)chunk * lda, &zero, T->part, N,
                                           (rocblas_stride)M * N, GRAD_SPLIT));
    hipLaunchKernelGGL(k_reduce_parts, grid1((long long)M * N), dim3(256), 0, T->st, T->part, GRAD_SPLIT, M, N, C, ldc);
    KPD_LAUNCH_CHECK();
    const int done = chunk * GRAD_SPLIT;
    if (done < K) KPD_TRY(grad_gemm(T, M, N, K - done, A + (size_t)done * lda, lda, B + (size_t)done * ldb, ldb, C, ldc));
    return KPD_OK;
}

kpd_status param(TrainCtx *T, const std::string &name, int rows, int cols, Param *out) {
    auto it = T->params.find(name);
    KPD_REQUIRE(it != T->params.end(), KPD_ERR_WEIGHTS, "parameter %s was not bound", name.c_str());
    KPD_REQUIRE(it->second.rows == rows && it->second.cols == cols, KPD_ERR_WEIGHTS, "parameter %s is [%d,%d], expected [%d,%d]",
                name.c_str(), it->second.rows, it->second.cols, rows, cols);
    *out = it->second;
    return KPD_OK;
}

}  // namespace
}  // namespace kpd
