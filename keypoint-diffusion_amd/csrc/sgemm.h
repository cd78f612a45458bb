// General fp32 MFMA GEMM of the training engines (sgemm.hip).
#pragma once
#include "common.h"

namespace kpd {

constexpr int SGEMM_MAX_SPLIT = 256;

// row-major C[M,N] = alpha op(A) op(B) + beta C.  With a scratch buffer `part`, a product whose output is a few tiles only (weight
// gradients: K = edge count) is cut along K into slices computed by grid.z of one launch and summed in slice order (no atomics).
kpd_status sgemm(bool tA, bool tB, int M, int N, int K, float alpha, const float *A, int lda, const float *B, int ldb, float beta,
                 float *C, int ldc, hipStream_t st, float *part = nullptr, size_t part_floats = 0, float *colsum = nullptr,
                 const float *silu_pre = nullptr, const float *bias = nullptr, float *act_out = nullptr);
// colsum (A^T B products only): colsum[m] += sum_k A[k][m] in the same pass over A -- the bias gradient of the Linear whose weight
// gradient the product is; summed in slab / slice order like the product itself.
// silu_pre: C = (alpha op(A) op(B) + beta C) * SiLU'(silu_pre[m][n]), silu_pre laid out like C -- the backward of an activation whose
// pre-activation was kept, fused into the product that produces its upstream gradient.
// bias: + bias[n]; act_out: a second output SiLU(C), laid out like C -- the Linear + bias + SiLU of a forward pass in one kernel, with
// the pre-activation (C) kept for the backward pass
// slices that give every CU about two workgroups for a [M,N] output, bounded by K / 256
int sgemm_split_slices(int M, int N, int K);
// Weight gradients of a GVP message chain, several products per launch (k_wgrad_tnx), every output ACCUMULATED (+=):
//   C [256, 256] += A^T B;   Cx1 [256, nb2] += A^T B2 (nb2 <= 31);   colsum [256] += column sums of A;
//   Cx2 [na2, 256] += A2^T B (na2 <= 32);   colsum2 [na2] += column sums of A2
//   C == nullptr ("top" products, all of a batch or none): no 256 x 256 block, and a second narrow block Cx3 [256, nb3] += A^T B3 (nb3 <= 32)
// A, B: [K, 256] (16-byte aligned, leading dimensions multiples of 4); the narrow operands [K, ld] at any alignment.
struct WgradItem {
    const float *A, *B;
    int lda, ldb, K;
    float *C;
    int ldc;
    const float *B2;
    int ldb2, nb2;
    float *Cx1;
    int ldx1;
    float *colsum;
    const float *A2;
    int lda2, na2;
    float *Cx2;
    int ldx2;
    float *colsum2;
    const float *B3;
    int ldb3, nb3;
    float *Cx3;
    int ldx3;
};
kpd_status wgrad_batch(const WgradItem *items, int n, float *part, size_t part_floats, hipStream_t st);
// 257 x 257 weight gradients C += A^T B (A, B: [K, >= 257] with 16-byte aligned rows; colsum [257] += column sums of A, optional), up to
// eight per launch of k_sgemm_tn256_batch: the EGNN trainer's second-Linear gradients of a layer
struct Grad257Item {
    const float *A, *B;
    int lda, ldb, K;
    float *C;
    int ldc;
    float *colsum;
};
kpd_status grad257_batch(const Grad257Item *items, int n, float *part, size_t part_floats, hipStream_t st);
// y[m * incy] = beta y + sum_k A[m][k] x[k * incx]
kpd_status sgemv_rows(int M, int K, const float *A, int lda, const float *x, int incx, float beta, float *y, int incy, hipStream_t st);

}  // namespace kpd
