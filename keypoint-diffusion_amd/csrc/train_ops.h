// Shared pieces of the training engines (egnn_train.hip, gvp_train.hip): parameter table entries, row-major GEMM /
// GEMV wrappers over the library's own MFMA GEMM (sgemm.hip) on the caller's stream, the split-K weight-gradient GEMM, and the generic elementwise,
// column-sum and reduction kernels.  Included by exactly those translation units; everything is file-local.
#pragma once

#include <stdlib.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "sgemm.h"

namespace kpd {
namespace {

// v_exp_f32 + v_rcp_f32 (1 ulp each) instead of the IEEE division sequence: a third of the VALU work of every activation / derivative
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
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

// out[0] += sum of v[0..n): double accumulation in a fixed order (a lone scalar gradient such as the attention bias is a heavily
// cancelling sum over all edges; float atomics in arbitrary order cost it two to three digits).  gridDim.x workgroups sum contiguous
// chunks; the one that arrives last (an integer ticket, the only atomic) adds the chunk sums in chunk order and resets the ticket, so
// the result depends on the chunking alone.  (One workgroup over all edges was 33 us per call, latency-bound on one CU.)
__global__ __launch_bounds__(1024) void k_sum_scalar(const float *__restrict__ v, int n, double *__restrict__ part, int *__restrict__ ticket,
                                                     float *__restrict__ out) {
    __shared__ double ws[16];
    __shared__ int s_last;
    const int chunk = (n + (int)gridDim.x - 1) / (int)gridDim.x, i0 = (int)blockIdx.x * chunk, i1 = min(n, i0 + chunk);
    double s = 0.0;
    for (int i = i0 + (int)threadIdx.x; i < i1; i += (int)blockDim.x) s += (double)v[i];
#pragma unroll
    for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += ws[w];
        __hip_atomic_store(&part[blockIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        s_last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
        if (s_last) {
            __threadfence();
            double tot = 0.0;
            for (int b = 0; b < (int)gridDim.x; ++b) tot += __hip_atomic_load(&part[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out[0] += (float)tot;
            *ticket = 0;
        }
    }
}

__global__ void k_sub_inplace(float *__restrict__ a, const float *__restrict__ b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] -= b[i];
}

// y[c * incy] += sum over rows r of A[r][c] * (x ? x[r] : 1): bias gradients and the A^T x products of the heads.
// Tall-skinny (hundreds of thousands of rows, <= 512 columns): a row block per workgroup, columns across threads
// (coalesced row reads).  Every workgroup writes its partial sums to part[block][0 | 1][c]; k_colsum_reduce adds the blocks in
// block order, so the result is bitwise reproducible (float atomics in arrival order were not).
constexpr int COLSUM_ROWS = 256, COLSUM_THREADS = 320;       // 320 threads: the 257 columns of a layer in one pass
constexpr int COLSUM_LD = 1088;                              // columns per partial row (K <= 1088: four 264-wide slots of the EGNN trainer's batched sums)
constexpr int HEAD_ROWS = 64;                                // rows per workgroup of the head-backward kernels that emit partials too
__global__ void k_colsum(const float *__restrict__ A, int lda, const float *__restrict__ x, int M, int K, int rows_per_block, float *__restrict__ part) {
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float *p = part + (size_t)blockIdx.x * 2 * COLSUM_LD;
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
        p[c] = s0 + s1;
        p[COLSUM_LD + c] = p0 + p1;
    }
}

// one workgroup of 64 columns x 16 block-slices: slice g adds the partials of blocks g, g + 16, ... in order, the 16 slice sums
// are combined in slice order -- a fixed tree, so the result does not depend on timing
__device__ __forceinline__ void colsum_reduce_body(const float *__restrict__ part, int blocks, int K, float *__restrict__ y, int incy,
                                                   float *__restrict__ y2) {
    __shared__ float s_s[16][64], s_p[16][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    float s = 0.0f, p = 0.0f;
    if (c < K) {
        int b = g;
        for (; b + 48 < blocks; b += 64) {                       // four partials in flight, added in block order as below
            float ts[4], tp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ts[u] = part[(size_t)(b + 16 * u) * 2 * COLSUM_LD + c];
                tp[u] = part[(size_t)(b + 16 * u) * 2 * COLSUM_LD + COLSUM_LD + c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s += ts[u];
                p += tp[u];
            }
        }
        for (; b < blocks; b += 16) {
            s += part[(size_t)b * 2 * COLSUM_LD + c];
            p += part[(size_t)b * 2 * COLSUM_LD + COLSUM_LD + c];
        }
    }
    s_s[g][cl] = s;
    s_p[g][cl] = p;
    __syncthreads();
    if (g == 0 && c < K) {
        float ts = 0.0f, tp = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            ts += s_s[k][cl];
            tp += s_p[k][cl];
        }
        if (y) y[(size_t)c * incy] += ts;
        if (y2) y2[c] += tp;
    }
}
__global__ __launch_bounds__(1024) void k_colsum_reduce(const float *__restrict__ part, int blocks, int K, float *__restrict__ y, int incy,
                                                        float *__restrict__ y2) {
    colsum_reduce_body(part, blocks, K, y, incy, y2);
}
// several such reductions in one launch (blockIdx.y = which): the per-(edge type, branch) head / bias gradients of an EGNN layer
struct ColsumRedBatch {
    struct One {
        const float *part;
        int blocks, K, incy;
        float *y, *y2;
    } r[8];
};
__global__ __launch_bounds__(1024) void k_colsum_reduce_batch(ColsumRedBatch b) {
    const ColsumRedBatch::One &o = b.r[blockIdx.y];
    colsum_reduce_body(o.part, o.blocks, o.K, o.y, o.incy, o.y2);
}

// ---- edges grouped by SOURCE node (the engines' edge lists are dst-sorted) ------------------------------------------------
// Sums over the out-edges of a node (the gradients that flow back to h_src / x_src) used to be float atomics; they are
// segmented sums over this index now: perm lists the edge ids of each source node in ascending order, rowptr [n_src + 1].
// Built per forward: integer histogram (atomics on counts: order-free), scan, scatter, then every node sorts its own
// short segment, which makes the summation order a function of the graph alone.
struct SrcCsr {
    int *rowptr = nullptr, *perm = nullptr;
};

__global__ void k_idx_count(const int *__restrict__ idx, int E, int *__restrict__ cnt) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) atomicAdd(&cnt[idx[e]], 1);
}

__global__ void k_idx_scan(const int *cnt, int n, int *__restrict__ rowptr, int *cursor) {      // cnt may alias cursor
    __shared__ int s_part[1024];
    __shared__ int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += blockDim.x) {
        const int i = base + threadIdx.x;
        const int v = i < n ? cnt[i] : 0;
        s_part[threadIdx.x] = v;
        __syncthreads();
        for (int st = 1; st < (int)blockDim.x; st <<= 1) {
            const int t = (int)threadIdx.x >= st ? s_part[threadIdx.x - st] : 0;
            __syncthreads();
            s_part[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n) {
            rowptr[i] = s_carry + s_part[threadIdx.x] - v;
            cursor[i] = rowptr[i];
        }
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) s_carry += s_part[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) rowptr[n] = s_carry;
}

__global__ void k_idx_fill(const int *__restrict__ idx, int E, int *__restrict__ cursor, int *__restrict__ perm) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) perm[atomicAdd(&cursor[idx[e]], 1)] = e;
}

// every source node orders its own segment by edge id: one wave per node, segment staged in LDS, each element's final
// position is its rank (ids are unique); segments longer than SEG_LDS fall back to an in-place insertion sort by one lane
constexpr int SEG_LDS = 2048;
__global__ __launch_bounds__(64) void k_idx_sort_segments(const int *__restrict__ rowptr, int n, int *__restrict__ perm) {
    __shared__ int s_key[SEG_LDS];
    const int v = blockIdx.x;
    const int lo = rowptr[v], d = rowptr[v + 1] - lo;
    if (d <= 1) return;
    if (d <= SEG_LDS) {
        for (int i = threadIdx.x; i < d; i += 64) s_key[i] = perm[lo + i];
        __syncthreads();
        for (int i = threadIdx.x; i < d; i += 64) {
            const int key = s_key[i];
            int rank = 0;
            for (int j = 0; j < d; ++j) rank += s_key[j] < key;
            perm[lo + rank] = key;
        }
    } else if (threadIdx.x == 0) {
        for (int i = lo + 1; i < lo + d; ++i) {
            const int key = perm[i];
            int j = i - 1;
            while (j >= lo && perm[j] > key) {
                perm[j + 1] = perm[j];
                --j;
            }
            perm[j + 1] = key;
        }
    }
}

// Segmented sums over the rows of a node:
//   out[v][c] (+)= f_v * sum over j in [rowptr[v], rowptr[v + 1]) of M[perm ? perm[j] : j][c0 + c],  c < cols,  f_v = alpha * (scale ? scale[v] : 1)
// A group of 2^gshift lanes owns a node (a wave for 256 columns, 16 lanes for the 48 of a vector row, 4 for coordinates), a lane VEC
// columns (16-byte pieces where the operands allow); rows are fetched four at a time and added in segment order, so the result is a
// function of the graph alone.  accumulate: nodes without rows are left untouched.
template <int VEC>
__global__ __launch_bounds__(256) void k_segsum_g(const float *__restrict__ M, int lda, int c0, int cw, const int *__restrict__ perm,
                                                  const int *__restrict__ rowptr, const float *__restrict__ scale, float alpha, int accumulate, int n,
                                                  float *__restrict__ out, int ldo, int gshift) {
    typedef float vt __attribute__((ext_vector_type(VEC)));
    const int v = blockIdx.x * (256 >> gshift) + (threadIdx.x >> gshift), lane = threadIdx.x & ((1 << gshift) - 1);
    if (v >= n) return;
    const int lo = rowptr[v], hi = rowptr[v + 1];
    if (lo == hi && accumulate) return;
    const float f = scale ? alpha * scale[v] : alpha;
    for (int c = lane; c < cw; c += 1 << gshift) {
        const float *base = M + c0 + VEC * c;
        vt s = 0.0f;
        int j = lo;
        for (; j + 4 <= hi; j += 4) {
            vt m[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) m[k] = *reinterpret_cast<const vt *>(base + (size_t)(perm ? perm[j + k] : j + k) * lda);
#pragma unroll
            for (int k = 0; k < 4; ++k) s += m[k];
        }
        for (; j < hi; ++j) s += *reinterpret_cast<const vt *>(base + (size_t)(perm ? perm[j] : j) * lda);
        vt *o = reinterpret_cast<vt *>(out + (size_t)v * ldo + VEC * c);
        if (accumulate) *o += s * f;
        else *o = s * f;
    }
}

inline kpd_status segsum(hipStream_t st, const float *M, int lda, int c0, int cols, const int *perm, const int *rowptr, const float *scale, float alpha,
                         bool accumulate, int n, float *out, int ldo) {
    if (n <= 0 || cols <= 0) return KPD_OK;
    const bool vec = (cols & 3) == 0 && (c0 & 3) == 0 && (lda & 3) == 0 && (ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(M) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    const int cw = vec ? cols / 4 : cols;
    int gshift = 0;
    while ((1 << gshift) < cw && gshift < 6) ++gshift;
    const dim3 grid((unsigned)((n + (256 >> gshift) - 1) / (256 >> gshift)));
    if (vec) hipLaunchKernelGGL(k_segsum_g<4>, grid, dim3(256), 0, st, M, lda, c0, cw, perm, rowptr, scale, alpha, accumulate ? 1 : 0, n, out, ldo, gshift);
    else hipLaunchKernelGGL(k_segsum_g<1>, grid, dim3(256), 0, st, M, lda, c0, cw, perm, rowptr, scale, alpha, accumulate ? 1 : 0, n, out, ldo, gshift);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// out[r] = A[idx[r]] * (scale ? scale[idx[r]] : 1), rows `cols` wide; 16-byte pieces where the operands allow
template <int VEC>
__global__ void k_gather_rows_g(const float *__restrict__ A, const int *__restrict__ idx, const float *__restrict__ scale, int rows, int cw,
                                float *__restrict__ out) {
    typedef float vt __attribute__((ext_vector_type(VEC)));
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)rows * cw) return;
    const int r = (int)(i / cw), c = (int)(i - (long long)r * cw);
    const int v = idx[r];
    const vt a = reinterpret_cast<const vt *>(A)[(size_t)v * cw + c];
    reinterpret_cast<vt *>(out)[i] = scale ? a * scale[v] : a;
}

inline kpd_status gather_rows(hipStream_t st, const float *A, const int *idx, const float *scale, int rows, int cols, float *out) {
    if (rows <= 0 || cols <= 0) return KPD_OK;
    const bool vec = (cols & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    const int cw = vec ? cols / 4 : cols;
    const dim3 grid((unsigned)(((long long)rows * cw + 255) / 256));
    if (vec) hipLaunchKernelGGL(k_gather_rows_g<4>, grid, dim3(256), 0, st, A, idx, scale, rows, cw, out);
    else hipLaunchKernelGGL(k_gather_rows_g<1>, grid, dim3(256), 0, st, A, idx, scale, rows, cw, out);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

inline dim3 grid1(long long n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }

struct Param {
    const float *w = nullptr;
    float *g = nullptr;
    int rows = 0, cols = 0;
};

// ---- reference tensors narrower than the engine's fixed layout -------------------------------------------------------------------
// The engines compute in fixed widths (EGNN: 256 hidden columns + the timestep column; GVP: 16 vector channels).  A model with a
// narrower reference shape (LigRecDynamics' default hidden_nf = 255, models/dynamics.py:300-302; vector_size < 16, models/
// dynamics_gvp.py:106-108) is trained through zero padding: every bound tensor whose axes carry such a width gets an engine-owned
// wide copy (weights) and a wide gradient; one batched launch stages all weights before the forward pass, one zeroes the wide
// gradients before the backward pass and one adds them into the caller's gradient tensors, in the reference shape, after it.
// Zero rows / columns keep the padding inert: padded activations are exactly 0 (SiLU(0) = 0, gates x 0), the normalisations
// skip them explicitly (their kernels take the live width), and whatever reaches a padded gradient entry is dropped by the map.
struct AxisSeg { int ref_n, wide_n; };          // ref_n consecutive reference entries at the start of wide_n engine entries

struct WideDesc {
    const float *ref_w;
    float *ref_g, *w, *g;
    const int *rmap, *cmap;                     // engine index -> reference index or -1
    int R, C, ref_ld, blk0;
};

// mode 0: w = widened ref_w; 1: g = 0; 2: ref_g += the mapped entries of g
__global__ void k_wide_batch(const WideDesc *__restrict__ tab, int n_tab, int mode) {
    int lo = 0, hi = n_tab - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].blk0 <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const WideDesc d = tab[lo];
    const int i = ((int)blockIdx.x - d.blk0) * 256 + (int)threadIdx.x;
    if (i >= d.R * d.C) return;
    const int r = d.rmap[i / d.C], c = d.cmap[i % d.C];
    const bool live = r >= 0 && c >= 0;
    if (mode == 0) d.w[i] = live ? d.ref_w[(size_t)r * d.ref_ld + c] : 0.0f;
    else if (!d.g) return;
    else if (mode == 1) d.g[i] = 0.0f;
    else if (live) d.ref_g[(size_t)r * d.ref_ld + c] += d.g[i];
}

struct WideSet {
    struct Entry {
        WideDesc d{};
        int *maps_dev = nullptr;
        float *w_dev = nullptr, *g_dev = nullptr;
        int ref_rows = 0, ref_cols = 0;
    };
    static constexpr int RING = 4;
    std::map<std::string, int> index;
    std::vector<Entry> e;
    WideDesc *tab_host[RING] = {nullptr, nullptr, nullptr, nullptr}, *tab_dev[RING] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t done[RING] = {nullptr, nullptr, nullptr, nullptr};
    int tab_cap = 0, turn = 0;

    void release() {
        for (Entry &x : e)
            if (x.maps_dev) (void)hipFree(x.maps_dev);          // one allocation per entry: maps, wide weight, wide gradient
        e.clear();
        index.clear();
        for (int k = 0; k < RING; ++k) {
            if (tab_host[k]) (void)hipHostFree(tab_host[k]);
            if (tab_dev[k]) (void)hipFree(tab_dev[k]);
            if (done[k]) (void)hipEventDestroy(done[k]);
            tab_host[k] = tab_dev[k] = nullptr;
            done[k] = nullptr;
        }
        tab_cap = 0;
    }
};

inline std::vector<std::string> split_name(const std::string &s, char c = '.') {
    std::vector<std::string> out;
    size_t a = 0;
    for (;;) {
        const size_t b = s.find(c, a);
        out.push_back(s.substr(a, b == std::string::npos ? b : b - a));
        if (b == std::string::npos) return out;
        a = b + 1;
    }
}

inline std::vector<int> axis_map(const std::vector<AxisSeg> &segs, int *ref_len) {
    std::vector<int> map;
    int ref = 0;
    for (const AxisSeg &s : segs) {
        for (int j = 0; j < s.wide_n; ++j) map.push_back(j < s.ref_n ? ref + j : -1);
        ref += s.ref_n;
    }
    *ref_len = ref;
    return map;
}

// what the wrappers need from an engine: the stream of the current call, split-K scratch, a ones vector
struct TrainCtx {
    hipStream_t st = nullptr;
    float *part = nullptr;
    size_t part_floats = 0;
    float *ones = nullptr;
    float *colpart = nullptr;          // [colpart_blocks][2][COLSUM_LD] partial column sums (k_colsum -> k_colsum_reduce)
    int colpart_blocks = 0;
    std::map<std::string, Param> params;
    WideSet wide;
    double *ss_part = nullptr;         // sum_scalar: chunk sums and the arrival ticket (allocated at first use, freed by release_scratch)
    int *ss_ticket = nullptr;
    void release_scratch() {
        if (ss_part) (void)hipFree(ss_part);
        if (ss_ticket) (void)hipFree(ss_ticket);
        ss_part = nullptr;
        ss_ticket = nullptr;
    }
};

inline size_t colpart_floats(int max_rows) { return (size_t)cdiv(std::max(max_rows, 1), HEAD_ROWS) * 2 * COLSUM_LD; }

// row-major C[M,N] = alpha op(A) op(B) + beta C
// silu_pre: C = (.) * SiLU'(silu_pre), element for element (silu_pre laid out like C): replaces a k_silu_bwd pass over C
// bias / act_out: C = (.) + bias[n], act_out = SiLU(C) (laid out like C): replaces a k_bias_silu / k_bias_add pass over C
kpd_status gemm(TrainCtx *T, bool tA, bool tB, int M, int N, int K, const float *A, int lda, const float *B, int ldb,
                float beta, float *C, int ldc, float alpha = 1.0f, const float *silu_pre = nullptr, const float *bias = nullptr,
                float *act_out = nullptr) {
    if (M == 0 || N == 0) return KPD_OK;
    if (K == 0 && (bias || act_out)) {
        KPD_REQUIRE(beta == 1.0f && !silu_pre && bias, KPD_ERR_INVALID, "gemm: empty K with a bias epilogue needs beta = 1 and a bias");
        if (act_out) hipLaunchKernelGGL(k_bias_silu, grid1((long long)M * N), dim3(256), 0, T->st, C, bias, (long long)M * N, N, ldc, act_out);
        else hipLaunchKernelGGL(k_bias_add, grid1((long long)M * N), dim3(256), 0, T->st, C, bias, (long long)M * N, N, ldc);
        KPD_LAUNCH_CHECK();
        return KPD_OK;
    }
    if (K == 0 && silu_pre) {
        KPD_REQUIRE(beta == 1.0f, KPD_ERR_INVALID, "gemm: empty K with an activation epilogue needs beta = 1");
        hipLaunchKernelGGL(k_silu_bwd, grid1((long long)M * N), dim3(256), 0, T->st, C, silu_pre, (long long)M * N, N, ldc);
        KPD_LAUNCH_CHECK();
        return KPD_OK;
    }
    if (K == 0) {
        if (beta == 0.0f) KPD_HIP(hipMemset2DAsync(C, (size_t)ldc * 4, 0, (size_t)N * 4, M, T->st));
        return KPD_OK;
    }
    // (the scratch lets a product of few tiles be cut along K: sgemm_split_slices; an activation epilogue keeps the product whole)
    // a product of a handful of tiles (ligand-sized: 1 600 rows) gains more from the split along K than from a fused epilogue, and the
    // epilogues live in the unsplit kernel: such a product takes the split and the elementwise pass
    static const bool fuse = tool_env_int("KPD_TRAIN_EPI", 1) != 0;          // A/B runs (TOOLS build): 0 = never fuse
    if ((silu_pre || bias || act_out) && (!fuse || (cdiv(M, 128) * cdiv(N, 128) * 4 <= cu_count() && K >= 128))) {
        KPD_TRY(sgemm(tA, tB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, T->st, T->part, T->part_floats));
        const long long tot = (long long)M * N;
        if (silu_pre) hipLaunchKernelGGL(k_silu_bwd, grid1(tot), dim3(256), 0, T->st, C, silu_pre, tot, N, ldc);
        if (act_out) hipLaunchKernelGGL(k_bias_silu, grid1(tot), dim3(256), 0, T->st, C, bias, tot, N, ldc, act_out);
        else if (bias) hipLaunchKernelGGL(k_bias_add, grid1(tot), dim3(256), 0, T->st, C, bias, tot, N, ldc);
        KPD_LAUNCH_CHECK();
        return KPD_OK;
    }
    return sgemm(tA, tB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, T->st, T->part, T->part_floats, nullptr, silu_pre, bias, act_out);
}

// y[M] (stride incy) = beta y + A[M,K] x (stride incx), A row-major
kpd_status gemv_n(TrainCtx *T, int M, int K, const float *A, int lda, const float *x, int incx, float beta, float *y,
                  int incy) {
    if (M == 0) return KPD_OK;
    return sgemv_rows(M, K, A, lda, x, incx, beta, y, incy, T->st);
}

// y[K] (stride incy) += A[M,K]^T x[M] (x = nullptr: column sums), A row-major
kpd_status gemv_t_colsum_acc(TrainCtx *T, int M, int K, const float *A, int lda, const float *x, float *y, int incy, float *y2);
kpd_status gemv_t_acc(TrainCtx *T, int M, int K, const float *A, int lda, const float *x, float *y, int incy) {
    return gemv_t_colsum_acc(T, M, K, A, lda, x, y, incy, nullptr);
}

// y[K] (stride incy) += A^T x and y2[K] += column sums of A, in one pass over A
kpd_status gemv_t_colsum_acc(TrainCtx *T, int M, int K, const float *A, int lda, const float *x, float *y, int incy, float *y2) {
    if (M == 0 || (!y && !y2)) return KPD_OK;
    // node-sized matrices (<= 20 800 rows at the contract shape) in 64-row blocks: 256-row blocks left two thirds of the CUs idle
    // (82 workgroups, 37 us per call, ~2.4 ms of an EGNN training step); edge-sized ones keep 256 rows per block
    // ... and 16-row blocks when such a matrix is wide as well (the EGNN trainer's 1 088-column sums over 19 200 nodes: 300 workgroups each
    // walked 64 rows four column passes long -- 61 us, latency-bound at 1.4 TB/s)
    int rows_per_block = M <= 65536 ? HEAD_ROWS : COLSUM_ROWS;
    if (M <= 65536 && K >= 512 && cdiv(M, 16) <= T->colpart_blocks) rows_per_block = 16;
    const int blocks = cdiv(M, rows_per_block);
    KPD_REQUIRE(K <= COLSUM_LD && blocks <= T->colpart_blocks && T->colpart, KPD_ERR_CAPACITY,
                "column-sum scratch too small (%d row blocks of %d, %d columns)", blocks, T->colpart_blocks, K);
    hipLaunchKernelGGL(k_colsum, dim3(blocks), dim3(COLSUM_THREADS), 0, T->st, A, lda, x, M, K, rows_per_block, T->colpart);
    KPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_colsum_reduce, dim3(cdiv(K, 64)), dim3(1024), 0, T->st, T->colpart, blocks, K, y, incy, y2);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// edges of one type grouped by source node: perm [E], rowptr [n_src + 1]; cursor: [n_src] scratch
kpd_status build_src_csr(TrainCtx *T, const int *src, int E, int n_src, int *cursor, const SrcCsr &out) {
    KPD_HIP(hipMemsetAsync(cursor, 0, (size_t)n_src * sizeof(int), T->st));
    if (E > 0) {
        hipLaunchKernelGGL(k_idx_count, grid1(E), dim3(256), 0, T->st, src, E, cursor);
        KPD_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_idx_scan, dim3(1), dim3(1024), 0, T->st, cursor, n_src, out.rowptr, cursor);
    KPD_LAUNCH_CHECK();
    if (E > 0) {
        hipLaunchKernelGGL(k_idx_fill, grid1(E), dim3(256), 0, T->st, src, E, cursor, out.perm);
        KPD_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_idx_sort_segments, dim3(n_src), dim3(64), 0, T->st, out.rowptr, n_src, out.perm);
        KPD_LAUNCH_CHECK();
    }
    return KPD_OK;
}

kpd_status colsum_acc(TrainCtx *T, int M, int K, const float *A, int lda, float *y) { return gemv_t_acc(T, M, K, A, lda, nullptr, y, 1); }

constexpr size_t GRAD_PART_FLOATS = (size_t)260 * (256 * 256 + 1024);     // split-K scratch of an engine: one slice per CU of a 257 x 257 gradient with its fringe and column-sum shares (k_sgemm_tn256)

// weight gradient C[M,N] += A[K,M]^T B[K,N] with K = rows of a tall activation matrix: the output is a few tiles only, so K is
// cut into slices (grid.z of one launch, partial products in scratch) that are summed in slice order -- no atomics
// bias_grad (optional): += the column sums of A, i.e. the gradient of the Linear's bias, in the same pass (replaces a colsum_acc over A)
kpd_status grad_gemm(TrainCtx *T, int M, int N, int K, const float *A, int lda, const float *B, int ldb, float *C, int ldc,
                     float *bias_grad = nullptr) {
    if (M == 0 || K == 0) return KPD_OK;
    if (!C || N == 0) return bias_grad ? colsum_acc(T, K, M, A, lda, bias_grad) : KPD_OK;
    // (Round 4 tried leaving the split-K reductions pending and sending four at a time as one launch: 63.2 vs 62.4 ms per egnn_train
    // step, same call -- a reduction is ~9 us of HBM time for its 34 MB of partials, not launch overhead; the simpler form stays.)
    return sgemm(true, false, M, N, K, 1.0f, A, lda, B, ldb, 1.0f, C, ldc, T->st, T->part, T->part_floats, bias_grad);
}

// out[0] += sum of v[0..n) (k_sum_scalar)
kpd_status sum_scalar(TrainCtx *T, const float *v, int n, float *out) {
    if (n <= 0 || !out) return KPD_OK;
    constexpr int SS_MAX = 64;
    if (!T->ss_part) {
        KPD_HIP(hipMalloc(reinterpret_cast<void **>(&T->ss_part), SS_MAX * sizeof(double)));
        KPD_HIP(hipMalloc(reinterpret_cast<void **>(&T->ss_ticket), sizeof(int)));
        KPD_HIP(hipMemset(T->ss_ticket, 0, sizeof(int)));
    }
    const int blocks = std::max(1, std::min(SS_MAX, n / 2048));
    hipLaunchKernelGGL(k_sum_scalar, dim3(blocks), dim3(1024), 0, T->st, v, n, T->ss_part, T->ss_ticket, out);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// bind one reference tensor through a wide copy: `rows` / `cols` describe its axes (cols empty = a vector).  The entry (maps, wide
// weight, wide gradient) is created at the first bind of the name; later binds only swap the caller's pointers.
kpd_status bind_wide(TrainCtx *T, const char *name, const float *weight, float *grad, const int64_t *shape, int ndim,
                     const std::vector<AxisSeg> &rows, const std::vector<AxisSeg> &cols) {
    WideSet &W = T->wide;
    int ref_r = 0, ref_c = 1;
    auto it = W.index.find(name);
    if (it == W.index.end()) {
        std::vector<int> rmap = axis_map(rows, &ref_r), cmap = cols.empty() ? std::vector<int>{0} : axis_map(cols, &ref_c);
        WideSet::Entry x;
        x.ref_rows = ref_r; x.ref_cols = ref_c;
        x.d.R = (int)rmap.size(); x.d.C = (int)cmap.size(); x.d.ref_ld = ref_c;
        const size_t n = (size_t)x.d.R * x.d.C;
        // (one allocation for maps, weight and gradient: a failure leaves nothing behind)
        const size_t map_bytes = ((rmap.size() + cmap.size()) * 4 + 255) & ~size_t(255), mat_bytes = (n * 4 + 255) & ~size_t(255);
        char *blob = nullptr;
        KPD_HIP(hipMalloc(reinterpret_cast<void **>(&blob), map_bytes + 2 * mat_bytes));
        x.maps_dev = reinterpret_cast<int *>(blob);
        x.w_dev = reinterpret_cast<float *>(blob + map_bytes);
        x.g_dev = reinterpret_cast<float *>(blob + map_bytes + mat_bytes);
        if (hipMemcpy(x.maps_dev, rmap.data(), rmap.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(x.maps_dev + rmap.size(), cmap.data(), cmap.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(blob);
            set_error("bind_wide: copying the axis maps of %s failed", name);
            return KPD_ERR_HIP;
        }
        x.d.rmap = x.maps_dev; x.d.cmap = x.maps_dev + rmap.size(); x.d.w = x.w_dev;
        W.e.push_back(x);
        it = W.index.emplace(name, (int)W.e.size() - 1).first;
    }
    WideSet::Entry &x = W.e[it->second];
    const bool ok = cols.empty() ? (ndim == 1 && shape[0] == x.ref_rows) : (ndim == 2 && shape[0] == x.ref_rows && shape[1] == x.ref_cols);
    KPD_REQUIRE(ok, KPD_ERR_WEIGHTS, "parameter %s has the wrong shape for this configuration (expected [%d%s%s])", name, x.ref_rows,
                cols.empty() ? "" : ", ", cols.empty() ? "" : std::to_string(x.ref_cols).c_str());
    x.d.ref_w = weight; x.d.ref_g = grad; x.d.g = grad ? x.g_dev : nullptr;
    Param p;
    p.w = x.w_dev; p.g = x.d.g; p.rows = x.d.R; p.cols = cols.empty() ? 1 : x.d.C;
    T->params[name] = p;
    return KPD_OK;
}

// one batched launch over every wide entry (k_wide_batch modes); a no-op for a model in the engine's own widths
kpd_status wide_run(TrainCtx *T, int mode) {
    WideSet &W = T->wide;
    const int n = (int)W.e.size();
    if (n == 0) return KPD_OK;
    if (n > W.tab_cap) {
        for (int k = 0; k < WideSet::RING; ++k) {
            if (W.done[k]) KPD_HIP(hipEventSynchronize(W.done[k]));
            if (W.tab_host[k]) (void)hipHostFree(W.tab_host[k]);
            if (W.tab_dev[k]) (void)hipFree(W.tab_dev[k]);
            KPD_HIP(hipHostMalloc(reinterpret_cast<void **>(&W.tab_host[k]), (size_t)(n + 64) * sizeof(WideDesc), hipHostMallocDefault));
            KPD_HIP(hipMalloc(reinterpret_cast<void **>(&W.tab_dev[k]), (size_t)(n + 64) * sizeof(WideDesc)));
            if (!W.done[k]) KPD_HIP(hipEventCreateWithFlags(&W.done[k], hipEventDisableTiming));
        }
        W.tab_cap = n + 64;
    }
    const int k = W.turn;
    W.turn = (W.turn + 1) % WideSet::RING;
    KPD_HIP(hipEventSynchronize(W.done[k]));           // the slot's previous upload has been consumed (normally long ago)
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        W.e[i].d.blk0 = blocks;
        W.tab_host[k][i] = W.e[i].d;
        blocks += cdiv(W.e[i].d.R * W.e[i].d.C, 256);
    }
    KPD_HIP(hipMemcpyAsync(W.tab_dev[k], W.tab_host[k], (size_t)n * sizeof(WideDesc), hipMemcpyHostToDevice, T->st));
    hipLaunchKernelGGL(k_wide_batch, dim3(blocks), dim3(256), 0, T->st, W.tab_dev[k], n, mode);
    KPD_LAUNCH_CHECK();
    KPD_HIP(hipEventRecord(W.done[k], T->st));
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
