// Per-step ligand graph build on the GPU: lig-lig radius graph and keypoint->ligand kNN,
// emitted directly as dst-sorted COO + CSR in preallocated buffers (no host sync, fixed
// launch geometry).  Replaces torch_cluster.radius_graph / torch_cluster.knn + DGL
// add_edges in models/dynamics.py:387-420 and models/dynamics_gvp.py:201-234.
//
// One workgroup per complex; ligand coordinates are staged in LDS.  These kernels are tiny
// (<= a few thousand distance tests per complex) and latency bound.
#include <algorithm>

#include "common.h"

namespace kpd {

// ---- ll: radius graph ---------------------------------------------------------------------
// Pass 1: in-degree of every ligand atom (neighbours j != i, same complex, |xi-xj| < r, at
// most `max_nn` in index order) and edges per complex.
__global__ void k_ll_count(const float *__restrict__ x, const int *__restrict__ ptr, float r2, int max_nn,
                           int *__restrict__ deg, int *__restrict__ per_graph) {
    extern __shared__ float sx[];
    const int b = blockIdx.x;
    const int lo = ptr[b], n = ptr[b + 1] - lo;
    for (int i = threadIdx.x; i < n * 3; i += blockDim.x) sx[i] = x[(size_t)lo * 3 + i];
    __shared__ int s_tot;
    if (threadIdx.x == 0) s_tot = 0;
    __syncthreads();
    int local = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float xi = sx[3 * i], yi = sx[3 * i + 1], zi = sx[3 * i + 2];
        int c = 0;
        for (int j = 0; j < n; ++j) {
            const float dx = sx[3 * j] - xi, dy = sx[3 * j + 1] - yi, dz = sx[3 * j + 2] - zi;
            const float d2 = dx * dx + dy * dy + dz * dz;
            c += (j != i && d2 < r2) ? 1 : 0;
        }
        c = min(c, max_nn);
        deg[lo + i] = c;
        local += c;
    }
    atomicAdd(&s_tot, local);
    __syncthreads();
    if (threadIdx.x == 0) per_graph[b] = s_tot;
}

// Pass 2 (single workgroup): exclusive scan of the per-complex ll counts; totals.
// counts[0] = E_ll, counts[1] = E_kl (= kl_off[B], data independent).
__global__ void k_scan_graph_counts(const int *__restrict__ per_graph, int B, int *__restrict__ off,
                                    const int *__restrict__ kl_off, int *__restrict__ counts) {
    __shared__ int s_part[1024];
    __shared__ int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < B; base += blockDim.x) {
        const int i = base + threadIdx.x;
        const int v = i < B ? per_graph[i] : 0;
        s_part[threadIdx.x] = v;
        __syncthreads();
        for (int s = 1; s < blockDim.x; s <<= 1) {          // Hillis-Steele inclusive scan
            int t = threadIdx.x >= s ? s_part[threadIdx.x - s] : 0;
            __syncthreads();
            s_part[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < B) off[i] = s_carry + s_part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) s_carry += s_part[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        off[B] = s_carry;
        counts[0] = s_carry;
        counts[1] = kl_off ? kl_off[B] : 0;
    }
}

// Pass 3: fill.  Edge order: dst (centre) major, src ascending.
__global__ void k_ll_fill(const float *__restrict__ x, const int *__restrict__ ptr, float r2, int max_nn,
                          const int *__restrict__ deg, const int *__restrict__ off, int n_total, int cap,
                          int *__restrict__ src, int *__restrict__ dst, int *__restrict__ rowptr) {
    extern __shared__ float sx[];
    const int b = blockIdx.x;
    const int lo = ptr[b], n = ptr[b + 1] - lo;
    int *s_start = reinterpret_cast<int *>(sx + 3 * n);
    for (int i = threadIdx.x; i < n * 3; i += blockDim.x) sx[i] = x[(size_t)lo * 3 + i];
    __syncthreads();
    if (threadIdx.x == 0) {                                  // n <= a few hundred: serial prefix
        int run = off[b];
        for (int i = 0; i < n; ++i) {
            s_start[i] = run;
            rowptr[lo + i] = run;
            run += deg[lo + i];
        }
        if (lo + n == n_total) rowptr[n_total] = run;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float xi = sx[3 * i], yi = sx[3 * i + 1], zi = sx[3 * i + 2];
        int w = s_start[i], c = 0;
        for (int j = 0; j < n && c < max_nn; ++j) {
            const float dx = sx[3 * j] - xi, dy = sx[3 * j + 1] - yi, dz = sx[3 * j + 2] - zi;
            const float d2 = dx * dx + dy * dy + dz * dz;
            if (j != i && d2 < r2) {
                if (w < cap) {
                    src[w] = lo + j;
                    dst[w] = lo + i;
                }
                ++w;
                ++c;
            }
        }
    }
}

// ---- kl / lk: for every keypoint its k nearest ligand atoms -------------------------------
// lk (src lig -> dst kp) is emitted kp-major, nearest first (ties: lower index).
// kl (src kp -> dst lig) is the same pair list sorted by (lig, kp): ranks come from a
// per-ligand-atom bitmask over keypoints kept in LDS.
// Dynamic LDS: 3*max_lig floats + max_lig*(words+1) ints + (max_lig+1) ints.
__global__ void k_kl_build(const float *__restrict__ lig_x, const int *__restrict__ lig_ptr,
                           const float *__restrict__ kp_x, const int *__restrict__ kp_ptr,
                           const int *__restrict__ kl_off, int k, int words, int n_lig_total, int n_kp_total,
                           int *__restrict__ kl_src, int *__restrict__ kl_dst, int *__restrict__ kl_rowptr,
                           int *__restrict__ lk_src, int *__restrict__ lk_dst, int *__restrict__ lk_rowptr) {
    extern __shared__ float smem[];
    const int b = blockIdx.x;
    const int llo = lig_ptr[b], nl = lig_ptr[b + 1] - llo;
    const int klo = kp_ptr[b], nk = kp_ptr[b + 1] - klo;
    float *sx = smem;
    unsigned *mask = reinterpret_cast<unsigned *>(smem + 3 * nl);
    int *start = reinterpret_cast<int *>(mask + (size_t)nl * words);
    const int kk = min(k, nl);
    const int base = kl_off[b];

    for (int i = threadIdx.x; i < nl * 3; i += blockDim.x) sx[i] = lig_x[(size_t)llo * 3 + i];
    for (int i = threadIdx.x; i < nl * words; i += blockDim.x) mask[i] = 0u;
    __syncthreads();

    // phase 1: top-k per keypoint, lk lists, selection bitmasks
    for (int p = threadIdx.x; p < nk; p += blockDim.x) {
        const float px = kp_x[(size_t)(klo + p) * 3], py = kp_x[(size_t)(klo + p) * 3 + 1],
                    pz = kp_x[(size_t)(klo + p) * 3 + 2];
        float bd[KL_KMAX];
        int bi[KL_KMAX];
#pragma unroll
        for (int j = 0; j < KL_KMAX; ++j) {
            bd[j] = 3.0e38f;
            bi[j] = -1;
        }
        for (int l = 0; l < nl; ++l) {
            const float dx = sx[3 * l] - px, dy = sx[3 * l + 1] - py, dz = sx[3 * l + 2] - pz;
            float d = dx * dx + dy * dy + dz * dz;
            // a diverged chain (NaN / Inf coordinates) must not leave slots of the best-list empty: non-finite distances count
            // as "farthest" and fill free slots in index order, so every emitted index is a real atom of this complex
            d = d < 3.0e38f ? d : 3.0e38f;
            int id = l;
            // insertion into the sorted best-list (strict < keeps the lower index on ties)
#pragma unroll
            for (int j = 0; j < KL_KMAX; ++j) {
                if (j >= kk) break;                       // (uniform: kk is the same for the whole workgroup)
                if (d < bd[j] || bi[j] < 0) {
                    const float td = bd[j];
                    const int ti = bi[j];
                    bd[j] = d;
                    bi[j] = id;
                    d = td;
                    id = ti;
                }
            }
        }
        lk_rowptr[klo + p] = base + p * kk;
#pragma unroll
        for (int j = 0; j < KL_KMAX; ++j) {
            if (j < kk) {
                lk_src[base + p * kk + j] = llo + bi[j];
                lk_dst[base + p * kk + j] = klo + p;
                atomicOr(&mask[(size_t)bi[j] * words + (p >> 5)], 1u << (p & 31));
            }
        }
    }
    if (threadIdx.x == 0 && klo + nk == n_kp_total) lk_rowptr[n_kp_total] = base + nk * kk;
    __syncthreads();

    // phase 2: kl row pointers (ligand atoms of this complex)
    if (threadIdx.x == 0) {
        int run = base;
        for (int l = 0; l < nl; ++l) {
            start[l] = run;
            kl_rowptr[llo + l] = run;
            int c = 0;
            for (int w = 0; w < words; ++w) c += __popc(mask[(size_t)l * words + w]);
            run += c;
        }
        if (llo + nl == n_lig_total) kl_rowptr[n_lig_total] = run;
    }
    __syncthreads();

    // phase 3: kl fill, (lig, kp)-sorted: one thread per (ligand atom, mask word)
    for (int it = threadIdx.x; it < nl * words; it += blockDim.x) {
        const int l = it / words, w = it - l * words;
        int pos = start[l];
        for (int ww = 0; ww < w; ++ww) pos += __popc(mask[(size_t)l * words + ww]);
        unsigned m = mask[(size_t)l * words + w];
        while (m) {
            const int bit = __ffs(m) - 1;
            m &= m - 1;
            kl_src[pos] = klo + w * 32 + bit;
            kl_dst[pos] = llo + l;
            ++pos;
        }
    }
}

// Per-complex static kl offsets: off[b] = sum_{b'<b} n_kp[b'] * min(k, n_lig[b']); off[B] = total.
__global__ void k_kl_offsets(const int *__restrict__ lig_ptr, const int *__restrict__ kp_ptr, int B, int k,
                             int *__restrict__ off) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int run = 0;
        for (int b = 0; b < B; ++b) {
            off[b] = run;
            run += (kp_ptr[b + 1] - kp_ptr[b]) * min(k, lig_ptr[b + 1] - lig_ptr[b]);
        }
        off[B] = run;
    }
}

// ---- ll: kNN graph (ll_k > 0, torch_cluster.knn_graph) --------------------------------------------------
// Edge counts are data independent: every atom has min(k, n - 1) in-edges.
__global__ void k_ll_knn_counts(const int *__restrict__ ptr, int B, int k, int *__restrict__ per_graph) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        const int n = ptr[b + 1] - ptr[b];
        per_graph[b] = n * min(k, max(n - 1, 0));
    }
}

// centre-major, nearest first (ties: lower index), src = neighbour, dst = centre
__global__ void k_ll_knn_fill(const float *__restrict__ x, const int *__restrict__ ptr, int k, const int *__restrict__ off,
                              int n_total, int cap, int *__restrict__ src, int *__restrict__ dst, int *__restrict__ rowptr) {
    extern __shared__ float sx[];
    const int b = blockIdx.x;
    const int lo = ptr[b], n = ptr[b + 1] - lo;
    for (int i = threadIdx.x; i < n * 3; i += blockDim.x) sx[i] = x[(size_t)lo * 3 + i];
    __syncthreads();
    const int kk = min(k, max(n - 1, 0)), base = off[b];
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float xi = sx[3 * i], yi = sx[3 * i + 1], zi = sx[3 * i + 2];
        float bd[KL_KMAX];
        int bi[KL_KMAX];
#pragma unroll
        for (int j = 0; j < KL_KMAX; ++j) {
            bd[j] = 3.0e38f;
            bi[j] = -1;
        }
        for (int l = 0; l < n; ++l) {
            if (l == i) continue;
            const float dx = sx[3 * l] - xi, dy = sx[3 * l + 1] - yi, dz = sx[3 * l + 2] - zi;
            float d = dx * dx + dy * dy + dz * dz;
            d = d < 3.0e38f ? d : 3.0e38f;               // non-finite: farthest, fills free slots in index order (see k_kl_build)
            int id = l;
#pragma unroll
            for (int j = 0; j < KL_KMAX; ++j) {
                if (j < kk && (d < bd[j] || bi[j] < 0)) {
                    const float td = bd[j];
                    const int ti = bi[j];
                    bd[j] = d;
                    bi[j] = id;
                    d = td;
                    id = ti;
                }
            }
        }
        rowptr[lo + i] = base + i * kk;
#pragma unroll
        for (int j = 0; j < KL_KMAX; ++j) {
            const int w = base + i * kk + j;
            if (j < kk && w < cap) {
                src[w] = lo + bi[j];
                dst[w] = lo + i;
            }
        }
    }
    if (threadIdx.x == 0 && lo + n == n_total) rowptr[n_total] = base + n * kk;
}

// ---- kl / lk: radius graph (kl_k == 0, torch_cluster.radius(x = lig, y = kp, r, max 100)) ------------------
// For every keypoint all ligand atoms of its complex with |.| < r, in index order (at most max_nn).  Pass 1 counts per
// complex, pass 2 (after the scan) rebuilds the selection and emits lk kp-major and kl (lig, kp)-sorted through the same
// per-ligand-atom bitmask as k_kl_build.  Dynamic LDS as k_kl_build plus nk ints.
__global__ void k_klr_count(const float *__restrict__ lig_x, const int *__restrict__ lig_ptr, const float *__restrict__ kp_x,
                            const int *__restrict__ kp_ptr, float r2, int max_nn, int *__restrict__ per_graph) {
    extern __shared__ float sx[];
    __shared__ int s_tot;
    const int b = blockIdx.x;
    const int llo = lig_ptr[b], nl = lig_ptr[b + 1] - llo;
    const int klo = kp_ptr[b], nk = kp_ptr[b + 1] - klo;
    for (int i = threadIdx.x; i < nl * 3; i += blockDim.x) sx[i] = lig_x[(size_t)llo * 3 + i];
    if (threadIdx.x == 0) s_tot = 0;
    __syncthreads();
    int local = 0;
    for (int p = threadIdx.x; p < nk; p += blockDim.x) {
        const float px = kp_x[(size_t)(klo + p) * 3], py = kp_x[(size_t)(klo + p) * 3 + 1], pz = kp_x[(size_t)(klo + p) * 3 + 2];
        int c = 0;
        for (int l = 0; l < nl; ++l) {
            const float dx = sx[3 * l] - px, dy = sx[3 * l + 1] - py, dz = sx[3 * l + 2] - pz;
            c += (dx * dx + dy * dy + dz * dz < r2) ? 1 : 0;
        }
        local += min(c, max_nn);
    }
    atomicAdd(&s_tot, local);
    __syncthreads();
    if (threadIdx.x == 0) per_graph[b] = s_tot;
}

__global__ void k_klr_fill(const float *__restrict__ lig_x, const int *__restrict__ lig_ptr, const float *__restrict__ kp_x,
                           const int *__restrict__ kp_ptr, const int *__restrict__ kl_off, float r2, int max_nn, int words,
                           int n_lig_total, int n_kp_total, int *__restrict__ kl_src, int *__restrict__ kl_dst,
                           int *__restrict__ kl_rowptr, int *__restrict__ lk_src, int *__restrict__ lk_dst,
                           int *__restrict__ lk_rowptr) {
    extern __shared__ float smem[];
    const int b = blockIdx.x;
    const int llo = lig_ptr[b], nl = lig_ptr[b + 1] - llo;
    const int klo = kp_ptr[b], nk = kp_ptr[b + 1] - klo;
    float *sx = smem;
    unsigned *mask = reinterpret_cast<unsigned *>(smem + 3 * nl);
    int *start = reinterpret_cast<int *>(mask + (size_t)nl * words);
    int *kdeg = start + nl + 1;
    const int base = kl_off[b];

    for (int i = threadIdx.x; i < nl * 3; i += blockDim.x) sx[i] = lig_x[(size_t)llo * 3 + i];
    for (int i = threadIdx.x; i < nl * words; i += blockDim.x) mask[i] = 0u;
    __syncthreads();
    // phase 1: selections per keypoint (bitmasks) and their counts
    for (int p = threadIdx.x; p < nk; p += blockDim.x) {
        const float px = kp_x[(size_t)(klo + p) * 3], py = kp_x[(size_t)(klo + p) * 3 + 1], pz = kp_x[(size_t)(klo + p) * 3 + 2];
        int c = 0;
        for (int l = 0; l < nl && c < max_nn; ++l) {
            const float dx = sx[3 * l] - px, dy = sx[3 * l + 1] - py, dz = sx[3 * l + 2] - pz;
            if (dx * dx + dy * dy + dz * dz < r2) {
                atomicOr(&mask[(size_t)l * words + (p >> 5)], 1u << (p & 31));
                ++c;
            }
        }
        kdeg[p] = c;
    }
    __syncthreads();
    // phase 2: row pointers of both lists (serial prefixes: a complex has a few hundred nodes at most)
    if (threadIdx.x == 0) {
        int run = base;
        for (int p = 0; p < nk; ++p) {
            lk_rowptr[klo + p] = run;
            const int c = kdeg[p];
            kdeg[p] = run;                       // becomes the write position of keypoint p
            run += c;
        }
        if (klo + nk == n_kp_total) lk_rowptr[n_kp_total] = run;
        run = base;
        for (int l = 0; l < nl; ++l) {
            start[l] = run;
            kl_rowptr[llo + l] = run;
            int c = 0;
            for (int w = 0; w < words; ++w) c += __popc(mask[(size_t)l * words + w]);
            run += c;
        }
        if (llo + nl == n_lig_total) kl_rowptr[n_lig_total] = run;
    }
    __syncthreads();
    // phase 3a: lk fill (kp-major, ligand index ascending): bit p of every ligand atom's mask
    for (int p = threadIdx.x; p < nk; p += blockDim.x) {
        int pos = kdeg[p];
        for (int l = 0; l < nl; ++l)
            if ((mask[(size_t)l * words + (p >> 5)] >> (p & 31)) & 1u) {
                lk_src[pos] = llo + l;
                lk_dst[pos] = klo + p;
                ++pos;
            }
    }
    // phase 3b: kl fill, (lig, kp)-sorted: one thread per (ligand atom, mask word)
    for (int it = threadIdx.x; it < nl * words; it += blockDim.x) {
        const int l = it / words, w = it - l * words;
        int pos = start[l];
        for (int ww = 0; ww < w; ++ww) pos += __popc(mask[(size_t)l * words + ww]);
        unsigned m = mask[(size_t)l * words + w];
        while (m) {
            const int bit = __ffs(m) - 1;
            m &= m - 1;
            kl_src[pos] = klo + w * 32 + bit;
            kl_dst[pos] = llo + l;
            ++pos;
        }
    }
}

// same-residue flag of every rr edge (pdbbind_processing.py:248: res_idx[src] == res_idx[dst])
__global__ void k_same_res(const int *__restrict__ src, const int *__restrict__ dst, const int *__restrict__ n_edges, int cap,
                           const int *__restrict__ res_idx, unsigned char *__restrict__ same_res) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < min(*n_edges, cap)) same_res[e] = res_idx[src[e]] == res_idx[dst[e]];
}

}  // namespace kpd

using namespace kpd;

// Host-side launchers shared by the engines (declared in engine.h).
namespace kpd {

// radius graph of one point set (torch_cluster.radius_graph): dst-sorted COO + CSR, per-graph counts
kpd_status launch_radius_graph(const float *x, const int *ptr, int B, int n_total, int max_per_graph, float r, int max_nn,
                               int cap, int *src, int *dst, int *rowptr, int *per_graph, int *deg_tmp, int *off_tmp,
                               const int *kl_off_for_counts, int *counts, hipStream_t st) {
    KPD_REQUIRE(max_per_graph >= 1 && max_per_graph <= 2048, KPD_ERR_INVALID, "radius graph: %d nodes per graph (max 2048)", max_per_graph);
    const int threads = 256;
    const float r2 = r * r;
    size_t lds = (size_t)max_per_graph * 3 * sizeof(float);
    hipLaunchKernelGGL(k_ll_count, dim3(B), dim3(threads), lds, st, x, ptr, r2, max_nn, deg_tmp, per_graph);
    KPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_scan_graph_counts, dim3(1), dim3(1024), 0, st, per_graph, B, off_tmp, kl_off_for_counts, counts);
    KPD_LAUNCH_CHECK();
    lds = (size_t)max_per_graph * (3 * sizeof(float) + sizeof(int));
    hipLaunchKernelGGL(k_ll_fill, dim3(B), dim3(threads), lds, st, x, ptr, r2, max_nn, deg_tmp, off_tmp, n_total, cap, src, dst,
                       rowptr);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// for every y its k nearest x of the same graph (torch_cluster.knn): y-major list (src = x, dst = y) and the
// same pairs x-major (src = y, dst = x), both with CSR row pointers
kpd_status launch_knn_bipartite(const float *x, const int *x_ptr, int n_x, int max_x, const float *y, const int *y_ptr, int n_y,
                                int max_y, int B, int k, int *off_tmp, int *xm_src, int *xm_dst, int *xm_rowptr, int *ym_src,
                                int *ym_dst, int *ym_rowptr, hipStream_t st) {
    KPD_REQUIRE(k >= 1 && k <= KL_KMAX, KPD_ERR_INVALID, "knn k=%d outside 1..%d", k, KL_KMAX);
    hipLaunchKernelGGL(k_kl_offsets, dim3(1), dim3(64), 0, st, x_ptr, y_ptr, B, k, off_tmp);
    KPD_LAUNCH_CHECK();
    const int words = cdiv(max_y, 32);
    const size_t lds = (size_t)max_x * (3 * sizeof(float) + (size_t)words * sizeof(unsigned) + sizeof(int)) + 16;
    KPD_REQUIRE(lds <= 150 * 1024, KPD_ERR_INVALID, "knn needs %zu B of LDS (max_x=%d, max_y=%d)", lds, max_x, max_y);
    KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_kl_build), 150 * 1024));
    // one thread per keypoint in phase 1: 320 threads take the 300 keypoints of an all-atom pocket in ONE pass (256 needed a second pass
    // with 44 active threads: 43.6 -> see profiles/r04_kl_build.txt)
    const int threads = max_y > 256 ? (max_y > 320 ? 512 : 320) : 256;
    hipLaunchKernelGGL(k_kl_build, dim3(B), dim3(threads), lds, st, x, x_ptr, y, y_ptr, off_tmp, k, words, n_x, n_y, xm_src, xm_dst,
                       xm_rowptr, ym_src, ym_dst, ym_rowptr);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// kNN lig-lig graph (torch_cluster.knn_graph), same outputs as launch_radius_graph
kpd_status launch_knn_graph(const float *x, const int *ptr, int B, int n_total, int max_per_graph, int k, int cap, int *src, int *dst,
                            int *rowptr, int *per_graph, int *off_tmp, const int *kl_off_for_counts, int *counts, hipStream_t st) {
    KPD_REQUIRE(k >= 1 && k <= KL_KMAX, KPD_ERR_INVALID, "ll_k=%d outside 1..%d", k, KL_KMAX);
    hipLaunchKernelGGL(k_ll_knn_counts, dim3(cdiv(B, 256)), dim3(256), 0, st, ptr, B, k, per_graph);
    KPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_scan_graph_counts, dim3(1), dim3(1024), 0, st, per_graph, B, off_tmp, kl_off_for_counts, counts);
    KPD_LAUNCH_CHECK();
    const int threads = std::min(256, std::max(64, (max_per_graph + 63) / 64 * 64));
    hipLaunchKernelGGL(k_ll_knn_fill, dim3(B), dim3(threads), (size_t)max_per_graph * 3 * sizeof(float), st, x, ptr, k, off_tmp, n_total,
                       cap, src, dst, rowptr);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// radius keypoint->ligand graph (torch_cluster.radius(x = lig, y = kp)): kl lig-major and lk kp-major lists; kl_off[B] = E_kl
kpd_status launch_radius_bipartite(const float *x, const int *x_ptr, int n_x, int max_x, const float *y, const int *y_ptr, int n_y,
                                   int max_y, int B, float r, int max_nn, int *per_graph_tmp, int *scratch2, int *off_tmp, int *xm_src,
                                   int *xm_dst, int *xm_rowptr, int *ym_src, int *ym_dst, int *ym_rowptr, hipStream_t st) {
    KPD_REQUIRE(r > 0.0f, KPD_ERR_INVALID, "kl_k = 0 needs graph_cutoffs['kl'] > 0 (got %f)", r);
    const int words = cdiv(max_y, 32);
    const size_t lds = (size_t)max_x * (3 * sizeof(float) + (size_t)words * sizeof(unsigned) + sizeof(int)) + (size_t)(max_y + 2) * sizeof(int);
    KPD_REQUIRE(lds <= 150 * 1024, KPD_ERR_INVALID, "radius kl graph needs %zu B of LDS (max_x=%d, max_y=%d)", lds, max_x, max_y);
    KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_klr_fill), 150 * 1024));
    hipLaunchKernelGGL(k_klr_count, dim3(B), dim3(256), (size_t)max_x * 3 * sizeof(float), st, x, x_ptr, y, y_ptr, r * r, max_nn,
                       per_graph_tmp);
    KPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_scan_graph_counts, dim3(1), dim3(1024), 0, st, per_graph_tmp, B, off_tmp, static_cast<const int *>(nullptr),
                       scratch2);
    KPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_klr_fill, dim3(B), dim3(256), lds, st, x, x_ptr, y, y_ptr, off_tmp, r * r, max_nn, words, n_x, n_y, xm_src,
                       xm_dst, xm_rowptr, ym_src, ym_dst, ym_rowptr);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// ll: radius graph (ll_k == 0) or kNN graph; kl / lk: kNN (kl_k > 0) or radius graph (kl_k == 0)     dynamics.py:387-420
kpd_status launch_lig_graph(const kpd_batch *bt, float ll_cutoff, int ll_k, float kl_cutoff, int kl_k, const kpd_lig_graph *g,
                            int *ll_deg_tmp, int *ll_off_tmp, int *kl_off_tmp, int *kl_pg_tmp, hipStream_t st) {
    KPD_REQUIRE(kl_k >= 0 && kl_k <= KL_KMAX, KPD_ERR_INVALID, "kl_k=%d outside 0..%d", kl_k, KL_KMAX);
    KPD_REQUIRE(ll_k >= 0 && ll_k <= KL_KMAX, KPD_ERR_INVALID, "ll_k=%d outside 0..%d", ll_k, KL_KMAX);
    KPD_REQUIRE(bt->max_lig >= 1 && bt->max_lig <= 1024, KPD_ERR_INVALID, "max_lig=%d outside 1..1024", bt->max_lig);
    const long need_kl = (long)bt->n_kp * (kl_k > 0 ? kl_k : std::min(bt->max_lig, 100));
    KPD_REQUIRE(g->cap_kl >= need_kl, KPD_ERR_CAPACITY, "cap_kl=%d < %ld", g->cap_kl, need_kl);
    const long need_ll = (long)bt->n_lig * (ll_k > 0 ? std::min(bt->max_lig - 1, ll_k) : std::min(bt->max_lig - 1, 200));
    KPD_REQUIRE(g->cap_ll >= need_ll, KPD_ERR_CAPACITY, "cap_ll=%d < %ld", g->cap_ll, need_ll);
    // kl first: its offsets table also yields E_kl for counts[1]
    if (kl_k > 0)
        KPD_TRY(launch_knn_bipartite(bt->lig_x, bt->lig_ptr, bt->n_lig, bt->max_lig, bt->kp_x, bt->kp_ptr, bt->n_kp, bt->max_kp, bt->B,
                                     kl_k, kl_off_tmp, g->kl_src, g->kl_dst, g->kl_rowptr, g->lk_src, g->lk_dst, g->lk_rowptr, st));
    else
        KPD_TRY(launch_radius_bipartite(bt->lig_x, bt->lig_ptr, bt->n_lig, bt->max_lig, bt->kp_x, bt->kp_ptr, bt->n_kp, bt->max_kp, bt->B,
                                        kl_cutoff, 100, kl_pg_tmp, kl_pg_tmp + bt->B, kl_off_tmp, g->kl_src, g->kl_dst, g->kl_rowptr,
                                        g->lk_src, g->lk_dst, g->lk_rowptr, st));
    if (ll_k > 0)
        return launch_knn_graph(bt->lig_x, bt->lig_ptr, bt->B, bt->n_lig, bt->max_lig, ll_k, g->cap_ll, g->ll_src, g->ll_dst, g->ll_rowptr,
                                g->ll_per_graph, ll_off_tmp, kl_off_tmp, g->counts, st);
    return launch_radius_graph(bt->lig_x, bt->lig_ptr, bt->B, bt->n_lig, bt->max_lig, ll_cutoff, 200, g->cap_ll, g->ll_src,
                               g->ll_dst, g->ll_rowptr, g->ll_per_graph, ll_deg_tmp, ll_off_tmp, kl_off_tmp, g->counts, st);
}
}  // namespace kpd

extern "C" int64_t kpd_rec_graph_scratch_bytes(int32_t n_rec, int32_t B) {
    if (n_rec < 0 || B < 0) return -1;
    return (int64_t)((size_t)n_rec + (size_t)B + 4) * (int64_t)sizeof(int);
}

extern "C" kpd_status kpd_build_rec_graph(const float *rec_x, const int32_t *rec_ptr, int32_t B, int32_t n_rec, int32_t max_rec,
                                          float r, int32_t max_nn, const int32_t *res_idx, int32_t cap, int32_t *src, int32_t *dst,
                                          int32_t *rowptr, int32_t *per_graph, uint8_t *same_res, int32_t *counts, void *scratch,
                                          void *stream) {
    KPD_REQUIRE(rec_x && rec_ptr && src && dst && rowptr && per_graph && counts && scratch, KPD_ERR_INVALID, "null argument");
    KPD_REQUIRE(B >= 1 && n_rec >= 1 && max_nn >= 1 && cap >= 0 && r > 0.0f, KPD_ERR_INVALID, "B=%d n_rec=%d max_nn=%d cap=%d r=%g", B,
                n_rec, max_nn, cap, (double)r);
    KPD_REQUIRE((res_idx == nullptr) == (same_res == nullptr), KPD_ERR_INVALID, "res_idx and same_res go together");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int *deg_tmp = static_cast<int *>(scratch), *off_tmp = deg_tmp + n_rec;
    KPD_TRY(launch_radius_graph(rec_x, rec_ptr, B, n_rec, max_rec, r, max_nn, cap, src, dst, rowptr, per_graph, deg_tmp, off_tmp,
                                nullptr, counts, st));
    if (same_res && cap) {
        hipLaunchKernelGGL(k_same_res, dim3(cdiv(cap, 256)), dim3(256), 0, st, src, dst, counts, cap, res_idx, same_res);
        KPD_LAUNCH_CHECK();
    }
    return KPD_OK;
}
