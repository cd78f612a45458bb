// Internal declarations shared between the translation units of libkpd_hip.so.
#pragma once
#include <type_traits>

#include "common.h"

namespace kpd {

// kl_pg_tmp: [B + 2] scratch (per-complex kl counts of the radius variant)
kpd_status launch_lig_graph(const kpd_batch *bt, float ll_cutoff, int ll_k, float kl_cutoff, int kl_k, const kpd_lig_graph *g,
                            int *ll_deg_tmp, int *ll_off_tmp, int *kl_off_tmp, int *kl_pg_tmp, hipStream_t st);

kpd_status launch_radius_graph(const float *x, const int *ptr, int B, int n_total, int max_per_graph, float r, int max_nn,
                               int cap, int *src, int *dst, int *rowptr, int *per_graph, int *deg_tmp, int *off_tmp,
                               const int *kl_off_for_counts, int *counts, hipStream_t st);
kpd_status launch_knn_bipartite(const float *x, const int *x_ptr, int n_x, int max_x, const float *y, const int *y_ptr, int n_y,
                                int max_y, int B, int k, int *off_tmp, int *xm_src, int *xm_dst, int *xm_rowptr, int *ym_src,
                                int *ym_dst, int *ym_rowptr, hipStream_t st);

// for every y all x of its graph within r, at most max_nn in index order (torch_cluster.radius): lists as launch_knn_bipartite;
// per_graph_tmp [B], scratch2 [2], off_tmp [B + 1] (off_tmp[B] = total)
kpd_status launch_radius_bipartite(const float *x, const int *x_ptr, int n_x, int max_x, const float *y, const int *y_ptr, int n_y,
                                   int max_y, int B, float r, int max_nn, int *per_graph_tmp, int *scratch2, int *off_tmp, int *xm_src,
                                   int *xm_dst, int *xm_rowptr, int *ym_src, int *ym_dst, int *ym_rowptr, hipStream_t st);

// Debug poisoning (environment KPD_POISON, read once; 0 = off, the production setting):
//   >= 1  every float buffer carved out of a WORKSPACE arena is filled with NaNs instead of the arena's zeros, and the LDS of every
//         CU is overwritten with NaNs before each dominant-kernel launch (poison_lds): a read of memory the current forward has not
//         written -- stale arena contents, pads assumed zero, LDS left by an earlier workgroup -- surfaces as a NaN in the output
//         instead of as a value that happens to equal the previous forward's;
//   >= 2  the packed-weight arenas are poisoned too (shows which pads of the packed blocks rely on the arena's zero fill).
// Integer buffers keep the zero fill (a poisoned index would fault).  tests/test_poison_gpu.py runs the denoisers this way.
int poison_level();
bool poison_selected();      // KPD_POISON_ONLY=<i>: poison only the i-th float buffer carved in this process (bisecting a NaN to its buffer)
void poison_floats(void *p, size_t bytes);
void zero_pad_columns(float *p, size_t rows, int stride, int valid);
kpd_status poison_lds(hipStream_t st);

// Grow-only device arena: one hipMalloc, carved with 256-B alignment, zero-filled.
struct Arena {
    char *base = nullptr;
    size_t cap = 0, used = 0;
    int poison_at = 1;          // KPD_POISON level from which this arena's float buffers are NaN-filled (weights: 2)
    kpd_status reserve(size_t bytes);
    void release();
    void reset() { used = 0; }
    template <typename T>
    T *take(size_t count) {
        size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
        if (used + bytes > cap) return nullptr;
        T *p = reinterpret_cast<T *>(base + used);
        used += bytes;
        if (std::is_floating_point<T>::value && poison_level() >= poison_at && poison_selected()) poison_floats(p, bytes);
        return p;
    }
    // rows x stride floats of which only the first `valid` columns of a row are ever written: the K-padding columns are part of
    // the layout's contract (they must read as zero: they meet zero weight rows in a GEMM) and keep the zero fill under poisoning
    float *take_rows(size_t rows, int stride, int valid) {
        float *p = take<float>(rows * stride);
        if (p && poison_level() >= poison_at) zero_pad_columns(p, rows, stride, valid);
        return p;
    }
};

// Weight repacking helpers (pack.hip).
// src: torch Linear weight [n_out, ld] row-major on device; uses columns [col0, col0 + K).
kpd_status pack_gemm_weight(const float *src, int n_out, int ld, int col0, int K, float *wp, float *wx, hipStream_t st);
kpd_status scale_inplace(float *p, int n, float f, hipStream_t st);
// f16x2 mode: the finished packed fp32 block wp as f16 hi / lo planes in 32x32x16 B-fragment order (WH_HALVES halves)
kpd_status pack_f16_split(const float *wp, void *wh, hipStream_t st);
// f16x2 mode of the projections: the finished chained block (16 k-slabs of pack_chain_frag, scalings in it) as f16 hi / lo planes in
// the A-fragment order of v_mfma_f32_16x16x32_f16 (CHH_HALVES halves)
kpd_status pack_proj_f16_split(const float *chain, void *chh, hipStream_t st);
// f16x2 mode of the GVP chains: the chunk buffer of a GVP with 256 scalar outputs re-packed unit by unit (pack.hip, k_pack_gvp_unit_h)
// Range guard of the f16 planes (pack.hip): bracket the f16 packing of one commit; f16_pack_end() == true means a scaled weight
// fell outside the finite f16 range and the f16x2 mode must not be used with these weights.
void f16_pack_begin();
bool f16_pack_end();
kpd_status f16_range_check_array(const float *p, int n, hipStream_t st);      // p[i] * 2^10 within the finite f16 range?
struct F16PackScope {           // RAII around the bracket (an early error return must not keep the guard's mutex)
    bool open = true;
    F16PackScope() { f16_pack_begin(); }
    bool overflowed() { open = false; return f16_pack_end(); }
    ~F16PackScope() { if (open) (void)f16_pack_end(); }
};
constexpr const char *F16_RANGE_ERROR =
    "these weights do not fit the f16x2 mode: a packed weight reaches |w| * 2^10 >= 65504 (|w| >= ~64 with the block's scalings) "
    "or is not finite, so its f16 planes would hold inf / NaN; use the exact mode (gemm=f32, the default)";
kpd_status pack_gvp_chain_h(const float *chain, float *chain_h, int head, int n_ht, hipStream_t st);
kpd_status pack_gvp_proj_h(const float *wproj, float *wproj_h, hipStream_t st);
kpd_status patch_bias_row(float *wp, float *wx, const float *bias, float f, int k, hipStream_t st);
// 16x16x4 MFMA A-operand fragments of a [n][k] matrix with element (n, k) at src[n * sn + k * sk]:
// dst[(mt * 64 + lane) * 4 + r] = element(16 mt + (lane & 15), k_base + 4 (lane >> 4) + r), zero outside
// n < n_valid / 4 (lane >> 4) + r < k_valid.  One call packs the 16 k-rows [k_base, k_base + 16) for n_tiles
// 16-wide output tiles (the chained GVP edge kernel, gvp_chain.hip).
kpd_status pack_chain_frag(const float *src, int sn, int sk, int n_valid, int k_base, int k_valid, int n_tiles, float *dst,
                           hipStream_t st);
// dst[0..n_dst) = src[0..n_src) then zeros.
kpd_status copy_pad(const float *src, int n_src, float *dst, int n_dst, hipStream_t st);
// dst[c][r] = src[r][c]
kpd_status transpose2d(const float *src, int rows, int cols, float *dst, hipStream_t st);
// strided column gather: dst[i] = src[i * ld + col] for i < n, zeros up to n_dst
kpd_status copy_col_pad(const float *src, int n, int ld, int col, float *dst, int n_dst, hipStream_t st);

// Weight-stationary tall-skinny GEMM with fused epilogues for the training engines (ws_gemm.hip).
enum { WS_BIAS_SILU = 0, WS_SILU_BWD = 1, WS_PLAIN = 2 };
int ws_gemm_pack_floats();
// rowdot (optional; WS_SILU_BWD: of Y, WS_BIAS_SILU: of the activated output A): rowdot_out[hf * rows + r] = sum over the columns of half hf
// of the row times rowdot_w[c * rowdot_stride] -- the product of every finished row with one more vector (a head of the MLP, the
// distance column in backward), taken from the registers the epilogue holds (two halves: the caller adds them)
// ext (256-wide form only): up to 17 extra inputs X2 (WS_BIAS_SILU) or extra outputs Y2 (WS_PLAIN) -- the vector-norm block of a GVP's
// to_feats_out; W addresses element (n = output, k = input) of the narrow block as W[n * sn + k * sk]; ld: row stride of X2 / Y2
struct WsgExtra {
    const float *X2 = nullptr;
    float *Y2 = nullptr;
    const float *W = nullptr;
    int sn = 0, sk = 0, n = 0, ld = 0;
    // with X2: G2[row][0..15] / G2b[row][0..15] = the shares of the two column halves of SiLU(Y) Wg^T (Wg [ng <= 16][256], row stride ldg);
    // the consumer adds them
    const float *Wg = nullptr;
    float *G2 = nullptr, *G2b = nullptr;
    int ldg = 0, ng = 0;
};
kpd_status ws_gemm(int mode, const float *X, int rows, int ldx, const float *W, int ldw, bool transpose_w, const float *bias,
                   const float *P, float *Y, float *A, int ldy, float *pack_scratch, hipStream_t st, bool has257 = true,
                   bool accumulate = false, const float *rowdot_w = nullptr, int rowdot_stride = 1, float *rowdot_out = nullptr,
                   const WsgExtra *ext = nullptr);

}  // namespace kpd
