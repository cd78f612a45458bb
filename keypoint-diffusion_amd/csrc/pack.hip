// Library-wide utilities: error text, device arena, weight repacking kernels.
#include <stdarg.h>

#include <map>
#include <mutex>
#include <utility>

#include "engine.h"

namespace kpd {

static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

kpd_status ensure_dynamic_lds(const void *kernel, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void *, int>, int> done;          // (kernel, device) -> bytes granted
    int dev = 0;
    KPD_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    int &have = done[std::make_pair(kernel, dev)];
    if (have >= bytes) return KPD_OK;
    KPD_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    have = bytes;
    return KPD_OK;
}

int cu_count() {
    static std::mutex mu;
    static std::map<int, int> cus;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    std::lock_guard<std::mutex> lock(mu);
    int &n = cus[dev];
    if (!n) {
        hipDeviceProp_t prop;
        n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return n;
}

int poison_level() {
    static const int level = [] {
        const char *e = getenv("KPD_POISON");
        return e ? atoi(e) : 0;
    }();
    return level;
}

bool poison_selected() {
    static const int only = tool_env_int("KPD_POISON_ONLY", -1);      // (TOOLS build: bisecting a NaN to its buffer)
    static int counter = 0;
    const int me = counter++;
    return only < 0 || only == me;
}

void poison_floats(void *p, size_t bytes) {
    (void)hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(p), 0x7fc0dead, bytes / 4);      // a quiet NaN with a recognisable payload
}

void zero_pad_columns(float *p, size_t rows, int stride, int valid) {
    (void)hipMemset2D(p + valid, (size_t)stride * 4, 0, (size_t)(stride - valid) * 4, rows);
}

// One workgroup per 80 KB of LDS, two per CU, several rounds of them: every byte of every CU's LDS ends up holding NaNs.
__global__ __launch_bounds__(256) void k_poison_lds(int words, unsigned *sink) {
    extern __shared__ unsigned lds_words[];
    for (int i = threadIdx.x; i < words; i += 256) lds_words[i] = 0x7fc0dead;
    __syncthreads();
    // keep the workgroup resident long enough for its siblings to land on the other CUs (and keep the stores alive)
    unsigned acc = 0;
    for (int r = 0; r < 64; ++r)
        for (int i = threadIdx.x; i < words; i += 256 * 64) acc += lds_words[i] >> 31;
    if (acc == 0xffffffffu) sink[0] = acc;
}

kpd_status poison_lds(hipStream_t st) {
    if (tool_env_int("KPD_POISON_ONLY", -1) >= 0) return KPD_OK;
    static unsigned *sink = nullptr;
    if (!sink) KPD_HIP(hipMalloc(reinterpret_cast<void **>(&sink), 256));
    const int bytes = 80 * 1024;
    KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_poison_lds), bytes));
    k_poison_lds<<<8 * cu_count(), 256, bytes, st>>>(bytes / 4, sink);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status Arena::reserve(size_t bytes) {
    if (bytes <= cap) {
        used = 0;
        return KPD_OK;
    }
    release();
    KPD_HIP(hipMalloc(reinterpret_cast<void **>(&base), bytes));
    KPD_HIP(hipMemset(base, 0, bytes));
    cap = bytes;
    used = 0;
    return KPD_OK;
}

void Arena::release() {
    if (base) (void)hipFree(base);
    base = nullptr;
    cap = used = 0;
}

__global__ void k_pack_gemm_weight(const float *__restrict__ src, int n_out, int ld, int col0, int K,
                                   float *__restrict__ wp, float *__restrict__ wx, int wp_floats) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < wp_floats) {
        const int j = idx & 3, nt = (idx >> 2) & 1, lane = (idx >> 3) & 63, wave = (idx >> 9) & 3, g = idx >> 11;
        const int k = 8 * g + 4 * (lane >> 5) + j;
        const int n = 64 * wave + 32 * nt + (lane & 31);
        wp[idx] = (k < K && n < n_out) ? src[(size_t)n * ld + col0 + k] : 0.0f;
    }
    if (wx && idx < KP) wx[idx] = (idx < K && n_out > 256) ? src[(size_t)256 * ld + col0 + idx] : 0.0f;
}

// Transposed source: element (k, n) of the logical weight sits at src[k * ld + n] (GVP Wh / Wu are
// stored [in, out]); otherwise as k_pack_gemm_weight.
__global__ void k_pack_gemm_weight_kn(const float *__restrict__ src, int n_out, int ld, int K, float *__restrict__ wp,
                                      int wp_floats) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < wp_floats) {
        const int j = idx & 3, nt = (idx >> 2) & 1, lane = (idx >> 3) & 63, wave = (idx >> 9) & 3, g = idx >> 11;
        const int k = 8 * g + 4 * (lane >> 5) + j;
        const int n = 64 * wave + 32 * nt + (lane & 31);
        wp[idx] = (k < K && n < n_out) ? src[(size_t)k * ld + n] : 0.0f;
    }
}

kpd_status pack_gemm_weight(const float *src, int n_out, int ld, int col0, int K, float *wp, float *wx,
                            hipStream_t st) {
    KPD_REQUIRE(K <= KP && n_out <= HW, KPD_ERR_WEIGHTS, "pack: K=%d n_out=%d exceed %d/%d", K, n_out, KP, HW);
    hipLaunchKernelGGL(k_pack_gemm_weight, dim3(cdiv(WP_FLOATS, 256)), dim3(256), 0, st, src, n_out, ld, col0, K, wp, wx,
                       WP_FLOATS);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// f16x2 mode (mfma_core.h, gemm_rows64_h): re-pack a finished fp32 block (all scalings and the bias row already in it) into two
// f16 planes, w ~ hi + lo, in the B-fragment order of v_mfma_f32_32x32x16_f16:
//   Wh[(((s * 4 + wave) * 2 + nt) * 2 + plane) * 64 + lane][i] = plane(W[n = 64 wave + 32 nt + (lane & 31)][k = 16 s + 8 (lane >> 5) + i])
// for s < KH_STEPS k-steps of 16 (K = 264 padded to 272 with zeros), i < 8: every fragment load of a wave (one dwordx4 per lane)
// covers 1 KB of contiguous memory.  One thread per (s, wave, lane, nt, i).
constexpr float H_SCALE_W_PACK = 1024.0f;       // = H_SCALE_W of mfma_core.h: keeps the lo plane out of the f16 subnormal range

// Range guard of the f16x2 planes: a scaled weight at or beyond the largest finite f16 (65504, i.e. |w| >= ~64 after every
// scaling the finished block carries) would become inf in the hi plane and NaN in the lo plane.  Every f16 packing kernel
// raises this flag instead of letting that through; the commit that ran the packing reads it (f16_pack_begin / f16_pack_end)
// and refuses the f16x2 mode for that engine with a clear error, the exact fp32 mode is unaffected.
__device__ unsigned g_f16_overflow;
static std::mutex g_f16_mu;

__device__ __forceinline__ void f16_range_check(float scaled) {
    if (!(fabsf(scaled) < 65504.0f)) atomicOr(&g_f16_overflow, 1u);      // also catches NaN / inf weights
}

void f16_pack_begin() {
    g_f16_mu.lock();
    const unsigned zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_f16_overflow), &zero, sizeof(zero));
}

bool f16_pack_end() {
    unsigned v = 1;
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_f16_overflow), sizeof(v));
    g_f16_mu.unlock();
    return v != 0;
}

__global__ void k_f16_range_check(const float *__restrict__ p, int n, float scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) f16_range_check(scale * p[i]);
}

kpd_status f16_range_check_array(const float *p, int n, hipStream_t st) {
    hipLaunchKernelGGL(k_f16_range_check, dim3(cdiv(n, 256)), dim3(256), 0, st, p, n, H_SCALE_W_PACK);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}
__global__ void k_pack_f16_split(const float *__restrict__ wp, __fp16 *__restrict__ wh, int total) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int i = idx & 7, nt = (idx >> 3) & 1, lane = (idx >> 4) & 63, wave = (idx >> 10) & 3, s = idx >> 12;
    const int k = 16 * s + 8 * (lane >> 5) + i, n31 = lane & 31;
    float w = 0.0f;
    if (k < KP) {
        const int g = k >> 3, h = (k >> 2) & 1, j = k & 3;
        w = H_SCALE_W_PACK * wp[((size_t)(g * 4 + wave) * 64 + (n31 + 32 * h)) * 8 + nt * 4 + j];
    }
    f16_range_check(w);
    const __fp16 hi = (__fp16)w;                       // round to nearest: |w - hi| <= 2^-11 |w|
    const __fp16 lo = (__fp16)(w - (float)hi);         // and the remainder again: |w - hi - lo| <= 2^-22 |w| (f16 normal range)
    const size_t base = ((((size_t)(s * 4 + wave) * 2 + nt) * 2) * 64 + lane) * 8;
    wh[base + i] = hi;
    wh[base + 64 * 8 + i] = lo;
}

kpd_status pack_f16_split(const float *wp, void *wh, hipStream_t st) {
    const int total = KH_STEPS * 4 * 64 * 2 * 8;
    hipLaunchKernelGGL(k_pack_f16_split, dim3(cdiv(total, 256)), dim3(256), 0, st, wp, static_cast<__fp16 *>(wh), total);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// Projections in f16x2 mode (k_proj_ws_h, egnn_chain.hip): W[n][k] of a 256 x 256 block, read from its chained fp32 form
// (slab kc = k >> 4: chain[kc * 4096 + ((n >> 4) * 64 + 16 * ((k & 15) >> 2) + (n & 15)) * 4 + (k & 3)]), as two f16 planes in the
// A-fragment order of v_mfma_f32_16x16x32_f16:
//   chh[(((kb * 16 + mt) * 2 + plane) * 64 + lane) * 8 + j] = plane(W[n = 16 mt + (lane & 15)][k = 32 kb + 8 (lane >> 4) + j])
__global__ void k_pack_proj_f16_split(const float *__restrict__ chain, __fp16 *__restrict__ chh) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // (kb, mt, lane, j)
    if (idx >= 8 * 16 * 64 * 8) return;
    const int j = idx & 7, lane = (idx >> 3) & 63, mt = (idx >> 9) & 15, kb = idx >> 13;
    const int k = 32 * kb + 8 * (lane >> 4) + j;
    const float w = H_SCALE_W_PACK * chain[(size_t)(k >> 4) * 4096 + (mt * 64 + 16 * ((k & 15) >> 2) + (lane & 15)) * 4 + (k & 3)];
    f16_range_check(w);
    const __fp16 hi = (__fp16)w;
    const __fp16 lo = (__fp16)(w - (float)hi);
    const size_t base = (((size_t)(kb * 16 + mt) * 2) * 64 + lane) * 8 + j;
    chh[base] = hi;
    chh[base + 64 * 8] = lo;
}

kpd_status pack_proj_f16_split(const float *chain, void *chh, hipStream_t st) {
    hipLaunchKernelGGL(k_pack_proj_f16_split, dim3(8 * 16 * 64 * 8 / 256), dim3(256), 0, st, chain, static_cast<__fp16 *>(chh));
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

__global__ void k_pack_chain_frag(const float *__restrict__ src, int sn, int sk, int n_valid, int k_base, int k_valid,
                                  int n_tiles, float *__restrict__ dst) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_tiles * 256) return;
    const int r = idx & 3, lane = (idx >> 2) & 63, mt = idx >> 8;
    const int n = 16 * mt + (lane & 15), i = 4 * (lane >> 4) + r;
    dst[idx] = (n < n_valid && i < k_valid) ? src[(size_t)n * sn + (size_t)(k_base + i) * sk] : 0.0f;
}

kpd_status pack_chain_frag(const float *src, int sn, int sk, int n_valid, int k_base, int k_valid, int n_tiles, float *dst,
                           hipStream_t st) {
    KPD_REQUIRE(n_tiles >= 1 && k_valid >= 0 && k_valid <= 16, KPD_ERR_WEIGHTS, "pack_chain_frag: bad tile (%d, %d)", n_tiles, k_valid);
    hipLaunchKernelGGL(k_pack_chain_frag, dim3(n_tiles), dim3(256), 0, st, src, sn, sk, n_valid, k_base, k_valid, n_tiles, dst);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// f16x2 mode of the GVP chains (k_gvp_chain<16, 1>, k_gvp_node_chain<16, 1>): the chunk buffer of a GVP with 256 scalar outputs,
// re-packed unit by unit (a unit = one 16-KB chunk of the LDS ring, 4096 floats).  Input (pack_chain_frag): chunk c = a 16-row
// k-slab of to_feats_out as 16 x 256 floats [(mt * 64 + lane) * 4 + r] = W[16 mt + (lane & 15)][k0(c) + 4 (lane >> 4) + r], or
// the gate slab [nt][lane][r] = Wg[lane & 15][16 nt + 4 (lane >> 4) + r].  Three unit kinds (all planes carry 2^10, hi / lo
// halves of a weight):
//   kind 0, a PAIR of x slabs (2 kb, 2 kb + 1) -> units 2 kb and 2 kb + 1: unit u holds output tiles 8 (u & 1) .. + 7 of the
//           32-wide k-block kb in the A-fragment order of v_mfma_f32_16x16x32_f16,
//           unit[((m * 2 + plane) * 64 + lane) * 8 + j], k = 32 kb + 4 q + j (j < 4) | 32 kb + 16 + 4 q + (j - 4): the eight slots
//           a lane fills from result tiles 2 kb and 2 kb + 1 of the previous product;
//   kind 1, ONE 16-row slab (rbf, sh) -> one unit in the A-fragment order of v_mfma_f32_16x16x16_f16,
//           unit[((mt * 2 + plane) * 64 + lane) * 4 + j] = W[16 mt + (lane & 15)][k0 + 4 q + j];
//   kind 2, the gate slab -> one unit, [kb][plane][lane][8] with the k-slots of kind 0.
__global__ void k_pack_gvp_unit_h(const float *__restrict__ chain, float *__restrict__ chain_h, int kind, int unit, int src_chunk) {
    const int rem = blockIdx.x * blockDim.x + threadIdx.x;       // < 4096
    __fp16 *dst = reinterpret_cast<__fp16 *>(chain_h + (size_t)unit * 4096);
    float w;
    int at;
    if (kind == 0) {
        const int j = rem & 7, lane = (rem >> 3) & 63, m = rem >> 9;
        const int mt = 8 * (unit & 1) + m;
        w = chain[(size_t)(src_chunk + (j >> 2)) * 4096 + (mt * 64 + lane) * 4 + (j & 3)];
        at = ((m * 2) * 64 + lane) * 8 + j;
        f16_range_check(H_SCALE_W_PACK * w);
        const __fp16 hi = (__fp16)(H_SCALE_W_PACK * w);
        dst[at] = hi;
        dst[at + 64 * 8] = (__fp16)(H_SCALE_W_PACK * w - (float)hi);
    } else if (kind == 1) {
        const int j = rem & 3, lane = (rem >> 2) & 63, mt = rem >> 8;
        w = chain[(size_t)src_chunk * 4096 + (mt * 64 + lane) * 4 + j];
        at = ((mt * 2) * 64 + lane) * 4 + j;
        f16_range_check(H_SCALE_W_PACK * w);
        const __fp16 hi = (__fp16)(H_SCALE_W_PACK * w);
        dst[at] = hi;
        dst[at + 64 * 4] = (__fp16)(H_SCALE_W_PACK * w - (float)hi);
    } else {
        if (rem >= 8 * 64 * 8) return;
        const int j = rem & 7, lane = (rem >> 3) & 63, kb = rem >> 9;
        w = chain[(size_t)src_chunk * 4096 + ((2 * kb + (j >> 2)) * 64 + lane) * 4 + (j & 3)];
        at = ((kb * 2) * 64 + lane) * 8 + j;
        f16_range_check(H_SCALE_W_PACK * w);
        const __fp16 hi = (__fp16)(H_SCALE_W_PACK * w);
        dst[at] = hi;
        dst[at + 64 * 8] = (__fp16)(H_SCALE_W_PACK * w - (float)hi);
    }
}

// chain_pos == 0 (head of a message chain): chunks [rbf | n_ht sh slabs | gate]; otherwise [16 x slabs | sh | gate]
// the 256 x 256 node block of a split first Linear (k_gvp_proj_chain): 16 x slabs -> 16 units of kind 0
kpd_status pack_gvp_proj_h(const float *wproj, float *wproj_h, hipStream_t st) {
    for (int u = 0; u < 16; ++u) hipLaunchKernelGGL(k_pack_gvp_unit_h, dim3(16), dim3(256), 0, st, wproj, wproj_h, 0, u, 2 * (u >> 1));
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status pack_gvp_chain_h(const float *chain, float *chain_h, int head, int n_ht, hipStream_t st) {
    auto unit = [&](int kind, int u, int src) { hipLaunchKernelGGL(k_pack_gvp_unit_h, dim3(16), dim3(256), 0, st, chain, chain_h, kind, u, src); };
    if (head) {
        for (int c = 0; c < 1 + n_ht; ++c) unit(1, c, c);
        unit(2, 1 + n_ht, 1 + n_ht);
    } else {
        for (int u = 0; u < 16; ++u) unit(0, u, 2 * (u >> 1));
        unit(1, 16, 16);
        unit(2, 17, 17);
    }
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

__global__ void k_scale_inplace(float *__restrict__ p, int n, float f) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] *= f;
}

kpd_status scale_inplace(float *p, int n, float f, hipStream_t st) {
    hipLaunchKernelGGL(k_scale_inplace, dim3(cdiv(n, 256)), dim3(256), 0, st, p, n, f);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// Writes the bias as weight row k (the A tile carries a constant 1 in column k): wp[..k..][n] = f * bias[n] for the
// 256 MFMA columns, wx[k] = f * bias[256] for the extra column.
__global__ void k_patch_bias_row(float *__restrict__ wp, float *__restrict__ wx, const float *__restrict__ bias, float f, int k) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < 256) {
        const int g = k >> 3, h = (k >> 2) & 1, j = k & 3;
        const int wave = n >> 6, nt = (n >> 5) & 1, lane = (n & 31) + 32 * h;
        wp[((size_t)(g * 4 + wave) * 64 + lane) * 8 + nt * 4 + j] = f * bias[n];
    }
    if (n == 256) wx[k] = f * bias[256];
}

kpd_status patch_bias_row(float *wp, float *wx, const float *bias, float f, int k, hipStream_t st) {
    hipLaunchKernelGGL(k_patch_bias_row, dim3(2), dim3(256), 0, st, wp, wx, bias, f, k);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

__global__ void k_copy_pad(const float *__restrict__ src, int n_src, float *__restrict__ dst, int n_dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_dst) dst[i] = i < n_src ? src[i] : 0.0f;
}

kpd_status copy_pad(const float *src, int n_src, float *dst, int n_dst, hipStream_t st) {
    hipLaunchKernelGGL(k_copy_pad, dim3(cdiv(n_dst, 256)), dim3(256), 0, st, src, n_src, dst, n_dst);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

__global__ void k_copy_col_pad(const float *__restrict__ src, int n, int ld, int col, float *__restrict__ dst,
                               int n_dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_dst) dst[i] = i < n ? src[(size_t)i * ld + col] : 0.0f;
}

kpd_status copy_col_pad(const float *src, int n, int ld, int col, float *dst, int n_dst, hipStream_t st) {
    hipLaunchKernelGGL(k_copy_col_pad, dim3(cdiv(n_dst, 256)), dim3(256), 0, st, src, n, ld, col, dst, n_dst);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

__global__ void k_transpose2d(const float *__restrict__ src, int rows, int cols, float *__restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * cols) {
        const int r = i / cols, c = i - r * cols;
        dst[(size_t)c * rows + r] = src[i];
    }
}

kpd_status transpose2d(const float *src, int rows, int cols, float *dst, hipStream_t st) {
    hipLaunchKernelGGL(k_transpose2d, dim3(cdiv(rows * cols, 256)), dim3(256), 0, st, src, rows, cols, dst);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

}  // namespace kpd

extern "C" const char *kpd_last_error(void) { return kpd::g_err; }
extern "C" int kpd_version(void) { return 100; }
extern "C" int kpd_build_flags(void) {
#ifdef KPD_TOOLS
    return 1;
#else
    return 0;
#endif
}
