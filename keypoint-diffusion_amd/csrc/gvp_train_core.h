// Generic pieces of the GVP training engines (gvp_train.hip: the denoiser; recenc_train.hip: the keypoint receptor encoder):
// the elementwise / reduction kernels of GVP.forward and GVPLayerNorm and their backward passes, edge geometry forward and
// backward, dropout streams, and GVP / GVPLayerNorm as host routines over sgemm.hip products -- templates over the engine type,
// which supplies the stream (TrainCtx), the scalar width S and the scratch buffers dgate, dsh, dVh, wsg_pack,
// tmp_s, tmp_v, U.  Vector features are [rows, 3, channels].  File-local in every translation unit that includes it.
#pragma once
#include "chain_core.h"
#include "egnn_kernels.h"
#include "engine.h"
#include "train_ops.h"

namespace kpd {
namespace {

constexpr int VC = 16;        // vector channels
constexpr int VH = 17;        // widest vector block (message GVP 0: x_diff + 16 channels)
constexpr int RBF = 16;

// ---- kernels ------------------------------------------------------------------------------------------------------------
// edge geometry (gvp.py:474-480): unit vector x_diff / (|x_diff|_nonan + 1e-8) and the rbf code of that length
__global__ void k_gvp_geom(const int *__restrict__ src, const int *__restrict__ dst, const float *__restrict__ xs,
                           const float *__restrict__ xd, int E, float dmax, float *__restrict__ unit, float *__restrict__ rbf) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int u = src[e], v = dst[e];
    const float dx = xs[3 * u] - xd[3 * v], dy = xs[3 * u + 1] - xd[3 * v + 1], dz = xs[3 * u + 2] - xd[3 * v + 2];
    const float d = sqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-8f)) + 1e-8f;
    const float inv = 1.0f / d;
    unit[3 * e] = dx * inv; unit[3 * e + 1] = dy * inv; unit[3 * e + 2] = dz * inv;
    const float sigma = dmax / RBF;
#pragma unroll
    for (int k = 0; k < RBF; ++k) {
        const float mu = dmax * (float)k / (float)(RBF - 1);
        const float q = (d - mu) / sigma;
        rbf[(size_t)e * RBF + k] = __expf(-q * q);
    }
}

// Backward of k_gvp_geom: the loss reaches the positions through the unit edge vector (channel 0 of the message input vectors,
// dvin [E, 3, ldv]) and through the rbf code (drbf [E, 16]).  With diff = x_src - x_dst, q = |diff|^2, n = sqrt(max(q, 1e-8)),
// d = n + 1e-8:  unit = diff / d,  rbf_k = exp(-((d - mu_k) / sigma)^2);  where the clamp is active n does not depend on diff.
// dxe[e] = dL/d diff (added to the source node's gradient, subtracted from the destination's).
__global__ void k_gvp_geom_bwd(const int *__restrict__ src, const int *__restrict__ dst, const float *__restrict__ xs,
                               const float *__restrict__ xd, int E, float dmax, const float *__restrict__ rbf,
                               const float *__restrict__ drbf, const float *__restrict__ dvin, int ldv, float *__restrict__ dxe) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const int u = src[e], v = dst[e];
    const float df[3] = {xs[3 * u] - xd[3 * v], xs[3 * u + 1] - xd[3 * v + 1], xs[3 * u + 2] - xd[3 * v + 2]};
    const float q = df[0] * df[0] + df[1] * df[1] + df[2] * df[2];
    const float n = sqrtf(fmaxf(q, 1e-8f)), d = n + 1e-8f, inv = 1.0f / d;
    const float sigma = dmax / RBF;
    float dd = 0.0f;
#pragma unroll
    for (int k = 0; k < RBF; ++k) {
        const float mu = dmax * (float)k / (float)(RBF - 1);
        dd += drbf[(size_t)e * RBF + k] * rbf[(size_t)e * RBF + k] * (-2.0f * (d - mu) / (sigma * sigma));
    }
    float du[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        du[c] = dvin[((size_t)e * 3 + c) * ldv];
        dd -= du[c] * df[c] * inv * inv;
    }
    const float k = q > 1e-8f ? dd / n : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) dxe[3 * e + c] = du[c] * inv + k * df[c];
}

// gx[v][0..3) += sign * sum over j in [rowptr[v], rowptr[v + 1]) of dxe[perm ? perm[j] : j]: one thread per node, edges in
// ascending order (the deterministic sums of the rest of the backward pass)
__global__ void k_seg3(const float *__restrict__ dxe, const int *__restrict__ perm, const int *__restrict__ rowptr, int n, float sign,
                       float *__restrict__ gx) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    float a = 0.0f, b = 0.0f, c = 0.0f;
    for (int j = rowptr[v]; j < rowptr[v + 1]; ++j) {
        const int e = perm ? perm[j] : j;
        a += dxe[3 * e]; b += dxe[3 * e + 1]; c += dxe[3 * e + 2];
    }
    gx[3 * v] += sign * a; gx[3 * v + 1] += sign * b; gx[3 * v + 2] += sign * c;
}

// message input vectors [E, 3, 17]: channel 0 = unit edge vector, channels 1..16 = v_src[src] (gvp.py:545)
__global__ void k_gvp_vin(const float *__restrict__ unit, const float *__restrict__ vsrc, const int *__restrict__ src, long long total,
                          float *__restrict__ vin) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ch = (int)(i % VH);
    const long long ec = i / VH;               // e * 3 + c
    const int e = (int)(ec / 3), c = (int)(ec - 3LL * e);
    vin[i] = ch == 0 ? unit[3 * e + c] : vsrc[((size_t)src[e] * 3 + c) * VC + ch - 1];
}

// gv[v, c, ch] += sum over the out-edges j of source node v (ascending edge order) of dvin[perm[j], c, 1 + ch]: the gradient of the source
// vectors from that of the message inputs [x_diff | v_src] ([E, 3, 17], channels 1..16), the three components in one launch.  A quarter
// wave per node, a lane per (component, channel) pair (48 of 64).
__global__ __launch_bounds__(256) void k_segsum_vin(const float *__restrict__ dvin, const int *__restrict__ perm, const int *__restrict__ rowptr, int n,
                                                    float *__restrict__ gv) {
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (v >= n || lane >= 3 * VC) return;
    const int lo = rowptr[v], hi = rowptr[v + 1];
    if (lo == hi) return;
    const int off = (lane / VC) * VH + 1 + lane % VC;
    float s = 0.0f;
    int j = lo;
    for (; j + 4 <= hi; j += 4) {
        float m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] = dvin[(size_t)perm[j + k] * (3 * VH) + off];
#pragma unroll
        for (int k = 0; k < 4; ++k) s += m[k];
    }
    for (; j < hi; ++j) s += dvin[(size_t)perm[j] * (3 * VH) + off];
    gv[(size_t)v * (3 * VC) + lane] += s;
}

// the same for all edge types of a conv in one launch: blockIdx.y = source node type, its edge types added in edge-type order (as the separate
// launches did)
struct SegsumVinArgs {
    const float *dvin[4];
    const int *perm[4], *rowptr[4];
    int live[4], src_nt[4];
    int n[2];
    float *gv[2];
};
__global__ __launch_bounds__(256) void k_segsum_vin_all(SegsumVinArgs a) {
    const int nt = blockIdx.y, v = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (v >= a.n[nt] || lane >= 3 * VC) return;
    const int off = (lane / VC) * VH + 1 + lane % VC;
    float acc = a.gv[nt][(size_t)v * (3 * VC) + lane];
    bool any = false;
#pragma unroll
    for (int et = 0; et < 4; ++et) {
        if (!a.live[et] || a.src_nt[et] != nt) continue;
        const int lo = a.rowptr[et][v], hi = a.rowptr[et][v + 1];
        if (lo == hi) continue;
        const float *dvin = a.dvin[et];
        const int *perm = a.perm[et];
        float s = 0.0f;
        int j = lo;
        for (; j + 4 <= hi; j += 4) {
            float m[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) m[k] = dvin[(size_t)perm[j + k] * (3 * VH) + off];
#pragma unroll
            for (int k = 0; k < 4; ++k) s += m[k];
        }
        for (; j < hi; ++j) s += dvin[(size_t)perm[j] * (3 * VH) + off];
        acc += s;
        any = true;
    }
    if (any) a.gv[nt][(size_t)v * (3 * VC) + lane] = acc;
}

// sh[m, j] = sqrt(max(sum_c Vh[m, c, j]^2, 1e-8)) (_norm_no_nan, gvp.py:12-19)
__global__ void k_gvp_sh(const float *__restrict__ Vh, long long total, int h, float *__restrict__ sh) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long m = i / h;
    const int j = (int)(i - m * h);
    const float a = Vh[(m * 3) * h + j], b = Vh[(m * 3 + 1) * h + j], c = Vh[(m * 3 + 2) * h + j];
    sh[i] = sqrtf(fmaxf(a * a + b * b + c * c, 1e-8f));
}

// dVh[m, c, j] += dsh[m, j] * Vh[m, c, j] / sh[m, j] where the clamp is inactive
__global__ void k_gvp_sh_bwd(const float *__restrict__ Vh, const float *__restrict__ sh, const float *__restrict__ dsh, long long total,
                             int h, float *__restrict__ dVh) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // over [m, c, j]
    if (i >= total) return;
    const int j = (int)(i % h);
    const long long m = i / (3LL * h);
    const float s = sh[m * h + j];
    if (s * s > 1e-8f) dVh[i] += dsh[m * h + j] * Vh[i] / s;
}

// Vector half of GVP.forward up to the gate (gvp.py:97-104) for one row per thread: Vh = v_in Wh, sh = |Vh| over the three components
// (clamped, _norm_no_nan), Vu = Vh Wu.  As three launches (two <= 33-wide products through the GEMM and k_gvp_sh) the [3 M x 17]
// arrays were read and written five times by kernels that do ~600 FMAs per row; here each is touched once.  The weights sit in LDS
// (every lane reads the same element: broadcast), a row's 3 x VI inputs and 3 x H hidden values in registers.
template <int VI, int H, int VO>
__global__ __launch_bounds__(128) void k_gvp_vec_fwd(const float *__restrict__ v_in, const float *__restrict__ Wh, const float *__restrict__ Wu,
                                                     int M, float *__restrict__ Vh, float *__restrict__ Vu, float *__restrict__ sh) {
    __shared__ float s_wh[VI * H], s_wu[H * VO];
    for (int i = threadIdx.x; i < VI * H; i += blockDim.x) s_wh[i] = Wh[i];
    for (int i = threadIdx.x; i < H * VO; i += blockDim.x) s_wu[i] = Wu[i];
    __syncthreads();
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float sh2[H];
#pragma unroll
    for (int j = 0; j < H; ++j) sh2[j] = 0.0f;
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        const float *x = v_in + ((size_t)m * 3 + c) * VI;
        float xv[VI], vh[H];
#pragma unroll
        for (int i = 0; i < VI; ++i) xv[i] = x[i];
#pragma unroll
        for (int j = 0; j < H; ++j) vh[j] = 0.0f;
#pragma unroll
        for (int i = 0; i < VI; ++i) {
            asm volatile("" ::: "memory");          // keeps the weight row's LDS reads here (hoisted, all VI x H weights would live in registers)
#pragma unroll
            for (int j = 0; j < H; ++j) vh[j] = fmaf(xv[i], s_wh[i * H + j], vh[j]);
        }
        float *oh = Vh + ((size_t)m * 3 + c) * H;
#pragma unroll
        for (int j = 0; j < H; ++j) {
            oh[j] = vh[j];
            sh2[j] = fmaf(vh[j], vh[j], sh2[j]);
        }
        float vu[VO];
#pragma unroll
        for (int u = 0; u < VO; ++u) vu[u] = 0.0f;
#pragma unroll
        for (int j = 0; j < H; ++j) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < VO; ++u) vu[u] = fmaf(vh[j], s_wu[j * VO + u], vu[u]);
        }
        float *ou = Vu + ((size_t)m * 3 + c) * VO;
#pragma unroll
        for (int u = 0; u < VO; ++u) ou[u] = vu[u];
    }
    float *os = sh + (size_t)m * H;
#pragma unroll
    for (int j = 0; j < H; ++j) os[j] = sqrtf(fmaxf(sh2[j], 1e-8f));
}

// ---- 16-channel GVPs: the vector half as register-chained 16x16x4 MFMA products (chain_core.h conventions) --------------------------
// A wave owns groups of 16 rows m (x 3 components).  Lane (e = lane & 15, q = lane >> 4) holds channels 4 q .. 4 q + 3 of row e as one
// float4: that is a 16-byte piece of the row in memory (coalesced: the four q-lanes cover the row's 64 B), the B operand of a product
// T^T[n][e] = sum_k W'[n][k] X^T[k][e], and the layout the product's result comes back in.  The 16 x 16 weights are A-fragments in
// registers for the whole kernel.  Forward: Vh = v_in Wh, Vu = Vh Wu, sh = |Vh| over the components: one read and three writes of
// [rows x 16] arrays instead of five passes by three launches.
__device__ __forceinline__ v4f chain16(const v4f &w, const v4f &x) {
    v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[r], x[r], acc, 0, 0, 0);
    return acc;
}

__global__ __launch_bounds__(256) void k_gvp_vec16_fwd(const float *__restrict__ v_in, const float *__restrict__ Wh, const float *__restrict__ Wu,
                                                       int M, float *__restrict__ Vh, float *__restrict__ Vu, float *__restrict__ sh) {
    const int lane = threadIdx.x & 63, e = lane & 15, q = lane >> 4;
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), n_wv = gridDim.x * 4, G = (M + 15) >> 4;
    v4f aWh, aWu;                 // W'[n = e][k = 4 q + r] = W[4 q + r][e]
#pragma unroll
    for (int r = 0; r < 4; ++r) { aWh[r] = Wh[(4 * q + r) * 16 + e]; aWu[r] = Wu[(4 * q + r) * 16 + e]; }
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int g = wv; g < G; g += n_wv) {
        const int m = 16 * g + e;
        const bool on = m < M;
        const size_t row0 = (size_t)min(m, M - 1) * 3;
        v4f s2 = zero;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const size_t at = (row0 + c) * 16 + 4 * q;
            v4f x = *reinterpret_cast<const v4f *>(v_in + at);
            if (!on) x = zero;
            const v4f vh = chain16(aWh, x);
            const v4f vu = chain16(aWu, vh);
            s2 += vh * vh;
            if (on) {
                *reinterpret_cast<v4f *>(Vh + at) = vh;
                *reinterpret_cast<v4f *>(Vu + at) = vu;
            }
        }
        if (on) {
            v4f o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = sqrtf(fmaxf(s2[r], 1e-8f));
            *reinterpret_cast<v4f *>(sh + (size_t)m * 16 + 4 * q) = o;
        }
    }
}

// Backward of the same half: dVh = dVu Wu^T + dsh Vh / sh (clamp inactive), dv_in = dVh Wh^T, and the two weight gradients
// Wu.g += Vh^T dVu, Wh.g += v_in^T dVh.  The gradients are products over ROWS: their operands want "channel on lane & 15, row on (lane >> 4, r)",
// the transpose of what a lane holds, so each 16 x 16 tile takes one trip through a 17-float-stride LDS tile of its wave.  Every wave adds
// its groups in a fixed order, a workgroup adds its four waves in wave order and writes ONE partial per gradient; k_gvp_vec_reduce adds
// the partials in workgroup order (no atomics).
// dVh is not written: nothing else reads it.
constexpr int VEC16_MAX_WAVES = 4096;
__global__ __launch_bounds__(256) void k_gvp_vec16_bwd(const float *__restrict__ dVu, const float *__restrict__ Vh, const float *__restrict__ sh,
                                                       const float *__restrict__ dsh, const float *__restrict__ v_in,
                                                       const float *__restrict__ Wh, const float *__restrict__ Wu, int M,
                                                       float *__restrict__ dv_in, float *__restrict__ part) {
    __shared__ float s_t[4][4][16 * 17];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, e = lane & 15, q = lane >> 4;
    const int wv = blockIdx.x * 4 + wave, n_wv = gridDim.x * 4, G = (M + 15) >> 4;
    v4f aWuT, aWhT;               // dVh: W'[n = j][k = u] = Wu[j][u];  dv_in: W'[n = i][k = j] = Wh[i][j]
#pragma unroll
    for (int r = 0; r < 4; ++r) { aWuT[r] = Wu[e * 16 + 4 * q + r]; aWhT[r] = Wh[e * 16 + 4 * q + r]; }
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
    v4f gWu = zero, gWh = zero;
    float *t_vh = s_t[wave][0], *t_du = s_t[wave][1], *t_x = s_t[wave][2], *t_g = s_t[wave][3];
    auto put = [&](float *t, const v4f &v) {              // row e, channels 4 q ..
#pragma unroll
        for (int r = 0; r < 4; ++r) t[e * 17 + 4 * q + r] = v[r];
    };
    auto get = [&](const float *t) {                      // channel e of rows 4 q ..
        v4f v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = t[(4 * q + r) * 17 + e];
        return v;
    };
#pragma unroll 1
    for (int g = wv; g < G; g += n_wv) {
        const int m = 16 * g + e;
        const bool on = m < M;
        const size_t mm = (size_t)min(m, M - 1);
        v4f k;
        {
            const v4f s = *reinterpret_cast<const v4f *>(sh + mm * 16 + 4 * q), d = *reinterpret_cast<const v4f *>(dsh + mm * 16 + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) k[r] = (on && s[r] * s[r] > 1e-8f) ? d[r] / s[r] : 0.0f;
        }
#pragma unroll 1
        for (int c = 0; c < 3; ++c) {
            const size_t at = (mm * 3 + c) * 16 + 4 * q;
            v4f du = *reinterpret_cast<const v4f *>(dVu + at), vh = *reinterpret_cast<const v4f *>(Vh + at),
                x = *reinterpret_cast<const v4f *>(v_in + at);
            if (!on) { du = zero; vh = zero; x = zero; }
            v4f gh = chain16(aWuT, du);
            gh += k * vh;
            if (dv_in) {
                const v4f dx = chain16(aWhT, gh);
                if (on) *reinterpret_cast<v4f *>(dv_in + at) = dx;
            }
            put(t_vh, vh); put(t_du, du); put(t_x, x); put(t_g, gh);
            const v4f vhT = get(t_vh), duT = get(t_du), xT = get(t_x), gT = get(t_g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                gWu = __builtin_amdgcn_mfma_f32_16x16x4f32(vhT[r], duT[r], gWu, 0, 0, 0);       // [j][u] += Vh[row][j] dVu[row][u]
                gWh = __builtin_amdgcn_mfma_f32_16x16x4f32(xT[r], gT[r], gWh, 0, 0, 0);        // [i][j] += v_in[row][i] dVh[row][j]
            }
        }
    }
    // result element (row 4 q + r, column e) of either gradient: the four waves' sums are added in wave order, one partial per workgroup
    __syncthreads();                                       // (the transpose tiles are free now: reused as the exchange buffer)
    float *xw = &s_t[0][0][0] + wave * 512;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        xw[(4 * q + r) * 16 + e] = gWu[r];
        xw[256 + (4 * q + r) * 16 + e] = gWh[r];
    }
    __syncthreads();
    const float *x0 = &s_t[0][0][0];
    float *p = part + (size_t)blockIdx.x * 512;
    for (int i = threadIdx.x; i < 512; i += 256) p[i] = ((x0[i] + x0[512 + i]) + x0[1024 + i]) + x0[1536 + i];
}

// g_u[i] += sum over workgroups of part[b][i] (i < n_u), g_h[i - n_u] += ... (n_u <= i < n_u + n_h); partials `stride` floats apart.
// Workgroups of 16 outputs x 16 slices: slice s adds partials s, s + 16, ... in order, the sixteen slice sums are combined in slice order:
// a fixed tree.
__global__ __launch_bounds__(256) void k_gvp_vec_reduce(const float *__restrict__ part, int n_part, int stride, int n_u, int n_h,
                                                        float *__restrict__ g_u, float *__restrict__ g_h) {
    __shared__ float s_p[16][16];
    const int o = threadIdx.x & 15, sl = threadIdx.x >> 4, i = blockIdx.x * 16 + o;
    float s = 0.0f;
    if (i < n_u + n_h)
        for (int b = sl; b < n_part; b += 16) s += part[(size_t)b * stride + i];
    s_p[sl][o] = s;
    __syncthreads();
    if (sl == 0 && i < n_u + n_h) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += s_p[k][o];
        if (i < n_u) { if (g_u) g_u[i] += t; }
        else if (g_h) g_h[i - n_u] += t;
    }
}

// several such reductions in one launch (blockIdx.y = which): the vector-weight gradients of a conv's GVPs, each with a partial region of its own
struct VecRedBatch {
    struct One {
        const float *part;
        int n_part, stride, n_u, n_h;
        float *g_u, *g_h;
    };
    One r[16];
};
__global__ __launch_bounds__(256) void k_gvp_vec_reduce_batch(VecRedBatch b) {
    const VecRedBatch::One &a = b.r[blockIdx.y];
    __shared__ float s_p[16][16];
    const int o = threadIdx.x & 15, sl = threadIdx.x >> 4, i = blockIdx.x * 16 + o;
    float s = 0.0f;
    if (i < a.n_u + a.n_h)
        for (int k = sl; k < a.n_part; k += 16) s += a.part[(size_t)k * a.stride + i];
    s_p[sl][o] = s;
    __syncthreads();
    if (sl == 0 && i < a.n_u + a.n_h) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += s_p[k][o];
        if (i < a.n_u) { if (a.g_u) a.g_u[i] += t; }
        else if (a.g_h) a.g_h[i - a.n_u] += t;
    }
}
constexpr size_t VEC_PART_REGION = (size_t)512 * 576;          // floats of one call's partials: at most 2 x 256 workgroups x VEC17_PART

// ---- the message head GVP: 17 vector inputs [x_diff | 16 v_src], 17 hidden channels, 16 outputs (gvp.py:545, 395-415) -------------------
// Same chained scheme on the 16 x 16 core (inputs 1 .. 16, hidden 0 .. 15); input channel 0 and hidden channel 16 are rank-1 / dot-product
// updates on the VALU (a row's 16 core channels are spread over the four q-lanes: dots finish with two shuffles).  Rows are 17 floats
// apart, so the core channels move as scalars (four per lane) instead of one float4.
struct Head17W {
    v4f core, w0, wc;             // A-fragment of the 16 x 16 core; W[0][4 q ..] (input 0 -> hidden core); W[1 + 4 q ..][16] (core inputs -> hidden 16)
    float w016;                   // W[0][16]
};
__device__ __forceinline__ float sum_q(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ float sum_e(float v) {         // over the 16 rows of a lane group (same q)
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void k_gvp_vec17_fwd(const float *__restrict__ v_in, const float *__restrict__ Wh, const float *__restrict__ Wu,
                                                       int M, float *__restrict__ Vh, float *__restrict__ Vu, float *__restrict__ sh) {
    const int lane = threadIdx.x & 63, e = lane & 15, q = lane >> 4;
    const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), n_wv = gridDim.x * 4, G = (M + 15) >> 4;
    v4f aWh, w0, wc, aWu, wu16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        aWh[r] = Wh[(1 + 4 * q + r) * 17 + e];            // hidden e <- input 1 + 4 q + r
        w0[r] = Wh[4 * q + r];                             // hidden 4 q + r <- input 0
        wc[r] = Wh[(1 + 4 * q + r) * 17 + 16];            // hidden 16 <- input 1 + 4 q + r
        aWu[r] = Wu[(4 * q + r) * 16 + e];                // output e <- hidden 4 q + r
        wu16[r] = Wu[16 * 16 + 4 * q + r];                // output 4 q + r <- hidden 16
    }
    const float w016 = Wh[16];
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int g = wv; g < G; g += n_wv) {
        const int m = 16 * g + e;
        const bool on = m < M;
        const size_t row0 = (size_t)min(m, M - 1) * 3;
        v4f s2 = zero;
        float s2_16 = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float *xr = v_in + (row0 + c) * 17;
            float x0 = xr[0];
            v4f xs;
#pragma unroll
            for (int r = 0; r < 4; ++r) xs[r] = xr[1 + 4 * q + r];
            if (!on) { x0 = 0.0f; xs = zero; }
            v4f vh = chain16(aWh, xs);
            vh += x0 * w0;
            const float vh16 = sum_q(xs[0] * wc[0] + xs[1] * wc[1] + xs[2] * wc[2] + xs[3] * wc[3]) + x0 * w016;
            v4f vu = chain16(aWu, vh);
            vu += vh16 * wu16;
            s2 += vh * vh;
            s2_16 = fmaf(vh16, vh16, s2_16);
            if (on) {
                float *oh = Vh + (row0 + c) * 17;
#pragma unroll
                for (int r = 0; r < 4; ++r) oh[4 * q + r] = vh[r];
                if (q == 0) oh[16] = vh16;
                *reinterpret_cast<v4f *>(Vu + (row0 + c) * 16 + 4 * q) = vu;
            }
        }
        if (on) {
            float *os = sh + (size_t)m * 17;
#pragma unroll
            for (int r = 0; r < 4; ++r) os[4 * q + r] = sqrtf(fmaxf(s2[r], 1e-8f));
            if (q == 0) os[16] = sqrtf(fmaxf(s2_16, 1e-8f));
        }
    }
}

constexpr int VEC17_PART = 576;          // floats per partial: Wu.g [17][16] (272), Wh.g [17][17] (289), padding
__global__ __launch_bounds__(256) void k_gvp_vec17_bwd(const float *__restrict__ dVu, const float *__restrict__ Vh, const float *__restrict__ sh,
                                                       const float *__restrict__ dsh, const float *__restrict__ v_in,
                                                       const float *__restrict__ Wh, const float *__restrict__ Wu, int M,
                                                       float *__restrict__ dv_in, float *__restrict__ part) {
    __shared__ float s_t[4][4][16 * 17];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, e = lane & 15, q = lane >> 4;
    const int wv = blockIdx.x * 4 + wave, n_wv = gridDim.x * 4, G = (M + 15) >> 4;
    v4f aWuT, wu16, aWhT, w0, wc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        aWuT[r] = Wu[e * 16 + 4 * q + r];                 // d hidden e <- d output 4 q + r
        wu16[r] = Wu[16 * 16 + 4 * q + r];                // d hidden 16 <- d output 4 q + r
        aWhT[r] = Wh[(1 + e) * 17 + 4 * q + r];           // d input 1 + e <- d hidden 4 q + r
        w0[r] = Wh[4 * q + r];                             // d input 0 <- d hidden 4 q + r
        wc[r] = Wh[(1 + 4 * q + r) * 17 + 16];            // d input 1 + 4 q + r <- d hidden 16
    }
    const float w016 = Wh[16];
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
    v4f gWu = zero, gWh = zero;                            // MFMA blocks: Wu.g[j < 16][u], Wh.g[1 + i'][j < 16]
    v4f gWu16 = zero, gWh0 = zero, gWhc = zero;            // per-lane sums: Wu.g[16][4 q ..], Wh.g[0][4 q ..], Wh.g[1 + 4 q ..][16]
    float gWh016 = 0.0f;                                   // Wh.g[0][16] (q == 0 lanes only)
    float *t_vh = s_t[wave][0], *t_du = s_t[wave][1], *t_x = s_t[wave][2], *t_g = s_t[wave][3];
    auto put = [&](float *t, const v4f &v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) t[e * 17 + 4 * q + r] = v[r];
    };
    auto get = [&](const float *t) {
        v4f v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = t[(4 * q + r) * 17 + e];
        return v;
    };
#pragma unroll 1
    for (int g = wv; g < G; g += n_wv) {
        const int m = 16 * g + e;
        const bool on = m < M;
        const size_t mm = (size_t)min(m, M - 1);
        v4f k;
        float k16;
        {
            const float *sr = sh + mm * 17, *dr = dsh + mm * 17;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sv = sr[4 * q + r];
                k[r] = (on && sv * sv > 1e-8f) ? dr[4 * q + r] / sv : 0.0f;
            }
            const float s16 = sr[16];
            k16 = (on && s16 * s16 > 1e-8f) ? dr[16] / s16 : 0.0f;
        }
#pragma unroll 1
        for (int c = 0; c < 3; ++c) {
            const size_t row = mm * 3 + c;
            v4f du = *reinterpret_cast<const v4f *>(dVu + row * 16 + 4 * q), vh, xs;
            const float *hr = Vh + row * 17, *xr = v_in + row * 17;
#pragma unroll
            for (int r = 0; r < 4; ++r) { vh[r] = hr[4 * q + r]; xs[r] = xr[1 + 4 * q + r]; }
            float vh16 = hr[16], x0 = xr[0];
            if (!on) { du = zero; vh = zero; xs = zero; vh16 = 0.0f; x0 = 0.0f; }
            v4f gh = chain16(aWuT, du);
            gh += k * vh;
            const float gh16 = sum_q(du[0] * wu16[0] + du[1] * wu16[1] + du[2] * wu16[2] + du[3] * wu16[3]) + k16 * vh16;
            if (dv_in) {
                v4f dxs = chain16(aWhT, gh);
                dxs += gh16 * wc;
                const float dx0 = sum_q(gh[0] * w0[0] + gh[1] * w0[1] + gh[2] * w0[2] + gh[3] * w0[3]) + gh16 * w016;
                if (on) {
                    float *od = dv_in + row * 17;
#pragma unroll
                    for (int r = 0; r < 4; ++r) od[1 + 4 * q + r] = dxs[r];
                    if (q == 0) od[0] = dx0;
                }
            }
            // weight gradients: the 16 x 16 blocks through transposed operands on the MFMA, the extra row / column as per-lane sums
            put(t_vh, vh); put(t_du, du); put(t_x, xs); put(t_g, gh);
            const v4f vhT = get(t_vh), duT = get(t_du), xT = get(t_x), gT = get(t_g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                gWu = __builtin_amdgcn_mfma_f32_16x16x4f32(vhT[r], duT[r], gWu, 0, 0, 0);       // [j][u] += Vh[row][j] dVu[row][u]
                gWh = __builtin_amdgcn_mfma_f32_16x16x4f32(xT[r], gT[r], gWh, 0, 0, 0);        // [i'][j] += v_in[row][1 + i'] dVh[row][j]
            }
            gWu16 += vh16 * du;                            // Wu.g[16][u]   += Vh[row][16] dVu[row][u]
            gWh0 += x0 * gh;                               // Wh.g[0][j]    += v_in[row][0] dVh[row][j]
            gWhc += gh16 * xs;                             // Wh.g[1+i'][16] += v_in[row][1 + i'] dVh[row][16]
            if (q == 0) gWh016 = fmaf(x0, gh16, gWh016);   // Wh.g[0][16]
        }
    }
    // per-lane sums -> per-wave sums over the 16 rows of the lane group
#pragma unroll
    for (int r = 0; r < 4; ++r) { gWu16[r] = sum_e(gWu16[r]); gWh0[r] = sum_e(gWh0[r]); gWhc[r] = sum_e(gWhc[r]); }
    gWh016 = sum_e(gWh016);
    __syncthreads();                                       // (the transpose tiles are free now: reused as the exchange buffer)
    float *xw = &s_t[0][0][0] + wave * VEC17_PART;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        xw[(4 * q + r) * 16 + e] = gWu[r];                             // Wu.g[j = 4 q + r][u = e]
        xw[272 + (1 + 4 * q + r) * 17 + e] = gWh[r];                   // Wh.g[1 + i'][j = e]
        if (e == 0) {
            xw[256 + 4 * q + r] = gWu16[r];                            // Wu.g[16][4 q + r]
            xw[272 + 4 * q + r] = gWh0[r];                             // Wh.g[0][4 q + r]
            xw[272 + (1 + 4 * q + r) * 17 + 16] = gWhc[r];             // Wh.g[1 + 4 q + r][16]
        }
    }
    if (lane == 0) xw[272 + 16] = gWh016;                              // Wh.g[0][16]
    __syncthreads();
    const float *x0p = &s_t[0][0][0];
    float *p = part + (size_t)blockIdx.x * VEC17_PART;
    for (int i = threadIdx.x; i < 272 + 289; i += 256)
        p[i] = ((x0p[i] + x0p[VEC17_PART + i]) + x0p[2 * VEC17_PART + i]) + x0p[3 * VEC17_PART + i];
}

inline bool vec_fused() {
    static const bool on = tool_env_int("KPD_TRAIN_VEC_FUSED", 1) != 0;          // A/B runs
    return on;
}

template <int VI, int H, int VO>
kpd_status launch_gvp_vec_fwd(const float *v_in, const float *Wh, const float *Wu, int M, float *Vh, float *Vu, float *sh, hipStream_t st) {
    hipLaunchKernelGGL((k_gvp_vec_fwd<VI, H, VO>), dim3(cdiv(M, 128)), dim3(128), 0, st, v_in, Wh, Wu, M, Vh, Vu, sh);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// V[m, c, u] = act(gate[m, u]) * Vu[m, c, u], act = sigmoid or identity (gvp.py:108-114); one thread per (m, VEC channels).
// part1 (optional): gate holds the first column half's share of the gate product (ws_gemm), part1 the second's: gate <- gate + part1 + bias
template <int VEC>
__global__ void k_gvp_gate(float *__restrict__ gate, const float *__restrict__ part1, const float *__restrict__ bias, const float *__restrict__ Vu,
                           int rows, int vw, int identity, float *__restrict__ V) {
    typedef float vt __attribute__((ext_vector_type(VEC)));
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // over [m][u / VEC]
    if (i >= rows * vw) return;
    const int m = i / vw, u = i - m * vw;
    vt g = reinterpret_cast<const vt *>(gate)[i];
    if (part1) {
        g += reinterpret_cast<const vt *>(part1)[i] + reinterpret_cast<const vt *>(bias)[u];
        reinterpret_cast<vt *>(gate)[i] = g;
    }
    if (!identity)
#pragma unroll
        for (int r = 0; r < VEC; ++r) g[r] = sigm(g[r]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const size_t k = ((size_t)m * 3 + c) * vw + u;
        reinterpret_cast<vt *>(V)[k] = g * reinterpret_cast<const vt *>(Vu)[k];
    }
}

// one thread per (m, VEC channels): dgate = sum_c dV Vu act'(gate); dV <- dV act(gate) (= dVu)
template <int VEC>
__global__ void k_gvp_gate_bwd(const float *__restrict__ gate, const float *__restrict__ Vu, int rows, int vw, int identity, float *__restrict__ dV,
                               float *__restrict__ dgate) {
    typedef float vt __attribute__((ext_vector_type(VEC)));
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // over [m][u / VEC]
    if (i >= rows * vw) return;
    const int m = i / vw, u = i - m * vw;
    vt a = reinterpret_cast<const vt *>(gate)[i], da = 1.0f;
    if (!identity)
#pragma unroll
        for (int r = 0; r < VEC; ++r) {
            a[r] = sigm(a[r]);
            da[r] = a[r] * (1.0f - a[r]);
        }
    vt s = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const size_t k = ((size_t)m * 3 + c) * vw + u;
        const vt d = reinterpret_cast<vt *>(dV)[k];
        s += d * reinterpret_cast<const vt *>(Vu)[k];
        reinterpret_cast<vt *>(dV)[k] = d * a;
    }
    reinterpret_cast<vt *>(dgate)[i] = s * da;
}

// LayerNorm over `cols` (<= 512) columns, one wave per row
__global__ void k_ln_fwd(const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta, int rows, int cols,
                         float *__restrict__ out) {
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float *xr = x + (size_t)r * cols;
    float s = 0.0f;
    for (int c = lane; c < cols; c += 64) s += xr[c];
    const float mean = wave_sum(s) / cols;
    float q = 0.0f;
    for (int c = lane; c < cols; c += 64) q += (xr[c] - mean) * (xr[c] - mean);
    const float rstd = rsqrtf(wave_sum(q) / cols + 1e-5f);
    for (int c = lane; c < cols; c += 64) out[(size_t)r * cols + c] = (xr[c] - mean) * rstd * gamma[c] + beta[c];
}

__global__ void k_ln_bwd_g(const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ dy, int rows, int cols,
                           float *__restrict__ dx, float *__restrict__ dyxhat) {
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float *xr = x + (size_t)r * cols, *dyr = dy + (size_t)r * cols;
    float s = 0.0f;
    for (int c = lane; c < cols; c += 64) s += xr[c];
    const float mean = wave_sum(s) / cols;
    float q = 0.0f;
    for (int c = lane; c < cols; c += 64) q += (xr[c] - mean) * (xr[c] - mean);
    const float rstd = rsqrtf(wave_sum(q) / cols + 1e-5f);
    float sg = 0.0f, sgx = 0.0f;
    for (int c = lane; c < cols; c += 64) {
        const float xh = (xr[c] - mean) * rstd, g = dyr[c] * gamma[c];
        sg += g;
        sgx += g * xh;
    }
    sg = wave_sum(sg) / cols;
    sgx = wave_sum(sgx) / cols;
    for (int c = lane; c < cols; c += 64) {
        const float xh = (xr[c] - mean) * rstd, d = dyr[c];
        dyxhat[(size_t)r * cols + c] = d * xh;
        dx[(size_t)r * cols + c] = rstd * (d * gamma[c] - sg - xh * sgx);
    }
}

// The same backward with the two parameter gradients' row sums taken on the way: a workgroup owns LN_GP_ROWS rows (a wave every fourth of them,
// one after the other: 64 rows per workgroup left the node-sized launches with one workgroup per CU and 16 dependent rows per wave -- 93 us against
// the 16 us of k_ln_bwd_g), keeps sum_r dy xhat and sum_r dy per column in registers and leaves them as part[block][0 | 1][c] -- the partial format
// of k_colsum, so k_colsum_reduce finishes gamma.grad and beta.grad in one launch (instead of a dy * xhat array, two column-sum launches over it
// and over dy, and their two reductions).  cols <= 256.
constexpr int LN_GP_ROWS = 8;
// dx may be dy: a lane writes only the entries it has read itself, after its last read of them.
__global__ __launch_bounds__(256) void k_ln_bwd_gp(const float *__restrict__ x, const float *__restrict__ gamma, const float *dy, int rows, int cols,
                                                   float *dx, float *__restrict__ part) {
    __shared__ float s_gx[4][256], s_g[4][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = blockIdx.x * LN_GP_ROWS, r1 = min(rows, r0 + LN_GP_ROWS);
    float agx[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ag[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int r = r0 + wave; r < r1; r += 4) {
        const float *xr = x + (size_t)r * cols, *dyr = dy + (size_t)r * cols;
        float s = 0.0f;
        for (int c = lane; c < cols; c += 64) s += xr[c];
        const float mean = wave_sum(s) / cols;
        float q = 0.0f;
        for (int c = lane; c < cols; c += 64) q += (xr[c] - mean) * (xr[c] - mean);
        const float rstd = rsqrtf(wave_sum(q) / cols + 1e-5f);
        float sg = 0.0f, sgx = 0.0f;
        for (int c = lane; c < cols; c += 64) {
            const float xh = (xr[c] - mean) * rstd, g = dyr[c] * gamma[c];
            sg += g;
            sgx += g * xh;
        }
        sg = wave_sum(sg) / cols;
        sgx = wave_sum(sgx) / cols;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane + 64 * j;
            if (c < cols) {
                const float xh = (xr[c] - mean) * rstd, d = dyr[c];
                agx[j] += d * xh;
                ag[j] += d;
                dx[(size_t)r * cols + c] = rstd * (d * gamma[c] - sg - xh * sgx);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s_gx[wave][lane + 64 * j] = agx[j];
        s_g[wave][lane + 64 * j] = ag[j];
    }
    __syncthreads();
    const int c = threadIdx.x;
    if (c < cols) {
        float *p = part + (size_t)blockIdx.x * 2 * COLSUM_LD;
        p[c] = ((s_gx[0][c] + s_gx[1][c]) + s_gx[2][c]) + s_gx[3][c];
        p[COLSUM_LD + c] = ((s_g[0][c] + s_g[1][c]) + s_g[2][c]) + s_g[3][c];
    }
}

// vector part of GVPLayerNorm (gvp.py:162-165): v / (sqrt(mean_ch(max(|v_ch|^2, 1e-8)) + 1e-5) + 1e-5), one thread per row
// vl = the model's vector_size: channels vl .. 15 are zero padding (a narrower model trained in the 16-channel layout) and take no part in the mean
__global__ void k_vnorm_fwd(const float *__restrict__ v, int rows, int vl, float *__restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float *p = v + (size_t)r * 3 * VC;
    float m = 0.0f;
#pragma unroll
    for (int ch = 0; ch < VC; ++ch)
        if (ch < vl) m += fmaxf(p[ch] * p[ch] + p[VC + ch] * p[VC + ch] + p[2 * VC + ch] * p[2 * VC + ch], 1e-8f);
    const float inv = 1.0f / (sqrtf(m / vl + 1e-5f) + 1e-5f);
    for (int k = 0; k < 3 * VC; ++k) out[(size_t)r * 3 * VC + k] = p[k] * inv;
}

// Sixteen lanes per row (lane = channel, its three components), four rows per wave: the row sums are 16-lane shuffles and every access is a 64-byte
// piece of the row (one thread per row read and wrote its 2 x 48 floats 192 bytes apart from its neighbours': 29 us for 19 200 rows).
// dv may be dout (every thread reads its three gradient entries before it writes them).
__global__ __launch_bounds__(256) void k_vnorm_bwd(const float *__restrict__ v, const float *dout, int rows, int vl, float *dv) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, r = t >> 4, ch = t & 15;
    const bool in = r < rows;
    const size_t base = (size_t)(in ? r : rows - 1) * 3 * VC + ch;
    const float p0 = v[base], p1 = v[base + VC], p2 = v[base + 2 * VC], d0 = dout[base], d1 = dout[base + VC], d2 = dout[base + 2 * VC];
    const float n2 = p0 * p0 + p1 * p1 + p2 * p2;
    const bool used = ch < vl, live = n2 > 1e-8f && used;
    float m = used ? fmaxf(n2, 1e-8f) : 0.0f, dot = used ? d0 * p0 + d1 * p1 + d2 * p2 : 0.0f;
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) {
        m += __shfl_xor(m, o);
        dot += __shfl_xor(dot, o);
    }
    const float root = sqrtf(m / vl + 1e-5f), vn = root + 1e-5f, inv = 1.0f / vn;
    // out = v / vn; dvn = -(dout . v) / vn^2; dvn/dv[ch, c] = v[ch, c] / (vl * root) where the clamp is inactive
    const float k = -dot * inv * inv / (vl * root);
    if (!in) return;
    dv[base] = used ? d0 * inv + (live ? k * p0 : 0.0f) : 0.0f;
    dv[base + VC] = used ? d1 * inv + (live ? k * p1 : 0.0f) : 0.0f;
    dv[base + 2 * VC] = used ? d2 * inv + (live ? k * p2 : 0.0f) : 0.0f;
}

__global__ void k_add(const float *__restrict__ a, const float *__restrict__ b, long long n, float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

__global__ void k_acc(float *__restrict__ a, const float *__restrict__ b, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += b[i];
}

// reference [rows, vl, 3] <-> engine [rows, 3, VC] (vl = the model's vector_size <= VC; engine channels vl .. 15 are zeros)
__global__ void k_v_transpose(const float *__restrict__ in, long long rows, int to_internal, int vl, float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (to_internal) {          // out[r][c][ch] = in[r][ch][c]
        if (i >= rows * 3 * VC) return;
        const long long r = i / (3 * VC);
        const int k = (int)(i - r * 3 * VC);
        const int c = k / VC, ch = k - c * VC;
        out[i] = ch < vl ? in[r * 3 * vl + ch * 3 + c] : 0.0f;
    } else {                    // out[r][ch][c] = in[r][c][ch]
        if (i >= rows * 3 * vl) return;
        const long long r = i / (3 * vl);
        const int k = (int)(i - r * 3 * vl);
        const int ch = k / 3, c = k - ch * 3;
        out[i] = in[r * 3 * VC + c * VC + ch];
    }
}

// encoder input rows [h_0, t[graph]] (dynamics_gvp.py:161-169)
__global__ void k_cat_time(const float *__restrict__ h, int F, const float *__restrict__ t, const int *__restrict__ bidx, long long total,
                           float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int r = (int)(i / (F + 1)), c = (int)(i - (long long)r * (F + 1));
    out[i] = c < F ? h[(size_t)r * F + c] : t[bidx[r]];
}

// per-node scale of the aggregated messages: 'mean' -> 1 / in-degree of this edge type; else 1 / norm (constant or z[graph])
__global__ void k_msg_scale(const int *__restrict__ rowptr, const float *__restrict__ z, const int *__restrict__ bidx, int n, int mode,
                            float norm, float *__restrict__ scale) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    if (mode == 1) {
        const int deg = rowptr[v + 1] - rowptr[v];
        scale[v] = deg > 0 ? 1.0f / deg : 0.0f;
    } else {
        scale[v] = 1.0f / (mode == 2 ? z[bidx[v]] : norm);
    }
}

// keep mask of one dropout stream: element i is kept iff its Philox word >= rate * 2^32; kept elements scale by 1 / (1 - rate)
__device__ __forceinline__ float dropout_scale(unsigned long long seed, unsigned stream, long long i, float rate) {
    unsigned c[4] = {(unsigned)(i >> 2), (unsigned)((unsigned long long)i >> 34), stream, 0x6b70646fu};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
    const unsigned w = c[i & 3];
    const unsigned thr = (unsigned)fminf(rate * 4294967296.0f, 4294967040.0f);
    return w >= thr ? 1.0f / (1.0f - rate) : 0.0f;
}

// out[r, k, c] = in[r, k, c] * mask(r, c): rows x inner x cols with the mask shared over `inner` (1 for scalars, 3 for the
// components of a vector channel)
// `live` <= cols: the mask stream is laid out [rows, live] (what a model of that width draws); columns past it are padding and stay 0
__global__ void k_dropout(const float *__restrict__ in, long long rows, int inner, int cols, int live, unsigned long long seed, unsigned stream,
                          float rate, float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * inner * cols) return;
    const int c = (int)(i % cols);
    const long long r = i / ((long long)inner * cols);
    out[i] = c < live ? in[i] * dropout_scale(seed, stream, r * live + c, rate) : 0.0f;
}

__global__ void k_dropout_mask(long long n, unsigned long long seed, unsigned stream, float rate, float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = dropout_scale(seed, stream, i, rate);
}

struct GvpP {
    Param Wh, Wu, Ws, bs, Wg, bg;
    int vi = 0, vo = 0, h = 0, si = 0, so = 0;
};

struct GvpBuf {
    float *Vh = nullptr, *Vu = nullptr, *sh = nullptr, *pre = nullptr, *s = nullptr, *gate = nullptr, *V = nullptr;
};


template <class TT>
kpd_status gvp_params(TT *T, const std::string &p, int vi, int vo, int si, int so, GvpP *g) {
    g->vi = vi; g->vo = vo; g->h = std::max(vi, vo); g->si = si; g->so = so;
    KPD_TRY(param(T, p + ".Wh", vi, g->h, &g->Wh));
    KPD_TRY(param(T, p + ".Wu", g->h, vo, &g->Wu));
    KPD_TRY(param(T, p + ".to_feats_out.0.weight", so, si + g->h, &g->Ws));
    KPD_TRY(param(T, p + ".to_feats_out.0.bias", so, 1, &g->bs));
    KPD_TRY(param(T, p + ".scalar_to_vector_gates.weight", vo, so, &g->Wg));
    KPD_TRY(param(T, p + ".scalar_to_vector_gates.bias", vo, 1, &g->bg));
    return KPD_OK;
}

// the 256 x 256 scalar block of a GVP can take the weight-stationary GEMM (KPD_TRAIN_WS=0: library GEMMs throughout)
bool ws_ok(const GvpP &g, int ld_s) {
    static const bool on = tool_env_int("KPD_TRAIN_WS", 1) != 0;
    return on && g.si == 256 && g.so == 256 && ld_s == 256;
}

// the 16 / 17 vector norms of to_feats_out's input ride along in the weight-stationary kernel (KPD_TRAIN_WS_EXTRA=0: separate products)
inline bool ws_extra() {
    static const bool on = tool_env_int("KPD_TRAIN_WS_EXTRA", 1) != 0;
    return on;
}

// GVP.forward (gvp.py:89-116).  s_in == nullptr: B.pre already holds the contribution of the scalar inputs (no bias).
template <class TT>
kpd_status gvp_fwd(TT *T, const GvpP &g, int M, const float *s_in, int ld_s, const float *v_in, const GvpBuf &B,
                   bool identity) {
    if (M == 0) return KPD_OK;
    bool fused = false;
    if (vec_fused()) {          // the shapes the engines use: message head [x_diff | v_src] (17), plain (16), noise head (16 -> 1), encoder rk head (33)
        fused = true;
        if (g.vi == 17 && g.h == 17 && g.vo == 16) {
            const int blocks = std::max(1, std::min(cdiv(cdiv(M, 16), 4), 4 * cu_count()));
            hipLaunchKernelGGL(k_gvp_vec17_fwd, dim3(blocks), dim3(256), 0, T->st, v_in, g.Wh.w, g.Wu.w, M, B.Vh, B.Vu, B.sh);
            KPD_LAUNCH_CHECK();
        }
        else if (g.vi == 16 && g.h == 16 && g.vo == 16) {
            const int blocks = std::max(1, std::min(cdiv(cdiv(M, 16), 4), 4 * cu_count()));
            hipLaunchKernelGGL(k_gvp_vec16_fwd, dim3(blocks), dim3(256), 0, T->st, v_in, g.Wh.w, g.Wu.w, M, B.Vh, B.Vu, B.sh);
            KPD_LAUNCH_CHECK();
        }
        else if (g.vi == 16 && g.h == 16 && g.vo == 1) KPD_TRY((launch_gvp_vec_fwd<16, 16, 1>(v_in, g.Wh.w, g.Wu.w, M, B.Vh, B.Vu, B.sh, T->st)));
        else if (g.vi == 33 && g.h == 33 && g.vo == 16) KPD_TRY((launch_gvp_vec_fwd<33, 33, 16>(v_in, g.Wh.w, g.Wu.w, M, B.Vh, B.Vu, B.sh, T->st)));
        else fused = false;
    }
    if (!fused) {
        KPD_TRY(gemm(T, false, false, 3 * M, g.h, g.vi, v_in, g.vi, g.Wh.w, g.h, 0.0f, B.Vh, g.h));
        KPD_TRY(gemm(T, false, false, 3 * M, g.vo, g.h, B.Vh, g.h, g.Wu.w, g.vo, 0.0f, B.Vu, g.vo));
        hipLaunchKernelGGL(k_gvp_sh, grid1((long long)M * g.h), dim3(256), 0, T->st, B.Vh, (long long)M * g.h, g.h, B.sh);
        KPD_LAUNCH_CHECK();
    }
    long long tot = (long long)M * g.so;
    bool gate_parts = false;
    if (s_in && ws_ok(g, ld_s)) {
        // the narrow vector-norm block first, then the 256 x 256 scalar block on the weight-stationary GEMM with the partial
        // pre-activation, the bias and the SiLU fused into its epilogue
        if (g.h <= 17 && ws_extra()) {           // ... with the vector-norm block as extra inputs of the same kernel, and the gate product
            WsgExtra x;                          // (16 outputs of the activated row) taken in its epilogue
            x.X2 = B.sh; x.W = g.Ws.w + g.si; x.sn = g.si + g.h; x.sk = 1; x.n = g.h; x.ld = g.h;
            if (g.vo == 16) { x.Wg = g.Wg.w; x.ldg = g.so; x.ng = g.vo; x.G2 = B.gate; x.G2b = T->dgate; gate_parts = true; }
            KPD_TRY(ws_gemm(WS_BIAS_SILU, s_in, M, ld_s, g.Ws.w, g.si + g.h, false, g.bs.w, nullptr, B.pre, B.s, g.so, T->wsg_pack, T->st, false, false,
                            nullptr, 1, nullptr, &x));
        } else {
            KPD_TRY(gemm(T, false, true, M, g.so, g.h, B.sh, g.h, g.Ws.w + g.si, g.si + g.h, 0.0f, B.pre, g.so));
            KPD_TRY(ws_gemm(WS_BIAS_SILU, s_in, M, ld_s, g.Ws.w, g.si + g.h, false, g.bs.w, nullptr, B.pre, B.s, g.so, T->wsg_pack, T->st, false, true));
        }
    } else {
        if (s_in) KPD_TRY(gemm(T, false, true, M, g.so, g.si, s_in, ld_s, g.Ws.w, g.si + g.h, 0.0f, B.pre, g.so));
        // + the vector-norm block, the bias and the SiLU in the epilogue: B.pre keeps the pre-activation, B.s its activation
        KPD_TRY(gemm(T, false, true, M, g.so, g.h, B.sh, g.h, g.Ws.w + g.si, g.si + g.h, 1.0f, B.pre, g.so, 1.0f, nullptr, g.bs.w, B.s));
    }
    if (!gate_parts) KPD_TRY(gemm(T, false, true, M, g.vo, g.so, B.s, g.so, g.Wg.w, g.so, 0.0f, B.gate, g.vo, 1.0f, nullptr, g.bg.w));
    tot = (long long)M * g.vo;
    const float *p1 = gate_parts ? T->dgate : nullptr;
    if ((g.vo & 3) == 0) hipLaunchKernelGGL(k_gvp_gate<4>, grid1(tot / 4), dim3(256), 0, T->st, B.gate, p1, g.bg.w, B.Vu, M, g.vo / 4, identity ? 1 : 0, B.V);
    else hipLaunchKernelGGL(k_gvp_gate<1>, grid1(tot), dim3(256), 0, T->st, B.gate, p1, g.bg.w, B.Vu, M, g.vo, identity ? 1 : 0, B.V);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// Backward of gvp_fwd.  ds [M, so] = dL/ds' (overwritten with dL/dpre), dV [M, 3, vo] = dL/dV' (overwritten with dL/dVu);
// ds_in [M, si] (ld so-independent: compact si) and dv_in [M, 3, vi] are written when non-null.
template <class TT>
kpd_status gvp_bwd(TT *T, const GvpP &g, int M, const float *s_in, int ld_s, const float *v_in, const GvpBuf &B,
                   bool identity, float *ds, float *dV, float *ds_in, float *dv_in) {
    if (M == 0) return KPD_OK;
    long long tot = (long long)M * g.vo;
    if ((g.vo & 3) == 0) hipLaunchKernelGGL(k_gvp_gate_bwd<4>, grid1(tot / 4), dim3(256), 0, T->st, B.gate, B.Vu, M, g.vo / 4, identity ? 1 : 0, dV, T->dgate);
    else hipLaunchKernelGGL(k_gvp_gate_bwd<1>, grid1(tot), dim3(256), 0, T->st, B.gate, B.Vu, M, g.vo, identity ? 1 : 0, dV, T->dgate);
    KPD_LAUNCH_CHECK();
    KPD_TRY(grad_gemm(T, g.vo, g.so, M, T->dgate, g.vo, B.s, g.so, g.Wg.g, g.so, g.bg.g));       // + gate bias gradient (column sums of dgate)
    // ds = (ds + dgate Wg) * SiLU'(pre): the activation derivative in the product's epilogue
    KPD_TRY(gemm(T, false, false, M, g.so, g.vo, T->dgate, g.vo, g.Wg.w, g.so, 1.0f, ds, g.so, 1.0f, B.pre));
    bool have_dsh = false;
    if (s_in) {
        if (g.Ws.g) KPD_TRY(grad_gemm(T, g.so, g.si, M, ds, g.so, s_in, ld_s, g.Ws.g, g.si + g.h));
        if (ds_in) {
            if (ws_ok(g, ld_s) && g.h <= 17 && ws_extra()) {       // ... with dsh = ds Ws[:, si:] as extra outputs of the same kernel
                WsgExtra x;
                x.Y2 = T->dsh; x.W = g.Ws.w + g.si; x.sn = 1; x.sk = g.si + g.h; x.n = g.h; x.ld = g.h;
                KPD_TRY(ws_gemm(WS_PLAIN, ds, M, g.so, g.Ws.w, g.si + g.h, true, nullptr, nullptr, ds_in, nullptr, g.si, T->wsg_pack, T->st, false, false,
                                nullptr, 1, nullptr, &x));
                have_dsh = true;
            } else if (ws_ok(g, ld_s)) KPD_TRY(ws_gemm(WS_PLAIN, ds, M, g.so, g.Ws.w, g.si + g.h, true, nullptr, nullptr, ds_in, nullptr, g.si, T->wsg_pack, T->st, false, false));
            else KPD_TRY(gemm(T, false, false, M, g.si, g.so, ds, g.so, g.Ws.w, g.si + g.h, 0.0f, ds_in, g.si));
        }
    }
    // the sh block of to_feats_out, with the bias gradient (column sums of ds) riding along
    KPD_TRY(grad_gemm(T, g.so, g.h, M, ds, g.so, B.sh, g.h, g.Ws.g ? g.Ws.g + g.si : nullptr, g.si + g.h, g.bs.g));
    if (!have_dsh) KPD_TRY(gemm(T, false, false, M, g.h, g.so, ds, g.so, g.Ws.w + g.si, g.si + g.h, 0.0f, T->dsh, g.h));
    if (vec_fused() && g.vi == 17 && g.h == 17 && g.vo == 16 && T->part && T->part_floats >= (size_t)VEC16_MAX_WAVES * VEC17_PART) {
        const int blocks = std::max(1, std::min(cdiv(cdiv(M, 16), 4), std::min(2 * cu_count(), VEC16_MAX_WAVES / 4)));
        hipLaunchKernelGGL(k_gvp_vec17_bwd, dim3(blocks), dim3(256), 0, T->st, dV, B.Vh, B.sh, T->dsh, v_in, g.Wh.w, g.Wu.w, M, dv_in, T->part);
        KPD_LAUNCH_CHECK();
        if (g.Wu.g || g.Wh.g) {
            hipLaunchKernelGGL(k_gvp_vec_reduce, dim3(cdiv(272 + 289, 16)), dim3(256), 0, T->st, T->part, blocks, VEC17_PART, 272, 289, g.Wu.g, g.Wh.g);
            KPD_LAUNCH_CHECK();
        }
        return KPD_OK;
    }
    if (vec_fused() && g.vi == 16 && g.h == 16 && g.vo == 16 && T->part && T->part_floats >= (size_t)VEC16_MAX_WAVES * 512) {
        const int blocks = std::max(1, std::min(cdiv(cdiv(M, 16), 4), std::min(2 * cu_count(), VEC16_MAX_WAVES / 4)));
        hipLaunchKernelGGL(k_gvp_vec16_bwd, dim3(blocks), dim3(256), 0, T->st, dV, B.Vh, B.sh, T->dsh, v_in, g.Wh.w, g.Wu.w, M, dv_in, T->part);
        KPD_LAUNCH_CHECK();
        if (g.Wu.g || g.Wh.g) {
            hipLaunchKernelGGL(k_gvp_vec_reduce, dim3(32), dim3(256), 0, T->st, T->part, blocks, 512, 256, 256, g.Wu.g, g.Wh.g);
            KPD_LAUNCH_CHECK();
        }
        return KPD_OK;
    }
    // (The same one-row-per-thread fusion of this half -- dVh = dVu Wu^T + the norm term, dv_in = dVh Wh^T -- was built and measured:
    // 94.5 vs 93.4 ms per gvp_train step, slower: five strided row streams per thread instead of two.  A chained MFMA form is the way there.)
    KPD_TRY(gemm(T, false, true, 3 * M, g.h, g.vo, dV, g.vo, g.Wu.w, g.vo, 0.0f, T->dVh, g.h));
    tot = (long long)M * 3 * g.h;
    hipLaunchKernelGGL(k_gvp_sh_bwd, grid1(tot), dim3(256), 0, T->st, B.Vh, B.sh, T->dsh, tot, g.h, T->dVh);
    KPD_LAUNCH_CHECK();
    KPD_TRY(grad_gemm(T, g.h, g.vo, 3 * M, B.Vh, g.h, dV, g.vo, g.Wu.g, g.vo));
    KPD_TRY(grad_gemm(T, g.vi, g.h, 3 * M, v_in, g.vi, T->dVh, g.h, g.Wh.g, g.h));
    if (dv_in) KPD_TRY(gemm(T, false, true, 3 * M, g.vi, g.h, T->dVh, g.h, g.Wh.w, g.h, 0.0f, dv_in, g.vi));
    return KPD_OK;
}


struct LnP {
    Param gamma, beta;
};

template <class TT>
kpd_status ln_params(TT *T, const std::string &p, LnP *l) {
    KPD_TRY(param(T, p + ".feat_norm.weight", T->S, 1, &l->gamma));
    KPD_TRY(param(T, p + ".feat_norm.bias", T->S, 1, &l->beta));
    return KPD_OK;
}

template <class TT>
kpd_status gvp_ln_fwd(TT *T, const LnP &l, int n, const float *s, const float *v, float *so, float *vo) {
    hipLaunchKernelGGL(k_ln_fwd, dim3(cdiv(n, 4)), dim3(256), 0, T->st, s, l.gamma.w, l.beta.w, n, T->S, so);
    KPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_vnorm_fwd, grid1(n), dim3(256), 0, T->st, v, n, T->V, vo);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// backward of GVPLayerNorm at input (s, v): dso / dvo in, ds / dv out (may alias the inputs' gradient buffers)
template <class TT>
kpd_status gvp_ln_bwd(TT *T, const LnP &l, int n, const float *s, const float *v, const float *dso, const float *dvo,
                      float *ds, float *dv) {
    // (through tmp_s / tmp_v only when the outputs alias the incoming gradients: dso is read again for the bias gradient)
    float *os = ds == dso ? T->tmp_s : ds, *ov = dv;          // (the vector kernel works in place)
    const int blocks = cdiv(n, LN_GP_ROWS);
    if (T->S <= 256 && l.gamma.g && l.beta.g && T->colpart && blocks <= T->colpart_blocks) {
        os = ds;          // (in place when ds == dso: see the kernel)
        hipLaunchKernelGGL(k_ln_bwd_gp, dim3(blocks), dim3(256), 0, T->st, s, l.gamma.w, dso, n, T->S, os, T->colpart);
        KPD_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_colsum_reduce, dim3(cdiv(T->S, 64)), dim3(1024), 0, T->st, T->colpart, blocks, T->S, l.gamma.g, 1, l.beta.g);
        KPD_LAUNCH_CHECK();
    } else {
        hipLaunchKernelGGL(k_ln_bwd_g, dim3(cdiv(n, 4)), dim3(256), 0, T->st, s, l.gamma.w, dso, n, T->S, os, T->U);
        KPD_LAUNCH_CHECK();
        KPD_TRY(colsum_acc(T, n, T->S, T->U, T->S, l.gamma.g));        // U (free outside the edge passes) = dy * xhat
        KPD_TRY(colsum_acc(T, n, T->S, dso, T->S, l.beta.g));
    }
    if (os != ds) KPD_HIP(hipMemcpyAsync(ds, os, (size_t)n * T->S * 4, hipMemcpyDeviceToDevice, T->st));
    hipLaunchKernelGGL(k_vnorm_bwd, grid1((long long)n * 16), dim3(256), 0, T->st, v, dvo, n, T->V, ov);
    KPD_LAUNCH_CHECK();
    if (ov != dv) KPD_HIP(hipMemcpyAsync(dv, ov, (size_t)n * 3 * VC * 4, hipMemcpyDeviceToDevice, T->st));
    return KPD_OK;
}


}  // namespace
}  // namespace kpd
