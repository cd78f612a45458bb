// Training path of the GVP denoiser: forward with saved conv states + backward.
//
// Gradients of LigRecDynamicsGVP.forward (models/dynamics_gvp.py:149-199; LigRecGVP :46-101, GVPMultiEdgeConv
// models/gvp.py:459-551, GVP :89-116, GVPLayerNorm :159-166, NoisePredictionBlock dynamics_gvp.py:10-44) with respect to
// every parameter, to the scalar / vector input features and -- when asked for -- to the ligand and keypoint positions, which
// enter through the unit edge vector and the rbf code of every edge (gvp.py:472-480): learned keypoints make them a function of
// the receptor encoder's parameters (k_gvp_geom_bwd).
//
// Same formulation as egnn_train.hip: parameters in place in the reference layout, gradients accumulated in that
// layout, dense products through rocBLAS on the caller's stream, everything else in the kernels below; only node-sized
// state per conv is kept between forward and backward, edge activations are recomputed one edge type at a time.
// Vector features are kept as [rows, 3, channels] (the reference holds [rows, channels, 3]), so that the channel
// mixes Wh / Wu are plain GEMMs over 3 x rows; the first scalar Linear of the message function is split as in the
// inference path (per-node block U = s_src W[:, :S]^T gathered per edge, + rbf and vector-norm blocks per edge).
// Dropout: GVPDropout (gvp.py:119-149) acts on the aggregated messages and on the update residual in training mode
// (feature dropout per element, vector dropout per channel, both scaled by 1 / (1 - rate)).  Masks come from Philox keyed
// by a per-forward seed and the (conv, node type, position, kind) stream, so the backward pass regenerates them instead
// of storing them; kpd_dropout_mask exposes the same stream for tests.
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
// dvin [E, 3, 17]) and through the rbf code (drbf [E, 16]).  With diff = x_src - x_dst, q = |diff|^2, n = sqrt(max(q, 1e-8)),
// d = n + 1e-8:  unit = diff / d,  rbf_k = exp(-((d - mu_k) / sigma)^2);  where the clamp is active n does not depend on diff.
// dxe[e] = dL/d diff (added to the source node's gradient, subtracted from the destination's).
__global__ void k_gvp_geom_bwd(const int *__restrict__ src, const int *__restrict__ dst, const float *__restrict__ xs,
                               const float *__restrict__ xd, int E, float dmax, const float *__restrict__ rbf,
                               const float *__restrict__ drbf, const float *__restrict__ dvin, float *__restrict__ dxe) {
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
        du[c] = dvin[((size_t)e * 3 + c) * VH];
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

__global__ void k_gather_rows(const float *__restrict__ A, const int *__restrict__ idx, const float *__restrict__ scale, long long total,
                              int cols, float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
    const int v = idx[r];
    out[i] = A[(size_t)v * cols + c] * (scale ? scale[v] : 1.0f);
}

// acc[v] += scale[v] * sum over the edges of dst node v of M[e] (rows `cols` wide), one workgroup per dst node
__global__ void k_segsum(const float *__restrict__ M, int cols, const int *__restrict__ rowptr, const float *__restrict__ scale,
                         float *__restrict__ acc) {
    const int v = blockIdx.x;
    const int e0 = rowptr[v], e1 = rowptr[v + 1];
    if (e0 == e1) return;
    const float sc = scale[v];
    for (int c = threadIdx.x; c < cols; c += blockDim.x) {
        float s = 0.0f;
        for (int e = e0; e < e1; ++e) s += M[(size_t)e * cols + c];
        acc[(size_t)v * cols + c] += s * sc;
    }
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

// V[m, c, u] = act(gate[m, u]) * Vu[m, c, u], act = sigmoid or identity (gvp.py:108-114)
__global__ void k_gvp_gate(const float *__restrict__ gate, const float *__restrict__ Vu, long long total, int vo, int identity,
                           float *__restrict__ V) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int u = (int)(i % vo);
    const long long m = i / (3LL * vo);
    const float g = gate[m * vo + u];
    V[i] = (identity ? g : sigm(g)) * Vu[i];
}

// one thread per (m, u): dgate = sum_c dV Vu act'(gate); dV <- dV act(gate) (= dVu)
__global__ void k_gvp_gate_bwd(const float *__restrict__ gate, const float *__restrict__ Vu, long long total, int vo, int identity,
                               float *__restrict__ dV, float *__restrict__ dgate) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // over [m, u]
    if (i >= total) return;
    const long long m = i / vo;
    const int u = (int)(i - m * vo);
    const float g = gate[i];
    const float a = identity ? g : sigm(g);
    const float da = identity ? 1.0f : a * (1.0f - a);
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const size_t k = ((size_t)m * 3 + c) * vo + u;
        s = fmaf(dV[k], Vu[k], s);
        dV[k] *= a;
    }
    dgate[i] = s * da;
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

// vector part of GVPLayerNorm (gvp.py:162-165): v / (sqrt(mean_ch(max(|v_ch|^2, 1e-8)) + 1e-5) + 1e-5), one thread per row
__global__ void k_vnorm_fwd(const float *__restrict__ v, int rows, float *__restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float *p = v + (size_t)r * 3 * VC;
    float m = 0.0f;
#pragma unroll
    for (int ch = 0; ch < VC; ++ch) m += fmaxf(p[ch] * p[ch] + p[VC + ch] * p[VC + ch] + p[2 * VC + ch] * p[2 * VC + ch], 1e-8f);
    const float inv = 1.0f / (sqrtf(m / VC + 1e-5f) + 1e-5f);
    for (int k = 0; k < 3 * VC; ++k) out[(size_t)r * 3 * VC + k] = p[k] * inv;
}

__global__ void k_vnorm_bwd(const float *__restrict__ v, const float *__restrict__ dout, int rows, float *__restrict__ dv) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float *p = v + (size_t)r * 3 * VC, *d = dout + (size_t)r * 3 * VC;
    float m = 0.0f, dot = 0.0f;
    bool live[VC];
#pragma unroll
    for (int ch = 0; ch < VC; ++ch) {
        const float n2 = p[ch] * p[ch] + p[VC + ch] * p[VC + ch] + p[2 * VC + ch] * p[2 * VC + ch];
        live[ch] = n2 > 1e-8f;
        m += fmaxf(n2, 1e-8f);
        dot += d[ch] * p[ch] + d[VC + ch] * p[VC + ch] + d[2 * VC + ch] * p[2 * VC + ch];
    }
    const float root = sqrtf(m / VC + 1e-5f), vn = root + 1e-5f, inv = 1.0f / vn;
    // out = v / vn; dvn = -(dout . v) / vn^2; dvn/dv[ch, c] = v[ch, c] / (VC * root) where the clamp is inactive
    const float k = -dot * inv * inv / (VC * root);
#pragma unroll
    for (int ch = 0; ch < VC; ++ch)
#pragma unroll
        for (int c = 0; c < 3; ++c) dv[(size_t)r * 3 * VC + c * VC + ch] = d[c * VC + ch] * inv + (live[ch] ? k * p[c * VC + ch] : 0.0f);
}

__global__ void k_add(const float *__restrict__ a, const float *__restrict__ b, long long n, float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

__global__ void k_acc(float *__restrict__ a, const float *__restrict__ b, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += b[i];
}

// [rows, VC, 3] <-> [rows, 3, VC]
__global__ void k_v_transpose(const float *__restrict__ in, long long rows, int to_internal, float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * 3 * VC) return;
    const long long r = i / (3 * VC);
    const int k = (int)(i - r * 3 * VC);
    if (to_internal) {          // out[r][c][ch] = in[r][ch][c]
        const int c = k / VC, ch = k - c * VC;
        out[i] = in[r * 3 * VC + ch * 3 + c];
    } else {                    // out[r][ch][c] = in[r][c][ch]
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
__global__ void k_dropout(const float *__restrict__ in, long long rows, int inner, int cols, unsigned long long seed, unsigned stream,
                          float rate, float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * inner * cols) return;
    const int c = (int)(i % cols);
    const long long r = i / ((long long)inner * cols);
    out[i] = in[i] * dropout_scale(seed, stream, r * cols + c, rate);
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

}  // namespace
}  // namespace kpd

using namespace kpd;

struct kpd_gvp_trainer : TrainCtx {
    kpd_gvp_config cfg{};
    Arena ws;
    int S = 256;
    int cap_B = 0, cap_lig = 0, cap_kp = 0, cap_kk = 0, cap_maxlig = 0, cap_maxkp = 0, cap_ll = 0, cap_kl = 0, cap_R = 0;
    kpd_batch bt{};
    const float *t_dev = nullptr;
    bool have_forward = false;
    int n[2] = {0, 0}, E[4] = {0, 0, 0, 0};
    const int *e_src[4] = {nullptr, nullptr, nullptr, nullptr}, *e_dst[4] = {nullptr, nullptr, nullptr, nullptr},
              *e_rowptr[4] = {nullptr, nullptr, nullptr, nullptr};
    kpd_lig_graph lg{};
    int *meta = nullptr, *ll_deg = nullptr, *ll_off = nullptr, *kl_off = nullptr, *kl_pg = nullptr, *bidx[2] = {nullptr, nullptr};
    SrcCsr scsr[4];                     // edges of each type grouped by source node (deterministic sums over out-edges)
    int *cursor = nullptr;
    float *z[2] = {nullptr, nullptr};
    // saved node state: ss[nt][i], vs[nt][i] = input of conv i (i = n_convs: output); sa / va = pre-LayerNorm sums of conv i
    std::vector<float *> ss[2], vs[2], sa[2], va[2];
    float *enc_in[2] = {nullptr, nullptr}, *enc_pre[2] = {nullptr, nullptr}, *enc_act[2] = {nullptr, nullptr};
    // scratch (row capacity cap_R = max edge type / node count)
    GvpBuf gb[4];
    // Kept forward activations of the message chains (KPD_TRAIN_STORE, default on): geometry, rbf, message inputs and the seven
    // buffers of every message GVP per (conv, edge type), so that the backward pass does not run the chain a second time.
    // ~8.6 KB per edge and conv (22 GB at gvp_all_atom, B = 64); falls back to recomputation if the allocation fails.
    struct MsgSlot {
        float *unit = nullptr, *rbf = nullptr, *vin = nullptr;
        GvpBuf gb[4];
    };
    bool store = false;
    char *store_base = nullptr;
    std::vector<MsgSlot> slots;                    // [conv * 4 + et]
    MsgSlot scratch;
    float *ds[2] = {nullptr, nullptr}, *dV[2] = {nullptr, nullptr}, *dVh = nullptr, *dsh = nullptr, *dgate = nullptr;
    float *unit = nullptr, *rbf = nullptr, *vin = nullptr, *U = nullptr, *scale = nullptr, *tmp_s = nullptr, *tmp_v = nullptr,
          *s1 = nullptr, *v1 = nullptr, *sb = nullptr, *vb = nullptr;
    float *gs[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}, *gv[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // [cur/nxt][nt]
    float *wsg_pack = nullptr;
    float *dxe = nullptr, *gx[2] = {nullptr, nullptr};     // position gradients: per edge, per node type (lig, kp)
    bool want_x = false;
    float dropout = 0.0f;
    unsigned long long seed = 0;
};

namespace {

const char *kCanon[4] = {"lig_ll_lig", "kp_kl_lig", "lig_lk_kp", "kp_kk_kp"};
const char *kNtName[2] = {"lig", "kp"};
const int kSrc[4] = {NT_LIG, NT_KP, NT_LIG, NT_KP};
const int kDst[4] = {NT_LIG, NT_LIG, NT_KP, NT_KP};

kpd_status gvp_params(kpd_gvp_trainer *T, const std::string &p, int vi, int vo, int si, int so, GvpP *g) {
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
    static const bool on = !(getenv("KPD_TRAIN_WS") && atoi(getenv("KPD_TRAIN_WS")) == 0);
    return on && g.si == 256 && g.so == 256 && ld_s == 256;
}

// GVP.forward (gvp.py:89-116).  s_in == nullptr: B.pre already holds the contribution of the scalar inputs (no bias).
kpd_status gvp_fwd(kpd_gvp_trainer *T, const GvpP &g, int M, const float *s_in, int ld_s, const float *v_in, const GvpBuf &B,
                   bool identity) {
    if (M == 0) return KPD_OK;
    KPD_TRY(gemm(T, false, false, 3 * M, g.h, g.vi, v_in, g.vi, g.Wh.w, g.h, 0.0f, B.Vh, g.h));
    KPD_TRY(gemm(T, false, false, 3 * M, g.vo, g.h, B.Vh, g.h, g.Wu.w, g.vo, 0.0f, B.Vu, g.vo));
    hipLaunchKernelGGL(k_gvp_sh, grid1((long long)M * g.h), dim3(256), 0, T->st, B.Vh, (long long)M * g.h, g.h, B.sh);
    KPD_LAUNCH_CHECK();
    long long tot = (long long)M * g.so;
    if (s_in && ws_ok(g, ld_s)) {
        // the narrow vector-norm block first, then the 256 x 256 scalar block on the weight-stationary GEMM with the partial
        // pre-activation, the bias and the SiLU fused into its epilogue
        KPD_TRY(gemm(T, false, true, M, g.so, g.h, B.sh, g.h, g.Ws.w + g.si, g.si + g.h, 0.0f, B.pre, g.so));
        KPD_TRY(ws_gemm(WS_BIAS_SILU, s_in, M, ld_s, g.Ws.w, g.si + g.h, false, g.bs.w, nullptr, B.pre, B.s, g.so, T->wsg_pack, T->st, false, true));
    } else {
        if (s_in) KPD_TRY(gemm(T, false, true, M, g.so, g.si, s_in, ld_s, g.Ws.w, g.si + g.h, 0.0f, B.pre, g.so));
        KPD_TRY(gemm(T, false, true, M, g.so, g.h, B.sh, g.h, g.Ws.w + g.si, g.si + g.h, 1.0f, B.pre, g.so));
        hipLaunchKernelGGL(k_bias_silu, grid1(tot), dim3(256), 0, T->st, B.pre, g.bs.w, tot, g.so, g.so, B.s);
        KPD_LAUNCH_CHECK();
    }
    KPD_TRY(gemm(T, false, true, M, g.vo, g.so, B.s, g.so, g.Wg.w, g.so, 0.0f, B.gate, g.vo));
    tot = (long long)M * g.vo;
    hipLaunchKernelGGL(k_bias_add, grid1(tot), dim3(256), 0, T->st, B.gate, g.bg.w, tot, g.vo, g.vo);
    KPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_gvp_gate, grid1(3 * tot), dim3(256), 0, T->st, B.gate, B.Vu, 3 * tot, g.vo, identity ? 1 : 0, B.V);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// Backward of gvp_fwd.  ds [M, so] = dL/ds' (overwritten with dL/dpre), dV [M, 3, vo] = dL/dV' (overwritten with dL/dVu);
// ds_in [M, si] (ld so-independent: compact si) and dv_in [M, 3, vi] are written when non-null.
kpd_status gvp_bwd(kpd_gvp_trainer *T, const GvpP &g, int M, const float *s_in, int ld_s, const float *v_in, const GvpBuf &B,
                   bool identity, float *ds, float *dV, float *ds_in, float *dv_in) {
    if (M == 0) return KPD_OK;
    long long tot = (long long)M * g.vo;
    hipLaunchKernelGGL(k_gvp_gate_bwd, grid1(tot), dim3(256), 0, T->st, B.gate, B.Vu, tot, g.vo, identity ? 1 : 0, dV, T->dgate);
    KPD_LAUNCH_CHECK();
    KPD_TRY(colsum_acc(T, M, g.vo, T->dgate, g.vo, g.bg.g));
    KPD_TRY(grad_gemm(T, g.vo, g.so, M, T->dgate, g.vo, B.s, g.so, g.Wg.g, g.so));
    KPD_TRY(gemm(T, false, false, M, g.so, g.vo, T->dgate, g.vo, g.Wg.w, g.so, 1.0f, ds, g.so));
    tot = (long long)M * g.so;
    hipLaunchKernelGGL(k_silu_bwd, grid1(tot), dim3(256), 0, T->st, ds, B.pre, tot, g.so, g.so);
    KPD_LAUNCH_CHECK();
    KPD_TRY(colsum_acc(T, M, g.so, ds, g.so, g.bs.g));
    if (s_in) {
        if (g.Ws.g) KPD_TRY(grad_gemm(T, g.so, g.si, M, ds, g.so, s_in, ld_s, g.Ws.g, g.si + g.h));
        if (ds_in) {
            if (ws_ok(g, ld_s)) KPD_TRY(ws_gemm(WS_PLAIN, ds, M, g.so, g.Ws.w, g.si + g.h, true, nullptr, nullptr, ds_in, nullptr, g.si, T->wsg_pack, T->st, false, false));
            else KPD_TRY(gemm(T, false, false, M, g.si, g.so, ds, g.so, g.Ws.w, g.si + g.h, 0.0f, ds_in, g.si));
        }
    }
    if (g.Ws.g) KPD_TRY(grad_gemm(T, g.so, g.h, M, ds, g.so, B.sh, g.h, g.Ws.g + g.si, g.si + g.h));
    KPD_TRY(gemm(T, false, false, M, g.h, g.so, ds, g.so, g.Ws.w + g.si, g.si + g.h, 0.0f, T->dsh, g.h));
    KPD_TRY(gemm(T, false, true, 3 * M, g.h, g.vo, dV, g.vo, g.Wu.w, g.vo, 0.0f, T->dVh, g.h));
    tot = (long long)M * 3 * g.h;
    hipLaunchKernelGGL(k_gvp_sh_bwd, grid1(tot), dim3(256), 0, T->st, B.Vh, B.sh, T->dsh, tot, g.h, T->dVh);
    KPD_LAUNCH_CHECK();
    KPD_TRY(grad_gemm(T, g.h, g.vo, 3 * M, B.Vh, g.h, dV, g.vo, g.Wu.g, g.vo));
    KPD_TRY(grad_gemm(T, g.vi, g.h, 3 * M, v_in, g.vi, T->dVh, g.h, g.Wh.g, g.h));
    if (dv_in) KPD_TRY(gemm(T, false, true, 3 * M, g.vi, g.h, T->dVh, g.h, g.Wh.w, g.h, 0.0f, dv_in, g.vi));
    return KPD_OK;
}

// chain of n GVPs (S, 16) -> (S, 16) on M rows from (s0, v0): forward into gb[0..n)
kpd_status chain_fwd(kpd_gvp_trainer *T, const std::string &prefix, int n, int M, const float *s0, const float *v0) {
    for (int j = 0; j < n; ++j) {
        GvpP g;
        KPD_TRY(gvp_params(T, prefix + "." + std::to_string(j), VC, VC, T->S, T->S, &g));
        KPD_TRY(gvp_fwd(T, g, M, j == 0 ? s0 : T->gb[j - 1].s, T->S, j == 0 ? v0 : T->gb[j - 1].V, T->gb[j], false));
    }
    return KPD_OK;
}

// backward through that chain: in ds[0] / dV[0] (gradients of the last outputs), out ds[0] / dV[0] (gradients of s0, v0)
kpd_status chain_bwd(kpd_gvp_trainer *T, const std::string &prefix, int n, int M, const float *s0, const float *v0) {
    for (int j = n - 1; j >= 0; --j) {
        GvpP g;
        KPD_TRY(gvp_params(T, prefix + "." + std::to_string(j), VC, VC, T->S, T->S, &g));
        KPD_TRY(gvp_bwd(T, g, M, j == 0 ? s0 : T->gb[j - 1].s, T->S, j == 0 ? v0 : T->gb[j - 1].V, T->gb[j], false, T->ds[0], T->dV[0],
                        T->ds[1], T->dV[1]));
        std::swap(T->ds[0], T->ds[1]);
        std::swap(T->dV[0], T->dV[1]);
    }
    return KPD_OK;
}

unsigned drop_stream(int conv, int nt, int pos, int kind) { return (unsigned)(((conv * 2 + nt) * 2 + pos) * 2 + kind); }

// GVPDropout on (s [n, S], v [n, 3, 16]) -> (so, vo); identity copy when the rate is 0 (so / vo may alias the inputs)
kpd_status dropout_apply(kpd_gvp_trainer *T, int conv, int nt, int pos, int n, const float *s, const float *v, float *so, float *vo) {
    const int S = T->S;
    if (T->dropout <= 0.0f) {
        if (so != s) KPD_HIP(hipMemcpyAsync(so, s, (size_t)n * S * 4, hipMemcpyDeviceToDevice, T->st));
        if (vo != v) KPD_HIP(hipMemcpyAsync(vo, v, (size_t)n * 3 * VC * 4, hipMemcpyDeviceToDevice, T->st));
        return KPD_OK;
    }
    hipLaunchKernelGGL(k_dropout, grid1((long long)n * S), dim3(256), 0, T->st, s, (long long)n, 1, S, T->seed, drop_stream(conv, nt, pos, 0),
                       T->dropout, so);
    KPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_dropout, grid1((long long)n * 3 * VC), dim3(256), 0, T->st, v, (long long)n, 3, VC, T->seed,
                       drop_stream(conv, nt, pos, 1), T->dropout, vo);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

bool conv_uses(const kpd_gvp_trainer *T, int conv, int et) {
    if (et < 2) return true;
    return T->cfg.update_kp && conv < T->cfg.n_convs - 1;           // dynamics_gvp.py:67-72
}

kpd_status edge_scale(kpd_gvp_trainer *T, int et) {
    const int d = kDst[et];
    hipLaunchKernelGGL(k_msg_scale, grid1(T->n[d]), dim3(256), 0, T->st, T->e_rowptr[et], T->z[d], T->bidx[d], T->n[d],
                       T->cfg.message_norm_mode, T->cfg.message_norm, T->scale);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// message function of one edge type (gvp.py:540-551) forward into gb[0 .. n_message_gvps)
// point the message buffers at the kept activations of (conv, edge type); conv < 0: back to the scratch set (which the node-update
// chains use as well)
void bind_msg(kpd_gvp_trainer *T, int conv, int et) {
    const kpd_gvp_trainer::MsgSlot &sl = (T->store && conv >= 0) ? T->slots[(size_t)conv * 4 + et] : T->scratch;
    T->unit = sl.unit; T->rbf = sl.rbf; T->vin = sl.vin;
    for (int j = 0; j < 4; ++j)
        if (sl.gb[j].pre) T->gb[j] = sl.gb[j];
}

// recompute = false: the buffers already hold this chain's forward pass (kept activations); only the parameters are looked up
kpd_status message_fwd(kpd_gvp_trainer *T, int conv, int et, GvpP *g0_out, bool recompute = true) {
    if (!recompute) {
        const std::string prefix0 = "noise_predictor.conv_layers." + std::to_string(conv) + ".edge_message_fns." + kCanon[et];
        return gvp_params(T, prefix0 + ".0", VH, VC, T->S + RBF, T->S, g0_out);
    }
    const int E = T->E[et], s = kSrc[et], d = kDst[et], S = T->S, nm = T->cfg.n_message_gvps;
    const std::string prefix = "noise_predictor.conv_layers." + std::to_string(conv) + ".edge_message_fns." + kCanon[et];
    hipLaunchKernelGGL(k_gvp_geom, grid1(E), dim3(256), 0, T->st, T->e_src[et], T->e_dst[et], s == NT_LIG ? T->bt.lig_x : T->bt.kp_x,
                       d == NT_LIG ? T->bt.lig_x : T->bt.kp_x, E, 15.0f, T->unit, T->rbf);
    KPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_gvp_vin, grid1((long long)E * 3 * VH), dim3(256), 0, T->st, T->unit, T->vs[s][conv], T->e_src[et],
                       (long long)E * 3 * VH, T->vin);
    KPD_LAUNCH_CHECK();
    GvpP g0;
    KPD_TRY(gvp_params(T, prefix + ".0", VH, VC, S + RBF, S, &g0));
    // scalar part of the first Linear: U[src] + rbf W[:, S:S+16]^T
    KPD_TRY(gemm(T, false, true, T->n[s], S, S, T->ss[s][conv], S, g0.Ws.w, g0.si + g0.h, 0.0f, T->U, S));
    hipLaunchKernelGGL(k_gather_rows, grid1((long long)E * S), dim3(256), 0, T->st, T->U, T->e_src[et], (const float *)nullptr,
                       (long long)E * S, S, T->gb[0].pre);
    KPD_LAUNCH_CHECK();
    KPD_TRY(gemm(T, false, true, E, S, RBF, T->rbf, RBF, g0.Ws.w + S, g0.si + g0.h, 1.0f, T->gb[0].pre, S));
    KPD_TRY(gvp_fwd(T, g0, E, nullptr, 0, T->vin, T->gb[0], false));
    for (int j = 1; j < nm; ++j) {
        GvpP g;
        KPD_TRY(gvp_params(T, prefix + "." + std::to_string(j), VC, VC, S, S, &g));
        KPD_TRY(gvp_fwd(T, g, E, T->gb[j - 1].s, S, T->gb[j - 1].V, T->gb[j], false));
    }
    if (g0_out) *g0_out = g0;
    return KPD_OK;
}

struct LnP {
    Param gamma, beta;
};

kpd_status ln_params(kpd_gvp_trainer *T, const std::string &p, LnP *l) {
    KPD_TRY(param(T, p + ".feat_norm.weight", T->S, 1, &l->gamma));
    KPD_TRY(param(T, p + ".feat_norm.bias", T->S, 1, &l->beta));
    return KPD_OK;
}

kpd_status gvp_ln_fwd(kpd_gvp_trainer *T, const LnP &l, int n, const float *s, const float *v, float *so, float *vo) {
    hipLaunchKernelGGL(k_ln_fwd, dim3(cdiv(n, 4)), dim3(256), 0, T->st, s, l.gamma.w, l.beta.w, n, T->S, so);
    KPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_vnorm_fwd, grid1(n), dim3(256), 0, T->st, v, n, vo);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// backward of GVPLayerNorm at input (s, v): dso / dvo in, ds / dv out (may alias the inputs' gradient buffers)
kpd_status gvp_ln_bwd(kpd_gvp_trainer *T, const LnP &l, int n, const float *s, const float *v, const float *dso, const float *dvo,
                      float *ds, float *dv) {
    hipLaunchKernelGGL(k_ln_bwd_g, dim3(cdiv(n, 4)), dim3(256), 0, T->st, s, l.gamma.w, dso, n, T->S, T->tmp_s, T->U);
    KPD_LAUNCH_CHECK();
    KPD_TRY(colsum_acc(T, n, T->S, T->U, T->S, l.gamma.g));        // U (free outside the edge passes) = dy * xhat
    KPD_TRY(colsum_acc(T, n, T->S, dso, T->S, l.beta.g));
    KPD_HIP(hipMemcpyAsync(ds, T->tmp_s, (size_t)n * T->S * 4, hipMemcpyDeviceToDevice, T->st));
    hipLaunchKernelGGL(k_vnorm_bwd, grid1(n), dim3(256), 0, T->st, v, dvo, n, T->tmp_v);
    KPD_LAUNCH_CHECK();
    KPD_HIP(hipMemcpyAsync(dv, T->tmp_v, (size_t)n * 3 * VC * 4, hipMemcpyDeviceToDevice, T->st));
    return KPD_OK;
}

// one GVPMultiEdgeConv forward (gvp.py:459-538): ss/vs[conv] -> ss/vs[conv + 1]; keeps sa/va[conv] (pre-LayerNorm sums)
kpd_status conv_fwd(kpd_gvp_trainer *T, int conv) {
    const int S = T->S, nm = T->cfg.n_message_gvps, nu = T->cfg.n_update_gvps;
    const std::string cp = "noise_predictor.conv_layers." + std::to_string(conv);
    bool is_dst[2] = {false, false};
    for (int et = 0; et < 4; ++et)
        if (conv_uses(T, conv, et)) is_dst[kDst[et]] = true;
    for (int nt = 0; nt < 2; ++nt) {
        if (!is_dst[nt]) {          // untouched node type: the conv passes it through ({**node, **new})
            T->ss[nt][conv + 1] = T->ss[nt][conv];
            T->vs[nt][conv + 1] = T->vs[nt][conv];
            continue;
        }
        KPD_HIP(hipMemsetAsync(T->sa[nt][conv], 0, (size_t)T->n[nt] * S * 4, T->st));          // aggregated messages first ...
        KPD_HIP(hipMemsetAsync(T->va[nt][conv], 0, (size_t)T->n[nt] * 3 * VC * 4, T->st));
    }
    for (int et = 0; et < 4; ++et) {
        if (!conv_uses(T, conv, et) || T->E[et] == 0) continue;
        const int d = kDst[et];
        bind_msg(T, conv, et);
        KPD_TRY(message_fwd(T, conv, et, nullptr));
        KPD_TRY(edge_scale(T, et));
        hipLaunchKernelGGL(k_segsum, dim3(T->n[d]), dim3(256), 0, T->st, T->gb[nm - 1].s, S, T->e_rowptr[et], T->scale, T->sa[d][conv]);
        KPD_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_segsum, dim3(T->n[d]), dim3(64), 0, T->st, T->gb[nm - 1].V, 3 * VC, T->e_rowptr[et], T->scale, T->va[d][conv]);
        KPD_LAUNCH_CHECK();
    }
    bind_msg(T, -1, -1);
    for (int nt = 0; nt < 2; ++nt) {
        if (!is_dst[nt]) continue;
        const int n = T->n[nt];
        LnP l1, l2;
        KPD_TRY(ln_params(T, cp + ".message_layer_norms." + kNtName[nt], &l1));
        KPD_TRY(ln_params(T, cp + ".update_layer_norms." + kNtName[nt], &l2));
        // ... then dropout on them and the residual: sa = s + dropout(msg) (gvp.py:516-518)
        KPD_TRY(dropout_apply(T, conv, nt, 0, n, T->sa[nt][conv], T->va[nt][conv], T->sa[nt][conv], T->va[nt][conv]));
        hipLaunchKernelGGL(k_acc, grid1((long long)n * S), dim3(256), 0, T->st, T->sa[nt][conv], T->ss[nt][conv], (long long)n * S);
        KPD_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_acc, grid1((long long)n * 3 * VC), dim3(256), 0, T->st, T->va[nt][conv], T->vs[nt][conv], (long long)n * 3 * VC);
        KPD_LAUNCH_CHECK();
        KPD_TRY(gvp_ln_fwd(T, l1, n, T->sa[nt][conv], T->va[nt][conv], T->s1, T->v1));
        KPD_TRY(chain_fwd(T, cp + ".node_update_fns." + kNtName[nt], nu, n, T->s1, T->v1));
        KPD_TRY(dropout_apply(T, conv, nt, 1, n, T->gb[nu - 1].s, T->gb[nu - 1].V, T->tmp_s, T->tmp_v));
        hipLaunchKernelGGL(k_add, grid1((long long)n * S), dim3(256), 0, T->st, T->s1, T->tmp_s, (long long)n * S, T->sb);
        KPD_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_add, grid1((long long)n * 3 * VC), dim3(256), 0, T->st, T->v1, T->tmp_v, (long long)n * 3 * VC, T->vb);
        KPD_LAUNCH_CHECK();
        KPD_TRY(gvp_ln_fwd(T, l2, n, T->sb, T->vb, T->ss[nt][conv + 1], T->vs[nt][conv + 1]));
    }
    return KPD_OK;
}

// backward of conv: gs/gv[cur] = gradients of the conv outputs, gs/gv[nxt] = gradients of its inputs
kpd_status conv_bwd(kpd_gvp_trainer *T, int conv, int cur, int nxt) {
    const int S = T->S, nm = T->cfg.n_message_gvps, nu = T->cfg.n_update_gvps;
    const std::string cp = "noise_predictor.conv_layers." + std::to_string(conv);
    bool is_dst[2] = {false, false};
    for (int et = 0; et < 4; ++et)
        if (conv_uses(T, conv, et)) is_dst[kDst[et]] = true;
    for (int nt = 0; nt < 2; ++nt) {
        const int n = T->n[nt];
        if (!is_dst[nt]) {
            KPD_HIP(hipMemcpyAsync(T->gs[nxt][nt], T->gs[cur][nt], (size_t)n * S * 4, hipMemcpyDeviceToDevice, T->st));
            KPD_HIP(hipMemcpyAsync(T->gv[nxt][nt], T->gv[cur][nt], (size_t)n * 3 * VC * 4, hipMemcpyDeviceToDevice, T->st));
            continue;
        }
        LnP l1, l2;
        KPD_TRY(ln_params(T, cp + ".message_layer_norms." + kNtName[nt], &l1));
        KPD_TRY(ln_params(T, cp + ".update_layer_norms." + kNtName[nt], &l2));
        const std::string up = cp + ".node_update_fns." + kNtName[nt];
        // recompute s1, v1, the update chain and the second pre-norm sums
        KPD_TRY(gvp_ln_fwd(T, l1, n, T->sa[nt][conv], T->va[nt][conv], T->s1, T->v1));
        KPD_TRY(chain_fwd(T, up, nu, n, T->s1, T->v1));
        KPD_TRY(dropout_apply(T, conv, nt, 1, n, T->gb[nu - 1].s, T->gb[nu - 1].V, T->tmp_s, T->tmp_v));
        hipLaunchKernelGGL(k_add, grid1((long long)n * S), dim3(256), 0, T->st, T->s1, T->tmp_s, (long long)n * S, T->sb);
        KPD_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_add, grid1((long long)n * 3 * VC), dim3(256), 0, T->st, T->v1, T->tmp_v, (long long)n * 3 * VC, T->vb);
        KPD_LAUNCH_CHECK();
        // second GVPLayerNorm: d(sb, vb) -> ds[0] / dV[0]
        KPD_TRY(gvp_ln_bwd(T, l2, n, T->sb, T->vb, T->gs[cur][nt], T->gv[cur][nt], T->ds[0], T->dV[0]));
        // residual: d s1 += d sb, d v1 += d vb -> keep them in gs/gv[nxt] for now; the update chain sees them through its dropout mask
        KPD_HIP(hipMemcpyAsync(T->gs[nxt][nt], T->ds[0], (size_t)n * S * 4, hipMemcpyDeviceToDevice, T->st));
        KPD_HIP(hipMemcpyAsync(T->gv[nxt][nt], T->dV[0], (size_t)n * 3 * VC * 4, hipMemcpyDeviceToDevice, T->st));
        KPD_TRY(dropout_apply(T, conv, nt, 1, n, T->ds[0], T->dV[0], T->ds[0], T->dV[0]));
        KPD_TRY(chain_bwd(T, up, nu, n, T->s1, T->v1));
        hipLaunchKernelGGL(k_acc, grid1((long long)n * S), dim3(256), 0, T->st, T->gs[nxt][nt], T->ds[0], (long long)n * S);
        KPD_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_acc, grid1((long long)n * 3 * VC), dim3(256), 0, T->st, T->gv[nxt][nt], T->dV[0], (long long)n * 3 * VC);
        KPD_LAUNCH_CHECK();
        // first GVPLayerNorm at (sa, va): gradients of the pre-norm sums = gradients of the inputs (residual) and of the messages
        KPD_TRY(gvp_ln_bwd(T, l1, n, T->sa[nt][conv], T->va[nt][conv], T->gs[nxt][nt], T->gv[nxt][nt], T->gs[nxt][nt], T->gv[nxt][nt]));
        // the gradient of the aggregated messages is the same tensor: keep a copy where the edge passes can read it while
        // gs/gv[nxt] accumulate the source-side contributions
        KPD_TRY(dropout_apply(T, conv, nt, 0, n, T->gs[nxt][nt], T->gv[nxt][nt], T->gs[cur][nt], T->gv[cur][nt]));
    }
    for (int et = 0; et < 4; ++et) {
        if (!conv_uses(T, conv, et) || T->E[et] == 0) continue;
        const int E = T->E[et], s = kSrc[et], d = kDst[et];
        const std::string prefix = cp + ".edge_message_fns." + kCanon[et];
        GvpP g0;
        bind_msg(T, conv, et);
        KPD_TRY(message_fwd(T, conv, et, &g0, !T->store));
        KPD_TRY(edge_scale(T, et));
        // d(message of edge e) = scale[dst] * d(aggregate)[dst]
        hipLaunchKernelGGL(k_gather_rows, grid1((long long)E * S), dim3(256), 0, T->st, T->gs[cur][d], T->e_dst[et], T->scale,
                           (long long)E * S, S, T->ds[0]);
        KPD_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_gather_rows, grid1((long long)E * 3 * VC), dim3(256), 0, T->st, T->gv[cur][d], T->e_dst[et], T->scale,
                           (long long)E * 3 * VC, 3 * VC, T->dV[0]);
        KPD_LAUNCH_CHECK();
        for (int j = nm - 1; j >= 1; --j) {
            GvpP g;
            KPD_TRY(gvp_params(T, prefix + "." + std::to_string(j), VC, VC, S, S, &g));
            KPD_TRY(gvp_bwd(T, g, E, T->gb[j - 1].s, S, T->gb[j - 1].V, T->gb[j], false, T->ds[0], T->dV[0], T->ds[1], T->dV[1]));
            std::swap(T->ds[0], T->ds[1]);
            std::swap(T->dV[0], T->dV[1]);
        }
        KPD_TRY(gvp_bwd(T, g0, E, nullptr, 0, T->vin, T->gb[0], false, T->ds[0], T->dV[0], nullptr, T->dV[1]));
        if (T->want_x) {
            // positions (gvp.py:472-480): d rbf = dpre W[:, S:S+16] (dsh is free again), d unit = channel 0 of d vin
            const float *xs = s == NT_LIG ? T->bt.lig_x : T->bt.kp_x, *xd = d == NT_LIG ? T->bt.lig_x : T->bt.kp_x;
            KPD_TRY(gemm(T, false, false, E, RBF, S, T->ds[0], S, g0.Ws.w + S, g0.si + g0.h, 0.0f, T->dsh, RBF));
            hipLaunchKernelGGL(k_gvp_geom_bwd, grid1(E), dim3(256), 0, T->st, T->e_src[et], T->e_dst[et], xs, xd, E, 15.0f, T->rbf, T->dsh,
                               T->dV[1], T->dxe);
            KPD_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_seg3, grid1(T->n[s]), dim3(256), 0, T->st, T->dxe, T->scsr[et].perm, T->scsr[et].rowptr, T->n[s], 1.0f, T->gx[s]);
            KPD_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_seg3, grid1(T->n[d]), dim3(256), 0, T->st, T->dxe, (const int *)nullptr, T->e_rowptr[et], T->n[d], -1.0f, T->gx[d]);
            KPD_LAUNCH_CHECK();
        }
        // ds[0] = dL/dpre of the first GVP: its scalar inputs were U[src] and rbf
        if (g0.Ws.g) KPD_TRY(grad_gemm(T, S, RBF, E, T->ds[0], S, T->rbf, RBF, g0.Ws.g + S, g0.si + g0.h));
        // sums over the out-edges of every source node, in ascending edge order (no float atomics)
        hipLaunchKernelGGL(k_segsum_perm, dim3(T->n[s]), dim3(256), 0, T->st, T->ds[0], S, 0, S, T->scsr[et].perm, T->scsr[et].rowptr, 1.0f, 0,
                           T->U, S);
        KPD_LAUNCH_CHECK();
        if (g0.Ws.g) KPD_TRY(grad_gemm(T, S, S, T->n[s], T->U, S, T->ss[s][conv], S, g0.Ws.g, g0.si + g0.h));
        KPD_TRY(gemm(T, false, false, T->n[s], S, S, T->U, S, g0.Ws.w, g0.si + g0.h, 1.0f, T->gs[nxt][s], S));
        for (int cc = 0; cc < 3; ++cc) {      // vector rows [E, 3, 17], channels 1..16 -> gv[src, 3, 16]
            hipLaunchKernelGGL(k_segsum_perm, dim3(T->n[s]), dim3(64), 0, T->st, T->dV[1], 3 * VH, cc * VH + 1, VC, T->scsr[et].perm,
                               T->scsr[et].rowptr, 1.0f, 1, T->gv[nxt][s] + cc * VC, 3 * VC);
            KPD_LAUNCH_CHECK();
        }
    }
    bind_msg(T, -1, -1);
    return KPD_OK;
}

kpd_status encoder_fwd(kpd_gvp_trainer *T, int nt) {
    const int F = nt == 0 ? T->cfg.n_lig_scalars : T->cfg.n_kp_scalars, n = T->n[nt], S = T->S;
    const std::string p = nt == 0 ? "lig_encoder" : "kp_encoder";
    Param W, b;
    LnP l;
    KPD_TRY(param(T, p + ".0.weight", S, F + 1, &W));
    KPD_TRY(param(T, p + ".0.bias", S, 1, &b));
    KPD_TRY(param(T, p + ".2.weight", S, 1, &l.gamma));
    KPD_TRY(param(T, p + ".2.bias", S, 1, &l.beta));
    const long long tot = (long long)n * (F + 1);
    hipLaunchKernelGGL(k_cat_time, grid1(tot), dim3(256), 0, T->st, nt == 0 ? T->bt.lig_h : T->bt.kp_h, F, T->t_dev, T->bidx[nt], tot,
                       T->enc_in[nt]);
    KPD_LAUNCH_CHECK();
    KPD_TRY(gemm(T, false, true, n, S, F + 1, T->enc_in[nt], F + 1, W.w, F + 1, 0.0f, T->enc_pre[nt], S));
    hipLaunchKernelGGL(k_bias_silu, grid1((long long)n * S), dim3(256), 0, T->st, T->enc_pre[nt], b.w, (long long)n * S, S, S, T->enc_act[nt]);
    KPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ln_fwd, dim3(cdiv(n, 4)), dim3(256), 0, T->st, T->enc_act[nt], l.gamma.w, l.beta.w, n, S, T->ss[nt][0]);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// gs[cur][nt] = gradient of the encoder output; d_h (may be null) [n, F]
kpd_status encoder_bwd(kpd_gvp_trainer *T, int nt, int cur, float *d_h) {
    const int F = nt == 0 ? T->cfg.n_lig_scalars : T->cfg.n_kp_scalars, n = T->n[nt], S = T->S;
    const std::string p = nt == 0 ? "lig_encoder" : "kp_encoder";
    Param W, b;
    LnP l;
    KPD_TRY(param(T, p + ".0.weight", S, F + 1, &W));
    KPD_TRY(param(T, p + ".0.bias", S, 1, &b));
    KPD_TRY(param(T, p + ".2.weight", S, 1, &l.gamma));
    KPD_TRY(param(T, p + ".2.bias", S, 1, &l.beta));
    hipLaunchKernelGGL(k_ln_bwd_g, dim3(cdiv(n, 4)), dim3(256), 0, T->st, T->enc_act[nt], l.gamma.w, T->gs[cur][nt], n, S, T->tmp_s, T->sb);
    KPD_LAUNCH_CHECK();
    KPD_TRY(colsum_acc(T, n, S, T->sb, S, l.gamma.g));
    KPD_TRY(colsum_acc(T, n, S, T->gs[cur][nt], S, l.beta.g));
    hipLaunchKernelGGL(k_silu_bwd, grid1((long long)n * S), dim3(256), 0, T->st, T->tmp_s, T->enc_pre[nt], (long long)n * S, S, S);
    KPD_LAUNCH_CHECK();
    KPD_TRY(colsum_acc(T, n, S, T->tmp_s, S, b.g));
    if (W.g) KPD_TRY(grad_gemm(T, S, F + 1, n, T->tmp_s, S, T->enc_in[nt], F + 1, W.g, F + 1));
    if (d_h) {
        KPD_TRY(gemm(T, false, false, n, F + 1, S, T->tmp_s, S, W.w, F + 1, 0.0f, T->sb, F + 1));
        hipLaunchKernelGGL(k_copy_rows, grid1((long long)n * F), dim3(256), 0, T->st, T->sb, F + 1, d_h, F, (long long)n * F, F);
        KPD_LAUNCH_CHECK();
    }
    return KPD_OK;
}

const int kHeadS = 64;      // scalar width of the last noise GVP (dynamics_gvp.py:24-31)

kpd_status noise_fwd(kpd_gvp_trainer *T, float *eps_h, float *eps_x) {
    const int S = T->S, nn = T->cfg.n_noise_gvps, n = T->n[0], L = T->cfg.n_convs, F = T->cfg.n_lig_scalars;
    const std::string p = "noise_predictor.noise_predictor";
    for (int j = 0; j < nn; ++j) {
        const bool last = j == nn - 1;
        GvpP g;
        KPD_TRY(gvp_params(T, p + ".gvps." + std::to_string(j), VC, last ? 1 : VC, S, last ? kHeadS : S, &g));
        KPD_TRY(gvp_fwd(T, g, n, j == 0 ? T->ss[0][L] : T->gb[j - 1].s, S, j == 0 ? T->vs[0][L] : T->gb[j - 1].V, T->gb[j], last));
    }
    Param W, b;
    KPD_TRY(param(T, p + ".to_scalar_output.weight", F, kHeadS, &W));
    KPD_TRY(param(T, p + ".to_scalar_output.bias", F, 1, &b));
    if (eps_h) {
        KPD_TRY(gemm(T, false, true, n, F, kHeadS, T->gb[nn - 1].s, kHeadS, W.w, kHeadS, 0.0f, eps_h, F));
        hipLaunchKernelGGL(k_bias_add, grid1((long long)n * F), dim3(256), 0, T->st, eps_h, b.w, (long long)n * F, F, F);
        KPD_LAUNCH_CHECK();
        KPD_HIP(hipMemcpyAsync(eps_x, T->gb[nn - 1].V, (size_t)n * 12, hipMemcpyDeviceToDevice, T->st));     // [n, 3, 1]
    }
    return KPD_OK;
}

}  // namespace

extern "C" kpd_status kpd_gvp_trainer_create(const kpd_gvp_config *cfg, kpd_gvp_trainer **out) {
    KPD_REQUIRE(cfg && out, KPD_ERR_INVALID, "null argument");
    KPD_REQUIRE(cfg->vector_size == VC, KPD_ERR_INVALID, "vector_size=%d: only 16 is supported", cfg->vector_size);
    KPD_REQUIRE(cfg->n_hidden_scalars == 128 || cfg->n_hidden_scalars == 256, KPD_ERR_INVALID, "n_hidden_scalars=%d", cfg->n_hidden_scalars);
    KPD_REQUIRE(cfg->n_convs >= 1 && cfg->n_convs <= 64 && cfg->n_message_gvps >= 1 && cfg->n_message_gvps <= 4 && cfg->n_update_gvps >= 1 &&
                    cfg->n_update_gvps <= 4 && cfg->n_noise_gvps >= 1 && cfg->n_noise_gvps <= 4,
                KPD_ERR_INVALID, "n_convs=%d gvps=%d/%d/%d", cfg->n_convs, cfg->n_message_gvps, cfg->n_update_gvps, cfg->n_noise_gvps);
    KPD_REQUIRE(cfg->n_lig_scalars >= 1 && cfg->n_lig_scalars <= 255 && cfg->n_kp_scalars >= 1 && cfg->n_kp_scalars <= 255, KPD_ERR_INVALID,
                "scalar input widths %d / %d", cfg->n_lig_scalars, cfg->n_kp_scalars);
    KPD_REQUIRE(cfg->update_kp || cfg->n_convs == 1, KPD_ERR_INVALID, "update_kp=False with more than one conv cannot run in the reference");
    KPD_REQUIRE(cfg->message_norm_mode >= 0 && cfg->message_norm_mode <= 2 && (cfg->message_norm_mode != 0 || cfg->message_norm > 0.0f),
                KPD_ERR_INVALID, "message_norm mode %d value %g", cfg->message_norm_mode, (double)cfg->message_norm);
    KPD_REQUIRE(cfg->ll_k >= 0 && cfg->ll_k <= 16 && cfg->kl_k >= 0 && cfg->kl_k <= KL_KMAX, KPD_ERR_INVALID, "ll_k=%d kl_k=%d", cfg->ll_k,
                cfg->kl_k);
    kpd_gvp_trainer *T = new kpd_gvp_trainer();
    T->cfg = *cfg;
    T->S = cfg->n_hidden_scalars;
    if (rocblas_create_handle(&T->blas) != rocblas_status_success) {
        delete T;
        set_error("rocblas_create_handle failed");
        return KPD_ERR_HIP;
    }
    (void)rocblas_set_atomics_mode(T->blas, rocblas_atomics_not_allowed);      // bitwise-reproducible products
    *out = T;
    return KPD_OK;
}

extern "C" void kpd_gvp_trainer_destroy(kpd_gvp_trainer *T) {
    if (!T) return;
    if (T->blas) rocblas_destroy_handle(T->blas);
    T->ws.release();
    if (T->store_base) (void)hipFree(T->store_base);
    delete T;
}

extern "C" kpd_status kpd_gvp_trainer_bind(kpd_gvp_trainer *T, const char *name, const float *weight, float *grad, const int64_t *shape,
                                           int32_t ndim) {
    KPD_REQUIRE(T && name && shape && (ndim == 1 || ndim == 2), KPD_ERR_INVALID, "bad argument");
    Param p;
    p.w = weight;
    p.g = grad;
    p.rows = (int)shape[0];
    p.cols = ndim == 2 ? (int)shape[1] : 1;
    T->params[name] = p;
    return KPD_OK;
}

extern "C" kpd_status kpd_gvp_trainer_set_dropout(kpd_gvp_trainer *T, float rate, uint64_t seed) {
    KPD_REQUIRE(T, KPD_ERR_INVALID, "null trainer");
    KPD_REQUIRE(rate >= 0.0f && rate < 1.0f, KPD_ERR_INVALID, "dropout rate %g outside [0, 1)", (double)rate);
    T->have_forward = false;        // a pending forward was drawn with the previous masks: it can no longer be differentiated
    T->dropout = rate;
    T->seed = seed;
    return KPD_OK;
}

extern "C" kpd_status kpd_dropout_mask(uint64_t seed, int32_t conv, int32_t node_type, int32_t position, int32_t kind, int64_t n,
                                       float rate, float *out, void *stream) {
    KPD_REQUIRE(out && n >= 0 && rate >= 0.0f && rate < 1.0f && conv >= 0 && (node_type | 1) == 1 && (position | 1) == 1 && (kind | 1) == 1,
                KPD_ERR_INVALID, "bad argument");
    if (n == 0) return KPD_OK;
    hipLaunchKernelGGL(k_dropout_mask, grid1(n), dim3(256), 0, static_cast<hipStream_t>(stream), (long long)n, (unsigned long long)seed,
                       drop_stream(conv, node_type, position, kind), rate, out);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

extern "C" kpd_status kpd_gvp_trainer_reserve(kpd_gvp_trainer *T, int32_t max_B, int32_t max_n_lig, int32_t max_n_kp, int32_t max_n_kk,
                                              int32_t max_lig_pg, int32_t max_kp_pg) {
    KPD_REQUIRE(T, KPD_ERR_INVALID, "null trainer");
    KPD_REQUIRE(max_B >= 1 && max_n_lig >= 1 && max_n_kp >= 1 && max_n_kk >= 0 && max_lig_pg >= 1 && max_kp_pg >= 1, KPD_ERR_INVALID,
                "bad capacities");
    if (max_B <= T->cap_B && max_n_lig <= T->cap_lig && max_n_kp <= T->cap_kp && max_n_kk <= T->cap_kk && max_lig_pg <= T->cap_maxlig &&
        max_kp_pg <= T->cap_maxkp)
        return KPD_OK;
    const kpd_gvp_config &c = T->cfg;
    max_B = std::max(max_B, T->cap_B); max_n_lig = std::max(max_n_lig, T->cap_lig); max_n_kp = std::max(max_n_kp, T->cap_kp);
    max_n_kk = std::max(max_n_kk, T->cap_kk); max_lig_pg = std::max(max_lig_pg, T->cap_maxlig); max_kp_pg = std::max(max_kp_pg, T->cap_maxkp);
    const int cap_ll = std::max<long>((long)max_n_lig * std::min(max_lig_pg - 1, c.ll_k > 0 ? c.ll_k : 200), 1);
    const int cap_kl = std::max<long>((long)max_n_kp * (c.kl_k > 0 ? c.kl_k : std::min(max_lig_pg, 100)), 1);
    const int R = std::max(std::max(std::max(cap_ll, cap_kl), std::max<int>(max_n_kk, 1)), std::max(max_n_lig, max_n_kp));
    const int L = c.n_convs, S = T->S;
    const int nn[2] = {max_n_lig, max_n_kp};
    for (int nt = 0; nt < 2; ++nt) {
        T->ss[nt].assign(L + 1, nullptr); T->vs[nt].assign(L + 1, nullptr);
        T->sa[nt].assign(L, nullptr); T->va[nt].assign(L, nullptr);
    }
    T->ws.release();
    // two passes over the same list: size, then carve
    for (int pass = 0; pass < 2; ++pass) {
        size_t bytes = 0;
        auto F = [&](float *&p, size_t count) {
            if (pass == 0) bytes += (count * 4 + 255) & ~size_t(255);
            else p = T->ws.take<float>(count);
        };
        auto I = [&](int *&p, size_t count) {
            if (pass == 0) bytes += (count * 4 + 255) & ~size_t(255);
            else p = T->ws.take<int>(count);
        };
        for (int nt = 0; nt < 2; ++nt) {
            for (int l = 0; l <= L; ++l) { F(T->ss[nt][l], (size_t)nn[nt] * S); F(T->vs[nt][l], (size_t)nn[nt] * 3 * VC); }
            for (int l = 0; l < L; ++l) { F(T->sa[nt][l], (size_t)nn[nt] * S); F(T->va[nt][l], (size_t)nn[nt] * 3 * VC); }
            F(T->enc_in[nt], (size_t)nn[nt] * 256); F(T->enc_pre[nt], (size_t)nn[nt] * S); F(T->enc_act[nt], (size_t)nn[nt] * S);
            I(T->bidx[nt], nn[nt]); F(T->z[nt], max_B);
            for (int k = 0; k < 2; ++k) { F(T->gs[k][nt], (size_t)nn[nt] * S); F(T->gv[k][nt], (size_t)nn[nt] * 3 * VC); }
        }
        for (int k = 0; k < 4; ++k) {
            GvpBuf &b = T->gb[k];
            F(b.Vh, (size_t)R * 3 * VH); F(b.Vu, (size_t)R * 3 * VC); F(b.sh, (size_t)R * VH); F(b.pre, (size_t)R * S); F(b.s, (size_t)R * S);
            F(b.gate, (size_t)R * VC); F(b.V, (size_t)R * 3 * VC);
        }
        for (int k = 0; k < 2; ++k) { F(T->ds[k], (size_t)R * (S + RBF)); F(T->dV[k], (size_t)R * 3 * VH); }
        F(T->dVh, (size_t)R * 3 * VH); F(T->dsh, (size_t)R * VH); F(T->dgate, (size_t)R * VC);
        F(T->unit, (size_t)R * 3); F(T->rbf, (size_t)R * RBF); F(T->vin, (size_t)R * 3 * VH);
        F(T->dxe, (size_t)R * 3); F(T->gx[0], (size_t)max_n_lig * 3); F(T->gx[1], (size_t)max_n_kp * 3);
        const size_t N = std::max(max_n_lig, max_n_kp);
        F(T->U, N * S); F(T->scale, N); F(T->tmp_s, N * S); F(T->tmp_v, N * 3 * VC); F(T->s1, N * S); F(T->v1, N * 3 * VC);
        F(T->sb, N * std::max(S, 256)); F(T->vb, N * 3 * VC);
        F(T->part, (size_t)GRAD_SPLIT * 264 * 520);
        F(T->wsg_pack, (size_t)ws_gemm_pack_floats());
        F(T->ones, 8);
        const int cap_et[4] = {cap_ll, cap_kl, cap_kl, std::max<int>(max_n_kk, 1)};
        for (int et = 0; et < 4; ++et) { I(T->scsr[et].perm, cap_et[et]); I(T->scsr[et].rowptr, nn[kSrc[et]] + 1); }
        I(T->cursor, std::max(max_n_lig, max_n_kp));
        F(T->colpart, colpart_floats(R));
        I(T->meta, 32); I(T->ll_deg, max_n_lig); I(T->ll_off, max_B + 1); I(T->kl_off, max_B + 1); I(T->kl_pg, max_B + 2);
        kpd_lig_graph &g = T->lg;
        I(g.ll_src, cap_ll); I(g.ll_dst, cap_ll); I(g.ll_rowptr, max_n_lig + 1);
        I(g.kl_src, cap_kl); I(g.kl_dst, cap_kl); I(g.kl_rowptr, max_n_lig + 1);
        I(g.lk_src, cap_kl); I(g.lk_dst, cap_kl); I(g.lk_rowptr, max_n_kp + 1);
        I(g.ll_per_graph, max_B); I(g.counts, 8);
        if (pass == 0) KPD_TRY(T->ws.reserve(bytes + 4096));
    }
    KPD_REQUIRE(T->lg.counts != nullptr, KPD_ERR_HIP, "workspace arena too small (internal sizing error)");
    T->part_floats = (size_t)GRAD_SPLIT * 264 * 520;
    T->lg.cap_ll = cap_ll; T->lg.cap_kl = cap_kl;
    T->colpart_blocks = cdiv(R, HEAD_ROWS);
    T->scratch.unit = T->unit; T->scratch.rbf = T->rbf; T->scratch.vin = T->vin;
    for (int j = 0; j < 4; ++j) T->scratch.gb[j] = T->gb[j];
    {
        if (T->store_base) (void)hipFree(T->store_base);
        T->store_base = nullptr;
        T->store = false;
        static const bool want = !(getenv("KPD_TRAIN_STORE") && atoi(getenv("KPD_TRAIN_STORE")) == 0);
        const int cap_et[4] = {cap_ll, cap_kl, cap_kl, std::max<int>(max_n_kk, 1)};
        const int nm = c.n_message_gvps;
        auto al = [](size_t floats) { return (floats * 4 + 255) & ~size_t(255); };
        auto slot_bytes = [&](size_t E) {
            size_t b = al(E * 3) + al(E * RBF) + al(E * 3 * VH);
            for (int j = 0; j < nm; ++j) b += al(E * 3 * VH) + al(E * 3 * VC) + al(E * VH) + 2 * al(E * S) + al(E * VC) + al(E * 3 * VC);
            return b;
        };
        size_t total = 0;
        for (int conv = 0; conv < L; ++conv)
            for (int et = 0; et < 4; ++et)
                if (conv_uses(T, conv, et)) total += slot_bytes(cap_et[et]);
        if (want && nm <= 4 && hipMalloc(reinterpret_cast<void **>(&T->store_base), std::max<size_t>(total, 256)) == hipSuccess) {
            T->store = true;
            T->slots.assign((size_t)L * 4, kpd_gvp_trainer::MsgSlot());
            char *p = T->store_base;
            auto take = [&](size_t floats) { float *r = reinterpret_cast<float *>(p); p += al(floats); return r; };
            for (int conv = 0; conv < L; ++conv)
                for (int et = 0; et < 4; ++et) {
                    if (!conv_uses(T, conv, et)) continue;
                    const size_t E = cap_et[et];
                    kpd_gvp_trainer::MsgSlot &sl = T->slots[(size_t)conv * 4 + et];
                    sl.unit = take(E * 3); sl.rbf = take(E * RBF); sl.vin = take(E * 3 * VH);
                    for (int j = 0; j < nm; ++j) {
                        GvpBuf &b = sl.gb[j];
                        b.Vh = take(E * 3 * VH); b.Vu = take(E * 3 * VC); b.sh = take(E * VH); b.pre = take(E * S); b.s = take(E * S);
                        b.gate = take(E * VC); b.V = take(E * 3 * VC);
                    }
                }
        } else {
            (void)hipGetLastError();
            T->store_base = nullptr;
        }
    }
    T->cap_B = max_B; T->cap_lig = max_n_lig; T->cap_kp = max_n_kp; T->cap_kk = max_n_kk; T->cap_maxlig = max_lig_pg;
    T->cap_maxkp = max_kp_pg; T->cap_ll = cap_ll; T->cap_kl = cap_kl; T->cap_R = R;
    T->have_forward = false;
    return KPD_OK;
}

extern "C" kpd_status kpd_gvp_trainer_forward(kpd_gvp_trainer *T, const kpd_batch *bt, const float *t_dev, float *eps_h, float *eps_x,
                                              void *stream) {
    KPD_REQUIRE(T && bt && t_dev && eps_h && eps_x, KPD_ERR_INVALID, "null argument");
    KPD_REQUIRE(bt->B >= 1 && bt->n_lig >= 1 && bt->n_kp >= 1 && bt->kp_v, KPD_ERR_INVALID, "empty batch or missing keypoint vectors");
    KPD_REQUIRE(bt->B <= T->cap_B && bt->n_lig <= T->cap_lig && bt->n_kp <= T->cap_kp && bt->n_kk <= T->cap_kk &&
                    bt->max_lig <= T->cap_maxlig && bt->max_kp <= T->cap_maxkp,
                KPD_ERR_CAPACITY, "batch exceeds the reserved workspace");
    KPD_REQUIRE(bt->kk_rowptr && (bt->n_kk == 0 || (bt->kk_src && bt->kk_dst)), KPD_ERR_INVALID, "kk edges missing");
    const kpd_gvp_config &c = T->cfg;
    hipStream_t st = static_cast<hipStream_t>(stream);
    T->st = st;
    KPD_BLAS(rocblas_set_stream(T->blas, st));
    T->bt = *bt;
    T->t_dev = t_dev;
    T->n[0] = bt->n_lig; T->n[1] = bt->n_kp;
    KPD_TRY(launch_node_graph_index(bt->lig_ptr, bt->B, bt->n_lig, T->bidx[0], st));
    KPD_TRY(launch_node_graph_index(bt->kp_ptr, bt->B, bt->n_kp, T->bidx[1], st));
    KPD_TRY(launch_lig_graph(bt, c.ll_cutoff, c.ll_k, c.kl_cutoff, c.kl_k, &T->lg, T->ll_deg, T->ll_off, T->kl_off, T->kl_pg, st));
    KPD_TRY(launch_egnn_meta(T->lg.counts, bt->n_kk, 0xF, 0xF, bt->lig_ptr, bt->kp_ptr, T->lg.ll_per_graph, bt->kk_rowptr, bt->B, T->kl_off,
                             c.message_norm_mode == 2 ? 0.0f : 1.0f, 1, T->meta, T->z[0], T->z[1], st));
    int counts[2];
    KPD_HIP(hipMemcpyAsync(counts, T->lg.counts, sizeof(counts), hipMemcpyDeviceToHost, st));
    KPD_HIP(hipStreamSynchronize(st));
    KPD_REQUIRE(counts[0] <= T->cap_ll && counts[1] <= T->cap_kl, KPD_ERR_CAPACITY, "edge lists overflow");
    T->E[ET_LL] = counts[0]; T->E[ET_KL] = counts[1]; T->E[ET_LK] = counts[1]; T->E[ET_KK] = bt->n_kk;
    T->e_src[ET_LL] = T->lg.ll_src; T->e_dst[ET_LL] = T->lg.ll_dst; T->e_rowptr[ET_LL] = T->lg.ll_rowptr;
    T->e_src[ET_KL] = T->lg.kl_src; T->e_dst[ET_KL] = T->lg.kl_dst; T->e_rowptr[ET_KL] = T->lg.kl_rowptr;
    T->e_src[ET_LK] = T->lg.lk_src; T->e_dst[ET_LK] = T->lg.lk_dst; T->e_rowptr[ET_LK] = T->lg.lk_rowptr;
    T->e_src[ET_KK] = bt->kk_src; T->e_dst[ET_KK] = bt->kk_dst; T->e_rowptr[ET_KK] = bt->kk_rowptr;
    for (int et = 0; et < 4; ++et) KPD_TRY(build_src_csr(T, T->e_src[et], T->E[et], T->n[kSrc[et]], T->cursor, T->scsr[et]));
    // node state of conv 0: encoders; ligand vectors start at zero, keypoint vectors are v_0 (dynamics_gvp.py:179-189)
    KPD_TRY(encoder_fwd(T, 0));
    KPD_TRY(encoder_fwd(T, 1));
    KPD_HIP(hipMemsetAsync(T->vs[0][0], 0, (size_t)bt->n_lig * 3 * VC * 4, st));
    hipLaunchKernelGGL(k_v_transpose, grid1((long long)bt->n_kp * 3 * VC), dim3(256), 0, st, bt->kp_v, (long long)bt->n_kp, 1, T->vs[1][0]);
    KPD_LAUNCH_CHECK();
    for (int i = 0; i < c.n_convs; ++i) KPD_TRY(conv_fwd(T, i));
    KPD_TRY(noise_fwd(T, eps_h, eps_x));
    T->have_forward = true;
    return KPD_OK;
}

extern "C" kpd_status kpd_gvp_trainer_backward(kpd_gvp_trainer *T, const float *d_eps_h, const float *d_eps_x, float *d_lig_h,
                                               float *d_kp_h, float *d_kp_v, float *d_lig_x, float *d_kp_x, void *stream) {
    KPD_REQUIRE(T && d_eps_h && d_eps_x, KPD_ERR_INVALID, "null argument");
    KPD_REQUIRE(T->have_forward, KPD_ERR_STATE, "kpd_gvp_trainer_backward before kpd_gvp_trainer_forward");
    const kpd_gvp_config &c = T->cfg;
    hipStream_t st = static_cast<hipStream_t>(stream);
    T->st = st;
    KPD_BLAS(rocblas_set_stream(T->blas, st));
    const int S = T->S, nn = c.n_noise_gvps, nl = T->n[0], nk = T->n[1], L = c.n_convs, F = c.n_lig_scalars;
    int cur = 0, nxt = 1;
    T->want_x = d_lig_x || d_kp_x;
    if (T->want_x)
        for (int nt = 0; nt < 2; ++nt) KPD_HIP(hipMemsetAsync(T->gx[nt], 0, (size_t)T->n[nt] * 12, st));
    for (int k = 0; k < 2; ++k)
        for (int nt = 0; nt < 2; ++nt) {
            KPD_HIP(hipMemsetAsync(T->gs[k][nt], 0, (size_t)T->n[nt] * S * 4, st));
            KPD_HIP(hipMemsetAsync(T->gv[k][nt], 0, (size_t)T->n[nt] * 3 * VC * 4, st));
        }
    // noise block (recomputed: the conv recomputation reuses its buffers)
    KPD_TRY(noise_fwd(T, nullptr, nullptr));
    {
        const std::string p = "noise_predictor.noise_predictor";
        Param W, b;
        KPD_TRY(param(T, p + ".to_scalar_output.weight", F, kHeadS, &W));
        KPD_TRY(param(T, p + ".to_scalar_output.bias", F, 1, &b));
        KPD_TRY(colsum_acc(T, nl, F, d_eps_h, F, b.g));
        if (W.g) KPD_TRY(grad_gemm(T, F, kHeadS, nl, d_eps_h, F, T->gb[nn - 1].s, kHeadS, W.g, kHeadS));
        KPD_TRY(gemm(T, false, false, nl, kHeadS, F, d_eps_h, F, W.w, kHeadS, 0.0f, T->ds[0], kHeadS));
        KPD_HIP(hipMemcpyAsync(T->dV[0], d_eps_x, (size_t)nl * 12, hipMemcpyDeviceToDevice, st));
        for (int j = nn - 1; j >= 0; --j) {
            const bool last = j == nn - 1;
            GvpP g;
            KPD_TRY(gvp_params(T, p + ".gvps." + std::to_string(j), VC, last ? 1 : VC, S, last ? kHeadS : S, &g));
            KPD_TRY(gvp_bwd(T, g, nl, j == 0 ? T->ss[0][L] : T->gb[j - 1].s, S, j == 0 ? T->vs[0][L] : T->gb[j - 1].V, T->gb[j], last,
                            T->ds[0], T->dV[0], T->ds[1], T->dV[1]));
            std::swap(T->ds[0], T->ds[1]);
            std::swap(T->dV[0], T->dV[1]);
        }
        KPD_HIP(hipMemcpyAsync(T->gs[cur][0], T->ds[0], (size_t)nl * S * 4, hipMemcpyDeviceToDevice, st));
        KPD_HIP(hipMemcpyAsync(T->gv[cur][0], T->dV[0], (size_t)nl * 3 * VC * 4, hipMemcpyDeviceToDevice, st));
    }
    for (int i = L - 1; i >= 0; --i) {
        KPD_TRY(conv_bwd(T, i, cur, nxt));
        std::swap(cur, nxt);
    }
    KPD_TRY(encoder_bwd(T, 0, cur, d_lig_h));
    KPD_TRY(encoder_bwd(T, 1, cur, d_kp_h));
    if (d_kp_v) {
        hipLaunchKernelGGL(k_v_transpose, grid1((long long)nk * 3 * VC), dim3(256), 0, st, T->gv[cur][1], (long long)nk, 0, d_kp_v);
        KPD_LAUNCH_CHECK();
    }
    if (d_lig_x) KPD_HIP(hipMemcpyAsync(d_lig_x, T->gx[0], (size_t)nl * 12, hipMemcpyDeviceToDevice, st));
    if (d_kp_x) KPD_HIP(hipMemcpyAsync(d_kp_x, T->gx[1], (size_t)nk * 12, hipMemcpyDeviceToDevice, st));
    T->have_forward = false;
    return KPD_OK;
}
