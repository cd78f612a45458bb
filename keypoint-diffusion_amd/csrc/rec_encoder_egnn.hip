// EGNN keypoint receptor encoder behind the kpd_recegnn_* C ABI (include/kpd.h).  Replaces ReceptorEncoder.forward
// (models/receptor_encoder.py:483-555) as a whole: the ReceptorConv stack on the rr graph (:14-154), the keypoint
// embedding of the mean receptor feature (:526-530), RecKeyConv (:182-236: attention-pooled keypoint positions, kNN
// rec->kp edges, mean neighbour feature + the k distances -> kp_feature_mlp) and the keypoint radius graph (:541).
// Runs once per pocket on widths <= 256 (128 in every shipped config), so the kernels are the simple kind: one thread
// per output unit, activations through LDS, weights stored transposed so that a wave reads them coalesced.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "egnn_kernels.h"
#include "mfma_core.h"
#include "rec_kernels.h"

using namespace kpd;

namespace kpd {

constexpr int RW = 256;          // widest layer / threads per workgroup

// sums two values over the workgroup (256 threads); every thread gets the totals
__device__ __forceinline__ void block_sum2(float &a, float &b, float (*s_red)[4], int tid) {
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
    }
    __syncthreads();                          // s_red may still be read from the previous call
    if (lane == 0) {
        s_red[0][wave] = a;
        s_red[1][wave] = b;
    }
    __syncthreads();
    a = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
    b = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
}

// LayerNorm of one row held one element per thread (tid < n valid)
__device__ __forceinline__ float block_layernorm(float y, bool on, int n, const float *__restrict__ lw, const float *__restrict__ lb,
                                                 float (*s_red)[4], int tid) {
    float s = on ? y : 0.0f, dummy = 0.0f;
    block_sum2(s, dummy, s_red, tid);
    const float mean = s / (float)n;
    float d = on ? y - mean : 0.0f, v = d * d;
    dummy = 0.0f;
    block_sum2(v, dummy, s_red, tid);
    const float rstd = 1.0f / sqrtf(v / (float)n + 1e-5f);
    return on ? d * rstd * lw[tid] + lb[tid] : 0.0f;
}

// Per-node blocks of the first Linear of edge_mlp / coord_mlp (they are linear in [h_src, h_dst, radial, a]):
// P[node] = {We[:, :in] h, We[:, in:2in] h + be, Wc[:, :in] h, Wc[:, in:2in] h + bc}, each `hid` wide.
struct RcProjArgs {
    const float *h;            // [n][in]
    int n, in, hid;
    const float *We_t, *Wc_t;  // transposed first Linear weights [2 in + 1 + ef][hid]; Wc_t null with fix_pos
    const float *be, *bc;
    float *P;                  // [n][4][hid]
};

__global__ __launch_bounds__(RW) void k_rc_proj(RcProjArgs a) {
    __shared__ float s_in[4][RW];
    const int node0 = blockIdx.x * 4, tid = threadIdx.x;
    for (int i = tid; i < 4 * a.in; i += RW) {
        const int j = i / a.in, k = i - j * a.in;
        s_in[j][k] = node0 + j < a.n ? a.h[(size_t)(node0 + j) * a.in + k] : 0.0f;
    }
    __syncthreads();
    if (tid >= a.hid) return;
#pragma unroll 1
    for (int part = 0; part < 4; ++part) {
        const float *Wt = part < 2 ? a.We_t : a.Wc_t;
        if (!Wt) continue;
        const float *w = Wt + (size_t)((part & 1) * a.in) * a.hid + tid;
        const float b = (part & 1) ? (part < 2 ? a.be[tid] : a.bc[tid]) : 0.0f;
        float y[4] = {b, b, b, b};
        for (int k = 0; k < a.in; ++k) {
            const float wv = w[(size_t)k * a.hid];
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = fmaf(wv, s_in[j][k], y[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (node0 + j < a.n) a.P[((size_t)(node0 + j) * 4 + part) * a.hid + tid] = y[j];
    }
}

// One workgroup per destination node: messages of its in-edges (ReceptorConv.message :68-96), their sums
// (:144-147), node_mlp, LayerNorm and the coordinate update (:149-152).  Edge order = CSR order: deterministic.
struct RcConvArgs {
    int n, in, hid, out, ef;
    const float *h, *x;        // [n][in], [n][3] layer input
    const float *P;            // [n][4][hid]
    const int *src, *rowptr;   // rr edges sorted by (dst, src)
    const float *same_res;     // [E] or null
    const int *bidx;
    const float *z;            // per-graph normaliser (message_norm == 0) or null
    float norm_const;
    const float *We_t, *Wc_t;  // rows 2 in (radial) and 2 in + 1 (same_res) of the transposed first Linears
    const float *W2_t, *b2;    // edge_mlp.2 transposed [hid][hid]
    const float *watt, *batt;  // soft_attention [hid], [1]
    const float *w3;           // coord_mlp.2 [hid]
    const float *Wn1_t, *bn1;  // node_mlp.0 transposed [in + hid][hid]
    const float *Wn2_t, *bn2;  // node_mlp.2 transposed [hid][out]
    const float *ln_w, *ln_b;  // null without norm
    int use_tanh, fix_pos;
    float coords_range;
    float *h_out, *x_out;      // [n][out], [n][3]
};

__global__ __launch_bounds__(RW) void k_rc_conv(RcConvArgs a) {
    __shared__ float s_f[2 * RW];
    __shared__ float s_red[2][4];
    const int v = blockIdx.x, tid = threadIdx.x;
    const bool on = tid < a.hid;
    const int hid = a.hid;
    const float *Pv = a.P + (size_t)v * 4 * hid;
    const float ped = on ? Pv[hid + tid] : 0.0f, pcd = on && !a.fix_pos ? Pv[3 * hid + tid] : 0.0f;
    const float wre = on ? a.We_t[(size_t)(2 * a.in) * hid + tid] : 0.0f;
    const float wae = on && a.ef ? a.We_t[(size_t)(2 * a.in + 1) * hid + tid] : 0.0f;
    const float wrc = on && !a.fix_pos ? a.Wc_t[(size_t)(2 * a.in) * hid + tid] : 0.0f;
    const float wac = on && !a.fix_pos && a.ef ? a.Wc_t[(size_t)(2 * a.in + 1) * hid + tid] : 0.0f;
    const float b2 = on ? a.b2[tid] : 0.0f, watt = on ? a.watt[tid] : 0.0f, w3 = on && !a.fix_pos ? a.w3[tid] : 0.0f;
    const float batt = a.batt[0];
    const float xv0 = a.x[(size_t)v * 3], xv1 = a.x[(size_t)v * 3 + 1], xv2 = a.x[(size_t)v * 3 + 2];
    float acc_h = 0.0f, ax = 0.0f, ay = 0.0f, az = 0.0f;
    for (int e = a.rowptr[v]; e < a.rowptr[v + 1]; ++e) {
        const int u = a.src[e];
        const float dx = a.x[(size_t)u * 3] - xv0, dy = a.x[(size_t)u * 3 + 1] - xv1, dz = a.x[(size_t)u * 3 + 2] - xv2;   // :137
        const float d = sqrtf(dx * dx + dy * dy + dz * dz);                                                           // :138
        const float sr = a.ef ? a.same_res[e] : 0.0f;
        const float *Pu = a.P + (size_t)u * 4 * hid;
        float cpart = 0.0f;
        if (on) {
            s_f[tid] = silu(Pu[tid] + ped + d * wre + sr * wae);                       // edge_mlp.0 + SiLU
            if (!a.fix_pos) cpart = silu(Pu[2 * hid + tid] + pcd + d * wrc + sr * wac) * w3;   // coord_mlp.0 + SiLU, then .2
        }
        __syncthreads();
        float m = 0.0f;
        if (on) {
            m = b2;
            for (int k = 0; k < hid; ++k) m = fmaf(a.W2_t[(size_t)k * hid + tid], s_f[k], m);
            m = silu(m);                                                               // edge_mlp.2 + SiLU
        }
        float apart = m * watt;
        block_sum2(apart, cpart, s_red, tid);
        acc_h = fmaf(m, sigmoidf_(apart + batt), acc_h);                               // :85-86
        if (!a.fix_pos) {
            const float c = a.use_tanh ? tanhf(cpart) * a.coords_range : cpart;        // :89-92
            const float inv = c / (d + 1.0f);                                          // x_diff / (radial + 1), :140-142
            ax = fmaf(inv, dx, ax);
            ay = fmaf(inv, dy, ay);
            az = fmaf(inv, dz, az);
        }
        __syncthreads();                                                               // s_f is rewritten by the next edge
    }
    const float zinv = 1.0f / (a.z ? a.z[a.bidx[v]] : a.norm_const);
    // node_mlp([h, h_neigh / z]) (:149)
    if (tid < a.in) s_f[tid] = a.h[(size_t)v * a.in + tid];
    if (on) s_f[RW + tid] = acc_h * zinv;
    __syncthreads();
    float n1 = 0.0f;
    if (on) {
        n1 = a.bn1[tid];
        for (int k = 0; k < a.in; ++k) n1 = fmaf(a.Wn1_t[(size_t)k * hid + tid], s_f[k], n1);
        for (int k = 0; k < hid; ++k) n1 = fmaf(a.Wn1_t[(size_t)(a.in + k) * hid + tid], s_f[RW + k], n1);
        n1 = silu(n1);
    }
    __syncthreads();
    if (on) s_f[tid] = n1;
    __syncthreads();
    const bool oon = tid < a.out;
    float y = 0.0f;
    if (oon) {
        y = a.bn2[tid];
        for (int k = 0; k < hid; ++k) y = fmaf(a.Wn2_t[(size_t)k * a.out + tid], s_f[k], y);
    }
    if (a.ln_w) y = block_layernorm(y, oon, a.out, a.ln_w, a.ln_b, s_red, tid);        // :152
    if (oon) a.h_out[(size_t)v * a.out + tid] = y;
    if (tid == 0) {                                                                    // :150
        a.x_out[(size_t)v * 3] = xv0 + ax * zinv;
        a.x_out[(size_t)v * 3 + 1] = xv1 + ay * zinv;
        a.x_out[(size_t)v * 3 + 2] = xv2 + az * zinv;
    }
}

// message_norm == 0: z[b] = rr edges of the graph / receptor nodes of the graph (no +1, :505-509)
__global__ void k_rc_z(const int *__restrict__ rowptr, const int *__restrict__ ptr, int B, float *__restrict__ z) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) z[b] = (float)(rowptr[ptr[b + 1]] - rowptr[ptr[b]]) / (float)(ptr[b + 1] - ptr[b]);
}

// keypoint_embedding (:529-530): out[b][j] = SiLU(W[j] . mean[b] + bias[j]), j < D K; W row-major [D K][D]
__global__ __launch_bounds__(RW) void k_rc_kp_embed(const float *__restrict__ mean, const float *__restrict__ W,
                                                    const float *__restrict__ bias, int D, int DK, float *__restrict__ out) {
    __shared__ float s_m[RW];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < D) s_m[tid] = mean[(size_t)b * D + tid];
    __syncthreads();
    const int j = blockIdx.y * RW + tid;
    if (j >= DK) return;
    float acc = bias[j];
    const float *w = W + (size_t)j * D;
    for (int k = 0; k < D; ++k) acc = fmaf(w[k], s_m[k], acc);
    out[(size_t)b * DK + j] = silu(acc);
}

// k_closest_feats + kp_feature_mlp + LayerNorm (:257-291, :231-236): one workgroup per keypoint; its k rec->kp edges are
// rk_src[kp * k .. +k) (nearest first).
__global__ __launch_bounds__(RW) void k_rc_kp_feat(const float *__restrict__ h, const float *__restrict__ x0,
                                                   const float *__restrict__ kp_x, const int *__restrict__ rk_src, int k, int D,
                                                   const float *__restrict__ W_t, const float *__restrict__ bias,
                                                   const float *__restrict__ ln_w, const float *__restrict__ ln_b,
                                                   float *__restrict__ kp_h) {
    __shared__ float s_f[RW + 32];
    __shared__ float s_red[2][4];
    const int kp = blockIdx.x, tid = threadIdx.x;
    const bool on = tid < D;
    if (on) {
        float acc = 0.0f;
        for (int i = 0; i < k; ++i) acc += h[(size_t)rk_src[kp * k + i] * D + tid];
        s_f[tid] = acc / (float)k;                                                     // fn.mean over the k edges, :284
    }
    if (tid < k) {
        const int r = rk_src[kp * k + tid];
        const float dx = x0[(size_t)r * 3] - kp_x[(size_t)kp * 3] + 1e-30f, dy = x0[(size_t)r * 3 + 1] - kp_x[(size_t)kp * 3 + 1] + 1e-30f,
                    dz = x0[(size_t)r * 3 + 2] - kp_x[(size_t)kp * 3 + 2] + 1e-30f;
        s_f[D + tid] = sqrtf(dx * dx + dy * dy + dz * dz);                             // :285-286
    }
    __syncthreads();
    float y = 0.0f;
    if (on) {
        y = bias[tid];
        for (int i = 0; i < D + k; ++i) y = fmaf(W_t[(size_t)i * D + tid], s_f[i], y);
        y = silu(y);
    }
    if (ln_w) y = block_layernorm(y, on, D, ln_w, ln_b, s_red, tid);
    if (on) kp_h[(size_t)kp * D + tid] = y;
}

// kp_rad_feats + kp_feature_mlp + LayerNorm (:238-262, :231-236): one workgroup per keypoint; its rec->kp radius edges are
// rk_src[rowptr[kp] .. rowptr[kp + 1]) (index order, at most 100), z = rk edges of the complex / keypoints of the complex + 1.
__global__ __launch_bounds__(RW) void k_rc_kp_radfeat(const float *__restrict__ h, const int *__restrict__ rk_src,
                                                      const int *__restrict__ rk_rowptr, const int *__restrict__ rk_off, int K, int D,
                                                      const float *__restrict__ W_t, const float *__restrict__ bias,
                                                      const float *__restrict__ ln_w, const float *__restrict__ ln_b,
                                                      float *__restrict__ kp_h) {
    __shared__ float s_f[RW];
    __shared__ float s_red[2][4];
    const int kp = blockIdx.x, tid = threadIdx.x, b = kp / K;
    const bool on = tid < D;
    if (on) {
        float acc = 0.0f;
        for (int e = rk_rowptr[kp]; e < rk_rowptr[kp + 1]; ++e) acc += h[(size_t)rk_src[e] * D + tid];      // fn.sum, :258
        const float z = (float)(rk_off[b + 1] - rk_off[b]) / (float)K + 1.0f;                                // :259-260
        s_f[tid] = acc / z;
    }
    __syncthreads();
    float y = 0.0f;
    if (on) {
        y = bias[tid];
        for (int i = 0; i < D; ++i) y = fmaf(W_t[(size_t)i * D + tid], s_f[i], y);
        y = silu(y);
    }
    if (ln_w) y = block_layernorm(y, on, D, ln_w, ln_b, s_red, tid);
    if (on) kp_h[(size_t)kp * D + tid] = y;
}

}  // namespace kpd

// ---- engine -----------------------------------------------------------------------------------------
struct ConvW {
    int in = 0, out = 0;
    float *We_t = nullptr, *be = nullptr, *W2_t = nullptr, *b2 = nullptr, *watt = nullptr, *batt = nullptr;
    float *Wc_t = nullptr, *bc = nullptr, *w3 = nullptr;
    float *Wn1_t = nullptr, *bn1 = nullptr, *Wn2_t = nullptr, *bn2 = nullptr, *ln_w = nullptr, *ln_b = nullptr;
};

struct kpd_recegnn {
    kpd_recegnn_config cfg;
    Arena warena, ws;
    std::vector<ConvW> conv;
    float *kpe_W = nullptr, *kpe_b = nullptr, *fc_src_t = nullptr, *kpf_W_t = nullptr, *kpf_b = nullptr, *kp_lw = nullptr, *kp_lb = nullptr;
    std::set<std::string> expected, loaded;
    bool committed = false;
    int cap_B = 0, cap_rec = 0, cap_rr = 0, cap_maxrec = 0;
    float *h[2], *x[2], *P, *z, *gmean, *kp_h0, *ft_src, *ft_dst;
    int *bidx, *kp_ptr, *off_tmp, *deg_tmp, *rad_tmp, *xm_src, *xm_dst, *xm_rowptr, *rk_rowptr, *kk_rowptr, *kk_off;
};

extern "C" kpd_status kpd_recegnn_create(const kpd_recegnn_config *cfg, kpd_recegnn **out) {
    KPD_REQUIRE(cfg && out, KPD_ERR_INVALID, "null argument");
    const int in = cfg->in_n_node_feat, hid = cfg->hidden_n_node_feat, D = cfg->out_n_node_feat, K = cfg->n_keypoints;
    KPD_REQUIRE(in >= 1 && in <= RW && hid >= 1 && hid <= RW && D >= 1 && D <= RW, KPD_ERR_INVALID,
                "feature widths (%d, %d, %d) must be in 1..%d", in, hid, D, RW);
    KPD_REQUIRE(cfg->n_convs >= 1 && cfg->n_convs <= 32, KPD_ERR_INVALID, "n_convs=%d", cfg->n_convs);
    KPD_REQUIRE((cfg->k_closest >= 1 && cfg->k_closest <= KL_KMAX && cfg->kp_rad == 0.0f) || (cfg->k_closest == 0 && cfg->kp_rad > 0.0f),
                KPD_ERR_INVALID, "keypoint features: either 1 <= k_closest <= %d with kp_rad = 0, or k_closest = 0 with kp_rad > 0 (got %d, %f)",
                KL_KMAX, cfg->k_closest, cfg->kp_rad);
    KPD_REQUIRE(K >= 1 && K <= 4096, KPD_ERR_INVALID, "n_keypoints=%d", K);
    KPD_REQUIRE(cfg->message_norm >= 0.0f, KPD_ERR_INVALID, "message_norm=%f", cfg->message_norm);
    kpd_recegnn *m = new kpd_recegnn();
    m->cfg = *cfg;
    const int ef = cfg->use_sameres_feat ? 1 : 0, f1 = 2 * RW + 2;
    size_t bytes = (size_t)cfg->n_convs * ((size_t)2 * f1 * RW + 3 * RW * RW + (size_t)2 * RW * RW + 16 * RW) * 4 +
                   ((size_t)D * K * D + (size_t)D * K + (size_t)D * D + (size_t)(D + 32) * D + 4 * RW) * 4 + (1 << 20);
    kpd_status st = m->warena.reserve(bytes);
    if (st != KPD_OK) {
        delete m;
        return st;
    }
    m->warena.poison_at = 2;          // packed weights: poisoned only at KPD_POISON >= 2 (engine.h)
    Arena &A = m->warena;
    m->conv.resize(cfg->n_convs);
    for (int i = 0; i < cfg->n_convs; ++i) {
        ConvW &c = m->conv[i];
        c.in = i == 0 ? in : hid;                                   // receptor_encoder.py:431-449
        c.out = i == cfg->n_convs - 1 ? D : hid;
        const int e_in = 2 * c.in + 1 + ef;
        c.We_t = A.take<float>((size_t)e_in * hid); c.be = A.take<float>(hid);
        c.W2_t = A.take<float>((size_t)hid * hid); c.b2 = A.take<float>(hid);
        c.watt = A.take<float>(hid); c.batt = A.take<float>(1);
        c.Wn1_t = A.take<float>((size_t)(c.in + hid) * hid); c.bn1 = A.take<float>(hid);
        c.Wn2_t = A.take<float>((size_t)hid * c.out); c.bn2 = A.take<float>(c.out);
        const std::string p = "rec_convs." + std::to_string(i) + ".";
        for (const char *s : {"edge_mlp.0.weight", "edge_mlp.0.bias", "edge_mlp.2.weight", "edge_mlp.2.bias", "soft_attention.0.weight",
                              "soft_attention.0.bias", "node_mlp.0.weight", "node_mlp.0.bias", "node_mlp.2.weight", "node_mlp.2.bias"})
            m->expected.insert(p + s);
        if (!cfg->fix_pos) {
            c.Wc_t = A.take<float>((size_t)e_in * hid); c.bc = A.take<float>(hid); c.w3 = A.take<float>(hid);
            for (const char *s : {"coord_mlp.0.weight", "coord_mlp.0.bias", "coord_mlp.2.weight"}) m->expected.insert(p + s);
        }
        if (cfg->norm) {
            c.ln_w = A.take<float>(c.out); c.ln_b = A.take<float>(c.out);
            m->expected.insert(p + "layer_norm.weight"); m->expected.insert(p + "layer_norm.bias");
        }
    }
    m->kpe_W = A.take<float>((size_t)D * K * D); m->kpe_b = A.take<float>((size_t)D * K);
    m->fc_src_t = A.take<float>((size_t)D * D);
    m->kpf_W_t = A.take<float>((size_t)(D + cfg->k_closest) * D); m->kpf_b = A.take<float>(D);
    for (const char *s : {"keypoint_embedding.0.weight", "keypoint_embedding.0.bias", "rec_kp_conv.fc_src.weight",
                          "rec_kp_conv.kp_feature_mlp.0.weight", "rec_kp_conv.kp_feature_mlp.0.bias"})
        m->expected.insert(s);
    if (cfg->norm) {
        m->kp_lw = A.take<float>(D); m->kp_lb = A.take<float>(D);
        m->expected.insert("rec_kp_conv.layer_norm.weight"); m->expected.insert("rec_kp_conv.layer_norm.bias");
    }
    if (!m->kpf_b || (cfg->norm && !m->kp_lb)) {
        set_error("recegnn weight arena too small (internal sizing error)");
        m->warena.release();
        delete m;
        return KPD_ERR_HIP;
    }
    *out = m;
    return KPD_OK;
}

extern "C" void kpd_recegnn_destroy(kpd_recegnn *m) {
    if (!m) return;
    m->warena.release();
    m->ws.release();
    delete m;
}

static kpd_status shape_is(const char *name, const int64_t *shape, int ndim, std::initializer_list<int64_t> want) {
    bool ok = ndim == (int)want.size();
    int i = 0;
    for (int64_t w : want) {
        if (ok && shape[i] != w) ok = false;
        ++i;
    }
    if (!ok) {
        std::string got, exp;
        for (int j = 0; j < ndim; ++j) got += std::to_string(shape[j]) + ",";
        for (int64_t w : want) exp += std::to_string(w) + ",";
        set_error("weight %s has shape [%s], expected [%s]", name, got.c_str(), exp.c_str());
        return KPD_ERR_WEIGHTS;
    }
    return KPD_OK;
}

extern "C" kpd_status kpd_recegnn_load_weight(kpd_recegnn *m, const char *name, const float *w, const int64_t *shape, int32_t ndim,
                                              void *stream) {
    KPD_REQUIRE(m && name && w && shape, KPD_ERR_INVALID, "null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const std::string n(name);
    const kpd_recegnn_config &c = m->cfg;
    const int hid = c.hidden_n_node_feat, D = c.out_n_node_feat, ef = c.use_sameres_feat ? 1 : 0;
    if (n == "rec_kp_conv.fc_dst.weight") return KPD_OK;           // exists upstream, never applied (:190-191)
    KPD_REQUIRE(m->expected.count(n), KPD_ERR_WEIGHTS, "unexpected weight '%s'", name);
    auto vec = [&](float *dst, int len) -> kpd_status {
        KPD_TRY(shape_is(name, shape, ndim, {len}));
        return copy_pad(w, len, dst, len, st);
    };
    auto mat_t = [&](float *dst, int rows, int cols) -> kpd_status {        // torch [rows][cols] -> dst [cols][rows]
        KPD_TRY(shape_is(name, shape, ndim, {rows, cols}));
        return transpose2d(w, rows, cols, dst, st);
    };
    if (n.rfind("rec_convs.", 0) == 0) {
        const size_t dot = n.find('.', 10);
        const int i = atoi(n.substr(10, dot - 10).c_str());
        KPD_REQUIRE(i >= 0 && i < c.n_convs, KPD_ERR_WEIGHTS, "conv index in '%s'", name);
        ConvW &cw = m->conv[i];
        const std::string r = n.substr(dot + 1);
        const int e_in = 2 * cw.in + 1 + ef;
        if (r == "edge_mlp.0.weight") KPD_TRY(mat_t(cw.We_t, hid, e_in));
        else if (r == "edge_mlp.0.bias") KPD_TRY(vec(cw.be, hid));
        else if (r == "edge_mlp.2.weight") KPD_TRY(mat_t(cw.W2_t, hid, hid));
        else if (r == "edge_mlp.2.bias") KPD_TRY(vec(cw.b2, hid));
        else if (r == "soft_attention.0.weight") { KPD_TRY(shape_is(name, shape, ndim, {1, hid})); KPD_TRY(copy_pad(w, hid, cw.watt, hid, st)); }
        else if (r == "soft_attention.0.bias") KPD_TRY(vec(cw.batt, 1));
        else if (r == "coord_mlp.0.weight") KPD_TRY(mat_t(cw.Wc_t, hid, e_in));
        else if (r == "coord_mlp.0.bias") KPD_TRY(vec(cw.bc, hid));
        else if (r == "coord_mlp.2.weight") { KPD_TRY(shape_is(name, shape, ndim, {1, hid})); KPD_TRY(copy_pad(w, hid, cw.w3, hid, st)); }
        else if (r == "node_mlp.0.weight") KPD_TRY(mat_t(cw.Wn1_t, hid, cw.in + hid));
        else if (r == "node_mlp.0.bias") KPD_TRY(vec(cw.bn1, hid));
        else if (r == "node_mlp.2.weight") KPD_TRY(mat_t(cw.Wn2_t, cw.out, hid));
        else if (r == "node_mlp.2.bias") KPD_TRY(vec(cw.bn2, cw.out));
        else if (r == "layer_norm.weight") KPD_TRY(vec(cw.ln_w, cw.out));
        else if (r == "layer_norm.bias") KPD_TRY(vec(cw.ln_b, cw.out));
    } else if (n == "keypoint_embedding.0.weight") {
        KPD_TRY(shape_is(name, shape, ndim, {(int64_t)D * c.n_keypoints, D}));
        KPD_TRY(copy_pad(w, D * c.n_keypoints * D, m->kpe_W, D * c.n_keypoints * D, st));
    } else if (n == "keypoint_embedding.0.bias") KPD_TRY(vec(m->kpe_b, D * c.n_keypoints));
    else if (n == "rec_kp_conv.fc_src.weight") KPD_TRY(mat_t(m->fc_src_t, D, D));
    else if (n == "rec_kp_conv.kp_feature_mlp.0.weight") KPD_TRY(mat_t(m->kpf_W_t, D, D + c.k_closest));
    else if (n == "rec_kp_conv.kp_feature_mlp.0.bias") KPD_TRY(vec(m->kpf_b, D));
    else if (n == "rec_kp_conv.layer_norm.weight") KPD_TRY(vec(m->kp_lw, D));
    else if (n == "rec_kp_conv.layer_norm.bias") KPD_TRY(vec(m->kp_lb, D));
    m->loaded.insert(n);
    return KPD_OK;
}

extern "C" kpd_status kpd_recegnn_commit(kpd_recegnn *m) {
    KPD_REQUIRE(m, KPD_ERR_INVALID, "null handle");
    for (const std::string &e : m->expected)
        KPD_REQUIRE(m->loaded.count(e), KPD_ERR_WEIGHTS, "missing weight '%s' (%zu of %zu loaded)", e.c_str(), m->loaded.size(),
                    m->expected.size());
    m->committed = true;
    return KPD_OK;
}

extern "C" kpd_status kpd_recegnn_reserve(kpd_recegnn *m, int32_t max_B, int32_t max_n_rec, int32_t max_n_rr, int32_t max_rec_pg) {
    KPD_REQUIRE(m, KPD_ERR_INVALID, "null handle");
    KPD_REQUIRE(max_B >= 1 && max_n_rec >= 1 && max_n_rr >= 0 && max_rec_pg >= 1, KPD_ERR_INVALID, "reserve: non-positive size");
    if (max_B <= m->cap_B && max_n_rec <= m->cap_rec && max_n_rr <= m->cap_rr && max_rec_pg <= m->cap_maxrec) return KPD_OK;
    max_B = std::max(max_B, m->cap_B); max_n_rec = std::max(max_n_rec, m->cap_rec);
    max_n_rr = std::max(max_n_rr, m->cap_rr); max_rec_pg = std::max(max_rec_pg, m->cap_maxrec);
    const kpd_recegnn_config &c = m->cfg;
    const int K = c.n_keypoints, D = c.out_n_node_feat, n_kp = max_B * K;
    const int cap_rk = n_kp * (c.k_closest > 0 ? c.k_closest : std::min(100, max_rec_pg));      // radius: at most 100 per keypoint (:246)
    const int n_max = std::max(max_n_rec, n_kp);
    size_t bytes = 1 << 20;
    auto add = [&](size_t cnt) { bytes += ((cnt * 4 + 255) & ~size_t(255)); };
    add((size_t)max_n_rec * RW); add((size_t)max_n_rec * RW); add((size_t)max_n_rec * 3); add((size_t)max_n_rec * 3);
    add((size_t)max_n_rec * 4 * RW); add(max_B); add((size_t)max_B * D); add((size_t)n_kp * D); add((size_t)max_n_rec * D); add((size_t)n_kp * D);
    add(max_n_rec); add(max_B + 1); add(max_B + 1); add(n_max); add(max_B + 8); add(cap_rk); add(cap_rk); add(max_n_rec + 1); add(n_kp + 1); add(n_kp + 1);
    add(max_B + 1);
    KPD_TRY(m->ws.reserve(bytes));
    Arena &W = m->ws;
    for (int i = 0; i < 2; ++i) m->h[i] = W.take<float>((size_t)max_n_rec * RW);
    for (int i = 0; i < 2; ++i) m->x[i] = W.take<float>((size_t)max_n_rec * 3);
    m->P = W.take<float>((size_t)max_n_rec * 4 * RW); m->z = W.take<float>(max_B);
    m->gmean = W.take<float>((size_t)max_B * D); m->kp_h0 = W.take<float>((size_t)n_kp * D);
    m->ft_src = W.take<float>((size_t)max_n_rec * D); m->ft_dst = W.take<float>((size_t)n_kp * D);
    m->bidx = W.take<int>(max_n_rec); m->kp_ptr = W.take<int>(max_B + 1); m->off_tmp = W.take<int>(max_B + 1); m->deg_tmp = W.take<int>(n_max);
    m->rad_tmp = W.take<int>(max_B + 8);
    m->xm_src = W.take<int>(cap_rk); m->xm_dst = W.take<int>(cap_rk); m->xm_rowptr = W.take<int>(max_n_rec + 1);
    m->rk_rowptr = W.take<int>(n_kp + 1); m->kk_rowptr = W.take<int>(n_kp + 1); m->kk_off = W.take<int>(max_B + 1);
    KPD_REQUIRE(m->kk_off != nullptr, KPD_ERR_HIP, "recegnn workspace arena too small (internal sizing error)");
    m->cap_B = max_B; m->cap_rec = max_n_rec; m->cap_rr = max_n_rr; m->cap_maxrec = max_rec_pg;
    return KPD_OK;
}

extern "C" kpd_status kpd_recegnn_forward(kpd_recegnn *m, const kpd_rec_batch *bt, const float *rr_same_res, const kpd_rec_out *out,
                                          float *rec_h_out, float *rec_x_out, void *stream) {
    KPD_REQUIRE(m && bt && out, KPD_ERR_INVALID, "null argument");
    KPD_REQUIRE(m->committed, KPD_ERR_STATE, "kpd_recegnn_forward before kpd_recegnn_commit");
    KPD_REQUIRE(bt->B >= 1 && bt->n_rec >= 1 && bt->rec_ptr && bt->rec_x && bt->rec_h && bt->rr_rowptr, KPD_ERR_INVALID, "bad batch");
    KPD_REQUIRE(bt->n_rr == 0 || (bt->rr_src && bt->rr_dst), KPD_ERR_INVALID, "rr edges missing");
    KPD_REQUIRE(bt->B <= m->cap_B && bt->n_rec <= m->cap_rec && bt->n_rr <= m->cap_rr && bt->max_rec <= m->cap_maxrec,
                KPD_ERR_CAPACITY, "batch exceeds reserved workspace (call kpd_recegnn_reserve)");
    const kpd_recegnn_config &c = m->cfg;
    KPD_REQUIRE(!c.use_sameres_feat || rr_same_res || bt->n_rr == 0, KPD_ERR_INVALID, "use_sameres_feat needs the rr same_res column");
    KPD_REQUIRE(out->kp_x && out->kp_h && out->rk_src && out->rk_dst && out->kk_src && out->kk_dst && out->kk_per_graph && out->counts,
                KPD_ERR_INVALID, "output buffers missing");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int hid = c.hidden_n_node_feat, D = c.out_n_node_feat, K = c.n_keypoints, B = bt->B, n_rec = bt->n_rec, n_kp = B * K;
    KPD_REQUIRE(out->cap_kk >= (long)n_kp * std::min(K - 1, 100), KPD_ERR_CAPACITY, "cap_kk=%d too small", out->cap_kk);

    KPD_TRY(launch_node_graph_index(bt->rec_ptr, B, n_rec, m->bidx, st));
    KPD_TRY(launch_iota_scaled(m->kp_ptr, B + 1, K, st));
    if (c.message_norm == 0.0f) {
        hipLaunchKernelGGL(k_rc_z, dim3(cdiv(B, 256)), dim3(256), 0, st, bt->rr_rowptr, bt->rec_ptr, B, m->z);
        KPD_LAUNCH_CHECK();
    }
    // ReceptorConv stack (:512-513); layer i reads (h, x) buffer i & 1 (layer 0: the inputs) and writes the other
    const float *h_in = bt->rec_h, *x_in = bt->rec_x;
    for (int i = 0; i < c.n_convs; ++i) {
        const ConvW &cw = m->conv[i];
        RcProjArgs pa;
        pa.h = h_in; pa.n = n_rec; pa.in = cw.in; pa.hid = hid; pa.We_t = cw.We_t; pa.Wc_t = cw.Wc_t; pa.be = cw.be; pa.bc = cw.bc; pa.P = m->P;
        hipLaunchKernelGGL(k_rc_proj, dim3(cdiv(n_rec, 4)), dim3(RW), 0, st, pa);
        KPD_LAUNCH_CHECK();
        RcConvArgs a;
        memset(&a, 0, sizeof(a));
        a.n = n_rec; a.in = cw.in; a.hid = hid; a.out = cw.out; a.ef = c.use_sameres_feat ? 1 : 0;
        a.h = h_in; a.x = x_in; a.P = m->P; a.src = bt->rr_src; a.rowptr = bt->rr_rowptr; a.same_res = rr_same_res; a.bidx = m->bidx;
        a.z = c.message_norm == 0.0f ? m->z : nullptr; a.norm_const = c.message_norm == 0.0f ? 1.0f : c.message_norm;
        a.We_t = cw.We_t; a.Wc_t = cw.Wc_t; a.W2_t = cw.W2_t; a.b2 = cw.b2; a.watt = cw.watt; a.batt = cw.batt; a.w3 = cw.w3;
        a.Wn1_t = cw.Wn1_t; a.bn1 = cw.bn1; a.Wn2_t = cw.Wn2_t; a.bn2 = cw.bn2; a.ln_w = cw.ln_w; a.ln_b = cw.ln_b;
        a.use_tanh = c.use_tanh; a.fix_pos = c.fix_pos; a.coords_range = c.coords_range;
        a.h_out = m->h[i & 1]; a.x_out = m->x[i & 1];
        hipLaunchKernelGGL(k_rc_conv, dim3(n_rec), dim3(RW), 0, st, a);
        KPD_LAUNCH_CHECK();
        h_in = a.h_out; x_in = a.x_out;
    }
    if (rec_h_out) KPD_HIP(hipMemcpyAsync(rec_h_out, h_in, (size_t)n_rec * D * 4, hipMemcpyDeviceToDevice, st));      // :516-517
    if (rec_x_out) KPD_HIP(hipMemcpyAsync(rec_x_out, x_in, (size_t)n_rec * 3 * 4, hipMemcpyDeviceToDevice, st));

    // keypoint features from the mean receptor feature (:526-530), attention-pooled positions (:188-222)
    KPD_TRY(launch_graph_mean(h_in, bt->rec_ptr, B, D, m->gmean, st));
    hipLaunchKernelGGL(k_rc_kp_embed, dim3(B, cdiv(D * K, RW)), dim3(RW), 0, st, m->gmean, m->kpe_W, m->kpe_b, D, D * K, m->kp_h0);
    KPD_LAUNCH_CHECK();
    KPD_TRY(launch_linear_rows(h_in, n_rec, D, m->fc_src_t, m->ft_src, st));
    KPD_TRY(launch_linear_rows(m->kp_h0, n_kp, D, m->fc_src_t, m->ft_dst, st));                // fc_src on both sides (:190-191)
    KPD_TRY(launch_kp_attention(m->ft_src, m->ft_dst, c.fix_pos ? bt->rec_x : x_in, bt->rec_ptr, n_kp, K, D, out->kp_x, st));

    if (c.k_closest > 0) {
        // k nearest receptor atoms of every keypoint by the ORIGINAL positions (:262-267); kp-major = rk edges
        KPD_TRY(launch_knn_bipartite(bt->rec_x, bt->rec_ptr, n_rec, bt->max_rec, out->kp_x, m->kp_ptr, n_kp, K, B, c.k_closest, m->off_tmp,
                                     m->xm_src, m->xm_dst, m->xm_rowptr, out->rk_src, out->rk_dst, m->rk_rowptr, st));
        hipLaunchKernelGGL(k_rc_kp_feat, dim3(n_kp), dim3(RW), 0, st, h_in, bt->rec_x, out->kp_x, out->rk_src, c.k_closest, D, m->kpf_W_t,
                           m->kpf_b, m->kp_lw, m->kp_lb, out->kp_h);
    } else {
        // receptor atoms within kp_rad of every keypoint (original positions, at most 100, index order; :238-262)
        KPD_TRY(launch_radius_bipartite(bt->rec_x, bt->rec_ptr, n_rec, bt->max_rec, out->kp_x, m->kp_ptr, n_kp, K, B, c.kp_rad, 100,
                                        m->rad_tmp, m->rad_tmp + B, m->off_tmp, m->xm_src, m->xm_dst, m->xm_rowptr, out->rk_src,
                                        out->rk_dst, m->rk_rowptr, st));
        hipLaunchKernelGGL(k_rc_kp_radfeat, dim3(n_kp), dim3(RW), 0, st, h_in, out->rk_src, m->rk_rowptr, m->off_tmp, K, D, m->kpf_W_t,
                           m->kpf_b, m->kp_lw, m->kp_lb, out->kp_h);
    }
    KPD_LAUNCH_CHECK();

    // keypoint-keypoint radius graph (:541); counts = {E_kk, E_rk}
    KPD_TRY(launch_radius_graph(out->kp_x, m->kp_ptr, B, n_kp, K, c.kk_cutoff, 100, out->cap_kk, out->kk_src, out->kk_dst, m->kk_rowptr,
                                out->kk_per_graph, m->deg_tmp, m->kk_off, m->off_tmp, out->counts, st));
    return KPD_OK;
}
