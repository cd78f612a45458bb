// GVP keypoint receptor encoder behind the kpd_recenc_* C ABI (include/kpd.h).  Replaces
// ReceptorEncoderGVP.forward (models/receptor_encoder_gvp.py:212-294) as a whole: scalar embedding,
// rec-rec GVP convolutions (GVPEdgeConv, models/gvp.py:249-341), KeypointInitializer (:40-93),
// kNN rec->kp edges (:297-321), rec-kp convolutions and the keypoint radius graph (:285-292).
// Runs once per pocket; the GVP convolutions reuse the denoiser's kernels (gvp_kernels.hip) with
// node type 0 = receptor atoms, node type 1 = keypoints.
#include <string.h>

#include "egnn_kernels.h"
#include "gvp_host.h"
#include "mfma_core.h"
#include "rec_kernels.h"

using namespace kpd;

namespace kpd {

// ---- small kernels ---------------------------------------------------------------------------------
// out = LN(SiLU(W1 SiLU(W0 in + b0) + b1))   (receptor_encoder_gvp.py:158-164, 221-222)
constexpr int REMB_NODES = 4;
__global__ __launch_bounds__(256) void k_rec_embed(const float *__restrict__ in, int n, int fin, const float *__restrict__ W0,
                                                   const float *__restrict__ b0, const float *__restrict__ W1t,
                                                   const float *__restrict__ b1, const float *__restrict__ lw,
                                                   const float *__restrict__ lb, int S, float *__restrict__ out) {
    __shared__ float s_in[REMB_NODES][64];
    __shared__ float s_hid[REMB_NODES][256];
    __shared__ float s_red[REMB_NODES][2][4];
    const int node0 = blockIdx.x * REMB_NODES, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < REMB_NODES * fin; i += 256) {
        const int j = i / fin, k = i - j * fin;
        s_in[j][k] = node0 + j < n ? in[(size_t)(node0 + j) * fin + k] : 0.0f;
    }
    __syncthreads();
    const bool on = tid < S;
    float y[REMB_NODES];
#pragma unroll
    for (int j = 0; j < REMB_NODES; ++j) y[j] = on ? b0[tid] : 0.0f;
    if (on)
        for (int k = 0; k < fin; ++k) {
            const float wv = W0[(size_t)tid * fin + k];
#pragma unroll
            for (int j = 0; j < REMB_NODES; ++j) y[j] = fmaf(wv, s_in[j][k], y[j]);
        }
#pragma unroll
    for (int j = 0; j < REMB_NODES; ++j) s_hid[j][tid] = on ? silu(y[j]) : 0.0f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < REMB_NODES; ++j) y[j] = on ? b1[tid] : 0.0f;
    if (on)
        for (int u = 0; u < S; ++u) {
            const float wv = W1t[(size_t)u * S + tid];
#pragma unroll
            for (int j = 0; j < REMB_NODES; ++j) y[j] = fmaf(wv, s_hid[j][u], y[j]);
        }
#pragma unroll
    for (int j = 0; j < REMB_NODES; ++j) y[j] = on ? silu(y[j]) : 0.0f;
    float mean[REMB_NODES];
#pragma unroll
    for (int j = 0; j < REMB_NODES; ++j) {
        float v = y[j];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) s_red[j][0][wave] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < REMB_NODES; ++j) {
        mean[j] = (s_red[j][0][0] + s_red[j][0][1] + s_red[j][0][2] + s_red[j][0][3]) / (float)S;
        const float d = on ? y[j] - mean[j] : 0.0f;
        float v = d * d;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) s_red[j][1][wave] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < REMB_NODES; ++j) {
        const float rstd = 1.0f / sqrtf((s_red[j][1][0] + s_red[j][1][1] + s_red[j][1][2] + s_red[j][1][3]) / (float)S + 1e-5f);
        const int v = node0 + j;
        if (on && v < n) out[(size_t)v * S + tid] = (y[j] - mean[j]) * rstd * lw[tid] + lb[tid];
    }
}

__global__ void k_iota_scaled(int *out, int n, int scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i * scale;
}

// meta for a launch that carries a single edge type: E = *count_dev (or count_host)
__global__ void k_meta_single(int et, const int *__restrict__ count_dev, int count_host, int *__restrict__ meta) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int E = count_dev ? *count_dev : count_host;
        const int T = (E + TM - 1) / TM;
        for (int e = 0; e < 4; ++e) {
            meta[e] = e == et ? E : 0;
            meta[4 + e] = e <= et ? 0 : T;
        }
        meta[8] = T;
    }
}

// message_norm == 0: z[b] = edges into the graph's dst nodes / number of dst nodes  (no +1 here:
// receptor_encoder_gvp.py:245-246, 268-269)
__global__ void k_z_indegree(const int *__restrict__ rowptr, const int *__restrict__ ptr, int B, float *__restrict__ z) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) z[b] = (float)(rowptr[ptr[b + 1]] - rowptr[ptr[b]]) / (float)(ptr[b + 1] - ptr[b]);
}

// per-graph mean of node scalars (dgl.readout_nodes mean, receptor_encoder_gvp.py:51)
__global__ __launch_bounds__(256) void k_graph_mean(const float *__restrict__ s, const int *__restrict__ ptr, int S,
                                                    float *__restrict__ out) {
    const int b = blockIdx.x, c = threadIdx.x;
    if (c >= S) return;
    float acc = 0.0f;
    for (int v = ptr[b]; v < ptr[b + 1]; ++v) acc += s[(size_t)v * S + c];
    out[(size_t)b * S + c] = acc / (float)(ptr[b + 1] - ptr[b]);
}

// keypoint embedding: LN_{S*K}(SiLU(W mean + b)) -> [B][K][S]  (receptor_encoder_gvp.py:31-35, 54-55)
constexpr int KPE_MAX = 10240;
__global__ __launch_bounds__(256) void k_kp_embed(const float *__restrict__ mean, const float *__restrict__ W,
                                                  const float *__restrict__ b, const float *__restrict__ lw,
                                                  const float *__restrict__ lb, int S, int SK, float *__restrict__ out) {
    __shared__ float s_m[256];
    __shared__ float s_buf[KPE_MAX];
    __shared__ float s_red[4];
    __shared__ float s_stat[2];
    const int g = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid < S) s_m[tid] = mean[(size_t)g * S + tid];
    __syncthreads();
    float part = 0.0f;
    for (int j = tid; j < SK; j += 256) {
        float acc = b[j];
        const float *w = W + (size_t)j * S;
        for (int k = 0; k < S; ++k) acc = fmaf(w[k], s_m[k], acc);
        acc = silu(acc);
        s_buf[j] = acc;
        part += acc;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o);
    if (lane == 0) s_red[wave] = part;
    __syncthreads();
    if (tid == 0) s_stat[0] = (s_red[0] + s_red[1] + s_red[2] + s_red[3]) / (float)SK;
    __syncthreads();
    const float mu = s_stat[0];
    part = 0.0f;
    for (int j = tid; j < SK; j += 256) {
        const float d = s_buf[j] - mu;
        part = fmaf(d, d, part);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o);
    __syncthreads();
    if (lane == 0) s_red[wave] = part;
    __syncthreads();
    if (tid == 0) s_stat[1] = 1.0f / sqrtf((s_red[0] + s_red[1] + s_red[2] + s_red[3]) / (float)SK + 1e-5f);
    __syncthreads();
    const float rstd = s_stat[1];
    for (int j = tid; j < SK; j += 256) out[(size_t)g * SK + j] = (s_buf[j] - mu) * rstd * lw[j] + lb[j];
}

// out[node][:] = Wt^T in[node][:]  (bias-free Linear with the weight stored transposed [in][out])
__global__ __launch_bounds__(256) void k_linear_rows(const float *__restrict__ in, int n, int S, const float *__restrict__ Wt,
                                                     float *__restrict__ out) {
    __shared__ float s_in[4][256];
    const int node0 = blockIdx.x * 4, tid = threadIdx.x;
    for (int i = tid; i < 4 * S; i += 256) {
        const int j = i / S, k = i - j * S;
        s_in[j][k] = node0 + j < n ? in[(size_t)(node0 + j) * S + k] : 0.0f;
    }
    __syncthreads();
    if (tid >= S) return;
    float y[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < S; ++k) {
        const float wv = Wt[(size_t)k * S + tid];
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = fmaf(wv, s_in[j][k], y[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (node0 + j < n) out[(size_t)(node0 + j) * S + tid] = y[j];
}

// attention-pooled keypoint positions (receptor_encoder_gvp.py:57-87): one workgroup per keypoint;
// logits are exponentiated without max-subtraction, exactly as upstream.
__global__ __launch_bounds__(256) void k_kp_attention(const float *__restrict__ ft_src, const float *__restrict__ ft_dst,
                                                      const float *__restrict__ rec_x, const int *__restrict__ rec_ptr, int K,
                                                      int S, float *__restrict__ kp_x) {
    __shared__ float s_q[256];
    __shared__ float s_part[256][4];
    const int kp = blockIdx.x, g = kp / K, tid = threadIdx.x;
    if (tid < S) s_q[tid] = ft_dst[(size_t)kp * S + tid];
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)S);
    float a_sum = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
    for (int r = rec_ptr[g] + tid; r < rec_ptr[g + 1]; r += 256) {
        const float *f = ft_src + (size_t)r * S;
        float dot = 0.0f;
        for (int k = 0; k < S; ++k) dot = fmaf(f[k], s_q[k], dot);
        const float a = expf(dot * scale);
        a_sum += a;
        ax = fmaf(a, rec_x[(size_t)r * 3], ax);
        ay = fmaf(a, rec_x[(size_t)r * 3 + 1], ay);
        az = fmaf(a, rec_x[(size_t)r * 3 + 2], az);
    }
    s_part[tid][0] = a_sum; s_part[tid][1] = ax; s_part[tid][2] = ay; s_part[tid][3] = az;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (tid < o) {
#pragma unroll
            for (int c = 0; c < 4; ++c) s_part[tid][c] += s_part[tid + o][c];
        }
        __syncthreads();
    }
    if (tid < 3) kp_x[(size_t)kp * 3 + tid] = s_part[0][1 + tid] / s_part[0][0];
}

kpd_status launch_iota_scaled(int *out, int n, int scale, hipStream_t st) {
    hipLaunchKernelGGL(k_iota_scaled, dim3(cdiv(n, 256)), dim3(256), 0, st, out, n, scale);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status launch_graph_mean(const float *s, const int *ptr, int B, int S, float *out, hipStream_t st) {
    KPD_REQUIRE(S <= 256, KPD_ERR_INVALID, "graph mean: S=%d > 256", S);
    hipLaunchKernelGGL(k_graph_mean, dim3(B), dim3(256), 0, st, s, ptr, S, out);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status launch_linear_rows(const float *in, int n, int S, const float *Wt, float *out, hipStream_t st) {
    KPD_REQUIRE(S <= 256, KPD_ERR_INVALID, "linear rows: S=%d > 256", S);
    hipLaunchKernelGGL(k_linear_rows, dim3(cdiv(n, 4)), dim3(256), 0, st, in, n, S, Wt, out);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status launch_kp_attention(const float *ft_src, const float *ft_dst, const float *rec_x, const int *rec_ptr, int n_kp, int K,
                               int S, float *kp_x, hipStream_t st) {
    KPD_REQUIRE(S <= 256, KPD_ERR_INVALID, "kp attention: S=%d > 256", S);
    hipLaunchKernelGGL(k_kp_attention, dim3(n_kp), dim3(256), 0, st, ft_src, ft_dst, rec_x, rec_ptr, K, S, kp_x);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

}  // namespace kpd

// ---- engine -----------------------------------------------------------------------------------------
struct kpd_recenc {
    kpd_recenc_config cfg;
    int S;
    Arena warena, ws;
    std::vector<std::vector<HostGvp>> rr_msg, rr_upd, rk_msg, rk_upd;   // [conv][j]
    std::vector<float *> rr_ln1w, rr_ln1b, rr_ln2w, rr_ln2b, rk_ln1w, rk_ln1b, rk_ln2w, rk_ln2b;
    float *emb_W0, *emb_b0, *emb_W1t, *emb_b1, *emb_lw, *emb_lb;
    float *kpe_W, *kpe_b, *kpe_lw, *kpe_lb, *src_Wt, *dst_Wt;
    std::set<std::string> expected, loaded, ignored;
    bool committed = false;
    // workspace
    int cap_B = 0, cap_rec = 0, cap_rr = 0, cap_maxrec = 0;
    float *s[2], *v[2], *s_tmp[2], *Psrc, *Pdst, *ms_main, *ms_cont, *mv_main, *mv_cont;
    float *gmean, *kp_emb, *ft_src, *ft_dst, *z;
    int *bidx[2], *kp_ptr, *meta, *off_tmp, *deg_tmp, *rad_tmp, *xm_src, *xm_dst, *xm_rowptr, *rk_rowptr, *kk_rowptr, *kk_off;
};

static void alloc_conv(kpd_recenc *m, Arena &A, std::vector<HostGvp> &msg, std::vector<HostGvp> &upd, bool use_dst,
                       const std::string &pre, float **ln) {
    const kpd_recenc_config &c = m->cfg;
    const int S = m->S;
    msg.resize(c.n_message_gvps);
    upd.resize(c.n_update_gvps);
    for (int j = 0; j < c.n_message_gvps; ++j) {
        HostGvp &g = msg[j];
        g.vin = j == 0 ? GV + 1 + (use_dst ? GV : 0) : GV;
        g.vout = GV;
        g.s_in = j == 0 ? S + 16 + (use_dst ? S : 0) : S;
        g.sout = S;
        g.split = j == 0 ? (use_dst ? SPLIT_SRC_DST : SPLIT_SRC) : SPLIT_NONE;
        g.S = S;
        g.chain_pos = j;
        g.vcut = GV - c.vector_size;             // narrower models: every 16-channel block carries vector_size channels (gvp_host.hip)
        alloc_gvp(A, g, m->expected, pre + "edge_message." + std::to_string(j));
    }
    for (int j = 0; j < c.n_update_gvps; ++j) {
        HostGvp &g = upd[j];
        g.vin = GV; g.vout = GV; g.s_in = S; g.sout = S;
        g.chain_pos = 1;
        g.vcut = GV - c.vector_size;
        alloc_gvp(A, g, m->expected, pre + "node_update." + std::to_string(j));
    }
    for (int i = 0; i < 4; ++i) ln[i] = A.take<float>(S);
    for (const char *s : {".feat_norm.weight", ".feat_norm.bias"}) {
        m->expected.insert(pre + "message_layer_norm" + s);
        m->expected.insert(pre + "update_layer_norm" + s);
    }
}

extern "C" kpd_status kpd_recenc_create(const kpd_recenc_config *cfg, kpd_recenc **out) {
    KPD_REQUIRE(cfg && out, KPD_ERR_INVALID, "null argument");
    KPD_REQUIRE(cfg->vector_size >= 1 && cfg->vector_size <= GV, KPD_ERR_INVALID, "vector_size=%d outside 1 .. %d", cfg->vector_size, GV);
    KPD_REQUIRE(cfg->out_scalar_size == 128 || cfg->out_scalar_size == 256, KPD_ERR_INVALID, "out_scalar_size=%d: supported 128, 256",
                cfg->out_scalar_size);
    KPD_REQUIRE(cfg->in_scalar_size >= 1 && cfg->in_scalar_size <= 64, KPD_ERR_INVALID, "in_scalar_size=%d", cfg->in_scalar_size);
    KPD_REQUIRE((cfg->k_closest >= 1 && cfg->k_closest <= KL_KMAX && cfg->kp_rad == 0.0f) || (cfg->k_closest == 0 && cfg->kp_rad > 0.0f),
                KPD_ERR_INVALID, "rec->kp graph: either 1 <= k_closest <= %d with kp_rad = 0, or k_closest = 0 with kp_rad > 0 (got %d, %f)",
                KL_KMAX, cfg->k_closest, cfg->kp_rad);
    KPD_REQUIRE(cfg->n_keypoints >= 1 && cfg->n_keypoints * cfg->out_scalar_size <= KPE_MAX, KPD_ERR_INVALID, "n_keypoints=%d",
                cfg->n_keypoints);
    KPD_REQUIRE(cfg->n_message_gvps >= 1 && cfg->n_message_gvps <= GVP_MAX_CHAIN && cfg->n_update_gvps >= 1 &&
                    cfg->n_update_gvps <= GVP_MAX_CHAIN, KPD_ERR_INVALID, "GVP chain lengths must be within 1..%d", GVP_MAX_CHAIN);
    KPD_REQUIRE(cfg->n_rr_convs >= 0 && cfg->n_rr_convs <= 16 && cfg->n_rk_convs >= 1 && cfg->n_rk_convs <= 16, KPD_ERR_INVALID,
                "conv counts");
    KPD_REQUIRE(cfg->message_norm_mode >= 0 && cfg->message_norm_mode <= 2, KPD_ERR_INVALID, "message_norm_mode");
    KPD_TRY(egnn_kernels_init());
    kpd_recenc *m = new kpd_recenc();
    m->cfg = *cfg;
    const int S = m->S = cfg->out_scalar_size, K = cfg->n_keypoints, F = cfg->in_scalar_size;
    size_t bytes = gvp_arena_bytes(S) * (size_t)(cfg->n_rr_convs + cfg->n_rk_convs) * (cfg->n_message_gvps + cfg->n_update_gvps) +
                   (size_t)(cfg->n_rr_convs + cfg->n_rk_convs) * 4 * (S * 4 + 256) +
                   ((size_t)S * F + 2 * S * S + (size_t)S * K * S + 3 * (size_t)S * K + 8 * S) * 4 + (1 << 20);
    kpd_status st = m->warena.reserve(bytes);
    if (st != KPD_OK) {
        delete m;
        return st;
    }
    m->warena.poison_at = 2;          // packed weights: poisoned only at KPD_POISON >= 2 (engine.h)
    Arena &A = m->warena;
    m->rr_msg.resize(cfg->n_rr_convs); m->rr_upd.resize(cfg->n_rr_convs);
    m->rk_msg.resize(cfg->n_rk_convs); m->rk_upd.resize(cfg->n_rk_convs);
    m->rr_ln1w.resize(cfg->n_rr_convs); m->rr_ln1b.resize(cfg->n_rr_convs); m->rr_ln2w.resize(cfg->n_rr_convs); m->rr_ln2b.resize(cfg->n_rr_convs);
    m->rk_ln1w.resize(cfg->n_rk_convs); m->rk_ln1b.resize(cfg->n_rk_convs); m->rk_ln2w.resize(cfg->n_rk_convs); m->rk_ln2b.resize(cfg->n_rk_convs);
    for (int i = 0; i < cfg->n_rr_convs; ++i) {
        float *ln[4];
        alloc_conv(m, A, m->rr_msg[i], m->rr_upd[i], false, "rr_conv_layers." + std::to_string(i) + ".", ln);
        m->rr_ln1w[i] = ln[0]; m->rr_ln1b[i] = ln[1]; m->rr_ln2w[i] = ln[2]; m->rr_ln2b[i] = ln[3];
    }
    for (int i = 0; i < cfg->n_rk_convs; ++i) {
        float *ln[4];
        alloc_conv(m, A, m->rk_msg[i], m->rk_upd[i], i != 0, "rk_conv_layers." + std::to_string(i) + ".", ln);   // :194-197
        m->rk_ln1w[i] = ln[0]; m->rk_ln1b[i] = ln[1]; m->rk_ln2w[i] = ln[2]; m->rk_ln2b[i] = ln[3];
    }
    m->emb_W0 = A.take<float>((size_t)S * F); m->emb_b0 = A.take<float>(S);
    m->emb_W1t = A.take<float>((size_t)S * S); m->emb_b1 = A.take<float>(S);
    m->emb_lw = A.take<float>(S); m->emb_lb = A.take<float>(S);
    m->kpe_W = A.take<float>((size_t)S * K * S); m->kpe_b = A.take<float>((size_t)S * K);
    m->kpe_lw = A.take<float>((size_t)S * K); m->kpe_lb = A.take<float>((size_t)S * K);
    m->src_Wt = A.take<float>((size_t)S * S); m->dst_Wt = A.take<float>((size_t)S * S);
    for (const char *s : {"scalar_embed.0.weight", "scalar_embed.0.bias", "scalar_embed.2.weight", "scalar_embed.2.bias",
                          "scalar_norm.weight", "scalar_norm.bias", "keypoint_initializer.src_net.weight",
                          "keypoint_initializer.dst_net.weight", "keypoint_initializer.keypoint_embedding.0.weight",
                          "keypoint_initializer.keypoint_embedding.0.bias", "keypoint_initializer.keypoint_embedding.2.weight",
                          "keypoint_initializer.keypoint_embedding.2.bias"})
        m->expected.insert(s);
    // present in the state dict, unused by forward (receptor_encoder_gvp.py:37)
    m->ignored.insert("keypoint_initializer.norm.weight");
    m->ignored.insert("keypoint_initializer.norm.bias");
    if (!m->dst_Wt) {
        set_error("recenc weight arena too small (internal sizing error)");
        kpd_recenc_destroy(m);
        return KPD_ERR_HIP;
    }
    *out = m;
    return KPD_OK;
}

extern "C" void kpd_recenc_destroy(kpd_recenc *m) {
    if (!m) return;
    m->warena.release();
    m->ws.release();
    delete m;
}

extern "C" kpd_status kpd_recenc_load_weight(kpd_recenc *m, const char *name, const float *w, const int64_t *shape, int32_t ndim,
                                             void *stream) {
    KPD_REQUIRE(m && name && w && shape, KPD_ERR_INVALID, "null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const std::string nm(name);
    if (m->ignored.count(nm)) return KPD_OK;
    if (!m->expected.count(nm)) {
        set_error("unknown or unused weight name '%s' for this configuration", name);
        return KPD_ERR_WEIGHTS;
    }
    const int S = m->S, K = m->cfg.n_keypoints, F = m->cfg.in_scalar_size;
    const std::vector<std::string> tk = split_dots(nm);
    auto tail_from = [&](size_t i) {
        std::string t;
        for (size_t k = i; k < tk.size(); ++k) t += (k > i ? "." : "") + tk[k];
        return t;
    };
    const bool is_w = tk.back() == "weight";
    if (tk[0] == "scalar_embed") {
        if (tk[1] == "0") {
            if (is_w) { KPD_TRY(want_shape(name, shape, ndim, {S, F})); KPD_TRY(copy_pad(w, S * F, m->emb_W0, S * F, st)); }
            else { KPD_TRY(want_shape(name, shape, ndim, {S})); KPD_TRY(copy_pad(w, S, m->emb_b0, S, st)); }
        } else {
            if (is_w) { KPD_TRY(want_shape(name, shape, ndim, {S, S})); KPD_TRY(transpose2d(w, S, S, m->emb_W1t, st)); }
            else { KPD_TRY(want_shape(name, shape, ndim, {S})); KPD_TRY(copy_pad(w, S, m->emb_b1, S, st)); }
        }
    } else if (tk[0] == "scalar_norm") {
        KPD_TRY(want_shape(name, shape, ndim, {S}));
        KPD_TRY(copy_pad(w, S, is_w ? m->emb_lw : m->emb_lb, S, st));
    } else if (tk[0] == "keypoint_initializer") {
        if (tk[1] == "src_net" || tk[1] == "dst_net") {
            KPD_TRY(want_shape(name, shape, ndim, {S, S}));
            KPD_TRY(transpose2d(w, S, S, tk[1] == "src_net" ? m->src_Wt : m->dst_Wt, st));
        } else {   // keypoint_embedding.{0,2}.{weight,bias}
            if (tk[2] == "0") {
                if (is_w) { KPD_TRY(want_shape(name, shape, ndim, {S * K, S})); KPD_TRY(copy_pad(w, S * K * S, m->kpe_W, S * K * S, st)); }
                else { KPD_TRY(want_shape(name, shape, ndim, {S * K})); KPD_TRY(copy_pad(w, S * K, m->kpe_b, S * K, st)); }
            } else {
                KPD_TRY(want_shape(name, shape, ndim, {S * K}));
                KPD_TRY(copy_pad(w, S * K, is_w ? m->kpe_lw : m->kpe_lb, S * K, st));
            }
        }
    } else {       // rr_conv_layers.<i>.<block>... | rk_conv_layers.<i>.<block>...
        const bool rr = tk[0] == "rr_conv_layers";
        const int i = atoi(tk[1].c_str());
        const std::string &blk = tk[2];
        if (blk == "edge_message" || blk == "node_update") {
            std::vector<HostGvp> &vec = blk == "edge_message" ? (rr ? m->rr_msg[i] : m->rk_msg[i]) : (rr ? m->rr_upd[i] : m->rk_upd[i]);
            KPD_TRY(load_gvp_tensor(vec[atoi(tk[3].c_str())], tail_from(4), name, w, shape, ndim, st));
        } else {   // message_layer_norm.feat_norm.<p> | update_layer_norm.feat_norm.<p>
            KPD_TRY(want_shape(name, shape, ndim, {S}));
            float *dst;
            if (blk == "message_layer_norm") dst = rr ? (is_w ? m->rr_ln1w[i] : m->rr_ln1b[i]) : (is_w ? m->rk_ln1w[i] : m->rk_ln1b[i]);
            else dst = rr ? (is_w ? m->rr_ln2w[i] : m->rr_ln2b[i]) : (is_w ? m->rk_ln2w[i] : m->rk_ln2b[i]);
            KPD_TRY(copy_pad(w, S, dst, S, st));
        }
    }
    m->loaded.insert(nm);
    m->committed = false;
    return KPD_OK;
}

extern "C" kpd_status kpd_recenc_commit(kpd_recenc *m) {
    KPD_REQUIRE(m, KPD_ERR_INVALID, "null handle");
    for (const std::string &n : m->expected)
        if (!m->loaded.count(n)) {
            set_error("weight '%s' was never loaded (%zu of %zu loaded)", n.c_str(), m->loaded.size(), m->expected.size());
            return KPD_ERR_WEIGHTS;
        }
    m->committed = true;
    return KPD_OK;
}

// rk edges per keypoint: k of the kNN graph, or at most 10 within kp_rad (receptor_encoder_gvp.py:302-306)
static inline int rk_per_kp(const kpd_recenc_config &c) { return c.k_closest > 0 ? c.k_closest : 10; }

extern "C" kpd_status kpd_recenc_reserve(kpd_recenc *m, int32_t max_B, int32_t max_n_rec, int32_t max_n_rr, int32_t max_rec_pg) {
    KPD_REQUIRE(m, KPD_ERR_INVALID, "null handle");
    KPD_REQUIRE(max_B >= 1 && max_n_rec >= 1 && max_n_rr >= 0 && max_rec_pg >= 1, KPD_ERR_INVALID, "reserve: non-positive size");
    if (max_B <= m->cap_B && max_n_rec <= m->cap_rec && max_n_rr <= m->cap_rr && max_rec_pg <= m->cap_maxrec) return KPD_OK;
    max_B = std::max(max_B, m->cap_B); max_n_rec = std::max(max_n_rec, m->cap_rec);
    max_n_rr = std::max(max_n_rr, m->cap_rr); max_rec_pg = std::max(max_rec_pg, m->cap_maxrec);
    const int S = m->S, K = m->cfg.n_keypoints, n_kp = max_B * K;
    const int cap_rk = n_kp * rk_per_kp(m->cfg);
    const int n[2] = {max_n_rec, n_kp};
    const int e_max = std::max(std::max(max_n_rr, cap_rk), 1);
    const int n_max = std::max(max_n_rec, n_kp);
    const int tiles = cdiv(e_max, TM) + 1;
    size_t bytes = 1 << 20;
    auto add = [&](size_t cnt) { bytes += ((cnt * 4 + 255) & ~size_t(255)); };
    for (int nt = 0; nt < 2; ++nt) { add((size_t)n[nt] * S); add((size_t)n[nt] * S); add((size_t)n[nt] * 48); add(n[nt]); }
    add((size_t)n_max * S); add((size_t)n_max * S); add((size_t)n_max * S); add((size_t)tiles * S); add((size_t)n_max * 48); add((size_t)tiles * 48);
    add((size_t)max_B * S); add((size_t)n_kp * S); add((size_t)max_n_rec * S); add((size_t)n_kp * S); add(max_B);
    add(max_B + 1); add(16); add(max_B + 1); add(n_max); add(max_B + 8); add(cap_rk); add(cap_rk); add(max_n_rec + 1); add(n_kp + 1); add(n_kp + 1); add(max_B + 1);
    KPD_TRY(m->ws.reserve(bytes));
    Arena &W = m->ws;
    for (int nt = 0; nt < 2; ++nt) {
        m->s[nt] = W.take<float>((size_t)n[nt] * S); m->s_tmp[nt] = W.take<float>((size_t)n[nt] * S);
        m->v[nt] = W.take<float>((size_t)n[nt] * 48); m->bidx[nt] = W.take<int>(n[nt]);
    }
    m->Psrc = W.take<float>((size_t)n_max * S); m->Pdst = W.take<float>((size_t)n_max * S);
    m->ms_main = W.take<float>((size_t)n_max * S); m->ms_cont = W.take<float>((size_t)tiles * S);
    m->mv_main = W.take<float>((size_t)n_max * 48); m->mv_cont = W.take<float>((size_t)tiles * 48);
    m->gmean = W.take<float>((size_t)max_B * S); m->kp_emb = W.take<float>((size_t)n_kp * S);
    m->ft_src = W.take<float>((size_t)max_n_rec * S); m->ft_dst = W.take<float>((size_t)n_kp * S); m->z = W.take<float>(max_B);
    m->kp_ptr = W.take<int>(max_B + 1); m->meta = W.take<int>(16); m->off_tmp = W.take<int>(max_B + 1); m->deg_tmp = W.take<int>(n_max);
    m->rad_tmp = W.take<int>(max_B + 8);
    m->xm_src = W.take<int>(cap_rk); m->xm_dst = W.take<int>(cap_rk); m->xm_rowptr = W.take<int>(max_n_rec + 1);
    m->rk_rowptr = W.take<int>(n_kp + 1); m->kk_rowptr = W.take<int>(n_kp + 1); m->kk_off = W.take<int>(max_B + 1);
    KPD_REQUIRE(m->kk_off != nullptr, KPD_ERR_HIP, "recenc workspace arena too small (internal sizing error)");
    m->cap_B = max_B; m->cap_rec = max_n_rec; m->cap_rr = max_n_rr; m->cap_maxrec = max_rec_pg;
    return KPD_OK;
}

static kpd_status run_conv(kpd_recenc *m, int et, int n_src, int n_dst, int n_edges_cap, const int *e_src, const int *e_dst,
                           const int *rowptr, const float *x_src, const float *x_dst, std::vector<HostGvp> &msg,
                           std::vector<HostGvp> &upd, float *const *ln, bool use_dst, float rbf_dmax, hipStream_t st) {
    const kpd_recenc_config &c = m->cfg;
    const int S = m->S;
    const int snt = (et == 2) ? 0 : 0, dnt = (et == 2) ? 1 : 0;      // et 0: rec->rec ("ll" slot), et 2: rec->kp ("lk" slot)
    GvpProjArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.S = S;
    pa.s[0] = m->s[snt]; pa.n[0] = n_src; pa.wp[0] = msg[0].wproj; pa.b[0] = msg[0].bproj; pa.P[0] = m->Psrc;
    pa.tiles_first[0] = 0; pa.tiles_first[1] = cdiv(n_src, TM);
    pa.n_slots = 1;
    if (use_dst) {
        pa.s[1] = m->s[dnt]; pa.n[1] = n_dst; pa.wp[1] = msg[0].wproj_dst; pa.b[1] = nullptr; pa.P[1] = m->Pdst;
        pa.tiles_first[2] = pa.tiles_first[1] + cdiv(n_dst, TM);
        pa.n_slots = 2;
    }
    KPD_TRY(launch_gvp_proj(pa, st));

    GvpEdgeArgs ea;
    memset(&ea, 0, sizeof(ea));
    ea.meta = m->meta;
    ea.x[0] = x_src; ea.x[1] = x_dst; ea.v[0] = m->v[0]; ea.v[1] = m->v[1];
    if (et == 0) { ea.x[1] = x_src; }
    ea.n_gvps = c.n_message_gvps; ea.S = S; ea.rbf_dmax = rbf_dmax; ea.use_dst = use_dst ? 1 : 0;
    ea.src[et] = e_src; ea.dst[et] = e_dst; ea.Psrc[et] = m->Psrc; ea.Pdst[et] = m->Pdst;
    for (int j = 0; j < c.n_message_gvps; ++j) ea.g[et][j] = msg[j].dev();
    ea.ms_main[et] = m->ms_main; ea.ms_cont[et] = m->ms_cont; ea.mv_main[et] = m->mv_main; ea.mv_cont[et] = m->mv_cont;
    KPD_TRY(launch_gvp_edge(ea, cdiv(std::max(n_edges_cap, 1), TM), st));

    GvpNodePair np;
    memset(&np, 0, sizeof(np));
    GvpNodeArgs &na = np.nt[dnt];
    na.n = n_dst; na.s = m->s[dnt]; na.v = m->v[dnt]; na.s_tmp = m->s_tmp[dnt]; na.bidx = m->bidx[dnt];
    na.mean = c.message_norm_mode == 1;
    na.z = c.message_norm_mode == 2 ? m->z : nullptr;
    na.norm_const = c.message_norm_mode == 0 ? c.message_norm : 1.0f;
    na.n_in = 1; na.rowptr[0] = rowptr;
    na.ms_main[0] = m->ms_main; na.ms_cont[0] = m->ms_cont; na.mv_main[0] = m->mv_main; na.mv_cont[0] = m->mv_cont;
    na.ln1_w = ln[0]; na.ln1_b = ln[1]; na.ln2_w = ln[2]; na.ln2_b = ln[3];
    na.n_gvps = c.n_update_gvps; na.S = S;
    na.ln_inv_n = 1.0f / (float)S; na.ln_pad = 0.0f;
    na.vn_inv_n = 1.0f / (float)c.vector_size; na.vn_pad = (float)(GV - c.vector_size);      // vector half of GVPLayerNorm over the model's channels
    for (int j = 0; j < c.n_update_gvps; ++j) na.g[j] = upd[j].dev();
    np.tiles0 = dnt == 0 ? cdiv(n_dst, TM) : 0;
    KPD_TRY(launch_gvp_node(np, st));
    return KPD_OK;
}

extern "C" kpd_status kpd_recenc_forward(kpd_recenc *m, const kpd_rec_batch *bt, const kpd_rec_out *out, void *stream) {
    KPD_REQUIRE(m && bt && out, KPD_ERR_INVALID, "null argument");
    KPD_REQUIRE(m->committed, KPD_ERR_STATE, "kpd_recenc_forward before kpd_recenc_commit");
    KPD_REQUIRE(bt->B >= 1 && bt->n_rec >= 1 && bt->rec_ptr && bt->rec_x && bt->rec_h && bt->rr_rowptr, KPD_ERR_INVALID, "bad batch");
    KPD_REQUIRE(bt->B <= m->cap_B && bt->n_rec <= m->cap_rec && bt->n_rr <= m->cap_rr && bt->max_rec <= m->cap_maxrec,
                KPD_ERR_CAPACITY, "batch exceeds reserved workspace (call kpd_recenc_reserve)");
    KPD_REQUIRE(out->kp_x && out->kp_h && out->kp_v && out->rk_src && out->rk_dst && out->kk_src && out->kk_dst &&
                    out->kk_per_graph && out->counts, KPD_ERR_INVALID, "output buffers missing");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const kpd_recenc_config &c = m->cfg;
    const int S = m->S, K = c.n_keypoints, B = bt->B, n_rec = bt->n_rec, n_kp = B * K;
    KPD_REQUIRE(out->cap_kk >= (long)n_kp * std::min(K - 1, 100), KPD_ERR_CAPACITY, "cap_kk=%d too small", out->cap_kk);

    KPD_TRY(launch_node_graph_index(bt->rec_ptr, B, n_rec, m->bidx[0], st));
    hipLaunchKernelGGL(k_iota_scaled, dim3(cdiv(B + 1, 256)), dim3(256), 0, st, m->kp_ptr, B + 1, K);
    KPD_LAUNCH_CHECK();
    KPD_TRY(launch_node_graph_index(m->kp_ptr, B, n_kp, m->bidx[1], st));
    hipLaunchKernelGGL(k_rec_embed, dim3(cdiv(n_rec, REMB_NODES)), dim3(256), 0, st, bt->rec_h, n_rec, c.in_scalar_size, m->emb_W0,
                       m->emb_b0, m->emb_W1t, m->emb_b1, m->emb_lw, m->emb_lb, S, m->s[0]);
    KPD_LAUNCH_CHECK();
    KPD_HIP(hipMemsetAsync(m->v[0], 0, (size_t)n_rec * 48 * 4, st));                               // :225

    // rec-rec convolutions (:240-254)
    if (c.message_norm_mode == 2) {
        hipLaunchKernelGGL(k_z_indegree, dim3(cdiv(B, 256)), dim3(256), 0, st, bt->rr_rowptr, bt->rec_ptr, B, m->z);
        KPD_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_meta_single, dim3(1), dim3(64), 0, st, 0, static_cast<const int *>(nullptr), bt->n_rr, m->meta);
    KPD_LAUNCH_CHECK();
    for (int i = 0; i < c.n_rr_convs; ++i) {
        float *ln[4] = {m->rr_ln1w[i], m->rr_ln1b[i], m->rr_ln2w[i], m->rr_ln2b[i]};
        KPD_TRY(run_conv(m, 0, n_rec, n_rec, bt->n_rr, bt->rr_src, bt->rr_dst, bt->rr_rowptr, bt->rec_x, bt->rec_x, m->rr_msg[i],
                         m->rr_upd[i], ln, false, c.rr_cutoff, st));
    }

    // keypoint positions (:40-93)
    hipLaunchKernelGGL(k_graph_mean, dim3(B), dim3(256), 0, st, m->s[0], bt->rec_ptr, S, m->gmean);
    KPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_kp_embed, dim3(B), dim3(256), 0, st, m->gmean, m->kpe_W, m->kpe_b, m->kpe_lw, m->kpe_lb, S, S * K, m->kp_emb);
    KPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_linear_rows, dim3(cdiv(n_rec, 4)), dim3(256), 0, st, m->s[0], n_rec, S, m->src_Wt, m->ft_src);
    KPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_linear_rows, dim3(cdiv(n_kp, 4)), dim3(256), 0, st, m->kp_emb, n_kp, S, m->dst_Wt, m->ft_dst);
    KPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_kp_attention, dim3(n_kp), dim3(256), 0, st, m->ft_src, m->ft_dst, bt->rec_x, bt->rec_ptr, K, S, out->kp_x);
    KPD_LAUNCH_CHECK();
    KPD_HIP(hipMemsetAsync(m->s[1], 0, (size_t)n_kp * S * 4, st));                                 // :90-91
    KPD_HIP(hipMemsetAsync(m->v[1], 0, (size_t)n_kp * 48 * 4, st));

    // rec -> kp edges (:297-321), kNN or radius (at most 10 per keypoint, index order): kp-major list = rk (src rec, dst kp),
    // dst-sorted; off_tmp[B] = E_rk either way
    if (c.k_closest > 0)
        KPD_TRY(launch_knn_bipartite(bt->rec_x, bt->rec_ptr, n_rec, bt->max_rec, out->kp_x, m->kp_ptr, n_kp, K, B, c.k_closest,
                                     m->off_tmp, m->xm_src, m->xm_dst, m->xm_rowptr, out->rk_src, out->rk_dst, m->rk_rowptr, st));
    else
        KPD_TRY(launch_radius_bipartite(bt->rec_x, bt->rec_ptr, n_rec, bt->max_rec, out->kp_x, m->kp_ptr, n_kp, K, B, c.kp_rad, 10,
                                        m->rad_tmp, m->rad_tmp + B, m->off_tmp, m->xm_src, m->xm_dst, m->xm_rowptr, out->rk_src,
                                        out->rk_dst, m->rk_rowptr, st));
    if (c.message_norm_mode == 2) {
        hipLaunchKernelGGL(k_z_indegree, dim3(cdiv(B, 256)), dim3(256), 0, st, m->rk_rowptr, m->kp_ptr, B, m->z);   // :266-269
        KPD_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_meta_single, dim3(1), dim3(64), 0, st, 2, m->off_tmp + B, 0, m->meta);
    KPD_LAUNCH_CHECK();
    for (int i = 0; i < c.n_rk_convs; ++i) {
        float *ln[4] = {m->rk_ln1w[i], m->rk_ln1b[i], m->rk_ln2w[i], m->rk_ln2b[i]};
        KPD_TRY(run_conv(m, 2, n_rec, n_kp, n_kp * std::min(rk_per_kp(c), bt->max_rec), out->rk_src, out->rk_dst, m->rk_rowptr, bt->rec_x, out->kp_x,
                         m->rk_msg[i], m->rk_upd[i], ln, i != 0, c.rk_cutoff, st));
    }
    KPD_HIP(hipMemcpyAsync(out->kp_h, m->s[1], (size_t)n_kp * S * 4, hipMemcpyDeviceToDevice, st));
    // kp v_0 [n_kp][vector_size][3]: the leading channels of the 16-channel rows
    if (c.vector_size == GV) KPD_HIP(hipMemcpyAsync(out->kp_v, m->v[1], (size_t)n_kp * 48 * 4, hipMemcpyDeviceToDevice, st));
    else KPD_HIP(hipMemcpy2DAsync(out->kp_v, (size_t)c.vector_size * 12, m->v[1], 48 * 4, (size_t)c.vector_size * 12, n_kp, hipMemcpyDeviceToDevice, st));

    // keypoint-keypoint radius graph (:285-292); counts = {E_kk, E_rk}
    KPD_TRY(launch_radius_graph(out->kp_x, m->kp_ptr, B, n_kp, K, c.kk_cutoff, 100, out->cap_kk, out->kk_src, out->kk_dst,
                                m->kk_rowptr, out->kk_per_graph, m->deg_tmp, m->kk_off, m->off_tmp, out->counts, st));
    return KPD_OK;
}
