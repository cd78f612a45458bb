// Training path of the GVP keypoint receptor encoder: forward with saved node states + backward (SURVEY.md 8(f) item 2 for
// row a8).  Gradients of ReceptorEncoderGVP.forward (models/receptor_encoder_gvp.py:212-294) -- scalar embedding (:158-164,
// :221-222), the rec-rec GVPEdgeConv stack (models/gvp.py:249-341), KeypointInitializer (:40-93: graph-mean feature ->
// keypoint embedding -> attention-pooled keypoint positions), the rec-kp GVPEdgeConv stack with destination features from
// the second convolution on (:194-197, gvp.py:323-337) -- with respect to every parameter, given the gradients of the three
// outputs the denoiser and the encoder loss consume: keypoint positions, scalars and vectors.
//
// Same formulation as gvp_train.hip (whose generic GVP / GVPLayerNorm routines it shares, gvp_train_core.h): parameters in place
// in the reference layout, dense products through sgemm.hip on the caller's stream, deterministic segmented sums (no float
// atomics), node-sized state kept per convolution, edge activations recomputed one convolution at a time.  The encoder runs
// once per batch on ~170 k rr edges and ~13 k rk edges (B = 64), a few per cent of a training step: the message inputs
// [s_src | rbf | s_dst] and [x_diff | v_src | v_dst] are simply materialised per edge instead of splitting the first Linear.
// Receptor positions are data; keypoint positions are a function of the parameters (attention pooling) and receive the
// gradient that reaches them through the geometry of the rk edges, the denoiser and the optimal-transport encoder loss.
// The rec->kp edge list (kNN / radius) and the kk radius graph are rebuilt from the positions and are not differentiable,
// as in the reference (torch_cluster ops).
#include "gvp_train_core.h"
#include "rec_kernels.h"

namespace kpd {
namespace {

constexpr int VHE = 33;       // widest vector block of a message GVP: [x_diff | 16 v_src | 16 v_dst]

// message scalars [E, si]: [s_src[src] (S) | rbf (16) | s_dst[dst] (S, only with destination features)]  (gvp.py:325-337)
__global__ void k_enc_sin(const float *__restrict__ s_src, const int *__restrict__ src, const float *__restrict__ rbf,
                          const float *__restrict__ s_dst, const int *__restrict__ dst, long long total, int S, int si,
                          float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int e = (int)(i / si), c = (int)(i - (long long)e * si);
    out[i] = c < S ? s_src[(size_t)src[e] * S + c] : c < S + RBF ? rbf[(size_t)e * RBF + c - S] : s_dst[(size_t)dst[e] * S + c - S - RBF];
}

// message vectors [E, 3, vi]: channel 0 = unit edge vector, 1..16 = v_src[src], 17..32 = v_dst[dst] (vi = 33 only)
__global__ void k_enc_vin(const float *__restrict__ unit, const float *__restrict__ v_src, const int *__restrict__ src,
                          const float *__restrict__ v_dst, const int *__restrict__ dst, long long total, int vi, float *__restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int ch = (int)(i % vi);
    const long long ec = i / vi;
    const int e = (int)(ec / 3), c = (int)(ec - 3LL * e);
    out[i] = ch == 0 ? unit[3 * e + c] : ch <= VC ? v_src[((size_t)src[e] * 3 + c) * VC + ch - 1] : v_dst[((size_t)dst[e] * 3 + c) * VC + ch - 1 - VC];
}

// message_norm == 0: z[graph] = edges into the graph's destination nodes / destination nodes (receptor_encoder_gvp.py:243-246, :266-269)
__global__ void k_enc_z(const int *__restrict__ rowptr, const int *__restrict__ ptr, int B, float *__restrict__ z) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) z[b] = (float)(rowptr[ptr[b + 1]] - rowptr[ptr[b]]) / (float)(ptr[b + 1] - ptr[b]);
}

// g[r, :] += dmean[graph(r), :] / n_graph   (backward of dgl.readout_nodes mean)
__global__ void k_mean_bwd(const float *__restrict__ dmean, const int *__restrict__ bidx, const int *__restrict__ ptr, long long total, int S,
                           float *__restrict__ g) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int r = (int)(i / S), c = (int)(i - (long long)r * S), b = bidx[r];
    g[i] += dmean[(size_t)b * S + c] / (float)(ptr[b + 1] - ptr[b]);
}

// KeypointInitializer attention (receptor_encoder_gvp.py:57-87): one workgroup per keypoint over the receptor atoms of its graph.
// w[r * K + k] = exp(<ft_src[r], ft_dst[kp]> / sqrt(S)) / sum over the graph (no max-subtraction, as upstream); kp_x = sum w x_r
__global__ __launch_bounds__(256) void k_att_fwd(const float *__restrict__ ft_src, const float *__restrict__ ft_dst, const float *__restrict__ rec_x,
                                                 const int *__restrict__ rec_ptr, int K, int S, float *__restrict__ w, float *__restrict__ kp_x) {
    __shared__ float s_q[256];
    __shared__ float s_part[256][4];
    const int kp = blockIdx.x, g = kp / K, k = kp - g * K, tid = threadIdx.x;
    if (tid < S) s_q[tid] = ft_dst[(size_t)kp * S + tid];
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)S);
    float a_sum = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
    for (int r = rec_ptr[g] + tid; r < rec_ptr[g + 1]; r += 256) {
        const float *f = ft_src + (size_t)r * S;
        float dot = 0.0f;
        for (int j = 0; j < S; ++j) dot = fmaf(f[j], s_q[j], dot);
        const float a = expf(dot * scale);
        w[(size_t)r * K + k] = a;
        a_sum += a;
        ax = fmaf(a, rec_x[(size_t)r * 3], ax);
        ay = fmaf(a, rec_x[(size_t)r * 3 + 1], ay);
        az = fmaf(a, rec_x[(size_t)r * 3 + 2], az);
    }
    s_part[tid][0] = a_sum; s_part[tid][1] = ax; s_part[tid][2] = ay; s_part[tid][3] = az;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (tid < o)
#pragma unroll
            for (int c = 0; c < 4; ++c) s_part[tid][c] += s_part[tid + o][c];
        __syncthreads();
    }
    const float inv = 1.0f / s_part[0][0];
    for (int r = rec_ptr[g] + tid; r < rec_ptr[g + 1]; r += 256) w[(size_t)r * K + k] *= inv;
    if (tid < 3) kp_x[(size_t)kp * 3 + tid] = s_part[0][1 + tid] * inv;
}

// d<ft_src[r], ft_dst[kp]> = w (<dkp_x, x_r> - <dkp_x, kp_x>) / sqrt(S), written over w (softmax backward of kp_x = sum w x)
__global__ __launch_bounds__(256) void k_att_bwd_logits(float *__restrict__ w, const float *__restrict__ rec_x, const int *__restrict__ rec_ptr,
                                                        int K, int S, const float *__restrict__ dkp_x, const float *__restrict__ kp_x) {
    const int kp = blockIdx.x, g = kp / K, k = kp - g * K;
    const float dx = dkp_x[(size_t)kp * 3], dy = dkp_x[(size_t)kp * 3 + 1], dz = dkp_x[(size_t)kp * 3 + 2];
    const float base = dx * kp_x[(size_t)kp * 3] + dy * kp_x[(size_t)kp * 3 + 1] + dz * kp_x[(size_t)kp * 3 + 2];
    const float scale = 1.0f / sqrtf((float)S);
    for (int r = rec_ptr[g] + threadIdx.x; r < rec_ptr[g + 1]; r += 256) {
        const float dw = dx * rec_x[(size_t)r * 3] + dy * rec_x[(size_t)r * 3 + 1] + dz * rec_x[(size_t)r * 3 + 2];
        w[(size_t)r * K + k] *= (dw - base) * scale;
    }
}

// dft_dst[kp, s] = sum over the graph's receptor atoms of G[r, k] ft_src[r, s]   (ascending r: deterministic)
__global__ __launch_bounds__(256) void k_att_bwd_dst(const float *__restrict__ G, const float *__restrict__ ft_src, const int *__restrict__ rec_ptr,
                                                     int K, int S, float *__restrict__ dft_dst) {
    const int kp = blockIdx.x, g = kp / K, k = kp - g * K, s = threadIdx.x;
    if (s >= S) return;
    float acc = 0.0f;
    for (int r = rec_ptr[g]; r < rec_ptr[g + 1]; ++r) acc = fmaf(G[(size_t)r * K + k], ft_src[(size_t)r * S + s], acc);
    dft_dst[(size_t)kp * S + s] = acc;
}

// dft_src[r, s] = sum over the graph's keypoints of G[r, k] ft_dst[kp_k, s]
__global__ __launch_bounds__(256) void k_att_bwd_src(const float *__restrict__ G, const float *__restrict__ ft_dst, const int *__restrict__ bidx,
                                                     int K, int S, float *__restrict__ dft_src) {
    const int r = blockIdx.x, s = threadIdx.x, g = bidx[r];
    if (s >= S) return;
    float acc = 0.0f;
    for (int k = 0; k < K; ++k) acc = fmaf(G[(size_t)r * K + k], ft_dst[((size_t)g * K + k) * S + s], acc);
    dft_src[(size_t)r * S + s] = acc;
}

}  // namespace
}  // namespace kpd

using namespace kpd;

struct kpd_recenc_trainer : TrainCtx {
    kpd_recenc_config cfg{};
    Arena ws;
    int S = 128;
    int V = VC;                         // the model's vector_size (<= 16: narrower models run zero-padded, train_ops.h WideSet)
    int cap_B = 0, cap_rec = 0, cap_rr = 0, cap_maxrec = 0, cap_rk = 0, cap_R = 0;
    kpd_rec_batch bt{};
    bool have_forward = false;
    int B = 0, n_rec = 0, n_kp = 0, E_rk = 0;
    // graph data
    int *bidx[2] = {nullptr, nullptr}, *kp_ptr = nullptr, *rk_src = nullptr, *rk_dst = nullptr, *rk_rowptr = nullptr;
    int *off_tmp = nullptr, *xm_src = nullptr, *xm_dst = nullptr, *xm_rowptr = nullptr, *rad_tmp = nullptr, *kk_rowptr = nullptr,
        *deg_tmp = nullptr, *kk_off = nullptr, *cursor = nullptr;
    SrcCsr scsr_rr, scsr_rk;
    float *z = nullptr, *scale = nullptr;
    // saved forward state
    float *e_pre0 = nullptr, *e_a0 = nullptr, *e_pre1 = nullptr, *e_a1 = nullptr;
    std::vector<float *> rs, rv, rsa, rva;          // rec state entering rr conv i (i = n_rr: final), pre-LayerNorm sums of conv i
    float *gmean = nullptr, *kpe_pre = nullptr, *kpe_act = nullptr, *kp_emb = nullptr, *ft_src = nullptr, *ft_dst = nullptr, *att = nullptr,
          *kp_x = nullptr;
    std::vector<float *> ks, kv, ksa, kva;          // keypoint state entering rk conv j, pre-LayerNorm sums
    // scratch
    GvpBuf gb[4];
    float *ds[2] = {nullptr, nullptr}, *dV[2] = {nullptr, nullptr}, *dVh = nullptr, *dsh = nullptr, *dgate = nullptr;
    float *unit = nullptr, *rbf = nullptr, *sin = nullptr, *vin = nullptr, *dxe = nullptr;
    float *U = nullptr, *tmp_s = nullptr, *tmp_v = nullptr, *s1 = nullptr, *v1 = nullptr, *sb = nullptr, *vb = nullptr;
    float *grs = nullptr, *grv = nullptr;           // gradient of the FINAL receptor state, then of the state entering each rr conv
    float *gro_s = nullptr, *gro_v = nullptr;       // the gradient of a rr conv's outputs, staged while grs / grv collect that of its inputs
    float *gks[2] = {nullptr, nullptr}, *gkv[2] = {nullptr, nullptr}, *gkx = nullptr;      // keypoint state / position gradients
    float *gn_s = nullptr, *gn_v = nullptr;         // node-sized gradient scratch (message-aggregate gradients)
    float *big = nullptr;                           // [B, S * K] scratch of the keypoint embedding's backward
    float *wsg_pack = nullptr;
    float dropout = 0.0f;
    unsigned long long seed = 0;
};

namespace {

unsigned enc_stream(int conv, int pos, int kind) { return 0x4000u + (unsigned)((conv * 2 + pos) * 2 + kind); }

kpd_status enc_dropout(kpd_recenc_trainer *T, int conv, int pos, int n, const float *s, const float *v, float *so, float *vo) {
    const int S = T->S;
    if (T->dropout <= 0.0f) {
        if (so != s) KPD_HIP(hipMemcpyAsync(so, s, (size_t)n * S * 4, hipMemcpyDeviceToDevice, T->st));
        if (vo != v) KPD_HIP(hipMemcpyAsync(vo, v, (size_t)n * 3 * VC * 4, hipMemcpyDeviceToDevice, T->st));
        return KPD_OK;
    }
    hipLaunchKernelGGL(k_dropout, grid1((long long)n * S), dim3(256), 0, T->st, s, (long long)n, 1, S, S, T->seed, enc_stream(conv, pos, 0),
                       T->dropout, so);
    KPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_dropout, grid1((long long)n * 3 * VC), dim3(256), 0, T->st, v, (long long)n, 3, VC, T->V, T->seed, enc_stream(conv, pos, 1),
                       T->dropout, vo);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// one GVPEdgeConv (gvp.py:249-341) on a dst-sorted edge list
struct Conv {
    std::string prefix;
    int id = 0;                       // dropout stream
    int E = 0, n_src = 0, n_dst = 0, bidx_dst = 0;
    const int *src = nullptr, *dst = nullptr, *rowptr = nullptr;
    const SrcCsr *scsr = nullptr;
    const float *xs = nullptr, *xd = nullptr;
    const float *s_src = nullptr, *v_src = nullptr, *s_dst = nullptr, *v_dst = nullptr;       // dst state = residual input
    float *sa = nullptr, *va = nullptr, *s_out = nullptr, *v_out = nullptr;
    bool use_dst = false;
    float dmax = 0.0f;
    int vi() const { return use_dst ? VHE : VC + 1; }
    int si(int S) const { return S + RBF + (use_dst ? S : 0); }
};

kpd_status conv_scale(kpd_recenc_trainer *T, const Conv &c) {
    hipLaunchKernelGGL(k_msg_scale, grid1(c.n_dst), dim3(256), 0, T->st, c.rowptr, T->z, T->bidx[c.bidx_dst], c.n_dst, T->cfg.message_norm_mode,
                       T->cfg.message_norm, T->scale);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// message function on every edge into gb[0 .. n_message_gvps); leaves unit / rbf / sin / vin for the backward pass
kpd_status conv_message_fwd(kpd_recenc_trainer *T, const Conv &c, GvpP *g0_out) {
    const int E = c.E, S = T->S, nm = T->cfg.n_message_gvps, vi = c.vi(), si = c.si(S);
    hipLaunchKernelGGL(k_gvp_geom, grid1(E), dim3(256), 0, T->st, c.src, c.dst, c.xs, c.xd, E, c.dmax, T->unit, T->rbf);
    KPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_enc_sin, grid1((long long)E * si), dim3(256), 0, T->st, c.s_src, c.src, T->rbf, c.s_dst, c.dst, (long long)E * si, S, si,
                       T->sin);
    KPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_enc_vin, grid1((long long)E * 3 * vi), dim3(256), 0, T->st, T->unit, c.v_src, c.src, c.v_dst, c.dst,
                       (long long)E * 3 * vi, vi, T->vin);
    KPD_LAUNCH_CHECK();
    GvpP g0;
    KPD_TRY(gvp_params(T, c.prefix + ".edge_message.0", vi, VC, si, S, &g0));
    KPD_TRY(gvp_fwd(T, g0, E, T->sin, si, T->vin, T->gb[0], false));
    for (int j = 1; j < nm; ++j) {
        GvpP g;
        KPD_TRY(gvp_params(T, c.prefix + ".edge_message." + std::to_string(j), VC, VC, S, S, &g));
        KPD_TRY(gvp_fwd(T, g, E, T->gb[j - 1].s, S, T->gb[j - 1].V, T->gb[j], false));
    }
    if (g0_out) *g0_out = g0;
    return KPD_OK;
}

kpd_status update_chain_fwd(kpd_recenc_trainer *T, const std::string &prefix, int n, const float *s0, const float *v0) {
    for (int j = 0; j < T->cfg.n_update_gvps; ++j) {
        GvpP g;
        KPD_TRY(gvp_params(T, prefix + "." + std::to_string(j), VC, VC, T->S, T->S, &g));
        KPD_TRY(gvp_fwd(T, g, n, j == 0 ? s0 : T->gb[j - 1].s, T->S, j == 0 ? v0 : T->gb[j - 1].V, T->gb[j], false));
    }
    return KPD_OK;
}

// everything after the aggregation: residual, LayerNorm, update chain, residual, LayerNorm (gvp.py:312-330).
// from_sa: sa / va already hold the pre-LayerNorm sums (backward-pass recomputation)
kpd_status conv_node_fwd(kpd_recenc_trainer *T, const Conv &c, bool write_out) {
    const int n = c.n_dst, S = T->S, nu = T->cfg.n_update_gvps;
    LnP l1, l2;
    KPD_TRY(ln_params(T, c.prefix + ".message_layer_norm", &l1));
    KPD_TRY(ln_params(T, c.prefix + ".update_layer_norm", &l2));
    KPD_TRY(gvp_ln_fwd(T, l1, n, c.sa, c.va, T->s1, T->v1));
    KPD_TRY(update_chain_fwd(T, c.prefix + ".node_update", n, T->s1, T->v1));
    KPD_TRY(enc_dropout(T, c.id, 1, n, T->gb[nu - 1].s, T->gb[nu - 1].V, T->tmp_s, T->tmp_v));
    hipLaunchKernelGGL(k_add, grid1((long long)n * S), dim3(256), 0, T->st, T->s1, T->tmp_s, (long long)n * S, T->sb);
    KPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_add, grid1((long long)n * 3 * VC), dim3(256), 0, T->st, T->v1, T->tmp_v, (long long)n * 3 * VC, T->vb);
    KPD_LAUNCH_CHECK();
    if (write_out) KPD_TRY(gvp_ln_fwd(T, l2, n, T->sb, T->vb, c.s_out, c.v_out));
    return KPD_OK;
}

kpd_status conv_fwd(kpd_recenc_trainer *T, const Conv &c) {
    const int n = c.n_dst, S = T->S, nm = T->cfg.n_message_gvps;
    KPD_HIP(hipMemsetAsync(c.sa, 0, (size_t)n * S * 4, T->st));
    KPD_HIP(hipMemsetAsync(c.va, 0, (size_t)n * 3 * VC * 4, T->st));
    if (c.E > 0) {
        KPD_TRY(conv_message_fwd(T, c, nullptr));
        KPD_TRY(conv_scale(T, c));
        KPD_TRY(segsum(T->st, T->gb[nm - 1].s, S, 0, S, nullptr, c.rowptr, T->scale, 1.0f, true, n, c.sa, S));
        KPD_TRY(segsum(T->st, T->gb[nm - 1].V, 3 * VC, 0, 3 * VC, nullptr, c.rowptr, T->scale, 1.0f, true, n, c.va, 3 * VC));
    }
    KPD_TRY(enc_dropout(T, c.id, 0, n, c.sa, c.va, c.sa, c.va));
    hipLaunchKernelGGL(k_acc, grid1((long long)n * S), dim3(256), 0, T->st, c.sa, c.s_dst, (long long)n * S);
    KPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_acc, grid1((long long)n * 3 * VC), dim3(256), 0, T->st, c.va, c.v_dst, (long long)n * 3 * VC);
    KPD_LAUNCH_CHECK();
    return conv_node_fwd(T, c, true);
}

// Backward of one convolution.  In: g_out_s / g_out_v = gradient of (s_out, v_out).  Out (all ACCUMULATED into):
// g_dst_s / g_dst_v (gradient of the destination state: residual + destination features), g_src_s / g_src_v (gradient of the
// source state; may alias g_dst_* when source and destination are the same nodes), g_xd (gradient of the destination positions,
// null when they are data).
kpd_status conv_bwd(kpd_recenc_trainer *T, const Conv &c, const float *g_out_s, const float *g_out_v, float *g_dst_s, float *g_dst_v,
                    float *g_src_s, float *g_src_v, float *g_xd) {
    const int n = c.n_dst, S = T->S, nm = T->cfg.n_message_gvps, nu = T->cfg.n_update_gvps, E = c.E, vi = c.vi(), si = c.si(S);
    LnP l1, l2;
    KPD_TRY(ln_params(T, c.prefix + ".message_layer_norm", &l1));
    KPD_TRY(ln_params(T, c.prefix + ".update_layer_norm", &l2));
    KPD_TRY(conv_node_fwd(T, c, false));                                   // s1, v1, update chain, sb, vb again
    KPD_TRY(gvp_ln_bwd(T, l2, n, T->sb, T->vb, g_out_s, g_out_v, T->ds[0], T->dV[0]));
    // residual: d(s1, v1) gets d(sb, vb) directly and through the update chain (behind its dropout mask)
    KPD_HIP(hipMemcpyAsync(T->gn_s, T->ds[0], (size_t)n * S * 4, hipMemcpyDeviceToDevice, T->st));
    KPD_HIP(hipMemcpyAsync(T->gn_v, T->dV[0], (size_t)n * 3 * VC * 4, hipMemcpyDeviceToDevice, T->st));
    KPD_TRY(enc_dropout(T, c.id, 1, n, T->ds[0], T->dV[0], T->ds[0], T->dV[0]));
    for (int j = nu - 1; j >= 0; --j) {
        GvpP g;
        KPD_TRY(gvp_params(T, c.prefix + ".node_update." + std::to_string(j), VC, VC, S, S, &g));
        KPD_TRY(gvp_bwd(T, g, n, j == 0 ? T->s1 : T->gb[j - 1].s, S, j == 0 ? T->v1 : T->gb[j - 1].V, T->gb[j], false, T->ds[0], T->dV[0],
                        T->ds[1], T->dV[1]));
        std::swap(T->ds[0], T->ds[1]);
        std::swap(T->dV[0], T->dV[1]);
    }
    hipLaunchKernelGGL(k_acc, grid1((long long)n * S), dim3(256), 0, T->st, T->gn_s, T->ds[0], (long long)n * S);
    KPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_acc, grid1((long long)n * 3 * VC), dim3(256), 0, T->st, T->gn_v, T->dV[0], (long long)n * 3 * VC);
    KPD_LAUNCH_CHECK();
    // first LayerNorm at (sa, va): the gradient of the pre-norm sums is the gradient of the residual input ...
    KPD_TRY(gvp_ln_bwd(T, l1, n, c.sa, c.va, T->gn_s, T->gn_v, T->gn_s, T->gn_v));
    hipLaunchKernelGGL(k_acc, grid1((long long)n * S), dim3(256), 0, T->st, g_dst_s, T->gn_s, (long long)n * S);
    KPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_acc, grid1((long long)n * 3 * VC), dim3(256), 0, T->st, g_dst_v, T->gn_v, (long long)n * 3 * VC);
    KPD_LAUNCH_CHECK();
    if (E == 0) return KPD_OK;
    // ... and, behind the message dropout mask, of the aggregated messages
    KPD_TRY(enc_dropout(T, c.id, 0, n, T->gn_s, T->gn_v, T->gn_s, T->gn_v));
    GvpP g0;
    KPD_TRY(conv_message_fwd(T, c, &g0));
    KPD_TRY(conv_scale(T, c));
    KPD_TRY(gather_rows(T->st, T->gn_s, c.dst, T->scale, E, S, T->ds[0]));
    KPD_TRY(gather_rows(T->st, T->gn_v, c.dst, T->scale, E, 3 * VC, T->dV[0]));
    for (int j = nm - 1; j >= 1; --j) {
        GvpP g;
        KPD_TRY(gvp_params(T, c.prefix + ".edge_message." + std::to_string(j), VC, VC, S, S, &g));
        KPD_TRY(gvp_bwd(T, g, E, T->gb[j - 1].s, S, T->gb[j - 1].V, T->gb[j], false, T->ds[0], T->dV[0], T->ds[1], T->dV[1]));
        std::swap(T->ds[0], T->ds[1]);
        std::swap(T->dV[0], T->dV[1]);
    }
    // first message GVP: ds[1] [E, si] = gradient of [s_src | rbf | s_dst], dV[1] [E, 3, vi] = gradient of [x_diff | v_src | v_dst]
    KPD_TRY(gvp_bwd(T, g0, E, T->sin, si, T->vin, T->gb[0], false, T->ds[0], T->dV[0], T->ds[1], T->dV[1]));
    // source side: sums over the out-edges of every source node, ascending edge order
    KPD_TRY(segsum(T->st, T->ds[1], si, 0, S, c.scsr->perm, c.scsr->rowptr, nullptr, 1.0f, true, c.n_src, g_src_s, S));
    for (int cc = 0; cc < 3; ++cc) {
        KPD_TRY(segsum(T->st, T->dV[1], 3 * vi, cc * vi + 1, VC, c.scsr->perm, c.scsr->rowptr, nullptr, 1.0f, true, c.n_src, g_src_v + cc * VC, 3 * VC));
    }
    if (c.use_dst) {                  // destination features: the edge list is dst-sorted, so the in-edges of a node are contiguous
        KPD_TRY(segsum(T->st, T->ds[1], si, S + RBF, S, nullptr, c.rowptr, nullptr, 1.0f, true, n, g_dst_s, S));
        for (int cc = 0; cc < 3; ++cc) {
            KPD_TRY(segsum(T->st, T->dV[1], 3 * vi, cc * vi + 1 + VC, VC, nullptr, c.rowptr, nullptr, 1.0f, true, n, g_dst_v + cc * VC, 3 * VC));
        }
    }
    if (g_xd) {                       // geometry: d rbf = columns S .. S + 16 of d sin, d unit = channel 0 of d vin
        hipLaunchKernelGGL(k_copy_rows, grid1((long long)E * RBF), dim3(256), 0, T->st, T->ds[1] + S, si, T->dsh, RBF, (long long)E * RBF, RBF);
        KPD_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_gvp_geom_bwd, grid1(E), dim3(256), 0, T->st, c.src, c.dst, c.xs, c.xd, E, c.dmax, T->rbf, T->dsh, T->dV[1], vi, T->dxe);
        KPD_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_seg3, grid1(n), dim3(256), 0, T->st, T->dxe, (const int *)nullptr, c.rowptr, n, -1.0f, g_xd);
        KPD_LAUNCH_CHECK();
    }
    return KPD_OK;
}

Conv rr_conv(kpd_recenc_trainer *T, int i) {
    Conv c;
    c.prefix = "rr_conv_layers." + std::to_string(i);
    c.id = i;
    c.E = T->bt.n_rr; c.n_src = c.n_dst = T->n_rec; c.bidx_dst = 0;
    c.src = T->bt.rr_src; c.dst = T->bt.rr_dst; c.rowptr = T->bt.rr_rowptr; c.scsr = &T->scsr_rr;
    c.xs = c.xd = T->bt.rec_x;
    c.s_src = c.s_dst = T->rs[i]; c.v_src = c.v_dst = T->rv[i];
    c.sa = T->rsa[i]; c.va = T->rva[i]; c.s_out = T->rs[i + 1]; c.v_out = T->rv[i + 1];
    c.use_dst = false; c.dmax = T->cfg.rr_cutoff;
    return c;
}

Conv rk_conv(kpd_recenc_trainer *T, int j) {
    const int R = T->cfg.n_rr_convs;
    Conv c;
    c.prefix = "rk_conv_layers." + std::to_string(j);
    c.id = R + j;
    c.E = T->E_rk; c.n_src = T->n_rec; c.n_dst = T->n_kp; c.bidx_dst = 1;
    c.src = T->rk_src; c.dst = T->rk_dst; c.rowptr = T->rk_rowptr; c.scsr = &T->scsr_rk;
    c.xs = T->bt.rec_x; c.xd = T->kp_x;
    c.s_src = T->rs[R]; c.v_src = T->rv[R]; c.s_dst = T->ks[j]; c.v_dst = T->kv[j];
    c.sa = T->ksa[j]; c.va = T->kva[j]; c.s_out = T->ks[j + 1]; c.v_out = T->kv[j + 1];
    c.use_dst = j != 0; c.dmax = T->cfg.rk_cutoff;
    return c;
}

int rk_per_kp(const kpd_recenc_config &c) { return c.k_closest > 0 ? c.k_closest : 10; }

// column sums of a [rows, cols] matrix with cols beyond the 512 columns one k_colsum pass takes
kpd_status colsum_wide(kpd_recenc_trainer *T, int rows, int cols, const float *A, float *y) {
    if (!y) return KPD_OK;
    for (int c0 = 0; c0 < cols; c0 += COLSUM_LD) KPD_TRY(colsum_acc(T, rows, std::min(COLSUM_LD, cols - c0), A + c0, cols, y + c0));
    return KPD_OK;
}

}  // namespace

extern "C" kpd_status kpd_recenc_trainer_create(const kpd_recenc_config *cfg, kpd_recenc_trainer **out) {
    KPD_REQUIRE(cfg && out, KPD_ERR_INVALID, "null argument");
    KPD_REQUIRE(cfg->vector_size >= 1 && cfg->vector_size <= VC, KPD_ERR_INVALID, "vector_size=%d outside 1 .. %d", cfg->vector_size, VC);
    KPD_REQUIRE(cfg->out_scalar_size >= 16 && cfg->out_scalar_size <= 256, KPD_ERR_INVALID, "out_scalar_size=%d (16..256)", cfg->out_scalar_size);
    KPD_REQUIRE(cfg->in_scalar_size >= 1 && cfg->in_scalar_size <= 256, KPD_ERR_INVALID, "in_scalar_size=%d", cfg->in_scalar_size);
    KPD_REQUIRE(cfg->n_rr_convs >= 0 && cfg->n_rr_convs <= 16 && cfg->n_rk_convs >= 1 && cfg->n_rk_convs <= 16 && cfg->n_message_gvps >= 1 &&
                    cfg->n_message_gvps <= 4 && cfg->n_update_gvps >= 1 && cfg->n_update_gvps <= 4,
                KPD_ERR_INVALID, "convs %d/%d gvps %d/%d", cfg->n_rr_convs, cfg->n_rk_convs, cfg->n_message_gvps, cfg->n_update_gvps);
    KPD_REQUIRE(cfg->message_norm_mode >= 0 && cfg->message_norm_mode <= 2 && (cfg->message_norm_mode != 0 || cfg->message_norm > 0.0f),
                KPD_ERR_INVALID, "message_norm mode %d value %g", cfg->message_norm_mode, (double)cfg->message_norm);
    KPD_REQUIRE((cfg->k_closest >= 1 && cfg->k_closest <= 16) || (cfg->k_closest == 0 && cfg->kp_rad > 0.0f), KPD_ERR_INVALID,
                "k_closest=%d kp_rad=%g", cfg->k_closest, (double)cfg->kp_rad);
    KPD_REQUIRE(cfg->n_keypoints >= 1 && cfg->n_keypoints <= 256, KPD_ERR_INVALID, "n_keypoints=%d", cfg->n_keypoints);
    kpd_recenc_trainer *T = new kpd_recenc_trainer();
    T->cfg = *cfg;
    T->S = cfg->out_scalar_size;
    T->V = cfg->vector_size;
    *out = T;
    return KPD_OK;
}

extern "C" void kpd_recenc_trainer_destroy(kpd_recenc_trainer *T) {
    if (!T) return;
    T->ws.release();
    T->wide.release();
    T->release_scratch();
    delete T;
}

extern "C" kpd_status kpd_recenc_trainer_bind(kpd_recenc_trainer *T, const char *name, const float *weight, float *grad, const int64_t *shape,
                                              int32_t ndim) {
    KPD_REQUIRE(T && name && shape && (ndim == 1 || ndim == 2), KPD_ERR_INVALID, "bad argument");
    if (T->V != VC && weight) {
        // vector_size < 16: GVP tensors whose axes count vector channels are trained through their 16-channel zero-padded form (as in
        // gvp_train.hip).  The first message GVP of a conv reads [x_diff | v_src] (1 + V channels) or, in the rk convs after the first,
        // [x_diff | v_src | v_dst] (1 + 2 V; models/gvp.py:206-213, 323-337): the destination block keeps its own 16-channel slot.
        const std::vector<std::string> tk = split_name(name);
        if (tk.size() >= 5 && (tk[0] == "rr_conv_layers" || tk[0] == "rk_conv_layers") && (tk[2] == "edge_message" || tk[2] == "node_update")) {
            const int V = T->V;
            const bool msg0 = tk[2] == "edge_message" && tk[3] == "0";
            const bool use_dst = msg0 && tk[0] == "rk_conv_layers" && tk[1] != "0";
            std::vector<AxisSeg> vi = msg0 ? std::vector<AxisSeg>{{V + 1, VC + 1}} : std::vector<AxisSeg>{{V, VC}};
            if (use_dst) vi.push_back({V, VC});
            const int h_ref = msg0 ? (use_dst ? 2 * V + 1 : V + 1) : V, h_wide = msg0 ? (use_dst ? VHE : VC + 1) : VC;
            const std::vector<AxisSeg> h = {{h_ref, h_wide}}, vo = {{V, VC}};
            const std::string &leaf = tk[4];
            std::vector<AxisSeg> rows, cols;
            if (leaf == "Wh") { rows = vi; cols = h; }
            else if (leaf == "Wu") { rows = h; cols = vo; }
            else if (leaf == "to_feats_out" && tk.back() == "weight" && ndim == 2 && shape[1] > h_ref) {
                rows = {{(int)shape[0], (int)shape[0]}};
                cols = {{(int)shape[1] - h_ref, (int)shape[1] - h_ref}, h[0]};
            } else if (leaf == "scalar_to_vector_gates") {
                rows = vo;
                if (tk.back() == "weight" && ndim == 2) cols = {{(int)shape[1], (int)shape[1]}};
            }
            if (!rows.empty()) return bind_wide(T, name, weight, grad, shape, ndim, rows, cols);
        }
    }
    Param p;
    p.w = weight;
    p.g = grad;
    p.rows = (int)shape[0];
    p.cols = ndim == 2 ? (int)shape[1] : 1;
    T->params[name] = p;
    return KPD_OK;
}

extern "C" kpd_status kpd_recenc_trainer_set_dropout(kpd_recenc_trainer *T, float rate, uint64_t seed) {
    KPD_REQUIRE(T, KPD_ERR_INVALID, "null trainer");
    KPD_REQUIRE(rate >= 0.0f && rate < 1.0f, KPD_ERR_INVALID, "dropout rate %g outside [0, 1)", (double)rate);
    T->have_forward = false;
    T->dropout = rate;
    T->seed = seed;
    return KPD_OK;
}

extern "C" kpd_status kpd_recenc_trainer_reserve(kpd_recenc_trainer *T, int32_t max_B, int32_t max_n_rec, int32_t max_n_rr, int32_t max_rec_pg) {
    KPD_REQUIRE(T, KPD_ERR_INVALID, "null trainer");
    KPD_REQUIRE(max_B >= 1 && max_n_rec >= 1 && max_n_rr >= 0 && max_rec_pg >= 1, KPD_ERR_INVALID, "bad capacities");
    if (max_B <= T->cap_B && max_n_rec <= T->cap_rec && max_n_rr <= T->cap_rr && max_rec_pg <= T->cap_maxrec) return KPD_OK;
    const kpd_recenc_config &c = T->cfg;
    max_B = std::max(max_B, T->cap_B); max_n_rec = std::max(max_n_rec, T->cap_rec); max_n_rr = std::max(max_n_rr, T->cap_rr);
    max_rec_pg = std::max(max_rec_pg, T->cap_maxrec);
    const int S = T->S, K = c.n_keypoints, n_kp = max_B * K, Rr = c.n_rr_convs, Rk = c.n_rk_convs;
    const int cap_rk = std::max(n_kp * std::min(rk_per_kp(c), max_rec_pg), 1);
    const int R = std::max(std::max(std::max<int>(max_n_rr, 1), cap_rk), std::max(max_n_rec, n_kp));
    const int SI = 2 * S + RBF;
    T->rs.assign(Rr + 1, nullptr); T->rv.assign(Rr + 1, nullptr); T->rsa.assign(Rr, nullptr); T->rva.assign(Rr, nullptr);
    T->ks.assign(Rk + 1, nullptr); T->kv.assign(Rk + 1, nullptr); T->ksa.assign(Rk, nullptr); T->kva.assign(Rk, nullptr);
    T->ws.release();
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
        const size_t nr = max_n_rec, nk = n_kp, N = std::max(nr, nk);
        F(T->e_pre0, nr * S); F(T->e_a0, nr * S); F(T->e_pre1, nr * S); F(T->e_a1, nr * S);
        for (int i = 0; i <= Rr; ++i) { F(T->rs[i], nr * S); F(T->rv[i], nr * 3 * VC); }
        for (int i = 0; i < Rr; ++i) { F(T->rsa[i], nr * S); F(T->rva[i], nr * 3 * VC); }
        for (int j = 0; j <= Rk; ++j) { F(T->ks[j], nk * S); F(T->kv[j], nk * 3 * VC); }
        for (int j = 0; j < Rk; ++j) { F(T->ksa[j], nk * S); F(T->kva[j], nk * 3 * VC); }
        F(T->gmean, (size_t)max_B * S); F(T->kpe_pre, nk * S); F(T->kpe_act, nk * S); F(T->kp_emb, nk * S); F(T->big, nk * S);
        F(T->ft_src, nr * S); F(T->ft_dst, nk * S); F(T->att, nr * K); F(T->kp_x, nk * 3);
        for (int k = 0; k < 4; ++k) {
            GvpBuf &b = T->gb[k];
            F(b.Vh, (size_t)R * 3 * VHE); F(b.Vu, (size_t)R * 3 * VC); F(b.sh, (size_t)R * VHE); F(b.pre, (size_t)R * S); F(b.s, (size_t)R * S);
            F(b.gate, (size_t)R * VC); F(b.V, (size_t)R * 3 * VC);
        }
        for (int k = 0; k < 2; ++k) { F(T->ds[k], (size_t)R * SI); F(T->dV[k], (size_t)R * 3 * VHE); }
        F(T->dVh, (size_t)R * 3 * VHE); F(T->dsh, (size_t)R * VHE); F(T->dgate, (size_t)R * VC);
        F(T->unit, (size_t)R * 3); F(T->rbf, (size_t)R * RBF); F(T->sin, (size_t)R * SI); F(T->vin, (size_t)R * 3 * VHE); F(T->dxe, (size_t)R * 3);
        F(T->U, N * S); F(T->scale, N); F(T->tmp_s, N * S); F(T->tmp_v, N * 3 * VC); F(T->s1, N * S); F(T->v1, N * 3 * VC);
        F(T->sb, N * S); F(T->vb, N * 3 * VC); F(T->gn_s, N * S); F(T->gn_v, N * 3 * VC);
        F(T->grs, nr * S); F(T->grv, nr * 3 * VC); F(T->gro_s, nr * S); F(T->gro_v, nr * 3 * VC);
        for (int k = 0; k < 2; ++k) { F(T->gks[k], nk * S); F(T->gkv[k], nk * 3 * VC); }
        F(T->gkx, nk * 3); F(T->z, max_B);
        F(T->part, GRAD_PART_FLOATS); F(T->wsg_pack, (size_t)ws_gemm_pack_floats()); F(T->ones, 8);
        F(T->colpart, colpart_floats(R));
        I(T->bidx[0], nr); I(T->bidx[1], nk); I(T->kp_ptr, max_B + 1);
        I(T->rk_src, cap_rk); I(T->rk_dst, cap_rk); I(T->rk_rowptr, nk + 1);
        I(T->off_tmp, max_B + 2); I(T->xm_src, cap_rk); I(T->xm_dst, cap_rk); I(T->xm_rowptr, nr + 1); I(T->rad_tmp, 2 * max_B + 4);
        I(T->cursor, N);
        I(T->scsr_rr.perm, std::max<int>(max_n_rr, 1)); I(T->scsr_rr.rowptr, nr + 1);
        I(T->scsr_rk.perm, cap_rk); I(T->scsr_rk.rowptr, nr + 1);
        I(T->kk_rowptr, nk + 1); I(T->deg_tmp, nk); I(T->kk_off, max_B + 1);
        if (pass == 0) KPD_TRY(T->ws.reserve(bytes + 4096));
    }
    KPD_REQUIRE(T->kk_off != nullptr, KPD_ERR_HIP, "workspace arena too small (internal sizing error)");
    T->part_floats = GRAD_PART_FLOATS;
    T->colpart_blocks = cdiv(R, HEAD_ROWS);
    T->cap_B = max_B; T->cap_rec = max_n_rec; T->cap_rr = max_n_rr; T->cap_maxrec = max_rec_pg; T->cap_rk = cap_rk; T->cap_R = R;
    T->have_forward = false;
    return KPD_OK;
}

extern "C" kpd_status kpd_recenc_trainer_forward(kpd_recenc_trainer *T, const kpd_rec_batch *bt, const kpd_rec_out *out, void *stream) {
    KPD_REQUIRE(T && bt && out, KPD_ERR_INVALID, "null argument");
    KPD_REQUIRE(bt->B >= 1 && bt->n_rec >= 1 && bt->rec_ptr && bt->rec_x && bt->rec_h && bt->rr_rowptr, KPD_ERR_INVALID, "bad batch");
    KPD_REQUIRE(bt->B <= T->cap_B && bt->n_rec <= T->cap_rec && bt->n_rr <= T->cap_rr && bt->max_rec <= T->cap_maxrec, KPD_ERR_CAPACITY,
                "batch exceeds the reserved workspace (call kpd_recenc_trainer_reserve)");
    KPD_REQUIRE(out->kp_x && out->kp_h && out->kp_v && out->rk_src && out->rk_dst && out->kk_src && out->kk_dst && out->kk_per_graph &&
                    out->counts, KPD_ERR_INVALID, "output buffers missing");
    const kpd_recenc_config &c = T->cfg;
    hipStream_t st = static_cast<hipStream_t>(stream);
    T->st = st;
    KPD_TRY(wide_run(T, 0));                   // vector_size < 16: stage the current weights in the engine's widths
    T->bt = *bt;
    const int S = T->S, K = c.n_keypoints, B = bt->B, n_rec = bt->n_rec, n_kp = B * K, F = c.in_scalar_size, Rr = c.n_rr_convs, Rk = c.n_rk_convs;
    T->B = B; T->n_rec = n_rec; T->n_kp = n_kp;
    KPD_REQUIRE(out->cap_kk >= (long)n_kp * std::min(K - 1, 100), KPD_ERR_CAPACITY, "cap_kk=%d too small", out->cap_kk);
    KPD_TRY(launch_node_graph_index(bt->rec_ptr, B, n_rec, T->bidx[0], st));
    KPD_TRY(launch_iota_scaled(T->kp_ptr, B + 1, K, st));
    KPD_TRY(launch_node_graph_index(T->kp_ptr, B, n_kp, T->bidx[1], st));
    KPD_TRY(build_src_csr(T, bt->rr_src, bt->n_rr, n_rec, T->cursor, T->scsr_rr));

    // scalar embedding (:158-164, :221-222): Linear - SiLU - Linear - SiLU - LayerNorm; receptor vectors start at zero (:225)
    {
        Param W0, b0, W1, b1, lw, lb;
        KPD_TRY(param(T, "scalar_embed.0.weight", S, F, &W0)); KPD_TRY(param(T, "scalar_embed.0.bias", S, 1, &b0));
        KPD_TRY(param(T, "scalar_embed.2.weight", S, S, &W1)); KPD_TRY(param(T, "scalar_embed.2.bias", S, 1, &b1));
        KPD_TRY(param(T, "scalar_norm.weight", S, 1, &lw)); KPD_TRY(param(T, "scalar_norm.bias", S, 1, &lb));
        const long long tot = (long long)n_rec * S;
        KPD_TRY(gemm(T, false, true, n_rec, S, F, bt->rec_h, F, W0.w, F, 0.0f, T->e_pre0, S, 1.0f, nullptr, b0.w, T->e_a0));      // bias + SiLU in the epilogue
        KPD_TRY(gemm(T, false, true, n_rec, S, S, T->e_a0, S, W1.w, S, 0.0f, T->e_pre1, S, 1.0f, nullptr, b1.w, T->e_a1));
        hipLaunchKernelGGL(k_ln_fwd, dim3(cdiv(n_rec, 4)), dim3(256), 0, st, T->e_a1, lw.w, lb.w, n_rec, S, T->rs[0]);
        KPD_LAUNCH_CHECK();
        KPD_HIP(hipMemsetAsync(T->rv[0], 0, (size_t)n_rec * 3 * VC * 4, st));
    }
    if (c.message_norm_mode == 2) {
        hipLaunchKernelGGL(k_enc_z, grid1(B), dim3(256), 0, st, bt->rr_rowptr, bt->rec_ptr, B, T->z);
        KPD_LAUNCH_CHECK();
    }
    for (int i = 0; i < Rr; ++i) KPD_TRY(conv_fwd(T, rr_conv(T, i)));

    // keypoint positions (:40-93)
    {
        Param W, b, lw, lb, Ws, Wd;
        KPD_TRY(param(T, "keypoint_initializer.keypoint_embedding.0.weight", S * K, S, &W));
        KPD_TRY(param(T, "keypoint_initializer.keypoint_embedding.0.bias", S * K, 1, &b));
        KPD_TRY(param(T, "keypoint_initializer.keypoint_embedding.2.weight", S * K, 1, &lw));
        KPD_TRY(param(T, "keypoint_initializer.keypoint_embedding.2.bias", S * K, 1, &lb));
        KPD_TRY(param(T, "keypoint_initializer.src_net.weight", S, S, &Ws));
        KPD_TRY(param(T, "keypoint_initializer.dst_net.weight", S, S, &Wd));
        KPD_TRY(launch_graph_mean(T->rs[Rr], bt->rec_ptr, B, S, T->gmean, st));
        const long long tot = (long long)B * S * K;
        KPD_TRY(gemm(T, false, true, B, S * K, S, T->gmean, S, W.w, S, 0.0f, T->kpe_pre, S * K, 1.0f, nullptr, b.w, T->kpe_act));
        hipLaunchKernelGGL(k_ln_fwd, dim3(cdiv(B, 4)), dim3(256), 0, st, T->kpe_act, lw.w, lb.w, B, S * K, T->kp_emb);      // 'b (k d) -> (b k) d'
        KPD_LAUNCH_CHECK();
        KPD_TRY(gemm(T, false, true, n_rec, S, S, T->rs[Rr], S, Ws.w, S, 0.0f, T->ft_src, S));
        KPD_TRY(gemm(T, false, true, n_kp, S, S, T->kp_emb, S, Wd.w, S, 0.0f, T->ft_dst, S));
        hipLaunchKernelGGL(k_att_fwd, dim3(n_kp), dim3(256), 0, st, T->ft_src, T->ft_dst, bt->rec_x, bt->rec_ptr, K, S, T->att, T->kp_x);
        KPD_LAUNCH_CHECK();
        KPD_HIP(hipMemsetAsync(T->ks[0], 0, (size_t)n_kp * S * 4, st));                                 // :90-91
        KPD_HIP(hipMemsetAsync(T->kv[0], 0, (size_t)n_kp * 3 * VC * 4, st));
    }
    // rec -> kp edges (:297-321): kNN, or radius with at most 10 receptor atoms per keypoint in index order; kp-major = dst-sorted
    if (c.k_closest > 0)
        KPD_TRY(launch_knn_bipartite(bt->rec_x, bt->rec_ptr, n_rec, bt->max_rec, T->kp_x, T->kp_ptr, n_kp, K, B, c.k_closest, T->off_tmp,
                                     T->xm_src, T->xm_dst, T->xm_rowptr, T->rk_src, T->rk_dst, T->rk_rowptr, st));
    else
        KPD_TRY(launch_radius_bipartite(bt->rec_x, bt->rec_ptr, n_rec, bt->max_rec, T->kp_x, T->kp_ptr, n_kp, K, B, c.kp_rad, 10, T->rad_tmp,
                                        T->rad_tmp + B, T->off_tmp, T->xm_src, T->xm_dst, T->xm_rowptr, T->rk_src, T->rk_dst, T->rk_rowptr, st));
    int e_rk = 0;
    KPD_HIP(hipMemcpyAsync(&e_rk, T->off_tmp + B, sizeof(int), hipMemcpyDeviceToHost, st));
    KPD_HIP(hipStreamSynchronize(st));
    KPD_REQUIRE(e_rk >= 0 && e_rk <= T->cap_rk, KPD_ERR_CAPACITY, "rk edge list overflow (%d of %d)", e_rk, T->cap_rk);
    T->E_rk = e_rk;
    KPD_TRY(build_src_csr(T, T->rk_src, e_rk, n_rec, T->cursor, T->scsr_rk));
    if (c.message_norm_mode == 2) {
        hipLaunchKernelGGL(k_enc_z, grid1(B), dim3(256), 0, st, T->rk_rowptr, T->kp_ptr, B, T->z);          // :266-269
        KPD_LAUNCH_CHECK();
    }
    for (int j = 0; j < Rk; ++j) KPD_TRY(conv_fwd(T, rk_conv(T, j)));

    KPD_HIP(hipMemcpyAsync(out->kp_x, T->kp_x, (size_t)n_kp * 12, hipMemcpyDeviceToDevice, st));
    KPD_HIP(hipMemcpyAsync(out->kp_h, T->ks[Rk], (size_t)n_kp * S * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_v_transpose, grid1((long long)n_kp * 3 * VC), dim3(256), 0, st, T->kv[Rk], (long long)n_kp, 0, T->V, out->kp_v);
    KPD_LAUNCH_CHECK();
    KPD_HIP(hipMemcpyAsync(out->rk_src, T->rk_src, (size_t)e_rk * 4, hipMemcpyDeviceToDevice, st));
    KPD_HIP(hipMemcpyAsync(out->rk_dst, T->rk_dst, (size_t)e_rk * 4, hipMemcpyDeviceToDevice, st));
    // keypoint-keypoint radius graph (:285-292); counts = {E_kk, E_rk}
    KPD_TRY(launch_radius_graph(T->kp_x, T->kp_ptr, B, n_kp, K, c.kk_cutoff, 100, out->cap_kk, out->kk_src, out->kk_dst, T->kk_rowptr,
                                out->kk_per_graph, T->deg_tmp, T->kk_off, T->off_tmp, out->counts, st));
    T->have_forward = true;
    return KPD_OK;
}

extern "C" kpd_status kpd_recenc_trainer_backward(kpd_recenc_trainer *T, const float *d_kp_x, const float *d_kp_h, const float *d_kp_v,
                                                  void *stream) {
    KPD_REQUIRE(T, KPD_ERR_INVALID, "null trainer");
    KPD_REQUIRE(T->have_forward, KPD_ERR_STATE, "kpd_recenc_trainer_backward before kpd_recenc_trainer_forward");
    const kpd_recenc_config &c = T->cfg;
    hipStream_t st = static_cast<hipStream_t>(stream);
    T->st = st;
    KPD_TRY(wide_run(T, 1));                   // vector_size < 16: zero the wide gradients
    const int S = T->S, K = c.n_keypoints, B = T->B, n_rec = T->n_rec, n_kp = T->n_kp, F = c.in_scalar_size, Rr = c.n_rr_convs, Rk = c.n_rk_convs;
    const kpd_rec_batch &bt = T->bt;
    // incoming gradients (null = zero); keypoint vectors arrive as [n, 16, 3]
    int cur = 0, nxt = 1;
    if (d_kp_h) KPD_HIP(hipMemcpyAsync(T->gks[cur], d_kp_h, (size_t)n_kp * S * 4, hipMemcpyDeviceToDevice, st));
    else KPD_HIP(hipMemsetAsync(T->gks[cur], 0, (size_t)n_kp * S * 4, st));
    if (d_kp_v) {
        hipLaunchKernelGGL(k_v_transpose, grid1((long long)n_kp * 3 * VC), dim3(256), 0, st, d_kp_v, (long long)n_kp, 1, T->V, T->gkv[cur]);
        KPD_LAUNCH_CHECK();
    } else KPD_HIP(hipMemsetAsync(T->gkv[cur], 0, (size_t)n_kp * 3 * VC * 4, st));
    if (d_kp_x) KPD_HIP(hipMemcpyAsync(T->gkx, d_kp_x, (size_t)n_kp * 12, hipMemcpyDeviceToDevice, st));
    else KPD_HIP(hipMemsetAsync(T->gkx, 0, (size_t)n_kp * 12, st));
    KPD_HIP(hipMemsetAsync(T->grs, 0, (size_t)n_rec * S * 4, st));
    KPD_HIP(hipMemsetAsync(T->grv, 0, (size_t)n_rec * 3 * VC * 4, st));

    // rec -> kp convolutions, last to first: the keypoint-state gradient moves from gks[cur] to gks[nxt]; the receptor-state and
    // keypoint-position gradients accumulate
    if (c.message_norm_mode == 2) {
        hipLaunchKernelGGL(k_enc_z, grid1(B), dim3(256), 0, st, T->rk_rowptr, T->kp_ptr, B, T->z);
        KPD_LAUNCH_CHECK();
    }
    for (int j = Rk - 1; j >= 0; --j) {
        KPD_HIP(hipMemsetAsync(T->gks[nxt], 0, (size_t)n_kp * S * 4, st));
        KPD_HIP(hipMemsetAsync(T->gkv[nxt], 0, (size_t)n_kp * 3 * VC * 4, st));
        KPD_TRY(conv_bwd(T, rk_conv(T, j), T->gks[cur], T->gkv[cur], T->gks[nxt], T->gkv[nxt], T->grs, T->grv, T->gkx));
        std::swap(cur, nxt);
    }
    // (the keypoint state entering the first rk convolution is the constant zero: its gradient ends here)

    // keypoint initializer: kp_x = sum softmax(<ft_src, ft_dst> / sqrt(S)) x_rec
    {
        Param W, b, lw, lb, Ws, Wd;
        KPD_TRY(param(T, "keypoint_initializer.keypoint_embedding.0.weight", S * K, S, &W));
        KPD_TRY(param(T, "keypoint_initializer.keypoint_embedding.0.bias", S * K, 1, &b));
        KPD_TRY(param(T, "keypoint_initializer.keypoint_embedding.2.weight", S * K, 1, &lw));
        KPD_TRY(param(T, "keypoint_initializer.keypoint_embedding.2.bias", S * K, 1, &lb));
        KPD_TRY(param(T, "keypoint_initializer.src_net.weight", S, S, &Ws));
        KPD_TRY(param(T, "keypoint_initializer.dst_net.weight", S, S, &Wd));
        hipLaunchKernelGGL(k_att_bwd_logits, dim3(n_kp), dim3(256), 0, st, T->att, bt.rec_x, bt.rec_ptr, K, S, T->gkx, T->kp_x);
        KPD_LAUNCH_CHECK();
        float *dft_dst = T->gks[nxt], *dft_src = T->U;                        // free node-sized buffers
        hipLaunchKernelGGL(k_att_bwd_dst, dim3(n_kp), dim3(256), 0, st, T->att, T->ft_src, bt.rec_ptr, K, S, dft_dst);
        KPD_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_att_bwd_src, dim3(n_rec), dim3(256), 0, st, T->att, T->ft_dst, T->bidx[0], K, S, dft_src);
        KPD_LAUNCH_CHECK();
        // ft_src = s_R Ws^T, ft_dst = kp_emb Wd^T
        if (Ws.g) KPD_TRY(grad_gemm(T, S, S, n_rec, dft_src, S, T->rs[Rr], S, Ws.g, S));
        KPD_TRY(gemm(T, false, false, n_rec, S, S, dft_src, S, Ws.w, S, 1.0f, T->grs, S));
        if (Wd.g) KPD_TRY(grad_gemm(T, S, S, n_kp, dft_dst, S, T->kp_emb, S, Wd.g, S));
        KPD_TRY(gemm(T, false, false, n_kp, S, S, dft_dst, S, Wd.w, S, 0.0f, T->big, S));      // d kp_emb as [B, S * K]
        // keypoint embedding: LayerNorm(S K) <- SiLU <- Linear(S, S K) of the graph-mean feature
        hipLaunchKernelGGL(k_ln_bwd_g, dim3(cdiv(B, 4)), dim3(256), 0, st, T->kpe_act, lw.w, T->big, B, S * K, T->kp_emb, T->ft_dst);
        KPD_LAUNCH_CHECK();                                                                    // kp_emb <- d act, ft_dst <- dy * xhat
        KPD_TRY(colsum_wide(T, B, S * K, T->ft_dst, lw.g));
        KPD_TRY(colsum_wide(T, B, S * K, T->big, lb.g));
        hipLaunchKernelGGL(k_silu_bwd, grid1((long long)B * S * K), dim3(256), 0, st, T->kp_emb, T->kpe_pre, (long long)B * S * K, S * K, S * K);
        KPD_LAUNCH_CHECK();
        KPD_TRY(colsum_wide(T, B, S * K, T->kp_emb, b.g));
        if (W.g) KPD_TRY(gemm(T, true, false, S * K, S, B, T->kp_emb, S * K, T->gmean, S, 1.0f, W.g, S));
        KPD_TRY(gemm(T, false, false, B, S, S * K, T->kp_emb, S * K, W.w, S, 0.0f, T->ft_dst, S));      // d gmean [B, S]
        hipLaunchKernelGGL(k_mean_bwd, grid1((long long)n_rec * S), dim3(256), 0, st, T->ft_dst, T->bidx[0], bt.rec_ptr, (long long)n_rec * S, S,
                           T->grs);
        KPD_LAUNCH_CHECK();
    }

    // rec-rec convolutions, last to first: grs / grv hold the gradient of the state LEAVING conv i and receive the gradient of the
    // state entering it (source and destination are the same nodes: residual, source-side and destination-side terms all add up)
    if (c.message_norm_mode == 2) {
        hipLaunchKernelGGL(k_enc_z, grid1(B), dim3(256), 0, st, bt.rr_rowptr, bt.rec_ptr, B, T->z);
        KPD_LAUNCH_CHECK();
    }
    for (int i = Rr - 1; i >= 0; --i) {
        KPD_HIP(hipMemcpyAsync(T->gro_s, T->grs, (size_t)n_rec * S * 4, hipMemcpyDeviceToDevice, st));
        KPD_HIP(hipMemcpyAsync(T->gro_v, T->grv, (size_t)n_rec * 3 * VC * 4, hipMemcpyDeviceToDevice, st));
        KPD_HIP(hipMemsetAsync(T->grs, 0, (size_t)n_rec * S * 4, st));
        KPD_HIP(hipMemsetAsync(T->grv, 0, (size_t)n_rec * 3 * VC * 4, st));
        KPD_TRY(conv_bwd(T, rr_conv(T, i), T->gro_s, T->gro_v, T->grs, T->grv, T->grs, T->grv, nullptr));
    }

    // scalar embedding: rs[0] = LayerNorm(e_a1), e_a1 = SiLU(e_pre1), e_pre1 = e_a0 W1^T + b1, e_a0 = SiLU(e_pre0), e_pre0 = h W0^T + b0
    {
        Param W0, b0, W1, b1, lw, lb;
        KPD_TRY(param(T, "scalar_embed.0.weight", S, F, &W0)); KPD_TRY(param(T, "scalar_embed.0.bias", S, 1, &b0));
        KPD_TRY(param(T, "scalar_embed.2.weight", S, S, &W1)); KPD_TRY(param(T, "scalar_embed.2.bias", S, 1, &b1));
        KPD_TRY(param(T, "scalar_norm.weight", S, 1, &lw)); KPD_TRY(param(T, "scalar_norm.bias", S, 1, &lb));
        const long long tot = (long long)n_rec * S;
        hipLaunchKernelGGL(k_ln_bwd_g, dim3(cdiv(n_rec, 4)), dim3(256), 0, st, T->e_a1, lw.w, T->grs, n_rec, S, T->tmp_s, T->sb);
        KPD_LAUNCH_CHECK();
        KPD_TRY(colsum_acc(T, n_rec, S, T->sb, S, lw.g));
        KPD_TRY(colsum_acc(T, n_rec, S, T->grs, S, lb.g));
        hipLaunchKernelGGL(k_silu_bwd, grid1(tot), dim3(256), 0, st, T->tmp_s, T->e_pre1, tot, S, S);
        KPD_LAUNCH_CHECK();
        KPD_TRY(grad_gemm(T, S, S, n_rec, T->tmp_s, S, T->e_a0, S, W1.g, S, b1.g));
        KPD_TRY(gemm(T, false, false, n_rec, S, S, T->tmp_s, S, W1.w, S, 0.0f, T->sb, S, 1.0f, T->e_pre0));      // * SiLU'(pre0) in the epilogue
        KPD_TRY(grad_gemm(T, S, F, n_rec, T->sb, S, bt.rec_h, F, W0.g, F, b0.g));
    }
    KPD_TRY(wide_run(T, 2));                   // vector_size < 16: add the wide gradients into the caller's tensors (reference shapes)
    T->have_forward = false;
    return KPD_OK;
}
