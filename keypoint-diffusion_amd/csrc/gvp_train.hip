// Training path of the GVP denoiser: forward with saved conv states + backward.
//
// Gradients of LigRecDynamicsGVP.forward (models/dynamics_gvp.py:149-199; LigRecGVP :46-101, GVPMultiEdgeConv
// models/gvp.py:459-551, GVP :89-116, GVPLayerNorm :159-166, NoisePredictionBlock dynamics_gvp.py:10-44) with respect to
// every parameter, to the scalar / vector input features and -- when asked for -- to the ligand and keypoint positions, which
// enter through the unit edge vector and the rbf code of every edge (gvp.py:472-480): learned keypoints make them a function of
// the receptor encoder's parameters (k_gvp_geom_bwd).
//
// Same formulation as egnn_train.hip: parameters in place in the reference layout, gradients accumulated in that
// layout, dense products through sgemm.hip on the caller's stream, everything else in the kernels below; only node-sized
// state per conv is kept between forward and backward, edge activations are recomputed one edge type at a time.
// Vector features are kept as [rows, 3, channels] (the reference holds [rows, channels, 3]), so that the channel
// mixes Wh / Wu are plain GEMMs over 3 x rows; the first scalar Linear of the message function is split as in the
// inference path (per-node block U = s_src W[:, :S]^T gathered per edge, + rbf and vector-norm blocks per edge).
// Dropout: GVPDropout (gvp.py:119-149) acts on the aggregated messages and on the update residual in training mode
// (feature dropout per element, vector dropout per channel, both scaled by 1 / (1 - rate)).  Masks come from Philox keyed
// by a per-forward seed and the (conv, node type, position, kind) stream, so the backward pass regenerates them instead
// of storing them; kpd_dropout_mask exposes the same stream for tests.
#include <cstring>

#include "gvp_kernels.h"
#include "gvp_train_core.h"

using namespace kpd;

struct kpd_gvp_trainer : TrainCtx {
    kpd_gvp_config cfg{};
    Arena ws;
    int S = 256;
    int V = VC;                         // the model's vector_size (<= 16: narrower models run zero-padded, train_ops.h WideSet)
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
    // Fused message forward (S = 256 with kept activations): one projection launch, ONE launch of the inference path's register-chained edge
    // kernel in its training form (k_gvp_chain<16, 0, 1>: all edge types of the conv, the whole 3-GVP chain, activations stored on the way)
    // and one launch that sums the per-tile message pieces, per conv.  The chain's weights are re-packed into the kernels' fragment order
    // from the current parameters by one launch per forward (descriptor table built once per binding).
    struct PackDesc {
        const float *src;
        float *dst;
        int kind, sn, sk, n_valid, k_base, k_valid, n_tiles;      // kind 0: fragments (pack.hip, k_pack_chain_frag); 1: dst[i < k_valid] = i < n_valid ? src[i] : 0
    };
    struct ChainPack {
        GvpW g[4];
        GvpBwdW bw[4];                               // the same GVPs in backward form (k_gvp_chain_bwd)
        const float *wproj = nullptr, *bproj = nullptr;
    };
    bool fused = false, pack_dirty = true;
    float *pack_base = nullptr;
    std::vector<ChainPack> packs;                  // [conv * 4 + et]
    PackDesc *desc_dev = nullptr;
    int n_desc = 0, desc_cap = 0;
    GvpTrainSlot *slots_dev = nullptr;             // [conv * 4 + et]
    // Fused node update (k_gvp_node_chain<16, 0, 1>): one launch per conv for both node types -- message sums, dropout, the two layer norms and
    // the update chain; what the backward pass reads is kept per (conv, node type) instead of being recomputed there.
    struct NodeSlot {
        float *s1 = nullptr, *v1 = nullptr, *sb = nullptr, *vb = nullptr;
        GvpBuf gb[4];
    };
    std::vector<NodeSlot> nslots;                  // [conv * 2 + nt]
    struct NodePack { GvpW g[4]; GvpBwdW bw[4]; };
    GvpBwdGvp nbslots[2][4];                       // per node type and update GVP: dpre / dgate / dVu / d|Vh| of the fused node-chain backward
    std::vector<NodePack> npacks;                  // [conv * 2 + nt]: the update GVPs in the kernels' fragment order
    float *Uet[4] = {nullptr, nullptr, nullptr, nullptr}, *dvin_et[4] = {nullptr, nullptr, nullptr, nullptr};      // per edge type: by-source sums of the head GVP's dpre [n_src][256];
                                                   // gradient of its input vectors [E][3][17] -- kept until the conv's batched launches ran
    float *vpart = nullptr;                        // [16][VEC_PART_REGION]: partial sums of the vector-weight kernels of a conv (one region per call)
    VecRedBatch vred;                              // ... and their pending reductions (one launch per 16: flush_vec_reduce)
    int n_vred = 0;
    GvpBwdSlot bslots[4];                          // per edge type: what the fused message backward leaves for the weight-gradient products
    GvpBwdSlot *bslots_dev = nullptr;              // [4]
    float *Psrc[4] = {nullptr, nullptr, nullptr, nullptr};
    float *ms_main[4] = {nullptr, nullptr, nullptr, nullptr}, *ms_cont[4] = {nullptr, nullptr, nullptr, nullptr};
    float *mv_main[4] = {nullptr, nullptr, nullptr, nullptr}, *mv_cont[4] = {nullptr, nullptr, nullptr, nullptr};
    void release_fused() {
        if (pack_base) (void)hipFree(pack_base);
        if (desc_dev) (void)hipFree(desc_dev);
        if (slots_dev) (void)hipFree(slots_dev);
        if (bslots_dev) (void)hipFree(bslots_dev);
        pack_base = nullptr; desc_dev = nullptr; slots_dev = nullptr; bslots_dev = nullptr;
        fused = false; pack_dirty = true; n_desc = desc_cap = 0;
    }
};

namespace {

const char *kCanon[4] = {"lig_ll_lig", "kp_kl_lig", "lig_lk_kp", "kp_kk_kp"};
const char *kNtName[2] = {"lig", "kp"};
const int kSrc[4] = {NT_LIG, NT_KP, NT_LIG, NT_KP};
const int kDst[4] = {NT_LIG, NT_LIG, NT_KP, NT_KP};

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
    hipLaunchKernelGGL(k_dropout, grid1((long long)n * S), dim3(256), 0, T->st, s, (long long)n, 1, S, S, T->seed, drop_stream(conv, nt, pos, 0),
                       T->dropout, so);
    KPD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_dropout, grid1((long long)n * 3 * VC), dim3(256), 0, T->st, v, (long long)n, 3, VC, T->V, T->seed,
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
    KPD_TRY(gather_rows(T->st, T->U, T->e_src[et], nullptr, E, S, T->gb[0].pre));
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

// ---- fused message forward -----------------------------------------------------------------------------------------------------
// every weight block of every message chain into the chained kernels' A-fragment order: block (x, y) = tile x of descriptor y
__global__ void k_gvp_train_pack(const kpd_gvp_trainer::PackDesc *__restrict__ descs) {
    const kpd_gvp_trainer::PackDesc d = descs[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (d.kind == 1) {
        if (idx < d.k_valid) d.dst[idx] = idx < d.n_valid ? d.src[idx] : 0.0f;
        return;
    }
    if ((int)blockIdx.x >= d.n_tiles) return;
    const int r = idx & 3, lane = (idx >> 2) & 63, mt = idx >> 8;
    const int n = 16 * mt + (lane & 15), i = 4 * (lane >> 4) + r;
    d.dst[idx] = (n < d.n_valid && i < d.k_valid) ? d.src[(size_t)n * d.sn + (size_t)(d.k_base + i) * d.sk] : 0.0f;
}

// split-sum scratch: one share per CU of a batched message-chain weight gradient (256 x 256 + the narrow riders, wgrad_batch) with slack for
// the rounding of the per-product slice counts
constexpr size_t GVP_PART_FLOATS = std::max<size_t>(GRAD_PART_FLOATS, (size_t)320 * (256 * 256 + 256 * 17 + 256 + 16 * 256));
constexpr size_t PK_CHUNK = 16 * 256;                                   // floats of one weight chunk (16 output tiles)
constexpr size_t PK_HEAD = 4 * PK_CHUNK + 9 * 256 + 2 * 256 + 256 + 64 + 16 * PK_CHUNK + 256 + 4 * PK_CHUNK;      // chain | whp | wup | b | bg | wproj | bproj | backward chain
constexpr size_t PK_GENERIC = 18 * PK_CHUNK + 256 + 256 + 256 + 64 + 18 * PK_CHUNK + 512;                            // chain | whp | wup | b | bg | backward chain | Wu^T | Wh^T
inline size_t pack_floats_per_chain(int nm) { return PK_HEAD + (size_t)(nm - 1) * PK_GENERIC; }
constexpr size_t PK_NODE_GVP = 18 * PK_CHUNK + 256 + 256 + 256 + 64 + 18 * PK_CHUNK + 512;          // an update GVP: chain | whp | wup | b | bg | backward chain | Wu^T | Wh^T

// carve the pack arena and (re)build the descriptor table from the bound parameters
kpd_status build_pack_table(kpd_gvp_trainer *T) {
    const int L = T->cfg.n_convs, nm = T->cfg.n_message_gvps, S = T->S;
    std::vector<kpd_gvp_trainer::PackDesc> descs;
    auto frag = [&](const float *src, int sn, int sk, int n_valid, int k_base, int k_valid, int n_tiles, float *dst) {
        descs.push_back(kpd_gvp_trainer::PackDesc{src, dst, 0, sn, sk, n_valid, k_base, k_valid, n_tiles});
    };
    auto copy = [&](const float *src, int n, int total, float *dst) { descs.push_back(kpd_gvp_trainer::PackDesc{src, dst, 1, 0, 0, n, 0, total, 1}); };
    T->packs.assign((size_t)L * 4, kpd_gvp_trainer::ChainPack());
    float *p = T->pack_base;
    auto take = [&](size_t n) { float *r = p; p += n; return r; };
    for (int conv = 0; conv < L; ++conv)
        for (int et = 0; et < 4; ++et) {
            if (!conv_uses(T, conv, et)) continue;
            kpd_gvp_trainer::ChainPack &cp = T->packs[(size_t)conv * 4 + et];
            const std::string prefix = "noise_predictor.conv_layers." + std::to_string(conv) + ".edge_message_fns." + kCanon[et];
            for (int j = 0; j < nm; ++j) {
                GvpP g;
                const bool head = j == 0;
                KPD_TRY(gvp_params(T, prefix + "." + std::to_string(j), head ? VH : VC, VC, head ? S + RBF : S, S, &g));
                const int k_all = g.si + g.h, chunks = head ? 4 : 18;
                float *chain = take((size_t)chunks * PK_CHUNK), *whp = take(head ? 9 * 256 : 256), *wup = take(head ? 2 * 256 : 256);
                float *b = take(256), *bg = take(64);
                GvpW &w = cp.g[j];
                w.b = b; w.bg = bg; w.vin = g.vi; w.h = g.h; w.vout = g.vo; w.sout = S; w.vec_sigmoid = 1;
                w.chain = chain; w.chain_h = nullptr; w.whp = whp; w.wup = wup;
                if (head) {
                    // Wh [17 in][17 hidden]: input tiles [source 1..16] (slot 0) and [x_diff 0] (slot 2) x two hidden tiles; Wu [17][16]
                    frag(g.Wh.w, 1, VH, VH, 1, 16, 2, whp);
                    frag(g.Wh.w, 1, VH, VH, 0, 1, 2, whp + 6 * 256);
                    frag(g.Wu.w, 1, VC, VC, 0, 16, 1, wup);
                    frag(g.Wu.w, 1, VC, VC, 16, 1, 1, wup + 256);
                    // to_feats_out [256][256 src | 16 rbf | 17 sh]: the source block per node (k_gvp_proj_chain), rbf and sh per edge
                    float *wproj = take(16 * PK_CHUNK), *bproj = take(256);
                    for (int kc = 0; kc < 16; ++kc) frag(g.Ws.w, k_all, 1, S, 16 * kc, 16, 16, wproj + (size_t)kc * PK_CHUNK);
                    copy(g.bs.w, S, 256, bproj);               // the bias rides with the per-node projection; b stays zero (arena memset)
                    cp.wproj = wproj; cp.bproj = bproj;
                    frag(g.Ws.w, k_all, 1, S, S, 16, 16, chain);
                    frag(g.Ws.w, k_all, 1, S, S + RBF, 16, 16, chain + PK_CHUNK);
                    frag(g.Ws.w, k_all, 1, S, S + RBF + 16, 1, 16, chain + 2 * PK_CHUNK);
                } else {
                    frag(g.Wh.w, 1, VC, VC, 0, 16, 1, whp);
                    frag(g.Wu.w, 1, VC, VC, 0, 16, 1, wup);
                    for (int kc = 0; kc < 16; ++kc) frag(g.Ws.w, k_all, 1, S, 16 * kc, 16, 16, chain + (size_t)kc * PK_CHUNK);
                    frag(g.Ws.w, k_all, 1, S, S, 16, 16, chain + 16 * PK_CHUNK);
                    copy(g.bs.w, S, 256, b);
                }
                float *gch = chain + (size_t)(chunks - 1) * PK_CHUNK;
                for (int nt = 0; nt < 16; ++nt) frag(g.Wg.w, S, 1, VC, 16 * nt, 16, 1, gch + nt * 256);
                copy(g.bg.w, VC, 16, bg);
                // backward form (gvp_kernels.h, GvpBwdW): Wg^T, then the transposed k-slabs of to_feats_out and its narrow blocks as one-tile slabs
                float *bch = take((size_t)chunks * PK_CHUNK);
                GvpBwdW &bw = cp.bw[j];
                bw.chain = bch; bw.wut = nullptr; bw.wht = nullptr;
                frag(g.Wg.w, 1, S, S, 0, 16, 16, bch);
                if (head) {
                    for (int nt = 0; nt < 16; ++nt) {
                        frag(g.Ws.w + S, 1, k_all, 16, 16 * nt, 16, 1, bch + PK_CHUNK + nt * 256);                       // rbf block
                        frag(g.Ws.w + S + RBF, 1, k_all, 16, 16 * nt, 16, 1, bch + 2 * PK_CHUNK + nt * 256);             // |Vh| channels 0..15
                        frag(g.Ws.w + S + RBF + 16, 1, k_all, 1, 16 * nt, 16, 1, bch + 3 * PK_CHUNK + nt * 256);        // |Vh| channel 16
                    }
                } else {
                    for (int c = 0; c < 16; ++c) frag(g.Ws.w, 1, k_all, S, 16 * c, 16, 16, bch + (size_t)(1 + c) * PK_CHUNK);
                    for (int nt = 0; nt < 16; ++nt) frag(g.Ws.w + S, 1, k_all, 16, 16 * nt, 16, 1, bch + 17 * PK_CHUNK + nt * 256);
                    float *wut = take(256), *wht = take(256);
                    frag(g.Wu.w, VC, 1, VC, 0, 16, 1, wut);          // dVh[h] = sum_o Wu[h][o] dVu[o]
                    frag(g.Wh.w, VC, 1, VC, 0, 16, 1, wht);          // dv_in[i] = sum_h Wh[i][h] dVh[h]
                    bw.wut = wut; bw.wht = wht;
                }
            }
        }
    const int nu = T->cfg.n_update_gvps;
    T->npacks.assign((size_t)L * 2, kpd_gvp_trainer::NodePack());
    for (int conv = 0; conv < L; ++conv)
        for (int nt = 0; nt < 2; ++nt) {
            if (!conv_uses(T, conv, nt == 0 ? 0 : 2)) continue;          // (lig is the destination of ll / kl, kp of lk / kk)
            const std::string prefix = "noise_predictor.conv_layers." + std::to_string(conv) + ".node_update_fns." + kNtName[nt];
            for (int j = 0; j < nu; ++j) {
                GvpP g;
                KPD_TRY(gvp_params(T, prefix + "." + std::to_string(j), VC, VC, S, S, &g));
                const int k_all = g.si + g.h;
                float *chain = take(18 * PK_CHUNK), *whp = take(256), *wup = take(256), *b = take(256), *bg = take(64);
                GvpW &w = T->npacks[(size_t)conv * 2 + nt].g[j];
                w.b = b; w.bg = bg; w.vin = g.vi; w.h = g.h; w.vout = g.vo; w.sout = S; w.vec_sigmoid = 1;
                w.chain = chain; w.chain_h = nullptr; w.whp = whp; w.wup = wup;
                frag(g.Wh.w, 1, VC, VC, 0, 16, 1, whp);
                frag(g.Wu.w, 1, VC, VC, 0, 16, 1, wup);
                for (int kc = 0; kc < 16; ++kc) frag(g.Ws.w, k_all, 1, S, 16 * kc, 16, 16, chain + (size_t)kc * PK_CHUNK);
                frag(g.Ws.w, k_all, 1, S, S, 16, 16, chain + 16 * PK_CHUNK);
                for (int nt2 = 0; nt2 < 16; ++nt2) frag(g.Wg.w, S, 1, VC, 16 * nt2, 16, 1, chain + 17 * PK_CHUNK + nt2 * 256);
                copy(g.bs.w, S, 256, b);
                copy(g.bg.w, VC, 16, bg);
                float *bch = take(18 * PK_CHUNK), *wut = take(256), *wht = take(256);
                GvpBwdW &bw = T->npacks[(size_t)conv * 2 + nt].bw[j];
                bw.chain = bch; bw.wut = wut; bw.wht = wht;
                frag(g.Wg.w, 1, S, S, 0, 16, 16, bch);
                for (int c = 0; c < 16; ++c) frag(g.Ws.w, 1, k_all, S, 16 * c, 16, 16, bch + (size_t)(1 + c) * PK_CHUNK);
                for (int nt2 = 0; nt2 < 16; ++nt2) frag(g.Ws.w + S, 1, k_all, 16, 16 * nt2, 16, 1, bch + 17 * PK_CHUNK + nt2 * 256);
                frag(g.Wu.w, VC, 1, VC, 0, 16, 1, wut);
                frag(g.Wh.w, VC, 1, VC, 0, 16, 1, wht);
            }
        }
    if ((int)descs.size() > T->desc_cap) {
        if (T->desc_dev) (void)hipFree(T->desc_dev);
        T->desc_dev = nullptr;
        KPD_HIP(hipMalloc(reinterpret_cast<void **>(&T->desc_dev), descs.size() * sizeof(kpd_gvp_trainer::PackDesc)));
        T->desc_cap = (int)descs.size();
    }
    KPD_HIP(hipMemcpy(T->desc_dev, descs.data(), descs.size() * sizeof(kpd_gvp_trainer::PackDesc), hipMemcpyHostToDevice));
    T->n_desc = (int)descs.size();
    T->pack_dirty = false;
    return KPD_OK;
}

// once per forward: the current parameters into the pack arena
kpd_status pack_chains(kpd_gvp_trainer *T) {
    if (T->pack_dirty) KPD_TRY(build_pack_table(T));
    if (T->n_desc == 0) return KPD_OK;
    hipLaunchKernelGGL(k_gvp_train_pack, dim3(16, T->n_desc), dim3(256), 0, T->st, T->desc_dev);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// the messages of one conv, all edge types: per-node projections, then the chained edge kernel in its training form (message pieces per tile)
kpd_status conv_messages_fused(kpd_gvp_trainer *T, int conv) {
    const int S = T->S, nm = T->cfg.n_message_gvps;
    const bool all4 = conv_uses(T, conv, 2);
    GvpProjArgs pa;
    memset(&pa, 0, sizeof(pa));
    GvpEdgeArgs ea;
    memset(&ea, 0, sizeof(ea));
    ea.meta = all4 ? T->meta : T->meta + 16;
    ea.x[0] = T->bt.lig_x; ea.x[1] = T->bt.kp_x; ea.v[0] = T->vs[0][conv]; ea.v[1] = T->vs[1][conv];
    ea.n_gvps = nm; ea.S = S; ea.rbf_dmax = 15.0f; ea.train = T->slots_dev + (size_t)conv * 4;
    int run = 0, tiles = 0;
    for (int et = 0; et < (all4 ? 4 : 2); ++et) {
        const kpd_gvp_trainer::ChainPack &cp = T->packs[(size_t)conv * 4 + et];
        const int s = kSrc[et];
        pa.s[pa.n_slots] = T->ss[s][conv]; pa.n[pa.n_slots] = T->n[s]; pa.wp[pa.n_slots] = cp.wproj; pa.b[pa.n_slots] = cp.bproj;
        pa.P[pa.n_slots] = T->Psrc[et]; pa.tiles_first[pa.n_slots] = run;
        run += cdiv(T->n[s], TM);
        ++pa.n_slots;
        ea.src[et] = T->e_src[et]; ea.dst[et] = T->e_dst[et]; ea.Psrc[et] = T->Psrc[et];
        for (int j = 0; j < nm; ++j) ea.g[et][j] = cp.g[j];
        ea.ms_main[et] = T->ms_main[et]; ea.ms_cont[et] = T->ms_cont[et]; ea.mv_main[et] = T->mv_main[et]; ea.mv_cont[et] = T->mv_cont[et];
        tiles += cdiv(T->E[et], TM);
    }
    pa.tiles_first[pa.n_slots] = run;
    pa.S = S;
    KPD_TRY(launch_gvp_proj(pa, T->st));
    KPD_TRY(launch_gvp_edge(ea, tiles, T->st));
    return KPD_OK;
}

// the node update of one conv, both node types, from the message pieces of conv_messages_fused
kpd_status conv_nodes_fused(kpd_gvp_trainer *T, int conv, const bool (&is_dst)[2]) {
    const int S = T->S, nu = T->cfg.n_update_gvps;
    const bool all4 = conv_uses(T, conv, 2);
    const std::string cp = "noise_predictor.conv_layers." + std::to_string(conv);
    GvpNodePair np;
    memset(&np, 0, sizeof(np));
    for (int nt = 0; nt < 2; ++nt) {
        if (!is_dst[nt]) continue;
        GvpNodeArgs &na = np.nt[nt];
        LnP l1, l2;
        KPD_TRY(ln_params(T, cp + ".message_layer_norms." + kNtName[nt], &l1));
        KPD_TRY(ln_params(T, cp + ".update_layer_norms." + kNtName[nt], &l2));
        na.n = T->n[nt]; na.bidx = T->bidx[nt];
        na.mean = T->cfg.message_norm_mode == 1;
        na.z = T->cfg.message_norm_mode == 2 ? T->z[nt] : nullptr;
        na.norm_const = T->cfg.message_norm_mode == 0 ? T->cfg.message_norm : 1.0f;
        for (int et = 0; et < (all4 ? 4 : 2); ++et) {
            if (kDst[et] != nt) continue;
            const int k = na.n_in++;
            na.rowptr[k] = T->e_rowptr[et];
            na.ms_main[k] = T->ms_main[et]; na.ms_cont[k] = T->ms_cont[et]; na.mv_main[k] = T->mv_main[et]; na.mv_cont[k] = T->mv_cont[et];
        }
        na.ln1_w = l1.gamma.w; na.ln1_b = l1.beta.w; na.ln2_w = l2.gamma.w; na.ln2_b = l2.beta.w;
        na.n_gvps = nu; na.S = S;
        na.ln_inv_n = 1.0f / (float)S; na.ln_pad = 0.0f;
        na.vn_inv_n = 1.0f / (float)T->V; na.vn_pad = (float)(VC - T->V);
        const kpd_gvp_trainer::NodeSlot &ns = T->nslots[(size_t)conv * 2 + nt];
        for (int j = 0; j < nu; ++j) {
            na.g[j] = T->npacks[(size_t)conv * 2 + nt].g[j];
            const GvpBuf &b = ns.gb[j];
            na.tr.g[j] = GvpTrainGvp{b.Vh, b.Vu, b.sh, b.pre, b.s, b.gate, b.V};
        }
        na.train = 1;
        GvpNodeTrain &t = na.tr;
        t.s_in = T->ss[nt][conv]; t.v_in = T->vs[nt][conv]; t.s_out = T->ss[nt][conv + 1]; t.v_out = T->vs[nt][conv + 1];
        t.sa = T->sa[nt][conv]; t.va = T->va[nt][conv]; t.s1 = ns.s1; t.v1 = ns.v1; t.sb = ns.sb; t.vb = ns.vb;
        t.rate = T->dropout; t.seed = T->seed; t.live_v = T->V;
        t.stream[0] = drop_stream(conv, nt, 0, 0); t.stream[1] = drop_stream(conv, nt, 0, 1);
        t.stream[2] = drop_stream(conv, nt, 1, 0); t.stream[3] = drop_stream(conv, nt, 1, 1);
    }
    np.tiles0 = cdiv(np.nt[0].n, TM);
    return launch_gvp_node(np, T->st);
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
        if (T->fused) continue;
        KPD_HIP(hipMemsetAsync(T->sa[nt][conv], 0, (size_t)T->n[nt] * S * 4, T->st));          // aggregated messages first ...
        KPD_HIP(hipMemsetAsync(T->va[nt][conv], 0, (size_t)T->n[nt] * 3 * VC * 4, T->st));
    }
    if (T->fused) {
        KPD_TRY(conv_messages_fused(T, conv));
        return conv_nodes_fused(T, conv, is_dst);
    }
    for (int et = 0; et < 4; ++et) {
        if (T->fused || !conv_uses(T, conv, et) || T->E[et] == 0) continue;
        const int d = kDst[et];
        bind_msg(T, conv, et);
        KPD_TRY(message_fwd(T, conv, et, nullptr));
        KPD_TRY(edge_scale(T, et));
        KPD_TRY(segsum(T->st, T->gb[nm - 1].s, S, 0, S, nullptr, T->e_rowptr[et], T->scale, 1.0f, true, T->n[d], T->sa[d][conv], S));
        KPD_TRY(segsum(T->st, T->gb[nm - 1].V, 3 * VC, 0, 3 * VC, nullptr, T->e_rowptr[et], T->scale, 1.0f, true, T->n[d], T->va[d][conv], 3 * VC));
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

// the pending reductions of the vector-weight kernels (k_gvp_vec16_bwd / k_gvp_vec17_bwd partials in T->vpart) in one launch
kpd_status flush_vec_reduce(kpd_gvp_trainer *T) {
    if (T->n_vred == 0) return KPD_OK;
    hipLaunchKernelGGL(k_gvp_vec_reduce_batch, dim3(cdiv(272 + 289, 16), T->n_vred), dim3(256), 0, T->st, T->vred);
    KPD_LAUNCH_CHECK();
    T->n_vred = 0;
    return KPD_OK;
}

// What is left of gvp_bwd (gvp_train_core.h) for a message GVP after k_gvp_chain_bwd: the parameter gradients, from the kernel's dpre / dgate /
// dVu / d|Vh| and the kept activations -- the gate matrix, to_feats_out's scalar block (s_in: null at the head, whose source block is
// differentiated per node) and its |Vh| block with the bias, and Wu / Wh through the vector kernels (which also leave dv_in: read at the
// head -- the gradient of [x_diff | v_src] -- and redundant elsewhere).
// skip: REST_SKIP_WG -- the gradients of the gate matrix and its bias ride in another product; REST_SKIP_WS -- so do to_feats_out's blocks
// and bias (wgrad_batch, sgemm.hip)
constexpr int REST_SKIP_WG = 1, REST_SKIP_WS = 2;
kpd_status gvp_bwd_rest(kpd_gvp_trainer *T, const GvpP &g, int M, const float *s_in, const float *v_in, const GvpBuf &B, const GvpBwdGvp &o,
                        float *dv_in, int skip) {
    if (M == 0) return KPD_OK;
    if (!(skip & REST_SKIP_WG)) KPD_TRY(grad_gemm(T, g.vo, g.so, M, o.dgate, g.vo, B.s, g.so, g.Wg.g, g.so, g.bg.g));
    if (!(skip & REST_SKIP_WS)) {
        if (s_in && g.Ws.g) KPD_TRY(grad_gemm(T, g.so, g.si, M, o.dpre, g.so, s_in, g.si, g.Ws.g, g.si + g.h));
        KPD_TRY(grad_gemm(T, g.so, g.h, M, o.dpre, g.so, B.sh, g.h, g.Ws.g ? g.Ws.g + g.si : nullptr, g.si + g.h, g.bs.g));
    }
    // Wu / Wh: every call leaves its per-workgroup partials in a region of its own; the reductions wait and go out sixteen per launch
    const int blocks = std::max(1, std::min(cdiv(cdiv(M, 16), 4), std::min(std::min(2 * cu_count(), 512), VEC16_MAX_WAVES / 4)));
    if (T->n_vred == 16) KPD_TRY(flush_vec_reduce(T));
    float *part = T->vpart + (size_t)T->n_vred * VEC_PART_REGION;
    if (g.vi == 17) {
        hipLaunchKernelGGL(k_gvp_vec17_bwd, dim3(blocks), dim3(256), 0, T->st, o.dVu, B.Vh, B.sh, o.dsh, v_in, g.Wh.w, g.Wu.w, M, dv_in, part);
        KPD_LAUNCH_CHECK();
        if (g.Wu.g || g.Wh.g) T->vred.r[T->n_vred++] = VecRedBatch::One{part, blocks, VEC17_PART, 272, 289, g.Wu.g, g.Wh.g};
        return KPD_OK;
    }
    hipLaunchKernelGGL(k_gvp_vec16_bwd, dim3(blocks), dim3(256), 0, T->st, o.dVu, B.Vh, B.sh, o.dsh, v_in, g.Wh.w, g.Wu.w, M, dv_in, part);
    KPD_LAUNCH_CHECK();
    if (g.Wu.g || g.Wh.g) T->vred.r[T->n_vred++] = VecRedBatch::One{part, blocks, 512, 256, 256, g.Wu.g, g.Wh.g};
    return KPD_OK;
}

// backward of conv: gs/gv[cur] = gradients of the conv outputs, gs/gv[nxt] = gradients of its inputs
kpd_status conv_bwd(kpd_gvp_trainer *T, int conv, int cur, int nxt) {
    const int S = T->S, nm = T->cfg.n_message_gvps, nu = T->cfg.n_update_gvps;
    std::vector<WgradItem> wq, wq_top, wq_u;    // weight-gradient products of this conv, sent out together at the end (wgrad_batch)
    SegsumVinArgs sva;
    memset(&sva, 0, sizeof(sva));
    T->n_vred = 0;
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
        const float *s1 = T->s1, *v1 = T->v1, *sb = T->sb, *vb = T->vb;
        if (T->fused) {
            // kept by the fused node update (conv_nodes_fused)
            const kpd_gvp_trainer::NodeSlot &ns = T->nslots[(size_t)conv * 2 + nt];
            s1 = ns.s1; v1 = ns.v1; sb = ns.sb; vb = ns.vb;
            for (int j = 0; j < nu; ++j) T->gb[j] = ns.gb[j];
        } else {
            // recompute s1, v1, the update chain and the second pre-norm sums
            KPD_TRY(gvp_ln_fwd(T, l1, n, T->sa[nt][conv], T->va[nt][conv], T->s1, T->v1));
            KPD_TRY(chain_fwd(T, up, nu, n, T->s1, T->v1));
            KPD_TRY(dropout_apply(T, conv, nt, 1, n, T->gb[nu - 1].s, T->gb[nu - 1].V, T->tmp_s, T->tmp_v));
            hipLaunchKernelGGL(k_add, grid1((long long)n * S), dim3(256), 0, T->st, T->s1, T->tmp_s, (long long)n * S, T->sb);
            KPD_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_add, grid1((long long)n * 3 * VC), dim3(256), 0, T->st, T->v1, T->tmp_v, (long long)n * 3 * VC, T->vb);
            KPD_LAUNCH_CHECK();
        }
        // second GVPLayerNorm: d(sb, vb) -> gs/gv[nxt] (the residual: d s1 += d sb, d v1 += d vb); the update chain sees the same gradients
        // through its dropout mask -> ds[0] / dV[0] (a copy when the rate is 0)
        KPD_TRY(gvp_ln_bwd(T, l2, n, sb, vb, T->gs[cur][nt], T->gv[cur][nt], T->gs[nxt][nt], T->gv[nxt][nt]));
        KPD_TRY(dropout_apply(T, conv, nt, 1, n, T->gs[nxt][nt], T->gv[nxt][nt], T->ds[0], T->dV[0]));
        bool chained = false;
        if (T->fused) {
            GvpP gp[4];
            bool grads = true;
            for (int j = 0; j < nu; ++j) {
                KPD_TRY(gvp_params(T, up + "." + std::to_string(j), VC, VC, S, S, &gp[j]));
                grads = grads && gp[j].Ws.g && gp[j].bs.g && gp[j].Wg.g;
            }
            if (grads) {
                // the update chain's backward in one launch (k_gvp_node_chain_bwd), its weight gradients with the conv's batched products
                const kpd_gvp_trainer::NodeSlot &ns = T->nslots[(size_t)conv * 2 + nt];
                const kpd_gvp_trainer::NodePack &pk = T->npacks[(size_t)conv * 2 + nt];
                GvpNodeBwdArgs nb;
                memset(&nb, 0, sizeof(nb));
                nb.n = n; nb.ds = T->ds[0]; nb.dV = T->dV[0]; nb.ds_in = T->ds[1]; nb.dv_in = T->dV[1]; nb.n_gvps = nu;
                for (int j = 0; j < nu; ++j) {
                    const GvpBuf &b = ns.gb[j];
                    nb.g[j] = pk.bw[j];
                    nb.f[j] = GvpTrainGvp{b.Vh, b.Vu, b.sh, b.pre, b.s, b.gate, b.V};
                    nb.o[j] = T->nbslots[nt][j];
                }
                KPD_TRY(launch_gvp_node_bwd(nb, T->st));
                std::swap(T->ds[0], T->ds[1]);
                std::swap(T->dV[0], T->dV[1]);
                for (int j = nu - 1; j >= 0; --j) {
                    const GvpBwdGvp &o = T->nbslots[nt][j];
                    WgradItem it;
                    memset(&it, 0, sizeof(it));
                    it.A = o.dpre; it.lda = S; it.B = j > 0 ? ns.gb[j - 1].s : s1; it.ldb = S; it.K = n;
                    it.C = gp[j].Ws.g; it.ldc = gp[j].si + gp[j].h;
                    it.B2 = ns.gb[j].sh; it.ldb2 = gp[j].h; it.nb2 = gp[j].h; it.Cx1 = gp[j].Ws.g + gp[j].si; it.ldx1 = it.ldc;
                    it.colsum = gp[j].bs.g;
                    if (j > 0) { it.A2 = T->nbslots[nt][j - 1].dgate; it.lda2 = VC; it.na2 = VC; it.Cx2 = gp[j - 1].Wg.g; it.ldx2 = S; it.colsum2 = gp[j - 1].bg.g; }
                    wq.push_back(it);
                    // Wu / Wh through the vector kernel (its dv_in is not needed here)
                    KPD_TRY(gvp_bwd_rest(T, gp[j], n, nullptr, j > 0 ? ns.gb[j - 1].V : v1, ns.gb[j], o, nullptr, REST_SKIP_WG | REST_SKIP_WS));
                }
                {   // the last GVP's gate matrix: rider-only
                    const GvpBwdGvp &o = T->nbslots[nt][nu - 1];
                    WgradItem it;
                    memset(&it, 0, sizeof(it));
                    it.A = o.dpre; it.lda = S; it.B = ns.gb[nu - 1].s; it.ldb = S; it.K = n;
                    it.A2 = o.dgate; it.lda2 = VC; it.na2 = VC; it.Cx2 = gp[nu - 1].Wg.g; it.ldx2 = S; it.colsum2 = gp[nu - 1].bg.g;
                    wq_top.push_back(it);
                }
                chained = true;
            }
        }
        if (!chained) KPD_TRY(chain_bwd(T, up, nu, n, s1, v1));
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
    if (T->fused) {
        // every edge type's message chain backward in one launch, gradients through registers (k_gvp_chain_bwd, gvp_chain.hip)
        const bool all4 = conv_uses(T, conv, 2);
        GvpEdgeBwdArgs ba;
        memset(&ba, 0, sizeof(ba));
        ba.meta = all4 ? T->meta : T->meta + 16;
        int tiles = 0;
        for (int et = 0; et < (all4 ? 4 : 2); ++et) {
            ba.dst[et] = T->e_dst[et]; ba.rowptr[et] = T->e_rowptr[et];
            for (int j = 0; j < nm; ++j) ba.g[et][j] = T->packs[(size_t)conv * 4 + et].bw[j];
            tiles += cdiv(T->E[et], TM);
        }
        for (int nt = 0; nt < 2; ++nt) { ba.gs[nt] = T->gs[cur][nt]; ba.gv[nt] = T->gv[cur][nt]; ba.z[nt] = T->z[nt]; ba.bidx[nt] = T->bidx[nt]; }
        ba.mode = T->cfg.message_norm_mode; ba.norm = T->cfg.message_norm; ba.n_gvps = nm;
        ba.fwd = T->slots_dev + (size_t)conv * 4; ba.out = T->bslots_dev;
        KPD_TRY(launch_gvp_edge_bwd(ba, tiles, T->st));
    }
    for (int et = 0; et < 4; ++et) {
        if (!conv_uses(T, conv, et) || T->E[et] == 0) continue;
        const int E = T->E[et], s = kSrc[et], d = kDst[et];
        const std::string prefix = cp + ".edge_message_fns." + kCanon[et];
        GvpP g0;
        bind_msg(T, conv, et);
        KPD_TRY(message_fwd(T, conv, et, &g0, !T->store));
        bool rbf_done = false;
        const float *dpre0 = nullptr, *drbf = nullptr;          // dL/dpre of the head GVP [E, S]; dL/d rbf [E, 16] (positions wanted)
        if (T->fused) {
            // Parameter gradients.  The 256 x 256 block of GVP j's to_feats_out (dpre_j^T s_{j-1}) takes its |Vh| block and bias and the gate
            // matrix of GVP j - 1 (dgate_{j-1}^T s_{j-1}: the same s) along as narrow riders; the products of all edge types wait in wq and
            // go out as one launch per conv (wgrad_batch).
            const GvpBwdSlot &bs = T->bslots[et];
            GvpP gp[4];
            gp[0] = g0;
            for (int j = 1; j < nm; ++j) KPD_TRY(gvp_params(T, prefix + "." + std::to_string(j), VC, VC, S, S, &gp[j]));
            int skip[4] = {0, 0, 0, 0};
            // ... and the head's rbf and |Vh| blocks with its bias (dpre_0^T [rbf | sh_0 | 1]) share a rider-only pass with the gate matrix of
            // the chain's last GVP (dgate^T s of that GVP)
            if (g0.Ws.g && g0.bs.g && gp[nm - 1].Wg.g) {
                WgradItem it;
                memset(&it, 0, sizeof(it));
                it.A = bs.g[0].dpre; it.lda = S; it.B = T->gb[nm - 1].s; it.ldb = S; it.K = E;
                it.B2 = T->rbf; it.ldb2 = RBF; it.nb2 = RBF; it.Cx1 = g0.Ws.g + S; it.ldx1 = g0.si + g0.h;
                it.colsum = g0.bs.g;
                it.B3 = T->gb[0].sh; it.ldb3 = g0.h; it.nb3 = g0.h; it.Cx3 = g0.Ws.g + g0.si; it.ldx3 = g0.si + g0.h;
                it.A2 = bs.g[nm - 1].dgate; it.lda2 = VC; it.na2 = VC; it.Cx2 = gp[nm - 1].Wg.g; it.ldx2 = S; it.colsum2 = gp[nm - 1].bg.g;
                wq_top.push_back(it);
                skip[0] |= REST_SKIP_WS;
                skip[nm - 1] |= REST_SKIP_WG;
                rbf_done = true;
            }
            for (int j = nm - 1; j >= 1; --j) {
                if (gp[j].Ws.g && gp[j].bs.g && gp[j - 1].Wg.g) {
                    WgradItem it;
                    memset(&it, 0, sizeof(it));
                    it.A = bs.g[j].dpre; it.lda = S; it.B = T->gb[j - 1].s; it.ldb = S; it.K = E;
                    it.C = gp[j].Ws.g; it.ldc = gp[j].si + gp[j].h;
                    it.B2 = T->gb[j].sh; it.ldb2 = gp[j].h; it.nb2 = gp[j].h; it.Cx1 = gp[j].Ws.g + gp[j].si; it.ldx1 = it.ldc;
                    it.colsum = gp[j].bs.g;
                    it.A2 = bs.g[j - 1].dgate; it.lda2 = VC; it.na2 = VC; it.Cx2 = gp[j - 1].Wg.g; it.ldx2 = S; it.colsum2 = gp[j - 1].bg.g;
                    wq.push_back(it);
                    skip[j] |= REST_SKIP_WS;
                    skip[j - 1] |= REST_SKIP_WG;
                }
                KPD_TRY(gvp_bwd_rest(T, gp[j], E, T->gb[j - 1].s, T->gb[j - 1].V, T->gb[j], bs.g[j], nullptr, skip[j]));       // (dv_in: the chained kernel had it)
            }
            KPD_TRY(gvp_bwd_rest(T, g0, E, nullptr, T->vin, T->gb[0], bs.g[0], T->dvin_et[et], skip[0]));
            dpre0 = bs.g[0].dpre;
            drbf = bs.drbf;
        } else {
            KPD_TRY(edge_scale(T, et));
            // d(message of edge e) = scale[dst] * d(aggregate)[dst]
            KPD_TRY(gather_rows(T->st, T->gs[cur][d], T->e_dst[et], T->scale, E, S, T->ds[0]));
            KPD_TRY(gather_rows(T->st, T->gv[cur][d], T->e_dst[et], T->scale, E, 3 * VC, T->dV[0]));
            for (int j = nm - 1; j >= 1; --j) {
                GvpP g;
                KPD_TRY(gvp_params(T, prefix + "." + std::to_string(j), VC, VC, S, S, &g));
                KPD_TRY(gvp_bwd(T, g, E, T->gb[j - 1].s, S, T->gb[j - 1].V, T->gb[j], false, T->ds[0], T->dV[0], T->ds[1], T->dV[1]));
                std::swap(T->ds[0], T->ds[1]);
                std::swap(T->dV[0], T->dV[1]);
            }
            KPD_TRY(gvp_bwd(T, g0, E, nullptr, 0, T->vin, T->gb[0], false, T->ds[0], T->dV[0], nullptr, T->dV[1]));
            dpre0 = T->ds[0];
            if (T->want_x) {        // d rbf = dpre W[:, S:S+16] (dsh is free again)
                KPD_TRY(gemm(T, false, false, E, RBF, S, T->ds[0], S, g0.Ws.w + S, g0.si + g0.h, 0.0f, T->dsh, RBF));
                drbf = T->dsh;
            }
        }
        if (T->want_x) {
            // positions (gvp.py:472-480): through the rbf code and through the unit vector = channel 0 of d vin
            const float *xs = s == NT_LIG ? T->bt.lig_x : T->bt.kp_x, *xd = d == NT_LIG ? T->bt.lig_x : T->bt.kp_x;
            hipLaunchKernelGGL(k_gvp_geom_bwd, grid1(E), dim3(256), 0, T->st, T->e_src[et], T->e_dst[et], xs, xd, E, 15.0f, T->rbf, drbf,
                               T->fused ? T->dvin_et[et] : T->dV[1], VH, T->dxe);
            KPD_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_seg3, grid1(T->n[s]), dim3(256), 0, T->st, T->dxe, T->scsr[et].perm, T->scsr[et].rowptr, T->n[s], 1.0f, T->gx[s]);
            KPD_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_seg3, grid1(T->n[d]), dim3(256), 0, T->st, T->dxe, (const int *)nullptr, T->e_rowptr[et], T->n[d], -1.0f, T->gx[d]);
            KPD_LAUNCH_CHECK();
        }
        // dpre0 = dL/dpre of the first GVP: its scalar inputs were U[src] and rbf
        if (g0.Ws.g && !rbf_done) KPD_TRY(grad_gemm(T, S, RBF, E, dpre0, S, T->rbf, RBF, g0.Ws.g + S, g0.si + g0.h));
        // sums over the out-edges of every source node, in ascending edge order (no float atomics)
        // (the source block's gradient U^T s_src: as one more item of the conv's rider-carrying batched launch it cost that launch more -- 7.4 -> 8.3
        //  ms per step -- than the 22 small products it replaced -- 0.6; the four of a conv go out as a batch of their own, without riders)
        float *U = T->fused ? T->Uet[et] : T->U;
        KPD_TRY(segsum(T->st, dpre0, S, 0, S, T->scsr[et].perm, T->scsr[et].rowptr, nullptr, 1.0f, false, T->n[s], U, S));
        if (g0.Ws.g && T->fused) {          // U^T s_src of the conv's edge types: one rider-less batched launch below
            WgradItem it;
            memset(&it, 0, sizeof(it));
            it.A = U; it.lda = S; it.B = T->ss[s][conv]; it.ldb = S; it.K = T->n[s]; it.C = g0.Ws.g; it.ldc = g0.si + g0.h;
            wq_u.push_back(it);
        } else if (g0.Ws.g) KPD_TRY(grad_gemm(T, S, S, T->n[s], U, S, T->ss[s][conv], S, g0.Ws.g, g0.si + g0.h));
        KPD_TRY(gemm(T, false, false, T->n[s], S, S, U, S, g0.Ws.w, g0.si + g0.h, 1.0f, T->gs[nxt][s], S));
        // vector rows [E, 3, 17], channels 1..16 -> gv[src, 3, 16]
        if (T->fused) {
            sva.dvin[et] = T->dvin_et[et]; sva.perm[et] = T->scsr[et].perm; sva.rowptr[et] = T->scsr[et].rowptr; sva.live[et] = 1; sva.src_nt[et] = s;
        } else {
            hipLaunchKernelGGL(k_segsum_vin, dim3(cdiv(T->n[s], 4)), dim3(256), 0, T->st, T->dV[1], T->scsr[et].perm, T->scsr[et].rowptr, T->n[s], T->gv[nxt][s]);
            KPD_LAUNCH_CHECK();
        }
    }
    if (T->fused) {          // the source vectors' gradients of every edge type: one launch
        for (int nt = 0; nt < 2; ++nt) { sva.n[nt] = T->n[nt]; sva.gv[nt] = T->gv[nxt][nt]; }
        hipLaunchKernelGGL(k_segsum_vin_all, dim3(cdiv(std::max(T->n[0], T->n[1]), 4), 2), dim3(256), 0, T->st, sva);
        KPD_LAUNCH_CHECK();
        for (size_t i = 0; i < wq_u.size(); i += 8) KPD_TRY(wgrad_batch(wq_u.data() + i, (int)std::min<size_t>(8, wq_u.size() - i), T->part, T->part_floats, T->st));
    }
    KPD_TRY(flush_vec_reduce(T));
    for (size_t i = 0; i < wq.size(); i += 8) KPD_TRY(wgrad_batch(wq.data() + i, (int)std::min<size_t>(8, wq.size() - i), T->part, T->part_floats, T->st));
    for (size_t i = 0; i < wq_top.size(); i += 8)
        KPD_TRY(wgrad_batch(wq_top.data() + i, (int)std::min<size_t>(8, wq_top.size() - i), T->part, T->part_floats, T->st));
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
    KPD_TRY(gemm(T, false, true, n, S, F + 1, T->enc_in[nt], F + 1, W.w, F + 1, 0.0f, T->enc_pre[nt], S, 1.0f, nullptr, b.w, T->enc_act[nt]));
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
    KPD_TRY(grad_gemm(T, S, F + 1, n, T->tmp_s, S, T->enc_in[nt], F + 1, W.g, F + 1, b.g));
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
        KPD_TRY(gemm(T, false, true, n, F, kHeadS, T->gb[nn - 1].s, kHeadS, W.w, kHeadS, 0.0f, eps_h, F, 1.0f, nullptr, b.w));
        KPD_HIP(hipMemcpyAsync(eps_x, T->gb[nn - 1].V, (size_t)n * 12, hipMemcpyDeviceToDevice, T->st));     // [n, 3, 1]
    }
    return KPD_OK;
}

}  // namespace

extern "C" kpd_status kpd_gvp_trainer_create(const kpd_gvp_config *cfg, kpd_gvp_trainer **out) {
    KPD_REQUIRE(cfg && out, KPD_ERR_INVALID, "null argument");
    KPD_REQUIRE(cfg->vector_size >= 1 && cfg->vector_size <= VC, KPD_ERR_INVALID, "vector_size=%d outside 1 .. %d", cfg->vector_size, VC);
    KPD_REQUIRE(cfg->n_hidden_scalars >= 1 && cfg->n_hidden_scalars <= 256, KPD_ERR_INVALID, "n_hidden_scalars=%d outside 1 .. 256", cfg->n_hidden_scalars);
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
    T->V = cfg->vector_size;
    *out = T;
    return KPD_OK;
}

extern "C" void kpd_gvp_trainer_destroy(kpd_gvp_trainer *T) {
    if (!T) return;
    T->ws.release();
    T->wide.release();
    T->release_scratch();
    if (T->store_base) (void)hipFree(T->store_base);
    T->release_fused();
    delete T;
}

extern "C" kpd_status kpd_gvp_trainer_bind(kpd_gvp_trainer *T, const char *name, const float *weight, float *grad, const int64_t *shape,
                                           int32_t ndim) {
    KPD_REQUIRE(T && name && shape && (ndim == 1 || ndim == 2), KPD_ERR_INVALID, "bad argument");
    T->pack_dirty = true;                       // (the fused forward's descriptor table names parameter addresses)
    if (T->V != VC && weight) {
        // vector_size < 16: the tensors of a GVP whose axes count vector channels (Wh [v_in, h], Wu [h, v_out], the |Vh| block of
        // to_feats_out [S_out, S_in + h], the gates [v_out, S_out]; h = max(v_in, v_out), models/gvp.py:60-87) are trained through their
        // 16-channel zero-padded form.  The first message GVP reads [x_diff | v_src] (v_in = V + 1 -> 17), the last noise GVP emits one vector.
        const std::vector<std::string> tk = split_name(name);
        int at = -1;                                         // index of the GVP's position token
        bool msg0 = false, last = false;
        if (tk.size() >= 7 && tk[1] == "conv_layers" && (tk[3] == "edge_message_fns" || tk[3] == "node_update_fns")) {
            at = 5;
            msg0 = tk[3] == "edge_message_fns" && tk[5] == "0";
        } else if (tk.size() >= 5 && tk[1] == "noise_predictor" && tk[2] == "gvps") {
            at = 3;
            last = atoi(tk[3].c_str()) == T->cfg.n_noise_gvps - 1;
        }
        if (at >= 0) {
            const int V = T->V;
            const std::vector<AxisSeg> vi = msg0 ? std::vector<AxisSeg>{{V + 1, VH}} : std::vector<AxisSeg>{{V, VC}};
            const std::vector<AxisSeg> vo = last ? std::vector<AxisSeg>{{1, 1}} : std::vector<AxisSeg>{{V, VC}};
            const std::vector<AxisSeg> h = msg0 ? std::vector<AxisSeg>{{V + 1, VH}} : std::vector<AxisSeg>{{V, VC}};
            const int h_ref = msg0 ? V + 1 : V;
            const std::string &leaf = tk[at + 1];
            std::vector<AxisSeg> rows, cols;
            if (leaf == "Wh") { rows = vi; cols = h; }
            else if (leaf == "Wu") { rows = h; cols = vo; }
            else if (leaf == "to_feats_out" && tk.back() == "weight" && ndim == 2 && shape[1] > h_ref) {
                rows = {{(int)shape[0], (int)shape[0]}};
                cols = {{(int)shape[1] - h_ref, (int)shape[1] - h_ref}, h[0]};
            } else if (leaf == "scalar_to_vector_gates" && !last) {
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
        F(T->part, GVP_PART_FLOATS);
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
    T->part_floats = GVP_PART_FLOATS;
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
        // the fused message forward wants the kept activations (it writes them) and the 256-wide kernels
        T->release_fused();
        static const bool want_fused = tool_env_int("KPD_TRAIN_FUSED", 1) != 0;      // TOOLS build: A/B against the per-GVP path
        if (T->store && S == 256 && want_fused) {
            size_t floats = 0;
            for (int conv = 0; conv < L; ++conv)
                for (int et = 0; et < 4; ++et)
                    if (conv_uses(T, conv, et)) floats += pack_floats_per_chain(nm);
            const int nu = c.n_update_gvps;
            for (int conv = 0; conv < L; ++conv)
                for (int nt = 0; nt < 2; ++nt)
                    if (conv_uses(T, conv, nt == 0 ? 0 : 2)) floats += (size_t)nu * PK_NODE_GVP;
            const size_t pack_floats = floats;
            // kept node-update activations per (conv, destination node type): s1, v1, sb, vb and the seven buffers of every update GVP
            std::vector<size_t> off_n((size_t)L * 2, 0);
            const size_t node_per_row = 2 * (256 + 48) + (size_t)nu * (3 * 16 + 3 * 16 + 16 + 256 + 256 + 16 + 3 * 16);
            for (int conv = 0; conv < L; ++conv)
                for (int nt = 0; nt < 2; ++nt)
                    if (conv_uses(T, conv, nt == 0 ? 0 : 2)) { off_n[(size_t)conv * 2 + nt] = floats; floats += (size_t)nn[nt] * node_per_row; }
            size_t off_P[4], off_main[4], off_cont[4], off_vmain[4], off_vcont[4];
            for (int et = 0; et < 4; ++et) {
                const size_t tiles = (size_t)cap_et[et] / TM + 2;
                floats = (floats + 63) & ~size_t(63);
                off_P[et] = floats; floats += (size_t)nn[kSrc[et]] * 256;
                off_main[et] = floats; floats += (size_t)nn[kDst[et]] * 256;
                off_cont[et] = floats; floats += tiles * 256;
                off_vmain[et] = floats; floats += (size_t)nn[kDst[et]] * 48;
                off_vcont[et] = floats; floats += tiles * 48;
            }
            size_t off_b[4];
            const size_t bwd_per_edge = (size_t)nm * (256 + 16 + 48 + 17) + 16;
            for (int et = 0; et < 4; ++et) {
                floats = (floats + 63) & ~size_t(63);          // (256-byte aligned blocks: the kernels store float4)
                off_b[et] = floats;
                floats += (((size_t)cap_et[et] * bwd_per_edge + 16 * (size_t)nm) + 63) & ~size_t(63);
            }
            size_t off_U[4], off_dvin[4];
            for (int et = 0; et < 4; ++et) {
                off_U[et] = floats; floats += (size_t)nn[kSrc[et]] * 256;
                off_dvin[et] = floats; floats += (((size_t)cap_et[et] * 3 * VH) + 63) & ~size_t(63);
            }
            const size_t off_vp = floats;
            floats += 16 * VEC_PART_REGION;
            size_t off_nb[2];
            for (int nt = 0; nt < 2; ++nt) { off_nb[nt] = floats; floats += (size_t)nn[nt] * nu * (256 + 16 + 48 + 16); }
            std::vector<GvpTrainSlot> hs((size_t)L * 4);
            memset(hs.data(), 0, hs.size() * sizeof(GvpTrainSlot));
            for (size_t i = 0; i < hs.size(); ++i) {
                const kpd_gvp_trainer::MsgSlot &sl = T->slots[i];
                hs[i].unit = sl.unit; hs[i].rbf = sl.rbf; hs[i].vin = sl.vin;
                for (int j = 0; j < nm; ++j) {
                    const GvpBuf &b = sl.gb[j];
                    hs[i].g[j] = GvpTrainGvp{b.Vh, b.Vu, b.sh, b.pre, b.s, b.gate, b.V};
                }
            }
            if (hipMalloc(reinterpret_cast<void **>(&T->pack_base), floats * 4) == hipSuccess &&
                hipMalloc(reinterpret_cast<void **>(&T->slots_dev), hs.size() * sizeof(GvpTrainSlot)) == hipSuccess &&
                hipMalloc(reinterpret_cast<void **>(&T->bslots_dev), 4 * sizeof(GvpBwdSlot)) == hipSuccess) {
                for (int et = 0; et < 4; ++et) {
                    float *p = T->pack_base + off_b[et];
                    const size_t E = cap_et[et];
                    GvpBwdSlot &bs = T->bslots[et];
                    memset(&bs, 0, sizeof(bs));
                    for (int j = 0; j < nm; ++j) {
                        bs.g[j].dpre = p; p += E * 256;
                        bs.g[j].dgate = p; p += E * 16;
                        bs.g[j].dVu = p; p += E * 48;
                        bs.g[j].dsh = p; p += (E * 17 + 3) & ~size_t(3);          // (keeps the next block 16-byte aligned)
                    }
                    bs.drbf = p;
                }
                KPD_HIP(hipMemcpy(T->bslots_dev, T->bslots, 4 * sizeof(GvpBwdSlot), hipMemcpyHostToDevice));
                for (int et = 0; et < 4; ++et) { T->Uet[et] = T->pack_base + off_U[et]; T->dvin_et[et] = T->pack_base + off_dvin[et]; }
                T->vpart = T->pack_base + off_vp;
                T->n_vred = 0;
                for (int nt = 0; nt < 2; ++nt) {
                    float *p = T->pack_base + off_nb[nt];
                    const size_t n = nn[nt];
                    for (int j = 0; j < nu; ++j) {
                        GvpBwdGvp &o = T->nbslots[nt][j];
                        o.dpre = p; p += n * 256; o.dgate = p; p += n * 16; o.dVu = p; p += n * 48; o.dsh = p; p += n * 16;
                    }
                }
                T->nslots.assign((size_t)L * 2, kpd_gvp_trainer::NodeSlot());
                for (int conv = 0; conv < L; ++conv)
                    for (int nt = 0; nt < 2; ++nt) {
                        if (!conv_uses(T, conv, nt == 0 ? 0 : 2)) continue;
                        float *p = T->pack_base + off_n[(size_t)conv * 2 + nt];
                        const size_t n = nn[nt];
                        kpd_gvp_trainer::NodeSlot &ns = T->nslots[(size_t)conv * 2 + nt];
                        ns.s1 = p; p += n * 256; ns.v1 = p; p += n * 48; ns.sb = p; p += n * 256; ns.vb = p; p += n * 48;
                        for (int j = 0; j < nu; ++j) {
                            GvpBuf &b = ns.gb[j];
                            b.Vh = p; p += n * 48; b.Vu = p; p += n * 48; b.sh = p; p += n * 16; b.pre = p; p += n * 256; b.s = p; p += n * 256;
                            b.gate = p; p += n * 16; b.V = p; p += n * 48;
                        }
                    }
                KPD_HIP(hipMemset(T->pack_base, 0, pack_floats * 4));           // zero biases of the head GVPs, unused fragment tiles
                // (debug, KPD_POISON: every activation / piece / gradient buffer behind the packs starts as NaNs, and so do the kept message
                //  activations: a read of something this step has not written shows -- tests/test_gvp_train_gpu.py)
                if (poison_level() >= 1) {
                    poison_floats(T->pack_base + pack_floats, (floats - pack_floats) * 4);
                    poison_floats(T->store_base, total);
                }
                KPD_HIP(hipMemcpy(T->slots_dev, hs.data(), hs.size() * sizeof(GvpTrainSlot), hipMemcpyHostToDevice));
                for (int et = 0; et < 4; ++et) {
                    T->Psrc[et] = T->pack_base + off_P[et];
                    T->ms_main[et] = T->pack_base + off_main[et]; T->ms_cont[et] = T->pack_base + off_cont[et];
                    T->mv_main[et] = T->pack_base + off_vmain[et]; T->mv_cont[et] = T->pack_base + off_vcont[et];
                }
                T->fused = true;
                T->pack_dirty = true;
            } else {
                (void)hipGetLastError();
                T->release_fused();
            }
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
    KPD_TRY(wide_run(T, 0));                   // vector_size < 16: stage the current weights in the engine's widths
    T->bt = *bt;
    T->t_dev = t_dev;
    T->n[0] = bt->n_lig; T->n[1] = bt->n_kp;
    KPD_TRY(launch_node_graph_index(bt->lig_ptr, bt->B, bt->n_lig, T->bidx[0], st));
    KPD_TRY(launch_node_graph_index(bt->kp_ptr, bt->B, bt->n_kp, T->bidx[1], st));
    KPD_TRY(launch_lig_graph(bt, c.ll_cutoff, c.ll_k, c.kl_cutoff, c.kl_k, &T->lg, T->ll_deg, T->ll_off, T->kl_off, T->kl_pg, st));
    // (tile tables for the fused message forward: all four edge types at meta[0..8], the ligand-bound pair of the last conv at meta[16..24])
    KPD_TRY(launch_egnn_meta(T->lg.counts, bt->n_kk, 0xF, 0x3, bt->lig_ptr, bt->kp_ptr, T->lg.ll_per_graph, bt->kk_rowptr, bt->B, T->kl_off,
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
    hipLaunchKernelGGL(k_v_transpose, grid1((long long)bt->n_kp * 3 * VC), dim3(256), 0, st, bt->kp_v, (long long)bt->n_kp, 1, T->V, T->vs[1][0]);
    KPD_LAUNCH_CHECK();
    if (T->fused) KPD_TRY(pack_chains(T));
    for (int i = 0; i < c.n_convs; ++i) KPD_TRY(conv_fwd(T, i));
    KPD_TRY(noise_fwd(T, eps_h, eps_x));
    T->have_forward = true;
    return KPD_OK;
}

extern "C" kpd_status kpd_gvp_trainer_last_counts(kpd_gvp_trainer *T, int32_t out[4]) {
    KPD_REQUIRE(T && out, KPD_ERR_INVALID, "null argument");
    for (int et = 0; et < 4; ++et) out[et] = T->E[et];
    return KPD_OK;
}

extern "C" kpd_status kpd_gvp_trainer_message_path(kpd_gvp_trainer *T, int32_t *path) {
    KPD_REQUIRE(T && path, KPD_ERR_INVALID, "null argument");
    *path = T->fused ? 1 : 0;
    return KPD_OK;
}

extern "C" kpd_status kpd_gvp_trainer_backward(kpd_gvp_trainer *T, const float *d_eps_h, const float *d_eps_x, float *d_lig_h,
                                               float *d_kp_h, float *d_kp_v, float *d_lig_x, float *d_kp_x, void *stream) {
    KPD_REQUIRE(T && d_eps_h && d_eps_x, KPD_ERR_INVALID, "null argument");
    KPD_REQUIRE(T->have_forward, KPD_ERR_STATE, "kpd_gvp_trainer_backward before kpd_gvp_trainer_forward");
    const kpd_gvp_config &c = T->cfg;
    hipStream_t st = static_cast<hipStream_t>(stream);
    T->st = st;
    KPD_TRY(wide_run(T, 1));                   // vector_size < 16: zero the wide gradients
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
    // noise block: recomputed when the per-GVP conv path ran in between (it reuses these buffers); the chained path leaves them alone
    if (!T->fused) KPD_TRY(noise_fwd(T, nullptr, nullptr));
    {
        const std::string p = "noise_predictor.noise_predictor";
        Param W, b;
        KPD_TRY(param(T, p + ".to_scalar_output.weight", F, kHeadS, &W));
        KPD_TRY(param(T, p + ".to_scalar_output.bias", F, 1, &b));
        KPD_TRY(grad_gemm(T, F, kHeadS, nl, d_eps_h, F, T->gb[nn - 1].s, kHeadS, W.g, kHeadS, b.g));
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
    KPD_TRY(wide_run(T, 2));                   // vector_size < 16: add the wide gradients into the caller's tensors (reference shapes)
    if (d_kp_v) {
        hipLaunchKernelGGL(k_v_transpose, grid1((long long)nk * 3 * VC), dim3(256), 0, st, T->gv[cur][1], (long long)nk, 0, T->V, d_kp_v);
        KPD_LAUNCH_CHECK();
    }
    if (d_lig_x) KPD_HIP(hipMemcpyAsync(d_lig_x, T->gx[0], (size_t)nl * 12, hipMemcpyDeviceToDevice, st));
    if (d_kp_x) KPD_HIP(hipMemcpyAsync(d_kp_x, T->gx[1], (size_t)nk * 12, hipMemcpyDeviceToDevice, st));
    T->have_forward = false;
    return KPD_OK;
}
