// GVP denoiser engine behind the kpd_gvp_* C ABI (include/kpd.h).  Replaces
// LigRecDynamicsGVP.forward (models/dynamics_gvp.py:149-199) as a whole: encoders, per-step edge
// build, the GVPMultiEdgeConv stack (models/gvp.py:459-551) and the NoisePredictionBlock.
#include <cstring>
#include <string.h>

#include <map>
#include <set>
#include <string>
#include <vector>

#include "egnn_kernels.h"
#include "gvp_host.h"

using namespace kpd;

namespace {

const int kSrcNtG[4] = {0, 1, 0, 1};     // ll, kl, lk, kk  (0 = lig, 1 = kp)
const int kDstNtG[4] = {0, 0, 1, 1};
const char *kCanon[4] = {"lig_ll_lig", "kp_kl_lig", "lig_lk_kp", "kp_kk_kp"};
const char *kNtNameG[2] = {"lig", "kp"};

}  // namespace

struct kpd_gvp {
    kpd_gvp_config cfg;
    int S, V;                                   // S: width of the kernels' scalar layout (128 or 256)
    int St = 0;                                 // n_hidden_scalars of the model (<= S; the S - St trailing features are padding)
    int Vt = GV;                                // vector_size of the model (<= 16 channels of the kernels' layout, likewise)
    Arena warena, ws;
    // weights
    std::vector<std::vector<std::vector<HostGvp>>> msg;   // [conv][et][j]
    std::vector<std::vector<std::vector<HostGvp>>> upd;   // [conv][nt][j]
    std::vector<std::vector<float *>> ln1w, ln1b, ln2w, ln2b;   // [conv][nt]
    std::vector<HostGvp> noise;
    float *enc_W[2], *enc_b[2], *enc_lw[2], *enc_lb[2];
    float *out_W, *out_b;
    std::set<std::string> expected, loaded;
    bool committed = false;
    bool f16_ok = true;                       // the committed weights fit the f16 planes (pack.hip range guard)
    int gemm_mode = 0;                        // 0 exact fp32; 1 f16x2 split in the message chain (KPD_GEMM=f16x2, "gemm=f16x2")
    int debug_convs = -1;
    unsigned long long *stamps = nullptr;     // device [32], diagnostics
    int coop_rows = 0;                        // row limit of the cooperative node-side kernels (0: default; kpd_gvp_debug_state "coop_rows=N")
    // optional HIP-event timing of the dominant kernel (k_gvp_chain), for bench.py's roofline
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;
    size_t prof_used = 0;
    // workspace
    int cap_B = 0, cap_lig = 0, cap_kp = 0, cap_kk = 0, cap_maxlig = 0, cap_maxkp = 0;
    float *s[2], *v[2], *s_tmp[2];
    float *Psrc[4];
    float *ms_main[4], *ms_cont[4], *mv_main[4], *mv_cont[4];
    int *bidx[2];
    float *z[2];
    int *meta4, *meta2, *ll_deg, *ll_off, *kl_off, *kl_pg;
    kpd_lig_graph lg;

    int n_et(int conv) const { return (cfg.update_kp && conv != cfg.n_convs - 1) ? 4 : 2; }
};

extern "C" kpd_status kpd_gvp_create(const kpd_gvp_config *cfg, kpd_gvp **out) {
    KPD_REQUIRE(cfg && out, KPD_ERR_INVALID, "null argument");
    KPD_REQUIRE(cfg->vector_size >= 1 && cfg->vector_size <= GV, KPD_ERR_INVALID, "vector_size=%d: supported sizes are 1 .. 16", cfg->vector_size);
    KPD_REQUIRE(cfg->n_hidden_scalars >= 1 && cfg->n_hidden_scalars <= 256, KPD_ERR_INVALID,
                "n_hidden_scalars=%d: supported widths are 1 .. 256", cfg->n_hidden_scalars);
    KPD_REQUIRE(cfg->ll_k >= 0 && cfg->ll_k <= KL_KMAX, KPD_ERR_INVALID, "ll_k=%d outside 0..%d (0 = radius graph)", cfg->ll_k, KL_KMAX);
    KPD_REQUIRE(cfg->kl_k >= 0 && cfg->kl_k <= KL_KMAX, KPD_ERR_INVALID, "kl_k=%d outside 0..%d (0 = radius graph)", cfg->kl_k, KL_KMAX);
    KPD_REQUIRE(cfg->kl_k > 0 || cfg->kl_cutoff > 0.0f, KPD_ERR_INVALID, "kl_k = 0 needs graph_cutoffs['kl'] > 0");
    KPD_REQUIRE(cfg->ll_k > 0 || cfg->ll_cutoff > 0.0f, KPD_ERR_INVALID, "ll_k = 0 needs graph_cutoffs['ll'] > 0");
    KPD_REQUIRE(cfg->n_convs >= 1 && cfg->n_convs <= 32, KPD_ERR_INVALID, "n_convs=%d", cfg->n_convs);
    KPD_REQUIRE(cfg->update_kp || cfg->n_convs == 1, KPD_ERR_INVALID,
                "update_kp=0 with more than one convolution cannot run in the reference (gvp.py:501, 536)");
    KPD_REQUIRE(cfg->n_message_gvps >= 1 && cfg->n_message_gvps <= GVP_MAX_CHAIN && cfg->n_update_gvps >= 1 &&
                    cfg->n_update_gvps <= GVP_MAX_CHAIN && cfg->n_noise_gvps >= 1 && cfg->n_noise_gvps <= GVP_MAX_CHAIN,
                KPD_ERR_INVALID, "GVP chain lengths must be within 1..%d", GVP_MAX_CHAIN);
    KPD_REQUIRE(cfg->n_lig_scalars >= 1 && cfg->n_lig_scalars <= 64 && cfg->n_kp_scalars >= 1 && cfg->n_kp_scalars <= 256,
                KPD_ERR_INVALID, "feature widths out of range");
    KPD_REQUIRE(cfg->message_norm_mode >= 0 && cfg->message_norm_mode <= 2, KPD_ERR_INVALID, "message_norm_mode");
    KPD_TRY(egnn_kernels_init());
    kpd_gvp *m = new kpd_gvp();
    m->cfg = *cfg;
    // any n_hidden_scalars up to 256 runs on the 128- or 256-wide kernels: weights are packed with zero rows / columns for the padding
    // features (gvp_host.hip), which therefore stay exactly 0 through SiLU, gates and residuals; the two LayerNorms take the true width
    m->St = cfg->n_hidden_scalars;
    m->S = m->St <= 128 ? 128 : 256;
    m->V = GV;
    m->Vt = cfg->vector_size;
    if (const char *e = getenv("KPD_GEMM")) m->gemm_mode = (!strcmp(e, "f16x2") && m->S == 256) ? 1 : 0;
    const int S = m->S, C = cfg->n_convs;
    size_t per_gvp = gvp_arena_bytes(S);
    size_t bytes = per_gvp * ((size_t)C * (4 * cfg->n_message_gvps + 2 * cfg->n_update_gvps) + cfg->n_noise_gvps) +
                   (size_t)C * 2 * 4 * (S * 4 + 256) + (size_t)2 * (S * 260 + 3 * S) * 4 + 64 * 64 * 4 + (1 << 20);
    kpd_status st = m->warena.reserve(bytes);
    if (st != KPD_OK) {
        delete m;
        return st;
    }
    m->warena.poison_at = 2;          // packed weights: poisoned only at KPD_POISON >= 2 (engine.h)
    Arena &A = m->warena;
    m->msg.resize(C); m->upd.resize(C); m->ln1w.resize(C); m->ln1b.resize(C); m->ln2w.resize(C); m->ln2b.resize(C);
    for (int i = 0; i < C; ++i) {
        const std::string pre = "noise_predictor.conv_layers." + std::to_string(i) + ".";
        const int net = m->n_et(i), nnt = net == 4 ? 2 : 1;
        m->msg[i].resize(4); m->upd[i].resize(2);
        m->ln1w[i].assign(2, nullptr); m->ln1b[i].assign(2, nullptr); m->ln2w[i].assign(2, nullptr); m->ln2b[i].assign(2, nullptr);
        for (int et = 0; et < net; ++et) {
            m->msg[i][et].resize(cfg->n_message_gvps);
            for (int j = 0; j < cfg->n_message_gvps; ++j) {
                HostGvp &g = m->msg[i][et][j];
                g.vin = j == 0 ? GV + 1 : GV; g.vout = GV;
                g.s_in = j == 0 ? S + 16 : S; g.sout = S;
                g.split = j == 0 ? SPLIT_SRC : SPLIT_NONE; g.S = S; g.cut = S - m->St; g.vcut = GV - m->Vt;
                g.chain_pos = j;
                alloc_gvp(A, g, m->expected, pre + "edge_message_fns." + kCanon[et] + "." + std::to_string(j));
            }
        }
        for (int nt = 0; nt < nnt; ++nt) {
            m->upd[i][nt].resize(cfg->n_update_gvps);
            for (int j = 0; j < cfg->n_update_gvps; ++j) {
                HostGvp &g = m->upd[i][nt][j];
                g.vin = GV; g.vout = GV; g.s_in = S; g.sout = S; g.S = S; g.cut = S - m->St; g.vcut = GV - m->Vt;
                g.chain_pos = 1;            // register-chained node kernel: same form as a non-head message GVP
                alloc_gvp(A, g, m->expected, pre + "node_update_fns." + kNtNameG[nt] + "." + std::to_string(j));
            }
            m->ln1w[i][nt] = A.take<float>(S); m->ln1b[i][nt] = A.take<float>(S);
            m->ln2w[i][nt] = A.take<float>(S); m->ln2b[i][nt] = A.take<float>(S);
            for (const char *s : {".feat_norm.weight", ".feat_norm.bias"}) {
                m->expected.insert(pre + "message_layer_norms." + kNtNameG[nt] + s);
                m->expected.insert(pre + "update_layer_norms." + kNtNameG[nt] + s);
            }
        }
    }
    m->noise.resize(cfg->n_noise_gvps);
    for (int j = 0; j < cfg->n_noise_gvps; ++j) {
        HostGvp &g = m->noise[j];
        const bool last = j == cfg->n_noise_gvps - 1;
        g.vin = GV; g.vout = last ? 1 : GV; g.s_in = S; g.sout = last ? 64 : S; g.S = S; g.cut = S - m->St; g.vcut = GV - m->Vt;
        g.vec_sigmoid = last ? 0 : 1;
        g.chain_pos = 1;                // register-chained noise head (gvp_chain.hip)
        alloc_gvp(A, g, m->expected, "noise_predictor.noise_predictor.gvps." + std::to_string(j));
    }
    const int fin[2] = {cfg->n_lig_scalars + 1, cfg->n_kp_scalars + 1};
    for (int nt = 0; nt < 2; ++nt) {
        m->enc_W[nt] = A.take<float>((size_t)S * fin[nt]); m->enc_b[nt] = A.take<float>(S);
        m->enc_lw[nt] = A.take<float>(S); m->enc_lb[nt] = A.take<float>(S);
        const std::string e = std::string(kNtNameG[nt]) + "_encoder.";
        for (const char *s : {"0.weight", "0.bias", "2.weight", "2.bias"}) m->expected.insert(e + s);
    }
    m->out_W = A.take<float>((size_t)cfg->n_lig_scalars * 64);
    m->out_b = A.take<float>(cfg->n_lig_scalars);
    m->expected.insert("noise_predictor.noise_predictor.to_scalar_output.weight");
    m->expected.insert("noise_predictor.noise_predictor.to_scalar_output.bias");
    if (!m->out_b) {
        set_error("gvp weight arena too small (internal sizing error)");
        kpd_gvp_destroy(m);
        return KPD_ERR_HIP;
    }
    *out = m;
    return KPD_OK;
}

extern "C" void kpd_gvp_destroy(kpd_gvp *m) {
    if (!m) return;
    for (hipEvent_t e : m->prof_ev) (void)hipEventDestroy(e);
    if (m->stamps) (void)hipFree(m->stamps);
    m->warena.release();
    m->ws.release();
    delete m;
}

extern "C" kpd_status kpd_gvp_load_weight(kpd_gvp *m, const char *name, const float *w, const int64_t *shape, int32_t ndim,
                                          void *stream) {
    KPD_REQUIRE(m && name && w && shape, KPD_ERR_INVALID, "null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const std::string nm(name);
    if (!m->expected.count(nm)) {
        set_error("unknown or unused weight name '%s' for this configuration", name);
        return KPD_ERR_WEIGHTS;
    }
    const int S = m->S, St = m->St;
    const std::vector<std::string> tk = split_dots(nm);
    auto tail_from = [&](size_t i) {
        std::string t;
        for (size_t k = i; k < tk.size(); ++k) t += (k > i ? "." : "") + tk[k];
        return t;
    };
    if (tk[0] == "lig_encoder" || tk[0] == "kp_encoder") {
        const int nt = tk[0] == "lig_encoder" ? 0 : 1;
        const int fin = (nt == 0 ? m->cfg.n_lig_scalars : m->cfg.n_kp_scalars) + 1;
        const bool is_w = tk[2] == "weight";
        if (tk[1] == "0") {       // rows St .. S - 1 stay zero
            if (is_w) { KPD_TRY(want_shape(name, shape, ndim, {St, fin})); KPD_TRY(copy_pad(w, St * fin, m->enc_W[nt], S * fin, st)); }
            else { KPD_TRY(want_shape(name, shape, ndim, {St})); KPD_TRY(copy_pad(w, St, m->enc_b[nt], S, st)); }
        } else {
            KPD_TRY(want_shape(name, shape, ndim, {St}));
            KPD_TRY(copy_pad(w, St, is_w ? m->enc_lw[nt] : m->enc_lb[nt], S, st));
        }
    } else if (tk[1] == "noise_predictor") {
        if (tk[2] == "to_scalar_output") {
            const int F = m->cfg.n_lig_scalars;
            if (tk[3] == "weight") { KPD_TRY(want_shape(name, shape, ndim, {F, 64})); KPD_TRY(copy_pad(w, F * 64, m->out_W, F * 64, st)); }
            else { KPD_TRY(want_shape(name, shape, ndim, {F})); KPD_TRY(copy_pad(w, F, m->out_b, F, st)); }
        } else {   // noise_predictor.noise_predictor.gvps.<j>.<param>
            const int j = atoi(tk[3].c_str());
            KPD_TRY(load_gvp_tensor(m->noise[j], tail_from(4), name, w, shape, ndim, st));
        }
    } else {       // noise_predictor.conv_layers.<i>.<block>.<key>...
        const int i = atoi(tk[2].c_str());
        const std::string &blk = tk[3];
        if (blk == "edge_message_fns") {
            int et = -1;
            for (int e = 0; e < 4; ++e)
                if (tk[4] == kCanon[e]) et = e;
            KPD_TRY(load_gvp_tensor(m->msg[i][et][atoi(tk[5].c_str())], tail_from(6), name, w, shape, ndim, st));
        } else if (blk == "node_update_fns") {
            const int nt = tk[4] == "lig" ? 0 : 1;
            KPD_TRY(load_gvp_tensor(m->upd[i][nt][atoi(tk[5].c_str())], tail_from(6), name, w, shape, ndim, st));
        } else {   // message_layer_norms / update_layer_norms .<nt>.feat_norm.<param>
            const int nt = tk[4] == "lig" ? 0 : 1;
            const bool is_w = tk[6] == "weight";
            KPD_TRY(want_shape(name, shape, ndim, {St}));
            float *dst = blk == "message_layer_norms" ? (is_w ? m->ln1w[i][nt] : m->ln1b[i][nt])
                                                      : (is_w ? m->ln2w[i][nt] : m->ln2b[i][nt]);
            KPD_TRY(copy_pad(w, St, dst, S, st));
        }
    }
    m->loaded.insert(nm);
    m->committed = false;
    return KPD_OK;
}

extern "C" kpd_status kpd_gvp_commit(kpd_gvp *m) {
    KPD_REQUIRE(m, KPD_ERR_INVALID, "null handle");
    for (const std::string &n : m->expected)
        if (!m->loaded.count(n)) {
            set_error("weight '%s' was never loaded (%zu of %zu loaded)", n.c_str(), m->loaded.size(), m->expected.size());
            return KPD_ERR_WEIGHTS;
        }
    // f16x2 mode: the 256 -> 256 message GVPs behind the head of every chain, re-packed from their finished fp32 chunks
    F16PackScope f16_scope;
    for (auto &conv : m->msg)
        for (auto &et : conv)
            for (HostGvp &g : et)
                {
                    if (g.has_h() && g.chain_h) KPD_TRY(pack_gvp_chain_h(g.chain, g.chain_h, g.chain_pos == 0, g.n_ht(), nullptr));
                    if (g.wproj_h) KPD_TRY(pack_gvp_proj_h(g.wproj, g.wproj_h, nullptr));
                    if (g.wproj_dst_h) KPD_TRY(pack_gvp_proj_h(g.wproj_dst, g.wproj_dst_h, nullptr));
                }
    for (auto &conv : m->upd)
        for (auto &nt : conv)
            for (HostGvp &g : nt)
                if (g.has_h() && g.chain_h) KPD_TRY(pack_gvp_chain_h(g.chain, g.chain_h, 0, g.n_ht(), nullptr));
    for (HostGvp &g : m->noise)
        if (g.has_h() && g.chain_h) KPD_TRY(pack_gvp_chain_h(g.chain, g.chain_h, 0, g.n_ht(), nullptr));
    m->f16_ok = !f16_scope.overflowed();           // (synchronises the device)
    KPD_REQUIRE(m->f16_ok || m->gemm_mode == 0, KPD_ERR_WEIGHTS, "%s", F16_RANGE_ERROR);       // KPD_GEMM=f16x2 asked for it
    KPD_HIP(hipDeviceSynchronize());
    m->committed = true;
    return KPD_OK;
}

extern "C" kpd_status kpd_gvp_reserve(kpd_gvp *m, int32_t max_B, int32_t max_n_lig, int32_t max_n_kp, int32_t max_n_kk,
                                      int32_t max_lig_pg, int32_t max_kp_pg) {
    KPD_REQUIRE(m, KPD_ERR_INVALID, "null handle");
    KPD_REQUIRE(max_B >= 1 && max_n_lig >= 1 && max_n_kp >= 1 && max_n_kk >= 0 && max_lig_pg >= 1 && max_kp_pg >= 1,
                KPD_ERR_INVALID, "reserve: non-positive size");
    if (max_B <= m->cap_B && max_n_lig <= m->cap_lig && max_n_kp <= m->cap_kp && max_n_kk <= m->cap_kk &&
        max_lig_pg <= m->cap_maxlig && max_kp_pg <= m->cap_maxkp)
        return KPD_OK;
    max_B = std::max(max_B, m->cap_B); max_n_lig = std::max(max_n_lig, m->cap_lig); max_n_kp = std::max(max_n_kp, m->cap_kp);
    max_n_kk = std::max(max_n_kk, m->cap_kk); max_lig_pg = std::max(max_lig_pg, m->cap_maxlig); max_kp_pg = std::max(max_kp_pg, m->cap_maxkp);
    const long cap_ll_l = (long)max_n_lig * std::min(max_lig_pg - 1, m->cfg.ll_k > 0 ? m->cfg.ll_k : 200);
    const long cap_kl_l = (long)max_n_kp * (m->cfg.kl_k > 0 ? m->cfg.kl_k : std::min(max_lig_pg, 100));
    KPD_REQUIRE(cap_ll_l < (1l << 30) && cap_kl_l < (1l << 30), KPD_ERR_CAPACITY, "edge capacity overflows int32");
    const int cap_ll = std::max<long>(cap_ll_l, 1), cap_kl = std::max<long>(cap_kl_l, 1);
    const int E_cap[4] = {cap_ll, cap_kl, cap_kl, std::max(max_n_kk, 1)};
    const int S = m->S, n[2] = {max_n_lig, max_n_kp};
    int tiles[4];
    size_t bytes = 1 << 20;
    auto add = [&](size_t cnt) { bytes += ((cnt * 4 + 255) & ~size_t(255)); };
    for (int nt = 0; nt < 2; ++nt) { add((size_t)n[nt] * S); add((size_t)n[nt] * S); add((size_t)n[nt] * 48); add(n[nt]); add(max_B); }
    for (int et = 0; et < 4; ++et) {
        tiles[et] = cdiv(E_cap[et], TM) + 1;
        add((size_t)n[kSrcNtG[et]] * S);
        add((size_t)n[kDstNtG[et]] * S); add((size_t)tiles[et] * S); add((size_t)n[kDstNtG[et]] * 48); add((size_t)tiles[et] * 48);
    }
    add(32); add(16); add(max_n_lig); add(max_B + 1); add(max_B + 1); add(max_B + 2);
    add(cap_ll); add(cap_ll); add(max_n_lig + 1);
    for (int i = 0; i < 4; ++i) add(cap_kl);
    add(max_n_lig + 1); add(max_n_kp + 1); add(max_B); add(8);
    KPD_TRY(m->ws.reserve(bytes));
    Arena &W = m->ws;
    for (int nt = 0; nt < 2; ++nt) {
        m->s[nt] = W.take<float>((size_t)n[nt] * S); m->s_tmp[nt] = W.take<float>((size_t)n[nt] * S);
        m->v[nt] = W.take<float>((size_t)n[nt] * 48);
        m->bidx[nt] = W.take<int>(n[nt]); m->z[nt] = W.take<float>(max_B);
    }
    for (int et = 0; et < 4; ++et) {
        m->Psrc[et] = W.take<float>((size_t)n[kSrcNtG[et]] * S);
        m->ms_main[et] = W.take<float>((size_t)n[kDstNtG[et]] * S); m->ms_cont[et] = W.take<float>((size_t)tiles[et] * S);
        m->mv_main[et] = W.take<float>((size_t)n[kDstNtG[et]] * 48); m->mv_cont[et] = W.take<float>((size_t)tiles[et] * 48);
    }
    m->meta4 = W.take<int>(32); m->meta2 = m->meta4 + 16;    // one k_egnn_meta launch fills both tables
    m->ll_deg = W.take<int>(max_n_lig); m->ll_off = W.take<int>(max_B + 1); m->kl_off = W.take<int>(max_B + 1);
    m->kl_pg = W.take<int>(max_B + 2);
    kpd_lig_graph &g = m->lg;
    g.cap_ll = cap_ll; g.cap_kl = cap_kl;
    g.ll_src = W.take<int>(cap_ll); g.ll_dst = W.take<int>(cap_ll); g.ll_rowptr = W.take<int>(max_n_lig + 1);
    g.kl_src = W.take<int>(cap_kl); g.kl_dst = W.take<int>(cap_kl); g.kl_rowptr = W.take<int>(max_n_lig + 1);
    g.lk_src = W.take<int>(cap_kl); g.lk_dst = W.take<int>(cap_kl); g.lk_rowptr = W.take<int>(max_n_kp + 1);
    g.ll_per_graph = W.take<int>(max_B);
    g.counts = W.take<int>(8);
    KPD_REQUIRE(g.counts != nullptr, KPD_ERR_HIP, "gvp workspace arena too small (internal sizing error)");
    m->cap_B = max_B; m->cap_lig = max_n_lig; m->cap_kp = max_n_kp; m->cap_kk = max_n_kk;
    m->cap_maxlig = max_lig_pg; m->cap_maxkp = max_kp_pg;
    return KPD_OK;
}

extern "C" kpd_status kpd_gvp_forward(kpd_gvp *m, const kpd_batch *bt, const float *t_dev, float *eps_h, float *eps_x,
                                      void *stream) {
    KPD_REQUIRE(m && bt && t_dev && eps_h && eps_x, KPD_ERR_INVALID, "null argument");
    KPD_REQUIRE(m->committed, KPD_ERR_STATE, "kpd_gvp_forward before kpd_gvp_commit");
    KPD_REQUIRE(bt->B >= 1 && bt->n_lig >= 1 && bt->n_kp >= 1, KPD_ERR_INVALID, "empty batch");
    KPD_REQUIRE(bt->kp_v, KPD_ERR_INVALID, "kp_v (keypoint vector features v_0) missing");
    KPD_REQUIRE(bt->B <= m->cap_B && bt->n_lig <= m->cap_lig && bt->n_kp <= m->cap_kp && bt->n_kk <= m->cap_kk &&
                    bt->max_lig <= m->cap_maxlig && bt->max_kp <= m->cap_maxkp,
                KPD_ERR_CAPACITY, "batch exceeds reserved workspace (call kpd_gvp_reserve)");
    KPD_REQUIRE(bt->kk_rowptr && (bt->n_kk == 0 || (bt->kk_src && bt->kk_dst)), KPD_ERR_INVALID, "kk edges missing");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const kpd_gvp_config &c = m->cfg;
    const int S = m->S;
    const int n[2] = {bt->n_lig, bt->n_kp};

    KPD_TRY(launch_node_graph_index(bt->lig_ptr, bt->B, bt->n_lig, m->bidx[0], st));
    KPD_TRY(launch_node_graph_index(bt->kp_ptr, bt->B, bt->n_kp, m->bidx[1], st));
    KPD_TRY(launch_lig_graph(bt, c.ll_cutoff, c.ll_k, c.kl_cutoff, c.kl_k, &m->lg, m->ll_deg, m->ll_off, m->kl_off, m->kl_pg, st));
    // tile tables for convs over all four edge types and over ll + kl only; z for message_norm == 0
    const float mn = c.message_norm_mode == 2 ? 0.0f : 1.0f;
    KPD_TRY(launch_egnn_meta(m->lg.counts, bt->n_kk, 0xF, 0x3, bt->lig_ptr, bt->kp_ptr, m->lg.ll_per_graph, bt->kk_rowptr, bt->B,
                             m->kl_off, mn, 1, m->meta4, m->z[0], m->z[1], st));
    KPD_TRY(launch_gvp_embed(bt->lig_h, bt->n_lig, c.n_lig_scalars, m->enc_W[0], m->enc_b[0], m->enc_lw[0], m->enc_lb[0],
                             t_dev, m->bidx[0], S, m->St, m->s[0], st));
    KPD_TRY(launch_gvp_embed(bt->kp_h, bt->n_kp, c.n_kp_scalars, m->enc_W[1], m->enc_b[1], m->enc_lw[1], m->enc_lb[1],
                             t_dev, m->bidx[1], S, m->St, m->s[1], st));
    KPD_HIP(hipMemsetAsync(m->v[0], 0, (size_t)bt->n_lig * 48 * 4, st));                      // dynamics_gvp.py:179-184
    if (m->Vt == GV) {
        KPD_HIP(hipMemcpyAsync(m->v[1], bt->kp_v, (size_t)bt->n_kp * 48 * 4, hipMemcpyDeviceToDevice, st));
    } else {               // [n_kp][Vt][3] into the 16-channel rows, padding channels zero
        KPD_HIP(hipMemsetAsync(m->v[1], 0, (size_t)bt->n_kp * 48 * 4, st));
        KPD_HIP(hipMemcpy2DAsync(m->v[1], 48 * 4, bt->kp_v, (size_t)m->Vt * 12, (size_t)m->Vt * 12, bt->n_kp, hipMemcpyDeviceToDevice, st));
    }

    const int e_kl_cap = bt->n_kp * (c.kl_k > 0 ? c.kl_k : std::min(bt->max_lig, 100));
    const int E_cap[4] = {std::max<int>((long)bt->n_lig * std::min(bt->max_lig - 1, c.ll_k > 0 ? c.ll_k : 200), 1), e_kl_cap, e_kl_cap,
                          bt->n_kk};
    const int *esrc[4] = {m->lg.ll_src, m->lg.kl_src, m->lg.lk_src, bt->kk_src};
    const int *edst[4] = {m->lg.ll_dst, m->lg.kl_dst, m->lg.lk_dst, bt->kk_dst};
    const int *rowptr[4] = {m->lg.ll_rowptr, m->lg.kl_rowptr, m->lg.lk_rowptr, bt->kk_rowptr};
    const float *x[2] = {bt->lig_x, bt->kp_x};
    const int n_convs = m->debug_convs >= 0 ? std::min(m->debug_convs, c.n_convs) : c.n_convs;

    for (int ci = 0; ci < n_convs; ++ci) {
        const int net = m->n_et(ci), nnt = net == 4 ? 2 : 1;
        GvpProjArgs pa;
        memset(&pa, 0, sizeof(pa));
        pa.S = S;
        int run = 0, tile_cap = 0;
        for (int et = 0; et < net; ++et) {
            const HostGvp &g = m->msg[ci][et][0];
            pa.tiles_first[et] = run;
            pa.s[et] = m->s[kSrcNtG[et]]; pa.n[et] = n[kSrcNtG[et]]; pa.wp[et] = g.wproj; pa.wp_h[et] = g.wproj_h; pa.b[et] = g.bproj; pa.P[et] = m->Psrc[et];
            run += cdiv(n[kSrcNtG[et]], TM);
            tile_cap += cdiv(E_cap[et], TM);
        }
        pa.n_slots = net;
        pa.tiles_first[net] = run;
        pa.gemm_mode = S == 256 ? m->gemm_mode : 0;
        pa.coop_rows = m->coop_rows;
        KPD_TRY(launch_gvp_proj(pa, st));

        GvpEdgeArgs ea;
        memset(&ea, 0, sizeof(ea));
        ea.meta = net == 4 ? m->meta4 : m->meta2;
        ea.x[0] = x[0]; ea.x[1] = x[1]; ea.v[0] = m->v[0]; ea.v[1] = m->v[1];
        ea.n_gvps = c.n_message_gvps; ea.S = S; ea.rbf_dmax = 15.0f;            // gvp.py:350 default, not overridden
        ea.stamps = m->stamps;
        ea.gemm_mode = (S == 256 && !m->stamps) ? m->gemm_mode : 0;
        for (int et = 0; et < net; ++et) {
            ea.src[et] = esrc[et]; ea.dst[et] = edst[et]; ea.Psrc[et] = m->Psrc[et];
            for (int j = 0; j < c.n_message_gvps; ++j) ea.g[et][j] = m->msg[ci][et][j].dev();
            ea.ms_main[et] = m->ms_main[et]; ea.ms_cont[et] = m->ms_cont[et];
            ea.mv_main[et] = m->mv_main[et]; ea.mv_cont[et] = m->mv_cont[et];
        }
        const bool prof = m->prof_on && m->prof_used + 2 <= m->prof_ev.size();
        if (prof) KPD_HIP(hipEventRecord(m->prof_ev[m->prof_used], st));
        KPD_TRY(launch_gvp_edge(ea, tile_cap, st));
        if (prof) {
            KPD_HIP(hipEventRecord(m->prof_ev[m->prof_used + 1], st));
            m->prof_used += 2;
        }

        GvpNodePair np;
        memset(&np, 0, sizeof(np));
        for (int nt = 0; nt < nnt; ++nt) {
            GvpNodeArgs &na = np.nt[nt];
            na.n = n[nt]; na.s = m->s[nt]; na.v = m->v[nt]; na.s_tmp = m->s_tmp[nt]; na.bidx = m->bidx[nt];
            na.mean = c.message_norm_mode == 1;
            na.z = c.message_norm_mode == 2 ? m->z[nt] : nullptr;
            na.norm_const = c.message_norm_mode == 0 ? c.message_norm : 1.0f;
            int k = 0;
            for (int et = 0; et < net; ++et)
                if (kDstNtG[et] == nt) {
                    na.rowptr[k] = rowptr[et];
                    na.ms_main[k] = m->ms_main[et]; na.ms_cont[k] = m->ms_cont[et];
                    na.mv_main[k] = m->mv_main[et]; na.mv_cont[k] = m->mv_cont[et];
                    ++k;
                }
            na.n_in = k;
            na.ln1_w = m->ln1w[ci][nt]; na.ln1_b = m->ln1b[ci][nt]; na.ln2_w = m->ln2w[ci][nt]; na.ln2_b = m->ln2b[ci][nt];
            na.n_gvps = c.n_update_gvps; na.S = S;
            na.ln_inv_n = 1.0f / (float)m->St; na.ln_pad = (float)(S - m->St);
            na.vn_inv_n = 1.0f / (float)m->Vt; na.vn_pad = (float)(GV - m->Vt);
            for (int j = 0; j < c.n_update_gvps; ++j) na.g[j] = m->upd[ci][nt][j].dev();
        }
        np.tiles0 = cdiv(n[0], TM);
        np.gemm_mode = S == 256 ? m->gemm_mode : 0;
        np.coop_rows = m->coop_rows;
        KPD_TRY(launch_gvp_node(np, st));
    }

    GvpNoiseArgs no;
    memset(&no, 0, sizeof(no));
    no.n = bt->n_lig; no.s = m->s[0]; no.v = m->v[0]; no.n_gvps = c.n_noise_gvps; no.S = S;
    no.gemm_mode = S == 256 ? m->gemm_mode : 0;
    no.coop_rows = m->coop_rows;
    for (int j = 0; j < c.n_noise_gvps; ++j) no.g[j] = m->noise[j].dev();
    no.Wout = m->out_W; no.bout = m->out_b; no.F = c.n_lig_scalars; no.eps_h = eps_h; no.eps_x = eps_x;
    KPD_TRY(launch_gvp_noise(no, st));
    return KPD_OK;
}

extern "C" kpd_status kpd_gvp_debug_state(kpd_gvp *m, const char *what, float *out, int64_t n_floats, void *stream) {
    KPD_REQUIRE(m && what, KPD_ERR_INVALID, "null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const std::string w(what);
    if (w.rfind("convs=", 0) == 0) {
        m->debug_convs = atoi(w.c_str() + 6);
        return KPD_OK;
    }
    if (w.rfind("coop_rows=", 0) == 0) {         // row limit of the cooperative node-side kernels: 0 default, -1 never, N up to N rows
        m->coop_rows = atoi(w.c_str() + 10);
        return KPD_OK;
    }
    if (w.rfind("gemm=", 0) == 0) {              // "gemm=f32" (exact, the default) | "gemm=f16x2" (split f16 products in the message chain)
        const std::string v = w.substr(5);
        KPD_REQUIRE(v == "f32" || v == "f16x2", KPD_ERR_INVALID, "gemm mode must be f32 or f16x2");
        KPD_REQUIRE(v == "f32" || m->S != 256 || !m->committed || m->f16_ok, KPD_ERR_WEIGHTS, "%s", F16_RANGE_ERROR);
        m->gemm_mode = (v == "f16x2" && m->S == 256) ? 1 : 0;
        return KPD_OK;
    }
    if (w == "stamps=1") {
        if (!m->stamps) KPD_HIP(hipMalloc(reinterpret_cast<void **>(&m->stamps), 32 * sizeof(unsigned long long)));
        KPD_HIP(hipMemsetAsync(m->stamps, 0, 32 * sizeof(unsigned long long), st));
        return KPD_OK;
    }
    if (w == "stamps") {
        KPD_REQUIRE(m->stamps && out && n_floats >= 64, KPD_ERR_INVALID, "stamps not enabled or buffer < 64 floats");
        KPD_HIP(hipMemcpyAsync(out, m->stamps, 32 * sizeof(unsigned long long), hipMemcpyDeviceToDevice, st));
        return KPD_OK;
    }
    const float *src = nullptr;
    if (w == "s_lig") src = m->s[0];
    else if (w == "s_kp") src = m->s[1];
    else if (w == "v_lig") src = m->v[0];
    else if (w == "v_kp") src = m->v[1];
    KPD_REQUIRE(src && out, KPD_ERR_INVALID, "unknown debug tap '%s'", what);
    KPD_HIP(hipMemcpyAsync(out, src, (size_t)n_floats * 4, hipMemcpyDeviceToDevice, st));
    return KPD_OK;
}

extern "C" kpd_status kpd_gvp_profile(kpd_gvp *m, int32_t enable) {
    KPD_REQUIRE(m, KPD_ERR_INVALID, "null handle");
    if (enable && m->prof_ev.empty()) {
        m->prof_ev.resize(2 * 8192);
        for (hipEvent_t &e : m->prof_ev) KPD_HIP(hipEventCreate(&e));
    }
    m->prof_on = enable != 0;
    m->prof_used = 0;
    return KPD_OK;
}

extern "C" kpd_status kpd_gvp_profile_read(kpd_gvp *m, double *total_ms, int32_t *launches) {
    KPD_REQUIRE(m && total_ms && launches, KPD_ERR_INVALID, "null argument");
    double tot = 0.0;
    for (size_t i = 0; i + 1 < m->prof_used; i += 2) {
        KPD_HIP(hipEventSynchronize(m->prof_ev[i + 1]));
        float ms = 0.0f;
        KPD_HIP(hipEventElapsedTime(&ms, m->prof_ev[i], m->prof_ev[i + 1]));
        tot += ms;
    }
    *total_ms = tot;
    *launches = (int32_t)(m->prof_used / 2);
    return KPD_OK;
}

extern "C" kpd_status kpd_gvp_last_counts(kpd_gvp *m, int32_t out[8], void *stream) {
    KPD_REQUIRE(m && out, KPD_ERR_INVALID, "null argument");
    for (int i = 0; i < 7; ++i) out[i] = 0;
    out[7] = (m->gemm_mode && !m->stamps) ? 1 : 0;    // GEMM mode the next forward's dominant kernel runs in: 0 exact fp32, 1 f16x2 (a phase-stamped diagnostic run keeps the exact chain)
    if (!m->meta4) return KPD_OK;                      // no forward yet: only the mode is meaningful
    hipStream_t st = static_cast<hipStream_t>(stream);
    int host[25];
    KPD_HIP(hipMemcpyAsync(host, m->meta4, sizeof(host), hipMemcpyDeviceToHost, st));
    KPD_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < 4; ++i) out[i] = host[i];
    out[4] = host[8];                                  // tiles of a conv over all four edge types
    out[5] = host[16 + 8];                             // tiles of the final conv (ll + kl)
    out[6] = host[16] + host[17];                      // edges of the final conv
    return KPD_OK;
}
