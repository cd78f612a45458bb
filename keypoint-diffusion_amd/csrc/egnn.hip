// EGNN denoiser engine: weights, workspace and the per-call launch sequence behind the
// kpd_egnn_* C ABI (include/kpd.h).  Replaces LigRecDynamics.forward
// (models/dynamics.py:342-385) as a whole.
#include <string.h>

#include <map>
#include <set>
#include <string>
#include <vector>

#include "egnn_kernels.h"
#include "mfma_core.h"

using namespace kpd;

namespace {

struct LayerW {
    // per edge type
    float *wp_e[4], *wx_e[4], *b_e[4], *wr_e[4], *watt[4];
    float *wp_c[4], *wx_c[4], *b_c[4], *wr_c[4], *w3[4];
    void *wh_e[4], *wh_c[4];                    // f16x2 mode: the finished wp_e / wp_c blocks as f16 hi / lo planes
    void *chh_p[2][NSLOT];                      //             the projection blocks ch_p
    void *wh_a[2], *wh_b[2], *wh_2[2];          //             and the node-MLP blocks wp_a / wp_b / wp_2
    // per node type, per projection slot
    float *wp_p[2][NSLOT], *wx_p[2][NSLOT], *b_p[2][NSLOT];
    float *ch_p[2][NSLOT], *wcol_p[2][NSLOT];   // k_proj_ws form of the same blocks (chain-chunk fragment order, column 256 apart)
    // node MLP + LayerNorm per node type
    float *wp_a[2], *wx_a[2], *wp_b[2], *wx_b[2], *b0[2], *wp_2[2], *wx_2[2], *b2[2], *ln_w[2], *ln_b[2];
};

const int kSrcNt[4] = {NT_LIG, NT_KP, NT_LIG, NT_KP};
const int kDstNt[4] = {NT_LIG, NT_LIG, NT_KP, NT_KP};
const int kSrcSlot[4] = {0, 0, 6, 4};
const int kDstSlot[4] = {2, 4, 2, 6};
const char *kEtName[4] = {"ll", "kl", "lk", "kk"};
const char *kNtName[2] = {"lig", "kp"};

std::vector<std::string> split(const std::string &s, char c) {
    std::vector<std::string> out;
    size_t p = 0;
    while (true) {
        size_t q = s.find(c, p);
        out.push_back(s.substr(p, q == std::string::npos ? q : q - p));
        if (q == std::string::npos) break;
        p = q + 1;
    }
    return out;
}

int et_index(const std::string &s) {
    for (int i = 0; i < 4; ++i)
        if (s == kEtName[i]) return i;
    return -1;
}
int nt_index(const std::string &s) {
    for (int i = 0; i < 2; ++i)
        if (s == kNtName[i]) return i;
    return -1;
}

}  // namespace

struct kpd_egnn {
    kpd_egnn_config cfg;
    int n_et;                 // 2 or 4 active edge types
    int n_upd;                // 1 or 2 updated node types
    bool rec_identity;
    Arena warena, ws;
    std::vector<LayerW> L;
    // encoders / decoder
    float *le_W0, *le_b0, *le_W1t, *le_b1;
    float *re_W0, *re_b0, *re_W1t, *re_b1;
    float *de_W0, *de_b0, *de_W1, *de_b1;
    std::set<std::string> expected, loaded;
    bool committed = false;
    int debug_layers = -1;
    bool f16_ok = true;                        // the committed weights fit the f16 planes (pack.hip range guard)
    float *widen_buf = nullptr;                // hidden_nf < 256: staging of one reference tensor in the 256-wide layout
    int *widen_map = nullptr;
    size_t widen_floats = 0, widen_ints = 0;
    int gemm_mode = 0;                         // 0 exact fp32 MFMA; 1 f16x2 split products in the EGNN GEMMs (KPD_GEMM=f16x2, "gemm=f16x2")
    int h_parts = 7;                           // diagnostics: which kernels take the f16x2 form (1 edge, 2 projections, 4 node update)
    int prune_last = 1;                        // final layer: only what feeds (h_lig, x_lig) is computed ("prune=0" restores all)
    // optional HIP-event timing of the dominant kernel (k_egnn_edge), for bench.py's roofline
    unsigned long long *stamps = nullptr;      // device [16], diagnostics (kpd_egnn_debug_state "stamps=1")
    float *edge_dbg = nullptr;                 // per-row taps of the f16x2 edge kernel's coordinate branch ("edge_dbg=1", -DKPD_EDGE_DBG builds)
    size_t edge_dbg_floats = 0;
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;
    size_t prof_used = 0;

    // workspace (valid after reserve)
    int cap_B = 0, cap_lig = 0, cap_kp = 0, cap_kk = 0, cap_ll = 0, cap_kl = 0, cap_maxlig = 0, cap_maxkp = 0;
    int tile_cap = 0, tiles_et_cap[4] = {0, 0, 0, 0};
    float *h[2], *x[2], *P[2];
    float *hn_main[4], *hn_cont[4], *xn_main[4], *xn_cont[4];
    int *bidx[2];
    float *z[2];
    int *meta, *ll_deg, *ll_off, *kl_off, *kl_pg;
    kpd_lig_graph lg;
};

static kpd_status build_weight_arena(kpd_egnn *m) {
    const kpd_egnn_config &c = m->cfg;
    const size_t wpB = (size_t)WP_FLOATS * 4 + 256, vB = (size_t)HS * 4 + 256;
    const int n_gemm_layer = m->n_et * 2 + 16 + m->n_upd * 3;
    size_t bytes = (size_t)c.n_layers * (n_gemm_layer * (wpB + vB) + (m->n_et * 6 + 16 + m->n_upd * 4) * vB);
    bytes += (size_t)(64 * c.atom_nf + 64 + 64 * 256 + 256 + 2 * c.rec_nf * c.rec_nf + 2 * c.rec_nf +
                      2 * c.rec_nf * 256 + 256 + 2 * c.atom_nf * 256 + 2 * c.atom_nf + 2 * c.atom_nf * c.atom_nf +
                      c.atom_nf) * 4 + 64 * 256;
    bytes += (size_t)c.n_layers * m->n_et * 4 * (16 * 4096 + HS + 64) * 4;
    bytes += (size_t)c.n_layers * m->n_et * 2 * ((size_t)WH_HALVES * 2 + 256);
    bytes += (size_t)c.n_layers * m->n_et * 4 * ((size_t)CHH_HALVES * 2 + 256);
    bytes += (size_t)c.n_layers * m->n_upd * 3 * ((size_t)WH_HALVES * 2 + 256);
    bytes += 1 << 20;
    kpd_status st = m->warena.reserve(bytes);
    if (st != KPD_OK) return st;
    m->warena.poison_at = 2;          // packed weights: poisoned only at KPD_POISON >= 2 (engine.h)
    Arena &A = m->warena;
    m->L.assign(c.n_layers, LayerW());
    auto wp = [&]() { return A.take<float>(WP_FLOATS); };
    auto vec = [&]() { return A.take<float>(HS); };
    for (int i = 0; i < c.n_layers; ++i) {
        LayerW &w = m->L[i];
        memset(&w, 0, sizeof(w));
        const std::string pre = "egnn.conv_layers." + std::to_string(i) + ".";
        for (int et = 0; et < m->n_et; ++et) {
            w.wp_e[et] = wp(); w.wx_e[et] = vec(); w.b_e[et] = vec(); w.wr_e[et] = vec(); w.watt[et] = vec();
            w.wp_c[et] = wp(); w.wx_c[et] = vec(); w.b_c[et] = vec(); w.wr_c[et] = vec(); w.w3[et] = vec();
            w.wh_e[et] = A.take<unsigned short>(WH_HALVES); w.wh_c[et] = A.take<unsigned short>(WH_HALVES);
            for (int var = 0; var < 2; ++var) {
                const int ss = kSrcSlot[et] + var, ds = kDstSlot[et] + var;
                w.wp_p[kSrcNt[et]][ss] = wp(); w.wx_p[kSrcNt[et]][ss] = vec();
                w.wp_p[kDstNt[et]][ds] = wp(); w.wx_p[kDstNt[et]][ds] = vec(); w.b_p[kDstNt[et]][ds] = vec();
                w.ch_p[kSrcNt[et]][ss] = A.take<float>(16 * 4096); w.wcol_p[kSrcNt[et]][ss] = vec();
                w.ch_p[kDstNt[et]][ds] = A.take<float>(16 * 4096); w.wcol_p[kDstNt[et]][ds] = vec();
                w.chh_p[kSrcNt[et]][ss] = A.take<unsigned short>(CHH_HALVES);
                w.chh_p[kDstNt[et]][ds] = A.take<unsigned short>(CHH_HALVES);
            }
            const std::string e = kEtName[et];
            for (const char *blk : {"edge_mlp.", "coord_mlp."})
                for (const char *s : {".0.weight", ".0.bias", ".2.weight", ".2.bias"}) m->expected.insert(pre + blk + e + s);
            m->expected.insert(pre + "coord_mlp." + e + ".4.weight");
            m->expected.insert(pre + "soft_attention." + e + ".0.weight");
            m->expected.insert(pre + "soft_attention." + e + ".0.bias");
        }
        for (int nt = 0; nt < m->n_upd; ++nt) {
            w.wp_a[nt] = wp(); w.wx_a[nt] = vec(); w.wp_b[nt] = wp(); w.wx_b[nt] = vec(); w.b0[nt] = vec();
            w.wp_2[nt] = wp(); w.wx_2[nt] = vec(); w.b2[nt] = vec(); w.ln_w[nt] = vec(); w.ln_b[nt] = vec();
            w.wh_a[nt] = A.take<unsigned short>(WH_HALVES); w.wh_b[nt] = A.take<unsigned short>(WH_HALVES);
            w.wh_2[nt] = A.take<unsigned short>(WH_HALVES);
            const std::string n = kNtName[nt];
            for (const char *s : {".0.weight", ".0.bias", ".2.weight", ".2.bias"}) m->expected.insert(pre + "node_mlp." + n + s);
            if (c.norm) {
                m->expected.insert(pre + "layer_norm." + n + ".weight");
                m->expected.insert(pre + "layer_norm." + n + ".bias");
            }
        }
    }
    m->le_W0 = A.take<float>(64 * c.atom_nf); m->le_b0 = A.take<float>(64);
    m->le_W1t = A.take<float>(64 * 256); m->le_b1 = A.take<float>(256);
    for (const char *s : {"lig_encoder.0.weight", "lig_encoder.0.bias", "lig_encoder.2.weight", "lig_encoder.2.bias",
                          "lig_decoder.0.weight", "lig_decoder.0.bias", "lig_decoder.2.weight", "lig_decoder.2.bias"})
        m->expected.insert(s);
    if (!m->rec_identity) {
        m->re_W0 = A.take<float>(2 * c.rec_nf * c.rec_nf); m->re_b0 = A.take<float>(2 * c.rec_nf);
        m->re_W1t = A.take<float>(2 * c.rec_nf * 256); m->re_b1 = A.take<float>(256);
        for (const char *s : {"rec_encoder.0.weight", "rec_encoder.0.bias", "rec_encoder.2.weight", "rec_encoder.2.bias"})
            m->expected.insert(s);
    } else {
        m->re_W0 = m->re_b0 = m->re_W1t = m->re_b1 = nullptr;
    }
    m->de_W0 = A.take<float>(2 * c.atom_nf * 256); m->de_b0 = A.take<float>(2 * c.atom_nf);
    m->de_W1 = A.take<float>(2 * c.atom_nf * c.atom_nf); m->de_b1 = A.take<float>(c.atom_nf);
    KPD_REQUIRE(m->de_b1 != nullptr, KPD_ERR_HIP, "weight arena too small (internal sizing error)");
    return KPD_OK;
}

extern "C" kpd_status kpd_egnn_create(const kpd_egnn_config *cfg, kpd_egnn **out) {
    KPD_REQUIRE(cfg && out, KPD_ERR_INVALID, "null argument");
    // hidden_nf < 256 runs on the same kernels: features live in columns 0 .. hidden_nf - 1 of the 256-wide layout, the timestep
    // stays in column 256, the columns between hold zeros (zero weight rows / columns, kpd_egnn_load_weight), LayerNorm takes its
    // width at run time
    KPD_REQUIRE(cfg->hidden_nf >= 1 && cfg->hidden_nf <= HID, KPD_ERR_INVALID, "hidden_nf=%d: the HIP path covers 1 .. 256", cfg->hidden_nf);
    KPD_REQUIRE(cfg->hidden_nf == HID || cfg->rec_nf != cfg->hidden_nf, KPD_ERR_INVALID,
                "rec_nf == hidden_nf = %d (identity keypoint encoder, dynamics.py:326-334) is implemented for hidden_nf = 256 only", cfg->hidden_nf);
    KPD_REQUIRE(cfg->ll_k >= 0 && cfg->ll_k <= KL_KMAX, KPD_ERR_INVALID, "ll_k=%d outside 0..%d (0 = radius graph)", cfg->ll_k, KL_KMAX);
    KPD_REQUIRE(cfg->kl_k >= 0 && cfg->kl_k <= KL_KMAX, KPD_ERR_INVALID, "kl_k=%d outside 0..%d (0 = radius graph)", cfg->kl_k, KL_KMAX);
    KPD_REQUIRE(cfg->kl_k > 0 || cfg->kl_cutoff > 0.0f, KPD_ERR_INVALID, "kl_k = 0 needs graph_cutoffs['kl'] > 0");
    KPD_REQUIRE(cfg->ll_k > 0 || cfg->ll_cutoff > 0.0f, KPD_ERR_INVALID, "ll_k = 0 needs graph_cutoffs['ll'] > 0");
    KPD_REQUIRE(cfg->n_layers >= 1 && cfg->n_layers <= 64, KPD_ERR_INVALID, "n_layers=%d", cfg->n_layers);
    KPD_REQUIRE(cfg->atom_nf >= 1 && cfg->atom_nf <= 32, KPD_ERR_INVALID, "atom_nf=%d outside 1..32", cfg->atom_nf);
    KPD_REQUIRE(cfg->rec_nf >= 1 && (cfg->rec_nf <= 128 || cfg->rec_nf == 256), KPD_ERR_INVALID, "rec_nf=%d", cfg->rec_nf);
    KPD_REQUIRE(cfg->message_norm >= 0.0f, KPD_ERR_INVALID, "message_norm=%f", cfg->message_norm);
    kpd_status st = egnn_kernels_init();
    if (st != KPD_OK) return st;
    kpd_egnn *m = new kpd_egnn();
    m->cfg = *cfg;
    if (const char *e = getenv("KPD_GEMM")) m->gemm_mode = !strcmp(e, "f16x2") ? 1 : 0;
    m->h_parts = tool_env_int("KPD_H_PARTS", m->h_parts);      // (TOOLS build only)
    m->n_et = cfg->update_kp_feat ? 4 : 2;
    m->n_upd = cfg->update_kp_feat ? 2 : 1;
    m->rec_identity = cfg->rec_nf == cfg->hidden_nf;   // dynamics.py:326-334
    st = build_weight_arena(m);
    if (st == KPD_OK && cfg->hidden_nf != HID) {
        m->widen_floats = (size_t)HW * (2 * HW + 1);
        m->widen_ints = (size_t)HW + 2 * HW + 1;
        if (hipMalloc(reinterpret_cast<void **>(&m->widen_buf), m->widen_floats * 4) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&m->widen_map), m->widen_ints * 4) != hipSuccess) {
            set_error("allocation of the weight staging buffer failed");
            st = KPD_ERR_HIP;
        }
    }
    if (st != KPD_OK) {
        kpd_egnn_destroy(m);
        return st;
    }
    *out = m;
    return KPD_OK;
}

extern "C" void kpd_egnn_destroy(kpd_egnn *m) {
    if (!m) return;
    for (hipEvent_t e : m->prof_ev) (void)hipEventDestroy(e);
    if (m->edge_dbg) (void)hipFree(m->edge_dbg);
    if (m->stamps) (void)hipFree(m->stamps);
    if (m->widen_buf) (void)hipFree(m->widen_buf);
    if (m->widen_map) (void)hipFree(m->widen_map);
    m->warena.release();
    m->ws.release();
    delete m;
}

static kpd_status expect_shape(const char *name, const int64_t *shape, int ndim, std::initializer_list<int64_t> want) {
    bool ok = ndim == (int)want.size();
    int i = 0;
    for (int64_t w : want) {
        if (ok && shape[i] != w) ok = false;
        ++i;
    }
    if (!ok) {
        std::string got, exp;
        for (int j = 0; j < ndim; ++j) got += std::to_string(shape[j]) + (j + 1 < ndim ? "," : "");
        for (int64_t w : want) exp += std::to_string(w) + ",";
        set_error("weight %s has shape [%s], expected [%s]", name, got.c_str(), exp.c_str());
        return KPD_ERR_WEIGHTS;
    }
    return KPD_OK;
}

// ---- hidden_nf < 256: reference tensors are staged into the 256-wide layout before the packing code below sees them ----------
// An axis of a reference tensor is a sequence of blocks: 'F' = a feature block of width hidden_nf + 1 ([hidden | timestep]) that
// becomes 257 wide (hidden -> columns 0 .. hidden_nf - 1, timestep -> column 256, zeros between), 'P' = a block of hidden_nf rows
// / columns padded with zeros to 256, 'R' = n entries kept as they are.
struct AxisBlock { char type; int n; };

static std::vector<int> axis_map(const std::vector<AxisBlock> &blocks, int H, int *ref_len) {
    std::vector<int> map;        // padded index -> reference index or -1
    int ref = 0;
    for (const AxisBlock &b : blocks) {
        if (b.type == 'F') {
            for (int j = 0; j < HW; ++j) map.push_back(j < H ? ref + j : (j == HID ? ref + H : -1));
            ref += H + 1;
        } else if (b.type == 'P') {
            for (int j = 0; j < HID; ++j) map.push_back(j < H ? ref + j : -1);
            ref += H;
        } else {
            for (int j = 0; j < b.n; ++j) map.push_back(ref + j);
            ref += b.n;
        }
    }
    *ref_len = ref;
    return map;
}

__global__ void k_widen(const float *__restrict__ src, int ld, const int *__restrict__ rmap, int rows, const int *__restrict__ cmap, int cols,
                        float *__restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int r = rmap[i / cols], c = cmap[i % cols];
    dst[i] = (r >= 0 && c >= 0) ? src[(size_t)r * ld + c] : 0.0f;
}

// stages w (reference shape) into m->widen_buf in the padded layout; *shape_out = the padded shape
static kpd_status widen_weight(kpd_egnn *m, const char *name, const float *w, const int64_t *shape, int ndim, const std::vector<AxisBlock> &rows,
                               const std::vector<AxisBlock> &cols, int64_t *shape_out, hipStream_t st) {
    const int H = m->cfg.hidden_nf;
    int ref_r = 0, ref_c = 0;
    std::vector<int> rmap = axis_map(rows, H, &ref_r), cmap = cols.empty() ? std::vector<int>{0} : axis_map(cols, H, &ref_c);
    if (cols.empty()) ref_c = 1;
    const bool ok = cols.empty() ? (ndim == 1 && shape[0] == ref_r) : (ndim == 2 && shape[0] == ref_r && shape[1] == ref_c);
    KPD_REQUIRE(ok, KPD_ERR_WEIGHTS, "weight %s has the wrong shape for hidden_nf=%d (expected [%d%s%s])", name, H, ref_r, cols.empty() ? "" : ", ",
                cols.empty() ? "" : std::to_string(ref_c).c_str());
    const int R = (int)rmap.size(), Cc = (int)cmap.size();
    KPD_REQUIRE((size_t)R * Cc <= m->widen_floats && (size_t)(R + Cc) <= m->widen_ints, KPD_ERR_HIP, "widen scratch too small (internal sizing error)");
    KPD_HIP(hipMemcpyAsync(m->widen_map, rmap.data(), (size_t)R * 4, hipMemcpyHostToDevice, st));
    KPD_HIP(hipMemcpyAsync(m->widen_map + R, cmap.data(), (size_t)Cc * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_widen, dim3(cdiv(R * Cc, 256)), dim3(256), 0, st, w, ref_c, m->widen_map, R, m->widen_map + R, Cc, m->widen_buf);
    KPD_LAUNCH_CHECK();
    KPD_HIP(hipStreamSynchronize(st));          // rmap / cmap are host vectors that die with this call
    shape_out[0] = R;
    shape_out[1] = Cc;
    return KPD_OK;
}

extern "C" kpd_status kpd_egnn_load_weight(kpd_egnn *m, const char *name, const float *w, const int64_t *shape,
                                           int32_t ndim, void *stream) {
    KPD_REQUIRE(m && name && w && shape, KPD_ERR_INVALID, "null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const kpd_egnn_config &c = m->cfg;
    const std::string nm(name);
    if (!m->expected.count(nm)) {
        set_error("unknown or unused weight name '%s' for this configuration", name);
        return KPD_ERR_WEIGHTS;
    }
    const std::vector<std::string> tk = split(nm, '.');
    const bool is_w = tk.back() == "weight";
    int64_t wide_shape[2];
    if (c.hidden_nf != HID) {
        // which axes of this tensor carry the hidden width (reference shapes: models/dynamics.py:37-87, 313-334)
        std::vector<AxisBlock> rows, cols;
        bool touch = true;
        const AxisBlock Fb{'F', 0}, Pb{'P', 0};
        if (tk[0] == "lig_encoder" || tk[0] == "rec_encoder") {
            if (tk[1] == "0") touch = false;
            else { rows = {Pb}; if (is_w) cols = {{'R', tk[0] == "lig_encoder" ? 64 : 2 * c.rec_nf}}; }
        } else if (tk[0] == "lig_decoder") {
            if (tk[1] == "0" && is_w) { rows = {{'R', 2 * c.atom_nf}}; cols = {Pb}; }
            else touch = false;
        } else {
            const std::string &blk = tk[3];
            if (blk == "layer_norm") rows = {Fb};
            else if (blk == "node_mlp") { rows = {Fb}; if (is_w) cols = tk[5] == "0" ? std::vector<AxisBlock>{Fb, Fb} : std::vector<AxisBlock>{Fb}; }
            else if (blk == "soft_attention") { if (is_w) { rows = {{'R', 1}}; cols = {Fb}; } else touch = false; }
            else if (tk[5] == "0") { rows = {Fb}; if (is_w) cols = {Fb, Fb, {'R', 1}}; }
            else if (tk[5] == "2") { rows = {Fb}; if (is_w) cols = {Fb}; }
            else { rows = {{'R', 1}}; cols = {Fb}; }                     // coord_mlp.<et>.4.weight
        }
        if (touch) {
            KPD_TRY(widen_weight(m, name, w, shape, ndim, rows, cols, wide_shape, st));
            w = m->widen_buf;
            shape = wide_shape;
        }
    }
    if (tk[0] == "lig_encoder") {
        if (tk[1] == "0") {
            if (is_w) { KPD_TRY(expect_shape(name, shape, ndim, {64, c.atom_nf})); KPD_TRY(copy_pad(w, 64 * c.atom_nf, m->le_W0, 64 * c.atom_nf, st)); }
            else { KPD_TRY(expect_shape(name, shape, ndim, {64})); KPD_TRY(copy_pad(w, 64, m->le_b0, 64, st)); }
        } else {
            if (is_w) { KPD_TRY(expect_shape(name, shape, ndim, {256, 64})); KPD_TRY(transpose2d(w, 256, 64, m->le_W1t, st)); }
            else { KPD_TRY(expect_shape(name, shape, ndim, {256})); KPD_TRY(copy_pad(w, 256, m->le_b1, 256, st)); }
        }
    } else if (tk[0] == "rec_encoder") {
        const int r = c.rec_nf;
        if (tk[1] == "0") {
            if (is_w) { KPD_TRY(expect_shape(name, shape, ndim, {2 * r, r})); KPD_TRY(copy_pad(w, 2 * r * r, m->re_W0, 2 * r * r, st)); }
            else { KPD_TRY(expect_shape(name, shape, ndim, {2 * r})); KPD_TRY(copy_pad(w, 2 * r, m->re_b0, 2 * r, st)); }
        } else {
            if (is_w) { KPD_TRY(expect_shape(name, shape, ndim, {256, 2 * r})); KPD_TRY(transpose2d(w, 256, 2 * r, m->re_W1t, st)); }
            else { KPD_TRY(expect_shape(name, shape, ndim, {256})); KPD_TRY(copy_pad(w, 256, m->re_b1, 256, st)); }
        }
    } else if (tk[0] == "lig_decoder") {
        const int a = c.atom_nf;
        if (tk[1] == "0") {
            if (is_w) { KPD_TRY(expect_shape(name, shape, ndim, {2 * a, 256})); KPD_TRY(copy_pad(w, 2 * a * 256, m->de_W0, 2 * a * 256, st)); }
            else { KPD_TRY(expect_shape(name, shape, ndim, {2 * a})); KPD_TRY(copy_pad(w, 2 * a, m->de_b0, 2 * a, st)); }
        } else {
            if (is_w) { KPD_TRY(expect_shape(name, shape, ndim, {a, 2 * a})); KPD_TRY(copy_pad(w, 2 * a * a, m->de_W1, 2 * a * a, st)); }
            else { KPD_TRY(expect_shape(name, shape, ndim, {a})); KPD_TRY(copy_pad(w, a, m->de_b1, a, st)); }
        }
    } else {
        // egnn.conv_layers.<i>.<block>.<et|nt>.<idx>.<param>  |  egnn.conv_layers.<i>.layer_norm.<nt>.<param>
        const int i = atoi(tk[2].c_str());
        LayerW &L = m->L[i];
        const std::string &blk = tk[3];
        if (blk == "layer_norm") {
            const int nt = nt_index(tk[4]);
            KPD_TRY(expect_shape(name, shape, ndim, {HW}));
            KPD_TRY(copy_pad(w, HW, is_w ? L.ln_w[nt] : L.ln_b[nt], HS, st));
        } else if (blk == "node_mlp") {
            const int nt = nt_index(tk[4]);
            if (tk[5] == "0") {
                if (is_w) {
                    KPD_TRY(expect_shape(name, shape, ndim, {HW, 2 * HW}));
                    KPD_TRY(pack_gemm_weight(w, HW, 2 * HW, 0, HW, L.wp_a[nt], L.wx_a[nt], st));
                    KPD_TRY(pack_gemm_weight(w, HW, 2 * HW, HW, HW, L.wp_b[nt], L.wx_b[nt], st));
                } else {
                    KPD_TRY(expect_shape(name, shape, ndim, {HW}));
                    KPD_TRY(copy_pad(w, HW, L.b0[nt], HS, st));
                }
            } else {
                if (is_w) {
                    KPD_TRY(expect_shape(name, shape, ndim, {HW, HW}));
                    KPD_TRY(pack_gemm_weight(w, HW, HW, 0, HW, L.wp_2[nt], L.wx_2[nt], st));
                } else {
                    KPD_TRY(expect_shape(name, shape, ndim, {HW}));
                    KPD_TRY(copy_pad(w, HW, L.b2[nt], HS, st));
                }
            }
        } else if (blk == "soft_attention") {
            const int et = et_index(tk[4]);
            if (is_w) {
                KPD_TRY(expect_shape(name, shape, ndim, {1, HW}));
                KPD_TRY(copy_pad(w, HW, L.watt[et], ATT_BIAS_AT, st));     // keeps [260] (bias) intact
                KPD_TRY(scale_inplace(L.watt[et], HW, 1.0f / SILU_C, st)); // consumes c * m (pre-scaled SiLU)
            } else {
                KPD_TRY(expect_shape(name, shape, ndim, {1}));
                KPD_TRY(copy_pad(w, 1, L.watt[et] + ATT_BIAS_AT, 1, st));
            }
        } else {   // edge_mlp / coord_mlp
            const int var = blk == "coord_mlp" ? 1 : 0;
            const int et = et_index(tk[4]);
            const int snt = kSrcNt[et], dnt = kDstNt[et], ss = kSrcSlot[et] + var, ds = kDstSlot[et] + var;
            if (tk[5] == "0") {
                if (is_w) {
                    KPD_TRY(expect_shape(name, shape, ndim, {HW, 2 * HW + 1}));
                    const int ld = 2 * HW + 1;
                    KPD_TRY(pack_gemm_weight(w, HW, ld, 0, HW, L.wp_p[snt][ss], L.wx_p[snt][ss], st));
                    KPD_TRY(pack_gemm_weight(w, HW, ld, HW, HW, L.wp_p[dnt][ds], L.wx_p[dnt][ds], st));
                    KPD_TRY(copy_col_pad(w, HW, ld, 2 * HW, var ? L.wr_c[et] : L.wr_e[et], HS, st));
                    // the first Linear produces c * (pre-activation): see silu_pre() in mfma_core.h
                    KPD_TRY(scale_inplace(L.wp_p[snt][ss], WP_FLOATS, SILU_C, st));
                    KPD_TRY(scale_inplace(L.wx_p[snt][ss], KP, SILU_C, st));
                    KPD_TRY(scale_inplace(L.wp_p[dnt][ds], WP_FLOATS, SILU_C, st));
                    KPD_TRY(scale_inplace(L.wx_p[dnt][ds], KP, SILU_C, st));
                    KPD_TRY(scale_inplace(var ? L.wr_c[et] : L.wr_e[et], HS, SILU_C, st));
                    for (int side = 0; side < 2; ++side) {      // the same two blocks in the chunk order k_proj_ws keeps resident
                        float *ch = side ? L.ch_p[dnt][ds] : L.ch_p[snt][ss], *wc = side ? L.wcol_p[dnt][ds] : L.wcol_p[snt][ss];
                        const float *blk0 = w + side * HW;
                        for (int kc = 0; kc < 16; ++kc) KPD_TRY(pack_chain_frag(blk0, ld, 1, 256, 16 * kc, 16, 16, ch + (size_t)kc * 4096, st));
                        KPD_TRY(copy_col_pad(blk0, 256, ld, 256, wc, HS, st));
                        KPD_TRY(scale_inplace(ch, 16 * 4096, SILU_C, st));
                        KPD_TRY(scale_inplace(wc, HS, SILU_C, st));
                    }
                } else {
                    KPD_TRY(expect_shape(name, shape, ndim, {HW}));
                    KPD_TRY(copy_pad(w, HW, L.b_p[dnt][ds], HS, st));
                    KPD_TRY(scale_inplace(L.b_p[dnt][ds], HS, SILU_C, st));
                }
            } else if (tk[5] == "2") {
                if (is_w) {
                    KPD_TRY(expect_shape(name, shape, ndim, {HW, HW}));
                    KPD_TRY(pack_gemm_weight(w, HW, HW, 0, HW, var ? L.wp_c[et] : L.wp_e[et], var ? L.wx_c[et] : L.wx_e[et], st));
                } else {
                    KPD_TRY(expect_shape(name, shape, ndim, {HW}));
                    KPD_TRY(copy_pad(w, HW, var ? L.b_c[et] : L.b_e[et], HS, st));
                }
            } else {   // coord_mlp.<et>.4.weight
                KPD_TRY(expect_shape(name, shape, ndim, {1, HW}));
                KPD_TRY(copy_pad(w, HW, L.w3[et], HS, st));
                KPD_TRY(scale_inplace(L.w3[et], HS, 1.0f / SILU_C, st));
            }
        }
    }
    m->loaded.insert(nm);
    m->committed = false;
    return KPD_OK;
}

extern "C" kpd_status kpd_egnn_commit(kpd_egnn *m) {
    KPD_REQUIRE(m, KPD_ERR_INVALID, "null handle");
    for (const std::string &n : m->expected)
        if (!m->loaded.count(n)) {
            set_error("weight '%s' was never loaded (%zu of %zu loaded)", n.c_str(), m->loaded.size(), m->expected.size());
            return KPD_ERR_WEIGHTS;
        }
    // the bias of edge_mlp.2 / coord_mlp.2 rides in the GEMM as weight row BIAS_K against a constant-1 column of
    // the A tile (scaled by c for the pre-scaled SiLU); patched here because weight and bias arrive separately
    F16PackScope f16_scope;
    for (LayerW &L : m->L)
        for (int et = 0; et < m->n_et; ++et) {
            KPD_TRY(patch_bias_row(L.wp_e[et], L.wx_e[et], L.b_e[et], SILU_C, BIAS_K, nullptr));
            KPD_TRY(patch_bias_row(L.wp_c[et], L.wx_c[et], L.b_c[et], SILU_C, BIAS_K, nullptr));
            KPD_TRY(pack_f16_split(L.wp_e[et], L.wh_e[et], nullptr));      // the same finished blocks for the f16x2 mode
            KPD_TRY(pack_f16_split(L.wp_c[et], L.wh_c[et], nullptr));
            KPD_TRY(f16_range_check_array(L.wx_e[et], KP, nullptr));        // W2[256, :]: split inside k_egnn_edge_h
            KPD_TRY(f16_range_check_array(L.wx_c[et], KP, nullptr));
        }
    for (LayerW &L : m->L)
        for (int nt = 0; nt < 2; ++nt)
            for (int s = 0; s < NSLOT; ++s)
                if (L.ch_p[nt][s]) KPD_TRY(pack_proj_f16_split(L.ch_p[nt][s], L.chh_p[nt][s], nullptr));
    for (LayerW &L : m->L)
        for (int nt = 0; nt < m->n_upd; ++nt) {
            KPD_TRY(pack_f16_split(L.wp_a[nt], L.wh_a[nt], nullptr));
            KPD_TRY(pack_f16_split(L.wp_b[nt], L.wh_b[nt], nullptr));
            KPD_TRY(pack_f16_split(L.wp_2[nt], L.wh_2[nt], nullptr));
        }
    m->f16_ok = !f16_scope.overflowed();           // (synchronises the device)
    KPD_REQUIRE(m->f16_ok || m->gemm_mode == 0, KPD_ERR_WEIGHTS, "%s", F16_RANGE_ERROR);       // KPD_GEMM=f16x2 asked for it
    KPD_HIP(hipDeviceSynchronize());
    m->committed = true;
    return KPD_OK;
}

extern "C" kpd_status kpd_egnn_reserve(kpd_egnn *m, int32_t max_B, int32_t max_n_lig, int32_t max_n_kp, int32_t max_n_kk,
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
    // the edge kernels address a node's row of P by a 32-bit byte offset (node * NSLOT * HS * 4)
    KPD_REQUIRE(((long)std::max(max_n_lig, max_n_kp) + TM) * NSLOT * HS * 4 < (1l << 32), KPD_ERR_CAPACITY,
                "batch of %d / %d nodes exceeds the 4 GB addressable per node type by the edge kernel's 32-bit row offsets (split the batch)",
                max_n_lig, max_n_kp);
    const int cap_ll = std::max<long>(cap_ll_l, 1), cap_kl = std::max<long>(cap_kl_l, 1);
    const int E_cap[4] = {cap_ll, cap_kl, cap_kl, std::max(max_n_kk, 1)};
    int tiles[4], tile_cap = 0;
    for (int et = 0; et < 4; ++et) {
        tiles[et] = cdiv(E_cap[et], 32) + 1;          // sized for the finer of the two edge-tile sizes
        tile_cap += tiles[et];
    }
    const int n[2] = {max_n_lig, max_n_kp};
    size_t bytes = 1 << 20;
    auto add = [&](size_t cnt, size_t sz) { bytes += ((cnt * sz + 255) & ~size_t(255)); };
    for (int nt = 0; nt < 2; ++nt) {
        add((size_t)(n[nt] + TM) * HS, 4); add((size_t)n[nt] * 3, 4); add((size_t)(n[nt] + TM) * NSLOT * HS, 4);
        add(n[nt], 4); add(max_B, 4);
    }
    for (int et = 0; et < 4; ++et) {
        add((size_t)n[kDstNt[et]] * HS, 4); add((size_t)tiles[et] * HS, 4);
        add((size_t)n[kDstNt[et]] * 4, 4); add((size_t)tiles[et] * 4, 4);
    }
    add(32, 4); add(max_n_lig, 4); add(max_B + 1, 4); add(max_B + 1, 4); add(max_B + 2, 4);
    add(cap_ll, 4); add(cap_ll, 4); add(max_n_lig + 1, 4);
    for (int i = 0; i < 4; ++i) add(cap_kl, 4);
    add(max_n_lig + 1, 4); add(max_n_kp + 1, 4); add(max_B, 4); add(8, 4);
    KPD_TRY(m->ws.reserve(bytes));
    Arena &W = m->ws;
    for (int nt = 0; nt < 2; ++nt) {
        m->h[nt] = W.take<float>((size_t)(n[nt] + TM) * HS);          // + one tile: kernels touch whole tiles
        m->x[nt] = W.take<float>((size_t)n[nt] * 3);
        m->P[nt] = W.take_rows((size_t)(n[nt] + TM) * NSLOT, HS, HW);
        m->bidx[nt] = W.take<int>(n[nt]);
        m->z[nt] = W.take<float>(max_B);
    }
    for (int et = 0; et < 4; ++et) {
        m->hn_main[et] = W.take_rows(n[kDstNt[et]], HS, HW);
        m->hn_cont[et] = W.take_rows(tiles[et], HS, HW);
        m->xn_main[et] = W.take<float>((size_t)n[kDstNt[et]] * 4);
        m->xn_cont[et] = W.take<float>((size_t)tiles[et] * 4);
        m->tiles_et_cap[et] = tiles[et];
    }
    m->meta = W.take<int>(32);                 // [0..8] all active edge types, [16..24] the final layer's subset
    m->ll_deg = W.take<int>(max_n_lig);
    m->ll_off = W.take<int>(max_B + 1);
    m->kl_off = W.take<int>(max_B + 1);
    m->kl_pg = W.take<int>(max_B + 2);
    kpd_lig_graph &g = m->lg;
    g.cap_ll = cap_ll; g.cap_kl = cap_kl;
    g.ll_src = W.take<int>(cap_ll); g.ll_dst = W.take<int>(cap_ll); g.ll_rowptr = W.take<int>(max_n_lig + 1);
    g.kl_src = W.take<int>(cap_kl); g.kl_dst = W.take<int>(cap_kl); g.kl_rowptr = W.take<int>(max_n_lig + 1);
    g.lk_src = W.take<int>(cap_kl); g.lk_dst = W.take<int>(cap_kl); g.lk_rowptr = W.take<int>(max_n_kp + 1);
    g.ll_per_graph = W.take<int>(max_B);
    g.counts = W.take<int>(8);
    KPD_REQUIRE(g.counts != nullptr, KPD_ERR_HIP, "workspace arena too small (internal sizing error)");
    m->cap_B = max_B; m->cap_lig = max_n_lig; m->cap_kp = max_n_kp; m->cap_kk = max_n_kk;
    m->cap_ll = cap_ll; m->cap_kl = cap_kl; m->cap_maxlig = max_lig_pg; m->cap_maxkp = max_kp_pg;
    m->tile_cap = tile_cap;
    return KPD_OK;
}

extern "C" kpd_status kpd_egnn_forward(kpd_egnn *m, const kpd_batch *bt, const float *t_dev, float *eps_h, float *eps_x,
                                       void *stream) {
    KPD_REQUIRE(m && bt && t_dev && eps_h && eps_x, KPD_ERR_INVALID, "null argument");
    KPD_REQUIRE(m->committed, KPD_ERR_STATE, "kpd_egnn_forward before kpd_egnn_commit");
    KPD_REQUIRE(bt->B >= 1 && bt->n_lig >= 1 && bt->n_kp >= 1, KPD_ERR_INVALID, "empty batch (B=%d n_lig=%d n_kp=%d)", bt->B, bt->n_lig, bt->n_kp);
    KPD_REQUIRE(bt->B <= m->cap_B && bt->n_lig <= m->cap_lig && bt->n_kp <= m->cap_kp && bt->n_kk <= m->cap_kk &&
                    bt->max_lig <= m->cap_maxlig && bt->max_kp <= m->cap_maxkp,
                KPD_ERR_CAPACITY, "batch (B=%d lig=%d kp=%d kk=%d maxlig=%d maxkp=%d) exceeds reserved workspace (%d %d %d %d %d %d)",
                bt->B, bt->n_lig, bt->n_kp, bt->n_kk, bt->max_lig, bt->max_kp, m->cap_B, m->cap_lig, m->cap_kp, m->cap_kk,
                m->cap_maxlig, m->cap_maxkp);
    KPD_REQUIRE(!m->cfg.update_kp_feat || bt->n_kk == 0 || (bt->kk_src && bt->kk_dst), KPD_ERR_INVALID, "kk edges missing");
    KPD_REQUIRE(bt->kk_rowptr, KPD_ERR_INVALID, "kk_rowptr missing");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const kpd_egnn_config &c = m->cfg;

    KPD_HIP(hipMemcpyAsync(m->x[NT_LIG], bt->lig_x, (size_t)bt->n_lig * 12, hipMemcpyDeviceToDevice, st));
    KPD_HIP(hipMemcpyAsync(m->x[NT_KP], bt->kp_x, (size_t)bt->n_kp * 12, hipMemcpyDeviceToDevice, st));
    KPD_TRY(launch_node_graph_index(bt->lig_ptr, bt->B, bt->n_lig, m->bidx[NT_LIG], st));
    KPD_TRY(launch_node_graph_index(bt->kp_ptr, bt->B, bt->n_kp, m->bidx[NT_KP], st));
    KPD_TRY(launch_lig_graph(bt, c.ll_cutoff, c.ll_k, c.kl_cutoff, c.kl_k, &m->lg, m->ll_deg, m->ll_off, m->kl_off, m->kl_pg, st));
    // LigRecEGNN.forward returns only (h_lig, x_lig) (dynamics.py:288-294): in the final layer the lk / kk messages and
    // the keypoint update feed nothing, so that layer runs the ll + kl edge types and the ligand update only (the GVP
    // reference drops those edge types itself, dynamics_gvp.py:67-72).  A consumer of the h_kp / x_kp debug taps asks for
    // the full layer with "prune=0".
    const int tr = TM;                                     // edges per tile of the edge kernel
    const int active = c.update_kp_feat ? 0xF : 0x3;
    const bool prune = c.update_kp_feat && m->prune_last;
    const int active_last = prune ? 0x3 : active;
    KPD_TRY(launch_egnn_meta(m->lg.counts, bt->n_kk, active, active_last, bt->lig_ptr, bt->kp_ptr, m->lg.ll_per_graph,
                             bt->kk_rowptr, bt->B, m->kl_off, c.message_norm, c.update_kp_feat, m->meta, m->z[NT_LIG],
                             m->z[NT_KP], st, tr));
    KPD_TRY(launch_embed(bt->lig_h, bt->n_lig, c.atom_nf, m->le_W0, m->le_b0, 64, m->le_W1t, m->le_b1, t_dev,
                         m->bidx[NT_LIG], m->h[NT_LIG], 0, st));
    KPD_TRY(launch_embed(bt->kp_h, bt->n_kp, c.rec_nf, m->re_W0, m->re_b0, 2 * c.rec_nf, m->re_W1t, m->re_b1, t_dev,
                         m->bidx[NT_KP], m->h[NT_KP], m->rec_identity ? 1 : 0, st));

    // tile capacity for this batch (host-known upper bound; the kernel exits early past the device-side total)
    const int e_kl_cap = bt->n_kp * (c.kl_k > 0 ? c.kl_k : std::min(bt->max_lig, 100));
    const int E_cap[4] = {std::max<int>((long)bt->n_lig * std::min(bt->max_lig - 1, c.ll_k > 0 ? c.ll_k : 200), 1), e_kl_cap, e_kl_cap,
                          bt->n_kk};
    int tile_cap = 0, tile_cap_last = 0;
    for (int et = 0; et < m->n_et; ++et) {
        tile_cap += cdiv(E_cap[et], tr);
        if ((active_last >> et) & 1) tile_cap_last += cdiv(E_cap[et], tr);
    }

    const int n[2] = {bt->n_lig, bt->n_kp};
    const int *esrc[4] = {m->lg.ll_src, m->lg.kl_src, m->lg.lk_src, bt->kk_src};
    const int *edst[4] = {m->lg.ll_dst, m->lg.kl_dst, m->lg.lk_dst, bt->kk_dst};
    const int *rowptr[4] = {m->lg.ll_rowptr, m->lg.kl_rowptr, m->lg.lk_rowptr, bt->kk_rowptr};
    const int n_layers = m->debug_layers >= 0 ? std::min(m->debug_layers, c.n_layers) : c.n_layers;

    // projection slots a layer needs = the (edge, coord) slot pairs of the endpoints of its active edge types
    auto slot_mask = [&](int etmask, int nt) {
        int mk = 0;
        for (int et = 0; et < 4; ++et)
            if ((etmask >> et) & 1) {
                if (kSrcNt[et] == nt) mk |= 3 << kSrcSlot[et];
                if (kDstNt[et] == nt) mk |= 3 << kDstSlot[et];
            }
        return mk;
    };
    auto layer_etmask = [&](int li) { return li == n_layers - 1 ? active_last : active; };
    for (int li = 0; li < n_layers; ++li) {
        const LayerW &L = m->L[li];
        {
            {
                ProjPair pp;
                memset(&pp, 0, sizeof(pp));
                for (int nt = 0; nt < 2; ++nt) {
                    ProjArgs &pa = pp.nt[nt];
                    pa.h = m->h[nt]; pa.n = n[nt]; pa.P = m->P[nt];
                    int k = 0;
                    const int mk = slot_mask(layer_etmask(li), nt);
                    for (int s = 0; s < NSLOT; ++s)
                        if (L.wp_p[nt][s] && ((mk >> s) & 1)) {
                            pa.wp[k] = L.wp_p[nt][s]; pa.wx[k] = L.wx_p[nt][s]; pa.bias[k] = L.b_p[nt][s]; pa.slot[k] = s;
                            pa.chain[k] = L.ch_p[nt][s]; pa.wcol[k] = L.wcol_p[nt][s]; pa.chain_h[k] = L.chh_p[nt][s];
                            ++k;
                        }
                    pp.n_slots[nt] = k;
                }
                pp.tiles0 = cdiv(n[0], TM);
                pp.gemm_mode = (m->h_parts & 2) ? m->gemm_mode : 0;
                KPD_TRY(launch_proj_chain(pp, st));
            }
        }
        EdgeArgs ea;
        memset(&ea, 0, sizeof(ea));
        const bool last = li == n_layers - 1;
        ea.meta = last ? m->meta + 16 : m->meta;
        ea.x[0] = m->x[0]; ea.x[1] = m->x[1]; ea.P[0] = m->P[0]; ea.P[1] = m->P[1];
        ea.use_tanh = c.use_tanh; ea.coords_range = c.coords_range;
        ea.stamps = m->stamps;
        ea.dbg = m->edge_dbg;
        ea.tile_rows = tr;
        ea.gemm_mode = (tr == TM && (m->h_parts & 1)) ? m->gemm_mode : 0;
        for (int et = 0; et < 4; ++et) {
            ea.src[et] = esrc[et]; ea.dst[et] = edst[et];
            ea.src_nt[et] = kSrcNt[et]; ea.dst_nt[et] = kDstNt[et]; ea.src_slot[et] = kSrcSlot[et]; ea.dst_slot[et] = kDstSlot[et];
            ea.wr_e[et] = L.wr_e[et]; ea.wr_c[et] = L.wr_c[et];
            ea.wp_e[et] = L.wp_e[et]; ea.wx_e[et] = L.wx_e[et]; ea.b_e[et] = L.b_e[et];
            ea.wp_c[et] = L.wp_c[et]; ea.wx_c[et] = L.wx_c[et]; ea.b_c[et] = L.b_c[et];
            ea.wh_e[et] = L.wh_e[et]; ea.wh_c[et] = L.wh_c[et];
            ea.watt[et] = L.watt[et]; ea.w3[et] = L.w3[et];
            ea.hn_main[et] = m->hn_main[et]; ea.hn_cont[et] = m->hn_cont[et];
            ea.xn_main[et] = m->xn_main[et]; ea.xn_cont[et] = m->xn_cont[et];
        }
        const bool prof = m->prof_on && m->prof_used + 2 <= m->prof_ev.size();
        if (prof) KPD_HIP(hipEventRecord(m->prof_ev[m->prof_used], st));
        KPD_TRY(launch_egnn_edge(ea, last ? tile_cap_last : tile_cap, st));
        if (prof) {
            KPD_HIP(hipEventRecord(m->prof_ev[m->prof_used + 1], st));
            m->prof_used += 2;
        }
        auto fill_update = [&](NodeArgs &na, int nt) {
            na.n = n[nt]; na.h = m->h[nt]; na.x = m->x[nt]; na.bidx = m->bidx[nt]; na.z = m->z[nt];
            int k = 0;
            for (int et = 0; et < m->n_et; ++et)
                if (kDstNt[et] == nt && ((layer_etmask(li) >> et) & 1)) {
                    na.rowptr[k] = rowptr[et];
                    na.hn_main[k] = m->hn_main[et]; na.hn_cont[k] = m->hn_cont[et];
                    na.xn_main[k] = m->xn_main[et]; na.xn_cont[k] = m->xn_cont[et];
                    ++k;
                }
            na.n_in = k;
            na.wp_a = L.wp_a[nt]; na.wx_a = L.wx_a[nt]; na.wp_b = L.wp_b[nt]; na.wx_b = L.wx_b[nt]; na.b0 = L.b0[nt];
            na.wp_2 = L.wp_2[nt]; na.wx_2 = L.wx_2[nt]; na.b2 = L.b2[nt]; na.ln_w = L.ln_w[nt]; na.ln_b = L.ln_b[nt];
            na.ln_inv_n = 1.0f / (float)(c.hidden_nf + 1); na.ln_pad = (float)(HID - c.hidden_nf);
            na.wh_a = L.wh_a[nt]; na.wh_b = L.wh_b[nt]; na.wh_2 = L.wh_2[nt];
            na.norm = c.norm;
            na.tile_shift = 6;
        };
        {
            NodeLayerPair lp;
            memset(&lp, 0, sizeof(lp));
            for (int nt = 0; nt < 2; ++nt) {
                NodeLayerArgs &na = lp.nt[nt];
                na.u.n = n[nt]; na.u.h = m->h[nt];
                if (nt < m->n_upd && !(nt == NT_KP && last && prune)) {
                    fill_update(na.u, nt);
                    na.do_update = 1;
                }
                if (!na.do_update && !na.do_proj) na.u.n = 0;          // nothing to do for this node type
            }
            lp.tiles0 = cdiv(lp.nt[0].u.n, TN);
            lp.stamps = m->stamps ? m->stamps + 16 : nullptr;
            lp.gemm_mode = (m->h_parts & 4) ? m->gemm_mode : 0;
            KPD_TRY(launch_node_layer(lp, st));
        }
    }
    KPD_TRY(launch_decode(m->h[NT_LIG], m->x[NT_LIG], bt->lig_x, bt->n_lig, c.atom_nf, 2 * c.atom_nf, m->de_W0, m->de_b0,
                          m->de_W1, m->de_b1, eps_h, eps_x, st));
    return KPD_OK;
}

extern "C" kpd_status kpd_egnn_debug_state(kpd_egnn *m, const char *what, float *out, int64_t n_floats, void *stream) {
    KPD_REQUIRE(m && what && out, KPD_ERR_INVALID, "null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const std::string w(what);
    const float *src = nullptr;
    if (w == "h_lig") src = m->h[0];
    else if (w == "h_kp") src = m->h[1];
    else if (w == "x_lig") src = m->x[0];
    else if (w == "x_kp") src = m->x[1];
    else if (w == "z_lig") src = m->z[0];
    else if (w == "z_kp") src = m->z[1];
    else if (w.size() == 4 && (w.rfind("xnm", 0) == 0 || w.rfind("xnc", 0) == 0 || w.rfind("hnm", 0) == 0 || w.rfind("hnc", 0) == 0) &&
             w[3] >= '0' && w[3] < '4') {          // segment-sum pieces of edge type w[3] as the last layer left them
        const int et = w[3] - '0';
        src = w[0] == 'x' ? (w[2] == 'm' ? m->xn_main[et] : m->xn_cont[et]) : (w[2] == 'm' ? m->hn_main[et] : m->hn_cont[et]);
    }
    else if (w.rfind("layers=", 0) == 0) {
        m->debug_layers = atoi(w.c_str() + 7);
        return KPD_OK;
    } else if (w.rfind("prune=", 0) == 0) {        // A/B switch of the final-layer pruning (tests: bit-identical eps)
        m->prune_last = atoi(w.c_str() + 6);
        return KPD_OK;
    } else if (w.rfind("gemm=", 0) == 0) {         // "gemm=f32" (exact fp32 MFMA, the contract path) | "gemm=f16x2" (split f16 products)
        const std::string v = w.substr(5);
        KPD_REQUIRE(v == "f32" || v == "f16x2", KPD_ERR_INVALID, "gemm mode must be f32 or f16x2");
        KPD_REQUIRE(v == "f32" || !m->committed || m->f16_ok, KPD_ERR_WEIGHTS, "%s", F16_RANGE_ERROR);
        m->gemm_mode = v == "f16x2" ? 1 : 0;
        return KPD_OK;
    } else if (w == "edge_dbg=1") {          // allocate the per-row tap buffer for the current workspace ([tile_cap][64][4] floats)
        KPD_REQUIRE(m->tile_cap > 0, KPD_ERR_STATE, "edge_dbg=1 needs a reserved workspace");
        if (m->edge_dbg) (void)hipFree(m->edge_dbg);
        m->edge_dbg_floats = (size_t)m->tile_cap * (TM * 4 + 3 * 4 * 64 * 12);      // per-row taps, then per-tile operand taps of rows 0..2 of every wave
        KPD_HIP(hipMalloc(reinterpret_cast<void **>(&m->edge_dbg), m->edge_dbg_floats * 4));
        KPD_HIP(hipMemsetAsync(m->edge_dbg, 0, m->edge_dbg_floats * 4, st));
        return KPD_OK;
    } else if (w == "edge_dbg") {
        KPD_REQUIRE(m->edge_dbg && (size_t)n_floats <= m->edge_dbg_floats, KPD_ERR_INVALID, "edge_dbg not enabled or request too large");
        KPD_HIP(hipMemcpyAsync(out, m->edge_dbg, (size_t)n_floats * 4, hipMemcpyDeviceToDevice, st));
        return KPD_OK;
    } else if (w == "stamps=1") {            // start accumulating per-phase cycle sums of the edge kernel
        if (!m->stamps) KPD_HIP(hipMalloc(reinterpret_cast<void **>(&m->stamps), 32 * sizeof(unsigned long long)));
        KPD_HIP(hipMemsetAsync(m->stamps, 0, 32 * sizeof(unsigned long long), st));
        return KPD_OK;
    } else if (w == "stamps") {              // read them back (as 32 floats: lo/hi 24-bit split is avoided by copying raw)
        KPD_REQUIRE(m->stamps && n_floats >= 64, KPD_ERR_INVALID, "stamps not enabled or buffer < 64 floats");
        KPD_HIP(hipMemcpyAsync(out, m->stamps, 32 * sizeof(unsigned long long), hipMemcpyDeviceToDevice, st));
        return KPD_OK;
    }
    KPD_REQUIRE(src, KPD_ERR_INVALID, "unknown debug tap '%s'", what);
    KPD_HIP(hipMemcpyAsync(out, src, (size_t)n_floats * 4, hipMemcpyDeviceToDevice, st));
    return KPD_OK;
}

extern "C" kpd_status kpd_egnn_profile(kpd_egnn *m, int32_t enable) {
    KPD_REQUIRE(m, KPD_ERR_INVALID, "null handle");
    if (enable && m->prof_ev.empty()) {
        m->prof_ev.resize(2 * 8192);
        for (hipEvent_t &e : m->prof_ev) KPD_HIP(hipEventCreate(&e));
    }
    m->prof_on = enable != 0;
    m->prof_used = 0;
    return KPD_OK;
}

extern "C" kpd_status kpd_egnn_profile_read(kpd_egnn *m, double *total_ms, int32_t *launches) {
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

extern "C" kpd_status kpd_egnn_last_counts(kpd_egnn *m, int32_t out[8], void *stream) {
    KPD_REQUIRE(m && out, KPD_ERR_INVALID, "null argument");
    for (int i = 0; i < 7; ++i) out[i] = 0;
    out[7] = m->gemm_mode;                                           // GEMM mode the next forward runs in: 0 exact fp32, 1 f16x2
    if (!m->ws.base) return KPD_OK;                                  // no forward yet: only the mode is meaningful
    hipStream_t st = static_cast<hipStream_t>(stream);
    int host[25];
    KPD_HIP(hipMemcpyAsync(host, m->meta, sizeof(host), hipMemcpyDeviceToHost, st));
    KPD_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < 4; ++i) out[i] = host[i];
    out[4] = host[8];
    out[5] = host[16 + 8];                                           // tiles of the final layer's edge launch
    out[6] = host[16] + host[17] + host[18] + host[19];              // edges of the final layer's edge launch
    return KPD_OK;
}

extern "C" kpd_status kpd_build_lig_graph(const kpd_batch *bt, float ll_cutoff, int32_t ll_k, float kl_cutoff, int32_t kl_k,
                                          const kpd_lig_graph *out, void *stream) {
    KPD_REQUIRE(bt && out, KPD_ERR_INVALID, "null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    // scratch: carve from a small per-call allocation (this entry point is for tests and
    // standalone use; the engines use their own workspace and never allocate per call)
    int *tmp = nullptr;
    const size_t cnt = (size_t)bt->n_lig + 3 * ((size_t)bt->B + 2);
    KPD_HIP(hipMalloc(reinterpret_cast<void **>(&tmp), cnt * sizeof(int)));
    kpd_status s = launch_lig_graph(bt, ll_cutoff, ll_k, kl_cutoff, kl_k, out, tmp, tmp + bt->n_lig, tmp + bt->n_lig + bt->B + 2,
                                    tmp + bt->n_lig + 2 * (bt->B + 2), st);
    hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    if (s != KPD_OK) return s;
    KPD_HIP(e);
    return KPD_OK;
}
