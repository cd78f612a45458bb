// Shared host-side GVP weight handling (see gvp_host.h).
#include "gvp_host.h"

namespace kpd {

std::vector<std::string> split_dots(const std::string &s) {
    std::vector<std::string> out;
    size_t p = 0;
    while (true) {
        size_t q = s.find('.', p);
        out.push_back(s.substr(p, q == std::string::npos ? q : q - p));
        if (q == std::string::npos) break;
        p = q + 1;
    }
    return out;
}

size_t gvp_arena_bytes(int S) {
    return (256 + 16 + 4 * (size_t)(S / 8) * 2048 + 256 + 2 * (size_t)(S / 16 + 2) * (S / 16) * 256 + 12 * 256) * 4 + 16384;
}

void alloc_gvp(Arena &A, HostGvp &g, std::set<std::string> &expected, const std::string &prefix) {
    g.h = std::max(g.vin, g.vout);
    g.b = A.take<float>(256);
    g.bg = A.take<float>(16);
    if (g.split != SPLIT_NONE) {
        g.wproj = A.take<float>((size_t)(g.S / 8) * 2048);
        g.bproj = A.take<float>(256);
        if (g.S == 256) g.wproj_h = A.take<float>((size_t)(g.S / 8) * 2048);
    }
    if (g.split == SPLIT_SRC_DST) {
        g.wproj_dst = A.take<float>((size_t)(g.S / 8) * 2048);
        if (g.S == 256) g.wproj_dst_h = A.take<float>((size_t)(g.S / 8) * 2048);
    }
    // a split first Linear exists only at the head of an edge-message chain
    if ((g.split != SPLIT_NONE) != (g.chain_pos == 0)) set_error("internal: GVP split/chain position mismatch");
    g.chain = A.take<float>((size_t)g.chain_chunks() * (g.sout / 16) * 256);
    if (g.has_h()) g.chain_h = A.take<float>((size_t)g.chain_chunks() * (g.sout / 16) * 256);
    g.whp = A.take<float>(g.chain_pos == 0 ? 9 * 256 : 256);
    g.wup = A.take<float>((size_t)g.n_ht() * 256);
    for (const char *s : {".Wh", ".Wu", ".to_feats_out.0.weight", ".to_feats_out.0.bias", ".scalar_to_vector_gates.weight",
                          ".scalar_to_vector_gates.bias"})
        expected.insert(prefix + s);
}

kpd_status want_shape(const char *name, const int64_t *shape, int ndim, std::initializer_list<int64_t> want) {
    bool ok = ndim == (int)want.size();
    int i = 0;
    for (int64_t w : want) {
        if (ok && shape[i] != w) ok = false;
        ++i;
    }
    if (!ok) {
        std::string got;
        for (int j = 0; j < ndim; ++j) got += std::to_string(shape[j]) + ",";
        std::string exp;
        for (int64_t w : want) exp += std::to_string(w) + ",";
        set_error("weight %s has shape [%s], expected [%s]", name, got.c_str(), exp.c_str());
        return KPD_ERR_WEIGHTS;
    }
    return KPD_OK;
}

kpd_status load_gvp_tensor(HostGvp &g, const std::string &param, const char *name, const float *w, const int64_t *shape,
                           int ndim, hipStream_t st) {
    // Reference shapes carry the true scalar width St = S - cut; the kernels' layouts are S wide.  Every S-wide block of a tensor
    // (output rows of to_feats_out / its bias, its leading input columns, the gate's input columns) is packed with its last `cut`
    // entries zero, and the blocks behind it (rbf, sh) are read from where they sit in the reference tensor.
    const int St = g.S - g.cut;
    const int sout_t = g.sout == g.S ? St : g.sout;                 // (the 64-wide output of the last noise GVP is not an S block)
    const int sin_t = g.s_in - g.cut * (g.split == SPLIT_SRC_DST ? 2 : 1);
    // vector channels: 16-channel blocks (source / destination features, outputs) hold Vt = 16 - vcut channels; x_diff and the
    // single output channel of the last noise GVP are not blocks.  The hidden width follows the reference: max(vin, vout) of the true sizes.
    const int Vt = GV - g.vcut;
    const int vin_t = g.vin - g.vcut * (g.vin / GV), vout_t = g.vout == GV ? Vt : g.vout;
    const int h_t = std::max(vin_t, vout_t);
    const int k_all = sin_t + h_t;
    auto hvalid = [&](int ht) { return std::max(0, std::min(16, h_t - 16 * ht)); };        // valid hidden channels of hidden tile ht
    auto kvalid = [&](int kc) { return std::max(0, std::min(16, St - 16 * kc)); };      // valid k of slab kc of an S block
    if (param == "Wh") {
        KPD_TRY(want_shape(name, shape, ndim, {vin_t, h_t}));
        if (g.chain_pos == 0) {
            // input vectors arrive as [x_diff | Vt source | (Vt destination)] (gvp.py:474-480); the kernel feeds them as
            // tiles [source], [destination], [x_diff]
            const bool dst = g.vin == 2 * GV + 1;
            KPD_TRY(pack_chain_frag(w, 1, h_t, h_t, 1, Vt, g.n_ht(), g.whp, st));
            if (dst) KPD_TRY(pack_chain_frag(w, 1, h_t, h_t, 1 + Vt, Vt, g.n_ht(), g.whp + 3 * 256, st));
            KPD_TRY(pack_chain_frag(w, 1, h_t, h_t, 0, 1, g.n_ht(), g.whp + 6 * 256, st));
        } else {
            KPD_TRY(pack_chain_frag(w, 1, h_t, h_t, 0, vin_t, 1, g.whp, st));
        }
    } else if (param == "Wu") {
        KPD_TRY(want_shape(name, shape, ndim, {h_t, vout_t}));
        for (int ht = 0; ht < g.n_ht(); ++ht)
                KPD_TRY(pack_chain_frag(w, 1, vout_t, vout_t, 16 * ht, hvalid(ht), 1, g.wup + ht * 256, st));
    } else if (param == "to_feats_out.0.weight") {
        KPD_TRY(want_shape(name, shape, ndim, {sout_t, k_all}));
        // node blocks of a split first Linear: S / 16 k-slabs in chunk order for k_gvp_proj_chain
        auto pack_block = [&](int col0, float *dst) -> kpd_status {
            const int nts = g.sout / 16;
            for (int kc = 0; kc < g.S / 16; ++kc)
                KPD_TRY(pack_chain_frag(w, k_all, 1, sout_t, col0 + 16 * kc, kvalid(kc), nts, dst + (size_t)kc * nts * 256, st));
            return KPD_OK;
        };
        if (g.split == SPLIT_SRC) {             // [h_src St | rbf 16 | sh h]
            KPD_TRY(pack_block(0, g.wproj));
        } else if (g.split == SPLIT_SRC_DST) {  // [h_src St | rbf 16 | h_dst St | sh h]
            KPD_TRY(pack_block(0, g.wproj));
            KPD_TRY(pack_block(St + 16, g.wproj_dst));
        }
        {
            const int nts = g.sout / 16, ch = nts * 256;
            int c = 0;
            if (g.chain_pos == 0) {
                const int sh0 = g.split == SPLIT_SRC_DST ? 2 * St + 16 : St + 16;
                KPD_TRY(pack_chain_frag(w, k_all, 1, sout_t, St, 16, nts, g.chain + (size_t)(c++) * ch, st));          // rbf
                for (int ht = 0; ht < g.n_ht(); ++ht)
                    KPD_TRY(pack_chain_frag(w, k_all, 1, sout_t, sh0 + 16 * ht, hvalid(ht), nts, g.chain + (size_t)(c++) * ch, st));
            } else {
                // scalar inputs: an S block (s_in == S) or a narrower input of its own width (no padding inside)
                for (int kc = 0; kc < g.s_in / 16; ++kc)
                    KPD_TRY(pack_chain_frag(w, k_all, 1, sout_t, 16 * kc, g.s_in == g.S ? kvalid(kc) : 16, nts, g.chain + (size_t)(c++) * ch, st));
                KPD_TRY(pack_chain_frag(w, k_all, 1, sout_t, sin_t, hvalid(0), nts, g.chain + (size_t)(c++) * ch, st));
            }
        }
    } else if (param == "to_feats_out.0.bias") {
        KPD_TRY(want_shape(name, shape, ndim, {sout_t}));
        // split: the bias rides with the per-node source projection; the per-edge stage adds nothing
        KPD_TRY(copy_pad(w, sout_t, g.split != SPLIT_NONE ? g.bproj : g.b, 256, st));
    } else if (param == "scalar_to_vector_gates.weight") {
        KPD_TRY(want_shape(name, shape, ndim, {vout_t, sout_t}));
        {
            float *gch = g.chain + (size_t)(g.chain_chunks() - 1) * (g.sout / 16) * 256;
            for (int nt = 0; nt < g.sout / 16; ++nt)
                KPD_TRY(pack_chain_frag(w, sout_t, 1, vout_t, 16 * nt, std::max(0, std::min(16, sout_t - 16 * nt)), 1, gch + nt * 256, st));
        }
    } else if (param == "scalar_to_vector_gates.bias") {
        KPD_TRY(want_shape(name, shape, ndim, {vout_t}));
        KPD_TRY(copy_pad(w, vout_t, g.bg, 16, st));
    } else {
        set_error("unknown GVP parameter '%s'", name);
        return KPD_ERR_WEIGHTS;
    }
    return KPD_OK;
}

}  // namespace kpd
