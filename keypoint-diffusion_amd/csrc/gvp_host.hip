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
    return ((size_t)NG_G * 2048 + 2 * GVH * GVH + 256 + 16 * 256 + 16 + 2 * (size_t)(S / 8) * 2048 + 256) * 4 + 8192;
}

void alloc_gvp(Arena &A, HostGvp &g, std::set<std::string> &expected, const std::string &prefix) {
    g.h = std::max(g.vin, g.vout);
    const int k_edge = g.edge_scalars() + g.h;
    g.ng = (k_edge + 7) / 8;
    g.Wh = A.take<float>(g.vin * g.h);
    g.Wu = A.take<float>(g.h * g.vout);
    g.wp = A.take<float>((size_t)g.ng * 2048);
    g.b = A.take<float>(256);
    g.wg = A.take<float>((size_t)(g.sout / 16) * 256);
    g.bg = A.take<float>(16);
    if (g.split != SPLIT_NONE) {
        g.wproj = A.take<float>((size_t)(g.S / 8) * 2048);
        g.bproj = A.take<float>(256);
    }
    if (g.split == SPLIT_SRC_DST) g.wproj_dst = A.take<float>((size_t)(g.S / 8) * 2048);
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
    const int k_all = g.s_in + g.h;
    if (param == "Wh") {
        KPD_TRY(want_shape(name, shape, ndim, {g.vin, g.h}));
        KPD_TRY(copy_pad(w, g.vin * g.h, g.Wh, g.vin * g.h, st));
    } else if (param == "Wu") {
        KPD_TRY(want_shape(name, shape, ndim, {g.h, g.vout}));
        KPD_TRY(copy_pad(w, g.h * g.vout, g.Wu, g.h * g.vout, st));
    } else if (param == "to_feats_out.0.weight") {
        KPD_TRY(want_shape(name, shape, ndim, {g.sout, k_all}));
        if (g.split == SPLIT_SRC) {             // [h_src S | rbf 16 | sh h]
            KPD_TRY(pack_gemm_weight_ng(w, g.sout, k_all, 0, g.S, g.S / 8, g.wproj, st));
            KPD_TRY(pack_gemm_weight_ng(w, g.sout, k_all, g.S, k_all - g.S, g.ng, g.wp, st));
        } else if (g.split == SPLIT_SRC_DST) {  // [h_src S | rbf 16 | h_dst S | sh h]
            KPD_TRY(pack_gemm_weight_ng(w, g.sout, k_all, 0, g.S, g.S / 8, g.wproj, st));
            KPD_TRY(pack_gemm_weight_ng(w, g.sout, k_all, g.S + 16, g.S, g.S / 8, g.wproj_dst, st));
            KPD_TRY(pack_gemm_weight_2ranges(w, g.sout, k_all, g.S, 16, 2 * g.S + 16, g.h, g.ng, g.wp, st));
        } else {
            KPD_TRY(pack_gemm_weight_ng(w, g.sout, k_all, 0, k_all, g.ng, g.wp, st));
        }
    } else if (param == "to_feats_out.0.bias") {
        KPD_TRY(want_shape(name, shape, ndim, {g.sout}));
        // split: the bias rides with the per-node source projection; the per-edge stage adds nothing
        KPD_TRY(copy_pad(w, g.sout, g.split != SPLIT_NONE ? g.bproj : g.b, 256, st));
    } else if (param == "scalar_to_vector_gates.weight") {
        KPD_TRY(want_shape(name, shape, ndim, {g.vout, g.sout}));
        KPD_TRY(pack_gate_weight(w, g.vout, g.sout, g.wg, st));
    } else if (param == "scalar_to_vector_gates.bias") {
        KPD_TRY(want_shape(name, shape, ndim, {g.vout}));
        KPD_TRY(copy_pad(w, g.vout, g.bg, 16, st));
    } else {
        set_error("unknown GVP parameter '%s'", name);
        return KPD_ERR_WEIGHTS;
    }
    return KPD_OK;
}

}  // namespace kpd
