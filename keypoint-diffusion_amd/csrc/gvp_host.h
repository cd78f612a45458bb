// Host-side description of one GVP's weights and their device buffers, shared by the GVP denoiser
// engine (gvp.hip) and the GVP receptor encoder engine (rec_encoder.hip).
#pragma once
#include <algorithm>
#include <set>
#include <string>
#include <vector>

#include "gvp_kernels.h"

namespace kpd {

// How the first Linear (to_feats_out) of a message GVP is split.  Its input is
// [h_src (S) | rbf (16) | (h_dst (S)) | sh (h)]; blocks that depend on one node only are applied per
// node by k_gvp_proj and enter the edge stage as gathered per-row terms.
enum GvpSplit { SPLIT_NONE = 0, SPLIT_SRC = 1, SPLIT_SRC_DST = 2 };

struct HostGvp {
    int vin = 0, h = 0, vout = 0, s_in = 0, sout = 0;   // s_in = scalar inputs of to_feats_out (without sh)
    int split = SPLIT_NONE;
    int S = 0;                                          // width of the node blocks when split
    int cut = 0;                                        // n_hidden_scalars below the kernels' width S: the last `cut` entries of every
                                                        // S-wide block are padding (zero weight rows / columns), see load_gvp_tensor
    int vcut = 0;                                       // vector_size below the kernels' 16 channels: likewise for every 16-channel block
    float *b = nullptr, *bg = nullptr;                  // to_feats_out bias (zero padded to 256), gate bias [16]
    float *wproj = nullptr, *bproj = nullptr;           // h_src block (+ bias)
    float *wproj_dst = nullptr;                         // h_dst block
    float *wproj_h = nullptr, *wproj_dst_h = nullptr;   // f16x2 re-packs (S = 256)
    int vec_sigmoid = 1;
    int chain_pos = 1;                                  // 0: head of an edge-message chain (split first Linear), >= 1: any other GVP
    float *chain = nullptr, *whp = nullptr, *wup = nullptr;   // 16x16x4 A fragments for the chained kernels (gvp_chain.hip)
    float *chain_h = nullptr;                           // f16x2 re-pack of `chain` (GVPs with 256 scalar outputs; 256 scalar inputs unless head)
    bool has_h() const { return sout == 256 && (chain_pos == 0 || s_in == 256); }
    int n_ht() const { return (h + 15) / 16; }
    // k-slabs of to_feats_out ([rbf | sh tiles] at the head of a message chain, [s_in / 16 scalar slabs | sh] otherwise) + gates
    int chain_chunks() const { return chain_pos == 0 ? 2 + n_ht() : s_in / 16 + 2; }
    GvpW dev() const {
        GvpW w;
        w.b = b; w.bg = bg;
        w.vin = vin; w.h = h; w.vout = vout;
        w.sout = sout; w.vec_sigmoid = vec_sigmoid;
        w.chain = chain; w.whp = whp; w.wup = wup;
        w.chain_h = chain_h;
        return w;
    }
};

std::vector<std::string> split_dots(const std::string &s);
void alloc_gvp(Arena &A, HostGvp &g, std::set<std::string> &expected, const std::string &prefix);
kpd_status want_shape(const char *name, const int64_t *shape, int ndim, std::initializer_list<int64_t> want);
kpd_status load_gvp_tensor(HostGvp &g, const std::string &param, const char *name, const float *w, const int64_t *shape,
                           int ndim, hipStream_t st);
// bytes one HostGvp can take from the arena (upper bound)
size_t gvp_arena_bytes(int S);

}  // namespace kpd
