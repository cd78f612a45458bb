// Argument blocks and launchers of the EGNN kernels (egnn_kernels.hip).
#pragma once
#include "engine.h"

namespace kpd {

constexpr int NSLOT = 8;            // projection slots per node row of P
constexpr int ATT_BIAS_AT = 260;
constexpr int BIAS_K = 260;          // A-tile column that holds the constant 1 multiplying the bias row of edge GEMMs: with the last real
                                     // feature at k = 256 it shares k-step 0 of the last k-group (mfma_core.h, TailSteps)    // soft_attention bias is parked in the pad of its weight row
constexpr int ET_LL = 0, ET_KL = 1, ET_LK = 2, ET_KK = 3;
constexpr int NT_LIG = 0, NT_KP = 1;

constexpr int EDGE_LDS_BYTES = TM * SA * 4 + (TM * 2 + TM + 3 * TM + TM + 3 * TM + 4 * HS + 8 + TM) * 4;   // ... misc[8], 64 column-256 values
// k_egnn_edge_h: two f16 planes of 64 x 280 halves (= the T tile's region), row data, two fp32 head rows, W2 row 256 as 2 x 2 x 272 halves
constexpr int EDGE_H_LDS_BYTES = 64 * 280 * 2 * 2 + (TM * 2 + TM + 3 * TM + TM + 3 * TM + 2 * HS + 8) * 4 + 2 * 2 * 272 * 2;

struct ProjArgs {
    const float *h;                 // [n][HS]
    int n;
    float *P;                       // [n][NSLOT][HS]
    const float *wp[NSLOT];
    const float *wx[NSLOT];
    const float *bias[NSLOT];       // nullptr for src slots
    int slot[NSLOT];
    const float *chain[NSLOT];      // k_proj_ws: the 256 x 256 block as 16 A-fragment chunks, and W[:, 256]
    const float *wcol[NSLOT];
    const void *chain_h[NSLOT];     // f16x2 mode: the same block as f16 hi / lo planes (pack_proj_f16_split)
};

struct ProjPair {
    ProjArgs nt[2];
    int tiles0;                     // workgroups (64-node tiles) of nt[0]
    int n_slots[2];
    int slots_per_block;            // consecutive slots one workgroup computes from its resident h registers
    int gemm_mode;                  // 0: exact fp32 MFMA, 1: f16x2 split (k_proj_ws_h)
};

struct EdgeArgs {
    const int *meta;                // [9] device: E[4], first tile[5]
    const int *src[4];
    const int *dst[4];
    const float *x[2];              // current coordinates [n][3]: lig, kp
    const float *P[2];
    int src_nt[4], dst_nt[4], src_slot[4], dst_slot[4];
    const float *wr_e[4], *wr_c[4];
    const float *wp_e[4], *wx_e[4], *b_e[4];
    const float *wp_c[4], *wx_c[4], *b_c[4];
    const float *watt[4];
    const float *w3[4];
    float *hn_main[4], *hn_cont[4];
    float *xn_main[4], *xn_cont[4];
    int use_tanh;
    float coords_range;
    unsigned long long *stamps;     // [16] phase-cycle sums, diagnostics only (null in production)
    const void *wh_e[4], *wh_c[4];  // f16x2 mode: W2 of edge_mlp / coord_mlp as f16 hi / lo planes (pack_f16_split)
    int gemm_mode;                  // 0: exact fp32 MFMA (contract path), 1: f16x2 split products (opt-in)
    int tile_rows;                  // edges per tile: 64 (TM)
    int ablate;                     // timing experiments only (KPD_EDGE_ABLATE), 0 in production
    int split_slots;                // workgroup slots per XCD (set by the launcher): the tiles of an XCD's last round run one branch per work item; 0: off
    float *dbg;                     // [tiles][64][4] per-row taps of the coordinate branch (builds with -DKPD_EDGE_DBG only; "edge_dbg=1")
};

// Forward edge kernel of the EGNN trainer (k_egnn_edge_train): the inference kernel's program on the current weights, keeping what the
// backward pass reads.
struct EdgeTrainArgs {
    const int *meta;                // as EdgeArgs (64-edge tiles)
    const int *src[4];
    const int *dst[4];
    const float *x[2];
    const float *P[2];              // the layer's projection blocks [n][NSLOT][HS] of both node types (b1 folded into the dst slots)
    int src_nt[4], dst_nt[4];
    int slot[4][2][2];              // [et][branch][src | dst] -> slot of P on that side's node type
    const float *wr[4][2], *wp[4][2], *wx[4][2];      // per (et, branch): radial row, packed second Linear (bias row at BIAS_K), its row 256
    const float *watt[4], *w3[4];   // per et: soft-attention row (bias at ATT_BIAS_AT), coordinate-head row
    float *hn_main[4], *hn_cont[4], *xn_main[4], *xn_cont[4];
    int use_tanh;
    float coords_range;
    float *keep[4][2][4];           // [et][branch][pre1, a1, pre2, a2], each [E][HS]
    float *att[4], *sc[4], *dij[4], *xdiff[4], *nvec[4];
    unsigned long long *stamps;     // [64] phase-cycle sums of both training edge kernels (TOOLS build, KPD_TRAIN_STAMPS; null in production)
    int skip;                       // TOOLS build (KPD_TR_SKIP): bit 0 / 1 / 2 = leave out the pre1 + a1 / pre2 / geometry stores (timing experiments; results are wrong)
};

struct EdgePackEntry {
    const float *W1, *W2, *b2;      // [257][515], [257][257], [257] of the branch
    const float *head, *head_b;     // soft_attention weight (+ bias) or coord_mlp.4 weight
    float *wp, *wx, *wr, *head_out; // outputs (head_out: watt or w3)
};
struct EdgePackTab {
    EdgePackEntry e[8];
    int n;
    int transposed;                 // 1: wp / wx hold W2^T without a bias row (backward: dpre1 = dpre2 W2)
};

// Backward edge kernel of the EGNN trainer (k_egnn_edge_bwd).  In place: dpre2 over keep[..][2], dpre1 over keep[..][0], ds over att,
// d dij over sc, dn over nvec.
struct EdgeBwdArgs {
    const int *meta;                // as EdgeArgs (64-edge tiles)
    const int *dst[4];
    int dst_nt[4];
    const float *dhn[2];            // dL / d(h_neigh / z) of the layer per node type [n][HS] (null where the type is not updated)
    const float *dxo[2];            // dL / d x_out [n][3]
    const float *zinv[2];
    float *keep[4][2][4];           // [et][branch][pre1 -> dpre1, a1, pre2 -> dpre2, a2]
    float *att[4], *sc[4], *nvec[4];
    const float *dij[4];
    const float *wa[4], *w3[4];     // head rows (padded to HS)
    const float *wpT[4][2], *wxT[4][2], *wr[4][2];
    float *dv_main[4][2], *dv_cont[4][2], *dvw_main[4][2], *dvw_cont[4][2];
    float *part[2];                 // per branch: [tiles of the layer][2][part_ld] column-sum partials (head weight, b2)
    int part_ld;
    int use_tanh;
    float coords_range;
    unsigned long long *stamps;     // as EdgeTrainArgs (slots 32 ..)
    int skip;                       // TOOLS build (KPD_TR_SKIP): bit 3 / 4 = leave out the dpre2 / dpre1 stores
};

struct NodeArgs {
    int n;
    float *h;                       // [n][HS] in/out
    float *x;                       // [n][3] in/out
    const int *bidx;
    const float *z;                 // [B]
    int n_in;
    const int *rowptr[2];
    const float *hn_main[2], *hn_cont[2];
    const float *xn_main[2], *xn_cont[2];
    const float *wp_a, *wx_a, *wp_b, *wx_b, *b0;
    const float *wp_2, *wx_2, *b2;
    const void *wh_a, *wh_b, *wh_2; // f16x2 mode: wp_a / wp_b / wp_2 as f16 hi / lo planes (pack_f16_split)
    const float *ln_w, *ln_b;
    int norm;
    float ln_inv_n, ln_pad;         // LayerNorm over hidden_nf + 1 features: 1 / (hidden_nf + 1), 256 - hidden_nf (pad columns hold zeros)
    int tile_shift;                 // log2 of the edge-tile size the segment pieces (main / cont) were written with
};

// Fused node kernel (32-node tiles): [update of layer i] -> [first-layer projections of layer i + 1].
struct NodeLayerArgs {
    NodeArgs u;                     // update part (do_update) -- u.n / u.h are always valid
    int do_update, do_proj;
    float *P;                       // [n][NSLOT][HS]
    int n_slots;
    const float *wp[NSLOT], *wx[NSLOT], *bias[NSLOT];
    int slot[NSLOT];
};

struct NodeLayerPair {
    NodeLayerArgs nt[2];
    int tiles0;                     // 32-node tiles of nt[0]
    unsigned long long *stamps;     // [16] phase-cycle sums (diagnostics only, null in production)
    int dbg;                        // ablation switches for timing experiments (KPD_NODE_ABLATE), 0 in production
    int gemm_mode;                  // 0: exact fp32 MFMA, 1: f16x2 split (k_node_update8_h)
};

constexpr int TN = 32;              // rows per workgroup of the fused node kernel
constexpr int NODE_LAYER_LDS_BYTES = TN * SA * 4 + 3 * TN * 4 + 3 * HS * 4;        // tile, z / mean / rstd per row, the three row-dot vectors
// k_node_update8_h: two f16 planes of 32 x 280 halves (35 840 B) shared with the fp32 tile (34 304 B)
constexpr int NODE_H_TILE_FLOATS = 2 * TN * 280 / 2;
constexpr int NODE_H_LDS_BYTES = NODE_H_TILE_FLOATS * 4 + 3 * TN * 4;


kpd_status egnn_kernels_init();
kpd_status launch_node_graph_index(const int *ptr, int B, int n, int *bidx, hipStream_t st);
kpd_status launch_egnn_meta(const int *counts, int e_kk, int active_mask, int active_last, const int *lig_ptr, const int *kp_ptr,
                            const int *ll_per_graph, const int *kk_rowptr, int B, const int *kl_off, float message_norm,
                            int update_kp, int *meta, float *z_lig, float *z_kp, hipStream_t st, int tile_rows = TM);
kpd_status launch_embed(const float *in, int n, int fin, const float *W0, const float *b0, int hid, const float *W1t,
                        const float *b1, const float *t, const int *bidx, float *out, int identity, hipStream_t st);
kpd_status launch_decode(const float *h, const float *x, const float *x0, int n, int atom_nf, int hid, const float *W0,
                         const float *b0, const float *W1, const float *b1, float *eps_h, float *eps_x, hipStream_t st);
kpd_status launch_egnn_edge(const EdgeArgs &a, int tile_cap, hipStream_t st);
kpd_status launch_egnn_edge_train(const EdgeTrainArgs &a, int tile_cap, hipStream_t st);
// h_neigh / x_neigh of both node types += zinv x (the per-tile message pieces of every live edge type into that type), in edge-type order
struct EdgePiecesSumArgs {
    const float *hn_main[4], *hn_cont[4], *xn_main[4], *xn_cont[4];
    const int *rowptr[4];
    int live[4], dst_nt[4];
    const float *zinv[2];
    int n[2];                     // 0: this node type receives nothing in this launch
    float *hn[2], *xn[2];
};
kpd_status launch_edge_pieces_sum(const EdgePiecesSumArgs &a, hipStream_t st);
kpd_status launch_edge_train_pack(const EdgePackTab &t, hipStream_t st);
kpd_status launch_egnn_edge_bwd(const EdgeBwdArgs &a, int tile_cap, hipStream_t st);
// dV / dVw of up to eight (edge type, branch) pairs from the per-tile pieces of k_egnn_edge_bwd (o2 may be null)
struct EdgePiecesBatch {
    struct One {
        const float *m1, *c1, *m2, *c2;
        const int *rowptr;
        int n;
        float *o1, *o2;
    } e[8];
    int ldo;
};
kpd_status launch_edge_pieces_set(const EdgePiecesBatch &b, int count, hipStream_t st);
kpd_status launch_proj_chain(const ProjPair &p, hipStream_t st);
kpd_status launch_node_layer(const NodeLayerPair &p, hipStream_t st);

}  // namespace kpd
