// Argument blocks and launchers of the GVP kernels (gvp_kernels.hip).
#pragma once
#include "engine.h"

namespace kpd {

constexpr int GV = 16;                 // vector channels (vector_size; 16 in every config)
constexpr int GVH = 33;                // largest hidden vector width (x_diff + 16 source + 16 destination vectors)

// One GVP (models/gvp.py:43-116) in kernel-ready form.
struct GvpW {
    const float *b;       // [256] to_feats_out bias, zero padded (zero for the split head of a message chain)
    const float *bg;      // [16] gate bias
    int vin, h, vout;
    int sout;             // scalar outputs (S, or 64 for the noise head)
    int vec_sigmoid;      // 1: sigmoid gate, 0: identity (last noise GVP)
    // 16x16x4 A-operand fragments, see HostGvp / load_gvp_tensor
    const float *chain;   // weight chunks: to_feats_out k-slabs, then the gate slab
    const float *chain_h; // f16x2 mode: the same chunks re-packed (pack_gvp_chain_h); null when the GVP has no such form
    const float *whp;     // Wh fragments ([3 input tiles][3 hidden tiles][256] at the head of a message chain, [256] otherwise)
    const float *wup;     // Wu fragments ([hidden tile][256])
};

constexpr int GVP_MAX_CHAIN = 4;

// Training forward (gvp_train.hip): where the chained edge kernel leaves what the backward pass reads, per edge type and message GVP, in
// the trainer's layouts -- rows = edges; scalars [E][256]; vectors [E][3][channels] with 17 channels at the head GVP's input / hidden
// side ([x_diff | 16 source channels]) and 16 elsewhere; the gate before its sigmoid.
struct GvpTrainGvp {
    float *Vh, *Vu, *sh, *pre, *s, *gate, *V;
};
struct GvpTrainSlot {
    float *unit, *rbf, *vin;          // [E][3], [E][16], [E][3][17]
    GvpTrainGvp g[4];
};

struct GvpEdgeArgs {
    const int *meta;              // [9]: E[4], first tile[5]
    const int *src[4], *dst[4];
    const float *x[2];            // positions (constant in the GVP denoiser)
    const float *v[2];            // [n][16][3]
    const float *Psrc[4];         // [n_src][S]: W0[:, :S] . s_src + b0, per edge type
    const float *Pdst[4];         // [n_dst][S]: h_dst block of the first message GVP (use_dst_feats, gvp.py:323-337)
    int use_dst;                  // message input also carries the destination node's scalars / vectors
    GvpW g[4][GVP_MAX_CHAIN];     // per edge type, message chain
    int n_gvps;
    int S;
    float rbf_dmax;
    float *ms_main[4], *ms_cont[4];   // [n_dst][S], [tiles][S]
    float *mv_main[4], *mv_cont[4];   // [n_dst][48], [tiles][48]
    unsigned long long *stamps;       // [32] phase-cycle sums (diagnostics only, null in production)
    int gemm_mode;                    // 0: exact fp32 MFMA; 1: f16x2 split in the 256 x 256 products of the non-head message GVPs
    int train_skip;                   // TOOLS build only (KPD_TR_SKIP): bit 0 / 1 / 2 = leave out the pre / s / remaining stores of the training form (timing experiments)
    const GvpTrainSlot *train;        // non-null: the training form of the kernel -- node vectors v arrive as [n][3][16], the vector pieces
                                      // mv_main / mv_cont leave as [3][16], and every activation the backward pass reads is stored (device table [4])
};

// ---- backward of the message chains (training) ----------------------------------------------------------------------------------
// One message GVP in backward form (packed from the current parameters by the trainer): 16-KB chunks in consumption order --
//   generic GVP: Wg^T (scalar gradient from the gates), 16 k-slabs of to_feats_out[:, :256]^T, the |Vh| block^T (16 one-tile slabs);
//   head GVP:    Wg^T, the rbf block^T, the two |Vh| tiles^T (its source-scalar block is differentiated per node by the caller);
// and the 16 x 16 fragments of Wu^T / Wh^T (generic GVPs only: the head's vector half stays with k_gvp_vec17_bwd).
struct GvpBwdW {
    const float *chain, *wut, *wht;
};
// per edge type and message GVP: what the kernel leaves for the weight-gradient products and the vector-weight kernels (rows = edges)
struct GvpBwdGvp {
    float *dpre;          // [E][256]  dL/d pre-activation of to_feats_out
    float *dgate;         // [E][16]   dL/d gate pre-activation
    float *dVu;           // [E][3][16] dL/d (Vh Wu)
    float *dsh;           // [E][16], head: [E][17]  dL/d |Vh|
};
struct GvpBwdSlot {
    GvpBwdGvp g[4];
    float *drbf;          // [E][16] dL/d rbf code (head GVP)
};
struct GvpEdgeBwdArgs {
    const int *meta;              // as GvpEdgeArgs
    const int *dst[4], *rowptr[4];
    const float *gs[2], *gv[2];   // gradients of the aggregated messages per destination node type: [n][256], [n][3][16]
    const float *z[2];            // per-graph normaliser (mode 2)
    const int *bidx[2];
    int mode;                     // message_norm_mode: 0 constant, 1 per-edge-type mean, 2 z[graph]
    float norm;
    GvpBwdW g[4][GVP_MAX_CHAIN];
    int n_gvps;
    const GvpTrainSlot *fwd;      // [4] kept activations (device table)
    const GvpBwdSlot *out;        // [4] (device table)
};
kpd_status launch_gvp_edge_bwd(const GvpEdgeBwdArgs &a, int tile_cap, hipStream_t st);

// backward of a node-update chain (all GVPs of the generic kind) for one node type: ds / dV = gradients of its outputs (after the update
// dropout), ds_in / dv_in = gradients of its inputs (the message LayerNorm's outputs)
struct GvpNodeBwdArgs {
    int n;
    const float *ds, *dV;         // [n][256], [n][3][16]
    float *ds_in, *dv_in;
    GvpBwdW g[GVP_MAX_CHAIN];     // chain: Wg^T, 16 k-slabs of to_feats_out[:, :256]^T, the |Vh| block^T; wut, wht
    int n_gvps;
    GvpTrainGvp f[GVP_MAX_CHAIN]; // kept activations of the update GVPs
    GvpBwdGvp o[GVP_MAX_CHAIN];   // dpre, dgate, dVu, d|Vh| per GVP
};
kpd_status launch_gvp_node_bwd(const GvpNodeBwdArgs &a, hipStream_t st);

// Training form of the node update (k_gvp_node_chain<16, 0, 1>): the conv's input state is read only, everything the backward pass reads is
// kept, GVPDropout (gvp.py:119-149) acts on the aggregated messages and on the update residual with the trainers' Philox streams
// (gvp_train_core.h, dropout_scale), vectors travel as [n][3][16].
struct GvpNodeTrain {
    const float *s_in, *v_in;         // [n][256], [n][3][16]
    float *s_out, *v_out;             // the conv's output state
    float *sa, *va;                   // s + dropout(aggregated messages): input of the message LayerNorm
    float *s1, *v1;                   // its output (the residual of the update block)
    float *sb, *vb;                   // s1 + dropout(update chain): input of the update LayerNorm
    GvpTrainGvp g[GVP_MAX_CHAIN];     // the update GVPs' activations
    float rate;                       // dropout rate (0: none)
    unsigned long long seed;
    unsigned stream[4];               // Philox streams: message scalars / vectors, update scalars / vectors
    int live_v;                       // the model's vector_size (layout of the vector masks)
};

struct GvpNodeArgs {
    int n;
    float *s;                     // [n][S] in/out
    float *v;                     // [n][16][3] in/out
    float *s_tmp;                 // [n][S] scratch (residual of the update block)
    const int *bidx;
    const float *z;               // [B] per-graph normaliser (message_norm == 0) or nullptr
    float norm_const;             // otherwise
    int mean;                     // per-edge-type mean instead of sum
    int n_in;
    const int *rowptr[2];
    const float *ms_main[2], *ms_cont[2], *mv_main[2], *mv_cont[2];
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    GvpW g[GVP_MAX_CHAIN];
    int n_gvps;
    int S;
    float ln_inv_n, ln_pad;       // LayerNorm over the model's n_hidden_scalars = S - ln_pad features: 1 / that width, and the padding count
    float vn_inv_n, vn_pad;       // its vector half over vector_size = 16 - vn_pad channels
    int train;                    // 1: the training form (tr; mv_main / mv_cont as [3][16])
    GvpNodeTrain tr;
};

struct GvpNodePair {
    GvpNodeArgs nt[2];
    int tiles0;
    int gemm_mode;                // 0 exact fp32; 1 f16x2 split in the update GVPs' 256 x 256 products
    int coop_rows;                // launches of at most this many rows take the cooperative form (0: COOP_ROWS_DEFAULT, < 0: never)
};

constexpr int GVP_PROJ_SLOTS = 8;
struct GvpProjArgs {
    const float *s[GVP_PROJ_SLOTS];    // scalar state to project, per slot
    int n[GVP_PROJ_SLOTS];
    const float *wp[GVP_PROJ_SLOTS], *b[GVP_PROJ_SLOTS];   // b may be nullptr
    const float *wp_h[GVP_PROJ_SLOTS];                     // f16x2 mode: wp re-packed as 16 units of kind 0 (pack.hip)
    int gemm_mode;
    float *P[GVP_PROJ_SLOTS];
    int tiles_first[GVP_PROJ_SLOTS + 1];
    int n_slots;
    int S;
    int coop_rows;                // as GvpNodePair::coop_rows
};

struct GvpNoiseArgs {
    int n;
    const float *s, *v;
    GvpW g[GVP_MAX_CHAIN];
    int n_gvps;
    int S;
    const float *Wout, *bout;     // [F][64], [F]
    int F;
    float *eps_h, *eps_x;
    int gemm_mode;                // 0 exact fp32; 1 f16x2 split in the generic GVPs of the head
    int coop_rows;                // as GvpNodePair::coop_rows
};

kpd_status launch_gvp_embed(const float *in, int n, int fin, const float *W, const float *b, const float *ln_w,
                            const float *ln_b, const float *t, const int *bidx, int S, int S_true, float *out, hipStream_t st);
kpd_status launch_gvp_proj(const GvpProjArgs &a, hipStream_t st);
// edge messages + segmented sums (gvp_chain.hip); message chains are HostGvp with chain_pos >= 0
kpd_status launch_gvp_edge(const GvpEdgeArgs &a, int tile_cap, hipStream_t st);
kpd_status launch_gvp_node(const GvpNodePair &p, hipStream_t st);
kpd_status launch_gvp_noise(const GvpNoiseArgs &a, hipStream_t st);

// Cooperative (column-split) forms of the three node-side launches (gvp_coop.hip): 16 rows per workgroup, the four waves split the
// output columns.  launch_gvp_node / _proj / _noise take them in the exact fp32 mode whenever the launch has at most coop_rows_max()
// rows -- a function of the row count (and the engine's setting) alone, so results stay bitwise repeatable; the two forms agree to rounding (gate sums).
constexpr int COOP_ROWS_DEFAULT = 8192;
// the row limit a launch uses: its own coop_rows field when set (kpd_gvp_debug_state "coop_rows=N": tests run both forms), else the default
int coop_rows_max(int requested);
kpd_status launch_gvp_node_coop(const GvpNodePair &p, hipStream_t st);
kpd_status launch_gvp_proj_coop(const GvpProjArgs &a, hipStream_t st);
kpd_status launch_gvp_noise_coop(const GvpNoiseArgs &a, hipStream_t st);

}  // namespace kpd
