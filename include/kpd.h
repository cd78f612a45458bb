/*
 * kpd.h -- C ABI of libkpd_hip.so: the MI355X (gfx950) denoising hot path of
 * Dunni3/keypoint-diffusion behind plain pointers and sizes.
 *
 * The reference is pure Python; its "FFI" for this path is the set of third-party native
 * ops it calls every reverse-diffusion step (torch_cluster.radius_graph / knn,
 * DGL apply_edges / multi_update_all, torch.nn.Linear / LayerNorm) from
 *   models/dynamics.py:342-441        LigRecDynamics.forward  (+ LigRecEGNN, LigRecConv)
 *   models/dynamics_gvp.py:149-255    LigRecDynamicsGVP.forward (+ gvp.py GVPMultiEdgeConv)
 *   models/receptor_encoder_gvp.py:212-321  ReceptorEncoderGVP.forward
 *   models/ligand_diffuser.py:497-538 KeypointDiffusion.sample_p_zs_given_zt
 * Each entry point below replaces one of those call sites as a whole and cites it.
 *
 * Conventions
 *   - every pointer named *_dev / marked [dev] is a device (HBM) pointer, fp32 or int32,
 *     contiguous, row-major; everything else is host memory;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, no hidden
 *     synchronisation, no allocation inside forward calls (kpd_*_reserve allocates);
 *   - node arrays are flat and graph-major: complex b owns rows [ptr[b], ptr[b+1]);
 *   - every call returns KPD_OK or a negative kpd_status; kpd_last_error() gives text.
 */
#ifndef KPD_H
#define KPD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum kpd_status {
    KPD_OK = 0,
    KPD_ERR_INVALID = -1,      /* bad argument / unsupported configuration              */
    KPD_ERR_CAPACITY = -2,     /* batch larger than the reserved workspace               */
    KPD_ERR_WEIGHTS = -3,      /* unknown weight name, wrong shape, or weights missing   */
    KPD_ERR_HIP = -4,          /* a HIP runtime call failed                              */
    KPD_ERR_STATE = -5         /* call order violated (e.g. forward before commit)       */
} kpd_status;

const char *kpd_last_error(void);
int kpd_version(void);
/* 0 = the product build.  Bit 0 set = the TOOLS build (`make tools`, -DKPD_TOOLS): the only build in which the A/B, ablation and
 * LDS-padding switches of profiles/tools are read from the environment (KPD_EDGE_ABLATE, KPD_EDGE_SPLIT, KPD_*_LDS_PAD,
 * KPD_SGEMM_*, KPD_TRAIN_EPI / _WS / _VEC_FUSED, ...).  bench.py refuses to print a contract line from a library whose flags are not 0.
 * (No reference counterpart: the reference has no build variants; the check exists because a timing library must be able to attest
 * that its kernels cannot be told to skip work.) */
int kpd_build_flags(void);

/* ---------------------------------------------------------------------------------------
 * Batch of complexes (the tensors the reference keeps in a batched DGL heterograph).
 * kk edges are static during sampling (ligand_diffuser.py:201-202 translates keypoints
 * rigidly) and are passed dst-sorted with their CSR row pointer.
 * ------------------------------------------------------------------------------------- */
typedef struct kpd_batch {
    int32_t B;                 /* complexes                                              */
    int32_t n_lig, n_kp;       /* total ligand atoms / keypoints                         */
    int32_t max_lig, max_kp;   /* largest per-complex counts (host-known)                */
    const int32_t *lig_ptr;    /* [dev] [B+1]                                            */
    const int32_t *kp_ptr;     /* [dev] [B+1]                                            */
    const float *lig_x;        /* [dev] [n_lig,3]        g.nodes['lig'].data['x_0']      */
    const float *lig_h;        /* [dev] [n_lig,atom_nf]  g.nodes['lig'].data['h_0']      */
    const float *kp_x;         /* [dev] [n_kp,3]                                         */
    const float *kp_h;         /* [dev] [n_kp,rec_nf]                                    */
    const float *kp_v;         /* [dev] [n_kp,V,3] (GVP only, else NULL)                 */
    int32_t n_kk;              /* kk edges                                               */
    const int32_t *kk_src;     /* [dev] [n_kk] sorted by (dst, src)                      */
    const int32_t *kk_dst;     /* [dev] [n_kk]                                           */
    const int32_t *kk_rowptr;  /* [dev] [n_kp+1]                                         */
} kpd_batch;

/* ---------------------------------------------------------------------------------------
 * Per-step ligand graph build.  Replaces add_lig_edges (models/dynamics.py:387-420,
 * models/dynamics_gvp.py:201-234): torch_cluster.radius_graph(lig, r=ll) and
 * torch_cluster.knn(x=lig, y=kp, k) plus the DGL add_edges / batch bookkeeping.
 * Output: dst-sorted COO + CSR row pointers for ll (dst lig), kl (src kp -> dst lig) and
 * lk (src lig -> dst kp), in caller buffers of the stated capacity.
 * ------------------------------------------------------------------------------------- */
typedef struct kpd_lig_graph {
    int32_t cap_ll, cap_kl;    /* capacities (edges) of the arrays below                 */
    int32_t *ll_src, *ll_dst, *ll_rowptr;   /* [dev] [cap_ll],[cap_ll],[n_lig+1]         */
    int32_t *kl_src, *kl_dst, *kl_rowptr;   /* [dev] [cap_kl],[cap_kl],[n_lig+1]         */
    int32_t *lk_src, *lk_dst, *lk_rowptr;   /* [dev] [cap_kl],[cap_kl],[n_kp+1]          */
    int32_t *ll_per_graph;     /* [dev] [B] ll edges per complex                          */
    int32_t *counts;           /* [dev] [2]: {E_ll, E_kl}                                 */
} kpd_lig_graph;

/* ll: radius graph of radius ll_cutoff (ll_k == 0, at most 200 neighbours) or kNN graph (ll_k in 1..16);
 * kl / lk: for every keypoint its kl_k nearest ligand atoms (kl_k in 1..16) or all ligand atoms within kl_cutoff
 * (kl_k == 0, at most 100).  Capacities: cap_ll >= n_lig * min(max_lig - 1, ll_k or 200),
 * cap_kl >= n_kp * (kl_k or min(max_lig, 100)). */
kpd_status kpd_build_lig_graph(const kpd_batch *batch, float ll_cutoff, int32_t ll_k, float kl_cutoff,
                               int32_t kl_k, const kpd_lig_graph *out, void *stream);

/* ---------------------------------------------------------------------------------------
 * EGNN denoiser.  Replaces LigRecDynamics.forward (models/dynamics.py:342-385) including
 * lig/rec encoders, edge build, the LigRecEGNN stack (:266-294, LigRecConv :89-217) and
 * the decoder.  Constructor fields mirror LigRecDynamics.__init__ (:300-340).
 * ------------------------------------------------------------------------------------- */
typedef struct kpd_egnn_config {
    int32_t atom_nf, rec_nf;
    int32_t n_layers, hidden_nf;       /* hidden_nf 1 .. 256 (the kernels are 256 + 1 wide; narrower models run zero padded: the reference's
                                          default ctor is 255); > 256 is refused (INTEGRATION.md section 1 lists the limits) */
    int32_t use_tanh, norm, update_kp_feat;
    float message_norm;                /* 0 => per-graph average in-degree + 1           */
    int32_t ll_k, kl_k;                /* 0 = radius graph (the cutoffs below), else kNN, <= 16 */
    float ll_cutoff, kl_cutoff;
    float coords_range;                /* 10 in the reference (dynamics.py:15)           */
} kpd_egnn_config;

typedef struct kpd_egnn kpd_egnn;

kpd_status kpd_egnn_create(const kpd_egnn_config *cfg, kpd_egnn **out);
void kpd_egnn_destroy(kpd_egnn *m);
/* One state-dict tensor of the reference `dynamics` module, by its reference name
 * (e.g. "egnn.conv_layers.3.edge_mlp.kl.2.weight"); repacked on device for the kernels. */
kpd_status kpd_egnn_load_weight(kpd_egnn *m, const char *name, const float *w_dev,
                                const int64_t *shape, int32_t ndim, void *stream);
kpd_status kpd_egnn_commit(kpd_egnn *m);          /* checks every tensor was loaded        */
/* Allocate workspace for batches up to these sizes (grow-only; not stream-ordered). */
kpd_status kpd_egnn_reserve(kpd_egnn *m, int32_t max_B, int32_t max_n_lig, int32_t max_n_kp,
                            int32_t max_n_kk, int32_t max_lig_per_graph, int32_t max_kp_per_graph);
/* eps_h [n_lig, atom_nf], eps_x [n_lig, 3];  t [B] in (0,1]. */
kpd_status kpd_egnn_forward(kpd_egnn *m, const kpd_batch *batch, const float *t_dev,
                            float *eps_h_dev, float *eps_x_dev, void *stream);
/* Debug/test taps and switches.  Taps copy engine state to out_dev: "h_lig" / "h_kp" (node state, row stride 264), "x_lig" /
 * "x_kp", "z_lig" / "z_kp", "xnm<et>" / "xnc<et>" / "hnm<et>" / "hnc<et>" (segment-sum pieces of edge type et as the last layer
 * left them), "stamps".  Switches (n_floats = 0, out_dev ignored but non-null): "layers=N" (run only the first N layers),
 * "prune=0|1", "stamps=1", and
 *   "gemm=f32"   exact fp32 MFMA in every GEMM -- the default and the contract path;
 *   "gemm=f16x2" EXPERIMENTAL, opt-in, never the default and never part of the benchmark's `value`: every fp32 product of the
 *                edge / projection / node-update GEMMs as three f16 MFMA products of hi / lo operand planes with fp32
 *                accumulation (the parity suite holds at the same 1e-4 in this mode, ~2x the step rate).  Status (round 4,
 *                DESIGN.md "f16x2 mode"): a build VARIANT of its edge kernel (batched distance read) showed a first-launch
 *                deviation of one LDS row in rounds 2 - 3; the shipped per-row form has never shown it in any detector, and a
 *                standalone kernel with the suspected ingredients (profiles/tools/f16_lds_row_probe.hip: 20 fresh processes,
 *                80 launches clean) does not reproduce it; in round 4 the variant build itself -- of today's sources and of the
 *                round-3 sources -- is clean in 24 fresh processes on 10 GPUs (profiles/r04_f16_variant_resample.txt), so it is
 *                neither explained nor reproducible any more, nor shown to be a hardware erratum.
 *                Until it is, the mode is frozen as experimental: use it for throughput experiments, not for results you keep.
 *                The environment variable KPD_GEMM=f16x2 selects it at kpd_egnn_create time. */
kpd_status kpd_egnn_debug_state(kpd_egnn *m, const char *what, float *out_dev, int64_t n_floats,
                                void *stream);
/* HIP-event timing of the dominant kernel (the fused edge kernel), recorded on the caller's
 * stream around each of its launches while enabled (up to 8192 launches per enable). */
kpd_status kpd_egnn_profile(kpd_egnn *m, int32_t enable);
kpd_status kpd_egnn_profile_read(kpd_egnn *m, double *total_ms, int32_t *launches);
/* Launch geometry of the last forward: {E_ll, E_kl, E_lk, E_kk, edge tiles of a full layer, edge tiles and edges of the
 * final layer, and in out[7] the GEMM mode in effect (0 exact fp32, 1 f16x2)} -- the final layer runs ll + kl only, since
 * LigRecEGNN.forward returns (h_lig, x_lig) alone (models/dynamics.py:288-294).  Before the first forward only out[7] is set. */
kpd_status kpd_egnn_last_counts(kpd_egnn *m, int32_t out[8], void *stream);

/* ---------------------------------------------------------------------------------------
 * EGNN denoiser, training path: forward that keeps the layer states and the backward pass of
 * LigRecDynamics.forward (models/dynamics.py:342-385), i.e. what torch autograd derives for the loss of
 * KeypointDiffusion.forward (models/ligand_diffuser.py:89-175) in train.py:423-524.  Parameters are bound once by their
 * reference state-dict names and read in place in the reference [out, in] layout (no repacking: they change every
 * optimizer step); each parameter's gradient is ACCUMULATED (+=) into the bound buffer of the same shape, the way
 * autograd accumulates into .grad.  grad may be NULL for a frozen parameter.
 *   forward : eps_h [n_lig, atom_nf], eps_x [n_lig, 3]; the batch tensors and `t` must stay alive until backward.
 *             One host read-back of the edge counts per call (they size the GEMMs).
 *   backward: d_eps_h / d_eps_x = dL/d(eps); any of d_lig_h [n_lig, atom_nf], d_lig_x [n_lig, 3], d_kp_h [n_kp, rec_nf],
 *             d_kp_x [n_kp, 3] may be NULL (written, not accumulated, when given).  Consumes the forward.
 * Memory: reserve() tries to keep the edge activations of every layer (13.5 GB at B = 64 x (300-atom pocket, 25-atom ligand)); if that
 * allocation fails it keeps one layer's worth and recomputes layer by layer in backward (same results, bit for bit).
 * Environment: KPD_TRAIN_STORE=0 (read once per process) selects the recompute mode.  The A/B switches of profiles/tools
 * (KPD_TRAIN_FUSED_*, KPD_SGEMM_*, KPD_TRAIN_WS*, KPD_TRAIN_VEC_FUSED) exist only in the TOOLS build of the library (kpd_build_flags).
 * profile / profile_read: HIP-event time of the two per-layer edge kernels of a training step on their launch stream, as
 * kpd_egnn_profile does for the inference kernel: index 0 = forward (k_egnn_edge_train), 1 = backward (k_egnn_edge_bwd); `edges` =
 * edges those launches processed (for a FLOP count).  No reference counterpart (bench.py's roofline of the training lines).
 * ------------------------------------------------------------------------------------- */
typedef struct kpd_egnn_trainer kpd_egnn_trainer;
kpd_status kpd_egnn_trainer_create(const kpd_egnn_config *cfg, kpd_egnn_trainer **out);
void kpd_egnn_trainer_destroy(kpd_egnn_trainer *t);
kpd_status kpd_egnn_trainer_bind(kpd_egnn_trainer *t, const char *name, const float *weight_dev, float *grad_dev,
                                 const int64_t *shape, int32_t ndim);
kpd_status kpd_egnn_trainer_reserve(kpd_egnn_trainer *t, int32_t max_B, int32_t max_n_lig, int32_t max_n_kp,
                                    int32_t max_n_kk, int32_t max_lig_per_graph, int32_t max_kp_per_graph);
kpd_status kpd_egnn_trainer_forward(kpd_egnn_trainer *t, const kpd_batch *batch, const float *t_dev, float *eps_h_dev,
                                    float *eps_x_dev, void *stream);
kpd_status kpd_egnn_trainer_backward(kpd_egnn_trainer *t, const float *d_eps_h, const float *d_eps_x, float *d_lig_h,
                                     float *d_lig_x, float *d_kp_h, float *d_kp_x, void *stream);
kpd_status kpd_egnn_trainer_profile(kpd_egnn_trainer *t, int32_t enable);
kpd_status kpd_egnn_trainer_profile_read(kpd_egnn_trainer *t, double total_ms[2], int32_t launches[2], double edges[2]);

/* ---------------------------------------------------------------------------------------
 * GVP denoiser.  Replaces LigRecDynamicsGVP.forward (models/dynamics_gvp.py:149-199): encoders
 * (:124-134), edge build (:201-234), the GVPMultiEdgeConv stack (models/gvp.py:343-551; GVP :43-116,
 * GVPLayerNorm :152-166) and the NoisePredictionBlock (:10-44).  Fields mirror
 * LigRecDynamicsGVP.__init__ (:106-147).  kpd_batch.kp_v carries the keypoint vector features v_0.
 * ------------------------------------------------------------------------------------- */
typedef struct kpd_gvp_config {
    int32_t n_lig_scalars, n_kp_scalars;
    int32_t vector_size;               /* 1 .. 16 (kernels are 16 channels wide; fewer are zero padded)           */
    int32_t n_convs, n_hidden_scalars; /* n_hidden_scalars 1 .. 256 (kernels are 128 / 256 wide, likewise); the   */
                                       /* training engine kpd_gvp_trainer_* takes the same ranges (narrower models */
                                       /* through zero-padded wide parameter copies)                               */
    int32_t update_kp;
    int32_t message_norm_mode;         /* 0: constant message_norm, 1: 'mean', 2: message_norm == 0
                                          (per-graph average in-degree + 1, gvp.py:504-507)  */
    float message_norm;
    int32_t ll_k, kl_k;                /* ll_k must be 0; kl_k in 1..16                    */
    float ll_cutoff, kl_cutoff;
    int32_t n_message_gvps, n_update_gvps, n_noise_gvps;   /* each in 1..4                 */
} kpd_gvp_config;

typedef struct kpd_gvp kpd_gvp;

kpd_status kpd_gvp_create(const kpd_gvp_config *cfg, kpd_gvp **out);
void kpd_gvp_destroy(kpd_gvp *m);
/* Reference state-dict names of the `dynamics` module, e.g.
 * "noise_predictor.conv_layers.2.edge_message_fns.kp_kl_lig.0.to_feats_out.0.weight". */
kpd_status kpd_gvp_load_weight(kpd_gvp *m, const char *name, const float *w_dev,
                               const int64_t *shape, int32_t ndim, void *stream);
kpd_status kpd_gvp_commit(kpd_gvp *m);
kpd_status kpd_gvp_reserve(kpd_gvp *m, int32_t max_B, int32_t max_n_lig, int32_t max_n_kp,
                           int32_t max_n_kk, int32_t max_lig_per_graph, int32_t max_kp_per_graph);
kpd_status kpd_gvp_forward(kpd_gvp *m, const kpd_batch *batch, const float *t_dev,
                           float *eps_h_dev, float *eps_x_dev, void *stream);
/* Debug/test taps: "convs=<n>" limits the conv stack; "s_lig", "s_kp", "v_lig", "v_kp" copy state; "gemm=f32" | "gemm=f16x2" as for
 * kpd_egnn_debug_state (the 256 x 256 products of the message / update chains; KPD_GEMM=f16x2 at create time; 256 scalars only). */
kpd_status kpd_gvp_debug_state(kpd_gvp *m, const char *what, float *out_dev, int64_t n_floats,
                               void *stream);
/* HIP-event timing of the dominant kernel (k_gvp_chain: the message chain of GVPMultiEdgeConv.message,
 * models/gvp.py:459-497, 545-549) around each of its launches while enabled, and the launch geometry of the last
 * forward {E_ll, E_kl, E_lk, E_kk, tiles of a four-edge-type conv, tiles of the final conv, edges of the final conv}
 * (the final conv runs ll + kl only, models/dynamics_gvp.py:67-72), out[7] = the GEMM mode in effect (0 exact fp32, 1 f16x2). */
kpd_status kpd_gvp_profile(kpd_gvp *m, int32_t enable);
kpd_status kpd_gvp_profile_read(kpd_gvp *m, double *total_ms, int32_t *launches);
kpd_status kpd_gvp_last_counts(kpd_gvp *m, int32_t out[8], void *stream);

/* Training path of the GVP denoiser: same contract as kpd_egnn_trainer_* above for
 * LigRecDynamicsGVP.forward (models/dynamics_gvp.py:149-199).  Gradients flow to every parameter and to the scalar and
 * vector input features: d_lig_h [n_lig, n_lig_scalars], d_kp_h [n_kp, n_kp_scalars], d_kp_v [n_kp, 16, 3] (each may be
 * NULL); positions receive no gradient (they enter through the unit edge vector and the rbf code only and are data in
 * every training configuration served). */
typedef struct kpd_gvp_trainer kpd_gvp_trainer;
kpd_status kpd_gvp_trainer_create(const kpd_gvp_config *cfg, kpd_gvp_trainer **out);
void kpd_gvp_trainer_destroy(kpd_gvp_trainer *t);
kpd_status kpd_gvp_trainer_bind(kpd_gvp_trainer *t, const char *name, const float *weight_dev, float *grad_dev,
                                const int64_t *shape, int32_t ndim);
/* GVPDropout of training mode (gvp.py:119-149): rate in [0, 1) and the seed of this step's masks; call before forward
 * (0 = identity, the default).  Feature dropout is per element, vector dropout per channel, kept entries scale by
 * 1 / (1 - rate); the masks are Philox streams keyed by (seed; conv, node type, position, kind) and are regenerated in
 * backward.  kpd_dropout_mask writes one such stream ({0, 1 / (1 - rate)} per entry; node_type 0 = lig, 1 = kp;
 * position 0 = aggregated messages, 1 = update residual; kind 0 = scalars [n, S] row-major, 1 = vector channels
 * [n, 16]) so that tests can replay a step on the oracle. */
kpd_status kpd_gvp_trainer_set_dropout(kpd_gvp_trainer *t, float rate, uint64_t seed);
kpd_status kpd_dropout_mask(uint64_t seed, int32_t conv, int32_t node_type, int32_t position, int32_t kind, int64_t n,
                            float rate, float *out_dev, void *stream);
kpd_status kpd_gvp_trainer_reserve(kpd_gvp_trainer *t, int32_t max_B, int32_t max_n_lig, int32_t max_n_kp,
                                   int32_t max_n_kk, int32_t max_lig_per_graph, int32_t max_kp_per_graph);
kpd_status kpd_gvp_trainer_forward(kpd_gvp_trainer *t, const kpd_batch *batch, const float *t_dev, float *eps_h_dev,
                                   float *eps_x_dev, void *stream);
/* d_lig_x [n_lig, 3] / d_kp_x [n_kp, 3] (either may be null): gradients with respect to the positions, which enter through the
 * unit edge vector and the rbf code of every edge (models/gvp.py:472-480); the edge lists themselves are not differentiable. */
/* E_ll, E_kl, E_lk, E_kk of the last forward (host values the forward already read back: no synchronisation).  No reference counterpart
 * (bench.py's FLOP count of the training lines). */
kpd_status kpd_gvp_trainer_last_counts(kpd_gvp_trainer *t, int32_t out[4]);
/* Which form the edge messages of the convs run in with the current reservation: 1 = the register-chained kernels (one forward, one
 * backward and two batched weight-gradient launches per conv: n_hidden_scalars = 256 and the kept-activation memory was granted),
 * 0 = one GVP at a time through the GEMM kernels (any width; recomputation when memory is short).  Same gradients to rounding.  No
 * reference counterpart (tests assert that the path they mean to cover is the one that ran). */
kpd_status kpd_gvp_trainer_message_path(kpd_gvp_trainer *t, int32_t *path);
kpd_status kpd_gvp_trainer_backward(kpd_gvp_trainer *t, const float *d_eps_h, const float *d_eps_x, float *d_lig_h,
                                    float *d_kp_h, float *d_kp_v, float *d_lig_x, float *d_kp_x, void *stream);

/* ---------------------------------------------------------------------------------------
 * GVP keypoint receptor encoder (once per pocket).  Replaces ReceptorEncoderGVP.forward
 * (models/receptor_encoder_gvp.py:212-294): scalar embedding, rec-rec GVPEdgeConv stack
 * (models/gvp.py:170-341), KeypointInitializer (:15-93), kNN rec->kp edges (update_rk_edges
 * :297-321), rec-kp GVPEdgeConv stack, keypoint radius graph.  Fields mirror
 * ReceptorEncoderGVP.__init__ (:99-114) + graph_cutoffs.
 * ------------------------------------------------------------------------------------- */
typedef struct kpd_recenc_config {
    int32_t in_scalar_size, out_scalar_size;   /* out_scalar_size in {128, 256}            */
    int32_t vector_size;                       /* 1 .. 16 (16-channel kernels, fewer are zero padded); kp_v of kpd_rec_out is [n_kp][vector_size][3] */
    int32_t n_rr_convs, n_rk_convs, n_message_gvps, n_update_gvps;
    int32_t message_norm_mode;                 /* 0 constant, 1 'mean', 2 message_norm == 0 */
    float message_norm;
    int32_t k_closest;                         /* kNN rec->kp, 1..16, or 0 with kp_rad > 0 */
    int32_t n_keypoints;
    float rr_cutoff, rk_cutoff, kk_cutoff;     /* graph_cutoffs['rr'|'rk'|'kk'] (rbf D_max, kk radius) */
    float kp_rad;                              /* > 0 (with k_closest == 0): radius rec->kp graph, at most 10 receptor atoms per
                                                * keypoint in index order (receptor_encoder_gvp.py:304-306) */
} kpd_recenc_config;

typedef struct kpd_rec_batch {
    int32_t B, n_rec, max_rec;
    const int32_t *rec_ptr;     /* [dev] [B+1]                                             */
    const float *rec_x;         /* [dev] [n_rec,3]                                         */
    const float *rec_h;         /* [dev] [n_rec,in_scalar_size]                            */
    int32_t n_rr;
    const int32_t *rr_src;      /* [dev] [n_rr] sorted by (dst, src)                       */
    const int32_t *rr_dst;
    const int32_t *rr_rowptr;   /* [dev] [n_rec+1]                                         */
} kpd_rec_batch;

typedef struct kpd_rec_out {
    float *kp_x;                /* [dev] [B*K,3]    keypoint positions                     */
    float *kp_h;                /* [dev] [B*K,S]    keypoint scalars                       */
    float *kp_v;                /* [dev] [B*K,16,3] keypoint vectors                       */
    int32_t *rk_src, *rk_dst;   /* [dev] [B*K*k_closest] rec->kp edges, kp-major           */
    int32_t cap_kk;             /* capacity of kk_src / kk_dst (>= B*K*min(K-1,100))       */
    int32_t *kk_src, *kk_dst;   /* [dev] kp-kp radius graph, dst-sorted                    */
    int32_t *kk_per_graph;      /* [dev] [B]                                               */
    int32_t *counts;            /* [dev] [2]: {E_kk, E_rk}                                 */
} kpd_rec_out;

typedef struct kpd_recenc kpd_recenc;

kpd_status kpd_recenc_create(const kpd_recenc_config *cfg, kpd_recenc **out);
void kpd_recenc_destroy(kpd_recenc *m);
/* Reference state-dict names of the `rec_encoder` module, e.g.
 * "rk_conv_layers.1.edge_message.0.to_feats_out.0.weight". */
kpd_status kpd_recenc_load_weight(kpd_recenc *m, const char *name, const float *w_dev,
                                  const int64_t *shape, int32_t ndim, void *stream);
kpd_status kpd_recenc_commit(kpd_recenc *m);
kpd_status kpd_recenc_reserve(kpd_recenc *m, int32_t max_B, int32_t max_n_rec, int32_t max_n_rr,
                              int32_t max_rec_per_graph);
kpd_status kpd_recenc_forward(kpd_recenc *m, const kpd_rec_batch *batch, const kpd_rec_out *out,
                              void *stream);

/* Training engine of the same encoder (SURVEY.md 8(f) item 2 for row a8): forward with saved node states, backward with
 * respect to every parameter given the gradients of the three outputs (keypoint positions, scalars, vectors -- what the GVP
 * denoiser's backward pass and the optimal-transport encoder loss, losses/rec_encoder_loss.py:49-82, hand back).  Parameters are
 * bound by reference state-dict name to live storage (read in place, gradients accumulated in place: bind zero-filled buffers);
 * dropout as in kpd_gvp_trainer_set_dropout (GVPDropout on the aggregated messages and on the update residual of every
 * GVPEdgeConv, models/gvp.py:316, 327).  Forward fills the same kpd_rec_out as kpd_recenc_forward; any of the three gradient
 * pointers of backward may be null (= zero). */
typedef struct kpd_recenc_trainer kpd_recenc_trainer;
kpd_status kpd_recenc_trainer_create(const kpd_recenc_config *cfg, kpd_recenc_trainer **out);
void kpd_recenc_trainer_destroy(kpd_recenc_trainer *t);
kpd_status kpd_recenc_trainer_bind(kpd_recenc_trainer *t, const char *name, const float *weight_dev, float *grad_dev,
                                   const int64_t *shape, int32_t ndim);
kpd_status kpd_recenc_trainer_set_dropout(kpd_recenc_trainer *t, float rate, uint64_t seed);
kpd_status kpd_recenc_trainer_reserve(kpd_recenc_trainer *t, int32_t max_B, int32_t max_n_rec, int32_t max_n_rr,
                                      int32_t max_rec_per_graph);
kpd_status kpd_recenc_trainer_forward(kpd_recenc_trainer *t, const kpd_rec_batch *batch, const kpd_rec_out *out, void *stream);
kpd_status kpd_recenc_trainer_backward(kpd_recenc_trainer *t, const float *d_kp_x, const float *d_kp_h, const float *d_kp_v,
                                       void *stream);

/* ---------------------------------------------------------------------------------------
 * EGNN keypoint receptor encoder (once per pocket; the encoder of the egnn_20kp / egnn_40kp models).  Replaces
 * ReceptorEncoder.forward (models/receptor_encoder.py:483-555): the ReceptorConv stack on the rr graph (:14-154),
 * keypoint_embedding of the mean receptor feature (:526-530), RecKeyConv (:182-297, k_closest features) and the
 * keypoint radius graph (:541).  Fields mirror ReceptorEncoder.__init__ (:383-400) + graph_cutoffs['kk'].
 * Batch and outputs reuse kpd_rec_batch / kpd_rec_out (rec_h width = in_n_node_feat, kp_h width =
 * out_n_node_feat, kp_v unused); rr_same_res [n_rr] is the rr `same_res` column as floats, in the order of the
 * sorted rr edges (null without use_sameres_feat); rec_h_out [n_rec,out] / rec_x_out [n_rec,3] (optional) receive
 * the learned receptor features / positions the reference stores as rec 'h' / 'x' (:516-517).
 * ------------------------------------------------------------------------------------- */
typedef struct kpd_recegnn_config {
    int32_t n_convs, n_keypoints;
    int32_t in_n_node_feat, hidden_n_node_feat, out_n_node_feat;   /* each in 1..256                        */
    int32_t use_sameres_feat, use_tanh, norm, fix_pos;
    float coords_range;
    float message_norm;          /* 0: z = rr edges / receptor nodes per graph (no +1, :505-509)          */
    int32_t k_closest;           /* kNN rec->kp features, 1..16, or 0 with kp_rad > 0                     */
    float kk_cutoff;             /* graph_cutoffs['kk']                                                   */
    float kp_rad;                /* > 0 (with k_closest == 0): keypoint features = sum of the receptor features within kp_rad (at most
                                  * 100 atoms, index order) / (rk edges per keypoint of the complex + 1) (receptor_encoder.py:238-262) */
} kpd_recegnn_config;

typedef struct kpd_recegnn kpd_recegnn;

kpd_status kpd_recegnn_create(const kpd_recegnn_config *cfg, kpd_recegnn **out);
void kpd_recegnn_destroy(kpd_recegnn *m);
/* Reference state-dict names of the `rec_encoder` module, e.g. "rec_convs.2.edge_mlp.0.weight",
 * "rec_kp_conv.kp_feature_mlp.0.bias" ("rec_kp_conv.fc_dst.weight" is accepted and ignored, as upstream never applies it). */
kpd_status kpd_recegnn_load_weight(kpd_recegnn *m, const char *name, const float *w_dev,
                                   const int64_t *shape, int32_t ndim, void *stream);
kpd_status kpd_recegnn_commit(kpd_recegnn *m);
kpd_status kpd_recegnn_reserve(kpd_recegnn *m, int32_t max_B, int32_t max_n_rec, int32_t max_n_rr,
                               int32_t max_rec_per_graph);
kpd_status kpd_recegnn_forward(kpd_recegnn *m, const kpd_rec_batch *batch, const float *rr_same_res,
                               const kpd_rec_out *out, float *rec_h_out, float *rec_x_out, void *stream);

/* Training engine of the same encoder (SURVEY.md 8(f) item 2 for row f1): forward with saved layer states, backward with respect
 * to every parameter given the gradients of the keypoint positions and keypoint features (either may be null = zero).  The
 * k_closest keypoint features only (every shipped config; create refuses kp_rad > 0).  Parameters are bound as in
 * kpd_recenc_trainer_bind.  Forward fills the same outputs as kpd_recegnn_forward. */
typedef struct kpd_recegnn_trainer kpd_recegnn_trainer;
kpd_status kpd_recegnn_trainer_create(const kpd_recegnn_config *cfg, kpd_recegnn_trainer **out);
void kpd_recegnn_trainer_destroy(kpd_recegnn_trainer *t);
kpd_status kpd_recegnn_trainer_bind(kpd_recegnn_trainer *t, const char *name, const float *weight_dev, float *grad_dev,
                                    const int64_t *shape, int32_t ndim);
kpd_status kpd_recegnn_trainer_reserve(kpd_recegnn_trainer *t, int32_t max_B, int32_t max_n_rec, int32_t max_n_rr,
                                       int32_t max_rec_per_graph);
kpd_status kpd_recegnn_trainer_forward(kpd_recegnn_trainer *t, const kpd_rec_batch *batch, const float *rr_same_res,
                                       const kpd_rec_out *out, float *rec_h_out, float *rec_x_out, void *stream);
kpd_status kpd_recegnn_trainer_backward(kpd_recegnn_trainer *t, const float *d_kp_x, const float *d_kp_h, void *stream);

/* ---------------------------------------------------------------------------------------
 * Exact optimal-transport plans between small uniform point clouds, on the HOST (no device work): what the reference gets from
 * POT's `ot.emd` in losses/rec_encoder_loss.py:11-18 (keypoints vs receptor atoms / interface points, once per complex and
 * training batch).  n_problems independent problems; problem p: cost matrix at cost + offsets[p] (row-major [n[p], m[p]],
 * doubles), masses 1 / n[p] and 1 / m[p]; the optimal plan is written to plan + offsets[p].  Up to n_threads host threads.
 * ------------------------------------------------------------------------------------- */
kpd_status kpd_ot_emd_uniform(int32_t n_problems, const int32_t *n, const int32_t *m, const int64_t *offsets,
                              const double *cost_host, double *plan_host, int32_t n_threads);

/* ---------------------------------------------------------------------------------------
 * The fp32 GEMM the training engines run their dense products on (csrc/sgemm.hip, v_mfma_f32_32x32x2_f32; no vendor BLAS in the
 * library): row-major C[M,N] = alpha op(A) op(B) + beta C on device pointers, any sizes, leading dimensions and alignments.  What the
 * reference gets from torch.nn.Linear / autograd inside models/dynamics.py:37-79, models/gvp.py:166-222 during train.py; exported so
 * that its parity against a plain fp32 product can be tested on its own.  `workspace` (device floats, may be NULL): scratch for the
 * partial products of a K-dominated shape (a weight gradient, K = edge count), which is then cut along K over the grid and summed in a
 * fixed order -- never with atomics.  `colsum` (may be NULL; A^T B products only): colsum[m] += sum_k A[k][m] in the same pass over A
 * -- the bias gradient of the Linear whose weight gradient the product is.
 * ------------------------------------------------------------------------------------- */
kpd_status kpd_sgemm(int32_t trans_a, int32_t trans_b, int32_t M, int32_t N, int32_t K, float alpha, const float *A, int32_t lda,
                     const float *B, int32_t ldb, float beta, float *C, int32_t ldc, float *colsum, float *workspace,
                     int64_t workspace_floats, void *stream);

/* The batched weight gradients of the training engines (csrc/sgemm.hip: k_wgrad_tnx, k_sgemm_tn256_batch), exported like kpd_sgemm so that
 * their parity against plain fp32 products can be tested on their own.  Up to eight products of one kind per call, every output ACCUMULATED
 * (+=), split along K over the CUs in proportion to the products' K and summed in a fixed order (no atomics).  A, B: [K, >= 256] device
 * arrays with 16-byte aligned rows (lda, ldb multiples of 4).
 *   kind 0 (GVP message / update chains: autograd of models/gvp.py:100-111 with respect to to_feats_out and scalar_to_vector_gates):
 *       C [256, 256] += A^T B;  Cx1 [256, nb2] += A^T B2 (nb2 <= 31);  colsum [256] += column sums of A;
 *       Cx2 [na2, 256] += A2^T B (na2 <= 32);  colsum2 [na2] += column sums of A2;
 *       C == NULL for all products of the call: no 256 x 256 block, and Cx3 [256, nb3] += A^T B3 (nb3 <= 32) as a second narrow block.
 *       Unused narrow operands: widths 0 and NULL pointers.
 *   kind 1 (EGNN second Linears, models/dynamics.py:37-79): C [257, 257] += A[:, :257]^T B[:, :257], colsum [257] += column sums of A (or
 *       NULL); the narrow fields are ignored. */
typedef struct {
    const float *A, *B;
    int32_t lda, ldb, K;
    float *C;
    int32_t ldc;
    const float *B2;
    int32_t ldb2, nb2;
    float *Cx1;
    int32_t ldx1;
    float *colsum;
    const float *A2;
    int32_t lda2, na2;
    float *Cx2;
    int32_t ldx2;
    float *colsum2;
    const float *B3;
    int32_t ldb3, nb3;
    float *Cx3;
    int32_t ldx3;
} kpd_wgrad_item;
kpd_status kpd_wgrad_batch(int32_t kind, int32_t n, const kpd_wgrad_item *items, float *workspace, int64_t workspace_floats, void *stream);

/* ---------------------------------------------------------------------------------------
 * The optimizer step of the training loop.  Replaces torch.nn.utils.clip_grad_value_(model.parameters(), clip_value) +
 * torch.optim.Adam(...).step() of train.py:430-433, 541-543 (torch/optim/adam.py, default flags: no amsgrad, no maximize) for every
 * parameter tensor in ONE launch; same arithmetic element for element (csrc/optim.hip).  `params_dev`: DEVICE array of n_params entries
 * -- device pointers to a parameter, its gradient, exp_avg, exp_avg_sq (fp32, contiguous) and the element count -- that the caller
 * keeps current (gradient tensors may move from step to step); max_numel = the largest count.  mode 0: Adam step number `step` (>= 1),
 * with the gradients clamped to +- clip_value first when clip_value > 0 (written back, as clip_grad_value_ does); mode 1: the clamp
 * alone.  No host synchronisation.
 * ------------------------------------------------------------------------------------- */
typedef struct {
    float *p;
    const float *g;
    float *m, *v;
    int64_t n;
} kpd_adam_param;
kpd_status kpd_adam_step(const kpd_adam_param *params_dev, int32_t n_params, int64_t max_numel, int32_t mode, double lr, double beta1, double beta2,
                         double eps, double weight_decay, int64_t step, double clip_value, void *stream);

/* ---------------------------------------------------------------------------------------
 * Reverse-diffusion update around the denoiser.  Replaces the elementwise part of
 * KeypointDiffusion.sample_p_zs_given_zt (models/ligand_diffuser.py:515-536):
 *   z_s = z_t / alpha_ts - var_terms * eps + sigma * noise, then ligand-COM removal from
 *   ligand and keypoints (remove_com :185-203).  coef [B,3] = {1/alpha_ts, var_terms, sigma}.
 * Updates lig_x, lig_h, kp_x in place.
 * ------------------------------------------------------------------------------------- */
kpd_status kpd_sample_update(int32_t B, const int32_t *lig_ptr, const int32_t *kp_ptr,
                             int32_t atom_nf, float *lig_x, float *lig_h, float *kp_x,
                             const float *eps_x, const float *eps_h,
                             const float *noise_x, const float *noise_h,
                             const float *coef, int32_t max_lig, void *stream);

/* Coefficients of that update for every complex of the batch, from the noise-schedule table.
 * Replaces PredefinedNoiseSchedule.forward (models/ligand_diffuser.py:654-690) and the gamma / sigma / alpha
 * arithmetic of sample_p_zs_given_zt (:505-526, sigma_and_alpha_t_given_s :552-566):
 *   gamma [n_gamma] device table (n_gamma = timesteps + 1), s, t [B] device, coef [B,3] device out =
 *   {alpha_t|s, sigma^2_t|s / alpha_t|s / sigma_t, sigma_t|s sigma_s / sigma_t}. */
kpd_status kpd_step_coefficients(const float *gamma, int32_t n_gamma, const float *s, const float *t,
                                 int32_t B, float *coef, void *stream);

/* Sharding-invariant N(0,1) noise for the ligand rows of a batch (opt-in replacement of the global torch.randn draws of
 * ligand_diffuser.py:367, 530-531): out [n_nodes, width] with rows of complex b =
 * [node_ptr[b], node_ptr[b+1]).  Philox4x32-10 keyed by (seed, complex_id[b]), counter (element, step, tag): the values do
 * not depend on batch composition or rank.  complex_id [B] device int64 (global index of the complex in the job). */
kpd_status kpd_complex_noise(int32_t B, const int32_t *node_ptr, int32_t width, const int64_t *complex_id,
                             uint64_t seed, int32_t step, int32_t tag, float *out, void *stream);

/* Input side: the receptor part of build_initial_complex_graph
 * (data_processing/pdbbind_processing.py:221-274) for a whole batch of pockets taken from the flat dataset arrays
 * (data_processing/crossdocked/dataset.py:62-76, 135-145): rr = torch_cluster.radius_graph(rec, r = cutoffs['rr'],
 * max_num_neighbors = 100) per pocket (:245) and same_res = res_idx[src] == res_idx[dst] per edge (:248).
 *   rec_x [n_rec,3], rec_ptr [B+1] (device; pocket b = rows [rec_ptr[b], rec_ptr[b+1]), at most max_rec <= 2048 each),
 *   res_idx [n_rec] or NULL.  Out: src/dst [cap] dst-major with src ascending (global row numbers), rowptr [n_rec+1],
 *   per_graph [B], same_res [cap] bytes or NULL, counts [2] = {n_edges, 0} (edges beyond cap are counted, not written);
 *   scratch: kpd_rec_graph_scratch_bytes(n_rec, B) device bytes.  The complete rec -> kp edge list (:251-253) is
 *   implicit (dst-major, every receptor atom of the pocket) and is never materialised by this library. */
int64_t kpd_rec_graph_scratch_bytes(int32_t n_rec, int32_t B);
kpd_status kpd_build_rec_graph(const float *rec_x, const int32_t *rec_ptr, int32_t B, int32_t n_rec, int32_t max_rec,
                               float r, int32_t max_nn, const int32_t *res_idx, int32_t cap, int32_t *src, int32_t *dst,
                               int32_t *rowptr, int32_t *per_graph, uint8_t *same_res, int32_t *counts, void *scratch,
                               void *stream);

/* Output side of sampling: element decode and XYZ text for a batch of sampled ligands.
 * Replaces the tensor -> text part of write_sampled_ligands (sample.py:66-90: torch.argmax over the feature columns,
 * dataset.lig_atom_idx_to_element) and write_xyz_file (utils.py:11-21: "<n>\n\n" + "<el> <x:.3f> <y:.3f> <z:.3f>\n"
 * per atom), as analysis/molecule_builder.py:47-48 calls it per ligand; bond perception (openbabel / rdkit) stays with
 * the caller.  The bytes equal Python's: "%.3f" of the exact fp32 value, round-half-even, '-0.000', 'nan', 'inf'.
 *   pos [n_atoms,3], feat [n_atoms,F], lig_ptr [B+1] (device); symbols [F] device, each element symbol as up to four
 *   NUL-padded bytes packed little-endian; elem [n_atoms] out (argmax, first maximum); text [capacity] out; text_ptr
 *   [B+1] int64 out (ligand b's block = text[text_ptr[b] : text_ptr[b+1]]; text_ptr[B] is the size needed, also when it
 *   exceeds capacity); status [1] out: bit 0 = a coordinate with |x| >= 2^53 was printed as '?', bit 1 = capacity too
 *   small (nothing written for the ligands that do not fit); scratch: kpd_xyz_scratch_bytes(n_atoms, B) device bytes. */
int64_t kpd_xyz_scratch_bytes(int32_t n_atoms, int32_t B);
kpd_status kpd_xyz_emit(const float *pos, const float *feat, const int32_t *lig_ptr, int32_t n_atoms, int32_t B,
                        int32_t F, const uint32_t *symbols, int32_t *elem, uint8_t *text, int64_t capacity,
                        int64_t *text_ptr, int32_t *status, void *scratch, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* KPD_H */
