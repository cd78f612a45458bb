// Training path of the EGNN denoiser: forward with saved layer states + backward.
//
// Gradients of LigRecDynamics.forward (models/dynamics.py:342-385; LigRecEGNN :266-294, LigRecConv :89-217) with respect
// to every parameter and to the four input tensors, for the loss of KeypointDiffusion.forward
// (models/ligand_diffuser.py:89-175), which train.py:423-524 differentiates with torch autograd.
//
// Formulation.  Parameters are read in place in the reference [out, in] layout (they change every optimizer step; what the kernels want
// in another order is packed per layer and step) and gradients are accumulated in the same layout.  The first-layer split of the
// inference path carries over to the backward pass: pre1[e] = U[src] + V[dst] + d_e w_r + b1 with U = h_src W1[:, :257]^T and
// V = h_dst W1[:, 257:514]^T, hence dW1 and dh need only the per-node sums of dpre1 (segmented by dst, scattered by src) and node-sized
// GEMMs -- batched over all edge MLPs of a layer per node type (layer_stage / layer_project / layer_cat_bwd).  The per-edge work of a
// layer is two kernels (egnn_kernels.hip): k_egnn_edge_train (forward: gather, SiLU, 257 x 257 product, SiLU, heads, segment pieces,
// keeping pre1 / a1 / pre2) and k_egnn_edge_bwd (head backward, dpre2 W2, SiLU backward, d dij, by-destination sums, in place over
// the kept arrays); dW2 = dpre2^T a1 and the by-source sums need whole matrices and stay separate (sgemm.hip, k_segsum264).  (The
// per-branch kernels those two replaced in round 4 -- k_edge_pre1 + ws_gemm + head / segmented-sum kernels -- were removed in round 5.)
// Memory: the edge activations of all layers when they fit (13.5 GB at C2, B = 64: pre1, a1, pre2 per branch), else one layer's slots
// (3 GB) and a recomputation per layer in backward; less than that is an out-of-memory error of kpd_egnn_trainer_reserve.
#include <cstring>

#include "egnn_kernels.h"
#include "engine.h"
#include "train_ops.h"

namespace kpd {
namespace {

constexpr int H = HW;         // 257
constexpr int LD = HS;        // 264: row stride of every activation matrix
typedef float f32x4 __attribute__((ext_vector_type(4)));




// By-source sums of the E x 257 gradient matrices of a layer (the backward of the gather h[src] in front of the edge MLPs): one WAVE per node, a
// lane owns four columns (lane 0 column 256 too), rows are fetched four at a time through the by-source permutation and added in edge order -- a
// fixed order per column.   out[v] = sum_j M[perm[j]],  j in [rowptr[v], rowptr[v + 1]);  out rows are ldo floats apart.
// All (edge type, branch) pairs of a layer in one launch (blockIdx.y = pair): eight launches of very different sizes before -- the ligand-sized
// ones pure latency, every one with its own tail.
constexpr int SEGSUM_BATCH = 8;
struct SegsumSrcBatch {
    struct One {
        const float *M;
        const int *perm, *rowptr;
        int n;
        float *out;
    } e[SEGSUM_BATCH];
    int ldo;
};
__global__ __launch_bounds__(256) void k_segsum264(SegsumSrcBatch b) {
    const SegsumSrcBatch::One &e = b.e[blockIdx.y];
    const float *__restrict__ M = e.M;
    const int *__restrict__ perm = e.perm;
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (v >= e.n) return;
    const int lo = e.rowptr[v], hi = e.rowptr[v + 1];
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    float t = 0.0f;
    int j = lo;
    for (; j + 4 <= hi; j += 4) {
        int r[4];
        f32x4 m[4];
        float mt[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = perm[j + k];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            m[k] = *reinterpret_cast<const f32x4 *>(M + (size_t)r[k] * LD + 4 * lane);
            mt[k] = lane == 0 ? M[(size_t)r[k] * LD + 256] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s += m[k];
            t += mt[k];
        }
    }
    for (; j < hi; ++j) {
        const int r = perm[j];
        s += *reinterpret_cast<const f32x4 *>(M + (size_t)r * LD + 4 * lane);
        if (lane == 0) t += M[(size_t)r * LD + 256];
    }
    *reinterpret_cast<f32x4 *>(e.out + (size_t)v * b.ldo + 4 * lane) = s;
    if (lane == 0) e.out[(size_t)v * b.ldo + 256] = t;
}

// h' = LayerNorm(h + q2 + b2) (or without the norm), one wave per node (dynamics.py:202-205)
__global__ void k_node_out(const float *__restrict__ h, const float *__restrict__ q2, const float *__restrict__ b2,
                           const float *__restrict__ gamma, const float *__restrict__ beta, int norm, int n, int hid,
                           float *__restrict__ out) {
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n) return;
    float u[5], s = 0.0f;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int c = lane + 64 * k;
        u[k] = (c < hid || c == H - 1) ? h[(size_t)r * LD + c] + q2[(size_t)r * LD + c] + b2[c] : 0.0f;
        s += u[k];
    }
    if (!norm) {
#pragma unroll
        for (int k = 0; k < 5; ++k)
            if (lane + 64 * k < H) out[(size_t)r * LD + lane + 64 * k] = u[k];
        return;
    }
    const float inv_n = 1.0f / (float)(hid + 1);      // live columns: [0, hid) and the timestep column (hid < 256: zero padding between)
    const float mean = wave_sum(s) * inv_n;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < 5; ++k)
        if (lane + 64 * k < hid || lane + 64 * k == H - 1) q += (u[k] - mean) * (u[k] - mean);
    const float rstd = rsqrtf(wave_sum(q) * inv_n + 1e-5f);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int c = lane + 64 * k;
        if (c < H) out[(size_t)r * LD + c] = (c < hid || c == H - 1) ? (u[k] - mean) * rstd * gamma[c] + beta[c] : 0.0f;
    }
}

__global__ void k_axpy3(const float *__restrict__ x, const float *__restrict__ xn, int n3, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) out[i] = x[i] + xn[i];
}

// node rows [h(256 after the encoder) | t]: out[r][0..255] = in (already there), out[r][256] = t[bidx[r]]
__global__ void k_set_time(float *__restrict__ hmat, const float *__restrict__ t, const int *__restrict__ bidx, int n) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) hmat[(size_t)r * LD + 256] = t[bidx[r]];
}


__global__ void k_zinv(const float *__restrict__ z, const int *__restrict__ bidx, int n, float *__restrict__ zinv) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) zinv[r] = 1.0f / z[bidx[r]];
}



// LayerNorm backward, one wave per node: u = h + q2 + b2 recomputed; du, and dy * xhat for the gamma gradient
__global__ void k_ln_bwd(const float *__restrict__ h, const float *__restrict__ q2, const float *__restrict__ b2,
                         const float *__restrict__ gamma, const float *__restrict__ dy, int n, int hid, float *__restrict__ du,
                         float *__restrict__ dyxhat) {
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n) return;
    float u[5], s = 0.0f;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int c = lane + 64 * k;
        u[k] = (c < hid || c == H - 1) ? h[(size_t)r * LD + c] + q2[(size_t)r * LD + c] + b2[c] : 0.0f;
        s += u[k];
    }
    const float inv_n = 1.0f / (float)(hid + 1);
    const float mean = wave_sum(s) * inv_n;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < 5; ++k)
        if (lane + 64 * k < hid || lane + 64 * k == H - 1) q += (u[k] - mean) * (u[k] - mean);
    const float rstd = rsqrtf(wave_sum(q) * inv_n + 1e-5f);
    float g[5], xh[5], sg = 0.0f, sgx = 0.0f;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int c = lane + 64 * k;
        const bool live = c < hid || c == H - 1;
        xh[k] = live ? (u[k] - mean) * rstd : 0.0f;
        const float d = live ? dy[(size_t)r * LD + c] : 0.0f;
        g[k] = live ? d * gamma[c] : 0.0f;
        sg += g[k];
        sgx += g[k] * xh[k];
        if (c < H) dyxhat[(size_t)r * LD + c] = d * xh[k];
    }
    sg = wave_sum(sg) * inv_n;
    sgx = wave_sum(sgx) * inv_n;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int c = lane + 64 * k;
        if (c < H) du[(size_t)r * LD + c] = (c < hid || c == H - 1) ? rstd * (g[k] - sg - xh[k] * sgx) : 0.0f;
    }
}

// The same with the row sums of the two parameter gradients taken on the way (as k_ln_bwd_gp of the GVP trainers): a workgroup owns LNP_ROWS
// rows, a wave every fourth of them, and leaves sum_r dy xhat / sum_r dy per column as part[block][0 | 1][c] in k_colsum's partial format --
// one k_colsum_reduce then finishes gamma.grad and beta.grad (instead of a dy * xhat array, two column-sum launches and two reductions).
constexpr int LNP_ROWS = 8;
__global__ __launch_bounds__(256) void k_ln_bwd_p(const float *__restrict__ h, const float *__restrict__ q2, const float *__restrict__ b2,
                                                  const float *__restrict__ gamma, const float *__restrict__ dy, int n, int hid, float *__restrict__ du,
                                                  float *__restrict__ part) {
    __shared__ float s_gx[4][320], s_g[4][320];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = blockIdx.x * LNP_ROWS, r1 = min(n, r0 + LNP_ROWS);
    float agx[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, ag[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const float inv_n = 1.0f / (float)(hid + 1);
    for (int r = r0 + wave; r < r1; r += 4) {
        float u[5], s = 0.0f;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int c = lane + 64 * k;
            u[k] = (c < hid || c == H - 1) ? h[(size_t)r * LD + c] + q2[(size_t)r * LD + c] + b2[c] : 0.0f;
            s += u[k];
        }
        const float mean = wave_sum(s) * inv_n;
        float q = 0.0f;
#pragma unroll
        for (int k = 0; k < 5; ++k)
            if (lane + 64 * k < hid || lane + 64 * k == H - 1) q += (u[k] - mean) * (u[k] - mean);
        const float rstd = rsqrtf(wave_sum(q) * inv_n + 1e-5f);
        float g[5], xh[5], sg = 0.0f, sgx = 0.0f;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int c = lane + 64 * k;
            const bool live = c < hid || c == H - 1;
            xh[k] = live ? (u[k] - mean) * rstd : 0.0f;
            const float d = live ? dy[(size_t)r * LD + c] : 0.0f;
            g[k] = live ? d * gamma[c] : 0.0f;
            sg += g[k];
            sgx += g[k] * xh[k];
            agx[k] += d * xh[k];
            ag[k] += c < H ? dy[(size_t)r * LD + c] : 0.0f;          // (the bias gradient sums dy over all H columns, as colsum_acc(dy) did)
        }
        sg = wave_sum(sg) * inv_n;
        sgx = wave_sum(sgx) * inv_n;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int c = lane + 64 * k;
            if (c < H) du[(size_t)r * LD + c] = (c < hid || c == H - 1) ? rstd * (g[k] - sg - xh[k] * sgx) : 0.0f;
        }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        s_gx[wave][lane + 64 * k] = agx[k];
        s_g[wave][lane + 64 * k] = ag[k];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 256) {
        float *p = part + (size_t)blockIdx.x * 2 * COLSUM_LD;
        p[c] = ((s_gx[0][c] + s_gx[1][c]) + s_gx[2][c]) + s_gx[3][c];
        p[COLSUM_LD + c] = ((s_g[0][c] + s_g[1][c]) + s_g[2][c]) + s_g[3][c];
    }
}

// dx[nt][v] += sum over the edge types that leave type nt of the per-edge position gradients of v's out-edges (by-source index, ascending edge
// order) - sum over the edge types that enter it of those of v's in-edges: one thread per node, both node types and all edge types of a layer in
// one launch, in a fixed order (no atomics).  Replaces two 3-wide segment-sum launches per edge type.
struct DxGatherArgs {
    const float *redge[4];
    const int *perm[4], *srowptr[4], *drowptr[4];
    int live[4], src_nt[4], dst_nt[4];
    int n[2];
    float *dx[2];
};
__global__ __launch_bounds__(256) void k_dx_gather(DxGatherArgs a) {
    // sixteen lanes per node: lane i takes edges i, i + 16, ... of every list; the lane sums are combined by a fixed shuffle tree
    const int nt = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x, v = t >> 4, l16 = t & 15;
    const bool in = v < a.n[nt];
    const int vv = in ? v : 0;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int et = 0; et < 4; ++et) {
        if (!a.live[et] || !in) continue;
        const float *r = a.redge[et];
        if (a.src_nt[et] == nt) {
            for (int j = a.srowptr[et][vv] + l16; j < a.srowptr[et][vv + 1]; j += 16) {
                const int e = a.perm[et][j];
                s0 += r[3 * e]; s1 += r[3 * e + 1]; s2 += r[3 * e + 2];
            }
        }
        if (a.dst_nt[et] == nt) {
            for (int j = a.drowptr[et][vv] + l16; j < a.drowptr[et][vv + 1]; j += 16) { s0 -= r[3 * j]; s1 -= r[3 * j + 1]; s2 -= r[3 * j + 2]; }
        }
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) {
        s0 += __shfl_xor(s0, o);
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if (in && l16 == 0) {
        float *o = a.dx[nt] + 3 * (size_t)v;
        o[0] += s0; o[1] += s1; o[2] += s2;
    }
}

// geometry backward: n = x_diff / (dij + 1), dij = |x_diff|; per-edge gradient of x_src (= minus that of x_dst)
__global__ void k_geom_bwd(const float *__restrict__ ddij, const float *__restrict__ dn, const float *__restrict__ xdiff,
                           const float *__restrict__ dij, int E, float *__restrict__ redge) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const float d = dij[e], inv = 1.0f / (d + 1.0f);
    const float x0 = xdiff[3 * e], x1 = xdiff[3 * e + 1], x2 = xdiff[3 * e + 2];
    const float g0 = dn[3 * e], g1 = dn[3 * e + 1], g2 = dn[3 * e + 2];
    const float dd = ddij[e] - (g0 * x0 + g1 * x1 + g2 * x2) * inv * inv;     // total gradient of dij
    const float k = d > 0.0f ? dd / d : 0.0f;
    const float r0 = g0 * inv + k * x0, r1 = g1 * inv + k * x1, r2 = g2 * inv + k * x2;
    redge[3 * e] = r0; redge[3 * e + 1] = r1; redge[3 * e + 2] = r2;       // + to x_src, - to x_dst: summed per node by the caller
}






// ---- the first Linear of every edge MLP of a layer, batched per node type --------------------------------------------------------
// pre1 = U[src] + V[dst] + ... with U = h_src W1[:, :257]^T, V = h_dst W1[:, 257:514]^T: a layer has up to 8 such per-node products
// per node type (4 edge types x 2 branches x {src, dst} over 2 node types).  Their weight blocks are staged side by side
// (wcat [slot][264][264], zero padded), so that per node type ONE product gives all projections (ucat [n][slot][264], the layout of the
// inference engine's P), ONE product adds all their contributions to dh (ducat . wcat) and ONE gives all weight gradients
// (ducat^T h, with the b1 gradients as its column sums) -- instead of 8 ligand-sized products each, which ran at a fraction of the
// GPU (13 row tiles) and paid a split-K reduction apiece.
struct CatSlot {
    const float *w, *b;         // W1 of the branch [257][515]; b1 (dst slots: it rides in the projection) or null
    float *g, *bg;              // the gradient of W1 (or null); b1 gradient (dst slots, or null)
    int col0, nt, slot, dvw;    // first column of the block in W1; node type; slot in wcat[nt]; index of the slot among the dst slots of nt (-1: src)
};
struct CatTab {
    CatSlot s[16];
    int n;
};
constexpr int CAT_LD = NSLOT * LD;            // row stride of ucat / ducat
__global__ void k_cat_stage(CatTab t, float *__restrict__ wcat0, float *__restrict__ wcat1, float *__restrict__ bcat0, float *__restrict__ bcat1) {
    const CatSlot &e = t.s[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= LD * LD) return;
    const int r = i / LD, c = i - r * LD;
    (e.nt ? wcat1 : wcat0)[(size_t)e.slot * LD * LD + i] = (r < H && c < H) ? e.w[(size_t)r * (2 * H + 1) + e.col0 + c] : 0.0f;
    if (i < LD) (e.nt ? bcat1 : bcat0)[e.slot * LD + i] = (e.b && i < H) ? e.b[i] : 0.0f;
}
// W1.g[:, block] += dwcat[slot]; b1.g += column sums of dV (dbcat); W1.g[:, 514] += column sums of dVw (dwr).  One thread per element.
__global__ void k_cat_scatter(CatTab t, const float *__restrict__ dwcat0, const float *__restrict__ dwcat1, const float *__restrict__ dbcat0,
                              const float *__restrict__ dbcat1, const float *__restrict__ dwr0, const float *__restrict__ dwr1) {
    const CatSlot &e = t.s[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * LD) return;
    const int r = i / LD, c = i - r * LD;
    if (c < H && e.g) e.g[(size_t)r * (2 * H + 1) + e.col0 + c] += (e.nt ? dwcat1 : dwcat0)[((size_t)e.slot * LD + r) * LD + c];
    if (c == H && e.bg) e.bg[r] += (e.nt ? dbcat1 : dbcat0)[e.slot * LD + r];
    if (c == H + 1 && e.dvw >= 0 && e.g) e.g[(size_t)r * (2 * H + 1) + 2 * H] += (e.nt ? dwr1 : dwr0)[e.dvw * LD + r];
}

}  // namespace
}  // namespace kpd

using namespace kpd;

struct kpd_egnn_trainer : TrainCtx {
    kpd_egnn_config cfg{};
    // HIP-event timing of the two per-layer edge kernels (kpd_egnn_trainer_profile): (start, stop) pairs, which kernel, how many edges
    std::vector<hipEvent_t> prof_ev;
    float *wsg_pack = nullptr;                 // per-call weight pack of ws_gemm (node MLPs)
    unsigned long long *stamps = nullptr;      // TOOLS build, KPD_TRAIN_STAMPS=1: phase-cycle sums of the two edge kernels (printed by profile_read)
    std::vector<int> prof_tag;
    std::vector<double> prof_edges;
    bool prof_on = false;
    size_t prof_used = 0;
    template <class F>
    kpd_status timed(int tag, double edges, F &&launch) {
        const bool on = prof_on && prof_used + 2 <= prof_ev.size();
        if (on) KPD_HIP(hipEventRecord(prof_ev[prof_used], st));
        KPD_TRY(launch());
        if (on) {
            KPD_HIP(hipEventRecord(prof_ev[prof_used + 1], st));
            prof_tag[prof_used / 2] = tag;
            prof_edges[prof_used / 2] = edges;
            prof_used += 2;
        }
        return KPD_OK;
    }
    Arena ws;
    int n_et = 2, n_upd = 1;
    bool rec_identity = false;
    // capacities
    int cap_B = 0, cap_lig = 0, cap_kp = 0, cap_kk = 0, cap_maxlig = 0, cap_maxkp = 0, cap_ll = 0, cap_kl = 0, cap_E = 0, cap_N = 0;
    // batch of the last forward
    kpd_batch bt{};
    const float *t_dev = nullptr;
    bool have_forward = false;
    int n[2] = {0, 0}, E[4] = {0, 0, 0, 0};
    const int *e_src[4] = {nullptr, nullptr, nullptr, nullptr}, *e_dst[4] = {nullptr, nullptr, nullptr, nullptr},
              *e_rowptr[4] = {nullptr, nullptr, nullptr, nullptr};
    // graph build
    kpd_lig_graph lg{};
    int *meta = nullptr, *ll_deg = nullptr, *ll_off = nullptr, *kl_off = nullptr, *kl_pg = nullptr, *bidx[2] = {nullptr, nullptr};
    float *z[2] = {nullptr, nullptr}, *zinv[2] = {nullptr, nullptr};
    SrcCsr scsr[4];                     // edges of each type grouped by source node (deterministic sums over out-edges)
    int *cursor = nullptr;
    // saved node states: index l = input of layer l (l = n_layers: output of the stack)
    std::vector<float *> hs[2], xs[2], hns[2], xns[2];
    // scratch
    float *nb[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};      // [cap_N, LD] each
    float *nbw[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};     // per node type: du, dq1 of node_bwd and (recompute mode) the node MLP's hidden activation, kept until the layer's batched weight gradients ran
    std::vector<Grad257Item> wq_nodes;                               // the node MLPs' weight gradients of the current layer (grad257_batch)
    float *dact = nullptr;                                                       // [cap_N, ENC_LD]
    // Kept forward activations (KPD_TRAIN_STORE, default on): pre1 / a1 / pre2 / a2 of both MLP branches of every (layer, edge
    // type), the attention weights, the geometry and the coordinate head, so that the backward pass reads them instead of running
    // the gather, the per-node projections and the 257 x 257 GEMM of every branch a second time (~12 % of a training step).  13.5 GB at
    // C2, B = 64 -- sized for a 288-GB part; if the allocation fails the engine falls back to recomputation.
    struct Slot {
        float *e[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
        float *att = nullptr, *dij = nullptr, *xdiff = nullptr, *nvec = nullptr, *sc = nullptr, *msgx = nullptr;
    };
    bool store = false;
    char *store_base = nullptr;
    std::vector<Slot> slots;                       // [layer * 4 + et]
    // batched first-layer products of the current layer (k_cat_stage): staged weights, projections, per-node gradients and their staging
    float *wcat[2] = {nullptr, nullptr}, *ucat[2] = {nullptr, nullptr}, *ducat[2] = {nullptr, nullptr}, *dvwcat[2] = {nullptr, nullptr},
          *dwcat[2] = {nullptr, nullptr}, *dbcat[2] = {nullptr, nullptr}, *dwr[2] = {nullptr, nullptr}, *bcat[2] = {nullptr, nullptr};
    CatTab cat{};
    // forward edge kernel (k_egnn_edge_train): per-step weight pack of the current layer, segment pieces, whether slots of a whole layer exist
    float *epack = nullptr, *hn_main[4] = {nullptr, nullptr, nullptr, nullptr}, *hn_cont[4] = {nullptr, nullptr, nullptr, nullptr},
          *xn_main[4] = {nullptr, nullptr, nullptr, nullptr}, *xn_cont[4] = {nullptr, nullptr, nullptr, nullptr};
    // backward edge kernel: pieces of the by-destination sums of dpre1 and dij * dpre1 per (edge type, branch); per-tile column-sum partials
    float *dv_main[4][2] = {}, *dv_cont[4][2] = {}, *dvw_main[4][2] = {}, *dvw_cont[4][2] = {}, *bpart[2] = {nullptr, nullptr};
    int bpart_tiles = 0;
    int cat_slots[2] = {0, 0}, cat_dvw[2] = {0, 0};
    int cat_of[4][2][2] = {}, cat_dvw_of[4][2] = {};   // [et][branch][src | dst] -> slot on that side's node type; [et][branch] -> dvw index
    std::vector<float *> nq[2][3];                 // kept node-MLP activations q1, c1, q2 of every (node type, layer), with the edge activations
    float *dh[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}, *dx[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    float *enc1[2] = {nullptr, nullptr}, *enc2[2] = {nullptr, nullptr}, *dec1 = nullptr, *dec2 = nullptr;   // encoder / decoder scratch
};

namespace {

const char *kEt[4] = {"ll", "kl", "lk", "kk"};
const char *kNt[2] = {"lig", "kp"};
const int kS[4] = {NT_LIG, NT_KP, NT_LIG, NT_KP};     // source node type of ll, kl, lk, kk
const int kD[4] = {NT_LIG, NT_LIG, NT_KP, NT_KP};


struct BranchParams {
    Param W1, b1, W2, b2, head, head_b;     // head = soft_attention weight [1,257] (+ bias) or coord_mlp.4 weight [1,257]
};

kpd_status branch_params(kpd_egnn_trainer *T, int layer, int et, int branch, BranchParams *p) {
    const std::string base = "egnn.conv_layers." + std::to_string(layer) + ".";
    const std::string mlp = base + (branch == 0 ? "edge_mlp." : "coord_mlp.") + kEt[et];
    KPD_TRY(param(T, mlp + ".0.weight", H, 2 * H + 1, &p->W1));
    KPD_TRY(param(T, mlp + ".0.bias", H, 1, &p->b1));
    KPD_TRY(param(T, mlp + ".2.weight", H, H, &p->W2));
    KPD_TRY(param(T, mlp + ".2.bias", H, 1, &p->b2));
    if (branch == 0) {
        KPD_TRY(param(T, base + "soft_attention." + kEt[et] + ".0.weight", 1, H, &p->head));
        KPD_TRY(param(T, base + "soft_attention." + kEt[et] + ".0.bias", 1, 1, &p->head_b));
    } else {
        KPD_TRY(param(T, mlp + ".4.weight", 1, H, &p->head));
    }
    return KPD_OK;
}

// LigRecEGNN.forward returns (h_lig, x_lig) only (dynamics.py:288-294): the final layer's lk / kk messages and its keypoint
// update feed nothing and their gradients are exactly zero, so that layer runs (forward and backward) on ll + kl and the
// ligand update alone -- the same pruning as the inference engine (egnn.hip).
inline int layer_n_et(const kpd_egnn_trainer *T, int l) { return l == T->cfg.n_layers - 1 ? 2 : T->n_et; }
inline int layer_n_upd(const kpd_egnn_trainer *T, int l) { return l == T->cfg.n_layers - 1 ? 1 : T->n_upd; }

// slots of layer l (edge types without edges get none), the staged weight blocks, and -- forward -- the projections of both node types
kpd_status layer_stage(kpd_egnn_trainer *T, int l) {
    CatTab &t = T->cat;
    t.n = 0;
    T->cat_slots[0] = T->cat_slots[1] = T->cat_dvw[0] = T->cat_dvw[1] = 0;
    for (int et = 0; et < layer_n_et(T, l); ++et) {
        if (T->E[et] == 0) continue;
        for (int br = 0; br < 2; ++br) {
            BranchParams p;
            KPD_TRY(branch_params(T, l, et, br, &p));
            for (int side = 0; side < 2; ++side) {
                const int nt = side ? kD[et] : kS[et];
                CatSlot &e = t.s[t.n++];
                e.w = p.W1.w; e.b = side ? p.b1.w : nullptr; e.g = p.W1.g; e.bg = side ? p.b1.g : nullptr;
                e.col0 = side * H; e.nt = nt; e.slot = T->cat_slots[nt]++;
                e.dvw = side ? T->cat_dvw[nt]++ : -1;
                T->cat_of[et][br][side] = e.slot;
                if (side) T->cat_dvw_of[et][br] = e.dvw;
            }
        }
    }
    if (t.n == 0) return KPD_OK;
    hipLaunchKernelGGL(k_cat_stage, dim3(cdiv(LD * LD, 256), t.n), dim3(256), 0, T->st, t, T->wcat[0], T->wcat[1], T->bcat[0], T->bcat[1]);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status layer_project(kpd_egnn_trainer *T, int l) {
    for (int nt = 0; nt < 2; ++nt)
        if (T->cat_slots[nt])
            KPD_TRY(gemm(T, false, true, T->n[nt], T->cat_slots[nt] * LD, H, T->hs[nt][l], LD, T->wcat[nt], LD, 0.0f, T->ucat[nt], CAT_LD, 1.0f, nullptr,
                         T->bcat[nt]));          // (+ b1 on the dst slots)
    return KPD_OK;
}

// after the edge types of layer l: the weight gradients of all first Linears and their contributions to dh, one product each per node type
kpd_status layer_cat_bwd(kpd_egnn_trainer *T, int l, int nxt) {
    const CatTab &t = T->cat;
    if (t.n == 0) return KPD_OK;
    bool want = false;
    for (int i = 0; i < t.n; ++i) want = want || t.s[i].g || t.s[i].bg;
    for (int nt = 0; nt < 2; ++nt) {
        const int ns = T->cat_slots[nt], n = T->n[nt];
        if (!ns) continue;
        if (want) {
            KPD_HIP(hipMemsetAsync(T->dbcat[nt], 0, (size_t)NSLOT * LD * 4, T->st));
            KPD_TRY(sgemm(true, false, ns * LD, H, n, 1.0f, T->ducat[nt], CAT_LD, T->hs[nt][l], LD, 0.0f, T->dwcat[nt], LD, T->st, T->part, T->part_floats,
                          T->dbcat[nt]));
            if (T->cat_dvw[nt]) {
                const int cols = T->cat_dvw[nt] * LD;
                KPD_HIP(hipMemsetAsync(T->dwr[nt], 0, (size_t)NSLOT * LD * 4, T->st));
                for (int c0 = 0; c0 < cols; c0 += COLSUM_LD)
                    KPD_TRY(colsum_acc(T, n, std::min(COLSUM_LD, cols - c0), T->dvwcat[nt] + c0, CAT_LD, T->dwr[nt] + c0));
            }
        }
        KPD_TRY(gemm(T, false, false, n, H, ns * LD, T->ducat[nt], CAT_LD, T->wcat[nt], LD, 1.0f, T->dh[nxt][nt], LD));
    }
    if (want) {
        hipLaunchKernelGGL(k_cat_scatter, dim3(cdiv(H * LD, 256), t.n), dim3(256), 0, T->st, t, T->dwcat[0], T->dwcat[1], T->dbcat[0], T->dbcat[1],
                           T->dwr[0], T->dwr[1]);
        KPD_LAUNCH_CHECK();
    }
    return KPD_OK;
}

// floats of the per-layer weight pack of the forward edge kernel: per (et, branch) wp, wx, wr; per et watt, w3
constexpr size_t EPACK_ENTRY = (size_t)WP_FLOATS + 2 * HS;
constexpr size_t EPACK_FLOATS = 8 * EPACK_ENTRY + 8 * HS;

// all edge types and both branches of layer l in one launch: messages, heads, segment pieces, kept activations.  sum_pieces: add the
// aggregated messages into hns / xns (the forward pass; a recomputation for the backward pass only refills the kept arrays)
kpd_status layer_edges_fused(kpd_egnn_trainer *T, int l, bool sum_pieces) {
    const kpd_egnn_config &c = T->cfg;
    EdgePackTab pk;
    pk.n = 0;
    pk.transposed = 0;
    EdgeTrainArgs a{};
    a.meta = T->meta + (l == c.n_layers - 1 ? 16 : 0);
    a.use_tanh = c.use_tanh; a.coords_range = c.coords_range;
    a.stamps = T->stamps;
    a.skip = tool_env_int("KPD_TR_SKIP", 0);
    int tiles = 0;
    for (int nt = 0; nt < 2; ++nt) { a.x[nt] = T->xs[nt][l]; a.P[nt] = T->ucat[nt]; }
    for (int et = 0; et < 4; ++et) {
        a.src[et] = T->e_src[et]; a.dst[et] = T->e_dst[et];
        a.src_nt[et] = kS[et]; a.dst_nt[et] = kD[et];
        if (et >= layer_n_et(T, l) || T->E[et] == 0) continue;
        tiles += cdiv(T->E[et], TM);
        const kpd_egnn_trainer::Slot &sl = T->slots[(size_t)l * 4 + et];
        float *heads = T->epack + 8 * EPACK_ENTRY + (size_t)et * 2 * HS;
        a.watt[et] = heads; a.w3[et] = heads + HS;
        for (int br = 0; br < 2; ++br) {
            BranchParams p;
            KPD_TRY(branch_params(T, l, et, br, &p));
            float *base = T->epack + (size_t)(et * 2 + br) * EPACK_ENTRY;
            EdgePackEntry &e = pk.e[pk.n++];
            e.W1 = p.W1.w; e.W2 = p.W2.w; e.b2 = p.b2.w; e.head = p.head.w; e.head_b = br == 0 ? p.head_b.w : nullptr;
            e.wp = base; e.wx = base + WP_FLOATS; e.wr = base + WP_FLOATS + HS; e.head_out = heads + (size_t)br * HS;
            a.wp[et][br] = e.wp; a.wx[et][br] = e.wx; a.wr[et][br] = e.wr;
            for (int side = 0; side < 2; ++side) a.slot[et][br][side] = T->cat_of[et][br][side];
            for (int k = 0; k < 4; ++k) a.keep[et][br][k] = sl.e[br][k];
        }
        a.att[et] = sl.att; a.sc[et] = sl.sc; a.dij[et] = sl.dij; a.xdiff[et] = sl.xdiff; a.nvec[et] = sl.nvec;
        a.hn_main[et] = T->hn_main[et]; a.hn_cont[et] = T->hn_cont[et]; a.xn_main[et] = T->xn_main[et]; a.xn_cont[et] = T->xn_cont[et];
    }
    if (tiles == 0 && !sum_pieces) return KPD_OK;          // (with sum_pieces the neighbour sums are still written: zeros)
    KPD_TRY(launch_edge_train_pack(pk, T->st));
    {
        double edges = 0.0;
        for (int et = 0; et < layer_n_et(T, l); ++et) edges += T->E[et];
        KPD_TRY(T->timed(0, edges, [&] { return launch_egnn_edge_train(a, tiles, T->st); }));
    }
    if (sum_pieces) {
        EdgePiecesSumArgs pa;
        memset(&pa, 0, sizeof(pa));
        for (int et = 0; et < layer_n_et(T, l); ++et) {
            if (T->E[et] == 0) continue;
            const int d = kD[et];
            pa.hn_main[et] = T->hn_main[et]; pa.hn_cont[et] = T->hn_cont[et]; pa.xn_main[et] = T->xn_main[et]; pa.xn_cont[et] = T->xn_cont[et];
            pa.rowptr[et] = T->e_rowptr[et]; pa.live[et] = 1; pa.dst_nt[et] = d;
        }
        for (int k = 0; k < layer_n_upd(T, l); ++k) {          // every updated node type gets its rows written, edges or not
            pa.zinv[k] = T->zinv[k]; pa.n[k] = T->n[k]; pa.hn[k] = T->hns[k][l]; pa.xn[k] = T->xns[k][l];
        }
        KPD_TRY(launch_edge_pieces_sum(pa, T->st));
    }
    return KPD_OK;
}

// one LigRecConv layer forward (dynamics.py:124-207) from the saved inputs hs[l], xs[l] into hs[l+1], xs[l+1], hns[l], xns[l]
kpd_status layer_fwd(kpd_egnn_trainer *T, int l) {
    // (h_neigh / x_neigh of the layer are written whole by k_edge_pieces_sum: no memset; their padding columns keep the arena's zeros)
    KPD_TRY(layer_stage(T, l));
    KPD_TRY(layer_project(T, l));
    return layer_edges_fused(T, l, true);
}

struct NodeParams {
    Param W1, b1, W2, b2, gamma, beta;
};

kpd_status node_params(kpd_egnn_trainer *T, int l, int nt, NodeParams *p) {
    const std::string base = "egnn.conv_layers." + std::to_string(l) + ".";
    KPD_TRY(param(T, base + "node_mlp." + kNt[nt] + ".0.weight", H, 2 * H, &p->W1));
    KPD_TRY(param(T, base + "node_mlp." + kNt[nt] + ".0.bias", H, 1, &p->b1));
    KPD_TRY(param(T, base + "node_mlp." + kNt[nt] + ".2.weight", H, H, &p->W2));
    KPD_TRY(param(T, base + "node_mlp." + kNt[nt] + ".2.bias", H, 1, &p->b2));
    if (T->cfg.norm) {
        KPD_TRY(param(T, base + "layer_norm." + kNt[nt] + ".weight", H, 1, &p->gamma));
        KPD_TRY(param(T, base + "layer_norm." + kNt[nt] + ".bias", H, 1, &p->beta));
    }
    return KPD_OK;
}

// node MLP of layer l for node type nt: q[0] = q1 (+ bias), q[1] = c1 = SiLU(q1), q[2] = q2 (without its bias) -- the kept buffers of
// (nt, l), or nb[2..4] when nothing is kept
struct NodeAct { float *q[3]; };
inline NodeAct node_act(kpd_egnn_trainer *T, int l, int nt) {
    NodeAct a;
    for (int k = 0; k < 3; ++k) a.q[k] = T->store ? T->nq[nt][k][l] : k == 1 ? T->nbw[nt][2] : T->nb[2 + k];       // (q[1] outlives node_bwd: its weight gradient is batched)
    return a;
}
// The node MLPs' products are [n, 257] x [257, 257]: K is seventeen 16-row slabs, and in the tiled kernel (600 workgroups of 128 x 64 outputs, every one
// filling its LDS ring and draining its accumulators for 17 slabs of work) a receptor-sized product took 86 us -- 30 - 60 TFLOP/s.  ws_gemm.hip keeps half of
// the weight matrix resident in a workgroup's LDS and walks 128-row tiles (the shape k_proj_ws has in the inference engine), with the same fused epilogues.
// Ligand-sized products (13 row tiles) stay on the tiled kernel.
inline bool node_ws(const kpd_egnn_trainer *T, int n) {
    static const int on = tool_env_int("KPD_NODE_WS", 1);          // A/B runs
    return on && T->wsg_pack && n >= 4096;
}
kpd_status node_mlp_fwd(kpd_egnn_trainer *T, const NodeParams &p, int l, int nt, const NodeAct &a) {
    const int n = T->n[nt];
    if (node_ws(T, n)) {
        KPD_TRY(ws_gemm(WS_PLAIN, T->hs[nt][l], n, LD, p.W1.w, 2 * H, false, nullptr, nullptr, a.q[0], nullptr, LD, T->wsg_pack, T->st));
        KPD_TRY(ws_gemm(WS_BIAS_SILU, T->hns[nt][l], n, LD, p.W1.w + H, 2 * H, false, p.b1.w, nullptr, a.q[0], a.q[1], LD, T->wsg_pack, T->st, true, true));
        return ws_gemm(WS_PLAIN, a.q[1], n, LD, p.W2.w, H, false, nullptr, nullptr, a.q[2], nullptr, LD, T->wsg_pack, T->st);
    }
    KPD_TRY(gemm(T, false, true, n, H, H, T->hs[nt][l], LD, p.W1.w, 2 * H, 0.0f, a.q[0], LD));
    KPD_TRY(gemm(T, false, true, n, H, H, T->hns[nt][l], LD, p.W1.w + H, 2 * H, 1.0f, a.q[0], LD, 1.0f, nullptr, p.b1.w, a.q[1]));      // + bias, SiLU -> c1
    KPD_TRY(gemm(T, false, true, n, H, H, a.q[1], LD, p.W2.w, H, 0.0f, a.q[2], LD));
    return KPD_OK;
}

kpd_status nodes_fwd(kpd_egnn_trainer *T, int l) {
    for (int nt = 0; nt < 2; ++nt) {
        const int n = T->n[nt];
        if (nt >= T->n_upd) {       // kp not updated: every layer reads the encoder output and x_0 again (dynamics.py:288-292)
            T->hs[nt][l + 1] = T->hs[nt][0];
            T->xs[nt][l + 1] = T->xs[nt][0];
            continue;
        }
        if (nt >= layer_n_upd(T, l)) continue;       // final layer: the keypoint output is never read
        NodeParams p;
        KPD_TRY(node_params(T, l, nt, &p));
        const NodeAct na = node_act(T, l, nt);
        KPD_TRY(node_mlp_fwd(T, p, l, nt, na));
        hipLaunchKernelGGL(k_node_out, dim3(cdiv(n, 4)), dim3(256), 0, T->st, T->hs[nt][l], na.q[2], p.b2.w, p.gamma.w, p.beta.w,
                           T->cfg.norm, n, T->cfg.hidden_nf, T->hs[nt][l + 1]);
        KPD_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_axpy3, grid1(3 * n), dim3(256), 0, T->st, T->xs[nt][l], T->xns[nt][l], 3 * n, T->xs[nt][l + 1]);
        KPD_LAUNCH_CHECK();
    }
    return KPD_OK;
}

// two-layer MLPs at the ends (dynamics.py:313-334): y = [SiLU](W2 SiLU(W0 x + b0) + b2)
struct MlpParams {
    Param W0, b0, W2, b2;
    int fin, hid, fout;
};

kpd_status mlp_params(kpd_egnn_trainer *T, const char *name, int fin, int hid, int fout, MlpParams *p) {
    const std::string b = name;
    KPD_TRY(param(T, b + ".0.weight", hid, fin, &p->W0));
    KPD_TRY(param(T, b + ".0.bias", hid, 1, &p->b0));
    KPD_TRY(param(T, b + ".2.weight", fout, hid, &p->W2));
    KPD_TRY(param(T, b + ".2.bias", fout, 1, &p->b2));
    p->fin = fin; p->hid = hid; p->fout = fout;
    return KPD_OK;
}

// pre1 [n, hid] (ld LD... hid <= 512 uses its own stride), act1, pre2 [n, fout]; final_act: out = SiLU(pre2) else out = pre2
kpd_status mlp_fwd(kpd_egnn_trainer *T, const MlpParams &p, const float *x, int ldx, int n, float *pre1, float *act1, int ld1,
                   float *pre2, int ld2, float *out, int ldo, bool final_act) {
    KPD_TRY(gemm(T, false, true, n, p.hid, p.fin, x, ldx, p.W0.w, p.fin, 0.0f, pre1, ld1, 1.0f, nullptr, p.b0.w, act1));                   // + bias, SiLU -> act1
    KPD_TRY(gemm(T, false, true, n, p.fout, p.hid, act1, ld1, p.W2.w, p.hid, 0.0f, pre2, ld2));
    long long tot = (long long)n * p.fout;
    if (final_act) {
        KPD_REQUIRE(ld2 == ldo, KPD_ERR_INVALID, "internal: mlp_fwd strides");
        hipLaunchKernelGGL(k_bias_silu, grid1(tot), dim3(256), 0, T->st, pre2, p.b2.w, tot, p.fout, ld2, out);
    } else {
        hipLaunchKernelGGL(k_bias_add, grid1(tot), dim3(256), 0, T->st, pre2, p.b2.w, tot, p.fout, ld2);
        KPD_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_copy_rows, grid1(tot), dim3(256), 0, T->st, pre2, ld2, out, ldo, tot, p.fout);
    }
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

// backward of mlp_fwd given dout (overwritten); pre1 / act1 / pre2 as left by mlp_fwd; dx (may be null) = gradient of x
kpd_status mlp_bwd(kpd_egnn_trainer *T, const MlpParams &p, const float *x, int ldx, int n, const float *pre1, const float *act1,
                   int ld1, const float *pre2, int ld2, float *dout, int ldo, bool final_act, float *dact1, float *dx, int lddx) {
    if (final_act) {
        const long long tot = (long long)n * p.fout;
        KPD_REQUIRE(ld2 == ldo, KPD_ERR_INVALID, "internal: mlp_bwd strides");
        hipLaunchKernelGGL(k_silu_bwd, grid1(tot), dim3(256), 0, T->st, dout, pre2, tot, p.fout, ldo);
        KPD_LAUNCH_CHECK();
    }
    KPD_TRY(grad_gemm(T, p.fout, p.hid, n, dout, ldo, act1, ld1, p.W2.g, p.hid, p.b2.g));
    KPD_TRY(gemm(T, false, false, n, p.hid, p.fout, dout, ldo, p.W2.w, p.hid, 0.0f, dact1, ld1, 1.0f, pre1));      // * SiLU'(pre1) in the epilogue
    KPD_TRY(grad_gemm(T, p.hid, p.fin, n, dact1, ld1, x, ldx, p.W0.g, p.fin, p.b0.g));
    if (dx) KPD_TRY(gemm(T, false, false, n, p.fin, p.hid, dact1, ld1, p.W0.w, p.fin, 0.0f, dx, lddx));
    return KPD_OK;
}

constexpr int ENC_LD = 512;       // row stride of the encoder / decoder hidden activations (hid = 64, 2 rec_nf <= 512, 20)

}  // namespace

extern "C" kpd_status kpd_egnn_trainer_create(const kpd_egnn_config *cfg, kpd_egnn_trainer **out) {
    KPD_REQUIRE(cfg && out, KPD_ERR_INVALID, "null argument");
    KPD_REQUIRE(cfg->hidden_nf >= 1 && cfg->hidden_nf <= HID, KPD_ERR_INVALID, "hidden_nf=%d outside 1 .. %d", cfg->hidden_nf, HID);
    KPD_REQUIRE(cfg->atom_nf >= 1 && cfg->atom_nf <= 256 && cfg->rec_nf >= 1 && cfg->rec_nf <= 256 && cfg->n_layers >= 1 &&
                    cfg->n_layers <= 64,
                KPD_ERR_INVALID, "atom_nf=%d rec_nf=%d n_layers=%d", cfg->atom_nf, cfg->rec_nf, cfg->n_layers);
    KPD_REQUIRE(cfg->ll_k >= 0 && cfg->ll_k <= 16 && cfg->kl_k >= 0 && cfg->kl_k <= KL_KMAX, KPD_ERR_INVALID, "ll_k=%d kl_k=%d",
                cfg->ll_k, cfg->kl_k);
    kpd_egnn_trainer *T = new kpd_egnn_trainer();
    T->cfg = *cfg;
    T->n_et = cfg->update_kp_feat ? 4 : 2;
    T->n_upd = cfg->update_kp_feat ? 2 : 1;
    T->rec_identity = cfg->rec_nf == cfg->hidden_nf;        // models/dynamics.py:326-334
    *out = T;
    return KPD_OK;
}

extern "C" kpd_status kpd_egnn_trainer_profile(kpd_egnn_trainer *T, int32_t enable) {
    KPD_REQUIRE(T, KPD_ERR_INVALID, "null handle");
    if (enable && T->prof_ev.empty()) {
        T->prof_ev.resize(2 * 4096);
        T->prof_tag.assign(4096, 0);
        T->prof_edges.assign(4096, 0.0);
        for (hipEvent_t &e : T->prof_ev) KPD_HIP(hipEventCreate(&e));
    }
    T->prof_on = enable != 0;
    T->prof_used = 0;
    if (enable && !T->stamps && tool_env_int("KPD_TRAIN_STAMPS", 0)) KPD_HIP(hipMalloc(reinterpret_cast<void **>(&T->stamps), 64 * sizeof(unsigned long long)));
    if (T->stamps) KPD_HIP(hipMemset(T->stamps, 0, 64 * sizeof(unsigned long long)));
    return KPD_OK;
}

extern "C" kpd_status kpd_egnn_trainer_profile_read(kpd_egnn_trainer *T, double total_ms[2], int32_t launches[2], double edges[2]) {
    KPD_REQUIRE(T && total_ms && launches && edges, KPD_ERR_INVALID, "null argument");
    total_ms[0] = total_ms[1] = edges[0] = edges[1] = 0.0;
    launches[0] = launches[1] = 0;
    for (size_t i = 0; i + 1 < T->prof_used; i += 2) {
        KPD_HIP(hipEventSynchronize(T->prof_ev[i + 1]));
        float ms = 0.0f;
        KPD_HIP(hipEventElapsedTime(&ms, T->prof_ev[i], T->prof_ev[i + 1]));
        const int tag = T->prof_tag[i / 2];
        total_ms[tag] += ms;
        edges[tag] += T->prof_edges[i / 2];
        ++launches[tag];
    }
    if (T->stamps) {
        unsigned long long h[64];
        KPD_HIP(hipDeviceSynchronize());
        KPD_HIP(hipMemcpy(h, T->stamps, sizeof h, hipMemcpyDeviceToHost));
        static const char *fn[10] = {"geometry + run structure", "gather -> pre1, a1 (feature)", "GEMM (feature)", "pre2 stores (feature)", "attention head",
                                     "segmented sum", "gather -> pre1, a1 (coordinate)", "GEMM (coordinate)", "pre2 stores (coordinate)", "coordinate head + scan"};
        static const char *bn[5] = {"head rows -> dpre2", "row dots + GEMM", "dpre1 = acc SiLU'(pre1)", "d dij + segmented sums", "column 256 scan + column sums"};
        const double tf = (double)h[15], tb = (double)h[63];
        fprintf(stderr, "k_egnn_edge_train: %.0f tiles, s_memtime ticks per tile and phase (100 MHz)\n", tf);
        for (int i = 0; i < 10; ++i) fprintf(stderr, "  %-36s %8.1f\n", fn[i], tf > 0 ? (double)h[i] / tf : 0.0);
        fprintf(stderr, "k_egnn_edge_bwd: %.0f tiles\n  %-36s %8.1f\n", tb, "row scalars + run structure", tb > 0 ? (double)h[32] / tb : 0.0);
        for (int br = 0; br < 2; ++br)
            for (int i = 0; i < 5; ++i) fprintf(stderr, "  [%s] %-29s %8.1f\n", br ? "coord" : "feat", bn[i], tb > 0 ? (double)h[33 + 8 * br + i] / tb : 0.0);
    }
    return KPD_OK;
}

extern "C" void kpd_egnn_trainer_destroy(kpd_egnn_trainer *T) {
    if (!T) return;
    for (hipEvent_t &e : T->prof_ev) (void)hipEventDestroy(e);
    T->ws.release();
    T->wide.release();
    T->release_scratch();
    if (T->store_base) (void)hipFree(T->store_base);
    if (T->stamps) (void)hipFree(T->stamps);
    delete T;
}

extern "C" kpd_status kpd_egnn_trainer_bind(kpd_egnn_trainer *T, const char *name, const float *weight, float *grad,
                                            const int64_t *shape, int32_t ndim) {
    KPD_REQUIRE(T && name && weight && shape && (ndim == 1 || ndim == 2), KPD_ERR_INVALID, "bad argument");
    const kpd_egnn_config &c = T->cfg;
    if (c.hidden_nf != HID) {
        // hidden_nf < 256 (the reference's default constructor has 255, models/dynamics.py:300-302): the tensors whose axes carry the
        // hidden width are trained through their 256-wide zero-padded form (train_ops.h, WideSet).  Which axes do: reference shapes
        // models/dynamics.py:37-87, 313-334 -- F = [hidden | timestep] -> 257 columns with the timestep last, P = hidden -> 256.
        const int h = c.hidden_nf;
        const std::vector<AxisSeg> F = {{h, HID}, {1, 1}}, P = {{h, HID}};
        auto cat = [](std::vector<AxisSeg> a, const std::vector<AxisSeg> &b) { a.insert(a.end(), b.begin(), b.end()); return a; };
        auto R = [](int n) { return std::vector<AxisSeg>{{n, n}}; };
        const std::vector<std::string> tk = split_name(name);
        const bool is_w = tk.back() == "weight";
        std::vector<AxisSeg> rows, cols;
        bool touch = true;
        if (tk[0] == "lig_encoder" || tk[0] == "rec_encoder") {
            if (tk.size() < 2 || tk[1] == "0") touch = false;
            else { rows = P; if (is_w) cols = R(tk[0] == "lig_encoder" ? 64 : 2 * c.rec_nf); }
        } else if (tk[0] == "lig_decoder") {
            if (tk.size() >= 2 && tk[1] == "0" && is_w) { rows = R(2 * c.atom_nf); cols = P; }
            else touch = false;
        } else if (tk.size() >= 6 && tk[0] == "egnn") {
            const std::string &blk = tk[3];
            if (blk == "layer_norm") rows = F;
            else if (blk == "node_mlp") { rows = F; if (is_w) cols = tk[5] == "0" ? cat(F, F) : F; }
            else if (blk == "soft_attention") { if (is_w) { rows = R(1); cols = F; } else touch = false; }
            else if (tk[5] == "0") { rows = F; if (is_w) cols = cat(cat(F, F), R(1)); }
            else if (tk[5] == "2") { rows = F; if (is_w) cols = F; }
            else { rows = R(1); cols = F; }                     // coord_mlp.<et>.4.weight
        } else {
            touch = false;
        }
        if (touch) return bind_wide(T, name, weight, grad, shape, ndim, rows, cols);
    }
    Param p;
    p.w = weight;
    p.g = grad;
    p.rows = (int)shape[0];
    p.cols = ndim == 2 ? (int)shape[1] : 1;
    T->params[name] = p;
    return KPD_OK;
}

extern "C" kpd_status kpd_egnn_trainer_reserve(kpd_egnn_trainer *T, int32_t max_B, int32_t max_n_lig, int32_t max_n_kp,
                                               int32_t max_n_kk, int32_t max_lig_pg, int32_t max_kp_pg) {
    KPD_REQUIRE(T, KPD_ERR_INVALID, "null trainer");
    KPD_REQUIRE(max_B >= 1 && max_n_lig >= 1 && max_n_kp >= 1 && max_n_kk >= 0 && max_lig_pg >= 1 && max_kp_pg >= 1, KPD_ERR_INVALID,
                "bad capacities");
    if (max_B <= T->cap_B && max_n_lig <= T->cap_lig && max_n_kp <= T->cap_kp && max_n_kk <= T->cap_kk && max_lig_pg <= T->cap_maxlig &&
        max_kp_pg <= T->cap_maxkp)
        return KPD_OK;
    const kpd_egnn_config &c = T->cfg;
    max_B = std::max(max_B, T->cap_B); max_n_lig = std::max(max_n_lig, T->cap_lig); max_n_kp = std::max(max_n_kp, T->cap_kp);
    max_n_kk = std::max(max_n_kk, T->cap_kk); max_lig_pg = std::max(max_lig_pg, T->cap_maxlig); max_kp_pg = std::max(max_kp_pg, T->cap_maxkp);
    const int cap_ll = std::max<long>((long)max_n_lig * std::min(max_lig_pg - 1, c.ll_k > 0 ? c.ll_k : 200), 1);
    const int cap_kl = std::max<long>((long)max_n_kp * (c.kl_k > 0 ? c.kl_k : std::min(max_lig_pg, 100)), 1);
    const int cap_E = std::max(std::max(cap_ll, cap_kl), std::max<int>(max_n_kk, 1));
    const int cap_N = std::max(max_n_lig, max_n_kp);
    KPD_REQUIRE(((long)cap_N + TM) * NSLOT * HS * 4 < (1l << 32), KPD_ERR_CAPACITY,
                "batch of %d / %d nodes exceeds the 4 GB addressable per node type by the edge kernels' 32-bit row offsets (split the batch)",
                max_n_lig, max_n_kp);
    const int L = c.n_layers;
    const int nn[2] = {max_n_lig, max_n_kp};
    size_t bytes = 0;
    auto add = [&](size_t count, size_t sz) { bytes += (count * sz + 255) & ~size_t(255); };
    for (int nt = 0; nt < 2; ++nt) {
        for (int l = 0; l <= L; ++l) { add((size_t)nn[nt] * LD, 4); add((size_t)nn[nt] * 3, 4); }
        for (int l = 0; l < L; ++l) { add((size_t)nn[nt] * LD, 4); add((size_t)nn[nt] * 3, 4); }
        add(nn[nt], 4); add(max_B, 4); add(nn[nt], 4);
        for (int k = 0; k < 2; ++k) { add((size_t)nn[nt] * LD, 4); add((size_t)nn[nt] * 3, 4); }
        add((size_t)nn[nt] * ENC_LD, 4); add((size_t)nn[nt] * ENC_LD, 4);
    }
    add((size_t)max_n_lig * ENC_LD, 4); add((size_t)max_n_lig * ENC_LD, 4);
    for (int k = 0; k < 6; ++k) add((size_t)cap_E * LD, 4);
    for (int k = 0; k < 7 + 6; ++k) add((size_t)cap_N * LD, 4);          // nb[7] + nbw[2][3]
    add((size_t)cap_N * ENC_LD, 4);
    for (int nt = 0; nt < 2; ++nt) {
        for (int k = 0; k < 3; ++k) add((size_t)nn[nt] * CAT_LD, 4);              // ucat, ducat, dvwcat
        add((size_t)NSLOT * LD * LD, 4); add((size_t)NSLOT * LD * LD, 4);          // wcat, dwcat
        add((size_t)NSLOT * LD, 4); add((size_t)NSLOT * LD, 4); add((size_t)NSLOT * LD, 4);   // dbcat, dwr, bcat
    }
    add(EPACK_FLOATS, 4);
    {
        int tl = 0;
        for (int et = 0; et < 4; ++et) {
            const int cap = et == 0 ? cap_ll : et == 3 ? std::max<int>(max_n_kk, 1) : cap_kl;
            tl += cdiv(cap, TM) + 1;
            for (int k = 0; k < 4; ++k) { add((size_t)nn[kD[et]] * LD, 4); add((size_t)(cdiv(cap, TM) + 1) * LD, 4); }
        }
        add((size_t)tl * 2 * COLSUM_LD, 4); add((size_t)tl * 2 * COLSUM_LD, 4);
    }
    for (int et = 0; et < 4; ++et) {
        const int cap = et == 0 ? cap_ll : et == 3 ? std::max<int>(max_n_kk, 1) : cap_kl;
        add((size_t)nn[kD[et]] * LD, 4); add((size_t)(cdiv(cap, TM) + 1) * LD, 4); add((size_t)nn[kD[et]] * 4, 4); add((size_t)(cdiv(cap, TM) + 1) * 4, 4);
    }
    add(GRAD_PART_FLOATS, 4);
    add(std::max(cap_E, cap_N), 4);                               // ones
    add(32, 4); add(max_n_lig, 4); add(max_B + 1, 4); add(max_B + 1, 4); add(max_B + 2, 4);
    add(cap_ll, 4); add(cap_ll, 4); add(max_n_lig + 1, 4);
    for (int i = 0; i < 4; ++i) add(cap_kl, 4);
    add(max_n_lig + 1, 4); add(max_n_kp + 1, 4); add(max_B, 4); add(8, 4);
    const int cap_et[4] = {cap_ll, cap_kl, cap_kl, std::max<int>(max_n_kk, 1)};
    for (int et = 0; et < 4; ++et) { add(cap_et[et], 4); add(nn[kS[et]] + 1, 4); }
    add(cap_N, 4);
    add(colpart_floats(std::max(cap_E, cap_N)), 4);
    add((size_t)ws_gemm_pack_floats(), 4);
    T->ws.release();
    KPD_TRY(T->ws.reserve(bytes + 4096));
    Arena &W = T->ws;
    for (int nt = 0; nt < 2; ++nt) {
        T->hs[nt].assign(L + 1, nullptr); T->xs[nt].assign(L + 1, nullptr);
        T->hns[nt].assign(L, nullptr); T->xns[nt].assign(L, nullptr);
        for (int l = 0; l <= L; ++l) { T->hs[nt][l] = W.take<float>((size_t)nn[nt] * LD); T->xs[nt][l] = W.take<float>((size_t)nn[nt] * 3); }
        for (int l = 0; l < L; ++l) { T->hns[nt][l] = W.take<float>((size_t)nn[nt] * LD); T->xns[nt][l] = W.take<float>((size_t)nn[nt] * 3); }
        T->bidx[nt] = W.take<int>(nn[nt]);
        T->z[nt] = W.take<float>(max_B);
        T->zinv[nt] = W.take<float>(nn[nt]);
        for (int k = 0; k < 2; ++k) { T->dh[k][nt] = W.take<float>((size_t)nn[nt] * LD); T->dx[k][nt] = W.take<float>((size_t)nn[nt] * 3); }
        T->enc1[nt] = W.take<float>((size_t)nn[nt] * ENC_LD);
        T->enc2[nt] = W.take<float>((size_t)nn[nt] * ENC_LD);
    }
    T->dec1 = W.take<float>((size_t)max_n_lig * ENC_LD);
    T->dec2 = W.take<float>((size_t)max_n_lig * ENC_LD);
    for (int k = 0; k < 7; ++k) T->nb[k] = W.take<float>((size_t)cap_N * LD);
    for (int nt = 0; nt < 2; ++nt)
        for (int k = 0; k < 3; ++k) T->nbw[nt][k] = W.take<float>((size_t)cap_N * LD);
    T->dact = W.take<float>((size_t)cap_N * ENC_LD);
    for (int nt = 0; nt < 2; ++nt) {
        T->ucat[nt] = W.take<float>((size_t)nn[nt] * CAT_LD);
        T->ducat[nt] = W.take<float>((size_t)nn[nt] * CAT_LD);
        T->dvwcat[nt] = W.take<float>((size_t)nn[nt] * CAT_LD);
        T->wcat[nt] = W.take<float>((size_t)NSLOT * LD * LD);
        T->dwcat[nt] = W.take<float>((size_t)NSLOT * LD * LD);
        T->dbcat[nt] = W.take<float>((size_t)NSLOT * LD);
        T->dwr[nt] = W.take<float>((size_t)NSLOT * LD);
        T->bcat[nt] = W.take<float>((size_t)NSLOT * LD);
        // (the padding columns 257 .. 263 of every slot are never written by the segmented sums and are read by the products: zeros)
        KPD_HIP(hipMemset(T->ducat[nt], 0, (size_t)nn[nt] * CAT_LD * 4));
        KPD_HIP(hipMemset(T->dvwcat[nt], 0, (size_t)nn[nt] * CAT_LD * 4));
    }
    T->epack = W.take<float>(EPACK_FLOATS);
    {
        int tl = 0;
        for (int et = 0; et < 4; ++et) {
            const int cap = et == 0 ? cap_ll : et == 3 ? std::max<int>(max_n_kk, 1) : cap_kl;
            tl += cdiv(cap, TM) + 1;
            for (int br = 0; br < 2; ++br) {
                T->dv_main[et][br] = W.take<float>((size_t)nn[kD[et]] * LD); T->dv_cont[et][br] = W.take<float>((size_t)(cdiv(cap, TM) + 1) * LD);
                T->dvw_main[et][br] = W.take<float>((size_t)nn[kD[et]] * LD); T->dvw_cont[et][br] = W.take<float>((size_t)(cdiv(cap, TM) + 1) * LD);
            }
        }
        T->bpart_tiles = tl;
        T->bpart[0] = W.take<float>((size_t)tl * 2 * COLSUM_LD); T->bpart[1] = W.take<float>((size_t)tl * 2 * COLSUM_LD);
    }
    for (int et = 0; et < 4; ++et) {
        const int cap = et == 0 ? cap_ll : et == 3 ? std::max<int>(max_n_kk, 1) : cap_kl;
        T->hn_main[et] = W.take<float>((size_t)nn[kD[et]] * LD); T->hn_cont[et] = W.take<float>((size_t)(cdiv(cap, TM) + 1) * LD);
        T->xn_main[et] = W.take<float>((size_t)nn[kD[et]] * 4); T->xn_cont[et] = W.take<float>((size_t)(cdiv(cap, TM) + 1) * 4);
    }
    T->part_floats = GRAD_PART_FLOATS;
    T->part = W.take<float>(T->part_floats);
    const int n_ones = std::max(cap_E, cap_N);
    T->ones = W.take<float>(n_ones);
    T->meta = W.take<int>(32);
    T->ll_deg = W.take<int>(max_n_lig);
    T->ll_off = W.take<int>(max_B + 1);
    T->kl_off = W.take<int>(max_B + 1);
    T->kl_pg = W.take<int>(max_B + 2);
    kpd_lig_graph &g = T->lg;
    g.cap_ll = cap_ll; g.cap_kl = cap_kl;
    g.ll_src = W.take<int>(cap_ll); g.ll_dst = W.take<int>(cap_ll); g.ll_rowptr = W.take<int>(max_n_lig + 1);
    g.kl_src = W.take<int>(cap_kl); g.kl_dst = W.take<int>(cap_kl); g.kl_rowptr = W.take<int>(max_n_lig + 1);
    g.lk_src = W.take<int>(cap_kl); g.lk_dst = W.take<int>(cap_kl); g.lk_rowptr = W.take<int>(max_n_kp + 1);
    g.ll_per_graph = W.take<int>(max_B);
    g.counts = W.take<int>(8);
    for (int et = 0; et < 4; ++et) { T->scsr[et].perm = W.take<int>(cap_et[et]); T->scsr[et].rowptr = W.take<int>(nn[kS[et]] + 1); }
    T->cursor = W.take<int>(cap_N);
    T->colpart_blocks = cdiv(std::max(cap_E, cap_N), HEAD_ROWS);
    T->colpart = W.take<float>(colpart_floats(std::max(cap_E, cap_N)));
    T->wsg_pack = W.take<float>((size_t)ws_gemm_pack_floats());
    KPD_REQUIRE(T->colpart != nullptr && T->wsg_pack != nullptr, KPD_ERR_HIP, "workspace arena too small (internal sizing error)");
    {
        if (T->store_base) (void)hipFree(T->store_base);
        T->store_base = nullptr;
        T->store = false;
        static const bool want = !(getenv("KPD_TRAIN_STORE") && atoi(getenv("KPD_TRAIN_STORE")) == 0);
        auto al = [](size_t floats) { return (floats * 4 + 255) & ~size_t(255); };
        size_t per_layer = 0;
        // (a2 = SiLU(pre2) is never read from memory -- the backward edge kernel recomputes it from the pre2 rows it streams: three kept
        // arrays per branch, pre1 / a1 / pre2)
        const int n_keep = 6;
        for (int et = 0; et < T->n_et; ++et) per_layer += n_keep * al((size_t)cap_et[et] * LD) + 3 * al(cap_et[et]) + 3 * al((size_t)cap_et[et] * 3);
        for (int nt = 0; nt < T->n_upd; ++nt) per_layer += 3 * al((size_t)nn[nt] * LD);
        // all layers (activations kept: backward recomputes nothing), else one layer's edge slots (the forward edge kernel fills a whole layer
        // at a time; backward recomputes layer by layer).  Less than one layer's slots (3 GB at C2, B = 64) is an out-of-memory error.
        size_t edge_layer = 0;
        for (int et = 0; et < T->n_et; ++et) edge_layer += n_keep * al((size_t)cap_et[et] * LD) + 3 * al(cap_et[et]) + 3 * al((size_t)cap_et[et] * 3);
        int keep_layers = 0;
        if (want && hipMalloc(reinterpret_cast<void **>(&T->store_base), per_layer * L) == hipSuccess) keep_layers = L;
        else {
            (void)hipGetLastError();              // a failed allocation is not an error: recompute instead
            T->store_base = nullptr;
            if (hipMalloc(reinterpret_cast<void **>(&T->store_base), edge_layer) == hipSuccess) keep_layers = 1;
            else { (void)hipGetLastError(); T->store_base = nullptr; }
        }
        KPD_REQUIRE(keep_layers >= 1, KPD_ERR_HIP, "out of device memory: the edge activations of one layer (%zu MB) do not fit", edge_layer >> 20);
        if (T->store_base && poison_level() >= 1) poison_floats(T->store_base, keep_layers == L ? per_layer * L : edge_layer);       // (debug: KPD_POISON)
        T->store = keep_layers == L && want;
        for (int nt = 0; nt < 2; ++nt)
            for (int k = 0; k < 3; ++k) T->nq[nt][k].assign(L, nullptr);
        {
            T->slots.assign((size_t)L * 4, kpd_egnn_trainer::Slot());
            char *p = T->store_base;
            auto take = [&](size_t floats) { float *r = reinterpret_cast<float *>(p); p += al(floats); return r; };
            for (int l = 0; l < (T->store ? L : 1); ++l)
                for (int et = 0; et < T->n_et; ++et) {
                    kpd_egnn_trainer::Slot &sl = T->slots[(size_t)l * 4 + et];
                    for (int br = 0; br < 2; ++br)
                        for (int k = 0; k < 4; ++k) sl.e[br][k] = k == 3 ? nullptr : take((size_t)cap_et[et] * LD);
                    sl.att = take(cap_et[et]); sl.dij = take(cap_et[et]); sl.sc = take(cap_et[et]);
                    sl.xdiff = take((size_t)cap_et[et] * 3); sl.nvec = take((size_t)cap_et[et] * 3); sl.msgx = take((size_t)cap_et[et] * 3);
                }
            if (!T->store)
                for (int l = 1; l < L; ++l)
                    for (int et = 0; et < 4; ++et) T->slots[(size_t)l * 4 + et] = T->slots[et];         // every layer uses the one set
            if (T->store)
                for (int nt = 0; nt < T->n_upd; ++nt)
                    for (int k = 0; k < 3; ++k)
                        for (int l = 0; l < L; ++l) T->nq[nt][k][l] = take((size_t)nn[nt] * LD);
        }
    }
    hipLaunchKernelGGL(k_fill, grid1(n_ones), dim3(256), 0, nullptr, T->ones, 1.0f, (long long)n_ones);
    KPD_LAUNCH_CHECK();
    KPD_HIP(hipDeviceSynchronize());
    T->cap_B = max_B; T->cap_lig = max_n_lig; T->cap_kp = max_n_kp; T->cap_kk = max_n_kk; T->cap_maxlig = max_lig_pg;
    T->cap_maxkp = max_kp_pg; T->cap_ll = cap_ll; T->cap_kl = cap_kl; T->cap_E = cap_E; T->cap_N = cap_N;
    T->have_forward = false;
    return KPD_OK;
}

namespace {

kpd_status encoders_fwd(kpd_egnn_trainer *T) {
    const kpd_egnn_config &c = T->cfg;
    MlpParams p;
    KPD_TRY(mlp_params(T, "lig_encoder", c.atom_nf, 64, 256, &p));
    KPD_TRY(mlp_fwd(T, p, T->bt.lig_h, c.atom_nf, T->n[0], T->enc1[0], T->enc2[0], ENC_LD, T->nb[0], LD, T->hs[0][0], LD, true));
    if (T->rec_identity) {
        const long long tot = (long long)T->n[1] * c.hidden_nf;
        if (c.hidden_nf != HID) KPD_HIP(hipMemsetAsync(T->hs[1][0], 0, (size_t)T->n[1] * LD * 4, T->st));       // padding columns are zeros
        hipLaunchKernelGGL(k_copy_rows, grid1(tot), dim3(256), 0, T->st, T->bt.kp_h, c.hidden_nf, T->hs[1][0], LD, tot, c.hidden_nf);
        KPD_LAUNCH_CHECK();
    } else {
        KPD_TRY(mlp_params(T, "rec_encoder", c.rec_nf, 2 * c.rec_nf, 256, &p));
        KPD_TRY(mlp_fwd(T, p, T->bt.kp_h, c.rec_nf, T->n[1], T->enc1[1], T->enc2[1], ENC_LD, T->nb[1], LD, T->hs[1][0], LD, true));
    }
    for (int nt = 0; nt < 2; ++nt) {
        hipLaunchKernelGGL(k_set_time, grid1(T->n[nt]), dim3(256), 0, T->st, T->hs[nt][0], T->t_dev, T->bidx[nt], T->n[nt]);
        KPD_LAUNCH_CHECK();
    }
    return KPD_OK;
}

}  // namespace

extern "C" kpd_status kpd_egnn_trainer_forward(kpd_egnn_trainer *T, const kpd_batch *bt, const float *t_dev, float *eps_h,
                                               float *eps_x, void *stream) {
    KPD_REQUIRE(T && bt && t_dev && eps_h && eps_x, KPD_ERR_INVALID, "null argument");
    KPD_REQUIRE(bt->B >= 1 && bt->n_lig >= 1 && bt->n_kp >= 1, KPD_ERR_INVALID, "empty batch");
    KPD_REQUIRE(bt->B <= T->cap_B && bt->n_lig <= T->cap_lig && bt->n_kp <= T->cap_kp && bt->n_kk <= T->cap_kk &&
                    bt->max_lig <= T->cap_maxlig && bt->max_kp <= T->cap_maxkp,
                KPD_ERR_CAPACITY, "batch exceeds the reserved workspace");
    KPD_REQUIRE(bt->kk_rowptr && (!T->cfg.update_kp_feat || bt->n_kk == 0 || (bt->kk_src && bt->kk_dst)), KPD_ERR_INVALID, "kk edges missing");
    const kpd_egnn_config &c = T->cfg;
    hipStream_t st = static_cast<hipStream_t>(stream);
    T->st = st;
    KPD_TRY(wide_run(T, 0));                   // hidden_nf < 256: stage the current weights in the engine's widths
    T->bt = *bt;
    T->t_dev = t_dev;
    T->n[0] = bt->n_lig; T->n[1] = bt->n_kp;
    const int L = c.n_layers;
    KPD_HIP(hipMemcpyAsync(T->xs[0][0], bt->lig_x, (size_t)bt->n_lig * 12, hipMemcpyDeviceToDevice, st));
    KPD_HIP(hipMemcpyAsync(T->xs[1][0], bt->kp_x, (size_t)bt->n_kp * 12, hipMemcpyDeviceToDevice, st));
    KPD_TRY(launch_node_graph_index(bt->lig_ptr, bt->B, bt->n_lig, T->bidx[0], st));
    KPD_TRY(launch_node_graph_index(bt->kp_ptr, bt->B, bt->n_kp, T->bidx[1], st));
    KPD_TRY(launch_lig_graph(bt, c.ll_cutoff, c.ll_k, c.kl_cutoff, c.kl_k, &T->lg, T->ll_deg, T->ll_off, T->kl_off, T->kl_pg, st));
    const int active = c.update_kp_feat ? 0xF : 0x3;
    KPD_TRY(launch_egnn_meta(T->lg.counts, bt->n_kk, active, 0x3, bt->lig_ptr, bt->kp_ptr, T->lg.ll_per_graph, bt->kk_rowptr, bt->B, T->kl_off,
                             c.message_norm, c.update_kp_feat, T->meta, T->z[0], T->z[1], st));
    // edge counts drive GEMM shapes: one read-back per training step
    int counts[2];
    KPD_HIP(hipMemcpyAsync(counts, T->lg.counts, sizeof(counts), hipMemcpyDeviceToHost, st));
    KPD_HIP(hipStreamSynchronize(st));
    KPD_REQUIRE(counts[0] <= T->cap_ll && counts[1] <= T->cap_kl, KPD_ERR_CAPACITY, "edge lists overflow (ll %d/%d, kl %d/%d)", counts[0],
                T->cap_ll, counts[1], T->cap_kl);
    T->E[ET_LL] = counts[0]; T->E[ET_KL] = counts[1]; T->E[ET_LK] = counts[1]; T->E[ET_KK] = bt->n_kk;
    T->e_src[ET_LL] = T->lg.ll_src; T->e_dst[ET_LL] = T->lg.ll_dst; T->e_rowptr[ET_LL] = T->lg.ll_rowptr;
    T->e_src[ET_KL] = T->lg.kl_src; T->e_dst[ET_KL] = T->lg.kl_dst; T->e_rowptr[ET_KL] = T->lg.kl_rowptr;
    T->e_src[ET_LK] = T->lg.lk_src; T->e_dst[ET_LK] = T->lg.lk_dst; T->e_rowptr[ET_LK] = T->lg.lk_rowptr;
    T->e_src[ET_KK] = bt->kk_src; T->e_dst[ET_KK] = bt->kk_dst; T->e_rowptr[ET_KK] = bt->kk_rowptr;
    for (int nt = 0; nt < 2; ++nt) {
        hipLaunchKernelGGL(k_zinv, grid1(T->n[nt]), dim3(256), 0, st, T->z[nt], T->bidx[nt], T->n[nt], T->zinv[nt]);
        KPD_LAUNCH_CHECK();
    }
    for (int et = 0; et < T->n_et; ++et) KPD_TRY(build_src_csr(T, T->e_src[et], T->E[et], T->n[kS[et]], T->cursor, T->scsr[et]));
    KPD_TRY(encoders_fwd(T));
    for (int l = 0; l < L; ++l) {
        KPD_TRY(layer_fwd(T, l));
        KPD_TRY(nodes_fwd(T, l));
    }
    // decoder on the first 256 columns (dynamics.py:376-381); eps_x = x_out - x_0
    MlpParams p;
    KPD_TRY(mlp_params(T, "lig_decoder", 256, 2 * c.atom_nf, c.atom_nf, &p));
    KPD_TRY(mlp_fwd(T, p, T->hs[0][L], LD, T->n[0], T->dec1, T->dec2, ENC_LD, T->nb[0], LD, eps_h, c.atom_nf, false));
    KPD_HIP(hipMemcpyAsync(eps_x, T->xs[0][L], (size_t)T->n[0] * 12, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_sub_inplace, grid1(3 * T->n[0]), dim3(256), 0, st, eps_x, T->xs[0][0], 3 * T->n[0]);
    KPD_LAUNCH_CHECK();
    T->have_forward = true;
    return KPD_OK;
}

namespace {

// backward of the node update of layer l for node type nt: consumes dh_out / dx_out (T->dh[cur], T->dx[cur]), writes the
// node part of dh_in / dx_in (T->dh[nxt], "="), leaves nb[5] = dL/d(h_neigh / z) rows (un-scaled by zinv) for the heads
kpd_status node_bwd(kpd_egnn_trainer *T, int l, int nt, int cur, int nxt, float *dhn_out) {
    const int n = T->n[nt];
    NodeParams p;
    KPD_TRY(node_params(T, l, nt, &p));
    const NodeAct na = node_act(T, l, nt);
    if (!T->store) KPD_TRY(node_mlp_fwd(T, p, l, nt, na));          // (kept otherwise)
    float *du = T->nbw[nt][0], *tmp = T->nbw[nt][1];
    const float *dy = T->dh[cur][nt];
    // The three 257 x 257 weight gradients of this node MLP wait for the other node type's and go out as ONE launch (layer_bwd): as split-K
    // products of their own (K = node count, nine 128 x 128 output tiles each) they ran at a quarter of the MFMA peak.  Their operands are kept
    // activations and the per-type buffers above (nbw), in both memory modes: the two modes stay bit-identical.
    const bool batch = p.W2.g && p.b2.g && p.W1.g && p.b1.g;
    if (T->cfg.norm) {
        const int blocks = cdiv(n, LNP_ROWS);
        if (p.gamma.g && p.beta.g && T->colpart && blocks <= T->colpart_blocks) {
            hipLaunchKernelGGL(k_ln_bwd_p, dim3(blocks), dim3(256), 0, T->st, T->hs[nt][l], na.q[2], p.b2.w, p.gamma.w, dy, n, T->cfg.hidden_nf, du, T->colpart);
            KPD_LAUNCH_CHECK();
            hipLaunchKernelGGL(k_colsum_reduce, dim3(cdiv(H, 64)), dim3(1024), 0, T->st, T->colpart, blocks, H, p.gamma.g, 1, p.beta.g);
            KPD_LAUNCH_CHECK();
        } else {
            hipLaunchKernelGGL(k_ln_bwd, dim3(cdiv(n, 4)), dim3(256), 0, T->st, T->hs[nt][l], na.q[2], p.b2.w, p.gamma.w, dy, n, T->cfg.hidden_nf, du, tmp);
            KPD_LAUNCH_CHECK();
            KPD_TRY(colsum_acc(T, n, H, tmp, LD, p.gamma.g));
            KPD_TRY(colsum_acc(T, n, H, dy, LD, p.beta.g));
        }
    } else {
        KPD_HIP(hipMemcpyAsync(du, dy, (size_t)n * LD * 4, hipMemcpyDeviceToDevice, T->st));
    }
    if (batch) T->wq_nodes.push_back(Grad257Item{du, na.q[1], LD, LD, n, p.W2.g, H, p.b2.g});
    else KPD_TRY(grad_gemm(T, H, H, n, du, LD, na.q[1], LD, p.W2.g, H, p.b2.g));
    float *dq1 = tmp;
    const bool ws = node_ws(T, n);
    if (ws) KPD_TRY(ws_gemm(WS_SILU_BWD, du, n, LD, p.W2.w, H, true, nullptr, na.q[0], dq1, nullptr, LD, T->wsg_pack, T->st));
    else KPD_TRY(gemm(T, false, false, n, H, H, du, LD, p.W2.w, H, 0.0f, dq1, LD, 1.0f, na.q[0]));                    // * SiLU'(pre) in the epilogue
    if (batch) {
        T->wq_nodes.push_back(Grad257Item{dq1, T->hs[nt][l], LD, LD, n, p.W1.g, 2 * H, p.b1.g});
        T->wq_nodes.push_back(Grad257Item{dq1, T->hns[nt][l], LD, LD, n, p.W1.g + H, 2 * H, nullptr});
    } else {
        KPD_TRY(grad_gemm(T, H, H, n, dq1, LD, T->hs[nt][l], LD, p.W1.g, 2 * H, p.b1.g));
        if (p.W1.g) KPD_TRY(grad_gemm(T, H, H, n, dq1, LD, T->hns[nt][l], LD, p.W1.g + H, 2 * H));
    }
    // dh_in = du (residual) + dq1 W1[:, :257];  d(h_neigh / z) = dq1 W1[:, 257:]
    KPD_HIP(hipMemcpyAsync(T->dh[nxt][nt], du, (size_t)n * LD * 4, hipMemcpyDeviceToDevice, T->st));
    if (ws) {
        KPD_TRY(ws_gemm(WS_PLAIN, dq1, n, LD, p.W1.w, 2 * H, true, nullptr, nullptr, T->dh[nxt][nt], nullptr, LD, T->wsg_pack, T->st, true, true));
        KPD_TRY(ws_gemm(WS_PLAIN, dq1, n, LD, p.W1.w + H, 2 * H, true, nullptr, nullptr, dhn_out, nullptr, LD, T->wsg_pack, T->st));
    } else {
        KPD_TRY(gemm(T, false, false, n, H, H, dq1, LD, p.W1.w, 2 * H, 1.0f, T->dh[nxt][nt], LD));
        KPD_TRY(gemm(T, false, false, n, H, H, dq1, LD, p.W1.w + H, 2 * H, 0.0f, dhn_out, LD));
    }
    KPD_HIP(hipMemcpyAsync(T->dx[nxt][nt], T->dx[cur][nt], (size_t)n * 12, hipMemcpyDeviceToDevice, T->st));   // x' = x + x_neigh
    return KPD_OK;
}

// the edge part of layer l's backward pass through k_egnn_edge_bwd: one launch for the head backward, the dpre2 W2 product, dpre1, d dij and
// the by-destination sums of every (edge type, branch); then, per (edge type, branch), what needs whole matrices: dW2 = dpre2^T a1, the
// by-source sums of dpre1, the reductions of the per-tile partials
kpd_status layer_edges_bwd_fused(kpd_egnn_trainer *T, int l, int cur, int nxt, float *dhn[2]) {
    const kpd_egnn_config &c = T->cfg;
    EdgePackTab pk;
    pk.n = 0;
    pk.transposed = 1;
    EdgeBwdArgs a{};
    a.meta = T->meta + (l == c.n_layers - 1 ? 16 : 0);
    a.use_tanh = c.use_tanh; a.coords_range = c.coords_range;
    a.part[0] = T->bpart[0]; a.part[1] = T->bpart[1]; a.part_ld = COLSUM_LD;
    a.stamps = T->stamps;
    a.skip = tool_env_int("KPD_TR_SKIP", 0);
    for (int nt = 0; nt < 2; ++nt) { a.dhn[nt] = dhn[nt]; a.dxo[nt] = T->dx[cur][nt]; a.zinv[nt] = T->zinv[nt]; }
    int tiles = 0, tile0[4] = {0, 0, 0, 0};
    for (int et = 0; et < 4; ++et) {
        a.dst[et] = T->e_dst[et]; a.dst_nt[et] = kD[et];
        if (et >= layer_n_et(T, l) || T->E[et] == 0) continue;
        tile0[et] = tiles;
        tiles += cdiv(T->E[et], TM);
        const kpd_egnn_trainer::Slot &sl = T->slots[(size_t)l * 4 + et];
        float *heads = T->epack + 8 * EPACK_ENTRY + (size_t)et * 2 * HS;
        a.wa[et] = heads; a.w3[et] = heads + HS;
        for (int br = 0; br < 2; ++br) {
            BranchParams p;
            KPD_TRY(branch_params(T, l, et, br, &p));
            float *base = T->epack + (size_t)(et * 2 + br) * EPACK_ENTRY;
            EdgePackEntry &e = pk.e[pk.n++];
            e.W1 = p.W1.w; e.W2 = p.W2.w; e.b2 = p.b2.w; e.head = p.head.w; e.head_b = br == 0 ? p.head_b.w : nullptr;
            e.wp = base; e.wx = base + WP_FLOATS; e.wr = base + WP_FLOATS + HS; e.head_out = heads + (size_t)br * HS;
            a.wpT[et][br] = e.wp; a.wxT[et][br] = e.wx; a.wr[et][br] = e.wr;
            for (int k = 0; k < 4; ++k) a.keep[et][br][k] = sl.e[br][k];
            a.dv_main[et][br] = T->dv_main[et][br]; a.dv_cont[et][br] = T->dv_cont[et][br];
            a.dvw_main[et][br] = T->dvw_main[et][br]; a.dvw_cont[et][br] = T->dvw_cont[et][br];
        }
        a.att[et] = sl.att; a.sc[et] = sl.sc; a.dij[et] = sl.dij; a.nvec[et] = sl.nvec;
    }
    if (tiles == 0) return KPD_OK;
    KPD_REQUIRE(tiles <= T->bpart_tiles, KPD_ERR_CAPACITY, "per-tile partial sums: %d tiles, room for %d", tiles, T->bpart_tiles);
    KPD_TRY(launch_edge_train_pack(pk, T->st));
    {
        double edges = 0.0;
        for (int et = 0; et < layer_n_et(T, l); ++et) edges += T->E[et];
        KPD_TRY(T->timed(1, edges, [&] { return launch_egnn_edge_bwd(a, tiles, T->st); }));
    }
    std::vector<Grad257Item> wq;
    ColsumRedBatch crb;
    memset(&crb, 0, sizeof(crb));
    DxGatherArgs dxa;
    memset(&dxa, 0, sizeof(dxa));
    EdgePiecesBatch epb;
    memset(&epb, 0, sizeof(epb));
    epb.ldo = CAT_LD;
    int n_crb = 0, n_epb = 0, n_ssb = 0, ssb_rows = 0;
    SegsumSrcBatch ssb;
    memset(&ssb, 0, sizeof(ssb));
    ssb.ldo = CAT_LD;
    for (int et = 0; et < layer_n_et(T, l); ++et) {
        const int E = T->E[et], s = kS[et], d = kD[et];
        if (E == 0) continue;
        const kpd_egnn_trainer::Slot &sl = T->slots[(size_t)l * 4 + et];
        for (int br = 0; br < 2; ++br) {
            BranchParams p;
            KPD_TRY(branch_params(T, l, et, br, &p));
            float *dpre1 = sl.e[br][0], *a1 = sl.e[br][1], *dpre2 = sl.e[br][2];
            crb.r[n_crb++] = ColsumRedBatch::One{T->bpart[br] + (size_t)tile0[et] * 2 * COLSUM_LD, cdiv(E, TM), H, 1, p.head.g, p.b2.g};       // (one launch below)
            if (br == 0 && p.head_b.g) KPD_TRY(sum_scalar(T, sl.att, E, p.head_b.g));           // (ds of the attention logits, left over att)
            if (p.W2.g) {       // dW2 += dpre2^T a1: with the layer's other edge-sized products in one launch below
                wq.push_back(Grad257Item{dpre2, a1, LD, LD, E, p.W2.g, H, nullptr});
            }
            float *dU = T->ducat[s] + (size_t)T->cat_of[et][br][0] * LD, *dV = T->ducat[d] + (size_t)T->cat_of[et][br][1] * LD;
            float *dVw = T->dvwcat[d] + (size_t)T->cat_dvw_of[et][br] * LD;
            ssb.e[n_ssb++] = SegsumSrcBatch::One{dpre1, T->scsr[et].perm, T->scsr[et].rowptr, T->n[s], dU};          // (one launch below)
            ssb_rows = std::max(ssb_rows, T->n[s]);
            epb.e[n_epb++] = EdgePiecesBatch::One{T->dv_main[et][br], T->dv_cont[et][br], T->dvw_main[et][br], T->dvw_cont[et][br], T->e_rowptr[et], T->n[d], dV,
                                                  p.W1.g ? dVw : nullptr};
        }
        float *redge = sl.msgx;
        hipLaunchKernelGGL(k_geom_bwd, grid1(E), dim3(256), 0, T->st, sl.sc, sl.nvec, sl.xdiff, sl.dij, E, redge);       // (d dij over sc, dn over nvec)
        KPD_LAUNCH_CHECK();
        dxa.redge[et] = redge; dxa.perm[et] = T->scsr[et].perm; dxa.srowptr[et] = T->scsr[et].rowptr; dxa.drowptr[et] = T->e_rowptr[et];
        dxa.live[et] = 1; dxa.src_nt[et] = s; dxa.dst_nt[et] = d;
    }
    {   // position gradients of both node types from every edge type's per-edge gradients: one launch
        for (int nt = 0; nt < 2; ++nt) { dxa.n[nt] = T->n[nt]; dxa.dx[nt] = T->dx[nxt][nt]; }
        hipLaunchKernelGGL(k_dx_gather, dim3(cdiv(std::max(T->n[0], T->n[1]), 16), 2), dim3(256), 0, T->st, dxa);
        KPD_LAUNCH_CHECK();
    }
    // the second-Linear weight gradients of every (edge type, branch) of the layer: one launch, a share of the CUs per product proportional
    // to its edge count (sgemm.hip, grad257_batch) instead of a launch, 256 partial tiles and a reduction each
    if (n_ssb) {          // the by-source gradient blocks of every (edge type, branch): one launch
        hipLaunchKernelGGL(k_segsum264, dim3(cdiv(ssb_rows, 4), n_ssb), dim3(256), 0, T->st, ssb);
        KPD_LAUNCH_CHECK();
    }
    KPD_TRY(launch_edge_pieces_set(epb, n_epb, T->st));          // the by-destination gradient blocks of every (edge type, branch): one launch
    if (n_crb) {          // head and second-bias gradients of every (edge type, branch): the per-tile partials of k_egnn_edge_bwd, summed in one launch
        hipLaunchKernelGGL(k_colsum_reduce_batch, dim3(cdiv(H, 64), n_crb), dim3(1024), 0, T->st, crb);
        KPD_LAUNCH_CHECK();
    }
    for (size_t i = 0; i < wq.size(); i += 8) KPD_TRY(grad257_batch(wq.data() + i, (int)std::min<size_t>(8, wq.size() - i), T->part, T->part_floats, T->st));
    return KPD_OK;
}

kpd_status layer_bwd(kpd_egnn_trainer *T, int l, int cur, int nxt, float *dhn[2]) {
    // (final layer, keypoints: dh_out = dx_out = 0, so dh_in / dx_in start from the zeros the caller left in dh[nxt] / dx[nxt])
    T->wq_nodes.clear();
    for (int nt = 0; nt < layer_n_upd(T, l); ++nt) KPD_TRY(node_bwd(T, l, nt, cur, nxt, dhn[nt]));
    KPD_TRY(grad257_batch(T->wq_nodes.data(), (int)T->wq_nodes.size(), T->part, T->part_floats, T->st));
    KPD_TRY(layer_stage(T, l));
    if (!T->store) KPD_TRY(layer_project(T, l));
    if (!T->store) KPD_TRY(layer_edges_fused(T, l, false));          // recompute mode: refill this layer's slots
    KPD_TRY(layer_edges_bwd_fused(T, l, cur, nxt, dhn));
    return layer_cat_bwd(T, l, nxt);
}

}  // namespace

extern "C" kpd_status kpd_egnn_trainer_backward(kpd_egnn_trainer *T, const float *d_eps_h, const float *d_eps_x, float *d_lig_h,
                                                float *d_lig_x, float *d_kp_h, float *d_kp_x, void *stream) {
    KPD_REQUIRE(T && d_eps_h && d_eps_x, KPD_ERR_INVALID, "null argument");
    KPD_REQUIRE(T->have_forward, KPD_ERR_STATE, "kpd_egnn_trainer_backward before kpd_egnn_trainer_forward");
    const kpd_egnn_config &c = T->cfg;
    hipStream_t st = static_cast<hipStream_t>(stream);
    T->st = st;
    KPD_TRY(wide_run(T, 1));                   // hidden_nf < 256: zero the wide gradients
    const int L = c.n_layers, nl = T->n[0], nk = T->n[1];
    int cur = 0, nxt = 1;
    if (T->n_upd == 1) {        // kp not updated: one gradient accumulator across layers (the same tensors feed every layer)
        T->dh[1][1] = T->dh[0][1];
        T->dx[1][1] = T->dx[0][1];
    }
    // decoder backward -> dh_out[lig][:, :256]; dh_out[kp] = 0; dx_out[lig] = d_eps_x; dx_out[kp] = 0
    for (int k = 0; k < 2; ++k)
        for (int nt = 0; nt < 2; ++nt) {
            KPD_HIP(hipMemsetAsync(T->dh[k][nt], 0, (size_t)T->n[nt] * LD * 4, st));
            KPD_HIP(hipMemsetAsync(T->dx[k][nt], 0, (size_t)T->n[nt] * 12, st));
        }
    {
        MlpParams p;
        KPD_TRY(mlp_params(T, "lig_decoder", 256, 2 * c.atom_nf, c.atom_nf, &p));
        // recompute the decoder activations (scratch may have been reused) and run its backward
        float *dout = T->nb[1];
        KPD_TRY(mlp_fwd(T, p, T->hs[0][L], LD, nl, T->dec1, T->dec2, ENC_LD, T->nb[0], LD, dout, LD, false));
        const long long tot = (long long)nl * c.atom_nf;
        hipLaunchKernelGGL(k_copy_rows, grid1(tot), dim3(256), 0, st, d_eps_h, c.atom_nf, dout, LD, tot, c.atom_nf);
        KPD_LAUNCH_CHECK();
        KPD_TRY(mlp_bwd(T, p, T->hs[0][L], LD, nl, T->dec1, T->dec2, ENC_LD, T->nb[0], LD, dout, LD, false, T->dact, T->dh[cur][0], LD));
        KPD_HIP(hipMemcpyAsync(T->dx[cur][0], d_eps_x, (size_t)nl * 12, hipMemcpyDeviceToDevice, st));
    }
    float *dhn[2] = {T->nb[5], T->nb[6]};
    for (int l = L - 1; l >= 0; --l) {
        KPD_TRY(layer_bwd(T, l, cur, nxt, dhn));
        std::swap(cur, nxt);
    }
    // encoders (the timestep column carries no parameter gradient)
    {
        MlpParams p;
        KPD_TRY(mlp_params(T, "lig_encoder", c.atom_nf, 64, 256, &p));
        KPD_TRY(mlp_fwd(T, p, T->bt.lig_h, c.atom_nf, nl, T->enc1[0], T->enc2[0], ENC_LD, T->nb[0], LD, T->nb[1], LD, true));
        KPD_TRY(mlp_bwd(T, p, T->bt.lig_h, c.atom_nf, nl, T->enc1[0], T->enc2[0], ENC_LD, T->nb[0], LD, T->dh[cur][0], LD, true, T->dact,
                        d_lig_h, c.atom_nf));
        if (T->rec_identity) {
            if (d_kp_h) {
                const long long tot = (long long)nk * c.hidden_nf;
                hipLaunchKernelGGL(k_copy_rows, grid1(tot), dim3(256), 0, st, T->dh[cur][1], LD, d_kp_h, c.hidden_nf, tot, c.hidden_nf);
                KPD_LAUNCH_CHECK();
            }
        } else {
            KPD_TRY(mlp_params(T, "rec_encoder", c.rec_nf, 2 * c.rec_nf, 256, &p));
            KPD_TRY(mlp_fwd(T, p, T->bt.kp_h, c.rec_nf, nk, T->enc1[1], T->enc2[1], ENC_LD, T->nb[0], LD, T->nb[1], LD, true));
            KPD_TRY(mlp_bwd(T, p, T->bt.kp_h, c.rec_nf, nk, T->enc1[1], T->enc2[1], ENC_LD, T->nb[0], LD, T->dh[cur][1], LD, true, T->dact,
                            d_kp_h, c.rec_nf));
        }
    }
    // eps_x = x_out - x_0: the direct term
    if (d_lig_x) {
        KPD_HIP(hipMemcpyAsync(d_lig_x, T->dx[cur][0], (size_t)nl * 12, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(k_sub_inplace, grid1(3 * nl), dim3(256), 0, st, d_lig_x, d_eps_x, 3 * nl);
        KPD_LAUNCH_CHECK();
    }
    if (d_kp_x) KPD_HIP(hipMemcpyAsync(d_kp_x, T->dx[cur][1], (size_t)nk * 12, hipMemcpyDeviceToDevice, st));
    KPD_TRY(wide_run(T, 2));                   // hidden_nf < 256: add the wide gradients into the caller's tensors (reference shapes)
    T->have_forward = false;
    return KPD_OK;
}
