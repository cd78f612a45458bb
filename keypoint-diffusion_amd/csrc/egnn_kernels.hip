// EGNN denoiser kernels (LigRecDynamics, models/dynamics.py:9-385) for gfx950.
//
// Per layer (LigRecConv.forward, dynamics.py:124-207) three kernels run:
//   k_proj_ws     (egnn_chain.hip) P[node][slot] = W1[:, h-part] . h[node] (+ b1 on dst slots)
//                 -- the first Linear(515, 257) of edge_mlp / coord_mlp is linear in
//                    [h_src, h_dst, d_ij], so its two 257-wide blocks are applied once per
//                    NODE instead of once per EDGE (3x fewer edge FLOPs);
//   k_egnn_edge   per tile of 64 same-type edges (dst-sorted): gather P_src + P_dst + d*w_r,
//                 SiLU, the 257x257 second Linear on fp32 MFMA, SiLU, soft attention,
//                 coordinate head, and the segmented sum over destination nodes -- all in
//                 one workgroup, intermediates never leave LDS/registers;
//   k_node_update8  h' = LN(h + node_mlp([h, h_neigh / z])), x' = x + x_neigh / z.
// Segment pieces: a tile writes the sum of each run of equal dst either to main[dst]
// (run starts the segment) or to cont[tile] (run continues a segment begun in an earlier
// tile); k_node_update8 adds main + cont pieces in tile order => deterministic, no atomics.
#include <stdlib.h>

#include <algorithm>

#include "egnn_kernels.h"
#include "mfma_core.h"

namespace kpd {

// ---- small helpers ------------------------------------------------------------------------
__global__ void k_node_graph_index(const int *__restrict__ ptr, int B, int n, int *__restrict__ bidx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int lo = 0, hi = B;                       // largest b with ptr[b] <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (ptr[mid] <= i) lo = mid; else hi = mid;
    }
    bidx[i] = lo;
}

// meta layout: [0..3] E per etype (ll, kl, lk, kk), [4..8] first tile of each etype (+total); the same nine
// entries at [16..24] for the final layer's edge-type subset (active_last).
// z: per-graph message normaliser (dynamics.py:277-285) -- always over ALL active edge types: the pruned
// final layer drops messages nobody reads, not terms of a sum somebody does.
__global__ void k_egnn_meta(const int *__restrict__ counts, int e_kk, int active_mask, int active_last, const int *__restrict__ lig_ptr,
                            const int *__restrict__ kp_ptr, const int *__restrict__ ll_per_graph,
                            const int *__restrict__ kk_rowptr, int B, const int *__restrict__ kl_off, float message_norm,
                            int update_kp, int tile_rows, int *__restrict__ meta, float *__restrict__ z_lig, float *__restrict__ z_kp) {
    if (blockIdx.x == 0 && threadIdx.x < 2) {
        const int mask = threadIdx.x ? active_last : active_mask;
        int *mt = meta + 16 * threadIdx.x;
        int E[4] = {counts[0], counts[1], counts[1], e_kk};
        int run = 0;
        for (int et = 0; et < 4; ++et) {
            if (!((mask >> et) & 1)) E[et] = 0;
            mt[et] = E[et];
            mt[4 + et] = run;
            run += (E[et] + tile_rows - 1) / tile_rows;
        }
        mt[8] = run;
    }
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
        const int nl = lig_ptr[b + 1] - lig_ptr[b], nk = kp_ptr[b + 1] - kp_ptr[b];
        if (message_norm == 0.0f) {
            const int e_kl = kl_off[b + 1] - kl_off[b];          // per-complex kl edges (kNN: nk min(k, nl); radius: counted)
            const int e_kk_b = kk_rowptr[kp_ptr[b + 1]] - kk_rowptr[kp_ptr[b]];
            z_lig[b] = (float)(ll_per_graph[b] + e_kl) / (float)nl + 1.0f;
            z_kp[b] = update_kp ? (float)(e_kl + e_kk_b) / (float)nk + 1.0f : 1.0f;
        } else {
            z_lig[b] = message_norm;
            z_kp[b] = message_norm;
        }
    }
}

// ---- encoders (dynamics.py:313-318, 326-334, 355-363) -------------------------------------
// out[node][0..255] = SiLU(W1 SiLU(W0 in + b0) + b1), out[node][256] = t[graph], pads 0.
// W1t is stored transposed [hid][256] so that consecutive threads read consecutive floats.
constexpr int EMB_NODES = 8;
__global__ __launch_bounds__(256) void k_embed(const float *__restrict__ in, int n, int fin,
                                               const float *__restrict__ W0, const float *__restrict__ b0, int hid,
                                               const float *__restrict__ W1t, const float *__restrict__ b1,
                                               const float *__restrict__ t, const int *__restrict__ bidx,
                                               float *__restrict__ out, int identity) {
    __shared__ float s_in[EMB_NODES][256];
    __shared__ float s_hid[EMB_NODES][256];
    const int node0 = blockIdx.x * EMB_NODES, tid = threadIdx.x;
    for (int i = tid; i < EMB_NODES * fin; i += 256) {
        const int j = i / fin, k = i - j * fin;
        s_in[j][k] = node0 + j < n ? in[(size_t)(node0 + j) * fin + k] : 0.0f;
    }
    __syncthreads();
    if (!identity) {
        for (int u = tid; u < hid; u += 256) {
            float a[EMB_NODES];
#pragma unroll
            for (int j = 0; j < EMB_NODES; ++j) a[j] = b0[u];
            for (int k = 0; k < fin; ++k) {
                const float w = W0[(size_t)u * fin + k];
#pragma unroll
                for (int j = 0; j < EMB_NODES; ++j) a[j] = fmaf(w, s_in[j][k], a[j]);
            }
#pragma unroll
            for (int j = 0; j < EMB_NODES; ++j) s_hid[j][u] = silu(a[j]);
        }
        __syncthreads();
        float a[EMB_NODES];
#pragma unroll
        for (int j = 0; j < EMB_NODES; ++j) a[j] = b1[tid];
        for (int u = 0; u < hid; ++u) {
            const float w = W1t[(size_t)u * 256 + tid];
#pragma unroll
            for (int j = 0; j < EMB_NODES; ++j) a[j] = fmaf(w, s_hid[j][u], a[j]);
        }
#pragma unroll
        for (int j = 0; j < EMB_NODES; ++j)
            if (node0 + j < n) out[(size_t)(node0 + j) * HS + tid] = silu(a[j]);
    } else {
#pragma unroll
        for (int j = 0; j < EMB_NODES; ++j)
            if (node0 + j < n) out[(size_t)(node0 + j) * HS + tid] = s_in[j][tid];
    }
    if (tid < EMB_NODES * 8) {
        const int j = tid >> 3, c = tid & 7;
        if (node0 + j < n) out[(size_t)(node0 + j) * HS + 256 + c] = c == 0 ? t[bidx[node0 + j]] : 0.0f;
    }
}

// ---- decoder (dynamics.py:320-324, 376-381) -----------------------------------------------
// One wave per ligand atom.  eps_h = W1 SiLU(W0 h[:256] + b0) + b1;  eps_x = x - x_0.
__global__ __launch_bounds__(64) void k_decode(const float *__restrict__ h, const float *__restrict__ x,
                                               const float *__restrict__ x0, int n, int atom_nf, int hid,
                                               const float *__restrict__ W0, const float *__restrict__ b0,
                                               const float *__restrict__ W1, const float *__restrict__ b1,
                                               float *__restrict__ eps_h, float *__restrict__ eps_x) {
    __shared__ float s_hid[64];
    const int v = blockIdx.x, lane = threadIdx.x;
    if (v >= n) return;
    const f32x4 hv = *reinterpret_cast<const f32x4 *>(h + (size_t)v * HS + 4 * lane);
    // four hidden units at a time: their weight rows are requested together and their cross-lane reductions interleave (one unit at
    // a time was a chain of `hid` dependent load + 6-shuffle sequences: 16.9 us for 1 600 atoms, all of it latency)
    for (int u0 = 0; u0 < hid; u0 += 4) {
        float s[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u = min(u0 + i, hid - 1);
            const f32x4 w = *reinterpret_cast<const f32x4 *>(W0 + (size_t)u * 256 + 4 * lane);
            s[i] = hv[0] * w[0] + hv[1] * w[1] + hv[2] * w[2] + hv[3] * w[3];
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] += __shfl_xor(s[i], o);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (lane == 0 && u0 + i < hid) s_hid[u0 + i] = silu(s[i] + b0[u0 + i]);
    }
    __syncthreads();
    if (lane < atom_nf) {
        float s = b1[lane];
        for (int u = 0; u < hid; ++u) s = fmaf(W1[(size_t)lane * hid + u], s_hid[u], s);
        eps_h[(size_t)v * atom_nf + lane] = s;
    }
    if (lane < 3) eps_x[(size_t)v * 3 + lane] = x[(size_t)v * 3 + lane] - x0[(size_t)v * 3 + lane];
}

// ---- fused edge kernel --------------------------------------------------------------------
constexpr unsigned PROW_B = NSLOT * HS * 4;       // bytes of one node's row of P

struct EdgeSmem {
    float *A;
    int *src, *dst;     // byte offsets of the endpoints' P rows (node * PROW_B)
    float *d, *xd, *att, *mx;
    float *wv;          // [4][HS]: soft-attention row (+bias at ATT_BIAS_AT), coordinate head row, W2[256, :] of edge_mlp / coord_mlp
    int *misc;          // [0] first run continues the previous tile, [2..3] segment-end mask, [4..5] head mask
};

__device__ __forceinline__ EdgeSmem edge_smem(float *smem) {
    EdgeSmem s;
    s.A = smem;
    s.src = reinterpret_cast<int *>(smem + TM * SA);
    s.dst = s.src + TM;
    s.d = reinterpret_cast<float *>(s.dst + TM);
    s.xd = s.d + TM;
    s.att = s.xd + 3 * TM;
    s.mx = s.att + TM;
    s.wv = s.mx + 3 * TM;
    s.misc = reinterpret_cast<int *>(s.wv + 4 * HS);
    return s;
}

// A[r][:] = SiLU(Ps[src_r] + Pd[dst_r] + d_r * w_r)   (first Linear of edge_mlp / coord_mlp;
// its bias is folded into Pd by the projection kernel).  Each of the NW waves owns 64 / NW rows.
// Split in two so that the 4-wave build (256 VGPRs per lane) can issue the gathers of the coordinate branch before the
// attention / segmented-sum phases of the feature branch and consume them after (the 8-wave build has no registers to
// spare for that: 61 spills, measured slower).
template <int NW>
struct EdgeGather {
    static constexpr int RPW = TM / NW;
    f32x4 ps[RPW], pd[RPW];     // columns 4 lane .. 4 lane + 3 of the wave's RPW rows
    f32x4 tps, tpd;             // columns 256 + 4 c .. of row lane >> 2 (lanes < 4 RPW, c = lane & 3 < 2)
};

template <int NW>
__device__ __forceinline__ void edge_gather_issue(EdgeGather<NW> &g, const EdgeSmem &s, const float *__restrict__ Ps,
                                                  const float *__restrict__ Pd, int wave, int lane) {
    constexpr int RPW = TM / NW;
    // s.src / s.dst hold the P rows' 32-bit byte offsets (node * PROW_B, premultiplied in phase 0; P stays far below 4 GB):
    // one full-rate add per row and side here instead of a quarter-rate integer multiply-add.  One 1-KiB row segment per
    // wave instruction, all rows in flight.
    const char *ps = reinterpret_cast<const char *>(Ps), *pd = reinterpret_cast<const char *>(Pd);
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave * RPW + rr;
        g.ps[rr] = *reinterpret_cast<const f32x4 *>(ps + ((unsigned)s.src[r] + 16u * lane));
        g.pd[rr] = *reinterpret_cast<const f32x4 *>(pd + ((unsigned)s.dst[r] + 16u * lane));
    }
    if (lane < 4 * RPW && (lane & 3) < 2) {
        const int r = wave * RPW + (lane >> 2), c = lane & 3;
        g.tps = *reinterpret_cast<const f32x4 *>(ps + ((unsigned)s.src[r] + 16u * (64 + c)));
        g.tpd = *reinterpret_cast<const f32x4 *>(pd + ((unsigned)s.dst[r] + 16u * (64 + c)));
    }
}

// The same gathers issued before the tile's LDS row data exists: lane (l & (RPW - 1)) of every wave loads the endpoints of row
// wave * RPW + (l & (RPW - 1)) itself, the row's (wave-uniform) offsets are read out of those lanes (v_readlane) and the loads go
// out while wave 0 is still in its geometry chain -- the gather latency then overlaps phase 0 and its barrier.
template <int NW>
__device__ __forceinline__ void edge_gather_issue_early(EdgeGather<NW> &g, const int *__restrict__ esrc, const int *__restrict__ edst, int e0,
                                                        int ne, const float *__restrict__ Ps, const float *__restrict__ Pd, int wave, int lane) {
    constexpr int RPW = TM / NW;
    const int rl = min(wave * RPW + (lane & (RPW - 1)), ne - 1);
    const int iu = esrc[e0 + rl], iv = edst[e0 + rl];
    const char *ps = reinterpret_cast<const char *>(Ps), *pd = reinterpret_cast<const char *>(Pd);
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const unsigned ou = (unsigned)__builtin_amdgcn_readlane(iu, rr) * PROW_B, ov = (unsigned)__builtin_amdgcn_readlane(iv, rr) * PROW_B;
        g.ps[rr] = *reinterpret_cast<const f32x4 *>(ps + (ou + 16u * lane));
        g.pd[rr] = *reinterpret_cast<const f32x4 *>(pd + (ov + 16u * lane));
    }
    // (the cross-lane reads stay outside the branch: a lane that the branch switches off supplies nothing to ds_bpermute)
    const unsigned ou = (unsigned)__shfl(iu, (lane >> 2) & (RPW - 1)) * PROW_B, ov = (unsigned)__shfl(iv, (lane >> 2) & (RPW - 1)) * PROW_B;
    if (lane < 4 * RPW && (lane & 3) < 2) {
        const int c = lane & 3;
        g.tps = *reinterpret_cast<const f32x4 *>(ps + (ou + 16u * (64 + c)));
        g.tpd = *reinterpret_cast<const f32x4 *>(pd + (ov + 16u * (64 + c)));
    }
}

template <int NW>
__device__ __forceinline__ void edge_gather_finish(const EdgeGather<NW> &g, const EdgeSmem &s, const float *__restrict__ wr, int wave,
                                                   int lane) {
    constexpr int RPW = TM / NW;
    const f32x4 w0 = reinterpret_cast<const f32x4 *>(wr)[lane];
    // the wave's RPW distances in RPW / 4 broadcast reads up front instead of a read + wait per row: -1.0 % on the kernel, same-call A/B
    // (0.877 vs 0.886 ms).  The f16x2 kernel keeps the per-row form (edge_gather_finish_h, KPD_H_BATCH_D): there the batched read came
    // with a first-forward deviation; every detector of that (profiles/tools/repro_*.py, cold_*_check.py, tests/test_cold_start_gpu.py)
    // is clean for this kernel.
    f32x4 dv[(RPW + 3) / 4];
#pragma unroll
    for (int i = 0; i < (RPW + 3) / 4; ++i) dv[i] = *reinterpret_cast<const f32x4 *>(s.d + wave * RPW + 4 * i);
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave * RPW + rr;
        f32x4 v = g.ps[rr] + g.pd[rr] + dv[rr >> 2][rr & 3] * w0;   // = c * pre-activation (P, w_r carry c)
        v = silu_pre4(v);
        *reinterpret_cast<f32x4 *>(s.A + r * SA + 4 * lane) = v;
    }
    // columns 256..263 of the wave's rows in one pass (lane = row * 4 + chunk): keeping this out of the
    // row loop halves the VALU work, which on gfx950 is paid in MFMA time
    if (lane < 4 * RPW && (lane & 3) < 2) {
        const int r = wave * RPW + (lane >> 2), c = lane & 3;
        const f32x4 w1 = reinterpret_cast<const f32x4 *>(wr)[64 + c];
        f32x4 u = g.tps + g.tpd + s.d[r] * w1;
        u[0] = silu_pre(u[0]); u[1] = silu_pre(u[1]); u[2] = silu_pre(u[2]); u[3] = silu_pre(u[3]);
        if (c == (BIAS_K - 256) / 4) u[(BIAS_K - 256) % 4] = 1.0f;   // constant-1 column: the GEMM adds the bias row itself
        *reinterpret_cast<f32x4 *>(s.A + r * SA + 256 + 4 * c) = u;
    }
}

template <int NW>
__device__ __forceinline__ void build_edge_A(const EdgeSmem &s, const float *__restrict__ Ps, const float *__restrict__ Pd,
                                             const float *__restrict__ wr, int wave, int lane) {
    EdgeGather<NW> g;
    edge_gather_issue<NW>(g, s, Ps, Pd, wave, lane);
    edge_gather_finish<NW>(g, s, wr, wave, lane);
}

// T[row][col] = SiLU(acc + b[col]) for the 257 valid columns.  PRE: the accumulator already holds
// c * (W a + b) (bias row in the GEMM, pre-scaled SiLU) and T receives c * SiLU(.).
template <int NW, bool PRE>
__device__ __forceinline__ void store_T_silu_w(float *T, const f32x16 (&acc)[2][WaveCols<NW>::NT], float ex,
                                               const float *__restrict__ b, int tid, int wave, int lane) {
    constexpr int TPR = NW;      // threads per row = 64 NW / 64
#pragma unroll
    for (int nt = 0; nt < WaveCols<NW>::NT; ++nt) {
        const int col = acc_col_w<NW>(nt, wave, lane);
        const float bb = PRE ? 0.0f : b[col];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if constexpr (PRE) {
#pragma unroll
                for (int r4 = 0; r4 < 16; r4 += 4) {
                    const f32x4 y = silu_pre4(f32x4{acc[mt][nt][r4], acc[mt][nt][r4 + 1], acc[mt][nt][r4 + 2], acc[mt][nt][r4 + 3]});
#pragma unroll
                    for (int i = 0; i < 4; ++i) T[acc_row(mt, r4 + i, lane) * SA + col] = y[i];
                }
            } else {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) T[acc_row(mt, reg, lane) * SA + col] = silu(acc[mt][nt][reg] + bb);
            }
        }
    }
    if ((tid % TPR) == 0) T[(tid / TPR) * SA + 256] = PRE ? silu_pre(ex) : silu(ex + b[256]);
}

// Phase stamps (diagnostic builds of the timeline only; a.stamps is null in production): wave 0 of
// every workgroup adds the s_memtime delta of each phase to a.stamps[phase].
#define KPD_STAMP(idx)                                                                     \
    if (a.stamps && tid == 0) {                                                            \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                      \
        atomicAdd(&a.stamps[idx], (unsigned long long)(now_ - t_prev_));                   \
        t_prev_ = now_;                                                                    \
    }

template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 4) void k_egnn_edge(EdgeArgs a) {
    constexpr int TPR = NW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const EdgeSmem s = edge_smem(smem);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    unsigned long long t_prev_ = a.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    // tile decode (XCD-aware: consecutive tiles -- neighbouring edges of one complex, which
    // share P rows -- go to the same XCD / L2)
    const int T = a.meta[8];
    const int chunk = (T + 7) >> 3;
    const int bi = blockIdx.x >> 3;
    // Tail: an XCD's tiles fill its workgroup slots (a.split_slots: two per CU) round after round, and the last round is usually almost
    // empty (775 tiles on 64 slots: 7 tiles keep the whole launch waiting for a full tile time).  When that remainder fits half the slots,
    // its tiles run as TWO work items each -- the feature branch and the coordinate branch -- so the last round lasts half a tile.
    int tl = bi, bsel = 3;                    // bit 0: feature branch, bit 1: coordinate branch
    const int rem = a.split_slots > 0 ? chunk % a.split_slots : 0;
    if (rem > 0 && 2 * rem <= a.split_slots) {
        const int full = chunk - rem;
        if (bi >= full) {
            const int j = bi - full;
            if (j >= 2 * rem) return;
            tl = full + (j >> 1);
            bsel = 1 << (j & 1);
        }
    } else if (bi >= chunk) return;
    const int tile = (blockIdx.x & 7) * chunk + tl;
    if (tile >= T) return;
    int et = 0;
#pragma unroll
    for (int e = 1; e < 4; ++e)
        if (tile >= a.meta[4 + e]) et = e;
    const int tile_in_et = tile - a.meta[4 + et];
    const int e0 = tile_in_et * TM;
    const int ne = min(TM, a.meta[et] - e0);
    const int snt = a.src_nt[et], dnt = a.dst_nt[et];
    const int *__restrict__ esrc = a.src[et];
    const int *__restrict__ edst = a.dst[et];

    const float *Ps = a.P[snt] + (size_t)a.src_slot[et] * HS;
    const float *Pd = a.P[dnt] + (size_t)a.dst_slot[et] * HS;
    // the feature branch's P rows start travelling before the geometry chain and the first barrier (-1.1 % on the kernel, same-call
    // A/B: 0.859 vs 0.869 ms)
    EdgeGather<NW> ge;
    if (bsel & 1) edge_gather_issue_early<NW>(ge, esrc, edst, e0, ne, Ps, Pd, wave, lane);
    __builtin_amdgcn_sched_barrier(0);
    // phase 0: edge endpoints and geometry (dynamics.py:160-169, 209-217); head weights to LDS
    if (tid < TM) {
        const int e = e0 + min(tid, ne - 1);
        const int u = esrc[e], v = edst[e];
        s.src[tid] = (int)((unsigned)u * PROW_B);        // byte offsets of the nodes' P rows
        s.dst[tid] = (int)((unsigned)v * PROW_B);
        const float *xs = a.x[snt] + (size_t)u * 3, *xd = a.x[dnt] + (size_t)v * 3;
        const float dx = xs[0] - xd[0], dy = xs[1] - xd[1], dz = xs[2] - xd[2];
        const float d = sqrtf(dx * dx + dy * dy + dz * dz);
        const float inv = 1.0f / (d + 1.0f);
        s.d[tid] = d;
        s.xd[3 * tid] = dx * inv;
        s.xd[3 * tid + 1] = dy * inv;
        s.xd[3 * tid + 2] = dz * inv;
        // run structure of the dst-sorted tile as two 64-bit masks (wave 0 == rows 0..63)
        const int vprev = tid > 0 ? edst[e0 + min(tid - 1, ne - 1)] : (e0 > 0 ? edst[e0 - 1] : -1);
        const int vnext = tid + 1 < ne ? edst[e0 + tid + 1] : -2;
        const unsigned long long heads = __ballot(tid < ne && (tid == 0 || vprev != v));
        const unsigned long long ends = __ballot(tid < ne && vnext != v);
        if (tid == 0) {
            s.misc[0] = (vprev == v) ? 1 : 0;
            s.misc[2] = (int)(ends & 0xffffffffu);
            s.misc[3] = (int)(ends >> 32);
            s.misc[4] = (int)(heads & 0xffffffffu);
            s.misc[5] = (int)(heads >> 32);
        }
    } else {
        // four HS rows to LDS: soft-attention row, coordinate-head row, and row 256 of W2 of either branch (the "+1" output
        // column is a per-row dot on the VALU; from LDS it needs no global round trip next to the GEMM)
        // (no pointer array: pointers that pass through private memory lose their address space and turn every later load
        // through them into a flat load with full waits)
        for (int i = tid - TM; i < 4 * 66; i += 64 * NW - TM) {
            const int which = i / 66, j = i - which * 66;      // 66 float4 = one HS row
            const float *row = which == 0 ? a.watt[et] : which == 1 ? a.w3[et] : which == 2 ? a.wx_e[et] : a.wx_c[et];
            reinterpret_cast<f32x4 *>(s.wv + which * HS)[j] = reinterpret_cast<const f32x4 *>(row)[j];
        }
    }
    [[maybe_unused]] BPrefetch bpre;
    if constexpr (NW == 4) gemm_b_prefetch(bpre, (bsel & 1) ? a.wp_e[et] : a.wp_c[et], wave, lane);     // lands during the gather / A-build
    lds_barrier();
    KPD_STAMP(0)

    const int first_is_cont = s.misc[0];
    const unsigned long long endmask =
        ((unsigned long long)(unsigned)s.misc[3] << 32) | (unsigned long long)(unsigned)s.misc[2];
    f32x16 acc[2][WaveCols<NW>::NT];
    float ex;
    EdgeGather<NW == 4 ? 4 : TM> gc;      // (one row per wave, unused, in the 8-wave build)

    // ---- feature messages: m = edge_mlp(f); msg_h = m * sigmoid(att(m)) (dynamics.py:111-112)
    // timing experiments of the TOOLS build only (KPD_EDGE_ABLATE: 1 no GEMM, 2 no A-build, 4 no epilogues); the constant 0 in the product
    const int abl = KPD_TOOL_SWITCH(a.ablate, 0);
    if (bsel & 1) {
    if (!(abl & 2)) edge_gather_finish<NW>(ge, s, a.wr_e[et], wave, lane);
    lds_barrier();
    KPD_STAMP(1)
    acc_zero_w<NW>(acc);
    ex = (abl & 4) ? 0.0f : row_dot_chunks<TPR>(s.A, s.wv + 2 * HS, KP / 4, tid);
    if (!(abl & 1)) {
        if constexpr (NW == 4) gemm_rows64_pre<NG, SA>(s.A, a.wp_e[et], acc, wave, lane, bpre);
        else gemm_rows64_w<NW, NG, SA>(s.A, a.wp_e[et], acc, wave, lane);
    }
    if constexpr (NW == 4) gemm_b_prefetch(bpre, a.wp_c[et], wave, lane);     // for the coordinate GEMM, four phases away
    lds_barrier();
    KPD_STAMP(2)
    if (!(abl & 4)) store_T_silu_w<NW, true>(s.A, acc, ex, a.b_e[et], tid, wave, lane);
    else if (acc[0][0][0] == 12345.0f) s.A[tid] = acc[1][NW == 4 ? 1 : 0][3] + acc[0][NW == 4 ? 1 : 0][5] + acc[1][0][7];
    if constexpr (NW == 4) {  // the coordinate branch's P rows start travelling now; consumed after the segmented sum below
        if (!(abl & 2) && (bsel & 2)) {
        edge_gather_issue<NW>(gc, s, Ps + HS, Pd + HS, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
        }
    }
    lds_barrier();
    KPD_STAMP(3)
    if (!(abl & 4)) {
        float dot = row_dot_chunks<TPR>(s.A, s.wv, 64, tid);
        const int row = tid / TPR;
        if ((tid % TPR) == 0) {
            dot = fmaf(s.A[row * SA + 256], s.wv[256], dot);
            // T holds c * m, w_att carries 1 / c; the returned weight carries 1 / c so that T * att = m * sigmoid(.)
            s.att[row] = row < ne ? sigmoidf_(dot + s.wv[ATT_BIAS_AT]) * (1.0f / SILU_C) : 0.0f;
        }
    }
    lds_barrier();
    KPD_STAMP(4)
    if (!(abl & 4)) {
        // segmented sum over dst (dynamics.py:182-185): thread = column, rows in order; the run
        // boundaries are wave-uniform (endmask), LDS reads are issued 16 rows at a time
        float *hmain = a.hn_main[et], *hcont = a.hn_cont[et] + (size_t)tile_in_et * HS;
        if (tid < 256) {
            float run = 0.0f;
            int piece = 0;
#pragma unroll 1
            for (int r0 = 0; r0 < TM; r0 += 16) {
                if (r0 >= ne) break;
                float v[16], w[16];
                int dvv[16];
                {
                    typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const i32x4 t = *reinterpret_cast<const i32x4 *>(s.dst + r0 + 4 * j);
                        dvv[4 * j] = t[0]; dvv[4 * j + 1] = t[1]; dvv[4 * j + 2] = t[2]; dvv[4 * j + 3] = t[3];
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    w[i] = s.att[r0 + i];
                    v[i] = s.A[(r0 + i) * SA + tid];
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    run = fmaf(v[i], w[i], run);
                    if ((endmask >> (r0 + i)) & 1ull) {
                        float *out = (piece == 0 && first_is_cont) ? hcont : hmain + ((unsigned)dvv[i] / (unsigned)(NSLOT * 4));
                        out[tid] = run;
                        run = 0.0f;
                        ++piece;
                    }
                }
            }
        }
        // column 256: the weighted values wait in LDS for the scan at the end of the kernel (with the coordinate messages)
        if (wave == NW - 1) reinterpret_cast<float *>(s.misc + 8)[lane] = s.A[lane * SA + 256] * s.att[lane];
    }
    lds_barrier();
    KPD_STAMP(5)
    }

    // ---- coordinate messages: msg_x = tanh(coord_mlp(f)) * x_diff * range (dynamics.py:113-120)
    if (bsel & 2) {
    if (!(abl & 2)) {
    if constexpr (NW == 4) {
        if (!(bsel & 1)) edge_gather_issue<NW>(gc, s, Ps + HS, Pd + HS, wave, lane);        // (a coordinate-only work item of the tail)
        edge_gather_finish<NW>(gc, s, a.wr_c[et], wave, lane);
    } else build_edge_A<NW>(s, Ps + HS, Pd + HS, a.wr_c[et], wave, lane);
    }
    lds_barrier();
    KPD_STAMP(6)
    acc_zero_w<NW>(acc);
    ex = (abl & 4) ? 0.0f : row_dot_chunks<TPR>(s.A, s.wv + 3 * HS, KP / 4, tid);
    if (!(abl & 1)) {
        if constexpr (NW == 4) gemm_rows64_pre<NG, SA>(s.A, a.wp_c[et], acc, wave, lane, bpre);
        else gemm_rows64_w<NW, NG, SA>(s.A, a.wp_c[et], acc, wave, lane);
    }
    lds_barrier();
    KPD_STAMP(7)
    if (!(abl & 4)) store_T_silu_w<NW, true>(s.A, acc, ex, a.b_c[et], tid, wave, lane);
    else if (acc[0][0][0] == 12345.0f) s.A[tid] = acc[1][NW == 4 ? 1 : 0][3] + acc[0][NW == 4 ? 1 : 0][5] + acc[1][0][7];
    lds_barrier();
    KPD_STAMP(8)
    if (!(abl & 4)) {
        float dot = row_dot_chunks<TPR>(s.A, s.wv + HS, 64, tid);
        const int row = tid / TPR;
        if ((tid % TPR) == 0) {
            dot = fmaf(s.A[row * SA + 256], s.wv[HS + 256], dot);
            float c = a.use_tanh ? tanhf(dot) * a.coords_range : dot;
            if (row >= ne) c = 0.0f;
            s.mx[3 * row] = c * s.xd[3 * row];
            s.mx[3 * row + 1] = c * s.xd[3 * row + 1];
            s.mx[3 * row + 2] = c * s.xd[3 * row + 2];
        }
    }
    lds_barrier();
    KPD_STAMP(9)
    }
    if (wave == 0 && !(abl & 4)) {
        // segmented inclusive scan across lanes (lane = row), then the last lane of every run writes
        const unsigned long long heads =
            ((unsigned long long)(unsigned)s.misc[5] << 32) | (unsigned long long)(unsigned)s.misc[4];
        const unsigned long long upto = lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull);
        const int start = 63 - __clzll((long long)((heads & upto) | 1ull));
        float vx = 0.0f, vy = 0.0f, vz = 0.0f;
        if (bsel & 2) { vx = s.mx[3 * lane]; vy = s.mx[3 * lane + 1]; vz = s.mx[3 * lane + 2]; }
        float vh = (bsel & 1) ? reinterpret_cast<const float *>(s.misc + 8)[lane] : 0.0f;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float tx = __shfl_up(vx, off), ty = __shfl_up(vy, off), tz = __shfl_up(vz, off);
            const float th = __shfl_up(vh, off);
            if (lane - off >= start) {
                vx += tx;
                vy += ty;
                vz += tz;
                vh += th;
            }
        }
        if ((endmask >> lane) & 1ull) {
            const int piece = __popcll(endmask & ((1ull << lane) - 1ull));
            const unsigned dsto = (unsigned)s.dst[lane];
            float *out = (piece == 0 && first_is_cont) ? a.xn_cont[et] + (size_t)tile_in_et * 4
                                                       : a.xn_main[et] + (size_t)(dsto / PROW_B) * 4;
            if (bsel & 2) {
                out[0] = vx;
                out[1] = vy;
                out[2] = vz;
            }
            float *oh = (piece == 0 && first_is_cont) ? a.hn_cont[et] + (size_t)tile_in_et * HS : a.hn_main[et] + (dsto / (unsigned)(NSLOT * 4));
            if (bsel & 1) oh[256] = vh;
        }
    }
    KPD_STAMP(10)
}


// ---- training form of the fused edge kernel (egnn_train.hip, forward pass) -----------------------------------------------------
// The same workgroup program as k_egnn_edge<4> -- gather P_src + P_dst + d w_r, SiLU, the 257 x 257 Linear on fp32 MFMA, SiLU, soft
// attention / coordinate head, segmented sum over the tile's destination runs -- on the CURRENT weights (packed per step, unscaled:
// plain SiLU instead of the pre-scaled form) and with everything the backward pass reads stored on the way: pre1, a1 (from the gather's
// registers, one 1-KiB row segment per wave store), pre2, a2 (from the accumulators: 32 consecutive columns of two rows per store), the
// attention weight, the coordinate scalar and the geometry.  It replaces k_edge_pre1 + k_ws_gemm<0> + the head and segmented-sum
// kernels of the forward pass (three passes over E x 257 matrices less per branch).
// phase-cycle sums of the training edge kernels (a.stamps is null in production: profiles/tools/train_stamps.sh)
#define TRAIN_STAMP(idx)                                                                   \
    if (a.stamps && tid == 0) {                                                            \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                      \
        atomicAdd(&a.stamps[idx], (unsigned long long)(now_ - t_prev_));                   \
        t_prev_ = now_;                                                                    \
    }

struct EdgeKeep {
    float *pre1, *a1, *pre2;               // rows e0 .. of the branch's kept arrays [E][HS]
};

// (addresses of the kept rows: a wave-uniform row base -- scalar registers -- plus one 32-bit lane offset, so that the sixteen rows of a
// wave cost no vector registers for addresses; the opaque asm keeps the compiler from carrying offsets from one phase to the next)
// (the kept arrays are written once and read in the backward pass, tens of milliseconds later; non-temporal stores for them were
// measured -- 1.354 vs 1.283 ms per launch, not better -- and removed)
template <class V>
__device__ __forceinline__ void keep_store(V *p, const V &v) {
    *p = v;
}

__device__ __forceinline__ void edge_gather_finish_train(const EdgeGather<4> &g, const EdgeSmem &s, const float *__restrict__ wr, int wave, int lane,
                                                         const EdgeKeep &k, int ne) {
    constexpr int RPW = TM / 4;
    const f32x4 w0 = reinterpret_cast<const f32x4 *>(wr)[lane];
    f32x4 dv[RPW / 4];
#pragma unroll
    for (int i = 0; i < RPW / 4; ++i) dv[i] = *reinterpret_cast<const f32x4 *>(s.d + wave * RPW + 4 * i);
    unsigned voff = 16u * lane;
    asm volatile("" : "+v"(voff));
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave * RPW + rr;
        const f32x4 v = g.ps[rr] + g.pd[rr] + dv[rr >> 2][rr & 3] * w0;          // (b1 rides in the dst projection)
        f32x4 a;
        a[0] = silu(v[0]); a[1] = silu(v[1]); a[2] = silu(v[2]); a[3] = silu(v[3]);
        *reinterpret_cast<f32x4 *>(s.A + r * SA + 4 * lane) = a;
        if (r < ne) {
            char *p1 = reinterpret_cast<char *>(k.pre1 + (size_t)r * HS), *pa = reinterpret_cast<char *>(k.a1 + (size_t)r * HS);
            keep_store(reinterpret_cast<f32x4 *>(p1 + voff), v);
            keep_store(reinterpret_cast<f32x4 *>(pa + voff), a);
        }
    }
    if (lane < 4 * RPW && (lane & 3) < 2) {
        const int r = wave * RPW + (lane >> 2), c = lane & 3;
        const f32x4 w1 = reinterpret_cast<const f32x4 *>(wr)[64 + c];
        const f32x4 u = g.tps + g.tpd + s.d[r] * w1;             // column 256; the padding columns come out as exact zeros
        f32x4 a;
        a[0] = silu(u[0]); a[1] = silu(u[1]); a[2] = silu(u[2]); a[3] = silu(u[3]);
        if (r < ne) {
            const unsigned o = (unsigned)(r * HS + 256 + 4 * c) * 4u;
            keep_store(reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(k.pre1) + o), u);
            keep_store(reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(k.a1) + o), a);
        }
        if (c == (BIAS_K - 256) / 4) a[(BIAS_K - 256) % 4] = 1.0f;   // constant-1 column of the LDS tile only: the GEMM adds the bias row itself
        *reinterpret_cast<f32x4 *>(s.A + r * SA + 256 + 4 * c) = a;
    }
}

// T = SiLU(acc) to LDS; pre2 = acc (bias included: bias row of the GEMM) and a2 = T to the kept arrays
// (a2 is not stored: the backward edge kernel recomputes it from the pre2 rows it streams anyway)
__device__ __forceinline__ void store_T_train(float *T, const f32x16 (&acc)[2][2], float ex, int tid, int wave, int lane, const EdgeKeep &k, int ne) {
    // element (row, col) of accumulator register reg of tile (mt, nt): row = 32 mt + 8 (reg >> 2) + (reg & 3) + 4 (lane >> 5),
    // col = 64 wave + 32 nt + (lane & 31): one lane offset, everything else is a constant (groups of four rows within the 4-KiB immediate)
    const int row0 = 4 * (lane >> 5), col0 = 64 * wave + (lane & 31);
    unsigned off0 = (unsigned)(row0 * HS + col0) * 4u;
    asm volatile("" : "+v"(off0));
    char *b2 = reinterpret_cast<char *>(k.pre2);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const unsigned off = off0 + (unsigned)((32 * mt + 8 * q4) * HS + 32 * nt) * 4u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = 32 * mt + 8 * q4 + j + row0;
                    const float v = acc[mt][nt][4 * q4 + j], a = silu(v);
                    T[row * SA + col0 + 32 * nt] = a;
                    if (row < ne) keep_store(reinterpret_cast<float *>(b2 + (off + (unsigned)(j * HS * 4))), v);
                }
            }
    if ((tid & 3) == 0) {
        const int row = tid >> 2;
        const float a = silu(ex);
        T[row * SA + 256] = a;
        if (row < ne) keep_store(k.pre2 + row * HS + 256, ex);
    }
}

__global__ __launch_bounds__(256, 2) void k_egnn_edge_train(EdgeTrainArgs a) {
    constexpr int NW = 4, TPR = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const EdgeSmem s = edge_smem(smem);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    unsigned long long t_prev_ = a.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    const int T = a.meta[8];
    const int chunk = (T + 7) >> 3;
    const int bi = blockIdx.x >> 3;
    if (bi >= chunk) return;
    const int tile = (blockIdx.x & 7) * chunk + bi;
    if (tile >= T) return;
    int et = 0;
#pragma unroll
    for (int e = 1; e < 4; ++e)
        if (tile >= a.meta[4 + e]) et = e;
    const int tile_in_et = tile - a.meta[4 + et];
    const int e0 = tile_in_et * TM;
    const int ne = min(TM, a.meta[et] - e0);
    const int snt = a.src_nt[et], dnt = a.dst_nt[et];
    const int ne1 = (a.skip & 1) ? 0 : ne, ne2 = (a.skip & 2) ? 0 : ne;      // rows whose kept values are stored (a.skip = 0 outside timing experiments)
    const int *__restrict__ esrc = a.src[et];
    const int *__restrict__ edst = a.dst[et];
    const float *Ps_e = a.P[snt] + (size_t)a.slot[et][0][0] * HS, *Pd_e = a.P[dnt] + (size_t)a.slot[et][0][1] * HS;
    const float *Ps_c = a.P[snt] + (size_t)a.slot[et][1][0] * HS, *Pd_c = a.P[dnt] + (size_t)a.slot[et][1][1] * HS;
    EdgeKeep ke, kc;
    ke.pre1 = a.keep[et][0][0] + (size_t)e0 * HS; ke.a1 = a.keep[et][0][1] + (size_t)e0 * HS;
    ke.pre2 = a.keep[et][0][2] + (size_t)e0 * HS;
    kc.pre1 = a.keep[et][1][0] + (size_t)e0 * HS; kc.a1 = a.keep[et][1][1] + (size_t)e0 * HS;
    kc.pre2 = a.keep[et][1][2] + (size_t)e0 * HS;

    EdgeGather<NW> ge;
    edge_gather_issue_early<NW>(ge, esrc, edst, e0, ne, Ps_e, Pd_e, wave, lane);
    __builtin_amdgcn_sched_barrier(0);
    // phase 0: endpoints and geometry (dynamics.py:160-169, 209-217), kept for the backward pass; head rows to LDS
    if (tid < TM) {
        const int e = e0 + min(tid, ne - 1);
        const int u = esrc[e], v = edst[e];
        s.src[tid] = (int)((unsigned)u * PROW_B);
        s.dst[tid] = (int)((unsigned)v * PROW_B);
        const float *xs = a.x[snt] + (size_t)u * 3, *xd = a.x[dnt] + (size_t)v * 3;
        const float dx = xs[0] - xd[0], dy = xs[1] - xd[1], dz = xs[2] - xd[2];
        const float d = sqrtf(dx * dx + dy * dy + dz * dz);
        const float inv = 1.0f / (d + 1.0f);
        s.d[tid] = d;
        s.xd[3 * tid] = dx * inv;
        s.xd[3 * tid + 1] = dy * inv;
        s.xd[3 * tid + 2] = dz * inv;
        if (tid < ne && !(a.skip & 4)) {
            a.dij[et][e] = d;
            float *xo = a.xdiff[et] + (size_t)e * 3, *no = a.nvec[et] + (size_t)e * 3;
            xo[0] = dx; xo[1] = dy; xo[2] = dz;
            no[0] = dx * inv; no[1] = dy * inv; no[2] = dz * inv;
        }
        const int vprev = tid > 0 ? edst[e0 + min(tid - 1, ne - 1)] : (e0 > 0 ? edst[e0 - 1] : -1);
        const int vnext = tid + 1 < ne ? edst[e0 + tid + 1] : -2;
        const unsigned long long heads = __ballot(tid < ne && (tid == 0 || vprev != v));
        const unsigned long long ends = __ballot(tid < ne && vnext != v);
        if (tid == 0) {
            s.misc[0] = (vprev == v) ? 1 : 0;
            s.misc[2] = (int)(ends & 0xffffffffu);
            s.misc[3] = (int)(ends >> 32);
            s.misc[4] = (int)(heads & 0xffffffffu);
            s.misc[5] = (int)(heads >> 32);
        }
    } else {
        for (int i = tid - TM; i < 4 * 66; i += 64 * NW - TM) {
            const int which = i / 66, j = i - which * 66;
            const float *row = which == 0 ? a.watt[et] : which == 1 ? a.w3[et] : which == 2 ? a.wx[et][0] : a.wx[et][1];
            reinterpret_cast<f32x4 *>(s.wv + which * HS)[j] = reinterpret_cast<const f32x4 *>(row)[j];
        }
    }
    BPrefetch bpre;
    gemm_b_prefetch(bpre, a.wp[et][0], wave, lane);
    lds_barrier();
    TRAIN_STAMP(0)

    // (the run structure is wave-uniform: as scalars -- three vector registers less in a kernel that sat at 256 with 6 spilled)
    const int first_is_cont = __builtin_amdgcn_readfirstlane(s.misc[0]);
    const unsigned long long endmask = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(s.misc[3]) << 32) |
                                       (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(s.misc[2]);
    f32x16 acc[2][2];
    float ex;

    // ---- feature messages (dynamics.py:103-112)
    edge_gather_finish_train(ge, s, a.wr[et][0], wave, lane, ke, ne1);
    lds_barrier();
    TRAIN_STAMP(1)
    acc_zero_w<NW>(acc);
    ex = row_dot_chunks<TPR>(s.A, s.wv + 2 * HS, KP / 4, tid);
    gemm_rows64_pre<NG, SA>(s.A, a.wp[et][0], acc, wave, lane, bpre);
    gemm_b_prefetch(bpre, a.wp[et][1], wave, lane);
    lds_barrier();
    TRAIN_STAMP(2)
    store_T_train(s.A, acc, ex, tid, wave, lane, ke, ne2);
    lds_barrier();
    TRAIN_STAMP(3)
    {
        float dot = row_dot_chunks<TPR>(s.A, s.wv, 64, tid);
        const int row = tid / TPR;
        if ((tid % TPR) == 0) {
            dot = fmaf(s.A[row * SA + 256], s.wv[256], dot);
            const float at = row < ne ? sigmoidf_(dot + s.wv[ATT_BIAS_AT]) : 0.0f;
            s.att[row] = at;
            if (row < ne) a.att[et][e0 + row] = at;
        }
    }
    lds_barrier();
    TRAIN_STAMP(4)
    EdgeGather<NW> gc;
    edge_gather_issue<NW>(gc, s, Ps_c, Pd_c, wave, lane);      // the coordinate branch's rows travel during the segmented sum
    __builtin_amdgcn_sched_barrier(0);
    {
        // segmented sum over dst (dynamics.py:182-185), as in k_egnn_edge
        float *hmain = a.hn_main[et], *hcont = a.hn_cont[et] + (size_t)tile_in_et * HS;
        if (tid < 256) {
            float run = 0.0f;
            int piece = 0;
#pragma unroll 1
            for (int r0 = 0; r0 < TM; r0 += 16) {
                if (r0 >= ne) break;
                float v[16], w[16];
                int dvv[16];
                {
                    typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const i32x4 t = *reinterpret_cast<const i32x4 *>(s.dst + r0 + 4 * j);
                        dvv[4 * j] = t[0]; dvv[4 * j + 1] = t[1]; dvv[4 * j + 2] = t[2]; dvv[4 * j + 3] = t[3];
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    w[i] = s.att[r0 + i];
                    v[i] = s.A[(r0 + i) * SA + tid];
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    run = fmaf(v[i], w[i], run);
                    if ((endmask >> (r0 + i)) & 1ull) {
                        float *out = (piece == 0 && first_is_cont) ? hcont : hmain + ((unsigned)dvv[i] / (unsigned)(NSLOT * 4));
                        out[tid] = run;
                        run = 0.0f;
                        ++piece;
                    }
                }
            }
        }
        if (wave == NW - 1) reinterpret_cast<float *>(s.misc + 8)[lane] = s.A[lane * SA + 256] * s.att[lane];
    }
    lds_barrier();
    TRAIN_STAMP(5)

    // ---- coordinate messages (dynamics.py:113-120)
    // (the second branch takes the thread index through an opaque copy: with the plain one the compiler keeps LDS addresses of the first
    //  branch alive across its GEMM for re-use here, and at 256 registers that meant three spilled -- each reload a vmcnt(0) in the store stream)
    int tid_c = tid;
    asm volatile("" : "+v"(tid_c));
    const int lane_c = tid_c & 63;
    edge_gather_finish_train(gc, s, a.wr[et][1], wave, lane_c, kc, ne1);
    lds_barrier();
    TRAIN_STAMP(6)
    acc_zero_w<NW>(acc);
    ex = row_dot_chunks<TPR>(s.A, s.wv + 3 * HS, KP / 4, tid_c);
    gemm_rows64_pre<NG, SA>(s.A, a.wp[et][1], acc, wave, lane_c, bpre);
    lds_barrier();
    TRAIN_STAMP(7)
    store_T_train(s.A, acc, ex, tid_c, wave, lane_c, kc, ne2);
    lds_barrier();
    TRAIN_STAMP(8)
    {
        float dot = row_dot_chunks<TPR>(s.A, s.wv + HS, 64, tid_c);
        const int row = tid_c / TPR;
        if ((tid_c % TPR) == 0) {
            dot = fmaf(s.A[row * SA + 256], s.wv[HS + 256], dot);
            float c = a.use_tanh ? tanhf(dot) * a.coords_range : dot;
            if (row >= ne) c = 0.0f;
            else a.sc[et][e0 + row] = dot;
            s.mx[3 * row] = c * s.xd[3 * row];
            s.mx[3 * row + 1] = c * s.xd[3 * row + 1];
            s.mx[3 * row + 2] = c * s.xd[3 * row + 2];
        }
    }
    lds_barrier();
    if (wave == 0) {
        const unsigned long long heads = ((unsigned long long)(unsigned)s.misc[5] << 32) | (unsigned long long)(unsigned)s.misc[4];
        const unsigned long long upto = lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull);
        const int start = 63 - __clzll((long long)((heads & upto) | 1ull));
        float vx = s.mx[3 * lane], vy = s.mx[3 * lane + 1], vz = s.mx[3 * lane + 2];
        float vh = reinterpret_cast<const float *>(s.misc + 8)[lane];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float tx = __shfl_up(vx, off), ty = __shfl_up(vy, off), tz = __shfl_up(vz, off), th = __shfl_up(vh, off);
            if (lane - off >= start) {
                vx += tx;
                vy += ty;
                vz += tz;
                vh += th;
            }
        }
        if ((endmask >> lane) & 1ull) {
            const int piece = __popcll(endmask & ((1ull << lane) - 1ull));
            const unsigned dsto = (unsigned)s.dst[lane];
            float *out = (piece == 0 && first_is_cont) ? a.xn_cont[et] + (size_t)tile_in_et * 4 : a.xn_main[et] + (size_t)(dsto / PROW_B) * 4;
            out[0] = vx;
            out[1] = vy;
            out[2] = vz;
            float *oh = (piece == 0 && first_is_cont) ? a.hn_cont[et] + (size_t)tile_in_et * HS : a.hn_main[et] + (dsto / (unsigned)(NSLOT * 4));
            oh[256] = vh;
        }
    }
    TRAIN_STAMP(9)
    if (a.stamps && tid == 0) atomicAdd(&a.stamps[15], 1ull);
}

// ---- backward form (egnn_train.hip, backward pass of the edge MLPs of a layer) ---------------------------------------------------
// Per 64-edge tile and branch: head backward (the rows of a2 / pre2 of the tile stream through registers: dpre2 = d(a2) SiLU'(pre2), written
// to the LDS tile and, over pre2, to HBM for the dW2 product), the 257 x 257 product with W2 on fp32 MFMA, dpre1 = (dpre2 W2) SiLU'(pre1)
// (pre1 read in the accumulator layout, dpre1 written over it for the by-source sums), its product with the radial column (d dij), and the
// segmented sums over the tile's destination runs of dpre1 (dV) and of dij * dpre1 (the radial weight gradient's per-node share) as
// main / continuation pieces -- the work of k_*_head_bwd, k_ws_gemm<WS_SILU_BWD>, k_add_halves and the by-destination k_segsum264 of
// every (edge type, branch) in one launch per layer.  The column sums of dpre2 (b2 gradient) and of a2 ds (head weight gradient) leave as one
// partial row per tile (k_colsum_reduce adds them in tile order).  In place: dpre2 over pre2, dpre1 over pre1, ds over att, d dij over sc,
// dn over nvec -- each element is read by the one workgroup that overwrites it.
struct EdgeBwdSmem {
    float *A;
    int *dst;                 // destination node of the row
    float *zi, *sa, *dd;      // zinv[dst]; att (feature) / sc (coordinate); dij
    float *g3, *nv;           // d x_out[dst] * zinv (3); nvec (3)
    float *ddij;              // d dij of the row, both branches
    float *c256;              // [2][TM]: column 256 of dpre1 and of dij * dpre1, for the scan
    int *misc;
    float *wv;                // [4][HS]: row 256 of W2^T and the radial column of W1, per branch -- the vectors of the per-row dots
};
constexpr int EDGE_BWD_LDS_BYTES = TM * SA * 4 + (TM * (1 + 3 + 3 + 3 + 1 + 2) + 8 + 4 * HS) * 4;

__device__ __forceinline__ EdgeBwdSmem edge_bwd_smem(float *smem) {
    EdgeBwdSmem s;
    s.A = smem;
    s.dst = reinterpret_cast<int *>(smem + TM * SA);
    s.zi = reinterpret_cast<float *>(s.dst + TM);
    s.sa = s.zi + TM;
    s.dd = s.sa + TM;
    s.g3 = s.dd + TM;
    s.nv = s.g3 + 3 * TM;
    s.ddij = s.nv + 3 * TM;
    s.c256 = s.ddij + TM;
    s.misc = reinterpret_cast<int *>(s.c256 + 2 * TM);
    s.wv = reinterpret_cast<float *>(s.misc + 8);
    return s;
}

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float silu_grad_(float x) {
    const float sg = sigmoidf_(x);
    return sg * (1.0f + x * (1.0f - sg));
}

// running column sums of a wave over its rows: dpre2 (-> b2 gradient) and a2 ds (-> head weight gradient); lane = columns 4 lane .. + 3, lane 0 column 256 too
struct HeadSums {
    f32x4 cs, ws;
    float cs_t, ws_t;
};

// head backward of one branch for the wave's 16 rows: A rows <- dpre2 (zero rows past ne), dpre2 over pre2 in HBM
template <bool FEAT>
__device__ __forceinline__ void edge_head_bwd_rows(const EdgeBwdSmem &s, const EdgeBwdArgs &a, int et, int e0, int ne, int wave, int lane,
                                                   const float *__restrict__ dhn, float *__restrict__ pre2,
                                                   const float *__restrict__ wh, float *__restrict__ ds_out, float *__restrict__ dn_out, HeadSums &hs) {
    constexpr int RPW = TM / 4, BATCH = 8;         // (16 rows in one batch -- one memory round trip per branch instead of two -- spills 440 B: measured on the ISA, not run)
    const f32x4 wv = reinterpret_cast<const f32x4 *>(wh)[lane];
    const float wv_t = wh[256];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int b0 = 0; b0 < RPW; b0 += BATCH) {
        f32x4 dm[BATCH], pv[BATCH];
        float dm_t[BATCH], pv_t[BATCH];
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            const int r = wave * RPW + b0 + i, rc = min(r, ne - 1);
            const size_t eo = (size_t)(e0 + rc) * HS;
            pv[i] = *reinterpret_cast<const f32x4 *>(pre2 + eo + 4 * lane);
            pv_t[i] = lane == 0 ? pre2[eo + 256] : 0.0f;
            if (FEAT) {
                const size_t vo = (size_t)s.dst[rc] * HS;
                dm[i] = *reinterpret_cast<const f32x4 *>(dhn + vo + 4 * lane);
                dm_t[i] = lane == 0 ? dhn[vo + 256] : 0.0f;
            }
        }
#pragma unroll
        for (int i = 0; i < BATCH; ++i) {
            const int r = wave * RPW + b0 + i;
            const bool on = r < ne;
            // a2 = SiLU(pre2) is not kept: the same expression the forward kernel put into its tile
            f32x4 av;
#pragma unroll
            for (int q = 0; q < 4; ++q) av[q] = silu(pv[i][q]);
            const float av_t = lane == 0 ? silu(pv_t[i]) : 0.0f;
            f32x4 g = zero;
            float gt = 0.0f, ds = 0.0f;
            if (FEAT) {
                const float zi = s.zi[min(r, ne - 1)], at = s.sa[min(r, ne - 1)];
                const f32x4 d4 = dm[i] * zi;
                const float d_t = dm_t[i] * zi;
                float sdot = d_t * av_t;
#pragma unroll
                for (int q = 0; q < 4; ++q) sdot = fmaf(d4[q], av[q], sdot);
                ds = wave_sum64(sdot) * at * (1.0f - at);
#pragma unroll
                for (int q = 0; q < 4; ++q) g[q] = (d4[q] * at + ds * wv[q]) * silu_grad_(pv[i][q]);
                gt = (d_t * at + ds * wv_t) * silu_grad_(pv_t[i]);
            } else {
                const int rc = min(r, ne - 1);
                const float sc = s.sa[rc];
                const float gx = s.g3[3 * rc], gy = s.g3[3 * rc + 1], gz = s.g3[3 * rc + 2];
                const float dcoef = gx * s.nv[3 * rc] + gy * s.nv[3 * rc + 1] + gz * s.nv[3 * rc + 2];
                // 1 - tanh^2 = 4 e / (1 + e)^2, e = exp(-2 |sc|): exact near saturation, where 1 - th * th cancels to nothing
                const float ex_ = expf(-2.0f * fabsf(sc)), sech2 = 4.0f * ex_ / ((1.0f + ex_) * (1.0f + ex_));
                ds = a.use_tanh ? dcoef * a.coords_range * sech2 : dcoef;
#pragma unroll
                for (int q = 0; q < 4; ++q) g[q] = ds * wv[q] * silu_grad_(pv[i][q]);
                gt = ds * wv_t * silu_grad_(pv_t[i]);
                if (on && lane == 0) {
                    const float coef = a.use_tanh ? tanhf(sc) * a.coords_range : sc;
                    float *dn = dn_out + (size_t)(e0 + r) * 3;         // (over nvec: the values above came from the LDS copy)
                    dn[0] = coef * gx; dn[1] = coef * gy; dn[2] = coef * gz;
                }
            }
            if (!on) { g = zero; gt = 0.0f; ds = 0.0f; }
            *reinterpret_cast<f32x4 *>(s.A + r * SA + 4 * lane) = g;
            if (lane < 2) {
                f32x4 t = zero;
                if (lane == 0) t[0] = gt;
                *reinterpret_cast<f32x4 *>(s.A + r * SA + 256 + 4 * lane) = t;
            }
            if (on && !(a.skip & 8)) {
                *reinterpret_cast<f32x4 *>(pre2 + (size_t)(e0 + r) * HS + 4 * lane) = g;
                if (lane == 0) {
                    pre2[(size_t)(e0 + r) * HS + 256] = gt;
                    if (FEAT) ds_out[e0 + r] = ds;
                }
            }
            hs.cs += g;
            hs.cs_t += gt;
#pragma unroll
            for (int q = 0; q < 4; ++q) hs.ws[q] = fmaf(av[q], ds, hs.ws[q]);
            hs.ws_t = fmaf(av_t, ds, hs.ws_t);
        }
    }
}

// T = acc * SiLU'(pre1) to LDS and, over pre1, to HBM; rows past ne are zeros (their A rows were).  All 64 values of pre1 a lane needs are
// requested before the first is used: one memory latency per tile and branch instead of one per group of rows.
__device__ __forceinline__ void store_T_bwd(float *T, const f32x16 (&acc)[2][2], float ex, int tid, int wave, int lane, float *__restrict__ pre1, int ne,
                                            int ne_st) {
    const int row0 = 4 * (lane >> 5), col0 = 64 * wave + (lane & 31);
    unsigned off0 = (unsigned)(row0 * HS + col0) * 4u;
    asm volatile("" : "+v"(off0));
    char *bp = reinterpret_cast<char *>(pre1);
    f32x16 p[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int rl = 32 * mt + 8 * (reg >> 2) + (reg & 3);
                p[mt][nt][reg] = rl + row0 < ne ? *reinterpret_cast<const float *>(bp + (off0 + (unsigned)(rl * HS + 32 * nt) * 4u)) : 0.0f;
            }
    const float p256 = ((tid & 3) == 0 && (tid >> 2) < ne) ? pre1[(tid >> 2) * HS + 256] : 0.0f;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int rl = 32 * mt + 8 * (reg >> 2) + (reg & 3), row = rl + row0;
                const float v = acc[mt][nt][reg] * silu_grad_(p[mt][nt][reg]);
                T[row * SA + col0 + 32 * nt] = v;
                if (row < ne_st) *reinterpret_cast<float *>(bp + (off0 + (unsigned)(rl * HS + 32 * nt) * 4u)) = v;
            }
    if ((tid & 3) == 0) {
        const int row = tid >> 2;
        float v = 0.0f;
        if (row < ne) {
            v = ex * silu_grad_(p256);
            pre1[row * HS + 256] = v;
        }
        T[row * SA + 256] = v;
    }
}

__global__ __launch_bounds__(256, 2) void k_egnn_edge_bwd(EdgeBwdArgs a) {
    constexpr int NW = 4, TPR = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const EdgeBwdSmem s = edge_bwd_smem(smem);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    unsigned long long t_prev_ = a.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    const int T = a.meta[8];
    const int chunk = (T + 7) >> 3;
    const int bi = blockIdx.x >> 3;
    if (bi >= chunk) return;
    const int tile = (blockIdx.x & 7) * chunk + bi;
    if (tile >= T) return;
    int et = 0;
#pragma unroll
    for (int e = 1; e < 4; ++e)
        if (tile >= a.meta[4 + e]) et = e;
    const int tile_in_et = tile - a.meta[4 + et];
    const int e0 = tile_in_et * TM;
    const int ne = min(TM, a.meta[et] - e0);
    const int dnt = a.dst_nt[et];
    const int *__restrict__ edst = a.dst[et];

    // phase 0: per-row scalars; run structure of the dst-sorted tile
    if (tid < TM) {
        const int e = e0 + min(tid, ne - 1);
        const int v = edst[e];
        const float zi = a.zinv[dnt][v];
        s.dst[tid] = v;
        s.zi[tid] = zi;
        s.sa[tid] = a.att[et][e];
        s.dd[tid] = a.dij[et][e];
        s.ddij[tid] = 0.0f;
        const float *dxo = a.dxo[dnt] + (size_t)v * 3, *nv = a.nvec[et] + (size_t)e * 3;
        s.g3[3 * tid] = dxo[0] * zi; s.g3[3 * tid + 1] = dxo[1] * zi; s.g3[3 * tid + 2] = dxo[2] * zi;
        s.nv[3 * tid] = nv[0]; s.nv[3 * tid + 1] = nv[1]; s.nv[3 * tid + 2] = nv[2];
        const int vprev = tid > 0 ? edst[e0 + min(tid - 1, ne - 1)] : (e0 > 0 ? edst[e0 - 1] : -1);
        const int vnext = tid + 1 < ne ? edst[e0 + tid + 1] : -2;
        const unsigned long long heads = __ballot(tid < ne && (tid == 0 || vprev != v));
        const unsigned long long ends = __ballot(tid < ne && vnext != v);
        if (tid == 0) {
            s.misc[0] = (vprev == v) ? 1 : 0;
            s.misc[2] = (int)(ends & 0xffffffffu);
            s.misc[3] = (int)(ends >> 32);
            s.misc[4] = (int)(heads & 0xffffffffu);
            s.misc[5] = (int)(heads >> 32);
        }
    } else {
        // the four vectors of the per-row dots to LDS, as the forward kernels do: read from global memory inside row_dot_chunks they were
        // 17 dependent L2 round trips per call (a guarded load and a vmcnt(0) per chunk), four calls per tile
        for (int i = tid - TM; i < 4 * 66; i += 64 * NW - TM) {
            const int which = i / 66, j = i - which * 66;
            const float *row = which == 0 ? a.wxT[et][0] : which == 1 ? a.wxT[et][1] : which == 2 ? a.wr[et][0] : a.wr[et][1];
            reinterpret_cast<f32x4 *>(s.wv + which * HS)[j] = reinterpret_cast<const f32x4 *>(row)[j];
        }
    }
    BPrefetch bpre;
    gemm_b_prefetch(bpre, a.wpT[et][0], wave, lane);
    lds_barrier();
    TRAIN_STAMP(32)
    const int first_is_cont = s.misc[0];
    const unsigned long long endmask = ((unsigned long long)(unsigned)s.misc[3] << 32) | (unsigned long long)(unsigned)s.misc[2];
    const unsigned long long headmask = ((unsigned long long)(unsigned)s.misc[5] << 32) | (unsigned long long)(unsigned)s.misc[4];
    f32x16 acc[2][2];

#pragma unroll 1
    for (int br = 0; br < 2; ++br) {
        // (an opaque copy of the thread index per branch: see k_egnn_edge_train -- nothing derived from it is carried across the loop)
        int tid_b = tid;
#ifndef KPD_BWD_LAUNDER_OFF
        asm volatile("" : "+v"(tid_b));
#endif
        const int lane_b = tid_b & 63;
        float *pre1 = a.keep[et][br][0] + (size_t)e0 * HS;
        HeadSums hs;
        hs.cs = f32x4{0.f, 0.f, 0.f, 0.f}; hs.ws = f32x4{0.f, 0.f, 0.f, 0.f}; hs.cs_t = 0.0f; hs.ws_t = 0.0f;
        if (br == 0) {
            edge_head_bwd_rows<true>(s, a, et, e0, ne, wave, lane_b, a.dhn[dnt], a.keep[et][0][2], a.wa[et], a.att[et], nullptr, hs);
        } else {
            if (tid_b < TM) s.sa[tid_b] = a.sc[et][e0 + min(tid_b, ne - 1)];        // the coordinate scalar replaces the attention weight
            lds_barrier();
            edge_head_bwd_rows<false>(s, a, et, e0, ne, wave, lane_b, nullptr, a.keep[et][1][2], a.w3[et], nullptr, a.nvec[et], hs);
        }
        lds_barrier();
        TRAIN_STAMP(33 + 8 * br)
        acc_zero_w<NW>(acc);
        const float ex = row_dot_chunks<TPR>(s.A, s.wv + br * HS, KP / 4, tid_b);
        gemm_rows64_pre<NG, SA>(s.A, a.wpT[et][br], acc, wave, lane_b, bpre);
        if (br == 0) gemm_b_prefetch(bpre, a.wpT[et][1], wave, lane_b);
        lds_barrier();
        TRAIN_STAMP(34 + 8 * br)
        store_T_bwd(s.A, acc, ex, tid_b, wave, lane_b, pre1, ne, (a.skip & 16) ? 0 : ne);
        lds_barrier();
        TRAIN_STAMP(35 + 8 * br)
        {   // d dij += dpre1 . W1[:, 514]
            const float dot = row_dot_chunks<TPR>(s.A, s.wv + (2 + br) * HS, KP / 4, tid_b);
            if ((tid_b % TPR) == 0) s.ddij[tid_b / TPR] += dot;
        }
        {   // segmented sums over dst of dpre1 (dV) and dij * dpre1 (dVw): thread = column, rows in order
            float *m1 = a.dv_main[et][br], *c1 = a.dv_cont[et][br] + (size_t)tile_in_et * HS;
            float *m2 = a.dvw_main[et][br], *c2 = a.dvw_cont[et][br] + (size_t)tile_in_et * HS;
            float run = 0.0f, run2 = 0.0f;
            int piece = 0;
#pragma unroll 1
            for (int r0 = 0; r0 < TM; r0 += 16) {
                if (r0 >= ne) break;
                float v[16], w[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    w[i] = s.dd[r0 + i];
                    v[i] = s.A[(r0 + i) * SA + tid_b];
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    run += v[i];
                    run2 = fmaf(v[i], w[i], run2);
                    if ((endmask >> (r0 + i)) & 1ull) {
                        const bool cont = piece == 0 && first_is_cont;
                        const size_t vo = (size_t)s.dst[r0 + i] * HS;
                        (cont ? c1 : m1 + vo)[tid_b] = run;
                        (cont ? c2 : m2 + vo)[tid_b] = run2;
                        run = 0.0f;
                        run2 = 0.0f;
                        ++piece;
                    }
                }
            }
            if (wave == NW - 1) {
                const float v256 = s.A[lane_b * SA + 256];
                s.c256[lane_b] = v256;
                s.c256[TM + lane_b] = v256 * s.dd[lane_b];
            }
        }
        lds_barrier();
        TRAIN_STAMP(36 + 8 * br)
        if (wave == 0) {    // column 256: segmented inclusive scan across lanes (lane_b = row), the last lane_b of every run writes
            const unsigned long long upto = lane_b == 63 ? ~0ull : ((1ull << (lane_b + 1)) - 1ull);
            const int start = 63 - __clzll((long long)((headmask & upto) | 1ull));
            float v1 = s.c256[lane_b], v2 = s.c256[TM + lane_b];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float t1 = __shfl_up(v1, off), t2 = __shfl_up(v2, off);
                if (lane_b - off >= start) {
                    v1 += t1;
                    v2 += t2;
                }
            }
            if ((endmask >> lane_b) & 1ull) {
                const int piece = __popcll(endmask & ((1ull << lane_b) - 1ull));
                const bool cont = piece == 0 && first_is_cont;
                const size_t vo = (size_t)s.dst[lane_b] * HS;
                (cont ? a.dv_cont[et][br] + (size_t)tile_in_et * HS : a.dv_main[et][br] + vo)[256] = v1;
                (cont ? a.dvw_cont[et][br] + (size_t)tile_in_et * HS : a.dvw_main[et][br] + vo)[256] = v2;
            }
        }
        // the tile's column sums (four waves in order) through the LDS tile, which is free now
        {
            float *sc = s.A, *sw = s.A + 4 * 320;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sc[wave * 320 + 4 * lane_b + q] = hs.cs[q];
                sw[wave * 320 + 4 * lane_b + q] = hs.ws[q];
            }
            if (lane_b == 0) {
                sc[wave * 320 + 256] = hs.cs_t;
                sw[wave * 320 + 256] = hs.ws_t;
            }
            lds_barrier();
            float *p = a.part[br] + (size_t)tile * 2 * a.part_ld;
            for (int c = tid_b; c < HW; c += 256) {
                p[c] = (sw[c] + sw[320 + c]) + (sw[640 + c] + sw[960 + c]);                       // slot 0 -> head weight
                p[a.part_ld + c] = (sc[c] + sc[320 + c]) + (sc[640 + c] + sc[960 + c]);          // slot 1 -> b2
            }
            lds_barrier();
        }
        TRAIN_STAMP(37 + 8 * br)
    }
    if (tid < ne) a.sc[et][e0 + tid] = s.ddij[tid];          // d dij of both branches, over sc
    if (a.stamps && tid == 0) atomicAdd(&a.stamps[63], 1ull);
}

// dV[v] = main[v] + the continuation pieces of the tiles v's in-edges span (zeros without in-edges), the same for dVw: the pieces of
// k_egnn_edge_bwd in tile order, into the layer's gradient blocks (rows ldo apart)
// (one launch for the (edge type, branch) pairs of a layer: blockIdx.y = pair)
__global__ __launch_bounds__(256) void k_edge_pieces_set(EdgePiecesBatch b) {
    const EdgePiecesBatch::One &e = b.e[blockIdx.y];
    const float *__restrict__ m1 = e.m1, *__restrict__ c1 = e.c1, *__restrict__ m2 = e.m2, *__restrict__ c2 = e.c2;
    const int *__restrict__ rowptr = e.rowptr;
    float *__restrict__ o1 = e.o1, *__restrict__ o2 = e.o2;
    const int n = e.n, ldo = b.ldo;
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (v >= n) return;
    const int lo = rowptr[v], hi = rowptr[v + 1];
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    float t1 = 0.0f, t2 = 0.0f;
    if (hi > lo) {
        s1 = *reinterpret_cast<const f32x4 *>(m1 + (size_t)v * HS + 4 * lane);
        s2 = *reinterpret_cast<const f32x4 *>(m2 + (size_t)v * HS + 4 * lane);
        if (lane == 0) { t1 = m1[(size_t)v * HS + 256]; t2 = m2[(size_t)v * HS + 256]; }
        for (int k = (lo >> 6) + 1; k <= ((hi - 1) >> 6); ++k) {
            s1 += *reinterpret_cast<const f32x4 *>(c1 + (size_t)k * HS + 4 * lane);
            s2 += *reinterpret_cast<const f32x4 *>(c2 + (size_t)k * HS + 4 * lane);
            if (lane == 0) { t1 += c1[(size_t)k * HS + 256]; t2 += c2[(size_t)k * HS + 256]; }
        }
    }
    *reinterpret_cast<f32x4 *>(o1 + (size_t)v * ldo + 4 * lane) = s1;
    if (lane == 0) o1[(size_t)v * ldo + 256] = t1;
    if (o2) {
        *reinterpret_cast<f32x4 *>(o2 + (size_t)v * ldo + 4 * lane) = s2;
        if (lane == 0) o2[(size_t)v * ldo + 256] = t2;
    }
}

// h_neigh[v] += zinv[v] * (main[v] + the continuation pieces of the tiles its in-edges span), x_neigh likewise: the pieces of
// k_egnn_edge_train summed in tile order (one wave per destination node; nodes without in-edges are left alone)
// All edge types of a layer in one launch: blockIdx.y = destination node type; a node's row is the sum over its edge types, in edge-type
// order, and is WRITTEN (zeros for a node nothing points at): no memset of h_neigh / x_neigh in front.
__global__ __launch_bounds__(256) void k_edge_pieces_sum(EdgePiecesSumArgs a) {
    const int nt = blockIdx.y, v = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (v >= a.n[nt]) return;
    const float zi = a.zinv[nt][v];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float tacc = 0.0f;
#pragma unroll
    for (int et = 0; et < 4; ++et) {
        if (!a.live[et] || a.dst_nt[et] != nt) continue;
        const float *__restrict__ hn_main = a.hn_main[et], *__restrict__ hn_cont = a.hn_cont[et], *__restrict__ xn_main = a.xn_main[et],
                    *__restrict__ xn_cont = a.xn_cont[et];
        const int lo = a.rowptr[et][v], hi = a.rowptr[et][v + 1];
        if (hi == lo) continue;
        const int t0 = (lo >> 6) + 1, t1 = (hi - 1) >> 6;
        f32x4 s = *reinterpret_cast<const f32x4 *>(hn_main + (size_t)v * HS + 4 * lane);
        float t = lane == 0 ? hn_main[(size_t)v * HS + 256] : (lane >= 1 && lane < 4) ? xn_main[(size_t)v * 4 + lane - 1] : 0.0f;
        for (int k = t0; k <= t1; ++k) {
            s += *reinterpret_cast<const f32x4 *>(hn_cont + (size_t)k * HS + 4 * lane);
            t += lane == 0 ? hn_cont[(size_t)k * HS + 256] : (lane >= 1 && lane < 4) ? xn_cont[(size_t)k * 4 + lane - 1] : 0.0f;
        }
        acc += s * zi;
        tacc += t * zi;
    }
    *reinterpret_cast<f32x4 *>(a.hn[nt] + (size_t)v * HS + 4 * lane) = acc;
    if (lane == 0) a.hn[nt][(size_t)v * HS + 256] = tacc;
    else if (lane < 4) a.xn[nt][(size_t)v * 3 + lane - 1] = tacc;
}

// The per-step weight pack of one layer for k_egnn_edge_train: for entry (et, branch) the 257 x 257 second Linear in MFMA fragment
// order with its bias as row BIAS_K (wp, wx), the radial column of the first Linear as a row (wr); per et the two head rows (watt with
// its bias at ATT_BIAS_AT, w3).
__global__ void k_edge_train_pack(EdgePackTab t) {
    const EdgePackEntry &e = t.e[blockIdx.y];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < WP_FLOATS) {
        const int j = idx & 3, nt = (idx >> 2) & 1, lane = (idx >> 3) & 63, wave = (idx >> 9) & 3, g = idx >> 11;
        const int k = 8 * g + 4 * (lane >> 5) + j;
        const int n = 64 * wave + 32 * nt + (lane & 31);
        if (t.transposed) e.wp[idx] = k < HW ? e.W2[(size_t)k * HW + n] : 0.0f;                     // "weight" of dpre1 = dpre2 W2: M[n][k] = W2[k][n], no bias
        else e.wp[idx] = k < HW ? e.W2[(size_t)n * HW + k] : k == BIAS_K ? e.b2[n] : 0.0f;
    } else if (idx < WP_FLOATS + HS) {
        const int k = idx - WP_FLOATS;
        if (t.transposed) e.wx[k] = k < HW ? e.W2[(size_t)k * HW + 256] : 0.0f;
        else e.wx[k] = k < HW ? e.W2[(size_t)256 * HW + k] : k == BIAS_K ? e.b2[256] : 0.0f;
    } else if (idx < WP_FLOATS + 2 * HS) {
        const int k = idx - WP_FLOATS - HS;
        e.wr[k] = k < HW ? e.W1[(size_t)k * (2 * HW + 1) + 2 * HW] : 0.0f;
    } else if (idx < WP_FLOATS + 3 * HS) {
        const int k = idx - WP_FLOATS - 2 * HS;
        if (e.head_out) e.head_out[k] = k < HW ? e.head[k] : (k == ATT_BIAS_AT && e.head_b) ? e.head_b[0] : 0.0f;
    }
}

// ---- f16x2 form of the fused edge kernel (opt-in: KPD_GEMM=f16x2 / "gemm=f16x2") ----------------------------------------------
// Same phases and the same fp32 epilogues as k_egnn_edge<4>; the two 257 x 257 products run as three f16 MFMA products of
// hi / lo operand planes with fp32 accumulation (mfma_core.h, gemm_rows64_h).  The A tile is written as two f16 planes by the
// A-build; T (fp32) reuses their LDS region after the GEMM exactly as it reuses the fp32 A tile in the exact kernel.
constexpr int EDGE_H_REGION0_FLOATS = TM * SAH;        // two planes of TM x SAH halves = TM * SAH floats (>= the TM x SA fp32 T tile)
static_assert(EDGE_H_REGION0_FLOATS >= TM * SA, "T tile must fit the A planes' region");

__device__ __forceinline__ EdgeSmem edge_smem_h(float *smem) {
    EdgeSmem s;
    s.A = smem;
    s.src = reinterpret_cast<int *>(smem + EDGE_H_REGION0_FLOATS);
    s.dst = s.src + TM;
    s.d = reinterpret_cast<float *>(s.dst + TM);
    s.xd = s.d + TM;
    s.att = s.xd + 3 * TM;
    s.mx = s.att + TM;
    s.wv = s.mx + 3 * TM;
    s.misc = reinterpret_cast<int *>(s.wv + 2 * HS);
    return s;
}

// A[r][:] = SiLU(Ps[src_r] + Pd[dst_r] + d_r w_r) as f16 hi / lo planes; columns 264..271 (K padding of the 16-wide k-steps) zero
// BATCH_D: read the wave's RPW distances with RPW / 4 broadcast ds_read_b128 up front instead of one ds_read_b32 (and an LDS round trip)
// per row.  OFF in production: with the batched read at both call sites of k_egnn_edge_h (0.394 instead of 0.402 ms) the FIRST forward
// of a process differed from all later ones in a few x pieces (first segments of ll / kl tiles, ~1 % of the piece) -- at either site
// alone, or with the per-row reads, it does not (profiles/tools/repro_layer.py, repro_pieces.py; six fresh processes each).  The
// batched form is semantically identical, so the cause is a timing-dependent hazard that was not found; the arrangement that has
// never shown it is the one that ships.
#ifndef KPD_H_BATCH_D
#define KPD_H_BATCH_D false
#endif
template <int NW, bool BATCH_D>
__device__ __forceinline__ void edge_gather_finish_h(const EdgeGather<NW> &g, const EdgeSmem &s, _Float16 *Ah, const float *__restrict__ wr,
                                                     int wave, int lane, float *dbg2 = nullptr) {
    constexpr int RPW = TM / NW;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const f32x4 w0 = reinterpret_cast<const f32x4 *>(wr)[lane];
    // the wave's RPW distances in one batch of broadcast reads (a read + wait per row costs an LDS round trip each, and the per-row
    // addresses were being kept live -- spilled -- across the GEMM)
    f32x4 dv[RPW / 4];
#ifdef KPD_HZ_VM0      // hazard hunt (profiles/tools/hz_variant.sh): every gathered row has landed before the first one is consumed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#pragma unroll
    for (int i = 0; i < RPW / 4; ++i) dv[i] = BATCH_D ? *reinterpret_cast<const f32x4 *>(s.d + wave * RPW + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef KPD_HZ_LGKM0    // hazard hunt: the batched distances have landed before anything else is issued
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave * RPW + rr;
#ifdef KPD_EDGE_DBG
        if (dbg2 && rr < 3) {       // the operands exactly as this row consumes them
            float *o = dbg2 + ((size_t)(rr * 4 + wave) * 64 + lane) * 12;
            *reinterpret_cast<f32x4 *>(o) = g.ps[rr];
            *reinterpret_cast<f32x4 *>(o + 4) = g.pd[rr];
            o[8] = BATCH_D ? dv[rr >> 2][rr & 3] : s.d[r]; o[9] = w0[0]; o[10] = __builtin_bit_cast(float, s.src[r]); o[11] = __builtin_bit_cast(float, s.dst[r]);
        }
#endif
        f32x4 v = g.ps[rr] + g.pd[rr] + (BATCH_D ? dv[rr >> 2][rr & 3] : s.d[r]) * w0;
        v[0] = silu_pre_x64(v[0]); v[1] = silu_pre_x64(v[1]); v[2] = silu_pre_x64(v[2]); v[3] = silu_pre_x64(v[3]);
        unsigned h0, h1, l0, l1;
        split_pair(v[0], v[1], h0, l0);
        split_pair(v[2], v[3], h1, l1);
        *reinterpret_cast<u32x2 *>(Ah + r * SAH + 4 * lane) = u32x2{h0, h1};
        *reinterpret_cast<u32x2 *>(Ah + PLANE_H + r * SAH + 4 * lane) = u32x2{l0, l1};
    }
    if (lane < 4 * RPW) {   // columns 256 .. 271 of the wave's rows: lane = row * 4 + chunk of four columns
        const int r = wave * RPW + (lane >> 2), c = lane & 3;
        f32x4 u = {0.f, 0.f, 0.f, 0.f};
        if (c < 2) {
            const f32x4 w1 = reinterpret_cast<const f32x4 *>(wr)[64 + c];
            u = g.tps + g.tpd + s.d[r] * w1;
            u[0] = silu_pre_x64(u[0]); u[1] = silu_pre_x64(u[1]); u[2] = silu_pre_x64(u[2]); u[3] = silu_pre_x64(u[3]);
            if (c == (BIAS_K - 256) / 4) u[(BIAS_K - 256) % 4] = H_SCALE_A;
        }
        unsigned h0, h1, l0, l1;
        split_pair(u[0], u[1], h0, l0);
        split_pair(u[2], u[3], h1, l1);
        *reinterpret_cast<u32x2 *>(Ah + r * SAH + 256 + 4 * c) = u32x2{h0, h1};
        *reinterpret_cast<u32x2 *>(Ah + PLANE_H + r * SAH + 256 + 4 * c) = u32x2{l0, l1};
    }
}

template <int NT>
__device__ __forceinline__ void unscale_acc(f32x16 (&acc)[2][NT], float &ex) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] *= H_UNSCALE;
    ex *= H_UNSCALE;
}

template <int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 4) void k_egnn_edge_h(EdgeArgs a) {
    constexpr int TPR = NW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const EdgeSmem s = edge_smem_h(smem);
    _Float16 *Ah = reinterpret_cast<_Float16 *>(smem);                 // two f16 planes of the A tile; T (fp32) reuses the region
    // W2[256, :] of edge_mlp then coord_mlp x H_SCALE_W as f16 planes: [hi 272 | lo 272] each
    _Float16 *wxs = reinterpret_cast<_Float16 *>(s.misc + 8);
    // (the wave index as a scalar, as in k_node_update8: the per-row LDS addresses of the A-builds become scalar base + immediate offset
    // instead of one VGPR per row kept live -- spilled -- across the GEMMs)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    unsigned long long t_prev_ = a.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    // tile decode (XCD-aware: consecutive tiles -- neighbouring edges of one complex, which
    // share P rows -- go to the same XCD / L2)
    const int T = a.meta[8];
    const int chunk = (T + 7) >> 3;
    const int bi = blockIdx.x >> 3;
    if (bi >= chunk) return;
    const int tile = (blockIdx.x & 7) * chunk + bi;
    if (tile >= T) return;
    int et = 0;
#pragma unroll
    for (int e = 1; e < 4; ++e)
        if (tile >= a.meta[4 + e]) et = e;
    const int tile_in_et = tile - a.meta[4 + et];
    const int e0 = tile_in_et * TM;
    const int ne = min(TM, a.meta[et] - e0);
    const int snt = a.src_nt[et], dnt = a.dst_nt[et];
    const int *__restrict__ esrc = a.src[et];
    const int *__restrict__ edst = a.dst[et];

    // phase 0: edge endpoints and geometry (dynamics.py:160-169, 209-217); head weights to LDS
    if (tid < TM) {
        const int e = e0 + min(tid, ne - 1);
        const int u = esrc[e], v = edst[e];
        s.src[tid] = (int)((unsigned)u * PROW_B);        // byte offsets of the nodes' P rows
        s.dst[tid] = (int)((unsigned)v * PROW_B);
        const float *xs = a.x[snt] + (size_t)u * 3, *xd = a.x[dnt] + (size_t)v * 3;
        const float dx = xs[0] - xd[0], dy = xs[1] - xd[1], dz = xs[2] - xd[2];
        const float d = sqrtf(dx * dx + dy * dy + dz * dz);
        const float inv = 1.0f / (d + 1.0f);
        s.d[tid] = d;
        s.xd[3 * tid] = dx * inv;
        s.xd[3 * tid + 1] = dy * inv;
        s.xd[3 * tid + 2] = dz * inv;
        // run structure of the dst-sorted tile as two 64-bit masks (wave 0 == rows 0..63)
        const int vprev = tid > 0 ? edst[e0 + min(tid - 1, ne - 1)] : (e0 > 0 ? edst[e0 - 1] : -1);
        const int vnext = tid + 1 < ne ? edst[e0 + tid + 1] : -2;
        const unsigned long long heads = __ballot(tid < ne && (tid == 0 || vprev != v));
        const unsigned long long ends = __ballot(tid < ne && vnext != v);
        if (tid == 0) {
            s.misc[0] = (vprev == v) ? 1 : 0;
            s.misc[2] = (int)(ends & 0xffffffffu);
            s.misc[3] = (int)(ends >> 32);
            s.misc[4] = (int)(heads & 0xffffffffu);
            s.misc[5] = (int)(heads >> 32);
        }
    } else {
        // soft-attention and coordinate-head rows (used on T) and row 256 of either W2 scaled like the weight planes (used on A)
        for (int i = tid - TM; i < 2 * 66; i += 64 * NW - TM) {
            const int which = i / 66, j = i - which * 66;
            const float *row = which == 0 ? a.watt[et] : a.w3[et];
            reinterpret_cast<f32x4 *>(s.wv + which * HS)[j] = reinterpret_cast<const f32x4 *>(row)[j];
        }
        for (int i = tid - TM; i < 2 * 272; i += 64 * NW - TM) {
            const int which = i / 272, kk = i - which * 272;
            const float w = kk < KP ? H_SCALE_W * (which == 0 ? a.wx_e[et] : a.wx_c[et])[kk] : 0.0f;
            const _Float16 hi = (_Float16)w;
            wxs[which * 544 + kk] = hi;
            wxs[which * 544 + 272 + kk] = (_Float16)(w - (float)hi);
        }
    }
    lds_barrier();
    KPD_STAMP(0)

    const float *Ps = a.P[snt] + (size_t)a.src_slot[et] * HS;
    const float *Pd = a.P[dnt] + (size_t)a.dst_slot[et] * HS;
    const int first_is_cont = s.misc[0];
    const unsigned long long endmask =
        ((unsigned long long)(unsigned)s.misc[3] << 32) | (unsigned long long)(unsigned)s.misc[2];
    f32x16 acc[2][WaveCols<NW>::NT];
    float ex;

    // ---- feature messages: m = edge_mlp(f); msg_h = m * sigmoid(att(m)) (dynamics.py:111-112)
    constexpr int abl = 0;
    {
        EdgeGather<NW> ge;
        edge_gather_issue<NW>(ge, s, Ps, Pd, wave, lane);
        edge_gather_finish_h<NW, KPD_H_BATCH_D>(ge, s, Ah, a.wr_e[et], wave, lane);
    }
    lds_barrier();
    KPD_STAMP(1)
    acc_zero_w<NW>(acc);
    ex = row_dot_h2<TPR>(Ah, wxs, tid);
#ifndef KPD_HZ_NOGEMM   // hazard hunt: the kernel without its matrix products (results meaningless, run-to-run equality still telling)
    if constexpr (NW == 4) gemm_rows64_h(Ah, a.wh_e[et], acc, wave, lane);
    else gemm_rows64_h8(Ah, a.wh_e[et], acc, wave, lane);
#endif
    unscale_acc(acc, ex);
    lds_barrier();
    KPD_STAMP(2)
    if (!(abl & 4)) store_T_silu_w<NW, true>(s.A, acc, ex, a.b_e[et], tid, wave, lane);
    else if (acc[0][0][0] == 12345.0f) s.A[tid] = acc[1][NW == 4 ? 1 : 0][3] + acc[0][NW == 4 ? 1 : 0][5] + acc[1][0][7];
    EdgeGather<NW> gc;
    {   // the coordinate branch's P rows start travelling now; consumed after the segmented sum below
        edge_gather_issue<NW>(gc, s, Ps + HS, Pd + HS, wave, lane);
        __builtin_amdgcn_sched_barrier(0);
    }
    lds_barrier();
    KPD_STAMP(3)
    if (!(abl & 4)) {
        float dot = row_dot_chunks<TPR>(s.A, s.wv, 64, tid);
        const int row = tid / TPR;
        if ((tid % TPR) == 0) {
            dot = fmaf(s.A[row * SA + 256], s.wv[256], dot);
            // T holds c * m, w_att carries 1 / c; the returned weight carries 1 / c so that T * att = m * sigmoid(.)
            s.att[row] = row < ne ? sigmoidf_(dot + s.wv[ATT_BIAS_AT]) * (1.0f / SILU_C) : 0.0f;
        }
    }
    lds_barrier();
    KPD_STAMP(4)
    if (!(abl & 4)) {
        // segmented sum over dst (dynamics.py:182-185): thread = column, rows in order; the run
        // boundaries are wave-uniform (endmask), LDS reads are issued 16 rows at a time
        float *hmain = a.hn_main[et], *hcont = a.hn_cont[et] + (size_t)tile_in_et * HS;
        if (tid < 256) {
            float run = 0.0f;
            int piece = 0;
#pragma unroll 1
            for (int r0 = 0; r0 < TM; r0 += 16) {
                if (r0 >= ne) break;
                float v[16], w[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    w[i] = s.att[r0 + i];
                    v[i] = s.A[(r0 + i) * SA + tid];
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    run = fmaf(v[i], w[i], run);
                    if ((endmask >> (r0 + i)) & 1ull) {
                        float *out = (piece == 0 && first_is_cont) ? hcont : hmain + ((unsigned)s.dst[r0 + i] / (unsigned)(NSLOT * 4));
                        out[tid] = run;
                        run = 0.0f;
                        ++piece;
                    }
                }
            }
        }
        // column 256: lane = row on the last wave, segmented inclusive scan across lanes
        if (wave == NW - 1) {
            const unsigned long long heads =
                ((unsigned long long)(unsigned)s.misc[5] << 32) | (unsigned long long)(unsigned)s.misc[4];
            const unsigned long long upto = lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull);
            const int start = 63 - __clzll((long long)((heads & upto) | 1ull));
            float v = s.A[lane * SA + 256] * s.att[lane];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const float t = __shfl_up(v, off);
                if (lane - off >= start) v += t;
            }
            if ((endmask >> lane) & 1ull) {
                const int pc = __popcll(endmask & ((1ull << lane) - 1ull));
                float *out = (pc == 0 && first_is_cont) ? hcont : hmain + ((unsigned)s.dst[lane] / (unsigned)(NSLOT * 4));
                out[256] = v;
            }
        }
    }
    lds_barrier();
    KPD_STAMP(5)

    // ---- coordinate messages: msg_x = tanh(coord_mlp(f)) * x_diff * range (dynamics.py:113-120)
#ifdef KPD_EDGE_DBG
    edge_gather_finish_h<NW, KPD_H_BATCH_D>(gc, s, Ah, a.wr_c[et], wave, lane, a.dbg ? a.dbg + (size_t)a.meta[8] * TM * 4 + (size_t)tile * (3 * 4 * 64 * 12) : nullptr);
#else
    edge_gather_finish_h<NW, KPD_H_BATCH_D>(gc, s, Ah, a.wr_c[et], wave, lane);
#endif
    lds_barrier();
    KPD_STAMP(6)
    acc_zero_w<NW>(acc);
    ex = row_dot_h2<TPR>(Ah, wxs + 544, tid);
#ifndef KPD_HZ_NOGEMM   // hazard hunt: the kernel without its matrix products (results meaningless, run-to-run equality still telling)
    if constexpr (NW == 4) gemm_rows64_h(Ah, a.wh_c[et], acc, wave, lane);
    else gemm_rows64_h8(Ah, a.wh_c[et], acc, wave, lane);
#endif
    unscale_acc(acc, ex);
    lds_barrier();
    KPD_STAMP(7)
    if (!(abl & 4)) store_T_silu_w<NW, true>(s.A, acc, ex, a.b_c[et], tid, wave, lane);
    else if (acc[0][0][0] == 12345.0f) s.A[tid] = acc[1][NW == 4 ? 1 : 0][3] + acc[0][NW == 4 ? 1 : 0][5] + acc[1][0][7];
    lds_barrier();
    KPD_STAMP(8)
    if (!(abl & 4)) {
        float dot = row_dot_chunks<TPR>(s.A, s.wv + HS, 64, tid);
        const int row = tid / TPR;
        if ((tid % TPR) == 0) {
            dot = fmaf(s.A[row * SA + 256], s.wv[HS + 256], dot);
            float c = a.use_tanh ? tanhf(dot) * a.coords_range : dot;
            if (row >= ne) c = 0.0f;
#ifdef KPD_EDGE_DBG
            if (a.dbg) {
                float *o = a.dbg + ((size_t)tile * TM + row) * 4;
                o[0] = dot; o[1] = s.d[row]; o[2] = s.A[row * SA + 256]; o[3] = s.A[row * SA + 7];
            }
#endif
            s.mx[3 * row] = c * s.xd[3 * row];
            s.mx[3 * row + 1] = c * s.xd[3 * row + 1];
            s.mx[3 * row + 2] = c * s.xd[3 * row + 2];
        }
    }
    lds_barrier();
    KPD_STAMP(9)
    if (wave == 0 && !(abl & 4)) {
        // segmented inclusive scan across lanes (lane = row), then the last lane of every run writes
        const unsigned long long heads =
            ((unsigned long long)(unsigned)s.misc[5] << 32) | (unsigned long long)(unsigned)s.misc[4];
        const unsigned long long upto = lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull);
        const int start = 63 - __clzll((long long)((heads & upto) | 1ull));
        float vx = s.mx[3 * lane], vy = s.mx[3 * lane + 1], vz = s.mx[3 * lane + 2];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const float tx = __shfl_up(vx, off), ty = __shfl_up(vy, off), tz = __shfl_up(vz, off);
            if (lane - off >= start) {
                vx += tx;
                vy += ty;
                vz += tz;
            }
        }
        if ((endmask >> lane) & 1ull) {
            const int piece = __popcll(endmask & ((1ull << lane) - 1ull));
            float *out = (piece == 0 && first_is_cont) ? a.xn_cont[et] + (size_t)tile_in_et * 4
                                                       : a.xn_main[et] + (size_t)((unsigned)s.dst[lane] / PROW_B) * 4;
            out[0] = vx;
            out[1] = vy;
            out[2] = vz;
        }
    }
    KPD_STAMP(10)
}



// ---- node update: 8 waves per 32-node tile (the projections of the next layer run in k_proj_ws, egnn_chain.hip) ---------
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_node_update8(NodeLayerPair p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *A = smem;
    float *s_z = smem + TN * SA;
    float *s_mean = s_z + TN;
    float *s_rstd = s_mean + TN;
    float *s_wx = s_rstd + TN;           // [3][HS]: row 256 of the three weight matrices, the vectors of the per-row dots (from global memory
                                         // inside row_dot_chunks they were five dependent L2 round trips per call)
    // the wave index as a scalar: every row-wise loop below addresses rows by wave, so row pointers, the CSR bounds of the h_neigh
    // gather and its branches are scalar work
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int which = blockIdx.x >= p.tiles0 ? 1 : 0;
    const NodeLayerArgs &L = p.nt[which];
    const NodeArgs &a = L.u;
    const int node0 = (blockIdx.x - (which ? p.tiles0 : 0)) * TN;
    constexpr int RPW = TN / 8;          // rows per wave in the copy loops
    constexpr int TPR = 512 / TN;        // threads per row in the row-wise passes
    unsigned long long t_prev_ = p.stamps ? __builtin_amdgcn_s_memtime() : 0ull;
#define NL_STAMP(idx)                                                                      \
    if (p.stamps && tid == 0) {                                                            \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                      \
        atomicAdd(&p.stamps[idx], (unsigned long long)(now_ - t_prev_));                   \
        t_prev_ = now_;                                                                    \
    }

    // rows of h to the tile: all of a wave's loads in flight together (the h array is padded to a whole tile; rows >= n hold zeros)
    auto load_h = [&]() {
        f32x4 val[RPW], val2[RPW];
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(a.h + (size_t)(node0 + wave * RPW + rr) * HS);
            val[rr] = src[lane];
            val2[rr] = src[64 + (lane & 1)];
        }
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = wave * RPW + rr;
            *reinterpret_cast<f32x4 *>(A + r * SA + 4 * lane) = val[rr];
            if (lane < 2) *reinterpret_cast<f32x4 *>(A + r * SA + 256 + 4 * lane) = val2[rr];
        }
    };
    // edge types whose pieces a node sums: n_in is 1 or 2; with one, the second slot repeats the first and is masked out
    const int two = a.n_in > 1 ? 1 : 0;
    const int *__restrict__ rp0 = a.rowptr[0], *__restrict__ rp1 = a.rowptr[two];

    f32x16 acc;
    {
        // coordinates: x' = x + x_neigh / z (dynamics.py:190-192, 206)
        if (tid < TN) {
            const int v = node0 + tid;
            float z = 1.0f;
            if (v < a.n) {
                // bounds and main pieces of both edge types requested together (a main piece exists for every node; it is used only where
                // the node has in-edges of the type), continuation pieces -- runs that cross a tile boundary -- in a loop behind them
                const int lo0 = rp0[v], hi0 = rp0[v + 1], lo1 = rp1[v], hi1 = two ? rp1[v + 1] : lo1;
                const float *pm0 = a.xn_main[0] + (size_t)v * 4, *pm1 = a.xn_main[two] + (size_t)v * 4;
                const float m0x = pm0[0], m0y = pm0[1], m0z = pm0[2], m1x = pm1[0], m1y = pm1[1], m1z = pm1[2];
                z = a.z[a.bidx[v]];
                float sx = 0.f, sy = 0.f, sz = 0.f;
                if (hi0 > lo0) {
                    sx += m0x; sy += m0y; sz += m0z;
                    for (int t = (lo0 >> a.tile_shift) + 1; t <= ((hi0 - 1) >> a.tile_shift); ++t) {
                        const float *q = a.xn_cont[0] + (size_t)t * 4;
                        sx += q[0]; sy += q[1]; sz += q[2];
                    }
                }
                if (hi1 > lo1) {
                    sx += m1x; sy += m1y; sz += m1z;
                    for (int t = (lo1 >> a.tile_shift) + 1; t <= ((hi1 - 1) >> a.tile_shift); ++t) {
                        const float *q = a.xn_cont[two] + (size_t)t * 4;
                        sx += q[0]; sy += q[1]; sz += q[2];
                    }
                }
                float *xv = a.x + (size_t)v * 3;
                xv[0] += sx / z; xv[1] += sy / z; xv[2] += sz / z;
            }
            s_z[tid] = z;
        } else if (tid >= 512 - 3 * 66) {
            const int i = tid - (512 - 3 * 66), which = i / 66, j = i - which * 66;
            const float *row = which == 0 ? a.wx_a : which == 1 ? a.wx_b : a.wx_2;
            reinterpret_cast<f32x4 *>(s_wx + which * HS)[j] = reinterpret_cast<const f32x4 *>(row)[j];
        }
        // GEMM 1a: W[:, :257] . h
        load_h();
        lds_barrier();
        NL_STAMP(0)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        gemm_rows32_t8<NG, SA>(A, a.wp_a, acc, wave, lane);
        float ex = row_dot_chunks<TPR>(A, s_wx, KP / 4, tid);
        lds_barrier();
        NL_STAMP(1)
        // GEMM 1b: + W[:, 257:] . (h_neigh / z); h_neigh = sum of segment pieces over the incoming edge
        // types in fixed order (multi_update_all cross_reducer='sum')
        {
            // (the bounds of a wave's four rows are scalar loads, the main pieces of both edge types of all four rows travel together; before,
            //  every row and edge type was a chain bounds -> branch -> piece, a quarter of the kernel's time)
            constexpr int GB = 2;        // rows per batch: 16 registers per row next to the live accumulators, in a kernel held to 80
#pragma unroll
            for (int r0 = 0; r0 < RPW; r0 += GB) {
                int lo0[GB], hi0[GB], lo1[GB], hi1[GB];
                f32x4 m0[GB], m1[GB];
                float t0[GB], t1[GB];            // columns 256 .. 263 of the row: one float on each of eight lanes
#pragma unroll
                for (int rr = 0; rr < GB; ++rr) {
                    const int v = min(node0 + wave * RPW + r0 + rr, a.n - 1);
                    lo0[rr] = rp0[v]; hi0[rr] = rp0[v + 1];
                    lo1[rr] = rp1[v]; hi1[rr] = two ? rp1[v + 1] : lo1[rr];
                    const f32x4 *p0 = reinterpret_cast<const f32x4 *>(a.hn_main[0] + (size_t)v * HS);
                    const f32x4 *p1 = reinterpret_cast<const f32x4 *>(a.hn_main[two] + (size_t)v * HS);
                    m0[rr] = p0[lane]; t0[rr] = reinterpret_cast<const float *>(p0)[256 + (lane & 7)];
                    m1[rr] = p1[lane]; t1[rr] = reinterpret_cast<const float *>(p1)[256 + (lane & 7)];
                }
#pragma unroll
                for (int rr = 0; rr < GB; ++rr) {
                    const int r = wave * RPW + r0 + rr, v = node0 + r;
                    f32x4 val = {0.f, 0.f, 0.f, 0.f};
                    float val2 = 0.0f;
                    if (v < a.n) {
                        if (hi0[rr] > lo0[rr]) {
                            val += m0[rr];
                            val2 += t0[rr];
                            for (int t = (lo0[rr] >> a.tile_shift) + 1; t <= ((hi0[rr] - 1) >> a.tile_shift); ++t) {
                                const f32x4 *q = reinterpret_cast<const f32x4 *>(a.hn_cont[0] + (size_t)t * HS);
                                val += q[lane];
                                val2 += reinterpret_cast<const float *>(q)[256 + (lane & 7)];
                            }
                        }
                        if (hi1[rr] > lo1[rr]) {
                            val += m1[rr];
                            val2 += t1[rr];
                            for (int t = (lo1[rr] >> a.tile_shift) + 1; t <= ((hi1[rr] - 1) >> a.tile_shift); ++t) {
                                const f32x4 *q = reinterpret_cast<const f32x4 *>(a.hn_cont[two] + (size_t)t * HS);
                                val += q[lane];
                                val2 += reinterpret_cast<const float *>(q)[256 + (lane & 7)];
                            }
                        }
                        const float z = s_z[r];
                        val /= z;
                        val2 /= z;
                    }
                    *reinterpret_cast<f32x4 *>(A + r * SA + 4 * lane) = val;
                    if (lane < 8) A[r * SA + 256 + lane] = val2;
                }
            }
        }
        lds_barrier();
        NL_STAMP(2)
        gemm_rows32_t8<NG, SA>(A, a.wp_b, acc, wave, lane);
        ex += row_dot_chunks<TPR>(A, s_wx + HS, KP / 4, tid);
        lds_barrier();
        NL_STAMP(3)
        // hidden = SiLU(. + b0) -> T (pad columns 257..263 stay 0 from the h_neigh tile)
        {
            const int col = 32 * wave + (lane & 31);
            const float bb = a.b0[col];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) A[acc_row32(reg, lane) * SA + col] = silu(acc[reg] + bb);
        }
        if ((tid % TPR) == 0) A[(tid / TPR) * SA + 256] = silu(ex + a.b0[256]);
        lds_barrier();
        NL_STAMP(4)
        // GEMM 2 + bias + residual (dynamics.py:201-203)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        gemm_rows32_t8<NG, SA>(A, a.wp_2, acc, wave, lane);
        ex = row_dot_chunks<TPR>(A, s_wx + 2 * HS, KP / 4, tid);
        lds_barrier();
        NL_STAMP(5)
        {
            const int col = 32 * wave + (lane & 31);
            const float bb = a.b2[col];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) A[acc_row32(reg, lane) * SA + col] = acc[reg] + bb;
        }
        if ((tid % TPR) == 0) A[(tid / TPR) * SA + 256] = ex + a.b2[256];
        lds_barrier();
        // residual h (row-wise, coalesced; the h array is padded to a whole tile, rows >= n read zeros)
#pragma unroll 4
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = wave * RPW + rr;
            const f32x4 *src = reinterpret_cast<const f32x4 *>(a.h + (size_t)(node0 + r) * HS);
            *reinterpret_cast<f32x4 *>(A + r * SA + 4 * lane) += src[lane];
            if (lane == 0) A[r * SA + 256] += a.h[(size_t)(node0 + r) * HS + 256];
        }
        lds_barrier();
        NL_STAMP(6)
        // LayerNorm(257) (dynamics.py:81-87, 204), biased variance, eps = 1e-5
        if (a.norm) {
            const int row = tid / TPR, q = tid % TPR;
            const float *tr = A + row * SA + q;
            float sum = 0.0f;
            for (int i = 0; i < 256 / TPR; ++i) sum += tr[TPR * i];
            if (q == 0) sum += A[row * SA + 256];
#pragma unroll
            for (int o = 1; o < TPR; o <<= 1) sum += __shfl_xor(sum, o);
            const float mean = sum * a.ln_inv_n;
            float var = 0.0f;
            for (int i = 0; i < 256 / TPR; ++i) {
                const float dlt = tr[TPR * i] - mean;
                var = fmaf(dlt, dlt, var);
            }
            if (q == 0) {
                const float dlt = A[row * SA + 256] - mean;
                var = fmaf(dlt, dlt, var);
            }
#pragma unroll
            for (int o = 1; o < TPR; o <<= 1) var += __shfl_xor(var, o);
            if (q == 0) {
                s_mean[row] = mean;
                s_rstd[row] = 1.0f / sqrtf((var - a.ln_pad * mean * mean) * a.ln_inv_n + 1e-5f);      // pad columns (hidden_nf < 256) hold 0: take their (0 - mean)^2 out
            }
        }
        lds_barrier();
        NL_STAMP(7)
        // normalise in place (the tile becomes the A operand of the projections) and write h' back
#pragma unroll 2
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = wave * RPW + rr, v = node0 + r;
            f32x4 val = *reinterpret_cast<const f32x4 *>(A + r * SA + 4 * lane);
            float last = A[r * SA + 256];
            if (a.norm) {
                const float mean = s_mean[r], rstd = s_rstd[r];
                const f32x4 w = reinterpret_cast<const f32x4 *>(a.ln_w)[lane];
                const f32x4 b = reinterpret_cast<const f32x4 *>(a.ln_b)[lane];
                val = (val - mean) * rstd * w + b;
                last = (last - mean) * rstd * a.ln_w[256] + a.ln_b[256];
            }
            if (v >= a.n) {
                val = f32x4{0.f, 0.f, 0.f, 0.f};
                last = 0.0f;
            }
            *reinterpret_cast<f32x4 *>(A + r * SA + 4 * lane) = val;
            const f32x4 t = {last, 0.f, 0.f, 0.f};
            if (lane == 0) *reinterpret_cast<f32x4 *>(A + r * SA + 256) = t;
            if (lane == 1) *reinterpret_cast<f32x4 *>(A + r * SA + 260) = f32x4{0.f, 0.f, 0.f, 0.f};
            if (v < a.n) {
                f32x4 *dst = reinterpret_cast<f32x4 *>(a.h + (size_t)v * HS);
                dst[lane] = val;
                if (lane == 0) dst[64] = t;
            }
        }
        lds_barrier();
        NL_STAMP(8)
    }
#undef NL_STAMP
}


// ---- node update, f16x2 split (opt-in gemm mode 1; same phases as k_node_update8) ----------------------------------------------
// The three GEMMs run on v_mfma_f32_32x32x16_f16 over hi / lo planes (gemm_rows32_h8); the tile is kept as two f16 planes of
// 2^6 x value while it is a GEMM operand and as fp32 (same LDS region) from the output of GEMM 2 on: bias, residual, LayerNorm and
// the h' written back are fp32 exactly as in the exact kernel.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 6))) void k_node_update8_h(NodeLayerPair p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *A = smem;                                                   // fp32 view [TN][SA] (phases after GEMM 2)
    _Float16 *Ah = reinterpret_cast<_Float16 *>(smem);                 // plane view [2][TN][SAH]
    constexpr int PN = TN * SAH;
    float *s_z = smem + NODE_H_TILE_FLOATS;
    float *s_mean = s_z + TN;
    float *s_rstd = s_mean + TN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int which = blockIdx.x >= p.tiles0 ? 1 : 0;
    const NodeLayerArgs &L = p.nt[which];
    const NodeArgs &a = L.u;
    const int node0 = (blockIdx.x - (which ? p.tiles0 : 0)) * TN;
    constexpr int RPW = TN / 8;
    constexpr int TPR = 512 / TN;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

    // row r of the tile <- 2^6 (val | val2) as planes; lanes 2, 3 zero the K padding 264 .. 271
    auto put_row = [&](int r, const f32x4 &val, const f32x4 &val2) {
        unsigned h0, h1, l0, l1;
        split_pair(H_SCALE_A * val[0], H_SCALE_A * val[1], h0, l0);
        split_pair(H_SCALE_A * val[2], H_SCALE_A * val[3], h1, l1);
        *reinterpret_cast<u32x2 *>(Ah + r * SAH + 4 * lane) = u32x2{h0, h1};
        *reinterpret_cast<u32x2 *>(Ah + PN + r * SAH + 4 * lane) = u32x2{l0, l1};
        if (lane < 4) {
            split_pair(H_SCALE_A * val2[0], H_SCALE_A * val2[1], h0, l0);
            split_pair(H_SCALE_A * val2[2], H_SCALE_A * val2[3], h1, l1);
            *reinterpret_cast<u32x2 *>(Ah + r * SAH + 256 + 4 * lane) = u32x2{h0, h1};
            *reinterpret_cast<u32x2 *>(Ah + PN + r * SAH + 256 + 4 * lane) = u32x2{l0, l1};
        }
    };
    auto put_elem = [&](int row, int col, float v) {
        const float sv = H_SCALE_A * v;
        const _Float16 hi = (_Float16)sv;
        Ah[row * SAH + col] = hi;
        Ah[PN + row * SAH + col] = (_Float16)(sv - (float)hi);
    };

    f32x16 acc;
    // coordinates: x' = x + x_neigh / z (dynamics.py:190-192, 206)
    if (tid < TN) {
        const int v = node0 + tid;
        float z = 1.0f;
        if (v < a.n) {
            z = a.z[a.bidx[v]];
            float sx = 0.f, sy = 0.f, sz = 0.f;
            for (int i = 0; i < a.n_in; ++i) {
                const int lo = a.rowptr[i][v], hi = a.rowptr[i][v + 1];
                if (hi > lo) {
                    const float *pm = a.xn_main[i] + (size_t)v * 4;
                    sx += pm[0]; sy += pm[1]; sz += pm[2];
                    for (int t = (lo >> a.tile_shift) + 1; t <= ((hi - 1) >> a.tile_shift); ++t) {
                        const float *q = a.xn_cont[i] + (size_t)t * 4;
                        sx += q[0]; sy += q[1]; sz += q[2];
                    }
                }
            }
            float *xv = a.x + (size_t)v * 3;
            xv[0] += sx / z; xv[1] += sy / z; xv[2] += sz / z;
        }
        s_z[tid] = z;
    }
    // GEMM 1a: W[:, :257] . h
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave * RPW + rr, v = node0 + r;
        f32x4 val = {0.f, 0.f, 0.f, 0.f}, val2 = {0.f, 0.f, 0.f, 0.f};
        if (v < a.n) {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(a.h + (size_t)v * HS);
            val = src[lane];
            if (lane < 2) val2 = src[64 + lane];
        }
        put_row(r, val, val2);
    }
    lds_barrier();
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    gemm_rows32_h8(Ah, PN, a.wh_a, acc, wave, lane);
    float ex = row_dot_planes<TPR>(Ah, PN, a.wx_a, tid);
    lds_barrier();
    // GEMM 1b: + W[:, 257:] . (h_neigh / z); h_neigh = sum of segment pieces over the incoming edge types in fixed order
#pragma unroll 2
    for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave * RPW + rr, v = node0 + r;
        f32x4 val = {0.f, 0.f, 0.f, 0.f}, val2 = {0.f, 0.f, 0.f, 0.f};
        if (v < a.n) {
            for (int i = 0; i < a.n_in; ++i) {
                const int lo = a.rowptr[i][v], hi = a.rowptr[i][v + 1];
                if (hi > lo) {
                    const f32x4 *pm = reinterpret_cast<const f32x4 *>(a.hn_main[i] + (size_t)v * HS);
                    val += pm[lane];
                    if (lane < 2) val2 += pm[64 + lane];
                    for (int t = (lo >> a.tile_shift) + 1; t <= ((hi - 1) >> a.tile_shift); ++t) {
                        const f32x4 *q = reinterpret_cast<const f32x4 *>(a.hn_cont[i] + (size_t)t * HS);
                        val += q[lane];
                        if (lane < 2) val2 += q[64 + lane];
                    }
                }
            }
            const float z = s_z[r];
            val /= z;
            val2 /= z;
        }
        put_row(r, val, val2);
    }
    lds_barrier();
    gemm_rows32_h8(Ah, PN, a.wh_b, acc, wave, lane);
    ex += row_dot_planes<TPR>(Ah, PN, a.wx_b, tid);
    lds_barrier();
    // hidden = SiLU(. + b0) -> planes (pad columns 257 .. 271 stay 0 from the h_neigh tile)
    {
        const int col = 32 * wave + (lane & 31);
        const float bb = a.b0[col];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) put_elem(acc_row32(reg, lane), col, silu(fmaf(acc[reg], H_UNSCALE, bb)));
    }
    if ((tid % TPR) == 0) put_elem(tid / TPR, 256, silu(fmaf(ex, 1.0f / H_SCALE_A, a.b0[256])));
    lds_barrier();
    // GEMM 2 + bias + residual (dynamics.py:201-203)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    gemm_rows32_h8(Ah, PN, a.wh_2, acc, wave, lane);
    ex = row_dot_planes<TPR>(Ah, PN, a.wx_2, tid);
    lds_barrier();
    {
        const int col = 32 * wave + (lane & 31);
        const float bb = a.b2[col];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) A[acc_row32(reg, lane) * SA + col] = fmaf(acc[reg], H_UNSCALE, bb);
    }
    if ((tid % TPR) == 0) A[(tid / TPR) * SA + 256] = fmaf(ex, 1.0f / H_SCALE_A, a.b2[256]);
    lds_barrier();
#pragma unroll 4
    for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave * RPW + rr;
        const f32x4 *src = reinterpret_cast<const f32x4 *>(a.h + (size_t)(node0 + r) * HS);
        *reinterpret_cast<f32x4 *>(A + r * SA + 4 * lane) += src[lane];
        if (lane == 0) A[r * SA + 256] += a.h[(size_t)(node0 + r) * HS + 256];
    }
    lds_barrier();
    // LayerNorm(257) (dynamics.py:81-87, 204), biased variance, eps = 1e-5
    if (a.norm) {
        const int row = tid / TPR, q = tid % TPR;
        const float *tr = A + row * SA + q;
        float sum = 0.0f;
        for (int i = 0; i < 256 / TPR; ++i) sum += tr[TPR * i];
        if (q == 0) sum += A[row * SA + 256];
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) sum += __shfl_xor(sum, o);
        const float mean = sum * a.ln_inv_n;
        float var = 0.0f;
        for (int i = 0; i < 256 / TPR; ++i) {
            const float dlt = tr[TPR * i] - mean;
            var = fmaf(dlt, dlt, var);
        }
        if (q == 0) {
            const float dlt = A[row * SA + 256] - mean;
            var = fmaf(dlt, dlt, var);
        }
#pragma unroll
        for (int o = 1; o < TPR; o <<= 1) var += __shfl_xor(var, o);
        if (q == 0) {
            s_mean[row] = mean;
            s_rstd[row] = 1.0f / sqrtf((var - a.ln_pad * mean * mean) * a.ln_inv_n + 1e-5f);      // pad columns (hidden_nf < 256) hold 0: take their (0 - mean)^2 out
        }
    }
    lds_barrier();
    // normalise and write h' back
#pragma unroll 2
    for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave * RPW + rr, v = node0 + r;
        if (v >= a.n) continue;
        f32x4 val = *reinterpret_cast<const f32x4 *>(A + r * SA + 4 * lane);
        float last = A[r * SA + 256];
        if (a.norm) {
            const float mean = s_mean[r], rstd = s_rstd[r];
            const f32x4 w = reinterpret_cast<const f32x4 *>(a.ln_w)[lane];
            const f32x4 b = reinterpret_cast<const f32x4 *>(a.ln_b)[lane];
            val = (val - mean) * rstd * w + b;
            last = (last - mean) * rstd * a.ln_w[256] + a.ln_b[256];
        }
        f32x4 *dst = reinterpret_cast<f32x4 *>(a.h + (size_t)v * HS);
        dst[lane] = val;
        if (lane == 0) dst[64] = f32x4{last, 0.f, 0.f, 0.f};
    }
}


// ---- launchers ----------------------------------------------------------------------------
// kept for the engines' create paths: everything it used to do now happens per (kernel, device) at launch time
kpd_status egnn_kernels_init() { return KPD_OK; }

kpd_status launch_node_graph_index(const int *ptr, int B, int n, int *bidx, hipStream_t st) {
    if (n == 0) return KPD_OK;
    hipLaunchKernelGGL(k_node_graph_index, dim3(cdiv(n, 256)), dim3(256), 0, st, ptr, B, n, bidx);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status launch_egnn_meta(const int *counts, int e_kk, int active_mask, int active_last, const int *lig_ptr, const int *kp_ptr,
                            const int *ll_per_graph, const int *kk_rowptr, int B, const int *kl_off, float message_norm,
                            int update_kp, int *meta, float *z_lig, float *z_kp, hipStream_t st, int tile_rows) {
    hipLaunchKernelGGL(k_egnn_meta, dim3(cdiv(B, 256)), dim3(256), 0, st, counts, e_kk, active_mask, active_last, lig_ptr, kp_ptr,
                       ll_per_graph, kk_rowptr, B, kl_off, message_norm, update_kp, tile_rows, meta, z_lig, z_kp);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status launch_embed(const float *in, int n, int fin, const float *W0, const float *b0, int hid, const float *W1t,
                        const float *b1, const float *t, const int *bidx, float *out, int identity, hipStream_t st) {
    if (n == 0) return KPD_OK;
    KPD_REQUIRE(fin <= 256 && hid <= 256, KPD_ERR_INVALID, "embed: fin=%d hid=%d exceed 256", fin, hid);
    KPD_REQUIRE(!identity || fin == 256, KPD_ERR_INVALID, "identity encoder needs 256 input features");
    hipLaunchKernelGGL(k_embed, dim3(cdiv(n, EMB_NODES)), dim3(256), 0, st, in, n, fin, W0, b0, hid, W1t, b1, t, bidx,
                       out, identity);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status launch_decode(const float *h, const float *x, const float *x0, int n, int atom_nf, int hid, const float *W0,
                         const float *b0, const float *W1, const float *b1, float *eps_h, float *eps_x,
                         hipStream_t st) {
    if (n == 0) return KPD_OK;
    KPD_REQUIRE(hid <= 64 && atom_nf <= 64, KPD_ERR_INVALID, "decode: hid=%d atom_nf=%d exceed 64", hid, atom_nf);
    hipLaunchKernelGGL(k_decode, dim3(n), dim3(64), 0, st, h, x, x0, n, atom_nf, hid, W0, b0, W1, b1, eps_h, eps_x);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status launch_egnn_edge(const EdgeArgs &a, int tile_cap, hipStream_t st) {
    if (tile_cap == 0) return KPD_OK;
    if (poison_level() >= 1) KPD_TRY(poison_lds(st));      // debug only (engine.h)
    // KPD_EDGE_LDS_PAD (diagnostics): extra dynamic LDS to force one workgroup per CU
    static const int pad = tool_env_int("KPD_EDGE_LDS_PAD", 0);
    static const int ablate = tool_env_int("KPD_EDGE_ABLATE", 0);
    EdgeArgs b = a;
    b.ablate = ablate;
    KPD_REQUIRE(a.tile_rows == TM, KPD_ERR_INVALID, "the edge kernels walk 64-edge tiles");
    // 4 waves per workgroup: 256 VGPRs per lane leave room to keep the coordinate branch's gathered P rows in registers across
    // the attention / segmented-sum phases (the 8-wave and the 32-row builds of rounds 1 - 3 lost to it and were removed in round 4)
    if (a.gemm_mode == 1) {
        for (int et = 0; et < 4; ++et)
            KPD_REQUIRE(!a.wp_e[et] || (a.wh_e[et] && a.wh_c[et]), KPD_ERR_STATE, "f16x2 weights of edge type %d were not packed", et);
        KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_egnn_edge_h<4>), EDGE_H_LDS_BYTES + pad));
        hipLaunchKernelGGL(k_egnn_edge_h<4>, dim3(8 * cdiv(tile_cap, 8)), dim3(256), EDGE_H_LDS_BYTES + pad, st, b);
        KPD_LAUNCH_CHECK();
        return KPD_OK;
    }
    KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_egnn_edge<4>), EDGE_LDS_BYTES + pad));
    // the tail of a launch split by branch (k_egnn_edge): two workgroup slots per CU, an eighth of them per XCD; KPD_EDGE_SPLIT=0: whole tiles only
    static const bool split = tool_env_int("KPD_EDGE_SPLIT", 1) != 0;
    b.split_slots = (split && pad == 0) ? 2 * cu_count() / 8 : 0;
    hipLaunchKernelGGL(k_egnn_edge<4>, dim3(8 * (cdiv(tile_cap, 8) + b.split_slots / 2)), dim3(256), EDGE_LDS_BYTES + pad, st, b);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status launch_node_layer(const NodeLayerPair &p, hipStream_t st) {
    const int tiles = p.tiles0 + cdiv(p.nt[1].u.n, TN);
    if (tiles == 0) return KPD_OK;
    if (poison_level() >= 1) KPD_TRY(poison_lds(st));      // debug only (engine.h)
    // KPD_NODE_LDS_PAD (diagnostics): extra dynamic LDS to lower the number of co-resident workgroups
    static const int pad = tool_env_int("KPD_NODE_LDS_PAD", 0);
    for (int nt = 0; nt < 2; ++nt)
        KPD_REQUIRE(p.nt[nt].u.n == 0 || (p.nt[nt].do_update && !p.nt[nt].do_proj), KPD_ERR_INVALID, "node launch: update-only node types expected");
    // k_node_update8 reads the segment bounds of (up to) two incoming edge types unconditionally: an updated node type has at least one, with its arrays
    for (int nt = 0; nt < 2; ++nt) {
        const NodeArgs &u = p.nt[nt].u;
        if (u.n == 0) continue;
        KPD_REQUIRE(u.n_in >= 1 && u.n_in <= 2, KPD_ERR_INVALID, "node launch: node type %d has %d incoming edge types (1 or 2 expected)", nt, u.n_in);
        for (int i = 0; i < u.n_in; ++i)
            KPD_REQUIRE(u.rowptr[i] && u.hn_main[i] && u.hn_cont[i] && u.xn_main[i] && u.xn_cont[i], KPD_ERR_INVALID,
                        "node launch: segment arrays of incoming edge type %d of node type %d are missing", i, nt);
    }
    if (p.gemm_mode == 1 && !p.stamps) {                   // (phase-stamped diagnostic launches keep the exact kernel)
        for (int nt = 0; nt < 2; ++nt)
            if (p.nt[nt].u.n > 0)
                KPD_REQUIRE(p.nt[nt].u.wh_a && p.nt[nt].u.wh_b && p.nt[nt].u.wh_2, KPD_ERR_STATE, "node weights of type %d have no f16 planes", nt);
        KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_node_update8_h), NODE_H_LDS_BYTES));
        hipLaunchKernelGGL(k_node_update8_h, dim3(tiles), dim3(512), NODE_H_LDS_BYTES, st, p);
        KPD_LAUNCH_CHECK();
        return KPD_OK;
    }
    KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_node_update8), NODE_LAYER_LDS_BYTES + pad));
    hipLaunchKernelGGL(k_node_update8, dim3(tiles), dim3(512), NODE_LAYER_LDS_BYTES + pad, st, p);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}


kpd_status launch_egnn_edge_train(const EdgeTrainArgs &a, int tile_cap, hipStream_t st) {
    if (tile_cap == 0) return KPD_OK;
    KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_egnn_edge_train), EDGE_LDS_BYTES));
    hipLaunchKernelGGL(k_egnn_edge_train, dim3(8 * cdiv(tile_cap, 8)), dim3(256), EDGE_LDS_BYTES, st, a);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status launch_edge_pieces_sum(const EdgePiecesSumArgs &a, hipStream_t st) {
    const int most = std::max(a.n[0], a.n[1]);
    if (most == 0) return KPD_OK;
    hipLaunchKernelGGL(k_edge_pieces_sum, dim3(cdiv(most, 4), 2), dim3(256), 0, st, a);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status launch_egnn_edge_bwd(const EdgeBwdArgs &a, int tile_cap, hipStream_t st) {
    if (tile_cap == 0) return KPD_OK;
    KPD_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k_egnn_edge_bwd), EDGE_BWD_LDS_BYTES));
    hipLaunchKernelGGL(k_egnn_edge_bwd, dim3(8 * cdiv(tile_cap, 8)), dim3(256), EDGE_BWD_LDS_BYTES, st, a);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status launch_edge_pieces_set(const EdgePiecesBatch &b, int count, hipStream_t st) {
    int most = 0;
    for (int i = 0; i < count; ++i) most = std::max(most, b.e[i].n);
    if (count == 0 || most == 0) return KPD_OK;
    KPD_REQUIRE(count <= 8, KPD_ERR_INVALID, "edge pieces: %d pairs in one launch (at most 8)", count);
    hipLaunchKernelGGL(k_edge_pieces_set, dim3(cdiv(most, 4), count), dim3(256), 0, st, b);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

kpd_status launch_edge_train_pack(const EdgePackTab &t, hipStream_t st) {
    if (t.n == 0) return KPD_OK;
    hipLaunchKernelGGL(k_edge_train_pack, dim3(cdiv(WP_FLOATS + 3 * HS, 256), t.n), dim3(256), 0, st, t);
    KPD_LAUNCH_CHECK();
    return KPD_OK;
}

}  // namespace kpd
